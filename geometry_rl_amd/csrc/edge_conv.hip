// Fused per-edge pipeline of the separable fiber-bundle convolution (reference rows 6-7 of SURVEY.md section 8a):
//   invariants -> PolynomialFeatures(2) -> Linear(14,64)+GELU -> Linear(64,64)+GELU        (hepi.py:76-82,109-123,145-157)
//   -> kernel Linear(64,64, no bias) -> message = kernel * x_src[src]                      (conv.py:79,115-117)
//   -> sum over the edges of each destination node                                         (conv.py:141-147)
// Nothing per-edge ever reaches HBM.  Every wave works alone on 32 rows (2 edges x 16 orientations) per pass, in
// destination-sorted edge order:
//   forward : a wave owns TD = 2 consecutive destination nodes, sums their messages in registers and stores the finished rows
//             with plain stores (no atomics, no workgroup barrier in the loop);
//   backward: two launches, each recomputing the chain (see "backward" below): the x kernel walks the edges in SOURCE-sorted order
//             and sums d x_src per source node in registers exactly like the forward sums per destination node; the w kernel
//             accumulates the five weight gradients in registers for the whole launch.  No per-edge scratch, no atomics.
// The next pass's indices / positions and this pass's x_src rows are requested before the MFMA chain starts, so the gather
// latency hides behind it.
#include "grl_common.h"
#include "grl_wimg.h"
#include <type_traits>
#define GRL_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)

namespace {

constexpr int C = 64;            // channels
constexpr int O = 16;            // orientations
constexpr int TD = 2;            // destination nodes per wave tile (forward)
constexpr int LDT = C + 4;       // padded row of an LDS activation tile
constexpr int LDW = GRL_LD(64);  // 68   fp32 images (backward: column reads)
constexpr int LDB = WI_LDB;      // 72   split-bf16 images (chain): ChainW, grl_wimg.h
constexpr int LDB1 = WI_LDB1;    // 24

struct EdgeParams {
  const st_t* x_src;     // [Ns,16,64] (storage type: grl_common.h)
  const float* pos_src;  // [Ns,3]
  const float* pos_dst;  // [Nd,3]
  const int* rowptr;     // [Na+1]  anchor-sorted CSR (anchor = dst in forward, src in backward)
  const int* e_src;      // [E] source node of each edge, in anchor-sorted order
  const int* e_dst;      // [E] destination node of each edge, same order
  const float* grid;     // [16,3] (z = 0 for the S1 grid)
  const float* W1;       // [64,14]
  const float* b1;       // [64]
  const float* W2;       // [64,64]
  const float* b2;       // [64]
  const float* Wk;       // [64,64]
  int n_anchor;
  int dim;
  // Attention aggregation (conv.py:21-26,58-61,138-139): the gradient that reaches the messages is PER EDGE ([E,16,64], rows in
  // destination-sorted edge order) instead of per destination node.  per_edge != 0: the backward kernels read row erow[i] of that
  // tensor for the i-th edge of THEIR order (erow == nullptr: row i, i.e. the kernel walks the destination-sorted order itself).
  const int* erow = nullptr;
  int per_edge = 0;
  const void* wimg = nullptr;   // optional pre-split ChainW image of this forward pass (grl_weight_images, kind WIMG_EDGE32)
};

// Polynomial features of (a, b) in the reference order (ponita.py:233-244):
//   k:  0  1 | 2   3   4   5 | 6    7    8    9    10   11   12   13
//       a  b | aa  ab  ba  bb| aaa  aab  aba  abb  baa  bab  bba  bbb        (columns 14, 15 are zero padding)
// split into this lane's two fragments: lane half h owns k = 4h..4h+3 and k = 8+4h..8+4h+3.
GRL_DEVINL void poly_frags(float a, float b, int h, float4& f0, float4& f1) {
  const float aa = a * a, ab = a * b, bb = b * b;
  if (h == 0) {
    f0 = make_float4(a, b, aa, ab);
    f1 = make_float4(ab * a, ab * b, ab * a, ab * b);
  } else {
    f0 = make_float4(ab, bb, aa * a, aa * b);
    f1 = make_float4(bb * a, bb * b, 0.f, 0.f);
  }
}

// One pass of the chain for this lane's row.  Returns the kernel fragments K (8 float4); optionally keeps the
// pre-activation derivatives for the backward pass.
// Weight images of the chain in LDS: bf16 hi/lo rows (grl_common.h "split-bf16 MFMA path"): struct ChainW, grl_wimg.h

// Split-bf16 fragments of the chain's activations (rows on the lanes): the B operands of the next layer and, in the backward,
// the inputs of the register-level transposes for the row-reduction products.
struct ChainFrags {
  bf16x8 ph[1], pl[1];    // polynomial features (16 columns, 14 used)
  bf16x8 g1h[4], g1l[4];  // gelu(z1)
  bf16x8 g2h[4], g2l[4];  // gelu(z2)
};

GRL_DEVINL f32x16 bias_frag(const float* bias_s, int n0, int h) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 bb = *reinterpret_cast<const float4*>(bias_s + n0 + 8 * q + 4 * h);
    acc[4 * q] = bb.x; acc[4 * q + 1] = bb.y; acc[4 * q + 2] = bb.z; acc[4 * q + 3] = bb.w;
  }
  return acc;
}

// One pass of the chain for this lane's row.  BWD keeps the activation derivatives; FENCED selects the two-waves-per-SIMD-safe
// MFMA grouping (grl_common.h); k_epilogue(nt, acc) receives the two 32-column tiles of the kernel layer K = Wk g2 (pass
// nullptr_t-like NoK to skip that layer).
// ---- in-kernel phase timing (diagnostic build only: -DGRL_PHASE_PROF; tools/edge_phase.py reads the totals) ----------------------------
// s_memtime stamps between the stages of the chain and of the kernels around it, accumulated by ONE wave per workgroup into a
// __device__ table [kernel][phase].  The stamps cost ~10 % of the wave's cycles and serialise nothing else; shares, not absolutes.
#ifdef GRL_PHASE_PROF
__device__ unsigned long long g_ephase[3][24];
struct PhaseClock {
  unsigned long long ph[24], last;
  GRL_DEVINL void start() { for (int i = 0; i < 24; ++i) ph[i] = 0; last = __builtin_amdgcn_s_memtime(); }
  GRL_DEVINL void stamp(int i) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph[i] += t - last; last = t; }
  GRL_DEVINL void flush(int kernel) {
    if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 1)
      for (int i = 0; i < 24; ++i) atomicAdd(&g_ephase[kernel][i], ph[i]);
  }
};
#define PHS(i) pc.stamp(i)
#else
struct PhaseClock { GRL_DEVINL void start() {} GRL_DEVINL void stamp(int) {} GRL_DEVINL void flush(int) {} };
#define PHS(i)
#endif
struct NoK {};
// GRL_CHAIN_PIPED (build switch, off): in the fenced (two waves per SIMD) form every MFMA group's weight fragments are requested
// while the previous group's activation epilogue runs (mma_wx_bf_piped) instead of in front of the group.  Measured round 2: no
// change (forward 0.52 vs 0.51 ms per step, +14 registers): the LDS latency of the fragments is already covered by the SIMD partner.
#ifndef GRL_CHAIN_PIPED
#define GRL_CHAIN_PIPED 0
#endif
#ifndef GRL_POS_EARLY
#define GRL_POS_EARLY 1
#endif
struct NoMid { GRL_DEVINL void operator()() const {} };
template <bool BWD, bool FENCED, class KEpi, class Mid = NoMid>
GRL_DEVINL void edge_chain(const ChainW& w, float a, float b, float4 (&g1)[8], float4 (&gp1)[8], float4 (&g2)[8], float4 (&gp2)[8],
                           ChainFrags& f, KEpi&& k_epilogue, float* sink_p, PhaseClock& pc, Mid&& mid = NoMid{}) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  float4 phi[2];
  poly_frags(a, b, h, phi[0], phi[1]);
  bf16x8 (&ph)[1] = f.ph, (&pl)[1] = f.pl;
  split_frags<16>(phi, ph, pl);
  PHS(2);   // (a, b) available (positions of this pass arrived) + polynomial features + split
#if GRL_CHAIN_PIPED
  if constexpr (FENCED && !BWD && !std::is_same<typename std::decay<KEpi>::type, NoK>::value) {
    float& sink = *sink_p;
    auto act1 = [&](int nt) {
      return [&, nt](const f32x16& acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) g1[4 * nt + q] = gelu4(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
      };
    };
    auto act2 = [&](int nt) {
      return [&, nt](const f32x16& acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) g2[4 * nt + q] = gelu4(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
      };
    };
    auto wp = [&](const unsigned short* base, int ld, int nt) { return base + (32 * nt + i) * ld + 8 * h; };
    WFrags<16> w1a, w1b;
    WFrags<64> wa, wb;
    load_wfrags<16>(w1a, wp(w.W1h, LDB1, 0), wp(w.W1l, LDB1, 0));
    mma_wx_bf_piped<16, 16>(w1a, ph, pl, bias_frag(w.b1s, 0, h), &w1b, wp(w.W1h, LDB1, 1), wp(w.W1l, LDB1, 1), sink, act1(0));
    mma_wx_bf_piped<16, 64>(w1b, ph, pl, bias_frag(w.b1s, 32, h), &wa, wp(w.W2h, LDB, 0), wp(w.W2l, LDB, 0), sink, act1(1));
    bf16x8 (&g1h)[4] = f.g1h, (&g1l)[4] = f.g1l;
    split_frags<64>(g1, g1h, g1l);
    mma_wx_bf_piped<64, 64>(wa, g1h, g1l, bias_frag(w.b2s, 0, h), &wb, wp(w.W2h, LDB, 1), wp(w.W2l, LDB, 1), sink, act2(0));
    mma_wx_bf_piped<64, 64>(wb, g1h, g1l, bias_frag(w.b2s, 32, h), &wa, wp(w.Wkh, LDB, 0), wp(w.Wkl, LDB, 0), sink, act2(1));
    bf16x8 (&g2h)[4] = f.g2h, (&g2l)[4] = f.g2l;
    split_frags<64>(g2, g2h, g2l);
    mma_wx_bf_piped<64, 64>(wa, g2h, g2l, zero16(), &wb, wp(w.Wkh, LDB, 1), wp(w.Wkl, LDB, 1), sink,
                            [&](const f32x16& acc) { k_epilogue(0, acc); });
    mma_wx_bf_piped<64, 64>(wb, g2h, g2l, zero16(), static_cast<WFrags<64>*>(nullptr), nullptr, nullptr, sink,
                            [&](const f32x16& acc) { k_epilogue(1, acc); });
    return;
  }
#endif
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    auto act = [&](const f32x16& acc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 zq = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        if (BWD) gelu_both4(zq, g1[4 * nt + q], gp1[4 * nt + q]);
        else g1[4 * nt + q] = gelu4(zq);
      }
    };
    const unsigned short* wh = w.W1h + (32 * nt + i) * LDB1 + 8 * h, *wl = w.W1l + (32 * nt + i) * LDB1 + 8 * h;
    if (FENCED) {
      mma_wx_bf_fenced<16>(wh, wl, ph, pl, bias_frag(w.b1s, 32 * nt, h), act);
    } else {
      f32x16 acc = bias_frag(w.b1s, 32 * nt, h);
      mma_wx_bf<16>(wh, wl, ph, pl, acc);
      act(acc);
    }
  }
  mid();    // the caller's early requests for the NEXT pass (positions): their latency hides behind layers 2 and 3
  PHS(3);   // layer 1: two tiles (fragment loads, 3 MFMAs, GELU each)
  bf16x8 (&g1h)[4] = f.g1h, (&g1l)[4] = f.g1l;
  split_frags<64>(g1, g1h, g1l);
  GRL_SCHED_BARRIER();
  PHS(4);   // split of g1
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    auto act = [&](const f32x16& acc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 zq = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        if (BWD) gelu_both4(zq, g2[4 * nt + q], gp2[4 * nt + q]);
        else g2[4 * nt + q] = gelu4(zq);
      }
    };
    const unsigned short* wh = w.W2h + (32 * nt + i) * LDB + 8 * h, *wl = w.W2l + (32 * nt + i) * LDB + 8 * h;
    if (FENCED) {
      mma_wx_bf_fenced<64>(wh, wl, g1h, g1l, bias_frag(w.b2s, 32 * nt, h), act);
    } else {
      f32x16 acc = bias_frag(w.b2s, 32 * nt, h);
      mma_wx_bf<64>(wh, wl, g1h, g1l, acc);
      act(acc);
    }
  }
  PHS(5);   // layer 2: two tiles (fragment loads, 12 MFMAs, GELU each)
  bf16x8 (&g2h)[4] = f.g2h, (&g2l)[4] = f.g2l;
  split_frags<64>(g2, g2h, g2l);
  GRL_SCHED_BARRIER();
  PHS(6);   // split of g2
  if constexpr (!std::is_same<typename std::decay<KEpi>::type, NoK>::value) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const unsigned short* wh = w.Wkh + (32 * nt + i) * LDB + 8 * h, *wl = w.Wkl + (32 * nt + i) * LDB + 8 * h;
      if (FENCED) {
        mma_wx_bf_fenced<64>(wh, wl, g2h, g2l, zero16(), [&](const f32x16& acc) { k_epilogue(nt, acc); });
      } else {
        f32x16 acc = zero16();
        mma_wx_bf<64>(wh, wl, g2h, g2l, acc);
        k_epilogue(nt, acc);
      }
    }
    GRL_SCHED_BARRIER();
    PHS(7);   // kernel layer: two tiles (fragment loads, 12 MFMAs, message epilogue each)
  }
}

// backward additionally keeps split-bf16 images of W2^T / Wk^T for dG = dZ W (rows of the transposed matrix = columns of W)
struct BwdW {
  unsigned short W2Th[64 * LDB], W2Tl[64 * LDB];
  unsigned short WkTh[64 * LDB], WkTl[64 * LDB];
};

GRL_DEVINL void load_chain_weights(ChainW& s, const EdgeParams& p) {
  if (p.wimg) {   // built once per forward pass by grl_weight_images: a linear copy
    copy_image<256>(&s, p.wimg, (int)sizeof(ChainW));
    return;
  }
  stage_split<64, 16, 14, 256>(s.W1h, s.W1l, p.W1, LDB1);
  stage_split<64, 64, 64, 256>(s.W2h, s.W2l, p.W2, LDB);
  stage_split<64, 64, 64, 256>(s.Wkh, s.Wkl, p.Wk, LDB);
  for (int idx = threadIdx.x; idx < 64; idx += blockDim.x) {
    s.b1s[idx] = p.b1[idx];
    s.b2s[idx] = p.b2[idx];
    s.grid_s[idx] = (idx < 48) ? p.grid[idx] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ per-pass metadata
struct PassMeta {
  int src, dst;   // node ids of this lane's edge
  int grow;       // row of the incoming-gradient tensor for this edge: its destination node, or its per-edge row (attention)
  float a, b;     // spatial invariants for this lane's (edge, orientation)
  float rx, ry, rz, qx, qy, qz;   // pos_src / pos_dst of the edge (requested early, consumed by meta_invariants_compute)
  bool valid;
};

GRL_DEVINL void meta_indices(const EdgeParams& p, int e, int e_end, PassMeta& m) {
  m.valid = e < e_end;
  const int ee = m.valid ? e : e_end - 1;
  m.src = p.e_src[ee];
  m.dst = p.e_dst[ee];
  m.grow = p.per_edge ? (p.erow ? p.erow[ee] : ee) : m.dst;
}
// The positions of a pass are requested as soon as its indices have arrived -- in the MIDDLE of the previous pass's chain (edge_chain's
// ``mid`` hook, behind layer 1) -- and turned into (a, b) at its end: the dependent index -> position round trip used to sit exposed
// at the end of every pass (phase timing, tools/edge_phase.py: 12 % of the forward, 14 % of the weights kernel).
GRL_DEVINL void meta_pos_load(const EdgeParams& p, PassMeta& m) {   // loads only: nothing here may wait for them
  m.rx = p.pos_src[3 * m.src]; m.ry = p.pos_src[3 * m.src + 1]; m.rz = p.pos_src[3 * m.src + 2];
  m.qx = p.pos_dst[3 * m.dst]; m.qy = p.pos_dst[3 * m.dst + 1]; m.qz = p.pos_dst[3 * m.dst + 2];
}
GRL_DEVINL void meta_invariants_compute(const EdgeParams& p, const float* grid_s, int o, PassMeta& m) {
  float rx = m.rx - m.qx, ry = m.ry - m.qy, rz = (p.dim == 2) ? 0.f : m.rz - m.qz;
  const float gx = grid_s[3 * o], gy = grid_s[3 * o + 1], gz = grid_s[3 * o + 2];
  m.a = rx * gx + ry * gy + rz * gz;                      // hepi.py:115
  rx -= m.a * gx; ry -= m.a * gy; rz -= m.a * gz;
  m.b = sqrtf(rx * rx + ry * ry + rz * rz);               // hepi.py:117
}
GRL_DEVINL void meta_invariants(const EdgeParams& p, const float* grid_s, int o, PassMeta& m) {
  float rx = p.pos_src[3 * m.src] - p.pos_dst[3 * m.dst];
  float ry = p.pos_src[3 * m.src + 1] - p.pos_dst[3 * m.dst + 1];
  float rz = (p.dim == 2) ? 0.f : p.pos_src[3 * m.src + 2] - p.pos_dst[3 * m.dst + 2];
  const float gx = grid_s[3 * o], gy = grid_s[3 * o + 1], gz = grid_s[3 * o + 2];
  m.a = rx * gx + ry * gy + rz * gz;                      // hepi.py:115
  rx -= m.a * gx; ry -= m.a * gy; rz -= m.a * gz;
  m.b = sqrtf(rx * rx + ry * ry + rz * rz);               // hepi.py:117
}

// ------------------------------------------------------------------------------------------------ forward
// A wave owns TD = 2 consecutive destination nodes (A, B).  Each lane keeps two register accumulators (one per node) for
// its (edge slot, orientation) row; after the last pass the two edge slots are folded with one cross-lane exchange
// (lane r <-> r^16) and the rows leave with plain stores.  No LDS traffic besides the weights (LDS float atomics cost
// ~200 LDS cycles per wave instruction on gfx950 -- measured, profiles/r01_*pmc* -- and made the first version LDS-bound).
// GRL_FENCED_2W: the forward and the d x_src kernel run two waves per SIMD and therefore use the fenced MFMA groups (DESIGN.md finding 3).
// -DGRL_FENCED_2W=false builds the hazard's in-situ reproducer (tools/det_check_all.py, tools/run_hazard_check.sh): never ship it.
#ifndef GRL_FENCED_2W
#define GRL_FENCED_2W true
#endif
#ifndef GRL_FWD_WAVES
#define GRL_FWD_WAVES 4
#endif
constexpr int FWD_WAVES = GRL_FWD_WAVES;
static_assert(FWD_WAVES >= 4, "load_chain_weights copies a pre-split image with 256 threads");
#ifndef GRL_FWD_SPLIT_TILES
#define GRL_FWD_SPLIT_TILES 512   // at most this many destination tiles: one workgroup per tile (edge_conv_fwd_kernel<true>)
#endif
#ifndef GRL_FWD_MAX_BLOCKS
#define GRL_FWD_MAX_BLOCKS 512   // two 4-wave workgroups per CU = two waves per SIMD (the chain is fenced for that)
#endif
// SPLIT: few destination tiles with long edge lists (e.g. the object -> gripper convolution of a small minibatch shard: a few
// hundred tiles of 16+ passes would occupy a fraction of the SIMDs for the whole launch).  A workgroup then owns ONE tile, its
// four waves take every fourth pass and the four partial messages are added in wave order through LDS.
template <bool SPLIT>
__global__ __launch_bounds__(64 * FWD_WAVES, 2) void edge_conv_fwd_kernel
(EdgeParams p, st_t* __restrict__ x1 /*[Nd,16,64]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  ChainW& s = *reinterpret_cast<ChainW*>(smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  load_chain_weights(s, p);
  __syncthreads();
  const int o = r & 15, el = r >> 4;
  const int n_tiles = (p.n_anchor + TD - 1) / TD;
  constexpr int ESTEP = SPLIT ? 2 * FWD_WAVES : 2;
  float sink = 0.f;   // keeps the accumulator fences of the pipelined chain alive (never stored, see the end of the kernel)
  PhaseClock pc;
  pc.start();
  for (int tl = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * FWD_WAVES + wave; tl < n_tiles;
       tl += SPLIT ? (int)gridDim.x : (int)gridDim.x * FWD_WAVES) {
    const int d0 = tl * TD, d1 = min(d0 + TD, p.n_anchor);
    const int e0 = p.rowptr[d0] + (SPLIT ? 2 * wave : 0), e1 = p.rowptr[d1];
    float4 accA[8], accB[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { accA[t] = make_float4(0.f, 0.f, 0.f, 0.f); accB[t] = accA[t]; }
    if (e1 > e0) {
      PassMeta cur;
      meta_indices(p, e0 + el, e1, cur);
      meta_invariants(p, s.grid_s, o, cur);
      PHS(0);   // tile head: rowptr, first indices + positions (dependent loads, exposed)
#pragma unroll 1
      for (int e = e0; e < e1; e += ESTEP) {
        PassMeta nxt;
        const bool more = e + ESTEP < e1;
        // next pass: indices in flight (requested unconditionally -- meta_indices clamps past the end -- so that nothing downstream
        // hangs on a branch: a guarded load's result is merged by register copies that wait for it on the spot)
        meta_indices(p, e + ESTEP + el, e1, nxt);
        const st_t* xs = p.x_src + ((size_t)cur.src * O + o) * C + 4 * h;
        float4 xv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) xv[t] = ld4(xs + 8 * t);                   // this pass: x_src row in flight
        PHS(1);   // pass top: next indices + this pass's rows requested
        float4 g1[8], gp1[8], g2[8], gp2[8];
        ChainFrags cf;
        const float wa = (cur.valid && cur.dst == d0) ? 1.f : 0.f;
        const float wb = (cur.valid && cur.dst != d0) ? 1.f : 0.f;
        // message = K * x_src, summed into the accumulator of the edge's destination node as each K tile leaves the matrix pipe
        edge_chain<false, GRL_FENCED_2W>(s, cur.a, cur.b, g1, gp1, g2, gp2, cf, [&](int nt, const f32x16& acc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = 4 * nt + q;
            const float4 m = f4_mul(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]), xv[t]);
            accA[t] = make_float4(fmaf(m.x, wa, accA[t].x), fmaf(m.y, wa, accA[t].y), fmaf(m.z, wa, accA[t].z), fmaf(m.w, wa, accA[t].w));
            accB[t] = make_float4(fmaf(m.x, wb, accB[t].x), fmaf(m.y, wb, accB[t].y), fmaf(m.z, wb, accB[t].z), fmaf(m.w, wb, accB[t].w));
          }
        }, &sink, pc, [&]() { if (GRL_POS_EARLY) meta_pos_load(p, nxt); });   // next pass: positions requested mid-chain
        if (GRL_POS_EARLY) meta_invariants_compute(p, s.grid_s, o, nxt);         // next pass: positions -> (a, b)
        else if (more) meta_invariants(p, s.grid_s, o, nxt);
        cur = nxt;
        PHS(8);   // (a, b) of the next pass
      }
    }
    // fold the two edge slots; slot 0 lanes store node A's rows, slot 1 lanes node B's
    const int node = d0 + el;
    st_t* dstp = x1 + ((size_t)node * O + o) * C + 4 * h;
    float4* red = reinterpret_cast<float4*>(smem_raw + sizeof(ChainW) / 4);   // SPLIT only: [FWD_WAVES][8][64]
    if (SPLIT) __syncthreads();   // the previous tile's sums have been read
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float4 a = accA[t], b = accB[t];
      a.x += __shfl_xor(a.x, 16, 64); a.y += __shfl_xor(a.y, 16, 64); a.z += __shfl_xor(a.z, 16, 64); a.w += __shfl_xor(a.w, 16, 64);
      b.x += __shfl_xor(b.x, 16, 64); b.y += __shfl_xor(b.y, 16, 64); b.z += __shfl_xor(b.z, 16, 64); b.w += __shfl_xor(b.w, 16, 64);
      // component-wise select: a struct-level `el == 0 ? a : b` is lowered through scratch memory by hipcc (store both, load
      // one by index) -- slow, and the store->load ordering proved unreliable with two waves of a block per SIMD
      const bool is_a = el == 0;
      const float4 v = make_float4(is_a ? a.x : b.x, is_a ? a.y : b.y, is_a ? a.z : b.z, is_a ? a.w : b.w);
      if (SPLIT) red[(wave * 8 + t) * 64 + lane] = v;
      else if (node < d1) st4(dstp + 8 * t, v);
    }
    PHS(9);   // tile tail: cross-lane fold + stores
    if (SPLIT) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8 / FWD_WAVES; ++u) {   // wave w adds up and stores fragments t = w, w + FWD_WAVES, ...
        const int t = wave + u * FWD_WAVES;
        float4 v = red[t * 64 + lane];
#pragma unroll
        for (int w_ = 1; w_ < FWD_WAVES; ++w_) v = f4_add(v, red[(w_ * 8 + t) * 64 + lane]);
        if (node < d1) st4(dstp + 8 * t, v);
      }
    }
  }
  if (sink == 123456.789f) st1(x1, sink);   // never true
  pc.flush(0);
}

// GRL_LEGACY32 (build switch, default 0): the 32-row kernels that the 16-row kernels of edge_conv16.hip superseded in round 2 -- the flat
// message kernel, the d x_src kernel and the weight-gradient kernel below, and the non-split instance of the forward above.  They are
// the A/B baselines of DESIGN.md findings 19 / 20 (-DGRL_LEGACY32=1 -DGRL_EDGE16=0 -DGRL_EDGE_BWD16=0) and are NOT compiled into the
// shipped library: nothing unreachable, nothing the parity / determinism suite does not cover.
constexpr int EDGE_PARTIAL = 64 * 14 + 64 + 64 * 64 + 64 + 64 * 64;   // partial row: [W1 64x14 | b1 64 | W2 64x64 | b2 64 | Wk 64x64]
#ifndef GRL_LEGACY32
#define GRL_LEGACY32 0
#endif
#if GRL_LEGACY32
// ------------------------------------------------------------------------------------------------ messages (attention aggregation)
// FiberBundleConv(aggr="AttentionalAggregation") (conv.py:21-26,58-61,138-139; configs/algorithm/pyg_agent/model/hepi_attention.yaml)
// gates every message before it is summed, so the messages m_e = K_e * x_src[src(e)] have to exist per edge: this kernel is the forward
// chain with a store in place of the per-destination sum -- flat passes of two edges in destination-sorted order, row e of
// msg [E,16,64] = edge e of that order.  The gate, the per-destination softmax and the weighted sum follow in grl_softmax_aggregate_*.
__global__ __launch_bounds__(256, 2) void edge_msg_fwd_kernel(EdgeParams p, st_t* __restrict__ msg /*[E,16,64]*/, int n_edges) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  ChainW& s = *reinterpret_cast<ChainW*>(smem_raw);
  load_chain_weights(s, p);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int o = r & 15, el = r >> 4;
  float sink = 0.f;
  PhaseClock pc;
  pc.start();
  const int n_pass = (n_edges + 1) >> 1;
  const int stride = gridDim.x * 4;
  int ps = blockIdx.x * 4 + wave;
  PassMeta cur;
  if (ps < n_pass) {
    meta_indices(p, 2 * ps + el, n_edges, cur);
    meta_invariants(p, s.grid_s, o, cur);
  }
#pragma unroll 1
  for (; ps < n_pass; ps += stride) {
    PassMeta nxt;
    const bool more = ps + stride < n_pass;
    meta_indices(p, 2 * (ps + stride) + el, n_edges, nxt);
    const st_t* xs = p.x_src + ((size_t)cur.src * O + o) * C + 4 * h;
    float4 xv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) xv[t] = ld4(xs + 8 * t);
    float4 g1[8], gp1[8], g2[8], gp2[8];
    ChainFrags cf;
    st_t* mrow = msg + ((size_t)(2 * ps + el) * O + o) * C + 4 * h;
    const bool ok = cur.valid;
    edge_chain<false, GRL_FENCED_2W>(s, cur.a, cur.b, g1, gp1, g2, gp2, cf, [&](int nt, const f32x16& acc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = 4 * nt + q;
        const float4 m = f4_mul(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]), xv[t]);
        if (ok) st4(mrow + 8 * t, m);
      }
    }, &sink, pc, [&]() { if (GRL_POS_EARLY) meta_pos_load(p, nxt); });
    if (GRL_POS_EARLY) meta_invariants_compute(p, s.grid_s, o, nxt);
    else if (more) meta_invariants(p, s.grid_s, o, nxt);
    cur = nxt;
  }
  if (sink == 123456.789f) st1(msg, sink);   // never true
}

// ------------------------------------------------------------------------------------------------ backward
// Two launches, each recomputing the cheap split-bf16 chain, so that the register file holds one launch's accumulators without
// spilling (everything in one kernel: 224 accumulator registers + the chain state, 170-220 spilled VGPRs, 30 % slower):
//   x kernel (edges in SOURCE-sorted order, a wave owns TD = 2 consecutive source nodes, exactly like the forward owns
//             destination nodes; two waves per SIMD, fenced MFMA groups): dM = d x1[dst]; d x_src[src] += dM * K accumulated in
//             registers and stored once per node (no per-edge scratch, no second pass, no atomics)
//   w kernel (destination-sorted order, flat passes of 2 edges x 16 orientations = 32 rows per wave; one wave per SIMD):
//             dK = dM * x_src; dWk += dK^T g2; dZ2 = (dK Wk) * gelu'(z2); dW2 += dZ2^T g1; db2;
//             dZ1 = (dZ2 W2) * gelu'(z1); dW1 += dZ1^T phi; db1
// Weight-gradient accumulators live in registers for the whole launch and leave as one partial row per wave:
//   partial[block][9216] = [W1 64x14 | b1 64 | W2 64x64 | b2 64 | Wk 64x64]   (the four waves folded through LDS at the end).


// p is the SOURCE-anchored view of the edge set (rowptr = rowptr_s, e_src / e_dst in source-sorted order, n_anchor = n_src).
__global__ __launch_bounds__(256, 2) void edge_conv_bwd_x_kernel(EdgeParams p, const st_t* __restrict__ dx1 /*[Nd,16,64]*/,
                                                                 st_t* __restrict__ dx_src /*[Ns,16,64]*/,
                                                                 const st_t* __restrict__ dres /*[Ns,16,64] or null*/) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  ChainW& s = *reinterpret_cast<ChainW*>(smem_raw);
  load_chain_weights(s, p);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int o = r & 15, el = r >> 4;
  float sink = 0.f;
  PhaseClock pc;
  pc.start();

  const int n_tiles = (p.n_anchor + TD - 1) / TD;
#pragma unroll 1
  for (int tl = blockIdx.x * 4 + wave; tl < n_tiles; tl += gridDim.x * 4) {
    const int s0 = tl * TD, s1 = min(s0 + TD, p.n_anchor);
    const int e0 = p.rowptr[s0], e1 = p.rowptr[s1];
    float4 accA[8], accB[8];               // d x_src rows of the tile's two source nodes
#pragma unroll
    for (int t = 0; t < 8; ++t) { accA[t] = make_float4(0.f, 0.f, 0.f, 0.f); accB[t] = accA[t]; }
    if (e1 > e0) {
      PassMeta cur;
      meta_indices(p, e0 + el, e1, cur);
      meta_invariants(p, s.grid_s, o, cur);
#pragma unroll 1
      for (int e = e0; e < e1; e += 2) {
        PassMeta nxt;
        const bool more = e + 2 < e1;
        meta_indices(p, e + 2 + el, e1, nxt);
        const st_t* dm = dx1 + ((size_t)cur.grow * O + o) * C + 4 * h;
        float4 dv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) dv[t] = ld4(dm + 8 * t);   // in flight behind the chain
        float4 g1[8], gp1[8], g2[8], gp2[8];
        ChainFrags cf;
        const float wa = (cur.valid && cur.src == s0) ? 1.f : 0.f;
        const float wb = (cur.valid && cur.src != s0) ? 1.f : 0.f;
        // d x_src row = dM * K, summed into the accumulator of the edge's source node as each K tile leaves the matrix pipe
        // (two waves share a SIMD: fenced MFMA groups, exactly like the forward kernel)
        edge_chain<false, GRL_FENCED_2W>(s, cur.a, cur.b, g1, gp1, g2, gp2, cf, [&](int nt, const f32x16& acc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = 4 * nt + q;
            const float4 m = f4_mul(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]), dv[t]);
            accA[t] = make_float4(fmaf(m.x, wa, accA[t].x), fmaf(m.y, wa, accA[t].y), fmaf(m.z, wa, accA[t].z), fmaf(m.w, wa, accA[t].w));
            accB[t] = make_float4(fmaf(m.x, wb, accB[t].x), fmaf(m.y, wb, accB[t].y), fmaf(m.z, wb, accB[t].z), fmaf(m.w, wb, accB[t].w));
          }
        }, &sink, pc, [&]() { if (GRL_POS_EARLY) meta_pos_load(p, nxt); });
        if (GRL_POS_EARLY) meta_invariants_compute(p, s.grid_s, o, nxt);
        else if (more) meta_invariants(p, s.grid_s, o, nxt);
        cur = nxt;
        PHS(8);
      }
    }
    // fold the two edge slots; slot 0 lanes store node A's rows, slot 1 lanes node B's (see the forward kernel)
    const int node = s0 + el;
    st_t* dstp = dx_src + ((size_t)node * O + o) * C + 4 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float4 a = accA[t], b = accB[t];
      a.x += __shfl_xor(a.x, 16, 64); a.y += __shfl_xor(a.y, 16, 64); a.z += __shfl_xor(a.z, 16, 64); a.w += __shfl_xor(a.w, 16, 64);
      b.x += __shfl_xor(b.x, 16, 64); b.y += __shfl_xor(b.y, 16, 64); b.z += __shfl_xor(b.z, 16, 64); b.w += __shfl_xor(b.w, 16, 64);
      const bool is_a = el == 0;
      float4 v = make_float4(is_a ? a.x : b.x, is_a ? a.y : b.y, is_a ? a.z : b.z, is_a ? a.w : b.w);
      if (node < s1) {
        if (dres) v = f4_add(v, ld4(dres + ((size_t)node * O + o) * C + 4 * h + 8 * t));  // + residual branch
        st4(dstp + 8 * t, v);
      }
    }
  }
  if (sink == 123456.789f) st1(dx_src, sink);   // never true
  pc.flush(1);
}

__global__ __launch_bounds__(256, 1) void edge_conv_bwd_w_kernel(EdgeParams p, const st_t* __restrict__ dx1 /*[Nd,16,64]*/,
                                                                  float* __restrict__ partial, int n_edges) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  ChainW& s = *reinterpret_cast<ChainW*>(smem_raw);
  BwdW& sb = *reinterpret_cast<BwdW*>(smem_raw + sizeof(ChainW) / 4);
  load_chain_weights(s, p);
  stage_split_T<256>(sb.W2Th, sb.W2Tl, p.W2, LDB);
  stage_split_T<256>(sb.WkTh, sb.WkTl, p.Wk, LDB);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int o = r & 15, el = r >> 4;
  bf16x8 sel0, sel1;
  make_selectors(sel0, sel1);

  PhaseClock pc;
  pc.start();
  f32x16 accA[2][2], accB[2], accK[2][2];  // accA = dW2, accB = dW1, accK = dWk
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_) {
    accB[a_] = zero16();
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) { accA[a_][b_] = zero16(); accK[a_][b_] = zero16(); }
  }
  float db1[2] = {0.f, 0.f}, db2[2] = {0.f, 0.f};  // column 32*nt + r, summed over this lane half's rows

  const int n_pass = (n_edges + 1) >> 1;
  const int stride = gridDim.x * 4;
  int ps = blockIdx.x * 4 + wave;
  PassMeta cur;
  if (ps < n_pass) {
    meta_indices(p, 2 * ps + el, n_edges, cur);
    meta_invariants(p, s.grid_s, o, cur);
  }
#pragma unroll 1
  for (; ps < n_pass; ps += stride) {
    PassMeta nxt;
    const bool more = ps + stride < n_pass;
    meta_indices(p, 2 * (ps + stride) + el, n_edges, nxt);
    const st_t* xs = p.x_src + ((size_t)cur.src * O + o) * C + 4 * h;
    const st_t* dm = dx1 + ((size_t)cur.grow * O + o) * C + 4 * h;
    float4 xv[8], dv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { xv[t] = ld4(xs + 8 * t); dv[t] = ld4(dm + 8 * t); }   // in flight behind the chain
    PHS(1);
    float4 g1[8], gp1[8], g2[8], gp2[8];
    ChainFrags cf;
    edge_chain<true, false>(s, cur.a, cur.b, g1, gp1, g2, gp2, cf, NoK{}, nullptr, pc,
                            [&]() { if (GRL_POS_EARLY) meta_pos_load(p, nxt); });
    if (GRL_POS_EARLY) meta_invariants_compute(p, s.grid_s, o, nxt);
    else if (more) meta_invariants(p, s.grid_s, o, nxt);
    PHS(8);

    float4 dK[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (!cur.valid) dv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      dK[t] = f4_mul(dv[t], xv[t]);
    }
    {
      // ---- dZ2 = (dK Wk) * gelu'(z2)      (split-bf16: rows of Wk^T)
      float4 dz2[8];
      {
        bf16x8 dh_[4], dl_[4];
        split_frags<64>(dK, dh_, dl_);
        // ---- dWk[c][k] += sum_r dK[r][c] g2[r][k]   (g2 fragments straight from the chain; consumed first so they die early)
        {
          const TTile tg0 = transpose_split(cf.g2h[0], cf.g2h[1], cf.g2l[0], cf.g2l[1], sel0, sel1);
          const TTile tg1 = transpose_split(cf.g2h[2], cf.g2h[3], cf.g2l[2], cf.g2l[3], sel0, sel1);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const TTile tk = transpose_split(dh_[2 * ct], dh_[2 * ct + 1], dl_[2 * ct], dl_[2 * ct + 1], sel0, sel1);
            mma_tn_bf_acc(tk, tg0, accK[ct][0]);
            mma_tn_bf_acc(tk, tg1, accK[ct][1]);
          }
        }
        GRL_SCHED_BARRIER();
        PHS(10);  // dK, its split, dWk: 4 transposed tiles (g2 x2, dK x2) + 4 x 6 MFMAs
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x16 acc = zero16();
          mma_wx_bf<64>(sb.WkTh + (32 * kt + r) * LDB + 8 * h, sb.WkTl + (32 * kt + r) * LDB + 8 * h, dh_, dl_, acc);
          float4 f0, f1, f2, f3;
          acc_to_frag(acc, f0, f1, f2, f3);
          dz2[4 * kt] = f4_mul(f0, gp2[4 * kt]);
          dz2[4 * kt + 1] = f4_mul(f1, gp2[4 * kt + 1]);
          dz2[4 * kt + 2] = f4_mul(f2, gp2[4 * kt + 2]);
          dz2[4 * kt + 3] = f4_mul(f3, gp2[4 * kt + 3]);
        }
      }
      GRL_SCHED_BARRIER();
      PHS(11);  // dZ2 = (dK Wk) * gelu'(z2): 24 MFMAs from LDS fragments
      bf16x8 zh[4], zl[4];
      split_frags<64>(dz2, zh, zl);
      // ---- dW2 += dZ2^T g1, db2 += column sums of dZ2      (register-level transposes, split-bf16 products)
      {
        const TTile tg0 = transpose_split(cf.g1h[0], cf.g1h[1], cf.g1l[0], cf.g1l[1], sel0, sel1);
        const TTile tg1 = transpose_split(cf.g1h[2], cf.g1h[3], cf.g1l[2], cf.g1l[3], sel0, sel1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const TTile tz = transpose_split(zh[2 * nt], zh[2 * nt + 1], zl[2 * nt], zl[2 * nt + 1], sel0, sel1, &db2[nt]);
          mma_tn_bf_acc(tz, tg0, accA[nt][0]);
          mma_tn_bf_acc(tz, tg1, accA[nt][1]);
        }
      }
      GRL_SCHED_BARRIER();
      PHS(12);  // split dZ2, dW2: 4 transposed tiles + 24 MFMAs
      // ---- dZ1 = (dZ2 W2) * gelu'(z1)     (split-bf16: rows of W2^T)
      float4 dz1[8];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16 acc = zero16();
        mma_wx_bf<64>(sb.W2Th + (32 * kt + r) * LDB + 8 * h, sb.W2Tl + (32 * kt + r) * LDB + 8 * h, zh, zl, acc);
        float4 f0, f1, f2, f3;
        acc_to_frag(acc, f0, f1, f2, f3);
        dz1[4 * kt] = f4_mul(f0, gp1[4 * kt]);
        dz1[4 * kt + 1] = f4_mul(f1, gp1[4 * kt + 1]);
        dz1[4 * kt + 2] = f4_mul(f2, gp1[4 * kt + 2]);
        dz1[4 * kt + 3] = f4_mul(f3, gp1[4 * kt + 3]);
      }
      GRL_SCHED_BARRIER();
      PHS(13);  // dZ1: 24 MFMAs
      // ---- dW1 += dZ1^T phi, db1 += column sums of dZ1      (phi: one 16-column fragment, columns 16..31 of its tile are zero)
      {
        bf16x8 yh[4], yl[4];
        split_frags<64>(dz1, yh, yl);
        u32x4 z4 = {0u, 0u, 0u, 0u};
        const bf16x8 zero8 = __builtin_bit_cast(bf16x8, z4);
        const TTile tp = transpose_split(cf.ph[0], zero8, cf.pl[0], zero8, sel0, sel1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const TTile ty = transpose_split(yh[2 * nt], yh[2 * nt + 1], yl[2 * nt], yl[2 * nt + 1], sel0, sel1, &db1[nt]);
          mma_tn_bf_acc(ty, tp, accB[nt]);
        }
      }
      GRL_SCHED_BARRIER();
      PHS(14);  // split dZ1, dW1: 3 transposed tiles + 12 MFMAs
    }
    cur = nxt;
  }
  pc.flush(2);

  // ---- fold the four waves' accumulators through LDS (the weight images are dead): pairs (1 -> 0, 3 -> 2), then 2 -> 0; every
  //      lane reads back exactly the slots its partner lane wrote ([register][lane]: conflict-free, no address arithmetic), so
  //      the order is fixed and ONE partial row per workgroup leaves (a quarter of the slab the folding launch has to read)
  constexpr int NACC = 10 * 16 + 4;
  float* fold = smem_raw;
  asm_acc_drain();   // the last in-place (asm) MFMAs of the weight-gradient accumulators have finished before they are read
  auto visit = [&](auto&& f) {
    int k = 0;
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_) {
#pragma unroll
      for (int i = 0; i < 16; ++i) accB[a_][i] = f(accB[a_][i], k++);
#pragma unroll
      for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accA[a_][b_][i] = f(accA[a_][b_][i], k++); accK[a_][b_][i] = f(accK[a_][b_][i], k++); }
      db1[a_] = f(db1[a_], k++);
      db2[a_] = f(db2[a_], k++);
    }
  };
  __syncthreads();
  if (wave & 1) visit([&](float v_, int k) { fold[((wave >> 1) * NACC + k) * 64 + lane] = v_; return v_; });
  __syncthreads();
  if (!(wave & 1)) visit([&](float v_, int k) { return v_ + fold[((wave >> 1) * NACC + k) * 64 + lane]; });
  __syncthreads();
  if (wave == 2) visit([&](float v_, int k) { fold[k * 64 + lane] = v_; return v_; });
  __syncthreads();
  if (wave != 0) return;
  visit([&](float v_, int k) { return v_ + fold[k * 64 + lane]; });
  // ---- write the workgroup's partial.  acc element rho of lane (j = r, h): D[n = 8q+4h+u][col j]
  float* out = partial + (size_t)blockIdx.x * EDGE_PARTIAL;
  float* oW1 = out, *ob1 = out + 64 * 14, *oW2 = ob1 + 64, *ob2 = oW2 + 64 * 64, *oWk = ob2 + 64;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) {
      const int n = 32 * nt + (rho & 3) + 8 * (rho >> 2) + 4 * h;
      if (r < 14) oW1[n * 14 + r] = accB[nt][rho];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        oW2[n * 64 + 32 * kt + r] = accA[nt][kt][rho];
        oWk[n * 64 + 32 * kt + r] = accK[nt][kt][rho];
      }
    }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {   // the other 16 rows of every column were summed by lane ^ 32
    const float v1 = db1[nt] + __shfl_xor(db1[nt], 32, 64), v2 = db2[nt] + __shfl_xor(db2[nt], 32, 64);
    if (h == 0) { ob1[32 * nt + r] = v1; ob2[32 * nt + r] = v2; }
  }
}
#endif   // GRL_LEGACY32

}  // namespace

extern "C" {

#if !GRL_PREC   // shape queries: shared by both precision builds of this file
// launch shape of the forward for n_dst destination nodes (host-side partitioning, ops.build_edge_set): the image kind its kernel copies
// (grl_weight_images: 1 = the one-workgroup-per-tile 32-row kernel of small launches, 0 = the 16-row kernel), the wave slots a balanced
// partition has to cover (0: the launch ignores partitions) and the nodes per round-robin chunk
int grl_edge_fwd_image_kind(int n_dst) { return (n_dst + TD - 1) / TD <= GRL_FWD_SPLIT_TILES ? 1 : 0; }
int grl_edge_fwd_chunk_nodes(int n_dst);   // edge_conv16.hip
int grl_edge_bwd_chunk_nodes(int n_src);
int grl_edge_fwd_slots(int n_dst) {
  if ((n_dst + TD - 1) / TD <= GRL_FWD_SPLIT_TILES) return 0;
  const int npw = grl_edge_fwd_chunk_nodes(n_dst), chunks = (n_dst + npw - 1) / npw;
  int blocks = (chunks + 3) / 4;
  const int cap = grl_edge_fwd_chunk_nodes(-1);   // (-1: the grid cap, workgroups)
  if (blocks > cap) blocks = cap;
  return 4 * blocks;
}
int grl_edge_partial_size() { return EDGE_PARTIAL; }
int grl_edge_bwd_blocks(int n_edges) {
  const int b = ((n_edges + 1) / 2 + 3) / 4;
  return b < 1 ? 1 : (b < 256 ? b : 256);
}
#else
int grl_edge_bwd_blocks(int n_edges);
#endif

// 16-row-tile kernels (edge_conv16.hip): forward, d x_src and messages; the weight-gradient kernel and the few-tile SPLIT forward stay here
#ifndef GRL_EDGE16
#define GRL_EDGE16 1
#endif
int GRL_ENTRY(grl_edge16_launch)(int mode, const st_t* x_in, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                 const int* e_dst, const int* erow, int per_edge, int n_anchor, int n_edges, int anchor_is_dst,
                                 const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* Wk, st_t* out, const st_t* dres, const int* split, int n_slots, const void* wimg,
                                 hipStream_t stream);

// fused 16-row backward (edge_conv16.hip edge_bwd16_kernel): d x_src and the weight gradients in one launch, one chain recompute
#ifndef GRL_EDGE_BWD16
#define GRL_EDGE_BWD16 1
#endif
int GRL_ENTRY(grl_edge_bwd16_launch)(const st_t* x_src, const st_t* dmsg, const float* pos_src, const float* pos_dst, const int* rowptr_s,
                                     const int* src_s, const int* dst_s, const int* erow, int per_edge, int n_src, int n_edges,
                                     const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                                     const float* Wk, const st_t* dres, st_t* dx_src, float* partial, int blocks, const int* split,
                                     const void* wimg, hipStream_t stream);

// grl_edge_conv_fwd_balanced: the same with split_d [n_slots + 1] (n_slots = a multiple of 4, at most 3072): node boundaries of an
// in-edge-balanced partition of the destination-sorted CSR over the launch's wave slots; NULL = round-robin chunks.
// wimg16 / wimg32: optional pre-split weight images of this forward pass (grl_weight_images kinds 0 / 1; the launch takes the one its
// kernel needs -- grl_edge_fwd_image_kind(n_dst) -- and stages the weights itself when that one is NULL).
int GRL_ENTRY(grl_edge_conv_fwd_balanced)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* Wk, st_t* x1, const int* split_d, int n_slots, const void* wimg16,
                      const void* wimg32, hipStream_t stream);
int GRL_ENTRY(grl_edge_conv_fwd)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* Wk, st_t* x1, hipStream_t stream) {
  return GRL_ENTRY(grl_edge_conv_fwd_balanced)(x_src, pos_src, pos_dst, rowptr, e_src, e_dst, n_dst, grid, dim, W1, b1, W2, b2, Wk, x1, nullptr,
                                               0, nullptr, nullptr, stream);
}
int GRL_ENTRY(grl_edge_conv_fwd_balanced)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* Wk, st_t* x1, const int* split_d, int n_slots, const void* wimg16,
                      const void* wimg32, hipStream_t stream) {
  if (n_dst <= 0) return 0;
  EdgeParams p{x_src, pos_src, pos_dst, rowptr, e_src, e_dst, grid, W1, b1, W2, b2, Wk, n_dst, dim};
  p.wimg = wimg32;
  const int n_tiles = (n_dst + TD - 1) / TD;
  const size_t smem = sizeof(ChainW), smem_split = smem + sizeof(float4) * FWD_WAVES * 8 * 64;
  GRL_ONCE(hipFuncSetAttribute((const void*)edge_conv_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_split));
  if (n_tiles <= GRL_FWD_SPLIT_TILES) {   // fewer tiles than SIMD groups: spread each tile's passes over a workgroup
    grl_prof_begin_replay("edge_conv_fwd_kernel", stream);
    hipLaunchKernelGGL(edge_conv_fwd_kernel<true>, dim3(n_tiles), dim3(64 * FWD_WAVES), smem_split, stream, p, x1);
    grl_prof_end_replay(stream);
    GRL_CHECK_LAUNCH();
    return 0;
  }
#if GRL_LEGACY32
  if (!GRL_EDGE16) {
    GRL_ONCE(hipFuncSetAttribute((const void*)edge_conv_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int blocks = (n_tiles + FWD_WAVES - 1) / FWD_WAVES;
    if (blocks > GRL_FWD_MAX_BLOCKS) blocks = GRL_FWD_MAX_BLOCKS;
    hipLaunchKernelGGL(edge_conv_fwd_kernel<false>, dim3(blocks), dim3(64 * FWD_WAVES), smem, stream, p, x1);
    GRL_CHECK_LAUNCH();
    return 0;
  }
#endif
  (void)smem;
  grl_prof_begin_replay("edge_conv_fwd_kernel", stream);
  const int rc16 = GRL_ENTRY(grl_edge16_launch)(0, x_src, pos_src, pos_dst, rowptr, e_src, e_dst, nullptr, 0, n_dst, 0, 1, grid, dim, W1, b1, W2,
                                                b2, Wk, x1, nullptr, split_d, n_slots, wimg16, stream);
  grl_prof_end_replay(stream);
  return rc16;
}

// The same edge set in both orders: destination-sorted (rowptr, e_src, e_dst: the forward's arrays) for the weight kernel and
// source-sorted (rowptr_s [n_src+1], src_s [E], dst_s [E]) for the d x_src kernel.  dx_src [n_src,16,64] is fully overwritten:
// dx_src = (dres ? dres : 0) + sum over out-edges; dres [n_src,16,64] = gradient of another use of x_src (the residual branch), or NULL.
// partial must hold grl_edge_bwd_blocks(n_edges) rows of grl_edge_partial_size() floats.
// grl_edge_conv_bwd_balanced: the same with ``split_s`` [n_slots_s + 1], n_slots_s = 4 * grl_edge_bwd_blocks(n_edges) -- node boundaries of
// an edge-balanced partition of the source-sorted CSR over the launch's wave slots (split_s[0] = 0, split_s[last] = n_src, non-decreasing);
// NULL = none; an array built for another slot count is IGNORED (round-robin chunks), never read past its end.
// wimg16: optional Edge16Image of this step's weights (grl_weight_images kind 0); NULL = the kernel stages the five images itself.
int GRL_ENTRY(grl_edge_conv_bwd_balanced)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s, int n_src,
                      const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                      const float* Wk, const st_t* dx1, const st_t* dres, st_t* dx_src, float* partial, const int* split_s,
                      int n_slots_s, const void* wimg16, hipStream_t stream);
int GRL_ENTRY(grl_edge_conv_bwd)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s, int n_src,
                      const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                      const float* Wk, const st_t* dx1, const st_t* dres, st_t* dx_src, float* partial, hipStream_t stream) {
  return GRL_ENTRY(grl_edge_conv_bwd_balanced)(x_src, pos_src, pos_dst, rowptr, e_src, e_dst, n_dst, n_edges, rowptr_s, src_s, dst_s, n_src,
                                               grid, dim, W1, b1, W2, b2, Wk, dx1, dres, dx_src, partial, nullptr, 0, nullptr, stream);
}
int GRL_ENTRY(grl_edge_conv_bwd_balanced)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s, int n_src,
                      const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                      const float* Wk, const st_t* dx1, const st_t* dres, st_t* dx_src, float* partial, const int* split_s,
                      int n_slots_s, const void* wimg16, hipStream_t stream) {
  if (split_s && n_slots_s != 4 * grl_edge_bwd_blocks(n_edges)) split_s = nullptr;   // built for another launch shape: not usable
  if (n_edges <= 0) {   // an empty edge set: d x_src is the residual branch alone, the weight gradients are zero (the caller sums
    if (n_src > 0) {    // grl_edge_bwd_blocks(n_edges) partial rows: they must hold zeros, not whatever the allocation held)
      if (dres) hipMemcpyAsync(dx_src, dres, sizeof(st_t) * (size_t)n_src * O * C, hipMemcpyDeviceToDevice, stream);
      else hipMemsetAsync(dx_src, 0, sizeof(st_t) * (size_t)n_src * O * C, stream);
    }
    hipMemsetAsync(partial, 0, sizeof(float) * (size_t)grl_edge_bwd_blocks(n_edges) * EDGE_PARTIAL, stream);
    return 0;
  }
  const int blocks = grl_edge_bwd_blocks(n_edges);
#if GRL_LEGACY32
  if (!GRL_EDGE_BWD16) {
    EdgeParams pd{x_src, pos_src, pos_dst, rowptr, e_src, e_dst, grid, W1, b1, W2, b2, Wk, n_dst, dim};
    EdgeParams ps{x_src, pos_src, pos_dst, rowptr_s, src_s, dst_s, grid, W1, b1, W2, b2, Wk, n_src, dim};
    const size_t smem_x = sizeof(ChainW);
    size_t smem_w = smem_x + sizeof(BwdW);
    if (smem_w < sizeof(float) * 2 * (10 * 16 + 4) * 64) smem_w = sizeof(float) * 2 * (10 * 16 + 4) * 64;   // the end-of-launch fold
    GRL_ONCE(hipFuncSetAttribute((const void*)edge_conv_bwd_x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_x); hipFuncSetAttribute((const void*)edge_conv_bwd_w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_w));
    grl_prof_begin("edge_conv_bwd_x_kernel", stream);
    if (GRL_EDGE16) {
      GRL_ENTRY(grl_edge16_launch)(1, dx1, pos_src, pos_dst, rowptr_s, src_s, dst_s, nullptr, 0, n_src, n_edges, 0, grid, dim, W1, b1, W2, b2,
                                   Wk, dx_src, dres, nullptr, 0, nullptr, stream);
    } else {
      const int n_tiles_s = (n_src + TD - 1) / TD;
      int xblocks = (n_tiles_s + 3) / 4;
      if (xblocks > GRL_FWD_MAX_BLOCKS) xblocks = GRL_FWD_MAX_BLOCKS;   // two 4-wave workgroups per CU, like the forward
      hipLaunchKernelGGL(edge_conv_bwd_x_kernel, dim3(xblocks < 1 ? 1 : xblocks), dim3(256), smem_x, stream, ps, dx1, dx_src, dres);
    }
    grl_prof_end(stream);
    GRL_CHECK_LAUNCH();
    grl_prof_begin("edge_conv_bwd_w_kernel", stream);
    hipLaunchKernelGGL(edge_conv_bwd_w_kernel, dim3(blocks), dim3(256), smem_w, stream, pd, dx1, partial, n_edges);
    grl_prof_end(stream);
    GRL_CHECK_LAUNCH();
    return 0;
  }
#endif
  (void)rowptr; (void)e_src; (void)e_dst; (void)n_dst;   // the destination-sorted view is the legacy weight kernel's
  grl_prof_begin("edge_bwd16_kernel", stream);
  const int rc = GRL_ENTRY(grl_edge_bwd16_launch)(x_src, dx1, pos_src, pos_dst, rowptr_s, src_s, dst_s, nullptr, 0, n_src, n_edges, grid, dim,
                                                  W1, b1, W2, b2, Wk, dres, dx_src, partial, blocks, split_s, wimg16, stream);
  grl_prof_end(stream);
  return rc;
}

// ---- attention aggregation, edge side: messages per edge (rows in destination-sorted edge order) and the backward for a per-edge
//      incoming gradient.  s2d [E]: for the i-th edge of the SOURCE-sorted order, its row in the destination-sorted order.
int GRL_ENTRY(grl_edge_messages_fwd)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                     const int* e_dst, int n_dst, int n_edges, const float* grid, int dim, const float* W1,
                                     const float* b1, const float* W2, const float* b2, const float* Wk, st_t* msg,
                                     hipStream_t stream) {
  if (n_edges <= 0) return 0;
#if GRL_LEGACY32
  if (!GRL_EDGE16) {
    EdgeParams p{x_src, pos_src, pos_dst, rowptr, e_src, e_dst, grid, W1, b1, W2, b2, Wk, n_dst, dim};
    const size_t smem = sizeof(ChainW);
    GRL_ONCE(hipFuncSetAttribute((const void*)edge_msg_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainW)));
    int blocks = ((n_edges + 1) / 2 + 3) / 4;
    if (blocks > GRL_FWD_MAX_BLOCKS) blocks = GRL_FWD_MAX_BLOCKS;
    hipLaunchKernelGGL(edge_msg_fwd_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), smem, stream, p, msg, n_edges);
    GRL_CHECK_LAUNCH();
    return 0;
  }
#endif
  return GRL_ENTRY(grl_edge16_launch)(2, x_src, pos_src, pos_dst, rowptr, e_src, e_dst, nullptr, 0, n_dst, n_edges, 1, grid, dim, W1, b1,
                                      W2, b2, Wk, msg, nullptr, nullptr, 0, nullptr, stream);
}

int GRL_ENTRY(grl_edge_messages_bwd)(const st_t* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                     const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                                     const int* s2d, int n_src, const float* grid, int dim, const float* W1, const float* b1,
                                     const float* W2, const float* b2, const float* Wk, const st_t* dmsg, const st_t* dres,
                                     st_t* dx_src, float* partial, hipStream_t stream) {
  if (n_edges <= 0) {   // an empty edge set: d x_src is the residual branch alone, the weight gradients are zero (the caller sums
    if (n_src > 0) {    // grl_edge_bwd_blocks(n_edges) partial rows: they must hold zeros, not whatever the allocation held)
      if (dres) hipMemcpyAsync(dx_src, dres, sizeof(st_t) * (size_t)n_src * O * C, hipMemcpyDeviceToDevice, stream);
      else hipMemsetAsync(dx_src, 0, sizeof(st_t) * (size_t)n_src * O * C, stream);
    }
    hipMemsetAsync(partial, 0, sizeof(float) * (size_t)grl_edge_bwd_blocks(n_edges) * EDGE_PARTIAL, stream);
    return 0;
  }
  const int blocks = grl_edge_bwd_blocks(n_edges);
#if GRL_LEGACY32
  if (!GRL_EDGE_BWD16) {
    EdgeParams pd{x_src, pos_src, pos_dst, rowptr, e_src, e_dst, grid, W1, b1, W2, b2, Wk, n_dst, dim};
    EdgeParams ps{x_src, pos_src, pos_dst, rowptr_s, src_s, dst_s, grid, W1, b1, W2, b2, Wk, n_src, dim};
    pd.per_edge = 1;            // the weight kernel walks the destination-sorted order: row = edge position
    ps.per_edge = 1;
    ps.erow = s2d;              // the d x_src kernel walks the source-sorted order
    const size_t smem_x = sizeof(ChainW);
    size_t smem_w = smem_x + sizeof(BwdW);
    if (smem_w < sizeof(float) * 2 * (10 * 16 + 4) * 64) smem_w = sizeof(float) * 2 * (10 * 16 + 4) * 64;
    GRL_ONCE(hipFuncSetAttribute((const void*)edge_conv_bwd_x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_x); hipFuncSetAttribute((const void*)edge_conv_bwd_w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_w));
    const int n_tiles_s = (n_src + TD - 1) / TD;
    int xblocks = (n_tiles_s + 3) / 4;
    if (xblocks > GRL_FWD_MAX_BLOCKS) xblocks = GRL_FWD_MAX_BLOCKS;
    if (GRL_EDGE16)
      GRL_ENTRY(grl_edge16_launch)(1, dmsg, pos_src, pos_dst, rowptr_s, src_s, dst_s, s2d, 1, n_src, n_edges, 0, grid, dim, W1, b1, W2, b2, Wk,
                                   dx_src, dres, nullptr, 0, nullptr, stream);
    else
      hipLaunchKernelGGL(edge_conv_bwd_x_kernel, dim3(xblocks < 1 ? 1 : xblocks), dim3(256), smem_x, stream, ps, dmsg, dx_src, dres);
    GRL_CHECK_LAUNCH();
    hipLaunchKernelGGL(edge_conv_bwd_w_kernel, dim3(blocks), dim3(256), smem_w, stream, pd, dmsg, partial, n_edges);
    GRL_CHECK_LAUNCH();
    return 0;
  }
#endif
  (void)rowptr; (void)e_src; (void)e_dst; (void)n_dst;
  return GRL_ENTRY(grl_edge_bwd16_launch)(x_src, dmsg, pos_src, pos_dst, rowptr_s, src_s, dst_s, s2d, 1, n_src, n_edges, grid, dim, W1, b1,
                                          W2, b2, Wk, dres, dx_src, partial, blocks, nullptr, nullptr, stream);
}

#if defined(GRL_PHASE_PROF) && !GRL_PREC
// diagnostic build only: out [3][24] = accumulated phase ticks of (forward, d x_src, weights) kernels; reset != 0 clears them
int grl_edge_phase_read(unsigned long long* out72, int reset) {
  hipMemcpyFromSymbol(out72, HIP_SYMBOL(g_ephase), sizeof(unsigned long long) * 72);
  if (reset) { unsigned long long z[72] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_ephase), z, sizeof(z)); }
  return 0;
}
#endif
}  // extern "C"
