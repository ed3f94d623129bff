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

// (The 32-row message / d x_src / weight-gradient kernels of round 1, superseded by edge_conv16.hip in round 2 and compiled out since, were
// removed in round 6; the A/B figures they served are DESIGN.md findings 19 / 20.)
constexpr int EDGE_PARTIAL = 64 * 14 + 64 + 64 * 64 + 64 + 64 * 64;   // partial row: [W1 64x14 | b1 64 | W2 64x64 | b2 64 | Wk 64x64]

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
int GRL_ENTRY(grl_edge16_launch)(int mode, const st_t* x_in, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                 const int* e_dst, const int* erow, int per_edge, int n_anchor, int n_edges, int anchor_is_dst,
                                 const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* Wk, st_t* out, const st_t* dres, const int* split, int n_slots, const void* wimg,
                                 hipStream_t stream);

// fused 16-row backward (edge_conv16.hip edge_bwd16_kernel): d x_src and the weight gradients in one launch, one chain recompute
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
