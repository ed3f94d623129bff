// Edge pipeline on 16-row tiles (round 2): forward, d x_src and message kernels of the separable fiber-bundle convolution with ONE EDGE
// (= its 16 orientation rows) per wave pass on v_mfma_f32_16x16x32_bf16, instead of two edges (32 rows) on 32x32x16.
//
// Why (DESIGN.md findings 13, 17; profiles/r02_edge_phase_*.txt): the 32-row kernels need 218-232 registers, so only two waves share a
// SIMD; a wave issues one vector instruction per ~5 cycles, the exact-erf GELU between the layers is ~190 dependent-ish instructions per
// 32x32 tile, and the phase timing shows every wave running at its own issue pace with the vector pipe ~50 % busy, plus 25-45 % of the
// time in exposed index -> position -> row round trips at pass and tile heads.  Half the rows per wave halves every per-row register
// array (activations, fragments, gathered row, destination accumulator): ~160 registers, THREE waves per SIMD.  One edge per pass also
// removes the half-empty passes of odd edge counts and the two-destination select of the message epilogue, and makes all per-pass
// metadata wave-uniform: it is fetched for 64 edges at a time (one coalesced load per array + one gather of the positions) and handed out
// with v_readlane -- no dependent global load is left inside the pass loop.
//
// Layout: lane l = (row r = l & 15 = orientation, k-group g = l >> 4).  D[n][r] += sum_k W[n][k] X[r][k] with the weight tile (16 output
// features n) on the A side and the activation rows on the B side; a 16x16 accumulator (f32x4) of n-tile nt holds features
// 16 nt + 4 g + u in element u -- two consecutive n-tiles ARE the 8 B-operand elements of one K-step of the next layer (k order inside a
// 32-block: position 8 g + j <-> feature 16 (j >> 2) + 4 g + (j & 3); the weight images are staged in that order), so the chain stays
// in registers exactly as in the 32-row kernels.  Three waves per SIMD: every MFMA group is fenced (all fragment loads, then the
// MFMAs, then a read of the accumulator: DESIGN.md finding 3).
#include "grl_common.h"
#include <cstdlib>

namespace {

constexpr int C = 64, O = 16;
typedef float f32x4v __attribute__((ext_vector_type(4)));
GRL_DEVINL f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

#ifndef GRL_E16_WAVES
#define GRL_E16_WAVES 4                  // waves per workgroup
#endif
#ifndef GRL_E16_WGS
#define GRL_E16_WGS 3                    // workgroups per CU the register budget is cut for (and the grid is capped at)
#endif
constexpr int E16_WAVES = GRL_E16_WAVES, E16_THREADS = 64 * E16_WAVES;
constexpr int LD1 = 32 + 8;   // bf16 elements per image row, layer 1 (K = 14 padded to one 32-deep step)
constexpr int LD2 = 64 + 8;   // layers 2 and 3
struct ChainW16 {
  unsigned short W1h[64 * LD1], W1l[64 * LD1];
  unsigned short W2h[64 * LD2], W2l[64 * LD2];
  unsigned short Wkh[64 * LD2], Wkl[64 * LD2];
  float b1s[64], b2s[64], grid_s[64];
};

// image[n][32 s + 8 g + j] = W[n][32 s + 16 (j >> 2) + 4 g + (j & 3)]   (zero beyond KSRC); one (n, s, g) item per thread and step:
// two 16-byte global loads (when aligned), one 16-byte LDS store per image
template <int KSRC, int KPAD, int NT>
GRL_DEVINL void stage16(unsigned short* hi, unsigned short* lo, const float* __restrict__ W, int ld) {
  constexpr int ITEMS = 64 * (KPAD / 32) * 4;
  for (int idx = threadIdx.x; idx < ITEMS; idx += NT) {
    const int g = idx & 3, s = (idx >> 2) % (KPAD / 32), n = idx / (4 * (KPAD / 32));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
      v[j] = f < KSRC ? W[n * KSRC + f] : 0.f;
    }
    bf16x8 h, l;
    split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), h, l);
    *reinterpret_cast<bf16x8*>(hi + n * ld + 32 * s + 8 * g) = h;
    GRL_LO(*reinterpret_cast<bf16x8*>(lo + n * ld + 32 * s + 8 * g) = l;)
  }
}

struct Edge16Params {
  const st_t* x_in;       // rows gathered per edge: x_src [Ns,16,64] (forward / messages) or dx1 [Nd or E,16,64] (d x_src kernel)
  const float* pos_src;   // [Ns,3]
  const float* pos_dst;   // [Nd,3]
  const int* rowptr;      // [Na+1] anchor CSR (anchor = destination in the forward, source in the backward)
  const int* e_src;       // [E] in anchor-sorted order
  const int* e_dst;       // [E]
  const int* erow;        // optional [E]: row of x_in for the i-th edge (attention backward: per-edge gradient rows); else the node id
  const float* grid, *W1, *b1, *W2, *b2, *Wk;
  int n_anchor, n_edges, dim;
  int anchor_is_dst;      // 1: forward order (gather by source, accumulate per destination); 0: source order (the reverse)
  int per_edge;           // 1: x_in rows are per edge (erow or the edge's own position)
  int npw;                // anchor nodes per wave chunk, 1..16 (host: as large as still leaves every wave slot several chunks)
};

GRL_DEVINL void load_w16(ChainW16& s, const Edge16Params& p) {
  stage16<14, 32, E16_THREADS>(s.W1h, s.W1l, p.W1, LD1);
  stage16<64, 64, E16_THREADS>(s.W2h, s.W2l, p.W2, LD2);
  stage16<64, 64, E16_THREADS>(s.Wkh, s.Wkl, p.Wk, LD2);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    s.b1s[i] = p.b1[i];
    s.b2s[i] = p.b2[i];
    s.grid_s[i] = i < 48 ? p.grid[i] : 0.f;
  }
}

// one fenced group: acc(16 features of n-tile nt x 16 rows) = init + sum over KS K-steps of W-tile . X, split-bf16
template <int KS, class Epi>
GRL_DEVINL void group16(const unsigned short* whi, const unsigned short* wlo, const bf16x8 (&xh)[KS], const bf16x8 (&xl)[KS], f32x4v acc,
                        Epi&& epi) {
  bf16x8 wh[KS], wl[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#ifdef GRL_E16_NOLDS
    wh[s] = xh[s];
    wl[s] = xl[s];
#else
    wh[s] = *reinterpret_cast<const bf16x8*>(whi + 32 * s);
    GRL_LO(wl[s] = *reinterpret_cast<const bf16x8*>(wlo + 32 * s);)
#endif
#ifdef GRL_E16_LDSONLY
    asm volatile("" ::"v"(wh[s]), "v"(wl[s]));
#endif
  }
  __builtin_amdgcn_sched_barrier(0);
#if !defined(GRL_E16_NOMFMA) && !defined(GRL_E16_LDSONLY)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    acc = mfma16(wh[s], xh[s], acc);
    GRL_LO(acc = mfma16(wl[s], xh[s], acc);)
    GRL_LO(acc = mfma16(wh[s], xl[s], acc);)
  }
#endif
  epi(acc);
  __builtin_amdgcn_sched_barrier(0);
}

GRL_DEVINL float4 v4(const f32x4v& a) { return make_float4(a[0], a[1], a[2], a[3]); }
// timing knock-outs (diagnostic builds only; results are wrong): -DGRL_E16_NOGELU, -DGRL_E16_NOMFMA, -DGRL_E16_NOGATHER
#ifdef GRL_E16_NOGELU
#define GELU16(x) (x)
#elif defined(GRL_E16_SCALAR_GELU)
GRL_DEVINL float4 gelu4s(float4 x) {
  float4 g, gp;
  gelu_both4(x, g, gp);
  return g;
}
#define GELU16(x) gelu4s(x)
#else
#define GELU16(x) gelu4(x)
#endif

// The chain for this lane's row: (a, b) -> K tiles handed to k_epi(nt, float4 of features 16 nt + 4 g + 0..3)
template <class KEpi>
GRL_DEVINL void chain16(const ChainW16& w, float a, float b, int r, int g, KEpi&& k_epi) {
  // polynomial features (ponita.py:233-244), this lane's four: f = 4 g .. 4 g + 3 of [a b | aa ab ba bb | aaa aab aba abb baa bab bba bbb]
  const float aa = a * a, ab = a * b, bb = b * b;
  float4 phi;
  if (g == 0) phi = make_float4(a, b, aa, ab);
  else if (g == 1) phi = make_float4(ab, bb, aa * a, aa * b);
  else if (g == 2) phi = make_float4(ab * a, ab * b, ab * a, ab * b);
  else phi = make_float4(bb * a, bb * b, 0.f, 0.f);
  bf16x8 ph[1], pl[1];
  split_pair(phi, make_float4(0.f, 0.f, 0.f, 0.f), ph[0], pl[0]);
  float4 g1[4], g2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 bq = *reinterpret_cast<const float4*>(w.b1s + 16 * nt + 4 * g);
    group16<1>(w.W1h + (16 * nt + r) * LD1 + 8 * g, w.W1l + (16 * nt + r) * LD1 + 8 * g, ph, pl, f32x4v{bq.x, bq.y, bq.z, bq.w},
               [&](const f32x4v& acc) { g1[nt] = GELU16(v4(acc)); });
  }
  bf16x8 xh[2], xl[2];
  split_pair(g1[0], g1[1], xh[0], xl[0]);
  split_pair(g1[2], g1[3], xh[1], xl[1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 bq = *reinterpret_cast<const float4*>(w.b2s + 16 * nt + 4 * g);
    group16<2>(w.W2h + (16 * nt + r) * LD2 + 8 * g, w.W2l + (16 * nt + r) * LD2 + 8 * g, xh, xl, f32x4v{bq.x, bq.y, bq.z, bq.w},
               [&](const f32x4v& acc) { g2[nt] = GELU16(v4(acc)); });
  }
  bf16x8 yh[2], yl[2];
  split_pair(g2[0], g2[1], yh[0], yl[0]);
  split_pair(g2[2], g2[3], yh[1], yl[1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
    group16<2>(w.Wkh + (16 * nt + r) * LD2 + 8 * g, w.Wkl + (16 * nt + r) * LD2 + 8 * g, yh, yl, f32x4v{0.f, 0.f, 0.f, 0.f},
               [&](const f32x4v& acc) { k_epi(nt, v4(acc)); });
}

constexpr int NPW_MAX = 16;   // anchor nodes per wave chunk (their rowptr entries live on the lanes)

// MODE 0: forward       out[anchor] = sum over its edges of K_e * x_in[other(e)]                    (anchor = destination)
// MODE 1: d x_src       out[anchor] = (dres ? dres[anchor] : 0) + sum of K_e * x_in[other(e) | row]   (anchor = source)
// MODE 2: messages      msg[edge position] = K_e * x_in[other(e)], no accumulation                    (anchor = destination)
template <int MODE>
__global__ __launch_bounds__(E16_THREADS, GRL_E16_WGS) void edge16_kernel(Edge16Params p, st_t* __restrict__ out, const st_t* __restrict__ dres) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  ChainW16& s = *reinterpret_cast<ChainW16*>(smem_raw);
  load_w16(s, p);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const float gx = s.grid_s[3 * r], gy = s.grid_s[3 * r + 1], gz = s.grid_s[3 * r + 2];
  const int* e_other = p.anchor_is_dst ? p.e_src : p.e_dst;
  const float* pos_anchor = p.anchor_is_dst ? p.pos_dst : p.pos_src;
  const float* pos_other = p.anchor_is_dst ? p.pos_src : p.pos_dst;
  const int NPW = p.npw;
  const int n_chunks = (p.n_anchor + NPW - 1) / NPW;
  for (int chunk = blockIdx.x * E16_WAVES + wave; chunk < n_chunks; chunk += gridDim.x * E16_WAVES) {
    const int n0 = chunk * NPW, nn = min(NPW, p.n_anchor - n0);
    // chunk metadata on the lanes: rowptr (lanes 0..nn) and the anchor nodes' positions (lanes 0..nn-1)
    const int rp = p.rowptr[n0 + min(lane, nn)];
    const int an = n0 + min(lane, nn - 1);
    const float pax = pos_anchor[3 * an], pay = pos_anchor[3 * an + 1], paz = pos_anchor[3 * an + 2];
    const int E0 = __builtin_amdgcn_readlane(rp, 0), E1 = __builtin_amdgcn_readlane(rp, nn);
    int node = 0;                         // index inside the chunk of the node being accumulated
    int node_end = __builtin_amdgcn_readlane(rp, 1);
    float4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&](int j) {             // rows of chunk node j are complete
      if (MODE == 2) return;
      st_t* o = out + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 v = acc[t];
        if (MODE == 1 && dres) v = f4_add(v, ld4(dres + ((size_t)(n0 + j) * O + r) * C + 4 * g + 16 * t));
        st4(o + 16 * t, v);
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    for (int eb = E0; eb < E1; eb += 64) {
      // metadata of the next (up to) 64 edges: the other end's node, its position, the row of x_in
      const int ee = min(eb + lane, E1 - 1);
      const int oth = e_other[ee];
      const int xrow = p.per_edge ? (p.erow ? p.erow[ee] : ee) : oth;
      const float pox = pos_other[3 * oth], poy = pos_other[3 * oth + 1], poz = pos_other[3 * oth + 2];
      const int nb = min(64, E1 - eb);
#pragma unroll 1
      for (int k = 0; k < nb; ++k) {
        const int e = eb + k;
        while (e >= node_end) {           // the edge belongs to a later node of the chunk: finish the nodes before it
          flush(node);
          ++node;
          node_end = __builtin_amdgcn_readlane(rp, node + 1);
        }
        const int row_in = __builtin_amdgcn_readlane(xrow, k);
        const st_t* xs = p.x_in + ((size_t)row_in * O + r) * C + 4 * g;
        float4 xv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#ifdef GRL_E16_NOGATHER
          xv[t] = make_float4(1.f, 1.f, 1.f, 1.f);
#else
          xv[t] = ld4(xs + 16 * t);      // this edge's gathered row, in flight behind the chain
#endif
        // r = pos_src - pos_dst (hepi.py:109-117), whichever end is the anchor
        float dx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pox), k)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pax), node));
        float dy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, poy), k)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pay), node));
        float dz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, poz), k)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, paz), node));
        if (!p.anchor_is_dst) { dx = -dx; dy = -dy; dz = -dz; }
        if (p.dim == 2) dz = 0.f;
        const float a = dx * gx + dy * gy + dz * gz;
        dx -= a * gx; dy -= a * gy; dz -= a * gz;
        const float b = sqrtf(dx * dx + dy * dy + dz * dz);
        st_t* mrow = MODE == 2 ? out + ((size_t)e * O + r) * C + 4 * g : nullptr;
        chain16(s, a, b, r, g, [&](int nt, const float4& kq) {
          const float4 m = f4_mul(kq, xv[nt]);
          if (MODE == 2) st4(mrow + 16 * nt, m);
          else acc[nt] = f4_add(acc[nt], m);
        });
      }
    }
    for (; node < nn; ++node) flush(node);   // the last node with edges and every trailing node without
  }
}

}  // namespace

extern "C" {

// Internal entry points used by edge_conv.hip's C-ABI functions when the 16-row kernels are selected (GRL_EDGE16).
int GRL_ENTRY(grl_edge16_launch)(int mode, const st_t* x_in, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                 const int* e_dst, const int* erow, int per_edge, int n_anchor, int n_edges, int anchor_is_dst,
                                 const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* Wk, st_t* out, const st_t* dres, hipStream_t stream) {
  if (n_anchor <= 0) return 0;
  // chunks: at least ~4 per wave slot of the chip (256 CUs x 12 waves) while the graph allows it
  int npw = n_anchor / (4 * 256 * E16_WAVES * GRL_E16_WGS);
  npw = npw < 1 ? 1 : (npw > NPW_MAX ? NPW_MAX : npw);
#ifdef GRL_E16_TUNE
  static const int env_npw = getenv("GRL_E16_NPW") ? atoi(getenv("GRL_E16_NPW")) : 0;
  if (env_npw > 0) npw = env_npw;
#endif
  Edge16Params p{x_in, pos_src, pos_dst, rowptr, e_src, e_dst, erow, grid, W1, b1, W2, b2, Wk, n_anchor, n_edges, dim, anchor_is_dst,
                 per_edge, npw};
  const int n_chunks = (n_anchor + npw - 1) / npw;
  int blocks = (n_chunks + E16_WAVES - 1) / E16_WAVES;
  int cap = 256 * GRL_E16_WGS;
#ifdef GRL_E16_TUNE                        // diagnostic builds: grid cap and chunk size from the environment
  static const int env_cap = getenv("GRL_E16_BLOCKS") ? atoi(getenv("GRL_E16_BLOCKS")) : 0;
  if (env_cap > 0) cap = env_cap;
#endif
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const size_t smem = sizeof(ChainW16);
  GRL_ONCE(hipFuncSetAttribute((const void*)edge16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainW16));
           hipFuncSetAttribute((const void*)edge16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainW16));
           hipFuncSetAttribute((const void*)edge16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainW16)));
  if (mode == 0) hipLaunchKernelGGL(edge16_kernel<0>, dim3(blocks), dim3(E16_THREADS), smem, stream, p, out, dres);
  else if (mode == 1) hipLaunchKernelGGL(edge16_kernel<1>, dim3(blocks), dim3(E16_THREADS), smem, stream, p, out, dres);
  else hipLaunchKernelGGL(edge16_kernel<2>, dim3(blocks), dim3(E16_THREADS), smem, stream, p, out, dres);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
