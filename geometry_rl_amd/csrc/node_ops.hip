// HBM-bound node-side kernels of the HEPi / EMPN actor:
//   lift + node encoder            (reference hepi.py:136-143, to_from_sphere.py:4-9)
//   depthwise fiber convolution    (reference conv.py:88-90,108-109: x2[n,p,c] = 1/16 sum_o x1[n,o,c] fk[o,p,c] + bias[c])
//   slab reduction of per-workgroup weight-gradient partials
// All are streaming kernels: one coalesced pass over [N,16,64] fp32 node tensors, weights held in registers / LDS.
#include "grl_common.h"
#include "grl_report.h"
#include "grl_feat.h"
#include "grl_wimg_kernel.h"

namespace {

constexpr int C = 64, O = 16;
constexpr int KF_MAX = 8;  // scalars + vectors per node (7 in every reference config)
struct FiberWfs { const float* w[4]; int n; };
struct FiberFks { float* p[4]; };
struct FiberDfks { const float* p[4]; };

// ------------------------------------------------------------------------------------------------ lift + encode
// x[n,o,c] = sum_s scal[n,s] W[c,s] + sum_v (vec[n,v,:] . grid[o,:]) W[c,S+v]
//          = A[n,c] + sum_d grid[o,d] Bv[n,c,d],   A = scal W_s^T,  Bv[.,.,d] = vec[.,.,d] W_v^T
// i.e. 3 multiply-adds per output instead of S+V; A and Bv cost 3+3V per (n,c).
// One WAVE per node and iteration: lane = (o4, c4) owns orientations o4, o4+4, o4+8, o4+12 of the channel quad c4, so a store
// instruction covers four whole consecutive rows of the node -- 1 KB contiguous -- and a node is four of them; the node index is
// wave-uniform (its S + 3V inputs are scalar loads, those of the next node are fetched before this node's rows are computed) and
// nodes are dealt to the resident waves one at a time (round 2: a thread per (node, channel quad), 2 688 workgroups' worth of
// work on a 2 048-workgroup grid -- half the chip idled through a second, partial round: 2.3 TB/s of stores).
// The inputs of a node are fetched by ONE vector load: lane i < 8 reads scalar feature min(i, S-1), lane 8 + 3v + d reads component d of
// vector feature min(v, V-1) (clamped duplicates meet zero weights / are dropped), and the values are broadcast from that register with
// v_readlane when they are used.  (Scalar loads of the same 32 values came out as seven s_waitcnt-separated groups per node: seven
// serialised scalar-cache round trips per iteration.)  `scal` / `vec` stand in for each other when S / V is zero (any readable floats).
struct LiftLane { const float* base; long long stride; int off; bool valid; };   // valid: the lane holds a real input (not a clamped duplicate)
GRL_DEVINL LiftLane lift_lane(const float* scal, const float* vec, int S, int V) {
  const int i = threadIdx.x & 31;        // lanes 32..63 mirror 0..31
  LiftLane L;
  if (i < 8) {
    L.base = S > 0 ? scal : vec; L.stride = S > 0 ? S : 0; L.off = i < S ? i : (S > 0 ? S - 1 : 0);
    L.valid = i < S;
  } else {
    const int v = (i - 8) / 3, d = (i - 8) - 3 * v;
    L.base = V > 0 ? vec : scal; L.stride = V > 0 ? 3 * V : 0; L.off = V > 0 ? 3 * (v < V ? v : V - 1) + d : 0;
    L.valid = v < V;
  }
  return L;
}
GRL_DEVINL float lift_fetch(const LiftLane& L, int n) { return L.base[(long long)n * L.stride + L.off]; }
#define LIFT_S(val, k) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), (k)))
#define LIFT_V(val, v, d) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), 8 + 3 * (v) + (d)))

constexpr int LIFT_CHUNK = 16;   // iterations per input refill (8 KB of LDS)
// (blk, n_blk): this workgroup's index among the workgroups that share the node set -- blockIdx.x / gridDim.x for the one-type launch, the
// position inside the type's block range for the multi-type launch (lift_encode_fwd_multi_kernel)
GRL_DEVINL void lift_encode_fwd_body(const float* __restrict__ scal, const float* __restrict__ vec, const float* __restrict__ grid,
                                     const float* __restrict__ Wenc, st_t* __restrict__ x, int N, int S, int V, int blk, int n_blk,
                                     float* stage /* [LIFT_CHUNK * 4 * 32] LDS */) {
  const int KF = S + V;
  const int lane = threadIdx.x & 63, c4 = lane & 15, o4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = 4 * blk + wv, n_waves = n_blk * 4;
  // wa[.][k]: weight of scalar feature k (0 for k >= S), wb[.][v]: of vector feature v (0 for v >= V) -- every slot takes the same
  // multiply-adds, no branch (a zero weight adds an exact zero for finite inputs; a non-finite input poisons its node either way)
  float wa[4][KF_MAX], wb[4][KF_MAX], g[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KF_MAX; ++k) {
      wa[j][k] = k < S ? Wenc[(4 * c4 + j) * KF + k] : 0.f;
      wb[j][k] = k < V ? Wenc[(4 * c4 + j) * KF + S + k] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int d = 0; d < 3; ++d) g[i][d] = grid[(4 * i + o4) * 3 + d];
  // The inputs of the workgroup's next LIFT_CHUNK iterations are staged in LDS by one burst of loads, so the node loop itself has no
  // vector-memory load: a load there would be waited for with vmcnt(0) -- the loop-carried count is merged conservatively -- which
  // also waits for the four stores issued after it: a store drain per node.
  const LiftLane L = lift_lane(scal, vec, S, V);
  const int n_iter = (N + n_waves - 1) / n_waves;
  for (int it0 = 0; it0 < n_iter; it0 += LIFT_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < LIFT_CHUNK * 4 * 32 / 256; ++m) {
      const int idx = threadIdx.x + 256 * m, i = idx >> 7, w_ = (idx >> 5) & 3;
      const long long node = (long long)(4 * blk + w_) + (long long)(it0 + i) * n_waves;
      // (clamped, not skipped: a branch per load would serialise them.)  Slots past S / V hold clamped duplicates -- with S == 0 or V == 0
      // even node 0's value of the OTHER array: they are zeroed where the register goes to LDS (never right behind the load, finding 31b),
      // so a non-finite input of one node cannot reach another node through NaN * 0
      const float fetched = lift_fetch(L, node < N ? (int)node : N - 1);
      stage[idx] = L.valid ? fetched : 0.f;
    }
    __syncthreads();
    for (int i = 0; i < LIFT_CHUNK; ++i) {
      const long long n = (long long)wave + (long long)(it0 + i) * n_waves;
      if (n >= N) break;
      const float cur = stage[(i * 4 + wv) * 32 + (lane & 31)];
      float A[4] = {0.f, 0.f, 0.f, 0.f}, Bx[4] = {0.f, 0.f, 0.f, 0.f}, By[4] = {0.f, 0.f, 0.f, 0.f}, Bz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KF_MAX; ++k) {
        const float sv = LIFT_S(cur, k);
#pragma unroll
        for (int j = 0; j < 4; ++j) A[j] = fmaf(sv, wa[j][k], A[j]);
      }
#pragma unroll
      for (int k = 0; k < KF_MAX; ++k) {
        const float vx = LIFT_V(cur, k, 0), vy = LIFT_V(cur, k, 1), vz = LIFT_V(cur, k, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) { Bx[j] = fmaf(vx, wb[j][k], Bx[j]); By[j] = fmaf(vy, wb[j][k], By[j]); Bz[j] = fmaf(vz, wb[j][k], Bz[j]); }
      }
      st_t* out = x + (size_t)n * (O * C) + o4 * C + 4 * c4;
#pragma unroll
      for (int r_ = 0; r_ < 4; ++r_) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = fmaf(g[r_][2], Bz[j], fmaf(g[r_][1], By[j], fmaf(g[r_][0], Bx[j], A[j])));
        st4(out + 4 * r_ * C, make_float4(r[0], r[1], r[2], r[3]));
      }
    }
  }
}

__global__ __launch_bounds__(256) void lift_encode_fwd_kernel(const float* __restrict__ scal, const float* __restrict__ vec,
                                                              const float* __restrict__ grid, const float* __restrict__ Wenc,
                                                              st_t* __restrict__ x, int N, int S, int V) {
  __shared__ float stage[LIFT_CHUNK * 4 * 32];
  lift_encode_fwd_body(scal, vec, grid, Wenc, x, N, S, V, (int)blockIdx.x, (int)gridDim.x, stage);
}
// Every node type of a graph in ONE launch (round 4: the two lift launches of a HEPi pass -- object points, actuators -- were two ~11 us
// links of the step's serial chain at small minibatches): the types share the encoder weights and the feature widths; a workgroup belongs
// to the type whose block range holds it.  The backward likewise; its partial rows of all types stack into one slab (one fold, one dW).
constexpr int LIFT_MAX_TYPES = 4;
struct LiftMulti {
  const float* scal[LIFT_MAX_TYPES];
  const float* vec[LIFT_MAX_TYPES];
  st_t* x[LIFT_MAX_TYPES];          // forward: outputs; backward: the incoming gradients (read only)
  int N[LIFT_MAX_TYPES];
  int blk0[LIFT_MAX_TYPES + 1];     // first workgroup of every type
  int n_types;
};
__global__ __launch_bounds__(256) void lift_encode_fwd_multi_kernel(LiftMulti m, const float* __restrict__ grid, const float* __restrict__ Wenc,
                                                                    int S, int V) {
  __shared__ float stage[LIFT_CHUNK * 4 * 32];
  int t = 0;
  while (t + 1 < m.n_types && (int)blockIdx.x >= m.blk0[t + 1]) ++t;
  lift_encode_fwd_body(m.scal[t], m.vec[t], grid, Wenc, m.x[t], m.N[t], S, V, (int)blockIdx.x - m.blk0[t], m.blk0[t + 1] - m.blk0[t], stage);
}

// dW[c,k] = sum_{n,o} dx[n,o,c] feat[n,o,k]; partial[block][64*KF].  With D0[n,c] = sum_o dx and Dd[n,c] = sum_o grid[o,d] dx:
// dW[c,s] = sum_n scal[n,s] D0,  dW[c,S+v] = sum_n vec[n,v,:] . D[n,c,:]  -- 4 multiply-adds per dx element instead of S+V.
// Same wave-per-node layout as the forward: a node's dx is four 1 KB loads (those of the next node are in flight while this one is
// folded); a lane folds ITS four orientations into D and straight on into its own dW partial -- dW is linear in D, so the sum over
// the four orientation groups of lanes waits until the end of the launch (two shuffles per accumulator, once).
GRL_DEVINL void lift_encode_bwd_body(const float* __restrict__ scal, const float* __restrict__ vec, const float* __restrict__ grid,
                                     const st_t* __restrict__ dx, float* __restrict__ partial_row, int N, int S, int V, int blk, int n_blk,
                                     float (*red)[C * KF_MAX] /* [4][C * KF_MAX] LDS */) {
  const int KF = S + V;
  const int lane = threadIdx.x & 63, c4 = lane & 15, o4 = lane >> 4, wv = threadIdx.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blk * 256 + threadIdx.x) >> 6)), n_waves = n_blk * 4;
  float g[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int d = 0; d < 3; ++d) g[i][d] = grid[(4 * i + o4) * 3 + d];
  float dwa[4][KF_MAX], dwb[4][KF_MAX];   // scalar slots k / vector slots v (slots past S / V collect clamped duplicates and are dropped)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KF_MAX; ++k) dwa[j][k] = dwb[j][k] = 0.f;
  const LiftLane L = lift_lane(scal, vec, S, V);
  // LB nodes per iteration: their loads (four storage quads + one input dword each) are issued in one burst and kept RAW until used
  // (grl_common.h raw4_t), 8 KB in flight per wave.  The former form -- node n + 1 loaded while node n is computed, handed over in
  // registers at the loop end -- did not survive the compiler: the loads were sunk behind the back-edge to their first use, and in the
  // bf16 build each of the four waited for the one before it (270 us for the rope minibatch's 180 MB, round 5).
  // Slots past N re-read the iteration's first node with their input zeroed (no branch around a load).
#ifndef GRL_LIFT_LB
#define GRL_LIFT_LB (GRL_PREC ? 4 : 2)
#endif
  constexpr int LB = GRL_LIFT_LB;
  for (long long n0 = wave; n0 < N; n0 += (long long)n_waves * LB) {
    raw4_t dq[LB][4];
    float in[LB];
#pragma unroll
    for (int b = 0; b < LB; ++b) {
      const long long nb = n0 + (long long)b * n_waves;
      const int nn = nb < N ? (int)nb : (int)n0;
      in[b] = lift_fetch(L, nn);
#pragma unroll
      for (int i = 0; i < 4; ++i) dq[b][i] = ld4_raw(dx + (size_t)nn * (O * C) + (4 * i + o4) * C + 4 * c4);
    }
#pragma unroll
    for (int b = 0; b < LB; ++b) {
      const float cur = n0 + (long long)b * n_waves < N ? in[b] : 0.f;
      float d0[4] = {0.f, 0.f, 0.f, 0.f}, d1[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f}, d3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 dw = widen4(dq[b][i]);
        const float dv[4] = {dw.x, dw.y, dw.z, dw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d0[j] += dv[j];
          d1[j] = fmaf(g[i][0], dv[j], d1[j]);
          d2[j] = fmaf(g[i][1], dv[j], d2[j]);
          d3[j] = fmaf(g[i][2], dv[j], d3[j]);
        }
      }
#pragma unroll
      for (int k = 0; k < KF_MAX; ++k) {
        const float sv = LIFT_S(cur, k), vx = LIFT_V(cur, k, 0), vy = LIFT_V(cur, k, 1), vz = LIFT_V(cur, k, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dwa[j][k] = fmaf(sv, d0[j], dwa[j][k]);
          dwb[j][k] += vx * d1[j] + vy * d2[j] + vz * d3[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KF_MAX; ++k) {
      float v = k < S ? dwa[j][k] : dwb[j][k - S < 0 ? 0 : k - S];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (o4 == 0) red[wv][(4 * c4 + j) * KF_MAX + k] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < C * KF; i += 256) {
    const int c = i / KF, k = i - c * KF;
    const int s_ = c * KF_MAX + k;
    partial_row[i] = red[0][s_] + red[1][s_] + red[2][s_] + red[3][s_];
  }
}
__global__ __launch_bounds__(256) void lift_encode_bwd_kernel(const float* __restrict__ scal, const float* __restrict__ vec,
                                                              const float* __restrict__ grid, const st_t* __restrict__ dx,
                                                              float* __restrict__ partial, int N, int S, int V) {
  __shared__ float red[4][C * KF_MAX];
  lift_encode_bwd_body(scal, vec, grid, dx, partial + (size_t)blockIdx.x * C * (S + V), N, S, V, (int)blockIdx.x, (int)gridDim.x, red);
}
__global__ __launch_bounds__(256) void lift_encode_bwd_multi_kernel(LiftMulti m, const float* __restrict__ grid, float* __restrict__ partial,
                                                                    int S, int V) {
  __shared__ float red[4][C * KF_MAX];
  int t = 0;
  while (t + 1 < m.n_types && (int)blockIdx.x >= m.blk0[t + 1]) ++t;
  lift_encode_bwd_body(m.scal[t], m.vec[t], grid, m.x[t], partial + (size_t)blockIdx.x * C * (S + V), m.N[t], S, V,
                       (int)blockIdx.x - m.blk0[t], m.blk0[t + 1] - m.blk0[t], red);
}

// ------------------------------------------------------------------------------------------------ fiber conv
// HBM-bound: 2 (forward) / 3 (backward) passes over [N,16,64].  A workgroup streams batches of FB nodes through LDS: all four waves load
// the batch with 16-byte accesses (every byte fetched once per workgroup, the loads of batch b+1 are in flight while batch b is computed
// from LDS), then thread (c, q) -- channel c, orientation quad q -- reads the 16 orientation values of its channel from LDS (bank = c:
// conflict-free) against its register slice of fk.  Round 2's kernels had each of the four waves load the node's 16 rows itself, one
// dword per lane and one node at a time: ~20 KB of distinct bytes in flight per CU, 3.1 / 2.4 TB/s.
#ifndef GRL_FIBER_FB
#define GRL_FIBER_FB 4
#endif
#ifndef GRL_FIBER_NT
#define GRL_FIBER_NT 0   // bit 0: non-temporal loads, bit 1: non-temporal stores
#endif
constexpr int FB = GRL_FIBER_FB;                  // nodes per batch
constexpr int FB_E = FB * O * C;                  // elements per batch
template <int NT> struct FiberRegs { raw4_t r[FB_E / 4 / NT]; };   // four-element pieces per thread and batch (NT threads per workgroup), RAW:
// widened where they go to LDS -- widened at the load (bf16 build) every prefetch was waited for before the batch in LDS was computed (round 5)
// Full batches only (the main loops are free of guards and branches: a conditional store or load in there makes the compiler's vmcnt
// bookkeeping assume the worst, and the wait for the prefetched loads becomes a wait for every store of the iteration as well).
template <int NT> GRL_DEVINL void fiber_load(FiberRegs<NT>& R, const st_t* __restrict__ src, long long batch) {
#pragma unroll
  for (int i = 0; i < FB_E / 4 / NT; ++i) {
    const st_t* p = src + batch * FB_E + 4 * (threadIdx.x + NT * i);
#if GRL_PREC
    R.r[i] = ld4_raw(p);
#else
    R.r[i] = (GRL_FIBER_NT & 1) ? ld4_nt(p) : ld4(p);
#endif
  }
}
template <int NT> GRL_DEVINL void fiber_put(const FiberRegs<NT>& R, float* tile) {
#pragma unroll
  for (int i = 0; i < FB_E / 4 / NT; ++i) *reinterpret_cast<float4*>(tile + 4 * (threadIdx.x + NT * i)) = widen4(R.r[i]);
}
template <int NT> GRL_DEVINL void fiber_store(st_t* __restrict__ dst, long long batch, const float* tile) {
#pragma unroll
  for (int i = 0; i < FB_E / 4 / NT; ++i) {
    const int e = 4 * (threadIdx.x + NT * i);
    if (GRL_FIBER_NT & 2) st4_nt(dst + batch * FB_E + e, *reinterpret_cast<const float4*>(tile + e));
    else st4(dst + batch * FB_E + e, *reinterpret_cast<const float4*>(tile + e));
  }
}
// the last, partial batch of a launch (N % FB nodes; one workgroup, once): guarded element-wise copies, rows past N are zero in LDS
GRL_DEVINL void fiber_tail_in(const st_t* __restrict__ src, long long first, int N, float* tile) {
  for (int e = threadIdx.x; e < FB_E; e += blockDim.x) tile[e] = first * (O * C) + e < (long long)N * (O * C) ? ld1(src + first * (O * C) + e) : 0.f;
}
GRL_DEVINL void fiber_tail_out(st_t* __restrict__ dst, long long first, int N, const float* tile) {
  for (int e = threadIdx.x; e < FB_E; e += blockDim.x)
    if (first * (O * C) + e < (long long)N * (O * C)) st1(dst + first * (O * C) + e, tile[e]);
}

// x2 rows of the FB nodes of an LDS batch into the LDS tile `to` (thread (c, q): channel c, output orientations 4q..4q+3)
GRL_DEVINL void fiber_fwd_batch(const float* tin, float* to, const float (&k)[O][4], float b, int c, int q) {
#pragma unroll
  for (int i = 0; i < FB; ++i) {
    float acc[4] = {b, b, b, b};
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float v = tin[(i * O + o) * C + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += v * k[o][j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) to[(i * O + 4 * q + j) * C + c] = acc[j];
  }
}

// ``flag_dst`` (optional): one thread copies ``flag_src[0]`` there when the launch STARTS, i.e. when everything in front of it on the
// stream has finished -- the signal another lane's hipStreamWaitValue32 waits for (PolicyUpdater: "the first edge convolution is done"),
// without a 4-us copy launch of its own on the step's chain.  System scope: the store must not linger in this XCD's L2.
GRL_DEVINL void lane_signal(int* flag_dst, const int* flag_src) {
  if (flag_dst && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(flag_dst, flag_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void fiber_conv_fwd_kernel(const st_t* __restrict__ x1, const float* __restrict__ fk,
                                                             const float* __restrict__ bias, st_t* __restrict__ x2, int N,
                                                             int* flag_dst, const int* flag_src) {
  lane_signal(flag_dst, flag_src);
  __shared__ __attribute__((aligned(16))) float tin[FB_E];   // the next batch waits in registers
  __shared__ __attribute__((aligned(16))) float to[FB_E];    // x2 of the batch: leaves as 16-byte stores
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  float k[O][4];
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int j = 0; j < 4; ++j) k[o][j] = fk[(o * O + 4 * q + j) * C + c] * (1.f / O);
  const float b = bias[c];
  const long long nb = N / FB;               // full batches
  FiberRegs<256> R;
  long long batch = blockIdx.x;
  if (batch < nb) { fiber_load(R, x1, batch); fiber_put(R, tin); }
  __syncthreads();
  for (; batch < nb; batch += gridDim.x) {
    const long long next = batch + gridDim.x;
    if (next < nb) fiber_load(R, x1, next);   // (clamped instead of guarded -- a branch-free loop -- measured: no gain in the bf16 step,
                                              //  -0.6 % in the fp32 one, profiles/r05_ab_nodeops.txt call 5)
    fiber_fwd_batch(tin, to, k, b, c, q);
    __syncthreads();
    fiber_store<256>(x2, batch, to);
    if (next < nb) fiber_put(R, tin);
    __syncthreads();
  }
  if (N % FB != 0 && blockIdx.x == nb % gridDim.x) {
    fiber_tail_in(x1, nb * FB, N, tin);
    __syncthreads();
    fiber_fwd_batch(tin, to, k, b, c, q);
    __syncthreads();
    fiber_tail_out(x2, nb * FB, N, to);
  }
}

// partial[block] = [dfk 16*16*64 | dbias 64]
constexpr int FIBER_PARTIAL = O * O * C + C;
// Backward: four waves, thread (c, q) owns channel c and the orientation quad 4q..4q+3: 64 registers of fk slice, 64 of dfk accumulators;
// two workgroups per CU.  (An eight-wave form -- two orientations per thread, ~120 VGPRs, four waves per SIMD -- ran the same 120 us on
// its own and cost 1-2 % of the step on every workload: sixteen resident waves per CU starve the critic's backward next to it.)
constexpr int FBW = 4;                       // waves per workgroup
constexpr int FBR = O / FBW;                 // orientations per thread
// Two forms of the backward batch.  GRL_FIBER_BWD_PK = 1 (the plain-bf16 build, where the kernel is instruction-bound: 656 -> 404 us on the
// rope minibatch's large layer): packed pairs, the own quad re-read from LDS.  0 (the fp32 build, HBM-bound either way: 154 us for 728 MB):
// round 4's form -- with the packed form the kernel itself is 10 % faster and the REPLAYED step 1.7 % slower on two boxes
// (profiles/r05_ab_nodeops.txt: 324.8 / 325.1 with this form, 319.1 / 319.4 with the packed one), so it stays.
#ifndef GRL_FIBER_BWD_PK
#define GRL_FIBER_BWD_PK GRL_PREC
#endif
#if !GRL_FIBER_BWD_PK
GRL_DEVINL void fiber_bwd_batch(const float* tx, const float* td, float* to, const float (&kq)[FBR][O], float (&dk)[O][FBR], float& db,
                                int c, int q) {
#pragma unroll 1
  for (int i = 0; i < FB; ++i) {
    float dv[O];
#pragma unroll
    for (int o = 0; o < O; ++o) dv[o] = td[(i * O + o) * C + c];
    float acc[FBR];
#pragma unroll
    for (int j = 0; j < FBR; ++j) {
      acc[j] = 0.f;
#pragma unroll
      for (int p = 0; p < O; ++p) acc[j] += dv[p] * kq[j][p];
      db += dv[FBR * q + j];
    }
#pragma unroll
    for (int j = 0; j < FBR; ++j) to[(i * O + FBR * q + j) * C + c] = acc[j];
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float xv = tx[(i * O + o) * C + c];
#pragma unroll
      for (int j = 0; j < FBR; ++j) dk[o][j] += xv * dv[FBR * q + j];
    }
  }
}
#endif
#if GRL_FIBER_BWD_PK
// The thread's own quad of dx2 (orientations FBR q .. FBR q + 3) is read from LDS a second time: picked out of the register array dv[] by
// the (wave-uniform, but not to the compiler) index q it cost 120 v_cmp / v_cndmask per node beside 128 FMAs (round 5: the bf16 build of
// this kernel is instruction-bound).  Both products run as packed pairs over the quad (v_pk_fma_f32: 64 per node).
GRL_DEVINL void fiber_bwd_batch(const float* tx, const float* td, float* to, const v2f (&kq)[O][FBR / 2], v2f (&dk)[O][FBR / 2], float& db,
                                int c, int q) {
#pragma unroll 1
  for (int i = 0; i < FB; ++i) {
    v2f dq[FBR / 2], acc[FBR / 2];
#pragma unroll
    for (int h = 0; h < FBR / 2; ++h) {
      dq[h] = v2f{td[(i * O + FBR * q + 2 * h) * C + c], td[(i * O + FBR * q + 2 * h + 1) * C + c]};
      acc[h] = v2f{0.f, 0.f};
      db += dq[h][0] + dq[h][1];
    }
#pragma unroll
    for (int p = 0; p < O; ++p) {
      const v2f dv = splat2(td[(i * O + p) * C + c]);
#pragma unroll
      for (int h = 0; h < FBR / 2; ++h) acc[h] = fma2(dv, kq[p][h], acc[h]);
    }
#pragma unroll
    for (int h = 0; h < FBR / 2; ++h) {
      to[(i * O + FBR * q + 2 * h) * C + c] = acc[h][0];
      to[(i * O + FBR * q + 2 * h + 1) * C + c] = acc[h][1];
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const v2f xv = splat2(tx[(i * O + o) * C + c]);
#pragma unroll
      for (int h = 0; h < FBR / 2; ++h) dk[o][h] = fma2(xv, dq[h], dk[o][h]);
    }
  }
}
#endif
__global__ __launch_bounds__(64 * FBW, 2) void fiber_conv_bwd_kernel(const st_t* __restrict__ x1, const float* __restrict__ fk,
                                                                     const st_t* __restrict__ dx2, st_t* __restrict__ dx1,
                                                                     float* __restrict__ partial, int N) {
  constexpr int NT = 64 * FBW;
  __shared__ __attribute__((aligned(16))) float tx[FB_E];   // single-buffered: the next batch waits in registers
  __shared__ __attribute__((aligned(16))) float td[FB_E];
  __shared__ __attribute__((aligned(16))) float to[FB_E];   // dx1 of the batch: leaves as 16-byte stores
  __shared__ float red[FBW * C];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
#if GRL_FIBER_BWD_PK
  v2f kq[O][FBR / 2];   // [p][j]: fk[o = FBR q + j][p][c] / 16   (rows this thread back-propagates to), pairs over j
  v2f dk[O][FBR / 2];   // d fk[o][p = FBR q + j][c], pairs over j
#pragma unroll
  for (int p = 0; p < O; ++p)
#pragma unroll
    for (int j = 0; j < FBR; ++j) kq[p][j >> 1][j & 1] = fk[((FBR * q + j) * O + p) * C + c] * (1.f / O);
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int h = 0; h < FBR / 2; ++h) dk[o][h] = v2f{0.f, 0.f};
  auto dk_at = [&](int o, int j) { return dk[o][j >> 1][j & 1]; };
#else
  float kq[FBR][O];   // fk[o = FBR q + j][p][c] / 16   (rows this thread back-propagates to)
  float dk[O][FBR];   // d fk[o][p = FBR q + j][c]
#pragma unroll
  for (int j = 0; j < FBR; ++j)
#pragma unroll
    for (int p = 0; p < O; ++p) kq[j][p] = fk[((FBR * q + j) * O + p) * C + c] * (1.f / O);
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int j = 0; j < FBR; ++j) dk[o][j] = 0.f;
  auto dk_at = [&](int o, int j) { return dk[o][j]; };
#endif
  float db = 0.f;
  const long long nb = N / FB;               // full batches
  FiberRegs<NT> RX, RD;
  long long batch = blockIdx.x;
  if (batch < nb) { fiber_load(RX, x1, batch); fiber_load(RD, dx2, batch); fiber_put(RX, tx); fiber_put(RD, td); }
  __syncthreads();
  for (; batch < nb; batch += gridDim.x) {
    const long long next = batch + gridDim.x;
    if (next < nb) { fiber_load(RX, x1, next); fiber_load(RD, dx2, next); }
    fiber_bwd_batch(tx, td, to, kq, dk, db, c, q);
    __syncthreads();
    fiber_store<NT>(dx1, batch, to);
    if (next < nb) { fiber_put(RX, tx); fiber_put(RD, td); }
    __syncthreads();
  }
  if (N % FB != 0 && blockIdx.x == nb % gridDim.x) {
    fiber_tail_in(x1, nb * FB, N, tx);
    fiber_tail_in(dx2, nb * FB, N, td);
    __syncthreads();
    fiber_bwd_batch(tx, td, to, kq, dk, db, c, q);
    __syncthreads();
    fiber_tail_out(dx1, nb * FB, N, to);
  }
  float* out = partial + (size_t)blockIdx.x * FIBER_PARTIAL;
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int j = 0; j < FBR; ++j) out[(o * O + FBR * q + j) * C + c] = dk_at(o, j) * (1.f / O);
  red[q * C + c] = db;
  __syncthreads();
  if (q == 0) {
    float t = red[c];
#pragma unroll
    for (int g = 1; g < FBW; ++g) t += red[g * C + c];
    out[O * O * C + c] = t;
  }
}

// ------------------------------------------------------------------------------------------------ attention aggregation
// PyG AttentionalAggregation (gate_nn = Linear + ReLU, no nn) as FiberBundleConv uses it (conv.py:21-26,58-61,138-139), vmapped over the
// orientations: per destination node d, orientation o and channel c
//     alpha_e = softmax over the in-edges e of d of gate[e,o,c]      (exp(g - max) / (sum + 1e-16): PyG 2.5.2 utils.softmax [upstream])
//     x1[d,o,c] = sum_e alpha_e msg[e,o,c]
// gate / msg rows are in destination-sorted edge order, so a destination's edges are the contiguous rows rowptr[d] .. rowptr[d+1].
// One wave per (d, o) row, lane = channel; the few edges of a destination are walked twice (max, then sums).  HBM-bound streaming.
// Backward (closed form): d msg_e = alpha_e dx1,  d gate_e = alpha_e dx1 (msg_e - x1).
__global__ __launch_bounds__(256) void softmax_agg_fwd_kernel(const float* __restrict__ gate, const st_t* __restrict__ msg,
                                                              const int* __restrict__ rowptr, int n_dst, st_t* __restrict__ x1) {
  const int c = threadIdx.x & 63;
  const long long n_rows = (long long)n_dst * O;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (long long)gridDim.x * 4) {
    const int d = (int)(row >> 4), o = (int)(row & 15);
    const int e0 = rowptr[d], e1 = rowptr[d + 1];
    float mx = -3.0e38f;
    for (int e = e0; e < e1; ++e) mx = fmaxf(mx, gate[((size_t)e * O + o) * C + c]);
    float den = 0.f, num = 0.f;
    for (int e = e0; e < e1; ++e) {
      const size_t i = ((size_t)e * O + o) * C + c;
      const float w = expf(gate[i] - mx);
      den += w;
      num = fmaf(w, ld1(msg + i), num);
    }
    st1(x1 + (size_t)row * C + c, e1 > e0 ? num / (den + 1e-16f) : 0.f);
  }
}
__global__ __launch_bounds__(256) void softmax_agg_bwd_kernel(const float* __restrict__ gate, const st_t* __restrict__ msg,
                                                              const st_t* __restrict__ x1, const st_t* __restrict__ dx1,
                                                              const int* __restrict__ rowptr, int n_dst, float* __restrict__ dgate,
                                                              st_t* __restrict__ dmsg) {
  const int c = threadIdx.x & 63;
  const long long n_rows = (long long)n_dst * O;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (long long)gridDim.x * 4) {
    const int d = (int)(row >> 4), o = (int)(row & 15);
    const int e0 = rowptr[d], e1 = rowptr[d + 1];
    float mx = -3.0e38f, den = 0.f;
    for (int e = e0; e < e1; ++e) mx = fmaxf(mx, gate[((size_t)e * O + o) * C + c]);
    for (int e = e0; e < e1; ++e) den += expf(gate[((size_t)e * O + o) * C + c] - mx);
    const float inv = 1.f / (den + 1e-16f), g = ld1(dx1 + (size_t)row * C + c), xo = ld1(x1 + (size_t)row * C + c);
    for (int e = e0; e < e1; ++e) {
      const size_t i = ((size_t)e * O + o) * C + c;
      const float a = expf(gate[i] - mx) * inv;
      st1(dmsg + i, a * g);
      dgate[i] = a * g * (ld1(msg + i) - xo);
    }
  }
}

// ------------------------------------------------------------------------------------------------ fiber kernel basis
// Phi = GELU(W2 GELU(W1 poly + b1) + b2) over the 256 (orientation, orientation) pairs (reference hepi.py:109-123,157;
// ponita.py:246-268: fiber_basis_fn on the degree-3 polynomial of o_i . o_j) and the fiber kernels fk_i = Phi Wf_i^T of up to
// FB_MAXC convolutions (conv.py:62,88): parameter-only, 256 rows -- one launch each way instead of ~30 rocBLAS / elementwise
// launches.  A workgroup owns 4 rows, thread = (row, column); weight rows are staged in LDS ([64][65], conflict-free by row and
// by column).  The backward leaves one partial row per workgroup: [dWf_0 .. dWf_{n-1} (4096 each) | dW2 4096 | db2 64 | dW1 192 | db1 64].
constexpr int FB_ROWS = 256, FB_RPB = 4, FB_MAXC = 4, FB_P = 3;
// 256 threads x 4 quads: every load is issued before the first LDS store (a scalar loop of 16 dependent loads per matrix made
// the staging, not the arithmetic, these two launches' run time)
GRL_DEVINL void fb_stage(float* dst /*[64][65]*/, const float* __restrict__ src /*[64][64]*/) {
  float4 q[4];
  const bool vec = (reinterpret_cast<size_t>(src) & 15) == 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = 4 * (threadIdx.x + 256 * u);
    q[u] = vec ? *reinterpret_cast<const float4*>(src + i) : make_float4(src[i], src[i + 1], src[i + 2], src[i + 3]);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = 4 * (threadIdx.x + 256 * u);
    float* d = dst + (i >> 6) * 65 + (i & 63);
    d[0] = q[u].x; d[1] = q[u].y; d[2] = q[u].z; d[3] = q[u].w;
  }
}
// LDS of the two bodies (static, so that they can ride in merged launches beside other roles: grl_step_head, grl_lift_fiber_basis_bwd):
// W2 | ONE Wf matrix at a time (round 5 staged all n of them: 17 KB each, dynamic LDS) | the row buffers
constexpr int FB_SMEM_FLOATS = 2 * 64 * 65 + FB_MAXC * FB_RPB * 64 + 4 * FB_RPB * 64;
struct FbFwd { const float *poly, *W1, *b1, *W2, *b2; FiberWfs wf; float* saved; FiberFks fk; };   // wf.n == 0: nothing to do
struct FbBwd { const float *poly, *W2; FiberWfs wf; const float* saved; FiberDfks dfk; float* partial; int partial_ld; };
GRL_DEVINL void fiber_basis_fwd_body(const FbFwd& A, int blk, float* fb_smem) {
  const FiberWfs& wf = A.wf;
  float* W2s = fb_smem;                    // [64][65]
  float* Wfs = W2s + 64 * 65;              // [64][65]: the convolution in progress
  float* hs = Wfs + 64 * 65;               // [2][FB_RPB][64]  h1 | h2
  fb_stage(W2s, A.W2);
  fb_stage(Wfs, wf.w[0]);
  const int rl = threadIdx.x >> 6, c = threadIdx.x & 63, r = blk * FB_RPB + rl;
  float z1 = A.b1[c];
#pragma unroll
  for (int k = 0; k < FB_P; ++k) z1 += A.W1[c * FB_P + k] * A.poly[r * FB_P + k];
  const float h1 = gelu_exact_f(z1);
  hs[rl * 64 + c] = h1;
  __syncthreads();
  float z2 = A.b2[c];
#pragma unroll 16
  for (int k = 0; k < 64; ++k) z2 += W2s[c * 65 + k] * hs[rl * 64 + k];
  const float h2 = gelu_exact_f(z2);
  hs[(FB_RPB + rl) * 64 + c] = h2;
  // saved for the backward: [z1 | h1 | z2 | h2] each [256][64]
  A.saved[(0 * FB_ROWS + r) * 64 + c] = z1;
  A.saved[(1 * FB_ROWS + r) * 64 + c] = h1;
  A.saved[(2 * FB_ROWS + r) * 64 + c] = z2;
  A.saved[(3 * FB_ROWS + r) * 64 + c] = h2;
  __syncthreads();
  for (int i = 0; i < wf.n; ++i) {
    if (i > 0) {
      __syncthreads();          // every thread has finished with the previous matrix
      fb_stage(Wfs, wf.w[i]);
      __syncthreads();
    }
    const float* w = Wfs + c * 65;
    float acc = 0.f;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) acc += w[k] * hs[(FB_RPB + rl) * 64 + k];
    A.fk.p[i][r * 64 + c] = acc;
  }
}
__global__ __launch_bounds__(256) void fiber_basis_fwd_kernel(FbFwd A) {
  __shared__ float fb_smem[FB_SMEM_FLOATS];
  fiber_basis_fwd_body(A, (int)blockIdx.x, fb_smem);
}

GRL_DEVINL void fiber_basis_bwd_body(const FbBwd& A, int blk, float* fb_smem) {
  const FiberWfs& wf = A.wf;
  float* W2s = fb_smem;                          // [64][65]
  float* Wfs = W2s + 64 * 65;                    // [64][65]: the convolution in progress
  float* dfs = Wfs + 64 * 65;                    // [n][FB_RPB][64]   dfk rows
  float* h2s = dfs + FB_MAXC * FB_RPB * 64;      // [FB_RPB][64]
  float* h1s = h2s + FB_RPB * 64;
  float* dz2s = h1s + FB_RPB * 64;
  float* dz1s = dz2s + FB_RPB * 64;
  const float* poly = A.poly;
  const float* saved = A.saved;
  fb_stage(W2s, A.W2);
  fb_stage(Wfs, wf.w[0]);
  const int rl = threadIdx.x >> 6, c = threadIdx.x & 63, r = blk * FB_RPB + rl;
  for (int i = 0; i < wf.n; ++i) dfs[(i * FB_RPB + rl) * 64 + c] = A.dfk.p[i] ? A.dfk.p[i][r * 64 + c] : 0.f;
  const float z1 = saved[(0 * FB_ROWS + r) * 64 + c], h1 = saved[(1 * FB_ROWS + r) * 64 + c];
  const float z2 = saved[(2 * FB_ROWS + r) * 64 + c], h2 = saved[(3 * FB_ROWS + r) * 64 + c];
  h2s[rl * 64 + c] = h2;
  h1s[rl * 64 + c] = h1;
  __syncthreads();
  // dPhi[r][k = c] = sum_i sum_c' dfk_i[r][c'] Wf_i[c'][k]
  float dphi = 0.f;
  for (int i = 0; i < wf.n; ++i) {
    if (i > 0) {
      __syncthreads();
      fb_stage(Wfs, wf.w[i]);
      __syncthreads();
    }
    const float* w = Wfs + c;
    const float* d = dfs + (i * FB_RPB + rl) * 64;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) dphi += d[k] * w[k * 65];
  }
  const float dz2 = dphi * gelu_exact_grad_f(z2);
  dz2s[rl * 64 + c] = dz2;
  __syncthreads();
  float dh1 = 0.f;
#pragma unroll 16
  for (int k = 0; k < 64; ++k) dh1 += dz2s[rl * 64 + k] * W2s[k * 65 + c];
  const float dz1 = dh1 * gelu_exact_grad_f(z1);
  dz1s[rl * 64 + c] = dz1;
  __syncthreads();
  // ---- this workgroup's partial row (sums over its FB_RPB rows)
  float* out = A.partial + (size_t)blk * A.partial_ld;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int cc = e >> 6, k = e & 63;
    for (int i = 0; i < wf.n; ++i) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < FB_RPB; ++q) t += dfs[(i * FB_RPB + q) * 64 + cc] * h2s[q * 64 + k];
      out[i * 4096 + e] = t;
    }
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < FB_RPB; ++q) t += dz2s[q * 64 + cc] * h1s[q * 64 + k];
    out[wf.n * 4096 + e] = t;
  }
  float* ob2 = out + (wf.n + 1) * 4096, *oW1 = ob2 + 64, *ob1 = oW1 + 64 * FB_P;
  if (threadIdx.x < 64) {
    float t2 = 0.f, t1 = 0.f, tw[FB_P] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < FB_RPB; ++q) {
      const int rr = blk * FB_RPB + q;
      t2 += dz2s[q * 64 + c];
      t1 += dz1s[q * 64 + c];
#pragma unroll
      for (int k = 0; k < FB_P; ++k) tw[k] += dz1s[q * 64 + c] * poly[rr * FB_P + k];
    }
    ob2[c] = t2;
    ob1[c] = t1;
#pragma unroll
    for (int k = 0; k < FB_P; ++k) oW1[c * FB_P + k] = tw[k];
  }
}
__global__ __launch_bounds__(256) void fiber_basis_bwd_kernel(FbBwd A) {
  __shared__ float fb_smem[FB_SMEM_FLOATS];
  fiber_basis_bwd_body(A, (int)blockIdx.x, fb_smem);
}

// ---- merged launches of the recorded step (round 6): roles that do not depend on each other share ONE launch, told apart by block range.
// Head of the actor's lane: node features (+ the step count) | fiber basis + fiber kernels | weight images -- three ~5 us launches of the
// step's serial chain (build_features -> lift needs only the first; the other two are parameter-only) become one.
__global__ __launch_bounds__(256) void step_head_kernel(FeatDescs feat, int n_desc, int nfx, int* __restrict__ bump, FbFwd fb, WimgJobs jobs) {
  __shared__ float fb_smem[FB_SMEM_FLOATS];
  const int nfb = fb.wf.n > 0 ? FB_ROWS / FB_RPB : 0, nwi = WIMG_PARTS * jobs.n;
  int b = (int)blockIdx.x;
  if (b < nfb) { fiber_basis_fwd_body(fb, b, fb_smem); return; }   // (the long roles first: they are the launch's critical path)
  b -= nfb;
  if (b < nwi) { weight_images_body(jobs, b / WIMG_PARTS, b % WIMG_PARTS); return; }
  b -= nwi;
  build_features_body(feat, bump, b % nfx, nfx, b / nfx, b == 0);
}
// End of the actor's backward: the lift's weight gradient (all node types) | the fiber basis backward -- both only feed the tail's fold.
__global__ __launch_bounds__(256) void lift_fiber_basis_bwd_kernel(LiftMulti m, const float* __restrict__ grid, float* __restrict__ partial,
                                                                   int S, int V, FbBwd fb) {
  __shared__ float smem[FB_SMEM_FLOATS];   // (the lift's [4][C * KF_MAX] reduction buffer aliases its head)
  static_assert(4 * C * KF_MAX <= FB_SMEM_FLOATS, "LDS of the lift backward must fit the shared block");
  const int nfb = fb.wf.n > 0 ? FB_ROWS / FB_RPB : 0;
  int b = (int)blockIdx.x;
  if (b < nfb) { fiber_basis_bwd_body(fb, b, smem); return; }
  b -= nfb;
  int t = 0;
  while (t + 1 < m.n_types && b >= m.blk0[t + 1]) ++t;
  lift_encode_bwd_body(m.scal[t], m.vec[t], grid, m.x[t], partial + (size_t)b * C * (S + V), m.N[t], S, V, b - m.blk0[t],
                       m.blk0[t + 1] - m.blk0[t], reinterpret_cast<float (*)[C * KF_MAX]>(smem));
}

// ------------------------------------------------------------------------------------------------ slab reduce
// out[j] += sum_w partial[w][j], bitwise reproducible: a workgroup owns 64 columns; its 8 waves sum interleaved row groups
// (wave g: rows g, g+8, ..., four independent running sums each, combined in a fixed order) and the 8 wave sums are folded
// through LDS in wave order.  No atomics, so the result does not depend on scheduling.
#ifndef GRL_RED_WAVES
#define GRL_RED_WAVES 2   // waves per fold workgroup.  Round 6 A/B (profiles/r06_ab_fold_waves.txt): 8 -> 4 -> 2 waves: 0.320 -> 0.304 -> 0.299 ms at 32 frames, 0.638 -> 0.618 -> 0.614 at 512, 3.113 -> 3.08 at 4096 -- short-lived 512-thread workgroups were the fold's cost, not its bytes
#endif
constexpr int RED_WAVES = GRL_RED_WAVES;
// RED_DEPTH independent loads in flight per wave: the slabs are row-strided, so a wave's walk down its rows is a chain of
// memory latencies -- with 4 in flight the longest columns (2048 rows: gradients fed by two convolutions) took 64 round trips.
constexpr int RED_DEPTH = 16;
GRL_DEVINL float column_sum(const float* __restrict__ src /*column base*/, size_t ld, int n_rows, int wave) {
  if (n_rows <= 0) return 0.f;   // (the clamped remainder loads below would index row -1)
  float a[RED_DEPTH];
#pragma unroll
  for (int u = 0; u < RED_DEPTH; ++u) a[u] = 0.f;
  int w = wave;
  for (; w + (RED_DEPTH - 1) * RED_WAVES < n_rows; w += RED_DEPTH * RED_WAVES) {
    float v[RED_DEPTH];
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) v[u] = src[(size_t)(w + u * RED_WAVES) * ld];
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) a[u] += v[u];
  }
  {  // remainder: all of its (< RED_DEPTH) loads in flight together (clamped rows, dropped at the add: see column_sum4)
    float v[RED_DEPTH];
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) {
      const int row = w + u * RED_WAVES;
      v[u] = src[(size_t)(row < n_rows ? row : n_rows - 1) * ld];
    }
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) a[u] += w + u * RED_WAVES < n_rows ? v[u] : 0.f;   // (a select, not a 0/1 factor: NaN * 0 = NaN)
  }
#pragma unroll
  for (int st = RED_DEPTH / 2; st > 0; st >>= 1)
#pragma unroll
    for (int u = 0; u < st; ++u) a[u] += a[u + st];
  return a[0];
}
GRL_DEVINL void fold_and_add(float v, float* __restrict__ dst, bool active, float (*red)[64], bool overwrite = false) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  red[wave][lane] = v;
  __syncthreads();
  if (wave == 0 && active) {
    float t = red[0][lane];
#pragma unroll
    for (int g = 1; g < RED_WAVES; ++g) t += red[g][lane];
    *dst = overwrite ? t : *dst + t;
  }
}
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                         int n_rows, int n) {
  __shared__ float red[RED_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bool active = j < n;
  const float v = active ? column_sum(partial + j, (size_t)n, n_rows, wave) : 0.f;
  fold_and_add(v, out + j, active, red);
}

// segmented variant: up to 8 column ranges of the partial rows are folded into 8 different destinations in ONE launch, so a
// backward kernel's weight gradients go straight into the (flat) parameter-gradient buffer
struct ReduceSegs {
  float* dst[8];
  int start[8];
  int len[8];
  int n_seg;
  int overwrite_mask;   // bit i set: dst[i] = sum (destination not read, need not be initialised) instead of dst[i] += sum
};
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_partials_seg_kernel(const float* __restrict__ partial, ReduceSegs segs,
                                                                             int n_rows, int ld) {
  __shared__ float red[RED_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = blockIdx.y;
  const int j = blockIdx.x * 64 + lane;
  if (blockIdx.x * 64 >= segs.len[seg]) return;   // whole workgroup outside this (shorter) segment
  const bool active = j < segs.len[seg];
  const float v = active ? column_sum(partial + segs.start[seg] + j, (size_t)ld, n_rows, wave) : 0.f;
  fold_and_add(v, segs.dst[seg] + j, active, red, (segs.overwrite_mask >> seg) & 1);
}

// every leaf-gradient fold of a backward pass in ONE launch: up to RED_MULTI_MAX source slabs, grouped by destination (several
// slabs may feed the same gradient, e.g. the basis MLP shared by all edge convolutions: one workgroup sums them in order)
constexpr int RED_MULTI_MAX = 64;
struct ReduceMulti {
  const float* partial[RED_MULTI_MAX];                                 // sources, grouped by destination
  int n_rows[RED_MULTI_MAX], ld[RED_MULTI_MAX], start[RED_MULTI_MAX];
  float* dst[RED_MULTI_MAX];                                           // destinations
  int len[RED_MULTI_MAX], first[RED_MULTI_MAX], count[RED_MULTI_MAX];
  int blk0[RED_MULTI_MAX + 1];                                         // first workgroup of destination d (64 columns per workgroup): a flat grid
  int n_dst;
  int overwrite;                                                       // 1: dst = sum (destination not read) instead of dst += sum
  unsigned char vec[RED_MULTI_MAX];                                    // destination group eligible for the float4 path
};
// The step's tail in ONE launch (one rank, no gradient clipping): while a workgroup still holds the freshly summed gradient entries it
// applies Adam to the parameters they belong to (the optimizer launch and its read of the gradient disappear; the gradient is stored as
// well: .grad stays inspectable), and one EXTRA workgroup folds the fused loss kernel's slots, evaluates the reported values and advances
// the optimizer's step count for the next step -- three more launches of the lane's tail gone (fold + Adam + report + count were
// 32 + 6 + 6 + 6 us at 512 frames).  Arithmetic of the update: adam_dev_kernel's (train_ops.hip), bit for bit.
struct FoldTail {
  long long d_param, d_m, d_v;     // element offsets from a gradient entry to its parameter / first moment / second moment
  const float* lr_dev;
  const int* step_dev;             // the count of the step in progress (advanced earlier on the lane: grl_build_features_bump)
  float b1, b2, eps;
  int adam;                        // 1: apply the update
  // report workgroup (blockIdx.x == number of fold workgroups)
  int report;
  const double* slots;
  int n_slot_blocks;
  double* sums;
  unsigned int* maxes;
  float ent_coef;
  float* out14;
  int* flag_dst;                   // optional lane signal written when the launch starts (lane_signal)
  const int* flag_src;
  float2* pairs_region;            // report == 2 (data parallel): the extra workgroup folds the slots into THIS rank's record of (hi, lo)
  int pairs_rank, pairs_world;     // float pairs (grl_report.h trpl_write_record_pairs) instead of evaluating the reported values
};
GRL_DEVINL void adam_apply(float g, float* __restrict__ gp, const FoldTail& t, float lr, float bc1, float bc2_sqrt) {
  float* p = gp + t.d_param;
  float* m = gp + t.d_m;
  float* v = gp + t.d_v;
  const float mi = t.b1 * *m + (1.f - t.b1) * g;
  const float vi = t.b2 * *v + (1.f - t.b2) * g * g;
  *m = mi;
  *v = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + t.eps;
  *p -= (lr / bc1) * (mi / denom);
}
// float4 variant: a wave covers 4 rows x 64 columns per load (16 lanes x float4 per row, lane >> 4 picks the row), so a
// workgroup still owns only 64 columns -- tall thin slabs (1024-2048 rows x 4096 columns) need that many workgroups: with 256
// columns per workgroup 16 CUs pulled the whole slab at ~95 GB/s each (tools/ubench/reduce_bench.py).
template <int RED_DEPTH>
GRL_DEVINL float4 column_sum4_d(const float* __restrict__ src, size_t ld, int n_rows, int row0) {
  constexpr int STEP = 4 * RED_WAVES;
  if (n_rows <= 0) return make_float4(0.f, 0.f, 0.f, 0.f);
  float4 a[RED_DEPTH];
#pragma unroll
  for (int u = 0; u < RED_DEPTH; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int w = row0;
  for (; w + (RED_DEPTH - 1) * STEP < n_rows; w += RED_DEPTH * STEP) {
    float4 v[RED_DEPTH];
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(w + u * STEP) * ld);
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) a[u] = f4_add(a[u], v[u]);
  }
  if (w < n_rows) {   // (one branch around the whole batch: slabs of exactly RED_DEPTH * STEP rows have no remainder)
     // remainder: all of its (< RED_DEPTH) loads in flight together -- rows past the end are loaded CLAMPED and dropped at the add: a guard
     // around each load is a branch per load with a wait at each join (finding 31a)
    float4 v[RED_DEPTH];
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) {
      const int row = w + u * STEP;
      v[u] = *reinterpret_cast<const float4*>(src + (size_t)(row < n_rows ? row : n_rows - 1) * ld);
    }
#pragma unroll
    for (int u = 0; u < RED_DEPTH; ++u) {
      const bool m = w + u * STEP < n_rows;   // (selects, not 0/1 factors: a non-finite clamped row must not turn into NaN * 0)
      a[u] = make_float4(a[u].x + (m ? v[u].x : 0.f), a[u].y + (m ? v[u].y : 0.f), a[u].z + (m ? v[u].z : 0.f), a[u].w + (m ? v[u].w : 0.f));
    }
  }
#pragma unroll
  for (int st = RED_DEPTH / 2; st > 0; st >>= 1)
#pragma unroll
    for (int u = 0; u < st; ++u) a[u] = f4_add(a[u], a[u + st]);
  return a[0];
}
// Loads in flight per lane sized to the slab (workgroup-uniform): the slabs of the 256-workgroup MFMA launches have exactly 256 rows = 8 loads
// of 32 rows -- with the fixed depth of 16 half of every lane's loads were clamped duplicates of the last row (round 3: 87 MB of slabs
// read as 174 MB of requests)
GRL_DEVINL float4 column_sum4(const float* __restrict__ src, size_t ld, int n_rows, int row0) {
  constexpr int STEP = 4 * RED_WAVES;
  if (n_rows <= 2 * STEP) return column_sum4_d<2>(src, ld, n_rows, row0);
  if (n_rows <= 4 * STEP) return column_sum4_d<4>(src, ld, n_rows, row0);
  if (n_rows <= 8 * STEP) return column_sum4_d<8>(src, ld, n_rows, row0);
  return column_sum4_d<16>(src, ld, n_rows, row0);
}
GRL_DEVINL float4 f4_shfl_xor(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64));
}
template <bool TAIL>
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_partials_multi_kernel(ReduceMulti m, FoldTail tail) {
  __shared__ float4 red[RED_WAVES][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lr = 0.f, bc1 = 1.f, bc2_sqrt = 1.f;
  if (TAIL) {
    lane_signal(tail.flag_dst, tail.flag_src);
    if (tail.report && (int)blockIdx.x == m.blk0[m.n_dst]) {   // the extra workgroup: reported values (its first 64 threads)
      __shared__ double sh[16], part[64 * RED_WAVES];
      if (tail.report == 2) {
        trpl_fold_columns<64 * RED_WAVES>(tail.slots, tail.n_slot_blocks, sh, part);
        trpl_write_record_pairs<64 * RED_WAVES>(sh, tail.pairs_region, tail.pairs_rank, tail.pairs_world);
      } else {
        trpl_report_body<64 * RED_WAVES>(tail.slots, tail.n_slot_blocks, tail.sums, tail.maxes, tail.ent_coef, tail.out14, sh, part);
      }
      return;
    }
    if (tail.adam) {
      const float t_ = (float)tail.step_dev[0];
      lr = tail.lr_dev[0];
      bc1 = 1.f - powf(tail.b1, t_);
      bc2_sqrt = sqrtf(1.f - powf(tail.b2, t_));
    }
  }
  // flat grid: exactly one workgroup per 64 columns of every destination (round 3 launched max_len / 64 x n_dst workgroups, ~8 of 10 of
  // which found themselves outside their destination and left at once -- ten thousand empty 512-thread workgroups per step)
  int d = 0;
  while (d + 1 < m.n_dst && (int)blockIdx.x >= m.blk0[d + 1]) ++d;
  const int bx = (int)blockIdx.x - m.blk0[d];
  if (m.vec[d]) {   // every source slab and the destination 16-byte aligned, lengths multiples of 4
    const int j = bx * 64 + 4 * (lane & 15);
    const bool active = j < m.len[d];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active)
      for (int q = m.first[d]; q < m.first[d] + m.count[d]; ++q)
        v = f4_add(v, column_sum4(m.partial[q] + m.start[q] + j, (size_t)m.ld[q], m.n_rows[q], 4 * wave + (lane >> 4)));
    v = f4_add(v, f4_shfl_xor(v, 16));
    v = f4_add(v, f4_shfl_xor(v, 32));
    if (lane < 16) red[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && lane < 16 && active) {
      float4 t = red[0][lane];
#pragma unroll
      for (int g = 1; g < RED_WAVES; ++g) t = f4_add(t, red[g][lane]);
      float4* dp = reinterpret_cast<float4*>(m.dst[d] + j);
      const float4 gq = m.overwrite ? t : f4_add(*dp, t);
      *dp = gq;
      if (TAIL && tail.adam) {
        float* gp = m.dst[d] + j;
        adam_apply(gq.x, gp, tail, lr, bc1, bc2_sqrt); adam_apply(gq.y, gp + 1, tail, lr, bc1, bc2_sqrt);
        adam_apply(gq.z, gp + 2, tail, lr, bc1, bc2_sqrt); adam_apply(gq.w, gp + 3, tail, lr, bc1, bc2_sqrt);
      }
    }
    return;
  }
  const int j = bx * 64 + lane;
  const bool active = j < m.len[d];
  float v = 0.f;
  if (active)
    for (int q = m.first[d]; q < m.first[d] + m.count[d]; ++q)
      v += column_sum(m.partial[q] + m.start[q] + j, (size_t)m.ld[q], m.n_rows[q], wave);
  __shared__ float reds[RED_WAVES][64];
  fold_and_add(v, m.dst[d] + j, active, reds, m.overwrite != 0);
  if (TAIL && tail.adam && wave == 0 && active) adam_apply(m.dst[d][j], m.dst[d] + j, tail, lr, bc1, bc2_sqrt);   // (this thread wrote the entry)
}

int cap_blocks(long long work, int per_block, int cap) {
  long long b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  return (int)(b < cap ? b : cap);
}
// one node per wave and iteration, >= 4 nodes per wave: the prologue (weights, grid, staged inputs) is paid per workgroup -- one node per
// wave on the step's small launches (a few thousand actuator nodes) took 17 instead of 11 us
int lift_blocks(int n_nodes) { return cap_blocks(n_nodes, 16, 1024); }

}  // namespace

extern "C" {

#if !GRL_PREC   // shape queries: shared by both precision builds
int grl_fiber_partial_size() { return FIBER_PARTIAL; }
// two workgroups per CU (register budget), all resident; small graphs: at least four batches per workgroup -- every workgroup costs a
// 66 KB partial row that the fold has to read back
int grl_fiber_bwd_blocks(int n_nodes) { const int b = cap_blocks(n_nodes, 4 * FB, 512); return b < 256 ? cap_blocks(n_nodes, FB, 256) : b; }
int grl_lift_bwd_blocks(int n_nodes) { return lift_blocks(n_nodes); }
int grl_lift_bwd_blocks_multi(int n_types, const int* n_nodes) {   // n_nodes: HOST array
  int b = 0;
  for (int t = 0; t < n_types; ++t)
    if (n_nodes[t] > 0) b += lift_blocks(n_nodes[t]);
  return b;
}
#else
int grl_fiber_bwd_blocks(int n_nodes);
int grl_lift_bwd_blocks(int n_nodes);
#endif

int GRL_ENTRY(grl_lift_encode_fwd)(const float* scal, const float* vec, const float* grid, const float* Wenc, st_t* x, int n_nodes,
                        int n_scal, int n_vec, hipStream_t stream) {
  if (n_nodes <= 0) return 0;
  if (n_scal + n_vec > KF_MAX) return -2;
  const int blocks = lift_blocks(n_nodes);
  hipLaunchKernelGGL(lift_encode_fwd_kernel, dim3(blocks), dim3(256), 0, stream, scal, vec, grid, Wenc, x, n_nodes, n_scal,
                     n_vec);
  GRL_CHECK_LAUNCH();
  return 0;
}

// n_types <= 4 node sets that share the encoder (hepi.py:136-143 lifts every node type with the one node_encoder) in ONE launch.
// scal / vec / x: HOST arrays of n_types device pointers, n_nodes: HOST int array; types with n_nodes <= 0 are skipped.
static int lift_multi_fill(LiftMulti& m, int n_types, const float* const* scal, const float* const* vec, st_t* const* x, const int* n_nodes) {
  if (n_types < 1 || n_types > LIFT_MAX_TYPES) return -2;
  m.n_types = 0;
  m.blk0[0] = 0;
  for (int t = 0; t < n_types; ++t) {
    if (n_nodes[t] <= 0) continue;
    const int k = m.n_types++;
    m.scal[k] = scal[t]; m.vec[k] = vec[t]; m.x[k] = x[t]; m.N[k] = n_nodes[t];
    m.blk0[k + 1] = m.blk0[k] + lift_blocks(n_nodes[t]);
  }
  return 0;
}
int GRL_ENTRY(grl_lift_encode_fwd_multi)(int n_types, const float* const* scal, const float* const* vec, const float* grid, const float* Wenc,
                                         st_t* const* x, const int* n_nodes, int n_scal, int n_vec, hipStream_t stream) {
  if (n_scal + n_vec > KF_MAX) return -2;
  LiftMulti m{};
  if (const int rc = lift_multi_fill(m, n_types, scal, vec, x, n_nodes)) return rc;
  if (m.n_types == 0) return 0;
  hipLaunchKernelGGL(lift_encode_fwd_multi_kernel, dim3(m.blk0[m.n_types]), dim3(256), 0, stream, m, grid, Wenc, n_scal, n_vec);
  GRL_CHECK_LAUNCH();
  return 0;
}
// partial: [grl_lift_bwd_blocks_multi(n_types, n_nodes)][64 * (n_scal + n_vec)] -- the types' rows stacked (sum ALL rows: one dW);
// dx: HOST array of n_types device pointers (a type without a gradient: n_nodes[t] = 0)
int GRL_ENTRY(grl_lift_encode_bwd_multi)(int n_types, const float* const* scal, const float* const* vec, const float* grid,
                                         const st_t* const* dx, float* partial, const int* n_nodes, int n_scal, int n_vec, hipStream_t stream) {
  if (n_scal + n_vec > KF_MAX) return -2;
  LiftMulti m{};
  if (const int rc = lift_multi_fill(m, n_types, scal, vec, const_cast<st_t* const*>(dx), n_nodes)) return rc;
  if (m.n_types == 0) return 0;
  hipLaunchKernelGGL(lift_encode_bwd_multi_kernel, dim3(m.blk0[m.n_types]), dim3(256), 0, stream, m, grid, partial, n_scal, n_vec);
  GRL_CHECK_LAUNCH();
  return 0;
}

// partial: [grl_lift_bwd_blocks(n_nodes)][64*(n_scal+n_vec)]
int GRL_ENTRY(grl_lift_encode_bwd)(const float* scal, const float* vec, const float* grid, const st_t* dx, float* partial, int n_nodes,
                        int n_scal, int n_vec, hipStream_t stream) {
  if (n_nodes <= 0) return 0;
  if (n_scal + n_vec > KF_MAX) return -2;
  hipLaunchKernelGGL(lift_encode_bwd_kernel, dim3(grl_lift_bwd_blocks(n_nodes)), dim3(256), 0, stream, scal, vec, grid, dx,
                     partial, n_nodes, n_scal, n_vec);
  GRL_CHECK_LAUNCH();
  return 0;
}

int GRL_ENTRY(grl_fiber_conv_fwd_sig)(const st_t* x1, const float* fk, const float* bias, st_t* x2, int n_nodes, int* flag_dst,
                                      const int* flag_src, hipStream_t stream);
int GRL_ENTRY(grl_fiber_conv_fwd)(const st_t* x1, const float* fk, const float* bias, st_t* x2, int n_nodes, hipStream_t stream) {
  return GRL_ENTRY(grl_fiber_conv_fwd_sig)(x1, fk, bias, x2, n_nodes, nullptr, nullptr, stream);
}
// the same; flag_dst (device int[1] or NULL) := flag_src[0] when the launch starts (see lane_signal)
int GRL_ENTRY(grl_fiber_conv_fwd_sig)(const st_t* x1, const float* fk, const float* bias, st_t* x2, int n_nodes, int* flag_dst,
                                      const int* flag_src, hipStream_t stream) {
  if (n_nodes <= 0) return flag_dst ? -2 : 0;   // (a signal needs a launch to ride on)
  if (flag_dst && !flag_src) return -2;
  hipLaunchKernelGGL(fiber_conv_fwd_kernel, dim3(cap_blocks(n_nodes, FB, 1024)), dim3(256), 0, stream, x1, fk, bias, x2,
                     n_nodes, flag_dst, flag_src);
  GRL_CHECK_LAUNCH();
  return 0;
}

// partial: [grl_fiber_bwd_blocks(n_nodes)][grl_fiber_partial_size()]
int GRL_ENTRY(grl_fiber_conv_bwd)(const st_t* x1, const float* fk, const st_t* dx2, st_t* dx1, float* partial, int n_nodes,
                       hipStream_t stream) {
  if (n_nodes <= 0) return 0;
  hipLaunchKernelGGL(fiber_conv_bwd_kernel, dim3(grl_fiber_bwd_blocks(n_nodes)), dim3(64 * FBW), 0, stream, x1, fk, dx2, dx1,
                     partial, n_nodes);
  GRL_CHECK_LAUNCH();
  return 0;
}

// gate [E,16,64] fp32 (the gate network runs as a plain library GEMM), msg [E,16,64] and x1 [n_dst,16,64] in the latent storage type,
// rowptr [n_dst+1] = destination CSR of the edge set; every row of x1 / dgate / dmsg is written
int GRL_ENTRY(grl_softmax_aggregate_fwd)(const float* gate, const st_t* msg, const int* rowptr, int n_dst, st_t* x1, hipStream_t stream) {
  if (n_dst <= 0) return 0;
  hipLaunchKernelGGL(softmax_agg_fwd_kernel, dim3(cap_blocks((long long)n_dst * O, 4, 4096)), dim3(256), 0, stream, gate, msg, rowptr,
                     n_dst, x1);
  GRL_CHECK_LAUNCH();
  return 0;
}
int GRL_ENTRY(grl_softmax_aggregate_bwd)(const float* gate, const st_t* msg, const st_t* x1, const st_t* dx1, const int* rowptr, int n_dst,
                                         float* dgate, st_t* dmsg, hipStream_t stream) {
  if (n_dst <= 0) return 0;
  hipLaunchKernelGGL(softmax_agg_bwd_kernel, dim3(cap_blocks((long long)n_dst * O, 4, 4096)), dim3(256), 0, stream, gate, msg, x1, dx1,
                     rowptr, n_dst, dgate, dmsg);
  GRL_CHECK_LAUNCH();
  return 0;
}

static int fb_fwd_fill(FbFwd& A, const float* poly, const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* const* wf, int n_conv, float* saved, float* const* fk) {
  if (n_conv < 1 || n_conv > FB_MAXC) return -2;
  A.poly = poly; A.W1 = W1; A.b1 = b1; A.W2 = W2; A.b2 = b2; A.saved = saved;
  A.wf.n = n_conv;
  for (int i = 0; i < n_conv; ++i) { A.wf.w[i] = wf[i]; A.fk.p[i] = fk[i]; }
  return 0;
}
#if !GRL_PREC   // parameter-only and reduction entry points exist once (fp32)
// poly [256,3]; W1 [64,3]; W2 [64,64]; wf: HOST array of n_conv <= 4 device pointers to fiber_kernel weights [64 channels, 64];
// fk: HOST array of n_conv device pointers to outputs [256,64]; saved: scratch [4,256,64] kept for the backward
int grl_fiber_basis_fwd(const float* poly, const float* W1, const float* b1, const float* W2, const float* b2, const float* const* wf,
                        int n_conv, float* saved, float* const* fk, hipStream_t stream) {
  FbFwd A{};
  if (const int rc = fb_fwd_fill(A, poly, W1, b1, W2, b2, wf, n_conv, saved, fk)) return rc;
  hipLaunchKernelGGL(fiber_basis_fwd_kernel, dim3(FB_ROWS / FB_RPB), dim3(256), 0, stream, A);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_fiber_basis_partial_size(int n_conv) { return (n_conv + 1) * 4096 + 64 + 64 * FB_P + 64; }
int grl_fiber_basis_blocks() { return FB_ROWS / FB_RPB; }
static int fb_bwd_fill(FbBwd& A, const float* poly, const float* W2, const float* const* wf, int n_conv, const float* saved,
                       const float* const* dfk, float* partial) {
  if (n_conv < 1 || n_conv > FB_MAXC) return -2;
  A.poly = poly; A.W2 = W2; A.saved = saved; A.partial = partial; A.partial_ld = grl_fiber_basis_partial_size(n_conv);
  A.wf.n = n_conv;
  for (int i = 0; i < n_conv; ++i) { A.wf.w[i] = wf[i]; A.dfk.p[i] = dfk[i]; }
  return 0;
}
// dfk: HOST array of n_conv device pointers [256,64] (NULL = no gradient); partial [grl_fiber_basis_blocks()][partial_size(n_conv)]
int grl_fiber_basis_bwd(const float* poly, const float* W2, const float* const* wf, int n_conv, const float* saved,
                        const float* const* dfk, float* partial, hipStream_t stream) {
  FbBwd A{};
  if (const int rc = fb_bwd_fill(A, poly, W2, wf, n_conv, saved, dfk, partial)) return rc;
  hipLaunchKernelGGL(fiber_basis_bwd_kernel, dim3(FB_ROWS / FB_RPB), dim3(256), 0, stream, A);
  GRL_CHECK_LAUNCH();
  return 0;
}
#endif   // !GRL_PREC

// ---- merged launches (round 6).  grl_step_head: grl_build_features_bump (descs, n_desc, bump) + grl_fiber_basis_fwd (n_conv may be 0:
// role absent) + grl_weight_images (n_img may be 0) in ONE launch; arguments as in those three entry points.  The images depend on the
// build's precision, hence the _bf16 twin.
int GRL_ENTRY(grl_step_head)(const long long* descs, int n_desc, int* bump, const float* poly, const float* W1, const float* b1,
                             const float* W2, const float* b2, const float* const* wf, int n_conv, float* saved, float* const* fk,
                             int n_img, const int* kinds, const float* const* srcs, void* const* outs, hipStream_t stream) {
  if (n_desc < 0 || n_desc > FEAT_MAX || n_img < 0) return -2;
  FeatDescs feat{};
  const int max_nodes = n_desc > 0 ? feat_fill(feat, descs, n_desc) : 1;
  FbFwd A{};
  if (n_conv > 0)
    if (const int rc = fb_fwd_fill(A, poly, W1, b1, W2, b2, wf, n_conv, saved, fk)) return rc;
  WimgJobs jobs{};
  if (n_img > 0)
    if (const int rc = wimg_fill(jobs, n_img, kinds, srcs, outs)) return rc;
  // feature workgroups: grid-stride over the nodes, at most 64 per descriptor (their 41 KB of static LDS -- the fiber-basis role's -- allows
  // four workgroups per compute unit)
  int nfx = (max_nodes + 255) / 256;
  if (nfx > 64) nfx = 64;
  const int blocks = (n_conv > 0 ? FB_ROWS / FB_RPB : 0) + WIMG_PARTS * jobs.n + nfx * n_desc;
  if (blocks <= 0) return 0;
  hipLaunchKernelGGL(step_head_kernel, dim3(blocks), dim3(256), 0, stream, feat, n_desc, nfx, bump, A, jobs);
  GRL_CHECK_LAUNCH();
  return 0;
}
// grl_lift_encode_bwd_multi + grl_fiber_basis_bwd in ONE launch (n_conv may be 0: the lift alone); arguments as in those two entry points
int GRL_ENTRY(grl_lift_fiber_basis_bwd)(int n_types, const float* const* scal, const float* const* vec, const float* grid,
                                        const st_t* const* dx, float* lift_partial, const int* n_nodes, int n_scal, int n_vec,
                                        const float* poly, const float* W2, const float* const* wf, int n_conv, const float* saved,
                                        const float* const* dfk, float* fb_partial, hipStream_t stream) {
  if (n_scal + n_vec > KF_MAX) return -2;
  LiftMulti m{};
  if (n_types > 0)
    if (const int rc = lift_multi_fill(m, n_types, scal, vec, const_cast<st_t* const*>(dx), n_nodes)) return rc;
  FbBwd A{};
  if (n_conv > 0) {
    A.poly = poly; A.W2 = W2; A.saved = saved; A.partial = fb_partial; A.partial_ld = (n_conv + 1) * 4096 + 64 + 64 * FB_P + 64;
    if (n_conv > FB_MAXC) return -2;
    A.wf.n = n_conv;
    for (int i = 0; i < n_conv; ++i) { A.wf.w[i] = wf[i]; A.dfk.p[i] = dfk[i]; }
  }
  const int blocks = (n_conv > 0 ? FB_ROWS / FB_RPB : 0) + (m.n_types > 0 ? m.blk0[m.n_types] : 0);
  if (blocks <= 0) return 0;
  hipLaunchKernelGGL(lift_fiber_basis_bwd_kernel, dim3(blocks), dim3(256), 0, stream, m, grid, lift_partial, n_scal, n_vec, A);
  GRL_CHECK_LAUNCH();
  return 0;
}

#if !GRL_PREC   // (the reductions exist once)
// out[j] += sum over the n_rows partial rows (out must be initialised by the caller)
int grl_reduce_partials(const float* partial, float* out, int n_rows, int n, hipStream_t stream) {
  if (n_rows <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 63) / 64), dim3(64 * RED_WAVES), 0, stream, partial, out, n_rows, n);
  GRL_CHECK_LAUNCH();
  return 0;
}

// dst[i][0..len[i]) += sum_rows partial[row][start[i] + j]   for i < n_seg <= 8 (host arrays of length n_seg)
int grl_reduce_partials_seg(const float* partial, int n_rows, int ld, int n_seg, float* const* dst, const int* start, const int* len,
                            int overwrite_mask, hipStream_t stream) {
  if (n_rows <= 0 || n_seg <= 0) return 0;
  if (n_seg > 8) return -2;
  ReduceSegs segs{};
  int max_len = 0;
  for (int i = 0; i < n_seg; ++i) {
    segs.dst[i] = dst[i];
    segs.start[i] = start[i];
    segs.len[i] = len[i];
    if (len[i] > max_len) max_len = len[i];
  }
  segs.n_seg = n_seg;
  segs.overwrite_mask = overwrite_mask;
  hipLaunchKernelGGL(reduce_partials_seg_kernel, dim3((max_len + 63) / 64, n_seg), dim3(64 * RED_WAVES), 0, stream, partial, segs,
                     n_rows, ld);
  GRL_CHECK_LAUNCH();
  return 0;
}

// n_seg <= 64 folds in one launch: dst[i][0..len[i]) += sum over n_rows[i] rows of partial[i][row*ld[i] + start[i] + j]
// (all arrays are HOST arrays of length n_seg).  Segments with the same destination are summed by the same workgroup, in order.
int grl_reduce_partials_multi_ow(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                                 const int* len, float* const* dst, int overwrite, hipStream_t stream);
int grl_reduce_partials_multi(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                              const int* len, float* const* dst, hipStream_t stream) {
  return grl_reduce_partials_multi_ow(n_seg, partial, n_rows, ld, start, len, dst, 0, stream);
}
// overwrite != 0: every destination is WRITTEN with the sum of its slabs (not accumulated into): the caller needs no zeroed gradient
// buffer, provided every slab of a destination is in THIS call (they are summed by one workgroup in the order given)
static int fold_fill(ReduceMulti& m, int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                     const int* len, float* const* dst, int overwrite);
int grl_reduce_partials_multi_ow(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                                 const int* len, float* const* dst, int overwrite, hipStream_t stream) {
  if (n_seg <= 0) return 0;
  ReduceMulti m{};
  if (const int rc = fold_fill(m, n_seg, partial, n_rows, ld, start, len, dst, overwrite)) return rc;
  if (m.blk0[m.n_dst] <= 0) return 0;
  hipLaunchKernelGGL(reduce_partials_multi_kernel<false>, dim3(m.blk0[m.n_dst]), dim3(64 * RED_WAVES), 0, stream, m, FoldTail{});
  GRL_CHECK_LAUNCH();
  return 0;
}
// Data parallel: the fold of a rank's slabs and, by one extra workgroup, its loss record as (hi, lo) float pairs in front of the flat gradient
// (grl_reduce_partials_multi_ow + grl_trpl_fold_record_pairs in ONE launch: both only feed the lane's all-reduce).
int grl_fold_record_pairs(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                          float* const* dst, int overwrite, const double* slots, int batch, int rank, int world, float* region,
                          hipStream_t stream) {
  if (!slots || !region || batch < 1 || world < 1 || rank < 0 || rank >= world) return -2;
  ReduceMulti m{};
  if (n_seg > 0)
    if (const int rc = fold_fill(m, n_seg, partial, n_rows, ld, start, len, dst, overwrite)) return rc;
  FoldTail t{};
  t.report = 2;
  t.slots = slots; t.n_slot_blocks = trpl_blocks(batch);
  t.pairs_region = reinterpret_cast<float2*>(region); t.pairs_rank = rank; t.pairs_world = world;
  hipLaunchKernelGGL(reduce_partials_multi_kernel<true>, dim3(m.blk0[m.n_dst] + 1), dim3(64 * RED_WAVES), 0, stream, m, t);
  GRL_CHECK_LAUNCH();
  return 0;
}
// The step's tail in one launch (one rank): the fold above, PLUS (adam != 0) the Adam update of every parameter entry whose gradient this
// launch produces -- grads / params / exp_avg / exp_avg_sq are parallel flat buffers (every dst lies inside grads; entries no slab feeds
// keep their zero gradient: Adam would not move them either), lr_dev float[1], step_dev int[1] = the count of THIS step --
// PLUS (slots != NULL) one extra workgroup doing grl_trpl_report's work.  Gradient clipping needs the finished norm first: not here.
int grl_fold_adam_report_sig(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                             float* const* dst, int overwrite, int adam, const float* grads, float* params, float* exp_avg, float* exp_avg_sq,
                             const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev, const double* slots, int batch,
                             double* sums, unsigned int* maxes, float entropy_coef, float* out14, int* flag_dst, const int* flag_src,
                             hipStream_t stream);
int grl_fold_adam_report(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                         float* const* dst, int overwrite, int adam, const float* grads, float* params, float* exp_avg, float* exp_avg_sq,
                         const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev, const double* slots, int batch,
                         double* sums, unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream) {
  return grl_fold_adam_report_sig(n_seg, partial, n_rows, ld, start, len, dst, overwrite, adam, grads, params, exp_avg, exp_avg_sq, lr_dev,
                                  beta1, beta2, eps, step_dev, slots, batch, sums, maxes, entropy_coef, out14, nullptr, nullptr, stream);
}
// the same; flag_dst (device int[1] or NULL) := flag_src[0] when the launch starts (a lane signal riding on the tail: lane_signal)
int grl_fold_adam_report_sig(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                             float* const* dst, int overwrite, int adam, const float* grads, float* params, float* exp_avg, float* exp_avg_sq,
                             const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev, const double* slots, int batch,
                             double* sums, unsigned int* maxes, float entropy_coef, float* out14, int* flag_dst, const int* flag_src,
                             hipStream_t stream) {
  ReduceMulti m{};
  if (n_seg > 0)
    if (const int rc = fold_fill(m, n_seg, partial, n_rows, ld, start, len, dst, overwrite)) return rc;
  FoldTail t{};
  if (adam) {
    if (!grads || !params || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return -2;
    t.adam = 1;
    t.d_param = params - grads; t.d_m = exp_avg - grads; t.d_v = exp_avg_sq - grads;
    t.lr_dev = lr_dev; t.step_dev = step_dev; t.b1 = beta1; t.b2 = beta2; t.eps = eps;
  }
  if (slots) {
    if (!sums || !maxes || !out14 || batch < 1) return -2;
    t.report = 1;
    t.slots = slots; t.n_slot_blocks = trpl_blocks(batch); t.sums = sums; t.maxes = maxes; t.ent_coef = entropy_coef; t.out14 = out14;
  }
  if (flag_dst && !flag_src) return -2;
  t.flag_dst = flag_dst; t.flag_src = flag_src;
  const int blocks = m.blk0[m.n_dst] + (t.report ? 1 : 0);
  if (blocks <= 0) return flag_dst ? -2 : 0;
  hipLaunchKernelGGL(reduce_partials_multi_kernel<true>, dim3(blocks), dim3(64 * RED_WAVES), 0, stream, m, t);
  GRL_CHECK_LAUNCH();
  return 0;
}
static int fold_fill(ReduceMulti& m, int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                     const int* len, float* const* dst, int overwrite) {
  if (n_seg > RED_MULTI_MAX) return -2;
  int n_dst = 0, n_src = 0, max_len = 0;
  bool used[RED_MULTI_MAX] = {false};
  for (int i = 0; i < n_seg; ++i) {
    if (used[i]) continue;
    m.dst[n_dst] = dst[i];
    m.len[n_dst] = len[i];
    m.first[n_dst] = n_src;
    bool vec = (len[i] & 3) == 0 && (reinterpret_cast<size_t>(dst[i]) & 15) == 0;
    for (int k = i; k < n_seg; ++k) {
      if (used[k] || dst[k] != dst[i]) continue;
      if (len[k] != len[i]) return -3;
      used[k] = true;
      vec = vec && (ld[k] & 3) == 0 && (reinterpret_cast<size_t>(partial[k] + start[k]) & 15) == 0;
      m.partial[n_src] = partial[k]; m.n_rows[n_src] = n_rows[k]; m.ld[n_src] = ld[k]; m.start[n_src] = start[k];
      ++n_src;
    }
    m.count[n_dst] = n_src - m.first[n_dst];
    m.vec[n_dst] = vec ? 1 : 0;
    if (len[i] > max_len) max_len = len[i];
    m.blk0[n_dst + 1] = m.blk0[n_dst] + (len[i] + 63) / 64;
    ++n_dst;
  }
  (void)max_len;
  m.n_dst = n_dst;
  m.overwrite = overwrite ? 1 : 0;
  return 0;
}
#endif   // !GRL_PREC

}  // extern "C"
