// DeepSets critic (reference deepsets.py:34-53, gnn_vf_net.py:50-86) with PyG's LayerNorm(mode="graph"):
//   h1 = Lin1(x)            [B, n, 64]      x: [B, n, d]  (d = n_types + 3 n_vec <= 16)
//   y1 = relu(LNg(h1))                      LNg: (h - mean_all) / (std_all + 1e-5) * gamma + beta  -- statistics over the
//   h2 = Lin2(y1); z = sum_n h2             whole tensor, i.e. over the whole (global) minibatch
//   u1 = Lin3(z); y2 = relu(LNg(u1)); u2 = Lin4(y2); V = Linv(u2)
// The whole-tensor statistics split the pass into three launches forward and three backward; each launch leaves the sums the
// next one needs as per-workgroup fp64 pairs in a slot array (NSLOT pairs, unused slots zeroed by the producer) which the
// consumer adds up in a fixed order -- no atomics (same-address fp64 atomics serialise in L2 at ~15 ns each: 1024 of them were a
// fixed 15 us per launch) and bit-reproducible.  A data-parallel run all-reduces the slot arrays between the launches.
// Four waves per workgroup, one wave per sample (or per node row), thread j owns feature j; 64x64 weights live in registers
// (row j, and column j where the transpose is needed) or LDS; weight gradients accumulate in registers over a grid-stride loop
// and leave as ONE partial row per workgroup.  Lin2 is followed directly by the sum over nodes, so z = W2 (sum_n y1_n) + n b2:
// one 64x64 product per SAMPLE instead of one per node row, forward as well as backward.
#include "grl_common.h"

namespace {

constexpr int H = 64;
constexpr int DMAX = 16;
constexpr float LNG_EPS = 1e-5f;

GRL_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// lane k's value to every lane through an SGPR (v_readlane_b32; k is a compile-time constant after unrolling).  __shfl is a
// ds_bpermute: 16 of them per row made the two row kernels LDS-crossbar-bound (cloth critic: 1 M rows, 0.45 ms for 0.07 ms of HBM time)
GRL_DEVINL float bcast_lane(float v, int k) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), k)); }
GRL_DEVINL double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int DS_WAVES = 4;
constexpr int NSLOT = 256;   // statistic slots = upper bound of every stage's grid

// Wave sums -> this workgroup's slot (plain stores); workgroup 0 clears the slots no workgroup owns.
template <int WAVES = DS_WAVES>
GRL_DEVINL void block_put_stats(double s0, double s1, double* __restrict__ slots) {
  __shared__ double red_stats[WAVES][2];
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red_stats[wave][0] = s0; red_stats[wave][1] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = red_stats[0][0], t1 = red_stats[0][1];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { t0 += red_stats[w][0]; t1 += red_stats[w][1]; }
    slots[2 * blockIdx.x] = t0;
    slots[2 * blockIdx.x + 1] = t1;
  }
  if (blockIdx.x == 0)
    for (int s_ = gridDim.x + threadIdx.x; s_ < NSLOT; s_ += blockDim.x) { slots[2 * s_] = 0.0; slots[2 * s_ + 1] = 0.0; }
}

// Totals of up to DS_WAVES slot arrays (wave w adds up array w: lane-strided, then a butterfly -- a fixed order), via LDS.
template <int K>
GRL_DEVINL void slot_totals(const double* const (&arr)[K], double (&tot)[K][2]) {
  __shared__ double tot_s[K][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < K) {
    const double2* sl = reinterpret_cast<const double2*>(arr[wave]);
    double a = 0, b = 0;
#pragma unroll
    for (int q = 0; q < NSLOT / 64; ++q) { const double2 v = sl[lane + 64 * q]; a += v.x; b += v.y; }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane == 0) { tot_s[wave][0] = a; tot_s[wave][1] = b; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) { tot[k][0] = tot_s[k][0]; tot[k][1] = tot_s[k][1]; }
}

struct LnStat { float mean, s, sigma; };  // s = sigma + eps
GRL_DEVINL LnStat ln_stat(const double (&sums)[2], double count) {
  const double m = sums[0] / count;
  double var = sums[1] / count - m * m;
  if (var < 0) var = 0;
  LnStat st;
  st.mean = (float)m;
  st.sigma = (float)sqrt(var);
  st.s = st.sigma + LNG_EPS;
  return st;
}

// row j of a row-major 64x64 matrix -> registers
GRL_DEVINL void load_row64(const float* __restrict__ Wm, int j, float (&w)[H]) {
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 a = *reinterpret_cast<const float4*>(Wm + j * H + k);
    w[k] = a.x; w[k + 1] = a.y; w[k + 2] = a.z; w[k + 3] = a.w;
  }
}
// 64-term dot product of an LDS vector (float4 broadcasts: 16 reads) with a register row
GRL_DEVINL float dot64(const float* __restrict__ ys /*LDS, 16-byte aligned*/, const float (&w)[H]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 y = *reinterpret_cast<const float4*>(ys + k);
    a0 = fmaf(y.x, w[k], a0); a1 = fmaf(y.y, w[k + 1], a1); a2 = fmaf(y.z, w[k + 2], a2); a3 = fmaf(y.w, w[k + 3], a3);
  }
  return (a0 + a1) + (a2 + a3);
}
// acc[k] += s * ys[k]
GRL_DEVINL void axpy64(const float* __restrict__ ys, float s_, float (&acc)[H]) {
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 y = *reinterpret_cast<const float4*>(ys + k);
    acc[k] = fmaf(s_, y.x, acc[k]); acc[k + 1] = fmaf(s_, y.y, acc[k + 1]);
    acc[k + 2] = fmaf(s_, y.z, acc[k + 2]); acc[k + 3] = fmaf(s_, y.w, acc[k + 3]);
  }
}
// the workgroup's waves' register rows acc[.] (thread j = matrix row j) summed in wave order into out[64][64]; buf: LDS [64][65]
template <int WAVES = DS_WAVES>
GRL_DEVINL void fold_rows64(const float (&acc)[H], float* __restrict__ buf, float* __restrict__ out) {
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w_ = 0; w_ < WAVES; ++w_) {
    if (wave == w_) {
#pragma unroll
      for (int k = 0; k < H; ++k) buf[j * (H + 1) + k] = w_ == 0 ? acc[k] : buf[j * (H + 1) + k] + acc[k];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < H * H; i += 64 * WAVES) out[i] = buf[(i >> 6) * (H + 1) + (i & 63)];
  __syncthreads();
}
// per-thread scalars of the workgroup's waves summed in wave order: out[q * 64 + j]; buf: LDS [Q][WAVES][64]
template <int Q, int WAVES = DS_WAVES>
GRL_DEVINL void fold_scalars(const float (&v)[Q], float* __restrict__ buf, float* const (&out)[Q]) {
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < Q; ++q) buf[(q * WAVES + wave) * H + j] = v[q];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float t = buf[(q * WAVES) * H + j];
#pragma unroll
      for (int w_ = 1; w_ < WAVES; ++w_) t += buf[(q * WAVES + w_) * H + j];
      out[q][j] = t;
    }
  }
  __syncthreads();
}

// ---- forward 1: h1 = x W1^T + b1 ; slot sums of (h1, h1^2).  Rows (sample, node) are independent: grid-stride over the
//      flattened rows, eight rows in flight per wave; a row's d inputs arrive with ONE load (lane k < d fetches x[row][k]) and
//      are handed round with v_readlane.
//      These two row kernels are latency-bound (one 60-byte and one 256-byte access per row): sixteen waves per workgroup keep
//      4096 waves x 4 rows in flight under the NSLOT-workgroup cap.
constexpr int ROWS_IN_FLIGHT = 4;
constexpr int ROW_WAVES = 16;
__global__ __launch_bounds__(64 * ROW_WAVES) void ds_fwd1(const float* __restrict__ x, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, float* __restrict__ h1,
                                                        double* __restrict__ slots, int B, int n, int d) {
  {   // group blockIdx.y of a grouped launch (time steps of the rollout pass): its own frames and its own statistic slots
    const size_t g = blockIdx.y;
    x += g * (size_t)B * n * d; h1 += g * (size_t)B * n * H; slots += g * 2 * NSLOT;
  }
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) w[k] = k < d ? W1[j * d + k] : 0.f;
  const float bj = b1[j];
  double s0 = 0, s1 = 0;
  const long long rows = (long long)B * n;
  const long long stride = (long long)gridDim.x * ROW_WAVES;
  for (long long row0 = blockIdx.x * ROW_WAVES + wave; row0 < rows; row0 += ROWS_IN_FLIGHT * stride) {
    float xv[ROWS_IN_FLIGHT];
#pragma unroll
    for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
      const long long row = row0 + u * stride;
      xv[u] = (row < rows && j < d) ? x[row * d + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
      const long long row = row0 + u * stride;
      if (row >= rows) break;
      float acc = bj;
#pragma unroll
      for (int k = 0; k < DMAX; ++k)
        if (k < d) acc += bcast_lane(xv[u], k) * w[k];
      h1[row * H + j] = acc;
      s0 += acc;
      s1 += (double)acc * acc;
    }
  }
  block_put_stats<ROW_WAVES>(s0, s1, slots);
}

// ---- forward 2: y1 = relu(LNg(h1)); z = W2 sum_n y1 + n b2; u1 = z W3^T + b3; slot sums of (u1, u1^2)
//      Thread j keeps row j of W2 and of W3 in registers; vectors are exchanged through wave-private LDS lines.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ds_fwd2(const float* __restrict__ h1, const double* __restrict__ slots1, double count1,
                                                        const float* __restrict__ g1, const float* __restrict__ be1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        const float* __restrict__ W3, const float* __restrict__ b3,
                                                        float* __restrict__ z, float* __restrict__ u1, double* __restrict__ slots2,
                                                        int B, int n) {
  {
    const size_t g = blockIdx.y;
    h1 += g * (size_t)B * n * H; slots1 += g * 2 * NSLOT; z += g * (size_t)B * H; u1 += g * (size_t)B * H; slots2 += g * 2 * NSLOT;
  }
  __shared__ __attribute__((aligned(16))) float ys_all[WAVES][2][H];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float (*ys)[H] = ys_all[wave];
  float w2[H], w3[H];
  load_row64(W2, j, w2);
  load_row64(W3, j, w3);
  double tot[1][2];
  const double* const arr[1] = {slots1};
  slot_totals<1>(arr, tot);
  const LnStat st = ln_stat(tot[0], count1);
  const float gj = g1[j], bj = be1[j], b2j = b2[j], b3j = b3[j];
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * WAVES + wave; b < B; b += gridDim.x * WAVES) {
    const float* hrow = h1 + (size_t)b * n * H + j;
    float ysum = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8) {   // eight rows in flight
      float hv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) hv[u] = i0 + u < n ? hrow[(size_t)(i0 + u) * H] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < n) ysum += fmaxf((hv[u] - st.mean) / st.s * gj + bj, 0.f);
    }
    ys[0][j] = ysum;
    __builtin_amdgcn_wave_barrier();
    const float zj = fmaf((float)n, b2j, dot64(ys[0], w2));
    z[(size_t)b * H + j] = zj;
    ys[1][j] = zj;
    __builtin_amdgcn_wave_barrier();
    const float acc = b3j + dot64(ys[1], w3);
    u1[(size_t)b * H + j] = acc;
    s0 += acc;
    s1 += (double)acc * acc;
    __builtin_amdgcn_wave_barrier();
  }
  block_put_stats<WAVES>(s0, s1, slots2);
}

// ---- forward 3: y2 = relu(LNg(u1)); u2 = y2 W4^T + b4; V = u2 . wv + bv
__global__ __launch_bounds__(64 * DS_WAVES) void ds_fwd3(const float* __restrict__ u1, const double* __restrict__ slots2, double count2,
                                                        const float* __restrict__ g2, const float* __restrict__ be2,
                                                        const float* __restrict__ W4, const float* __restrict__ b4,
                                                        const float* __restrict__ wv, const float* __restrict__ bv,
                                                        float* __restrict__ value, int B) {
  {
    const size_t g = blockIdx.y;
    u1 += g * (size_t)B * H; slots2 += g * 2 * NSLOT; value += g * (size_t)B;
  }
  __shared__ __attribute__((aligned(16))) float ys_all[DS_WAVES][H];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ys = ys_all[wave];
  float w4[H];
  load_row64(W4, j, w4);
  double tot[1][2];
  const double* const arr[1] = {slots2};
  slot_totals<1>(arr, tot);
  const LnStat st = ln_stat(tot[0], count2);
  const float gj = g2[j], bj = be2[j], b4j = b4[j], wvj = wv[j], bv0 = bv[0];
  for (int b = blockIdx.x * DS_WAVES + wave; b < B; b += gridDim.x * DS_WAVES) {
    ys[j] = fmaxf((u1[(size_t)b * H + j] - st.mean) / st.s * gj + bj, 0.f);
    __builtin_amdgcn_wave_barrier();
    const float acc = b4j + dot64(ys, w4);
    const float v = wave_sum(acc * wvj);
    if (j == 0) value[b] = v + bv0;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- backward 3: from dV.  Writes q2 = dy2 * relu'(.) * gamma2 (the LNg input-gradient numerator), slot sums of
//      (q2, q2 * xhat2); weight grads of Lin4, Linv and LNg2 affine.  Row j AND column j of W4 in registers.
// partial row: [dW4 64x64 | db4 64 | dwv 64 | dg2 64 | dbe2 64 | dbv 1 | 3 unused]  (every segment 16-byte aligned)
constexpr int P3 = H * H + 4 * H + 4;
__global__ __launch_bounds__(64 * DS_WAVES) void ds_bwd3(const float* __restrict__ u1, const double* __restrict__ slots2, double count2,
                                                        const float* __restrict__ g2, const float* __restrict__ be2,
                                                        const float* __restrict__ W4, const float* __restrict__ b4,
                                                        const float* __restrict__ wv, const float* __restrict__ dvalue,
                                                        float* __restrict__ q2, double* __restrict__ bslots2,
                                                        float* __restrict__ partial, int B) {
  __shared__ __attribute__((aligned(16))) float lines[DS_WAVES][2][H];
  __shared__ float fold[H * (H + 1)];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ys = lines[wave][0];
  float* du = lines[wave][1];
  float w4r[H], w4c[H], dW4[H];
  load_row64(W4, j, w4r);
#pragma unroll
  for (int k = 0; k < H; ++k) { w4c[k] = W4[k * H + j]; dW4[k] = 0.f; }
  double tot[1][2];
  const double* const arr[1] = {slots2};
  slot_totals<1>(arr, tot);
  const LnStat st = ln_stat(tot[0], count2);
  const float gj = g2[j], bj = be2[j], b4j = b4[j], wvj = wv[j];
  float db4 = 0.f, dwv = 0.f, dbv = 0.f, dg = 0.f, dbe = 0.f;
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * DS_WAVES + wave; b < B; b += gridDim.x * DS_WAVES) {
    const float xh = (u1[(size_t)b * H + j] - st.mean) / st.s;
    const float pre = xh * gj + bj;
    ys[j] = fmaxf(pre, 0.f);
    const float dv = dvalue[b];
    const float du2 = dv * wvj;
    du[j] = du2;
    __builtin_amdgcn_wave_barrier();
    const float u2 = b4j + dot64(ys, w4r);
    dwv += dv * u2;
    dbv += dv;
    db4 += du2;
    axpy64(ys, du2, dW4);
    const float dy = dot64(du, w4c);
    const float dpre = pre > 0.f ? dy : 0.f;
    dg += dpre * xh;
    dbe += dpre;
    const float q = dpre * gj;
    q2[(size_t)b * H + j] = q;
    s0 += q;
    s1 += (double)q * xh;
    __builtin_amdgcn_wave_barrier();
  }
  block_put_stats(s0, s1, bslots2);
  float* out = partial + (size_t)blockIdx.x * P3;
  fold_rows64(dW4, fold, out);
  const float sc[5] = {db4, dwv, dg, dbe, dbv};
  float* const outs[5] = {out + H * H, out + H * H + H, out + H * H + 2 * H, out + H * H + 3 * H, fold + 5 * DS_WAVES * H};
  fold_scalars<5>(sc, fold, outs);
  if (threadIdx.x == 0) out[H * H + 4 * H] = fold[5 * DS_WAVES * H];   // dbv is the same in every lane: lane 0's folded value
}

// ---- backward 2: du1 (LNg2 backward) -> Lin3 -> dz; dz fans out to every node row of the sample -> Lin2 -> dy1 (same for
//      all rows) -> q1 = dy1 * relu'(.) * gamma1 written per row; slot sums of (q1, q1 xhat1).
// partial row: [dW3 64x64 | db3 64 | dW2 64x64 | db2 64 | dg1 64 | dbe1 64]
constexpr int P2 = H * H + H + H * H + H + H + H;
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ds_bwd2(const float* __restrict__ h1, const double* __restrict__ slots1, double count1,
                                                        const float* __restrict__ g1, const float* __restrict__ be1,
                                                        const float* __restrict__ W2, const float* __restrict__ W3,
                                                        const float* __restrict__ z, const float* __restrict__ u1,
                                                        const double* __restrict__ slots2, double count2,
                                                        const float* __restrict__ q2, const double* __restrict__ bslots2,
                                                        float* __restrict__ q1, double* __restrict__ bslots1,
                                                        float* __restrict__ partial, int B, int n) {
  // the weight images (read by column) live in LDS and are shared by the four waves; their space is reused for the folds
  __shared__ __attribute__((aligned(16))) float W2s[H * (H + 1)], W3s[H * (H + 1)];
  __shared__ __attribute__((aligned(16))) float sh_all[WAVES][H], sy_all[WAVES][H];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sh = sh_all[wave];
  float* sy = sy_all[wave];
  {
    constexpr int NT = 64 * WAVES, NQ = H * H / 4 / NT;   // quads per matrix and thread: all loads in flight before the first LDS store
    float4 a[NQ], c[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      a[u] = reinterpret_cast<const float4*>(W2)[threadIdx.x + NT * u];
      c[u] = reinterpret_cast<const float4*>(W3)[threadIdx.x + NT * u];
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int i = 4 * (threadIdx.x + NT * u), r_ = i >> 6, k_ = i & 63;
      float* d2 = W2s + r_ * (H + 1) + k_;
      float* d3 = W3s + r_ * (H + 1) + k_;
      d2[0] = a[u].x; d2[1] = a[u].y; d2[2] = a[u].z; d2[3] = a[u].w;
      d3[0] = c[u].x; d3[1] = c[u].y; d3[2] = c[u].z; d3[3] = c[u].w;
    }
  }
  double tot[3][2];
  const double* const arr[3] = {slots1, slots2, bslots2};
  slot_totals<3>(arr, tot);   // (its barrier also publishes the weight images)
  const LnStat st1 = ln_stat(tot[0], count1), st2 = ln_stat(tot[1], count2);
  const float mq = (float)(tot[2][0] / count2);                               // mean(q2)
  const float cq = st2.sigma > 0.f ? (float)(tot[2][1] / count2) / st2.sigma : 0.f;  // sum(q2 xhat2) / (N sigma2)
  const float gj = g1[j], bj = be1[j];
  float dW3[H], dW2[H];
#pragma unroll
  for (int k = 0; k < H; ++k) { dW3[k] = 0.f; dW2[k] = 0.f; }
  float db3 = 0.f, db2 = 0.f, dg = 0.f, dbe = 0.f;
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * WAVES + wave; b < B; b += gridDim.x * WAVES) {
    // LNg2 backward: du1 = (q2 - mean(q2)) / s - xhat2 * sum(q2 xhat2) / (N sigma)
    const float xh2 = (u1[(size_t)b * H + j] - st2.mean) / st2.s;
    const float du1 = (q2[(size_t)b * H + j] - mq) / st2.s - xh2 * cq;
    db3 += du1;
    sh[j] = du1;
    sy[j] = z[(size_t)b * H + j];
    __builtin_amdgcn_wave_barrier();
    axpy64(sy, du1, dW3);
    float dz = 0.f;
#pragma unroll 16
    for (int k = 0; k < H; ++k) dz += sh[k] * W3s[k * (H + 1) + j];
    __builtin_amdgcn_wave_barrier();
    db2 += dz * n;
    sh[j] = dz;
    __builtin_amdgcn_wave_barrier();
    float dy1 = 0.f;  // identical for every node row of this sample
#pragma unroll 16
    for (int k = 0; k < H; ++k) dy1 += sh[k] * W2s[k * (H + 1) + j];
    float ysum = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8) {   // eight rows in flight
      float hv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) hv[u] = i0 + u < n ? h1[((size_t)b * n + i0 + u) * H + j] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i0 + u >= n) break;
        const float xh = (hv[u] - st1.mean) / st1.s;
        const float pre = xh * gj + bj;
        ysum += fmaxf(pre, 0.f);
        const float dpre = pre > 0.f ? dy1 : 0.f;
        dg += dpre * xh;
        dbe += dpre;
        const float q = dpre * gj;
        q1[((size_t)b * n + i0 + u) * H + j] = q;
        s0 += q;
        s1 += (double)q * xh;
      }
    }
    sy[j] = ysum;
    __builtin_amdgcn_wave_barrier();
    axpy64(sy, dz, dW2);
    __builtin_amdgcn_wave_barrier();
  }
  block_put_stats<WAVES>(s0, s1, bslots1);
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * P2;
  fold_rows64<WAVES>(dW3, W3s, out);
  fold_rows64<WAVES>(dW2, W2s, out + H * H + H);
  const float sc[4] = {db3, db2, dg, dbe};
  float* const outs[4] = {out + H * H, out + 2 * H * H + H, out + 2 * H * H + 2 * H, out + 2 * H * H + 3 * H};
  fold_scalars<4, WAVES>(sc, W3s, outs);
}

// ---- backward 1: dh1 (LNg1 backward) -> dW1, db1.   partial row: [dW1 64 x d | db1 64]   (rows flattened, 16 waves per workgroup)
__global__ __launch_bounds__(64 * ROW_WAVES) void ds_bwd1(const float* __restrict__ x, const float* __restrict__ h1,
                                                        const double* __restrict__ slots1, double count1,
                                                        const float* __restrict__ q1, const double* __restrict__ bslots1,
                                                        float* __restrict__ partial, int B, int n, int d) {
  __shared__ float red[ROW_WAVES / 2][DMAX + 1][H];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double tot[2][2];
  const double* const arr[2] = {slots1, bslots1};
  slot_totals<2>(arr, tot);
  const LnStat st = ln_stat(tot[0], count1);
  const float mq = (float)(tot[1][0] / count1);
  const float cq = st.sigma > 0.f ? (float)(tot[1][1] / count1) / st.sigma : 0.f;
  float dW1[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) dW1[k] = 0.f;
  float db1 = 0.f;
  const long long rows = (long long)B * n;
  const long long stride = (long long)gridDim.x * ROW_WAVES;
  for (long long row0 = blockIdx.x * ROW_WAVES + wave; row0 < rows; row0 += ROWS_IN_FLIGHT * stride) {
    float xv[ROWS_IN_FLIGHT], hv[ROWS_IN_FLIGHT], qv[ROWS_IN_FLIGHT];
#pragma unroll
    for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
      const long long row = row0 + u * stride;
      const bool ok = row < rows;
      xv[u] = (ok && j < d) ? x[row * d + j] : 0.f;
      hv[u] = ok ? h1[row * H + j] : 0.f;
      qv[u] = ok ? q1[row * H + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
      if (row0 + u * stride >= rows) break;
      const float xh = (hv[u] - st.mean) / st.s;
      const float dh = (qv[u] - mq) / st.s - xh * cq;
      db1 += dh;
#pragma unroll
      for (int k = 0; k < DMAX; ++k)
        if (k < d) dW1[k] += dh * bcast_lane(xv[u], k);
    }
  }
  // fixed-order fold of the sixteen waves: upper half -> LDS -> added by the lower half -> LDS -> wave 0
  constexpr int HW = ROW_WAVES / 2;
  if (wave >= HW) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) red[wave - HW][k][j] = dW1[k];
    red[wave - HW][DMAX][j] = db1;
  }
  __syncthreads();
  if (wave < HW) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) dW1[k] += red[wave][k][j];
    db1 += red[wave][DMAX][j];
  }
  __syncthreads();
  if (wave < HW) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) red[wave][k][j] = dW1[k];
    red[wave][DMAX][j] = db1;
  }
  __syncthreads();
  if (wave == 0) {
    float* out = partial + (size_t)blockIdx.x * (H * d + H);
    for (int k = 0; k < d; ++k) {
      float t = red[0][k][j];
#pragma unroll
      for (int w_ = 1; w_ < HW; ++w_) t += red[w_][k][j];
      out[j * d + k] = t;
    }
    float t = red[0][DMAX][j];
#pragma unroll
    for (int w_ = 1; w_ < HW; ++w_) t += red[w_][DMAX][j];
    out[H * d + j] = t;
  }
}

// ================================================================================================================================
// Many-row variants (QUAD_MIN_N or more rows per sample: the cloth critic has 239, the rope critic 162).  The kernels above keep lane =
// channel and move one 256-byte row per access -- right for the rigid tasks' 33 rows per sample, where the stages are a few tens of
// microseconds of prologue and latency; with ~1 M rows they ran at 1.0-1.5 TB/s (cloth: forward 2 alone 0.56 ms inside the step).
// ================================================================================================================================
constexpr int QUAD_MIN_N = 64;
constexpr int LANE8_MIN_BATCH = 8 * NSLOT;   // lane kernels of the per-sample passes: eight waves per workgroup from 2 048 samples up
// ---- Row passes (forward 1 / 2, backward 2 / 1) walk [rows][64] fp32 arrays FOUR ROWS PER INSTRUCTION: lane = (r = lane >> 4, c4 = lane & 15)
//      owns channels 4 c4 .. 4 c4 + 3 of row 4 q + r, so a load or store is 1 KB contiguous and a wave keeps 8-16 of them in flight.
//      (Round 2: lane = channel, one 256-byte dword access per row, 4-8 rows in flight, and only 1 024 waves on the per-sample passes:
//      1.0-1.5 TB/s; the cloth critic -- 1 M rows -- spent 0.56 ms in forward 2 alone, rocprof per-launch medians in profiles/.)
//      The d <= 16 inputs of four consecutive rows are 4 d <= 64 consecutive floats: ONE load (lane i = element i), handed to the lanes
//      of row r by ds_bpermute (index r d + k).
GRL_DEVINL float4 ld_f4(const float* p) { return *reinterpret_cast<const float4*>(p); }
GRL_DEVINL void st_f4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
GRL_DEVINL float4 quad_fold(float4 v) {   // sum over the four row groups of lanes (r): every lane ends with the total
  v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
  v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
  return v;
}

// ---- forward 1, four rows per instruction: grid-stride over quads of rows
constexpr int QUADS_IN_FLIGHT = 4;     // forward 1 (one 4-byte load per quad)
constexpr int QUADS_IN_FLIGHT_B = 2;   // backward 1 (one 4-byte + two 16-byte loads per quad, 68 accumulators)
constexpr int XS_LINE = 80;
__global__ __launch_bounds__(64 * ROW_WAVES) void ds_fwd1_quad(const float* __restrict__ x, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, float* __restrict__ h1,
                                                        double* __restrict__ slots, int B, int n, int d) {
  {   // group blockIdx.y of a grouped launch (time steps of the rollout pass): its own frames and its own statistic slots
    const size_t g = blockIdx.y;
    x += g * (size_t)B * n * d; h1 += g * (size_t)B * n * H; slots += g * 2 * NSLOT;
  }
  const int lane = threadIdx.x & 63, r = lane >> 4, c4 = lane & 15, wave = threadIdx.x >> 6;
  float w[4][DMAX];   // rows 4 c4 + i of W1, zero past d: every k takes the same multiply-add (the shuffled value is some finite input)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      const float v = W1[(4 * c4 + i) * d + (k < d ? k : d - 1)];   // (clamped + select: 64 guarded loads spilled their addresses)
      w[i][k] = k < d ? v : 0.f;
    }
  const float4 bj = ld_f4(b1 + 4 * c4);
  // a quad's 4 d inputs, wave-private: written by one store (lane i = element i), read back at r d + k (k = 0 .. 15: immediate offsets);
  // 16 zeroed floats behind the 64 catch the reads past the last row's inputs (k >= d meets a zero weight, but NaN x 0 is NaN)
  __shared__ float xs_all[ROW_WAVES][QUADS_IN_FLIGHT][XS_LINE];
  float (*xs)[XS_LINE] = xs_all[wave];
  if (lane < XS_LINE - 64)
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT; ++u) xs[u][64 + lane] = 0.f;
  double s0 = 0, s1 = 0;
  const long long rows = (long long)B * n, quads = (rows + 3) >> 2, last = rows * d - 1;
  const long long stride = (long long)gridDim.x * ROW_WAVES;
  for (long long q0 = (long long)blockIdx.x * ROW_WAVES + wave; q0 < quads; q0 += QUADS_IN_FLIGHT * stride) {
    float xq[QUADS_IN_FLIGHT];
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT; ++u) {
      const long long e = (q0 + u * stride) * 4 * d + lane;
      xq[u] = x[e < last ? e : last];          // clamped, not guarded: no branch around the load
    }
    __builtin_amdgcn_wave_barrier();           // (the previous round's reads of the lines are done)
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT; ++u) xs[u][lane] = xq[u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT; ++u) {
      const long long q = q0 + u * stride;
      if (q >= quads) break;
      float acc[4] = {bj.x, bj.y, bj.z, bj.w};
      const float* xr = xs[u] + r * d;
#pragma unroll
      for (int k = 0; k < DMAX; ++k) {
        const float xv = xr[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += xv * w[i][k];
      }
      const long long row = 4 * q + r;
      if (row < rows) {
        st_f4(h1 + row * H + 4 * c4, make_float4(acc[0], acc[1], acc[2], acc[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += acc[i]; s1 += (double)acc[i] * acc[i]; }
      }
      __builtin_amdgcn_sched_barrier(0);   // one quad at a time: sixteen shuffled inputs live, not sixty-four (128-VGPR budget)
    }
  }
  block_put_stats<ROW_WAVES>(s0, s1, slots);
}

// ---- forward 2, four rows per load: one wave per sample, EIGHT waves per workgroup (2 048 waves under the NSLOT cap); the sample's n rows are summed four rows per
//      load, 32 rows in flight; thread j keeps row j of W2 and of W3 in registers, vectors are exchanged through wave-private LDS lines.
constexpr int S_WAVES = 8;        // waves per workgroup of the per-sample passes (forward 2, backward 2)
constexpr int S_RU = 8;           // row quads in flight per wave, forward 2
constexpr int S_RU_B = 4;         // ... backward 2
__global__ __launch_bounds__(64 * S_WAVES) void ds_fwd2_quad(const float* __restrict__ h1, const double* __restrict__ slots1, double count1,
                                                       const float* __restrict__ g1, const float* __restrict__ be1,
                                                       const float* __restrict__ W2, const float* __restrict__ b2,
                                                       const float* __restrict__ W3, const float* __restrict__ b3,
                                                       float* __restrict__ z, float* __restrict__ u1, double* __restrict__ slots2,
                                                       int B, int n) {
  {
    const size_t g = blockIdx.y;
    h1 += g * (size_t)B * n * H; slots1 += g * 2 * NSLOT; z += g * (size_t)B * H; u1 += g * (size_t)B * H; slots2 += g * 2 * NSLOT;
  }
  __shared__ __attribute__((aligned(16))) float ys_all[S_WAVES][2][H];
  const int j = threadIdx.x & 63, r = j >> 4, c4 = j & 15, wave = threadIdx.x >> 6;
  float (*ys)[H] = ys_all[wave];
  float w2[H], w3[H];
  load_row64(W2, j, w2);
  load_row64(W3, j, w3);
  double tot[1][2];
  const double* const arr[1] = {slots1};
  slot_totals<1>(arr, tot);
  const LnStat st = ln_stat(tot[0], count1);
  const float4 g4 = ld_f4(g1 + 4 * c4), be4 = ld_f4(be1 + 4 * c4);
  const float b2j = b2[j], b3j = b3[j];
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * S_WAVES + wave; b < B; b += gridDim.x * S_WAVES) {
    const float* hs = h1 + (size_t)b * n * H + 4 * c4;
    float4 ysum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = 0; i0 < n; i0 += 4 * S_RU) {
      float4 hv[S_RU];
#pragma unroll
      for (int u = 0; u < S_RU; ++u) {
        const int row = i0 + 4 * u + r;
        hv[u] = ld_f4(hs + (size_t)(row < n ? row : n - 1) * H);
      }
#pragma unroll
      for (int u = 0; u < S_RU; ++u) {
        const float m = i0 + 4 * u + r < n ? 1.f : 0.f;     // rows past the sample's end were loaded clamped: they add nothing
        ysum.x += m * fmaxf((hv[u].x - st.mean) / st.s * g4.x + be4.x, 0.f);
        ysum.y += m * fmaxf((hv[u].y - st.mean) / st.s * g4.y + be4.y, 0.f);
        ysum.z += m * fmaxf((hv[u].z - st.mean) / st.s * g4.z + be4.z, 0.f);
        ysum.w += m * fmaxf((hv[u].w - st.mean) / st.s * g4.w + be4.w, 0.f);
      }
    }
    ysum = quad_fold(ysum);
    if (r == 0) st_f4(ys[0] + 4 * c4, ysum);
    __builtin_amdgcn_wave_barrier();
    const float zj = fmaf((float)n, b2j, dot64(ys[0], w2));
    z[(size_t)b * H + j] = zj;
    ys[1][j] = zj;
    __builtin_amdgcn_wave_barrier();
    const float acc = b3j + dot64(ys[1], w3);
    u1[(size_t)b * H + j] = acc;
    s0 += acc;
    s1 += (double)acc * acc;
    __builtin_amdgcn_wave_barrier();
  }
  block_put_stats<S_WAVES>(s0, s1, slots2);
}

// ---- backward 2, four rows per load (the stages are those of ds_bwd2 above)
__global__ __launch_bounds__(64 * S_WAVES) void ds_bwd2_quad(const float* __restrict__ h1, const double* __restrict__ slots1, double count1,
                                                       const float* __restrict__ g1, const float* __restrict__ be1,
                                                       const float* __restrict__ W2, const float* __restrict__ W3,
                                                       const float* __restrict__ z, const float* __restrict__ u1,
                                                       const double* __restrict__ slots2, double count2,
                                                       const float* __restrict__ q2, const double* __restrict__ bslots2,
                                                       float* __restrict__ q1, double* __restrict__ bslots1,
                                                       float* __restrict__ partial, int B, int n) {
  // the weight images (read by column) live in LDS and are shared by the waves; their space is reused for the folds
  __shared__ __attribute__((aligned(16))) float W2s[H * (H + 1)], W3s[H * (H + 1)];
  __shared__ __attribute__((aligned(16))) float sh_all[S_WAVES][H], sy_all[S_WAVES][H];
  const int j = threadIdx.x & 63, r = j >> 4, c4 = j & 15, wave = threadIdx.x >> 6;
  float* sh = sh_all[wave];
  float* sy = sy_all[wave];
  {
    constexpr int NT = 64 * S_WAVES, NQ = H * H / 4 / NT;   // quads per matrix and thread: all loads in flight before the first LDS store
    float4 a[NQ], c[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      a[u] = reinterpret_cast<const float4*>(W2)[threadIdx.x + NT * u];
      c[u] = reinterpret_cast<const float4*>(W3)[threadIdx.x + NT * u];
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int i = 4 * (threadIdx.x + NT * u), r_ = i >> 6, k_ = i & 63;
      float* d2 = W2s + r_ * (H + 1) + k_;
      float* d3 = W3s + r_ * (H + 1) + k_;
      d2[0] = a[u].x; d2[1] = a[u].y; d2[2] = a[u].z; d2[3] = a[u].w;
      d3[0] = c[u].x; d3[1] = c[u].y; d3[2] = c[u].z; d3[3] = c[u].w;
    }
  }
  double tot[3][2];
  const double* const arr[3] = {slots1, slots2, bslots2};
  slot_totals<3>(arr, tot);   // (its barrier also publishes the weight images)
  const LnStat st1 = ln_stat(tot[0], count1), st2 = ln_stat(tot[1], count2);
  const float mq = (float)(tot[2][0] / count2);                               // mean(q2)
  const float cq = st2.sigma > 0.f ? (float)(tot[2][1] / count2) / st2.sigma : 0.f;  // sum(q2 xhat2) / (N sigma2)
  const float4 g4 = ld_f4(g1 + 4 * c4), be4 = ld_f4(be1 + 4 * c4);
  float dW3[H], dW2[H];
#pragma unroll
  for (int k = 0; k < H; ++k) { dW3[k] = 0.f; dW2[k] = 0.f; }
  float db3 = 0.f, db2 = 0.f;
  float dg4[4] = {0.f, 0.f, 0.f, 0.f}, dbe4[4] = {0.f, 0.f, 0.f, 0.f};   // channels 4 c4 + i, this lane's rows (r, r + 4, ...) only
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * S_WAVES + wave; b < B; b += gridDim.x * S_WAVES) {
    // LNg2 backward: du1 = (q2 - mean(q2)) / s - xhat2 * sum(q2 xhat2) / (N sigma)
    const float xh2 = (u1[(size_t)b * H + j] - st2.mean) / st2.s;
    const float du1 = (q2[(size_t)b * H + j] - mq) / st2.s - xh2 * cq;
    db3 += du1;
    sh[j] = du1;
    sy[j] = z[(size_t)b * H + j];
    __builtin_amdgcn_wave_barrier();
    axpy64(sy, du1, dW3);
    float dz = 0.f;
#pragma unroll 16
    for (int k = 0; k < H; ++k) dz += sh[k] * W3s[k * (H + 1) + j];
    __builtin_amdgcn_wave_barrier();
    db2 += dz * n;
    sh[j] = dz;
    __builtin_amdgcn_wave_barrier();
    float dy1 = 0.f;  // identical for every node row of this sample
#pragma unroll 16
    for (int k = 0; k < H; ++k) dy1 += sh[k] * W2s[k * (H + 1) + j];
    sy[j] = dy1;
    __builtin_amdgcn_wave_barrier();
    const float4 dy4v = ld_f4(sy + 4 * c4);
    const float dy4[4] = {dy4v.x, dy4v.y, dy4v.z, dy4v.w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {be4.x, be4.y, be4.z, be4.w};
    float ysum[4] = {0.f, 0.f, 0.f, 0.f};
    const float* hs = h1 + (size_t)b * n * H + 4 * c4;
    float* qs = q1 + (size_t)b * n * H + 4 * c4;
    for (int i0 = 0; i0 < n; i0 += 4 * S_RU_B) {   // four rows per load, 16 rows in flight (register budget: two 64-float accumulator rows per thread)
      float4 hv[S_RU_B];
#pragma unroll
      for (int u = 0; u < S_RU_B; ++u) {
        const int row = i0 + 4 * u + r;
        hv[u] = ld_f4(hs + (size_t)(row < n ? row : n - 1) * H);
      }
#pragma unroll
      for (int u = 0; u < S_RU_B; ++u) {
        const int row = i0 + 4 * u + r;
        if (row < n) {
          const float hh[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w};
          float q[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float xh = (hh[i] - st1.mean) / st1.s;
            const float pre = xh * gg[i] + bb[i];
            ysum[i] += fmaxf(pre, 0.f);
            const float dpre = pre > 0.f ? dy4[i] : 0.f;
            dg4[i] += dpre * xh;
            dbe4[i] += dpre;
            q[i] = dpre * gg[i];
            s0 += q[i];
            s1 += (double)q[i] * xh;
          }
          st_f4(qs + (size_t)row * H, make_float4(q[0], q[1], q[2], q[3]));
        }
      }
    }
    const float4 ys4 = quad_fold(make_float4(ysum[0], ysum[1], ysum[2], ysum[3]));
    __builtin_amdgcn_wave_barrier();
    if (r == 0) st_f4(sy + 4 * c4, ys4);
    __builtin_amdgcn_wave_barrier();
    axpy64(sy, dz, dW2);
    __builtin_amdgcn_wave_barrier();
  }
  block_put_stats<S_WAVES>(s0, s1, bslots1);
  // dgamma / dbeta: lanes hold channel quads over their own rows -> totals over the four row groups -> thread j = channel j
  const float4 dgt = quad_fold(make_float4(dg4[0], dg4[1], dg4[2], dg4[3])), dbt = quad_fold(make_float4(dbe4[0], dbe4[1], dbe4[2], dbe4[3]));
  __builtin_amdgcn_wave_barrier();
  if (r == 0) { st_f4(sh + 4 * c4, dgt); st_f4(sy + 4 * c4, dbt); }
  __builtin_amdgcn_wave_barrier();
  const float dg = sh[j], dbe = sy[j];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * P2;
  fold_rows64<S_WAVES>(dW3, W3s, out);
  fold_rows64<S_WAVES>(dW2, W2s, out + H * H + H);
  const float sc[4] = {db3, db2, dg, dbe};
  float* const outs[4] = {out + H * H, out + 2 * H * H + H, out + 2 * H * H + 2 * H, out + 2 * H * H + 3 * H};
  fold_scalars<4, S_WAVES>(sc, W3s, outs);
}

// ---- backward 1, rows flattened in quads (16 waves per workgroup)
__global__ __launch_bounds__(64 * ROW_WAVES) void ds_bwd1_quad(const float* __restrict__ x, const float* __restrict__ h1,
                                                        const double* __restrict__ slots1, double count1,
                                                        const float* __restrict__ q1, const double* __restrict__ bslots1,
                                                        float* __restrict__ partial, int B, int n, int d) {
  __shared__ float red[ROW_WAVES / 2][DMAX + 1][H];
  __shared__ float xs_raw[ROW_WAVES * QUADS_IN_FLIGHT_B * XS_LINE];
  const int lane = threadIdx.x & 63, r = lane >> 4, c4 = lane & 15, wave = threadIdx.x >> 6;
  double tot[2][2];
  const double* const arr[2] = {slots1, bslots1};
  slot_totals<2>(arr, tot);
  const LnStat st = ln_stat(tot[0], count1);
  const float mq = (float)(tot[1][0] / count1);
  const float cq = st.sigma > 0.f ? (float)(tot[1][1] / count1) / st.sigma : 0.f;
  float dW1[4][DMAX];   // channels 4 c4 + i, this lane's rows only (slots k >= d collect products with stray finite inputs and are dropped)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < DMAX; ++k) dW1[i][k] = 0.f;
  float db1[4] = {0.f, 0.f, 0.f, 0.f};
  float (*xs)[XS_LINE] = reinterpret_cast<float (*)[XS_LINE]>(xs_raw + wave * QUADS_IN_FLIGHT_B * XS_LINE);   // as in forward 1
  if (lane < XS_LINE - 64)
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT_B; ++u) xs[u][64 + lane] = 0.f;
  const long long rows = (long long)B * n, quads = (rows + 3) >> 2, last = rows * d - 1;
  const long long stride = (long long)gridDim.x * ROW_WAVES;
  for (long long q0 = (long long)blockIdx.x * ROW_WAVES + wave; q0 < quads; q0 += QUADS_IN_FLIGHT_B * stride) {
    float xq[QUADS_IN_FLIGHT_B];
    float4 hv[QUADS_IN_FLIGHT_B], qv[QUADS_IN_FLIGHT_B];
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT_B; ++u) {
      const long long q = q0 + u * stride;
      const long long e = q * 4 * d + lane;
      xq[u] = x[e < last ? e : last];
      long long row = 4 * q + r;
      row = row < rows ? row : rows - 1;      // clamped, masked below
      hv[u] = ld_f4(h1 + row * H + 4 * c4);
      qv[u] = ld_f4(q1 + row * H + 4 * c4);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT_B; ++u) xs[u][lane] = xq[u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < QUADS_IN_FLIGHT_B; ++u) {
      const long long q = q0 + u * stride;
      if (q >= quads) break;
      const float m = 4 * q + r < rows ? 1.f : 0.f;
      const float hh[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w}, qq[4] = {qv[u].x, qv[u].y, qv[u].z, qv[u].w};
      float dh[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (hh[i] - st.mean) / st.s;
        dh[i] = m * ((qq[i] - mq) / st.s - xh * cq);
        db1[i] += dh[i];
      }
      const float* xr = xs[u] + r * d;
#pragma unroll
      for (int k = 0; k < DMAX; ++k) {
        const float xv = xr[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) dW1[i][k] += dh[i] * xv;
      }
    }
  }
  // totals over the four row groups of lanes, then the fixed-order fold of the sixteen waves: upper half -> LDS -> added by the lower
  // half -> LDS -> wave 0 (thread j = channel j)
#pragma unroll
  for (int k = 0; k < DMAX; ++k) {
    const float4 t = quad_fold(make_float4(dW1[0][k], dW1[1][k], dW1[2][k], dW1[3][k]));
    dW1[0][k] = t.x; dW1[1][k] = t.y; dW1[2][k] = t.z; dW1[3][k] = t.w;
  }
  {
    const float4 t = quad_fold(make_float4(db1[0], db1[1], db1[2], db1[3]));
    db1[0] = t.x; db1[1] = t.y; db1[2] = t.z; db1[3] = t.w;
  }
  constexpr int HW = ROW_WAVES / 2;
  if (wave >= HW && r == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k) red[wave - HW][k][4 * c4 + i] = dW1[i][k];
      red[wave - HW][DMAX][4 * c4 + i] = db1[i];
    }
  }
  __syncthreads();
  if (wave < HW && r == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k) dW1[i][k] += red[wave][k][4 * c4 + i];
      db1[i] += red[wave][DMAX][4 * c4 + i];
    }
  }
  __syncthreads();
  if (wave < HW && r == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k) red[wave][k][4 * c4 + i] = dW1[i][k];
      red[wave][DMAX][4 * c4 + i] = db1[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int j = lane;
    float* out = partial + (size_t)blockIdx.x * (H * d + H);
    for (int k = 0; k < d; ++k) {
      float t = red[0][k][j];
#pragma unroll
      for (int w_ = 1; w_ < HW; ++w_) t += red[w_][k][j];
      out[j * d + k] = t;
    }
    float t = red[0][DMAX][j];
#pragma unroll
    for (int w_ = 1; w_ < HW; ++w_) t += red[w_][DMAX][j];
    out[H * d + j] = t;
  }
}

int ds_blocks(long long units) {   // one wave per unit (sample or row), four waves per workgroup, at most NSLOT workgroups
  const long long b = (units + DS_WAVES - 1) / DS_WAVES;
  return b < 1 ? 1 : (b < NSLOT ? (int)b : NSLOT);
}

}  // namespace

extern "C" {

int grl_deepsets_blocks(int batch) { return ds_blocks(batch); }
int grl_deepsets_stat_slots(void) { return NSLOT; }
int grl_deepsets_partial3() { return P3; }
int grl_deepsets_partial2() { return P2; }

// Forward, stage k of 3.  slots1/slots2: fp64[grl_deepsets_stat_slots()][2], fully written by the producing stage (no
// initialisation needed); in a data-parallel run the caller all-reduces them (sum) between the stages and passes the GLOBAL
// element counts count1 = B_glob*n*64, count2 = B_glob*64.
// The *_groups forms run `groups` independent batches in one launch each (grid.y = group): x [groups][batch][n_nodes][d], every group with
// its own whole-tensor LayerNorm statistics (slot arrays [groups][slots][2]) -- the critic pass over the T + 1 frames of a rollout
// (gnn_vf_net.py:72-80 loops over the time steps: statistics per time step).  The per-group grid is the single-batch grid, so each group's
// result is bitwise the one a separate call gives.
int grl_deepsets_fwd1_groups(const float* x, const float* W1, const float* b1, float* h1, double* slots1, int batch, int n_nodes, int d,
                             int groups, hipStream_t stream) {
  if (d > DMAX) return -2;
  if (groups < 1 || groups > 65535) return -5;
  const long long rows = (long long)batch * n_nodes;
  if (n_nodes >= QUAD_MIN_N) {
    const long long wq = (rows + 4 * ROW_WAVES * QUADS_IN_FLIGHT - 1) / (4 * ROW_WAVES * QUADS_IN_FLIGHT);
    hipLaunchKernelGGL(ds_fwd1_quad, dim3(wq < 1 ? 1 : (wq < NSLOT ? (int)wq : NSLOT), groups), dim3(64 * ROW_WAVES), 0, stream, x, W1, b1,
                       h1, slots1, batch, n_nodes, d);
    GRL_CHECK_LAUNCH();
    return 0;
  }
  const long long wg = (rows + ROW_WAVES * ROWS_IN_FLIGHT - 1) / (ROW_WAVES * ROWS_IN_FLIGHT);
  hipLaunchKernelGGL(ds_fwd1, dim3(wg < 1 ? 1 : (wg < NSLOT ? (int)wg : NSLOT), groups), dim3(64 * ROW_WAVES), 0, stream, x, W1, b1, h1,
                     slots1, batch, n_nodes, d);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_fwd2_groups(const float* h1, const double* slots1, double count1, const float* g1, const float* be1, const float* W2,
                             const float* b2, const float* W3, const float* b3, float* z, float* u1, double* slots2, int batch,
                             int n_nodes, int groups, hipStream_t stream) {
  if (groups < 1 || groups > 65535) return -5;
  if (n_nodes >= QUAD_MIN_N)
    hipLaunchKernelGGL(ds_fwd2_quad, dim3(ds_blocks(batch), groups), dim3(64 * S_WAVES), 0, stream, h1, slots1, count1, g1, be1, W2, b2, W3,
                       b3, z, u1, slots2, batch, n_nodes);
  else if (batch >= LANE8_MIN_BATCH)   // the workgroup count is capped (statistic slots): eight waves per workgroup, two samples per wave
    hipLaunchKernelGGL(ds_fwd2<8>, dim3(ds_blocks(batch), groups), dim3(64 * 8), 0, stream, h1, slots1, count1, g1, be1, W2, b2, W3, b3,
                       z, u1, slots2, batch, n_nodes);
  else
    hipLaunchKernelGGL(ds_fwd2<DS_WAVES>, dim3(ds_blocks(batch), groups), dim3(64 * DS_WAVES), 0, stream, h1, slots1, count1, g1, be1, W2, b2,
                       W3, b3, z, u1, slots2, batch, n_nodes);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_fwd3_groups(const float* u1, const double* slots2, double count2, const float* g2, const float* be2, const float* W4,
                             const float* b4, const float* wv, const float* bv, float* value, int batch, int groups, hipStream_t stream) {
  if (groups < 1 || groups > 65535) return -5;
  hipLaunchKernelGGL(ds_fwd3, dim3(ds_blocks(batch), groups), dim3(64 * DS_WAVES), 0, stream, u1, slots2, count2, g2, be2, W4, b4, wv, bv,
                     value, batch);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_fwd1(const float* x, const float* W1, const float* b1, float* h1, double* slots1, int batch, int n_nodes, int d,
                      hipStream_t stream) {
  return grl_deepsets_fwd1_groups(x, W1, b1, h1, slots1, batch, n_nodes, d, 1, stream);
}
int grl_deepsets_fwd2(const float* h1, const double* slots1, double count1, const float* g1, const float* be1, const float* W2,
                      const float* b2, const float* W3, const float* b3, float* z, float* u1, double* slots2, int batch,
                      int n_nodes, hipStream_t stream) {
  return grl_deepsets_fwd2_groups(h1, slots1, count1, g1, be1, W2, b2, W3, b3, z, u1, slots2, batch, n_nodes, 1, stream);
}
int grl_deepsets_fwd3(const float* u1, const double* slots2, double count2, const float* g2, const float* be2, const float* W4,
                      const float* b4, const float* wv, const float* bv, float* value, int batch, hipStream_t stream) {
  return grl_deepsets_fwd3_groups(u1, slots2, count2, g2, be2, W4, b4, wv, bv, value, batch, 1, stream);
}

// Backward stages (reverse order).  bslots1/bslots2: slot arrays like the forward's; all-reduced between stages when data
// parallel.  Every partial slab has grl_deepsets_blocks(batch) rows.
int grl_deepsets_bwd3(const float* u1, const double* slots2, double count2, const float* g2, const float* be2, const float* W4,
                      const float* b4, const float* wv, const float* dvalue, float* q2, double* bslots2, float* partial,
                      int batch, hipStream_t stream) {
  hipLaunchKernelGGL(ds_bwd3, dim3(ds_blocks(batch)), dim3(64 * DS_WAVES), 0, stream, u1, slots2, count2, g2, be2, W4, b4, wv, dvalue,
                     q2, bslots2, partial, batch);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_bwd2(const float* h1, const double* slots1, double count1, const float* g1, const float* be1, const float* W2,
                      const float* W3, const float* z, const float* u1, const double* slots2, double count2, const float* q2,
                      const double* bslots2, float* q1, double* bslots1, float* partial, int batch, int n_nodes,
                      hipStream_t stream) {
  if (n_nodes >= QUAD_MIN_N)
    hipLaunchKernelGGL(ds_bwd2_quad, dim3(ds_blocks(batch)), dim3(64 * S_WAVES), 0, stream, h1, slots1, count1, g1, be1, W2, W3, z, u1,
                       slots2, count2, q2, bslots2, q1, bslots1, partial, batch, n_nodes);
  else if (batch >= LANE8_MIN_BATCH)
    hipLaunchKernelGGL(ds_bwd2<8>, dim3(ds_blocks(batch)), dim3(64 * 8), 0, stream, h1, slots1, count1, g1, be1, W2, W3, z, u1,
                       slots2, count2, q2, bslots2, q1, bslots1, partial, batch, n_nodes);
  else
    hipLaunchKernelGGL(ds_bwd2<DS_WAVES>, dim3(ds_blocks(batch)), dim3(64 * DS_WAVES), 0, stream, h1, slots1, count1, g1, be1, W2, W3, z, u1,
                       slots2, count2, q2, bslots2, q1, bslots1, partial, batch, n_nodes);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_bwd1(const float* x, const float* h1, const double* slots1, double count1, const float* q1,
                      const double* bslots1, float* partial, int batch, int n_nodes, int d, hipStream_t stream) {
  if (d > DMAX) return -2;
  if (n_nodes >= QUAD_MIN_N)
    hipLaunchKernelGGL(ds_bwd1_quad, dim3(ds_blocks(batch)), dim3(64 * ROW_WAVES), 0, stream, x, h1, slots1, count1, q1, bslots1, partial,
                       batch, n_nodes, d);
  else
    hipLaunchKernelGGL(ds_bwd1, dim3(ds_blocks(batch)), dim3(64 * ROW_WAVES), 0, stream, x, h1, slots1, count1, q1, bslots1, partial,
                       batch, n_nodes, d);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
