// DeepSets critic (reference deepsets.py:34-53, gnn_vf_net.py:50-86) with PyG's LayerNorm(mode="graph"):
//   h1 = Lin1(x)            [B, n, 64]      x: [B, n, d]  (d = n_types + 3 n_vec <= 16)
//   y1 = relu(LNg(h1))                      LNg: (h - mean_all) / (std_all + 1e-5) * gamma + beta  -- statistics over the
//   h2 = Lin2(y1); z = sum_n h2             whole tensor, i.e. over the whole (global) minibatch
//   u1 = Lin3(z); y2 = relu(LNg(u1)); u2 = Lin4(y2); V = Linv(u2)
// The whole-tensor statistics split the pass into three launches forward and three backward; each launch leaves the
// sums the next one needs in a tiny fp64 buffer (which is also what a data-parallel run all-reduces between launches).
// One wave per sample; thread j owns feature j; weights live in LDS; weight gradients accumulate in registers over a
// grid-stride loop and leave as per-block partial rows.
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
GRL_DEVINL double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Wave sums -> one pair of fp64 atomics per WORKGROUP: same-address atomics serialise in L2 (~15 ns each, measured: 8192 of
// them cost more than the kernel they ended), so their number, not the arithmetic, decides these kernels' run time.
template <int WAVES>
GRL_DEVINL void block_add_stats(double s0, double s1, double* __restrict__ out) {
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
    atomicAdd(out, t0);
    atomicAdd(out + 1, t1);
  }
}

struct LnStat { float mean, s, sigma; };  // s = sigma + eps
GRL_DEVINL LnStat ln_stat(const double* sums, double count) {
  const double m = sums[0] / count;
  double var = sums[1] / count - m * m;
  if (var < 0) var = 0;
  LnStat st;
  st.mean = (float)m;
  st.sigma = (float)sqrt(var);
  st.s = st.sigma + LNG_EPS;
  return st;
}

// ---- forward 1: h1 = x W1^T + b1 ; stats1 += (sum h1, sum h1^2).  Rows (sample, node) are independent: 4 waves per workgroup,
//      grid-stride over the flattened rows (one wave per sample left every load latency exposed: 1 wave per SIMD).
constexpr int DS_WAVES = 4;
__global__ __launch_bounds__(64 * DS_WAVES) void ds_fwd1(const float* __restrict__ x, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, float* __restrict__ h1,
                                                        double* __restrict__ stats, int B, int n, int d) {
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) w[k] = k < d ? W1[j * d + k] : 0.f;
  const float bj = b1[j];
  double s0 = 0, s1 = 0;
  // four rows in flight per wave; a row's d inputs arrive with ONE load (lane k < d fetches x[row][k]) and are handed round
  // with v_readlane -- d dependent broadcast loads per row left the kernel latency-bound
  const long long rows = (long long)B * n;
  const long long stride = (long long)gridDim.x * DS_WAVES;
  for (long long row0 = blockIdx.x * DS_WAVES + wave; row0 < rows; row0 += 4 * stride) {
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row = row0 + u * stride;
      xv[u] = (row < rows && j < d) ? x[row * d + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row = row0 + u * stride;
      if (row >= rows) break;
      float acc = bj;
#pragma unroll
      for (int k = 0; k < DMAX; ++k)
        if (k < d) acc += __shfl(xv[u], k, 64) * w[k];
      h1[row * H + j] = acc;
      s0 += acc;
      s1 += (double)acc * acc;
    }
  }
  block_add_stats<DS_WAVES>(s0, s1, stats);
}

// ---- forward 2: y1 = relu(LNg(h1)); h2 = y1 W2^T + b2; z = sum_n h2; u1 = z W3^T + b3; stats2 += (sum u1, sum u1^2)
//      Thread j keeps row j of W2 and of W3 in registers; the activation vector is exchanged through a wave-private LDS line and
//      read back as float4 broadcasts (16 LDS reads per 64-term dot product instead of 128).  4 waves (samples) per workgroup.
GRL_DEVINL float dot64(const float* __restrict__ ys /*LDS, 16-byte aligned*/, const float (&w)[H]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 y = *reinterpret_cast<const float4*>(ys + k);
    a0 = fmaf(y.x, w[k], a0); a1 = fmaf(y.y, w[k + 1], a1); a2 = fmaf(y.z, w[k + 2], a2); a3 = fmaf(y.w, w[k + 3], a3);
  }
  return (a0 + a1) + (a2 + a3);
}
__global__ __launch_bounds__(64 * DS_WAVES) void ds_fwd2(const float* __restrict__ h1, const double* __restrict__ stats1, double count1,
                                                        const float* __restrict__ g1, const float* __restrict__ be1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        const float* __restrict__ W3, const float* __restrict__ b3,
                                                        float* __restrict__ z, float* __restrict__ u1, double* __restrict__ stats2,
                                                        int B, int n) {
  __shared__ __attribute__((aligned(16))) float ys_all[DS_WAVES][2][H];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float (*ys)[H] = ys_all[wave];
  float w2[H], w3[H];
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 a = *reinterpret_cast<const float4*>(W2 + j * H + k), c = *reinterpret_cast<const float4*>(W3 + j * H + k);
    w2[k] = a.x; w2[k + 1] = a.y; w2[k + 2] = a.z; w2[k + 3] = a.w;
    w3[k] = c.x; w3[k + 1] = c.y; w3[k + 2] = c.z; w3[k + 3] = c.w;
  }
  const LnStat st = ln_stat(stats1, count1);
  const float gj = g1[j], bj = be1[j], b2j = b2[j], b3j = b3[j];
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * DS_WAVES + wave; b < B; b += gridDim.x * DS_WAVES) {
    float zj = 0.f;
    const float* hrow = h1 + (size_t)b * n * H + j;
    float hv = hrow[0];
    for (int i = 0; i < n; ++i) {   // wave-private double-buffered line: no barrier, LDS ops of a wave complete in order
      const float hn = i + 1 < n ? hrow[(size_t)(i + 1) * H] : 0.f;
      float* line = ys[i & 1];
      line[j] = fmaxf((hv - st.mean) / st.s * gj + bj, 0.f);
      __builtin_amdgcn_wave_barrier();
      zj += b2j + dot64(line, w2);
      hv = hn;
    }
    z[(size_t)b * H + j] = zj;
    float* line = ys[n & 1];
    line[j] = zj;
    __builtin_amdgcn_wave_barrier();
    const float acc = b3j + dot64(line, w3);
    u1[(size_t)b * H + j] = acc;
    s0 += acc;
    s1 += (double)acc * acc;
  }
  block_add_stats<DS_WAVES>(s0, s1, stats2);
}

// ---- forward 3: y2 = relu(LNg(u1)); u2 = y2 W4^T + b4; V = u2 . wv + bv
__global__ __launch_bounds__(64) void ds_fwd3(const float* __restrict__ u1, const double* __restrict__ stats2, double count2,
                                             const float* __restrict__ g2, const float* __restrict__ be2,
                                             const float* __restrict__ W4, const float* __restrict__ b4,
                                             const float* __restrict__ wv, const float* __restrict__ bv, float* __restrict__ value,
                                             int B) {
  __shared__ float W4s[H * (H + 1)], ys[H];
  const int j = threadIdx.x;
  for (int i = j; i < H * H; i += 64) W4s[(i / H) * (H + 1) + (i % H)] = W4[i];
  __syncthreads();
  const LnStat st = ln_stat(stats2, count2);
  const float gj = g2[j], bj = be2[j], b4j = b4[j], wvj = wv[j], bv0 = bv[0];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    ys[j] = fmaxf((u1[(size_t)b * H + j] - st.mean) / st.s * gj + bj, 0.f);
    __syncthreads();
    float acc = b4j;
#pragma unroll 16
    for (int k = 0; k < H; ++k) acc += ys[k] * W4s[j * (H + 1) + k];
    const float v = wave_sum(acc * wvj);
    if (j == 0) value[b] = v + bv0;
    __syncthreads();
  }
}

// ---- backward 3: from dV.  Writes q2 = dy2 * relu'(.) * gamma2 (the LNg input-gradient numerator) and accumulates
//      bstats2 += (sum q2, sum q2 * xhat2); weight grads of Lin4, Linv and LNg2 affine.
// partial row: [dW4 64x64 | db4 64 | dwv 64 | dbv 1 | dg2 64 | dbe2 64]
constexpr int P3 = H * H + H + H + 1 + H + H;
__global__ __launch_bounds__(64) void ds_bwd3(const float* __restrict__ u1, const double* __restrict__ stats2, double count2,
                                             const float* __restrict__ g2, const float* __restrict__ be2,
                                             const float* __restrict__ W4, const float* __restrict__ b4,
                                             const float* __restrict__ wv, const float* __restrict__ dvalue,
                                             float* __restrict__ q2, double* __restrict__ bstats2, float* __restrict__ partial,
                                             int B) {
  __shared__ float W4s[H * (H + 1)], ys[H], du[H];
  const int j = threadIdx.x;
  for (int i = j; i < H * H; i += 64) W4s[(i / H) * (H + 1) + (i % H)] = W4[i];
  __syncthreads();
  const LnStat st = ln_stat(stats2, count2);
  const float gj = g2[j], bj = be2[j], b4j = b4[j], wvj = wv[j];
  float dW4[H];
#pragma unroll
  for (int k = 0; k < H; ++k) dW4[k] = 0.f;
  float db4 = 0.f, dwv = 0.f, dbv = 0.f, dg = 0.f, dbe = 0.f;
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float xh = (u1[(size_t)b * H + j] - st.mean) / st.s;
    const float pre = xh * gj + bj;
    const float y = fmaxf(pre, 0.f);
    ys[j] = y;
    __syncthreads();
    float u2 = b4j;
#pragma unroll 16
    for (int k = 0; k < H; ++k) u2 += ys[k] * W4s[j * (H + 1) + k];
    const float dv = dvalue[b];
    const float du2 = dv * wvj;
    dwv += dv * u2;
    dbv += dv;
    db4 += du2;
#pragma unroll
    for (int k = 0; k < H; ++k) dW4[k] += du2 * ys[k];
    du[j] = du2;
    __syncthreads();
    float dy = 0.f;
#pragma unroll 16
    for (int k = 0; k < H; ++k) dy += du[k] * W4s[k * (H + 1) + j];
    const float dpre = pre > 0.f ? dy : 0.f;
    dg += dpre * xh;
    dbe += dpre;
    const float q = dpre * gj;
    q2[(size_t)b * H + j] = q;
    s0 += q;
    s1 += (double)q * xh;
    __syncthreads();
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (j == 0) { atomicAdd(bstats2, s0); atomicAdd(bstats2 + 1, s1); }
  float* out = partial + (size_t)blockIdx.x * P3;
#pragma unroll
  for (int k = 0; k < H; ++k) out[j * H + k] = dW4[k];
  out[H * H + j] = db4;
  out[H * H + H + j] = dwv;
  if (j == 0) out[H * H + 2 * H] = dbv;
  out[H * H + 2 * H + 1 + j] = dg;
  out[H * H + 3 * H + 1 + j] = dbe;
}

// ---- backward 2: du1 (LNg2 backward) -> Lin3 -> dz; dz fans out to every node row of the sample -> Lin2 -> dy1 (same for
//      all rows) -> q1 = dy1 * relu'(.) * gamma1 written per row; bstats1 += (sum q1, sum q1 xhat1).
// partial row: [dW3 64x64 | db3 64 | dW2 64x64 | db2 64 | dg1 64 | dbe1 64]
constexpr int P2 = H * H + H + H * H + H + H + H;
__global__ __launch_bounds__(64 * DS_WAVES) void ds_bwd2(const float* __restrict__ h1, const double* __restrict__ stats1, double count1,
                                                        const float* __restrict__ g1, const float* __restrict__ be1,
                                                        const float* __restrict__ W2, const float* __restrict__ W3,
                                                        const float* __restrict__ z, const float* __restrict__ u1,
                                                        const double* __restrict__ stats2, double count2,
                                                        const float* __restrict__ q2, const double* __restrict__ bstats2,
                                                        float* __restrict__ q1, double* __restrict__ bstats1,
                                                        float* __restrict__ partial, int B, int n) {
  // 4 waves (samples) per workgroup share the weight images; every wave keeps its own gradient accumulators in registers and
  // the four sets are folded through LDS at the end, so the workgroup still leaves ONE partial row
  __shared__ __attribute__((aligned(16))) float W2s[H * (H + 1)], W3s[H * (H + 1)], sh_all[DS_WAVES][H], sy_all[DS_WAVES][H];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sh = sh_all[wave];
  float* sy = sy_all[wave];
  for (int i = threadIdx.x; i < H * H; i += 64 * DS_WAVES) {
    W2s[(i / H) * (H + 1) + (i % H)] = W2[i];
    W3s[(i / H) * (H + 1) + (i % H)] = W3[i];
  }
  __syncthreads();
  const LnStat st1 = ln_stat(stats1, count1), st2 = ln_stat(stats2, count2);
  const float mq = (float)(bstats2[0] / count2);                               // mean(q2)
  const float cq = st2.sigma > 0.f ? (float)(bstats2[1] / count2) / st2.sigma : 0.f;  // sum(q2 xhat2) / (N sigma2)
  const float gj = g1[j], bj = be1[j];
  float dW3[H], dW2[H];
#pragma unroll
  for (int k = 0; k < H; ++k) { dW3[k] = 0.f; dW2[k] = 0.f; }
  float db3 = 0.f, db2 = 0.f, dg = 0.f, dbe = 0.f;
  double s0 = 0, s1 = 0;
  for (int b = blockIdx.x * DS_WAVES + wave; b < B; b += gridDim.x * DS_WAVES) {
    // LNg2 backward: du1 = (q2 - mean(q2)) / s - xhat2 * sum(q2 xhat2) / (N sigma)
    const float xh2 = (u1[(size_t)b * H + j] - st2.mean) / st2.s;
    const float du1 = (q2[(size_t)b * H + j] - mq) / st2.s - xh2 * cq;
    db3 += du1;
    sh[j] = du1;
    sy[j] = z[(size_t)b * H + j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < H; ++k) dW3[k] += du1 * sy[k];
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
    for (int i = 0; i < n; ++i) {
      const size_t row = (size_t)b * n + i;
      const float xh = (h1[row * H + j] - st1.mean) / st1.s;
      const float pre = xh * gj + bj;
      ysum += fmaxf(pre, 0.f);
      const float dpre = pre > 0.f ? dy1 : 0.f;
      dg += dpre * xh;
      dbe += dpre;
      const float q = dpre * gj;
      q1[row * H + j] = q;
      s0 += q;
      s1 += (double)q * xh;
    }
    sy[j] = ysum;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < H; ++k) dW2[k] += dz * sy[k];
    __builtin_amdgcn_wave_barrier();
  }
  block_add_stats<DS_WAVES>(s0, s1, bstats1);
  // fold the four waves (fixed order) through the weight images' LDS space, then one partial row per workgroup
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * P2;
  for (int pass = 0; pass < 2; ++pass) {
    float* acc = pass == 0 ? W3s : W2s;
    for (int w_ = 0; w_ < DS_WAVES; ++w_) {
      if (wave == w_) {
#pragma unroll
        for (int k = 0; k < H; ++k) {
          const float v = pass == 0 ? dW3[k] : dW2[k];
          acc[j * (H + 1) + k] = w_ == 0 ? v : acc[j * (H + 1) + k] + v;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < H * H; i += 64 * DS_WAVES) {
    out[i] = W3s[(i / H) * (H + 1) + (i % H)];
    out[H * H + H + i] = W2s[(i / H) * (H + 1) + (i % H)];
  }
  __syncthreads();
  float* red = W3s;   // [4 quantities][DS_WAVES][64]
  red[(0 * DS_WAVES + wave) * H + j] = db3;
  red[(1 * DS_WAVES + wave) * H + j] = db2;
  red[(2 * DS_WAVES + wave) * H + j] = dg;
  red[(3 * DS_WAVES + wave) * H + j] = dbe;
  __syncthreads();
  if (wave == 0) {
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      t[q] = red[(q * DS_WAVES) * H + j];
#pragma unroll
      for (int w_ = 1; w_ < DS_WAVES; ++w_) t[q] += red[(q * DS_WAVES + w_) * H + j];
    }
    out[H * H + j] = t[0];
    out[2 * H * H + H + j] = t[1];
    out[2 * H * H + 2 * H + j] = t[2];
    out[2 * H * H + 3 * H + j] = t[3];
  }
}

// ---- backward 1: dh1 (LNg1 backward) -> dW1, db1.   partial row: [dW1 64 x d | db1 64]   (rows flattened, 4 waves per workgroup)
__global__ __launch_bounds__(64 * DS_WAVES) void ds_bwd1(const float* __restrict__ x, const float* __restrict__ h1,
                                                        const double* __restrict__ stats1, double count1,
                                                        const float* __restrict__ q1, const double* __restrict__ bstats1,
                                                        float* __restrict__ partial, int B, int n, int d) {
  __shared__ float red[DS_WAVES][H][DMAX + 1];
  const int j = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const LnStat st = ln_stat(stats1, count1);
  const float mq = (float)(bstats1[0] / count1);
  const float cq = st.sigma > 0.f ? (float)(bstats1[1] / count1) / st.sigma : 0.f;
  float dW1[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) dW1[k] = 0.f;
  float db1 = 0.f;
  const long long rows = (long long)B * n;
  const long long stride = (long long)gridDim.x * DS_WAVES;
  for (long long row0 = blockIdx.x * DS_WAVES + wave; row0 < rows; row0 += 4 * stride) {   // four rows in flight (see ds_fwd1)
    float xv[4], hv[4], qv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row = row0 + u * stride;
      const bool ok = row < rows;
      xv[u] = (ok && j < d) ? x[row * d + j] : 0.f;
      hv[u] = ok ? h1[row * H + j] : 0.f;
      qv[u] = ok ? q1[row * H + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (row0 + u * stride >= rows) break;
      const float xh = (hv[u] - st.mean) / st.s;
      const float dh = (qv[u] - mq) / st.s - xh * cq;
      db1 += dh;
#pragma unroll
      for (int k = 0; k < DMAX; ++k)
        if (k < d) dW1[k] += dh * __shfl(xv[u], k, 64);
    }
  }
#pragma unroll
  for (int k = 0; k < DMAX; ++k) red[wave][j][k] = dW1[k];
  red[wave][j][DMAX] = db1;
  __syncthreads();
  if (wave == 0) {
    float* out = partial + (size_t)blockIdx.x * (H * d + H);
    for (int k = 0; k < d; ++k) {
      float t = red[0][j][k];
#pragma unroll
      for (int w_ = 1; w_ < DS_WAVES; ++w_) t += red[w_][j][k];
      out[j * d + k] = t;
    }
    float t = red[0][j][DMAX];
#pragma unroll
    for (int w_ = 1; w_ < DS_WAVES; ++w_) t += red[w_][j][DMAX];
    out[H * d + j] = t;
  }
}

int ds_blocks(int B) { return B < 1024 ? (B < 1 ? 1 : B) : 1024; }

}  // namespace

extern "C" {

int grl_deepsets_blocks(int batch) { return ds_blocks(batch); }
int grl_deepsets_partial3() { return P3; }
int grl_deepsets_partial2() { return P2; }

// Forward, stage k of 3.  stats1/stats2: fp64[2] zero-initialised by the caller; in a data-parallel run the caller
// all-reduces them (sum) between the stages and passes the GLOBAL element counts count1 = B_glob*n*64, count2 = B_glob*64.
int grl_deepsets_fwd1(const float* x, const float* W1, const float* b1, float* h1, double* stats1, int batch, int n_nodes, int d,
                      hipStream_t stream) {
  if (d > DMAX) return -2;
  hipLaunchKernelGGL(ds_fwd1, dim3(512), dim3(64 * DS_WAVES), 0, stream, x, W1, b1, h1, stats1, batch, n_nodes, d);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_fwd2(const float* h1, const double* stats1, double count1, const float* g1, const float* be1, const float* W2,
                      const float* b2, const float* W3, const float* b3, float* z, float* u1, double* stats2, int batch,
                      int n_nodes, hipStream_t stream) {
  hipLaunchKernelGGL(ds_fwd2, dim3((batch + DS_WAVES - 1) / DS_WAVES < 512 ? (batch + DS_WAVES - 1) / DS_WAVES : 512),
                     dim3(64 * DS_WAVES), 0, stream, h1, stats1, count1, g1, be1, W2, b2, W3, b3, z, u1, stats2, batch, n_nodes);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_fwd3(const float* u1, const double* stats2, double count2, const float* g2, const float* be2, const float* W4,
                      const float* b4, const float* wv, const float* bv, float* value, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(ds_fwd3, dim3(ds_blocks(batch)), dim3(64), 0, stream, u1, stats2, count2, g2, be2, W4, b4, wv, bv, value,
                     batch);
  GRL_CHECK_LAUNCH();
  return 0;
}
// Backward stages (reverse order).  bstats1/bstats2: fp64[2] zero-initialised; all-reduced between stages when data parallel.
int grl_deepsets_bwd3(const float* u1, const double* stats2, double count2, const float* g2, const float* be2, const float* W4,
                      const float* b4, const float* wv, const float* dvalue, float* q2, double* bstats2, float* partial,
                      int batch, hipStream_t stream) {
  hipLaunchKernelGGL(ds_bwd3, dim3(ds_blocks(batch)), dim3(64), 0, stream, u1, stats2, count2, g2, be2, W4, b4, wv, dvalue, q2,
                     bstats2, partial, batch);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_bwd2(const float* h1, const double* stats1, double count1, const float* g1, const float* be1, const float* W2,
                      const float* W3, const float* z, const float* u1, const double* stats2, double count2, const float* q2,
                      const double* bstats2, float* q1, double* bstats1, float* partial, int batch, int n_nodes,
                      hipStream_t stream) {
  hipLaunchKernelGGL(ds_bwd2, dim3(ds_blocks(batch)), dim3(64 * DS_WAVES), 0, stream, h1, stats1, count1, g1, be1, W2, W3, z, u1, stats2,
                     count2, q2, bstats2, q1, bstats1, partial, batch, n_nodes);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_deepsets_bwd1(const float* x, const float* h1, const double* stats1, double count1, const float* q1,
                      const double* bstats1, float* partial, int batch, int n_nodes, int d, hipStream_t stream) {
  if (d > DMAX) return -2;
  hipLaunchKernelGGL(ds_bwd1, dim3(ds_blocks(batch)), dim3(64 * DS_WAVES), 0, stream, x, h1, stats1, count1, q1, bstats1, partial, batch,
                     n_nodes, d);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
