// Training-loop kernels around the loss: fused Adam over a flat parameter buffer (reference examples/torchrl/train.py:145-146,
// 313-316: two torch.optim.Adam(lr, eps=1e-5) -> one launch over the concatenated actor+critic buffer), the shifted GAE
// scan (train.py:134-140,249-251; torchrl GAE(shifted=True, average_gae=False)), global grad-norm clipping
// (train.py:308-310) and the per-sample kNN topology (rigid_tasks_data.py:285-287).
#include "grl_common.h"
#include "grl_feat.h"
#include "grl_report.h"

namespace {

// torch.optim.Adam (no amsgrad, no weight decay): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps).   scale = optional gradient pre-scale (1/world, clip coefficient)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int n, float lr, float b1, float b2, float eps,
                                                  float bc1, float bc2_sqrt, const float* __restrict__ scale_dev,
                                                  float scale_host) {
  const float scale = scale_host * (scale_dev ? scale_dev[0] : 1.f);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// Same update with the step count read from device memory (bias corrections computed in-kernel), so that the launch can be
// recorded once into a hipGraph and replayed: *step_dev is advanced by the caller before each replay.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, int n, const float* __restrict__ lr_dev, float b1,
                                                      float b2, float eps, const int* __restrict__ step_dev,
                                                      const float* __restrict__ scale_dev, float scale_host) {
  const float t = (float)step_dev[0];
  const float lr = lr_dev[0];   // learning rate in device memory: an annealed rate (train.py:264-271) reaches a replayed graph
  const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
  const float scale = scale_host * (scale_dev ? scale_dev[0] : 1.f);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// Data parallel, behind the lane's all-reduce: the Adam update of the reduced slice and, by one extra workgroup, the reported values from the
// ranks' delivered loss records (grl_adam_step_dev + grl_trpl_report_record_pairs in ONE launch).
__global__ __launch_bounds__(256) void adam_dev_report_pairs_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                   float* __restrict__ v, int n, const float* __restrict__ lr_dev, float b1,
                                                                   float b2, float eps, const int* __restrict__ step_dev, int n_adam_blocks,
                                                                   const double* __restrict__ records, int n_records,
                                                                   double* __restrict__ sums, unsigned int* __restrict__ maxes,
                                                                   float entropy_coef, float* __restrict__ out14) {
  if ((int)blockIdx.x == n_adam_blocks) {
    __shared__ double sh[16], part[256];
    trpl_report_body<256, true>(records, n_records, sums, maxes, entropy_coef, out14, sh, part);
    return;
  }
  const float t = (float)step_dev[0];
  const float lr = lr_dev[0];
  const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += n_adam_blocks * blockDim.x) {
    const float gi = g[i] * 1.f;            // (the same arithmetic as adam_dev_kernel with scale 1: bitwise its update)
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// clip coefficient of torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (||g||_2 + 1e-6))
// ONE workgroup, fixed summation order (eight independent 16-byte loads in flight per thread, thread-strided partial sums, wave butterflies,
// the sixteen waves in order): the squared norm -- and with it the clip coefficient -- is bitwise reproducible (until round 3: fp64 atomics
// of up to 1 024 wave sums, in whatever order they arrived).  sqnorm[0] = the sum (written: round 4 -- the slot need not be zeroed), and the
// coefficient comes out of the same launch (its own one-thread kernel was a launch on the step's tail in every clipped configuration).
__global__ __launch_bounds__(1024) void sqnorm_kernel(const float* __restrict__ g, int n, double* __restrict__ out, float max_norm,
                                                     float* __restrict__ coef) {
  __shared__ double red[16];
  double s = 0;
  // views of the flat gradient start anywhere: up to three leading elements, then 16-byte pieces, then the tail
  int head = (int)(((16 - (reinterpret_cast<size_t>(g) & 15)) & 15) >> 2);
  head = head < n ? head : n;
  const int n4 = (n - head) / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g + head);
  if ((int)threadIdx.x < head) s += (double)g[threadIdx.x] * g[threadIdx.x];
  for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * 1024) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * 1024; v[u] = g4[i < n4 ? i : n4 - 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 1024 < n4) s += ((double)v[u].x * v[u].x + (double)v[u].y * v[u].y) + ((double)v[u].z * v[u].z + (double)v[u].w * v[u].w);
  }
  for (int i = head + 4 * n4 + threadIdx.x; i < n; i += 1024) s += (double)g[i] * g[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += red[w];
    out[0] = t;
    coef[0] = fminf(1.f, max_norm / ((float)sqrt(t) + 1e-6f));
  }
}

// One thread per environment, sequential over time (the recursion is inherently serial in t; N envs run in parallel).
// Inputs are [N, T] row-major (values [N, T+1]); a 64x(T) tile is transposed through LDS so global accesses stay coalesced.
constexpr int GAE_TT = 64;
__global__ __launch_bounds__(64) void gae_kernel(const float* __restrict__ reward, const unsigned char* __restrict__ done,
                                                const unsigned char* __restrict__ terminated, const float* __restrict__ values,
                                                float* __restrict__ adv, float* __restrict__ target, int N, int T, float gamma,
                                                float lmbda) {
  __shared__ float s_r[64][GAE_TT + 1], s_v[64][GAE_TT + 2], s_a[64][GAE_TT + 1];
  __shared__ unsigned char s_d[64][GAE_TT + 4], s_t[64][GAE_TT + 4];
  const int env0 = blockIdx.x * 64, lane = threadIdx.x;
  float run = 0.f;
  for (int t1 = T; t1 > 0; t1 -= GAE_TT) {
    const int t0 = max(t1 - GAE_TT, 0), len = t1 - t0;
    // cooperative coalesced load: row e of the tile is env0+e, columns t0..t1 (values: t0..t1 inclusive)
    for (int e = 0; e < 64; ++e) {
      const int env = env0 + e;
      if (env >= N) break;
      if (lane < len) {
        s_r[e][lane] = reward[(size_t)env * T + t0 + lane];
        s_d[e][lane] = done[(size_t)env * T + t0 + lane];
        s_t[e][lane] = terminated[(size_t)env * T + t0 + lane];
      }
      if (lane < len) s_v[e][lane] = values[(size_t)env * (T + 1) + t0 + lane];
      if (lane == 0) s_v[e][len] = values[(size_t)env * (T + 1) + t0 + len];  // the 65th value of a full 64-step tile
    }
    __syncthreads();
    if (env0 + lane < N) {
      for (int k = len - 1; k >= 0; --k) {
        const float nt = s_t[lane][k] ? 0.f : 1.f, nd = s_d[lane][k] ? 0.f : 1.f;
        const float delta = s_r[lane][k] + gamma * nt * s_v[lane][k + 1] - s_v[lane][k];
        run = delta + gamma * lmbda * nd * run;
        s_a[lane][k] = run;
      }
    }
    __syncthreads();
    for (int e = 0; e < 64; ++e) {
      const int env = env0 + e;
      if (env >= N) break;
      if (lane < len) {
        const float a = s_a[e][lane];
        adv[(size_t)env * T + t0 + lane] = a;
        target[(size_t)env * T + t0 + lane] = a + s_v[e][lane];
      }
    }
    __syncthreads();
  }
}

// Brute-force k nearest OTHER points among the first n_valid[b] of P points of each sample (P <= 128).
// out_nbr [B, P, k] (local indices, -1 where fewer than k neighbours exist / point is padding).  Ties: lower index first.
constexpr int KNN_PMAX = 128;
__global__ __launch_bounds__(128) void knn_kernel(const float* __restrict__ pos /*[B,P,3]*/, const int* __restrict__ n_valid,
                                                 int* __restrict__ out_nbr, int B, int P, int k) {
  __shared__ float px[KNN_PMAX], py[KNN_PMAX], pz[KNN_PMAX];
  const int b = blockIdx.x, i = threadIdx.x;
  const int nv = n_valid ? min(n_valid[b], P) : P;
  if (i < P) {
    px[i] = pos[((size_t)b * P + i) * 3];
    py[i] = pos[((size_t)b * P + i) * 3 + 1];
    pz[i] = pos[((size_t)b * P + i) * 3 + 2];
  }
  __syncthreads();
  if (i >= P) return;
  int chosen[8];
  for (int s = 0; s < k; ++s) {
    int best = -1;
    float bd = 3.4e38f;
    if (i < nv) {
      for (int j = 0; j < nv; ++j) {
        if (j == i) continue;
        bool used = false;
        for (int u = 0; u < s; ++u) used |= (chosen[u] == j);
        if (used) continue;
        const float dx = px[j] - px[i], dy = py[j] - py[i], dz = pz[j] - pz[i];
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < bd) { bd = d; best = j; }
      }
    }
    chosen[s] = best;
    out_nbr[((size_t)b * P + i) * k + s] = best;
  }
}

}  // namespace

// ---- node features of the batched graph: the body lives in grl_feat.h (shared with the step's merged head launch, node_ops.hip) ----
__global__ __launch_bounds__(256) void build_features_kernel(FeatDescs all, int* __restrict__ bump) {
  build_features_body(all, bump, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, blockIdx.x == 0 && blockIdx.y == 0);
}

// ---- lane gate as a kernel: one thread waits until flag[0] >= count[0] + add (both device int32), at most timeout_ticks of the 100 MHz wall clock
// -- what hipStreamWaitValue32 does on this runtime as well (its __amd_rocclr_streamOpsWait is a one-thread kernel), but capturable into a
// hipGraph, with the target read from DEVICE memory (a recorded step waits for ITS count) and bounded: a gate is a scheduling hint, so on a
// timeout the lane simply goes on.  One wave on one SIMD; no LDS, a handful of registers.
__global__ __launch_bounds__(64) void wait_flag_ge_kernel(const int* __restrict__ flag, const int* __restrict__ count, int add,
                                                          unsigned long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  const int target = count[0] + add;
  const unsigned long long deadline = wall_clock64() + timeout_ticks;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target < 0) {
    if (wall_clock64() > deadline) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

// ---- several small device-to-device copies in one launch (refreshing the static input buffers of a recorded step) -------------
constexpr int COPY_MAX = 24;
struct CopyJobs {
  void* dst[COPY_MAX];
  const void* src[COPY_MAX];
  long long bytes[COPY_MAX];
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyJobs jobs) {
  const int j = blockIdx.y;
  const long long n = jobs.bytes[j];
  const long long n16 = ((reinterpret_cast<size_t>(jobs.dst[j]) | reinterpret_cast<size_t>(jobs.src[j])) & 15) == 0 ? n >> 4 : 0;
  const uint4* s16 = reinterpret_cast<const uint4*>(jobs.src[j]);
  uint4* d16 = reinterpret_cast<uint4*>(jobs.dst[j]);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) d16[i] = s16[i];
  const unsigned char* s1 = reinterpret_cast<const unsigned char*>(jobs.src[j]);
  unsigned char* d1 = reinterpret_cast<unsigned char*>(jobs.dst[j]);
  for (long long i = (n16 << 4) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    d1[i] = s1[i];
}

// ---- minibatch assembly from a device-resident rollout: rows idx[0..n_rows) of several [R, width] tensors in one launch ---------
// (replaces the reference's CPU replay storage + sampler + per-minibatch H2D copy, examples/torchrl/train.py:120,128,258-261)
struct GatherJobs {
  unsigned int* dst[COPY_MAX];
  const unsigned int* src[COPY_MAX];
  int words[COPY_MAX];   // 4-byte words per row
};
// ``count`` (optional): the index row is picked on the DEVICE -- idx points at a matrix [n_idx_rows, n_rows] and this launch reads its row
// (count[0] - base[0]) mod n_idx_rows: a recorded step gathers "the next minibatch of the epoch" without the host touching its arguments
__global__ __launch_bounds__(256) void gather_rows_many_kernel(GatherJobs jobs, const long long* __restrict__ idx, int n_rows,
                                                               const int* __restrict__ count, const int* __restrict__ base, int n_idx_rows) {
  if (count) {
    int r = (count[0] - base[0]) % n_idx_rows;
    if (r < 0) r += n_idx_rows;
    idx += (long long)r * n_rows;
  }
  const int k = blockIdx.y;
  const int wpr = jobs.words[k];
  const long long total = (long long)n_rows * wpr;
  const unsigned int* __restrict__ src = jobs.src[k];
  unsigned int* __restrict__ dst = jobs.dst[k];
  // the usual case: 32-bit index arithmetic, four independent (index, word) load pairs in flight per thread.  `w0 + u * stride` and
  // `w0 += 4 * stride` must not wrap: the fast path is taken only while total + 8 * stride stays below 2^31
  if (total + 8ll * gridDim.x * blockDim.x < (1ll << 31)) {
    const int tot = (int)total, stride = gridDim.x * blockDim.x;
    for (int w0 = blockIdx.x * blockDim.x + threadIdx.x; w0 < tot; w0 += 4 * stride) {
      long long row[4];
      int c[4];
      unsigned int v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = w0 + u * stride < tot ? w0 + u * stride : tot - 1;   // clamped, not guarded
        const int r = w / wpr;
        c[u] = w - r * wpr;
        row[u] = idx[r];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[row[u] * wpr + c[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (w0 + u * stride < tot) dst[w0 + u * stride] = v[u];
    }
    return;
  }
  for (long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(w / wpr), c = (int)(w - (long long)r * wpr);
    dst[w] = src[idx[r] * wpr + c];
  }
}

// ---- running observation normalisation + clip (collector side: geometry_rl/torchrl/envs/transforms.py:141-163 NDVecNorm on
//      torchrl's VecNorm; configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:47-72) ----------------------------------------------
// x viewed as [rows, K] (K = 3 for the vector groups: one statistic per axis shared by all points and environments; K = D for the
// scalar group).  state = [sum K | ssq K | count 1] (float32 like the reference's tensordict entries):
//   sum = decay*sum + colsum(x); ssq = decay*ssq + colsum(x^2); count = decay*count + rows
//   mean = sum/count; std = sqrt(max(ssq/count - mean^2, eps)); y = clip((x - mean)/max(std, eps), lo, hi)
constexpr int VN_KMAX = 64, VN_BLOCKS = 256;
__global__ __launch_bounds__(256) void vecnorm_partial_kernel(const float* __restrict__ x, long long rows, int K,
                                                              double* __restrict__ partial /*[VN_BLOCKS][2*K]*/) {
  __shared__ double red[256][2];
  // thread t handles column t % K of rows (t / K) + m * rows_per_iter: coalesced along the row-major layout
  const int per = 256 / K;                     // rows covered by one sweep of the workgroup
  const int c = threadIdx.x % K, r0 = threadIdx.x / K;
  double s = 0, q = 0;
  if (r0 < per)
    for (long long r = (long long)blockIdx.x * per + r0; r < rows; r += (long long)gridDim.x * per) {
      const double v = x[r * K + c];
      s += v;
      q += v * v;
    }
  red[threadIdx.x][0] = s;
  red[threadIdx.x][1] = q;
  __syncthreads();
  if (threadIdx.x < K) {
    double ts = 0, tq = 0;
    for (int j = 0; j < per; ++j) { ts += red[j * K + threadIdx.x][0]; tq += red[j * K + threadIdx.x][1]; }
    partial[(size_t)blockIdx.x * 2 * K + threadIdx.x] = ts;
    partial[(size_t)blockIdx.x * 2 * K + K + threadIdx.x] = tq;
  }
}
__global__ void vecnorm_update_kernel(const double* __restrict__ partial, int n_blocks, long long rows, int K, float decay, float eps,
                                      int update, float* __restrict__ state, float* __restrict__ mean_std /*[2*K]*/) {
  const int c = threadIdx.x;   // one wave; lanes >= K idle
  float sum = 0.f, ssq = 0.f, count = 1.f;
  if (c < K) {
    sum = state[c]; ssq = state[K + c]; count = state[2 * K];
    if (update) {
      double ts = 0, tq = 0;
      for (int b = 0; b < n_blocks; ++b) { ts += partial[(size_t)b * 2 * K + c]; tq += partial[(size_t)b * 2 * K + K + c]; }
      sum = sum * decay + (float)ts;
      ssq = ssq * decay + (float)tq;
      count = count * decay + (float)rows;
    }
    const float mean = sum / count;
    const float std_ = sqrtf(fmaxf(ssq / count - mean * mean, eps));
    mean_std[c] = mean;
    mean_std[K + c] = fmaxf(std_, eps);
  }
  __syncthreads();   // every lane has read the old count before lane 0 overwrites it
  if (update && c < K) {
    state[c] = sum;
    state[K + c] = ssq;
    if (c == 0) state[2 * K] = count;
  }
}
__global__ __launch_bounds__(256) void vecnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean_std, long long n,
                                                            int K, float lo, float hi, float* __restrict__ y_norm,
                                                            float* __restrict__ y_clip) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % K);
    const float v = x[i];
    if (y_norm) y_norm[i] = fminf(fmaxf((v - mean_std[c]) / mean_std[K + c], lo), hi);
    if (y_clip) y_clip[i] = fminf(fmaxf(v, lo), hi);
  }
}

// ---- per-kernel timing (off by default; bench.py's roofline leg switches it on for a few steps) ---------------------------------
// mode 1: HIP events around the kernels inside multi-kernel entry points (eagerly issued launches).
// mode 2: one-thread kernels that store the device's wall clock (wall_clock64, constant rate) in front of and behind a launch.  They are
//         ordinary kernel nodes when the step is recorded into a hipGraph, so they are re-executed by every replay: the durations of
//         REPLAYED launches (events cannot do that here: external event records are refused by the HIP runtime torch bundles, DESIGN
//         finding 30).  The interval includes the dispatch of the launch behind the first stamp and of the second stamp: a few us.
#include <string>
#include <vector>
namespace {
struct ProfRec { const char* name; hipEvent_t e0, e1; int slot; };
int g_prof_mode = 0;
std::vector<ProfRec> g_prof;
hipEvent_t g_prof_open = nullptr;
const char* g_prof_open_name = nullptr;
constexpr int PROF_STAMPS = 256;             // stamp pairs per recording
unsigned long long* g_stamps = nullptr;      // device [PROF_STAMPS][2]
int g_stamp_next = 0, g_stamp_open = -1;
double g_wall_khz = 100000.0;
__global__ void prof_stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }
}  // namespace
void grl_prof_begin(const char* name, hipStream_t stream) {
  if (g_prof_mode == 0) return;
  if (g_prof_mode == 2) {
    if (g_stamp_next >= PROF_STAMPS) return;
    g_stamp_open = g_stamp_next++;
    hipLaunchKernelGGL(prof_stamp_kernel, dim3(1), dim3(1), 0, stream, g_stamps + 2 * g_stamp_open);
    g_prof_open_name = name;
    return;
  }
  hipEventCreate(&g_prof_open);
  hipEventRecord(g_prof_open, stream);
  g_prof_open_name = name;
}
void grl_prof_end(hipStream_t stream) {
  if (g_prof_mode == 2) {
    if (g_stamp_open < 0) return;
    hipLaunchKernelGGL(prof_stamp_kernel, dim3(1), dim3(1), 0, stream, g_stamps + 2 * g_stamp_open + 1);
    g_prof.push_back({g_prof_open_name, nullptr, nullptr, g_stamp_open});
    g_stamp_open = -1;
    return;
  }
  if (g_prof_mode == 0 || !g_prof_open) return;
  hipEvent_t e1;
  hipEventCreate(&e1);
  hipEventRecord(e1, stream);
  g_prof.push_back({g_prof_open_name, g_prof_open, e1, -1});
  g_prof_open = nullptr;
}
// wrappers of single-kernel entry points: only the replay mode needs them (eagerly, the caller's events around the entry point do)
void grl_prof_begin_replay(const char* name, hipStream_t stream) { if (g_prof_mode == 2) grl_prof_begin(name, stream); }
void grl_prof_end_replay(hipStream_t stream) { if (g_prof_mode == 2) grl_prof_end(stream); }

extern "C" {

// ABI version of this library: major * 10000 + minor * 100 + patch.  Bumped whenever a declared signature changes
// (include/grl_hip.h GRL_HIP_VERSION must agree: geometry_rl_amd/hip.py checks it at load time).
int grl_version(void) { return 205; }
// The current stream waits (a one-thread kernel: capturable) until flag[0] >= count[0] + add, at most timeout_us microseconds.
int grl_wait_flag_ge(const int* flag, const int* count, int add, int timeout_us, hipStream_t stream) {
  if (!flag || !count) return -2;
  hipLaunchKernelGGL(wait_flag_ge_kernel, dim3(1), dim3(64), 0, stream, flag, count, add,
                     (unsigned long long)(timeout_us > 0 ? timeout_us : 200000) * 100ull);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_can_stream_wait_value(void) {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) return 0;
  return v ? 1 : 0;
}

// step = 1-based Adam step count.  scale_dev: optional device scalar multiplied into the gradient (clip coefficient).
int grl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1, float beta2,
                  float eps, int step, const float* scale_dev, float scale_host, hipStream_t stream) {
  if (n <= 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                     bc1, bc2s, scale_dev, scale_host);
  GRL_CHECK_LAUNCH();
  return 0;
}

// step_dev: device int[1] holding the (1-based) optimizer step of THIS update; lr_dev: device float[1], the learning rate of THIS
// update (both are read by the kernel, so a recorded launch follows the host's schedule without being re-recorded)
int grl_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, const float* lr_dev, float beta1,
                      float beta2, float eps, const int* step_dev, const float* scale_dev, float scale_host, hipStream_t stream) {
  if (n <= 0) return 0;
  if (!lr_dev || !step_dev) return -2;
  const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, n, lr_dev, beta1,
                     beta2, eps, step_dev, scale_dev, scale_host);
  GRL_CHECK_LAUNCH();
  return 0;
}

int grl_adam_report_record_pairs(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, const float* lr_dev, float beta1,
                                 float beta2, float eps, const int* step_dev, const float* region, int n_records, double* sums,
                                 unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream) {
  if (n <= 0 || !lr_dev || !step_dev || !region || !sums || !maxes || !out14 || n_records < 1) return -2;
  const int blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(adam_dev_report_pairs_kernel, dim3(blocks + 1), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, n, lr_dev, beta1,
                     beta2, eps, step_dev, blocks, reinterpret_cast<const double*>(region), n_records, sums, maxes, entropy_coef, out14);
  GRL_CHECK_LAUNCH();
  return 0;
}

// sqnorm: device fp64[1] (written: the squared norm), coef: device float[1]; one launch
int grl_clip_coef(const float* grads, int n, float max_norm, double* sqnorm, float* coef, hipStream_t stream) {
  hipLaunchKernelGGL(sqnorm_kernel, dim3(1), dim3(1024), 0, stream, grads, n, sqnorm, max_norm, coef);
  GRL_CHECK_LAUNCH();
  return 0;
}

// reward [N,T] f32, done/terminated [N,T] u8 (torch.bool), values [N,T+1] -> advantage, value_target [N,T]
int grl_gae_scan(const float* reward, const unsigned char* done, const unsigned char* terminated, const float* values,
                 float* advantage, float* value_target, int n_env, int n_steps, float gamma, float lmbda, hipStream_t stream) {
  if (n_env <= 0 || n_steps <= 0) return 0;
  hipLaunchKernelGGL(gae_kernel, dim3((n_env + 63) / 64), dim3(64), 0, stream, reward, done, terminated, values, advantage,
                     value_target, n_env, n_steps, gamma, lmbda);
  GRL_CHECK_LAUNCH();
  return 0;
}

// descs: HOST array of n_desc <= 24 records of 18 8-byte words each:
//   [out, a, b, gather (pointers; 0 = absent), out_row_stride, out_col, rows_per_sample, row_off, n_nodes, n_per,
//    a_stride, a_off, a_bcast, b_stride, b_off, b_bcast, onehot_col, n_types]
int grl_build_features_bump(const long long* descs, int n_desc, int* bump, hipStream_t stream);
int grl_build_features(const long long* descs, int n_desc, hipStream_t stream) { return grl_build_features_bump(descs, n_desc, nullptr, stream); }
// the same; bump (device int[1] or NULL) is advanced by one by this launch
int grl_build_features_bump(const long long* descs, int n_desc, int* bump, hipStream_t stream) {
  if (n_desc <= 0) return 0;
  if (n_desc > FEAT_MAX) return -2;
  FeatDescs all{};
  const int max_nodes = feat_fill(all, descs, n_desc);
  const int bx = (max_nodes + 255) / 256 < 256 ? (max_nodes + 255) / 256 : 256;
  hipLaunchKernelGGL(build_features_kernel, dim3(bx, n_desc), dim3(256), 0, stream, all, bump);
  GRL_CHECK_LAUNCH();
  return 0;
}

// dst / src: HOST arrays of n <= 24 device pointers, bytes: HOST array of sizes
int grl_copy_many(void* const* dst, const void* const* src, const long long* bytes, int n, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > COPY_MAX) return -2;
  CopyJobs jobs{};
  long long mx = 0;
  for (int i = 0; i < n; ++i) {
    jobs.dst[i] = dst[i];
    jobs.src[i] = src[i];
    jobs.bytes[i] = bytes[i];
    if (bytes[i] > mx) mx = bytes[i];
  }
  long long bx = (mx / 16 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(copy_many_kernel, dim3((int)bx, n), dim3(256), 0, stream, jobs);
  GRL_CHECK_LAUNCH();
  return 0;
}

// dst[k][i, :] = src[k][idx[i], :] for k < n <= 24 tensors with rows of row_bytes[k] bytes (multiples of 4); idx: DEVICE int64[n_rows]
int grl_gather_rows_many_cur(void* const* dst, const void* const* src, const long long* row_bytes, int n, const long long* idx, int n_rows,
                             const int* count, const int* base, int n_idx_rows, hipStream_t stream);
int grl_gather_rows_many(void* const* dst, const void* const* src, const long long* row_bytes, int n, const long long* idx, int n_rows,
                         hipStream_t stream) {
  return grl_gather_rows_many_cur(dst, src, row_bytes, n, idx, n_rows, nullptr, nullptr, 1, stream);
}
// the same with the index row chosen on the device: idx = a matrix [n_idx_rows, n_rows] (device int64), the launch gathers the rows of its
// line (count[0] - base[0]) mod n_idx_rows (count, base: device int32[1]; count = NULL: idx is the row itself)
int grl_gather_rows_many_cur(void* const* dst, const void* const* src, const long long* row_bytes, int n, const long long* idx, int n_rows,
                             const int* count, const int* base, int n_idx_rows, hipStream_t stream) {
  if (n <= 0 || n_rows <= 0) return 0;
  if (count && (!base || n_idx_rows < 1)) return -2;
  if (n > COPY_MAX) return -2;
  GatherJobs jobs{};
  long long mx = 0;
  for (int i = 0; i < n; ++i) {
    if (row_bytes[i] % 4) return -3;
    jobs.dst[i] = reinterpret_cast<unsigned int*>(dst[i]);
    jobs.src[i] = reinterpret_cast<const unsigned int*>(src[i]);
    jobs.words[i] = (int)(row_bytes[i] / 4);
    if (row_bytes[i] / 4 * n_rows > mx) mx = row_bytes[i] / 4 * n_rows;
  }
  long long bx = (mx + 1023) / 1024;
  if (bx < 1) bx = 1;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(gather_rows_many_kernel, dim3((int)bx, n), dim3(256), 0, stream, jobs, idx, n_rows, count, base, n_idx_rows);
  GRL_CHECK_LAUNCH();
  return 0;
}

// x [rows, K] (K <= 64); state: device float[2K+1] = [sum | ssq | count] (updated in place when update != 0, as in training; frozen
// statistics otherwise); scratch: device, >= 256*2K doubles + 2K floats; y_norm / y_clip: optional outputs [rows, K]
int grl_vecnorm(const float* x, long long rows, int K, float decay, float eps, int update, float lo, float hi, float* state,
                void* scratch, float* y_norm, float* y_clip, hipStream_t stream) {
  if (K < 1 || K > VN_KMAX) return -2;
  if (rows <= 0) return 0;
  double* partial = reinterpret_cast<double*>(scratch);
  float* mean_std = reinterpret_cast<float*>(partial + (size_t)VN_BLOCKS * 2 * K);
  const int per = 256 / K;
  long long nb = (rows + per - 1) / per;
  if (nb > VN_BLOCKS) nb = VN_BLOCKS;
  if (update) {
    hipLaunchKernelGGL(vecnorm_partial_kernel, dim3((int)nb), dim3(256), 0, stream, x, rows, K, partial);
    GRL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(vecnorm_update_kernel, dim3(1), dim3(64), 0, stream, partial, (int)nb, rows, K, decay, eps, update, state, mean_std);
  GRL_CHECK_LAUNCH();
  const long long n = rows * K;
  long long ab = (n + 1023) / 1024;
  if (ab > 1024) ab = 1024;
  hipLaunchKernelGGL(vecnorm_apply_kernel, dim3((int)(ab < 1 ? 1 : ab)), dim3(256), 0, stream, x, mean_std, n, K, lo, hi, y_norm, y_clip);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_vecnorm_scratch_bytes(int K) { return (int)(sizeof(double) * VN_BLOCKS * 2 * K + sizeof(float) * 2 * K); }

int grl_knn_topology(const float* pos, const int* n_valid, int* out_nbr, int batch, int n_points, int k, hipStream_t stream) {
  if (n_points > KNN_PMAX || k > 8 || k < 1) return -2;
  hipLaunchKernelGGL(knn_kernel, dim3(batch), dim3(128), 0, stream, pos, n_valid, out_nbr, batch, n_points, k);
  GRL_CHECK_LAUNCH();
  return 0;
}


// Switch the per-kernel timing on (1: events around eagerly issued launches, 2: wall-clock stamp kernels, for recorded steps; clears old
// records) or off (0).
int grl_prof_enable(int on) {
  for (auto& r : g_prof) if (r.e0) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_mode = on == 2 ? 2 : (on != 0 ? 1 : 0);
  g_stamp_next = 0;
  g_stamp_open = -1;
  if (g_prof_mode == 2 && !g_stamps) {
    if (hipMalloc(&g_stamps, sizeof(unsigned long long) * 2 * PROF_STAMPS) != hipSuccess) { g_prof_mode = 0; return 1; }
    int dev = 0, khz = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) g_wall_khz = khz;
  }
  return 0;
}
int grl_prof_count() { return (int)g_prof.size(); }
// Record i -> kernel name (NUL-terminated, at most cap bytes) and its duration in ms; waits for the record's end event.
int grl_prof_get(int i, char* name, int cap, float* ms) {
  if (i < 0 || i >= (int)g_prof.size() || cap < 1) return 1;
  if (g_prof[i].slot >= 0) {   // stamp pair of the LATEST execution (eager or replayed) of that launch; the caller has synchronised
    unsigned long long t[2];
    if (hipMemcpy(t, g_stamps + 2 * g_prof[i].slot, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    *ms = (float)((double)(t[1] - t[0]) / g_wall_khz);
  } else {
    hipEventSynchronize(g_prof[i].e1);
    if (hipEventElapsedTime(ms, g_prof[i].e0, g_prof[i].e1) != hipSuccess) return 2;
  }
  int k = 0;
  for (; k < cap - 1 && g_prof[i].name[k]; ++k) name[k] = g_prof[i].name[k];
  name[k] = 0;
  return 0;
}
}  // extern "C"
