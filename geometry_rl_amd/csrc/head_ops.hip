// Actor read-out + contextual-std head, and the fused TRPL objective (forward values AND analytic gradients).
//
// Read-out (reference hepi.py:173-190 == ponita_gcn.py:129-146; std head gnn_gaussian_policy_diag.py:65-87):
//   y = lat W_dec^T + b (per orientation); scalar part averaged over the grid, vector part projected on the grid,
//   mean[v,:] = vec[v,:] * scal[v]; hidden = mean_o lat; sigma = softplus(hidden W_s^T + b_s + shift) + min_std.
// TRPL (reference objectives/trpl.py:231-321, projections/base_projection_layer.py:71-100,292-384,
//   projections/kl_projection_layer.py:15-111, utils/projection_utils.py:34-67, objectives/utils.py:5-28; ITPAL's diagonal
//   covariance projection restated from its KKT system): one thread per frame, fp64 inside.
#include "grl_common.h"
#include "grl_report.h"

namespace {

constexpr int C = 64, O = 16;
constexpr int JMAX = 4;   // output_dim + output_dim_vec
constexpr int APER_MAX = 6;

GRL_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
GRL_DEVINL float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
GRL_DEVINL float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// one wave per actuator node
__global__ __launch_bounds__(64) void readout_fwd_kernel(const float* __restrict__ lat, const float* __restrict__ grid,
                                                        const float* __restrict__ Wd, const float* __restrict__ bd,
                                                        const float* __restrict__ Ws, const float* __restrict__ bs, float shift,
                                                        float min_std, float* __restrict__ mean, float* __restrict__ sigma,
                                                        float* __restrict__ hidden, int n_nodes, int od, int ov) {
  const int c = threadIdx.x;
  const int J = od + ov, aper = 3 * ov;
  float wd[JMAX], ws[APER_MAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) wd[j] = j < J ? Wd[j * C + c] : 0.f;
#pragma unroll
  for (int a = 0; a < APER_MAX; ++a) ws[a] = a < aper ? Ws[a * C + c] : 0.f;
  for (int n = blockIdx.x; n < n_nodes; n += gridDim.x) {
    const float* l = lat + (size_t)n * O * C + c;
    // y[o][j] = lat[o] . wd[j] + bd[j] is linear in lat, so the sums over the orientations are taken per lane FIRST and each
    // output needs ONE wave reduction (14 per node instead of 70):
    //   sum_o y[o][j] = (sum_o lat[o]) . wd[j] + 16 bd[j],   sum_o y[o][j] g_o = (sum_o lat[o] g_o) . wd[j] + bd[j] sum_o g_o
    float hsum = 0.f, lgx = 0.f, lgy = 0.f, lgz = 0.f, sgx = 0.f, sgy = 0.f, sgz = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float v = l[o * C];
      const float gx = grid[3 * o], gy = grid[3 * o + 1], gz = grid[3 * o + 2];
      hsum += v;
      lgx += v * gx; lgy += v * gy; lgz += v * gz;
      sgx += gx; sgy += gy; sgz += gz;
    }
    float sc[JMAX] = {0.f, 0.f, 0.f, 0.f};      // sum_o y[o][j]          (j < od)
    float vx[JMAX] = {0.f, 0.f, 0.f, 0.f}, vy[JMAX] = {0.f, 0.f, 0.f, 0.f}, vz[JMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      if (j < od) sc[j] = wave_sum(hsum * wd[j]) + O * bd[j];
      else if (j < J) {
        vx[j - od] = wave_sum(lgx * wd[j]) + bd[j] * sgx;
        vy[j - od] = wave_sum(lgy * wd[j]) + bd[j] * sgy;
        vz[j - od] = wave_sum(lgz * wd[j]) + bd[j] * sgz;
      }
    }
    const float hid = hsum * (1.f / O);
    hidden[(size_t)n * C + c] = hid;
#pragma unroll
    for (int a = 0; a < APER_MAX; ++a) {
      if (a < aper) {
        const float pre = wave_sum(hid * ws[a]) + bs[a];
        if (c == 0) sigma[(size_t)n * aper + a] = softplus_f(pre + shift) + min_std;
      }
    }
    if (c == 0) {
      for (int v = 0; v < ov; ++v) {
        const float s = sc[v] * (1.f / O);
        mean[((size_t)n * ov + v) * 3 + 0] = vx[v] * (1.f / O) * s;
        mean[((size_t)n * ov + v) * 3 + 1] = vy[v] * (1.f / O) * s;
        mean[((size_t)n * ov + v) * 3 + 2] = vz[v] * (1.f / O) * s;
      }
    }
  }
}

// partial row: [dWd JMAX*64 | dbd JMAX | dWs APER_MAX*64 | dbs APER_MAX]
constexpr int RO_PARTIAL = JMAX * C + JMAX + APER_MAX * C + APER_MAX;
// Four waves per workgroup, one node per wave and iteration (round 3: one wave per workgroup, four nodes in a row per wave at 4096 frames --
// a node is one long dependent chain of loads, ten wave reductions and the std head: 41 us of latency).  The waves' gradient accumulators are
// summed in wave order through LDS: still one partial row per workgroup.
constexpr int RO_WAVES = 4;
__global__ __launch_bounds__(64 * RO_WAVES) void readout_bwd_kernel(const float* __restrict__ lat, const float* __restrict__ grid,
                                                        const float* __restrict__ Wd, const float* __restrict__ bd,
                                                        const float* __restrict__ Ws, const float* __restrict__ bs, float shift,
                                                        const float* __restrict__ dmean, const float* __restrict__ dsigma,
                                                        const float* __restrict__ dhidden_ext, float* __restrict__ dlat,
                                                        float* __restrict__ partial, int n_nodes, int od, int ov) {
  __shared__ float red[RO_WAVES][2 * (JMAX + APER_MAX)][C];
  const int c = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int J = od + ov, aper = 3 * ov;
  float wd[JMAX], ws[APER_MAX], dwd[JMAX], dws[APER_MAX], dbd[JMAX], dbs[APER_MAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) { wd[j] = j < J ? Wd[j * C + c] : 0.f; dwd[j] = 0.f; dbd[j] = 0.f; }
#pragma unroll
  for (int a = 0; a < APER_MAX; ++a) { ws[a] = a < aper ? Ws[a * C + c] : 0.f; dws[a] = 0.f; dbs[a] = 0.f; }
  for (int n = blockIdx.x * RO_WAVES + wave; n < n_nodes; n += gridDim.x * RO_WAVES) {
    const float* l = lat + (size_t)n * O * C + c;
    float lv[O], hsum = 0.f, lgx = 0.f, lgy = 0.f, lgz = 0.f, sgx = 0.f, sgy = 0.f, sgz = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) {
      lv[o] = l[o * C];
      const float gx = grid[3 * o], gy = grid[3 * o + 1], gz = grid[3 * o + 2];
      hsum += lv[o];
      lgx += lv[o] * gx; lgy += lv[o] * gy; lgz += lv[o] * gz;
      sgx += gx; sgy += gy; sgz += gz;
    }
    float sc[JMAX] = {0.f, 0.f, 0.f, 0.f}, vx[JMAX] = {0.f, 0.f, 0.f, 0.f}, vy[JMAX] = {0.f, 0.f, 0.f, 0.f},
          vz[JMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {   // one wave reduction per output (see the forward kernel)
      if (j < od) sc[j] = wave_sum(hsum * wd[j]) + O * bd[j];
      else if (j < J) {
        vx[j - od] = wave_sum(lgx * wd[j]) + bd[j] * sgx;
        vy[j - od] = wave_sum(lgy * wd[j]) + bd[j] * sgy;
        vz[j - od] = wave_sum(lgz * wd[j]) + bd[j] * sgz;
      }
    }
    const float hid = hsum * (1.f / O);
    // std head
    float dhid = dhidden_ext ? dhidden_ext[(size_t)n * C + c] : 0.f;
#pragma unroll
    for (int a = 0; a < APER_MAX; ++a) {
      if (a < aper) {
        const float pre = wave_sum(hid * ws[a]) + bs[a];
        const float dpre = dsigma[(size_t)n * aper + a] * sigmoid_f(pre + shift);
        dws[a] += dpre * hid;
        dbs[a] += dpre;
        dhid += dpre * ws[a];
      }
    }
    // mean = vec * scal
    float ds[JMAX] = {0.f, 0.f, 0.f, 0.f}, dvx[JMAX], dvy[JMAX], dvz[JMAX];
#pragma unroll
    for (int v = 0; v < JMAX; ++v) {
      dvx[v] = dvy[v] = dvz[v] = 0.f;
      if (v < ov) {
        const float s = sc[v] * (1.f / O);
        const float* dm = dmean + ((size_t)n * ov + v) * 3;
        ds[v] = (dm[0] * vx[v] + dm[1] * vy[v] + dm[2] * vz[v]) * (1.f / O);
        dvx[v] = dm[0] * s; dvy[v] = dm[1] * s; dvz[v] = dm[2] * s;
      }
    }
    float* dl = dlat + (size_t)n * O * C + c;
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float gx = grid[3 * o], gy = grid[3 * o + 1], gz = grid[3 * o + 2];
      float g = dhid * (1.f / O);
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        if (j < J) {
          const float dy = (j < od) ? ds[j] * (1.f / O) : (dvx[j - od] * gx + dvy[j - od] * gy + dvz[j - od] * gz) * (1.f / O);
          g += dy * wd[j];
          dwd[j] += dy * lv[o];
          dbd[j] += dy;
        }
      }
      dl[o * C] = g;
    }
  }
#pragma unroll
  for (int j = 0; j < JMAX; ++j) { red[wave][j][c] = dwd[j]; red[wave][JMAX + j][c] = dbd[j]; }
#pragma unroll
  for (int a = 0; a < APER_MAX; ++a) { red[wave][2 * JMAX + a][c] = dws[a]; red[wave][2 * JMAX + APER_MAX + a][c] = dbs[a]; }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w_ = 1; w_ < RO_WAVES; ++w_) {
#pragma unroll
    for (int j = 0; j < JMAX; ++j) { dwd[j] += red[w_][j][c]; dbd[j] += red[w_][JMAX + j][c]; }
#pragma unroll
    for (int a = 0; a < APER_MAX; ++a) { dws[a] += red[w_][2 * JMAX + a][c]; dbs[a] += red[w_][2 * JMAX + APER_MAX + a][c]; }
  }
  float* out = partial + (size_t)blockIdx.x * RO_PARTIAL;
#pragma unroll
  for (int j = 0; j < JMAX; ++j) out[j * C + c] = dwd[j];
#pragma unroll
  for (int a = 0; a < APER_MAX; ++a) out[JMAX * C + JMAX + a * C + c] = dws[a];
  if (c == 0) {
#pragma unroll
    for (int j = 0; j < JMAX; ++j) out[JMAX * C + j] = dbd[j];
#pragma unroll
    for (int a = 0; a < APER_MAX; ++a) out[JMAX * C + JMAX + APER_MAX * C + a] = dbs[a];
  }
}

// ------------------------------------------------------------------------------------------------ TRPL
constexpr int AMAX = 12;
// sums layout (fp64): 0 loss_objective 1 loss_trust_region 2 entropy(dist) 3 loss_critic 4 sum exp(lw) 5 sum exp(2 lw)
//                     6 mean_constraint 7 cov_constraint 8 entropy(p) 9 entropy_diff 10 count 11 kl(p || proj_p)
// (6, 7: the projection's own trust-region measure of (p, proj_p); equal to the KL parts for the KL projection)
// maxes layout (fp32 bits, values >= 0): 0 mean_constraint_max 1 cov_constraint_max
struct TrplCfg {
  double mean_bound, cov_bound, tr_coeff, ent_coef, critic_coef, clip_value, inv_batch, adv_count;
  int A;
  int adv_local;   // 1: the advantage statistics are those of THIS launch's batch, summed inside the kernel (one rank: no statistics launch)
};

GRL_DEVINL double kl_of_eta(double eta, const double* t, const double* o, int A) {
  double kl = 0.0;
  _Pragma("unroll") for (int i = 0; i < A; ++i) {
    const double v = (eta + 1.0) / (eta / o[i] + 1.0 / t[i]);
    kl += v / o[i] - 1.0 - log(v) + log(o[i]);
  }
  return 0.5 * kl;
}

// PROJ: 0 = KL (kl_projection_layer.py + ITPAL), 1 = Frobenius (frob_projection_layer.py:10-88), 2 = Wasserstein, commutative,
//       precision-scaled (w2_projection_layer.py:15-76, projection_utils.py:107-149); diagonal policy throughout
template <int AT, int PROJ>
__global__ __launch_bounds__(128) void trpl_kernel(TrplCfg cfg, const float* __restrict__ mean, const float* __restrict__ sigma,
                                                  const float* __restrict__ action, const float* __restrict__ old_mean,
                                                  const float* __restrict__ old_var, const float* __restrict__ old_logp,
                                                  const float* __restrict__ advantage, const float* __restrict__ value,
                                                  const float* __restrict__ old_value, const float* __restrict__ value_target,
                                                  float* __restrict__ dmean, float* __restrict__ dsigma, float* __restrict__ dvalue,
                                                  float* __restrict__ proj_mean_out, float* __restrict__ proj_var_out,
                                                  const double* __restrict__ adv_stats, double* __restrict__ slots,
                                                  const float* __restrict__ tgt_mean, const float* __restrict__ tgt_S, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  // external target (boundary methods get_trust_region_loss / compute_metrics, base_projection_layer.py:292-384): the projection is
  // skipped, (tgt_mean, tgt_S) stands for the DETACHED proj_p / q and only the regression term's direct gradient is produced
  const bool ext = tgt_mean != nullptr;
  const int A = AT > 0 ? AT : cfg.A;   // compile-time action width: the per-dimension loops unroll, arrays stay in registers
  double acc[11];
  for (int i = 0; i < 11; ++i) acc[i] = 0.0;
  float mmax = 0.f, cmax = 0.f;
  // One rank: every workgroup sums the batch's advantages itself (trpl.py:248-252, 286-289) -- B floats, thread-strided partial sums, wave
  // butterflies, the two waves in order: the SAME order in every workgroup and every run (bitwise reproducible), and one launch plus one
  // cross-lane dependency fewer on the step's chain than the separate statistics kernel (which the data-parallel step keeps: its sums
  // are all-reduced).
  __shared__ double adv_sh[2][2];
  double adv_sum0 = 0.0, adv_sum1 = 0.0;
  if (cfg.adv_local) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
      const double a_ = advantage[i];
      adv_sum0 += a_;
      adv_sum1 += a_ * a_;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { adv_sum0 += __shfl_xor(adv_sum0, off, 64); adv_sum1 += __shfl_xor(adv_sum1, off, 64); }
    if ((threadIdx.x & 63) == 0) { adv_sh[threadIdx.x >> 6][0] = adv_sum0; adv_sh[threadIdx.x >> 6][1] = adv_sum1; }
    __syncthreads();
    adv_sum0 = adv_sh[0][0] + adv_sh[1][0];
    adv_sum1 = adv_sh[0][1] + adv_sh[1][1];
  }
  if (b < B) {
    double mu[AMAX], S[AMAX], mo[AMAX], So[AMAX], t[AMAX], o[AMAX], a[AMAX];
    _Pragma("unroll") for (int i = 0; i < A; ++i) {
      mu[i] = mean[(size_t)b * A + i];
      const double sg = sigma[(size_t)b * A + i];
      S[i] = sg * sg;                      // policy covariance diagonal == "std" seen by the projection (trpl.py:241)
      mo[i] = old_mean[(size_t)b * A + i];
      So[i] = old_var[(size_t)b * A + i];
      a[i] = action[(size_t)b * A + i];
      t[i] = S[i] * S[i];                  // kl_projection_layer.py:60-63: covariance(std) = std**2
      o[i] = So[i] * So[i];
    }
    // ---- mean projection (base_projection_layer.py:71-100)
    double mp = 0.0;
    _Pragma("unroll") for (int i = 0; i < A; ++i) { const double d = (mu[i] - mo[i]) / So[i]; mp += d * d; }
    if (PROJ == 0) mp *= 0.5;   // KL: 1/2 maha (projection_utils.py:34-67); Frobenius / W2: maha (mean_distance, :9-31)
    const bool m_act = mp > cfg.mean_bound;
    double omega = 0.0, D = 1.0, pm[AMAX];
    if (m_act) { omega = sqrt(mp / cfg.mean_bound) - 1.0; D = 1.0 + omega + 1e-16; }
    _Pragma("unroll") for (int i = 0; i < A; ++i) pm[i] = m_act ? (mu[i] + omega * mo[i]) / D : mu[i];
    // ---- covariance projection: eta >= 0 with KL_cov(eta) = cov_bound.  With rho_i = v_i/o_i = (eta+1)/(eta+c_i), c_i = o_i/t_i:
    //      KL = 1/2 sum(rho_i - 1 - log rho_i),  dKL/deta = -1/2 sum (1-c_i)^2 / ((eta+1)(eta+c_i)^2) < 0, KL convex in eta:
    //      Newton from eta = 0 approaches the root monotonically from the left (never overshoots).
    double eta = 0.0;
    bool c_act;
    double v[AMAX], pS[AMAX];
    if (PROJ == 0) {
      c_act = kl_of_eta(0.0, t, o, A) > cfg.cov_bound;
      if (c_act) {
        double cr[AMAX];
        _Pragma("unroll") for (int i = 0; i < A; ++i) cr[i] = o[i] / t[i];
        // Phase 1, fp32 (hardware log2 / reciprocal): the same Newton iteration, cheap, until the step is below 1e-5 of eta.
        {
          float crf[AMAX], e32 = 0.f;
          _Pragma("unroll") for (int i = 0; i < A; ++i) crf[i] = (float)cr[i];
          const float bound = (float)cfg.cov_bound;
          for (int it = 0; it < 60; ++it) {
            float f = 0.f, df = 0.f;
            const float r1 = __builtin_amdgcn_rcpf(e32 + 1.f);
            _Pragma("unroll") for (int i = 0; i < A; ++i) {
              const float rden = __builtin_amdgcn_rcpf(e32 + crf[i]), rho = (e32 + 1.f) * rden, om = 1.f - crf[i];
              f += rho - 1.f - 0.69314718056f * __builtin_amdgcn_logf(rho);
              df += om * om * r1 * rden * rden;
            }
            f = 0.5f * f - bound;
            df *= 0.5f;
            if (!(f > 0.f) || !(df > 0.f)) break;
            const float step = f / df;
            e32 += step;
            if (!(step > 1e-5f * e32)) break;
          }
          if (e32 == e32 && e32 >= 0.f && e32 < 3.0e38f) eta = (double)e32;   // otherwise phase 2 starts from 0 as before
        }
        // Phase 2, fp64 polish.  KL is convex and decreasing in eta, so ONE Newton step from either side of the root lands left
        // of it (the tangent lies below the curve); from there the iteration is monotone as before and stops at double precision.
        for (int it = 0; it < 100; ++it) {
          double f = 0.0, df = 0.0;
          _Pragma("unroll") for (int i = 0; i < A; ++i) {
            const double den = eta + cr[i], rho = (eta + 1.0) / den, om = 1.0 - cr[i];
            f += rho - 1.0 - log(rho);
            df += om * om / ((eta + 1.0) * den * den);
          }
          f = 0.5 * f - cfg.cov_bound;
          df *= 0.5;
          if (df <= 0.0 || (it > 0 && f <= 0.0)) break;
          const double step = f / df;
          eta = fmax(eta + step, 0.0);
          if (fabs(step) <= 1e-15 * eta) break;
        }
      }
      _Pragma("unroll") for (int i = 0; i < A; ++i) { v[i] = (eta + 1.0) / (eta / o[i] + 1.0 / t[i]); pS[i] = sqrt(v[i]); }
    } else {
      // closed forms: eta = sqrt(part / bound) - 1 where the bound is violated
      double part = 0.0;
      _Pragma("unroll") for (int i = 0; i < A; ++i) {
        const double d = PROJ == 1 ? o[i] - t[i] : 1.0 - S[i] / So[i];   // |S_o^2 - S^2|_F^2  |  tr(I + S_o^-1 S^2 S_o^-1 - 2 S_o^-1 S)
        part += d * d;
      }
      c_act = part > cfg.cov_bound;
      if (c_act) eta = fabs(sqrt(part / cfg.cov_bound) - 1.0);
      const double den = 1.0 + eta + 1e-16;
      _Pragma("unroll") for (int i = 0; i < A; ++i) {
        if (PROJ == 1) { v[i] = c_act ? (t[i] + eta * o[i]) / den : t[i]; pS[i] = c_act ? sqrt(v[i]) : S[i]; }   // chol of the mixed covariance
        else { pS[i] = c_act ? (S[i] + eta * So[i]) / den : S[i]; v[i] = pS[i] * pS[i]; }
      }
    }
    if (ext) {
      _Pragma("unroll") for (int i = 0; i < A; ++i) {
        pm[i] = tgt_mean[(size_t)b * A + i];
        pS[i] = tgt_S[(size_t)b * A + i];
        v[i] = pS[i] * pS[i];
      }
    }
    // ---- log-prob under the projected distribution (covariance = pS), importance weight, objective
    const double LOG2PI = 1.8378770664093454836;
    double q = 0.0, sl = 0.0;
    _Pragma("unroll") for (int i = 0; i < A; ++i) { const double d = a[i] - pm[i]; q += d * d / pS[i]; sl += log(pS[i]); }
    const double lw = -0.5 * (q + A * LOG2PI + sl) - (double)old_logp[b];
    const double ratio = exp(lw);
    // advantage normalisation (trpl.py:286-289): batch mean / unbiased std (clamped at 1e-6) from the device-side sums
    double adv = (double)advantage[b];
    if ((adv_stats || cfg.adv_local) && cfg.adv_count > 1.0) {
      const double s0_ = cfg.adv_local ? adv_sum0 : adv_stats[0], s1_ = cfg.adv_local ? adv_sum1 : adv_stats[1];
      const double am = s0_ / cfg.adv_count;
      double var = (s1_ - cfg.adv_count * am * am) / (cfg.adv_count - 1.0);
      double sd = var > 0.0 ? sqrt(var) : 0.0;
      if (sd < 1e-6) sd = 1e-6;
      adv = (adv - am) / sd;
    }
    acc[0] = -ratio * adv;
    acc[4] = ratio;
    acc[5] = ratio * ratio;
    const double ent = 0.5 * (A * (1.0 + LOG2PI) + sl);
    acc[2] = ent;
    // ---- trust-region regression loss and metrics.  KL of (p, proj_p) is always reported; the constraint metrics are the
    //      projection's own measure of (p, proj_p) (base_projection_layer.py:332-384), the loss is base.py:292-327 with that
    //      measure (KL, W2: proj_p detached) or frob_projection_layer.py:73-88 (maha by the live S + squared distance, NOT detached)
    double mk = 0.0, ck = 0.0, ldS = 0.0, ldP = 0.0, cd = 0.0, mS = 0.0, sq = 0.0;
    _Pragma("unroll") for (int i = 0; i < A; ++i) {
      const double d = (mu[i] - pm[i]) / pS[i];
      mk += d * d;
      const double rr = S[i] / pS[i];
      ck += rr * rr;
      ldS += log(S[i]);
      ldP += log(pS[i]);
      if (PROJ == 1) {
        const double f = pS[i] * pS[i] - S[i] * S[i], dm = (mu[i] - pm[i]) / S[i], ds = S[i] - pS[i];
        cd += f * f; mS += dm * dm; sq += ds * ds;
      }
      if (PROJ == 2) cd += (1.0 - rr) * (1.0 - rr);
    }
    const double md = mk;            // maha(mean, proj_mean, proj_S)
    mk *= 0.5;
    ck = 0.5 * (ck - A + 2.0 * ldP - 2.0 * ldS);
    acc[10] = mk + ck;
    if (PROJ == 0) { acc[1] = (mk + ck) * cfg.tr_coeff; acc[6] = mk; acc[7] = ck; }
    if (PROJ == 1) { acc[1] = (mS + sq) * cfg.tr_coeff; acc[6] = md; acc[7] = cd; }
    if (PROJ == 2) { acc[1] = (md + cd) * cfg.tr_coeff; acc[6] = md; acc[7] = cd; }
    const double c_ent = 0.5 * A * 2.8378770664093454836;  // 0.5 k log(2 pi e)
    acc[8] = c_ent + ldS;                                   // policy.entropy(p) with S as "std"
    acc[9] = (c_ent + ldP) - (c_ent + ldS);
    mmax = (float)acc[6];
    cmax = (float)fmax(acc[7], 0.0);
    // ---- gradients of actor_loss = objective + entropy bonus + trust region  (all already scaled by 1/B)
    const double w_obj = -ratio * adv * cfg.inv_batch;
    double g_pm[AMAX], g_pS[AMAX];
    _Pragma("unroll") for (int i = 0; i < A; ++i) {
      const double d = a[i] - pm[i];
      g_pm[i] = w_obj * d / pS[i];
      g_pS[i] = w_obj * 0.5 * (d * d / (pS[i] * pS[i]) - 1.0 / pS[i]) - cfg.ent_coef * cfg.inv_batch * 0.5 / pS[i];
    }
    const double ctr = cfg.tr_coeff * cfg.inv_batch;
    if (ext) {   // detached target: nothing flows through the projection
      _Pragma("unroll") for (int i = 0; i < A; ++i) { g_pm[i] = 0.0; g_pS[i] = 0.0; }
    } else if (PROJ == 1) {   // the Frobenius regression loss also reaches the parameters THROUGH the projection
      _Pragma("unroll") for (int i = 0; i < A; ++i) {
        g_pm[i] -= ctr * 2.0 * (mu[i] - pm[i]) / (S[i] * S[i]);
        g_pS[i] -= ctr * 2.0 * (S[i] - pS[i]);
      }
    }
    double gmu[AMAX], gS[AMAX];
    if (m_act && !ext) {
      double dot = 0.0;
      _Pragma("unroll") for (int i = 0; i < A; ++i) dot += g_pm[i] * (mo[i] - pm[i]) / D;
      const double k = dot / (2.0 * (omega + 1.0) * cfg.mean_bound) * (PROJ == 0 ? 1.0 : 2.0);   // d(mean part)/d maha = 1/2 | 1
      _Pragma("unroll") for (int i = 0; i < A; ++i) gmu[i] = g_pm[i] / D + k * (mu[i] - mo[i]) / (So[i] * So[i]);
    } else {
      _Pragma("unroll") for (int i = 0; i < A; ++i) gmu[i] = g_pm[i];
    }
    if (ext) {
      _Pragma("unroll") for (int i = 0; i < A; ++i) gS[i] = 0.0;
    } else if (PROJ == 0) {
      double gv[AMAX];
      _Pragma("unroll") for (int i = 0; i < A; ++i) gv[i] = g_pS[i] / (2.0 * pS[i]);
      if (c_act) {
        double dvt[AMAX], dve[AMAX], gk[AMAX], denom = 0.0, num = 0.0;
        _Pragma("unroll") for (int i = 0; i < A; ++i) {
          dvt[i] = v[i] * v[i] / (t[i] * t[i] * (eta + 1.0));
          dve[i] = -v[i] * v[i] * (1.0 / o[i] - 1.0 / t[i]) / ((eta + 1.0) * (eta + 1.0));
          gk[i] = 0.5 * (1.0 / o[i] - 1.0 / v[i]);
          denom += gk[i] * dve[i];
          num += gv[i] * dve[i];
        }
        _Pragma("unroll") for (int i = 0; i < A; ++i) {
          const double gt = gv[i] * dvt[i] - num * gk[i] * dvt[i] / denom;
          gS[i] = gt * 2.0 * S[i];
        }
      } else {
        _Pragma("unroll") for (int i = 0; i < A; ++i) gS[i] = gv[i] * 2.0 * S[i];
      }
    } else if (c_act) {
      // proj = (x + eta y) / (1 + eta) with eta = sqrt(part / bound) - 1: d proj_i / d S_j = delta_ij x'_j / den + (y_i - proj_i) / den * d eta / d S_j
      const double den = 1.0 + eta + 1e-16, deta = 1.0 / (2.0 * (eta + 1.0) * cfg.cov_bound);
      double st_ = 0.0;
      _Pragma("unroll") for (int i = 0; i < A; ++i)
        st_ += PROJ == 1 ? g_pS[i] / (2.0 * pS[i]) * (o[i] - v[i]) / den : g_pS[i] * (So[i] - pS[i]) / den;
      _Pragma("unroll") for (int i = 0; i < A; ++i) {
        if (PROJ == 1) gS[i] = g_pS[i] / (2.0 * pS[i]) * 2.0 * S[i] / den + st_ * deta * (-4.0 * (o[i] - t[i]) * S[i]);
        else gS[i] = g_pS[i] / den + st_ * deta * (-2.0 * (1.0 - S[i] / So[i]) / So[i]);
      }
    } else {
      _Pragma("unroll") for (int i = 0; i < A; ++i) gS[i] = g_pS[i];
    }
    _Pragma("unroll") for (int i = 0; i < A; ++i) {
      if (PROJ == 0) {
        gmu[i] += ctr * (mu[i] - pm[i]) / (pS[i] * pS[i]);
        gS[i] += ctr * (S[i] / (pS[i] * pS[i]) - 1.0 / S[i]);
      } else if (PROJ == 1) {
        const double dm = mu[i] - pm[i];
        gmu[i] += ctr * 2.0 * dm / (S[i] * S[i]);
        gS[i] += ctr * (-2.0 * dm * dm / (S[i] * S[i] * S[i]) + 2.0 * (S[i] - pS[i]));
      } else {
        gmu[i] += ctr * 2.0 * (mu[i] - pm[i]) / (pS[i] * pS[i]);
        gS[i] += ctr * (-2.0 * (1.0 - S[i] / pS[i]) / pS[i]);
      }
      dmean[(size_t)b * A + i] = (float)gmu[i];
      dsigma[(size_t)b * A + i] = (float)(gS[i] * 2.0 * (double)sigma[(size_t)b * A + i]);
      if (proj_mean_out) { proj_mean_out[(size_t)b * A + i] = (float)pm[i]; proj_var_out[(size_t)b * A + i] = (float)pS[i]; }
    }
    // ---- clipped value loss (trpl.py:213-228, objectives/utils.py:5-28), l2
    if (value) {
      const double V = value[b], Vo = old_value[b], R = value_target[b];
      const double l1 = (V - R) * (V - R);
      double l = l1, g = 2.0 * (V - R);
      if (cfg.clip_value > 0.0) {
        const double dlt = V - Vo;
        const bool inside = dlt >= -cfg.clip_value && dlt <= cfg.clip_value;
        const double Vc = Vo + fmin(fmax(dlt, -cfg.clip_value), cfg.clip_value);
        const double l2 = (Vc - R) * (Vc - R);
        if (l2 > l1) { l = l2; g = inside ? 2.0 * (Vc - R) : 0.0; }
      }
      acc[3] = l * cfg.critic_coef;
      dvalue[b] = (float)(g * cfg.critic_coef * cfg.inv_batch);
    }
  }
  // block reduction (2 waves) -> this workgroup's own slot (plain stores; trpl_fold_kernel adds the slots up in a fixed order:
  // no atomics, the reported values are bitwise reproducible)
  __shared__ double red[2][11];
  __shared__ float redm[2][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = 0; i < 11; ++i) {
    double x = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) red[wv][i] = x;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mmax = fmaxf(mmax, __shfl_xor(mmax, off, 64));
    cmax = fmaxf(cmax, __shfl_xor(cmax, off, 64));
  }
  if (lane == 0) { redm[wv][0] = mmax; redm[wv][1] = cmax; }
  __syncthreads();
  double* slot = slots + (size_t)blockIdx.x * TRPL_SLOT;
  if (threadIdx.x < 10) slot[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];
  if (threadIdx.x == 13) slot[11] = red[0][10] + red[1][10];
  if (threadIdx.x == 10) {
    const int n_here = min(B - (int)(blockIdx.x * blockDim.x), (int)blockDim.x);
    slot[10] = (double)(n_here > 0 ? n_here : 0);
  }
  if (threadIdx.x == 11) slot[12] = (double)fmaxf(redm[0][0], redm[1][0]);
  if (threadIdx.x == 12) slot[13] = (double)fmaxf(redm[0][1], redm[1][1]);
}

// slots [n_blocks][14] -> sums[12] (written, not accumulated) and maxes[2] (float bits), block order fixed
__global__ __launch_bounds__(64) void trpl_fold_kernel(const double* __restrict__ slots, int n_blocks, double* __restrict__ sums,
                                                      unsigned int* __restrict__ maxes) {
  const int i = threadIdx.x;
  if (i < 12) {
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += slots[(size_t)b * TRPL_SLOT + i];
    sums[i] = s;
  } else if (i < 14) {
    double m = 0.0;
    for (int b = 0; b < n_blocks; ++b) m = fmax(m, slots[(size_t)b * TRPL_SLOT + i]);
    maxes[i - 12] = __float_as_uint((float)m);
  }
}

// sum and sum of squares of the advantages (fp64) WRITTEN to stats[0..1] (round 4: the slots need not be zeroed) -- ONE workgroup, fixed summation order (thread-strided partial sums,
// wave butterflies, the sixteen waves in order): bitwise reproducible.  (Until round 3 every wave of a multi-workgroup grid added its sums
// with fp64 atomics: the order of 64 additions, hence the last bits of the normalised advantages, depended on the run.)
__global__ __launch_bounds__(1024) void adv_stats_kernel(const float* __restrict__ adv, double* __restrict__ stats, int B) {
  __shared__ double red[16][2];
  double s0 = 0, s1 = 0;
  for (int i = threadIdx.x; i < B; i += 1024) {
    const double a = adv[i];
    s0 += a;
    s1 += a * a;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s0; red[threadIdx.x >> 6][1] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = red[0][0], t1 = red[0][1];
#pragma unroll
    for (int w = 1; w < 16; ++w) { t0 += red[w][0]; t1 += red[w][1]; }
    stats[0] = t0;
    stats[1] = t1;
  }
}

// ---- the critic's share of the loss on the critic's lane (one rank): clipped value loss (trpl.py:213-228, objectives/utils.py:5-28, l2) and
//      its gradient d loss_critic / d V -- elementwise in the frame, so it needs nothing from the fused actor kernel: the critic's forward,
//      loss, backward, fold and optimizer step form ONE chain that never meets the actor's.  ONE workgroup, fixed summation order.
//      out2: [0] = sum over the frames of critic_coef * loss (fp64), [1] = that sum / n_global (the reported loss_critic, as a double)
__global__ __launch_bounds__(1024) void value_loss_kernel(const float* __restrict__ value, const float* __restrict__ old_value,
                                                         const float* __restrict__ value_target, double clip_value, double critic_coef,
                                                         double inv_batch, float* __restrict__ dvalue, double* __restrict__ out2,
                                                         float* __restrict__ mean_out, int B) {
  __shared__ double red[16];
  double s = 0.0;
  for (int b = threadIdx.x; b < B; b += 1024) {
    const double V = value[b], Vo = old_value[b], R = value_target[b];
    const double l1 = (V - R) * (V - R);
    double l = l1, g = 2.0 * (V - R);
    if (clip_value > 0.0) {
      const double dlt = V - Vo;
      const bool inside = dlt >= -clip_value && dlt <= clip_value;
      const double Vc = Vo + fmin(fmax(dlt, -clip_value), clip_value);
      const double l2 = (Vc - R) * (Vc - R);
      if (l2 > l1) { l = l2; g = inside ? 2.0 * (Vc - R) : 0.0; }
    }
    s += l * critic_coef;
    dvalue[b] = (float)(g * critic_coef * inv_batch);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += red[w];
    out2[0] = t;
    out2[1] = t * inv_batch;
    if (mean_out) mean_out[0] = (float)(t * inv_batch);
  }
}

__global__ void loss_values_kernel(const double* __restrict__ sums, const unsigned int* __restrict__ maxes, float entropy_coef,
                                   float* __restrict__ out) {
  const double n = sums[10];
  const float tr = (float)(sums[1] / n), ent = -entropy_coef * (float)(sums[2] / n);
  const float actor = (float)((sums[0] + sums[1] - (double)entropy_coef * sums[2]) / n);
  out[0] = actor;
  out[1] = (float)(sums[3] / n);
  out[2] = tr;
  out[3] = ent;
  out[4] = (float)(sums[4] * sums[4] / sums[5] / n);   // exp(2 lse(lw) - lse(2 lw)) / B   (trpl.py:294-300,316)
  const float mc = (float)(sums[6] / n), cc = (float)(sums[7] / n);
  out[5] = (float)(sums[11] / n);
  out[6] = mc;
  out[7] = __uint_as_float(maxes[0]);
  out[8] = cc;
  out[9] = __uint_as_float(maxes[1]);
  out[10] = (float)(sums[8] / n);
  out[11] = (float)(sums[9] / n);
  out[12] = actor - (tr + ent);
  out[13] = mc + cc;   // "constraint": the projection's own measure (= kl for the KL projection)
}

// slots -> sums / maxes (trpl_fold_kernel) AND the reported values (loss_values_kernel) in ONE launch (body: grl_report.h)
__global__ __launch_bounds__(64) void trpl_report_kernel(const double* __restrict__ slots, int n_blocks, double* __restrict__ sums,
                                                        unsigned int* __restrict__ maxes, float entropy_coef, float* __restrict__ out) {
  __shared__ double sh[12];
  __shared__ unsigned int shm[2];
  trpl_report_body(slots, n_blocks, sums, maxes, entropy_coef, out, sh, shm);
}

// ---- collector-side action sampling: torch.distributions.MultivariateNormal(loc, covariance_matrix = diag(sigma^2)).rsample()
//      with return_log_prob (utils_algo_graph.py:146-158, configs/algorithm/policy/default.yaml:6): action = loc + sigma * eps,
//      log p = -1/2 sum eps_eff^2 - sum log sigma - A/2 log(2 pi), eps_eff = (action - loc) / sigma as the distribution computes it
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ loc, const float* __restrict__ sigma,
                                                             const float* __restrict__ eps, float* __restrict__ action,
                                                             float* __restrict__ logp, float* __restrict__ var, int B, int A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float q = 0.f, ls = 0.f;
  for (int i = 0; i < A; ++i) {
    const size_t k = (size_t)b * A + i;
    const float m = loc[k], sg = sigma[k];
    const float a = fmaf(sg, eps[k], m);
    const float d = (a - m) / sg;
    action[k] = a;
    if (var) var[k] = sg * sg;
    q = fmaf(d, d, q);
    ls += logf(sg);
  }
  logp[b] = -0.5f * q - ls - 0.5f * (float)A * 1.8378770664093454836f;
}

}  // namespace

extern "C" {

int grl_readout_partial_size() { return RO_PARTIAL; }
int grl_readout_blocks(int n_nodes) { const int b = (n_nodes + RO_WAVES - 1) / RO_WAVES; return b < 1 ? 1 : (b < 1024 ? b : 1024); }

int grl_readout_fwd(const float* lat, const float* grid, const float* Wd, const float* bd, const float* Ws, const float* bs,
                    float shift, float min_std, float* mean, float* sigma, float* hidden, int n_nodes, int output_dim,
                    int output_dim_vec, hipStream_t stream) {
  if (output_dim != output_dim_vec || output_dim + output_dim_vec > JMAX || 3 * output_dim_vec > APER_MAX) return -2;
  hipLaunchKernelGGL(readout_fwd_kernel, dim3(n_nodes < 4096 ? n_nodes : 4096), dim3(64), 0, stream, lat, grid, Wd, bd, Ws, bs,
                     shift, min_std, mean, sigma, hidden, n_nodes, output_dim, output_dim_vec);
  GRL_CHECK_LAUNCH();
  return 0;
}

// partial: [grl_readout_blocks(n_nodes)][grl_readout_partial_size()]; dhidden_ext may be NULL
int grl_readout_bwd(const float* lat, const float* grid, const float* Wd, const float* bd, const float* Ws, const float* bs,
                    float shift, const float* dmean, const float* dsigma, const float* dhidden_ext, float* dlat, float* partial,
                    int n_nodes, int output_dim, int output_dim_vec, hipStream_t stream) {
  if (output_dim != output_dim_vec || output_dim + output_dim_vec > JMAX || 3 * output_dim_vec > APER_MAX) return -2;
  hipLaunchKernelGGL(readout_bwd_kernel, dim3(grl_readout_blocks(n_nodes)), dim3(64 * RO_WAVES), 0, stream, lat, grid, Wd, bd, Ws, bs, shift,
                     dmean, dsigma, dhidden_ext, dlat, partial, n_nodes, output_dim, output_dim_vec);
  GRL_CHECK_LAUNCH();
  return 0;
}

int grl_adv_stats(const float* advantage, double* stats, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(1024), 0, stream, advantage, stats, batch);
  GRL_CHECK_LAUNCH();
  return 0;
}

// cfg9 (HOST pointer): TEN doubles since ABI 203 {mean_bound, cov_bound, trust_region_coeff, entropy_coef, critic_coef, clip_value,
// 1/B_global, B_global, projection type (0 KL, 1 Frobenius, 2 Wasserstein), adv_local (1: the advantage statistics are summed inside the
// kernel from this launch's batch -- adv_stats is then ignored; 0: adv_stats as below)}.  adv_stats: device fp64[2] = (sum, sum of squares) of the GLOBAL batch's advantages (from
// grl_adv_stats, all-reduced when data parallel) or NULL for no normalisation.  sums: fp64[12], maxes: u32[2], zeroed by the caller.  value/old_value/value_target/dvalue may be
// NULL together (actor-only call); proj_mean/proj_var may be NULL.
int grl_trpl_slot_doubles(int batch) { return TRPL_SLOT * ((batch + 127) / 128 < 1 ? 1 : (batch + 127) / 128); }

// value / old_value / value_target [batch], dvalue [batch], out2 fp64[2] (sum, sum / n_global), mean_out float[1] or NULL (the mean again, as
// the float the loss dict reports): see value_loss_kernel
int grl_value_loss(const float* value, const float* old_value, const float* value_target, double clip_value, double critic_coef,
                   double inv_batch, float* dvalue, double* out2, float* mean_out, int batch, hipStream_t stream) {
  if (batch < 1 || !value || !old_value || !value_target || !dvalue || !out2) return -2;
  hipLaunchKernelGGL(value_loss_kernel, dim3(1), dim3(1024), 0, stream, value, old_value, value_target, clip_value, critic_coef, inv_batch,
                     dvalue, out2, mean_out, batch);
  GRL_CHECK_LAUNCH();
  return 0;
}

static int trpl_launch(const double* cfg9, int action_dim, const float* mean, const float* sigma, const float* action,
                       const float* old_mean, const float* old_var, const float* old_logp, const float* advantage,
                       const float* value, const float* old_value, const float* value_target, float* dmean, float* dsigma,
                       float* dvalue, float* proj_mean, float* proj_var, const double* adv_stats, double* sums,
                       unsigned int* maxes, double* slots, const float* tgt_mean, const float* tgt_S, int batch, hipStream_t stream) {
  if (action_dim > AMAX || action_dim < 1 || batch < 1 || !slots) return -2;
  TrplCfg c{cfg9[0], cfg9[1], cfg9[2], cfg9[3], cfg9[4], cfg9[5], cfg9[6], cfg9[7], action_dim, (int)cfg9[9]};
  const int proj = (int)cfg9[8];
  if (proj < 0 || proj > 2) return -3;
#define GRL_TRPL_LAUNCH(AT, PJ)                                                                                               \
  hipLaunchKernelGGL((trpl_kernel<AT, PJ>), dim3((batch + 127) / 128), dim3(128), 0, stream, c, mean, sigma, action, old_mean,  \
                     old_var, old_logp, advantage, value, old_value, value_target, dmean, dsigma, dvalue, proj_mean, proj_var,   \
                     adv_stats, slots, tgt_mean, tgt_S, batch)
  if (proj == 0) {
    switch (action_dim) {   // the action widths of the reference tasks (G * n_vec * 3) get unrolled instances
      case 3: GRL_TRPL_LAUNCH(3, 0); break;
      case 6: GRL_TRPL_LAUNCH(6, 0); break;
      case 12: GRL_TRPL_LAUNCH(12, 0); break;
      default: GRL_TRPL_LAUNCH(0, 0); break;
    }
  } else if (proj == 1) {
    if (action_dim == 6) GRL_TRPL_LAUNCH(6, 1); else GRL_TRPL_LAUNCH(0, 1);
  } else {
    if (action_dim == 6) GRL_TRPL_LAUNCH(6, 2); else GRL_TRPL_LAUNCH(0, 2);
  }
#undef GRL_TRPL_LAUNCH
  GRL_CHECK_LAUNCH();
  if (sums) {   // sums == NULL: the caller folds the slots later (grl_trpl_fold, on a stream of its choice: the sums are reported values only)
    hipLaunchKernelGGL(trpl_fold_kernel, dim3(1), dim3(64), 0, stream, slots, (batch + 127) / 128, sums, maxes);
    GRL_CHECK_LAUNCH();
  }
  return 0;
}

int grl_trpl_fold(const double* slots, int batch, double* sums, unsigned int* maxes, hipStream_t stream) {
  if (!slots || !sums || !maxes || batch < 1) return -2;
  hipLaunchKernelGGL(trpl_fold_kernel, dim3(1), dim3(64), 0, stream, slots, (batch + 127) / 128, sums, maxes);
  GRL_CHECK_LAUNCH();
  return 0;
}

int grl_trpl_fwd_bwd(const double* cfg9, int action_dim, const float* mean, const float* sigma, const float* action,
                     const float* old_mean, const float* old_var, const float* old_logp, const float* advantage,
                     const float* value, const float* old_value, const float* value_target, float* dmean, float* dsigma,
                     float* dvalue, float* proj_mean, float* proj_var, const double* adv_stats, double* sums,
                     unsigned int* maxes, double* slots, int batch, hipStream_t stream) {
  return trpl_launch(cfg9, action_dim, mean, sigma, action, old_mean, old_var, old_logp, advantage, value, old_value, value_target,
                     dmean, dsigma, dvalue, proj_mean, proj_var, adv_stats, sums, maxes, slots, nullptr, nullptr, batch, stream);
}

// Boundary methods of the projection layer (base_projection_layer.py:292-327 get_trust_region_loss, :332-384 compute_metrics) for
// an ARBITRARY detached target distribution (tgt_mean, tgt_S = the target's "std" diagonal as the layer sees it, i.e. the covariance
// diagonal of the policy): the same kernel with its projection step skipped.  sums / maxes as grl_trpl_fwd_bwd (objective, entropy
// and critic entries are meaningless here); dmean / dsigma = gradient of trust_region_coeff * mean(measure(p, target)).
int grl_trpl_target_terms(const double* cfg9, int action_dim, const float* mean, const float* sigma, const float* tgt_mean,
                          const float* tgt_S, float* dmean, float* dsigma, double* sums, unsigned int* maxes, double* slots,
                          const float* zeros_b /* device float[batch] of zeros: advantage / old log-prob stand-ins */, int batch,
                          hipStream_t stream) {
  if (!tgt_mean || !tgt_S || !zeros_b) return -2;
  // action := target mean, old distribution := target (finite arithmetic in the skipped projection), advantage := 0
  return trpl_launch(cfg9, action_dim, mean, sigma, tgt_mean, tgt_mean, tgt_S, zeros_b, zeros_b, nullptr, nullptr, nullptr, dmean,
                     dsigma, nullptr, nullptr, nullptr, nullptr, sums, maxes, slots, tgt_mean, tgt_S, batch, stream);
}

// Data parallel: a rank's slots -> ONE 14-double record (12 sums, 2 maxes as doubles); the ranks' records are all-gathered (one
// collective for sums and maxes, where an all-reduce needs two: SUM and MAX) and grl_trpl_report_records sums / maximises over them and
// evaluates the reported values -- the same code as over the workgroups' slots of one rank.
__global__ __launch_bounds__(64) void trpl_fold_record_kernel(const double* __restrict__ slots, int n_blocks, double* __restrict__ rec) {
  const int i = threadIdx.x;
  if (i < 12) {
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += slots[(size_t)b * TRPL_SLOT + i];
    rec[i] = s;
  } else if (i < 14) {
    double m = 0.0;
    for (int b = 0; b < n_blocks; ++b) m = fmax(m, slots[(size_t)b * TRPL_SLOT + i]);
    rec[i] = m;
  }
}
int grl_trpl_fold_record(const double* slots, int batch, double* rec14, hipStream_t stream) {
  if (!slots || !rec14 || batch < 1) return -2;
  hipLaunchKernelGGL(trpl_fold_record_kernel, dim3(1), dim3(64), 0, stream, slots, (batch + 127) / 128, rec14);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_trpl_report_records(const double* records, int n_records, double* sums, unsigned int* maxes, float entropy_coef, float* out14,
                            hipStream_t stream) {
  if (!records || !sums || !maxes || !out14 || n_records < 1) return -2;
  hipLaunchKernelGGL(trpl_report_kernel, dim3(1), dim3(64), 0, stream, records, n_records, sums, maxes, entropy_coef, out14);
  GRL_CHECK_LAUNCH();
  return 0;
}

// grl_trpl_fold + grl_trpl_loss_values in one launch (one rank: nothing to all-reduce in between)
int grl_trpl_report(const double* slots, int batch, double* sums, unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream) {
  if (!slots || !sums || !maxes || !out14 || batch < 1) return -2;
  hipLaunchKernelGGL(trpl_report_kernel, dim3(1), dim3(64), 0, stream, slots, (batch + 127) / 128, sums, maxes, entropy_coef, out14);
  GRL_CHECK_LAUNCH();
  return 0;
}

// Reported values from the (globally reduced) sums / maxes of the fused kernel (trpl.py:280-321), one tiny launch instead of a
// chain of scalar tensor ops:  out = [actor loss, critic loss, loss_trust_region, loss_entropy, ESS, kl, mean_constraint,
// mean_constraint_max, cov_constraint, cov_constraint_max, entropy, entropy_diff, loss_objective, constraint]
int grl_trpl_loss_values(const double* sums, const unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream) {
  hipLaunchKernelGGL(loss_values_kernel, dim3(1), dim3(1), 0, stream, sums, maxes, entropy_coef, out14);
  GRL_CHECK_LAUNCH();
  return 0;
}

// action [B,A] = loc + sigma * eps, logp [B], var [B,A] = sigma^2 (optional, NULL to skip); eps: standard normal draws [B,A]
int grl_gaussian_sample(const float* loc, const float* sigma, const float* eps, float* action, float* logp, float* var, int batch,
                        int action_dim, hipStream_t stream) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, loc, sigma, eps, action, logp, var,
                     batch, action_dim);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
