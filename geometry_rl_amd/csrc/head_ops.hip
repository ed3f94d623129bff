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
// sums layout (fp64): 0 loss_objective 1 loss_trust_region 2 entropy(dist) 3 loss_critic 4 sum exp(lw) 5 sum exp(2 lw)
//                     6 mean_constraint 7 cov_constraint 8 entropy(p) 9 entropy_diff 10 count 11 kl(p || proj_p)
// (6, 7: the projection's own trust-region measure of (p, proj_p); equal to the KL parts for the KL projection)
// maxes layout (fp32 bits, values >= 0): 0 mean_constraint_max 1 cov_constraint_max
struct TrplCfg {
  double mean_bound, cov_bound, tr_coeff, ent_coef, critic_coef, clip_value, inv_batch, adv_count;
  int A;
  int adv_local;   // 1: the advantage statistics are those of THIS launch's batch, summed inside the kernel (one rank: no statistics launch)
};

// ------------------------------------------------------------------------------------------------ TRPL, lane-parallel form (round 4)
// PROJ: 0 = KL (kl_projection_layer.py + ITPAL), 1 = Frobenius (frob_projection_layer.py:10-88), 2 = Wasserstein, commutative,
//       precision-scaled (w2_projection_layer.py:15-76, projection_utils.py:107-149); diagonal policy throughout.
// Covariance projection (KL): eta >= 0 with KL_cov(eta) = cov_bound.  With rho_i = v_i/o_i = (eta+1)/(eta+c_i), c_i = o_i/t_i:
//   KL = 1/2 sum(rho_i - 1 - log rho_i),  dKL/deta = -1/2 sum (1-c_i)^2 / ((eta+1)(eta+c_i)^2) < 0, KL convex in eta: Newton from eta = 0
//   approaches the root monotonically from the left (never overshoots).  Phase 1 runs the iteration in fp32 (hardware log2 / reciprocal)
//   until the step is below 1e-5 of eta; phase 2 polishes in fp64: ONE Newton step from either side of the root lands left of it (the
//   tangent lies below the curve), from there the iteration is monotone and stops at double precision.
// External target (boundary methods get_trust_region_loss / compute_metrics, base_projection_layer.py:292-384): tgt_mean != NULL skips the
//   projection; (tgt_mean, tgt_S) stands for the DETACHED proj_p / q and only the regression term's direct gradient is produced.
// Rounds 1-3 ran ONE FRAME PER THREAD (per-dimension arrays in registers, instances unrolled per action width): ~8 000 fp64 instructions
// issued per wave whatever the batch -- 31 us on the step's chain at every size (its wave count, not its arithmetic, shrinks with the
// batch; DESIGN.md finding 9).  Here a frame's A action dimensions sit on L = 4 / 8 / 16 adjacent
// lanes (A <= 4 / 8 / 16); every per-dimension array of the form above is a scalar, every sum over the dimensions a butterfly over the L
// lanes (xor 1, 2, .. : each level adds the same pair on both sides, so all L lanes end with the bitwise-identical sum and every
// per-frame decision -- bound active, Newton exit -- is uniform across the group).  16 frames per workgroup: slots record
// b = frame / 16.  Same mathematics, line by line (kl_of_eta(0) is written with one log: v / o - 1 - log v + log o = rho - 1 - log rho
// for rho = v / o; sum log pS is taken once for the log-probability, the entropy and the KL); sums over the dimensions are tree sums
// instead of left-to-right ones: the last bits of the fp64 intermediates move, nothing at the fp32 outputs' resolution.
// sums over the L lanes of a frame on the DPP network (no LDS round trips: ~40 of them sit on the kernel's dependent chain): quad_perm
// xor 1, xor 2, then row_half_mirror (lane j <-> 7 - j: the other quad of the 8) and row_mirror (j <-> 15 - j: the other half of the 16);
// both sides of every exchange add the same two numbers, so all L lanes hold the bitwise-identical sum
template <int CTRL>
GRL_DEVINL float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
GRL_DEVINL double dpp_d(double v) {
  const long long u = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
template <int L>
GRL_DEVINL double gsum(double x) {
  x += dpp_d<0xB1>(x);
  x += dpp_d<0x4E>(x);
  if (L >= 8) x += dpp_d<0x141>(x);
  if (L >= 16) x += dpp_d<0x140>(x);
  return x;
}
template <int L>
GRL_DEVINL float gsumf(float x) {
  x += dpp_f<0xB1>(x);
  x += dpp_f<0x4E>(x);
  if (L >= 8) x += dpp_f<0x141>(x);
  if (L >= 16) x += dpp_f<0x140>(x);
  return x;
}
// Pointers of one launch.  mean / sigma / dmean / dsigma are indexed by (frame - ms_row0): the stand-alone kernel reads and writes the [B, A]
// arrays (ms_row0 = 0), the fused head kernel (below) hands over LDS images of its workgroup's 16 frames (ms_row0 = its first frame).
struct TrplPtrs {
  const float *mean, *sigma, *action, *old_mean, *old_var, *old_logp, *advantage, *value, *old_value, *value_target;
  float *dmean, *dsigma, *dvalue, *proj_mean_out, *proj_var_out;
  const double* adv_stats;
  double* slots;
  const float *tgt_mean, *tgt_S;
  int ms_row0;
};
// The body for workgroup `blk` of a launch whose workgroups have NT threads: the first TRPL_FPB * L of them carry the frames (whole waves:
// 64 / 128 / 256 threads), the others only take part in the advantage sums and in the barriers.
template <int L, int PROJ, int NT>
GRL_DEVINL void trpl_lanes_body(const TrplCfg& cfg, const TrplPtrs& q_, int B, int blk, int tid) {
  const float* __restrict__ mean = q_.mean; const float* __restrict__ sigma = q_.sigma; const float* __restrict__ action = q_.action;
  const float* __restrict__ old_mean = q_.old_mean; const float* __restrict__ old_var = q_.old_var;
  const float* __restrict__ old_logp = q_.old_logp; const float* __restrict__ advantage = q_.advantage;
  const float* __restrict__ value = q_.value; const float* __restrict__ old_value = q_.old_value;
  const float* __restrict__ value_target = q_.value_target;
  float* __restrict__ dmean = q_.dmean; float* __restrict__ dsigma = q_.dsigma; float* __restrict__ dvalue = q_.dvalue;
  float* __restrict__ proj_mean_out = q_.proj_mean_out; float* __restrict__ proj_var_out = q_.proj_var_out;
  const double* __restrict__ adv_stats = q_.adv_stats; double* __restrict__ slots = q_.slots;
  const float* __restrict__ tgt_mean = q_.tgt_mean; const float* __restrict__ tgt_S = q_.tgt_S;
  constexpr int NW = (NT + 63) / 64, NWL = (TRPL_FPB * L + 63) / 64;
  static_assert((TRPL_FPB * L) % 64 == 0 && NT >= TRPL_FPB * L, "the frames sit on whole waves");
  const int A = cfg.A;
  const bool live = tid < TRPL_FPB * L;
  const int frame = blk * TRPL_FPB + (int)(tid / L), i = tid & (L - 1);
  const bool fr = live && frame < B, act = fr && i < A;
  const int b = fr ? frame : B - 1, ii = i < A ? i : A - 1;   // clamped: the padding lanes compute duplicates that every sum masks out
  const bool ext = tgt_mean != nullptr;
#define M(x) (act ? (x) : 0.0)
  // One rank: every workgroup sums the batch's advantages itself (trpl.py:248-252, 286-289) -- B floats, thread-strided partial sums,
  // wave butterflies, the waves in order: the SAME order in every workgroup and every run (bitwise reproducible), and one launch plus
  // one cross-lane dependency fewer on the step's chain than the separate statistics kernel (which the data-parallel step keeps: its
  // sums are all-reduced).
  __shared__ double adv_sh[NW][2];
  double adv_sum0 = 0.0, adv_sum1 = 0.0;
  if (cfg.adv_local) {
    for (int k = tid; k < B; k += NT) {
      const double a_ = advantage[k];
      adv_sum0 += a_;
      adv_sum1 += a_ * a_;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { adv_sum0 += __shfl_xor(adv_sum0, off, 64); adv_sum1 += __shfl_xor(adv_sum1, off, 64); }
    if ((tid & 63) == 0) { adv_sh[tid >> 6][0] = adv_sum0; adv_sh[tid >> 6][1] = adv_sum1; }
    __syncthreads();
    adv_sum0 = adv_sh[0][0];
    adv_sum1 = adv_sh[0][1];
#pragma unroll
    for (int w = 1; w < NW; ++w) { adv_sum0 += adv_sh[w][0]; adv_sum1 += adv_sh[w][1]; }
  }
  __shared__ double red[NWL][11];
  __shared__ float redm[NWL][2];
  if (live) {   // (wave-uniform: the frames sit on whole waves)
  const size_t k = (size_t)b * A + ii, kms = (size_t)(b - q_.ms_row0) * A + ii;
  const double mu = mean[kms], sg = sigma[kms], mo = old_mean[k], So = old_var[k], ac = action[k];
  const double S = sg * sg;                  // policy covariance diagonal == "std" seen by the projection (trpl.py:241)
  const double t = S * S, o = So * So;       // kl_projection_layer.py:60-63: covariance(std) = std**2
  // ---- mean projection (base_projection_layer.py:71-100)
  double mp;
  { const double d = (mu - mo) / So; mp = gsum<L>(M(d * d)); }
  if (PROJ == 0) mp *= 0.5;
  const bool m_act = mp > cfg.mean_bound;
  double omega = 0.0, D = 1.0;
  if (m_act) { omega = sqrt(mp / cfg.mean_bound) - 1.0; D = 1.0 + omega + 1e-16; }
  double pm = m_act ? (mu + omega * mo) / D : mu;
  // ---- covariance projection (derivation: the header above)
  double eta = 0.0, v, pS;
  bool c_act;
  if (PROJ == 0) {
    const double rho0 = t / o;
    c_act = 0.5 * gsum<L>(M(rho0 - 1.0 - log(rho0))) > cfg.cov_bound;
    if (c_act) {
      const double cr = o / t;
      {   // phase 1, fp32 (hardware log2 / reciprocal)
        const float crf = (float)cr, bound = (float)cfg.cov_bound, om = 1.f - crf;
        float e32 = 0.f;
        for (int it = 0; it < 60; ++it) {
          const float r1 = __builtin_amdgcn_rcpf(e32 + 1.f);
          const float rden = __builtin_amdgcn_rcpf(e32 + crf), rho = (e32 + 1.f) * rden;
          float f = gsumf<L>(act ? rho - 1.f - 0.69314718056f * __builtin_amdgcn_logf(rho) : 0.f);
          float df = gsumf<L>(act ? om * om * r1 * rden * rden : 0.f);
          f = 0.5f * f - bound;
          df *= 0.5f;
          if (!(f > 0.f) || !(df > 0.f)) break;
          const float step = f / df;
          e32 += step;
          if (!(step > 1e-5f * e32)) break;
        }
        if (e32 == e32 && e32 >= 0.f && e32 < 3.0e38f) eta = (double)e32;
      }
      const double om = 1.0 - cr;
      for (int it = 0; it < 100; ++it) {   // phase 2, fp64 polish
        const double den = eta + cr, rho = (eta + 1.0) / den;
        double f = gsum<L>(M(rho - 1.0 - log(rho)));
        double df = gsum<L>(M(om * om / ((eta + 1.0) * den * den)));
        f = 0.5 * f - cfg.cov_bound;
        df *= 0.5;
        if (df <= 0.0 || (it > 0 && f <= 0.0)) break;
        const double step = f / df;
        eta = fmax(eta + step, 0.0);
        if (fabs(step) <= 1e-15 * eta) break;
      }
    }
    v = (eta + 1.0) / (eta / o + 1.0 / t);
    pS = sqrt(v);
  } else {
    const double d = PROJ == 1 ? o - t : 1.0 - S / So;
    const double part = gsum<L>(M(d * d));
    c_act = part > cfg.cov_bound;
    if (c_act) eta = fabs(sqrt(part / cfg.cov_bound) - 1.0);
    const double den = 1.0 + eta + 1e-16;
    if (PROJ == 1) { v = c_act ? (t + eta * o) / den : t; pS = c_act ? sqrt(v) : S; }
    else { pS = c_act ? (S + eta * So) / den : S; v = pS * pS; }
  }
  if (ext) {
    pm = tgt_mean[k];
    pS = tgt_S[k];
    v = pS * pS;
  }
  // ---- log-prob under the projected distribution, importance weight, objective
  const double LOG2PI = 1.8378770664093454836;
  const double da = ac - pm;
  const double q = gsum<L>(M(da * da / pS)), sl = gsum<L>(M(log(pS)));
  const double lw = -0.5 * (q + A * LOG2PI + sl) - (double)old_logp[b];
  const double ratio = exp(lw);
  double adv = (double)advantage[b];
  if ((adv_stats || cfg.adv_local) && cfg.adv_count > 1.0) {
    const double s0_ = cfg.adv_local ? adv_sum0 : adv_stats[0], s1_ = cfg.adv_local ? adv_sum1 : adv_stats[1];
    const double am = s0_ / cfg.adv_count;
    double var = (s1_ - cfg.adv_count * am * am) / (cfg.adv_count - 1.0);
    double sd = var > 0.0 ? sqrt(var) : 0.0;
    if (sd < 1e-6) sd = 1e-6;
    adv = (adv - am) / sd;
  }
  double acc[11];
  for (int j = 0; j < 11; ++j) acc[j] = 0.0;
  acc[0] = -ratio * adv;
  acc[4] = ratio;
  acc[5] = ratio * ratio;
  acc[2] = 0.5 * (A * (1.0 + LOG2PI) + sl);
  // ---- trust-region regression loss and metrics
  const double dmk = (mu - pm) / pS, rr = S / pS;
  double mk = gsum<L>(M(dmk * dmk)), ck = gsum<L>(M(rr * rr));
  const double ldS = gsum<L>(M(log(S))), ldP = sl;
  double cd = 0.0, mS = 0.0, sq = 0.0;
  if (PROJ == 1) {
    const double f = pS * pS - S * S, dm = (mu - pm) / S, ds = S - pS;
    cd = gsum<L>(M(f * f)); mS = gsum<L>(M(dm * dm)); sq = gsum<L>(M(ds * ds));
  }
  if (PROJ == 2) cd = gsum<L>(M((1.0 - rr) * (1.0 - rr)));
  const double md = mk;
  mk *= 0.5;
  ck = 0.5 * (ck - A + 2.0 * ldP - 2.0 * ldS);
  acc[10] = mk + ck;
  if (PROJ == 0) { acc[1] = (mk + ck) * cfg.tr_coeff; acc[6] = mk; acc[7] = ck; }
  if (PROJ == 1) { acc[1] = (mS + sq) * cfg.tr_coeff; acc[6] = md; acc[7] = cd; }
  if (PROJ == 2) { acc[1] = (md + cd) * cfg.tr_coeff; acc[6] = md; acc[7] = cd; }
  const double c_ent = 0.5 * A * 2.8378770664093454836;
  acc[8] = c_ent + ldS;
  acc[9] = (c_ent + ldP) - (c_ent + ldS);
  float mmax = (float)acc[6], cmax = (float)fmax(acc[7], 0.0);
  // ---- gradients of actor_loss = objective + entropy bonus + trust region  (all already scaled by 1/B)
  const double w_obj = -ratio * adv * cfg.inv_batch;
  double g_pm = w_obj * da / pS;
  double g_pS = w_obj * 0.5 * (da * da / (pS * pS) - 1.0 / pS) - cfg.ent_coef * cfg.inv_batch * 0.5 / pS;
  const double ctr = cfg.tr_coeff * cfg.inv_batch;
  if (ext) { g_pm = 0.0; g_pS = 0.0; }
  else if (PROJ == 1) {
    g_pm -= ctr * 2.0 * (mu - pm) / (S * S);
    g_pS -= ctr * 2.0 * (S - pS);
  }
  double gmu, gS;
  if (m_act && !ext) {
    const double dot = gsum<L>(M(g_pm * (mo - pm) / D));
    const double kk = dot / (2.0 * (omega + 1.0) * cfg.mean_bound) * (PROJ == 0 ? 1.0 : 2.0);
    gmu = g_pm / D + kk * (mu - mo) / (So * So);
  } else gmu = g_pm;
  if (ext) gS = 0.0;
  else if (PROJ == 0) {
    const double gv = g_pS / (2.0 * pS);
    if (c_act) {
      const double dvt = v * v / (t * t * (eta + 1.0));
      const double dve = -v * v * (1.0 / o - 1.0 / t) / ((eta + 1.0) * (eta + 1.0));
      const double gk = 0.5 * (1.0 / o - 1.0 / v);
      const double denom = gsum<L>(M(gk * dve)), num = gsum<L>(M(gv * dve));
      gS = (gv * dvt - num * gk * dvt / denom) * 2.0 * S;
    } else gS = gv * 2.0 * S;
  } else if (c_act) {
    const double den = 1.0 + eta + 1e-16, deta = 1.0 / (2.0 * (eta + 1.0) * cfg.cov_bound);
    const double st_ = gsum<L>(M(PROJ == 1 ? g_pS / (2.0 * pS) * (o - v) / den : g_pS * (So - pS) / den));
    if (PROJ == 1) gS = g_pS / (2.0 * pS) * 2.0 * S / den + st_ * deta * (-4.0 * (o - t) * S);
    else gS = g_pS / den + st_ * deta * (-2.0 * (1.0 - S / So) / So);
  } else gS = g_pS;
  if (PROJ == 0) {
    gmu += ctr * (mu - pm) / (pS * pS);
    gS += ctr * (S / (pS * pS) - 1.0 / S);
  } else if (PROJ == 1) {
    const double dm = mu - pm;
    gmu += ctr * 2.0 * dm / (S * S);
    gS += ctr * (-2.0 * dm * dm / (S * S * S) + 2.0 * (S - pS));
  } else {
    gmu += ctr * 2.0 * (mu - pm) / (pS * pS);
    gS += ctr * (-2.0 * (1.0 - S / pS) / pS);
  }
  if (act) {
    dmean[kms] = (float)gmu;
    dsigma[kms] = (float)(gS * 2.0 * sg);
    if (proj_mean_out) { proj_mean_out[k] = (float)pm; proj_var_out[k] = (float)pS; }
  }
  // ---- clipped value loss (trpl.py:213-228, objectives/utils.py:5-28), l2: once per frame
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
    if (fr && i == 0) dvalue[b] = (float)(g * cfg.critic_coef * cfg.inv_batch);
  }
#undef M
  // workgroup reduction over its frames: lane 0 of every frame's group contributes; fixed order (wave butterflies, the waves in order)
  const bool lead = fr && i == 0;
  const int lane = tid & 63, wv = tid >> 6;
  for (int j = 0; j < 11; ++j) {
    double x = lead ? acc[j] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) red[wv][j] = x;
  }
  mmax = lead ? mmax : 0.f;
  cmax = lead ? cmax : 0.f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mmax = fmaxf(mmax, __shfl_xor(mmax, off, 64));
    cmax = fmaxf(cmax, __shfl_xor(cmax, off, 64));
  }
  if (lane == 0) { redm[wv][0] = mmax; redm[wv][1] = cmax; }
  }   // live
  __syncthreads();
  double* slot = slots + (size_t)blk * TRPL_SLOT;
  if (tid < 11) {
    double x = red[0][tid];
#pragma unroll
    for (int w = 1; w < NWL; ++w) x += red[w][tid];
    slot[tid < 10 ? tid : 11] = x;
  }
  if (tid == 11) {
    const int n_here = min(B - (int)(blk * TRPL_FPB), TRPL_FPB);
    slot[10] = (double)(n_here > 0 ? n_here : 0);
  }
  if (tid == 12 || tid == 13) {
    float x = redm[0][tid - 12];
#pragma unroll
    for (int w = 1; w < NWL; ++w) x = fmaxf(x, redm[w][tid - 12]);
    slot[tid] = (double)x;
  }
}
template <int L, int PROJ>
__global__ __launch_bounds__(TRPL_FPB * L) void trpl_lanes_kernel(TrplCfg cfg, TrplPtrs q_, int B) {
  trpl_lanes_body<L, PROJ, TRPL_FPB * L>(cfg, q_, B, (int)blockIdx.x, (int)threadIdx.x);
}

// slots [n_blocks][14] -> sums[12] (written, not accumulated) and maxes[2] (float bits); fixed order (grl_report.h trpl_fold_columns)
constexpr int FOLD_NT = 256;
__global__ __launch_bounds__(FOLD_NT) void trpl_fold_kernel(const double* __restrict__ slots, int n_blocks, double* __restrict__ sums,
                                                           unsigned int* __restrict__ maxes) {
  __shared__ double sh[16], part[FOLD_NT];
  trpl_fold_columns<FOLD_NT>(slots, n_blocks, sh, part);
  const int i = threadIdx.x;
  if (i < 12) sums[i] = sh[i];
  else if (i < 14) maxes[i - 12] = __float_as_uint((float)sh[i]);
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
__global__ __launch_bounds__(FOLD_NT) void trpl_report_kernel(const double* __restrict__ slots, int n_blocks, double* __restrict__ sums,
                                                             unsigned int* __restrict__ maxes, float entropy_coef, float* __restrict__ out) {
  __shared__ double sh[16], part[FOLD_NT];
  trpl_report_body<FOLD_NT>(slots, n_blocks, sums, maxes, entropy_coef, out, sh, part);
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
int grl_trpl_slot_doubles(int batch) { return TRPL_SLOT * (trpl_blocks(batch) < 1 ? 1 : trpl_blocks(batch)); }

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
  if (action_dim > 16 || action_dim < 1 || batch < 1 || !slots) return -2;
  TrplCfg c{cfg9[0], cfg9[1], cfg9[2], cfg9[3], cfg9[4], cfg9[5], cfg9[6], cfg9[7], action_dim, (int)cfg9[9]};
  const int proj = (int)cfg9[8];
  if (proj < 0 || proj > 2) return -3;
  const TrplPtrs tp{mean, sigma, action, old_mean, old_var, old_logp, advantage, value, old_value, value_target, dmean, dsigma, dvalue,
                    proj_mean, proj_var, adv_stats, slots, tgt_mean, tgt_S, 0};
#define GRL_TRPL_LAUNCH(LL, PJ)                                                                                                       \
  hipLaunchKernelGGL((trpl_lanes_kernel<LL, PJ>), dim3(trpl_blocks(batch)), dim3(TRPL_FPB * LL), 0, stream, c, tp, batch)
#define GRL_TRPL_WIDTH(PJ)                                        \
  do {                                                            \
    if (action_dim <= 4) GRL_TRPL_LAUNCH(4, PJ);                  \
    else if (action_dim <= 8) GRL_TRPL_LAUNCH(8, PJ);             \
    else GRL_TRPL_LAUNCH(16, PJ);                                 \
  } while (0)
  if (proj == 0) GRL_TRPL_WIDTH(0);
  else if (proj == 1) GRL_TRPL_WIDTH(1);
  else GRL_TRPL_WIDTH(2);
#undef GRL_TRPL_WIDTH
#undef GRL_TRPL_LAUNCH
  GRL_CHECK_LAUNCH();
  if (sums) {   // sums == NULL: the caller folds the slots later (grl_trpl_fold, on a stream of its choice: the sums are reported values only)
    hipLaunchKernelGGL(trpl_fold_kernel, dim3(1), dim3(FOLD_NT), 0, stream, slots, trpl_blocks(batch), sums, maxes);
    GRL_CHECK_LAUNCH();
  }
  return 0;
}

int grl_trpl_fold(const double* slots, int batch, double* sums, unsigned int* maxes, hipStream_t stream) {
  if (!slots || !sums || !maxes || batch < 1) return -2;
  hipLaunchKernelGGL(trpl_fold_kernel, dim3(1), dim3(FOLD_NT), 0, stream, slots, trpl_blocks(batch), sums, maxes);
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
__global__ __launch_bounds__(FOLD_NT) void trpl_fold_record_kernel(const double* __restrict__ slots, int n_blocks, double* __restrict__ rec) {
  __shared__ double sh[16], part[FOLD_NT];
  trpl_fold_columns<FOLD_NT>(slots, n_blocks, sh, part);
  if (threadIdx.x < TRPL_SLOT) rec[threadIdx.x] = sh[threadIdx.x];
}
// The same as (hi, lo) float pairs into a [world][14] region that travels with the flat gradient: this rank's row, zeros in the others
__global__ __launch_bounds__(FOLD_NT) void trpl_fold_record_pairs_kernel(const double* __restrict__ slots, int n_blocks, float2* __restrict__ region,
                                                                        int rank, int world) {
  __shared__ double sh[16], part[FOLD_NT];
  trpl_fold_columns<FOLD_NT>(slots, n_blocks, sh, part);
  trpl_write_record_pairs<FOLD_NT>(sh, region, rank, world);
}
__global__ __launch_bounds__(FOLD_NT) void trpl_report_pairs_kernel(const double* __restrict__ records, int n_records, double* __restrict__ sums,
                                                                   unsigned int* __restrict__ maxes, float entropy_coef, float* __restrict__ out) {
  __shared__ double sh[16], part[FOLD_NT];
  trpl_report_body<FOLD_NT, true>(records, n_records, sums, maxes, entropy_coef, out, sh, part);
}
int grl_trpl_fold_record_pairs(const double* slots, int batch, int rank, int world, float* region, hipStream_t stream) {
  if (!slots || !region || batch < 1 || world < 1 || rank < 0 || rank >= world) return -2;
  hipLaunchKernelGGL(trpl_fold_record_pairs_kernel, dim3(1), dim3(FOLD_NT), 0, stream, slots, trpl_blocks(batch),
                     reinterpret_cast<float2*>(region), rank, world);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_trpl_report_record_pairs(const float* region, int n_records, double* sums, unsigned int* maxes, float entropy_coef, float* out14,
                                 hipStream_t stream) {
  if (!region || !sums || !maxes || !out14 || n_records < 1) return -2;
  hipLaunchKernelGGL(trpl_report_pairs_kernel, dim3(1), dim3(FOLD_NT), 0, stream, reinterpret_cast<const double*>(region), n_records, sums,
                     maxes, entropy_coef, out14);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_trpl_fold_record(const double* slots, int batch, double* rec14, hipStream_t stream) {
  if (!slots || !rec14 || batch < 1) return -2;
  hipLaunchKernelGGL(trpl_fold_record_kernel, dim3(1), dim3(FOLD_NT), 0, stream, slots, trpl_blocks(batch), rec14);
  GRL_CHECK_LAUNCH();
  return 0;
}
int grl_trpl_report_records(const double* records, int n_records, double* sums, unsigned int* maxes, float entropy_coef, float* out14,
                            hipStream_t stream) {
  if (!records || !sums || !maxes || !out14 || n_records < 1) return -2;
  hipLaunchKernelGGL(trpl_report_kernel, dim3(1), dim3(FOLD_NT), 0, stream, records, n_records, sums, maxes, entropy_coef, out14);
  GRL_CHECK_LAUNCH();
  return 0;
}

// grl_trpl_fold + grl_trpl_loss_values in one launch (one rank: nothing to all-reduce in between)
int grl_trpl_report(const double* slots, int batch, double* sums, unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream) {
  if (!slots || !sums || !maxes || !out14 || batch < 1) return -2;
  hipLaunchKernelGGL(trpl_report_kernel, dim3(1), dim3(FOLD_NT), 0, stream, slots, trpl_blocks(batch), sums, maxes, entropy_coef, out14);
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
