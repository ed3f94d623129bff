// Fold of the fused TRPL kernel's per-workgroup slots + evaluation of the reported values (reference objectives/trpl.py:280-321) as a device
// function: the stand-alone launch (head_ops.hip grl_trpl_report) and the extra workgroup of the step's fused tail launch
// (node_ops.hip grl_fold_adam_report) run the same code.
#pragma once
#include "grl_common.h"

constexpr int TRPL_SLOT = 14;   // per-workgroup record of trpl_kernel: the 12 sums + the 2 maxes

// the first 64 threads of the workgroup take part (i = threadIdx.x < 64); sh / shm: LDS scratch [12] / [2]; the caller's __syncthreads
// must be reachable by every thread of the workgroup, so the barrier sits in the caller-visible part below
GRL_DEVINL void trpl_report_body(const double* __restrict__ slots, int n_blocks, double* __restrict__ sums, unsigned int* __restrict__ maxes,
                                 float entropy_coef, float* __restrict__ out, double* sh, unsigned int* shm) {
  const int i = threadIdx.x;
  if (i < 12) {
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += slots[(size_t)b * TRPL_SLOT + i];
    sums[i] = s;
    sh[i] = s;
  } else if (i < 14) {
    double m = 0.0;
    for (int b = 0; b < n_blocks; ++b) m = fmax(m, slots[(size_t)b * TRPL_SLOT + i]);
    maxes[i - 12] = __float_as_uint((float)m);
    shm[i - 12] = __float_as_uint((float)m);
  }
  __syncthreads();
  if (i == 0) {
    const double n = sh[10];
    const float tr = (float)(sh[1] / n), ent = -entropy_coef * (float)(sh[2] / n);
    const float actor = (float)((sh[0] + sh[1] - (double)entropy_coef * sh[2]) / n);
    out[0] = actor;
    out[1] = (float)(sh[3] / n);
    out[2] = tr;
    out[3] = ent;
    out[4] = (float)(sh[4] * sh[4] / sh[5] / n);   // exp(2 lse(lw) - lse(2 lw)) / B   (trpl.py:294-300,316)
    const float mc = (float)(sh[6] / n), cc = (float)(sh[7] / n);
    out[5] = (float)(sh[11] / n);
    out[6] = mc;
    out[7] = __uint_as_float(shm[0]);
    out[8] = cc;
    out[9] = __uint_as_float(shm[1]);
    out[10] = (float)(sh[8] / n);
    out[11] = (float)(sh[9] / n);
    out[12] = actor - (tr + ent);
    out[13] = mc + cc;   // "constraint": the projection's own measure (= kl for the KL projection)
  }
}
