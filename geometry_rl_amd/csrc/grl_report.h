// Fold of the fused TRPL kernel's per-workgroup slots + evaluation of the reported values (reference objectives/trpl.py:280-321) as a device
// function: the stand-alone launch (head_ops.hip grl_trpl_report) and the extra workgroup of the step's fused tail launch
// (node_ops.hip grl_fold_adam_report) run the same code.
#pragma once
#include "grl_common.h"

constexpr int TRPL_SLOT = 14;   // per-workgroup record of the fused loss kernel: the 12 sums + the 2 maxes
constexpr int TRPL_FPB = 16;    // frames per workgroup of that kernel (trpl_lanes_kernel, head_ops.hip)
inline int trpl_blocks(int batch) { return (batch + TRPL_FPB - 1) / TRPL_FPB; }

// Column sums (12) and maxes (2) of the slot records [n_blocks][14] by a whole workgroup: thread (part p = tid / 16, column c = tid % 16)
// adds the records p, p + NP, ... (four independent running sums: the loads of a 256-record fold -- 4096 frames -- were one dependent
// ~0.4 us round trip each when a single thread walked a column: +92 us on the step's tail), the NP partial results are combined in part
// order.  A fixed order: bitwise reproducible.  Result in sh[0..13] (sums 0..11, maxes 12..13 as doubles), valid after the closing barrier.
// PAIRS: a record value is stored as two floats (hi, lo) with hi + lo = the double to ~2^-48 -- the form that survives a float SUM
// all-reduce of a buffer in which every rank fills its own row and zeroes the others (x + 0 is exact): the ranks' loss records then
// travel with the flat gradient in ONE collective (PolicyUpdater._plan).  Same 8 bytes per value as a double.
template <bool PAIRS>
GRL_DEVINL double trpl_slot_value(const double* __restrict__ slots, size_t i) {
  if constexpr (PAIRS) {
    const float2 v = reinterpret_cast<const float2*>(slots)[i];
    return (double)v.x + (double)v.y;
  } else {
    return slots[i];
  }
}

template <int NT, bool PAIRS = false>
GRL_DEVINL void trpl_fold_columns(const double* __restrict__ slots_, int n_blocks, double* sh /* [14] */, double* part /* [NT / 16][16] */) {
  struct { const double* p; GRL_DEVINL double operator[](size_t i) const { return trpl_slot_value<PAIRS>(p, i); } } slots{slots_};
  constexpr int NP = NT / 16;
  const int c = threadIdx.x & 15, p = threadIdx.x >> 4;
  if (c < TRPL_SLOT && (int)threadIdx.x < NT) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = p;
    if (c < 12) {
      for (; b + 3 * NP < n_blocks; b += 4 * NP) {
        a0 += slots[(size_t)b * TRPL_SLOT + c]; a1 += slots[(size_t)(b + NP) * TRPL_SLOT + c];
        a2 += slots[(size_t)(b + 2 * NP) * TRPL_SLOT + c]; a3 += slots[(size_t)(b + 3 * NP) * TRPL_SLOT + c];
      }
      for (; b < n_blocks; b += NP) a0 += slots[(size_t)b * TRPL_SLOT + c];
      part[p * 16 + c] = (a0 + a1) + (a2 + a3);
    } else {
      for (; b < n_blocks; b += NP) a0 = fmax(a0, slots[(size_t)b * TRPL_SLOT + c]);
      part[p * 16 + c] = a0;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < TRPL_SLOT) {
    double r = part[threadIdx.x];
#pragma unroll
    for (int q = 1; q < NP; ++q) r = threadIdx.x < 12 ? r + part[q * 16 + threadIdx.x] : fmax(r, part[q * 16 + threadIdx.x]);
    sh[threadIdx.x] = r;
  }
  __syncthreads();
}

// fold + reported values; NT threads of the workgroup take part (all of the workgroup's threads must call: barriers inside)
template <int NT, bool PAIRS = false>
GRL_DEVINL void trpl_report_body(const double* __restrict__ slots, int n_blocks, double* __restrict__ sums, unsigned int* __restrict__ maxes,
                                 float entropy_coef, float* __restrict__ out, double* sh /* [14] */, double* part /* [NT / 16][16] */) {
  trpl_fold_columns<NT, PAIRS>(slots, n_blocks, sh, part);
  const int i = threadIdx.x;
  if (i < 12) sums[i] = sh[i];
  else if (i < 14) maxes[i - 12] = __float_as_uint((float)sh[i]);
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
    out[7] = (float)sh[12];
    out[8] = cc;
    out[9] = (float)sh[13];
    out[10] = (float)(sh[8] / n);
    out[11] = (float)(sh[9] / n);
    out[12] = actor - (tr + ent);
    out[13] = mc + cc;   // "constraint": the projection's own measure (= kl for the KL projection)
  }
}

// This rank's record (sh[0..13], from trpl_fold_columns) as (hi, lo) float pairs into its row of a [world][14] region, zeros in the other rows
// (the region travels with the flat gradient in ONE float SUM all-reduce: x + 0 is exact, hi + lo restores the double to ~2^-48).
template <int NT>
GRL_DEVINL void trpl_write_record_pairs(const double* sh, float2* __restrict__ region, int rank, int world) {
  for (int i = threadIdx.x; i < world * TRPL_SLOT; i += NT) {
    float2 v = make_float2(0.f, 0.f);
    if (i / TRPL_SLOT == rank) {
      const double x = sh[i % TRPL_SLOT];
      v.x = (float)x;
      v.y = (v.x - v.x == 0.f) ? (float)(x - (double)v.x) : 0.f;   // (inf / nan stay what they are)
    }
    region[i] = v;
  }
}
