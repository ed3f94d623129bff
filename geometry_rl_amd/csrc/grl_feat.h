// Node features of the batched graph (pyg_data/rigid_tasks_data.py:150-250 and the cloth / rope counterparts) as a device BODY, shared by
// the stand-alone launch (train_ops.hip: grl_build_features[_bump]) and the step's merged head launch (node_ops.hip: grl_step_head).
// Every 3-vector feature of every node type is "slice A of an observation group [- slice B]", gathered per node; one launch
// writes them all (the torch formulation is ~30 tiny split / index / stack / cat launches per network and step).
#pragma once
#include "grl_common.h"

namespace {

struct FeatDesc {
  float* out;            // output rows
  const float* a;        // term A: group tensor [B, a_stride], vector j of a sample at a + b*a_stride + a_off + 3*j (j = 0 if a_bcast)
  const float* b;        // optional term B (subtracted)
  const long long* gather;  // optional: node n -> b*n_per + j (compacted main node type); else n = b*n_per + j
  int out_row_stride, out_col;      // floats
  int rows_per_sample, row_off;     // > 0: output row = b*rows_per_sample + row_off + j (dense critic input); else row = n
  int n_nodes, n_per;
  int a_stride, a_off, a_bcast;
  int b_stride, b_off, b_bcast;
  int onehot_col, n_types;          // onehot_col >= 0: also write the node-type one-hot into columns [0, n_types)
};
constexpr int FEAT_MAX = 24;
struct FeatDescs { FeatDesc d[FEAT_MAX]; };

// workgroup (bx of nbx) of descriptor `desc`; `first`: the one workgroup of the launch that advances the optional step count
GRL_DEVINL void build_features_body(const FeatDescs& all, int* __restrict__ bump, int bx, int nbx, int desc, bool first) {
  // (optional) the optimizer's step count rides on this launch -- the first of a lane's recorded step: one thread advances it, every
  // later kernel of the lane (Adam) reads the new value; a separate one-element launch was ~6 us of every step's chain
  if (bump && first && threadIdx.x == 0) bump[0] += 1;
  const FeatDesc& f = all.d[desc];
  for (int n = bx * 256 + (int)threadIdx.x; n < f.n_nodes; n += nbx * 256) {
    const long long flat = f.gather ? f.gather[n] : (long long)n;
    const int b = (int)(flat / f.n_per), j = (int)(flat - (long long)b * f.n_per);
    float x = 0.f, y = 0.f, z = 0.f;
    if (f.a) {
      const float* p = f.a + (size_t)b * f.a_stride + f.a_off + (f.a_bcast ? 0 : 3 * j);
      x = p[0]; y = p[1]; z = p[2];
    }
    if (f.b) {
      const float* p = f.b + (size_t)b * f.b_stride + f.b_off + (f.b_bcast ? 0 : 3 * j);
      x -= p[0]; y -= p[1]; z -= p[2];
    }
    const size_t row = f.rows_per_sample > 0 ? (size_t)b * f.rows_per_sample + f.row_off + j : (size_t)n;
    float* o = f.out + row * f.out_row_stride;
    o[f.out_col] = x; o[f.out_col + 1] = y; o[f.out_col + 2] = z;
    if (f.onehot_col >= 0)
      for (int c = 0; c < f.n_types; ++c) o[c] = c == f.onehot_col ? 1.f : 0.f;
  }
}

// descs: HOST array of n_desc <= FEAT_MAX records of 18 8-byte words each (include/grl_hip.h grl_build_features); -> the largest node count
inline int feat_fill(FeatDescs& all, const long long* descs, int n_desc) {
  int max_nodes = 1;
  for (int i = 0; i < n_desc; ++i) {
    const long long* w = descs + 18 * i;
    FeatDesc& f = all.d[i];
    f.out = reinterpret_cast<float*>(w[0]);
    f.a = reinterpret_cast<const float*>(w[1]);
    f.b = reinterpret_cast<const float*>(w[2]);
    f.gather = reinterpret_cast<const long long*>(w[3]);
    f.out_row_stride = (int)w[4]; f.out_col = (int)w[5]; f.rows_per_sample = (int)w[6]; f.row_off = (int)w[7];
    f.n_nodes = (int)w[8]; f.n_per = (int)w[9];
    f.a_stride = (int)w[10]; f.a_off = (int)w[11]; f.a_bcast = (int)w[12];
    f.b_stride = (int)w[13]; f.b_off = (int)w[14]; f.b_bcast = (int)w[15];
    f.onehot_col = (int)w[16]; f.n_types = (int)w[17];
    if (f.n_nodes > max_nodes) max_nodes = f.n_nodes;
  }
  return max_nodes;
}

}  // namespace
