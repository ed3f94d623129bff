// Fused per-edge pipeline of the separable fiber-bundle convolution (reference rows 6-7 of SURVEY.md section 8a):
//   invariants -> PolynomialFeatures(2) -> Linear(14,64)+GELU -> Linear(64,64)+GELU        (hepi.py:76-82,109-123,145-157)
//   -> kernel Linear(64,64, no bias) -> message = kernel * x_src[src]                      (conv.py:79,115-117)
//   -> sum over the edges of each destination node                                         (conv.py:141-147)
// Nothing per-edge ever reaches HBM.  Forward walks edges grouped by destination (CSR by dst), backward walks the same
// edges grouped by SOURCE so that d x_src accumulates in an LDS tile too -- no global atomics in either direction.
#include "grl_common.h"

namespace {

constexpr int C = 64;            // channels
constexpr int O = 16;            // orientations
constexpr int GROUP = 8;         // anchor nodes per workgroup iteration
constexpr int LDT = C + 4;       // padded row of an LDS activation tile
constexpr int LDW = GRL_LD(64);  // 68
constexpr int LDW1 = GRL_LD(16); // 20

struct EdgeParams {
  const float* x_src;    // [Ns,16,64]
  const float* pos_src;  // [Ns,3]
  const float* pos_dst;  // [Nd,3]
  const int* rowptr;     // [Na+1]  anchor-sorted CSR (anchor = dst in forward, src in backward)
  const int* e_src;      // [E] source node of each edge, in anchor-sorted order
  const int* e_dst;      // [E] destination node of each edge, same order
  const float* grid;     // [16,3] (z = 0 for the S1 grid)
  const float* W1;       // [64,14]
  const float* b1;       // [64]
  const float* W2;       // [64,64]
  const float* b2;       // [64]
  const float* Wk;       // [64,64]
  int n_anchor;
  int dim;
};

// Polynomial features of (a, b) in the reference order (ponita.py:233-244):
//   k:  0  1 | 2   3   4   5 | 6    7    8    9    10   11   12   13
//       a  b | aa  ab  ba  bb| aaa  aab  aba  abb  baa  bab  bba  bbb        (columns 14, 15 are zero padding)
// split into this lane's two fragments: lane half h owns k = 4h..4h+3 and k = 8+4h..8+4h+3.
GRL_DEVINL void poly_frags(float a, float b, int h, float4& f0, float4& f1) {
  const float aa = a * a, ab = a * b, bb = b * b;
  if (h == 0) {
    f0 = make_float4(a, b, aa, ab);
    f1 = make_float4(ab * a, ab * b, ab * a, ab * b);
  } else {
    f0 = make_float4(ab, bb, aa * a, aa * b);
    f1 = make_float4(bb * a, bb * b, 0.f, 0.f);
  }
}

// One pass of the chain for this lane's row.  Returns the kernel fragments K (8 float4); optionally keeps the
// pre-activation derivatives for the backward pass.
template <bool BWD>
GRL_DEVINL void edge_chain(const float* W1s, const float* b1s, const float* W2s, const float* b2s, const float* Wks,
                           float a, float b, float4 (&kf)[8], float4 (&g1)[8], float4 (&gp1)[8], float4 (&g2)[8],
                           float4 (&gp2)[8], float4 (&phi)[2]) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  poly_frags(a, b, h, phi[0], phi[1]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bb = *reinterpret_cast<const float4*>(b1s + 32 * nt + 8 * q + 4 * h);
      acc[4 * q] = bb.x; acc[4 * q + 1] = bb.y; acc[4 * q + 2] = bb.z; acc[4 * q + 3] = bb.w;
    }
    mma_wx<16>(W1s + (32 * nt + i) * LDW1 + 4 * h, phi, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g1[4 * nt + q] = make_float4(gelu_f(acc[4 * q]), gelu_f(acc[4 * q + 1]), gelu_f(acc[4 * q + 2]), gelu_f(acc[4 * q + 3]));
      if (BWD)
        gp1[4 * nt + q] = make_float4(gelu_grad_f(acc[4 * q]), gelu_grad_f(acc[4 * q + 1]), gelu_grad_f(acc[4 * q + 2]),
                                      gelu_grad_f(acc[4 * q + 3]));
    }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bb = *reinterpret_cast<const float4*>(b2s + 32 * nt + 8 * q + 4 * h);
      acc[4 * q] = bb.x; acc[4 * q + 1] = bb.y; acc[4 * q + 2] = bb.z; acc[4 * q + 3] = bb.w;
    }
    mma_wx<64>(W2s + (32 * nt + i) * LDW + 4 * h, g1, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g2[4 * nt + q] = make_float4(gelu_f(acc[4 * q]), gelu_f(acc[4 * q + 1]), gelu_f(acc[4 * q + 2]), gelu_f(acc[4 * q + 3]));
      if (BWD)
        gp2[4 * nt + q] = make_float4(gelu_grad_f(acc[4 * q]), gelu_grad_f(acc[4 * q + 1]), gelu_grad_f(acc[4 * q + 2]),
                                      gelu_grad_f(acc[4 * q + 3]));
    }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    f32x16 acc = zero16();
    mma_wx<64>(Wks + (32 * nt + i) * LDW + 4 * h, g2, acc);
    acc_to_frag(acc, kf[4 * nt], kf[4 * nt + 1], kf[4 * nt + 2], kf[4 * nt + 3]);
  }
}

GRL_DEVINL void edge_invariants(const EdgeParams& p, const float* grid_s, int e, int o, float& a, float& b) {
  const int s = p.e_src[e], d = p.e_dst[e];
  float rx = p.pos_src[3 * s] - p.pos_dst[3 * d];
  float ry = p.pos_src[3 * s + 1] - p.pos_dst[3 * d + 1];
  float rz = (p.dim == 2) ? 0.f : p.pos_src[3 * s + 2] - p.pos_dst[3 * d + 2];
  const float gx = grid_s[3 * o], gy = grid_s[3 * o + 1], gz = grid_s[3 * o + 2];
  a = rx * gx + ry * gy + rz * gz;                      // hepi.py:115
  rx -= a * gx; ry -= a * gy; rz -= a * gz;
  b = sqrtf(rx * rx + ry * ry + rz * rz);               // hepi.py:117
}

// LDS: weights + the anchor tile
struct Smem {
  float W1s[64 * LDW1];
  float W2s[64 * LDW];
  float Wks[64 * LDW];
  float b1s[64];
  float b2s[64];
  float grid_s[64];
};

GRL_DEVINL void load_weights(Smem& s, const EdgeParams& p) {
  for (int idx = threadIdx.x; idx < 64 * LDW1; idx += blockDim.x) {
    const int r = idx / LDW1, c = idx - r * LDW1;
    s.W1s[idx] = (c < 14) ? p.W1[r * 14 + c] : 0.f;
  }
  stage_matrix(s.W2s, p.W2, 64, 64, LDW);
  stage_matrix(s.Wks, p.Wk, 64, 64, LDW);
  for (int idx = threadIdx.x; idx < 64; idx += blockDim.x) {
    s.b1s[idx] = p.b1[idx];
    s.b2s[idx] = p.b2[idx];
    s.grid_s[idx] = (idx < 48) ? p.grid[idx] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void edge_conv_fwd_kernel(EdgeParams p, float* __restrict__ x1 /*[Nd,16,64]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  Smem& s = *reinterpret_cast<Smem*>(smem_raw);
  float* tile = smem_raw + sizeof(Smem) / 4;  // [GROUP*16][LDT]
  load_weights(s, p);
  for (int idx = threadIdx.x; idx < GROUP * O * LDT; idx += blockDim.x) tile[idx] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int o = r & 15, el = r >> 4;
  const int n_groups = (p.n_anchor + GROUP - 1) / GROUP;
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int d0 = g * GROUP, d1 = min(d0 + GROUP, p.n_anchor);
    const int e0 = p.rowptr[d0], e1 = p.rowptr[d1];
    const int n_pass = (e1 - e0 + 1) >> 1;
    for (int ps = wave; ps < n_pass; ps += 4) {
      const int e = e0 + 2 * ps + el;
      const bool valid = e < e1;
      const int ee = valid ? e : e0;
      float a, b;
      edge_invariants(p, s.grid_s, ee, o, a, b);
      float4 kf[8], g1[8], gp1[8], g2[8], gp2[8], phi[2];
      edge_chain<false>(s.W1s, s.b1s, s.W2s, s.b2s, s.Wks, a, b, kf, g1, gp1, g2, gp2, phi);
      if (valid) {
        const float4* xs = reinterpret_cast<const float4*>(p.x_src + ((size_t)p.e_src[ee] * O + o) * C) + h;
        float* trow = tile + ((p.e_dst[ee] - d0) * O + o) * LDT + 4 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float4 m = f4_mul(kf[t], xs[2 * t]);  // float4 index 2t + h  <->  floats 8t + 4h
          atomicAdd(trow + 8 * t, m.x);
          atomicAdd(trow + 8 * t + 1, m.y);
          atomicAdd(trow + 8 * t + 2, m.z);
          atomicAdd(trow + 8 * t + 3, m.w);
        }
      }
    }
    __syncthreads();
    // flush the tile (coalesced float4 rows) and clear it
    const int n_rows = (d1 - d0) * O;
    for (int idx = threadIdx.x; idx < n_rows * (C / 4); idx += blockDim.x) {
      const int row = idx >> 4, c4 = idx & 15;
      float4* src = reinterpret_cast<float4*>(tile + row * LDT) + c4;
      reinterpret_cast<float4*>(x1 + ((size_t)d0 * O + row) * C)[c4] = *src;
      *src = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Walks edges grouped by SOURCE node.  Per pass (2 edges x 16 orientations = 32 rows per wave):
//   recompute chain; dM = d x1[dst]; dK = dM * x_src; d x_src += dM * K (LDS tile);
//   dWk += dK^T g2; dG2 = dK Wk; dZ2 = dG2 * gelu'(z2); dW2 += dZ2^T g1; db2 += colsum dZ2;
//   dG1 = dZ2 W2; dZ1 = dG1 * gelu'(z1); dW1 += dZ1^T phi; db1 += colsum dZ1.
// Weight-gradient accumulators live in registers for the whole kernel (one wave per SIMD, 512-VGPR budget) and are
// written once per wave to a partial slab: partial[(block*4 + wave)][9344] = [W1 64x14 | b1 64 | W2 64x64 | b2 64 | Wk 64x64].
constexpr int EDGE_PARTIAL = 64 * 14 + 64 + 64 * 64 + 64 + 64 * 64;

GRL_DEVINL void store_frags_rowmajor(float* buf /*wave-private [32][LDT]*/, int r, int h, const float4 (&f)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) *reinterpret_cast<float4*>(buf + r * LDT + 8 * t + 4 * h) = f[t];
}

__global__ __launch_bounds__(256, 1) void edge_conv_bwd_kernel(EdgeParams p, const float* __restrict__ dx1 /*[Nd,16,64]*/,
                                                                float* __restrict__ dx_src /*[Ns,16,64]*/,
                                                                float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  Smem& s = *reinterpret_cast<Smem*>(smem_raw);
  float* tile = smem_raw + sizeof(Smem) / 4;          // [GROUP*16][LDT]  d x_src accumulators
  float* tbuf = tile + GROUP * O * LDT;               // 4 waves x 2 x [32][LDT]
  load_weights(s, p);
  for (int idx = threadIdx.x; idx < GROUP * O * LDT; idx += blockDim.x) tile[idx] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int o = r & 15, el = r >> 4;
  float* T0 = tbuf + wave * 2 * 32 * LDT;
  float* T1 = T0 + 32 * LDT;

  f32x16 dWk[2][2], dW2[2][2], dW1[2];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_) {
    dW1[a_] = zero16();
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) { dWk[a_][b_] = zero16(); dW2[a_][b_] = zero16(); }
  }
  float db1 = 0.f, db2 = 0.f;  // lane = column (64 columns)

  const int n_groups = (p.n_anchor + GROUP - 1) / GROUP;
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int s0 = g * GROUP, s1 = min(s0 + GROUP, p.n_anchor);
    const int e0 = p.rowptr[s0], e1 = p.rowptr[s1];
    const int n_pass = (e1 - e0 + 1) >> 1;
    for (int ps = wave; ps < n_pass; ps += 4) {
      const int e = e0 + 2 * ps + el;
      const bool valid = e < e1;
      const int ee = valid ? e : e0;
      float a, b;
      edge_invariants(p, s.grid_s, ee, o, a, b);
      float4 kf[8], g1[8], gp1[8], g2[8], gp2[8], phi[2];
      edge_chain<true>(s.W1s, s.b1s, s.W2s, s.b2s, s.Wks, a, b, kf, g1, gp1, g2, gp2, phi);

      const int src = p.e_src[ee], dst = p.e_dst[ee];
      const float4* xs = reinterpret_cast<const float4*>(p.x_src + ((size_t)src * O + o) * C) + h;
      const float4* dm = reinterpret_cast<const float4*>(dx1 + ((size_t)dst * O + o) * C) + h;
      float4 dK[8];
      float* trow = tile + ((src - s0) * O + o) * LDT + 4 * h;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float4 d = dm[2 * t];
        if (!valid) d = make_float4(0.f, 0.f, 0.f, 0.f);
        dK[t] = f4_mul(d, xs[2 * t]);
        const float4 dx = f4_mul(d, kf[t]);
        if (valid) {
          atomicAdd(trow + 8 * t, dx.x);
          atomicAdd(trow + 8 * t + 1, dx.y);
          atomicAdd(trow + 8 * t + 2, dx.z);
          atomicAdd(trow + 8 * t + 3, dx.w);
        }
      }
      // ---- Wk: dWk[c][k] += sum_r dK[r][c] g2[r][k]
      store_frags_rowmajor(T0, r, h, dK);
      store_frags_rowmajor(T1, r, h, g2);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          mma_tn<32>(T0 + 4 * h * LDT + 32 * ct + r, LDT, T1 + 4 * h * LDT + 32 * kt + r, LDT, dWk[ct][kt]);
      // ---- dZ2 = (dK Wk) * gelu'(z2)
      float4 dz2[8];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16 acc = zero16();
        mma_wTy<64>(s.Wks + 4 * h * LDW + 32 * kt + r, LDW, dK, acc);
        float4 f0, f1, f2, f3;
        acc_to_frag(acc, f0, f1, f2, f3);
        dz2[4 * kt] = f4_mul(f0, gp2[4 * kt]);
        dz2[4 * kt + 1] = f4_mul(f1, gp2[4 * kt + 1]);
        dz2[4 * kt + 2] = f4_mul(f2, gp2[4 * kt + 2]);
        dz2[4 * kt + 3] = f4_mul(f3, gp2[4 * kt + 3]);
      }
      store_frags_rowmajor(T0, r, h, dz2);
      store_frags_rowmajor(T1, r, h, g1);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          mma_tn<32>(T0 + 4 * h * LDT + 32 * nt + r, LDT, T1 + 4 * h * LDT + 32 * kt + r, LDT, dW2[nt][kt]);
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) db2 += T0[rr * LDT + lane];
      // ---- dZ1 = (dZ2 W2) * gelu'(z1)
      float4 dz1[8];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16 acc = zero16();
        mma_wTy<64>(s.W2s + 4 * h * LDW + 32 * kt + r, LDW, dz2, acc);
        float4 f0, f1, f2, f3;
        acc_to_frag(acc, f0, f1, f2, f3);
        dz1[4 * kt] = f4_mul(f0, gp1[4 * kt]);
        dz1[4 * kt + 1] = f4_mul(f1, gp1[4 * kt + 1]);
        dz1[4 * kt + 2] = f4_mul(f2, gp1[4 * kt + 2]);
        dz1[4 * kt + 3] = f4_mul(f3, gp1[4 * kt + 3]);
      }
      store_frags_rowmajor(T0, r, h, dz1);
      // phi as row-major [32][16] inside T1 (columns 16..31 of the tile are never used downstream)
      *reinterpret_cast<float4*>(T1 + r * LDT + 4 * h) = phi[0];
      *reinterpret_cast<float4*>(T1 + r * LDT + 8 + 4 * h) = phi[1];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        mma_tn<32>(T0 + 4 * h * LDT + 32 * nt + r, LDT, T1 + 4 * h * LDT + r, LDT, dW1[nt]);
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) db1 += T0[rr * LDT + lane];
    }
    __syncthreads();
    const int n_rows = (s1 - s0) * O;
    for (int idx = threadIdx.x; idx < n_rows * (C / 4); idx += blockDim.x) {
      const int row = idx >> 4, c4 = idx & 15;
      float4* src = reinterpret_cast<float4*>(tile + row * LDT) + c4;
      reinterpret_cast<float4*>(dx_src + ((size_t)s0 * O + row) * C)[c4] = *src;
      *src = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
  }

  // ---- write this wave's weight-gradient partial.  acc element rho of lane (j = r, h): D[n = 8q+4h+u][col j]
  float* out = partial + (size_t)(blockIdx.x * 4 + wave) * EDGE_PARTIAL;
  float* oW1 = out, *ob1 = out + 64 * 14, *oW2 = ob1 + 64, *ob2 = oW2 + 64 * 64, *oWk = ob2 + 64;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) {
      const int n = 32 * nt + (rho & 3) + 8 * (rho >> 2) + 4 * h;
      if (r < 14) oW1[n * 14 + r] = dW1[nt][rho];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        oW2[n * 64 + 32 * kt + r] = dW2[nt][kt][rho];
        oWk[n * 64 + 32 * kt + r] = dWk[nt][kt][rho];
      }
    }
  ob1[lane] = db1;
  ob2[lane] = db2;
}

}  // namespace

extern "C" {

int grl_edge_partial_size() { return EDGE_PARTIAL; }
int grl_edge_bwd_blocks(int n_anchor) {
  const int n_groups = (n_anchor + GROUP - 1) / GROUP;
  return n_groups < 256 ? n_groups : 256;
}

int grl_edge_conv_fwd(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* Wk, float* x1, hipStream_t stream) {
  if (n_dst <= 0) return 0;
  EdgeParams p{x_src, pos_src, pos_dst, rowptr, e_src, e_dst, grid, W1, b1, W2, b2, Wk, n_dst, dim};
  const int n_groups = (n_dst + GROUP - 1) / GROUP;
  const int blocks = n_groups < 512 ? n_groups : 512;
  const size_t smem = sizeof(Smem) + sizeof(float) * GROUP * O * LDT;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)edge_conv_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  hipLaunchKernelGGL(edge_conv_fwd_kernel, dim3(blocks), dim3(256), smem, stream, p, x1);
  GRL_CHECK_LAUNCH();
  return 0;
}

// rowptr/e_src/e_dst here are the SOURCE-sorted CSR of the same edge set.  partial must hold
// grl_edge_bwd_blocks(n_src)*4 rows of grl_edge_partial_size() floats.
int grl_edge_conv_bwd(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr_s,
                      const int* e_src_s, const int* e_dst_s, int n_src, const float* grid, int dim, const float* W1,
                      const float* b1, const float* W2, const float* b2, const float* Wk, const float* dx1, float* dx_src,
                      float* partial, hipStream_t stream) {
  if (n_src <= 0) return 0;
  EdgeParams p{x_src, pos_src, pos_dst, rowptr_s, e_src_s, e_dst_s, grid, W1, b1, W2, b2, Wk, n_src, dim};
  const int blocks = grl_edge_bwd_blocks(n_src);
  const size_t smem = sizeof(Smem) + sizeof(float) * (GROUP * O * LDT + 4 * 2 * 32 * LDT);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)edge_conv_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  hipLaunchKernelGGL(edge_conv_bwd_kernel, dim3(blocks), dim3(256), smem, stream, p, dx1, dx_src, partial);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
