// Shared device helpers for the gfx950 (CDNA4) kernels of the geometry_rl policy-update path.
//
// MFMA convention used by every dense kernel in this directory ("transposed register chain"):
//   v_mfma_f32_32x32x2_f32 computes D[i][j] += A[i][k] * B[k][j] with lane l supplying A[i=l&31][k=l>>5] and
//   B[k=l>>5][j=l&31]; D element rho of lane l is D[(rho&3) + 8*(rho>>2) + 4*(l>>5)][l&31].
//   We always put the WEIGHT on the A side (lane index = output feature n) and the ACTIVATION row on the B side
//   (lane index = row r).  An activation row r is held as K/8 float4 "fragments": lane (r, h=l>>5) owns
//   X[r][8t + 4h + u], t = 0..K/8-1, u = 0..3.  With that k-ordering the 32x32 accumulator of one product
//   (16 floats per lane: features n = 32*nt + 8q + 4h + u in acc[4q+u]) IS the fragment set {4nt+q} of the next
//   product -- chains of Linear layers stay in registers, no LDS round trip for activations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GRL_DEVINL __device__ __forceinline__

// padded LDS leading dimension for a [rows][K] fp32 weight matrix read with ds_read_b128 (one access width of padding)
#define GRL_LD(K) ((K) + 4)

GRL_DEVINL f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

GRL_DEVINL f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// acc(32 n x 32 r) += W[n0 + i][0..K) . X[r][0..K)      (W row-major in LDS, leading dim ldw, b128 reads)
//   wrow = &W[(n0 + (lane&31)) * ldw + 4*(lane>>5)]
template <int K>
GRL_DEVINL void mma_wx(const float* wrow, const float4 (&x)[K / 8], f32x16& acc) {
#pragma unroll
  for (int t = 0; t < K / 8; ++t) {
    const float4 w = *reinterpret_cast<const float4*>(wrow + 8 * t);
    acc = mfma32(w.x, x[t].x, acc);
    acc = mfma32(w.y, x[t].y, acc);
    acc = mfma32(w.z, x[t].z, acc);
    acc = mfma32(w.w, x[t].w, acc);
  }
}

// acc(32 k x 32 r) += sum_n W[n][k0 + i] * Y[r][n], n = 0..N)   (same LDS image of W, read "down a column", b32 reads)
//   wcol = &W[(4*(lane>>5)) * ldw + k0 + (lane&31)]
template <int N>
GRL_DEVINL void mma_wTy(const float* wcol, int ldw, const float4 (&y)[N / 8], f32x16& acc) {
#pragma unroll
  for (int t = 0; t < N / 8; ++t) {
    const float* p = wcol + (8 * t) * ldw;
    acc = mfma32(p[0], y[t].x, acc);
    acc = mfma32(p[ldw], y[t].y, acc);
    acc = mfma32(p[2 * ldw], y[t].z, acc);
    acc = mfma32(p[3 * ldw], y[t].w, acc);
  }
}

// acc(32 n x 32 k) += sum_{r<R} P[r][n0 + i] * Q[r][k0 + j]   (both operands row-major [r][ld] in LDS, b32 reads)
//   pcol = &P[(4*(lane>>5)) * ldp + n0 + (lane&31)],  qcol = &Q[(4*(lane>>5)) * ldq + k0 + (lane&31)]
template <int R>
GRL_DEVINL void mma_tn(const float* pcol, int ldp, const float* qcol, int ldq, f32x16& acc) {
#pragma unroll
  for (int t = 0; t < R / 8; ++t) {
    const float* p = pcol + (8 * t) * ldp;
    const float* q = qcol + (8 * t) * ldq;
    acc = mfma32(p[0], q[0], acc);
    acc = mfma32(p[ldp], q[ldq], acc);
    acc = mfma32(p[2 * ldp], q[2 * ldq], acc);
    acc = mfma32(p[3 * ldp], q[3 * ldq], acc);
  }
}

// accumulator tile -> the 4 fragments it represents (see header comment)
GRL_DEVINL void acc_to_frag(const f32x16& a, float4& f0, float4& f1, float4& f2, float4& f3) {
  f0 = make_float4(a[0], a[1], a[2], a[3]);
  f1 = make_float4(a[4], a[5], a[6], a[7]);
  f2 = make_float4(a[8], a[9], a[10], a[11]);
  f3 = make_float4(a[12], a[13], a[14], a[15]);
}

// erf-GELU (torch.nn.GELU() default; reference hepi.py:73, conv.py:67) and its derivative, branch-free:
//   erf(z) = sign(z) (1 - (a1 t + .. + a5 t^5) exp(-z^2)),  t = 1/(1 + p |z|)     (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7)
// with z = x/sqrt(2), so exp(-z^2) = exp(-x^2/2) is also the Gaussian pdf factor the derivative needs: one v_exp_f32 and
// one v_rcp_f32 per element give both gelu(x) and gelu'(x).  (libm erff costs ~3x the instructions and diverges.)
GRL_DEVINL void gelu_both(float x, float& g, float& gp) {
  const float az = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  const float e = __expf(-0.5f * x * x);
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float erf_abs = fmaf(-poly, e, 1.0f);
  const float cdf = fmaf(0.5f, copysignf(erf_abs, x), 0.5f);
  g = x * cdf;
  gp = fmaf(x * e, 0.39894228040143267794f, cdf);
}
GRL_DEVINL float gelu_f(float x) {
  float g, gp;
  gelu_both(x, g, gp);
  return g;
}
GRL_DEVINL float gelu_grad_f(float x) {
  float g, gp;
  gelu_both(x, g, gp);
  return gp;
}

GRL_DEVINL float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
GRL_DEVINL float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
GRL_DEVINL float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// copy a row-major [rows][K] fp32 matrix from global into an LDS image with leading dim ld (pads untouched)
GRL_DEVINL void stage_matrix(float* dst, const float* __restrict__ src, int rows, int K, int ld) {
  for (int idx = threadIdx.x; idx < rows * K; idx += blockDim.x) {
    const int r = idx / K, c = idx - r * K;
    dst[r * ld + c] = src[idx];
  }
}

#define GRL_CHECK_LAUNCH()                       \
  do {                                           \
    hipError_t e_ = hipGetLastError();           \
    if (e_ != hipSuccess) return -1000 - (int)e_; \
  } while (0)
