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

// ---- diagnostic switches that produce WRONG RESULTS (timing knock-outs: a part of a kernel removed to see what it costs) ------------
// They compile only together with -DGRL_DIAG, and a GRL_DIAG object exports `grl_diag_build`: geometry_rl_amd/hip.py refuses to build
// such flags into libgrl_hip.so and refuses to load a library that exports the symbol as the product library (tools/build_variants.sh
// builds them under _variants/, bench.py loads those only with GRL_ALLOW_DIAG_LIB=1).  One mis-set flag can no longer ship a wrong kernel.
#if defined(GRL_KNOCK_MFMA) || defined(GRL_KNOCK_STAGE) || defined(GRL_E16_NOLDS) || defined(GRL_E16_NOGELU) || defined(GRL_E16_NOGATHER) || \
    defined(GRL_E16_LDSONLY) || defined(GRL_E16_NOMFMA) || defined(GRL_B16_NOROWMMA) || defined(GRL_B16_NOGELU) || defined(GRL_B16_NOGATHER) || \
    defined(GRL_MLPB_NOMFMA) || defined(GRL_MLPB_NOGELU) || defined(GRL_MLPB_NOBARRIER) || defined(GRL_M16_KNOCK) || \
    (defined(GRL_FENCED_2W) && !(GRL_FENCED_2W))   /* only a command-line definition is visible here */
#ifndef GRL_DIAG
#error "timing knock-out switches produce wrong results: they need -DGRL_DIAG (and can then not be linked into the product library)"
#endif
#endif
#ifdef GRL_DIAG
extern "C" __attribute__((weak, visibility("default"))) int grl_diag_build() { return 1; }
#endif

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

// erf-GELU (torch.nn.GELU() default; reference hepi.py:73, conv.py:67) and its derivative, branch-free, two elements at a time so
// that the multiplies / FMAs become packed VALU instructions (v_pk_mul_f32 / v_pk_fma_f32):
//   erfc(|z|) = (a1 t + .. + a5 t^5) exp(-z^2),  t = 1/(1 + p |z|)        (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7)
// with z = x/sqrt(2): exp(-z^2) = exp(-x^2/2) is also the Gaussian pdf factor the derivative needs, so one v_exp_f32 and one
// v_rcp_f32 per element give both gelu(x) and gelu'(x).  (libm erff costs ~3x the instructions and diverges.)
//   gelu(x)  = x/2 + |x|/2 (1 - q),  q = erfc(|z|)          gelu'(x) = 1/2 + sign(x)/2 (1 - q) + x pdf(x)
#ifndef GRL_PREC
#define GRL_PREC 0   // (documented below, at the split-bf16 section: 1 = the plain-bf16 build of BASELINE config 5)
#endif
// GRL_PREC = 1 only: GELU through the logistic approximation of the normal CDF, Phi(x) ~ sigma(1.5976 x + 0.07056 x^3) (|error| <= 1.4e-4,
// far below the 4e-3 of a bf16 operand; the build's tolerance is 2e-2): value 5 plain + 2 transcendental
// instructions per element instead of 12 + 2, value + derivative 10 + 2 instead of 14 + 2 -- GELU is two thirds of the vector instructions of
// the bf16 kernels (profiles/r03_pmc_table_rope_hepi_bf16.txt).  The derivative is the exact derivative of the approximant.
#ifndef GRL_GELU_LOGISTIC
#define GRL_GELU_LOGISTIC GRL_PREC
#endif
GRL_DEVINL float gelu_logistic(float x) {
  const float w = x * fmaf(x * x, -0.07056f * 1.44269504088896f, -1.5976f * 1.44269504088896f);   // -(u log2 e)
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(w));
}
GRL_DEVINL void gelu_logistic_both(float x, float& g, float& gp) {
  const float x2 = x * x;
  const float w = x * fmaf(x2, -0.07056f * 1.44269504088896f, -1.5976f * 1.44269504088896f);
  const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(w));
  g = x * s;
  gp = fmaf(fmaf(-s, s, s), x * fmaf(x2, 3.f * 0.07056f, 1.5976f), s);
}
typedef float v2f __attribute__((ext_vector_type(2)));
GRL_DEVINL v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
GRL_DEVINL v2f splat2(float a) { return v2f{a, a}; }
// GRL_GELU_V2 (round 3, default): the same A&S 7.1.26 evaluation with fewer issue slots -- what the MFMA kernels are short of
// (DESIGN.md finding 22).  Per PAIR of elements, value + derivative: 4 transcendental + 4 plain + 12 packed instructions (was 4 + 6 + 15):
//   e2 = exp2(-x^2/2 log2 e + log2(1/sqrt(2 pi))) = pdf(x)            (the 1/sqrt(2 pi) rides in the exponent: an FMA instead of a multiply)
//   hq = (b1 t + .. + b5 t^5) e2 = erfc(|x|/sqrt 2)/2 = Phi(-|x|)      (b_i = a_i sqrt(2 pi)/2)
//   cdf = 1/2 + copysign(1/2 - hq, x);   gelu = x cdf;   gelu' = cdf + x e2
// value only: gelu = max(x, 0) - |x| hq with hq = (a1/2 t + ..) exp(-x^2/2): 4 transcendental + 6 plain (|x| and the sign are operand
// modifiers of v_fma_f32) + 8 packed (was 4 + 4 + 11).  Max |error| against erf-GELU in fp32: 4.7e-7 / 3.3e-7 (old form: 3.3e-7), derivative 3.0e-7.
// A transcendental-free odd polynomial x P(x^2) for Phi - 1/2 was priced and dropped: 13 coefficients for 6e-7 on |x| <= 5 (26 packed-FMA
// issue slots per pair against 17 for the rcp + exp it replaces) and a degree-25 Horner form cancels catastrophically in fp32.
#ifndef GRL_GELU_V2
#define GRL_GELU_V2 1
#endif
// Packed form of the logistic GELU (round 5, plain-bf16 build): the same operations as gelu_logistic / gelu_logistic_both on TWO elements per
// instruction.  A lone wave issues a v_pk_mul_f32 / v_pk_fma_f32 in ~5.2 cycles (profiles/r02_valu_rates.txt) -- the cost of ONE plain
// instruction -- so the ten plain operations per element become ten per PAIR (66 -> 43 cycles per element with the two transcendentals).
// Packed f32 does not overlap with an MFMA of the same wave (+20 cycles per instruction inside a dependent-MFMA gap,
// profiles/r05_valu16_rates.txt): in the bf16 build that does not matter, its layers issue their few MFMAs as one burst and the
// epilogues behind it; the fp32 build (MFMAs between the epilogue's instructions) keeps the scalar forms (DESIGN.md finding 23).
GRL_DEVINL void gelu_logistic_pair(v2f x, v2f& g) {
  const v2f w = x * fma2(x * x, splat2(-0.07056f * 1.44269504088896f), splat2(-1.5976f * 1.44269504088896f));
  v2f a;
  a.x = __builtin_amdgcn_exp2f(w.x);
  a.y = __builtin_amdgcn_exp2f(w.y);
  a = a + splat2(1.f);
  v2f s_;
  s_.x = __builtin_amdgcn_rcpf(a.x);
  s_.y = __builtin_amdgcn_rcpf(a.y);
  g = x * s_;
}
GRL_DEVINL void gelu_logistic_both_pair(v2f x, v2f& g, v2f& gp) {
  const v2f x2 = x * x;
  const v2f w = x * fma2(x2, splat2(-0.07056f * 1.44269504088896f), splat2(-1.5976f * 1.44269504088896f));
  v2f a;
  a.x = __builtin_amdgcn_exp2f(w.x);
  a.y = __builtin_amdgcn_exp2f(w.y);
  a = a + splat2(1.f);
  v2f s_;
  s_.x = __builtin_amdgcn_rcpf(a.x);
  s_.y = __builtin_amdgcn_rcpf(a.y);
  g = x * s_;
  gp = fma2(fma2(-s_, s_, s_), x * fma2(x2, splat2(3.f * 0.07056f), splat2(1.5976f)), s_);   // s (1 - s) as s - s^2: one packed instruction less
}
template <bool WITH_GRAD> GRL_DEVINL void gelu_pair_as(v2f x, v2f& g, v2f& gp);
template <bool WITH_GRAD>
GRL_DEVINL void gelu_pair(v2f x, v2f& g, v2f& gp) {
#if GRL_GELU_LOGISTIC
  if (WITH_GRAD) gelu_logistic_both_pair(x, g, gp);
  else gelu_logistic_pair(x, g);
  return;
#endif
  gelu_pair_as<WITH_GRAD>(x, g, gp);
}
// the A&S 7.1.26 form in EVERY build (gelu_pair's form where GRL_GELU_LOGISTIC is off): parameter-only code such as the fiber basis must
// not change its numbers with the precision of the build it happens to be compiled into (gelu_exact_f / gelu_exact_grad_f below)
template <bool WITH_GRAD>
GRL_DEVINL void gelu_pair_as(v2f x, v2f& g, v2f& gp) {
  const float kp = 0.3275911f * 0.70710678118654752440f;
  v2f t, e;
  t.x = __builtin_amdgcn_rcpf(fmaf(fabsf(x.x), kp, 1.0f));
  t.y = __builtin_amdgcn_rcpf(fmaf(fabsf(x.y), kp, 1.0f));
#if GRL_GELU_V2
  if (WITH_GRAD) {
    const v2f arg = fma2(x * x, splat2(-0.72134752044448170368f), splat2(-1.32574806473615910f));   // log2 pdf(x)
    e.x = __builtin_amdgcn_exp2f(arg.x);
    e.y = __builtin_amdgcn_exp2f(arg.y);
    v2f poly = fma2(t, splat2(1.33027442959f), splat2(-1.82125597911f));     // a_i * sqrt(2 pi) / 2
    poly = fma2(poly, t, splat2(1.78147793657f));
    poly = fma2(poly, t, splat2(-0.35656378125f));
    poly = fma2(poly, t, splat2(0.31938153026f));
    const v2f hq = (poly * t) * e;                                         // Phi(-|x|)
    const v2f w = splat2(0.5f) - hq;
    v2f ws;
    ws.x = copysignf(w.x, x.x);
    ws.y = copysignf(w.y, x.y);
    const v2f cdf = ws + splat2(0.5f);
    g = x * cdf;
    gp = fma2(x, e, cdf);
  } else {
    const v2f arg = (x * x) * splat2(-0.72134752044448170368f);
    e.x = __builtin_amdgcn_exp2f(arg.x);
    e.y = __builtin_amdgcn_exp2f(arg.y);
    v2f poly = fma2(t, splat2(0.5307027145f), splat2(-0.7265760135f));     // a_i / 2
    poly = fma2(poly, t, splat2(0.7107068705f));
    poly = fma2(poly, t, splat2(-0.142248368f));
    poly = fma2(poly, t, splat2(0.127414796f));
    const v2f hq = (poly * t) * e;
    g.x = fmaf(-fabsf(x.x), hq.x, fmaxf(x.x, 0.f));
    g.y = fmaf(-fabsf(x.y), hq.y, fmaxf(x.y, 0.f));
  }
#else
  const v2f arg = (x * x) * splat2(-0.72134752044448170368f);   // -x^2/2 * log2(e)
  e.x = __builtin_amdgcn_exp2f(arg.x);
  e.y = __builtin_amdgcn_exp2f(arg.y);
  v2f poly = fma2(t, splat2(1.061405429f), splat2(-1.453152027f));
  poly = fma2(poly, t, splat2(1.421413741f));
  poly = fma2(poly, t, splat2(-0.284496736f));
  poly = fma2(poly, t, splat2(0.254829592f));
  const v2f q = (poly * t) * e;
  const v2f hx = x * splat2(0.5f);
  v2f s;
  s.x = fabsf(hx.x);
  s.y = fabsf(hx.y);
  g = fma2(-s, q, s + hx);
  if (WITH_GRAD) {
    v2f cs;
    cs.x = copysignf(0.5f, x.x);
    cs.y = copysignf(0.5f, x.y);
    const v2f cdf = fma2(-cs, q, cs + splat2(0.5f));
    gp = fma2(x * e, splat2(0.39894228040143267794f), cdf);
  }
#endif
}
// value only, scalar form (plain f32 instructions; 12 + 2 transcendental per element): for forward kernels built without packed math
GRL_DEVINL float gelu_val(float x) {
#if GRL_GELU_LOGISTIC
  return gelu_logistic(x);
#endif
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
  float poly = fmaf(t, 0.5307027145f, -0.7265760135f);     // a_i / 2
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  return fmaf(-fabsf(x), (poly * t) * e, fmaxf(x, 0.f));
}
#ifndef GRL_GELU4_SCALAR
#define GRL_GELU4_SCALAR 0
#endif
GRL_DEVINL float4 gelu4(float4 x) {
#if GRL_GELU4_SCALAR
  return make_float4(gelu_val(x.x), gelu_val(x.y), gelu_val(x.z), gelu_val(x.w));
#endif
  v2f g0, g1, d0, d1;
  gelu_pair<false>(v2f{x.x, x.y}, g0, d0);
  gelu_pair<false>(v2f{x.z, x.w}, g1, d1);
  return make_float4(g0.x, g0.y, g1.x, g1.y);
}
// value + derivative, scalar form: plain (unpacked) vector instructions only -- the form for code that runs beside MFMAs of the same
// wave (packed f32 operations do not overlap with the matrix pipe: MI355X_MICROARCH.md cycle constants; DESIGN.md finding 23)
GRL_DEVINL void gelu_both_as(float x, float& g, float& gp);
GRL_DEVINL void gelu_both(float x, float& g, float& gp) {
#if GRL_GELU_LOGISTIC
  gelu_logistic_both(x, g, gp);
  return;
#endif
  gelu_both_as(x, g, gp);
}
GRL_DEVINL void gelu_both_as(float x, float& g, float& gp) {   // (the A&S form in every build: see gelu_pair_as)
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752440f, 1.0f));
#if GRL_GELU_V2
  const float e = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170368f, -1.32574806473615910f));   // pdf(x)
  float poly = fmaf(t, 1.33027442959f, -1.82125597911f);
  poly = fmaf(poly, t, 1.78147793657f);
  poly = fmaf(poly, t, -0.35656378125f);
  poly = fmaf(poly, t, 0.31938153026f);
  const float hq = (poly * t) * e;                       // Phi(-|x|)
  const float cdf = copysignf(0.5f - hq, x) + 0.5f;
  g = x * cdf;
  gp = fmaf(x, e, cdf);
#else
  const float e = __builtin_amdgcn_exp2f((x * -0.72134752044448170368f) * x);   // exp(-x^2/2)
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float q = (poly * t) * e;                       // erfc(|x| / sqrt 2)
  const float cs = copysignf(0.5f, x);
  const float cdf = fmaf(-cs, q, cs + 0.5f);            // 1/2 + sign(x)/2 (1 - q)
  g = x * cdf;
  gp = fmaf(x * e, 0.39894228040143267794f, cdf);
#endif
}
GRL_DEVINL void gelu_both4(float4 x, float4& g, float4& gp) {
  gelu_both(x.x, g.x, gp.x);
  gelu_both(x.y, g.y, gp.y);
  gelu_both(x.z, g.z, gp.z);
  gelu_both(x.w, g.w, gp.w);
}
GRL_DEVINL float4 gelu_grad4(float4 x) {
  float4 g, gp;
  gelu_both4(x, g, gp);
  return gp;
}
GRL_DEVINL float gelu_f(float x) {   // scalar forms (tails, tests)
  v2f g, gp;
  gelu_pair<false>(v2f{x, x}, g, gp);
  return g.x;
}
GRL_DEVINL float gelu_grad_f(float x) {
  float g, gp;
  gelu_both(x, g, gp);
  return gp;
}
// ... and the same two with the A&S form whatever GRL_PREC says: bitwise gelu_f / gelu_grad_f of the fp32 build
GRL_DEVINL float gelu_exact_f(float x) {
  v2f g, gp;
  gelu_pair_as<false>(v2f{x, x}, g, gp);
  return g.x;
}
GRL_DEVINL float gelu_exact_grad_f(float x) {
  float g, gp;
  gelu_both_as(x, g, gp);
  return gp;
}

// streaming (non-temporal) 16-byte accesses: global_load/store_dwordx4 ... nt -- for data touched once per launch
typedef float f32x4n __attribute__((ext_vector_type(4)));
GRL_DEVINL float4 load_nt4(const float* p) {
  const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
GRL_DEVINL void store_nt4(float* p, const float4& v) {
  __builtin_nontemporal_store(f32x4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4n*>(p));
}
#if (GRL_PREC || (defined(GRL_B16_BURST) && GRL_B16_BURST)) && defined(GRL_PK_F4)   // plain-bf16 build of a file that asks for it: element-wise products and sums as packed pairs (see gelu_logistic_pair)
GRL_DEVINL float4 f4_mul(float4 a, float4 b) {
  const v2f lo = v2f{a.x, a.y} * v2f{b.x, b.y}, hi = v2f{a.z, a.w} * v2f{b.z, b.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
GRL_DEVINL float4 f4_add(float4 a, float4 b) {
  const v2f lo = v2f{a.x, a.y} + v2f{b.x, b.y}, hi = v2f{a.z, a.w} + v2f{b.z, b.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
#else
GRL_DEVINL float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
GRL_DEVINL float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
#endif
GRL_DEVINL float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// copy a row-major [rows][K] fp32 matrix from global into an LDS image with leading dim ld (pads untouched)
GRL_DEVINL void stage_matrix(float* dst, const float* __restrict__ src, int rows, int K, int ld) {
  for (int idx = threadIdx.x; idx < rows * K; idx += blockDim.x) {
    const int r = idx / K, c = idx - r * K;
    dst[r * ld + c] = src[idx];
  }
}

// Optional per-kernel HIP-event timing for entry points that launch more than one kernel (grl_prof_* in train_ops.hip).
void grl_prof_begin(const char* name, hipStream_t stream);
void grl_prof_end(hipStream_t stream);
void grl_prof_begin_replay(const char* name, hipStream_t stream);   // (active in the stamp mode only: single-kernel entry points)
void grl_prof_end_replay(hipStream_t stream);

// One-time, thread-safe initialisation of per-process kernel attributes (max dynamic LDS): replaces the former `static bool`
// latches.  One process drives one device, so "once per process" is "once per device".
#include <mutex>
#define GRL_ONCE(...)                               \
  do {                                              \
    static std::once_flag grl_once_flag_;           \
    std::call_once(grl_once_flag_, [&] { __VA_ARGS__; }); \
  } while (0)

#define GRL_CHECK_LAUNCH()                       \
  do {                                           \
    hipError_t e_ = hipGetLastError();           \
    if (e_ != hipSuccess) return -1000 - (int)e_; \
  } while (0)

// ------------------------------------------------------------------------------------------------ split-bf16 MFMA path
// fp32 GEMMs on the bf16 matrix pipe: x = hi + lo with hi = upper 16 bits of x (truncation, exact), lo = bf16(x - hi);
// a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo  (three v_mfma_f32_32x32x16_bf16 at 16x the fp32-MFMA rate; dropped term and the
// rounding of lo are ~2^-17 relative -- measured <= 4e-5 absolute on O(5) activations, tests keep the 1e-4 bar).
// Operand maps of v_mfma_f32_32x32x16_bf16: lane (i = l&31, h = l>>5) supplies A[i][8h+j], B[8h+j][i], j = 0..7; C/D as the
// fp32 32x32 form.  To keep the "accumulator is the next operand" chain, element j of lane half h of K-step s stands for
// k = 16s + 8(j>>2) + 4h + (j&3): exactly fragments x[2s] (j<4) and x[2s+1] (j>=4) of the fp32 convention above.  Weight rows
// are staged in LDS as bf16 with the two middle quads of every 16-block swapped, so a lane reads its 8 weights with one
// ds_read_b128.
// GRL_PREC (per translation unit): 0 = split-bf16 products (three MFMAs per product, fp32-accurate: the default library),
// 1 = plain bf16 products (ONE MFMA per product, operands rounded to nearest bf16, fp32 accumulation) -- the reduced-precision
// variant of BASELINE config 5 (rope_shaping_hepi_trpl, bf16): the same kernels compiled a second time with -DGRL_PREC=1 and the
// entry points suffixed _bf16.  In that build the "lo" halves below are never formed and every lo product is dropped.
#ifndef GRL_PREC
#define GRL_PREC 0
#endif
#if GRL_PREC
#define GRL_LO(...)
#define GRL_ENTRY(name) name##_bf16
#else
#define GRL_LO(...) __VA_ARGS__
#define GRL_ENTRY(name) name
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

GRL_DEVINL f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef GRL_KNOCK_MFMA   // timing knock-out (diagnostic builds of one file; results are wrong)
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

GRL_DEVINL unsigned pack_hi(float a, float b) {  // upper halves of a (low word) and b (high word)
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
GRL_DEVINL float trunc_bf16(float a) { return __uint_as_float(__float_as_uint(a) & 0xFFFF0000u); }
GRL_DEVINL unsigned pack_rn(float a, float b) {  // round-to-nearest bf16 of a (low) and b (high)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// fragments x[2s], x[2s+1] (fp32) -> hi / lo bf16 operands of K-step s
GRL_DEVINL void split_pair(const float4& f0, const float4& f1, bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
#if GRL_PREC
  h[0] = pack_rn(f0.x, f0.y); h[1] = pack_rn(f0.z, f0.w); h[2] = pack_rn(f1.x, f1.y); h[3] = pack_rn(f1.z, f1.w);
  l[0] = l[1] = l[2] = l[3] = 0u;   // never used: every lo product is compiled out
#else
  h[0] = pack_hi(f0.x, f0.y); h[1] = pack_hi(f0.z, f0.w); h[2] = pack_hi(f1.x, f1.y); h[3] = pack_hi(f1.z, f1.w);
  l[0] = pack_rn(f0.x - trunc_bf16(f0.x), f0.y - trunc_bf16(f0.y));
  l[1] = pack_rn(f0.z - trunc_bf16(f0.z), f0.w - trunc_bf16(f0.w));
  l[2] = pack_rn(f1.x - trunc_bf16(f1.x), f1.y - trunc_bf16(f1.y));
  l[3] = pack_rn(f1.z - trunc_bf16(f1.z), f1.w - trunc_bf16(f1.w));
#endif
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
template <int K>
GRL_DEVINL void split_frags(const float4 (&x)[K / 8], bf16x8 (&hi)[K / 16], bf16x8 (&lo)[K / 16]) {
#pragma unroll
  for (int s = 0; s < K / 16; ++s) split_pair(x[2 * s], x[2 * s + 1], hi[s], lo[s]);
}

// ---- storage type of the node latents ([N,16,64] tensors x, x1, x2 and their gradients) -----------------------------------------
// GRL_PREC = 0: float.  GRL_PREC = 1 (BASELINE config 5, "bf16 storage / MFMA, fp32 accumulate"): bf16 in HBM, widened to fp32 in
// registers; every accumulation stays fp32 and the rounding to nearest bf16 happens once, at the store.  Weights, positions, weight-
// gradient partial slabs and everything at the loss stay fp32.
#if GRL_PREC
typedef unsigned short st_t;   // bf16 bits
GRL_DEVINL float ld1(const st_t* p) { return __uint_as_float((unsigned)(*p) << 16); }
GRL_DEVINL void st1(st_t* p, float v) { *p = (unsigned short)(pack_rn(v, 0.f) & 0xFFFFu); }
GRL_DEVINL float4 ld4(const st_t* p) {   // 4 consecutive elements, 8-byte aligned
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
GRL_DEVINL void st4(st_t* p, const float4& v) {
  uint2 u;
  u.x = pack_rn(v.x, v.y);
  u.y = pack_rn(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}
GRL_DEVINL float4 ld4_nt(const st_t* p) { return ld4(p); }
GRL_DEVINL void st4_nt(st_t* p, const float4& v) { st4(p, v); }
// A prefetched quad stays RAW until it is used: widened where it is loaded, the shift is the load's first use and the compiler puts the
// s_waitcnt of every prefetch directly behind it (round 5: lift_encode_bwd's four loads per node ran one after the other, 270 us where
// the bytes need 60).
typedef uint2 raw4_t;
GRL_DEVINL raw4_t ld4_raw(const st_t* p) { return *reinterpret_cast<const uint2*>(p); }
GRL_DEVINL float4 widen4(const raw4_t& u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
#else
typedef float st_t;
GRL_DEVINL float ld1(const st_t* p) { return *p; }
GRL_DEVINL void st1(st_t* p, float v) { *p = v; }
GRL_DEVINL float4 ld4(const st_t* p) { return *reinterpret_cast<const float4*>(p); }
GRL_DEVINL void st4(st_t* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
GRL_DEVINL float4 ld4_nt(const st_t* p) { return load_nt4(p); }
GRL_DEVINL void st4_nt(st_t* p, const float4& v) { store_nt4(p, v); }
typedef float4 raw4_t;
GRL_DEVINL raw4_t ld4_raw(const st_t* p) { return load_nt4(p); }
GRL_DEVINL float4 widen4(const raw4_t& u) { return u; }
#endif

// LDS leading dimension (in bf16 elements) of a [rows][K] split-weight image: +8 elements (16 B) keeps ds_read_b128 conflict-free
#define GRL_LDB(K) ((K) + 8)

// stage W[ROWS][K] (fp32, global, row length KSRC <= K) as two bf16 images (hi, lo) with the per-16-block quad swap described
// above.  One quad (4 consecutive k = 4 consecutive image positions) per thread and iteration: a 16-byte global load, two 8-byte
// LDS stores; the loads of up to 8 iterations are issued back to back before the first store, so the prologue costs a couple of
// L2 round trips (the former scalar loop -- one dependent load per element -- was a fixed 25-35 us at the head of every launch).
GRL_DEVINL void put_split_quad(unsigned short* hi, unsigned short* lo, const float4& w) {
  uint2 h, l;
#if GRL_PREC
  h.x = pack_rn(w.x, w.y); h.y = pack_rn(w.z, w.w);
  *reinterpret_cast<uint2*>(hi) = h;
  (void)lo; (void)l;
#else
  h.x = pack_hi(w.x, w.y); h.y = pack_hi(w.z, w.w);
  l.x = pack_rn(w.x - trunc_bf16(w.x), w.y - trunc_bf16(w.y));
  l.y = pack_rn(w.z - trunc_bf16(w.z), w.w - trunc_bf16(w.w));
  *reinterpret_cast<uint2*>(hi) = h;
  *reinterpret_cast<uint2*>(lo) = l;
#endif
}
template <int ROWS, int K, int KSRC, int NT>
GRL_DEVINL void stage_split(unsigned short* hi, unsigned short* lo, const float* __restrict__ src, int ld) {
  constexpr int Q = K / 4, N = ROWS * Q, IT = (N + NT - 1) / NT, G = IT < 8 ? IT : 8;
  const bool vec = KSRC % 4 == 0 && (reinterpret_cast<size_t>(src) & 15) == 0;   // block-uniform
#pragma unroll 1
  for (int base = 0; base < IT; base += G) {
    float4 w[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int idx = threadIdx.x + (base + g) * NT;
      const int r = idx / Q, pq = idx - r * Q, q = pq & 3;
      const int k = ((4 * pq) & ~15) + ((q == 1) ? 8 : (q == 2) ? 4 : 4 * q);
      w[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < N) {
        const float* sp = src + (size_t)r * KSRC + k;
        if (vec) {
          w[g] = *reinterpret_cast<const float4*>(sp);
        } else {
          w[g].x = k < KSRC ? sp[0] : 0.f; w[g].y = k + 1 < KSRC ? sp[1] : 0.f;
          w[g].z = k + 2 < KSRC ? sp[2] : 0.f; w[g].w = k + 3 < KSRC ? sp[3] : 0.f;
        }
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int idx = threadIdx.x + (base + g) * NT;
      const int r = idx / Q, pq = idx - r * Q;
      if (idx < N) put_split_quad(hi + r * ld + 4 * pq, lo + r * ld + 4 * pq, w[g]);
    }
  }
}
// the same for the TRANSPOSE of src [64][64]: image row k holds src[.][k] (lanes run along k: coalesced 4-byte loads)
template <int NT>
GRL_DEVINL void stage_split_T(unsigned short* hi, unsigned short* lo, const float* __restrict__ src, int ld) {
  constexpr int N = 64 * 16, IT = (N + NT - 1) / NT;
  float4 w[IT];
#pragma unroll
  for (int g = 0; g < IT; ++g) {
    const int idx = threadIdx.x + g * NT, k = idx & 63, pq = idx >> 6, q = pq & 3;
    const int n = ((4 * pq) & ~15) + ((q == 1) ? 8 : (q == 2) ? 4 : 4 * q);
    w[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < N) w[g] = make_float4(src[n * 64 + k], src[(n + 1) * 64 + k], src[(n + 2) * 64 + k], src[(n + 3) * 64 + k]);
  }
#pragma unroll
  for (int g = 0; g < IT; ++g) {
    const int idx = threadIdx.x + g * NT, k = idx & 63, pq = idx >> 6;
    if (idx < N) put_split_quad(hi + k * ld + 4 * pq, lo + k * ld + 4 * pq, w[g]);
  }
}

// acc(32 n x 32 r) += W[n0+i][0..K) . X[r][0..K)   with split operands.
//   whi/wlo = &image[(n0 + (lane&31)) * ld + 8*(lane>>5)]
template <int K>
GRL_DEVINL void mma_wx_bf(const unsigned short* whi, const unsigned short* wlo, const bf16x8 (&xh)[K / 16], const bf16x8 (&xl)[K / 16],
                          f32x16& acc) {
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(whi + 16 * s);
    GRL_LO(const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wlo + 16 * s);)
    acc = mfma_bf(wh, xh[s], acc);
    GRL_LO(acc = mfma_bf(wl, xh[s], acc);)
    GRL_LO(acc = mfma_bf(wh, xl[s], acc);)
  }
}

// Fenced form for kernels that run two waves per SIMD.  Measured on gfx950 (tools/det_check_all.py, tools/variant_check.sh):
// when LDS operand loads are interleaved with the MFMAs of a dependent chain (the form above) and a second wave shares the SIMD,
// a few 1e-4 of the tiles come out wrong, different ones on every run -- a later load lands in a register that an earlier,
// still queued MFMA has not read yet (the register allocator reuses dead operand registers; with one wave per SIMD the queue
// never gets deep enough).  The cure is structural: load EVERY operand fragment of the group first, then issue the MFMAs, then
// run an epilogue that reads the accumulator (it cannot start before the group has finished), and only then let the next loads go.
// GRL_MFMA_PRIO (build switch, off): the wave raises its issue priority for the duration of an MFMA group, so that a SIMD
// partner's VALU stream cannot delay the issue of this wave's dependent MFMAs (MI355X_MICROARCH.md "Two waves per SIMD", items 2
// and 4).  Measured round 2 (tools/run_variants.sh, one box, two alternating rounds): 237-245 steps/s against 239-240 without --
// inside the box-internal noise for every kernel; left off.
#ifndef GRL_MFMA_PRIO
#define GRL_MFMA_PRIO 0
#endif
#if GRL_MFMA_PRIO
#define GRL_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define GRL_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define GRL_PRIO_HI()
#define GRL_PRIO_LO()
#endif
template <int K>
struct WFrags { bf16x8 h[K / 16], l[K / 16]; };
template <int K>
GRL_DEVINL void load_wfrags(WFrags<K>& w, const unsigned short* whi, const unsigned short* wlo) {
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    w.h[s] = *reinterpret_cast<const bf16x8*>(whi + 16 * s);
    GRL_LO(w.l[s] = *reinterpret_cast<const bf16x8*>(wlo + 16 * s);)
  }
}
// a real VALU read of the accumulator: everything after it is ordered behind the completion of the MFMA group that produced acc
GRL_DEVINL void acc_fence(const f32x16& acc, float& sink) {
  sink += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, acc[0]), 0xE4, 0xF, 0xF, false));
}
template <int K, class Epi>
GRL_DEVINL void mma_wx_bf_fenced(const unsigned short* whi, const unsigned short* wlo, const bf16x8 (&xh)[K / 16],
                                 const bf16x8 (&xl)[K / 16], f32x16 acc, Epi&& epilogue) {
  WFrags<K> w;
  load_wfrags<K>(w, whi, wlo);
  __builtin_amdgcn_sched_barrier(0);
  GRL_PRIO_HI();
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    acc = mfma_bf(w.h[s], xh[s], acc);
    GRL_LO(acc = mfma_bf(w.l[s], xh[s], acc);)
    GRL_LO(acc = mfma_bf(w.h[s], xl[s], acc);)
  }
  GRL_PRIO_LO();
  epilogue(acc);
  __builtin_amdgcn_sched_barrier(0);
}
// Pipelined form of the fenced group: the weight fragments of THIS group were loaded by the previous call (``cur``); the fragments
// of the NEXT group (``nhi`` / ``nlo``, K_NEXT deep) are requested right after a real read of the finished accumulator -- no MFMA
// is in flight any more (the hazard the fence exists for) -- and land while the epilogue (activation, splits) runs, instead of
// exposing their LDS latency in front of the next group.
template <int K, int K_NEXT, class Epi>
GRL_DEVINL void mma_wx_bf_piped(const WFrags<K>& cur, const bf16x8 (&xh)[K / 16], const bf16x8 (&xl)[K / 16], f32x16 acc,
                                WFrags<K_NEXT>* next, const unsigned short* nhi, const unsigned short* nlo, float& sink,
                                Epi&& epilogue) {
  __builtin_amdgcn_sched_barrier(0);
  GRL_PRIO_HI();
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    acc = mfma_bf(cur.h[s], xh[s], acc);
    GRL_LO(acc = mfma_bf(cur.l[s], xh[s], acc);)
    GRL_LO(acc = mfma_bf(cur.h[s], xl[s], acc);)
  }
  GRL_PRIO_LO();
  acc_fence(acc, sink);
  __builtin_amdgcn_sched_barrier(0);
  if (next) load_wfrags<K_NEXT>(*next, nhi, nlo);
  epilogue(acc);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- register-level transposes on the matrix pipe -------------------------------------------------------------------------
// X (32 rows x 32 columns) given as split-bf16 A operands (lane = row) times a 0/1 selection matrix gives X back in the
// ACCUMULATOR layout, i.e. with the column on the lane and the rows in the registers (acc row order == bf16 k-order): exactly the
// operand layout a product that sums over X's rows needs.  Two MFMAs per 32x32 tile and per hi/lo part, no LDS, exact (x * 1.0).
//   sel0 / sel1: lane (j = l&31, h = l>>5), element jj = 1.0 iff 8*(jj>>2) + 4h + (jj&3) == j (for j < 16) resp. j - 16 (j >= 16)
GRL_DEVINL void make_selectors(bf16x8& sel0, bf16x8& sel1) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  u32x4 s0, s1;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned v0 = 0, v1 = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int jj = 2 * w + e;
      const int kappa = 8 * (jj >> 2) + 4 * h + (jj & 3);
      if (j < 16 && kappa == j) v0 |= 0x3F80u << (16 * e);
      if (j >= 16 && kappa == j - 16) v1 |= 0x3F80u << (16 * e);
    }
    s0[w] = v0;
    s1[w] = v1;
  }
  sel0 = __builtin_bit_cast(bf16x8, s0);
  sel1 = __builtin_bit_cast(bf16x8, s1);
}
// columns 0..15 come from operand c0, columns 16..31 from operand c1 (two consecutive K-step fragments of the row-major tile)
GRL_DEVINL f32x16 transpose32(const bf16x8& c0, const bf16x8& c1, const bf16x8& sel0, const bf16x8& sel1) {
  f32x16 t = zero16();
  t = mfma_bf(c0, sel0, t);
  t = mfma_bf(c1, sel1, t);
  return t;
}
// accumulator tile holding bf16-exact values -> the two K-step operand fragments (rows 0..15 / 16..31 of the reduction)
GRL_DEVINL void acc_to_bf(const f32x16& t, bf16x8& k0, bf16x8& k1) {
  u32x4 a, b;
  a[0] = pack_hi(t[0], t[1]); a[1] = pack_hi(t[2], t[3]); a[2] = pack_hi(t[4], t[5]); a[3] = pack_hi(t[6], t[7]);
  b[0] = pack_hi(t[8], t[9]); b[1] = pack_hi(t[10], t[11]); b[2] = pack_hi(t[12], t[13]); b[3] = pack_hi(t[14], t[15]);
  k0 = __builtin_bit_cast(bf16x8, a);
  k1 = __builtin_bit_cast(bf16x8, b);
}

// A 32-column tile of a row-major activation block (rows on the lanes), transposed and split: lane = column, K-steps 0/1 =
// rows 0..15 / 16..31 (accumulator row order), hi and lo bf16 parts.  Built from the tile's two 16-column fragment pairs.
struct TTile { bf16x8 h0, h1, l0, l1; };
GRL_DEVINL TTile transpose_split(const bf16x8& ch0, const bf16x8& ch1, const bf16x8& cl0, const bf16x8& cl1, const bf16x8& sel0,
                                 const bf16x8& sel1, float* colsum = nullptr) {
  TTile t;
  const f32x16 th = transpose32(ch0, ch1, sel0, sel1);
#if GRL_PREC
  if (colsum) {
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc += th[q];
    *colsum += sacc;
  }
  acc_to_bf(th, t.h0, t.h1);
  t.l0 = t.h0; t.l1 = t.h1;   // placeholders, never used
#else
  const f32x16 tl = transpose32(cl0, cl1, sel0, sel1);
  if (colsum) {  // sum over this lane's 16 rows of (hi + lo) = the column sum restricted to them; the other 16 rows: lane ^ 32
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc += th[q] + tl[q];
    *colsum += sacc;
  }
  acc_to_bf(th, t.h0, t.h1);
  acc_to_bf(tl, t.l0, t.l1);
#endif
  return t;
}
// acc[m][n] += sum over the 32 rows r of P[r][m] Q[r][n]  (both operands transposed-split; split-bf16, 6 MFMAs)
GRL_DEVINL void mma_tn_bf(const TTile& p, const TTile& q, f32x16& acc) {
  acc = mfma_bf(p.h0, q.h0, acc); GRL_LO(acc = mfma_bf(p.l0, q.h0, acc); acc = mfma_bf(p.h0, q.l0, acc);)
  acc = mfma_bf(p.h1, q.h1, acc); GRL_LO(acc = mfma_bf(p.l1, q.h1, acc); acc = mfma_bf(p.h1, q.l1, acc);)
}
// The same for accumulators that live for a whole launch (weight gradients).  hipcc (ROCm 7.2) splits the live ranges of such
// loop-carried 16-register tuples and pays for it with AGPR-to-AGPR copies in front of the MFMA groups and at the loop header
// (256 v_accvgpr_mov per pass of edge_conv_bwd_w: 13 % of its vector issue slots).  With the accumulator tied as a read-write AGPR
// operand of an asm MFMA there is nothing to copy at the instruction: it is updated in place.  Measured round 2: the register
// allocator then places the same number of copies elsewhere (288 instead of 256 v_accvgpr_mov per pass) and the launch time does
// not move (1.20 vs 1.19 ms per step): GRL_ASM_ACC stays 0 (builtin form); the switch is kept for compiler upgrades.
// The asm is opaque to the compiler's hazard recognizer: the s_nop covers a VALU-written operand, and whoever reads such an
// accumulator with VALU code later must let the last MFMA drain first (asm_acc_drain()).
#ifndef GRL_ASM_ACC
#define GRL_ASM_ACC 0
#endif
GRL_DEVINL void mfma_acc(const bf16x8& a, const bf16x8& b, f32x16& c) {
#if GRL_ASM_ACC
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
  c = mfma_bf(a, b, c);
#endif
}
GRL_DEVINL void asm_acc_drain() {
#if GRL_ASM_ACC
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // 32 wait states > the 18 an MFMA result needs before a VALU read
#endif
}
GRL_DEVINL void mma_tn_bf_acc(const TTile& p, const TTile& q, f32x16& acc) {
  mfma_acc(p.h0, q.h0, acc); GRL_LO(mfma_acc(p.l0, q.h0, acc); mfma_acc(p.h0, q.l0, acc);)
  mfma_acc(p.h1, q.h1, acc); GRL_LO(mfma_acc(p.l1, q.h1, acc); mfma_acc(p.h1, q.l1, acc);)
}
