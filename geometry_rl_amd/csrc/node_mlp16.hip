// Fused backward of the ConvNeXt node block (reference conv.py:64-69,112; ponita.py:219-230) on 16-row tiles, round 3.
//   forward:  out = x_dst + W4 GELU(W3 LN(x2) + b3) + b4         backward (ONE launch): dx2, dW3, db3, dW4, db4, dgamma, dbeta
//
// Why a second form of node_mlp_bwd_fused_kernel (node_mlp.hip): that kernel walks 32-row chunks with eight phase-locked waves and FOUR
// barriers per chunk; its profile (profiles/r02_pmc_table_v4.txt) shows the matrix pipe 18 % busy, the vector pipe 18 % busy and 42 % of
// the wave cycles parked at waits / barriers: per chunk a SIMD needs ~4 700 matrix-pipe cycles and ~3 900 vector issue cycles and takes
// ~15 900.  Every stage (LayerNorm, fragment images, products, dA reduction, LayerNorm backward) exposes its LDS / L2 round trips in
// turn, and both waves of a SIMD do the same kind of work at the same time.
//
// This kernel is built like the fused edge backward (edge_conv16.hip, DESIGN.md finding 20):
//   * one workgroup of FOUR waves per CU, one wave per SIMD, 512 registers: wave w owns hidden units [64 w, 64 w + 64) -- its W3 / W4^T /
//     W3^T operand fragments (192 registers) and its dW3 / dW4 slices (128 accumulator registers, pinned AGPRs) live in registers for
//     the whole launch; no weight image in LDS, no W3 slab in L2;
//   * a chunk is ONE node (its 16 orientation rows): chain products (z, dH, dA) on v_mfma_f32_16x16x32_bf16 with the rows on the N side,
//     weight-gradient products on v_mfma_f32_32x32x16_bf16 with K = the 16 rows, operands through transposed LDS reads (grl_tile16.h);
//   * a three-stage software pipeline with ONE barrier per chunk: iteration i runs stage 1 of chunk i + 1 (LayerNorm, split-bf16 operand
//     images), the products of chunk i and stage 4 of chunk i - 1 (sum of the four partial dA rows, LayerNorm backward, dx2 store); the
//     three are independent, so the vector work of stages 1 and 4 fills the issue slots beside the chunk's MFMA groups (the only overlap a
//     SIMD gives: DESIGN.md finding 18), and nothing waits for another wave inside an iteration.  Images and partial dA rows are
//     double-buffered.
// Partial slab per workgroup (unchanged): [dW3 256x64 | db3 256 | dW4 64x256 | db4 64 | dgamma 64 | dbeta 64].
#include "grl_tile16.h"
#include "grl_wimg.h"

namespace {

constexpr int C = 64, W = 256;
constexpr float LN_EPS = 1e-5f;
constexpr int MLP_PARTIAL = W * C + W + C * W + C + C + C;
constexpr int LDBI = 64 + 16;   // bf16 elements per row of a B-operand image: 160-byte rows, conflict-free ds_read_b128 (finding 20b)
constexpr int LDDA = 64 + 4;    // floats per row of a partial dA image

struct Img16 {
  unsigned short aBh[16 * LDBI], aBl[16 * LDBI];   // a = LN(x2): B-operand layout [row][channel]
  unsigned short dBh[16 * LDBI], dBl[16 * LDBI];   // dOut
  unsigned short aTh[STG], aTl[STG];               // the same rows in the transposed-read layout (stg_off)
  unsigned short dTh[STG], dTl[STG];
};
struct Priv16 {
  unsigned short zh[STG], zl[STG];   // dZ rows of this wave's 64 hidden units; then (same bytes) h = GELU(z): a wave's LDS operations
                                     // execute in order, so the transposed reads of dZ have their data before h overwrites it
};
struct Mlp16Smem {
  Img16 img[2];
  float DA[2][4][16 * LDDA];
  Priv16 priv[4];
  u32x4 W4F[4][4][2][2][64];         // [wave][n-tile][k-step][hi | lo][lane]: the W4^T operand fragments (the third static set does not
                                     // fit the register file next to 128 accumulator and 128 static operand registers: it spilled)
};

template <int CTRL>
GRL_DEVINL float dpp_read(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
GRL_DEVINL float row16_sum(float v) {   // sum over the 16 lanes of a DPP row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror)
  v += dpp_read<0xB1>(v);
  v += dpp_read<0x4E>(v);
  v += dpp_read<0x141>(v);
  v += dpp_read<0x140>(v);
  return v;
}
GRL_DEVINL void split4(const float4& v, uint2& hi, uint2& lo) {
#if GRL_PREC
  hi.x = pack_rn(v.x, v.y); hi.y = pack_rn(v.z, v.w);
  lo = make_uint2(0u, 0u);
#else
  hi.x = pack_hi(v.x, v.y); hi.y = pack_hi(v.z, v.w);
  lo.x = pack_rn(v.x - trunc_bf16(v.x), v.y - trunc_bf16(v.y));
  lo.y = pack_rn(v.z - trunc_bf16(v.z), v.w - trunc_bf16(v.w));
#endif
}
// GRL_M16_SCALAR (default): plain f32 vector instructions only in this kernel (scalar GELU, file compiled with -fno-slp-vectorize)
#ifndef GRL_M16_SCALAR
#define GRL_M16_SCALAR (!GRL_PREC)   // the plain-bf16 build takes the packed logistic GELU (grl_common.h gelu_logistic_both_pair): +1.3 % on the
#endif                               // rope workload's step, A/B profiles/r05_ab_mlp_pk.txt; the fp32 build stays scalar (finding 23)
GRL_DEVINL void gelu_both4_pk(const float4& x, float4& gv, float4& gpv) {
#if GRL_M16_SCALAR
  gelu_both4(x, gv, gpv);
  return;
#endif
  v2f g0, g1, d0, d1;
  gelu_pair<true>(v2f{x.x, x.y}, g0, d0);
  gelu_pair<true>(v2f{x.z, x.w}, g1, d1);
  gv = make_float4(g0.x, g0.y, g1.x, g1.y);
  gpv = make_float4(d0.x, d0.y, d1.x, d1.y);
}
// static operand fragments of one 64 x 64 weight slice for the 16x16x32 chain: [n-tile][k-step], hi / lo
struct WFrag16 {
  bf16x8 h[4][2], l[4][2];
};
#define BAR() __builtin_amdgcn_sched_barrier(0)
#ifdef GRL_M16_PHASE   // diagnostic build: s_memtime ticks per stage of an iteration, wave 0 of every workgroup (tools/mlp_bwd_bench.py)
__device__ unsigned long long g_m16phase[16];
#define M16_PH(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[i] += t_ - tl_; tl_ = t_; } while (0)
#else
#define M16_PH(i)
#endif

__global__ __launch_bounds__(256, 1) void node_mlp_bwd16_kernel(const st_t* __restrict__ x2, const st_t* __restrict__ dout,
                                                                 const float* __restrict__ W3, const float* __restrict__ b3,
                                                                 const float* __restrict__ W4, const float* __restrict__ gam,
                                                                 const float* __restrict__ bet, st_t* __restrict__ dx2,
                                                                 float* __restrict__ partial, st_t* __restrict__ dump, int n_chunks,
                                                                 const Mlp16Image* __restrict__ wimg) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  Mlp16Smem& sm = *reinterpret_cast<Mlp16Smem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int srow = tid >> 4, cq = tid & 15;   // stages 1 / 4: thread = (row of the chunk, channel quad)
  const float4 gq = *reinterpret_cast<const float4*>(gam + 4 * cq), bq = *reinterpret_cast<const float4*>(bet + 4 * cq);
  Priv16& pv = sm.priv[wave];

  // ---- this wave's static operand fragments (hidden units j0 .. j0 + 63)
  const int j0 = 64 * wave;
  WFrag16 w3f, w3t;
  float4 b3q[4];
#ifdef GRL_KNOCK_STAGE   // timing knock-out: constant fragments instead of the strided weight loads (results are wrong)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    b3q[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32x4 c = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
      w3f.h[nt][s] = w3f.l[nt][s] = w3t.h[nt][s] = w3t.l[nt][s] = __builtin_bit_cast(bf16x8, c);
      sm.W4F[wave][nt][s][0][lane] = c; sm.W4F[wave][nt][s][1][lane] = c;
    }
  }
#else
  if (wimg) {   // the fragments of this step's weights, pre-split once by grl_weight_images (kind 3): 48 coalesced 16-byte loads per lane
                // instead of 16 vector + 128 strided dword loads and 48 splits
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      b3q[nt] = *reinterpret_cast<const float4*>(b3 + j0 + 16 * nt + 4 * g);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w3f.h[nt][s] = __builtin_bit_cast(bf16x8, wimg->w3f[wave][nt][s][0][lane]);
        w3t.h[nt][s] = __builtin_bit_cast(bf16x8, wimg->w3t[wave][nt][s][0][lane]);
        sm.W4F[wave][nt][s][0][lane] = wimg->w4f[wave][nt][s][0][lane];
#if !GRL_PREC
        w3f.l[nt][s] = __builtin_bit_cast(bf16x8, wimg->w3f[wave][nt][s][1][lane]);
        w3t.l[nt][s] = __builtin_bit_cast(bf16x8, wimg->w3t[wave][nt][s][1][lane]);
        sm.W4F[wave][nt][s][1][lane] = wimg->w4f[wave][nt][s][1][lane];
#endif
      }
    }
  } else {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    b3q[nt] = *reinterpret_cast<const float4*>(b3 + j0 + 16 * nt + 4 * g);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      {   // z^T = W3 a^T: A[m = hidden 16 nt + r][k = channel 32 s + 8 g + j]
        const float* p = W3 + (size_t)(j0 + 16 * nt + r) * C + 32 * s + 8 * g;
        split_pair(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), w3f.h[nt][s], w3f.l[nt][s]);
      }
      {   // dH^T = W4^T dOut^T: A[m = hidden][k = channel c] = W4[c][hidden]
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W4[(size_t)(32 * s + 8 * g + j) * W + j0 + 16 * nt + r];
        bf16x8 fh, fl;
        split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), fh, fl);
        sm.W4F[wave][nt][s][0][lane] = __builtin_bit_cast(u32x4, fh);
        GRL_LO(sm.W4F[wave][nt][s][1][lane] = __builtin_bit_cast(u32x4, fl);)
      }
      {   // dA^T = W3^T dZ^T: A[m = channel 16 nt + r][k = hidden, chain order 32 s + 16 (j >> 2) + 4 g + (j & 3)]
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W3[(size_t)(j0 + 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)) * C + 16 * nt + r];
        split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), w3t.h[nt][s], w3t.l[nt][s]);
      }
    }
  }
  }
#endif
#ifndef GRL_M16_PIN_STATIC
#define GRL_M16_PIN_STATIC 1
#endif
#if GRL_M16_PIN_STATIC
  // the 32 static fragments are defined in the accumulator half of the register file (MFMA A operands may be AGPRs): without this the
  // allocator keeps them as AGPR spill slots of ordinary registers and reads them back with 128 v_accvgpr_read per chunk
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      asm volatile("" : "+a"(w3f.h[nt][s]));  asm volatile("" : "+a"(w3t.h[nt][s]));
      GRL_LO(asm volatile("" : "+a"(w3f.l[nt][s]));  asm volatile("" : "+a"(w3t.l[nt][s]));)
    }
#endif
  f32x16 aW3[2][2], aW4[2][2];   // dW3[hidden tile][channel tile], dW4[channel tile][hidden tile]
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) { aW3[a_][b_] = zero16(); aW4[a_][b_] = zero16(); }
  float4 db3[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) db3[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam, db4 = dgam;

  // chunks of this workgroup: blockIdx.x, + gridDim.x, ...
  const int n_mine = blockIdx.x < n_chunks ? (n_chunks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto chunk_of = [&](int i) { return blockIdx.x + i * gridDim.x; };
  // the quads in flight stay RAW storage (grl_common.h raw4_t) until stage 1 reads them: widened where `px = pxn` hands them over, the
  // bf16 build waited for the loads of chunk it + 2 at the END of iteration it -- 0.9 instead of 1.7 iterations behind their issue (round 5)
  raw4_t px = raw4_t{}, pd = px, pxn = px, pdn = px;
  float4 pxw = make_float4(0.f, 0.f, 0.f, 0.f), pdw = pxw;
  auto fetch = [&](int i, raw4_t& fx, raw4_t& fd) {   // this thread's quad of chunk i (clamped: the loads of a non-existent chunk are never used)
#if defined(GRL_M16_KNOCK) && (GRL_M16_KNOCK & 1)   // timing knock-out (GRL_DIAG builds only): no row loads behind the first two chunks
    if (i >= 2) return;
#endif
    const size_t gofs = ((size_t)chunk_of(i < n_mine ? i : n_mine - 1) * 16 + srow) * C + 4 * cq;
    fx = ld4_raw(x2 + gofs);
    fd = ld4_raw(dout + gofs);
  };
  // LayerNorm state of the chunks in flight: [0] = chunk i - 1 (stage 4 of this iteration), [1] = chunk i, [2] = chunk i + 1 (stage 1)
  float4 xh0 = pxw, xh1 = pxw, xh2 = pxw;
  float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f;

  // Stage 1 (LayerNorm + operand images of the NEXT chunk) and stage 4 (LayerNorm backward of the PREVIOUS chunk) are cut into pieces that
  // are placed by hand between the weight-gradient MFMAs (asm statements: the compiler moves nothing across them that touches memory,
  // and clumps the rest); state of the pieces:
  float4 s1_xc, s1_d, s4_da, s4_gg;
  float s1_sum, s1_sq, s4_s1, s4_s2;
  uint2 s1_ah, s1_al, s1_dh, s1_dl;
  auto s1_a = [&](const float4& x, const float4& d) {   // row mean
    s1_d = d;
    s1_sum = row16_sum((x.x + x.y) + (x.z + x.w)) * (1.f / C);
    s1_xc = make_float4(x.x - s1_sum, x.y - s1_sum, x.z - s1_sum, x.w - s1_sum);
  };
  auto s1_b = [&](float4& xh, float& rstd) {            // variance, normalised row, a = LN(x2)
    s1_sq = row16_sum((s1_xc.x * s1_xc.x + s1_xc.y * s1_xc.y) + (s1_xc.z * s1_xc.z + s1_xc.w * s1_xc.w));
    rstd = rsqrtf(s1_sq * (1.f / C) + LN_EPS);
    xh = f4_scale(s1_xc, rstd);
  };
  auto s1_c = [&](const float4& xh, float valid) {      // split-bf16 quads of a and dOut (valid = 0: a chunk past the end, its images are
    const float4 a = make_float4(xh.x * gq.x + bq.x, xh.y * gq.y + bq.y, xh.z * gq.z + bq.z, xh.w * gq.w + bq.w);   // never read)
    split4(a, s1_ah, s1_al);
    split4(s1_d, s1_dh, s1_dl);
    db4 = make_float4(fmaf(s1_d.x, valid, db4.x), fmaf(s1_d.y, valid, db4.y), fmaf(s1_d.z, valid, db4.z), fmaf(s1_d.w, valid, db4.w));
  };
  auto s1_d_ = [&](int buf) {                           // both images of both operands
    Img16& im = sm.img[buf];
    const int ob = srow * LDBI + 4 * cq, ot = stg_off(srow, cq);
    *reinterpret_cast<uint2*>(im.aBh + ob) = s1_ah;
    *reinterpret_cast<uint2*>(im.dBh + ob) = s1_dh;
    *reinterpret_cast<uint2*>(im.aTh + ot) = s1_ah;
    *reinterpret_cast<uint2*>(im.dTh + ot) = s1_dh;
#if !GRL_PREC
    *reinterpret_cast<uint2*>(im.aBl + ob) = s1_al;
    *reinterpret_cast<uint2*>(im.dBl + ob) = s1_dl;
    *reinterpret_cast<uint2*>(im.aTl + ot) = s1_al;
    *reinterpret_cast<uint2*>(im.dTl + ot) = s1_dl;
#endif
  };
  auto stage1 = [&](int buf, const float4& x, const float4& d, float4& xh, float& rstd) {
    s1_a(x, d); s1_b(xh, rstd); s1_c(xh, 1.f); s1_d_(buf);
  };
  float4 s4_p[4];
  auto s4_load = [&](int buf) {                         // the four partial dA rows of this thread's quad
    const float* dap = &sm.DA[buf][0][0] + srow * LDDA + 4 * cq;
#pragma unroll
    for (int w_ = 0; w_ < 4; ++w_) s4_p[w_] = *reinterpret_cast<const float4*>(dap + w_ * 16 * LDDA);
  };
  auto s4_a = [&](const float4& xh) {
    s4_da = f4_add(f4_add(s4_p[0], s4_p[1]), f4_add(s4_p[2], s4_p[3]));
    s4_gg = f4_mul(s4_da, gq);
    s4_s1 = (s4_gg.x + s4_gg.y) + (s4_gg.z + s4_gg.w);
    s4_s2 = (s4_gg.x * xh.x + s4_gg.y * xh.y) + (s4_gg.z * xh.z + s4_gg.w * xh.w);
  };
  auto s4_b = [&]() {
    s4_s1 = row16_sum(s4_s1) * (1.f / C);
    s4_s2 = row16_sum(s4_s2) * (1.f / C);
  };
  // (the store is unconditional -- a branch around it makes every later wait for a load a wait for vmcnt(0), i.e. for the store's whole
  // round trip: the "previous chunk" of a workgroup's first iteration goes to a 16-row dump area behind the partial rows)
  auto s4_c = [&](st_t* rows, const float4& xh, float rstd) {
    const float4 dx = make_float4(rstd * (s4_gg.x - s4_s1 - xh.x * s4_s2), rstd * (s4_gg.y - s4_s1 - xh.y * s4_s2),
                                  rstd * (s4_gg.z - s4_s1 - xh.z * s4_s2), rstd * (s4_gg.w - s4_s1 - xh.w * s4_s2));
    st4_nt(rows + (size_t)srow * C + 4 * cq, dx);
  };
  auto s4_d = [&](const float4& xh) {
    dgam = make_float4(fmaf(s4_da.x, xh.x, dgam.x), fmaf(s4_da.y, xh.y, dgam.y), fmaf(s4_da.z, xh.z, dgam.z), fmaf(s4_da.w, xh.w, dgam.w));
    dbet = f4_add(dbet, s4_da);
  };

  // ---- the same two stages as twelve micro-steps of ~6 vector instructions (k is a constant after unrolling)
  // (volatile asm statements keep their order: an empty one that re-defines a step's input keeps the step behind the MFMA in front of it,
  // one that re-defines its output keeps it in front of the next MFMA -- without them every step sinks below the twelve MFMAs)
#define PIN(x) asm volatile("" : "+v"(x))
  float4 m_t0, m_t1, m_dx;
  float m_sa, m_sb;
  auto s4_step = [&](int k, st_t* rows) {
    const float4& xh = xh0;
    switch (k) {
      case 0: PIN(s4_p[0].x); m_t0 = f4_add(s4_p[0], s4_p[1]); PIN(m_t0.x); PIN(m_t0.w); break;
      case 1: PIN(s4_p[2].x); m_t1 = f4_add(s4_p[2], s4_p[3]); PIN(m_t1.x); PIN(m_t1.w); break;
      case 2: PIN(m_t0.y); s4_da = f4_add(m_t0, m_t1); PIN(s4_da.x); PIN(s4_da.w); break;
      case 3: PIN(s4_da.y); s4_gg = f4_mul(s4_da, gq); PIN(s4_gg.x); PIN(s4_gg.w); break;
      case 4: PIN(s4_gg.y); m_sa = (s4_gg.x + s4_gg.y) + (s4_gg.z + s4_gg.w); m_sb = s4_gg.x * xh.x + s4_gg.y * xh.y; PIN(m_sa); PIN(m_sb); break;
      case 5: PIN(m_sb); m_sb = fmaf(s4_gg.z, xh.z, fmaf(s4_gg.w, xh.w, m_sb)); m_sa += dpp_read<0xB1>(m_sa); m_sb += dpp_read<0xB1>(m_sb); PIN(m_sa); PIN(m_sb); break;
      case 6: PIN(m_sa); m_sa += dpp_read<0x4E>(m_sa); m_sb += dpp_read<0x4E>(m_sb); m_sa += dpp_read<0x141>(m_sa); m_sb += dpp_read<0x141>(m_sb); PIN(m_sa); PIN(m_sb); break;
      case 7: PIN(m_sa); m_sa += dpp_read<0x140>(m_sa); m_sb += dpp_read<0x140>(m_sb); s4_s1 = m_sa * (1.f / C); s4_s2 = m_sb * (1.f / C); PIN(s4_s1); PIN(s4_s2); break;
      case 8: PIN(s4_s1); m_dx.x = rs0 * (s4_gg.x - s4_s1 - xh.x * s4_s2); m_dx.y = rs0 * (s4_gg.y - s4_s1 - xh.y * s4_s2); PIN(m_dx.x); PIN(m_dx.y); break;
      case 9: PIN(s4_s2); m_dx.z = rs0 * (s4_gg.z - s4_s1 - xh.z * s4_s2); m_dx.w = rs0 * (s4_gg.w - s4_s1 - xh.w * s4_s2); PIN(m_dx.z); PIN(m_dx.w); break;
      case 10: st4_nt(rows + (size_t)srow * C + 4 * cq, m_dx); break;
      default: PIN(s4_da.z); s4_d(xh); PIN(dgam.x); PIN(dbet.x); break;
    }
  };
  float4 m_a;
  auto s1_step = [&](int k, int buf, float valid) {
    switch (k) {
      case 0: PIN(px.x); pxw = widen4(px); m_sa = (pxw.x + pxw.y) + (pxw.z + pxw.w); m_sa += dpp_read<0xB1>(m_sa); m_sa += dpp_read<0x4E>(m_sa); PIN(m_sa); break;
      case 1: PIN(m_sa); m_sa += dpp_read<0x141>(m_sa); m_sa += dpp_read<0x140>(m_sa); m_sa *= (1.f / C); PIN(m_sa); break;
      case 2: PIN(m_sa); s1_xc = make_float4(pxw.x - m_sa, pxw.y - m_sa, pxw.z - m_sa, pxw.w - m_sa);
              m_sb = (s1_xc.x * s1_xc.x + s1_xc.y * s1_xc.y) + (s1_xc.z * s1_xc.z + s1_xc.w * s1_xc.w); PIN(m_sb); break;
      case 3: PIN(m_sb); m_sb += dpp_read<0xB1>(m_sb); m_sb += dpp_read<0x4E>(m_sb); m_sb += dpp_read<0x141>(m_sb); PIN(m_sb); break;
      case 4: PIN(m_sb); m_sb += dpp_read<0x140>(m_sb); rs2 = rsqrtf(m_sb * (1.f / C) + LN_EPS); PIN(rs2); break;
      case 5: PIN(rs2); xh2 = f4_scale(s1_xc, rs2); PIN(xh2.x); PIN(xh2.w); break;
      case 6: PIN(xh2.y); m_a = make_float4(xh2.x * gq.x + bq.x, xh2.y * gq.y + bq.y, xh2.z * gq.z + bq.z, xh2.w * gq.w + bq.w); PIN(m_a.x); PIN(m_a.w); break;
      case 7: PIN(m_a.y); split4(m_a, s1_ah, s1_al); PIN(s1_ah.x); PIN(s1_al.y); break;
      case 8: PIN(pd.x); pdw = widen4(pd); split4(pdw, s1_dh, s1_dl); PIN(s1_dh.x); PIN(s1_dl.y); break;
      case 9: PIN(pdw.y); db4 = make_float4(fmaf(pdw.x, valid, db4.x), fmaf(pdw.y, valid, db4.y), fmaf(pdw.z, valid, db4.z), fmaf(pdw.w, valid, db4.w)); PIN(db4.x); PIN(db4.w); break;
      case 10: s1_d_(buf); break;
      default: break;
    }
  };

  // ---- prologue: stage 1 of chunk 0, loads of chunk 1; the "previous chunk" of the first iteration is all zeros (its store is skipped)
  for (int i = tid; i < 4 * 16 * LDDA; i += 256) (&sm.DA[1][0][0])[i] = 0.f;
  if (n_mine > 0) {
    fetch(0, px, pd);
    fetch(1, pxn, pdn);
    stage1(0, widen4(px), widen4(pd), xh1, rs1);
    px = pxn; pd = pdn;
  }
  __syncthreads();

#ifndef GRL_M16_BURST
#define GRL_M16_BURST GRL_PREC   // plain-bf16 build: z and dH of a chunk as ONE burst of sixteen MFMAs (the W4^T fragments resident in registers:
#endif                           // no lo halves, they fit), the GELUs and dZ as packed f32 pairs behind it -- the edge backward's recipe (round 5)
#if GRL_M16_BURST
  bf16x8 w4r[4][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int s = 0; s < 2; ++s) w4r[nt][s] = __builtin_bit_cast(bf16x8, sm.W4F[wave][nt][s][0][lane]);
#endif
#ifdef GRL_M16_PHASE
  unsigned long long ph_[12] = {0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
  for (int it = 0; it < n_mine; ++it) {
    const int cb = it & 1;
    const Img16& im = sm.img[cb];
    M16_PH(0);   // barrier wait
    // ---- operand fragments of this chunk (B side of the chain products), prefetch of chunk it + 2
    bf16x8 ah[2], al[2], dh_[2], dl_[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ah[s] = *reinterpret_cast<const bf16x8*>(im.aBh + r * LDBI + 32 * s + 8 * g);
      GRL_LO(al[s] = *reinterpret_cast<const bf16x8*>(im.aBl + r * LDBI + 32 * s + 8 * g);)
      dh_[s] = *reinterpret_cast<const bf16x8*>(im.dBh + r * LDBI + 32 * s + 8 * g);
      GRL_LO(dl_[s] = *reinterpret_cast<const bf16x8*>(im.dBl + r * LDBI + 32 * s + 8 * g);)
    }
    bf16x8 w4h[2][2], w4l[2][2];   // W4^T fragments of the dH groups, requested one n-tile ahead
    auto w4_load = [&](int nt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w4h[nt & 1][s] = __builtin_bit_cast(bf16x8, sm.W4F[wave][nt][s][0][lane]);
        GRL_LO(w4l[nt & 1][s] = __builtin_bit_cast(bf16x8, sm.W4F[wave][nt][s][1][lane]);)
      }
    };
    fetch(it + 2, pxn, pdn);
    BAR();
    M16_PH(1);   // fragment reads issued
    // ---- z = W3 a + b3 (hidden on the registers, row on the lane), then dH = W4^T dOut.  GRL_M16_PAIRS (build switch, off: measured equal, profiles/r03_mlp16_log.txt): the n-tiles go in PAIRS --
    // a region holds the twelve MFMAs of two tiles and the GELU (value + derivative) of the two tiles before: four independent packed
    // chains for the vector pipe instead of two (a lone wave pays ~8 cycles per DEPENDENT instruction, 4 per independent one)
#ifndef GRL_M16_PAIRS
#define GRL_M16_PAIRS 0
#endif
    float4 hv[4], gp[4];
    bf16x8 zh[2], zl[2], hh[2], hl[2];
    {
      f32x4v c[4], e[4];
      float4 dz[4];
      auto z_tile = [&](int nt) {
        c[nt] = f32x4v{b3q[nt].x, b3q[nt].y, b3q[nt].z, b3q[nt].w};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          c[nt] = mfma16(w3f.h[nt][s], ah[s], c[nt]);
          GRL_LO(c[nt] = mfma16(w3f.l[nt][s], ah[s], c[nt]);)
          GRL_LO(c[nt] = mfma16(w3f.h[nt][s], al[s], c[nt]);)
        }
      };
      auto dh_tile = [&](int nt) {
        e[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          e[nt] = mfma16(w4h[nt & 1][s], dh_[s], e[nt]);
          GRL_LO(e[nt] = mfma16(w4l[nt & 1][s], dh_[s], e[nt]);)
          GRL_LO(e[nt] = mfma16(w4h[nt & 1][s], dl_[s], e[nt]);)
        }
      };
      auto dz_tile = [&](int nt) {
        dz[nt] = f4_mul(v4(e[nt]), gp[nt]);
        db3[nt] = f4_add(db3[nt], dz[nt]);
      };
#if GRL_M16_BURST
      static_assert(GRL_PREC, "the burst form keeps W4^T resident: plain-bf16 build only");
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { c[nt] = f32x4v{b3q[nt].x, b3q[nt].y, b3q[nt].z, b3q[nt].w}; e[nt] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w3f.h[nt][s], ah[s], c[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) e[nt] = mfma16(w4r[nt][s], dh_[s], e[nt]);
      }
      BAR();
      M16_PH(2);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) gelu_both4_pk(v4(c[nt]), hv[nt], gp[nt]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dz_tile(nt);
      split_pair(dz[0], dz[1], zh[0], zl[0]);
#elif GRL_M16_PAIRS
      z_tile(0); z_tile(1); w4_load(0); w4_load(1);
      BAR();
      z_tile(2); z_tile(3);
      gelu_both4_pk(v4(c[0]), hv[0], gp[0]); gelu_both4_pk(v4(c[1]), hv[1], gp[1]);
      BAR();
      M16_PH(2);   // z: 24 MFMA, two GELU tiles
      dh_tile(0); dh_tile(1);
      gelu_both4_pk(v4(c[2]), hv[2], gp[2]); gelu_both4_pk(v4(c[3]), hv[3], gp[3]);
      BAR();
      w4_load(2); w4_load(3);
      BAR();
      dh_tile(2); dh_tile(3);
      dz_tile(0); dz_tile(1);
      split_pair(dz[0], dz[1], zh[0], zl[0]);
      BAR();
      dz_tile(2); dz_tile(3);
#else
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        z_tile(nt);
        if (nt > 0) gelu_both4_pk(v4(c[nt - 1]), hv[nt - 1], gp[nt - 1]);
        if (nt == 3) w4_load(0);
        BAR();
      }
      M16_PH(2);   // z: 24 MFMA, three GELU tiles
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt < 3) w4_load(nt + 1);
        dh_tile(nt);
        if (nt == 0) gelu_both4_pk(v4(c[3]), hv[3], gp[3]);
        else dz_tile(nt - 1);
        if (nt == 2) split_pair(dz[0], dz[1], zh[0], zl[0]);
        BAR();
      }
      dz_tile(3);
#endif
      M16_PH(3);   // dH: 24 MFMA, last GELU tile, dZ
      // ---- split dZ (B operand of dA, A operand of dW3); staged for the transposed reads, which are requested at once
      split_pair(dz[2], dz[3], zh[1], zl[1]);
      stage_put<2>(pv.zh, pv.zl, zh, zl, r, g);
    }
    bf16x8 fzh[2], fzl[2], fah[2], fal[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fzh[t] = tr_frag(pv.zh, t, lane);
      GRL_LO(fzl[t] = tr_frag(pv.zl, t, lane);)
      fah[t] = tr_frag(im.aTh, t, lane);
      GRL_LO(fal[t] = tr_frag(im.aTl, t, lane);)
    }
    BAR();
    M16_PH(4);   // split + staging of dZ, transposed reads requested
    // ---- dA^T (partial over this wave's hidden units) = W3^T dZ^T: four channel tiles; the split of h beside them
    {
      f32x4v da[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        da[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          da[ct] = mfma16(w3t.h[ct][s], zh[s], da[ct]);
          GRL_LO(da[ct] = mfma16(w3t.l[ct][s], zh[s], da[ct]);)
          GRL_LO(da[ct] = mfma16(w3t.h[ct][s], zl[s], da[ct]);)
        }
        if (ct == 0) split_pair(hv[0], hv[1], hh[0], hl[0]);
        if (ct == 1) split_pair(hv[2], hv[3], hh[1], hl[1]);
        BAR();
      }
      M16_PH(5);   // dA: 24 MFMA, split of h
      // partial dA rows of this wave (row r, channels 16 ct + 4 g .. + 3); h over dZ in the staging image (whose transposed reads were
      // issued above: in-order LDS); the operands of dW4 and the previous chunk's partial dA rows requested
      float* drow = &sm.DA[cb][wave][0] + r * LDDA + 4 * g;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4*>(drow + 16 * ct) = v4(da[ct]);
    }
    stage_put<2>(pv.zh, pv.zl, hh, hl, r, g);
    bf16x8 fdh[2], fdl[2], fhh[2], fhl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fdh[t] = tr_frag(im.dTh, t, lane);
      GRL_LO(fdl[t] = tr_frag(im.dTl, t, lane);)
      fhh[t] = tr_frag(pv.zh, t, lane);
      GRL_LO(fhl[t] = tr_frag(pv.zl, t, lane);)
    }
    s4_load(cb ^ 1);
    BAR();
    M16_PH(6);   // partial dA rows written, h staged, reads requested
    // ---- dW3 += dZ^T a (hidden x channel): K = the 16 rows; stage 4 of the previous chunk in the gaps of its twelve MFMAs
    // One MFMA (32 matrix-pipe cycles, 8 of them holding the issue port) hides ~5-6 independent vector instructions of the same wave;
    // asm MFMAs are opaque to the scheduler, so the stages are cut into twelve micro-steps each and placed by hand: MFMA, step, MFMA, ...
    st_t* s4_rows = it > 0 ? dx2 + (size_t)chunk_of(it - 1) * 16 * C : dump;
    const float s1_valid = it + 1 < n_mine ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      {
        const int tn = k / 6, tk = (k / 3) & 1, term = k % 3;
#if GRL_PREC
        if (term == 0) mfma32_acc(fzh[tn], fah[tk], aW3[tn][tk]);
#else
        mfma32_acc(term == 1 ? fzl[tn] : fzh[tn], term == 2 ? fal[tk] : fah[tk], aW3[tn][tk]);
#endif
      }
      s4_step(k, s4_rows);
      BAR();
    }
    M16_PH(7);   // dW3: 12 MFMA 32x32, stage 4 of the previous chunk
    // ---- dW4 += dOut^T h (channel x hidden); stage 1 of the next chunk in the gaps
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      {
        const int tn = k / 6, tk = (k / 3) & 1, term = k % 3;
#if GRL_PREC
        if (term == 0) mfma32_acc(fdh[tn], fhh[tk], aW4[tn][tk]);
#else
        mfma32_acc(term == 1 ? fdl[tn] : fdh[tn], term == 2 ? fhl[tk] : fhh[tk], aW4[tn][tk]);
#endif
      }
      s1_step(k, cb ^ 1, s1_valid);
      BAR();
    }
    M16_PH(8);   // dW4: 12 MFMA 32x32, stage 1 of the next chunk
    xh0 = xh1; rs0 = rs1; xh1 = xh2; rs1 = rs2;
    px = pxn; pd = pdn;
    __syncthreads();
  }
  if (n_mine > 0) {
    s4_load((n_mine - 1) & 1);
    s4_a(xh0); s4_b(); s4_c(dx2 + (size_t)chunk_of(n_mine - 1) * 16 * C, xh0, rs0); s4_d(xh0);
  }

#ifdef GRL_M16_PHASE
  if (lane == 0 && wave == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(&g_m16phase[i], ph_[i]);
#endif

  // ---- partial slab of this workgroup
  acc_drain(aW3[0][0], aW3[0][1], aW3[1][0], aW3[1][1], aW4[0][0], aW4[0][1], aW4[1][0], aW4[1][1]);
  float* out = partial + (size_t)blockIdx.x * MLP_PARTIAL;
  float* oW3 = out, *ob3 = oW3 + W * C, *oW4 = ob3 + W, *ob4 = oW4 + C * W, *og = ob4 + C, *obt = og + C;
  {   // 32x32 accumulator element rho of lane (column j = lane & 31, hh = lane >> 5): D[n = 8 (rho >> 2) + 4 hh + (rho & 3)][j]
    const int j = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) {
          const int n = (rho & 3) + 8 * (rho >> 2) + 4 * hh;
          oW3[(size_t)(j0 + 32 * tn + n) * C + 32 * tk + j] = aW3[tn][tk][rho];
          oW4[(size_t)(32 * tn + n) * W + j0 + 32 * tk + j] = aW4[tn][tk][rho];
        }
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {   // db3: sums over the 16 rows (the lanes that share g)
    const float s0 = row16_sum(db3[nt].x), s1 = row16_sum(db3[nt].y), s2 = row16_sum(db3[nt].z), s3 = row16_sum(db3[nt].w);
    if (r == 0) *reinterpret_cast<float4*>(ob3 + j0 + 16 * nt + 4 * g) = make_float4(s0, s1, s2, s3);
  }
  __syncthreads();
  float* red = &sm.DA[0][0][0];   // [16 rows][3][64]
  *reinterpret_cast<float4*>(red + (srow * 3 + 0) * C + 4 * cq) = db4;
  *reinterpret_cast<float4*>(red + (srow * 3 + 1) * C + 4 * cq) = dgam;
  *reinterpret_cast<float4*>(red + (srow * 3 + 2) * C + 4 * cq) = dbet;
  __syncthreads();
  if (tid < 3 * C) {
    const int which = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int g_ = 0; g_ < 16; ++g_) t += red[(g_ * 3 + which) * C + c];
    (which == 0 ? ob4 : which == 1 ? og : obt)[c] = t;
  }
}

}  // namespace

extern "C" {

// Internal entry point (called by grl_node_mlp_bwd of node_mlp.hip when GRL_MLP_BWD16 is on): n_rows must be a multiple of 16 (it is
// n_nodes * 16), `blocks` workgroups are launched and each writes its partial row.
int GRL_ENTRY(grl_node_mlp_bwd16_launch)(const st_t* x2, const st_t* dout, const float* W3, const float* b3, const float* W4,
                                         const float* gamma, const float* beta, st_t* dx2, float* partial, int n_rows, int blocks,
                                         const void* wimg, hipStream_t stream) {
  if (n_rows % 16) return -3;
  static_assert(sizeof(Mlp16Smem) > 80 * 1024, "one workgroup per CU (one wave per SIMD) is a precondition of the unfenced MFMA groups");
  GRL_ONCE(hipFuncSetAttribute((const void*)node_mlp_bwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Mlp16Smem)));
  // partial holds blocks + 1 rows (ABI 201); the first 4 KB of the last one take the dump rows (garbage, shared by all workgroups)
  st_t* dump = reinterpret_cast<st_t*>(partial + (size_t)blocks * MLP_PARTIAL);
  hipLaunchKernelGGL(node_mlp_bwd16_kernel, dim3(blocks), dim3(256), sizeof(Mlp16Smem), stream, x2, dout, W3, b3, W4, gamma, beta, dx2,
                     partial, dump, n_rows / 16, reinterpret_cast<const Mlp16Image*>(wimg));
  GRL_CHECK_LAUNCH();
  return 0;
}

#if defined(GRL_M16_PHASE) && !GRL_PREC
int grl_mlp16_phase_read(unsigned long long* out16, int reset) {
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_m16phase), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_m16phase), z, sizeof(z)); }
  return 0;
}
#endif

}  // extern "C"
