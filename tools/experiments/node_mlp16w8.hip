// Fused backward of the ConvNeXt node block on 16-row tiles with EIGHT waves per workgroup -- two per SIMD (round 4).
// EXPERIMENT, OFF BY DEFAULT (GRL_MLP_BWD_W8=1 selects it): correct -- the parity and determinism tests pass with it -- and slower than
// the four-wave kernel in both builds: split-bf16 0.77 -> 2.50 ms per 4096-frame step (its hi + lo fragments need ~290 registers: eight
// static fragments live in scratch and are re-read every chunk, 5.6 GB per step), plain bf16 (no spills) 1.69 -> 2.29 ms per step of the
// rope workload.  Halving a wave's slice does not halve its instructions (fragment reads, addressing, waits: 575 -> ~420 per chunk), and
// two waves per SIMD issue these mixes barely faster than one (+8 %): DESIGN.md finding 47, profiles/r04_mlp_bwd_w8.txt.
//
// The idea it tested:
// node_mlp_bwd16_kernel (node_mlp16.hip) runs one wave per SIMD: wave w owns 64 hidden units, 128 static operand registers and 128
// accumulator registers.  A lone wave issues one instruction of any kind per ~5 cycles (DESIGN.md findings 13, 43): its ~900
// instructions per chunk are ~4 500 cycles whatever the pipes could do.  This kernel keeps that kernel's architecture and halves the
// slice: wave w of EIGHT owns hidden units [32 w, 32 w + 32) -- 64 static operand registers (W3 and W3^T fragments; the W4^T fragments
// stay in LDS), 64 accumulator registers, half the MFMAs, GELUs and splits per chunk -- so that two waves share a SIMD and the SIMD
// issues twice as often.  What changes with it:
//   * the per-chunk row stages are divided by ROLE instead of being run by every wave: waves 4-7 run stage 1 of chunk i + 1 (LayerNorm,
//     split-bf16 operand images, db4), waves 0-3 stage 4 of chunk i - 1 (sum of the partial dA rows, LayerNorm backward, dx2 store,
//     dgamma / dbeta); a row's (mean, rstd) travel from stage 1 to stage 4 through a 128-byte LDS slot per chunk, stage 4 re-reads its
//     quad of x2 (an L2 hit: the chunk was read two iterations earlier);
//   * eight partial dA rows per chunk instead of four: the LDS budget (36 KB images + 64 KB W4^T fragments + 16 KB staging) leaves room
//     for ONE buffer of them, so a chunk has two barriers: the stage-4 waves sum the previous chunk's partial rows right behind the
//     first, every wave writes its partial rows behind the second;
//   * two waves per SIMD: no operand register may be re-loaded while an MFMA that reads it can still be queued (DESIGN.md findings 3,
//     15).  Inside an iteration every fragment is requested after the results of all earlier MFMA groups have been consumed by vector
//     instructions; the weight-gradient MFMAs (asm, accumulate-only) end an iteration, so a one-MFMA fence whose result IS read closes
//     it (the matrix pipe is in order: when the fence has finished, everything in front of it has read its operands).
// Same arithmetic per element as node_mlp_bwd16_kernel; the partial dA rows are summed as ((0+1)+(2+3)) + ((4+5)+(6+7)) instead of
// (0+1)+(2+3) over 64-unit slices, so dx2 differs in the last bits.  Partial slab per workgroup unchanged.
#include "grl_tile16.h"
#include "grl_wimg.h"

namespace {

constexpr int C = 64, W = 256;
constexpr float LN_EPS = 1e-5f;
constexpr int MLP_PARTIAL = W * C + W + C * W + C + C + C;
constexpr int LDBI = 64 + 16;   // bf16 elements per row of a B-operand image (160-byte rows)
constexpr int LDDA = 64 + 4;    // floats per row of a partial dA image
constexpr int NW8 = 8;          // waves per workgroup
constexpr int STG32 = 16 * 32;  // bf16 elements of a 16-row x 32-feature staging image

// staging image of 16 rows x 32 features: the layout of grl_tile16.h's stg_off with two feature blocks per block row instead of four
GRL_DEVINL int stg_off32(int row, int feat_quad /* feature >> 2, 0..7 */) {
  const int fb = feat_quad >> 2;
  return (((row >> 2) * 2 + fb) * 64) + (((row + fb) & 3) * 16) + (((feat_quad & 3) ^ (row >> 2)) << 2);
}
// fragment (8 rows x this lane's feature) of such an image for a 32x32x16 operand: lane (m = l & 31, h = l >> 5) gets rows 8 h + j of feature m
GRL_DEVINL bf16x8 tr_frag32(const unsigned short* img, int lane) {
  const int h = lane >> 5, fb = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int qs = ((q + fb) & 3) * 16;
  const unsigned short* base = img + ((2 * h) * 2 + fb) * 64 + qs + ((p ^ (2 * h)) << 2);
  const unsigned short* base1 = img + ((2 * h + 1) * 2 + fb) * 64 + qs + ((p ^ (2 * h + 1)) << 2);
  const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)base);
  const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)base1);
  typedef short v8s16 __attribute__((ext_vector_type(8)));
  const v8s16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// this lane's chain-layout fragment (row r, features 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3) -> image[row][feature]
GRL_DEVINL void stage_put32(unsigned short* ih, unsigned short* il, const bf16x8& fh, const bf16x8& fl, int r, int g) {
  const int o0 = stg_off32(r, g), o1 = stg_off32(r, 4 + g);
  const u32x4 h = __builtin_bit_cast(u32x4, fh);
  *reinterpret_cast<uint2*>(ih + o0) = make_uint2(h[0], h[1]);
  *reinterpret_cast<uint2*>(ih + o1) = make_uint2(h[2], h[3]);
#if !GRL_PREC
  const u32x4 l = __builtin_bit_cast(u32x4, fl);
  *reinterpret_cast<uint2*>(il + o0) = make_uint2(l[0], l[1]);
  *reinterpret_cast<uint2*>(il + o1) = make_uint2(l[2], l[3]);
#else
  (void)il; (void)fl;
#endif
}

struct Img16 {
  unsigned short aBh[16 * LDBI], aBl[16 * LDBI];   // a = LN(x2): B-operand layout [row][channel]
  unsigned short dBh[16 * LDBI], dBl[16 * LDBI];   // dOut
  unsigned short aTh[STG], aTl[STG];               // the same rows in the transposed-read layout (stg_off)
  unsigned short dTh[STG], dTl[STG];
};
struct Priv32 {
  unsigned short zh[STG32], zl[STG32];   // dZ rows of this wave's 32 hidden units; then (same bytes) h = GELU(z)
};
struct Mlp16w8Smem {
  Img16 img[2];
  float DA[NW8][16 * LDDA];          // the eight partial dA rows of ONE chunk
  Priv32 priv[NW8];
  u32x4 W4F[16][2][2][64];           // [hidden tile][k-step][hi | lo][lane]: the W4^T operand fragments of all sixteen 16-unit tiles
  float ln[4][16][2];                // (mean, rstd) of a chunk's rows, slot = chunk index & 3: stage 1 -> stage 4
};
static_assert(sizeof(Mlp16w8Smem) <= 160 * 1024, "LDS budget");
static_assert(sizeof(Mlp16w8Smem) > 80 * 1024, "one workgroup per CU");

template <int CTRL>
GRL_DEVINL float dpp_read(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
GRL_DEVINL float row16_sum(float v) {
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
GRL_DEVINL void mfma32_acc_v(const bf16x8& a, const bf16x8& b, f32x16& c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define BAR() __builtin_amdgcn_sched_barrier(0)
#define PIN(x) asm volatile("" : "+v"(x))

__global__ __launch_bounds__(64 * NW8, 1) void node_mlp_bwd16w8_kernel(const st_t* __restrict__ x2, const st_t* __restrict__ dout,
                                                                       const float* __restrict__ W3, const float* __restrict__ b3,
                                                                       const float* __restrict__ W4, const float* __restrict__ gam,
                                                                       const float* __restrict__ bet, st_t* __restrict__ dx2,
                                                                       float* __restrict__ partial, st_t* __restrict__ dump, int n_chunks,
                                                                       const Mlp16Image* __restrict__ wimg) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  Mlp16w8Smem& sm = *reinterpret_cast<Mlp16w8Smem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const bool role4 = wave < 4;                          // waves 0-3: stage 4 of the previous chunk; waves 4-7: stage 1 of the next
  const int srow = (tid & 255) >> 4, cq = tid & 15;     // the role's thread = (row of the chunk, channel quad)
  // (gamma / beta quads are re-read where a row stage uses them: 8 registers less across the MFMA phases)
#define GQ() (*reinterpret_cast<const float4*>(gam + 4 * cq))
#define BQ() (*reinterpret_cast<const float4*>(bet + 4 * cq))
  Priv32& pv = sm.priv[wave];

  // ---- this wave's static operand fragments (hidden units j0 .. j0 + 31 = 16-unit tiles 2 wave, 2 wave + 1)
  const int j0 = 32 * wave;
  bf16x8 w3fh[2][2], w3fl[2][2];   // z^T = W3 a^T:      A[m = hidden 16 nt + r][k = channel 32 s + 8 g + j]
  bf16x8 w3th[4], w3tl[4];         // dA^T = W3^T dZ^T:  A[m = channel 16 ct + r][k = this wave's hidden, chain order 16 (j >> 2) + 4 g + (j & 3)]
  if (wimg) {   // the 4-wave kernel's image (grl_weight_images kind 3): its (wave, n-tile) pairs are the sixteen hidden tiles in order, and
                // k-step s of its wave v is hidden units 64 v + 32 s .. + 31 = this kernel's wave 2 v + s
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int tile = 2 * wave + nt;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w3fh[nt][s] = __builtin_bit_cast(bf16x8, wimg->w3f[tile >> 2][tile & 3][s][0][lane]);
        GRL_LO(w3fl[nt][s] = __builtin_bit_cast(bf16x8, wimg->w3f[tile >> 2][tile & 3][s][1][lane]);)
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      w3th[ct] = __builtin_bit_cast(bf16x8, wimg->w3t[wave >> 1][ct][wave & 1][0][lane]);
      GRL_LO(w3tl[ct] = __builtin_bit_cast(bf16x8, wimg->w3t[wave >> 1][ct][wave & 1][1][lane]);)
    }
    // W4^T fragments of ALL tiles into LDS: 16 * 2 * 2 * 64 entries of 16 bytes, 512 threads
    const u32x4* src = &wimg->w4f[0][0][0][0][0];
    u32x4* dst = &sm.W4F[0][0][0][0];
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[tid + 512 * i] = src[tid + 512 * i];
  } else {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        {
          const float* p = W3 + (size_t)(j0 + 16 * nt + r) * C + 32 * s + 8 * g;
          split_pair(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), w3fh[nt][s], w3fl[nt][s]);
        }
        {   // dH^T = W4^T dOut^T: A[m = hidden][k = channel c] = W4[c][hidden]
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = W4[(size_t)(32 * s + 8 * g + j) * W + j0 + 16 * nt + r];
          bf16x8 fh, fl;
          split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), fh, fl);
          sm.W4F[2 * wave + nt][s][0][lane] = __builtin_bit_cast(u32x4, fh);
          GRL_LO(sm.W4F[2 * wave + nt][s][1][lane] = __builtin_bit_cast(u32x4, fl);)
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = W3[(size_t)(j0 + 16 * (j >> 2) + 4 * g + (j & 3)) * C + 16 * ct + r];
      split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), w3th[ct], w3tl[ct]);
    }
  }
  // (no AGPR is used by this kernel: with accumulators or static fragments pinned to AGPRs the compiler splits the 256 registers of a
  //  two-wave kernel 128 / 128 and spills ~160 of the working set; all-VGPR leaves ~40 static fragment registers in scratch)
  f32x16 aW3[2], aW4[2];   // dW3[this wave's 32 hidden][channel tile tk], dW4[channel tile tn][this wave's 32 hidden]
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_) { aW3[a_] = zero16(); aW4[a_] = zero16(); }
  float4 db3[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) db3[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam, db4 = dgam;

  const int n_mine = blockIdx.x < n_chunks ? (n_chunks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto chunk_of = [&](int i) { return blockIdx.x + i * gridDim.x; };
  auto clampi = [&](int i) { return i < 0 ? 0 : (i < n_mine ? i : n_mine - 1); };
  // stage 1's inputs: this thread's quad of x2 and dOut of chunk i (clamped: the loads of a non-existent chunk are never used)
  float4 px = make_float4(0.f, 0.f, 0.f, 0.f), pd = px;
  auto fetch1 = [&](int i, float4& fx, float4& fd) {
    const size_t gofs = ((size_t)chunk_of(clampi(i)) * 16 + srow) * C + 4 * cq;
    fx = ld4_nt(x2 + gofs);
    fd = ld4_nt(dout + gofs);
  };
  // stage 4's input: the quad of x2 again (stage 1 ran in other waves two iterations earlier)
  float4 qx = px;
  auto fetch4 = [&](int i) { qx = ld4(x2 + ((size_t)chunk_of(clampi(i)) * 16 + srow) * C + 4 * cq); };

  // ---- stage 1 pieces (role 1).  The pieces and their order follow node_mlp16.hip; xh / rstd are not kept: (mean, rstd) go to sm.ln
  float4 s1_xc;
  uint2 s1_ah, s1_al, s1_dh, s1_dl;
  float m_sa = 0.f, m_sb = 0.f, s1_rs = 0.f, s1_mean = 0.f;
  float4 m_a;
  auto s1_images = [&](int buf) {
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
  auto s1_step = [&](int k, int buf, int slot, float valid) {
    switch (k) {
      case 0: PIN(px.x); m_sa = (px.x + px.y) + (px.z + px.w); m_sa += dpp_read<0xB1>(m_sa); m_sa += dpp_read<0x4E>(m_sa); PIN(m_sa); break;
      case 1: PIN(m_sa); m_sa += dpp_read<0x141>(m_sa); m_sa += dpp_read<0x140>(m_sa); m_sa *= (1.f / C); s1_mean = m_sa; PIN(m_sa); break;
      case 2: PIN(m_sa); s1_xc = make_float4(px.x - m_sa, px.y - m_sa, px.z - m_sa, px.w - m_sa);
              m_sb = (s1_xc.x * s1_xc.x + s1_xc.y * s1_xc.y) + (s1_xc.z * s1_xc.z + s1_xc.w * s1_xc.w); PIN(m_sb); break;
      case 3: PIN(m_sb); m_sb += dpp_read<0xB1>(m_sb); m_sb += dpp_read<0x4E>(m_sb); m_sb += dpp_read<0x141>(m_sb); PIN(m_sb); break;
      case 4: PIN(m_sb); m_sb += dpp_read<0x140>(m_sb); s1_rs = rsqrtf(m_sb * (1.f / C) + LN_EPS); PIN(s1_rs); break;
      case 5: PIN(s1_rs); s1_xc = f4_scale(s1_xc, s1_rs); PIN(s1_xc.x); PIN(s1_xc.w); break;
      case 6: PIN(s1_xc.y); { const float4 gq = GQ(), bq = BQ(); m_a = make_float4(s1_xc.x * gq.x + bq.x, s1_xc.y * gq.y + bq.y, s1_xc.z * gq.z + bq.z, s1_xc.w * gq.w + bq.w); } PIN(m_a.x); PIN(m_a.w); break;
      case 7: PIN(m_a.y); split4(m_a, s1_ah, s1_al); PIN(s1_ah.x); PIN(s1_al.y); break;
      case 8: PIN(pd.x); split4(pd, s1_dh, s1_dl); PIN(s1_dh.x); PIN(s1_dl.y); break;
      case 9: PIN(pd.y); db4 = make_float4(fmaf(pd.x, valid, db4.x), fmaf(pd.y, valid, db4.y), fmaf(pd.z, valid, db4.z), fmaf(pd.w, valid, db4.w)); PIN(db4.x); PIN(db4.w); break;
      case 10: s1_images(buf); break;
      default: if (cq == 0) { sm.ln[slot][srow][0] = s1_mean; sm.ln[slot][srow][1] = s1_rs; } break;
    }
  };
  // ---- stage 4 pieces (role 0)
  float4 s4_da = px, s4_gg = px, s4_xh = px, m_dx = px;
  float s4_s1 = 0.f, s4_s2 = 0.f, s4_rs = 0.f, s4_mean = 0.f;
  auto s4_sum = [&]() {   // the eight partial dA rows of this thread's quad, pairwise in a fixed order
    const float* dap = &sm.DA[0][0] + srow * LDDA + 4 * cq;
    float4 p_[NW8];
#pragma unroll
    for (int w_ = 0; w_ < NW8; ++w_) p_[w_] = *reinterpret_cast<const float4*>(dap + w_ * 16 * LDDA);
    s4_da = f4_add(f4_add(f4_add(p_[0], p_[1]), f4_add(p_[2], p_[3])), f4_add(f4_add(p_[4], p_[5]), f4_add(p_[6], p_[7])));
  };
  auto s4_ln = [&](int slot) { s4_mean = sm.ln[slot][srow][0]; s4_rs = sm.ln[slot][srow][1]; };
  auto s4_step = [&](int k, st_t* rows) {
    switch (k) {
      case 0: PIN(qx.x); s4_xh = make_float4((qx.x - s4_mean) * s4_rs, (qx.y - s4_mean) * s4_rs, (qx.z - s4_mean) * s4_rs, (qx.w - s4_mean) * s4_rs); PIN(s4_xh.x); PIN(s4_xh.w); break;
      case 1: PIN(s4_da.y); s4_gg = f4_mul(s4_da, GQ()); PIN(s4_gg.x); PIN(s4_gg.w); break;
      case 2: PIN(s4_gg.y); m_sa = (s4_gg.x + s4_gg.y) + (s4_gg.z + s4_gg.w); m_sb = s4_gg.x * s4_xh.x + s4_gg.y * s4_xh.y; PIN(m_sa); PIN(m_sb); break;
      case 3: PIN(m_sb); m_sb = fmaf(s4_gg.z, s4_xh.z, fmaf(s4_gg.w, s4_xh.w, m_sb)); m_sa += dpp_read<0xB1>(m_sa); m_sb += dpp_read<0xB1>(m_sb); PIN(m_sa); PIN(m_sb); break;
      case 4: PIN(m_sa); m_sa += dpp_read<0x4E>(m_sa); m_sb += dpp_read<0x4E>(m_sb); m_sa += dpp_read<0x141>(m_sa); m_sb += dpp_read<0x141>(m_sb); PIN(m_sa); PIN(m_sb); break;
      case 5: PIN(m_sa); m_sa += dpp_read<0x140>(m_sa); m_sb += dpp_read<0x140>(m_sb); s4_s1 = m_sa * (1.f / C); s4_s2 = m_sb * (1.f / C); PIN(s4_s1); PIN(s4_s2); break;
      case 6: PIN(s4_s1); m_dx.x = s4_rs * (s4_gg.x - s4_s1 - s4_xh.x * s4_s2); m_dx.y = s4_rs * (s4_gg.y - s4_s1 - s4_xh.y * s4_s2); PIN(m_dx.x); PIN(m_dx.y); break;
      case 7: PIN(s4_s2); m_dx.z = s4_rs * (s4_gg.z - s4_s1 - s4_xh.z * s4_s2); m_dx.w = s4_rs * (s4_gg.w - s4_s1 - s4_xh.w * s4_s2); PIN(m_dx.z); PIN(m_dx.w); break;
      case 8: st4_nt(rows + (size_t)srow * C + 4 * cq, m_dx); break;
      case 9: PIN(s4_da.z);
              dgam = make_float4(fmaf(s4_da.x, s4_xh.x, dgam.x), fmaf(s4_da.y, s4_xh.y, dgam.y), fmaf(s4_da.z, s4_xh.z, dgam.z), fmaf(s4_da.w, s4_xh.w, dgam.w));
              PIN(dgam.x); break;
      case 10: dbet = f4_add(dbet, s4_da); PIN(dbet.x); break;
      default: break;
    }
  };

  // ---- prologue: zero partial rows and LayerNorm slots (the "previous chunk" of the first iteration), stage 1 of chunk 0
  for (int i = tid; i < NW8 * 16 * LDDA; i += 64 * NW8) (&sm.DA[0][0])[i] = 0.f;
  if (tid < 4 * 16 * 2) (&sm.ln[0][0][0])[tid] = 0.f;
  __syncthreads();   // (the slots are zeroed before stage 1 of chunk 0 writes slot 0)
  if (n_mine > 0 && !role4) {
    fetch1(0, px, pd);
#pragma unroll
    for (int k = 0; k < 12; ++k) s1_step(k, 0, 0, 1.f);
  }
  __syncthreads();

#pragma unroll 1
  for (int it = 0; it < n_mine; ++it) {
    const int cb = it & 1;
    const Img16& im = sm.img[cb];
    // ---- operand fragments of this chunk (B side of the chain products), both W4^T tiles; the roles' loads
    bf16x8 ah[2], al[2], dh_[2], dl_[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ah[s] = *reinterpret_cast<const bf16x8*>(im.aBh + r * LDBI + 32 * s + 8 * g);
      GRL_LO(al[s] = *reinterpret_cast<const bf16x8*>(im.aBl + r * LDBI + 32 * s + 8 * g);)
    }
    float4 b3q[2];   // (re-read per chunk -- an L1 hit -- instead of 8 registers held across the whole loop)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b3q[nt] = *reinterpret_cast<const float4*>(b3 + j0 + 16 * nt + 4 * g);
    bf16x8 w4h[2], w4l[2];
    auto w4_load = [&](int nt) {   // one hidden tile's W4^T fragments at a time (16 registers instead of 32)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w4h[s] = __builtin_bit_cast(bf16x8, sm.W4F[2 * wave + nt][s][0][lane]);
        GRL_LO(w4l[s] = __builtin_bit_cast(bf16x8, sm.W4F[2 * wave + nt][s][1][lane]);)
      }
    };
    auto dh_load = [&]() {   // requested behind the first z tile (every register they can land in was last read by an MFMA whose result has been consumed)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dh_[s] = *reinterpret_cast<const bf16x8*>(im.dBh + r * LDBI + 32 * s + 8 * g);
        GRL_LO(dl_[s] = *reinterpret_cast<const bf16x8*>(im.dBl + r * LDBI + 32 * s + 8 * g);)
      }
      w4_load(0);
    };
    if (role4) {   // previous chunk: its partial dA rows (summed now: the buffer is rewritten behind the second barrier), its x2 quad, its LayerNorm row
      s4_sum();
      s4_ln((it - 1) & 3);
      fetch4(it - 1);
    } else {
      fetch1(it + 1, px, pd);   // used by this iteration's stage 1, behind the chain products (~1 500 cycles later)
    }
    BAR();
    // ---- z = W3 a + b3 (this wave's two hidden tiles), GELU with derivative, dH = W4^T dOut, dZ = dH * gelu'
    float4 hv[2], gp[2];
    bf16x8 zh, zl, hh, hl;
    {
      f32x4v c[2], e[2];
      float4 dz[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        c[nt] = f32x4v{b3q[nt].x, b3q[nt].y, b3q[nt].z, b3q[nt].w};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          c[nt] = mfma16(w3fh[nt][s], ah[s], c[nt]);
          GRL_LO(c[nt] = mfma16(w3fl[nt][s], ah[s], c[nt]);)
          GRL_LO(c[nt] = mfma16(w3fh[nt][s], al[s], c[nt]);)
        }
        if (nt > 0) { gelu_both4(v4(c[0]), hv[0], gp[0]); dh_load(); }
        BAR();
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        e[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          e[nt] = mfma16(w4h[s], dh_[s], e[nt]);
          GRL_LO(e[nt] = mfma16(w4l[s], dh_[s], e[nt]);)
          GRL_LO(e[nt] = mfma16(w4h[s], dl_[s], e[nt]);)
        }
        if (nt == 0) {
          gelu_both4(v4(c[1]), hv[1], gp[1]);
          BAR();
          // tile 0's result is consumed BEFORE tile 1's fragments are requested: they may land in tile 0's registers (two waves per SIMD)
          dz[0] = f4_mul(v4(e[0]), gp[0]);
          db3[0] = f4_add(db3[0], dz[0]);
          BAR();
          w4_load(1);
        }
        BAR();
      }
      dz[1] = f4_mul(v4(e[1]), gp[1]);
      db3[1] = f4_add(db3[1], dz[1]);
      split_pair(dz[0], dz[1], zh, zl);
      stage_put32(pv.zh, pv.zl, zh, zl, r, g);
    }
    bf16x8 fzh, fzl, fah[2], fal[2];
    fzh = tr_frag32(pv.zh, lane);
    GRL_LO(fzl = tr_frag32(pv.zl, lane);)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fah[t] = tr_frag(im.aTh, t, lane);
      GRL_LO(fal[t] = tr_frag(im.aTl, t, lane);)
    }
    BAR();
    // ---- dA^T (partial over this wave's 32 hidden units) = W3^T dZ^T: four channel tiles, one k-step; the split of h beside them
    f32x4v da[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      da[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
      da[ct] = mfma16(w3th[ct], zh, da[ct]);
      GRL_LO(da[ct] = mfma16(w3tl[ct], zh, da[ct]);)
      GRL_LO(da[ct] = mfma16(w3th[ct], zl, da[ct]);)
      if (ct == 1) split_pair(hv[0], hv[1], hh, hl);
      BAR();
    }
    // second barrier of the chunk: the stage-4 waves have summed the previous chunk's partial rows (at the iteration's top)
    __syncthreads();
    {
      float* drow = &sm.DA[wave][0] + r * LDDA + 4 * g;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4*>(drow + 16 * ct) = v4(da[ct]);
    }
    stage_put32(pv.zh, pv.zl, hh, hl, r, g);   // h over dZ (whose transposed reads were issued above: in-order LDS)
    // ---- dW3 += dZ^T a (this wave's 32 hidden x 64 channels): six asm MFMAs; then -- behind a fence, so that the second set of fragments
    //      cannot land in registers a queued MFMA still reads -- dW4 += dOut^T h (64 channels x 32 hidden): six more.  The role's row
    //      stage in the gaps, one micro-step per MFMA (stage 4 of the previous chunk | stage 1 of the next).
    st_t* s4_rows = it > 0 ? dx2 + (size_t)chunk_of(it - 1) * 16 * C : dump;
    const float s1_valid = it + 1 < n_mine ? 1.f : 0.f;
    auto w3_mfma = [&](int k) {
      const int tile = (k / 3) & 1, term = k % 3;
#if GRL_PREC
      if (term == 0) mfma32_acc_v(fzh, fah[tile], aW3[tile]);
#else
      mfma32_acc_v(term == 1 ? fzl : fzh, term == 2 ? fal[tile] : fah[tile], aW3[tile]);
#endif
    };
    auto fence = [&]() {   // one more MFMA whose result IS read: the matrix pipe is in order
      f32x4v f = mfma16(w3th[0], zh, f32x4v{0.f, 0.f, 0.f, 0.f});
      float sink = f[0];
      asm volatile("" :: "v"(sink));
    };
    if (role4) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { w3_mfma(k); s4_step(k, s4_rows); BAR(); }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) { w3_mfma(k); s1_step(k, cb ^ 1, (it + 1) & 3, s1_valid); BAR(); }
    }
    fence();
    bf16x8 fdh[2], fdl[2], fhh, fhl;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fdh[t] = tr_frag(im.dTh, t, lane);
      GRL_LO(fdl[t] = tr_frag(im.dTl, t, lane);)
    }
    fhh = tr_frag32(pv.zh, lane);
    GRL_LO(fhl = tr_frag32(pv.zl, lane);)
    BAR();
    auto w4_mfma = [&](int k) {
      const int tile = (k / 3) & 1, term = k % 3;
#if GRL_PREC
      if (term == 0) mfma32_acc_v(fdh[tile], fhh, aW4[tile]);
#else
      mfma32_acc_v(term == 1 ? fdl[tile] : fdh[tile], term == 2 ? fhl : fhh, aW4[tile]);
#endif
    };
    if (role4) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { w4_mfma(k); s4_step(6 + k, s4_rows); BAR(); }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) { w4_mfma(k); s1_step(6 + k, cb ^ 1, (it + 1) & 3, s1_valid); BAR(); }
    }
    fence();
    __syncthreads();
  }
  if (n_mine > 0 && role4) {   // stage 4 of the last chunk
    s4_sum();
    s4_ln((n_mine - 1) & 3);
    fetch4(n_mine - 1);
    st_t* rows = dx2 + (size_t)chunk_of(n_mine - 1) * 16 * C;
#pragma unroll
    for (int k = 0; k < 12; ++k) s4_step(k, rows);
  }

  // ---- partial slab of this workgroup
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  asm volatile("" : "+v"(aW3[0])); asm volatile("" : "+v"(aW3[1])); asm volatile("" : "+v"(aW4[0])); asm volatile("" : "+v"(aW4[1]));
  float* out = partial + (size_t)blockIdx.x * MLP_PARTIAL;
  float* oW3 = out, *ob3 = oW3 + W * C, *oW4 = ob3 + W, *ob4 = oW4 + C * W, *og = ob4 + C, *obt = og + C;
  {   // 32x32 accumulator element rho of lane (column j = lane & 31, hh = lane >> 5): D[n = 8 (rho >> 2) + 4 hh + (rho & 3)][j]
    const int j = lane & 31, hh_ = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rho = 0; rho < 16; ++rho) {
        const int n = (rho & 3) + 8 * (rho >> 2) + 4 * hh_;
        oW3[(size_t)(j0 + n) * C + 32 * t + j] = aW3[t][rho];       // D[n = hidden][j = channel of tile t]
        oW4[(size_t)(32 * t + n) * W + j0 + j] = aW4[t][rho];       // D[n = channel of tile t][j = hidden]
      }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {   // db3: sums over the 16 rows (the lanes that share g)
    const float s0 = row16_sum(db3[nt].x), s1 = row16_sum(db3[nt].y), s2 = row16_sum(db3[nt].z), s3 = row16_sum(db3[nt].w);
    if (r == 0) *reinterpret_cast<float4*>(ob3 + j0 + 16 * nt + 4 * g) = make_float4(s0, s1, s2, s3);
  }
  __syncthreads();
  float* red = &sm.DA[0][0];   // [16 rows][3][64]
  if (!role4) *reinterpret_cast<float4*>(red + (srow * 3 + 0) * C + 4 * cq) = db4;
  else {
    *reinterpret_cast<float4*>(red + (srow * 3 + 1) * C + 4 * cq) = dgam;
    *reinterpret_cast<float4*>(red + (srow * 3 + 2) * C + 4 * cq) = dbet;
  }
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

// The eight-wave form of grl_node_mlp_bwd16_launch (same arguments, same partial slab; node_mlp.hip selects it).
int GRL_ENTRY(grl_node_mlp_bwd16w8_launch)(const st_t* x2, const st_t* dout, const float* W3, const float* b3, const float* W4,
                                           const float* gamma, const float* beta, st_t* dx2, float* partial, int n_rows, int blocks,
                                           const void* wimg, hipStream_t stream) {
  if (n_rows % 16) return -3;
  GRL_ONCE(hipFuncSetAttribute((const void*)node_mlp_bwd16w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Mlp16w8Smem)));
  st_t* dump = reinterpret_cast<st_t*>(partial + (size_t)blocks * MLP_PARTIAL);
  hipLaunchKernelGGL(node_mlp_bwd16w8_kernel, dim3(blocks), dim3(64 * NW8), sizeof(Mlp16w8Smem), stream, x2, dout, W3, b3, W4, gamma, beta,
                     dx2, partial, dump, n_rows / 16, reinterpret_cast<const Mlp16Image*>(wimg));
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
