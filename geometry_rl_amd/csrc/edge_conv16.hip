// Edge pipeline on 16-row tiles (round 2) of the separable fiber-bundle convolution (reference: hepi.py:76-82,109-123,145-157,
// ponita/conv.py:79-86,115-149): ONE EDGE (= its 16 orientation rows) per wave pass on v_mfma_f32_16x16x32_bf16, instead of two edges
// (32 rows) on 32x32x16.  Kernels of this file:
//   edge16_kernel<0> forward (three waves per SIMD), <2> per-edge messages, <1> d x_src alone (not used by the library any more);
//   edge_bwd16_kernel: the WHOLE backward (d x_src + the five weight gradients) in one launch, one wave per SIMD -- see its header.
// Entry points: grl_edge16_launch / grl_edge_bwd16_launch (called by the C-ABI functions of edge_conv.hip; internal).
//
// Why (DESIGN.md findings 13, 17; profiles/r02_edge_phase_*.txt): the 32-row kernels need 218-232 registers, so only two waves share a
// SIMD; a wave issues one vector instruction per ~5 cycles, the exact-erf GELU between the layers is ~190 dependent-ish instructions per
// 32x32 tile, and the phase timing shows every wave running at its own issue pace with the vector pipe ~50 % busy, plus 25-45 % of the
// time in exposed index -> position -> row round trips at pass and tile heads.  Half the rows per wave halves every per-row register
// array (activations, fragments, gathered row, destination accumulator): ~160 registers, THREE waves per SIMD.  One edge per pass also
// removes the half-empty passes of odd edge counts and the two-destination select of the message epilogue, and makes all per-pass
// metadata wave-uniform: it is fetched for 64 edges at a time (one coalesced load per array + one gather of the positions) and handed out
// with v_readlane -- no dependent global load is left inside the pass loop.
//
// Layout: lane l = (row r = l & 15 = orientation, k-group g = l >> 4).  D[n][r] += sum_k W[n][k] X[r][k] with the weight tile (16 output
// features n) on the A side and the activation rows on the B side; a 16x16 accumulator (f32x4) of n-tile nt holds features
// 16 nt + 4 g + u in element u -- two consecutive n-tiles ARE the 8 B-operand elements of one K-step of the next layer (k order inside a
// 32-block: position 8 g + j <-> feature 16 (j >> 2) + 4 g + (j & 3); the weight images are staged in that order), so the chain stays
// in registers exactly as in the 32-row kernels.  Three waves per SIMD: every MFMA group is fenced (all fragment loads, then the
// MFMAs, then a read of the accumulator: DESIGN.md finding 3).
#ifndef GRL_B16_BURST
#define GRL_B16_BURST 1   // fp32 build: MFMA bursts per layer + packed epilogues in the fused backward (see edge_bwd16_kernel; 0 = the grouped form of rounds 2-4)
#endif
#define GRL_PK_F4 1   // (plain-bf16 build, or the fp32 burst experiment: packed f32 pairs for the element-wise products, grl_common.h)
#include "grl_tile16.h"
#include "grl_wimg.h"
#include <cstdlib>

namespace {

constexpr int C = 64, O = 16;

#ifndef GRL_E16_WAVES
#define GRL_E16_WAVES 4                  // waves per workgroup
#endif
#ifndef GRL_E16_WGS
#define GRL_E16_WGS 3                    // workgroups per CU the register budget is cut for (and the grid is capped at)
#endif
constexpr int E16_WAVES = GRL_E16_WAVES, E16_THREADS = 64 * E16_WAVES;
constexpr int LD1 = WI_LD1, LD2 = WI_LD2;   // image row lengths, ChainW16 / Edge16Image, stage16: grl_wimg.h (shared with the image producer)

struct Edge16Params {
  const st_t* x_in;       // rows gathered per edge: x_src [Ns,16,64] (forward / messages) or dx1 [Nd or E,16,64] (d x_src kernel)
  const float* pos_src;   // [Ns,3]
  const float* pos_dst;   // [Nd,3]
  const int* rowptr;      // [Na+1] anchor CSR (anchor = destination in the forward, source in the backward)
  const int* e_src;       // [E] in anchor-sorted order
  const int* e_dst;       // [E]
  const int* erow;        // optional [E]: row of x_in for the i-th edge (attention backward: per-edge gradient rows); else the node id
  const float* grid, *W1, *b1, *W2, *b2, *Wk;
  int n_anchor, n_edges, dim;
  int anchor_is_dst;      // 1: forward order (gather by source, accumulate per destination); 0: source order (the reverse)
  int per_edge;           // 1: x_in rows are per edge (erow or the edge's own position)
  int npw;                // anchor nodes per wave chunk, 1..16 (host: as large as still leaves every wave slot several chunks)
  const int* split;       // optional [4 gridDim.x + 1] (forward only): wave slot s walks the anchor nodes split[s] .. split[s + 1]
  const void* wimg;       // optional pre-split image (grl_weight_images: a ChainW16 for the forward kernels, an Edge16Image for the backward)
};

GRL_DEVINL void load_w16(ChainW16& s, const Edge16Params& p) {
#ifdef GRL_KNOCK_STAGE   // timing knock-out: no weight staging (results are wrong)
  return;
#endif
  if (p.wimg) {   // the image of this forward pass (built once per step): a linear copy
    copy_image<E16_THREADS>(&s, p.wimg, (int)sizeof(ChainW16));
    return;
  }
  stage16<14, 32, E16_THREADS>(s.W1h, s.W1l, p.W1, LD1);
  stage16<64, 64, E16_THREADS>(s.W2h, s.W2l, p.W2, LD2);
  stage16<64, 64, E16_THREADS>(s.Wkh, s.Wkl, p.Wk, LD2);
  stage_chain16_small<E16_THREADS>(s, p.b1, p.b2, p.grid);
}

// one fenced group: acc(16 features of n-tile nt x 16 rows) = init + sum over KS K-steps of W-tile . X, split-bf16
template <int KS, bool FENCED = true, class Epi>
GRL_DEVINL void group16(const unsigned short* whi, const unsigned short* wlo, const bf16x8 (&xh)[KS], const bf16x8 (&xl)[KS], f32x4v acc,
                        Epi&& epi) {
  bf16x8 wh[KS], wl[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#ifdef GRL_E16_NOLDS
    wh[s] = xh[s];
    wl[s] = xl[s];
#else
    wh[s] = *reinterpret_cast<const bf16x8*>(whi + 32 * s);
    GRL_LO(wl[s] = *reinterpret_cast<const bf16x8*>(wlo + 32 * s);)
#endif
#ifdef GRL_E16_LDSONLY
    asm volatile("" ::"v"(wh[s]), "v"(wl[s]));
#endif
  }
  if (FENCED) __builtin_amdgcn_sched_barrier(0);
#if !defined(GRL_E16_NOMFMA) && !defined(GRL_E16_LDSONLY)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    acc = mfma16(wh[s], xh[s], acc);
    GRL_LO(acc = mfma16(wl[s], xh[s], acc);)
    GRL_LO(acc = mfma16(wh[s], xl[s], acc);)
  }
#endif
  epi(acc);
  if (FENCED) __builtin_amdgcn_sched_barrier(0);
}


// timing knock-outs (diagnostic builds only; results are wrong): -DGRL_E16_NOGELU, -DGRL_E16_NOMFMA, -DGRL_E16_NOGATHER
#ifdef GRL_E16_NOGELU
#define GELU16(x) (x)
#elif defined(GRL_E16_SCALAR_GELU)
GRL_DEVINL float4 gelu4s(float4 x) {
  float4 g, gp;
  gelu_both4(x, g, gp);
  return g;
}
#define GELU16(x) gelu4s(x)
#else
#define GELU16(x) gelu4(x)
#endif
// weight fragments of one chain group (n-tile nt of a 64 x 64 image), requested one group ahead of their MFMAs
struct WF2 {
  bf16x8 h[2], l[2];
  float4 bias;   // the group's accumulator start (requested with the fragments: a read issued inside the group's own region would make
                 // its wait cover the next group's fragment reads as well -- LDS returns in order)
};
GRL_DEVINL void wf_load(WF2& f, const unsigned short* whi, const unsigned short* wlo, const float* bias = nullptr) {
  f.bias = bias ? *reinterpret_cast<const float4*>(bias) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    f.h[s] = *reinterpret_cast<const bf16x8*>(whi + 32 * s);
    GRL_LO(f.l[s] = *reinterpret_cast<const bf16x8*>(wlo + 32 * s);)
  }
}
GRL_DEVINL f32x4v wf_mma(const WF2& f, const bf16x8 (&xh)[2], const bf16x8 (&xl)[2], f32x4v acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    acc = mfma16(f.h[s], xh[s], acc);
    GRL_LO(acc = mfma16(f.l[s], xh[s], acc);)
    GRL_LO(acc = mfma16(f.h[s], xl[s], acc);)
  }
  return acc;
}

// The same chain cut into scheduling regions like the fused backward's pass (GRL_E16_REGIONS): the fragments (and bias) of group k + 1
// are requested before group k's MFMAs, and region k runs the MFMAs of group k + 1 beside the epilogue (GELU) of group k -- matrix and
// vector work of the SAME wave, the only overlap a SIMD gives (DESIGN.md finding 18).  No LDS read sits between the MFMAs of a chain.
// Measured (tools/run_variants.sh, one box): 0.46-0.51 ms against 0.48-0.52 for the fenced groups -- inside the noise at three waves per
// SIMD; the default stays the fenced form (no fragment register is ever re-loaded while an MFMA that reads it may be queued).
#ifndef GRL_E16_REGIONS
#define GRL_E16_REGIONS 0
#endif
template <class KEpi>
GRL_DEVINL void chain16_regions(const ChainW16& w, float a, float b, int r, int g, KEpi&& k_epi) {
#define BAR() __builtin_amdgcn_sched_barrier(0)
  bf16x8 w1h[4], w1l[4];
  float4 b1q[4];
  WF2 wf[2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    w1h[nt] = *reinterpret_cast<const bf16x8*>(w.W1h + (16 * nt + r) * LD1 + 8 * g);
    GRL_LO(w1l[nt] = *reinterpret_cast<const bf16x8*>(w.W1l + (16 * nt + r) * LD1 + 8 * g);)
    b1q[nt] = *reinterpret_cast<const float4*>(w.b1s + 16 * nt + 4 * g);
  }
  wf_load(wf[0], w.W2h + r * LD2 + 8 * g, w.W2l + r * LD2 + 8 * g, w.b2s + 4 * g);
  const float aa = a * a, ab = a * b, bb = b * b;
  float4 phi;
  if (g == 0) phi = make_float4(a, b, aa, ab);
  else if (g == 1) phi = make_float4(ab, bb, aa * a, aa * b);
  else if (g == 2) phi = make_float4(ab * a, ab * b, ab * a, ab * b);
  else phi = make_float4(bb * a, bb * b, 0.f, 0.f);
  bf16x8 ph, pl;
  split_pair(phi, make_float4(0.f, 0.f, 0.f, 0.f), ph, pl);
  BAR();
  bf16x8 xh[2], xl[2], yh[2], yl[2];
  {
    float4 g1[4];
    f32x4v c[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      c[nt] = f32x4v{b1q[nt].x, b1q[nt].y, b1q[nt].z, b1q[nt].w};
      c[nt] = mfma16(w1h[nt], ph, c[nt]);
      GRL_LO(c[nt] = mfma16(w1l[nt], ph, c[nt]);)
      GRL_LO(c[nt] = mfma16(w1h[nt], pl, c[nt]);)
      if (nt > 0) g1[nt - 1] = GELU16(v4(c[nt - 1]));
      BAR();
    }
    g1[3] = GELU16(v4(c[3]));
    split_pair(g1[0], g1[1], xh[0], xl[0]);
    split_pair(g1[2], g1[3], xh[1], xl[1]);
    BAR();
  }
  auto layer64 = [&](int base, const unsigned short* mh, const unsigned short* ml, const unsigned short* nh, const unsigned short* nl,
                     const float* nbias, const bf16x8 (&ih)[2], const bf16x8 (&il)[2], auto&& epi, auto&& tail) {
    // groups of image (mh, ml); the group after the last one is tile 0 of (nh, nl) (nullptr: none)
    f32x4v c[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (nt < 3) wf_load(wf[(base + nt + 1) & 1], mh + (16 * (nt + 1) + r) * LD2 + 8 * g, ml + (16 * (nt + 1) + r) * LD2 + 8 * g,
                          base == 0 ? w.b2s + 16 * (nt + 1) + 4 * g : nullptr);
      else if (nh) wf_load(wf[(base + nt + 1) & 1], nh + r * LD2 + 8 * g, nl + r * LD2 + 8 * g, nbias);
      BAR();
      const float4 bq = wf[(base + nt) & 1].bias;
      c[nt] = wf_mma(wf[(base + nt) & 1], ih, il, f32x4v{bq.x, bq.y, bq.z, bq.w});
      if (nt > 0) epi(nt - 1, c[nt - 1]);
      BAR();
    }
    epi(3, c[3]);
    tail();
    BAR();
  };
  {
    float4 g2[4];
    layer64(0, w.W2h, w.W2l, w.Wkh, w.Wkl, nullptr, xh, xl, [&](int nt, const f32x4v& c) { g2[nt] = GELU16(v4(c)); },
            [&]() {
              split_pair(g2[0], g2[1], yh[0], yl[0]);
              split_pair(g2[2], g2[3], yh[1], yl[1]);
            });
  }
  layer64(4, w.Wkh, w.Wkl, nullptr, nullptr, nullptr, yh, yl, [&](int nt, const f32x4v& c) { k_epi(nt, v4(c)); }, [&]() {});
#undef BAR
}

// The chain for this lane's row: (a, b) -> K tiles handed to k_epi(nt, float4 of features 16 nt + 4 g + 0..3)
template <class KEpi>
GRL_DEVINL void chain16(const ChainW16& w, float a, float b, int r, int g, KEpi&& k_epi) {
  // polynomial features (ponita.py:233-244), this lane's four: f = 4 g .. 4 g + 3 of [a b | aa ab ba bb | aaa aab aba abb baa bab bba bbb]
  const float aa = a * a, ab = a * b, bb = b * b;
  float4 phi;
  if (g == 0) phi = make_float4(a, b, aa, ab);
  else if (g == 1) phi = make_float4(ab, bb, aa * a, aa * b);
  else if (g == 2) phi = make_float4(ab * a, ab * b, ab * a, ab * b);
  else phi = make_float4(bb * a, bb * b, 0.f, 0.f);
  bf16x8 ph[1], pl[1];
  split_pair(phi, make_float4(0.f, 0.f, 0.f, 0.f), ph[0], pl[0]);
  float4 g1[4], g2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 bq = *reinterpret_cast<const float4*>(w.b1s + 16 * nt + 4 * g);
    group16<1>(w.W1h + (16 * nt + r) * LD1 + 8 * g, w.W1l + (16 * nt + r) * LD1 + 8 * g, ph, pl, f32x4v{bq.x, bq.y, bq.z, bq.w},
               [&](const f32x4v& acc) { g1[nt] = GELU16(v4(acc)); });
  }
  bf16x8 xh[2], xl[2];
  split_pair(g1[0], g1[1], xh[0], xl[0]);
  split_pair(g1[2], g1[3], xh[1], xl[1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 bq = *reinterpret_cast<const float4*>(w.b2s + 16 * nt + 4 * g);
    group16<2>(w.W2h + (16 * nt + r) * LD2 + 8 * g, w.W2l + (16 * nt + r) * LD2 + 8 * g, xh, xl, f32x4v{bq.x, bq.y, bq.z, bq.w},
               [&](const f32x4v& acc) { g2[nt] = GELU16(v4(acc)); });
  }
  bf16x8 yh[2], yl[2];
  split_pair(g2[0], g2[1], yh[0], yl[0]);
  split_pair(g2[2], g2[3], yh[1], yl[1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
    group16<2>(w.Wkh + (16 * nt + r) * LD2 + 8 * g, w.Wkl + (16 * nt + r) * LD2 + 8 * g, yh, yl, f32x4v{0.f, 0.f, 0.f, 0.f},
               [&](const f32x4v& acc) { k_epi(nt, v4(acc)); });
}

#ifndef GRL_E16_RESIDENT
#define GRL_E16_RESIDENT GRL_PREC   // forward: weight fragments resident in registers (bf16 build: always; fp32 build: with GRL_E16_WGS=2, experiment)
#endif
#if GRL_E16_RESIDENT
// Resident weight fragments (round 5): a lane's weight fragments depend on (row, k-group, n-tile, K-step) only -- the SAME operand registers
// for every pass.  Plain-bf16 build: W2 8 + Wk 8 fragments = 64 registers at three waves per SIMD (W1's four would push the kernel over 168
// registers: they are requested with the biases at the head of the pass).  fp32 build: hi + lo = 128 registers, two waves per SIMD
// (GRL_E16_WGS=2).  The pass loop then has no fragment read inside the chain at all (the fenced groups exposed one LDS round trip per
// group: 12 per pass, wave cycles wait 0.41 + issue stall 0.23 at three waves per SIMD in the bf16 build, profiles/r05_pmc_table_rope_hepi_bf16_a.txt);
// the only LDS reads left are W1's fragments and the two layers' biases, requested at the head of the pass behind a scheduling barrier
// (no MFMA of this wave is in flight there: the previous pass's last accumulators have been consumed).  A layer's MFMAs run as four
// independent chains, round-robin.  Same products in the same order per accumulator: bit-identical results.
struct ChainRegs {
  bf16x8 w2[4][2], wk[4][2];
  GRL_LO(bf16x8 w2l[4][2]; bf16x8 wkl[4][2];)
};
GRL_DEVINL void chain_regs_load(ChainRegs& cr, const ChainW16& w, int r, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      cr.w2[nt][s] = *reinterpret_cast<const bf16x8*>(w.W2h + (16 * nt + r) * LD2 + 8 * g + 32 * s);
      cr.wk[nt][s] = *reinterpret_cast<const bf16x8*>(w.Wkh + (16 * nt + r) * LD2 + 8 * g + 32 * s);
      GRL_LO(cr.w2l[nt][s] = *reinterpret_cast<const bf16x8*>(w.W2l + (16 * nt + r) * LD2 + 8 * g + 32 * s);)
      GRL_LO(cr.wkl[nt][s] = *reinterpret_cast<const bf16x8*>(w.Wkl + (16 * nt + r) * LD2 + 8 * g + 32 * s);)
    }
  }
}
template <class KEpi>
GRL_DEVINL void chain16_resident(const ChainRegs& cr, const ChainW16& w, float a, float b, int r, int g, KEpi&& k_epi) {
  __builtin_amdgcn_sched_barrier(0);
  float4 b1q[4], b2q[4];
  bf16x8 w1[4];
  GRL_LO(bf16x8 w1l[4];)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    w1[nt] = *reinterpret_cast<const bf16x8*>(w.W1h + (16 * nt + r) * LD1 + 8 * g);
    GRL_LO(w1l[nt] = *reinterpret_cast<const bf16x8*>(w.W1l + (16 * nt + r) * LD1 + 8 * g);)
    b1q[nt] = *reinterpret_cast<const float4*>(w.b1s + 16 * nt + 4 * g);
    b2q[nt] = *reinterpret_cast<const float4*>(w.b2s + 16 * nt + 4 * g);
  }
  const float aa = a * a, ab = a * b, bb = b * b;
  float4 phi;
  if (g == 0) phi = make_float4(a, b, aa, ab);
  else if (g == 1) phi = make_float4(ab, bb, aa * a, aa * b);
  else if (g == 2) phi = make_float4(ab * a, ab * b, ab * a, ab * b);
  else phi = make_float4(bb * a, bb * b, 0.f, 0.f);
  bf16x8 ph, pl;
  split_pair(phi, make_float4(0.f, 0.f, 0.f, 0.f), ph, pl);
  f32x4v c[4];
  float4 g1[4], g2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w1[nt], ph, f32x4v{b1q[nt].x, b1q[nt].y, b1q[nt].z, b1q[nt].w});
#if !GRL_PREC
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w1l[nt], ph, c[nt]);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w1[nt], pl, c[nt]);
#endif
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) g1[nt] = GELU16(v4(c[nt]));
  bf16x8 xh[2], xl[2];
  split_pair(g1[0], g1[1], xh[0], xl[0]);
  split_pair(g1[2], g1[3], xh[1], xl[1]);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) c[nt] = f32x4v{b2q[nt].x, b2q[nt].y, b2q[nt].z, b2q[nt].w};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(cr.w2[nt][s], xh[s], c[nt]);
#if !GRL_PREC
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(cr.w2l[nt][s], xh[s], c[nt]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(cr.w2[nt][s], xl[s], c[nt]);
#endif
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) g2[nt] = GELU16(v4(c[nt]));
  bf16x8 yh[2], yl[2];
  split_pair(g2[0], g2[1], yh[0], yl[0]);
  split_pair(g2[2], g2[3], yh[1], yl[1]);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) c[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(cr.wk[nt][s], yh[s], c[nt]);
#if !GRL_PREC
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(cr.wkl[nt][s], yh[s], c[nt]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(cr.wk[nt][s], yl[s], c[nt]);
#endif
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) k_epi(nt, v4(c[nt]));
  __builtin_amdgcn_sched_barrier(0);
}
#define E16_CHAIN(S, A, B, R, G, EPI) chain16_resident(cregs, S, A, B, R, G, EPI)
#elif GRL_E16_REGIONS
#define E16_CHAIN chain16_regions
#else
#define E16_CHAIN chain16
#endif
constexpr int NPW_MAX = 16;   // anchor nodes per wave chunk (their rowptr entries live on the lanes)

// MODE 0: forward       out[anchor] = sum over its edges of K_e * x_in[other(e)]                    (anchor = destination)
// MODE 1: d x_src       out[anchor] = (dres ? dres[anchor] : 0) + sum of K_e * x_in[other(e) | row]   (anchor = source)
// MODE 2: messages      msg[edge position] = K_e * x_in[other(e)], no accumulation                    (anchor = destination)
template <int MODE>
__global__ __launch_bounds__(E16_THREADS, GRL_E16_WGS) void edge16_kernel(Edge16Params p, st_t* __restrict__ out, const st_t* __restrict__ dres) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  ChainW16& s = *reinterpret_cast<ChainW16*>(smem_raw);
  load_w16(s, p);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const float gx = s.grid_s[3 * r], gy = s.grid_s[3 * r + 1], gz = s.grid_s[3 * r + 2];
#if GRL_E16_RESIDENT
  ChainRegs cregs;
  chain_regs_load(cregs, s, r, g);
#endif
  const int* e_other = p.anchor_is_dst ? p.e_src : p.e_dst;
  const float* pos_anchor = p.anchor_is_dst ? p.pos_dst : p.pos_src;
  const float* pos_other = p.anchor_is_dst ? p.pos_src : p.pos_dst;
  // chunks of this wave: round-robin chunks of npw nodes, or (p.split) a contiguous node range with ~E / slots in-edges -- 13 108 chunks of
  // five nodes over 3 072 wave slots leave some waves five chunks and others four (75 passes against 60: the launch takes the 75)
  const int NPW = p.split ? NPW_MAX : p.npw;
  const int n_chunks = (p.n_anchor + NPW - 1) / NPW;
  const int slot = blockIdx.x * E16_WAVES + wave;
  const int n_lo = p.split ? p.split[slot] : 0, n_hi = p.split ? p.split[slot + 1] : p.n_anchor;
  for (int it = 0;; ++it) {
    int n0, nn;
    if (p.split) {
      n0 = n_lo + it * NPW_MAX;
      if (n0 >= n_hi) break;
      nn = min(NPW_MAX, n_hi - n0);
    } else {
      const int chunk = slot + it * (int)gridDim.x * E16_WAVES;
      if (chunk >= n_chunks) break;
      n0 = chunk * NPW;
      nn = min(NPW, p.n_anchor - n0);
    }
    // chunk metadata on the lanes: rowptr (lanes 0..nn) and the anchor nodes' positions (lanes 0..nn-1)
    const int rp = p.rowptr[n0 + min(lane, nn)];
    const int an = n0 + min(lane, nn - 1);
    const float pax = pos_anchor[3 * an], pay = pos_anchor[3 * an + 1], paz = pos_anchor[3 * an + 2];
    const int E0 = __builtin_amdgcn_readlane(rp, 0), E1 = __builtin_amdgcn_readlane(rp, nn);
    int node = 0;                         // index inside the chunk of the node being accumulated
    int node_end = __builtin_amdgcn_readlane(rp, 1);
    float4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&](int j) {             // rows of chunk node j are complete
      if (MODE == 2) return;
      st_t* o = out + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 v = acc[t];
        if (MODE == 1 && dres) v = f4_add(v, ld4(dres + ((size_t)(n0 + j) * O + r) * C + 4 * g + 16 * t));
        st4(o + 16 * t, v);
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    for (int eb = E0; eb < E1; eb += 64) {
      // metadata of the next (up to) 64 edges: the other end's node, its position, the row of x_in
      const int ee = min(eb + lane, E1 - 1);
      const int oth = e_other[ee];
      const int xrow = p.per_edge ? (p.erow ? p.erow[ee] : ee) : oth;
      const float pox = pos_other[3 * oth], poy = pos_other[3 * oth + 1], poz = pos_other[3 * oth + 2];
      const int nb = min(64, E1 - eb);
#pragma unroll 1
      for (int k = 0; k < nb; ++k) {
        const int e = eb + k;
        while (e >= node_end) {           // the edge belongs to a later node of the chunk: finish the nodes before it
          flush(node);
          ++node;
          node_end = __builtin_amdgcn_readlane(rp, node + 1);
        }
        const int row_in = __builtin_amdgcn_readlane(xrow, k);
        const st_t* xs = p.x_in + ((size_t)row_in * O + r) * C + 4 * g;
        float4 xv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#ifdef GRL_E16_NOGATHER
          xv[t] = make_float4(1.f, 1.f, 1.f, 1.f);
#else
          xv[t] = ld4(xs + 16 * t);      // this edge's gathered row, in flight behind the chain
#endif
        // r = pos_src - pos_dst (hepi.py:109-117), whichever end is the anchor
        float dx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pox), k)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pax), node));
        float dy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, poy), k)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pay), node));
        float dz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, poz), k)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, paz), node));
        if (!p.anchor_is_dst) { dx = -dx; dy = -dy; dz = -dz; }
        if (p.dim == 2) dz = 0.f;
        const float a = dx * gx + dy * gy + dz * gz;
        dx -= a * gx; dy -= a * gy; dz -= a * gz;
        const float b = sqrtf(dx * dx + dy * dy + dz * dz);
        st_t* mrow = MODE == 2 ? out + ((size_t)e * O + r) * C + 4 * g : nullptr;
        E16_CHAIN(s, a, b, r, g, [&](int nt, const float4& kq) {
          const float4 m = f4_mul(kq, xv[nt]);
          if (MODE == 2) st4(mrow + 16 * nt, m);
          else acc[nt] = f4_add(acc[nt], m);
        });
      }
    }
    for (; node < nn; ++node) flush(node);   // the last node with edges and every trailing node without
  }
}

// ---------------------------------------------------------------------------------------------------------------- fused backward
// ONE kernel for the whole edge backward (round 1 needed two, each rebuilding the basis-MLP chain; as one 32-row kernel it spilled
// 170-220 registers, DESIGN.md "what comes next" 1).  On 16-row tiles the chain state is half as wide and it fits:
//   walk: SOURCE-sorted edges, one edge per pass, a chunk of source nodes per wave (exactly the d x_src kernel above), one wave per SIMD
//   per pass: chain with derivatives (g1, g1', g2, g2'), K = Wk g2;  d x_src[src] += K * dM (registers, stored once per node);
//             dK = dM * x_src;  dWk += dK^T g2;  dZ2 = (Wk^T dK) * g2';  dW2 += dZ2^T g1;  db2;  dZ1 = (W2^T dZ2) * g1';  dW1 += dZ1^T phi;  db1
//   the products along the chain (contraction over features) are 16x16x32 MFMAs with the edge's 16 rows on the N side;
//   the weight-gradient products contract over the ROWS: v_mfma_f32_32x32x16_bf16 with K = the edge's 16 rows.  Their operands
//   need the rows in the registers and the feature on the lane -- the transpose of what the chain leaves -- so each operand is written
//   once to a wave-private LDS image [row][feature] (bf16 hi / lo, ds_write_b64) and read back with ds_read_b64_tr_b16 (the
//   hardware transpose read: cdna_hip_programming.md T10); no register transposes on the matrix pipe, no barrier (same wave).
//   Weight-gradient accumulators: 160 registers per wave for the whole launch (dWk 64, dW2 64, dW1 | db1 32), one partial row per
//   workgroup at the end in the layout of the 32-row kernel (edge_conv.hip EDGE_PARTIAL).
// (staging image layout, transposed fragment reads, pinned accumulators: grl_tile16.h)
struct Stage16 {
  unsigned short Ah[STG], Al[STG];   // A side: dK / dZ2 / dZ1
  unsigned short Bh[STG], Bl[STG];   // B side: g2 / g1 / phi
};
struct Bwd16Smem {
  Edge16Image img;   // chain images + the two transposes (grl_wimg.h)
  Stage16 st[4];
};
constexpr int BWD16_PARTIAL = 64 * 14 + 64 + 64 * 64 + 64 + 64 * 64;   // = EDGE_PARTIAL of edge_conv.hip

template <int NTK>
struct RFrags {
  bf16x8 ah[2], al[2], bh[NTK], bl[NTK];
};
// the transposed operand fragments of one weight-gradient product: requested here, consumed by rowred_mma a chain step later
template <int NTK>
GRL_DEVINL void rowred_load(const Stage16& st, int lane, RFrags<NTK>& f) {
#ifdef GRL_B16_NOROWMMA
  return;
#endif
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f.ah[t] = tr_frag(st.Ah, t, lane);
    GRL_LO(f.al[t] = tr_frag(st.Al, t, lane);)
  }
#pragma unroll
  for (int t = 0; t < NTK; ++t) {
    f.bh[t] = tr_frag(st.Bh, t, lane);
    GRL_LO(f.bl[t] = tr_frag(st.Bl, t, lane);)
  }
}
// acc[tn][tk] (32 x 32: output feature n on the registers, input feature k on the lane) += A^T B over the 16 staged rows
template <int NTK>
GRL_DEVINL void rowred_mma(const RFrags<NTK>& f, f32x16 (&acc)[2][NTK]) {
#ifdef GRL_B16_NOROWMMA
  return;
#endif
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int tk = 0; tk < NTK; ++tk) {
      mfma32_acc(f.ah[tn], f.bh[tk], acc[tn][tk]);
      GRL_LO(mfma32_acc(f.al[tn], f.bh[tk], acc[tn][tk]);)
      GRL_LO(mfma32_acc(f.ah[tn], f.bl[tk], acc[tn][tk]);)
    }
}

#ifdef GRL_B16_NOGELU
#define B16_GELU(x, gv, gpv) ((gv) = (x), (gpv) = (x))
#elif !defined(GRL_B16_SCALAR_GELU) || GRL_PREC || GRL_B16_BURST   // packed pairs: one wave per SIMD issues a v_pk_* in the time of a scalar op; the plain-bf16
                                                  // build always (its MFMAs come in bursts, not between the epilogue's instructions)
GRL_DEVINL void gelu_both4_pk(const float4& x, float4& gv, float4& gpv) {
  v2f g0, g1, d0, d1;
  gelu_pair<true>(v2f{x.x, x.y}, g0, d0);
  gelu_pair<true>(v2f{x.z, x.w}, g1, d1);
  gv = make_float4(g0.x, g0.y, g1.x, g1.y);
  gpv = make_float4(d0.x, d0.y, d1.x, d1.y);
}
#define B16_GELU(x, gv, gpv) gelu_both4_pk((x), (gv), (gpv))
#else
#define B16_GELU(x, gv, gpv) gelu_both4((x), (gv), (gpv))
#endif
#ifdef GRL_B16_NOGATHER
#define B16_LD(p) make_float4(0.5f, 0.25f, -0.5f, 1.f)
#else
#define B16_LD(p) ld4(p)
#endif
#ifdef GRL_B16_PHASE   // diagnostic build: s_memtime ticks per stage of the pass, wave 0 of every workgroup (tools/edge_phase.py --bwd16)
__device__ unsigned long long g_b16phase[16];
#define B16_PH(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[i] += t_ - tl_; tl_ = t_; } while (0)
#else
#define B16_PH(i)
#endif
struct Bwd16Params {
  Edge16Params e;         // the SOURCE-anchored view; e.x_in = dM rows (d x1 [Nd,16,64] or per-edge rows)
  const st_t* x_src;      // [Ns,16,64]
  const st_t* dres;       // optional [Ns,16,64]: gradient of the other use of x_src, added into d x_src
  st_t* dx_src;           // [Ns,16,64], fully written
  float* partial;         // [gridDim.x][BWD16_PARTIAL]
  const int* split;       // optional [4 gridDim.x + 1]: wave slot s walks the source nodes split[s] .. split[s + 1] (an edge-balanced partition
                          // built once per topology: ops.build_edge_set); NULL: chunks of npw nodes dealt round-robin
};

__global__ __launch_bounds__(256, 1) void edge_bwd16_kernel(Bwd16Params bp) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  Bwd16Smem& sm = *reinterpret_cast<Bwd16Smem*>(smem_raw);
  const Edge16Params& p = bp.e;
#ifndef GRL_KNOCK_STAGE
  if (p.wimg) {   // the five images of this step, built once by grl_weight_images: a linear copy (91 KB, 23 16-byte loads per thread)
    copy_image<256>(&sm.img, p.wimg, (int)sizeof(Edge16Image));
  } else {
    stage16<14, 32, 256>(sm.img.w.W1h, sm.img.w.W1l, p.W1, LD1);
    stage16<64, 64, 256>(sm.img.w.W2h, sm.img.w.W2l, p.W2, LD2);
    stage16<64, 64, 256>(sm.img.w.Wkh, sm.img.w.Wkl, p.Wk, LD2);
    stage16<64, 64, 256, true>(sm.img.WkTh, sm.img.WkTl, p.Wk, LD2);
    stage16<64, 64, 256, true>(sm.img.W2Th, sm.img.W2Tl, p.W2, LD2);
    stage_chain16_small<256>(sm.img.w, p.b1, p.b2, p.grid);
  }
#else
  stage_chain16_small<256>(sm.img.w, p.b1, p.b2, p.grid);
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  Stage16& st = sm.st[wave];
  for (int i = lane; i < STG / 2; i += 64) {
    reinterpret_cast<unsigned*>(st.Ah)[i] = 0u; reinterpret_cast<unsigned*>(st.Al)[i] = 0u;
    reinterpret_cast<unsigned*>(st.Bh)[i] = 0u; reinterpret_cast<unsigned*>(st.Bl)[i] = 0u;
  }
  __syncthreads();
  const ChainW16& w = sm.img.w;
  const float gx = w.grid_s[3 * r], gy = w.grid_s[3 * r + 1], gz = w.grid_s[3 * r + 2];
  const bool g0 = g == 0, g1_ = g == 1, g2_ = g == 2;

#if GRL_PREC
  // Plain-bf16 build (round 5): a 64-deep group is TWO MFMAs here (fp32 build: six), so a group's region is too short to hide the LDS
  // latency of the next group's fragments -- the K / dZ2 / dZ1 layers (four multiply-adds of epilogue per group) stood at s_waitcnt
  // lgkmcnt in front of every MFMA.  The four groups' fragments of a WHOLE layer (8 ds_read_b128 = 32 registers; no lo halves in this
  // build) are therefore requested one layer ahead into two alternating buffers:
  //   bufA: W2 (layer 2) -> Wk^T (dZ2) -> W2 of the NEXT pass;   bufB: Wk (K) -> W2^T (dZ1) -> Wk of the next pass
  // each refilled in the tail of the layer that has just consumed it, a full layer before its next use; inside a layer the eight MFMAs
  // run as four independent two-deep chains.  Measured against fragments two groups ahead in three rotating slots (24 registers): 2.66
  // vs 2.82 ms per step for the three launches of the rope workload.  Same products in the same order per accumulator.
  struct LayerFrags { bf16x8 h[4][2]; };
  auto lf_load = [&](LayerFrags& f, const unsigned short* img) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) f.h[nt][s_] = *reinterpret_cast<const bf16x8*>(img + (16 * nt + r) * LD2 + 8 * g + 32 * s_);
  };
  LayerFrags bufA, bufB;
  lf_load(bufA, w.W2h);
  lf_load(bufB, w.Wkh);
#elif GRL_B16_BURST
  // fp32 build (round 5; GRL_B16_BURST=0 restores the grouped form): a layer's 24 MFMAs as ONE burst of four independent six-deep chains from a layer-wide fragment
  // buffer (hi + lo: 64 registers, ONE buffer: it is refilled for the next layer right behind the burst -- in-order issue, one wave per
  // SIMD: every MFMA of the burst has been issued, i.e. has read its operands, before the first of these reads is issued -- and the
  // reads land during the epilogue), and the epilogues as PACKED f32 pairs behind it (no MFMA between their instructions: a lone wave issues a
  // v_pk_* in the time of a plain instruction, but not inside an MFMA's shadow).  Bitwise the grouped form's results (same products in the
  // same order per accumulator).  A/B on one box, three alternating rounds: 310.4-312.8 -> 312.2-313.7 steps/s (+0.5 %), edge_bwd16 1.19 ->
  // 1.17 ms per step (profiles/r05_ab_burst.txt): the packed epilogues save ~12 % of the pass's vector issue, the burst gives back the
  // little MFMA / VALU overlap the grouped form had (finding 18: the kernel is additive either way).
  struct LayerFrags { bf16x8 h[4][2], l[4][2]; };
  auto lf_load = [&](LayerFrags& f, const unsigned short* imh, const unsigned short* iml) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        f.h[nt][s_] = *reinterpret_cast<const bf16x8*>(imh + (16 * nt + r) * LD2 + 8 * g + 32 * s_);
        f.l[nt][s_] = *reinterpret_cast<const bf16x8*>(iml + (16 * nt + r) * LD2 + 8 * g + 32 * s_);
      }
  };
  LayerFrags buf;
  lf_load(buf, w.W2h, w.W2l);
#endif
  f32x16 accK[2][2], accA[2][2], accB[2][1];   // dWk, dW2, dW1 (| db1 in column 14)
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_) {
    accB[a_][0] = zero16();
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) { accK[a_][b_] = zero16(); accA[a_][b_] = zero16(); }
  }
  float4 db2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) db2[t] = make_float4(0.f, 0.f, 0.f, 0.f);

#ifdef GRL_B16_PHASE
  unsigned long long ph_[12] = {0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  // The chunks of this wave.  Round-robin chunks of npw nodes are balanced when every chunk carries the same number of edges (whole frames
  // of a large minibatch); for small graphs the OUT-degrees of a kNN graph vary (mean 3, up to ~10) and a wave with a few more edges than
  // the others sets the launch time (512-frame shard: 24 passes per wave on average, 37 for the slowest).  With bp.split every wave slot
  // gets a contiguous node range holding ~E / slots edges instead -- a function of the topology alone, so the partial rows (and with them
  // the weight gradients) stay bitwise reproducible, which a dynamic work queue would not.
  const int NPW = bp.split ? NPW_MAX : p.npw;
  const int slot = blockIdx.x * 4 + wave;
  const int n_lo = bp.split ? bp.split[slot] : 0, n_hi = bp.split ? bp.split[slot + 1] : p.n_anchor;
  const int n_chunks = (p.n_anchor + NPW - 1) / NPW;
#pragma unroll 1
  for (int it = 0;; ++it) {
    int n0, nn;
    if (bp.split) {
      n0 = n_lo + it * NPW_MAX;
      if (n0 >= n_hi) break;
      nn = min(NPW_MAX, n_hi - n0);
    } else {
      const int chunk = slot + it * (int)gridDim.x * 4;
      if (chunk >= n_chunks) break;
      n0 = chunk * NPW;
      nn = min(NPW, p.n_anchor - n0);
    }
    const int rp = p.rowptr[n0 + min(lane, nn)];
    const int an = n0 + min(lane, nn - 1);
    const float pax = p.pos_src[3 * an], pay = p.pos_src[3 * an + 1], paz = p.pos_src[3 * an + 2];
    const int E0 = __builtin_amdgcn_readlane(rp, 0), E1 = __builtin_amdgcn_readlane(rp, nn);
    int node = 0;
    int node_end = __builtin_amdgcn_readlane(rp, 1);
    float4 acc[4], xv[4];
    // `dres ? load : 0` written on the load EXPRESSION compiles to a branch around every load with an s_waitcnt vmcnt(0) at its end (the
    // value merges with a constant): four serialised memory round trips per node change -- 22 % of the bf16 build's pass, 7 % of the fp32
    // build's (profiles/r05_edge_bwd16_phases_*.txt).  The loads are therefore unconditional -- from x_src when there is no dres: the rows
    // the same node loads anyway, so no extra traffic where it matters -- and the SELECT is on the loaded value.
    const st_t* dres_b = bp.dres ? bp.dres : bp.x_src;
    const unsigned dmask = bp.dres ? 0xFFFFFFFFu : 0u;   // the select as a bit mask (a ?: on a wave-uniform bool becomes branches again)
    auto keep = [&](const float4& v) {
      return make_float4(__uint_as_float(__float_as_uint(v.x) & dmask), __uint_as_float(__float_as_uint(v.y) & dmask),
                         __uint_as_float(__float_as_uint(v.z) & dmask), __uint_as_float(__float_as_uint(v.w) & dmask));
    };
#ifndef GRL_B16_ROWS
#define GRL_B16_ROWS 0   // plain-bf16 build, how the source node's own rows are fetched: 0 = at the node change, widened at the first consumer;
#endif                   // 1 = "uniform passes" (below).  A/B on one box (tools/r05_ab_rows.sh): 0 wins, see the comment at the #else
#if GRL_PREC && GRL_B16_ROWS
    // Plain-bf16 build ("uniform passes", round 5).  The rows arrive as bf16 bits and are widened in registers, a ~2.6 us pass is shorter
    // than a row gather under load, and s_waitcnt vmcnt counts loads AND stores in issue order: any load or store whose presence depends
    // on the pass (a node change's row loads) makes the compiler wait for the smallest count over all paths, i.e. for operations that have
    // only just been issued (phase stamps: 22 % of the pass at the node change, then 41 % at the first consumer when the widening was
    // merely deferred; knock-out without gathers: -20 % of the launch).  Here EVERY pass issues the same twelve loads at its top -- for the
    // NEXT edge: its dM row, the x_src row and the dres row of its source node (the node's rows again for each of its edges: L1 / L2 hits)
    // -- and takes last pass's twelve over right before: one vmcnt(0) per pass, at a point where everything outstanding is a full pass
    // old.  No row load is tied to a node change; the dres row stays raw (aq) and is added by flush(), the accumulator starts at zero.
    struct RawRows { uint2 d[4], x[4], a[4]; };
    RawRows nq;
    uint2 aq[4];
    auto widen = [](const uint2& u) {
      return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
    };
    auto flush = [&](int j) {
      st_t* o = bp.dx_src + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) st4(o + 16 * t, f4_add(acc[t], widen(make_uint2(aq[t].x & dmask, aq[t].y & dmask))));
    };
#elif GRL_PREC
    // Plain-bf16 build, rows at the node change (the form that measured best: 2.66 ms per step for the rope workload's three launches
    // against 2.82-2.9 for the uniform passes above and for rows requested a node ahead -- the extra loads, copies and per-pass widenings
    // cost more than the residual wait): the raw rows are kept as loaded (xq, aq) and widened by materialise() in front of the K layer of
    // the node's first pass; a widening right behind the load would be a use that waits for it (node_begin stood a memory round trip at
    // every node change: 22 % of the pass).  dM rows are gathered two edges ahead (dq1, dq2).
    uint2 xq[4], aq[4];
    bool fresh = false;
    auto widen = [](const uint2& u) {
      return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
    };
    auto node_begin = [&](int j) {
      const st_t* xs = bp.x_src + ((size_t)(n0 + j) * O + r) * C + 4 * g;
      const st_t* ds = dres_b + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#ifdef GRL_B16_NOGATHER
        xq[t] = aq[t] = make_uint2(0x3f003e80u, 0xbf003f80u);
#else
        xq[t] = *reinterpret_cast<const uint2*>(xs + 16 * t);
        aq[t] = *reinterpret_cast<const uint2*>(ds + 16 * t);
#endif
      }
      fresh = true;
    };
    auto materialise = [&]() {
      if (fresh) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          xv[t] = widen(xq[t]);
          acc[t] = widen(make_uint2(aq[t].x & dmask, aq[t].y & dmask));
        }
        fresh = false;
      }
    };
    auto flush = [&](int j) {
      materialise();
      st_t* o = bp.dx_src + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) st4(o + 16 * t, acc[t]);
    };
#else
#ifndef GRL_B16_FP32_ROWS
#define GRL_B16_FP32_ROWS 0   // fp32 build: 0 = round 4's node rows (dres ? load : 0 on the load expression, node change in front of the dM take-over),
#endif                        // 1 = unconditional loads + bit-mask select + take-over first (what the bf16 build needs).  See the A/B note below.
    auto node_begin = [&](int j) {   // the source node's own row: the same for all of its edges
      const st_t* xs = bp.x_src + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) {   // the accumulator starts from the other branch's gradient row (dres), long before it is needed
        xv[t] = B16_LD(xs + 16 * t);
#if GRL_B16_FP32_ROWS
        const float4 dr = B16_LD(dres_b + ((size_t)(n0 + j) * O + r) * C + 4 * g + 16 * t);   // UNCONDITIONAL load, then a select
        acc[t] = keep(dr);
#else
        acc[t] = bp.dres ? B16_LD(bp.dres + ((size_t)(n0 + j) * O + r) * C + 4 * g + 16 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
      }
    };
    auto flush = [&](int j) {
      st_t* o = bp.dx_src + ((size_t)(n0 + j) * O + r) * C + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) st4(o + 16 * t, acc[t]);
    };
#endif
    // nodes [from, to) have no out-edges (padded points come in runs): their rows are dres or zero.  Four nodes per round trip --
    // a load -> store per node would expose an HBM latency each (indices are clamped, not branched on: duplicates are harmless)
    auto skip_empty_run = [&](int from, int to) {
#pragma unroll 1
      for (int j = from; j < to; j += 4) {
#if GRL_PREC
        uint2 v[4][4];   // (bf16 rows are copied as they are: no widening / rounding round trip)
#else
        float4 v[4][4];
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t row = ((size_t)(n0 + min(j + u, to - 1)) * O + r) * C + 4 * g;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
#if GRL_PREC
            const uint2 dr = *reinterpret_cast<const uint2*>(dres_b + row + 16 * t);
            v[u][t] = make_uint2(dr.x & dmask, dr.y & dmask);
#else
#if GRL_B16_FP32_ROWS
            const float4 dr = B16_LD(dres_b + row + 16 * t);
            v[u][t] = keep(dr);
#else
            v[u][t] = bp.dres ? B16_LD(bp.dres + row + 16 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
#endif
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t row = ((size_t)(n0 + min(j + u, to - 1)) * O + r) * C + 4 * g;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#if GRL_PREC
            *reinterpret_cast<uint2*>(bp.dx_src + row + 16 * t) = v[u][t];
#else
            st4(bp.dx_src + row + 16 * t, v[u][t]);
#endif
        }
      }
    };
#if GRL_PREC && GRL_B16_ROWS
    if (E0 < E1) {   // the leading nodes without edges are copied; the first node WITH edges becomes the current one
      while (__builtin_amdgcn_readlane(rp, node + 1) <= E0) ++node;
      skip_empty_run(0, node);
      node_end = __builtin_amdgcn_readlane(rp, node + 1);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    int pnode = node, pnode_end = node_end;   // the node of the edge whose rows are requested next (runs one edge ahead of `node`)
#else
    node_begin(0);
#endif
    for (int eb = E0; eb < E1; eb += 64) {
      const int ee = min(eb + lane, E1 - 1);
      const int oth = p.e_dst[ee];
      const int xrow = p.per_edge ? (p.erow ? p.erow[ee] : ee) : oth;
      const float pox = p.pos_dst[3 * oth], poy = p.pos_dst[3 * oth + 1], poz = p.pos_dst[3 * oth + 2];
      const int nb = min(64, E1 - eb);
      // dM rows are gathered AHEAD of their pass (a single wave per SIMD cannot hide an HBM round trip behind another wave): one edge ahead
      // in the fp32 build (a pass is ~4.3 us), TWO edges ahead in the plain-bf16 build, whose ~2.6 us pass is shorter than a gather under
      // load -- the one-ahead form stood ~0.5 us per pass at the loop tail's vmcnt (profiles/r05_edge_bwd16_phases_rope_bf16.txt: 22 % of
      // the pass; knock-out without gathers: -20 % of the launch).  The prefetched row is TAKEN OVER BEFORE the node change below: vmcnt
      // counts in issue order, and with the node change's stores and loads issued in front of the take-over the wait for the old gather
      // would also wait for the row loads that have only just been requested.
#if GRL_PREC && GRL_B16_ROWS
      auto rows_issue = [&](int kk) {   // the twelve loads of edge eb + kk (pnode: its source node)
        const int en = eb + kk;
        while (en >= pnode_end) { ++pnode; pnode_end = __builtin_amdgcn_readlane(rp, pnode + 1); }
        const int row_in = __builtin_amdgcn_readlane(xrow, kk);
        const st_t* dm = p.x_in + ((size_t)row_in * O + r) * C + 4 * g;
        const st_t* xs = bp.x_src + ((size_t)(n0 + pnode) * O + r) * C + 4 * g;
        const st_t* ds = dres_b + ((size_t)(n0 + pnode) * O + r) * C + 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#ifdef GRL_B16_NOGATHER
          nq.d[t] = nq.x[t] = nq.a[t] = make_uint2(0x3f003e80u, 0xbf003f80u);
#else
          nq.d[t] = *reinterpret_cast<const uint2*>(dm + 16 * t);
          nq.x[t] = *reinterpret_cast<const uint2*>(xs + 16 * t);
          nq.a[t] = *reinterpret_cast<const uint2*>(ds + 16 * t);
#endif
        }
      };
      rows_issue(0);
#elif GRL_PREC
      uint2 dq1[4], dq2[4];   // raw bf16 dM rows of the next two edges (widened when taken over)
      auto dq_issue = [&](uint2 (&q)[4], int kk) {
        const int row_in = __builtin_amdgcn_readlane(xrow, kk);
        const st_t* dm = p.x_in + ((size_t)row_in * O + r) * C + 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#ifdef GRL_B16_NOGATHER
          q[t] = make_uint2(0x3f003e80u, 0xbf003f80u);
#else
          q[t] = *reinterpret_cast<const uint2*>(dm + 16 * t);
#endif
      };
      dq_issue(dq1, 0);
      dq_issue(dq2, min(1, nb - 1));
#else
      float4 dvn[4];
      auto dv_issue = [&](int kk) {
        const int row_in = __builtin_amdgcn_readlane(xrow, kk);
        const st_t* dm = p.x_in + ((size_t)row_in * O + r) * C + 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t) dvn[t] = B16_LD(dm + 16 * t);
      };
      dv_issue(0);
#endif
#pragma unroll 1
      for (int k = 0; k < nb; ++k) {
        const int e = eb + k;
        B16_PH(7);   // (diagnostic build) loop tail of the previous pass
        float4 dv[4];
#if GRL_PREC && GRL_B16_ROWS
        // take-over of the twelve rows requested by the previous pass, PINNED here (the empty asm statements are uses the compiler
        // cannot sink): the one vmcnt wait of the pass, with nothing younger than a pass in flight
        uint2 dq[4], xq[4], an[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dq[t] = nq.d[t]; xq[t] = nq.x[t]; an[t] = nq.a[t];
          asm volatile("" : "+v"(dq[t].x), "+v"(dq[t].y), "+v"(xq[t].x), "+v"(xq[t].y), "+v"(an[t].x), "+v"(an[t].y));
        }
#elif GRL_PREC
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // taken over BEFORE the node change below (vmcnt counts in issue order)
          dv[t] = widen(dq1[t]);
          dq1[t] = dq2[t];
        }
#elif GRL_B16_FP32_ROWS
#pragma unroll
        for (int t = 0; t < 4; ++t) dv[t] = dvn[t];
#endif
        B16_PH(8);   // this edge's prefetched rows taken over (vmcnt wait)
        if (e >= node_end) {            // the edge belongs to a later node of the chunk
          flush(node);
          int m = node + 1;
          while (__builtin_amdgcn_readlane(rp, m + 1) <= e) ++m;
          skip_empty_run(node + 1, m);
          node = m;
          node_end = __builtin_amdgcn_readlane(rp, m + 1);
#if GRL_PREC && GRL_B16_ROWS
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
          node_begin(m);
#endif
        }
#if GRL_PREC && GRL_B16_ROWS
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // (behind the flush, which adds the OLD node's dres row)
          aq[t] = an[t];
          dv[t] = widen(dq[t]);
          xv[t] = widen(xq[t]);
        }
        rows_issue(min(k + 1, nb - 1));
#elif GRL_PREC
        dq_issue(dq2, min(k + 2, nb - 1));
#else
#if !GRL_B16_FP32_ROWS
#pragma unroll
        for (int t = 0; t < 4; ++t) dv[t] = dvn[t];
#endif
        dv_issue(min(k + 1, nb - 1));
#endif
        B16_PH(10);  // node change (flush, skip of empty nodes, node_begin), next gather issued
        // layer 1's weight fragments and biases, and the first 64-deep group: requested before the invariants are computed
        bf16x8 w1h[4], w1l[4];
        float4 b1q[4];
        WF2 wf[2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          w1h[nt] = *reinterpret_cast<const bf16x8*>(w.W1h + (16 * nt + r) * LD1 + 8 * g);
          GRL_LO(w1l[nt] = *reinterpret_cast<const bf16x8*>(w.W1l + (16 * nt + r) * LD1 + 8 * g);)
          b1q[nt] = *reinterpret_cast<const float4*>(w.b1s + 16 * nt + 4 * g);
        }
#if GRL_PREC || GRL_B16_BURST
        float4 b2q[4];   // layer 2's biases (its fragments are resident in bufA)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b2q[nt] = *reinterpret_cast<const float4*>(w.b2s + 16 * nt + 4 * g);
#else
        wf_load(wf[0], w.W2h + r * LD2 + 8 * g, w.W2l + r * LD2 + 8 * g, w.b2s + 4 * g);
#endif
        // dW1 (| db1) of the PREVIOUS pass: its operands were staged at that pass's end (zero tiles before a wave's first pass); the six
        // MFMAs run beside this pass's first GELU instead of standing alone behind an exposed LDS round trip
        RFrags<1> rf1;
        rowred_load<1>(st, lane, rf1);
        __builtin_amdgcn_sched_barrier(0);
        B16_PH(9);   // layer 1's fragments / biases and the transposed dW1 operands requested (the stamp itself waits for them: LDS latency)
        // rel = pos_src - pos_dst (hepi.py:109-117); the source is the anchor here
        float dx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pax), node)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pox), k));
        float dy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pay), node)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, poy), k));
        float dz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, paz), node)) -
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, poz), k));
        if (p.dim == 2) dz = 0.f;
        const float a = dx * gx + dy * gy + dz * gz;
        dx -= a * gx; dy -= a * gy; dz -= a * gz;
        const float b = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);   // 1 ulp; the IEEE expansion costs 15 instructions
        // ---- chain with derivatives
        // polynomial features 4 g .. 4 g + 3 of this lane (ponita.py:233-244), by selects; column 14 carries a one: the dW1 product then
        // leaves db1 in column 14 (W1's image has zeros there)
        const float aa = a * a, ab = a * b, bb = b * b, aba = ab * a, abb = ab * b;   // (products grouped as the reference's outer products)
        float4 phi;
        phi.x = g0 ? a : g1_ ? ab : g2_ ? aba : bb * a;
        phi.y = g0 ? b : g1_ ? bb : g2_ ? abb : bb * b;
        phi.z = g0 ? aa : g1_ ? aa * a : g2_ ? aba : 1.f;
        phi.w = g0 ? ab : g1_ ? aa * b : g2_ ? abb : 0.f;
        bf16x8 ph[1], pl[1];
        split_pair(phi, make_float4(0.f, 0.f, 0.f, 0.f), ph[0], pl[0]);
        B16_PH(0);   // pass top: node change, gathers issued, invariants, phi
        float4 gp1[4], gp2[4];
        bf16x8 xh[2], xl[2], yh[2], yl[2];
        // One wave per SIMD: nobody else hides an LDS round trip, and the compiler places a fragment load next to its MFMA.  The
        // pass is therefore cut into scheduling regions (BAR) by hand: region k issues the weight fragments of group k + 2, runs the
        // MFMAs of group k + 1 and the epilogue (GELU / products) of group k -- loads a group ahead, matrix pipe beside the vector work.
        // The sixteen 64-deep groups, in order: W2 (4), Wk (4), Wk^T (4), W2^T (4).
#define BAR() __builtin_amdgcn_sched_barrier(0)
        auto wptr_h = [&](int i) { const unsigned short* m = i < 4 ? w.W2h : i < 8 ? w.Wkh : i < 12 ? sm.img.WkTh : sm.img.W2Th; return m + (16 * (i & 3) + r) * LD2 + 8 * g; };
        auto wptr_l = [&](int i) { const unsigned short* m = i < 4 ? w.W2l : i < 8 ? w.Wkl : i < 12 ? sm.img.WkTl : sm.img.W2Tl; return m + (16 * (i & 3) + r) * LD2 + 8 * g; };
#if GRL_PREC
        // a 64-deep layer from a resident buffer: four independent two-deep MFMA chains, then the epilogues, then the tail (which refills
        // the buffer for its next use -- behind a scheduling barrier, so every MFMA that reads the buffer has delivered its result first)
        auto layer64 = [&](int base, const bf16x8 (&ih)[2], const bf16x8 (&il)[2], auto&& epi, auto&& tail) {
          (void)il;
          LayerFrags& f = (base == 0 || base == 8) ? bufA : bufB;
          f32x4v c[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[nt] = base == 0 ? f32x4v{b2q[nt].x, b2q[nt].y, b2q[nt].z, b2q[nt].w} : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(f.h[nt][s_], ih[s_], c[nt]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) epi(nt, c[nt]);
          BAR();
          lf_load(f, base == 0 ? sm.img.WkTh : base == 4 ? sm.img.W2Th : base == 8 ? w.W2h : w.Wkh);
          tail();
          BAR();
        };
#elif GRL_B16_BURST
        auto layer64 = [&](int base, const bf16x8 (&ih)[2], const bf16x8 (&il)[2], auto&& epi, auto&& tail) {
          f32x4v c[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[nt] = base == 0 ? f32x4v{b2q[nt].x, b2q[nt].y, b2q[nt].z, b2q[nt].w} : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_) {   // (per accumulator the same six products in the same order as the grouped form: bitwise equal)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(buf.h[nt][s_], ih[s_], c[nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(buf.l[nt][s_], ih[s_], c[nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(buf.h[nt][s_], il[s_], c[nt]);
          }
          BAR();
          if (base == 0) lf_load(buf, w.Wkh, w.Wkl);
          else if (base == 4) lf_load(buf, sm.img.WkTh, sm.img.WkTl);
          else if (base == 8) lf_load(buf, sm.img.W2Th, sm.img.W2Tl);
          else lf_load(buf, w.W2h, w.W2l);
          BAR();
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) epi(nt, c[nt]);
          tail();
          BAR();
        };
#else
        // a 64-deep layer: groups base .. base + 3 (wf[base & 1] already requested); epi(nt, c) consumes tile nt one region later
        auto layer64 = [&](int base, const bf16x8 (&ih)[2], const bf16x8 (&il)[2], auto&& epi, auto&& tail) {
          f32x4v c[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            if (base + nt + 1 < 16)
              wf_load(wf[(nt + 1) & 1], wptr_h(base + nt + 1), wptr_l(base + nt + 1), base + nt + 1 < 4 ? w.b2s + 16 * (nt + 1) + 4 * g : nullptr);
            BAR();
            const float4 bq = wf[nt & 1].bias;
            c[nt] = wf_mma(wf[nt & 1], ih, il, f32x4v{bq.x, bq.y, bq.z, bq.w});
            if (nt > 0) epi(nt - 1, c[nt - 1]);
            BAR();
          }
          epi(3, c[3]);
          tail();
          BAR();
        };
#endif
#if GRL_B16_BURST && !GRL_PREC
        {
          float4 g1[4];
          BAR();
          f32x4v c[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[nt] = f32x4v{b1q[nt].x, b1q[nt].y, b1q[nt].z, b1q[nt].w};
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w1h[nt], ph[0], c[nt]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w1l[nt], ph[0], c[nt]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[nt] = mfma16(w1h[nt], pl[0], c[nt]);
          rowred_mma<1>(rf1, accB);
          BAR();
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) B16_GELU(v4(c[nt]), g1[nt], gp1[nt]);
          split_pair(g1[0], g1[1], xh[0], xl[0]);
          split_pair(g1[2], g1[3], xh[1], xl[1]);
          BAR();
        }
#else
        {
          float4 g1[4];
          BAR();
          f32x4v c[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            c[nt] = f32x4v{b1q[nt].x, b1q[nt].y, b1q[nt].z, b1q[nt].w};
            c[nt] = mfma16(w1h[nt], ph[0], c[nt]);
            GRL_LO(c[nt] = mfma16(w1l[nt], ph[0], c[nt]);)
            GRL_LO(c[nt] = mfma16(w1h[nt], pl[0], c[nt]);)
            if (nt > 0) B16_GELU(v4(c[nt - 1]), g1[nt - 1], gp1[nt - 1]);
            BAR();
          }
          B16_GELU(v4(c[3]), g1[3], gp1[3]);
          rowred_mma<1>(rf1, accB);
          split_pair(g1[0], g1[1], xh[0], xl[0]);
          split_pair(g1[2], g1[3], xh[1], xl[1]);
          BAR();
        }
#endif
        B16_PH(1);   // layer 1 (12 MFMA, GELU + derivative, split)
        {
          float4 g2[4];
          layer64(0, xh, xl, [&](int nt, const f32x4v& c) { B16_GELU(v4(c), g2[nt], gp2[nt]); },
                  [&]() {
                    split_pair(g2[0], g2[1], yh[0], yl[0]);
                    split_pair(g2[2], g2[3], yh[1], yl[1]);
                  });
        }
        B16_PH(2);   // layer 2 (24 MFMA, GELU + derivative, split)
#if GRL_PREC && !GRL_B16_ROWS
        materialise();   // the node's own rows, requested at the node change, are widened here -- first use below
#endif
        // ---- K = Wk g2: d x_src row += K * dM;  dK = dM * x_src, staged with g2 for dWk += dK^T g2 (consumed after the dZ2 groups)
        bf16x8 kh[2], kl[2];
        RFrags<2> rf;
        layer64(4, yh, yl, [&](int nt, const f32x4v& c) { acc[nt] = f4_add(acc[nt], f4_mul(v4(c), dv[nt])); },
                [&]() {
                  split_pair(f4_mul(dv[0], xv[0]), f4_mul(dv[1], xv[1]), kh[0], kl[0]);
                  split_pair(f4_mul(dv[2], xv[2]), f4_mul(dv[3], xv[3]), kh[1], kl[1]);
                  stage_put<2>(st.Ah, st.Al, kh, kl, r, g);
                  stage_put<2>(st.Bh, st.Bl, yh, yl, r, g);
                  rowred_load<2>(st, lane, rf);
                });
        B16_PH(3);   // K (24 MFMA), d x_src, dK, staging dK | g2, transposed reads issued
        // ---- dZ2 = (Wk^T dK) * gelu'(z2);  db2;  then dWk's MFMAs beside the split of dZ2
        bf16x8 zh[2], zl[2];
        {
          float4 dz2[4];
          layer64(8, kh, kl,
                  [&](int nt, const f32x4v& c) {
                    dz2[nt] = f4_mul(v4(c), gp2[nt]);
                    db2[nt] = f4_add(db2[nt], dz2[nt]);
                  },
                  [&]() {
                    rowred_mma<2>(rf, accK);
                    split_pair(dz2[0], dz2[1], zh[0], zl[0]);
                    split_pair(dz2[2], dz2[3], zh[1], zl[1]);
                  });
        }
        B16_PH(4);   // dZ2 (24 MFMA), dWk (12 MFMA 32x32), split dZ2
        // ---- dW2 += dZ2^T g1 (fragments requested now, product after the dZ1 groups)
        stage_put<2>(st.Ah, st.Al, zh, zl, r, g);
        stage_put<2>(st.Bh, st.Bl, xh, xl, r, g);
        rowred_load<2>(st, lane, rf);
        // ---- dZ1 = (W2^T dZ2) * gelu'(z1)
        bf16x8 uh[2], ul[2];
        {
          float4 dz1[4];
          layer64(12, zh, zl, [&](int nt, const f32x4v& c) { dz1[nt] = f4_mul(v4(c), gp1[nt]); },
                  [&]() {
                    rowred_mma<2>(rf, accA);
                    split_pair(dz1[0], dz1[1], uh[0], ul[0]);
                    split_pair(dz1[2], dz1[3], uh[1], ul[1]);
                  });
        }
        B16_PH(5);   // staging dZ2 | g1, dZ1 (24 MFMA), dW2 (12 MFMA 32x32), split dZ1
        // ---- dW1 (| db1) += dZ1^T (phi | 1)
        stage_put<2>(st.Ah, st.Al, uh, ul, r, g);
        {   // phi image: features 4 g .. 4 g + 3 of columns 0..15; columns 16..31 of the tile are zeroed (g1 was there)
          const u32x4 h4 = __builtin_bit_cast(u32x4, ph[0]);
          *reinterpret_cast<uint2*>(st.Bh + stg_off(r, g)) = make_uint2(h4[0], h4[1]);
          *reinterpret_cast<uint2*>(st.Bh + stg_off(r, 4 + g)) = make_uint2(0u, 0u);
#if !GRL_PREC
          const u32x4 l4 = __builtin_bit_cast(u32x4, pl[0]);
          *reinterpret_cast<uint2*>(st.Bl + stg_off(r, g)) = make_uint2(l4[0], l4[1]);
          *reinterpret_cast<uint2*>(st.Bl + stg_off(r, 4 + g)) = make_uint2(0u, 0u);
#endif
        }
        B16_PH(6);   // staging dZ1 | phi (their product runs in the next pass)
#undef BAR
      }
    }
#if GRL_PREC && GRL_B16_ROWS
    if (E0 < E1) { flush(node); skip_empty_run(node + 1, nn); }   // the last node with edges, then the trailing nodes without
    else skip_empty_run(0, nn);
#else
    flush(node);                  // the last node with edges (or node 0 of a chunk without any), then the trailing nodes without
    skip_empty_run(node + 1, nn);
#endif
  }

  {   // dW1 of the wave's last pass
    RFrags<1> rf1;
    rowred_load<1>(st, lane, rf1);
    rowred_mma<1>(rf1, accB);
  }
#ifdef GRL_B16_PHASE
  if (lane == 0 && wave == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(&g_b16phase[i], ph_[i]);
#endif
  // ---- fold the four waves through LDS (images dead): 1 -> 0 and 3 -> 2, then 2 -> 0; fixed order, one partial row per workgroup
  constexpr int NACC = 10 * 16 + 16;
  float* fold = smem_raw;
  // the last asm MFMAs have written the accumulators before they are read (every tile is an operand of the drain: nothing moves above it)
  acc_drain(accB[0][0], accB[1][0], accA[0][0], accA[0][1], accA[1][0], accA[1][1], accK[0][0], accK[0][1], accK[1][0], accK[1][1]);
  auto visit = [&](auto&& f) {
    int k = 0;
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_) {
#pragma unroll
      for (int i = 0; i < 16; ++i) accB[a_][0][i] = f(accB[a_][0][i], k++);
#pragma unroll
      for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accA[a_][b_][i] = f(accA[a_][b_][i], k++); accK[a_][b_][i] = f(accK[a_][b_][i], k++); }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      db2[t].x = f(db2[t].x, k++); db2[t].y = f(db2[t].y, k++); db2[t].z = f(db2[t].z, k++); db2[t].w = f(db2[t].w, k++);
    }
  };
  __syncthreads();
  if (wave & 1) visit([&](float v_, int k) { fold[((wave >> 1) * NACC + k) * 64 + lane] = v_; return v_; });
  __syncthreads();
  if (!(wave & 1)) visit([&](float v_, int k) { return v_ + fold[((wave >> 1) * NACC + k) * 64 + lane]; });
  __syncthreads();
  if (wave == 2) visit([&](float v_, int k) { fold[k * 64 + lane] = v_; return v_; });
  __syncthreads();
  if (wave != 0) return;
  visit([&](float v_, int k) { return v_ + fold[k * 64 + lane]; });
  // 32x32 accumulator element rho of lane (column j = lane & 31, h = lane >> 5): D[n = 8 (rho >> 2) + 4 h + (rho & 3)][j]
  const int j = lane & 31, hh = lane >> 5;
  float* out = bp.partial + (size_t)blockIdx.x * BWD16_PARTIAL;
  float* oW1 = out, *ob1 = out + 64 * 14, *oW2 = ob1 + 64, *ob2 = oW2 + 64 * 64, *oWk = ob2 + 64;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) {
      const int n = 32 * nt + (rho & 3) + 8 * (rho >> 2) + 4 * hh;
      if (j < 14) oW1[n * 14 + j] = accB[nt][0][rho];
      if (j == 14) ob1[n] = accB[nt][0][rho];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        oW2[n * 64 + 32 * kt + j] = accA[nt][kt][rho];
        oWk[n * 64 + 32 * kt + j] = accK[nt][kt][rho];
      }
    }
  // db2: this lane's row sums of features 16 t + 4 g + u -> over the 16 rows (lanes sharing g)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float v[4] = {db2[t].x, db2[t].y, db2[t].z, db2[t].w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float x = v[u];
      x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
      if (r == 0) ob2[16 * t + 4 * g + u] = x;
    }
  }
}

}  // namespace

extern "C" {

#if !GRL_PREC   // launch-shape queries (host-side partitioning): the constants live here, ops.build_edge_set asks instead of copying them
static_assert(E16_WAVES == 4, "grl_edge_fwd_slots (edge_conv.hip) counts four wave slots per workgroup");
int grl_edge_fwd_chunk_nodes(int n_dst) {   // nodes per round-robin chunk of the forward; n_dst < 0: the grid cap in workgroups
  if (n_dst < 0) return 256 * GRL_E16_WGS;
  const int npw = n_dst / (4 * 256 * E16_WAVES * GRL_E16_WGS);
  return npw < 1 ? 1 : (npw > NPW_MAX ? NPW_MAX : npw);
}
int grl_edge_bwd_chunk_nodes(int n_src) {
  const int npw = n_src / (4 * 1024);            // ~4 chunks per wave (256 CUs x 4 waves)
  return npw < 1 ? 1 : (npw > NPW_MAX ? NPW_MAX : npw);
}
#else
int grl_edge_fwd_chunk_nodes(int n_dst);
int grl_edge_bwd_chunk_nodes(int n_src);
#endif

// Internal entry points used by edge_conv.hip's C-ABI functions.
int GRL_ENTRY(grl_edge16_launch)(int mode, const st_t* x_in, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                 const int* e_dst, const int* erow, int per_edge, int n_anchor, int n_edges, int anchor_is_dst,
                                 const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* Wk, st_t* out, const st_t* dres, const int* split, int n_slots, const void* wimg,
                                 hipStream_t stream) {
  if (n_anchor <= 0) return 0;
  // chunks: at least ~4 per wave slot of the chip (256 CUs x 12 waves) while the graph allows it
  int npw = grl_edge_fwd_chunk_nodes(n_anchor);
#ifdef GRL_E16_TUNE
  static const int env_npw = getenv("GRL_E16_NPW") ? atoi(getenv("GRL_E16_NPW")) : 0;
  if (env_npw > 0) npw = env_npw;
#endif
  Edge16Params p{x_in, pos_src, pos_dst, rowptr, e_src, e_dst, erow, grid, W1, b1, W2, b2, Wk, n_anchor, n_edges, dim, anchor_is_dst,
                 per_edge, npw, (mode == 0 && split && n_slots >= E16_WAVES && n_slots % E16_WAVES == 0) ? split : nullptr, wimg};
  const int n_chunks = (n_anchor + npw - 1) / npw;
  int blocks = p.split ? n_slots / E16_WAVES : (n_chunks + E16_WAVES - 1) / E16_WAVES;
  int cap = 256 * GRL_E16_WGS;
#ifdef GRL_E16_TUNE                        // diagnostic builds: grid cap and chunk size from the environment
  static const int env_cap = getenv("GRL_E16_BLOCKS") ? atoi(getenv("GRL_E16_BLOCKS")) : 0;
  if (env_cap > 0) cap = env_cap;
#endif
  if (blocks > cap && !p.split) blocks = cap;
  if (blocks < 1) blocks = 1;
  const size_t smem = sizeof(ChainW16);
  GRL_ONCE(hipFuncSetAttribute((const void*)edge16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainW16));
           hipFuncSetAttribute((const void*)edge16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainW16)));
  if (mode == 0) hipLaunchKernelGGL(edge16_kernel<0>, dim3(blocks), dim3(E16_THREADS), smem, stream, p, out, dres);
  else if (mode == 1) return -4;   // (the d x_src-alone instance served round 1's two-launch backward: removed with it in round 6)
  else hipLaunchKernelGGL(edge16_kernel<2>, dim3(blocks), dim3(E16_THREADS), smem, stream, p, out, dres);
  GRL_CHECK_LAUNCH();
  return 0;
}

// The whole edge backward in one launch (edge_bwd16_kernel).  The arrays are the SOURCE-sorted view of the edge set; partial must
// hold `blocks` rows of grl_edge_partial_size() floats and exactly `blocks` workgroups are launched (every row is written).
int GRL_ENTRY(grl_edge_bwd16_launch)(const st_t* x_src, const st_t* dmsg, const float* pos_src, const float* pos_dst, const int* rowptr_s,
                                     const int* src_s, const int* dst_s, const int* erow, int per_edge, int n_src, int n_edges,
                                     const float* grid, int dim, const float* W1, const float* b1, const float* W2, const float* b2,
                                     const float* Wk, const st_t* dres, st_t* dx_src, float* partial, int blocks, const int* split,
                                     const void* wimg, hipStream_t stream) {
  const int npw = grl_edge_bwd_chunk_nodes(n_src);
  Bwd16Params bp{{dmsg, pos_src, pos_dst, rowptr_s, src_s, dst_s, erow, grid, W1, b1, W2, b2, Wk, n_src, n_edges, dim, 0, per_edge, npw,
                  nullptr, wimg},
                 x_src, dres, dx_src, partial, split};
  // one wave per SIMD is a precondition of the unfenced MFMA groups of this kernel (DESIGN.md finding 3): more than half the LDS per
  // workgroup makes a second workgroup on the CU impossible, whatever the register allocation of a future compiler
  static_assert(sizeof(Bwd16Smem) > 80 * 1024, "edge_bwd16_kernel must not share a CU with a second workgroup");
  GRL_ONCE(hipFuncSetAttribute((const void*)edge_bwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Bwd16Smem)));
  hipLaunchKernelGGL(edge_bwd16_kernel, dim3(blocks), dim3(256), sizeof(Bwd16Smem), stream, bp);
  GRL_CHECK_LAUNCH();
  return 0;
}

#if defined(GRL_B16_PHASE)   // (each precision twin has its own counters: grl_edge_bwd16_phase_read / ..._bf16)
int GRL_ENTRY(grl_edge_bwd16_phase_read)(unsigned long long* out16, int reset) {
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_b16phase), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_b16phase), z, sizeof(z)); }
  return 0;
}
#endif

}  // extern "C"
