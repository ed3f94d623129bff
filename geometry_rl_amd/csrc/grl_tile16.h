// 16-row tile helpers shared by the kernels that run ONE 16-row tile (an edge's or a node's 16 orientations) per wave pass
// (edge_conv16.hip, node_mlp16.hip).
//
// Lane l = (row r = l & 15, k-group g = l >> 4).  Chain products D[n][r] += sum_k W[n][k] X[r][k] run on v_mfma_f32_16x16x32_bf16 with
// the weight tile (16 outputs n) on the A side and the 16 rows on the B side; the f32x4 accumulator of n-tile nt holds outputs
// 16 nt + 4 g + u in element u, so two consecutive n-tiles ARE the 8 B-operand elements of one K-step of the next product, in the
// k-order  position 32 s + 8 g + j  <->  feature 32 s + 16 (j >> 2) + 4 g + (j & 3).
// Products that contract over the 16 ROWS (weight gradients) run on v_mfma_f32_32x32x16_bf16 (K = the 16 rows): each operand is written
// once to a wave-private (or workgroup-shared) LDS image [row][feature] and read back transposed with ds_read_b64_tr_b16.
#pragma once
#include "grl_common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
GRL_DEVINL f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
GRL_DEVINL float4 v4(const f32x4v& a) { return make_float4(a[0], a[1], a[2], a[3]); }

// A staging image holds 16 rows x 64 features of bf16 in 4-row x 16-feature blocks of 128 B (the unit ds_read_b64_tr_b16 fetches per
// 16-lane group), block (row >> 2, feature >> 4) at ((row >> 2) * 4 + (feature >> 4)) * 128 B.  Inside a block the four rows are rotated
// by the feature block (row slot = (row + (feature >> 4)) & 3, 32 B each) and the four 8-byte feature quads of a row by the block row
// (quad ^ (row >> 2)).  Three access patterns then touch every bank once: a ds_write_b64 group of 16 ROWS with one quad (the chain
// layout, stage_put), a ds_write_b64 group of 16 QUADS of one row (row-major producers: the LayerNorm stage of node_mlp16.hip; without
// the row rotation its four feature blocks, 128 B apart, met in the same banks: measured conflict ratio 0.42), and the 32 lanes of a
// transposed read (two adjacent blocks).  2 KB per image, no padding.
constexpr int STG = 16 * 64;   // bf16 elements per image
GRL_DEVINL int stg_off(int row, int feat_quad /* feature >> 2, 0..15 */) {
  const int fb = feat_quad >> 2;
  return (((row >> 2) * 4 + fb) * 64) + (((row + fb) & 3) * 16) + (((feat_quad & 3) ^ (row >> 2)) << 2);
}

typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s16 lds_v4s16;
// fragment (8 rows x this lane's feature) of a staging image for a 32x32x16 operand: tile t = 32 features, lane (m = l & 31, h = l >> 5)
// gets rows 8 h + j of feature 32 t + m.  Two transposed block reads (4 rows x 16 features per 16-lane group each).
GRL_DEVINL bf16x8 tr_frag(const unsigned short* img, int t, int lane) {
  const int h = lane >> 5, half16 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  // block row 2 h (rows 8 h .. 8 h + 3) and 2 h + 1, feature block 2 t + half16; this lane supplies row q, quad p of the block
  const int fb = 2 * t + half16, qs = ((q + fb) & 3) * 16;
  const unsigned short* base = img + ((2 * h) * 4 + fb) * 64 + qs + ((p ^ (2 * h)) << 2);
  const unsigned short* base1 = img + ((2 * h + 1) * 4 + fb) * 64 + qs + ((p ^ (2 * h + 1)) << 2);
  const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)base);
  const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)base1);
  typedef short v8s16 __attribute__((ext_vector_type(8)));
  const v8s16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// this lane's chain-layout fragments (row r, features 32 s + {4 g .. 4 g + 3} and 32 s + 16 + {4 g .. 4 g + 3}) -> image[row][feature]
template <int KS>
GRL_DEVINL void stage_put(unsigned short* ih, unsigned short* il, const bf16x8 (&fh)[KS], const bf16x8 (&fl)[KS], int r, int g) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int o0 = stg_off(r, 8 * s + g), o1 = stg_off(r, 8 * s + 4 + g);   // features 32 s + 4 g .. and 32 s + 16 + 4 g ..
    const u32x4 h = __builtin_bit_cast(u32x4, fh[s]);
    *reinterpret_cast<uint2*>(ih + o0) = make_uint2(h[0], h[1]);
    *reinterpret_cast<uint2*>(ih + o1) = make_uint2(h[2], h[3]);
#if !GRL_PREC
    const u32x4 l = __builtin_bit_cast(u32x4, fl[s]);
    *reinterpret_cast<uint2*>(il + o0) = make_uint2(l[0], l[1]);
    *reinterpret_cast<uint2*>(il + o1) = make_uint2(l[2], l[3]);
#endif
  }
}
// Weight-gradient accumulators are pinned to the accumulator half of the register file: the MFMA is an asm statement with the tile as
// a read-write "a" operand (updated in place, never copied), while the files that use it are compiled with
// -mllvm -amdgpu-mfma-vgpr-form so that the chain's builtin MFMAs keep their results in ordinary VGPRs, where the activation reads them
// (with the default selection a 512-register kernel puts EVERY MFMA result into AGPRs: 450 v_accvgpr moves per pass, a third of the
// vector issue slots).  asm is opaque to the hazard recognizer; the operands come from LDS (counted loads: the compiler waits for them).
// Round 4: the "s_nop 1" that used to stand in front of every asm MFMA is gone (30 instructions of the edge backward's 1 366 per pass, 24 of
// the node-MLP backward's 828 per chunk: a lone wave pays ~5 cycles for each).  What it covered -- an operand register written by a VALU
// instruction less than two wait states before the MFMA reads it (the compiler inserts those wait states for builtin MFMAs; asm is opaque
// to its hazard recognizer) -- is now CHECKED on the generated ISA at build time instead: tools/isa_acc_lint.py (tests/test_isa_lint.py)
// fails on any vector write to a source register of an asm MFMA inside that window.  -DGRL_MFMA_ASM_NOP=1 restores the nop.
#ifndef GRL_MFMA_ASM_NOP
#define GRL_MFMA_ASM_NOP 0
#endif
GRL_DEVINL void mfma32_acc(const bf16x8& a, const bf16x8& b, f32x16& c) {
#if GRL_MFMA_ASM_NOP
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
}
// Before the accumulators are read with vector instructions at the end of a launch the last asm MFMA must have drained (an MFMA result
// needs 18 wait states before a VALU read, invisible to the compiler for asm MFMAs).  Every tile is re-defined by an (empty, volatile) asm statement behind the drain, so no v_accvgpr_read of a tile can be scheduled above it (with a bare memory clobber the compiler hoisted reads
// of pinned tiles above the s_nops; correct only because those tiles happened to be idle for long enough).
// (volatile asm statements keep their order: the nops first, then one empty statement per tile that re-defines it)
GRL_DEVINL void acc_pin(f32x16& t) { asm volatile("" : "+a"(t)); }
template <class... Tiles>
GRL_DEVINL void acc_drain(Tiles&... tiles) {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  (acc_pin(tiles), ...);
}
