// Pre-split weight images (round 4).  Until round 3 every MFMA launch rebuilt its split-bf16 weight images in its own prologue: each
// of the 256 workgroups of each of the eight launches of a step read the fp32 weights, split them into hi / lo bf16 (two of the edge
// backward's five images through strided dword loads: the transposes) and stored them to LDS -- 15-40 us at the head of every launch
// whatever the batch size (DESIGN.md findings 8, 25).  The weights change once per step (Adam), so the images are
// now built ONCE per forward pass by one small launch (weight_images.hip, grl_weight_images) into a device buffer whose bytes ARE the
// kernels' LDS structs; a kernel's prologue is a linear 16-byte copy global -> LDS.  The backward kernels of the same step reuse the
// images of the forward (the weights do not change in between).  A NULL image pointer keeps the in-kernel staging (stand-alone calls).
//
// This header holds the LDS / image layouts shared by the consumer kernels and the producer, and the copy.
#pragma once
#include "grl_common.h"

// ---- 16-row edge chain (edge_conv16.hip): k-order inside a 32-block: position 8 g + j <-> feature 16 (j >> 2) + 4 g + (j & 3)
#ifndef GRL_LD1
#define GRL_LD1 40
#endif
constexpr int WI_LD1 = GRL_LD1;  // bf16 elements per image row, layer 1 (K = 14 padded to one 32-deep step): 40 (80-B rows) is 2-way conflicted for
                                 // the ds_read_b128 lane groups, 48 (96-B rows) conflict-free (tools: the bank model of MI355X_MICROARCH.md)
constexpr int WI_LD2 = 64 + 16;  // layers 2 and 3: 160-B rows put the 16 lanes of every ds_read_b128 group on disjoint banks (72: 2-way)
struct ChainW16 {
  unsigned short W1h[64 * WI_LD1], W1l[64 * WI_LD1];
  unsigned short W2h[64 * WI_LD2], W2l[64 * WI_LD2];
  unsigned short Wkh[64 * WI_LD2], Wkl[64 * WI_LD2];
  float b1s[64], b2s[64], grid_s[64];
};
// the fused edge backward's images: the forward's chain + the transposes of Wk and W2 (dg = dZ W products)
struct Edge16Image {
  ChainW16 w;
  unsigned short WkTh[64 * WI_LD2], WkTl[64 * WI_LD2];
  unsigned short W2Th[64 * WI_LD2], W2Tl[64 * WI_LD2];
};
static_assert(sizeof(ChainW16) % 16 == 0 && sizeof(Edge16Image) % 16 == 0, "images are copied in 16-byte units");

// image[n][32 s + 8 g + j] = W[n][32 s + 16 (j >> 2) + 4 g + (j & 3)]   (zero beyond KSRC); one (n, s, g) item per thread and step:
// two 16-byte global loads (when aligned), one 16-byte store per image
//   TRANS: the image of W^T (row n of the image = column n of W [64,64]): the backward chain's dg = dZ W products
template <int KSRC, int KPAD, int NT, bool TRANS = false>
GRL_DEVINL void stage16(unsigned short* hi, unsigned short* lo, const float* __restrict__ W, int ld) {
  constexpr int ITEMS = 64 * (KPAD / 32) * 4;
  for (int idx = threadIdx.x; idx < ITEMS; idx += NT) {
    const int g = idx & 3, s = (idx >> 2) % (KPAD / 32), n = idx / (4 * (KPAD / 32));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
      v[j] = f < KSRC ? (TRANS ? W[f * 64 + n] : W[n * KSRC + f]) : 0.f;
    }
    bf16x8 h, l;
    split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), h, l);
    *reinterpret_cast<bf16x8*>(hi + n * ld + 32 * s + 8 * g) = h;
    GRL_LO(*reinterpret_cast<bf16x8*>(lo + n * ld + 32 * s + 8 * g) = l;)
  }
}
template <int NT>
GRL_DEVINL void stage_chain16_small(ChainW16& s, const float* b1, const float* b2, const float* grid) {
  for (int i = threadIdx.x; i < 64; i += NT) {
    s.b1s[i] = b1[i];
    s.b2s[i] = b2[i];
    s.grid_s[i] = i < 48 ? grid[i] : 0.f;
  }
}

// ---- 32-row edge chain (edge_conv.hip: the one-workgroup-per-tile forward of launches with few tiles)
constexpr int WI_LDB = GRL_LDB(64);   // 72   split-bf16 images (chain)
constexpr int WI_LDB1 = GRL_LDB(16);  // 24
struct ChainW {
  unsigned short W1h[64 * WI_LDB1], W1l[64 * WI_LDB1];
  unsigned short W2h[64 * WI_LDB], W2l[64 * WI_LDB];
  unsigned short Wkh[64 * WI_LDB], Wkl[64 * WI_LDB];
  float b1s[64];
  float b2s[64];
  float grid_s[64];
};
static_assert(sizeof(ChainW) % 16 == 0, "images are copied in 16-byte units");

// ---- ConvNeXt node block forward (node_mlp.hip): W3 [256,64] and W4 [64,256] as 32-row split-bf16 images + the small vectors
constexpr int WI_LB3 = GRL_LDB(64);   // 72
constexpr int WI_LB4 = GRL_LDB(256);  // 264
struct MlpSmemBf {
  unsigned short W3h[256 * WI_LB3], W3l[256 * WI_LB3];
  unsigned short W4h[64 * WI_LB4], W4l[64 * WI_LB4];
  float b3s[256];
  float b4s[64];
  float gam[64];
  float bet[64];
};
static_assert(sizeof(MlpSmemBf) % 16 == 0, "images are copied in 16-byte units");

// ---- ConvNeXt node block backward (node_mlp16.hip): the per-lane static operand fragments of the four waves (wave w owns hidden units
//      64 w .. 64 w + 63): [wave][n-tile][k-step][hi | lo][lane], 16 bytes each -- a wave's load instruction covers 1 KB contiguous
struct Mlp16Image {
  u32x4 w3f[4][4][2][2][64];   // z^T = W3 a^T:        A[m = hidden 16 nt + r][k = channel 32 s + 8 g + j]
  u32x4 w3t[4][4][2][2][64];   // dA^T = W3^T dZ^T:    A[m = channel 16 nt + r][k = hidden, chain order]
  u32x4 w4f[4][4][2][2][64];   // dH^T = W4^T dOut^T:  A[m = hidden][k = channel] = W4[channel][hidden]   (kept in LDS by the kernel)
};
// one (wave, n-tile) slice of the three fragment sets, by the 64 threads `lane` of a producer (the former kernel prologue)
GRL_DEVINL void mlp16_fragments(Mlp16Image& im, const float* __restrict__ W3, const float* __restrict__ W4, int wave, int nt, int lane) {
  constexpr int C = 64, W = 256;
  const int r = lane & 15, g = lane >> 4, j0 = 64 * wave;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 fh, fl;
    {
      const float* p = W3 + (size_t)(j0 + 16 * nt + r) * C + 32 * s + 8 * g;
      split_pair(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), fh, fl);
      im.w3f[wave][nt][s][0][lane] = __builtin_bit_cast(u32x4, fh);
      GRL_LO(im.w3f[wave][nt][s][1][lane] = __builtin_bit_cast(u32x4, fl);)
    }
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = W4[(size_t)(32 * s + 8 * g + j) * W + j0 + 16 * nt + r];
      split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), fh, fl);
      im.w4f[wave][nt][s][0][lane] = __builtin_bit_cast(u32x4, fh);
      GRL_LO(im.w4f[wave][nt][s][1][lane] = __builtin_bit_cast(u32x4, fl);)
    }
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = W3[(size_t)(j0 + 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)) * C + 16 * nt + r];
      split_pair(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), fh, fl);
      im.w3t[wave][nt][s][0][lane] = __builtin_bit_cast(u32x4, fh);
      GRL_LO(im.w3t[wave][nt][s][1][lane] = __builtin_bit_cast(u32x4, fl);)
    }
  }
}

// ---- the copy: `bytes` (a multiple of 16) from a 16-byte-aligned global image into LDS, NT threads, up to 8 loads in flight per thread
template <int NT>
GRL_DEVINL void copy_image(void* lds, const void* __restrict__ img, int bytes) {
  const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(img);
  u32x4* dst = reinterpret_cast<u32x4*>(lds);
  const int n = bytes >> 4;
  int i = threadIdx.x;
  for (; i + 7 * NT < n; i += 8 * NT) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * NT];
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * NT] = v[u];
  }
  for (; i < n; i += NT) dst[i] = src[i];
}

// kinds of images the producer builds (grl_weight_images), and their sizes
enum { WIMG_EDGE16 = 0, WIMG_EDGE32 = 1, WIMG_MLP_FWD = 2, WIMG_MLP_BWD16 = 3, WIMG_KINDS = 4 };
