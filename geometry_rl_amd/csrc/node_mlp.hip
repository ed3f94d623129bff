// ConvNeXt node block of FiberBundleConv (reference conv.py:64-69,112; ponita.py:219-230):
//   out = x_dst + Linear(256,64)( GELU( Linear(64,256)( LayerNorm64(x2) ) ) )
// Rows are (node, orientation) pairs; each wave owns 32 rows (2 nodes) per step and keeps the whole chain in registers
// (see grl_common.h): LayerNorm by lane-pair shuffles, W3/W4 staged once per workgroup in LDS as MFMA A operands.
//
// Backward is split in two launches:
//   node_mlp_bwd_data    : per-row chain (recompute z, dH = dOut W4, dZ = dH*gelu', dA = dZ W3, LayerNorm backward) and a
//                          hand-off of (dA, dZ) rows to HBM (1.25 KB/row; both kernels run at the HBM roof, so H and xhat are
//                          NOT handed over but recomputed by the second kernel),
//   node_mlp_bwd_weights : recomputes LayerNorm and its hidden tile h (split-bf16 MFMA, W3 tile register-stationary), then the
//                          row-reduction GEMMs  dW3 = dZ^T A, dW4 = dOut^T H  (+ bias / LayerNorm-affine sums) into
//                          per-workgroup partial slabs.
#include "grl_common.h"

namespace {

constexpr int C = 64, O = 16, W = 256;
constexpr int LD3 = GRL_LD(64);   // 68   W3s[256][68]
constexpr int LD4 = GRL_LD(256);  // 260  W4s[64][260]
constexpr float LN_EPS = 1e-5f;

struct MlpSmem {
  float W3s[W * LD3];
  float W4s[C * LD4];
  float b3s[W];
  float b4s[C];
  float gam[C];
  float bet[C];
};

GRL_DEVINL void mlp_stage(MlpSmem& s, const float* W3, const float* b3, const float* W4, const float* b4, const float* gam,
                          const float* bet) {
  stage_matrix(s.W3s, W3, W, C, LD3);
  stage_matrix(s.W4s, W4, C, W, LD4);
  for (int i = threadIdx.x; i < W; i += blockDim.x) s.b3s[i] = b3[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) { s.b4s[i] = b4[i]; s.gam[i] = gam[i]; s.bet[i] = bet[i]; }
}

// split-bf16 weight images for the forward kernel (same bytes as the fp32 image)
constexpr int LB3 = GRL_LDB(64);   // 72
constexpr int LB4 = GRL_LDB(256);  // 264
struct MlpSmemBf {
  unsigned short W3h[W * LB3], W3l[W * LB3];
  unsigned short W4h[C * LB4], W4l[C * LB4];
  float b3s[W];
  float b4s[C];
  float gam[C];
  float bet[C];
};

GRL_DEVINL float pair_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// loads this lane's 8 fragments of row `row` (64 floats)
GRL_DEVINL void load_row(const float* base, size_t row, int h, float4 (&f)[8]) {
  const float4* p = reinterpret_cast<const float4*>(base + row * C) + h;
#pragma unroll
  for (int t = 0; t < 8; ++t) f[t] = p[2 * t];
}
GRL_DEVINL void store_row(float* base, size_t row, int h, const float4 (&f)[8]) {
  float4* p = reinterpret_cast<float4*>(base + row * C) + h;
#pragma unroll
  for (int t = 0; t < 8; ++t) p[2 * t] = f[t];
}

// LayerNorm over the 64 channels of a row split across the lane pair (l, l^32); returns xhat and the affine output
GRL_DEVINL void layer_norm_row(const MlpSmem& s, int h, const float4 (&x)[8], float4 (&xh)[8], float4 (&a)[8], float& rstd) {
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) sum += (x[t].x + x[t].y) + (x[t].z + x[t].w);
  const float mean = pair_sum(sum) * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    xh[t] = make_float4(x[t].x - mean, x[t].y - mean, x[t].z - mean, x[t].w - mean);
    sq += (xh[t].x * xh[t].x + xh[t].y * xh[t].y) + (xh[t].z * xh[t].z + xh[t].w * xh[t].w);
  }
  rstd = rsqrtf(pair_sum(sq) * (1.f / C) + LN_EPS);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 g = *reinterpret_cast<const float4*>(s.gam + 8 * t + 4 * h);
    const float4 b = *reinterpret_cast<const float4*>(s.bet + 8 * t + 4 * h);
    xh[t] = f4_scale(xh[t], rstd);
    a[t] = make_float4(xh[t].x * g.x + b.x, xh[t].y * g.y + b.y, xh[t].z * g.z + b.z, xh[t].w * g.w + b.w);
  }
}

GRL_DEVINL f32x16 bias_acc(const float* bias, int n0, int h) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b = *reinterpret_cast<const float4*>(bias + n0 + 8 * q + 4 * h);
    acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------ forward
// Both GEMMs run on the bf16 matrix pipe with split operands (grl_common.h): per 32-row tile 2 x 96 bf16 MFMAs of 32 cycles
// instead of 2 x 256 fp32 MFMAs of 64 cycles.
__global__ __launch_bounds__(512) void node_mlp_fwd_kernel(const float* __restrict__ x2, const float* __restrict__ x_dst,
                                                           const float* W3, const float* b3, const float* W4, const float* b4,
                                                           const float* gam, const float* bet, float* __restrict__ out,
                                                           int n_rows, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  MlpSmemBf& s = *reinterpret_cast<MlpSmemBf*>(smem_raw);
  stage_split(s.W3h, s.W3l, W3, W, C, C, LB3);
  stage_split(s.W4h, s.W4l, W4, C, W, W, LB4);
  for (int i = threadIdx.x; i < W; i += blockDim.x) s.b3s[i] = b3[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) { s.b4s[i] = b4[i]; s.gam[i] = gam[i]; s.bet[i] = bet[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int n_tiles = (n_rows + 31) >> 5;
  for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += gridDim.x * 8) {
    const int row = tile * 32 + r;
    const bool valid = row < n_rows;
    const size_t rr = valid ? row : 0;
    float4 x[8], xh[8], a[8];
    float rstd;
    load_row(x2, rr, h, x);
    {  // LayerNorm (same arithmetic as layer_norm_row, on the bf16 image's gamma/beta)
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) sum += (x[t].x + x[t].y) + (x[t].z + x[t].w);
      const float mean = pair_sum(sum) * (1.f / C);
      float sq = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        xh[t] = make_float4(x[t].x - mean, x[t].y - mean, x[t].z - mean, x[t].w - mean);
        sq += (xh[t].x * xh[t].x + xh[t].y * xh[t].y) + (xh[t].z * xh[t].z + xh[t].w * xh[t].w);
      }
      rstd = rsqrtf(pair_sum(sq) * (1.f / C) + LN_EPS);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 g = *reinterpret_cast<const float4*>(s.gam + 8 * t + 4 * h);
        const float4 b = *reinterpret_cast<const float4*>(s.bet + 8 * t + 4 * h);
        a[t] = make_float4(xh[t].x * rstd * g.x + b.x, xh[t].y * rstd * g.y + b.y, xh[t].z * rstd * g.z + b.z,
                           xh[t].w * rstd * g.w + b.w);
      }
    }
    bf16x8 ah[4], al[4];
    split_frags<64>(a, ah, al);
    f32x16 o0 = bias_acc(s.b4s, 0, h), o1 = bias_acc(s.b4s, 32, h);
#pragma unroll 1
    for (int nt = 0; nt < 8; ++nt) {
      f32x16 acc = bias_acc(s.b3s, 32 * nt, h);
      mma_wx_bf<64>(s.W3h + (32 * nt + r) * LB3 + 8 * h, s.W3l + (32 * nt + r) * LB3 + 8 * h, ah, al, acc);
      float4 hq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        hq[q] = gelu4(make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
      bf16x8 hh[2], hl[2];
      split_frags<32>(hq, hh, hl);
      mma_wx_bf<32>(s.W4h + r * LB4 + 32 * nt + 8 * h, s.W4l + r * LB4 + 32 * nt + 8 * h, hh, hl, o0);
      mma_wx_bf<32>(s.W4h + (32 + r) * LB4 + 32 * nt + 8 * h, s.W4l + (32 + r) * LB4 + 32 * nt + 8 * h, hh, hl, o1);
    }
    float4 res[8], y[8];
    load_row(x_dst, rr, h, res);
    if (accumulate) {
      load_row(out, rr, h, y);
#pragma unroll
      for (int t = 0; t < 8; ++t) res[t] = f4_add(res[t], y[t]);
    }
    acc_to_frag(o0, y[0], y[1], y[2], y[3]);
    acc_to_frag(o1, y[4], y[5], y[6], y[7]);
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = f4_add(y[t], res[t]);
    if (valid) store_row(out, rr, h, y);
  }
}

// ------------------------------------------------------------------------------------------------ backward, data path
// All three products per hidden tile run split-bf16: z = W3[nt] a (rows of W3), dH = W4[:,nt]^T dOut (rows of W4^T, second LDS
// image), dA += W3[nt]^T dZ -- the transposed W3 tile is produced in registers from the same fragments that fed z
// (transpose32, grl_common.h), so no third weight image is needed.
struct MlpSmemBwd {
  unsigned short W3h[W * LB3], W3l[W * LB3];    // [256 n][64 k]
  unsigned short W4Th[W * LB3], W4Tl[W * LB3];  // [256 n][64 m]  (W4 transposed)
  float b3s[W];
  float gam[C];
  float bet[C];
};

__global__ __launch_bounds__(512) void node_mlp_bwd_data_kernel(const float* __restrict__ x2, const float* __restrict__ dout,
                                                                const float* W3, const float* b3, const float* W4,
                                                                const float* b4, const float* gam, const float* bet,
                                                                float* __restrict__ dx2, float* __restrict__ da_buf,
                                                                float* __restrict__ dz_buf, int n_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  MlpSmemBwd& s = *reinterpret_cast<MlpSmemBwd*>(smem_raw);
  stage_split(s.W3h, s.W3l, W3, W, C, C, LB3);
  for (int idx = threadIdx.x; idx < W * C; idx += blockDim.x) {  // W4^T image: row n holds W4[m][n] over (permuted) m
    const int n = idx >> 6, p = idx & 63;
    const int q = (p >> 2) & 3;
    const int m = (p & ~15) + ((q == 1) ? 8 : (q == 2) ? 4 : 4 * q) + (p & 3);
    const float w = W4[m * W + n];
    s.W4Th[n * LB3 + p] = (unsigned short)(__float_as_uint(w) >> 16);
    s.W4Tl[n * LB3 + p] = (unsigned short)(pack_rn(w - trunc_bf16(w), 0.f) & 0xFFFFu);
  }
  for (int i = threadIdx.x; i < W; i += blockDim.x) s.b3s[i] = b3[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) { s.gam[i] = gam[i]; s.bet[i] = bet[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  bf16x8 sel0, sel1;
  make_selectors(sel0, sel1);
  const int n_tiles = (n_rows + 31) >> 5;
  for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += gridDim.x * 8) {
    const int row = tile * 32 + r;
    const bool valid = row < n_rows;
    const size_t rr = valid ? row : 0;
    float rstd, mean;
    bf16x8 ah[4], al[4], dyh[4], dyl[4];
    {
      float4 x[8], a[8], dy[8];
      load_row(x2, rr, h, x);
      load_row(dout, rr, h, dy);
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) sum += (x[t].x + x[t].y) + (x[t].z + x[t].w);
      mean = pair_sum(sum) * (1.f / C);
      float sq = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        x[t] = make_float4(x[t].x - mean, x[t].y - mean, x[t].z - mean, x[t].w - mean);
        sq += (x[t].x * x[t].x + x[t].y * x[t].y) + (x[t].z * x[t].z + x[t].w * x[t].w);
      }
      rstd = rsqrtf(pair_sum(sq) * (1.f / C) + LN_EPS);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 g = *reinterpret_cast<const float4*>(s.gam + 8 * t + 4 * h);
        const float4 b = *reinterpret_cast<const float4*>(s.bet + 8 * t + 4 * h);
        a[t] = make_float4(x[t].x * rstd * g.x + b.x, x[t].y * rstd * g.y + b.y, x[t].z * rstd * g.z + b.z, x[t].w * rstd * g.w + b.w);
      }
      if (!valid) {
#pragma unroll
        for (int t = 0; t < 8; ++t) dy[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      split_frags<64>(a, ah, al);
      split_frags<64>(dy, dyh, dyl);
    }
    f32x16 da0 = zero16(), da1 = zero16();
#pragma unroll 1
    for (int nt = 0; nt < 8; ++nt) {
      // this hidden tile's W3 rows as A operands (lane = hidden unit)
      bf16x8 wh[4], wl[4];
      const unsigned short* w3h = s.W3h + (32 * nt + r) * LB3 + 8 * h;
      const unsigned short* w3l = s.W3l + (32 * nt + r) * LB3 + 8 * h;
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        wh[sidx] = *reinterpret_cast<const bf16x8*>(w3h + 16 * sidx);
        wl[sidx] = *reinterpret_cast<const bf16x8*>(w3l + 16 * sidx);
      }
      f32x16 z = bias_acc(s.b3s, 32 * nt, h);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        z = mfma_bf(wh[sidx], ah[sidx], z);
        z = mfma_bf(wl[sidx], ah[sidx], z);
        z = mfma_bf(wh[sidx], al[sidx], z);
      }
      f32x16 dh = zero16();
      mma_wx_bf<64>(s.W4Th + (32 * nt + r) * LB3 + 8 * h, s.W4Tl + (32 * nt + r) * LB3 + 8 * h, dyh, dyl, dh);
      float4 dz[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dz[q] = f4_mul(make_float4(dh[4 * q], dh[4 * q + 1], dh[4 * q + 2], dh[4 * q + 3]),
                       gelu_grad4(make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3])));
      if (valid) {
        float4* zp = reinterpret_cast<float4*>(dz_buf + rr * W + 32 * nt) + h;
#pragma unroll
        for (int q = 0; q < 4; ++q) zp[2 * q] = dz[q];
      }
      bf16x8 dzh[2], dzl[2];
      split_frags<32>(dz, dzh, dzl);
      // dA[k][r] += sum_n W3[32nt+n][k] dZ[r][n]: transposed W3 tile (lane = k) from the fragments above
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        bf16x8 th0, th1, tl0, tl1;
        acc_to_bf(transpose32(wh[2 * kt], wh[2 * kt + 1], sel0, sel1), th0, th1);
        acc_to_bf(transpose32(wl[2 * kt], wl[2 * kt + 1], sel0, sel1), tl0, tl1);
        f32x16& da = kt == 0 ? da0 : da1;
        da = mfma_bf(th0, dzh[0], da); da = mfma_bf(tl0, dzh[0], da); da = mfma_bf(th0, dzl[0], da);
        da = mfma_bf(th1, dzh[1], da); da = mfma_bf(tl1, dzh[1], da); da = mfma_bf(th1, dzl[1], da);
      }
    }
    float4 da[8], xh[8];
    acc_to_frag(da0, da[0], da[1], da[2], da[3]);
    acc_to_frag(da1, da[4], da[5], da[6], da[7]);
    load_row(x2, rr, h, xh);  // xhat is rebuilt from the (cache-resident) row instead of living in 32 registers across the loop
#pragma unroll
    for (int t = 0; t < 8; ++t)
      xh[t] = make_float4((xh[t].x - mean) * rstd, (xh[t].y - mean) * rstd, (xh[t].z - mean) * rstd, (xh[t].w - mean) * rstd);
    // LayerNorm backward:  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = da * gamma
    float sg = 0.f, sgx = 0.f;
    float4 g[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 gm = *reinterpret_cast<const float4*>(s.gam + 8 * t + 4 * h);
      g[t] = f4_mul(da[t], gm);
      sg += (g[t].x + g[t].y) + (g[t].z + g[t].w);
      sgx += (g[t].x * xh[t].x + g[t].y * xh[t].y) + (g[t].z * xh[t].z + g[t].w * xh[t].w);
    }
    const float mg = pair_sum(sg) * (1.f / C), mgx = pair_sum(sgx) * (1.f / C);
    float4 dx[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      dx[t] = make_float4(rstd * (g[t].x - mg - xh[t].x * mgx), rstd * (g[t].y - mg - xh[t].y * mgx),
                          rstd * (g[t].z - mg - xh[t].z * mgx), rstd * (g[t].w - mg - xh[t].w * mgx));
    if (valid) {
      store_row(dx2, rr, h, dx);
      store_row(da_buf, rr, h, da);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, weights
// partial slab per workgroup: [dW3 256x64 | db3 256 | dW4 64x256 | db4 64 | dgamma 64 | dbeta 64]
constexpr int MLP_PARTIAL = W * C + W + C * W + C + C + C;
constexpr int LDH = W + 4;  // 260
constexpr int LDA = C + 4;  // 68

__global__ __launch_bounds__(512) void node_mlp_bwd_weights_kernel(const float* __restrict__ x2,
                                                                   const float* __restrict__ da_buf,
                                                                   const float* __restrict__ dz_buf,
                                                                   const float* __restrict__ dout, const float* __restrict__ W3,
                                                                   const float* __restrict__ b3, const float* gam,
                                                                   const float* bet, float* __restrict__ partial, int n_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  float* DZ = smem_raw;              // [32][260]  handed over by the data kernel
  float* H = DZ + 32 * LDH;          // [32][260]  recomputed here: h = gelu(W3 a + b3) (split-bf16 MFMA, W3 tile in registers)
  float* A = H + 32 * LDH;           // [32][68]   a = LayerNorm(x2) (recomputed here)
  float* DO = A + 32 * LDA;          // [32][68]
  float* DA = DO + 32 * LDA;         // [32][68]
  float* XH = DA + 32 * LDA;         // [32][68]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  f32x16 dW3[2], dW4[2];
  dW3[0] = zero16(); dW3[1] = zero16(); dW4[0] = zero16(); dW4[1] = zero16();
  float colsum = 0.f;  // tid<256: db3[tid]; 256..319: db4; 320..383: dgamma; 384..447: dbeta
  // this wave's hidden tile of W3 (rows 32*wave .. +31) as stationary split-bf16 A operands
  bf16x8 wh[4], wl[4];
  {
    const float* wrow = W3 + (size_t)(32 * wave + r) * C;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
      split_pair(*reinterpret_cast<const float4*>(wrow + 16 * sidx + 4 * h), *reinterpret_cast<const float4*>(wrow + 16 * sidx + 8 + 4 * h),
                 wh[sidx], wl[sidx]);
  }
  const int n_chunks = (n_rows + 31) >> 5;
  // register-staged software pipeline: the next chunk's rows are requested before this chunk's MFMAs and written to LDS
  // after them, so the HBM latency hides behind the matrix work (one LDS image, two barriers/chunk)
  float4 pw_dz[4], p_x, p_da, p_dy;
  const int nr = tid >> 4, nc4 = tid & 15;
  auto fetch = [&](int ch) {
    const int row0 = ch * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 512 * q, rr = idx >> 6, c4 = idx & 63;
      const bool ok = row0 + rr < n_rows;
      const size_t g = (size_t)(ok ? row0 + rr : 0) * W + 4 * c4;
      pw_dz[q] = *reinterpret_cast<const float4*>(dz_buf + g);
      if (!ok) pw_dz[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool ok = row0 + nr < n_rows;
    const size_t g = (size_t)(ok ? row0 + nr : 0) * C + 4 * nc4;
    p_x = *reinterpret_cast<const float4*>(x2 + g);
    p_da = *reinterpret_cast<const float4*>(da_buf + g);
    p_dy = *reinterpret_cast<const float4*>(dout + g);
    if (!ok) { p_da = make_float4(0.f, 0.f, 0.f, 0.f); p_dy = p_da; }
  };
  const float4 gm = *reinterpret_cast<const float4*>(gam + 4 * nc4), bt = *reinterpret_cast<const float4*>(bet + 4 * nc4);
  int ch = blockIdx.x;
  if (ch < n_chunks) fetch(ch);
  for (; ch < n_chunks; ch += gridDim.x) {
    const bool ok_row = ch * 32 + nr < n_rows;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 512 * q, rr = idx >> 6, c4 = idx & 63;
      *reinterpret_cast<float4*>(DZ + rr * LDH + 4 * c4) = pw_dz[q];
    }
    {
      // LayerNorm of this row: its 64 values sit in the 16 consecutive lanes that share nr
      float sm = (p_x.x + p_x.y) + (p_x.z + p_x.w);
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) sm += __shfl_xor(sm, off, 64);
      const float mean = sm * (1.f / C);
      float4 xc = make_float4(p_x.x - mean, p_x.y - mean, p_x.z - mean, p_x.w - mean);
      float sq = (xc.x * xc.x + xc.y * xc.y) + (xc.z * xc.z + xc.w * xc.w);
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) sq += __shfl_xor(sq, off, 64);
      const float rstd = rsqrtf(sq * (1.f / C) + LN_EPS);
      float4 xh = f4_scale(xc, rstd);
      float4 a = make_float4(xh.x * gm.x + bt.x, xh.y * gm.y + bt.y, xh.z * gm.z + bt.z, xh.w * gm.w + bt.w);
      if (!ok_row) { a = make_float4(0.f, 0.f, 0.f, 0.f); xh = a; }
      *reinterpret_cast<float4*>(A + nr * LDA + 4 * nc4) = a;
      *reinterpret_cast<float4*>(DO + nr * LDA + 4 * nc4) = p_dy;
      *reinterpret_cast<float4*>(DA + nr * LDA + 4 * nc4) = p_da;
      *reinterpret_cast<float4*>(XH + nr * LDA + 4 * nc4) = xh;
    }
    __syncthreads();
    if (ch + (int)gridDim.x < n_chunks) fetch(ch + gridDim.x);
    // ---- recompute this wave's hidden tile: h[r][32w + n] = gelu(W3[32w+n] . a[r] + b3)
    {
      float4 af[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) af[t] = *reinterpret_cast<const float4*>(A + r * LDA + 8 * t + 4 * h);
      bf16x8 ah[4], al[4];
      split_frags<64>(af, ah, al);
      f32x16 z = bias_acc(b3, 32 * wave, h);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        z = mfma_bf(wh[sidx], ah[sidx], z);
        z = mfma_bf(wl[sidx], ah[sidx], z);
        z = mfma_bf(wh[sidx], al[sidx], z);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(H + r * LDH + 32 * wave + 8 * q + 4 * h) =
            gelu4(make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]));
    }
    // wave w owns hidden tile nt = w:  dW3[32w..32w+31][0..63] and dW4[0..63][32w..32w+31]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      mma_tn<32>(DZ + 4 * h * LDH + 32 * wave + r, LDH, A + 4 * h * LDA + 32 * kt + r, LDA, dW3[kt]);
      mma_tn<32>(DO + 4 * h * LDA + 32 * kt + r, LDA, H + 4 * h * LDH + 32 * wave + r, LDH, dW4[kt]);
    }
    if (tid < 256) {
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) colsum += DZ[rr * LDH + tid];
    } else if (tid < 320) {
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) colsum += DO[rr * LDA + (tid - 256)];
    } else if (tid < 384) {
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) colsum += DA[rr * LDA + (tid - 320)] * XH[rr * LDA + (tid - 320)];
    } else if (tid < 448) {
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) colsum += DA[rr * LDA + (tid - 384)];
    }
  }
  float* out = partial + (size_t)blockIdx.x * MLP_PARTIAL;
  float* oW3 = out, *ob3 = oW3 + W * C, *oW4 = ob3 + W, *ob4 = oW4 + C * W, *og = ob4 + C, *obt = og + C;
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) {
    const int n = (rho & 3) + 8 * (rho >> 2) + 4 * h;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      oW3[(32 * wave + n) * C + 32 * kt + r] = dW3[kt][rho];        // D[n][k]
      oW4[(32 * kt + n) * W + 32 * wave + r] = dW4[kt][rho];        // D[m][n']
    }
  }
  if (tid < 256) ob3[tid] = colsum;
  else if (tid < 320) ob4[tid - 256] = colsum;
  else if (tid < 384) og[tid - 320] = colsum;
  else if (tid < 448) obt[tid - 384] = colsum;
}

int blocks_for(int n_rows, int rows_per_block, int cap) {
  const int b = (n_rows + rows_per_block - 1) / rows_per_block;
  return b < 1 ? 1 : (b < cap ? b : cap);
}

}  // namespace

extern "C" {

int grl_node_mlp_partial_size() { return MLP_PARTIAL; }
int grl_node_mlp_bwd_blocks(int n_rows) { return blocks_for(n_rows, 32, 256); }

// rows = n_nodes*16.  out = (accumulate ? out : 0) + x_dst + MLP(LN(x2))
int grl_node_mlp_fwd(const float* x2, const float* x_dst, const float* W3, const float* b3, const float* W4, const float* b4,
                     const float* gamma, const float* beta, float* out, int n_rows, int accumulate, hipStream_t stream) {
  if (n_rows <= 0) return 0;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)node_mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmemBf));
    attr = true;
  }
  hipLaunchKernelGGL(node_mlp_fwd_kernel, dim3(blocks_for(n_rows, 256, 256)), dim3(512), sizeof(MlpSmemBf), stream, x2, x_dst,
                     W3, b3, W4, b4, gamma, beta, out, n_rows, accumulate);
  GRL_CHECK_LAUNCH();
  return 0;
}

// Scratch: da_buf [n_rows,64], dz_buf [n_rows,256]; partial [grl_node_mlp_bwd_blocks(n_rows)][partial_size].
// d x_dst is simply dout (residual) and is not produced here.
int grl_node_mlp_bwd(const float* x2, const float* dout, const float* W3, const float* b3, const float* W4, const float* b4,
                     const float* gamma, const float* beta, float* dx2, float* da_buf, float* dz_buf, float* partial, int n_rows,
                     hipStream_t stream) {
  if (n_rows <= 0) return 0;
  static bool attr = false;
  const size_t smem_w = sizeof(float) * (2 * 32 * LDH + 4 * 32 * LDA);
  if (!attr) {
    hipFuncSetAttribute((const void*)node_mlp_bwd_data_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmemBwd));
    hipFuncSetAttribute((const void*)node_mlp_bwd_weights_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_w);
    attr = true;
  }
  grl_prof_begin("node_mlp_bwd_data_kernel", stream);
  hipLaunchKernelGGL(node_mlp_bwd_data_kernel, dim3(blocks_for(n_rows, 256, 256)), dim3(512), sizeof(MlpSmemBwd), stream, x2,
                     dout, W3, b3, W4, b4, gamma, beta, dx2, da_buf, dz_buf, n_rows);
  grl_prof_end(stream);
  GRL_CHECK_LAUNCH();
  grl_prof_begin("node_mlp_bwd_weights_kernel", stream);
  hipLaunchKernelGGL(node_mlp_bwd_weights_kernel, dim3(grl_node_mlp_bwd_blocks(n_rows)), dim3(512), smem_w, stream, x2, da_buf,
                     dz_buf, dout, W3, b3, gamma, beta, partial, n_rows);
  grl_prof_end(stream);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
