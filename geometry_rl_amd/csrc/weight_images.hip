// Producer of the pre-split weight images (grl_wimg.h): ONE launch per forward pass builds, for every convolution of the pass, the
// split-bf16 images its MFMA kernels used to rebuild in each of their 256 workgroups' prologues --
//   kind 0  Edge16Image   W1 | W2 | Wk | Wk^T | W2^T (16-row k-order) + b1, b2, grid       edge16_kernel<*> (prefix), edge_bwd16_kernel
//   kind 1  ChainW        W1 | W2 | Wk (32-row k-order) + b1, b2, grid                       edge_conv_fwd_kernel<true> (few-tile launches)
//   kind 2  MlpSmemBf     W3 | W4 (32-row k-order) + b3, b4, gamma, beta                    node_mlp_fwd_kernel
//   kind 3  Mlp16Image    per-lane operand fragments W3, W3^T, W4^T of the four waves        node_mlp_bwd16_kernel
// The image bytes are exactly what the kernels' own staging code writes to LDS (the same device functions, pointed at global memory),
// so a launch with an image and one without produce bitwise identical results (tests/test_gpu_weight_images.py).
// Reference maths the images feed: hepi.py:76-89,109-123 (basis MLP), ponita/conv.py:64-69,79,112-117 (kernel layer, ConvNeXt block).
#include "grl_wimg.h"

namespace {

constexpr int WIMG_MAX_JOBS = 24;
constexpr int WIMG_PARTS = 8;   // grid.x: parts of one image, built by different workgroups
struct WimgJobs {
  int n;
  int kind[WIMG_MAX_JOBS];
  const float* src[WIMG_MAX_JOBS][6];   // kind 0 / 1: W1 b1 W2 b2 Wk grid;  kind 2: W3 b3 W4 b4 gamma beta;  kind 3: W3 W4
  void* out[WIMG_MAX_JOBS];
};

__global__ __launch_bounds__(256) void weight_images_kernel(WimgJobs jobs) {
  const int j = blockIdx.y, part = blockIdx.x;
  const float* const* a = jobs.src[j];
  switch (jobs.kind[j]) {
    case WIMG_EDGE16: {
      Edge16Image& im = *reinterpret_cast<Edge16Image*>(jobs.out[j]);
      if (part == 0) {
        stage16<14, 32, 256>(im.w.W1h, im.w.W1l, a[0], WI_LD1);
        stage_chain16_small<256>(im.w, a[1], a[3], a[5]);
      } else if (part == 1) stage16<64, 64, 256>(im.w.W2h, im.w.W2l, a[2], WI_LD2);
      else if (part == 2) stage16<64, 64, 256>(im.w.Wkh, im.w.Wkl, a[4], WI_LD2);
      else if (part == 3) stage16<64, 64, 256, true>(im.WkTh, im.WkTl, a[4], WI_LD2);
      else if (part == 4) stage16<64, 64, 256, true>(im.W2Th, im.W2Tl, a[2], WI_LD2);
      break;
    }
    case WIMG_EDGE32: {
      ChainW& im = *reinterpret_cast<ChainW*>(jobs.out[j]);
      if (part == 0) {
        stage_split<64, 16, 14, 256>(im.W1h, im.W1l, a[0], WI_LDB1);
        for (int i = threadIdx.x; i < 64; i += 256) {
          im.b1s[i] = a[1][i];
          im.b2s[i] = a[3][i];
          im.grid_s[i] = i < 48 ? a[5][i] : 0.f;
        }
      } else if (part == 1) stage_split<64, 64, 64, 256>(im.W2h, im.W2l, a[2], WI_LDB);
      else if (part == 2) stage_split<64, 64, 64, 256>(im.Wkh, im.Wkl, a[4], WI_LDB);
      break;
    }
    case WIMG_MLP_FWD: {
      MlpSmemBf& im = *reinterpret_cast<MlpSmemBf*>(jobs.out[j]);
      if (part < 4) {   // W3 [256,64]: 64 rows per part
        stage_split<64, 64, 64, 256>(im.W3h + part * 64 * WI_LB3, im.W3l + part * 64 * WI_LB3, a[0] + part * 64 * 64, WI_LB3);
      } else if (part < 8) {   // W4 [64,256]: 16 rows per part
        const int p4 = part - 4;
        stage_split<16, 256, 256, 256>(im.W4h + p4 * 16 * WI_LB4, im.W4l + p4 * 16 * WI_LB4, a[2] + p4 * 16 * 256, WI_LB4);
        if (p4 == 0) {
          for (int i = threadIdx.x; i < 256; i += 256) im.b3s[i] = a[1][i];
          for (int i = threadIdx.x; i < 64; i += 256) { im.b4s[i] = a[3][i]; im.gam[i] = a[4][i]; im.bet[i] = a[5][i]; }
        }
      }
      break;
    }
    case WIMG_MLP_BWD16: {
      Mlp16Image& im = *reinterpret_cast<Mlp16Image*>(jobs.out[j]);
      if (part < 4) mlp16_fragments(im, a[0], a[1], part /*wave*/, threadIdx.x >> 6 /*n-tile*/, threadIdx.x & 63);
      break;
    }
    default: break;
  }
}

}  // namespace

extern "C" {

#if !GRL_PREC
// bytes of an image of the given kind (16-byte multiples; the caller aligns every image to 16 bytes)
int grl_wimg_bytes(int kind) {
  switch (kind) {
    case WIMG_EDGE16: return (int)sizeof(Edge16Image);
    case WIMG_EDGE32: return (int)sizeof(ChainW);
    case WIMG_MLP_FWD: return (int)sizeof(MlpSmemBf);
    case WIMG_MLP_BWD16: return (int)sizeof(Mlp16Image);
    default: return -1;
  }
}
int grl_wimg_max_jobs() { return WIMG_MAX_JOBS; }
#endif

// n images in one launch.  kinds [n]; srcs [n][6] device pointers (kind 0 / 1: W1 [64,14], b1, W2 [64,64], b2, Wk [64,64], grid [16,3];
// kind 2: W3 [256,64], b3, W4 [64,256], b4, gamma, beta; kind 3: W3 (16-byte aligned), W4, the rest unused); outs [n]: device
// buffers of grl_wimg_bytes(kind) bytes, 16-byte aligned, fully written except for the lo halves in the bf16 build (never read there).
int GRL_ENTRY(grl_weight_images)(int n, const int* kinds, const float* const* srcs, void* const* outs, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > WIMG_MAX_JOBS) return -2;
  WimgJobs jobs;
  jobs.n = n;
  for (int j = 0; j < n; ++j) {
    if (kinds[j] < 0 || kinds[j] >= WIMG_KINDS || !outs[j]) return -3;
    jobs.kind[j] = kinds[j];
    jobs.out[j] = outs[j];
    const int need = kinds[j] == WIMG_MLP_BWD16 ? 2 : 6;
    for (int k = 0; k < 6; ++k) {
      jobs.src[j][k] = srcs[j * 6 + k];
      if (k < need && !jobs.src[j][k]) return -4;
      if (k == 0 && kinds[j] == WIMG_MLP_BWD16 && (reinterpret_cast<size_t>(jobs.src[j][k]) & 15)) return -4;   // W3 rows: 16-byte loads
    }
  }
  hipLaunchKernelGGL(weight_images_kernel, dim3(WIMG_PARTS, n), dim3(256), 0, stream, jobs);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
