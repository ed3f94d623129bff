// Producer of the pre-split weight images (grl_wimg.h): ONE launch per forward pass builds, for every convolution of the pass, the
// split-bf16 images its MFMA kernels used to rebuild in each of their 256 workgroups' prologues --
//   kind 0  Edge16Image   W1 | W2 | Wk | Wk^T | W2^T (16-row k-order) + b1, b2, grid       edge16_kernel<*> (prefix), edge_bwd16_kernel
//   kind 1  ChainW        W1 | W2 | Wk (32-row k-order) + b1, b2, grid                       edge_conv_fwd_kernel<true> (few-tile launches)
//   kind 2  MlpSmemBf     W3 | W4 (32-row k-order) + b3, b4, gamma, beta                    node_mlp_fwd_kernel
//   kind 3  Mlp16Image    per-lane operand fragments W3, W3^T, W4^T of the four waves        node_mlp_bwd16_kernel
// The image bytes are exactly what the kernels' own staging code writes to LDS (the same device functions, pointed at global memory),
// so a launch with an image and one without produce bitwise identical results (tests/test_gpu_weight_images.py).
// Reference maths the images feed: hepi.py:76-89,109-123 (basis MLP), ponita/conv.py:64-69,79,112-117 (kernel layer, ConvNeXt block).
#include "grl_wimg_kernel.h"

namespace {

__global__ __launch_bounds__(256) void weight_images_kernel(WimgJobs jobs) { weight_images_body(jobs, (int)blockIdx.y, (int)blockIdx.x); }

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
  if (const int rc = wimg_fill(jobs, n, kinds, srcs, outs)) return rc;
  hipLaunchKernelGGL(weight_images_kernel, dim3(WIMG_PARTS, n), dim3(256), 0, stream, jobs);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
