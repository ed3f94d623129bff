// Micro-benchmark: what limits an fp32-MFMA register chain on gfx950?  (hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../geometry_rl_amd/csrc/grl_common.h"

constexpr int LDW = 68;
// MODE 0: dependent MFMA chain from registers only
// MODE 1: + weight fragments from LDS (ds_read_b128 per 4 MFMAs)
// MODE 2: + GELU on the 16 accumulator values after every 32 MFMAs
// MODE 3: like 2 but two independent accumulators interleaved
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float W[256 * LDW];
  for (int i = threadIdx.x; i < 256 * LDW; i += blockDim.x) W[i] = 1e-3f * (i % 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  float4 x[8];
  for (int t = 0; t < 8; ++t) x[t] = make_float4(0.01f * lane, 0.02f, 0.03f * t, 0.04f);
  f32x16 acc = zero16(), acc2 = zero16();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int nt = 0; nt < 8; ++nt) {
      if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          acc = mfma32(x[t].x, x[t].y, acc); acc = mfma32(x[t].y, x[t].z, acc);
          acc = mfma32(x[t].z, x[t].w, acc); acc = mfma32(x[t].w, x[t].x, acc);
        }
      } else {
        mma_wx<64>(W + (32 * nt + r) * LDW + 4 * h, x, acc);
        if (MODE == 3) mma_wx<64>(W + (32 * ((nt + 1) & 7) + r) * LDW + 4 * h, x, acc2);
      }
      if (MODE >= 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = gelu_f(acc[q]);
        if (MODE == 3) {
#pragma unroll
          for (int q = 0; q < 16; ++q) acc2[q] = gelu_f(acc2[q]);
        }
      }
    }
  }
  float s = 0;
  for (int q = 0; q < 16; ++q) s += acc[q] + acc2[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads, int mfma_per_iter) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 200;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double mf = (double)256 * (threads / 64) * iters * mfma_per_iter;
  printf("%-40s waves/CU=%d  %.3f ms  %.1f TFLOP/s  ns/MFMA/SIMD=%.1f\n", name, threads / 64, ms, mf * 4096 / ms / 1e9,
         ms * 1e6 / (mf / 1024));
  hipFree(d);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("regs only, dependent chain", threads, 256);
    run<1>("+ LDS weight fragments", threads, 256);
    run<2>("+ GELU per 32 MFMAs", threads, 256);
    run<3>("two chains + GELU", threads, 512);
  }
  return 0;
}
