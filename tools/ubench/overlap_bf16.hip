// Micro-benchmark: do bf16 MFMAs of one wave overlap VALU work of the same / a partner wave on one SIMD (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 overlap_bf16.hip -o overlap_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#ifdef INT_VALU
__device__ inline float VOP(float x) { asm volatile("v_xor_b32 %0, 0x5bd1e995, %0" : "+v"(x)); return x; }
#elif defined(CVT_VALU)
__device__ inline float VOP(float x) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x)); return x; }
#elif defined(ASM_FMA)
__device__ inline float VOP(float x) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x)); return x; }
#else
__device__ inline float VOP(float x) { return __builtin_fmaf(x, 1.0001f, 0.5f); }
#endif

// per iteration: NM dependent 32x32x16 bf16 MFMAs, then NV dependent-free VALU fmas on 16 registers
template <int NM, int NV, bool F32, bool STAGGER = false>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
  if (STAGGER && (threadIdx.x >> 8)) {   // second wave of each SIMD: run one VALU phase first so the two waves alternate
#pragma unroll
    for (int n = 0; n < NV / 16; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = VOP(v[i]);
    asm volatile("" ::: "memory");
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
#ifdef PHASED
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int n = 0; n < NV / 16; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = VOP(v[i]);
#ifdef PHASED
    __builtin_amdgcn_sched_barrier(0);
#endif
    asm volatile("" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, bool F32, bool STAGGER = false>
void run(const char* name, int threads) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k<NM, NV, F32, STAGGER>), dim3(256), dim3(threads), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, F32, STAGGER>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s waves/SIMD=%d  %.3f ms  ns/iter/wave-slot=%.1f\n", name, threads / 256, ms, ms * 1e6 / iters);
  hipFree(d);
}

int main() {
  for (int threads : {256, 512}) {
    run<16, 0, false>("bf16: 16 MFMA", threads);
    run<0, 128, false>("128 VALU", threads);
    run<16, 128, false>("bf16: 16 MFMA + 128 VALU", threads);
    run<16, 256, false>("bf16: 16 MFMA + 256 VALU", threads);
    run<16, 128, false, true>("bf16: 16 MFMA + 128 VALU stag", threads);
    run<16, 256, false, true>("bf16: 16 MFMA + 256 VALU stag", threads);
    run<8, 128, true, true>("f32: 8 MFMA + 128 VALU stag", threads);
    run<8, 0, true>("f32: 8 MFMA", threads);
    run<8, 128, true>("f32: 8 MFMA + 128 VALU", threads);
  }
  return 0;
}
