// Box calibration for bench.py: two FIXED kernels whose rates identify the speed of the box a bench line was taken
// on -- the boxes of the pool differ by several per cent in the clock they hold under load (DESIGN.md finding 32), more than the changes a
// round makes.  Neither is part of the policy-update path; nothing in the package calls them.
//   grl_calib_mfma: every SIMD of the chip issues v_mfma_f32_32x32x16_bf16 back to back on random operands held in registers (four
//                   independent accumulator tiles per wave, one wave per SIMD: 256 workgroups x 4 waves); FLOPs = 1024 waves x iters x 16 x 32768.
//   grl_calib_copy: a float4 grid-stride copy (16 B per lane per access), bytes moved = 2 x bytes.
#include <hip/hip_runtime.h>

namespace {
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 cbf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float cf32x16;

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256, 1) void calib_mfma_kernel(int iters, float* __restrict__ out) {
  // random bf16 operands in [-1, 1): sign + exponent 0x3f / 0x3e + 7 random mantissa bits, from a hash of the lane's global index
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  unsigned short ra[8][8], rb[8][8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned h = hash32(gid * 131u + t * 17u + e);
      ra[t][e] = (unsigned short)(((h & 1u) << 15) | (0x3e80u + ((h >> 1) & 0xffu)));
      rb[t][e] = (unsigned short)((((h >> 9) & 1u) << 15) | (0x3e80u + ((h >> 10) & 0xffu)));
    }
  cbf16x8 a[8], b[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    a[t] = __builtin_bit_cast(cbf16x8, ra[t]);
    b[t] = __builtin_bit_cast(cbf16x8, rb[t]);
  }
  cf32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(2 * k + (q & 1)) & 7], b[(2 * k + (q >> 1) + 3) & 7], acc[q], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[q][i];
  out[gid] = s;
}

// one wave that does nothing for `us` microseconds (s_memrealtime: 100 MHz), sleeping between polls: a delay node for lane-placement experiments
__global__ __launch_bounds__(64) void calib_spin_kernel(int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < 100LL * us) __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" {
// out: 256 * 256 floats (written: keeps the accumulators alive).  FLOPs of one call: 1024 * iters * 16 * 32768.
int grl_calib_mfma(int iters, float* out, hipStream_t stream) {
  if (iters < 1 || !out) return -2;
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(256), dim3(256), 0, stream, iters, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// a one-wave kernel that idles for `us` microseconds (experiments with where the critic's lane starts relative to the actor's kernels)
int grl_calib_spin(int us, hipStream_t stream) {
  if (us <= 0) return 0;
  hipLaunchKernelGGL(calib_spin_kernel, dim3(1), dim3(64), 0, stream, us);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// bytes: a multiple of 16.  Bytes moved by one call: 2 * bytes.
int grl_calib_copy(const void* src, void* dst, long long bytes, hipStream_t stream) {
  if (bytes <= 0 || (bytes & 15) || !src || !dst) return -2;
  hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 8), dim3(256), 0, stream, (const float4*)src, (float4*)dst, bytes / 16);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
}
