// Micro-benchmark (gfx950), round 5: issue cost of the 16-bit vector instructions a packed-f16 GELU epilogue would use -- alone on a SIMD
// (one wave, the regime of the one-wave MFMA kernels; two and four waves beside it) and as FILLERS between dependent MFMAs of the same wave
// (cycles per MFMA gap with n fillers of a kind in it).  Prints s_memtime ticks per instruction / per gap.
//   hipcc --offload-arch=gfx950 -O3 valu16_rates.hip -o /tmp/valu16_rates && /tmp/valu16_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  unsigned v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0x3c003800u + threadIdx.x + 7 * i;   // two f16 near 1.0 / 0.5
  const unsigned c = 0x3c013bffu;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#define OP(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 1) {
#define OP(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 2) {
#define OP(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 3) {
#define OP(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 4) {   // hi half in place (SDWA)
#define OP(i) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 5) {
#define OP(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 6) {
#define OP(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 7) {
#define OP(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 8) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 9) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 10) {
#define OP(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 11) {  // the pair pattern of a packed logistic GELU: 2 transcendentals on the halves + 3 packed ops
#define OP(i) asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\tv_pk_fma_f16 %1, %1, %2, %1\n\tv_pk_mul_f16 %3, %3, %2\n\tv_pk_add_f16 %4, %4, %2" : "+v"(v[i & 3]), "+v"(v[4 + (i & 3)]), "+v"(v[8 + (i & 3)]), "+v"(v[12 + (i & 3)]) : "v"(c));
      REP16(OP)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
  for (int i = 0; i < 16; ++i) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// fillers between DEPENDENT 16x16x32 f16 MFMAs of one wave per SIMD: NF fillers of kind FK per gap
template <int FK, int NF>
__global__ __launch_bounds__(256, 1) void kg(float* out, unsigned long long* cyc, int iters) {
  unsigned v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0x3c003800u + threadIdx.x + 7 * i;
  const unsigned c = 0x3c013bffu;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (_Float16)(0.02f * (threadIdx.x % 5 + i)); }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int i = (g * NF + f) & 15;
        if (FK == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
        else if (FK == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
        else if (FK == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
        else if (FK == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        else if (FK == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        else if (FK == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<unsigned long long*>(&v[(2 * i) & 14])) : "v"(*reinterpret_cast<const unsigned long long*>(&v[(2 * i + 2) & 14])));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
  for (int i = 0; i < 16; ++i) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s) + acc[0] + acc[1] + acc[2] + acc[3];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static float* d;
static unsigned long long* cbuf;
template <int KIND>
void run(const char* name, int per_iter) {
  for (int threads : {256, 512, 1024}) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, cbuf, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, cbuf, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    hipMemcpy(h.data(), cbuf, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ticks = (double)h[h.size() / 2];
    const int wps = threads / 256;
    printf("%-44s waves/SIMD=%d  ticks/instr/wave=%7.2f  ticks/instr/SIMD=%7.2f\n", name, wps, ticks / ((double)iters * per_iter),
           ticks / ((double)iters * per_iter * wps));
  }
}
template <int FK, int NF>
void rung(const char* name) {
  const int iters = 2000;
  hipLaunchKernelGGL((kg<FK, NF>), dim3(256), dim3(256), 0, 0, d, cbuf, 10);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((kg<FK, NF>), dim3(256), dim3(256), 0, 0, d, cbuf, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 4);
  hipMemcpy(h.data(), cbuf, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("gap of a dependent 16x16x32 f16 MFMA chain + %d x %-18s : %6.2f ticks per gap\n", NF, name, (double)h[h.size() / 2] / (iters * 16.0));
}

int main() {
  hipMalloc(&d, 256 * 1024 * 4);
  hipMalloc(&cbuf, 256 * 16 * 8);
  run<8>("v_fma_f32 (reference)", 64);
  run<9>("v_exp_f32 (reference)", 64);
  run<0>("v_pk_fma_f16", 64);
  run<1>("v_pk_mul_f16", 64);
  run<10>("v_pk_add_f16", 64);
  run<2>("v_exp_f16", 64);
  run<3>("v_rcp_f16", 64);
  run<4>("v_exp_f16_sdwa (hi half in place)", 64);
  run<5>("v_cvt_pk_f16_f32", 64);
  run<6>("v_cvt_pkrtz_f16_f32", 64);
  run<7>("v_cvt_f32_f16", 64);
  run<11>("GELU pair pattern: 2 exp_f16 + 3 packed (per 5)", 16);
  rung<0, 0>("(bare)");
  rung<0, 1>("v_fma_f32"); rung<0, 2>("v_fma_f32"); rung<0, 3>("v_fma_f32"); rung<0, 4>("v_fma_f32");
  rung<1, 1>("v_pk_fma_f16"); rung<1, 2>("v_pk_fma_f16"); rung<1, 3>("v_pk_fma_f16"); rung<1, 4>("v_pk_fma_f16");
  rung<5, 1>("v_pk_fma_f32"); rung<5, 2>("v_pk_fma_f32");
  rung<2, 1>("v_exp_f16"); rung<2, 2>("v_exp_f16");
  rung<3, 1>("v_exp_f32"); rung<3, 2>("v_exp_f32");
  rung<4, 1>("v_cvt_pk_f16_f32"); rung<4, 2>("v_cvt_pk_f16_f32");
  return 0;
}
