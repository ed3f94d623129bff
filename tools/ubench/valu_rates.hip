// Micro-benchmark (gfx950): what does one wave-instruction of each VALU class cost on a SIMD, alone and with one to three more waves on the
// SIMD?  Independent instructions on 16 registers, unrolled; every CU runs 1 or 2 waves per SIMD.  Prints cycles per
// wave-instruction per SIMD (s_memtime ticks / instructions issued on that SIMD).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  float v[16];
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p[8];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i) + 0.5f;
  for (int i = 0; i < 8; ++i) p[i] = v2{v[2 * i], v[2 * i + 1]};
  const float c = 1.0001f;
  const v2 c2 = {1.0001f, 0.9999f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7]) : "v"(c2));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 2) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 3) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 4) {
#define OP(i) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 5) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 6) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 7) {   // dependent chain of v_fma (latency)
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 8) {   // dependent chain of v_pk_fma
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[0]) : "v"(c2));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 9) {   // mix: exp and fma alternating (does the transcendental unit run beside the main VALU?)
#define OP(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %1" : "+v"(v[i & 7]), "+v"(v[8 + (i & 7)]) : "v"(c));
      REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 10) {  // mix: exp + 3 fma
#define OP(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %3, %3, %2, %3\n\tv_fma_f32 %4, %4, %2, %4" : "+v"(v[i & 3]), "+v"(v[4 + (i & 3)]), "+v"(v[8 + (i & 3)]), "+v"(v[12 + (i & 3)]) : "v"(c));
      REP16(OP)
#undef OP
    } else if (KIND == 11) {  // v_mul_f32 with abs modifier (VOP3)
#define OP(i) asm volatile("v_fma_f32 %0, |%0|, %1, 1.0" : "+v"(v[i]) : "v"(c));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    } else if (KIND == 12) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(c2));
      REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter) {
  float* d;
  unsigned long long* c;
  hipMalloc(&d, 256 * 1024 * 4);
  hipMalloc(&c, 256 * 16 * 8);
  for (int threads : {256, 512, 768, 1024}) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, c, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, c, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * threads / 64);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ticks = (double)h[h.size() / 2];
    const int wps = threads / 256;
    // s_memtime counts at a fixed 100 MHz-derived rate on some parts: report both the tick-based and the wall-based figure
    printf("%-34s waves/SIMD=%d  ticks/instr/wave=%7.2f  ticks/instr/SIMD=%7.2f   wall ns/instr/SIMD=%6.3f\n", name, wps,
           ticks / ((double)iters * per_iter), ticks / ((double)iters * per_iter * wps), ms * 1e6 / ((double)iters * per_iter * wps));
  }
  hipFree(d);
  hipFree(c);
}

int main() {
  run<0>("v_fma_f32 (independent)", 64);
  run<1>("v_pk_fma_f32 (independent)", 64);
  run<12>("v_pk_mul_f32 (independent)", 64);
  run<2>("v_exp_f32", 64);
  run<3>("v_rcp_f32", 64);
  run<4>("v_and_b32", 64);
  run<5>("v_cvt_pk_bf16_f32", 64);
  run<6>("v_perm_b32", 64);
  run<11>("v_fma_f32 with |x| (VOP3)", 64);
  run<7>("v_fma_f32 dependent chain", 64);
  run<8>("v_pk_fma_f32 dependent chain", 64);
  run<9>("pairs: v_exp + v_fma (per pair)", 32);
  run<10>("quads: v_exp + 3 v_fma (per quad)", 16);
  return 0;
}
