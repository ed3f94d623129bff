// Micro-benchmark (gfx950): do one wave's MFMAs run beside ANOTHER wave's vector instructions on the same SIMD?
// 512-thread workgroups, one per CU: waves 0-3 and 4-7 land on SIMDs 0-3 pairwise.  Role of a wave by (wave >> 2): role A runs a chain of
// dependent v_mfma_f32_16x16x32_bf16 (or 32x32x16), role B a stream of independent v_fma_f32 / v_exp_f32 / ds_read_b128.
// Every pair is run twice: plain, and with the B wave at s_setprio 3.  Three launches per pair: A alone (B waves exit), B alone, both.  If the pipes overlap, T(both) ~ max(T(A), T(B)); if the SIMD serialises
// them, T(both) ~ T(A) + T(B).  Prints s_memtime ticks (100 MHz) per loop for each role.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// AKIND: 0 = 16x16x32 one dependent chain, 1 = 32x32x16 one chain, 2 = 16x16x32 four independent chains
// BKIND: 0 = v_fma_f32 independent, 1 = v_exp_f32, 2 = ds_read_b128 (conflict-free), 3 = v_pk_fma_f32
template <int AKIND, int BKIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int iters, int run_a, int run_b, int prio_b) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, role = wave >> 2;
  for (int i = threadIdx.x; i < 64 * 4 * 8; i += 512) lds[i] = 0.001f * i;
  __syncthreads();
  if (role == 0 && !run_a) return;
  if (role == 1 && !run_b) return;
  float sink = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  if (role == 0) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d = {0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      if (AKIND == 0) {
#pragma unroll
        for (int u = 0; u < 64; ++u) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      } else if (AKIND == 1) {
#pragma unroll
        for (int u = 0; u < 32; ++u) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    sink = c0[0] + c1[1] + c2[2] + c3[3] + d[0];
  } else {
    if (prio_b) __builtin_amdgcn_s_setprio(3);   // the vector-work wave asks for issue priority over the MFMA wave
    float v[16];
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p[8];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i) + 0.5f;
    for (int i = 0; i < 8; ++i) p[i] = v2{v[2 * i], v[2 * i + 1]};
    const float c = 1.0001f;
    const v2 cc = {1.0001f, 0.9999f};
    f32x4 q[4];
    const unsigned addr = (unsigned)(size_t)(lds) + lane * 16;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      if (BKIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
        REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
      } else if (BKIND == 1) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
      } else if (BKIND == 2) {
#define OP(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 3]) : "v"(addr), "n"((i & 7) * 1024));
        REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7]) : "v"(cc));
        REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) sink += v[i];
    for (int i = 0; i < 8; ++i) sink += p[i].x + p[i].y;
    if (BKIND == 2) sink += q[0][0] + q[1][1] + q[2][2] + q[3][3];
  }
  if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
  if (sink == 123.456f) out[threadIdx.x] = sink;
}

template <int AKIND, int BKIND>
void run(const char* an, const char* bn, int prio_b, float* out, unsigned long long* ticks) {
  const int iters = 2000, blocks = 256;
  std::vector<unsigned long long> h(blocks * 8);
  double res[3][2];
  for (int mode = 0; mode < 3; ++mode) {
    const int ra = mode != 1, rb = mode != 0;
    hipMemset(ticks, 0, sizeof(unsigned long long) * blocks * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<AKIND, BKIND>), dim3(blocks), dim3(512), 0, 0, out, ticks, iters, ra, rb, prio_b);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
    res[mode][0] = sa / (blocks * 4) / iters;
    res[mode][1] = sb / (blocks * 4) / iters;
  }
  printf("A = %-24s B = %-14s%s | ticks/loop  A alone %7.3f  B alone %7.3f | together: A %7.3f  B %7.3f | (A slows %.2fx, B slows %.2fx)\n", an, bn,
         res[0][0], res[1][1], res[2][0], res[2][1], res[2][0] / res[0][0], res[2][1] / res[1][1]);
}

int main() {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 4096);
  hipMalloc(&ticks, sizeof(unsigned long long) * 256 * 8);
  printf("per loop: A = 64 x 16x16x32 (or 32 x 32x32x16) MFMAs, B = 64 instructions; s_memtime ticks at 100 MHz\n");
run<0, 0>("16x16x32 one chain", "v_fma_f32", 0, out, ticks);
  run<0, 1>("16x16x32 one chain", "v_exp_f32", 0, out, ticks);
  run<0, 2>("16x16x32 one chain", "ds_read_b128", 0, out, ticks);
  run<0, 3>("16x16x32 one chain", "v_pk_fma_f32", 0, out, ticks);
  run<2, 0>("16x16x32 four chains", "v_fma_f32", 0, out, ticks);
  run<1, 0>("32x32x16 one chain", "v_fma_f32", 0, out, ticks);
  run<1, 1>("32x32x16 one chain", "v_exp_f32", 0, out, ticks);
  run<1, 2>("32x32x16 one chain", "ds_read_b128", 0, out, ticks);
  run<1, 3>("32x32x16 one chain", "v_pk_fma_f32", 0, out, ticks);
run<0, 0>("16x16x32 one chain", "v_fma_f32", 1, out, ticks);
  run<0, 1>("16x16x32 one chain", "v_exp_f32", 1, out, ticks);
  run<0, 2>("16x16x32 one chain", "ds_read_b128", 1, out, ticks);
  run<0, 3>("16x16x32 one chain", "v_pk_fma_f32", 1, out, ticks);
  run<2, 0>("16x16x32 four chains", "v_fma_f32", 1, out, ticks);
  run<1, 0>("32x32x16 one chain", "v_fma_f32", 1, out, ticks);
  run<1, 1>("32x32x16 one chain", "v_exp_f32", 1, out, ticks);
  run<1, 2>("32x32x16 one chain", "ds_read_b128", 1, out, ticks);
  run<1, 3>("32x32x16 one chain", "v_pk_fma_f32", 1, out, ticks);
  hipFree(out);
  hipFree(ticks);
  return 0;
}
