// Minimal reproducer attempt for DESIGN.md finding 3 ("MFMA operand hazard with two waves per SIMD", gfx950 / ROCm 7.2).
//
// The production kernels saw ~3e-4 of their 32-row tiles come out wrong, different ones on every run, when (a) the split-bf16 weight
// fragments of a dependent MFMA chain were loaded from LDS BETWEEN the MFMAs of the chain (mma_wx_bf, csrc/grl_common.h) and (b) a
// second wave shared the SIMD.  The cure was structural (mma_wx_bf_fenced: all fragment loads -> MFMAs -> a VALU read of the accumulator).
// This file isolates the pattern: the three-layer 64-wide chain of the edge kernels (W on the A side from LDS, activation rows on the B
// side from registers, exact-erf GELU between the layers), nothing else -- no graph, no gathers, no accumulation across tiles.
//   variant U: unfenced groups (ds_read_b128 interleaved with the dependent v_mfma_f32_32x32x16_bf16 chain)
//   variant F: fenced groups
// Both perform the same arithmetic in the same order, so every output must be BITWISE equal to variant F's, run after run.
// Occupancy: 256-thread workgroups, __launch_bounds__(256, 2); grid = 512 (two workgroups per CU = two waves per SIMD) or 256 (one).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_lds_hazard.hip -o mfma_lds_hazard && ./mfma_lds_hazard [repeats]
// Output: per configuration the number of runs / tiles that differ from the fenced reference.  The ISA of variant U's loop is kept in
// profiles/r02_mfma_hazard_isa_unfenced.s (hipcc -S of this file).
#include "../../geometry_rl_amd/csrc/grl_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int LDBW = GRL_LDB(64);
struct Smem {
  unsigned short Wh[3][64 * LDBW], Wl[3][64 * LDBW];
  float bias[3][64];
};

// GATHER: like the production forward, every chain runs with eight 16-byte global loads of a gathered row in flight (issued in
// front of the chain, consumed by a multiply behind it) and the next tile's index load issued under it.
template <bool FENCED, bool GATHER>
__global__ __launch_bounds__(256, 2) void chain_kernel(const float* __restrict__ W /*[3][64][64]*/, const float* __restrict__ bias,
                                                       const float* __restrict__ X /*[rows][64]*/, float* __restrict__ Y, int n_tiles,
                                                       int inner, const int* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  Smem& s = *reinterpret_cast<Smem*>(smem_raw);
  for (int l = 0; l < 3; ++l) stage_split<64, 64, 64, 256>(s.Wh[l], s.Wl[l], W + l * 4096, LDBW);
  for (int i = threadIdx.x; i < 192; i += blockDim.x) s.bias[i / 64][i % 64] = bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    float4 x[8];
    const float4* xp = reinterpret_cast<const float4*>(X + ((size_t)tile * 32 + r) * 64) + h;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = xp[2 * t];
    for (int rep = 0; rep < inner; ++rep) {   // the same chain again on its own output: keeps the SIMD in the pattern for long
      float4 gv[8];
      if (GATHER) {
        const int row = idx[(tile * 32 + r + rep * 7919) % (n_tiles * 32)];
        const float4* gp = reinterpret_cast<const float4*>(X + (size_t)row * 64) + h;
#pragma unroll
        for (int t = 0; t < 8; ++t) gv[t] = gp[2 * t];     // in flight behind the chain
      }
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        bf16x8 xh[4], xl[4];
        split_frags<64>(x, xh, xl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x16 acc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(s.bias[l] + 32 * nt + 8 * q + 4 * h);
            acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
          }
          const unsigned short* wh = s.Wh[l] + (32 * nt + r) * LDBW + 8 * h, *wl = s.Wl[l] + (32 * nt + r) * LDBW + 8 * h;
          auto epi = [&](const f32x16& a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) x[4 * nt + q] = gelu4(make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]));
          };
          if (FENCED) {
            mma_wx_bf_fenced<64>(wh, wl, xh, xl, acc, epi);
          } else {
            mma_wx_bf<64>(wh, wl, xh, xl, acc);
            epi(acc);
          }
        }
      }
      if (GATHER) {
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = make_float4(fmaf(x[t].x, 0.5f, 0.25f * gv[t].x), fmaf(x[t].y, 0.5f, 0.25f * gv[t].y),
                                                       fmaf(x[t].z, 0.5f, 0.25f * gv[t].z), fmaf(x[t].w, 0.5f, 0.25f * gv[t].w));
      }
    }
    float4* yp = reinterpret_cast<float4*>(Y + ((size_t)tile * 32 + r) * 64) + h;
#pragma unroll
    for (int t = 0; t < 8; ++t) yp[2 * t] = x[t];
  }
}

int main(int argc, char** argv) {
  const int repeats = argc > 1 ? atoi(argv[1]) : 40;
  const int n_tiles = 32768, rows = n_tiles * 32, inner = 4;
  std::vector<float> hW(3 * 4096), hb(192), hX((size_t)rows * 64);
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hW) v = 0.25f * rnd();
  for (auto& v : hb) v = 0.1f * rnd();
  for (auto& v : hX) v = 2.0f * rnd();
  float *W, *b, *X, *Y;
  hipMalloc(&W, hW.size() * 4); hipMalloc(&b, hb.size() * 4); hipMalloc(&X, hX.size() * 4); hipMalloc(&Y, hX.size() * 4);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  std::vector<int> hidx(rows);
  for (auto& v : hidx) { seed = seed * 1664525u + 1013904223u; v = (int)((seed >> 4) % (unsigned)rows); }
  int* idx;
  hipMalloc(&idx, hidx.size() * 4);
  hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)chain_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
  hipFuncSetAttribute((const void*)chain_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
  hipFuncSetAttribute((const void*)chain_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
  hipFuncSetAttribute((const void*)chain_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
  std::vector<float> ref(hX.size()), out(hX.size());
  int total_bad = 0;
  for (int gather = 0; gather < 2; ++gather) {
  if (gather) hipLaunchKernelGGL((chain_kernel<true, true>), dim3(256), dim3(256), sizeof(Smem), 0, W, b, X, Y, n_tiles, inner, idx);
  else hipLaunchKernelGGL((chain_kernel<true, false>), dim3(256), dim3(256), sizeof(Smem), 0, W, b, X, Y, n_tiles, inner, idx);   // one wave per SIMD, fenced
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), Y, ref.size() * 4, hipMemcpyDeviceToHost);
  double cs = 0;
  for (float v : ref) cs += v;
  printf("== %s: reference (fenced, 1 wave/SIMD): checksum %.6f\n", gather ? "chain with gathered rows in flight" : "bare chain", cs);
  for (int grid : {256, 512, 768, 1024}) {
    for (int fenced = 1; fenced >= 0; --fenced) {
      int bad_runs = 0;
      long long bad_tiles = 0;
      for (int it = 0; it < repeats; ++it) {
        hipMemset(Y, 0, hX.size() * 4);
        if (fenced && gather) hipLaunchKernelGGL((chain_kernel<true, true>), dim3(grid), dim3(256), sizeof(Smem), 0, W, b, X, Y, n_tiles, inner, idx);
        else if (fenced) hipLaunchKernelGGL((chain_kernel<true, false>), dim3(grid), dim3(256), sizeof(Smem), 0, W, b, X, Y, n_tiles, inner, idx);
        else if (gather) hipLaunchKernelGGL((chain_kernel<false, true>), dim3(grid), dim3(256), sizeof(Smem), 0, W, b, X, Y, n_tiles, inner, idx);
        else hipLaunchKernelGGL((chain_kernel<false, false>), dim3(grid), dim3(256), sizeof(Smem), 0, W, b, X, Y, n_tiles, inner, idx);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), Y, out.size() * 4, hipMemcpyDeviceToHost);
        long long bt = 0;
        for (int t = 0; t < n_tiles; ++t)
          if (memcmp(out.data() + (size_t)t * 2048, ref.data() + (size_t)t * 2048, 2048 * 4) != 0) ++bt;
        bad_tiles += bt;
        bad_runs += bt > 0;
      }
      printf("grid %3d (%d wave%s/SIMD) %-8s: %d of %d runs differ from the reference, %lld of %lld tiles\n", grid, grid / 256,
             grid == 256 ? "" : "s (up to)", fenced ? "fenced" : "unfenced", bad_runs, repeats, bad_tiles, (long long)repeats * n_tiles);
      total_bad += bad_runs;
    }
  }
  }
  printf("runs with any difference: %d\n", total_bad);
  hipFree(idx);
  hipFree(W); hipFree(b); hipFree(X); hipFree(Y);
  return 0;
}
