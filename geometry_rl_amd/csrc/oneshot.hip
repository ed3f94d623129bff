// One-shot all-reduce of the data-parallel step's ONE bandwidth-relevant collective (the actor's 0.54 MB gradient
// slice with the ranks' loss records in front of it) over directly mapped peer buffers -- hipIpc handles across processes on an xGMI node,
// plain device pointers in the single-process test.  The reference has no distributed code at all; the semantics to
// keep are examples/torchrl/train.py:304-316: every replica applies Adam to the SAME summed gradient.
//
// Ring all-reduce over 8 GPUs is 14 dependent hops of 68 KB: latency, not bytes.  Here every rank
//   1. writes its contribution to chunk p straight into peer p's staging row (one xGMI traversal, all seven links at once), raises a flag;
//   2. waits for the W contributions to ITS chunk, sums them in RANK ORDER (bitwise the same on every run and -- because each chunk is
//      reduced by exactly one rank -- identical on all ranks), writes the result into every peer's buffer (second traversal), raises a flag;
//   3. waits for the W result chunks.
// Two link traversals + two flag hand-offs.  Flags carry a sequence number that only grows: nothing is ever reset, a kernel of call k
// cannot be confused by a flag of call k - 1.  Payload stores are followed by a system-scope release before the flag; the reader
// acquires at system scope behind the flag (peer memory is not coherent through this GPU's L2).  Every wait is BOUNDED: on a timeout the
// kernel writes a status word and leaves instead of hanging the device.
#include "grl_common.h"

namespace {

constexpr int OS_MAX_WORLD = 8;
constexpr int OS_BLOCKS = 32;      // workgroups per rank: all ranks' kernels must be resident together (the single-process test: W x 32)
struct OneShotPeers {
  float* buf[OS_MAX_WORLD];        // rank p's payload [n]
  float* stage[OS_MAX_WORLD];      // rank p's staging rows [W][chunk]
  unsigned* flags[OS_MAX_WORLD];   // rank p's flags [2][W][OS_BLOCKS]
};

GRL_DEVINL bool os_wait(const unsigned* flag, unsigned seq, unsigned long long deadline) {
  // (seq only grows: "reached" is >=, so a flag that already moved on to a later call can never strand this one)
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
    if (wall_clock64() > deadline) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

// rank < 0: ALL ranks in this one launch, rank = blockIdx.y (the single-process test of the protocol: W x OS_BLOCKS workgroups that are
// resident together by construction -- W separate launches on W streams of one process may share a hardware queue and serialise)
__global__ __launch_bounds__(256) void oneshot_allreduce_kernel(OneShotPeers P, int rank_, int W, int n, int chunk, unsigned seq,
                                                                unsigned long long timeout_ticks, int* __restrict__ status) {
  __shared__ int ok;
  const int rank = rank_ >= 0 ? rank_ : (int)blockIdx.y;
  if (rank_ < 0) status += rank;          // (one status word per stand-in rank)
  const int b = blockIdx.x;
  const unsigned long long deadline = wall_clock64() + timeout_ticks;
  // this workgroup's share of every chunk: [lo, hi) in units of four floats
  const int quads = chunk >> 2, per = (quads + OS_BLOCKS - 1) / OS_BLOCKS;
  const int lo = b * per, hi = lo + per < quads ? lo + per : quads;
  if (threadIdx.x == 0) ok = 1;
  // ---- 1. my contribution to every chunk -> the owner's staging row `rank`
  for (int k = 1; k <= W; ++k) {
    const int p = (rank + k) % W;                                  // (start at the next rank: the W ranks load W different links)
    const float4* src = reinterpret_cast<const float4*>(P.buf[rank] + (size_t)p * chunk);
    float4* dst = reinterpret_cast<float4*>(P.stage[p] + (size_t)rank * chunk);
    const int qmax = (n - p * chunk + 3) >> 2;                     // quads of chunk p that exist (the last chunk may be short)
    for (int i = lo + (int)threadIdx.x; i < hi && i < qmax; i += 256) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < W)
    __hip_atomic_store(P.flags[threadIdx.x] + (0 * W + rank) * OS_BLOCKS + b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // ---- 2. my chunk: wait for the W contributions, sum in rank order, hand the result to everybody
  if (threadIdx.x < W && !os_wait(P.flags[rank] + (0 * W + threadIdx.x) * OS_BLOCKS + b, seq, deadline)) ok = 0;
  __syncthreads();
  if (!ok) { if (threadIdx.x == 0) atomicExch(status, 1); return; }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                    // (system scope: drop what L2 may hold of the staging rows)
  {
    const int qmax = (n - rank * chunk + 3) >> 2;
    for (int i = lo + (int)threadIdx.x; i < hi && i < qmax; i += 256) {
      float4 s = reinterpret_cast<const float4*>(P.stage[rank])[i];                       // rank 0's contribution
      for (int p = 1; p < W; ++p) s = f4_add(s, reinterpret_cast<const float4*>(P.stage[rank] + (size_t)p * chunk)[i]);
      for (int p = 0; p < W; ++p) reinterpret_cast<float4*>(P.buf[p] + (size_t)rank * chunk)[i] = s;
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < W)
    __hip_atomic_store(P.flags[threadIdx.x] + (1 * W + rank) * OS_BLOCKS + b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // ---- 3. every chunk of MY buffer has arrived
  if (threadIdx.x < W && !os_wait(P.flags[rank] + (1 * W + threadIdx.x) * OS_BLOCKS + b, seq, deadline)) ok = 0;
  __syncthreads();
  if (!ok && threadIdx.x == 0) atomicExch(status, 2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

}  // namespace

extern "C" {

// floats per chunk (a multiple of 4): the payload of n floats is cut into `world` chunks, chunk r is reduced by rank r
int grl_oneshot_chunk_floats(int n, int world) {
  if (n <= 0 || world < 1 || world > OS_MAX_WORLD) return -1;
  return (((n + world - 1) / world) + 3) & ~3;
}
// sizes of a rank's staging area (floats) and flag area (32-bit words, zero-initialised ONCE by the caller)
int grl_oneshot_stage_floats(int n, int world) { const int c = grl_oneshot_chunk_floats(n, world); return c < 0 ? -1 : c * world; }
int grl_oneshot_flag_words(int world) { return world < 1 || world > OS_MAX_WORLD ? -1 : 2 * world * OS_BLOCKS; }
int grl_oneshot_blocks(void) { return OS_BLOCKS; }

// One rank's side of the all-reduce: bufs / stages / flags are HOST arrays of `world` device pointers (entry p = rank p's areas, mapped
// into this process: hipIpcOpenMemHandle for peers); bufs[p] holds the n payload floats (n a multiple of 4; nothing past n is touched).  seq: the call's sequence number, the same on every rank, strictly increasing from 1.  status: device int, written only on
// a timeout (1: a contribution did not arrive, 2: a result chunk did not arrive) -- the caller checks it when it next synchronises.
// Every rank of the group must enqueue the call with the same (n, seq); the kernels wait for each other ON THE DEVICE.
int grl_oneshot_allreduce(float* const* bufs, float* const* stages, unsigned* const* flags, int rank, int world, int n, unsigned seq,
                          int timeout_ms, int* status, hipStream_t stream) {
  if (world < 1 || world > OS_MAX_WORLD || rank < 0 || rank >= world || n <= 0 || (n & 3) || seq == 0 || !status) return -2;   // (n: whole quads)
  OneShotPeers P{};
  for (int p = 0; p < world; ++p) {
    if (!bufs[p] || !stages[p] || !flags[p]) return -3;
    if ((reinterpret_cast<size_t>(bufs[p]) | reinterpret_cast<size_t>(stages[p])) & 15) return -4;
    P.buf[p] = bufs[p]; P.stage[p] = stages[p]; P.flags[p] = flags[p];
  }
  const unsigned long long ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ull;   // wall_clock64: 100 MHz
  hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3(OS_BLOCKS), dim3(256), 0, stream, P, rank, world, n,
                     grl_oneshot_chunk_floats(n, world), seq, ticks, status);
  GRL_CHECK_LAUNCH();
  return 0;
}

// The same protocol with all `world` stand-in ranks in ONE launch (rank = blockIdx.y): the single-process test on one GPU.  status: world ints.
int grl_oneshot_allreduce_local(float* const* bufs, float* const* stages, unsigned* const* flags, int world, int n, unsigned seq,
                                int timeout_ms, int* status, hipStream_t stream) {
  if (world < 1 || world > OS_MAX_WORLD || n <= 0 || (n & 3) || seq == 0 || !status) return -2;
  OneShotPeers P{};
  for (int p = 0; p < world; ++p) {
    if (!bufs[p] || !stages[p] || !flags[p]) return -3;
    P.buf[p] = bufs[p]; P.stage[p] = stages[p]; P.flags[p] = flags[p];
  }
  const unsigned long long ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ull;
  hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3(OS_BLOCKS, world), dim3(256), 0, stream, P, -1, world, n,
                     grl_oneshot_chunk_floats(n, world), seq, ticks, status);
  GRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
