// What a dependent phase change costs on MI355X for the shape of the DQN online chain: a kernel
// boundary (two launches per round, what pa_dqn_learn does) against ONE persistent kernel with a
// grid barrier between the phases (flat counter, and XCD-hierarchical as MI355X_MICROARCH.md
// recommends).  DESIGN.md §3.3 argued the megakernel away from a 2.2 us boundary; this measures it.
//
// A round has the chain's two all-to-all dependencies:
//   phase A ("row pass"):   NA workgroups; each reads EVERYTHING phase B wrote last round (the
//                           "weights": NB x WB bytes), then writes its own slab of SA bytes
//   phase B ("weight grad"): NB workgroups; each reads a 1/8 slice of EVERY phase-A slab, then
//                           writes its own WB bytes
// Arithmetic is a checksum, so both variants must print the same final value (a stale read shows).
//
//   hipcc -O3 --offload-arch=gfx950 tools/chain_boundary_bench.hip -o tools/chain_boundary_bench
//   tools/chain_boundary_bench [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NA = 64, NB = 113, NWG = 128;      // chain shape: 64 row workgroups, 113 dW workgroups, 128 CUs
constexpr int SA = 48 * 1024 / 4;                // floats per phase-A slab (h1, h2, dZ1 of 16 rows)
constexpr int WB = 4 * 1024 / 4;                 // floats per phase-B output (one 32 x 32 weight tile)

struct Bufs {
  float* slabA;      // [NA][SA]
  float* outB;       // [NB][WB]
  unsigned* bar;     // [0] flat counter, [1] generation, [8..15] per-XCC counters, [16] top, [32..39] per-XCC population
  float* result;
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
// WT: 16-byte write-through stores (sc0 sc1): the bytes leave the XCD's L2 while the phase runs, so
// the end-of-kernel / release write-back finds nothing dirty
template <bool WT>
__device__ __forceinline__ void store4(float4* p, float a, float b, float c, float d) {
  if constexpr (WT) {
    const f32x4_t v = {a, b, c, d};
#ifdef NOP_AFTER
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 4" ::"v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#endif
  } else {
    *p = make_float4(a, b, c, d);
  }
}

template <bool WT>
__device__ __forceinline__ void phase_a(const Bufs& b, int wg, int round, int tid) {
  // read all of phase B's output (NB * WB floats), 16 bytes per lane and load
  float acc = 0.f;
  const float4* src = reinterpret_cast<const float4*>(b.outB);
  for (int i = tid; i < NB * WB / 4; i += 512) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  // reduce over the workgroup so every element written depends on everything read
  __shared__ float red[512];
  red[tid] = acc;
  __syncthreads();
  for (int w = 256; w >= 1; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const float s = red[0] * 1e-6f;
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(b.slabA + (size_t)wg * SA);
  for (int i = tid; i < SA / 4; i += 512) {
    const float v = s + (float)((i + wg + round) & 255) * 0.001f;
    store4<WT>(dst + i, v, v + 1.f, v + 2.f, v + 3.f);
  }
}

template <bool WT>
__device__ __forceinline__ void phase_b(const Bufs& b, int wg, int round, int tid) {
  // a 1/8 slice of every phase-A slab (a dW tile reads 2 of 16 column blocks of every row)
  float acc = 0.f;
  const int part = wg & 7;
  for (int s = 0; s < NA; ++s) {
    const float4* src = reinterpret_cast<const float4*>(b.slabA + (size_t)s * SA + part * (SA / 8));
    for (int i = tid; i < SA / 32; i += 512) {
      const float4 v = src[i];
      acc += v.x + v.y + v.z + v.w;
    }
  }
  __shared__ float red[512];
  red[tid] = acc;
  __syncthreads();
  for (int w = 256; w >= 1; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const float s = red[0] * 1e-6f;
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(b.outB + (size_t)wg * WB);
  for (int i = tid; i < WB / 4; i += 512) {
    const float v = s + (float)((i + wg) & 63) * 0.01f;
    store4<WT>(dst + i, v, v, v, v);
  }
}

template <bool WT>
__global__ __launch_bounds__(512) void kernel_a(Bufs b, int round) { phase_a<WT>(b, blockIdx.x, round, threadIdx.x); }
template <bool WT>
__global__ __launch_bounds__(512) void kernel_b(Bufs b, int round) { phase_b<WT>(b, blockIdx.x, round, threadIdx.x); }
__global__ void empty_kernel() {}

// ---- grid barriers (all NWG workgroups resident: one per CU on 128 CUs) -------------------------
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void barrier_flat(unsigned* bar, unsigned& gen, int tid) {
#ifndef NO_WAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the asm stores are invisible to the compiler's counters)
#endif
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned target = gen + 1;
    if (atomicAdd(&bar[0], 1u) == (unsigned)NWG * target - 1u)
      __hip_atomic_store(&bar[1], target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else
      while (ld_relaxed(&bar[1]) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  gen += 1;
  __syncthreads();
}
__device__ __forceinline__ void barrier_xcd(unsigned* bar, unsigned& gen, int xcc, int tid) {
#ifndef NO_WAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  if (tid == 0) {
    const unsigned target = gen + 1;
    const unsigned pop = ld_relaxed(&bar[32 + xcc]);
    if (atomicAdd(&bar[8 + xcc], 1u) == pop * target - 1u) {
      // XCD leader: one release for the XCD's L2, then the top counter
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (atomicAdd(&bar[16], 1u) == 8u * target - 1u)
        __hip_atomic_store(&bar[1], target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (ld_relaxed(&bar[1]) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  gen += 1;
  __syncthreads();
}

// census: how many of the persistent kernel's workgroups sit on each XCC (b % 8 in practice)
__global__ void census_kernel(unsigned* bar) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;  // HW_REG_XCC_ID
    atomicAdd(&bar[32 + xcc], 1u);
  }
}

// mode 0: no synchronisation between phases (WRONG results; the cost of the phases themselves)
// mode 1: flat barrier   mode 2: XCD-hierarchical barrier
template <int MODE, bool WT>
__global__ __launch_bounds__(512) void persistent_kernel(Bufs b, int rounds) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7;
  unsigned gen = 0;
  for (int r = 0; r < rounds; ++r) {
    if (wg < NA) phase_a<WT>(b, wg, r, tid);
    if (MODE == 1) barrier_flat(b.bar, gen, tid);
    if (MODE == 2) barrier_xcd(b.bar, gen, xcc, tid);
    if (wg < NB) phase_b<WT>(b, wg, r, tid);
    if (MODE == 1) barrier_flat(b.bar, gen, tid);
    if (MODE == 2) barrier_xcd(b.bar, gen, xcc, tid);
  }
}

__global__ void checksum_kernel(Bufs b) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < NB * WB; i += 256) acc += b.outB[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) b.result[0] = red[0];
}

static void reset(const Bufs& b) {
  CK(hipMemset(b.slabA, 0, (size_t)NA * SA * 4));
  CK(hipMemset(b.outB, 0, (size_t)NB * WB * 4));
  CK(hipMemset(b.bar, 0, 32 * 4));   // counters and generation; the census (bar[32..]) stays
}
static float checksum(const Bufs& b) {
  hipLaunchKernelGGL(checksum_kernel, dim3(1), dim3(256), 0, 0, b);
  float r;
  CK(hipMemcpy(&r, b.result, 4, hipMemcpyDeviceToHost));
  return r;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  Bufs b;
  CK(hipMalloc((void**)&b.slabA, (size_t)NA * SA * 4));
  CK(hipMalloc((void**)&b.outB, (size_t)NB * WB * 4));
  CK(hipMalloc((void**)&b.bar, 64 * 4));
  CK(hipMalloc((void**)&b.result, 4));
  CK(hipMemset(b.bar, 0, 64 * 4));
  hipLaunchKernelGGL(census_kernel, dim3(NWG), dim3(64), 0, 0, b.bar);
  CK(hipDeviceSynchronize());
  unsigned pop[8];
  CK(hipMemcpy(pop, b.bar + 32, 32, hipMemcpyDeviceToHost));
  printf("workgroups per XCC in a %d-workgroup grid:", NWG);
  for (int i = 0; i < 8; ++i) printf(" %u", pop[i]);
  printf("\n");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  // back-to-back dependent EMPTY launches: the floor of a boundary with nothing to write back
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 2 * rounds; ++r) hipLaunchKernelGGL(empty_kernel, dim3(NB), dim3(512), 0, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("empty launch, back to back            : %7.2f us each\n", 1e3f * ms / (2 * rounds));
  }
  for (int wt = 0; wt < 2; ++wt) {
    printf("-- %s stores\n", wt ? "write-through (sc0 sc1)" : "plain");
    for (int rep = 0; rep < 2; ++rep) {   // second repetition is the reported one (warm clocks)
      // ---- two launches per round
      reset(b);
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < rounds; ++r) {
        if (wt) {
          hipLaunchKernelGGL(kernel_a<true>, dim3(NA), dim3(512), 0, 0, b, r);
          hipLaunchKernelGGL(kernel_b<true>, dim3(NB), dim3(512), 0, 0, b, r);
        } else {
          hipLaunchKernelGGL(kernel_a<false>, dim3(NA), dim3(512), 0, 0, b, r);
          hipLaunchKernelGGL(kernel_b<false>, dim3(NB), dim3(512), 0, 0, b, r);
        }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const float ref = checksum(b);
      if (rep) printf("two launches per round                : %7.2f us/round   checksum %.6e\n", 1e3f * ms / rounds, ref);
      // ---- persistent variants
      for (int mode = 0; mode < 3; ++mode) {
        reset(b);
        CK(hipEventRecord(e0, 0));
        if (mode == 0 && !wt) hipLaunchKernelGGL((persistent_kernel<0, false>), dim3(NWG), dim3(512), 0, 0, b, rounds);
        if (mode == 1 && !wt) hipLaunchKernelGGL((persistent_kernel<1, false>), dim3(NWG), dim3(512), 0, 0, b, rounds);
        if (mode == 2 && !wt) hipLaunchKernelGGL((persistent_kernel<2, false>), dim3(NWG), dim3(512), 0, 0, b, rounds);
        if (mode == 0 && wt) hipLaunchKernelGGL((persistent_kernel<0, true>), dim3(NWG), dim3(512), 0, 0, b, rounds);
        if (mode == 1 && wt) hipLaunchKernelGGL((persistent_kernel<1, true>), dim3(NWG), dim3(512), 0, 0, b, rounds);
        if (mode == 2 && wt) hipLaunchKernelGGL((persistent_kernel<2, true>), dim3(NWG), dim3(512), 0, 0, b, rounds);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float cs = checksum(b);
        const char* names[3] = {"persistent, NO barrier (wrong results)", "persistent, flat barrier              ",
                                "persistent, XCD-hierarchical barrier  "};
        if (rep)
          printf("%s: %7.2f us/round   checksum %.6e%s\n", names[mode], 1e3f * ms / rounds, cs,
                 mode && cs != ref ? "   MISMATCH" : "");
      }
    }
  }
  return 0;
}
