// Where the 13 us of the learn() prologue's index sampler go (round 6): the shipped
// sample_indices_block (sampler.hpp) with wall-clock stamps at its phase boundaries, and the
// whole-kernel duration by HIP events, for R workgroups of B = 1024 positions over n = 1 M.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. tools/sampler_bench.hip -o tools/sampler_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../pearl_amd/csrc/sampler.hpp"

using namespace pa;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(SAMPLE_THREADS) void shipped_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long table[];
  sample_indices_block(a, (int)blockIdx.x, table);
}

// the same body with stamps: [block][8] = {start, table cleared, round-0 proposals in, round-0 resolved,
//                                         loop done, iterations}
__global__ __launch_bounds__(SAMPLE_THREADS) void stamped_kernel(SampleArgs a, long long* st) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long table[];
  long long* my = st + (size_t)blockIdx.x * 8;
  const int tid = threadIdx.x;
  if (tid == 0) my[0] = wall_clock64();
  int64_t* __restrict__ idx_out = a.idx_out + (int64_t)blockIdx.x * a.B;
  const uint32_t n = a.n, seed_lo = a.seed_lo, seed_hi = a.seed_hi;
  const int B = a.B, hs = a.hs;
  const uint64_t offset = a.offset0 + blockIdx.x;
  const uint32_t off_lo = (uint32_t)offset, off_hi = (uint32_t)(offset >> 32);
  int* pending_p = reinterpret_cast<int*>(table + hs);
  for (int i = tid; i < hs; i += SAMPLE_THREADS) table[i] = EMPTY_ENTRY;
  bool done = tid >= B;
  uint32_t val = 0;
  const uint32_t mask = (uint32_t)hs - 1;
  __syncthreads();
  if (tid == 0) my[1] = wall_clock64();
  uint32_t t = 0;
  for (; t < 65535u; ++t) {
    if (!done) {
      uint32_t w[4];
      philox4x32_10(tid, t, off_lo, off_hi, seed_lo, seed_hi, w);
      const uint32_t v = bounded_draw(w, n);
      val = v;
      const unsigned long long entry = ((unsigned long long)v << 32) | (t << 16) | tid;
      uint32_t h = (v * 0x9E3779B1u) & mask;
      for (int probe = 0; probe < hs; ++probe) {
        const unsigned long long old = atomicCAS(&table[h], EMPTY_ENTRY, entry);
        if (old == EMPTY_ENTRY) break;
        if ((uint32_t)(old >> 32) == v) { atomicMin(&table[h], entry); break; }
        h = (h + 1) & mask;
      }
    }
    if (tid == 0) *pending_p = 0;
    __syncthreads();
    if (tid == 0 && t == 0) my[2] = wall_clock64();
    bool any = false;
    if (!done) {
      const uint32_t v = val;
      uint32_t h = (v * 0x9E3779B1u) & mask;
      for (int probe = 0; probe < hs && (uint32_t)(table[h] >> 32) != v; ++probe) h = (h + 1) & mask;
      const unsigned long long e = table[h];
      if ((uint32_t)(e & 0xFFFFu) == (uint32_t)tid && (uint32_t)((e >> 16) & 0xFFFFu) == t) {
        done = true;
        idx_out[tid] = (int64_t)v;
      } else any = true;
    }
    if (any) atomicOr(pending_p, 1);
    __syncthreads();
    const int p = *pending_p;
    __syncthreads();
    if (tid == 0 && t == 0) my[3] = wall_clock64();
    if (!p) break;
  }
  if (tid == 0) { my[4] = wall_clock64(); my[5] = t + 1; }
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 20;
  const int B = 1024;
  int64_t* idx; long long* st;
  CHECK(hipMalloc(&idx, (size_t)R * B * 8)); CHECK(hipMalloc(&st, (size_t)R * 8 * 8));
  SampleArgs a; a.idx_out = idx; a.n = 1000000u; a.B = B; a.hs = sample_table_slots(B);
  a.seed_lo = 123u; a.seed_hi = 456u; a.offset0 = 0;
  const size_t smem = sample_smem_bytes(a.hs);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int v = 0; v < 2; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 30; ++rep) {
      a.offset0 = 1000 * rep;
      CHECK(hipEventRecord(e0, 0));
      if (v == 0) hipLaunchKernelGGL(shipped_kernel, dim3(R), dim3(SAMPLE_THREADS), smem, 0, a);
      else hipLaunchKernelGGL(stamped_kernel, dim3(R), dim3(SAMPLE_THREADS), smem, 0, a, st);
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%s: %d workgroups, best of 30 launches %.2f us (event to event)\n", v == 0 ? "shipped" : "stamped", R, best * 1e3f);
  }
  std::vector<long long> h((size_t)R * 8);
  CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  long long first = h[0];
  for (int b = 0; b < R; ++b) if (h[b * 8] < first) first = h[b * 8];
  printf("block: start | table cleared | round-0 proposals in | round-0 resolved | done   (us since the first block's start; 100 MHz clock) iterations\n");
  for (int b = 0; b < R && b < 6; ++b)
    printf("  %2d: %6.2f | %6.2f | %6.2f | %6.2f | %6.2f   %lld\n", b, (h[b * 8] - first) * 0.01, (h[b * 8 + 1] - first) * 0.01,
           (h[b * 8 + 2] - first) * 0.01, (h[b * 8 + 3] - first) * 0.01, (h[b * 8 + 4] - first) * 0.01, h[b * 8 + 5]);
  return 0;
}
