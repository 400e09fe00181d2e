// Microbenchmark: fp32 MFMA issue rate for the shapes the learner kernels use.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_bench.hip -o tools/mfma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a, float b) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Random operands (different per lane and per k-step): the chip clocks to its power budget, and
// toggling operands draw more than constants do.
template <int NACC>
__global__ __launch_bounds__(512) void k32r(float* out, int iters, const float* in) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float av[8], bv[8];
  for (int u = 0; u < 8; ++u) {
    av[u] = in[(threadIdx.x * 8 + u) & 4095];
    bv[u] = in[(threadIdx.x * 8 + u + 2048 + blockIdx.x) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

void run_random(const char* name, int threads, int blocks, int iters) {
  float *out, *in;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&in, 4096 * 4);
  float h[4096];
  unsigned s = 12345;
  for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 16777216.0f) * 2.0f - 1.0f; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k32r<2>, dim3(blocks), dim3(threads), 0, 0, out, 10, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k32r<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)iters * 8 * 2 * ((double)blocks * threads / 64) * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
  printf("%-36s blocks=%d thr=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, iters, ms, tf);
  hipFree(out); hipFree(in);
}

template <typename K>
void run(const char* name, K kern, int threads, int blocks, int nacc, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * 8 * nacc;
  const double waves = (double)blocks * threads / 64;
  const double tf = mfma_per_wave * waves * flop_per_mfma / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD at 2.4 GHz: waves per SIMD = threads/64/4 (1 block per CU)
  const double wps = threads / 64.0 / 4.0;
  printf("%-28s blocks=%d thr=%d: %.3f ms  %.1f TFLOP/s  -> %.1f cyc/MFMA/SIMD @2.4GHz\n", name, blocks,
         threads, ms, tf, ms * 1e-3 * 2.4e9 / (mfma_per_wave * wps));
  hipFree(out);
}

int main() {
  run("16x16x4 acc=1 512thr", k16<1>, 512, 256, 1, 2.0 * 16 * 16 * 4);
  run("16x16x4 acc=2 512thr", k16<2>, 512, 256, 2, 2.0 * 16 * 16 * 4);
  run("16x16x4 acc=4 512thr", k16<4>, 512, 256, 4, 2.0 * 16 * 16 * 4);
  run("16x16x4 acc=2 256thr", k16<2>, 256, 256, 2, 2.0 * 16 * 16 * 4);
  run("16x16x4 acc=2 512thr 64blk", k16<2>, 512, 64, 2, 2.0 * 16 * 16 * 4);
  run("32x32x2 acc=1 512thr", k32<1>, 512, 256, 1, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=2 512thr", k32<2>, 512, 256, 2, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=2 256thr", k32<2>, 256, 256, 2, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=2 512thr 64blk", k32<2>, 512, 64, 2, 2.0 * 32 * 32 * 2);
  run_random("32x32x2 acc=2 RANDOM operands", 512, 256, 2000);
  run_random("32x32x2 acc=2 RANDOM operands", 512, 256, 40000);
  run_random("32x32x2 acc=2 RANDOM operands 2/CU", 512, 512, 40000);
  run("32x32x2 acc=2 512thr (const, again)", k32<2>, 512, 256, 2, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=2 512thr const 2/CU", k32<2>, 512, 512, 2, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=1 512thr const 2/CU", k32<1>, 512, 512, 1, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=2 256thr const 2/CU", k32<2>, 256, 512, 2, 2.0 * 32 * 32 * 2);
  run("32x32x2 acc=2 256thr const 4/CU", k32<2>, 256, 1024, 2, 2.0 * 32 * 32 * 2);
  run("16x16x4 acc=2 512thr const 2/CU", k16<2>, 512, 512, 2, 2.0 * 16 * 16 * 4);
  run_random("RANDOM 512thr 1/CU long", 512, 256, 100000);
  run_random("RANDOM 512thr 2/CU short", 512, 512, 10000);
  run_random("RANDOM 512thr 2/CU mid", 512, 512, 20000);
  run_random("RANDOM 512thr 1/CU long again", 512, 256, 200000);
  run_random("32x32x2 acc=2 RANDOM 256thr 2/CU", 256, 512, 40000);
  run_random("32x32x2 acc=2 RANDOM 256thr 1/CU", 256, 256, 40000);
  return 0;
}
