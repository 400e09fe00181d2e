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
  return 0;
}
