// Issue cost of the VALU instructions a bf16x3 operand split is made of, on gfx950: cycles per
// wave64 instruction with 1 and 2 waves per SIMD (s_memtime around 64 x 32 independent instructions).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_bench tools/valu_rate_bench.hip && tools/valu_rate_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X

#define KERNEL(NAME, ASM)                                                                   \
  __global__ void NAME(float* out, long long* cyc, int iters) {                             \
    float a0 = threadIdx.x * 1.5f + 1.0f, a1 = a0 + 3.f, a2 = a0 * 0.37f, a3 = a1 * 1.3f;    \
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;                                            \
    const long long t0 = __builtin_readcyclecounter();                                      \
    for (int i = 0; i < iters; ++i) {                                                        \
      asm volatile(REP32(ASM) : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); \
    }                                                                                        \
    const long long t1 = __builtin_readcyclecounter();                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = b0 + b1 + b2 + b3;                          \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
  }

// four independent destinations per group so that no instruction waits for the previous one
KERNEL(k_add, "v_add_f32 %0, %4, %5\n v_add_f32 %1, %5, %6\n v_add_f32 %2, %6, %7\n v_add_f32 %3, %7, %4\n")
KERNEL(k_sub, "v_sub_f32 %0, %4, %5\n v_sub_f32 %1, %5, %6\n v_sub_f32 %2, %6, %7\n v_sub_f32 %3, %7, %4\n")
KERNEL(k_and, "v_and_b32 %0, %4, %5\n v_and_b32 %1, %5, %6\n v_and_b32 %2, %6, %7\n v_and_b32 %3, %7, %4\n")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 16, %4\n v_lshlrev_b32 %1, 16, %5\n v_lshlrev_b32 %2, 16, %6\n v_lshlrev_b32 %3, 16, %7\n")
KERNEL(k_perm, "v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %5, %6, %7\n v_perm_b32 %2, %6, %7, %4\n v_perm_b32 %3, %7, %4, %5\n")
KERNEL(k_cvt, "v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n")
KERNEL(k_fma, "v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %5, %6, %7\n v_fma_f32 %2, %6, %7, %4\n v_fma_f32 %3, %7, %4, %5\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %4, 16, 16\n v_bfe_u32 %1, %5, 16, 16\n v_bfe_u32 %2, %6, 16, 16\n v_bfe_u32 %3, %7, 16, 16\n")

__global__ void k_pkadd(float* out, long long* cyc, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {threadIdx.x * 1.5f + 1.0f, 2.f}, a1 = a0 + 3.f, b0 = {0.f, 0.f}, b1 = b0, b2 = b0, b3 = b0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP32("v_pk_add_f32 %0, %4, %5\n v_pk_add_f32 %1, %5, %4\n v_pk_add_f32 %2, %4, %4\n v_pk_add_f32 %3, %5, %5\n")
                 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "v"(a1));
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = b0[0] + b1[1] + b2[0] + b3[1];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// matrix pipe next to the VALU: 4 independent accumulators; `NV` plain VALU per MFMA from the SAME wave
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
template <int NV>
__global__ void k_mix(float* out, long long* cyc, int iters) {
  bf16x8_ a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.01f + e); b[e] = (__bf16)(1.0f - e * 0.1f); }
  f32x4_ c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 + 7, v3 = v0 ^ 5, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      if (NV >= 1) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w0) : "v"(v0), "v"(v1));
      if (NV >= 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w1) : "v"(v1), "v"(v2));
      if (NV >= 3) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w2) : "v"(v2), "v"(v3));
      if (NV >= 4) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w3) : "v"(v3), "v"(v0));
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      if (NV >= 1) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w0) : "v"(v0), "v"(v1));
      if (NV >= 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w1) : "v"(v1), "v"(v2));
      if (NV >= 3) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w2) : "v"(v2), "v"(v3));
      if (NV >= 4) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w3) : "v"(v3), "v"(v0));
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      if (NV >= 1) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w0) : "v"(v0), "v"(v1));
      if (NV >= 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w1) : "v"(v1), "v"(v2));
      if (NV >= 3) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w2) : "v"(v2), "v"(v3));
      if (NV >= 4) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w3) : "v"(v3), "v"(v0));
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      if (NV >= 1) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w0) : "v"(v0), "v"(v1));
      if (NV >= 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w1) : "v"(v1), "v"(v2));
      if (NV >= 3) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w2) : "v"(v2), "v"(v3));
      if (NV >= 4) asm volatile("v_and_b32 %0, %1, %2" : "=v"(w3) : "v"(v3), "v"(v0));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (float)(w0 + w1 + w2 + w3);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// waves 0-3 of a 512-thread block only issue MFMAs, waves 4-7 (their SIMD partners) only VALU
__global__ void k_pair(float* out, long long* cyc, int iters) {
  bf16x8_ a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.01f + e); b[e] = (__bf16)(1.0f - e * 0.1f); }
  f32x4_ c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned v0 = threadIdx.x, v1 = v0 * 3, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  const bool mf = threadIdx.x < 256;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (mf) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
    } else {
      asm volatile(REP32("v_and_b32 %0, %4, %5\n v_and_b32 %1, %5, %4\n v_and_b32 %2, %4, %4\n v_and_b32 %3, %5, %5\n")
                   : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(v0), "v"(v1));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (float)(w0 + w1 + w2 + w3);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int waves_per_simd) {
  float* out; long long* cyc;
  const int threads = 256 * waves_per_simd, iters = 64;
  hipMalloc(&out, 256 * threads * 4); hipMalloc(&cyc, 256 * 8 * 8);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
  // s_memtime ticks at 100 MHz on this stack?  report raw ticks per instruction and let the ratio to
  // v_add_f32 speak
  printf("%-10s %d wave(s)/SIMD: %8.3f ticks per wave-instruction (slowest wave)\n", name, waves_per_simd,
         mx / (iters * 32.0 * 4.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run("v_add_f32", k_add, w); run("v_sub_f32", k_sub, w); run("v_fma_f32", k_fma, w);
    run("v_and_b32", k_and, w); run("v_lshlrev", k_lshl, w); run("v_bfe_u32", k_bfe, w);
    run("v_perm_b32", k_perm, w); run("v_cvt_pk_bf16", k_cvt, w); run("v_pk_add_f32", k_pkadd, w);
  }
  // MFMA (32 per iteration) with 0 .. 4 VALU behind each, one wave per SIMD: ticks per iteration of 32 groups
  for (int w = 1; w <= 2; ++w) {
    float* out; long long* cyc;
    const int threads = 256 * w, iters = 64;
    hipMalloc(&out, 256 * threads * 4); hipMalloc(&cyc, 256 * 8 * 8);
    auto rep = [&](const char* name, double groups) {
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
      double mx = 0, mn = 1e30;
      for (int i = 0; i < threads / 64; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
      printf("%-28s %d wave(s)/SIMD: %8.2f .. %8.2f ticks per MFMA group\n", name, w, mn / (iters * groups), mx / (iters * groups));
    };
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_mix<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    rep("mfma only", 32);
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_mix<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    rep("mfma + 1 valu", 32);
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_mix<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    rep("mfma + 2 valu", 32);
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_mix<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    rep("mfma + 4 valu", 32);
    if (w == 2) {
      for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      rep("pair: 32 mfma || 128 valu", 1);
    }
    hipFree(out); hipFree(cyc);
  }
  return 0;
}
