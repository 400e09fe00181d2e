// What can run beside an fp32 MFMA stream on the same SIMD?  One 1024-thread workgroup per CU:
// waves 0-7 (two per SIMD) run a v_mfma_f32_32x32x2_f32 loop, waves 8-15 run a partner loop
// (nothing / VALU fma / global loads / LDS traffic).  Both report their own wall-clock duration.
//   hipcc --offload-arch=gfx950 -O3 tools/corun_bench.hip -o tools/corun_bench && ./tools/corun_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(1024) void corun(float* out, const float* in, long long* t, int mfma_iters,
                                              int mode, int partner_iters, int partner_prio) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float a = in[tid & 1023], b = in[(tid + 512) & 1023];
  for (int i = tid; i < 8192; i += 1024) lds[i] = a;
  __syncthreads();
  if (wave >= 8 && partner_prio) __builtin_amdgcn_s_setprio(3);
  const long long t0 = wall_clock64();
  float res = 0.f;
  if (wave < 8) {
    f16v acc0, acc1;
    for (int j = 0; j < 16; ++j) { acc0[j] = 0; acc1[j] = 0; }
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      }
    }
    for (int j = 0; j < 16; ++j) res += acc0[j] + acc1[j];
  } else if (mode == 1) {          // dependent-free VALU fma
    float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
    for (int it = 0; it < partner_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      }
    }
    res = x0 + x1 + x2 + x3;
  } else if (mode == 2) {          // global loads (L2-resident 64 KB), 4 independent 16-byte loads in flight
    const float4* p = reinterpret_cast<const float4*>(in);
    float4 s = make_float4(0, 0, 0, 0);
    int idx = tid & 4095;
    for (int it = 0; it < partner_iters; ++it) {
      float4 v0 = p[idx], v1 = p[(idx + 1024) & 4095], v2 = p[(idx + 2048) & 4095], v3 = p[(idx + 3072) & 4095];
      s.x += v0.x + v1.x + v2.x + v3.x;
      idx = (idx + 64 + (int)(s.x * 0.f)) & 4095;
    }
    res = s.x;
  } else if (mode == 3) {          // LDS reads
    float s = 0;
    int idx = lane * 4;
    for (int it = 0; it < partner_iters; ++it) {
      const float4 v = *reinterpret_cast<const float4*>(lds + (idx & 8188));
      s += v.x + v.y;
      idx += 256;
    }
    res = s;
  }
  const long long t1 = wall_clock64();
  if (lane == 0) t[blockIdx.x * 16 + wave] = t1 - t0;
  out[blockIdx.x * 1024 + tid] = res;
}

int main() {
  const int blocks = 256;
  float *out, *in;
  long long* t;
  hipMalloc(&out, blocks * 1024 * 4);
  hipMalloc(&in, 16384 * 4);
  hipMalloc(&t, blocks * 16 * 8);
  float h[16384];
  unsigned s = 1;
  for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 16777216.0f) * 0.5f; }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[] = {"MFMA alone            ", "MFMA + VALU fma       ", "MFMA + global loads   ",
                         "MFMA + LDS reads      "};
  const int piters[] = {0, 6000, 3000, 20000};
  long long ht[blocks * 16];
  for (int prio = 0; prio < 2; ++prio)
  for (int pass = 0; pass < 2; ++pass)
    for (int mode = 0; mode < 4; ++mode)
      for (int mf = 0; mf < 2; ++mf) {     // mf = 0: partner alone (no MFMA work), 1: both
        if (mode == 0 && mf == 0) continue;
        hipMemset(t, 0, blocks * 16 * 8);
        hipLaunchKernelGGL(corun, dim3(blocks), dim3(1024), 0, 0, out, in, t, mf ? 2000 : 0, mode, piters[mode], prio);
        hipDeviceSynchronize();
        hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
        double m = 0, p = 0;
        for (int b = 0; b < blocks; ++b)
          for (int w = 0; w < 16; ++w) (w < 8 ? m : p) += ht[b * 16 + w] / 100.0;
        if (pass) printf("[partner prio %d] %s %s: MFMA waves %8.1f us   partner waves %8.1f us\n", prio * 3, names[mode],
                         mf ? "together" : "partner alone", m / (blocks * 8), p / (blocks * 8));
      }
  return 0;
}
