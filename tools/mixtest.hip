// v_fma_mix{lo,hi}_f16 as the fp16x2 split of online_f16_kernel.hpp: hi = f16(x s), lo = f16(x s - hi), one rounding
// each, bitwise the cvt / sub / cvt sequence.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mixtest.hip -o tools/mixtest
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int D = 0> __device__ __forceinline__ unsigned pair_hi(float x0, float x1, float s) {
  unsigned h;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  return h;
}
template <int D = 0> __device__ __forceinline__ unsigned pair_lo(float x0, float x1, float s, unsigned h) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
  return l;
}
__global__ void k(const float* x, float s, unsigned* o, float* chk) {
  int i = threadIdx.x;
  float a = x[2*i], b = x[2*i+1];
  unsigned h = pair_hi(a, b, s), l = pair_lo(a, b, s, h);
  o[2*i] = h; o[2*i+1] = l;
  // reference
  _Float16 h0 = (_Float16)(a*s), h1 = (_Float16)(b*s);
  _Float16 l0 = (_Float16)(a*s - (float)h0), l1 = (_Float16)(b*s - (float)h1);
  unsigned short hb0 = __builtin_bit_cast(unsigned short, h0), hb1 = __builtin_bit_cast(unsigned short, h1);
  unsigned short lb0 = __builtin_bit_cast(unsigned short, l0), lb1 = __builtin_bit_cast(unsigned short, l1);
  chk[i] = (h == ((unsigned)hb0 | ((unsigned)hb1 << 16)) && l == ((unsigned)lb0 | ((unsigned)lb1 << 16))) ? 1.f : 0.f;
}
int main() {
  const int N = 256; float hx[2*N]; srand(3);
  for (int i = 0; i < 2*N; ++i) hx[i] = ((float)rand()/RAND_MAX*2-1) * (i%7==0 ? 1e-4f : 1.f) * 0.07f;
  float *dx, *dc; unsigned* d_o;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&d_o, 2*N*4); hipMalloc(&dc, N*4);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(N), 0, 0, dx, 262144.f, d_o, dc);
  float hc[N]; hipMemcpy(hc, dc, N*4, hipMemcpyDeviceToHost);
  int ok = 0; for (int i = 0; i < N; ++i) ok += hc[i] == 1.f;
  printf("pairs matching the cvt reference: %d / %d\n", ok, N);
  return 0;
}
