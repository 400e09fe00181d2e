// fp16x2 split arithmetic shared by the kernels that run fp32 GEMMs on the fp16 matrix pipe
// (online_f16_kernel.hpp: the online row pass; target_h2_kernel.hpp: the target tile): every operand is
// scaled by an exact power of two into fp16's range and split into two fp16 terms,
// x 2^s = hi + lo with |x 2^s - hi - lo| <= 2^-24 |x 2^s|; a product is hi hi | hi lo + lo hi.
#pragma once
#include "dqn_kernels.hpp"

namespace pa {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// exponent field of a magnitude (sign bit cleared) -> the field the scaling works with
__host__ __device__ inline int h2_field(unsigned abs_bits) {
  const int e = (int)(abs_bits >> 23) & 0xff;
  return e == 255 ? 141 : (e < 15 ? 15 : e);
}
// 2^(141 - field): the maximum lands in [2^14, 2^15)
__device__ __forceinline__ float h2_scale(int field) {
  return __uint_as_float((unsigned)(268 - field) << 23);
}
// (templates: inline asm with "v" constraints must not be parsed by the host pass)
template <int D = 0>
__device__ __forceinline__ void h2_pair(float x0, float x1, float s0, float s1, unsigned& hi, unsigned& lo) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(s0));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(s1));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(s0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(s1), "v"(hi));
}
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

}  // namespace pa
