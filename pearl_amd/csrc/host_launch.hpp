// Host-side launch helpers shared by dqn.hip and mlp.hip.
#pragma once
#include "dqn_kernels.hpp"

namespace pa {
namespace {

template <typename K>
int set_max_smem(K kernel, size_t bytes) {
  PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PA_OK;
}

// One launch, one or two independent problems (blockIdx.z).
template <bool B_KS>
int launch_linear(const GemmArgs* probs, int nprob, hipStream_t s) {
  constexpr int KW = 4;
  static size_t configured = 0;
  auto kern = linear_kernel<B_KS, KW>;
  LinArgs a;
  memset(&a, 0, sizeof(a));
  size_t smem = 0;
  int gx = 0, gy = 0;
  for (int i = 0; i < nprob; ++i) {
    a.p[i] = probs[i];
    const size_t b = linear_smem_bytes<B_KS, KW>(probs[i].K);
    smem = b > smem ? b : smem;
    gx = (int)ceil_div(probs[i].N, G_BN) > gx ? (int)ceil_div(probs[i].N, G_BN) : gx;
    gy = (int)ceil_div(probs[i].M, G_BM) > gy ? (int)ceil_div(probs[i].M, G_BM) : gy;
  }
  if (smem > configured) {
    int rc = set_max_smem(kern, smem);
    if (rc != PA_OK) return rc;
    configured = smem;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy, (unsigned)nprob), dim3(128 * KW), smem,
                     s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}


}  // namespace
}  // namespace pa
