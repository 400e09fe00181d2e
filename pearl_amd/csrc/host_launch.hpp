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

// weight_grad_kernel with the split-K policy: batches of 2048 rows and more are cut into slices of
// ~1024 rows per workgroup (a 64 x 32 tile over 4096 rows is 27 us of MFMA on one CU).  The scratch
// for the partial tiles is one buffer per process, grown on demand; launches are ordered by their
// stream like every other use of a learner handle.
inline int launch_weight_grad(DwArgs& a, bool loss_wg, hipStream_t s) {
  static float* scratch = nullptr;
  static unsigned* tickets = nullptr;
  static size_t scratch_floats = 0, ticket_count = 0;
  int ks = 1;
  if (a.B >= 2048) {
    ks = a.B / 1024;
    if (ks > 8) ks = 8;
  }
  a.ksplit = ks;
  a.kscratch = nullptr;
  a.ktickets = nullptr;
  if (ks > 1) {
    const size_t need = (size_t)a.total_tiles * ks * (DW_TM * DW_TN + DW_TM);
    if (need > scratch_floats) {
      PA_HIP(hipDeviceSynchronize());
      if (scratch) (void)hipFree(scratch);
      PA_HIP(hipMalloc((void**)&scratch, need * sizeof(float)));
      scratch_floats = need;
    }
    if ((size_t)a.total_tiles > ticket_count) {
      PA_HIP(hipDeviceSynchronize());
      if (tickets) (void)hipFree(tickets);
      const size_t n = (size_t)a.total_tiles * 2;
      PA_HIP(hipMalloc((void**)&tickets, n * sizeof(unsigned)));
      PA_HIP(hipMemset(tickets, 0, n * sizeof(unsigned)));
      ticket_count = n;
    }
    a.kscratch = scratch;
    a.ktickets = tickets;
  }
  const unsigned grid = (unsigned)(a.total_tiles * ks) + (loss_wg ? 1u : 0u);
  hipLaunchKernelGGL(weight_grad_kernel, dim3(grid), dim3(512), 0, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

}  // namespace
}  // namespace pa
