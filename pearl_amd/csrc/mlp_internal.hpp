// Internals of the generic MLP engine shared between mlp.hip and the fused learner steps
// (sac_step.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pearl_amd.h"

struct pa_mlp {
  pa_mlp_desc d;
  pa_mlp_buffers bufs;
  bool bound;
  int L;
  int64_t woff[PA_MLP_MAX_LAYERS], boff[PA_MLP_MAX_LAYERS], P;
  // desc.layer_norm: gamma / beta of hidden layer l's LayerNorm (flat offsets, behind the W / b block),
  // what its backward needs of the kept forward (the normalised values and 1 / sqrt(var + eps) per
  // row), and the column sums' partials; norm0 = first float of the norm block (P when none)
  int64_t goff[PA_MLP_MAX_LAYERS], betaoff[PA_MLP_MAX_LAYERS], norm0;
  float* xhat[PA_MLP_MAX_LAYERS];
  float* rstd[PA_MLP_MAX_LAYERS];
  float* norm_part;
  // desc.batch_norm: weight / bias offsets (behind the LayerNorm block), the activation output in front
  // of the BatchNorm (its backward and the activation's need it), the batch's column mean and
  // 1 / sqrt(var + eps), the caller's running statistics [online | target]
  int64_t bn_goff[PA_MLP_MAX_LAYERS], bn_boff[PA_MLP_MAX_LAYERS];
  float* hpre[PA_MLP_MAX_LAYERS];
  float* bn_mean[PA_MLP_MAX_LAYERS];
  float* bn_rstd[PA_MLP_MAX_LAYERS];
  float* bn_tmp;                  // [4][max width]: statistics of forwards that are not kept; d gamma / d beta
                                  // of a backward that writes no parameter gradients
  float* bn_run_mean[2][PA_MLP_MAX_LAYERS];
  float* bn_run_var[2][PA_MLP_MAX_LAYERS];
  long long* bn_nbt[2][PA_MLP_MAX_LAYERS];
  // desc.dropout: the caller's keep masks for the next forward, and the kept forward's
  const float* drop_mask[PA_MLP_MAX_LAYERS]; int drop_ld[PA_MLP_MAX_LAYERS];
  const float* drop_kept[PA_MLP_MAX_LAYERS]; int drop_kept_ld[PA_MLP_MAX_LAYERS];
  // desc.residual: d out of a wrapped block, saved in front of the block's own backward
  float* dres[PA_MLP_MAX_LAYERS + 1];
  float* act[PA_MLP_MAX_LAYERS];  // hidden activations kept for the backward pass [max_batch, d]
  float* dz[PA_MLP_MAX_LAYERS];   // pre-activation gradient of every hidden layer [max_batch, d]
                                  // (all kept: the weight gradients of all layers are one launch)
  float* db_scratch;              // column sums of a bias-free last layer go here
  float* loss_scratch;
  int kept_B;                     // batch size of the kept forward (0 = none)
  // pa_mlp_q_all (allocated on first use): fragment-major copy of W2 and the first layer's state
  // product [max_batch, H1], the two operands target_fused_kernel needs beside the parameters
  float* qa_w2f;
  float* qa_u;
  void* qa_w2sp;                  // the same W2 as bf16 split planes (H1 = H2 = 256)
  // row-pass path (mlp_rowpass.hpp): fragment-major copies of every layer — online W_l and W_l^T,
  // target W_l — rebuilt lazily by ONE launch when the parameters may have changed (bind, AdamW,
  // soft update, pa_mlp_invalidate)
  bool row_ok;                    // shape fits the row-pass kernels
  float* wf[PA_MLP_MAX_LAYERS];
  float* wtf[PA_MLP_MAX_LAYERS];
  float* wf_t[PA_MLP_MAX_LAYERS];
  // the online W_l as bf16x3 split planes (wsp16_index, dqn_kernels.hpp): the fused row step's
  // forward GEMMs on the bf16 matrix pipe (mlp_rowstep.hpp, launches of 32 rows per workgroup);
  // kept current with wf (repack launch, optimizer epilogue)
  void* wsp[PA_MLP_MAX_LAYERS];
  // ... and W_l^T ([d_l units][d_{l+1}]) the same way, l >= 1: the row step's backward GEMMs
  void* wtsp[PA_MLP_MAX_LAYERS];
  bool packed_ok, packed_t_ok;
  // fp16x2 row kernels (sac_rows.hpp, H2 instantiations): max |w| per row of layer 1's weights (the
  // hidden 256 x 256 layer of a three-layer network), bit patterns — [online | target] x two buffers
  // of dims[2] entries (um_cur: the one that describes the parameters as they are; the other is zero,
  // ready for the atomic maxima of the next fused optimizer launch).  Allocated and computed on
  // first use (mlp_ensure_um); kept current by the weight-gradient launch's AdamW / soft-update
  // epilogue while um_ok; every other way the parameters change clears um_ok with packed_ok.
  unsigned* um[2];
  int um_cur[2];
  bool um_ok[2];
  // weight gradients deferred to pa_mlp_adam (want_dw = 2): the operands of the kept backward
  struct Pending {
    bool active;
    const float* x; int ldx; int B;
    const float* dzs[PA_MLP_MAX_LAYERS];
    int ldzs[PA_MLP_MAX_LAYERS];
  } pend;
  // set by a fused learner step whose rows exchange data between workgroups of one launch
  // (sac_step.hip): the device error word of that exchange; pa_mlp_adam hands it to the
  // weight-gradient kernel, which then leaves the optimizer alone (AdamFuse::guard)
  const int* adam_guard;
  // per-network state owned by such a step (its exchange buffers), released with the network
  void* aux;
  void (*aux_free)(void*);
};

namespace pa {
// fragment-major copies of the online (or target) parameters are current after this
int mlp_ensure_packed(pa_mlp* h, bool target, hipStream_t s);
// row maxima of layer 1's weights (online or target) are current after this; the buffer to read
int mlp_ensure_um(pa_mlp* h, bool target, hipStream_t s);
inline const unsigned* mlp_um(const pa_mlp* h, bool target) {
  const int w = target ? 1 : 0;
  return (h->um[w] && h->um_ok[w]) ? h->um[w] + (size_t)h->um_cur[w] * h->d.dims[2] : nullptr;
}
// the operands of a backward pass whose weight gradients pa_mlp_adam will form (want_dw = 2)
void mlp_set_pending(pa_mlp* h, const float* x, int ldx, int B, const float* const* dzs,
                     const int* ldzs);
// Two networks with pending weight gradients and ONE optimizer configuration (twin critics):
// dW + AdamW (+ soft target update when soft_tau >= 0) of both in one launch.
bool mlp_pair_fusable(const pa_mlp* a, const pa_mlp* b, bool soft);
struct TailJob;
// `tail` (nullable): end-of-step scalar work that rides the launch as one extra workgroup
int mlp_adam_pair(pa_mlp* a, pa_mlp* b, int64_t step, float soft_tau, hipStream_t s,
                  const TailJob* tail);
}  // namespace pa
