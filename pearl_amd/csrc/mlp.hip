// Generic fully-connected network engine + the actor-critic kernels (PPO, continuous SAC).
//
// Reference being replaced (file:line under /root/reference):
//   mlp_block                         pearl/neural_networks/common/utils.py:75-152
//   VanillaValueNetwork.forward       pearl/neural_networks/common/value_networks.py:35-59
//   VanillaActorNetwork               .../sequential_decision_making/actor_networks.py:107-176
//   GaussianActorNetwork              .../actor_networks.py:488-629
//   TwinCritic / critic losses        .../twin_critic.py:22-91, utils/functional_utils/learning/critic_utils.py:139-203
//   ActorCriticBase.learn_batch       policy_learners/sequential_decision_making/actor_critic_base.py:309-366
//   PPO losses + GAE                  .../ppo.py:152-293
//   ContinuousSoftActorCritic         .../soft_actor_critic_continuous.py:131-231
//
// A pa_mlp is Linear+ReLU hidden layers and a linear last layer over flat, caller-owned fp32
// buffers (parameters, optional target copy, gradient, AdamW state) — the same ownership rule as
// pa_dqn.  Forward/backward reuse the fp32-MFMA kernels of dqn_kernels.hpp (linear_kernel,
// weight_grad_kernel, adamw_kernel); the algorithm heads below are small fused elementwise /
// reduction kernels with torch's op order (one rounding per op, -ffp-contract=off).
#include <math.h>

#include <new>

#include "host_launch.hpp"
#include "mlp_rowpass.hpp"
#include "mlp_rowstep.hpp"
#include "mlp_internal.hpp"
#include "mlp_norm_act.hpp"
#include "sac_rows.hpp"

using namespace pa;

// struct pa_mlp: mlp_internal.hpp (shared with sac_step.hip)

namespace {

int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }

int layout(const pa_mlp_desc* d, int64_t* woff, int64_t* boff, int64_t* total) {
  PA_REQUIRE(d && d->n_layers >= 1 && d->n_layers <= PA_MLP_MAX_LAYERS, PA_ERR_INVALID,
             "n_layers must be in [1, %d]", PA_MLP_MAX_LAYERS);
  int64_t o = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    PA_REQUIRE(d->dims[l] > 0 && d->dims[l + 1] > 0, PA_ERR_INVALID, "layer dims must be positive");
    woff[l] = o; o = align4(o + (int64_t)d->dims[l + 1] * d->dims[l]);
    boff[l] = o; o = align4(o + d->dims[l + 1]);
  }
  PA_REQUIRE(d->hidden_act >= 0 && d->hidden_act < ACT_COUNT, PA_ERR_INVALID,
             "hidden_act %d is not an activation (0 relu, 1 leaky_relu, 2 tanh, 3 softplus, 4 sigmoid)",
             d->hidden_act);
  PA_REQUIRE(d->layer_norm >= 0 && (d->n_layers <= 1 ? d->layer_norm == 0
                                                       : d->layer_norm < (1 << (d->n_layers - 1))),
             PA_ERR_INVALID, "layer_norm is a bit mask over the %d hidden layers", d->n_layers - 1);
  const int hidden_bits = d->n_layers <= 1 ? 0 : (1 << (d->n_layers - 1)) - 1;
  PA_REQUIRE((d->batch_norm & ~hidden_bits) == 0 && (d->dropout & ~hidden_bits) == 0, PA_ERR_INVALID,
             "batch_norm / dropout are bit masks over the %d hidden layers", d->n_layers - 1);
  PA_REQUIRE((d->residual & ~((1 << d->n_layers) - 1)) == 0, PA_ERR_INVALID,
             "residual is a bit mask over the %d layers", d->n_layers);
  for (int l = 0; l < d->n_layers; ++l)
    PA_REQUIRE(!((d->residual >> l) & 1) || d->dims[l] == d->dims[l + 1], PA_ERR_INVALID,
               "residual layer %d needs d_in == d_out (%d vs %d)", l, d->dims[l], d->dims[l + 1]);
  *total = o;
  return PA_OK;
}
// the LayerNorm parameters of the hidden layers follow the W / b block: gamma_l, beta_l, ...
int norm_layout(const pa_mlp_desc* d, int64_t wb_total, int64_t* goff, int64_t* betaoff, int64_t* total) {
  int64_t o = wb_total;
  for (int l = 0; l + 1 < d->n_layers; ++l) {
    goff[l] = betaoff[l] = -1;
    if (!((d->layer_norm >> l) & 1)) continue;
    goff[l] = o; o = align4(o + d->dims[l + 1]);
    betaoff[l] = o; o = align4(o + d->dims[l + 1]);
  }
  *total = o;
  return PA_OK;
}
// ... and the BatchNorm1d weights / biases follow the LayerNorm block
int bn_layout(const pa_mlp_desc* d, int64_t ln_total, int64_t* goff, int64_t* boff, int64_t* total) {
  int64_t o = ln_total;
  for (int l = 0; l + 1 < d->n_layers; ++l) {
    goff[l] = boff[l] = -1;
    if (!((d->batch_norm >> l) & 1)) continue;
    goff[l] = o; o = align4(o + d->dims[l + 1]);
    boff[l] = o; o = align4(o + d->dims[l + 1]);
  }
  *total = o;
  return PA_OK;
}
// mlp_block's plain form (Linear + ReLU): what the fused row kernels compute
inline bool plain_net(const pa_mlp* h) {
  return h->d.hidden_act == 0 && h->d.layer_norm == 0 && h->d.batch_norm == 0 && h->d.dropout == 0 &&
         h->d.residual == 0;
}
// layer-by-layer forms that need the row-local kernel between a hidden GEMM and the next one
inline bool rowlocal_net(const pa_mlp* h) {
  return h->d.hidden_act != 0 || h->d.layer_norm != 0 || h->d.dropout != 0;
}

AdamScalars adam_scalars(const pa_mlp_desc& d, int64_t step) {
  const double bc1 = 1.0 - pow(d.beta1, (double)step);
  const double bc2 = 1.0 - pow(d.beta2, (double)step);
  AdamScalars c;
  c.decay = (float)(1.0 - d.lr * d.weight_decay);
  c.w1 = (float)(1.0 - d.beta1);
  c.beta2 = (float)d.beta2;
  c.omb2 = (float)(1.0 - d.beta2);
  c.bc2_sqrt = (float)sqrt(bc2);
  c.neg_step = (float)(-(d.lr / bc1));
  c.eps = (float)d.eps;
  c.amsgrad = d.amsgrad;
  return c;
}

}  // namespace

extern "C" int64_t pa_mlp_param_count(const pa_mlp_desc* d) {
  int64_t w[PA_MLP_MAX_LAYERS], b[PA_MLP_MAX_LAYERS], total = 0;
  if (layout(d, w, b, &total) != PA_OK) return -1;
  (void)norm_layout(d, total, w, b, &total);
  (void)bn_layout(d, total, w, b, &total);
  return total;
}

extern "C" int pa_mlp_bn_offsets(const pa_mlp_desc* d, int64_t* offsets) {
  PA_REQUIRE(offsets, PA_ERR_INVALID, "null output");
  int64_t w[PA_MLP_MAX_LAYERS], b[PA_MLP_MAX_LAYERS], total = 0;
  int rc = layout(d, w, b, &total);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(d->batch_norm, PA_ERR_INVALID, "pa_mlp_bn_offsets: the network has no BatchNorm1d");
  (void)norm_layout(d, total, w, b, &total);
  (void)bn_layout(d, total, w, b, &total);
  for (int l = 0; l + 1 < d->n_layers; ++l) {
    offsets[2 * l] = w[l];
    offsets[2 * l + 1] = b[l];
  }
  return PA_OK;
}

extern "C" int pa_mlp_norm_offsets(const pa_mlp_desc* d, int64_t* offsets) {
  PA_REQUIRE(offsets, PA_ERR_INVALID, "null output");
  int64_t w[PA_MLP_MAX_LAYERS], b[PA_MLP_MAX_LAYERS], total = 0;
  int rc = layout(d, w, b, &total);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(d->layer_norm, PA_ERR_INVALID, "pa_mlp_norm_offsets: the network has no LayerNorm");
  (void)norm_layout(d, total, w, b, &total);
  for (int l = 0; l + 1 < d->n_layers; ++l) {
    offsets[2 * l] = w[l];
    offsets[2 * l + 1] = b[l];
  }
  return PA_OK;
}

extern "C" int pa_mlp_param_offsets(const pa_mlp_desc* d, int64_t* offsets) {
  PA_REQUIRE(offsets, PA_ERR_INVALID, "null output");
  int64_t w[PA_MLP_MAX_LAYERS], b[PA_MLP_MAX_LAYERS], total = 0;
  int rc = layout(d, w, b, &total);
  if (rc != PA_OK) return rc;
  for (int l = 0; l < d->n_layers; ++l) {
    offsets[2 * l] = w[l];
    offsets[2 * l + 1] = b[l];
  }
  return PA_OK;
}

extern "C" int pa_mlp_destroy(pa_mlp* h) {
  if (!h) return PA_OK;
  (void)hipSetDevice(h->d.device);
  (void)hipDeviceSynchronize();
  for (int l = 0; l < PA_MLP_MAX_LAYERS; ++l)
    if (h->act[l]) (void)hipFree(h->act[l]);
  for (int i = 0; i < PA_MLP_MAX_LAYERS; ++i)
    if (h->dz[i]) (void)hipFree(h->dz[i]);
  for (int i = 0; i < PA_MLP_MAX_LAYERS; ++i) {
    if (h->hpre[i]) (void)hipFree(h->hpre[i]);
    if (h->bn_mean[i]) (void)hipFree(h->bn_mean[i]);
    if (h->bn_rstd[i]) (void)hipFree(h->bn_rstd[i]);
  }
  for (int i = 0; i <= PA_MLP_MAX_LAYERS; ++i)
    if (h->dres[i]) (void)hipFree(h->dres[i]);
  if (h->bn_tmp) (void)hipFree(h->bn_tmp);
  for (int i = 0; i < PA_MLP_MAX_LAYERS; ++i) {
    if (h->xhat[i]) (void)hipFree(h->xhat[i]);
    if (h->rstd[i]) (void)hipFree(h->rstd[i]);
  }
  if (h->norm_part) (void)hipFree(h->norm_part);
  if (h->db_scratch) (void)hipFree(h->db_scratch);
  if (h->loss_scratch) (void)hipFree(h->loss_scratch);
  if (h->qa_w2f) (void)hipFree(h->qa_w2f);
  if (h->qa_u) (void)hipFree(h->qa_u);
  if (h->qa_w2sp) (void)hipFree(h->qa_w2sp);
  if (h->aux && h->aux_free) h->aux_free(h->aux);
  for (int w = 0; w < 2; ++w)
    if (h->um[w]) (void)hipFree(h->um[w]);
  for (int l = 0; l < PA_MLP_MAX_LAYERS; ++l) {
    if (h->wf[l]) (void)hipFree(h->wf[l]);
    if (h->wtf[l]) (void)hipFree(h->wtf[l]);
    if (h->wf_t[l]) (void)hipFree(h->wf_t[l]);
    if (h->wsp[l]) (void)hipFree(h->wsp[l]);
    if (h->wtsp[l]) (void)hipFree(h->wtsp[l]);
  }
  delete h;
  release_process_device();
  return PA_OK;
}

extern "C" int pa_mlp_create(pa_mlp** out, const pa_mlp_desc* desc) {
  PA_REQUIRE(out && desc, PA_ERR_INVALID, "pa_mlp_create: null argument");
  PA_REQUIRE(desc->max_batch > 0, PA_ERR_INVALID, "max_batch must be positive");
  const int ndev = pa_device_count();
  PA_REQUIRE(desc->device >= 0 && desc->device < ndev, PA_ERR_HIP,
             "HIP device %d not available (%d visible): pa_mlp is HIP-only and has no CPU fallback",
             desc->device, ndev);
  pa_mlp* h = new (std::nothrow) pa_mlp();
  PA_REQUIRE(h, PA_ERR_NOMEM, "out of host memory");
  memset(h, 0, sizeof(*h));
  h->d = *desc;
  h->L = desc->n_layers;
  int rc = layout(desc, h->woff, h->boff, &h->P);
  if (rc != PA_OK) {
    delete h;
    return rc;
  }
  h->norm0 = h->P;
  (void)norm_layout(desc, h->P, h->goff, h->betaoff, &h->P);
  (void)bn_layout(desc, h->P, h->bn_goff, h->bn_boff, &h->P);
  {
    // (from here on every failure goes through pa_mlp_destroy, which releases the binding)
    int rc_dev = bind_process_device(desc->device);
    if (rc_dev != PA_OK) {
      delete h;
      return rc_dev;
    }
  }
  PA_HIP(hipSetDevice(desc->device));
  int maxh = 1;
  for (int l = 0; l < h->L; ++l) maxh = desc->dims[l + 1] > maxh ? desc->dims[l + 1] : maxh;
  auto alloc = [&](float** p, int64_t floats) {
    return hipMalloc((void**)p, (size_t)(floats * 4)) == hipSuccess;
  };
  bool ok = true;
  for (int l = 0; l + 1 < h->L; ++l)
    ok = ok && alloc(&h->act[l], (int64_t)desc->max_batch * desc->dims[l + 1]);
  for (int l = 1; l < h->L; ++l)   // dz[l]: gradient w.r.t. the output of layer l - 1
    ok = ok && alloc(&h->dz[l], (int64_t)desc->max_batch * desc->dims[l]);
  ok = ok && alloc(&h->db_scratch, maxh);
  ok = ok && alloc(&h->loss_scratch, 4);
  if (desc->layer_norm) {
    for (int l = 0; l + 1 < h->L; ++l) {
      if (!((desc->layer_norm >> l) & 1)) continue;
      ok = ok && alloc(&h->xhat[l], (int64_t)desc->max_batch * desc->dims[l + 1]);
      ok = ok && alloc(&h->rstd[l], desc->max_batch);
    }
    ok = ok && alloc(&h->norm_part, (int64_t)NP_BLOCKS * 2 * maxh);
  }
  // the activation's own output, kept where something behind it (BatchNorm, the residual add)
  // overwrites act[l]: act' is formed from it
  for (int l = 0; l + 1 < h->L; ++l)
    if (((desc->batch_norm | desc->residual) >> l) & 1)
      ok = ok && alloc(&h->hpre[l], (int64_t)desc->max_batch * desc->dims[l + 1]);
  if (desc->batch_norm) {
    for (int l = 0; l + 1 < h->L; ++l) {
      if (!((desc->batch_norm >> l) & 1)) continue;
      ok = ok && alloc(&h->bn_mean[l], desc->dims[l + 1]);
      ok = ok && alloc(&h->bn_rstd[l], desc->dims[l + 1]);
    }
    if (!h->norm_part) ok = ok && alloc(&h->norm_part, (int64_t)NP_BLOCKS * 2 * maxh);
    ok = ok && alloc(&h->bn_tmp, (int64_t)4 * maxh);
  }
  for (int l = 0; l < h->L; ++l)
    if ((desc->residual >> l) & 1) ok = ok && alloc(&h->dres[l], (int64_t)desc->max_batch * desc->dims[l + 1]);
  {
    static const bool enabled = []() {
      const char* v = getenv("PEARL_AMD_MLP_ROWPASS");
      return !(v && *v == '0');
    }();
    // (LayerNorm / other hidden activations: layer by layer, mlp_norm_act.hpp)
    h->row_ok = enabled && plain_net(h) && h->L <= ROW_MAX_LAYERS && desc->dims[0] <= ROW_MAX_IN;
    for (int l = 0; l < h->L; ++l) h->row_ok = h->row_ok && desc->dims[l + 1] <= ROW_MAX_OUT;
    if (h->row_ok) {
      for (int l = 0; l < h->L; ++l) {
        ok = ok && alloc(&h->wf[l], wf16_floats(desc->dims[l + 1], desc->dims[l]));
        ok = ok && alloc(&h->wf_t[l], wf16_floats(desc->dims[l + 1], desc->dims[l]));
        ok = ok && alloc(&h->wtf[l], wf16_floats(desc->dims[l], desc->dims[l + 1]));
        ok = ok && hipMalloc(&h->wsp[l], (size_t)wsp16_bytes(desc->dims[l + 1], desc->dims[l])) == hipSuccess;
        if (l >= 1)
          ok = ok && hipMalloc(&h->wtsp[l], (size_t)wsp16_bytes(desc->dims[l], desc->dims[l + 1])) == hipSuccess;
      }
    }
  }
  if (!ok) {
    set_error("hipMalloc(mlp workspace) failed");
    pa_mlp_destroy(h);
    return PA_ERR_NOMEM;
  }
  *out = h;
  return PA_OK;
}

extern "C" int pa_mlp_bind(pa_mlp* h, const pa_mlp_buffers* b) {
  PA_REQUIRE(h && b && b->p, PA_ERR_INVALID, "pa_mlp_bind: null argument");
  const float* all[] = {b->p, b->p_target, b->grad, b->exp_avg, b->exp_avg_sq, b->max_exp_avg_sq};
  for (const float* p : all)
    PA_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, PA_ERR_INVALID,
               "flat buffers must be 16-byte aligned");
  h->bufs = *b;
  h->bound = true;
  h->packed_ok = h->packed_t_ok = false;
  return PA_OK;
}

extern "C" int pa_mlp_bind_batch_norm(pa_mlp* h, int32_t use_target, int32_t layer, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked) {
  PA_REQUIRE(h && layer >= 0 && layer + 1 < h->L && ((h->d.batch_norm >> layer) & 1), PA_ERR_INVALID,
             "pa_mlp_bind_batch_norm: hidden layer %d has no BatchNorm1d", layer);
  const int w = use_target ? 1 : 0;
  h->bn_run_mean[w][layer] = running_mean;
  h->bn_run_var[w][layer] = running_var;
  h->bn_nbt[w][layer] = reinterpret_cast<long long*>(num_batches_tracked);
  return PA_OK;
}

extern "C" int pa_mlp_set_dropout(pa_mlp* h, int32_t layer, const float* mask, int32_t ldm) {
  PA_REQUIRE(h && layer >= 0 && layer + 1 < h->L && ((h->d.dropout >> layer) & 1), PA_ERR_INVALID,
             "pa_mlp_set_dropout: hidden layer %d has no Dropout", layer);
  PA_REQUIRE(!mask || ldm >= h->d.dims[layer + 1], PA_ERR_INVALID, "pa_mlp_set_dropout: mask pitch %d < %d",
             ldm, h->d.dims[layer + 1]);
  h->drop_mask[layer] = mask;
  h->drop_ld[layer] = ldm;
  return PA_OK;
}

extern "C" int pa_mlp_invalidate(pa_mlp* h) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null mlp");
  h->packed_ok = h->packed_t_ok = false;
  return PA_OK;
}

namespace {

// fragment-major copies of the parameter set a row pass is about to read
int ensure_packed(pa_mlp* h, bool target, hipStream_t s) {
  bool& ok = target ? h->packed_t_ok : h->packed_ok;
  if (ok) return PA_OK;
  RowPackArgs a;
  memset(&a, 0, sizeof(a));
  a.P = target ? h->bufs.p_target : h->bufs.p;
  a.L = h->L;
  for (int l = 0; l <= h->L; ++l) a.dims[l] = h->d.dims[l];
  for (int l = 0; l < h->L; ++l) {
    a.woff[l] = h->woff[l];
    a.Wf[l] = target ? h->wf_t[l] : h->wf[l];
    a.Wtf[l] = target ? nullptr : h->wtf[l];
    a.Wsp[l] = target ? nullptr : h->wsp[l];
    a.Wtsp[l] = target ? nullptr : h->wtsp[l];
  }
  hipLaunchKernelGGL(mlp_rowpack_kernel, dim3(128), dim3(256), 0, s, a);
  PA_LAUNCH_CHECK();
  ok = true;
  h->um_ok[target ? 1 : 0] = false;   // whatever made the copies stale changed the parameters
  return PA_OK;
}

// max |w| per row of layer 1's weights -> out[0 .. M), zero[0 .. M) cleared: one wave per row
__global__ __launch_bounds__(256) void mlp_unit_max_kernel(const float* __restrict__ W, int M, int K,
                                                           unsigned* __restrict__ out,
                                                           unsigned* __restrict__ zero) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t unit = w0; unit < M; unit += nw) {
    unsigned m = 0u;
    for (int k = lane; k < K; k += 64) m = umaxu(m, abs_bits(W[unit * K + k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = umaxu(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) {
      out[unit] = m;
      zero[unit] = 0u;
    }
  }
}
int ensure_um(pa_mlp* h, bool target, hipStream_t s) {
  const int w = target ? 1 : 0;
  PA_REQUIRE(h->L >= 2 && h->row_ok, PA_ERR_INVALID, "no hidden layer to keep row maxima of");
  const int M = h->d.dims[2], K = h->d.dims[1];
  if (!h->um[w]) {
    PA_HIP(hipMalloc((void**)&h->um[w], sizeof(unsigned) * 2 * (size_t)M));
    h->um_ok[w] = false;
  }
  if (h->um_ok[w]) return PA_OK;
  const float* P = target ? h->bufs.p_target : h->bufs.p;
  PA_REQUIRE(P, PA_ERR_INVALID, "no parameters bound");
  h->um_cur[w] = 0;
  hipLaunchKernelGGL(mlp_unit_max_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, s,
                     P + h->woff[1], M, K, h->um[w], h->um[w] + M);
  PA_LAUNCH_CHECK();
  h->um_ok[w] = true;
  return PA_OK;
}

void fill_fwd(const pa_mlp* h, bool target, float* out, int ldo, bool keep, RowNetFwd& n) {
  const float* P = target ? h->bufs.p_target : h->bufs.p;
  n.L = h->L;
  n.relu = 0;
  for (int l = 0; l <= h->L; ++l) n.dims[l] = h->d.dims[l];
  for (int l = 0; l < h->L; ++l) {
    const bool last = l == h->L - 1;
    n.Wf[l] = target ? h->wf_t[l] : h->wf[l];
    n.Wsp[l] = target ? nullptr : h->wsp[l];
    n.bias[l] = (last && h->d.no_last_bias) ? nullptr : P + h->boff[l];
    n.act[l] = (!last && keep) ? h->act[l] : nullptr;
    if (!last && !((h->d.identity_layers >> l) & 1)) n.relu |= 1 << l;
  }
  n.out = out;
  n.ldo = ldo;
}

int launch_rowfwd(RowFwdArgs& a, int nnet, int d0max, hipStream_t s) {
  static size_t configured = 0;
  const size_t smem = rowfwd_smem_bytes(d0max);
  if (smem > configured) {
    int rc = set_max_smem(mlp_rowfwd_kernel, smem);
    if (rc != PA_OK) return rc;
    configured = smem;
  }
  hipLaunchKernelGGL(mlp_rowfwd_kernel, dim3((unsigned)ceil_div(a.B, RP_ROWS), (unsigned)nnet),
                     dim3(512), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

void fill_bwd(const pa_mlp* h, const float* d_out, int ldd, float* d_x, int lddx, RowNetBwd& n) {
  n.L = h->L;
  n.relu = 0;
  for (int l = 0; l <= h->L; ++l) n.dims[l] = h->d.dims[l];
  for (int l = 0; l < h->L; ++l) {
    n.Wtf[l] = h->wtf[l];
    n.Wtsp[l] = h->wtsp[l];
    n.act[l] = h->act[l];
    n.dz[l] = h->dz[l];
    if (l + 1 < h->L && !((h->d.identity_layers >> l) & 1)) n.relu |= 1 << l;
  }
  n.d_out = d_out; n.ldd = ldd;
  n.d_x = d_x; n.lddx = lddx;
}

int launch_rowbwd(RowBwdArgs& a, int nnet, hipStream_t s) {
  static bool configured = false;
  const size_t smem = rowbwd_smem_bytes();
  if (!configured) {
    int rc = set_max_smem(mlp_rowbwd_kernel, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  hipLaunchKernelGGL(mlp_rowbwd_kernel, dim3((unsigned)ceil_div(a.B, RP_ROWS), (unsigned)nnet),
                     dim3(512), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// ---- the [K0, H1, H2, DO] forward through sac_rows.hpp's building blocks (rows3_fwd_kernel) -------
bool rows3_enabled() {
  static const bool on = []() {
    const char* v = getenv("PEARL_AMD_ROWS3");
    return !(v && *v == '0');
  }();
  return on;
}
bool rows3_shape(const pa_mlp* h) {
  return h->row_ok && h->L == 3 && h->d.identity_layers == 0 && !h->d.no_last_bias &&
         h->d.dims[1] <= 256 && h->d.dims[2] <= 256 && h->d.dims[3] <= 32 &&
         h->d.dims[0] <= ROW_MAX_IN;
}
// all networks of a launch: same kind (critics: one output; or heads of 2..32 outputs), a grid that
// fits the chip once (beyond that — PPO's 4096-row minibatch — the three-workgroups-per-CU generic
// kernel is already MFMA-bound)
bool rows3_usable(pa_mlp* const* hs, int nnet, int B) {
  if (!rows3_enabled() || (int64_t)ceil_div(B, RP_ROWS) * nnet > 256) return false;
  for (int i = 0; i < nnet; ++i) {
    if (!rows3_shape(hs[i])) return false;
    if ((hs[i]->d.dims[3] == 1) != (hs[0]->d.dims[3] == 1)) return false;
    if (hs[i]->d.dims[0] != hs[0]->d.dims[0]) return false;
  }
  return true;
}
void fill_rows3(const pa_mlp* h, bool target, bool keep, SacMlp3& n) {
  const float* P = target ? h->bufs.p_target : h->bufs.p;
  float* const* wf = target ? h->wf_t : h->wf;
  memset(&n, 0, sizeof(n));
  n.W1f = wf[0]; n.b1 = P + h->boff[0];
  n.W2f = wf[1]; n.b2 = P + h->boff[1];
  n.W3f = wf[2]; n.b3 = P + h->boff[2];
  n.w3 = P + h->woff[2];
  n.act1 = keep ? h->act[0] : nullptr;
  n.act2 = keep ? h->act[1] : nullptr;
  n.K0 = h->d.dims[0]; n.H1 = h->d.dims[1]; n.H2 = h->d.dims[2]; n.DO = h->d.dims[3];
}
template <int NGH, bool CRITIC>
int launch_rows3_t(const Rows3FwdArgs& a, int nnet, int k0, hipStream_t s) {
  static size_t configured = 0;
  const size_t smem = sac_rows_smem_floats(k0) * sizeof(float);
  if (smem > configured) {
    int rc = set_max_smem(rows3_fwd_kernel<NGH, CRITIC>, smem);
    if (rc != PA_OK) return rc;
    configured = smem;
  }
  hipLaunchKernelGGL((rows3_fwd_kernel<NGH, CRITIC>), dim3((unsigned)ceil_div(a.B, RP_ROWS), (unsigned)nnet),
                     dim3(512), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
int launch_rows3(pa_mlp* const* hs, int nnet, bool target, const float* x, int ldx, int B,
                 float* const* outs, const int* ldos, bool keep, hipStream_t s) {
  Rows3FwdArgs a;
  memset(&a, 0, sizeof(a));
  bool st = true;
  for (int i = 0; i < nnet; ++i) {
    fill_rows3(hs[i], target, keep, a.net[i]);
    a.out[i] = outs[i]; a.ldo[i] = ldos[i];
    st = st && wf16_nkg(hs[i]->d.dims[1]) == 16 && wf16_nkg(hs[i]->d.dims[2]) == 16;
  }
  a.x = x; a.ldx = ldx; a.B = B;
  const bool critic = hs[0]->d.dims[3] == 1;
  const int k0 = hs[0]->d.dims[0];
  if (critic) return st ? launch_rows3_t<16, true>(a, nnet, k0, s) : launch_rows3_t<0, true>(a, nnet, k0, s);
  return st ? launch_rows3_t<16, false>(a, nnet, k0, s) : launch_rows3_t<0, false>(a, nnet, k0, s);
}

}  // namespace

// out = W_L-1(relu(... relu(W_0 x + b_0) ...)) + b_L-1.  keep = 1 retains the hidden activations for
// pa_mlp_backward (one kept forward at a time).
extern "C" int pa_mlp_forward(pa_mlp* h, int32_t use_target, const float* x, int32_t ldx, int32_t B,
                              float* out, int32_t ldo, int32_t keep, void* stream) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "mlp has no bound parameter buffers");
  PA_REQUIRE(x && out && B > 0 && B <= h->d.max_batch, PA_ERR_INVALID,
             "pa_mlp_forward: bad argument (B=%d, max_batch=%d)", B, h->d.max_batch);
  const float* P = use_target ? h->bufs.p_target : h->bufs.p;
  PA_REQUIRE(P, PA_ERR_INVALID, "no target parameters bound");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(h->d.device));
  if (h->row_ok) {
    // the whole network in one launch (mlp_rowpass.hpp)
    int rc = ensure_packed(h, use_target != 0, s);
    if (rc != PA_OK) return rc;
    if (rows3_usable(&h, 1, B)) {
      float* outs[1] = {out};
      const int ldos[1] = {ldo};
      rc = launch_rows3(&h, 1, use_target != 0, x, ldx, B, outs, ldos, keep != 0, s);
      if (rc != PA_OK) return rc;
      h->kept_B = keep ? B : 0;
      return PA_OK;
    }
    RowFwdArgs a;
    memset(&a, 0, sizeof(a));
    fill_fwd(h, use_target != 0, out, ldo, keep != 0, a.net[0]);
    a.x = x; a.ldx = ldx; a.B = B;
    rc = launch_rowfwd(a, 1, h->d.dims[0], s);
    if (rc != PA_OK) return rc;
    h->kept_B = keep ? B : 0;
    return PA_OK;
  }
  const float* in = x;
  int ldin = ldx;
  for (int l = 0; l < h->L; ++l) {
    const bool last = (l == h->L - 1);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = in; g.lda = ldin;
    g.Bm = P + h->woff[l]; g.ldb = h->d.dims[l];
    g.C = last ? out : h->act[l]; g.ldc = last ? ldo : h->d.dims[l + 1];
    g.bias = P + h->boff[l];
    g.M = B; g.N = h->d.dims[l + 1]; g.K = h->d.dims[l];
    const bool ident = ((h->d.identity_layers >> l) & 1) != 0;
    const bool relu = !last && !ident && plain_net(h);
    g.epi = relu ? EPI_BIAS_RELU : ((last && h->d.no_last_bias) ? EPI_NONE : EPI_BIAS);
    int rc = launch_linear<false>(&g, 1, s);
    if (rc != PA_OK) return rc;
    const bool ln = !last && ((h->d.layer_norm >> l) & 1);
    const bool drop = !last && ((h->d.dropout >> l) & 1) && h->drop_mask[l] != nullptr;
    const bool kept = keep && !use_target;   // (only the ONLINE network's kept forward feeds a backward)
    if (!last && kept) {
      h->drop_kept[l] = drop ? h->drop_mask[l] : nullptr;
      h->drop_kept_ld[l] = h->drop_ld[l];
    }
    if (!last && !plain_net(h) && (ln || !ident || drop)) {
      // LayerNorm (optional), dropout (optional) and the hidden activation, row by row, in place
      // (mlp_norm_act.hpp)
      NormActArgs na;
      memset(&na, 0, sizeof(na));
      na.z = h->act[l]; na.ldz = h->d.dims[l + 1];
      if (ln) {
        na.gamma = P + h->goff[l]; na.beta = P + h->betaoff[l];
        na.xhat = kept ? h->xhat[l] : nullptr;
        na.rstd = kept ? h->rstd[l] : nullptr;
      }
      if (drop) { na.drop = h->drop_mask[l]; na.ldd = h->drop_ld[l]; }
      na.B = B; na.d = h->d.dims[l + 1]; na.act = h->d.hidden_act; na.identity = ident ? 1 : 0;
      na.eps = 1e-5f;
      hipLaunchKernelGGL(norm_act_fwd_kernel, dim3((unsigned)ceil_div(B, NA_ROWS)), dim3(64 * NA_ROWS), 0, s, na);
      PA_LAUNCH_CHECK();
    }
    if (!last && ((h->d.batch_norm >> l) & 1)) {
      // BatchNorm1d after the activation, training mode: the statistics of THIS batch (utils.py:119-121)
      const int which = use_target ? 1 : 0, dd = h->d.dims[l + 1];
      BnArgs b;
      memset(&b, 0, sizeof(b));
      b.y = h->act[l]; b.ldy = dd;
      b.hpre = kept ? h->hpre[l] : nullptr;
      b.gamma = P + h->bn_goff[l]; b.beta = P + h->bn_boff[l];
      b.mean = kept ? h->bn_mean[l] : h->bn_tmp;
      b.rstd = kept ? h->bn_rstd[l] : h->bn_tmp + dd;
      b.run_mean = h->bn_run_mean[which][l]; b.run_var = h->bn_run_var[which][l]; b.nbt = h->bn_nbt[which][l];
      b.part = h->norm_part; b.B = B; b.d = dd; b.eps = 1e-5f; b.momentum = 0.1f;
      const int rows_per = (int)ceil_div(B, NP_BLOCKS);
      const int nb = (int)ceil_div(B, rows_per);
      const dim3 gcol((unsigned)ceil_div(dd, 64), (unsigned)nb), g1((unsigned)ceil_div(dd, 64));
      hipLaunchKernelGGL(col_partial_kernel, gcol, dim3(64), 0, s, (const float*)b.y, b.ldy, B, dd,
                         (const float*)nullptr, b.part, rows_per, 0);
      hipLaunchKernelGGL(bn_stats_kernel, g1, dim3(64), 0, s, b, nb, 0);
      hipLaunchKernelGGL(col_partial_kernel, gcol, dim3(64), 0, s, (const float*)b.y, b.ldy, B, dd,
                         (const float*)b.mean, b.part, rows_per, 1);
      hipLaunchKernelGGL(bn_stats_kernel, g1, dim3(64), 0, s, b, nb, 1);
      hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)ceil_div((int64_t)B * dd, 256)), dim3(256), 0, s, b);
      PA_LAUNCH_CHECK();
    }
    if ((h->d.residual >> l) & 1) {
      // ResidualWrapper: out = in + block(in) (d_l == d_{l+1})
      float* dst = last ? out : h->act[l];
      const int ldd_ = last ? ldo : h->d.dims[l + 1];
      if (!last && kept && !((h->d.batch_norm >> l) & 1)) {   // (BatchNorm has kept it already)
        hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)ceil_div((int64_t)B * h->d.dims[l + 1], 256)), dim3(256),
                           0, s, h->hpre[l], h->d.dims[l + 1], (const float*)h->act[l], h->d.dims[l + 1], B,
                           h->d.dims[l + 1], 0);
      }
      hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)ceil_div((int64_t)B * h->d.dims[l + 1], 256)), dim3(256),
                         0, s, dst, ldd_, in, ldin, B, h->d.dims[l + 1], 1);
      PA_LAUNCH_CHECK();
    }
    in = h->act[l];
    ldin = h->d.dims[l + 1];
  }
  h->kept_B = keep ? B : 0;
  return PA_OK;
}

// Copy the kept output of hidden layer `layer` (0-based, after its ReLU) of the last keep = 1
// forward: e.g. NeuralLinearRegression's "nn_output" (neural_linear_regression.py:140-157).
extern "C" int pa_mlp_copy_activation(pa_mlp* h, int32_t layer, int32_t B, float* out, int32_t ldo,
                                      void* stream) {
  PA_REQUIRE(h && out && layer >= 0 && layer + 1 < h->L, PA_ERR_INVALID,
             "pa_mlp_copy_activation: bad layer");
  PA_REQUIRE(h->kept_B == B && B > 0, PA_ERR_INVALID, "no kept forward of batch %d", B);
  const int w = h->d.dims[layer + 1];
  PA_HIP(hipMemcpy2DAsync(out, (size_t)ldo * 4, h->act[layer], (size_t)w * 4, (size_t)w * 4, (size_t)B,
                          hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
  return PA_OK;
}

// The kept activation itself (device pointer and row pitch in floats) instead of a copy: valid until
// the next forward of this network; the optimizer step does not touch it.
extern "C" int pa_mlp_activation(pa_mlp* h, int32_t layer, float** ptr_out, int32_t* ld_out) {
  PA_REQUIRE(h && ptr_out && ld_out && layer >= 0 && layer + 1 < h->L && h->act[layer], PA_ERR_INVALID,
             "pa_mlp_activation: no such kept activation");
  PA_REQUIRE(h->kept_B > 0, PA_ERR_INVALID, "pa_mlp_activation: no forward with keep = 1 precedes it");
  *ptr_out = h->act[layer];
  *ld_out = h->d.dims[layer + 1];
  return PA_OK;
}

namespace {

// dW/db of every layer of one network — or of TWO networks that share an optimizer (twin
// critics) — in one launch per three layers.  adam_step > 0: the workgroup that finishes a tile
// also applies AdamW(amsgrad) to it and refreshes the fragment-major copies the row-pass kernels
// read (the DQN treatment: no AdamW launch, no repack launch); soft_tau >= 0 on top of that: the
// target parameters (and their packed copies, when current) take their soft update in the same
// epilogue (update_target_network, common/utils.py:214-226).
long long* g_mlp_dw_prof = nullptr;   // pa_debug_mlp_dw_prof
// Live kernel timing for the bench lines (bench_algos.py, bench.py's other_configs): HIP events on
// the launch stream around the first kMlpTimed fused row-step launches (slot 0) and weight-gradient
// launches (slot 1) after pa_mlp_timing(1).  An event record costs ~6 us of GPU idle on this stack:
// the timed steps are not the ones a throughput figure is taken from.
constexpr int kMlpTimed = 64;
struct MlpTimers {
  bool on = false, made = false;
  int n[2] = {0, 0};
  hipEvent_t ev[2][kMlpTimed][2];
} g_mt;
struct MlpTimedLaunch {
  int slot, i;
  hipStream_t s;
  MlpTimedLaunch(int slot_, hipStream_t s_) : slot(slot_), i(-1), s(s_) {
    if (g_mt.on && g_mt.n[slot] < kMlpTimed) {
      i = g_mt.n[slot]++;
      (void)hipEventRecord(g_mt.ev[slot][i][0], s);
    }
  }
  ~MlpTimedLaunch() {
    if (i >= 0) (void)hipEventRecord(g_mt.ev[slot][i][1], s);
  }
};
struct DwOperands {
  const float* x; int ldx;
  const float* const* dzs; const int* ldzs;
};
// `extra` (nullable): one more X^T dZ product over the same B rows that is NOT a parameter gradient
// (DwProblem::raw); it rides the last launch when that has a free problem slot, a launch of its own
// otherwise.  Its tile0 / tiles_n are filled in here.
int run_weight_grads_n(pa_mlp* const* hs, const DwOperands* ops, int nnet, int B, int64_t adam_step,
                       float soft_tau, hipStream_t s, const TailJob* tail = nullptr,
                       const DwProblem* extra = nullptr) {
  pa_mlp* h0 = hs[0];
  const int L = h0->L;
  // 32-row tiles (weight_grad_kernel32: twice the workgroups, half the MFMA time each, bitwise the
  // same sums) when the launch would otherwise leave most of the chip idle: small batches (no
  // split-K) whose 32-row tiling still fits one workgroup per CU.  PEARL_AMD_MLP_DW_TM=64: never.
  static const int dw_tm_env = []() {
    const char* v = getenv("PEARL_AMD_MLP_DW_TM");
    return v && *v ? atoi(v) : 0;
  }();
  // up to DW_MAX_PROB problems per launch: three layers each for a pair of networks, every layer of
  // one network with up to six (the bandit's four-layer trunk + head used to be two launches, the
  // second a 17 us single-output GEMV on two workgroups)
  const int LPL = DW_MAX_PROB / nnet;
  for (int l0 = 0; l0 < L; l0 += LPL) {
    DwArgs a;
    memset(&a, 0, sizeof(a));
    int t0 = 0;
    int TM = DW_TM;
    {
      int tiles32 = 0;
      for (int ni = 0; ni < nnet; ++ni)
        for (int l = l0; l < L && l < l0 + LPL; ++l)
          tiles32 += (int)(ceil_div(hs[ni]->d.dims[l + 1], 32) * ceil_div(hs[ni]->d.dims[l], DW_TN));
      // (big batches stay on 64-row tiles: at PPO's 4096 rows the 32-row tiling is 272 workgroups,
      //  every one of them MFMA-bound — a 16-row output layer's padded tile as much as a full one —
      //  and the 16 CUs that get two finish at 45 us against 27 for the rest: no gain over 144
      //  64-row tiles at 43 us; stamps in tools/prof_rowstep.py)
      if (dw_tm_env == 32 || (dw_tm_env == 0 && B < 2048 && tiles32 + 1 <= 232)) TM = 32;
    }
    a.tm = TM;
    for (int ni = 0; ni < nnet; ++ni) {
      pa_mlp* h = hs[ni];
      for (int l = l0; l < L && l < l0 + LPL; ++l) {
        DwProblem& pr = a.p[a.nprob++];
        pr.dZ = ops[ni].dzs[l]; pr.ldz = ops[ni].ldzs[l];
        pr.X = l > 0 ? h->act[l - 1] : ops[ni].x;
        pr.ldx = l > 0 ? h->d.dims[l] : ops[ni].ldx;
        pr.dW = h->bufs.grad + h->woff[l]; pr.ldw = h->d.dims[l];
        // a bias-free last layer (NeuralLinearRegression.linear_layer_e2e) keeps a zero bias
        // slot: its column sums go to scratch so AdamW never moves it
        const bool frozen = (l == L - 1 && h->d.no_last_bias);
        pr.db = frozen ? h->db_scratch : h->bufs.grad + h->boff[l];
        pr.bias_frozen = frozen ? 1 : 0;
        pr.M = h->d.dims[l + 1]; pr.N = h->d.dims[l];
        pr.tiles_n = (int)ceil_div(h->d.dims[l], DW_TN);
        pr.tile0 = t0;
        pr.kind = 2;
        pr.net = ni;
        if (adam_step > 0) {
          pr.kind = 3;
          if (h->row_ok) {
            pr.pkf = h->wf[l]; pr.nkgf = wf16_nkg(h->d.dims[l]);
            pr.pktf = h->wtf[l]; pr.nkgtf = wf16_nkg(h->d.dims[l + 1]);
            pr.pks = h->wsp[l]; pr.nks = wsp16_nks(h->d.dims[l]);
            pr.pkts = h->wtsp[l]; pr.nkts = wsp16_nks(h->d.dims[l + 1]);
            if (soft_tau >= 0.f && h->packed_t_ok) pr.pkf_t = h->wf_t[l];
          }
          // row maxima for the fp16x2 row kernels, while something keeps them (mlp_ensure_um)
          if (l == 1 && h->um[0] && h->um_ok[0]) {
            const int M = h->d.dims[2];
            const bool vec_ok = (h->d.dims[1] & 3) == 0 && (h->woff[1] & 3) == 0;
            if (vec_ok) {
              pr.um_clear = h->um[0] + (size_t)h->um_cur[0] * M;
              h->um_cur[0] ^= 1;
              pr.um_acc = h->um[0] + (size_t)h->um_cur[0] * M;
            } else {
              h->um_ok[0] = false;
            }
            if (soft_tau >= 0.f && h->um[1] && h->um_ok[1]) {
              if (vec_ok && pr.um_acc && h->packed_t_ok && h->bufs.p_target) {
                pr.um_clear_t = h->um[1] + (size_t)h->um_cur[1] * M;
                h->um_cur[1] ^= 1;
                pr.um_acc_t = h->um[1] + (size_t)h->um_cur[1] * M;
              } else {
                h->um_ok[1] = false;
              }
            }
          } else if (l == 1 && soft_tau >= 0.f) {
            h->um_ok[1] = false;
          }
        }
        t0 += (int)ceil_div(h->d.dims[l + 1], TM) * pr.tiles_n;
      }
    }
    if (extra && l0 + LPL >= L && a.nprob < DW_MAX_PROB) {
      DwProblem& pr = a.p[a.nprob++];
      pr = *extra;
      pr.raw = 1;
      pr.tiles_n = (int)ceil_div(pr.N, DW_TN);
      pr.tile0 = t0;
      t0 += (int)ceil_div(pr.M, TM) * pr.tiles_n;
      extra = nullptr;
    }
    a.total_tiles = t0;
    a.B = B;
    a.prof = g_mlp_dw_prof;
    if (adam_step > 0) {
      a.ad.enabled = 1;
      a.ad.guard = h0->adam_guard ? h0->adam_guard : (nnet > 1 ? hs[1]->adam_guard : nullptr);
      a.ad.c = adam_scalars(h0->d, adam_step);
      a.ad.st.p = h0->bufs.p; a.ad.st.m = h0->bufs.exp_avg; a.ad.st.v = h0->bufs.exp_avg_sq;
      a.ad.st.vmax = h0->bufs.max_exp_avg_sq;
      a.ad.grad_base = h0->bufs.grad;
      if (nnet > 1) {
        pa_mlp* h1 = hs[1];
        a.net2.st.p = h1->bufs.p; a.net2.st.m = h1->bufs.exp_avg; a.net2.st.v = h1->bufs.exp_avg_sq;
        a.net2.st.vmax = h1->bufs.max_exp_avg_sq;
        a.net2.grad_base = h1->bufs.grad;
      }
      if (soft_tau >= 0.f) {
        a.ad.soft_next = 1;
        a.ad.tau = soft_tau;
        a.ad.one_minus_tau = (float)(1.0 - (double)soft_tau);
        a.ad.tgt = h0->bufs.p_target;
        if (nnet > 1) a.net2.tgt = hs[1]->bufs.p_target;
      }
    }
    // the step's scalar tail rides the last launch as one extra workgroup
    const bool with_tail = tail && l0 + LPL >= L;
    if (with_tail) a.tail = *tail;
    int rc;
    {
      MlpTimedLaunch timed(1, s);
      rc = launch_weight_grad(a, with_tail, s);
    }
    if (rc != PA_OK) return rc;
  }
  if (adam_step > 0 && soft_tau >= 0.f)
    for (int ni = 0; ni < nnet; ++ni)
      if (!hs[ni]->row_ok) hs[ni]->packed_t_ok = false;
  if (extra) {   // no free slot: its own launch
    DwArgs a;
    memset(&a, 0, sizeof(a));
    a.nprob = 1;
    a.p[0] = *extra;
    a.p[0].raw = 1;
    a.p[0].tiles_n = (int)ceil_div(extra->N, DW_TN);
    a.p[0].tile0 = 0;
    a.total_tiles = (int)ceil_div(extra->M, DW_TM) * a.p[0].tiles_n;
    a.B = B;
    int rc = launch_weight_grad(a, false, s);
    if (rc != PA_OK) return rc;
  }
  return PA_OK;
}

int run_weight_grads(pa_mlp* h, const float* x, int ldx, int B, const float* const* dzs,
                     const int* ldzs, int64_t adam_step, hipStream_t s) {
  DwOperands op = {x, ldx, dzs, ldzs};
  return run_weight_grads_n(&h, &op, 1, B, adam_step, -1.f, s);
}

void set_pending(pa_mlp* h, const float* x, int ldx, int B, const float* const* dzs, const int* ldzs) {
  h->pend.active = true;
  h->pend.x = x; h->pend.ldx = ldx; h->pend.B = B;
  for (int l = 0; l < h->L; ++l) {
    h->pend.dzs[l] = dzs[l];
    h->pend.ldzs[l] = ldzs[l];
  }
}

}  // namespace

// Backward of the kept forward.  d_out[B, d_L] is the gradient w.r.t. the network output.
// want_dw: write dW/db of every layer into bufs.grad.  d_x (nullable): gradient w.r.t. the input.
extern "C" int pa_mlp_backward(pa_mlp* h, const float* x, int32_t ldx, int32_t B, const float* d_out,
                               int32_t ldd, int32_t want_dw, float* d_x, int32_t lddx,
                               void* stream) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "mlp has no bound parameter buffers");
  PA_REQUIRE(h->kept_B == B && B > 0, PA_ERR_INVALID,
             "pa_mlp_backward: no kept forward of batch %d (kept %d)", B, h->kept_B);
  PA_REQUIRE(x && d_out, PA_ERR_INVALID, "pa_mlp_backward: null argument");
  PA_REQUIRE(!want_dw || h->bufs.grad, PA_ERR_INVALID, "no gradient buffer bound");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(h->d.device));
  const float* P = h->bufs.p;
  // ---- dX chain first: dzs[l] = gradient w.r.t. the pre-activation output of layer l
  const float* dzs[PA_MLP_MAX_LAYERS];
  int ldzs[PA_MLP_MAX_LAYERS];
  dzs[h->L - 1] = d_out;
  ldzs[h->L - 1] = ldd;
  if (h->row_ok && (h->L > 1 || d_x)) {
    // pre-activation gradients of every layer (and d_x) in one launch
    int rc = ensure_packed(h, false, s);
    if (rc != PA_OK) return rc;
    RowBwdArgs a;
    memset(&a, 0, sizeof(a));
    fill_bwd(h, d_out, ldd, d_x, lddx, a.net[0]);
    a.B = B;
    rc = launch_rowbwd(a, 1, s);
    if (rc != PA_OK) return rc;
    for (int l = h->L - 1; l > 0; --l) {
      dzs[l - 1] = h->dz[l];
      ldzs[l - 1] = h->d.dims[l];
    }
  }
  for (int l = h->L - 1; l >= 0 && !h->row_ok; --l) {
    if (l > 0 || d_x) {
      // dIn = dZ W_l (masked by relu'(in) for hidden inputs)
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = dzs[l]; g.lda = ldzs[l];
      g.Bm = P + h->woff[l]; g.ldb = h->d.dims[l];
      float* dst = l > 0 ? h->dz[l] : d_x;
      g.C = dst; g.ldc = l > 0 ? h->d.dims[l] : lddx;
      g.M = B; g.N = h->d.dims[l]; g.K = h->d.dims[l + 1];
      const bool ident = l > 0 && ((h->d.identity_layers >> (l - 1)) & 1) != 0;
      if (l > 0 && !ident && plain_net(h)) {
        g.Hmask = h->act[l - 1]; g.ldh = h->d.dims[l];
        g.epi = EPI_MASK;
      } else {
        g.epi = EPI_NONE;
      }
      int rc = launch_linear<true>(&g, 1, s);
      if (rc != PA_OK) return rc;
      if ((h->d.residual >> l) & 1) {
        // layer l's block was wrapped: its input also received d out directly
        const float* dsrc = (l == h->L - 1) ? d_out : h->dres[l];
        const int lds_ = (l == h->L - 1) ? ldd : h->d.dims[l + 1];
        hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)ceil_div((int64_t)B * h->d.dims[l], 256)), dim3(256),
                           0, s, dst, (int)g.ldc, dsrc, lds_, B, h->d.dims[l], 1);
        PA_LAUNCH_CHECK();
      }
      if (l > 0 && ((h->d.residual >> (l - 1)) & 1)) {
        // d out of hidden layer l - 1, before its own block's backward rewrites it in place
        hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)ceil_div((int64_t)B * h->d.dims[l], 256)), dim3(256),
                           0, s, h->dres[l - 1], h->d.dims[l], (const float*)dst, h->d.dims[l], B, h->d.dims[l], 0);
        PA_LAUNCH_CHECK();
      }
      if (l > 0 && ((h->d.batch_norm >> (l - 1)) & 1)) {
        // through the BatchNorm1d behind hidden layer l - 1's activation (torch's batch_norm backward)
        const int dd = h->d.dims[l];
        BnArgs b;
        memset(&b, 0, sizeof(b));
        b.y = dst; b.ldy = dd; b.hpre = h->hpre[l - 1];
        b.gamma = P + h->bn_goff[l - 1];
        b.mean = h->bn_mean[l - 1]; b.rstd = h->bn_rstd[l - 1];
        b.part = h->norm_part; b.B = B; b.d = dd;
        float* dgam = want_dw ? h->bufs.grad + h->bn_goff[l - 1] : h->bn_tmp + 2 * dd;
        float* dbet = want_dw ? h->bufs.grad + h->bn_boff[l - 1] : h->bn_tmp + 3 * dd;
        const int rows_per = (int)ceil_div(B, NP_BLOCKS);
        const int nb = (int)ceil_div(B, rows_per);
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3((unsigned)ceil_div(dd, 64), (unsigned)nb), dim3(64), 0, s, b, rows_per);
        hipLaunchKernelGGL(norm_param_sum_kernel, dim3((unsigned)ceil_div(dd, 64)), dim3(64), 0, s,
                           (const float*)h->norm_part, nb, dd, dgam, dbet);
        hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)ceil_div((int64_t)B * dd, 256)), dim3(256), 0, s,
                           b, (const float*)dgam, (const float*)dbet);
        PA_LAUNCH_CHECK();
      }
      const bool ln = l > 0 && ((h->d.layer_norm >> (l - 1)) & 1);
      const bool drop = l > 0 && h->drop_kept[l - 1] != nullptr;
      if (l > 0 && !plain_net(h) && (ln || !ident || drop)) {
        // dh -> dz of hidden layer l - 1 through its activation and LayerNorm (mlp_norm_act.hpp);
        // the LayerNorm's own parameter gradients first (they read dh)
        NormActArgs na;
        memset(&na, 0, sizeof(na));
        na.z = dst; na.ldz = h->d.dims[l];
        // the activation's OUTPUT: in front of the BatchNorm / residual add when the layer has them
        na.h = (((h->d.batch_norm | h->d.residual) >> (l - 1)) & 1) ? h->hpre[l - 1] : h->act[l - 1];
        na.ldh = h->d.dims[l];
        na.B = B; na.d = h->d.dims[l]; na.act = h->d.hidden_act; na.identity = ident ? 1 : 0;
        na.eps = 1e-5f;
        if (drop) { na.drop = h->drop_kept[l - 1]; na.ldd = h->drop_kept_ld[l - 1]; }
        if (ln) {
          na.gamma = P + h->goff[l - 1]; na.beta = P + h->betaoff[l - 1];
          na.xhat = h->xhat[l - 1]; na.rstd = h->rstd[l - 1];
          if (want_dw) {
            const int rows_per = (int)ceil_div(B, NP_BLOCKS);
            const int nb = (int)ceil_div(B, rows_per);
            hipLaunchKernelGGL(norm_param_grad_kernel, dim3((unsigned)ceil_div(na.d, 64), (unsigned)nb), dim3(64),
                               0, s, na, h->norm_part, rows_per);
            hipLaunchKernelGGL(norm_param_sum_kernel, dim3((unsigned)ceil_div(na.d, 64)), dim3(64), 0, s,
                               h->norm_part, nb, na.d, h->bufs.grad + h->goff[l - 1],
                               h->bufs.grad + h->betaoff[l - 1]);
          }
        }
        hipLaunchKernelGGL(norm_act_bwd_kernel, dim3((unsigned)ceil_div(B, NA_ROWS)), dim3(64 * NA_ROWS), 0, s, na);
        PA_LAUNCH_CHECK();
      }
      if (l > 0) {
        dzs[l - 1] = dst;
        ldzs[l - 1] = h->d.dims[l];
      }
    }
  }
  // ---- then the weight gradients (three layers per launch), now or — want_dw = 2 — together with
  // AdamW in pa_mlp_adam
  if (want_dw == 2) set_pending(h, x, ldx, B, dzs, ldzs);
  else if (want_dw) return run_weight_grads(h, x, ldx, B, dzs, ldzs, 0, s);
  return PA_OK;
}

// Q(s_b, a_i) of a state-action critic for EVERY action of an action set, q_out[b * A + i]
// (TwinCritic.get_q_values on (B, A, AD) actions, twin_critic.py:75-91 / q_value_networks.py:152-174,
// which the reference computes on the (B, A, S + AD) expansion of the states).  For the
// two-hidden-layer critics this is exactly what DQN's target_fused_kernel computes before its row
// max: the state half of layer 1 once per state (U = s W1s^T + b1, one linear launch), the action
// half, layer 2 out of one LDS tile and layer 3 per 64-row tile of (state, action) pairs — no
// (B A, S + AD) input and no hidden activations in HBM.  `rows` = number of STATES (<= max_batch).
extern "C" int pa_mlp_q_all(pa_mlp* h, int32_t use_target, const float* state, int32_t ld_state,
                            const float* rep, int64_t rep_bstride, int32_t rows, int32_t A,
                            int32_t AD, float* q_out, void* stream) {
  PA_REQUIRE(h && h->bound && state && rep && q_out && rows > 0 && A > 0 && AD > 0, PA_ERR_INVALID,
             "pa_mlp_q_all: bad argument");
  const pa_mlp_desc& d = h->d;
  const int S = d.dims[0] - AD;
  PA_REQUIRE(plain_net(h), PA_ERR_UNSUPPORTED, "pa_mlp_q_all: a plain Linear + ReLU critic only");
  PA_REQUIRE(h->L == 3 && d.dims[3] == 1 && S > 0 && d.identity_layers == 0 && !d.no_last_bias &&
                 d.dims[1] <= 256 && d.dims[2] <= 256 && A <= T_ROWS && rows <= d.max_batch,
             PA_ERR_UNSUPPORTED,
             "pa_mlp_q_all: needs a [S + AD, H1 <= 256, H2 <= 256, 1] ReLU critic, A <= %d and "
             "rows <= max_batch",
             T_ROWS);
  const float* P = use_target ? h->bufs.p_target : h->bufs.p;
  PA_REQUIRE(P, PA_ERR_INVALID, "no target parameters bound");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(d.device));
  const int H1 = d.dims[1], H2 = d.dims[2];
  // 256 x 256 hidden layers take the bf16x3 split kernel (target_split_kernel.hpp: fp32 accuracy at
  // 2.67x the fp32 matrix rate) — PEARL_AMD_TARGET_SPLIT=0: the fp32-MFMA kernel for every shape
  static const bool use_split = []() {
    const char* v = getenv("PEARL_AMD_TARGET_SPLIT");
    return !(v && *v == '0');
  }();
  const bool split = use_split && H1 == TS_H && H2 == TS_H;
  if (!h->qa_w2f) {
    PA_HIP(hipMalloc((void**)&h->qa_w2f, (size_t)w2f_floats(H2, H1) * sizeof(float)));
    PA_HIP(hipMalloc((void**)&h->qa_u, (size_t)d.max_batch * H1 * sizeof(float)));
  }
  if (split && !h->qa_w2sp) PA_HIP(hipMalloc(&h->qa_w2sp, (size_t)w2sp_bytes()));
  // the parameters may have stepped since the last call: rebuild the fragment-major W2 (3 us)
  hipLaunchKernelGGL(repack_w2_kernel, dim3(64), dim3(256), 0, s, P + h->woff[1], H2, H1, h->qa_w2f,
                     split ? h->qa_w2sp : nullptr);
  PA_LAUNCH_CHECK();
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = state; g.lda = ld_state;
  g.Bm = P + h->woff[0]; g.ldb = d.dims[0];
  g.C = h->qa_u; g.ldc = H1;
  g.bias = P + h->boff[0];
  g.M = rows; g.N = H1; g.K = S;
  g.epi = EPI_BIAS;
  int rc = launch_linear<false>(&g, 1, s);
  if (rc != PA_OK) return rc;
  TargetArgs a;
  memset(&a, 0, sizeof(a));
  a.U = h->qa_u; a.ldu = H1;
  a.feat = rep; a.feat_bstride = rep_bstride;
  a.W1a = P + h->woff[0] + S; a.ldw1 = d.dims[0];
  a.W2f = h->qa_w2f;
  a.W2sp = split ? h->qa_w2sp : nullptr;
  a.b2 = P + h->boff[1]; a.w3 = P + h->woff[2]; a.b3 = P + h->boff[2];
  a.q_all = q_out;
  a.B = rows; a.A = A; a.AD = AD; a.H1 = H1; a.H2 = H2;
  a.bpw = T_ROWS / A;
  a.ntiles = (int)ceil_div(rows, a.bpw);
  return launch_target(a, s);
}

// Both critics of a twin on every action of an action set (TwinCritic.get_q_values,
// twin_critic.py:75-91): pa_mlp_q_all twice with the launches that can be shared shared — one repack
// launch and one first-layer GEMM launch (two problems) instead of two each.
extern "C" int pa_mlp_q_all2(pa_mlp* h1, pa_mlp* h2, int32_t use_target, const float* state,
                             int32_t ld_state, const float* rep, int64_t rep_bstride, int32_t rows,
                             int32_t A, int32_t AD, float* q1_out, float* q2_out, void* stream) {
  PA_REQUIRE(h1 && h2 && h1->bound && h2->bound && state && rep && q1_out && q2_out && rows > 0 &&
                 A > 0 && AD > 0,
             PA_ERR_INVALID, "pa_mlp_q_all2: bad argument");
  pa_mlp* hs[2] = {h1, h2};
  float* outs[2] = {q1_out, q2_out};
  const pa_mlp_desc& d = h1->d;
  const int S = d.dims[0] - AD, H1 = d.dims[1], H2 = d.dims[2];
  const float* Ps[2];
  for (int i = 0; i < 2; ++i) {
    const pa_mlp_desc& di = hs[i]->d;
    PA_REQUIRE(plain_net(hs[i]), PA_ERR_UNSUPPORTED, "pa_mlp_q_all2: plain Linear + ReLU critics only");
    PA_REQUIRE(hs[i]->L == 3 && di.dims[3] == 1 && S > 0 && di.identity_layers == 0 &&
                   !di.no_last_bias && di.dims[0] == d.dims[0] && di.dims[1] == H1 &&
                   di.dims[2] == H2 && di.device == d.device && H1 <= 256 && H2 <= 256 &&
                   A <= T_ROWS && rows <= di.max_batch,
               PA_ERR_UNSUPPORTED,
               "pa_mlp_q_all2: needs two [S + AD, H1 <= 256, H2 <= 256, 1] ReLU critics of one shape, "
               "A <= %d and rows <= max_batch", T_ROWS);
    Ps[i] = use_target ? hs[i]->bufs.p_target : hs[i]->bufs.p;
    PA_REQUIRE(Ps[i], PA_ERR_INVALID, "no target parameters bound");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(d.device));
  static const bool use_split = []() {
    const char* v = getenv("PEARL_AMD_TARGET_SPLIT");
    return !(v && *v == '0');
  }();
  const bool split = use_split && H1 == TS_H && H2 == TS_H;
  for (int i = 0; i < 2; ++i) {
    pa_mlp* h = hs[i];
    if (!h->qa_w2f) {
      PA_HIP(hipMalloc((void**)&h->qa_w2f, (size_t)w2f_floats(H2, H1) * sizeof(float)));
      PA_HIP(hipMalloc((void**)&h->qa_u, (size_t)h->d.max_batch * H1 * sizeof(float)));
    }
    if (split && !h->qa_w2sp) PA_HIP(hipMalloc(&h->qa_w2sp, (size_t)w2sp_bytes()));
  }
  hipLaunchKernelGGL(repack_w2_pair_kernel, dim3(64, 2), dim3(256), 0, s, Ps[0] + h1->woff[1],
                     Ps[1] + h2->woff[1], H2, H1, h1->qa_w2f, h2->qa_w2f,
                     split ? h1->qa_w2sp : nullptr, split ? h2->qa_w2sp : nullptr);
  PA_LAUNCH_CHECK();
  GemmArgs g[2];
  memset(g, 0, sizeof(g));
  for (int i = 0; i < 2; ++i) {
    g[i].A = state; g[i].lda = ld_state;
    g[i].Bm = Ps[i] + hs[i]->woff[0]; g[i].ldb = d.dims[0];
    g[i].C = hs[i]->qa_u; g[i].ldc = H1;
    g[i].bias = Ps[i] + hs[i]->boff[0];
    g[i].M = rows; g[i].N = H1; g[i].K = S;
    g[i].epi = EPI_BIAS;
  }
  int rc = launch_linear<false>(g, 2, s);
  if (rc != PA_OK) return rc;
  for (int i = 0; i < 2; ++i) {
    pa_mlp* h = hs[i];
    TargetArgs a;
    memset(&a, 0, sizeof(a));
    a.U = h->qa_u; a.ldu = H1;
    a.feat = rep; a.feat_bstride = rep_bstride;
    a.W1a = Ps[i] + h->woff[0] + S; a.ldw1 = d.dims[0];
    a.W2f = h->qa_w2f;
    a.W2sp = split ? h->qa_w2sp : nullptr;
    a.b2 = Ps[i] + h->boff[1]; a.w3 = Ps[i] + h->woff[2]; a.b3 = Ps[i] + h->boff[2];
    a.q_all = outs[i];
    a.B = rows; a.A = A; a.AD = AD; a.H1 = H1; a.H2 = H2;
    a.bpw = T_ROWS / A;
    a.ntiles = (int)ceil_div(rows, a.bpw);
    rc = launch_target(a, s);
    if (rc != PA_OK) return rc;
  }
  return PA_OK;
}

// ---- two networks of the same depth on the same input in lock-step (TwinCritic,
// twin_critic.py:22-91; PPO's actor and critic, ppo.py:152-192): every layer of both is ONE launch
// (linear_kernel and weight_grad_kernel take several problems per launch), which halves the launch
// count of the paired passes.  Arithmetic per network is unchanged.
namespace {
int check_pair(const pa_mlp* h1, const pa_mlp* h2) {
  PA_REQUIRE(h1 && h2 && h1->bound && h2->bound, PA_ERR_INVALID, "mlp pair: unbound network");
  // same depth and the same input; widths, activations and bias flags are per network (every
  // problem of a linear_kernel / weight_grad_kernel launch carries its own shape)
  PA_REQUIRE(h1->L == h2->L && h1->d.device == h2->d.device && h1->d.dims[0] == h2->d.dims[0],
             PA_ERR_INVALID, "mlp pair: the two networks differ in depth, device or input width");
  PA_REQUIRE(plain_net(h1) && plain_net(h2), PA_ERR_UNSUPPORTED,
             "mlp pair: LayerNorm / non-ReLU hidden activations run through pa_mlp_forward / "
             "pa_mlp_backward (one network at a time)");
  return PA_OK;
}
}  // namespace

extern "C" int pa_mlp_forward2(pa_mlp* h1, pa_mlp* h2, int32_t use_target, const float* x,
                               int32_t ldx, int32_t B, float* out1, int32_t ldo1, float* out2,
                               int32_t ldo2, int32_t keep, void* stream) {
  int rc = check_pair(h1, h2);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(x && out1 && out2 && B > 0 && B <= h1->d.max_batch && B <= h2->d.max_batch,
             PA_ERR_INVALID, "pa_mlp_forward2: bad argument (B=%d)", B);
  pa_mlp* hs[2] = {h1, h2};
  float* outs[2] = {out1, out2};
  const int ldos[2] = {ldo1, ldo2};
  const float* Ps[2];
  for (int i = 0; i < 2; ++i) {
    Ps[i] = use_target ? hs[i]->bufs.p_target : hs[i]->bufs.p;
    PA_REQUIRE(Ps[i], PA_ERR_INVALID, "no target parameters bound");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(h1->d.device));
  if (h1->row_ok && h2->row_ok) {
    // both networks, every layer: one launch (blockIdx.y = network)
    RowFwdArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 2; ++i) {
      rc = ensure_packed(hs[i], use_target != 0, s);
      if (rc != PA_OK) return rc;
    }
    if (rows3_usable(hs, 2, B)) {
      rc = launch_rows3(hs, 2, use_target != 0, x, ldx, B, outs, ldos, keep != 0, s);
      if (rc != PA_OK) return rc;
      h1->kept_B = h2->kept_B = keep ? B : 0;
      return PA_OK;
    }
    for (int i = 0; i < 2; ++i) fill_fwd(hs[i], use_target != 0, outs[i], ldos[i], keep != 0, a.net[i]);
    a.x = x; a.ldx = ldx; a.B = B;
    rc = launch_rowfwd(a, 2, h1->d.dims[0], s);
    if (rc != PA_OK) return rc;
    h1->kept_B = h2->kept_B = keep ? B : 0;
    return PA_OK;
  }
  for (int l = 0; l < h1->L; ++l) {
    const bool last = (l == h1->L - 1);
    GemmArgs g[2];
    memset(g, 0, sizeof(g));
    for (int i = 0; i < 2; ++i) {
      pa_mlp* h = hs[i];
      g[i].A = l == 0 ? x : h->act[l - 1];
      g[i].lda = l == 0 ? ldx : h->d.dims[l];
      g[i].Bm = Ps[i] + h->woff[l]; g[i].ldb = h->d.dims[l];
      g[i].C = last ? outs[i] : h->act[l]; g[i].ldc = last ? ldos[i] : h->d.dims[l + 1];
      g[i].bias = Ps[i] + h->boff[l];
      g[i].M = B; g[i].N = h->d.dims[l + 1]; g[i].K = h->d.dims[l];
      const bool relu = !last && !((h->d.identity_layers >> l) & 1);
      g[i].epi = relu ? EPI_BIAS_RELU : ((last && h->d.no_last_bias) ? EPI_NONE : EPI_BIAS);
    }
    rc = launch_linear<false>(g, 2, s);
    if (rc != PA_OK) return rc;
  }
  h1->kept_B = h2->kept_B = keep ? B : 0;
  return PA_OK;
}

extern "C" int pa_mlp_backward2(pa_mlp* h1, pa_mlp* h2, const float* x, int32_t ldx, int32_t B,
                                const float* d_out1, int32_t ldd1, const float* d_out2, int32_t ldd2,
                                int32_t want_dw, float* d_x1, float* d_x2, int32_t lddx,
                                void* stream) {
  int rc = check_pair(h1, h2);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(h1->kept_B == B && h2->kept_B == B && B > 0, PA_ERR_INVALID,
             "pa_mlp_backward2: no kept forward of batch %d", B);
  PA_REQUIRE(x && d_out1 && d_out2 && (!d_x1 == !d_x2), PA_ERR_INVALID, "pa_mlp_backward2: bad argument");
  PA_REQUIRE(!want_dw || (h1->bufs.grad && h2->bufs.grad), PA_ERR_INVALID, "no gradient buffer bound");
  pa_mlp* hs[2] = {h1, h2};
  float* dxs[2] = {d_x1, d_x2};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(h1->d.device));
  const int L = h1->L;
  const float* dzs[2][PA_MLP_MAX_LAYERS];
  int ldzs[2][PA_MLP_MAX_LAYERS];
  dzs[0][L - 1] = d_out1; ldzs[0][L - 1] = ldd1;
  dzs[1][L - 1] = d_out2; ldzs[1][L - 1] = ldd2;
  const bool rowp = h1->row_ok && h2->row_ok;
  if (rowp && (L > 1 || d_x1)) {
    RowBwdArgs a;
    memset(&a, 0, sizeof(a));
    const float* douts[2] = {d_out1, d_out2};
    const int ldds[2] = {ldd1, ldd2};
    for (int i = 0; i < 2; ++i) {
      rc = ensure_packed(hs[i], false, s);
      if (rc != PA_OK) return rc;
      fill_bwd(hs[i], douts[i], ldds[i], dxs[i], lddx, a.net[i]);
      for (int l = L - 1; l > 0; --l) {
        dzs[i][l - 1] = hs[i]->dz[l];
        ldzs[i][l - 1] = hs[i]->d.dims[l];
      }
    }
    a.B = B;
    rc = launch_rowbwd(a, 2, s);
    if (rc != PA_OK) return rc;
  }
  for (int l = L - 1; l >= 0 && !rowp; --l) {
    if (l > 0 || d_x1) {
      GemmArgs g[2];
      memset(g, 0, sizeof(g));
      for (int i = 0; i < 2; ++i) {
        pa_mlp* h = hs[i];
        g[i].A = dzs[i][l]; g[i].lda = ldzs[i][l];
        g[i].Bm = h->bufs.p + h->woff[l]; g[i].ldb = h->d.dims[l];
        float* dst = l > 0 ? h->dz[l] : dxs[i];
        g[i].C = dst; g[i].ldc = l > 0 ? h->d.dims[l] : lddx;
        g[i].M = B; g[i].N = h->d.dims[l]; g[i].K = h->d.dims[l + 1];
        if (l > 0 && !((h->d.identity_layers >> (l - 1)) & 1)) {
          g[i].Hmask = h->act[l - 1]; g[i].ldh = h->d.dims[l];
          g[i].epi = EPI_MASK;
        } else {
          g[i].epi = EPI_NONE;
        }
        if (l > 0) {
          dzs[i][l - 1] = dst;
          ldzs[i][l - 1] = h->d.dims[l];
        }
      }
      rc = launch_linear<true>(g, 2, s);
      if (rc != PA_OK) return rc;
    }
  }
  for (int i = 0; i < 2 && want_dw; ++i) {
    if (want_dw == 2) {
      set_pending(hs[i], x, ldx, B, dzs[i], ldzs[i]);
    } else {
      rc = run_weight_grads(hs[i], x, ldx, B, dzs[i], ldzs[i], 0, s);
      if (rc != PA_OK) return rc;
    }
  }
  return PA_OK;
}

// ---- forward + loss head + backward of one or two networks as ONE launch (mlp_rowstep.hpp) ------
namespace {
bool rowstep_enabled() {
  static const bool on = []() {
    const char* v = getenv("PEARL_AMD_ROWSTEP");
    return !(v && *v == '0');
  }();
  return on;
}
// per-tile partial sums of both networks, the chosen-action probabilities, the ticket: per stream
// (stream_scratch, common.hpp) — two learners on two streams do not share them
// heads[i].d_out / target / ... filled by the caller; p_rows, partials and the ticket are set here
long long* g_rowstep_prof = nullptr;   // pa_debug_rowstep_prof
int g_rowstep_last_split = 0;          // 1: the last fused row step ran the bf16x3 forward
int g_rowstep_split_mode = -1;         // pa_debug_set_rowstep_split: -1 default (on), 0 off
int rowstep_split_mode() { return g_rowstep_split_mode; }
// split_out (nullable): what THIS launch ran — 0 fp32, 1 bf16x3 forward, 2 forward and backward
int run_rowstep(pa_mlp* const* hs, int nnet, const float* x, int ldx, int B, RowHead* heads,
                float* const* outs, const int* ldos, float* losses, int sum_losses, hipStream_t s,
                int* split_out = nullptr) {
  RowStepArgs a;
  memset(&a, 0, sizeof(a));
  // 32 rows per workgroup (one workgroup per CU, every weight fragment used for two row tiles)
  // once the 16-row tiling would put more than one workgroup on a CU; the (row, action) heads need
  // rows x actions <= 512 threads.  PEARL_AMD_ROWSTEP_RT=1|2 forces either.
  static const int rt_env = []() {
    const char* v = getenv("PEARL_AMD_ROWSTEP_RT");
    return v && *v ? atoi(v) : 0;
  }();
  int RT = (ceil_div(B, RP_ROWS) * nnet > 256) ? 2 : 1;
  if (rt_env == 1 || rt_env == 2) RT = rt_env;
  for (int i = 0; i < nnet; ++i)
    if (heads[i].kind != RS_HEAD_MSE && heads[i].kind != RS_HEAD_WMSE1 &&
        hs[i]->d.dims[hs[i]->L] * 2 * RP_ROWS > 512)
      RT = 1;
  const unsigned gx = (unsigned)ceil_div(B, RP_ROWS * RT);
  float* scbuf = stream_scratch(SCR_ROWSTEP, s, (size_t)4 + 4 * gx + 2 * (size_t)B);
  if (!scbuf) return PA_ERR_NOMEM;
  int rc = PA_OK;
  a.ticket = reinterpret_cast<unsigned*>(scbuf);
  a.partials = scbuf + 4;
  int d0max = 0;
  for (int i = 0; i < nnet; ++i) {
    pa_mlp* h = hs[i];
    PA_REQUIRE(h->row_ok && B > 0 && B <= h->d.max_batch, PA_ERR_UNSUPPORTED,
               "row step: network outside the row-pass kernels' shapes, or batch above max_batch");
    rc = ensure_packed(h, false, s);
    if (rc != PA_OK) return rc;
    const bool fwd_only = heads[i].kind == RS_HEAD_DSAC_TARGET;
    fill_fwd(h, false, outs[i], ldos[i], !fwd_only, a.fwd[i]);
    fill_bwd(h, nullptr, 0, nullptr, 0, a.bwd[i]);
    if (fwd_only) a.bwd[i].L = 0;
    a.head[i] = heads[i];
    a.head[i].p_rows = scbuf + 4 + 4 * gx + (size_t)i * B;
    d0max = h->d.dims[0] > d0max ? h->d.dims[0] : d0max;
  }
  a.x = x; a.ldx = ldx; a.B = B;
  a.losses = losses;
  a.sum_losses = sum_losses;
  a.prof = g_rowstep_prof;
  // 32-row launches whose every layer input is at most 256 wide run their forward GEMMs on the bf16
  // matrix pipe at fp32 accuracy (mlp_rowstep_kernel<2, true>, mlp_rowstep.hpp).
  // PEARL_AMD_ROWSTEP_SPLIT=0: the fp32-MFMA forward everywhere.
  static const bool split_env = []() {
    const char* v = getenv("PEARL_AMD_ROWSTEP_SPLIT");
    return !(v && *v == '0');
  }();
  bool splitf = split_env && RT == 2 && rowstep_split_mode() != 0;
  for (int i = 0; i < nnet && splitf; ++i)
    for (int l = 0; l < hs[i]->L; ++l)
      splitf = splitf && hs[i]->d.dims[l] <= ROW_MAX_OUT && hs[i]->wsp[l] != nullptr;
  // PPO with epsilon = 0 (the reference's default, ppo.py:105): clamp(ratio, 1, 1) passes a gradient
  // only where ratio * gae <= gae, and in the first round after preprocess_replay_buffer the
  // reference's ratio is EXACTLY 1 in every row (its minibatch forward reproduces its rollout
  // forward bit for bit: ppo_cfg4_eps0.pt records 4096 of 4096).  That needs this forward to be
  // the rollout's arithmetic — the fp32-MFMA row pass, bitwise forward_pair's — so such a step
  // does not take the bf16x3 forward (whose logits differ from the rollout's in the last bits
  // and would switch the gradient of about half the rows off).
  for (int i = 0; i < nnet; ++i)
    if (heads[i].kind == RS_HEAD_PPO && heads[i].eps == 0.f) splitf = false;
  // ... and then the backward GEMMs too (W_l^T planes, dz planes in LDS); a network without them
  // (or PEARL_AMD_ROWSTEP_SPLIT_BWD=0) keeps the fp32-MFMA backward
  static const bool splitb_env = []() {
    const char* v = getenv("PEARL_AMD_ROWSTEP_SPLIT_BWD");
    return !(v && *v == '0');
  }();
  bool splitb = splitf && splitb_env && rowstep_split_mode() != 1;
  for (int i = 0; i < nnet && splitb; ++i)
    for (int l = 1; l < hs[i]->L; ++l) splitb = splitb && hs[i]->wtsp[l] != nullptr;
  a.split_bwd = splitb ? 1 : 0;
  static size_t configured[4] = {0, 0, 0, 0};
  const int slot = splitf ? 3 : RT;
  const size_t smem = splitf ? rowstep_split_smem_bytes()
                             : (RT == 2 ? rowstep_smem_bytes_t<2>(d0max) : rowstep_smem_bytes_t<1>(d0max));
  if (smem > configured[slot]) {
    rc = splitf ? set_max_smem(mlp_rowstep_kernel<2, true>, smem)
                : (RT == 2 ? set_max_smem(mlp_rowstep_kernel<2>, smem) : set_max_smem(mlp_rowstep_kernel<1>, smem));
    if (rc != PA_OK) return rc;
    configured[slot] = smem;
  }
  {
    MlpTimedLaunch timed(0, s);
    if (splitf) hipLaunchKernelGGL((mlp_rowstep_kernel<2, true>), dim3(gx, (unsigned)nnet), dim3(512), smem, s, a);
    else if (RT == 2) hipLaunchKernelGGL(mlp_rowstep_kernel<2>, dim3(gx, (unsigned)nnet), dim3(512), smem, s, a);
    else hipLaunchKernelGGL(mlp_rowstep_kernel<1>, dim3(gx, (unsigned)nnet), dim3(512), smem, s, a);
  }
  g_rowstep_last_split = splitf ? (splitb ? 2 : 1) : 0;
  if (split_out) *split_out = g_rowstep_last_split;
  PA_LAUNCH_CHECK();
  // the state a kept forward + a want_dw = 2 backward leave behind: the weight gradients (and
  // AdamW) of the next pa_mlp_adam / pa_mlp_adam2 / pa_mlp_flush_grads2 on these networks
  for (int i = 0; i < nnet; ++i) {
    pa_mlp* h = hs[i];
    if (heads[i].kind == RS_HEAD_DSAC_TARGET) continue;   // forward only: nothing kept, nothing pending
    const int L = h->L;
    const float* dzs[PA_MLP_MAX_LAYERS];
    int ldzs[PA_MLP_MAX_LAYERS];
    dzs[L - 1] = heads[i].d_out; ldzs[L - 1] = heads[i].ldd;
    for (int l = L - 1; l > 0; --l) {
      dzs[l - 1] = h->dz[l];
      ldzs[l - 1] = h->d.dims[l];
    }
    h->kept_B = B;
    set_pending(h, x, ldx, B, dzs, ldzs);
  }
  return PA_OK;
}
}  // namespace

// HIP-event timing of the engine's fused row-step and weight-gradient launches (see MlpTimers)
extern "C" int pa_mlp_timing(int32_t enable) {
  if (enable && !g_mt.made) {
    for (int k = 0; k < 2; ++k)
      for (int i = 0; i < kMlpTimed; ++i)
        for (int e = 0; e < 2; ++e) PA_HIP(hipEventCreate(&g_mt.ev[k][i][e]));
    g_mt.made = true;
  }
  g_mt.on = enable != 0;
  g_mt.n[0] = g_mt.n[1] = 0;
  return PA_OK;
}
// which: 0 = fused row step, 1 = weight gradients (+ AdamW); average duration over the launches
// timed so far; synchronises with their events
extern "C" int pa_mlp_timing_read(int32_t which, double* avg_us, int64_t* launches) {
  PA_REQUIRE((which == 0 || which == 1) && avg_us && launches, PA_ERR_INVALID,
             "pa_mlp_timing_read: bad argument");
  double sum = 0.0;
  for (int i = 0; i < g_mt.n[which]; ++i) {
    float ms = 0.f;
    PA_HIP(hipEventSynchronize(g_mt.ev[which][i][1]));
    PA_HIP(hipEventElapsedTime(&ms, g_mt.ev[which][i][0], g_mt.ev[which][i][1]));
    sum += ms;
  }
  *launches = g_mt.n[which];
  *avg_us = g_mt.n[which] ? 1e3 * sum / g_mt.n[which] : 0.0;
  return PA_OK;
}

// the same for the engine's weight-gradient launches (weight_grad_kernel's stamps, [workgroup][8][16])
extern "C" int pa_debug_mlp_dw_prof(long long* stamps) {
  g_mlp_dw_prof = stamps;
  return PA_OK;
}
// which GEMMs the fused row steps take: -1 = default (bf16x3 forward and backward where eligible),
// 0 = fp32 MFMA, 1 = bf16x3 forward with the fp32-MFMA backward;
// pa_rowstep_last_split: what the most recent fused row step ran — 0 fp32, 1 bf16x3 forward, 2 both
extern "C" int pa_debug_set_rowstep_split(int32_t mode) {
  PA_REQUIRE(mode >= -1 && mode <= 1, PA_ERR_INVALID, "pa_debug_set_rowstep_split: mode is -1, 0 or 1");
  g_rowstep_split_mode = mode;
  return PA_OK;
}
extern "C" int pa_rowstep_last_split(void) { return g_rowstep_last_split; }

// tuning aid (tools/prof_rowstep.py): in-kernel phase stamps of the next fused row-step launches,
// [workgroup][8 waves][16] wall-clock ticks; NULL switches them off
extern "C" int pa_debug_rowstep_prof(long long* stamps) {
  g_rowstep_prof = stamps;
  return PA_OK;
}

// 1 when pa_ppo_rowstep / pa_mse_rowstep2 take these networks (shapes the row-pass kernels cover;
// PEARL_AMD_ROWSTEP=0: never): callers fall back to forward2 -> heads -> backward2 otherwise.
extern "C" int pa_rowstep_supported(const pa_mlp* h1, const pa_mlp* h2, int32_t ppo_actions) {
  if (rowstep_enabled() && h1 && !h2 && ppo_actions > 0)   // one softmax actor (discrete SAC)
    return h1->bound && h1->row_ok && h1->d.dims[h1->L] == ppo_actions && ppo_actions <= 32;
  if (rowstep_enabled() && h1 && !h2 && ppo_actions == 0)  // one single-output network (bandit)
    return h1->bound && h1->row_ok && h1->d.dims[h1->L] == 1;
  if (!rowstep_enabled() || !h1 || !h2 || !h1->bound || !h2->bound) return 0;
  if (!h1->row_ok || !h2->row_ok || h1->L != h2->L || h1->d.dims[0] != h2->d.dims[0]) return 0;
  if (ppo_actions > 0 && (h1->d.dims[h1->L] != ppo_actions || ppo_actions > 32 ||
                          h2->d.dims[h2->L] != 1))
    return 0;
  if (ppo_actions == 0 && (h1->d.dims[h1->L] != 1 || h2->d.dims[h2->L] != 1)) return 0;
  return 1;
}

// One PPO minibatch step down to the pre-activation gradients (ppo.py:152-192 on
// actor_critic_base.py:309-366): actor and critic forward (activations kept), the surrogate's and
// the value loss's row-local gradients, both backward passes — ONE launch.  Leaves both networks'
// weight gradients pending like pa_mlp_backward2(want_dw = 2): follow with pa_mlp_adam2 (or
// pa_mlp_flush_grads2 -> all-reduce -> pa_mlp_adamw2).  d_value = value_grad_scale (v - target);
// losses[0] = actor loss, losses[1] = critic loss (mean squared error).
extern "C" int pa_ppo_rowstep(pa_mlp* actor, pa_mlp* critic, const float* x, int32_t ldx, int32_t B,
                              const float* action_rep, int32_t lda, const float* p_old,
                              const float* gae, float epsilon, float entropy_scale,
                              const float* value_target, float value_grad_scale, float* logits_out,
                              int32_t ldl, float* value_out, int32_t ldv, float* d_logits,
                              int32_t ldd, float* d_value, float* losses, void* stream) {
  PA_REQUIRE(actor && critic && x && action_rep && p_old && gae && value_target && d_logits &&
                 d_value && losses && B > 0,
             PA_ERR_INVALID, "pa_ppo_rowstep: bad argument");
  PA_REQUIRE(pa_rowstep_supported(actor, critic, actor->d.dims[actor->L]), PA_ERR_UNSUPPORTED,
             "pa_ppo_rowstep: needs an actor with <= 32 outputs and a one-output critic of the same "
             "depth and input width, every layer <= 256 wide");
  PA_HIP(hipSetDevice(actor->d.device));
  pa_mlp* hs[2] = {actor, critic};
  RowHead heads[2];
  memset(heads, 0, sizeof(heads));
  heads[0].kind = RS_HEAD_PPO;
  heads[0].d_out = d_logits; heads[0].ldd = ldd;
  heads[0].arep = action_rep; heads[0].lda = lda; heads[0].p_old = p_old; heads[0].gae = gae;
  heads[0].eps = epsilon; heads[0].ent_scale = entropy_scale;
  heads[1].kind = RS_HEAD_MSE;
  heads[1].d_out = d_value; heads[1].ldd = 1;
  heads[1].target = value_target; heads[1].grad_scale = value_grad_scale; heads[1].loss_scale = 1.0f;
  float* outs[2] = {logits_out, value_out};
  const int ldos[2] = {ldl, ldv};
  return run_rowstep(hs, 2, x, ldx, B, heads, outs, ldos, losses, 0,
                     reinterpret_cast<hipStream_t>(stream));
}

// Discrete SoftActorCritic's actor step down to the pre-activation gradients
// (soft_actor_critic.py:254-287): forward (kept) -> softmax policy loss against min(q1, q2) on every
// action -> backward, one launch; what pa_mlp_forward(keep) -> pa_dsac_actor_head -> pa_mlp_backward
// (want_dw = 2) compute.  The weight gradients stay pending for pa_mlp_adam.
extern "C" int pa_dsac_actor_rowstep(pa_mlp* actor, const float* x, int32_t ldx, int32_t B,
                                     const float* q1, const float* q2, const uint8_t* mask,
                                     const float* alpha, float* d_logits, int32_t ldd, float* h_out,
                                     float* loss_out, void* stream) {
  PA_REQUIRE(actor && x && q1 && q2 && alpha && d_logits && h_out && loss_out && B > 0,
             PA_ERR_INVALID, "pa_dsac_actor_rowstep: bad argument");
  PA_REQUIRE(pa_rowstep_supported(actor, nullptr, actor->d.dims[actor->L]), PA_ERR_UNSUPPORTED,
             "pa_dsac_actor_rowstep: needs an actor with <= 32 outputs, every layer <= 256 wide");
  PA_HIP(hipSetDevice(actor->d.device));
  pa_mlp* hs[1] = {actor};
  RowHead head;
  memset(&head, 0, sizeof(head));
  head.kind = RS_HEAD_DSAC_ACTOR;
  head.d_out = d_logits; head.ldd = ldd;
  head.q1 = q1; head.q2 = q2; head.mask = mask; head.alpha = alpha; head.h_out = h_out;
  float* outs[1] = {nullptr};
  const int ldos[1] = {0};
  return run_rowstep(hs, 1, x, ldx, B, &head, outs, ldos, loss_out, 1,
                     reinterpret_cast<hipStream_t>(stream));
}
// The expected next-state value under the policy and the Bellman target
// (soft_actor_critic.py:180-252): actor forward on the next states -> y, one launch, nothing kept.
extern "C" int pa_dsac_target_rowstep(pa_mlp* actor, const float* next_state, int32_t ldx, int32_t B,
                                      const float* q1, const float* q2, const uint8_t* mask,
                                      const float* alpha, const float* reward, const uint8_t* term,
                                      float gamma, float* y, void* stream) {
  PA_REQUIRE(actor && next_state && q1 && q2 && alpha && reward && term && y && B > 0,
             PA_ERR_INVALID, "pa_dsac_target_rowstep: bad argument");
  PA_REQUIRE(pa_rowstep_supported(actor, nullptr, actor->d.dims[actor->L]), PA_ERR_UNSUPPORTED,
             "pa_dsac_target_rowstep: needs an actor with <= 32 outputs, every layer <= 256 wide");
  PA_HIP(hipSetDevice(actor->d.device));
  pa_mlp* hs[1] = {actor};
  RowHead head;
  memset(&head, 0, sizeof(head));
  head.kind = RS_HEAD_DSAC_TARGET;
  head.q1 = q1; head.q2 = q2; head.mask = mask; head.alpha = alpha;
  head.reward = reward; head.term = term; head.gamma = gamma; head.y = y;
  float* outs[1] = {nullptr};
  const int ldos[1] = {0};
  return run_rowstep(hs, 1, next_state, ldx, B, &head, outs, ldos, nullptr, 1,
                     reinterpret_cast<hipStream_t>(stream));
}

// The neural-linear bandit's network step with unit weights (neural_linear_bandit.py:176-199):
// forward (kept: pa_mlp_copy_activation still serves the features) -> the criterion's gradient
// (d_pred = 2 (pred - y) / B for MSE) -> backward, one launch; loss_out[0] = mean criterion.  What
// pa_mlp_forward(keep) -> pa_weighted_loss_head(w = NULL) -> pa_mlp_backward(want_dw = 2) compute.
extern "C" int pa_wloss_rowstep(pa_mlp* net, const float* x, int32_t ldx, int32_t B, const float* y,
                                int32_t loss_kind, int32_t out_act, float* pred_pre_out,
                                float* pred_out, float* d_pred, float* loss_out, void* stream) {
  PA_REQUIRE(net && x && y && d_pred && loss_out && B > 0, PA_ERR_INVALID,
             "pa_wloss_rowstep: bad argument");
  PA_REQUIRE(loss_kind >= PA_LOSS_MSE && loss_kind <= PA_LOSS_BCE && out_act >= PA_OUT_LINEAR &&
             out_act <= PA_OUT_SIGMOID, PA_ERR_UNSUPPORTED,
             "pa_wloss_rowstep: loss is mse / mae / cross-entropy, output activation linear / sigmoid");
  PA_REQUIRE(pa_rowstep_supported(net, nullptr, 0), PA_ERR_UNSUPPORTED,
             "pa_wloss_rowstep: needs a one-output network, every layer <= 256 wide");
  PA_HIP(hipSetDevice(net->d.device));
  pa_mlp* hs[1] = {net};
  RowHead head;
  memset(&head, 0, sizeof(head));
  head.kind = RS_HEAD_WMSE1;
  head.d_out = d_pred; head.ldd = 1;
  head.target = y;
  head.loss_kind = loss_kind; head.out_act = out_act;
  head.out_post = out_act == PA_OUT_LINEAR ? nullptr : pred_out;
  // [B] network outputs (pre-activation); with a linear output activation they ARE the predictions
  float* outs[1] = {out_act == PA_OUT_LINEAR && !pred_pre_out ? pred_out : pred_pre_out};
  const int ldos[1] = {1};
  return run_rowstep(hs, 1, x, ldx, B, &head, outs, ldos, loss_out, 1,
                     reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_wmse_rowstep(pa_mlp* net, const float* x, int32_t ldx, int32_t B, const float* y,
                               float* pred_out, float* d_pred, float* loss_out, void* stream) {
  return pa_wloss_rowstep(net, x, ldx, B, y, PA_LOSS_MSE, PA_OUT_LINEAR, nullptr, pred_out, d_pred,
                          loss_out, stream);
}

// Twin critics on (state, action) rows against one target (twin_critic_action_value_loss,
// critic_utils.py:170-203): forward (kept), d_q_i = grad_scale (q_i - target), backward — one
// launch; loss_out[0] = loss_scale (mse_1 + mse_2).  Weight gradients pending as above.
extern "C" int pa_mse_rowstep2(pa_mlp* c1, pa_mlp* c2, const float* x, int32_t ldx, int32_t B,
                               const float* target, float grad_scale, float loss_scale,
                               float* q1_out, float* q2_out, float* d_q1, float* d_q2,
                               float* loss_out, void* stream) {
  PA_REQUIRE(c1 && c2 && x && target && d_q1 && d_q2 && loss_out && B > 0, PA_ERR_INVALID,
             "pa_mse_rowstep2: bad argument");
  PA_REQUIRE(pa_rowstep_supported(c1, c2, 0), PA_ERR_UNSUPPORTED,
             "pa_mse_rowstep2: needs two one-output networks of the same depth and input width, "
             "every layer <= 256 wide");
  PA_HIP(hipSetDevice(c1->d.device));
  pa_mlp* hs[2] = {c1, c2};
  RowHead heads[2];
  memset(heads, 0, sizeof(heads));
  float* dqs[2] = {d_q1, d_q2};
  for (int i = 0; i < 2; ++i) {
    heads[i].kind = RS_HEAD_MSE;
    heads[i].d_out = dqs[i]; heads[i].ldd = 1;
    heads[i].target = target; heads[i].grad_scale = grad_scale; heads[i].loss_scale = loss_scale;
  }
  float* outs[2] = {q1_out, q2_out};
  const int ldos[2] = {1, 1};
  return run_rowstep(hs, 2, x, ldx, B, heads, outs, ldos, loss_out, 1,
                     reinterpret_cast<hipStream_t>(stream));
}

// Weight gradients a want_dw = 2 backward left pending, without the optimizer (data parallel: the
// gradient is all-reduced before AdamW).
extern "C" int pa_mlp_flush_grads(pa_mlp* h, void* stream) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "mlp has no bound parameter buffers");
  if (!h->pend.active) return PA_OK;
  PA_HIP(hipSetDevice(h->d.device));
  h->pend.active = false;
  return run_weight_grads(h, h->pend.x, h->pend.ldx, h->pend.B, h->pend.dzs, h->pend.ldzs, 0,
                          reinterpret_cast<hipStream_t>(stream));
}

// optim.AdamW(amsgrad) step `step` (1-based) on bufs.grad.
extern "C" int pa_mlp_adam(pa_mlp* h, int64_t step, void* stream) {
  PA_REQUIRE(h && h->bound && h->bufs.grad && h->bufs.exp_avg && h->bufs.exp_avg_sq,
             PA_ERR_INVALID, "pa_mlp_adam: optimizer buffers not bound");
  PA_REQUIRE(!h->d.amsgrad || h->bufs.max_exp_avg_sq, PA_ERR_INVALID, "amsgrad needs max_exp_avg_sq");
  PA_REQUIRE(step >= 1, PA_ERR_INVALID, "adam step must be >= 1");
  PA_HIP(hipSetDevice(h->d.device));
  if (h->pend.active) {
    // the kept backward's weight gradients and this step in one pass; the fragment-major copies
    // stay current (refreshed by the same epilogue) if they were current before
    h->pend.active = false;
    int rc = run_weight_grads(h, h->pend.x, h->pend.ldx, h->pend.B, h->pend.dzs, h->pend.ldzs, step,
                              reinterpret_cast<hipStream_t>(stream));
    if (rc != PA_OK || h->norm0 >= h->P) return rc;
    // the LayerNorm parameters (their gradients were formed by the backward): AdamW on the block
    // behind W / b
    AdamArgs n;
    memset(&n, 0, sizeof(n));
    const int64_t o = h->norm0;
    n.st.p = h->bufs.p + o; n.st.m = h->bufs.exp_avg + o; n.st.v = h->bufs.exp_avg_sq + o;
    n.st.vmax = h->bufs.max_exp_avg_sq ? h->bufs.max_exp_avg_sq + o : nullptr;
    n.g = h->bufs.grad + o;
    n.n = h->P - o;
    n.c = adam_scalars(h->d, step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)ceil_div(n.n, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), n);
    PA_LAUNCH_CHECK();
    return PA_OK;
  }
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  a.st.p = h->bufs.p; a.st.m = h->bufs.exp_avg; a.st.v = h->bufs.exp_avg_sq;
  a.st.vmax = h->bufs.max_exp_avg_sq;
  a.g = h->bufs.grad;
  a.n = h->P;
  a.c = adam_scalars(h->d, step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)ceil_div(h->P, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  h->packed_ok = false;
  return PA_OK;
}

// update_target_network (common/utils.py:214-226)
extern "C" int pa_mlp_soft_update(pa_mlp* h, float tau, void* stream) {
  PA_REQUIRE(h && h->bound && h->bufs.p_target, PA_ERR_INVALID, "no target parameters bound");
  PA_HIP(hipSetDevice(h->d.device));
  if (h->row_ok && h->packed_t_ok) {
    RowSoftArgs a;
    memset(&a, 0, sizeof(a));
    a.tgt = h->bufs.p_target; a.src = h->bufs.p; a.n = h->P;
    a.tau = tau; a.one_minus_tau = (float)(1.0 - (double)tau);
    a.L = h->L;
    for (int l = 0; l <= h->L; ++l) a.dims[l] = h->d.dims[l];
    for (int l = 0; l < h->L; ++l) {
      a.woff[l] = h->woff[l];
      a.Wf[l] = h->wf_t[l];
    }
    hipLaunchKernelGGL(mlp_soft_update_kernel, dim3((unsigned)ceil_div(h->P, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    PA_LAUNCH_CHECK();
    h->um_ok[1] = false;
    return PA_OK;     // the target's packed copies were refreshed in place
  }
  hipLaunchKernelGGL(soft_update_kernel, dim3((unsigned)ceil_div(h->P, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), h->bufs.p_target, h->bufs.p, h->P, tau,
                     (float)(1.0 - (double)tau));
  PA_LAUNCH_CHECK();
  h->packed_t_ok = false;
  return PA_OK;
}

// =============================================================================================
// heads
// =============================================================================================
namespace {

// block-wide sum in a fixed order (256 threads)
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// softmax over A logits of one row, by one thread (A is an action count: small)
__device__ __forceinline__ float softmax_row_prob(const float* __restrict__ z, int A,
                                                  const float* __restrict__ arep, float* probs_out,
                                                  float* sum_out) {
  float m = z[0];
  for (int j = 1; j < A; ++j) m = fmaxf(m, z[j]);
  float s = 0.f;
  for (int j = 0; j < A; ++j) s += expf(z[j] - m);
  float p = 0.f;
  for (int j = 0; j < A; ++j) {
    const float pj = expf(z[j] - m) / s;
    if (probs_out) probs_out[j] = pj;
    p += pj * arep[j];
  }
  if (sum_out) *sum_out = s;
  return p;
}

// VanillaActorNetwork.get_action_prob (actor_networks.py:155-176): softmax(logits) . action_rep
__global__ __launch_bounds__(256) void action_prob_kernel(const float* __restrict__ logits, int ldl,
                                                          const float* __restrict__ arep, int lda,
                                                          int B, int A, float* __restrict__ probs,
                                                          float* __restrict__ aprob) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  aprob[b] = softmax_row_prob(logits + (int64_t)b * ldl, A, arep + (int64_t)b * lda,
                              probs ? probs + (int64_t)b * A : nullptr, nullptr);
}

// PPO clipped-surrogate actor loss and its gradient w.r.t. the logits (ppo.py:152-183).
// The gradient is row-local; only the reported loss needs batch-wide sums (the entropy bonus treats
// the B chosen-action probabilities as ONE categorical distribution — a detached scalar — which
// needs their batch sum first).  One workgroup per 256 rows writes its partial sums and its rows'
// probabilities; the last workgroup to arrive (ticket) adds the partials in block order and forms
// the entropy term — deterministic, and 4096 rows no longer run on one CU (307 us -> ~10 us).
// MSELoss(mean) head: loss = mean((pred - target)^2) * scale_loss, d_pred = grad_scale * (pred - target)
struct MseArgs {
  const float* pred; int ldp;
  const float* target;
  int B;
  float grad_scale;   // 2 / B (single critic) or 1 / B (each of the twin critics: (mse1 + mse2) / 2)
  float* d_pred;
  float* loss_out;    // += or = mean squared error * loss_scale
  float loss_scale;
  int accumulate;
};
__device__ __forceinline__ void mse_head_body(const MseArgs& a, float* red) {
  float part = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float d = __fsub_rn(a.pred[(int64_t)b * a.ldp], a.target[b]);
    part += d * d;
    if (a.d_pred) a.d_pred[b] = __fmul_rn(a.grad_scale, d);
  }
  const float s = block_sum_256(part, red);
  if (threadIdx.x == 0 && a.loss_out) {
    const float v = (s / (float)a.B) * a.loss_scale;
    a.loss_out[0] = a.accumulate ? a.loss_out[0] + v : v;
  }
}

struct PpoActorArgs {
  MseArgs critic;       // has_critic: the value head of the same minibatch rides the launch as one
  int has_critic;       // extra workgroup (the LAST block; pa_ppo_heads)
  const float* logits; int ldl;
  const float* arep; int lda;
  const float* p_old; const float* gae;
  int B, A;
  float eps, ent_scale;
  float* d_logits; int ldd;
  float* loss_out;
  float* p_rows;        // [B] scratch: chosen-action probability of every row
  float* partials;      // [2 * gridDim] scratch: per-block sums of (-min term, p)
  unsigned* ticket;     // zero on entry, zero again on exit
};

// ---- element-parallel form (A <= 256): one thread per (row, action) -------------------------
// The row-per-thread kernel below is instruction-issue bound — ~300 instructions per logit on ONE
// wave per SIMD of 16 CUs, 21 us for 4096 x 16 — so here a workgroup covers 256 / A rows, every
// thread owns one logit, and the per-row reductions (max, sum of exp, chosen-action probability)
// are serial loops over the row's LDS slots in action order: the same values in the same order as
// the serial loop, on 16x the lanes.
__global__ __launch_bounds__(256) void ppo_actor_elem_kernel(PpoActorArgs a) {
  __shared__ float va[256], vb[256], red[256];
  __shared__ unsigned last;
  const unsigned nact = gridDim.x - (a.has_critic ? 1u : 0u);   // blocks of the actor head
  if (blockIdx.x == nact) {
    mse_head_body(a.critic, red);
    return;
  }
  const float lo = 1.0f - a.eps, hi = 1.0f + a.eps;
  const int rpw = 256 / a.A;
  const int r = threadIdx.x / a.A, j = threadIdx.x - r * a.A;
  const int b = blockIdx.x * rpw + r;
  const bool live = r < rpw && b < a.B;
  const float z = live ? a.logits[(int64_t)b * a.ldl + j] : 0.f;
  const float ar = live ? a.arep[(int64_t)b * a.lda + j] : 0.f;
  const float g = live ? a.gae[b] : 0.f;
  const float pold = live ? a.p_old[b] : 1.f;
  const int base = r * a.A;
  va[threadIdx.x] = z;
  __syncthreads();
  float m = 0.f, s = 0.f, p = 0.f;
  if (live) {
    m = va[base];
    for (int k = 1; k < a.A; ++k) m = fmaxf(m, va[base + k]);
  }
  const float e = expf(z - m);
  vb[threadIdx.x] = e;
  __syncthreads();
  if (live)
    for (int k = 0; k < a.A; ++k) s += vb[base + k];
  const float y = live ? e / s : 0.f;
  va[threadIdx.x] = y * ar;       // every thread is past its reads of va (barrier above)
  __syncthreads();
  float part_loss = 0.f, part_p = 0.f;
  if (live) {
    for (int k = 0; k < a.A; ++k) p += va[base + k];
    const float rt = p / pold;
    const float clip = fminf(fmaxf(rt, lo), hi);
    const float s1 = rt * g, s2 = clip * g;
    const float inr = (rt >= lo && rt <= hi) ? 1.f : 0.f;
    float dr;
    if (s1 < s2) dr = g;
    else if (s1 > s2) dr = g * inr;
    else dr = 0.5f * g + 0.5f * g * inr;
    const float dp = -dr / pold;
    const float dot = dp * p;
    a.d_logits[(int64_t)b * a.ldd + j] = y * (dp * ar - dot);
    if (j == 0) {
      part_loss = -fminf(s1, s2);
      part_p = p;
      a.p_rows[b] = p;
    }
  }
  const float bl = block_sum_256(part_loss, red);
  const float bp = block_sum_256(part_p, red);
  if (threadIdx.x == 0) {
    a.partials[2 * blockIdx.x] = bl;
    a.partials[2 * blockIdx.x + 1] = bp;
    __threadfence();
    last = (atomicAdd(a.ticket, 1u) == nact - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();  // the other blocks' partials and p_rows are visible from here on
  float pl = 0.f, pp = 0.f;
  for (unsigned k = threadIdx.x; k < nact; k += 256) {   // fixed order: strided, then the tree
    pl += __builtin_nontemporal_load(a.partials + 2 * k);
    pp += __builtin_nontemporal_load(a.partials + 2 * k + 1);
  }
  const float loss = block_sum_256(pl, red);
  const float psum = block_sum_256(pp, red);
  float part_e = 0.f;
  const float tiny = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
#pragma unroll 8
  for (int i = threadIdx.x; i < a.B; i += 256) {
    const float pr = __builtin_nontemporal_load(a.p_rows + i);
    const float pn = pr / psum;
    const float pc = fminf(fmaxf(pn, tiny), 1.0f - tiny);
    float lg = logf(pc);
    lg = fmaxf(lg, -3.4028234663852886e+38f);
    part_e += lg * pn;
  }
  const float ent = -block_sum_256(part_e, red);
  if (threadIdx.x == 0) {
    a.loss_out[0] = loss - a.ent_scale * ent;
    *a.ticket = 0u;
  }
}

// STAGED (2 * 256 * (A + 1) floats of LDS fit): the workgroup's 256 logit / representation rows are
// staged through LDS with coalesced loads (pitch A + 1: conflict-free row-per-lane access), the
// row-per-thread arithmetic — unchanged, same operation order — runs out of LDS, and the logit
// gradients go back the same way.  Row-per-thread global access touches 64 cache lines per load.
template <bool STAGED>
__global__ __launch_bounds__(256) void ppo_actor_kernel(PpoActorArgs a) {
  extern __shared__ __attribute__((aligned(16))) float stage[];
  __shared__ float red[256];
  __shared__ unsigned last;
  const float lo = 1.0f - a.eps, hi = 1.0f + a.eps;
  float part_loss = 0.f, part_p = 0.f;
  const int b0 = blockIdx.x * 256;
  const int b = b0 + threadIdx.x;
  const int P = a.A + 1;
  if (STAGED) {
    for (int e = threadIdx.x; e < 256 * a.A; e += 256) {
      const int r = e / a.A, j = e - r * a.A;
      const bool in = b0 + r < a.B;
      stage[r * P + j] = in ? a.logits[(int64_t)(b0 + r) * a.ldl + j] : 0.f;
      stage[(256 + r) * P + j] = in ? a.arep[(int64_t)(b0 + r) * a.lda + j] : 0.f;
    }
    __syncthreads();
  }
  if (b < a.B) {
    // STAGED: the row's exp / softmax values replace its logits in LDS as they are formed, so each
    // is computed once (same values, same summation order as recomputing them in every pass — and a
    // third of the transcendental work on what is a one-wave-per-SIMD, latency-bound kernel)
    float* zs = stage + threadIdx.x * P;
    const float* z = STAGED ? zs : a.logits + (int64_t)b * a.ldl;
    const float* ar = STAGED ? stage + (256 + threadIdx.x) * P : a.arep + (int64_t)b * a.lda;
    float m = z[0];
    for (int j = 1; j < a.A; ++j) m = fmaxf(m, z[j]);
    float s = 0.f;
    for (int j = 0; j < a.A; ++j) {
      const float e = expf(z[j] - m);
      if (STAGED) zs[j] = e;
      s += e;
    }
    float p = 0.f;
    for (int j = 0; j < a.A; ++j) {
      const float yj = (STAGED ? zs[j] : expf(z[j] - m)) / s;
      if (STAGED) zs[j] = yj;
      p += yj * ar[j];
    }
    const float g = a.gae[b];
    const float r = p / a.p_old[b];
    const float clip = fminf(fmaxf(r, lo), hi);
    const float s1 = r * g, s2 = clip * g;
    part_loss = -fminf(s1, s2);
    part_p = p;
    a.p_rows[b] = p;
    // d(-min(s1, s2))/dr: torch.minimum splits ties evenly; clamp passes the gradient inside
    // [lo, hi] (bounds included)
    const float inr = (r >= lo && r <= hi) ? 1.f : 0.f;
    float dr;
    if (s1 < s2) dr = g;
    else if (s1 > s2) dr = g * inr;
    else dr = 0.5f * g + 0.5f * g * inr;
    const float dp = -dr / a.p_old[b];
    // p = sum_j y_j ar_j, y = softmax(z): dz_j = y_j (dp ar_j - sum_k dp ar_k y_k)
    const float dot = dp * p;
    // STAGED: in place over the row's softmax values
    float* dz = STAGED ? zs : a.d_logits + (int64_t)b * a.ldd;
    for (int j = 0; j < a.A; ++j) {
      const float yj = STAGED ? zs[j] : expf(z[j] - m) / s;
      dz[j] = yj * (dp * ar[j] - dot);
    }
  }
  if (STAGED) {
    __syncthreads();
    for (int e = threadIdx.x; e < 256 * a.A; e += 256) {
      const int r = e / a.A, j = e - r * a.A;
      if (b0 + r < a.B) a.d_logits[(int64_t)(b0 + r) * a.ldd + j] = stage[r * P + j];
    }
  }
  const float bl = block_sum_256(part_loss, red);
  const float bp = block_sum_256(part_p, red);
  if (threadIdx.x == 0) {
    a.partials[2 * blockIdx.x] = bl;
    a.partials[2 * blockIdx.x + 1] = bp;
    __threadfence();
    last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();  // the other blocks' partials and p_rows are visible from here on
  float loss = 0.f, psum = 0.f;
  const float* parts = a.partials;
#pragma unroll 8
  for (unsigned k = 0; k < gridDim.x; ++k) {  // block order: fixed
    loss += __builtin_nontemporal_load(parts + 2 * k);
    psum += __builtin_nontemporal_load(parts + 2 * k + 1);
  }
  // entropy of Categorical(probs = p / sum p) with torch's clamping of probs / logits
  float part_e = 0.f;
  const float tiny = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
  // plain loads: the acquire fence above ordered them after every block's release, and unlike
  // atomic loads the compiler keeps many of them in flight (16 dependent L2 round trips otherwise)
  const float* prow = a.p_rows;
#pragma unroll 8
  for (int i = threadIdx.x; i < a.B; i += 256) {
    const float p = __builtin_nontemporal_load(prow + i);
    const float pn = p / psum;
    const float pc = fminf(fmaxf(pn, tiny), 1.0f - tiny);
    float lg = logf(pc);
    lg = fmaxf(lg, -3.4028234663852886e+38f);
    part_e += lg * pn;
  }
  const float ent = -block_sum_256(part_e, red);
  if (threadIdx.x == 0) {
    a.loss_out[0] = loss - a.ent_scale * ent;
    *a.ticket = 0u;
  }
}

__global__ __launch_bounds__(256) void mse_head_kernel(MseArgs a) {
  __shared__ float red[256];
  mse_head_body(a, red);
}

// GAE / truncated lambda return (ppo.py:271-293).  The reference walks the rollout from the newest
// transition to the oldest; gae_i = td_i + gamma*lambda*[not (term_i or trunc_i)] * gae_{i+1}.
// The recurrence restarts wherever that factor is 0, so every such boundary starts an independent
// backward walk: thread i is active iff it is a boundary (or the newest transition) and walks back
// to the previous boundary — the same sequential fp32 arithmetic as the reference, in parallel
// across episodes.  Index i is the LOGICAL index (0 = oldest).
struct GaeArgs {
  const float* reward; const uint8_t* term; const uint8_t* trunc;
  const float* values;           // [N] critic(state_i)
  const float* next_value_last;  // critic(next_state of the newest transition), device scalar
  float gamma; float gl;         // gamma, (float)(gamma * lambda)
  int64_t N;
  float* gae; float* lam_return;
};
__global__ __launch_bounds__(256) void gae_kernel(GaeArgs a) {
  const int64_t i1 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i1 >= a.N) return;
  const bool boundary = (i1 == a.N - 1) | (((unsigned)a.term[i1] | (unsigned)a.trunc[i1]) != 0);
  if (!boundary) return;
  // The walk is a chain of dependent fp32 operations, but nothing it LOADS depends on the chain:
  // first find where the walk ends (the previous boundary; four flag pairs per trip instead of one
  // dependent load per element), then run it in blocks of four whose loads are all issued before
  // the block's arithmetic — 0.42 us per element (one exposed memory latency each) otherwise,
  // 210 us for 500-transition episodes.  The arithmetic and its order are unchanged.
  int64_t lo = i1;          // first element of this walk (inclusive)
  while (lo > 0) {
    bool d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = lo - 1 - k;
      const int64_t jc = j >= 0 ? j : 0;          // (unconditional loads: no short-circuit branches)
      const unsigned f = (unsigned)a.term[jc] | (unsigned)a.trunc[jc];
      d[k] = (j < 0) | (f != 0);
    }
    int k = 0;
    while (k < 4 && !d[k]) ++k;
    lo -= k;
    if (k < 4) break;
  }
  float gae = 0.f;
  auto step = [&](int64_t i, float nv, float rw, float vi, bool term, bool done) {
    // td = reward + gamma * next_value * (~terminated) - V[i]
    const float t0 = __fmul_rn(a.gamma, nv);
    const float t1 = __fmul_rn(t0, term ? 0.f : 1.f);
    const float t2 = __fadd_rn(rw, t1);
    const float td = __fsub_rn(t2, vi);
    const float c = done ? 0.f : a.gl;
    gae = __fadd_rn(td, __fmul_rn(c, gae));
    a.gae[i] = gae;
    a.lam_return[i] = __fadd_rn(gae, vi);
  };
  int64_t i = i1;
  for (; i - 3 >= lo; i -= 4) {
    float nv[4], rw[4], vi[4];
    bool tm[4], dn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = i - k;
      nv[k] = (j == a.N - 1) ? a.next_value_last[0] : a.values[j + 1];
      rw[k] = a.reward[j];
      vi[k] = a.values[j];
      const unsigned t8 = a.term[j], u8 = a.trunc[j];
      tm[k] = t8 != 0;
      dn[k] = (t8 | u8) != 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) step(i - k, nv[k], rw[k], vi[k], tm[k], dn[k]);
  }
  for (; i >= lo; --i) {
    const unsigned t8 = a.term[i], u8 = a.trunc[i];
    step(i, (i == a.N - 1) ? a.next_value_last[0] : a.values[i + 1], a.reward[i], a.values[i], t8 != 0,
         (t8 | u8) != 0);
  }
}

// ---- continuous SAC -------------------------------------------------------------------------
// GaussianActorNetwork.forward tail + sample_action (actor_networks.py:537-591):
//   log_std = -5 + 3.5 * (tanh(raw) + 1); u = mean + exp(log_std) * noise; n = tanh(u)
//   action = ((high - low) * (n + 1)) / 2 + low
//   log_prob = sum_j [ Normal(mean, std).log_prob(u) - log(bound * (1 - n^2) + 1e-6) ]
struct GaussArgs {
  const float* head; int ldh;      // [B, 2A]: mean | raw log_std
  const float* noise; int ldn;     // [B, A] standard normal draws (host-supplied in parity mode)
  const float* low; const float* high;  // [A]
  int B, A;
  float* action; int lda;          // [B, A] (may point into a [B, S+A] critic input)
  float* log_prob;                 // [B]
};
__device__ __forceinline__ void gauss_elem(float mean, float raw, float eps, float low, float high,
                                           float& t, float& log_std, float& stdv, float& u,
                                           float& n) {
  t = tanhf(raw);
  log_std = -5.0f + 3.5f * (t + 1.0f);
  stdv = expf(log_std);
  u = mean + stdv * eps;
  n = tanhf(u);
}
// One thread per (row, action component): a workgroup covers 256 / A rows, the per-row sum of the
// A log-prob terms is taken in component order by the row's first thread (the same order as a
// serial loop, so the result does not depend on the launch shape).  Needs A <= 256.
__global__ __launch_bounds__(256) void gauss_sample_kernel(GaussArgs a) {
  __shared__ float terms[256];
  const int rows_per_wg = 256 / a.A;
  const int r = threadIdx.x / a.A, j = threadIdx.x - r * a.A;
  const int b = blockIdx.x * rows_per_wg + r;
  const bool live = r < rows_per_wg && b < a.B;
  if (live) {
    const float* hd = a.head + (int64_t)b * a.ldh;
    float t, ls, sd, u, n;
    const float eps = a.noise[(int64_t)b * a.ldn + j];
    const float mean = hd[j], lo = a.low[j], hi = a.high[j];
    gauss_elem(mean, hd[a.A + j], eps, lo, hi, t, ls, sd, u, n);
    const float act = (((hi - lo) * (n + 1.0f)) / 2.0f) + lo;
    a.action[(int64_t)b * a.lda + j] = act;
    // Normal.log_prob: -((u - mean)^2) / (2 var) - log(std) - log(sqrt(2 pi))
    const float var = sd * sd;
    const float diff = u - mean;
    float l = -(diff * diff) / (2.0f * var) - logf(sd) - 0.9189385332046727f;
    const float bound = (hi - lo) / 2.0f;
    l -= logf(bound * (1.0f - n * n) + 1e-6f);
    terms[threadIdx.x] = l;
  }
  __syncthreads();
  if (live && j == 0) {
    float lp = 0.f;
    for (int k = 0; k < a.A; ++k) lp += terms[threadIdx.x + k];
    a.log_prob[b] = lp;
  }
}

// Gradient of mean_b(alpha * log_prob_b - q_b) w.r.t. the actor head, given dq_b/da (the critic's
// input gradient, already scaled by dL/dq = -1/B * min-weights).
struct GaussGradArgs {
  const float* head; int ldh;
  const float* noise; int ldn;
  const float* low; const float* high;
  const float* dl_da; const float* dl_da2; int ldda;  // [B, A] dL/d action through critic 1 (+ critic 2)
  const float* alpha;              // device scalar (entropy coefficient)
  int B, A;
  float* d_head; int lddh;         // [B, 2A]
};
__global__ __launch_bounds__(256) void gauss_grad_kernel(GaussGradArgs a) {
  // one thread per (row, action component)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.B * a.A) return;
  const int b = (int)(e / a.A), j = (int)(e - (int64_t)b * a.A);
  const float* hd = a.head + (int64_t)b * a.ldh;
  const float coef = a.alpha[0] / (float)a.B;   // dL/dlog_prob_b
  float t, ls, sd, u, n;
  const float eps = a.noise[(int64_t)b * a.ldn + j];
  gauss_elem(hd[j], hd[a.A + j], eps, a.low[j], a.high[j], t, ls, sd, u, n);
  const float bound = (a.high[j] - a.low[j]) / 2.0f;
  const float one_m_n2 = 1.0f - n * n;
  // d log_prob / du (the Normal term is -noise^2/2: constant under the reparameterisation)
  const float dlp_du = (2.0f * bound * n * one_m_n2) / (bound * one_m_n2 + 1e-6f);
  const float da_du = bound * one_m_n2;
  float dla = a.dl_da[(int64_t)b * a.ldda + j];
  if (a.dl_da2) dla += a.dl_da2[(int64_t)b * a.ldda + j];
  const float dl_du = coef * dlp_du + dla * da_du;
  const float dl_dls = dl_du * eps * sd - coef;     // -log(std) term: -1
  a.d_head[(int64_t)b * a.lddh + j] = dl_du;
  a.d_head[(int64_t)b * a.lddh + a.A + j] = dl_dls * 3.5f * (1.0f - t * t);
}

// ---- deterministic policies (DDPG / TD3) ------------------------------------------------------
// VanillaContinuousActorNetwork.sample_action (actor_networks.py:448-485): the network's last
// activation is tanh, then action_scaling (:29-51):  a = ((high - low) (tanh(z) + 1)) / 2 + low.
// With `noise` (TD3's target policy smoothing, td3.py:151-175): the N(0, sigma^2) draws are clamped
// to [-clip, clip], rescaled  noise (high - low) / 2, added, and the sum clamped to [low, high].
struct TanhActArgs {
  const float* head; int ldh;       // [B, A] pre-tanh outputs of the actor
  const float* noise; int ldn;      // [B, A] or null
  const float* low; const float* high;
  float clip;
  int B, A;
  float* action; int lda;           // [B, A] (may point into a [B, S+A] critic input)
};
__global__ __launch_bounds__(256) void tanh_action_kernel(TanhActArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.B * a.A) return;
  const int b = (int)(e / a.A), j = (int)(e - (int64_t)b * a.A);
  const float lo = a.low[j], hi = a.high[j];
  const float t = tanhf(a.head[(int64_t)b * a.ldh + j]);
  float act = (((hi - lo) * (t + 1.0f)) / 2.0f) + lo;
  if (a.noise) {
    float n = a.noise[(int64_t)b * a.ldn + j];
    n = fminf(fmaxf(n, -a.clip), a.clip);
    n = (n * (hi - lo)) / 2.0f;
    act = fminf(fmaxf(act + n, lo), hi);
  }
  a.action[(int64_t)b * a.lda + j] = act;
}

// d_head = dL/da . da/dz for the un-noised action:  (g / 2) (high - low) (1 - tanh(z)^2)
struct TanhGradArgs {
  const float* head; int ldh;
  const float* low; const float* high;
  const float* dl_da; int ldda;
  int B, A;
  float* d_head; int lddh;
};
__global__ __launch_bounds__(256) void tanh_action_grad_kernel(TanhGradArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.B * a.A) return;
  const int b = (int)(e / a.A), j = (int)(e - (int64_t)b * a.A);
  const float t = tanhf(a.head[(int64_t)b * a.ldh + j]);
  const float g = a.dl_da[(int64_t)b * a.ldda + j];
  const float gt = (g / 2.0f) * (a.high[j] - a.low[j]);
  a.d_head[(int64_t)b * a.lddh + j] = gt * (1.0f - t * t);
}

// DDPG's actor objective (ddpg.py:106-121): loss = -mean(q1), d loss / d q1 = -1/B
__global__ __launch_bounds__(256) void neg_mean_head_kernel(const float* __restrict__ q, int ldq,
                                                            int B, float* __restrict__ dq,
                                                            float* __restrict__ loss_out) {
  __shared__ float red[256];
  float part = 0.f;
  const float g = -1.0f / (float)B;
  for (int i = threadIdx.x; i < B; i += 256) {
    part += q[(int64_t)i * ldq];
    dq[i] = g;
  }
  const float sum = block_sum_256(part, red);
  if (threadIdx.x == 0) loss_out[0] = -(sum / (float)B);
}

// ---- discrete SAC (soft_actor_critic.py:180-287) ----------------------------------------------
// x[b * A + i] = state[b] || rep(available_actions[b, i])  — the (B, A, S + AD) critic input of
// TwinCritic.get_q_values on an action set (q_value_networks.py:152-174 expands the state)
__global__ __launch_bounds__(256) void expand_state_actions_kernel(
    const float* __restrict__ state, int lds_, const float* __restrict__ rep, int64_t rep_bstride,
    float* __restrict__ x, int B, int A, int S, int AD) {
  const int W = S + AD;
  const int64_t total = (int64_t)B * A * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / W;
    const int j = (int)(e - row * W);
    const int64_t b = row / A;
    const int i = (int)(row - b * A);
    x[e] = (j < S) ? state[b * lds_ + j] : rep[b * rep_bstride + (int64_t)i * AD + (j - S)];
  }
}

// Row-local pieces of both losses.  P = softmax(logits_b) (the policy ignores availability masks,
// actor_networks.py:136-153), logP = log(P + 1e-8), q = min(q1, q2) with unavailable actions at 0.
//   mode 0, actor (:254-287): loss = mean over (B, A) of P (alpha logP - q); d_logits through the
//           softmax; h_b = sum_j P_j logP_j (= -entropy_b, what the autotune step needs, :134-151)
//   mode 1, target (:180-252): v_b = sum_j (q_j - alpha log(P_j + 1e-8)) P_j;
//           y_b = v_b gamma (1 - term_b) + r_b
struct DsacArgs {
  const float* logits; int ldl;
  const float* q1; const float* q2;      // [B * A], row b * A + j
  const uint8_t* mask;                   // [B, A] 1 = unavailable, or null
  const float* alpha;
  int B, A, mode;
  float* d_logits; int ldd; float* loss_out; float* h_out;
  const float* reward; const uint8_t* term; float gamma; float* y;
  float* partials; unsigned* ticket;     // element-parallel form, mode 0
};
// Element-parallel form (A <= 256): one thread per (row, action), 256 / A rows per workgroup; the
// per-row reductions are serial loops over the row's LDS slots in action order — the arithmetic of
// the row-per-thread kernel below, which for 1024 x 16 ran 16 384 softmax elements on ONE
// workgroup (67 us a launch, two launches a step: 29 % of discrete SAC's step).
__global__ __launch_bounds__(256) void dsac_elem_kernel(DsacArgs a) {
  __shared__ float va[256], vb[256], vc[256], red[256];
  __shared__ unsigned last;
  const float alpha = a.alpha[0];
  const float inv_n = 1.0f / ((float)a.B * (float)a.A);
  const int rpw = 256 / a.A;
  const int r = threadIdx.x / a.A, j = threadIdx.x - r * a.A;
  const int b = blockIdx.x * rpw + r;
  const bool live = r < rpw && b < a.B;
  const int base = r * a.A;
  const float z = live ? a.logits[(int64_t)b * a.ldl + j] : 0.f;
  float q = 0.f;
  if (live && !(a.mask && a.mask[(int64_t)b * a.A + j]))
    q = fminf(a.q1[(int64_t)b * a.A + j], a.q2[(int64_t)b * a.A + j]);
  va[threadIdx.x] = z;
  __syncthreads();
  float m = 0.f, s = 0.f;
  if (live) {
    m = va[base];
    for (int k = 1; k < a.A; ++k) m = fmaxf(m, va[base + k]);
  }
  const float e = expf(z - m);
  vb[threadIdx.x] = e;
  __syncthreads();
  if (live)
    for (int k = 0; k < a.A; ++k) s += vb[base + k];
  const float p = live ? e / s : 0.f;
  if (a.mode == 1) {
    va[threadIdx.x] = (q - alpha * logf(p + 1e-8f)) * p;
    __syncthreads();
    if (live && j == 0) {
      float v = 0.f;
      for (int k = 0; k < a.A; ++k) v += va[base + k];
      const float lv = 1.0f - (a.term[b] ? 1.0f : 0.0f);
      a.y[b] = __fadd_rn(__fmul_rn(__fmul_rn(v, a.gamma), lv), a.reward[b]);
    }
    return;
  }
  const float lp = logf(p + 1e-8f);
  const float f = alpha * lp - q;
  const float g = (f + p * (alpha / (p + 1e-8f))) * inv_n;   // dL/dP_j
  va[threadIdx.x] = p * f;
  vc[threadIdx.x] = g * p;
  __syncthreads();            // (every thread is past its reads of vb: the barrier above the p line)
  vb[threadIdx.x] = p * lp;
  float dot = 0.f;
  if (live)
    for (int k = 0; k < a.A; ++k) dot += vc[base + k];
  if (live) a.d_logits[(int64_t)b * a.ldd + j] = p * (g - dot);
  __syncthreads();
  float part = 0.f;
  if (live && j == 0) {
    float h = 0.f;
    for (int k = 0; k < a.A; ++k) {
      part += va[base + k];
      h += vb[base + k];
    }
    a.h_out[b] = h;
  }
  const float bl = block_sum_256(part, red);
  if (threadIdx.x == 0) {
    a.partials[blockIdx.x] = bl;
    __threadfence();
    last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float pl = 0.f;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += 256)
    pl += __builtin_nontemporal_load(a.partials + k);
  const float sum = block_sum_256(pl, red);
  if (threadIdx.x == 0) {
    a.loss_out[0] = sum * inv_n;
    *a.ticket = 0u;
  }
}
__global__ __launch_bounds__(256) void dsac_kernel(DsacArgs a) {
  __shared__ float red[256];
  const float alpha = a.alpha[0];
  const float inv_n = 1.0f / ((float)a.B * (float)a.A);
  float part = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float* z = a.logits + (int64_t)b * a.ldl;
    const float* q1 = a.q1 + (int64_t)b * a.A;
    const float* q2 = a.q2 + (int64_t)b * a.A;
    const uint8_t* mk = a.mask ? a.mask + (int64_t)b * a.A : nullptr;
    float m = z[0];
    for (int j = 1; j < a.A; ++j) m = fmaxf(m, z[j]);
    float s = 0.f;
    for (int j = 0; j < a.A; ++j) s += expf(z[j] - m);
    if (a.mode == 1) {
      float v = 0.f;
      for (int j = 0; j < a.A; ++j) {
        const float p = expf(z[j] - m) / s;
        const float q = (mk && mk[j]) ? 0.f : fminf(q1[j], q2[j]);
        v += (q - alpha * logf(p + 1e-8f)) * p;
      }
      const float live = 1.0f - (a.term[b] ? 1.0f : 0.0f);
      a.y[b] = __fadd_rn(__fmul_rn(__fmul_rn(v, a.gamma), live), a.reward[b]);
      continue;
    }
    float dot = 0.f, h = 0.f;
    for (int j = 0; j < a.A; ++j) {
      const float p = expf(z[j] - m) / s;
      const float lp = logf(p + 1e-8f);
      const float q = (mk && mk[j]) ? 0.f : fminf(q1[j], q2[j]);
      const float f = alpha * lp - q;
      part += p * f;
      h += p * lp;
      const float g = (f + p * (alpha / (p + 1e-8f))) * inv_n;   // dL/dP_j
      dot += g * p;
    }
    a.h_out[b] = h;
    float* dz = a.d_logits + (int64_t)b * a.ldd;
    for (int j = 0; j < a.A; ++j) {
      const float p = expf(z[j] - m) / s;
      const float lp = logf(p + 1e-8f);
      const float q = (mk && mk[j]) ? 0.f : fminf(q1[j], q2[j]);
      const float g = ((alpha * lp - q) + p * (alpha / (p + 1e-8f))) * inv_n;
      dz[j] = p * (g - dot);
    }
  }
  if (a.mode == 0) {
    const float sum = block_sum_256(part, red);
    if (threadIdx.x == 0) a.loss_out[0] = sum * inv_n;
  }
}

// ---- implicit Q-learning (implicit_q_learning.py:159-285) --------------------------------------
// Value head (:186-196, :271-285): expectile regression of V(s) towards a target critic's Q(s, a),
//   loss = mean(w d^2), d = tq_value - v, w = expectile if d > 0 else 1 - expectile; dv = -2 w d / B.
// Advantage weights of the policy extraction (:203-215), detached in the reference:
//   adv = min(exp((tq_actor - v) temperature), clamp).
struct IqlValueArgs {
  const float* tq_value; const float* tq_actor; const float* v; int ldv;
  float expectile, temperature, adv_clamp;
  int B;
  float* dv; float* adv; float* loss_out;
};
__global__ __launch_bounds__(256) void iql_value_kernel(IqlValueArgs a) {
  __shared__ float red[256];
  float part = 0.f;
  const float invB = 1.0f / (float)a.B;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float v = a.v[(int64_t)b * a.ldv];
    const float d = a.tq_value[b] - v;
    const float w = d > 0.f ? a.expectile : (1.0f - a.expectile);
    part += w * (d * d);
    a.dv[b] = -2.0f * w * d * invB;
    a.adv[b] = fminf(expf((a.tq_actor[b] - v) * a.temperature), a.adv_clamp);
  }
  const float sum = block_sum_256(part, red);
  if (threadIdx.x == 0) a.loss_out[0] = sum * invB;
}

// Advantage-weighted regression heads (:197-246).
//   mode 0, deterministic actor: loss = mean_b(adv_b mean_j (pred - action)^2);
//           d_pred = adv_b 2 (pred - action) / (A B)
//   mode 1, softmax actor: idx = argmax_j action_b (one-hot), loss = -mean_b(adv_b log P[idx]);
//           d_logits_j = -(adv_b / B) ([j == idx] - P_j)
struct AwrArgs {
  const float* x; int ldx;           // mode 0: predicted actions [B, A]; mode 1: logits [B, A]
  const float* action; int lda;      // [B, A]
  const float* adv;
  int B, A, mode;
  float* dx; int lddx; float* loss_out;
};
__global__ __launch_bounds__(256) void awr_kernel(AwrArgs a) {
  __shared__ float red[256];
  float part = 0.f;
  const float invB = 1.0f / (float)a.B;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float* x = a.x + (int64_t)b * a.ldx;
    const float* act = a.action + (int64_t)b * a.lda;
    float* dx = a.dx + (int64_t)b * a.lddx;
    const float w = a.adv[b];
    if (a.mode == 0) {
      float sq = 0.f;
      const float g = w * 2.0f * invB / (float)a.A;
      for (int j = 0; j < a.A; ++j) {
        const float d = x[j] - act[j];
        sq += d * d;
        dx[j] = g * d;
      }
      part += w * (sq / (float)a.A);
    } else {
      int idx = 0;
      float best = act[0];
      for (int j = 1; j < a.A; ++j)
        if (act[j] > best) { best = act[j]; idx = j; }
      float m = x[0];
      for (int j = 1; j < a.A; ++j) m = fmaxf(m, x[j]);
      float s = 0.f;
      for (int j = 0; j < a.A; ++j) s += expf(x[j] - m);
      const float g = -w * invB;
      for (int j = 0; j < a.A; ++j) {
        const float p = expf(x[j] - m) / s;
        dx[j] = g * ((j == idx ? 1.0f : 0.0f) - p);
        if (j == idx) part += -w * logf(p);
      }
    }
  }
  const float sum = block_sum_256(part, red);
  if (threadIdx.x == 0) a.loss_out[0] = sum * invB;
}

// ---- IQL policy extraction with a GaussianActorNetwork (implicit_q_learning.py:231-243,
// actor_networks.py:593-629 get_log_probability): advantage-weighted regression
//   loss = -mean_b(adv_b log pi(a_b | s_b)),
//   log pi = sum_j [ Normal(mean_j, std_j).log_prob(atanh(n_j)) - log(bound_j (1 - n_j^2) + 1e-6) ],
//   n = clip(((a - low) / (high - low)) 2 - 1, -1 + 1e-6, 1 - 1e-6)   (action_unscaling, :54-65)
// from the head [B, 2A] = mean | raw log_std (log_std = -5 + 3.5 (tanh(raw) + 1), :537-549).
// The dataset action is a constant: only mean and log_std carry gradient,
//   d logp / d mean = (u - mean) / var,   d logp / d log_std = (u - mean)^2 / var - 1.
// One thread per (row, component); the per-row sum is taken in component order by the row's first
// thread, the batch sum block-ordered through a ticket (same shape as gauss_sample_kernel).
struct GaussAwrArgs {
  const float* head; int ldh;
  const float* action; int lda;
  const float* low; const float* high;
  const float* adv;
  int B, A;
  float* d_head; int lddh;
  float* log_prob;              // [B] scratch / probe output
  float* loss_out;
};
__global__ __launch_bounds__(256) void gauss_awr_kernel(GaussAwrArgs a) {
  __shared__ float terms[256];
  const int rows_per_wg = 256 / a.A;
  const int r = threadIdx.x / a.A, j = threadIdx.x - r * a.A;
  const int b = blockIdx.x * rows_per_wg + r;
  const bool live = r < rows_per_wg && b < a.B;
  if (live) {
    const float* hd = a.head + (int64_t)b * a.ldh;
    const float mean = hd[j], raw = hd[a.A + j], lo = a.low[j], hi = a.high[j];
    const float t = tanhf(raw);
    const float log_std = -5.0f + 3.5f * (t + 1.0f);
    const float sd = expf(log_std);
    float n = (((a.action[(int64_t)b * a.lda + j] - lo) / (hi - lo)) * 2.0f) - 1.0f;
    n = fminf(fmaxf(n, -1.0f + 1e-6f), 1.0f - 1e-6f);
    const float u = atanhf(n);
    const float var = sd * sd;
    const float diff = u - mean;
    const float bound = (hi - lo) / 2.0f;
    float l = -(diff * diff) / (2.0f * var) - logf(sd) - 0.9189385332046727f;
    l -= logf(bound * (1.0f - n * n) + 1e-6f);
    terms[threadIdx.x] = l;
    const float g = -a.adv[b] / (float)a.B;       // dL / d logp_b
    a.d_head[(int64_t)b * a.lddh + j] = g * (diff / var);
    a.d_head[(int64_t)b * a.lddh + a.A + j] = g * ((diff * diff) / var - 1.0f) * 3.5f * (1.0f - t * t);
  }
  __syncthreads();
  if (live && j == 0) {
    float lp = 0.f;
    for (int k = 0; k < a.A; ++k) lp += terms[threadIdx.x + k];
    a.log_prob[b] = lp;
  }
}
// loss = -mean(adv * log_prob), fixed order (one workgroup)
__global__ __launch_bounds__(256) void gauss_awr_loss_kernel(const float* __restrict__ adv,
                                                             const float* __restrict__ logp, int B,
                                                             float* __restrict__ loss_out) {
  __shared__ float red[256];
  float part = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) part += adv[b] * logp[b];
  const float sum = block_sum_256(part, red);
  if (threadIdx.x == 0) loss_out[0] = -(sum / (float)B);
}

// ---- conservative Q-learning (deep_td_learning.py:292-331, loss_fn_utils.py:17-72) -------------
// One head for the (B + B A)-row pass of DeepQLearning(is_conservative=True):
//   rows [0, B):  Q(s_b, a_b);  Bellman part  dq = 2 (q - y) / B,  reported loss mean |q - y|
//   rows B + b A + i:  Q(s_b, available action i) (padded, unmasked — as the reference);
//     cql = mean_b logsumexp_i q_all[b, i] - mean over (b, j) of q_all[b, long(action[b, j])]
//     (the reference gathers with batch.action.long(), whatever batch.action holds: with a one-hot
//     action that is column 0, A - 1 times, and column 1, once);
//     dq = alpha (softmax_i / B - count[b, i] / (B AD))
struct CqlArgs {
  const float* q; const float* y;
  const float* action; int lda;     // [B, AD] batch.action after preprocess_batch
  int B, A, AD;
  float alpha;
  float* dq; float* loss_out;       // loss_out[0] = mean |q - y|, loss_out[1] = total loss
};
__global__ __launch_bounds__(256) void cql_head_kernel(CqlArgs a) {
  __shared__ float red[256];
  const float invB = 1.0f / (float)a.B;
  float abs_part = 0.f, loss_part = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float d = a.q[b] - a.y[b];
    a.dq[b] = 2.0f * d * invB;
    abs_part += fabsf(d);
    loss_part += d * d * invB;
    const float* qa = a.q + a.B + (int64_t)b * a.A;
    float* dqa = a.dq + a.B + (int64_t)b * a.A;
    float m = qa[0];
    for (int i = 1; i < a.A; ++i) m = fmaxf(m, qa[i]);
    float s = 0.f;
    for (int i = 0; i < a.A; ++i) s += expf(qa[i] - m);
    loss_part += a.alpha * (m + logf(s)) * invB;
    for (int i = 0; i < a.A; ++i) dqa[i] = a.alpha * (expf(qa[i] - m) / s) * invB;
    const float w = a.alpha * invB / (float)a.AD;
    for (int j = 0; j < a.AD; ++j) {
      const int idx = (int)(long long)a.action[(int64_t)b * a.lda + j];
      if (idx >= 0 && idx < a.A) {
        dqa[idx] -= w;
        loss_part -= w * qa[idx];
      }
    }
  }
  const float s_abs = block_sum_256(abs_part, red);
  const float s_loss = block_sum_256(loss_part, red);
  if (threadIdx.x == 0) {
    a.loss_out[0] = s_abs * invB;
    a.loss_out[1] = s_loss;
  }
}

// Twin-critic plumbing for SAC (soft_actor_critic_continuous.py:155-231).
//   mode 0 (actor loss):  loss = mean(alpha * logp - min(q1, q2)); dq1/dq2 = -w/B with torch.minimum's
//                         even split on ties
//   mode 1 (critic target): y = (min(q1', q2') - alpha * logp') * gamma * (1 - term) + reward
struct TwinArgs {
  const float* q1; const float* q2; const float* logp; const float* alpha;
  const float* reward; const uint8_t* term;
  float gamma;
  int B, mode;
  float* out1; float* out2;   // mode 0: dq1, dq2; mode 1: out1 = y
  float* loss_out;            // mode 0
};
__global__ __launch_bounds__(256) void twin_kernel(TwinArgs a) {
  __shared__ float red[256];
  const float al = a.alpha[0];
  float part = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float x1 = a.q1[b], x2 = a.q2[b];
    const float mn = fminf(x1, x2);
    if (a.mode == 0) {
      part += al * a.logp[b] - mn;
      const float w1 = x1 < x2 ? 1.f : (x1 == x2 ? 0.5f : 0.f);
      a.out1[b] = -w1 / (float)a.B;
      a.out2[b] = -(1.f - w1) / (float)a.B;
    } else {
      const float v = mn - al * a.logp[b];
      const float live = 1.0f - (a.term[b] ? 1.0f : 0.0f);
      a.out1[b] = __fadd_rn(__fmul_rn(__fmul_rn(v, a.gamma), live), a.reward[b]);
    }
  }
  if (a.mode == 0) {
    const float s = block_sum_256(part, red);
    if (threadIdx.x == 0) a.loss_out[0] = s / (float)a.B;
  }
}

// Entropy-coefficient autotune (soft_actor_critic_continuous.py:134-151): scalar AdamW(amsgrad) on
// log_alpha with gradient mean(-exp(log_alpha) * (logp + target_entropy)); alpha = exp(log_alpha).
struct AlphaArgs {
  float* log_alpha; float* m; float* v; float* vmax; float* alpha;
  const float* logp; int B; float target_entropy;
  AdamScalars c;
  float* loss_out;
};
__global__ __launch_bounds__(256) void alpha_kernel(AlphaArgs a) {
  __shared__ float red[256];
  const float ea = expf(a.log_alpha[0]);
  float part = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) part += -ea * (a.logp[b] + a.target_entropy);
  const float s = block_sum_256(part, red);
  if (threadIdx.x == 0) {
    const float g = s / (float)a.B;   // d loss / d log_alpha == the loss itself
    if (a.loss_out) a.loss_out[0] = g;
    AdamState st;
    st.p = a.log_alpha; st.m = a.m; st.v = a.v; st.vmax = a.vmax;
    const float p = adam_update(a.c, st, 0, g);
    a.alpha[0] = expf(p);
  }
}

// out[b, 0:S] = state[b], out[b, S:S+A] = action[b]   (q_value_networks.py:166-168 torch.cat)
__global__ __launch_bounds__(256) void concat_kernel(const float* __restrict__ s, int lds_,
                                                     const float* __restrict__ act, int lda,
                                                     float* __restrict__ out, int B, int S, int A) {
  const int W = S + A;
  const int64_t total = (int64_t)B * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / W;
    const int j = (int)(e - b * W);
    out[e] = (j < S) ? s[b * lds_ + j] : act[b * lda + (j - S)];
  }
}

}  // namespace


namespace {

// Weighted loss of the neural-linear bandit (neural_linear_bandit.py:176-199; LossType,
// neural_networks/common/utils.py:60-72): pred = act(z), l_b = criterion(pred_b, y_b) unreduced,
//   loss = sum_b w_b l_b / sum_b w_b;  d_z_b = dl_b/dpred_b * (w_b / sum w) * act'(z_b)
// with torch's own backward formulas (one rounding per torch op):
//   mse_loss   l = (p - y)^2                 dl/dp = 2 (p - y) g
//   l1_loss    l = |p - y|                   dl/dp = sign(p - y) g
//   binary_cross_entropy
//              l = (y - 1) max(log(1 - p), -100) - y max(log p, -100)
//                                            dl/dp = g (p - y) / max((1 - p) p, 1e-12)
//   sigmoid    p = 1 / (1 + exp(-z))         dz = dp (1 - p) p
// wloss_row is shared with the fused row step (mlp_rowstep.hpp, RS_HEAD_WMSE1).
struct WmseArgs {
  const float* pred; int ldp; const float* y; const float* w;  // w may be null (= ones)
  int B;
  float* d_pred; float* loss_out; float* wsum_out;
  int loss_kind, out_act;      // PA_LOSS_*, PA_OUT_*
  float* pred_out;             // [B] post-activation predictions, or null
};
__global__ __launch_bounds__(256) void wmse_kernel(WmseArgs a) {
  __shared__ float red[256];
  float pw = 0.f, pl = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float w = a.w ? a.w[b] : 1.0f;
    const float p = wloss_act(a.pred[(int64_t)b * a.ldp], a.out_act);
    pw += w;
    pl += wloss_value(p, a.y[b], a.loss_kind) * w;
  }
  const float wsum = block_sum_256(pw, red);
  const float lsum = block_sum_256(pl, red);
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const float w = a.w ? a.w[b] : 1.0f;
    const float p = wloss_act(a.pred[(int64_t)b * a.ldp], a.out_act);
    if (a.pred_out) a.pred_out[b] = p;
    a.d_pred[b] = (wsum != 0.f) ? wloss_grad(p, a.y[b], w, wsum, a.loss_kind, a.out_act) : 0.f;
  }
  if (threadIdx.x == 0) {
    a.loss_out[0] = (wsum != 0.f) ? lsum / wsum : 0.f;
    if (a.wsum_out) a.wsum_out[0] = wsum;
  }
}

// LinearRegression.learn_batch operands (linear_regression.py:192-219): X = [1 | nn_out] and
// R = [X * w | y * w], so that X^T R = [delta_A (before symmetrisation) | delta_b].
// (ldX >= D, ldR >= D + 1: row pitches; pad columns are written as zeros)
__global__ __launch_bounds__(256) void linreg_operands_kernel(const float* __restrict__ f, int ldf,
                                                              const float* __restrict__ y,
                                                              const float* __restrict__ w, int B,
                                                              int d, float* __restrict__ X, int ldX,
                                                              float* __restrict__ R, int ldR) {
  const int D = d + 1;
  const int W = ldR > ldX ? ldR : ldX;
  const int64_t total = (int64_t)B * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / W;
    const int j = (int)(e - b * W);
    const float wb = w ? w[b] : 1.0f;
    const float x = (j == 0) ? 1.0f : (j < D ? f[b * ldf + (j - 1)] : 0.f);
    if (j < ldX) X[b * ldX + j] = x;
    if (j < ldR) R[b * ldR + j] = j < D ? x * wb : (j == D ? y[b] * wb : 0.f);
  }
}

// A += (dA + dA^T) / 2; b += db; sum_weight += dsw     (linear_regression.py:204-216)
// A_snap / b_snap (nullable): the updated A and b written a second time — the operands of the
// asynchronous solve (two device-to-device copy launches otherwise)
__global__ __launch_bounds__(256) void linreg_apply_kernel(const float* __restrict__ delta, int D,
                                                           float* __restrict__ A,
                                                           float* __restrict__ bvec,
                                                           float* __restrict__ sw,
                                                           float* __restrict__ A_snap,
                                                           float* __restrict__ b_snap,
                                                           const float* __restrict__ dsw) {
  const int64_t total = (int64_t)D * D;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / D), j = (int)(e - (int64_t)i * D);
    const float s = (delta[(int64_t)i * (D + 1) + j] + delta[(int64_t)j * (D + 1) + i]) / 2.0f;
    const float an = A[e] + s;
    A[e] = an;
    if (A_snap) A_snap[e] = an;
    if (j == 0) {
      const float bn = bvec[i] + delta[(int64_t)i * (D + 1) + D];
      bvec[i] = bn;
      if (b_snap) b_snap[i] = bn;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sw[0] += dsw[0];
}

// inv(A + lambda I) by Gauss-Jordan elimination with partial pivoting in fp64 (one workgroup; the
// augmented [D x 2D] system lives in a caller-provided fp64 workspace), then coefs = inv_A b.
// torch.linalg.inv (linear_regression.py:154-170, :252-259) is LAPACK's fp32 LU; working in fp64
// keeps the result within fp32 rounding of the exact inverse, i.e. inside LAPACK's own error.
struct SolveArgs {
  const float* A; const float* bvec; float lambda; int D;
  double* work;          // [D][2D]
  float* invA; float* coefs;
  int* singular;         // set to 1 if a pivot vanished
  const int* only_if;    // non-null: run only when this device word is non-zero (fallback launch)
};
__global__ __launch_bounds__(1024) void linreg_solve_kernel(SolveArgs a) {
  __shared__ double pv[1024];
  __shared__ int pi_[1024];
  __shared__ int prow;
  const int D = a.D, W = 2 * D, tid = threadIdx.x;
  for (int e = tid; e < D * W; e += 1024) {
    const int i = e / W, j = e - i * W;
    double v;
    if (j < D) v = (double)a.A[i * D + j] + (i == j ? (double)a.lambda : 0.0);
    else v = (j - D == i) ? 1.0 : 0.0;
    a.work[e] = v;
  }
  __syncthreads();
  for (int k = 0; k < D; ++k) {
    // pivot search in column k over rows k..D-1
    double best = -1.0;
    int bi = k;
    for (int i = k + tid; i < D; i += 1024) {
      const double v = fabs(a.work[i * W + k]);
      if (v > best) { best = v; bi = i; }
    }
    pv[tid] = best; pi_[tid] = bi;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) {
      if (tid < s) {
        if (pv[tid + s] > pv[tid] || (pv[tid + s] == pv[tid] && pi_[tid + s] < pi_[tid])) {
          pv[tid] = pv[tid + s]; pi_[tid] = pi_[tid + s];
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      prow = pi_[0];
      if (!(pv[0] > 0.0)) a.singular[0] = 1;
    }
    __syncthreads();
    const int p = prow;
    if (p != k) {
      for (int j = tid; j < W; j += 1024) {
        const double t = a.work[k * W + j];
        a.work[k * W + j] = a.work[p * W + j];
        a.work[p * W + j] = t;
      }
    }
    __syncthreads();
    const double piv = a.work[k * W + k];
    __syncthreads();
    for (int j = tid; j < W; j += 1024) a.work[k * W + j] /= piv;
    __syncthreads();
    // eliminate column k from every other row
    for (int e = tid; e < D * W; e += 1024) {
      const int i = e / W, j = e - i * W;
      if (i == k || j == k) continue;
      a.work[e] -= a.work[i * W + k] * a.work[k * W + j];
    }
    __syncthreads();
    for (int i = tid; i < D; i += 1024)
      if (i != k) a.work[i * W + k] = 0.0;
    __syncthreads();
  }
  for (int e = tid; e < D * D; e += 1024) {
    const int i = e / D, j = e - i * D;
    a.invA[e] = (float)a.work[i * W + D + j];
  }
  __syncthreads();
  for (int i = tid; i < D; i += 1024) {
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += a.work[i * W + D + j] * (double)a.bvec[j];
    a.coefs[i] = (float)s;
  }
}

// The same elimination (same pivots; the pivot row is scaled by one fp64 reciprocal instead of 2D
// divisions) with the augmented matrix in LDS, the element -> (row, column) map of every thread
// computed once, and four barriers per pivot step instead of seventeen on a global-memory work
// area: 419 us -> ~100 us for D = 65.  Used whenever [D][2D] doubles fit.
__global__ __launch_bounds__(1024) void linreg_solve_lds_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  if (a.only_if && a.only_if[0] == 0) return;      // the pivot-free kernel succeeded
  const int D = a.D, W = 2 * D, tid = threadIdx.x, lane = tid & 63;
  double* work = lds_work;          // [D][W]
  double* prowv = work + D * W;     // [W] normalised pivot row
  double* colk = prowv + W;         // [D] column k before the elimination
  __shared__ int prow;
  // this thread's elements e = tid + 1024 t of the [D][W] matrix: (row, column) advance by a fixed
  // stride, so the k loop needs no integer division
  const int i_first = tid / W, j_first = tid - i_first * W;
  const int di = 1024 / W, dj = 1024 - di * W;
  for (int e = tid; e < D * W; e += 1024) {
    const int i = e / W, j = e - i * W;
    double v;
    if (j < D) v = (double)a.A[i * D + j] + (i == j ? (double)a.lambda : 0.0);
    else v = (j - D == i) ? 1.0 : 0.0;
    work[e] = v;
  }
  __syncthreads();
  for (int k = 0; k < D; ++k) {
    // pivot search in column k over rows k..D-1 (one wave; largest |v|, lowest row on ties)
    if (tid < 64) {
      double best = -1.0;
      int bi = k;
      for (int i = k + lane; i < D; i += 64) {
        const double v = fabs(work[i * W + k]);
        if (v > best) { best = v; bi = i; }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) {
        prow = bi;
        if (!(best > 0.0)) a.singular[0] = 1;
      }
    }
    __syncthreads();
    const int p = prow;
    if (p != k) {
      for (int j = tid; j < W; j += 1024) {
        const double t = work[k * W + j];
        work[k * W + j] = work[p * W + j];
        work[p * W + j] = t;
      }
      __syncthreads();
    }
    const double rpiv = 1.0 / work[k * W + k];
    for (int j = tid; j < W; j += 1024) prowv[j] = (j == k) ? 1.0 : work[k * W + j] * rpiv;
    for (int i = tid; i < D; i += 1024) colk[i] = work[i * W + k];
    __syncthreads();
    // eliminate column k from every other row; row k becomes the normalised pivot row
    {
      int i = i_first, j = j_first;
      for (int e = tid; e < D * W; e += 1024) {
        if (i == k) work[e] = prowv[j];
        else if (j == k) work[e] = 0.0;
        else work[e] -= colk[i] * prowv[j];
        i += di;
        j += dj;
        if (j >= W) { j -= W; ++i; }
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < D * D; e += 1024) {
    const int i = e / D, j = e - i * D;
    a.invA[e] = (float)work[i * W + D + j];
  }
  for (int i = tid; i < D; i += 1024) {
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += work[i * W + D + j] * (double)a.bvec[j];
    a.coefs[i] = (float)s;
  }
}

// A + lambda I of the regression is symmetric positive definite (a weighted Gram matrix plus a
// ridge), and Gauss-Jordan on an SPD matrix needs no pivoting: every pivot is a Schur-complement
// diagonal, positive.  Without row exchanges each thread can keep ONE COLUMN of the augmented system
// in registers for the whole solve, provided every register index is static — which a ROTATING
// frame gives: at step k the logical row (k + r) mod DR lives in register r, so the pivot row is
// always register 0, the update writes row r into register r - 1, and the finished pivot row
// re-enters at the end; after DR steps the frame is back where it started.  The system is padded
// to DR = 72 rows with an identity block (the inverse of diag(M, I) is diag(inv M, I)).  A pivot
// step is then: the owner of column k publishes its registers through LDS, one barrier, and 71
// fused multiply-adds per thread out of registers — a 2.4 KB loop.  181 us (the LDS kernel above:
// four barriers and ~25 LDS operations per element per step across 16 waves) -> ~45 us for D = 65.
// (A fully unrolled version with fixed rows was tried first: 200 KB of straight-line code ran at
// instruction-fetch speed, 550 us.)  A pivot that is not positive (lambda = 0 with rank-deficient
// data, or a negative weight) raises `need_pivot`; the pivoting kernel above then runs as a second
// launch that otherwise exits at once.
constexpr int SOLVE_DR = 72;
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linreg_solve_spd_kernel(SolveArgs a, int* need_pivot) {
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  constexpr int DR = SOLVE_DR, W = 2 * SOLVE_DR;
  const int D = a.D, t = threadIdx.x;
  double* bcast = lds_work;                   // [2][DR]: the pivot column, double-buffered
  double* work = lds_work + 2 * DR;           // [DR][W]: staging of the input, then of the result
  __shared__ int bad;
  // this workgroup shares its CU with the learner stream's launches (the solve runs beside the next
  // step): its three waves go first at the issue ports (104 us -> see profiles/r04 when they did not)
  __builtin_amdgcn_s_setprio(3);
  if (t == 0) {
    bad = 0;
    need_pivot[0] = 0;       // (the two status words: no memset launches in front of this kernel)
    a.singular[0] = 0;
  }
  const bool own = t < W;
  // diag(A + lambda I, I) | I  staged through LDS with coalesced loads
  for (int e = t; e < DR * W; e += 192) {
    const int i = e / W, j = e - i * W;
    double v;
    if (j < DR) {
      if (i < D && j < D) v = (double)a.A[i * D + j] + (i == j ? (double)a.lambda : 0.0);
      else v = (i == j) ? 1.0 : 0.0;
    } else {
      v = (j - DR == i) ? 1.0 : 0.0;
    }
    work[e] = v;
  }
  __syncthreads();
  double col[DR];
#pragma unroll
  for (int r = 0; r < DR; ++r) col[r] = own ? work[r * W + t] : 0.0;
  __syncthreads();
  // only the D real pivots: a step on a padding row (identity pivot, f = e_0) is a pure rotation of
  // the frame, accounted for when the registers are written back
  for (int k = 0; k < D; ++k) {
    double* bc = bcast + (k & 1) * DR;
    if (t == k) {
#pragma unroll
      for (int r = 0; r < DR; ++r) bc[r] = col[r];
    }
    __syncthreads();
    // the pivot column, two doubles per LDS read (every lane reads the same address: broadcast)
    double f[DR];
#pragma unroll
    for (int r = 0; r < DR; r += 2) {
      const double2 v = *reinterpret_cast<const double2*>(bc + r);
      f[r] = v.x;
      f[r + 1] = v.y;
    }
    const double piv = f[0];
    if (!(piv > 0.0) && t == 0) bad = 1;
    const double pr = col[0] * (1.0 / piv);
#pragma unroll
    for (int r = 1; r < DR; ++r) col[r - 1] = __builtin_fma(-f[r], pr, col[r]);
    col[DR - 1] = pr;
  }
  // ---- outputs: inv(A + lambda I) = rows / columns < D of the right half; coefs = inv b
  // (after D steps register r holds logical row (D + r) mod DR)
  if (own) {
#pragma unroll
    for (int r = 0; r < DR; ++r) {
      int row = D + r;
      if (row >= DR) row -= DR;
      work[row * W + t] = col[r];
    }
  }
  __syncthreads();
  if (bad) {
    if (t == 0) need_pivot[0] = 1;
    return;
  }
  for (int e = t; e < D * D; e += 192) {
    const int i = e / D, j = e - i * D;
    a.invA[e] = (float)work[i * W + DR + j];
  }
  for (int i = t; i < D; i += 192) {
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += work[i * W + DR + j] * (double)a.bvec[j];
    a.coefs[i] = (float)s;
  }
}

// Round 5: the same elimination IN PLACE, each column shared by NS lanes.
// The right half of the augmented system above is never more than bookkeeping: before step k its
// column k is still the unit vector e_k, after step k it holds column k of the partial inverse, and
// the left half's column k is dead from then on.  So column k of ONE [DR][DR] matrix can carry both
// (classic in-place Gauss-Jordan): the owner of column k publishes it, then continues as if it held
// e_k.  Every surviving fused multiply-add has the same operands in the same order as in
// linreg_solve_spd_kernel, so the two kernels agree bit for bit (tests/test_gpu_actor_critic.py);
// half of the columns are gone, and with them half of the work.
// The freed lanes split each column's rows: lane c * NS + h keeps frame rows [h HR, (h + 1) HR),
// HR = DR / NS, of column c.  The rotating frame crosses the lane boundary once per step — the
// updated first row of part h + 1 becomes the last row of part h — and the pivot row's entry lives in
// part 0: two DPP moves inside a quad (no LDS, no barrier).  A pivot step is then HR - 1 FMAs and
// HR / 2 + 1 LDS reads per lane instead of 71 and 36.
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int quad_perm(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

template <int NS, bool PRIO>
__global__ __launch_bounds__(SOLVE_DR * NS)
void linreg_solve_spd_inplace_kernel(SolveArgs a, int* need_pivot) {
  static_assert(NS == 2 || NS == 4, "a column's parts exchange rows inside one quad");
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  constexpr int DR = SOLVE_DR, HR = DR / NS, NT = DR * NS;
  static_assert(HR * NS == DR && HR % 2 == 0, "parts of whole double2s");
  const int D = a.D, t = threadIdx.x, c = t / NS, h = t % NS;
  double* bcast = lds_work;                   // [2][DR]: the pivot column, double-buffered
  double* work = lds_work + 2 * DR;           // [DR][DR]: staging of the input, then of the result
  __shared__ int bad;
  if (PRIO) __builtin_amdgcn_s_setprio(3);    // (as above: the solve shares its CU with the next step)
  if (t == 0) {
    bad = 0;
    need_pivot[0] = 0;
    a.singular[0] = 0;
  }
  // diag(A + lambda I, I) staged through LDS with coalesced loads
  for (int e = t; e < DR * DR; e += NT) {
    const int i = e / DR, j = e - i * DR;
    double v;
    if (i < D && j < D) v = (double)a.A[i * D + j] + (i == j ? (double)a.lambda : 0.0);
    else v = (i == j) ? 1.0 : 0.0;
    work[e] = v;
  }
  __syncthreads();
  double col[HR];
#pragma unroll
  for (int r = 0; r < HR; ++r) col[r] = work[(h * HR + r) * DR + c];
  __syncthreads();
  for (int k = 0; k < D; ++k) {
    double* bc = bcast + (k & 1) * DR;
    const bool owner = c == k;
    if (owner) {
#pragma unroll
      for (int r = 0; r < HR; r += 2)
        *reinterpret_cast<double2*>(bc + h * HR + r) = make_double2(col[r], col[r + 1]);
    }
    __syncthreads();
    double f[HR];
#pragma unroll
    for (int r = 0; r < HR; r += 2) {
      const double2 v = *reinterpret_cast<const double2*>(bc + h * HR + r);
      f[r] = v.x;
      f[r + 1] = v.y;
    }
    const double piv = bc[0];
    if (!(piv > 0.0) && t == 0) bad = 1;
    if (owner) {                              // from here on this column is e_k's
#pragma unroll
      for (int r = 0; r < HR; ++r) col[r] = 0.0;
      if (h == 0) col[0] = 1.0;
    }
    // the pivot row's entry of this column: part 0's first register, to every part of the column
    const double pr = dpp_quad<NS == 2 ? quad_perm(0, 0, 2, 2) : quad_perm(0, 0, 0, 0)>(col[0] * (1.0 / piv));
    // part h + 1's updated first row is part h's new last row (lane + 1 of the quad)
    const double first = __builtin_fma(-f[0], pr, col[0]);
    const double from_next = dpp_quad<NS == 2 ? quad_perm(1, 0, 3, 2) : quad_perm(1, 2, 3, 3)>(first);
#pragma unroll
    for (int r = 1; r < HR; ++r) col[r - 1] = __builtin_fma(-f[r], pr, col[r]);
    col[HR - 1] = (h == NS - 1) ? pr : from_next;
  }
  // after D steps register r of part h holds logical row (D + h HR + r) mod DR
#pragma unroll
  for (int r = 0; r < HR; ++r) {
    int row = D + h * HR + r;
    if (row >= DR) row -= DR;
    work[row * DR + c] = col[r];
  }
  __syncthreads();
  if (bad) {
    if (t == 0) need_pivot[0] = 1;
    return;
  }
  for (int e = t; e < D * D; e += NT) {
    const int i = e / D, j = e - i * D;
    a.invA[e] = (float)work[i * DR + j];
  }
  for (int i = t; i < D; i += NT) {
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += work[i * DR + j] * (double)a.bvec[j];
    a.coefs[i] = (float)s;
  }
}

// force_pinv with l2_reg_lambda = 0 (linear_regression.py:138-157): torch.linalg.pinv(A, hermitian=True)
// of a possibly singular A — eigh, then 1 / lambda_j for the eigenvalues above rtol * max |lambda|
// (rtol = D * eps of float32, torch's default), zero for the rest.  Here: a one-sided (Hestenes) Jacobi
// iteration in fp64 on the columns of G = M V (V orthogonal, starts as I): a rotation of columns p, q
// of G and V that makes G's two columns orthogonal; when all pairs are, M V = V Lambda, i.e.
// lambda_j = v_j . g_j.  Pairs of one round-robin round are disjoint, so 36 groups of 16 lanes rotate
// 36 pairs between two barriers; 71 rounds are a sweep, 6 - 10 sweeps converge to fp64 rounding.
// The matrix is padded to order 72 with zeros (their eigenvalue 0 is dropped like every other null
// direction).  One workgroup, ~1 ms: this is the rarely used corner of the bandit (an unregularised,
// rank-deficient regression), built for completeness, not for speed.
constexpr int PINV_N = 72, PINV_LD = 73, PINV_GROUPS = PINV_N / 2, PINV_THREADS = PINV_GROUPS * 16;
struct PinvArgs {
  const float* A; const float* bvec; float lambda; int D; float rtol;
  float* invA; float* coefs; int* rank; int max_sweeps;
};
__device__ __forceinline__ double group16_sum(double v) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o);     // (xor butterfly: every lane ends with the same bits)
  return v;
}
__global__ __launch_bounds__(PINV_THREADS) void linreg_pinv_kernel(PinvArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_work[];
  constexpr int N = PINV_N, LD = PINV_LD;
  double* G = lds_work;               // column j of M V at G + j * LD
  double* V = G + N * LD;             // column j of V
  double* lam = V + N * LD;           // [N] eigenvalues, then their pseudo-inverse weights
  double* proj = lam + N;             // [N] w_j (v_j . b)
  __shared__ int rotated;
  __shared__ double cut, scale2;
  const int D = a.D, t = threadIdx.x, grp = t >> 4, l16 = t & 15;
  // M = A + lambda I from the LOWER triangle (what eigh reads), zero padding; V = I
  for (int e = t; e < N * N; e += PINV_THREADS) {
    const int j = e / N, i = e - j * N;
    double m = 0.0;
    if (i < D && j < D) {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      m = (double)a.A[hi * D + lo] + (i == j ? (double)a.lambda : 0.0);
    }
    G[j * LD + i] = m;
    V[j * LD + i] = (i == j) ? 1.0 : 0.0;
  }
  if (t == 0) scale2 = 0.0;
  __syncthreads();
  // the largest squared column norm: columns below 1e-14 of it (norm) are numerically null and are
  // left alone (rotating one against a real column would move that column by 1e-14 of an angle)
  if (t < N) {
    double n2 = 0.0;
    for (int i = 0; i < N; ++i) n2 = __builtin_fma(G[t * LD + i], G[t * LD + i], n2);
    lam[t] = n2;
  }
  __syncthreads();
  if (t == 0) {
    double m = 0.0;
    for (int j = 0; j < N; ++j) m = lam[j] > m ? lam[j] : m;
    scale2 = m;
  }
  __syncthreads();
  const double null2 = 1e-28 * scale2;
  for (int sweep = 0; sweep < a.max_sweeps; ++sweep) {
    if (t == 0) rotated = 0;
    __syncthreads();
    for (int step = 0; step < N - 1; ++step) {
      // round-robin round `step`: player N - 1 stays, the others sit on a circle of N - 1 seats
      int pa_, pb_;
      if (grp == 0) {
        pa_ = N - 1;
        pb_ = step;
      } else {
        pa_ = (step + grp) % (N - 1);
        pb_ = (step - grp + (N - 1)) % (N - 1);
      }
      const int p = pa_ < pb_ ? pa_ : pb_, q = pa_ < pb_ ? pb_ : pa_;
      double* gp = G + p * LD;
      double* gq = G + q * LD;
      double al = 0.0, be = 0.0, ga = 0.0;
      for (int i = l16; i < N; i += 16) {
        const double x = gp[i], y = gq[i];
        al = __builtin_fma(x, x, al);
        be = __builtin_fma(y, y, be);
        ga = __builtin_fma(x, y, ga);
      }
      al = group16_sum(al);
      be = group16_sum(be);
      ga = group16_sum(ga);
      if (al > null2 && be > null2 && fabs(ga) > 1e-15 * sqrt(al * be)) {
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + tt * tt), sn = c * tt;
        double* vp = V + p * LD;
        double* vq = V + q * LD;
        for (int i = l16; i < N; i += 16) {
          const double x = gp[i], y = gq[i];
          gp[i] = c * x - sn * y;
          gq[i] = sn * x + c * y;
          const double u = vp[i], w = vq[i];
          vp[i] = c * u - sn * w;
          vq[i] = sn * u + c * w;
        }
        if (l16 == 0) rotated = 1;
      }
      __syncthreads();
    }
    const int again = rotated;
    __syncthreads();
    if (!again) break;
  }
  // lambda_j = v_j . g_j ; proj_j = v_j . b
  for (int j = grp; j < N; j += PINV_GROUPS) {
    double l = 0.0, pb = 0.0;
    for (int i = l16; i < N; i += 16) {
      const double v = V[j * LD + i];
      l = __builtin_fma(v, G[j * LD + i], l);
      if (i < D) pb = __builtin_fma(v, (double)a.bvec[i], pb);
    }
    l = group16_sum(l);
    pb = group16_sum(pb);
    if (l16 == 0) {
      lam[j] = l;
      proj[j] = pb;
    }
  }
  __syncthreads();
  if (t == 0) {
    double m = 0.0;
    for (int j = 0; j < N; ++j) m = fabs(lam[j]) > m ? fabs(lam[j]) : m;
    cut = (double)a.rtol * m;
    int r = 0;
    for (int j = 0; j < N; ++j) r += fabs(lam[j]) > cut ? 1 : 0;
    a.rank[0] = r;
  }
  __syncthreads();
  if (t < N) {
    const double w = fabs(lam[t]) > cut ? 1.0 / lam[t] : 0.0;
    lam[t] = w;
    proj[t] *= w;
  }
  __syncthreads();
  for (int e = t; e < D * D; e += PINV_THREADS) {
    const int r = e / D, c = e - r * D;
    double sum = 0.0;
    for (int j = 0; j < N; ++j) sum = __builtin_fma(lam[j] * V[j * LD + r], V[j * LD + c], sum);
    a.invA[e] = (float)sum;
  }
  for (int r = t; r < D; r += PINV_THREADS) {
    double sum = 0.0;
    for (int j = 0; j < N; ++j) sum = __builtin_fma(proj[j], V[j * LD + r], sum);
    a.coefs[r] = (float)sum;
  }
}

// sigma[b] = sqrt([1 | f_b] inv_A [1 | f_b]^T)      (linear_regression.py:261-270)
__global__ __launch_bounds__(256) void linreg_sigma_kernel(const float* __restrict__ f, int ldf,
                                                           const float* __restrict__ invA, int B,
                                                           int d, float* __restrict__ sigma) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int D = d + 1;
  float acc = 0.f;
  for (int j = lane; j < D; j += 64) {
    float t = 0.f;
    for (int i = 0; i < D; ++i) {
      const float xi = (i == 0) ? 1.0f : f[(int64_t)b * ldf + (i - 1)];
      t += xi * invA[(int64_t)i * D + j];
    }
    const float xj = (j == 0) ? 1.0f : f[(int64_t)b * ldf + (j - 1)];
    acc += t * xj;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) sigma[b] = sqrtf(acc);
}

}  // namespace

extern "C" int pa_weighted_loss_head(const float* pred, int32_t ldp, const float* y, const float* w,
                                     int32_t B, int32_t loss_kind, int32_t out_act, float* pred_out,
                                     float* d_pred, float* loss_out, float* wsum_out, void* stream) {
  PA_REQUIRE(pred && y && d_pred && loss_out && B > 0, PA_ERR_INVALID,
             "pa_weighted_loss_head: bad argument");
  PA_REQUIRE(loss_kind >= PA_LOSS_MSE && loss_kind <= PA_LOSS_BCE && out_act >= PA_OUT_LINEAR &&
             out_act <= PA_OUT_SIGMOID, PA_ERR_UNSUPPORTED,
             "pa_weighted_loss_head: loss is mse / mae / cross-entropy, output activation linear / sigmoid");
  WmseArgs a;
  a.pred = pred; a.ldp = ldp; a.y = y; a.w = w; a.B = B; a.d_pred = d_pred; a.loss_out = loss_out;
  a.wsum_out = wsum_out; a.loss_kind = loss_kind; a.out_act = out_act; a.pred_out = pred_out;
  hipLaunchKernelGGL(wmse_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_weighted_mse_head(const float* pred, int32_t ldp, const float* y, const float* w,
                                    int32_t B, float* d_pred, float* loss_out, float* wsum_out,
                                    void* stream) {
  return pa_weighted_loss_head(pred, ldp, y, w, B, PA_LOSS_MSE, PA_OUT_LINEAR, nullptr, d_pred,
                               loss_out, wsum_out, stream);
}

// padded: 0 = the packed layout of the original entry point (X [B][D], R [B][D + 1]); 1 = row
// pitches rounded up to whole 16- / 8-byte vectors (X [B][Dp], Dp = D rounded up to 4; R [B][Rp],
// Rp = D + 1 rounded up to 2; pad columns zero), which is what lets batches of >= 2048 contexts take
// the bf16x3 weight-gradient loop for this GEMM too
struct LinregPitch { int ldX, ldR; };
static LinregPitch linreg_pitch(int D, int padded) {
  LinregPitch p;
  p.ldX = padded ? (D + 3) & ~3 : D;
  p.ldR = padded ? (D + 2) & ~1 : D + 1;
  return p;
}
static int linreg_operands(const float* features, int32_t ldf, const float* y, const float* w,
                           int32_t B, int32_t d, float* x_scratch, float* r_scratch, int padded,
                           hipStream_t s) {
  const int D = d + 1;
  const LinregPitch lp = linreg_pitch(D, padded);
  const int W = lp.ldR > lp.ldX ? lp.ldR : lp.ldX;
  const int64_t total = (int64_t)B * W;
  const unsigned grid = (unsigned)(ceil_div(total, 256) > 2048 ? 2048 : ceil_div(total, 256));
  hipLaunchKernelGGL(linreg_operands_kernel, dim3(grid), dim3(256), 0, s, features, ldf, y, w, B, d,
                     x_scratch, lp.ldX, r_scratch, lp.ldR);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
// X^T R = [delta_A | delta_b] ([D][D+1]) as a weight-gradient problem; the "bias" output of the
// kernel (column sums of X) goes to the scratch tail: its first entry (the ones column sum) is NOT
// the weight sum — that is delta_A[0][0] = sum_b 1 * 1 * w_b.
static DwProblem linreg_delta_problem(int32_t B, int32_t d, float* x_scratch, float* r_scratch,
                                      float* delta_out, int padded) {
  const int D = d + 1;
  const LinregPitch lp = linreg_pitch(D, padded);
  DwProblem pr;
  memset(&pr, 0, sizeof(pr));
  pr.dZ = x_scratch; pr.ldz = lp.ldX;
  pr.X = r_scratch; pr.ldx = lp.ldR;
  pr.dW = delta_out; pr.ldw = D + 1;
  pr.db = x_scratch + (int64_t)B * lp.ldX;   // scratch tail: D floats
  pr.M = D; pr.N = D + 1;
  pr.tiles_n = (int)ceil_div(D + 1, DW_TN);
  pr.tile0 = 0;
  pr.kind = 2;
  return pr;
}
static int linreg_delta_impl(const float* features, int32_t ldf, const float* y, const float* w,
                             int32_t B, int32_t d, float* x_scratch, float* r_scratch,
                             float* delta_out, int padded, hipStream_t s) {
  const int D = d + 1;
  int rc = linreg_operands(features, ldf, y, w, B, d, x_scratch, r_scratch, padded, s);
  if (rc != PA_OK) return rc;
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.nprob = 1;
  a.p[0] = linreg_delta_problem(B, d, x_scratch, r_scratch, delta_out, padded);
  a.total_tiles = (int)ceil_div(D, DW_TM) * a.p[0].tiles_n;
  a.B = B;
  {
    int rcw = launch_weight_grad(a, false, s);
    if (rcw != PA_OK) return rcw;
  }
  // the weight sum of the batch rides behind the matrix (one message for a data-parallel step)
  PA_HIP(hipMemcpyAsync(delta_out + (int64_t)D * (D + 1), delta_out, sizeof(float),
                        hipMemcpyDeviceToDevice, s));
  return PA_OK;
}
extern "C" int pa_linreg_delta(const float* features, int32_t ldf, const float* y, const float* w,
                               int32_t B, int32_t d, float* x_scratch, float* r_scratch,
                               float* delta_out, void* stream) {
  PA_REQUIRE(features && y && x_scratch && r_scratch && delta_out && B > 0 && d > 0, PA_ERR_INVALID,
             "pa_linreg_delta: bad argument");
  return linreg_delta_impl(features, ldf, y, w, B, d, x_scratch, r_scratch, delta_out, 0,
                           reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pa_linreg_delta2(const float* features, int32_t ldf, const float* y, const float* w,
                                int32_t B, int32_t d, float* x_scratch, float* r_scratch,
                                float* delta_out, void* stream) {
  PA_REQUIRE(features && y && x_scratch && r_scratch && delta_out && B > 0 && d > 0, PA_ERR_INVALID,
             "pa_linreg_delta2: bad argument");
  PA_REQUIRE(((reinterpret_cast<uintptr_t>(x_scratch) | reinterpret_cast<uintptr_t>(r_scratch)) & 15) == 0,
             PA_ERR_INVALID, "pa_linreg_delta2: scratch buffers must be 16-byte aligned");
  return linreg_delta_impl(features, ldf, y, w, B, d, x_scratch, r_scratch, delta_out, 1,
                           reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_linreg_apply2(const float* delta, int32_t d, float* A, float* b, float* sum_weight,
                                float* A_snap, float* b_snap, void* stream) {
  PA_REQUIRE(delta && A && b && sum_weight && d > 0, PA_ERR_INVALID, "pa_linreg_apply: bad argument");
  PA_REQUIRE((A_snap == nullptr) == (b_snap == nullptr), PA_ERR_INVALID,
             "pa_linreg_apply2: both snapshots or none");
  const int D = d + 1;
  hipLaunchKernelGGL(linreg_apply_kernel, dim3((unsigned)ceil_div((int64_t)D * D, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), delta, D, A, b, sum_weight, A_snap, b_snap,
                     delta + (int64_t)D * (D + 1));
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// One NeuralLinearBandit.learn_batch on unit weights, single process, as one call (neural_linear_
// bandit.py:139-214): the fused row step (forward kept, loss head, backward), the LinUCB operands
// from the kept features, ONE weight-gradient launch that forms the network's gradients with their
// AdamW step AND the [D x (D+1)] moment update X^T R, and the in-place update of A, b, sum_weight
// (with the snapshot the asynchronous solve reads).  Four launches on `stream`.
extern "C" int pa_bandit_step(const pa_bandit_step_args* g, void* stream) {
  PA_REQUIRE(g && g->net && g->x && g->y && g->pred && g->d_pred && g->scalars && g->x_scratch &&
                 g->r_scratch && g->delta && g->A && g->b && g->sum_weight && g->B > 0 && g->d > 0,
             PA_ERR_INVALID, "pa_bandit_step: bad argument");
  PA_REQUIRE((g->A_snap == nullptr) == (g->b_snap == nullptr), PA_ERR_INVALID,
             "pa_bandit_step: both snapshots or none");
  PA_REQUIRE(g->loss_kind >= PA_LOSS_MSE && g->loss_kind <= PA_LOSS_BCE &&
                 g->out_act >= PA_OUT_LINEAR && g->out_act <= PA_OUT_SIGMOID, PA_ERR_UNSUPPORTED,
             "pa_bandit_step: loss is mse / mae / cross-entropy, output activation linear / sigmoid");
  pa_mlp* net = g->net;
  PA_REQUIRE(pa_rowstep_supported(net, nullptr, 0), PA_ERR_UNSUPPORTED,
             "pa_bandit_step: needs a one-output network, every layer <= 256 wide");
  PA_REQUIRE(net->bound && net->bufs.grad && net->bufs.exp_avg && net->bufs.exp_avg_sq &&
                 (!net->d.amsgrad || net->bufs.max_exp_avg_sq), PA_ERR_INVALID,
             "pa_bandit_step: optimizer buffers not bound");
  PA_REQUIRE(g->adam_step >= 1, PA_ERR_INVALID, "adam step must be >= 1");
  PA_REQUIRE(!g->side_stream || (g->A_snap && g->ev_ready && g->ev_done && g->work && g->inv_A &&
                                 g->coefs && g->singular), PA_ERR_INVALID,
             "pa_bandit_step: the side-stream solve needs the snapshot pair, two events and its buffers");
  PA_REQUIRE(net->L >= 2 && net->d.dims[net->L - 1] == g->d, PA_ERR_INVALID,
             "pa_bandit_step: the regression's feature width is not the trunk's output width");
  PA_REQUIRE(((reinterpret_cast<uintptr_t>(g->x_scratch) | reinterpret_cast<uintptr_t>(g->r_scratch)) & 15) == 0,
             PA_ERR_INVALID, "pa_bandit_step: scratch buffers must be 16-byte aligned");
  PA_HIP(hipSetDevice(net->d.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  pa_mlp* hs[1] = {net};
  RowHead head;
  memset(&head, 0, sizeof(head));
  head.kind = RS_HEAD_WMSE1;
  head.d_out = g->d_pred; head.ldd = 1;
  head.target = g->y;
  head.loss_kind = g->loss_kind; head.out_act = g->out_act;
  head.out_post = g->out_act == PA_OUT_LINEAR ? nullptr : g->pred;
  head.mean_out = g->scalars + 1;
  // the LinUCB operands straight from the row step's feature tile (fp32-forward launches: the
  // bf16x3 forward keeps no fp32 feature tile in LDS, linreg_operands_kernel runs then)
  {
    const LinregPitch lp = linreg_pitch(g->d + 1, 1);
    head.lin_x = g->x_scratch; head.lin_r = g->r_scratch;
    head.ldX = lp.ldX; head.ldR = lp.ldR;
  }
  // (a linear output: the network outputs ARE the predictions; a sigmoid: the pre-activation
  //  outputs are not kept)
  float* outs[1] = {g->out_act == PA_OUT_LINEAR ? g->pred : nullptr};
  const int ldos[1] = {1};
  int ran_split = 0;
  int rc = run_rowstep(hs, 1, g->x, g->ldx, g->B, &head, outs, ldos, g->scalars, 1, s, &ran_split);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(net->pend.active, PA_ERR_INVALID, "pa_bandit_step: the row step left no pending gradients");
  // the features of THIS forward (the trunk's output, kept by the row step) feed the regression
  if (ran_split != 0) {
    rc = linreg_operands(net->act[net->L - 2], net->d.dims[net->L - 1], g->y, nullptr, g->B, g->d,
                         g->x_scratch, g->r_scratch, 1, s);
    if (rc != PA_OK) return rc;
  }
  const DwProblem extra = linreg_delta_problem(g->B, g->d, g->x_scratch, g->r_scratch, g->delta, 1);
  net->pend.active = false;
  DwOperands op = {net->pend.x, net->pend.ldx, net->pend.dzs, net->pend.ldzs};
  rc = run_weight_grads_n(hs, &op, 1, net->pend.B, g->adam_step, -1.f, s, nullptr, &extra);
  if (rc != PA_OK) return rc;
  const int D = g->d + 1;
  // (two steps back: normally long finished — then no wait packet in front of the update, ~6 us)
  if (g->ev_slot_free && hipEventQuery(reinterpret_cast<hipEvent_t>(g->ev_slot_free)) != hipSuccess) {
    (void)hipGetLastError();   // hipErrorNotReady is not an error here
    PA_HIP(hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(g->ev_slot_free), 0));
  }
  hipLaunchKernelGGL(linreg_apply_kernel, dim3((unsigned)ceil_div((int64_t)D * D, 256)), dim3(256), 0,
                     s, g->delta, D, g->A, g->b, g->sum_weight, g->A_snap, g->b_snap, g->delta);
  PA_LAUNCH_CHECK();
  if (g->side_stream) {
    hipStream_t side = reinterpret_cast<hipStream_t>(g->side_stream);
    PA_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(g->ev_ready), s));
    PA_HIP(hipStreamWaitEvent(side, reinterpret_cast<hipEvent_t>(g->ev_ready), 0));
    rc = pa_linreg_solve(g->A_snap, g->b_snap, g->l2_reg_lambda, g->d, g->work, g->inv_A, g->coefs,
                         g->singular, g->side_stream);
    if (rc != PA_OK) return rc;
    PA_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(g->ev_done), side));
  }
  return PA_OK;
}
extern "C" int pa_linreg_apply(const float* delta, int32_t d, float* A, float* b, float* sum_weight,
                               void* stream) {
  return pa_linreg_apply2(delta, d, A, b, sum_weight, nullptr, nullptr, stream);
}

extern "C" int pa_linreg_solve(const float* A, const float* b, float l2_reg_lambda, int32_t d,
                               double* work, float* inv_A_out, float* coefs_out, int32_t* singular_out,
                               void* stream) {
  PA_REQUIRE(A && b && work && inv_A_out && coefs_out && singular_out && d > 0, PA_ERR_INVALID,
             "pa_linreg_solve: bad argument");
  SolveArgs a;
  a.A = A; a.bvec = b; a.lambda = l2_reg_lambda; a.D = d + 1; a.work = work; a.invA = inv_A_out;
  a.coefs = coefs_out; a.singular = singular_out;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int D = d + 1;
  const size_t lds = sizeof(double) * ((size_t)D * 2 * D + 2 * D + D);
  a.only_if = nullptr;
  static const bool spd_on = []() {
    const char* v = getenv("PEARL_AMD_SOLVE_SPD");
    return !(v && *v == '0');
  }();
  if (spd_on && D <= SOLVE_DR) {
    // SPD fast path; `work` (caller's fp64 scratch, unused by the LDS kernels) carries the
    // "a pivot was not positive" word that arms the pivoting kernel below
    // (the kernel clears both status words itself: two 5 us memset launches less in the chain)
    int* need_pivot = reinterpret_cast<int*>(work);
    // PEARL_AMD_SOLVE_SPLIT: lanes per column of the in-place kernel (2 or 4, default 4); 1 = the augmented
    // kernel of round 4 (kept: the bit-for-bit comparison of tests/ runs against it)
    const int split = []() {
      const char* v = getenv("PEARL_AMD_SOLVE_SPLIT");
      return (v && *v) ? atoi(v) : 4;
    }();
    PA_REQUIRE(split == 1 || split == 2 || split == 4, PA_ERR_INVALID,
               "PEARL_AMD_SOLVE_SPLIT must be 1, 2 or 4");
    if (split == 1) {
      const size_t lds_spd = sizeof(double) * (2 * SOLVE_DR + (size_t)SOLVE_DR * 2 * SOLVE_DR);
      static size_t configured_spd = 0;
      if (lds_spd > configured_spd) {
        int rc = set_max_smem(linreg_solve_spd_kernel, lds_spd);
        if (rc != PA_OK) return rc;
        configured_spd = lds_spd;
      }
      hipLaunchKernelGGL(linreg_solve_spd_kernel, dim3(1), dim3(192), lds_spd, s, a, need_pivot);
    } else {
      const size_t lds_in = sizeof(double) * (2 * SOLVE_DR + (size_t)SOLVE_DR * SOLVE_DR);   // 41.6 KB
      // no raised priority by default: the solve is off the learner stream's chain, and five waves that
      // go first at the issue ports of the CU they share cost the next step's row pass 2.5 % (measured:
      // bandit 46.9 M contexts/s with, 48.3 M without — what the augmented kernel's three waves gave)
      const char* pv = getenv("PEARL_AMD_SOLVE_PRIO");
      const bool prio = pv && *pv == '1';
      void (*k)(SolveArgs, int*) =
          split == 2 ? (prio ? linreg_solve_spd_inplace_kernel<2, true> : linreg_solve_spd_inplace_kernel<2, false>)
                     : (prio ? linreg_solve_spd_inplace_kernel<4, true> : linreg_solve_spd_inplace_kernel<4, false>);
      hipLaunchKernelGGL(k, dim3(1), dim3(SOLVE_DR * split), lds_in, s, a, need_pivot);
    }
    PA_LAUNCH_CHECK();
    a.only_if = need_pivot;
  } else {
    PA_HIP(hipMemsetAsync(singular_out, 0, sizeof(int32_t), s));
  }
  if (lds <= 150 * 1024) {
    static size_t configured = 0;
    if (lds > configured) {
      int rc = set_max_smem(linreg_solve_lds_kernel, lds);
      if (rc != PA_OK) return rc;
      configured = lds;
    }
    hipLaunchKernelGGL(linreg_solve_lds_kernel, dim3(1), dim3(1024), lds, s, a);
  } else {
    hipLaunchKernelGGL(linreg_solve_kernel, dim3(1), dim3(1024), 0, s, a);
  }
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_linreg_pinv(const float* A, const float* b, float l2_reg_lambda, int32_t d,
                              float* inv_A_out, float* coefs_out, int32_t* rank_out, void* stream) {
  PA_REQUIRE(A && b && inv_A_out && coefs_out && rank_out && d > 0, PA_ERR_INVALID,
             "pa_linreg_pinv: bad argument");
  PA_REQUIRE(d + 1 <= PINV_N, PA_ERR_UNSUPPORTED,
             "pa_linreg_pinv: the Jacobi kernel holds systems up to order 72 (got %d)", d + 1);
  PinvArgs a;
  a.A = A; a.bvec = b; a.lambda = l2_reg_lambda; a.D = d + 1;
  a.rtol = (float)(d + 1) * 1.1920929e-07f;      // torch.linalg.pinv's default: max(m, n) * eps(float32)
  a.invA = inv_A_out; a.coefs = coefs_out; a.rank = rank_out; a.max_sweeps = 40;
  const size_t lds = sizeof(double) * (2 * (size_t)PINV_N * PINV_LD + 2 * PINV_N);
  static bool configured = false;
  if (!configured) {
    int rc = set_max_smem(linreg_pinv_kernel, lds);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  hipLaunchKernelGGL(linreg_pinv_kernel, dim3(1), dim3(PINV_THREADS), lds,
                     reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_linreg_sigma(const float* features, int32_t ldf, const float* inv_A, int32_t B,
                               int32_t d, float* sigma_out, void* stream) {
  PA_REQUIRE(features && inv_A && sigma_out && B > 0 && d > 0, PA_ERR_INVALID,
             "pa_linreg_sigma: bad argument");
  hipLaunchKernelGGL(linreg_sigma_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), features, ldf, inv_A, B, d, sigma_out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_softmax_action_prob(const float* logits, int32_t ldl, const float* action_rep,
                                      int32_t lda, int32_t B, int32_t A, float* probs_out,
                                      float* action_prob_out, void* stream) {
  PA_REQUIRE(logits && action_rep && action_prob_out && B > 0 && A > 0, PA_ERR_INVALID,
             "pa_softmax_action_prob: bad argument");
  hipLaunchKernelGGL(action_prob_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), logits, ldl, action_rep, lda, B, A,
                     probs_out, action_prob_out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

namespace {
int ppo_actor_launch(const float* logits, int32_t ldl, const float* action_rep, int32_t lda,
                     const float* p_old, const float* gae, int32_t B, int32_t A, float epsilon,
                     float entropy_scale, float* d_logits, int32_t ldd, float* loss_out,
                     const MseArgs* critic, void* stream);
}
extern "C" int pa_ppo_actor_loss(const float* logits, int32_t ldl, const float* action_rep,
                                 int32_t lda, const float* p_old, const float* gae, int32_t B,
                                 int32_t A, float epsilon, float entropy_scale, float* d_logits,
                                 int32_t ldd, float* loss_out, void* stream) {
  return ppo_actor_launch(logits, ldl, action_rep, lda, p_old, gae, B, A, epsilon, entropy_scale,
                          d_logits, ldd, loss_out, nullptr, stream);
}
// The actor head and the critic's MSE head of one PPO minibatch (ppo.py:152-192) in ONE launch:
// the value head (a single workgroup: serial sum in fixed order) runs beside the actor head's
// blocks instead of after them.  Same values as pa_ppo_actor_loss + pa_mse_head(…, 2 / B, 1, 0, …).
extern "C" int pa_ppo_heads(const float* logits, int32_t ldl, const float* action_rep, int32_t lda,
                            const float* p_old, const float* gae, int32_t B, int32_t A,
                            float epsilon, float entropy_scale, float* d_logits, int32_t ldd,
                            const float* value, int32_t ldv, const float* value_target,
                            float* d_value, float* losses, void* stream) {
  PA_REQUIRE(value && value_target && d_value && losses, PA_ERR_INVALID, "pa_ppo_heads: bad argument");
  MseArgs m;
  m.pred = value; m.ldp = ldv; m.target = value_target; m.B = B; m.grad_scale = 2.0f / (float)B;
  m.d_pred = d_value; m.loss_out = losses + 1; m.loss_scale = 1.0f; m.accumulate = 0;
  return ppo_actor_launch(logits, ldl, action_rep, lda, p_old, gae, B, A, epsilon, entropy_scale,
                          d_logits, ldd, losses, &m, stream);
}
namespace {
int ppo_actor_launch(const float* logits, int32_t ldl, const float* action_rep, int32_t lda,
                     const float* p_old, const float* gae, int32_t B, int32_t A, float epsilon,
                     float entropy_scale, float* d_logits, int32_t ldd, float* loss_out,
                     const MseArgs* critic, void* stream) {
  PA_REQUIRE(logits && action_rep && p_old && gae && d_logits && loss_out && B > 0 && A > 0,
             PA_ERR_INVALID, "pa_ppo_actor_loss: bad argument");
  PpoActorArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = logits; a.ldl = ldl; a.arep = action_rep; a.lda = lda; a.p_old = p_old; a.gae = gae;
  a.B = B; a.A = A; a.eps = epsilon; a.ent_scale = entropy_scale;
  a.d_logits = d_logits; a.ldd = ldd; a.loss_out = loss_out;
  // scratch (row probabilities, per-block partial sums, ticket): one buffer per stream
  // (stream_scratch, common.hpp), grown on demand
  const bool elem = A <= 256;
  const unsigned grid = elem ? (unsigned)ceil_div(B, 256 / A) : (unsigned)ceil_div(B, 256);
  float* scratch = stream_scratch(SCR_PPO_HEAD, reinterpret_cast<hipStream_t>(stream), (size_t)B + 2 * grid + 4);
  if (!scratch) return PA_ERR_NOMEM;
  a.ticket = reinterpret_cast<unsigned*>(scratch);
  a.partials = scratch + 4;
  a.p_rows = scratch + 4 + 2 * grid;
  const size_t lds = (size_t)2 * 256 * (A + 1) * sizeof(float);
  if (critic && !elem) {   // wide action sets: the row-per-thread kernels, then the value head
    hipLaunchKernelGGL(mse_head_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       *critic);
    PA_LAUNCH_CHECK();
  }
  if (elem) {
    if (critic) {
      a.critic = *critic;
      a.has_critic = 1;
    }
    hipLaunchKernelGGL(ppo_actor_elem_kernel, dim3(grid + (critic ? 1u : 0u)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
  } else if (lds <= 48 * 1024)
    hipLaunchKernelGGL(ppo_actor_kernel<true>, dim3(grid), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(ppo_actor_kernel<false>, dim3(grid), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
}  // namespace

extern "C" int pa_mse_head(const float* pred, int32_t ldp, const float* target, int32_t B,
                           float grad_scale, float loss_scale, int32_t accumulate, float* d_pred,
                           float* loss_out, void* stream) {
  PA_REQUIRE(pred && target && B > 0, PA_ERR_INVALID, "pa_mse_head: bad argument");
  MseArgs a;
  a.pred = pred; a.ldp = ldp; a.target = target; a.B = B; a.grad_scale = grad_scale;
  a.d_pred = d_pred; a.loss_out = loss_out; a.loss_scale = loss_scale; a.accumulate = accumulate;
  hipLaunchKernelGGL(mse_head_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_ppo_gae(const float* reward, const uint8_t* terminated, const uint8_t* truncated,
                          const float* values, const float* next_value_last, float gamma, float lam,
                          int64_t N, float* gae_out, float* lam_return_out, void* stream) {
  PA_REQUIRE(reward && terminated && truncated && values && next_value_last && gae_out &&
                 lam_return_out && N > 0,
             PA_ERR_INVALID, "pa_ppo_gae: bad argument");
  GaeArgs a;
  a.reward = reward; a.term = terminated; a.trunc = truncated; a.values = values;
  a.next_value_last = next_value_last;
  a.gamma = gamma;
  a.gl = (float)((double)gamma * (double)lam);
  a.N = N; a.gae = gae_out; a.lam_return = lam_return_out;
  hipLaunchKernelGGL(gae_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_gauss_sample(const float* head, int32_t ldh, const float* noise, int32_t ldn,
                               const float* low, const float* high, int32_t B, int32_t A,
                               float* action, int32_t lda, float* log_prob, void* stream) {
  PA_REQUIRE(head && noise && low && high && action && log_prob && B > 0 && A > 0, PA_ERR_INVALID,
             "pa_gauss_sample: bad argument");
  GaussArgs a;
  a.head = head; a.ldh = ldh; a.noise = noise; a.ldn = ldn; a.low = low; a.high = high;
  a.B = B; a.A = A; a.action = action; a.lda = lda; a.log_prob = log_prob;
  PA_REQUIRE(A <= 256, PA_ERR_UNSUPPORTED, "pa_gauss_sample: action dimension %d > 256", A);
  hipLaunchKernelGGL(gauss_sample_kernel, dim3((unsigned)ceil_div(B, 256 / A)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_gauss_actor_grad(const float* head, int32_t ldh, const float* noise, int32_t ldn,
                                   const float* low, const float* high, const float* dl_daction,
                                   const float* dl_daction2, int32_t ldda, const float* alpha,
                                   int32_t B, int32_t A,
                                   float* d_head, int32_t lddh, void* stream) {
  PA_REQUIRE(head && noise && low && high && dl_daction && alpha && d_head && B > 0 && A > 0,
             PA_ERR_INVALID, "pa_gauss_actor_grad: bad argument");
  GaussGradArgs a;
  a.head = head; a.ldh = ldh; a.noise = noise; a.ldn = ldn; a.low = low; a.high = high;
  a.dl_da = dl_daction; a.dl_da2 = dl_daction2; a.ldda = ldda; a.alpha = alpha; a.B = B; a.A = A;
  a.d_head = d_head; a.lddh = lddh;
  hipLaunchKernelGGL(gauss_grad_kernel, dim3((unsigned)ceil_div((int64_t)B * A, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_expand_state_actions(const float* state, int32_t lds_, const float* rep,
                                       int64_t rep_bstride, int32_t B, int32_t A, int32_t S,
                                       int32_t AD, float* x_out, void* stream) {
  PA_REQUIRE(state && rep && x_out && B > 0 && A > 0 && S > 0 && AD > 0, PA_ERR_INVALID,
             "pa_expand_state_actions: bad argument");
  const int64_t total = (int64_t)B * A * (S + AD);
  const unsigned grid = (unsigned)(ceil_div(total, 256) > 4096 ? 4096 : ceil_div(total, 256));
  hipLaunchKernelGGL(expand_state_actions_kernel, dim3(grid), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), state, lds_, rep, rep_bstride, x_out, B,
                     A, S, AD);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

namespace {
int launch_dsac(DsacArgs& a, hipStream_t s) {
  if (a.A > 256) {
    hipLaunchKernelGGL(dsac_kernel, dim3(1), dim3(256), 0, s, a);
    PA_LAUNCH_CHECK();
    return PA_OK;
  }
  // per-block partial sums + ticket: one buffer per stream (stream_scratch, common.hpp)
  const unsigned grid = (unsigned)ceil_div(a.B, 256 / a.A);
  float* scratch = stream_scratch(SCR_DSAC_HEAD, s, (size_t)grid + 4);
  if (!scratch) return PA_ERR_NOMEM;
  a.ticket = reinterpret_cast<unsigned*>(scratch);
  a.partials = scratch + 4;
  hipLaunchKernelGGL(dsac_elem_kernel, dim3(grid), dim3(256), 0, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
}  // namespace

extern "C" int pa_dsac_actor_head(const float* logits, int32_t ldl, const float* q1, const float* q2,
                                  const uint8_t* mask, const float* alpha, int32_t B, int32_t A,
                                  float* d_logits, int32_t ldd, float* loss_out, float* h_out,
                                  void* stream) {
  PA_REQUIRE(logits && q1 && q2 && alpha && d_logits && loss_out && h_out && B > 0 && A > 0,
             PA_ERR_INVALID, "pa_dsac_actor_head: bad argument");
  DsacArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = logits; a.ldl = ldl; a.q1 = q1; a.q2 = q2; a.mask = mask; a.alpha = alpha;
  a.B = B; a.A = A; a.mode = 0;
  a.d_logits = d_logits; a.ldd = ldd; a.loss_out = loss_out; a.h_out = h_out;
  return launch_dsac(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_dsac_target(const float* logits, int32_t ldl, const float* q1, const float* q2,
                              const uint8_t* mask, const float* alpha, const float* reward,
                              const uint8_t* terminated, float gamma, int32_t B, int32_t A,
                              float* y, void* stream) {
  PA_REQUIRE(logits && q1 && q2 && alpha && reward && terminated && y && B > 0 && A > 0,
             PA_ERR_INVALID, "pa_dsac_target: bad argument");
  DsacArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = logits; a.ldl = ldl; a.q1 = q1; a.q2 = q2; a.mask = mask; a.alpha = alpha;
  a.B = B; a.A = A; a.mode = 1;
  a.reward = reward; a.term = terminated; a.gamma = gamma; a.y = y;
  return launch_dsac(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_iql_value_head(const float* tq_value, const float* tq_actor, const float* v,
                                 int32_t ldv, float expectile, float temperature, float adv_clamp,
                                 int32_t B, float* dv, float* adv_out, float* loss_out,
                                 void* stream) {
  PA_REQUIRE(tq_value && tq_actor && v && dv && adv_out && loss_out && B > 0, PA_ERR_INVALID,
             "pa_iql_value_head: bad argument");
  IqlValueArgs a;
  a.tq_value = tq_value; a.tq_actor = tq_actor; a.v = v; a.ldv = ldv;
  a.expectile = expectile; a.temperature = temperature; a.adv_clamp = adv_clamp; a.B = B;
  a.dv = dv; a.adv = adv_out; a.loss_out = loss_out;
  hipLaunchKernelGGL(iql_value_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_awr_head(int32_t mode, const float* x, int32_t ldx, const float* action,
                           int32_t lda, const float* adv, int32_t B, int32_t A, float* dx,
                           int32_t lddx, float* loss_out, void* stream) {
  PA_REQUIRE(x && action && adv && dx && loss_out && B > 0 && A > 0 && (mode == 0 || mode == 1),
             PA_ERR_INVALID, "pa_awr_head: bad argument");
  AwrArgs a;
  a.x = x; a.ldx = ldx; a.action = action; a.lda = lda; a.adv = adv; a.B = B; a.A = A; a.mode = mode;
  a.dx = dx; a.lddx = lddx; a.loss_out = loss_out;
  hipLaunchKernelGGL(awr_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_gauss_awr_head(const float* head, int32_t ldh, const float* action, int32_t lda,
                                 const float* low, const float* high, const float* adv, int32_t B,
                                 int32_t A, float* d_head, int32_t lddh, float* log_prob,
                                 float* loss_out, void* stream) {
  PA_REQUIRE(head && action && low && high && adv && d_head && log_prob && loss_out && B > 0 &&
                 A > 0 && A <= 256,
             PA_ERR_INVALID, "pa_gauss_awr_head: bad argument");
  GaussAwrArgs a;
  a.head = head; a.ldh = ldh; a.action = action; a.lda = lda; a.low = low; a.high = high;
  a.adv = adv; a.B = B; a.A = A; a.d_head = d_head; a.lddh = lddh; a.log_prob = log_prob;
  a.loss_out = loss_out;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gauss_awr_kernel, dim3((unsigned)ceil_div(B, 256 / A)), dim3(256), 0, s, a);
  PA_LAUNCH_CHECK();
  hipLaunchKernelGGL(gauss_awr_loss_kernel, dim3(1), dim3(256), 0, s, adv, log_prob, B, loss_out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_cql_head(const float* q_rows, const float* y, const float* action, int32_t lda,
                           int32_t B, int32_t A, int32_t AD, float alpha, float* dq_rows,
                           float* loss_out, void* stream) {
  PA_REQUIRE(q_rows && y && action && dq_rows && loss_out && B > 0 && A > 0 && AD > 0,
             PA_ERR_INVALID, "pa_cql_head: bad argument");
  CqlArgs a;
  a.q = q_rows; a.y = y; a.action = action; a.lda = lda; a.B = B; a.A = A; a.AD = AD;
  a.alpha = alpha; a.dq = dq_rows; a.loss_out = loss_out;
  hipLaunchKernelGGL(cql_head_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_tanh_action(const float* head, int32_t ldh, const float* noise, int32_t ldn,
                              const float* low, const float* high, float noise_clip, int32_t B,
                              int32_t A, float* action, int32_t lda, void* stream) {
  PA_REQUIRE(head && low && high && action && B > 0 && A > 0, PA_ERR_INVALID,
             "pa_tanh_action: bad argument");
  TanhActArgs a;
  a.head = head; a.ldh = ldh; a.noise = noise; a.ldn = ldn; a.low = low; a.high = high;
  a.clip = noise_clip; a.B = B; a.A = A; a.action = action; a.lda = lda;
  hipLaunchKernelGGL(tanh_action_kernel, dim3((unsigned)ceil_div((int64_t)B * A, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_tanh_action_grad(const float* head, int32_t ldh, const float* low,
                                   const float* high, const float* dl_daction, int32_t ldda,
                                   int32_t B, int32_t A, float* d_head, int32_t lddh,
                                   void* stream) {
  PA_REQUIRE(head && low && high && dl_daction && d_head && B > 0 && A > 0, PA_ERR_INVALID,
             "pa_tanh_action_grad: bad argument");
  TanhGradArgs a;
  a.head = head; a.ldh = ldh; a.low = low; a.high = high; a.dl_da = dl_daction; a.ldda = ldda;
  a.B = B; a.A = A; a.d_head = d_head; a.lddh = lddh;
  hipLaunchKernelGGL(tanh_action_grad_kernel, dim3((unsigned)ceil_div((int64_t)B * A, 256)),
                     dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_neg_mean_head(const float* q, int32_t ldq, int32_t B, float* dq, float* loss_out,
                                void* stream) {
  PA_REQUIRE(q && dq && loss_out && B > 0, PA_ERR_INVALID, "pa_neg_mean_head: bad argument");
  hipLaunchKernelGGL(neg_mean_head_kernel, dim3(1), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), q, ldq, B, dq, loss_out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_sac_twin(int32_t mode, const float* q1, const float* q2, const float* log_prob,
                           const float* alpha, const float* reward, const uint8_t* terminated,
                           float gamma, int32_t B, float* out1, float* out2, float* loss_out,
                           void* stream) {
  PA_REQUIRE(q1 && q2 && log_prob && alpha && out1 && B > 0 && (mode == 0 || mode == 1),
             PA_ERR_INVALID, "pa_sac_twin: bad argument");
  PA_REQUIRE(mode == 0 ? (out2 && loss_out) : (reward && terminated), PA_ERR_INVALID,
             "pa_sac_twin: missing mode-specific argument");
  TwinArgs a;
  a.q1 = q1; a.q2 = q2; a.logp = log_prob; a.alpha = alpha; a.reward = reward; a.term = terminated;
  a.gamma = gamma; a.B = B; a.mode = mode; a.out1 = out1; a.out2 = out2; a.loss_out = loss_out;
  hipLaunchKernelGGL(twin_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_sac_alpha_step(float* log_alpha, float* exp_avg, float* exp_avg_sq,
                                 float* max_exp_avg_sq, float* alpha_out, const float* log_prob,
                                 int32_t B, float target_entropy, double lr, double beta1,
                                 double beta2, double eps, double weight_decay, int32_t amsgrad,
                                 int64_t step, float* loss_out, void* stream) {
  PA_REQUIRE(log_alpha && exp_avg && exp_avg_sq && alpha_out && log_prob && B > 0 && step >= 1,
             PA_ERR_INVALID, "pa_sac_alpha_step: bad argument");
  PA_REQUIRE(!amsgrad || max_exp_avg_sq, PA_ERR_INVALID, "amsgrad needs max_exp_avg_sq");
  pa_mlp_desc d;
  memset(&d, 0, sizeof(d));
  d.lr = lr; d.beta1 = beta1; d.beta2 = beta2; d.eps = eps; d.weight_decay = weight_decay;
  d.amsgrad = amsgrad;
  AlphaArgs a;
  a.log_alpha = log_alpha; a.m = exp_avg; a.v = exp_avg_sq; a.vmax = max_exp_avg_sq;
  a.alpha = alpha_out; a.logp = log_prob; a.B = B; a.target_entropy = target_entropy;
  a.c = adam_scalars(d, step);
  a.loss_out = loss_out;
  hipLaunchKernelGGL(alpha_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_concat_cols(const float* left, int32_t ldl, const float* right, int32_t ldr,
                              float* out, int32_t B, int32_t nl, int32_t nr, void* stream) {
  PA_REQUIRE(left && right && out && B > 0 && nl > 0 && nr > 0, PA_ERR_INVALID,
             "pa_concat_cols: bad argument");
  const int64_t total = (int64_t)B * (nl + nr);
  const unsigned grid = (unsigned)(ceil_div(total, 256) > 2048 ? 2048 : ceil_div(total, 256));
  hipLaunchKernelGGL(concat_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     left, ldl, right, ldr, out, B, nl, nr);
  PA_LAUNCH_CHECK();
  return PA_OK;
}


// Twin networks that share one optimizer configuration (twin critics), both with weight gradients
// deferred by pa_mlp_backward*(want_dw = 2): dW + AdamW of BOTH in one launch per three layers and,
// with soft_tau >= 0, their soft target updates in the same epilogue (update_target_network,
// common/utils.py:214-226).  PA_ERR_UNSUPPORTED when the pair does not qualify (the caller then
// steps them one by one).
extern "C" int pa_mlp_adam2(pa_mlp* a, pa_mlp* b, int64_t step, float soft_tau, void* stream) {
  PA_REQUIRE(a && b && a != b && a->bound && b->bound && step >= 1, PA_ERR_INVALID,
             "pa_mlp_adam2: bad argument");
  PA_REQUIRE(pa::mlp_pair_fusable(a, b, soft_tau >= 0.f), PA_ERR_UNSUPPORTED,
             "pa_mlp_adam2: the networks do not share shape, batch and optimizer configuration, or "
             "have no deferred weight gradients");
  PA_HIP(hipSetDevice(a->d.device));
  return pa::mlp_adam_pair(a, b, step, soft_tau, reinterpret_cast<hipStream_t>(stream), nullptr);
}

// The data-parallel step of such a pair (PPO's actor + critic under torch.distributed): the two
// networks' deferred weight gradients WITHOUT the optimizer in one launch ...
extern "C" int pa_mlp_flush_grads2(pa_mlp* a, pa_mlp* b, void* stream) {
  PA_REQUIRE(a && b && a != b && a->bound && b->bound, PA_ERR_INVALID, "pa_mlp_flush_grads2: bad argument");
  PA_REQUIRE(a->pend.active && b->pend.active && a->pend.B == b->pend.B && a->L == b->L && a->L <= 3,
             PA_ERR_UNSUPPORTED,
             "pa_mlp_flush_grads2: the networks have no deferred weight gradients of one batch");
  PA_HIP(hipSetDevice(a->d.device));
  pa_mlp* hs[2] = {a, b};
  DwOperands ops[2] = {{a->pend.x, a->pend.ldx, a->pend.dzs, a->pend.ldzs},
                       {b->pend.x, b->pend.ldx, b->pend.dzs, b->pend.ldzs}};
  a->pend.active = b->pend.active = false;
  return run_weight_grads_n(hs, ops, 2, a->pend.B, 0, -1.f, reinterpret_cast<hipStream_t>(stream));
}
// ... and, after the caller's all-reduce of the gradient buffers, AdamW(amsgrad) step `step` of
// both in one launch (each with its own optimizer configuration).
extern "C" int pa_mlp_adamw2(pa_mlp* a, pa_mlp* b, int64_t step_a, int64_t step_b, void* stream) {
  PA_REQUIRE(a && b && a != b && a->bound && b->bound && step_a >= 1 && step_b >= 1, PA_ERR_INVALID,
             "pa_mlp_adamw2: bad argument");
  AdamArgs args[2];
  pa_mlp* hs[2] = {a, b};
  const int64_t steps[2] = {step_a, step_b};
  for (int i = 0; i < 2; ++i) {
    pa_mlp* h = hs[i];
    PA_REQUIRE(h->bufs.grad && h->bufs.exp_avg && h->bufs.exp_avg_sq &&
                   (!h->d.amsgrad || h->bufs.max_exp_avg_sq) && !h->pend.active,
               PA_ERR_INVALID, "pa_mlp_adamw2: optimizer buffers not bound, or gradients still deferred");
    memset(&args[i], 0, sizeof(AdamArgs));
    args[i].st.p = h->bufs.p; args[i].st.m = h->bufs.exp_avg; args[i].st.v = h->bufs.exp_avg_sq;
    args[i].st.vmax = h->bufs.max_exp_avg_sq;
    args[i].g = h->bufs.grad;
    args[i].n = h->P;
    args[i].c = adam_scalars(h->d, steps[i]);
    h->packed_ok = false;
  }
  PA_HIP(hipSetDevice(a->d.device));
  const int64_t nmax = a->P > b->P ? a->P : b->P;
  hipLaunchKernelGGL(adamw2_kernel, dim3((unsigned)ceil_div(nmax, 256), 2), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), args[0], args[1]);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// ---- internal entry points for the fused learner steps (sac_step.hip) ---------------------------
namespace pa {
int mlp_ensure_packed(pa_mlp* h, bool target, hipStream_t s) { return ensure_packed(h, target, s); }
int mlp_ensure_um(pa_mlp* h, bool target, hipStream_t s) { return ensure_um(h, target, s); }
void mlp_set_pending(pa_mlp* h, const float* x, int ldx, int B, const float* const* dzs,
                     const int* ldzs) {
  set_pending(h, x, ldx, B, dzs, ldzs);
  h->kept_B = B;
}
bool mlp_pair_fusable(const pa_mlp* a, const pa_mlp* b, bool soft) {
  if (!a->pend.active || !b->pend.active || a->pend.B != b->pend.B) return false;
  if (a->L != b->L || a->L > 3) return false;
  const pa_mlp_desc &x = a->d, &y = b->d;
  if (x.lr != y.lr || x.beta1 != y.beta1 || x.beta2 != y.beta2 || x.eps != y.eps ||
      x.weight_decay != y.weight_decay || x.amsgrad != y.amsgrad)
    return false;
  for (const pa_mlp* h : {a, b}) {
    if (!h->bufs.grad || !h->bufs.exp_avg || !h->bufs.exp_avg_sq) return false;
    if (h->d.amsgrad && !h->bufs.max_exp_avg_sq) return false;
    if (soft && !h->bufs.p_target) return false;
  }
  return true;
}
int mlp_adam_pair(pa_mlp* a, pa_mlp* b, int64_t step, float soft_tau, hipStream_t s,
                  const TailJob* tail) {
  pa_mlp* hs[2] = {a, b};
  DwOperands ops[2] = {{a->pend.x, a->pend.ldx, a->pend.dzs, a->pend.ldzs},
                       {b->pend.x, b->pend.ldx, b->pend.dzs, b->pend.ldzs}};
  a->pend.active = b->pend.active = false;
  return run_weight_grads_n(hs, ops, 2, a->pend.B, step, soft_tau, s, tail);
}
}  // namespace pa
