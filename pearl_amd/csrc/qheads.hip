// Row-local heads of the GENERIC TD learner: the pieces around the pa_mlp engine that
// DeepQLearning / DoubleDQN / DeepSARSA need for Q-network architectures the fused DQN kernels do
// not cover — VanillaQValueNetwork of any depth / width, VanillaQValueMultiHeadNetwork
// (q_value_networks.py:185-249) and DuelingQValueNetwork (:352-508).  All of them are HBM-bound
// element-wise passes over (B, A)-shaped values: one thread per output element, per-row sums in
// index order (the order of a serial loop, independent of the launch shape).
//
//   pa_td_target        get_next_state_values + Bellman target (deep_q_learning.py:130-167,
//                       double_dqn.py:29-57, deep_td_learning.py:313-317)
//   pa_td_head          MSELoss(mean) gradient + reported mean |Q - target| (deep_td_learning.py:319-359)
//   pa_rows_dot         multi-head: Q(s, a) = onehot(a) . f(s)  (torch.bmm, q_value_networks.py:232-238)
//   pa_rows_scale       its gradient: d f[b, :] = dq[b] onehot(a_b)
//   pa_rows_bmm         multi-head over an action set: q[b, i] = rep[b, i, :] . f(s_b)
//   pa_dueling_q        Q = V + A - mean(A)  (q_value_networks.py:474-506)
//   pa_dueling_grad     gradient of that for the taken-action rows + the available-action rows
//   pa_dueling_feat_grad  sum of the advantage tower's input gradients over the rows of a state
//   pa_dueling_cql_grad, pa_rows_bmm_t   the CQL term's all-actions table on dueling / multi-head nets
#include <math.h>

#include "common.hpp"

using namespace pa;

namespace {

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

struct TdTargetArgs {
  const float* q_val; int ldv;       // [B, A] values that are returned (target network)
  const float* q_sel; int lds_;      // [B, A] values the action is chosen with (online net) or null
  const uint8_t* mask; int ldm;      // [B, A] 1 = unavailable, or null
  const float* reward; const uint8_t* term;
  float gamma;
  int B, A;
  float* next_v; float* y;           // either may be null
};
__global__ __launch_bounds__(256) void td_target_kernel(TdTargetArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const float* sel = (a.q_sel ? a.q_sel + (int64_t)b * a.lds_ : a.q_val + (int64_t)b * a.ldv);
  const uint8_t* mk = a.mask ? a.mask + (int64_t)b * a.ldm : nullptr;
  // torch.max(1): first maximum, the first NaN wins; masked entries are -inf
  float m = (mk && mk[0]) ? -INFINITY : sel[0];
  int mi = 0;
  for (int i = 1; i < a.A; ++i) {
    const float x = (mk && mk[i]) ? -INFINITY : sel[i];
    const bool take = (x > m || x != x) && !(m != m);
    m = take ? x : m;
    mi = take ? i : mi;
  }
  // DoubleDQN values the chosen action with the OTHER network, unmasked (double_dqn.py:49-56)
  const float v = a.q_sel ? a.q_val[(int64_t)b * a.ldv + mi] : m;
  if (a.next_v) a.next_v[b] = v;
  if (a.y) {
    const float live = 1.0f - (a.term[b] ? 1.0f : 0.0f);
    a.y[b] = __fadd_rn(__fmul_rn(__fmul_rn(v, a.gamma), live), a.reward[b]);
  }
}

// Bellman target from ONE value per row (DeepSARSA, deep_sarsa.py:59-97)
__global__ __launch_bounds__(256) void td_target1_kernel(const float* __restrict__ v, int ldv,
                                                         const float* __restrict__ reward,
                                                         const uint8_t* __restrict__ term, float gamma,
                                                         int B, float* __restrict__ y) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float live = 1.0f - (term[b] ? 1.0f : 0.0f);
  y[b] = __fadd_rn(__fmul_rn(__fmul_rn(v[(int64_t)b * ldv], gamma), live), reward[b]);
}

__global__ __launch_bounds__(256) void td_head_kernel(const float* __restrict__ q, int ldq,
                                                      const float* __restrict__ y, int B, float norm,
                                                      float* __restrict__ dq,
                                                      float* __restrict__ loss_out) {
  __shared__ float red[256];
  float pa_ = 0.f, ps = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float d = __fsub_rn(q[(int64_t)b * ldq], y[b]);
    dq[b] = __fmul_rn(norm, d);
    pa_ += fabsf(d);
    ps += d * d;
  }
  const float sa = block_sum_256(pa_, red);
  const float ss = block_sum_256(ps, red);
  if (threadIdx.x == 0) {
    loss_out[0] = sa / (float)B;   // mean |Q - target| (the reported "loss")
    loss_out[1] = ss / (float)B;   // MSE
  }
}

__global__ __launch_bounds__(256) void rows_dot_kernel(const float* __restrict__ f, int ldf,
                                                       const float* __restrict__ rep, int ldr, int B,
                                                       int A, float* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int j = 0; j < A; ++j) s = fmaf(rep[(int64_t)b * ldr + j], f[(int64_t)b * ldf + j], s);
  out[b] = s;
}

__global__ __launch_bounds__(256) void rows_scale_kernel(const float* __restrict__ dq,
                                                         const float* __restrict__ rep, int ldr,
                                                         int B, int A, float* __restrict__ out,
                                                         int ldo) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * A) return;
  const int b = (int)(e / A), j = (int)(e - (int64_t)b * A);
  out[(int64_t)b * ldo + j] = dq[b] * rep[(int64_t)b * ldr + j];
}

__global__ __launch_bounds__(256) void rows_bmm_kernel(const float* __restrict__ rep,
                                                       int64_t rep_bstride, const float* __restrict__ f,
                                                       int ldf, int B, int Q, int A,
                                                       float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * Q) return;
  const int b = (int)(e / Q), i = (int)(e - (int64_t)b * Q);
  const float* r = rep + (int64_t)b * rep_bstride + (int64_t)i * A;
  float s = 0.f;
  for (int j = 0; j < A; ++j) s = fmaf(r[j], f[(int64_t)b * ldf + j], s);
  out[e] = s;
}

// q[b, i] = (v[b] + adv_q[b, i]) - mean_j adv_m[b, j]   (adv_m = adv_q when there is no separate
// set of available actions: the reference then averages over the query actions themselves)
__global__ __launch_bounds__(256) void dueling_q_kernel(const float* __restrict__ v,
                                                        const float* __restrict__ adv_q, int Q,
                                                        const float* __restrict__ adv_m, int M, int B,
                                                        float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * Q) return;
  const int b = (int)(e / Q);
  const float* am = adv_m ? adv_m + (int64_t)b * M : adv_q + (int64_t)b * Q;
  const int n = adv_m ? M : Q;
  float s = 0.f;
  for (int j = 0; j < n; ++j) s += am[j];
  const float mean = s / (float)n;
  out[e] = __fsub_rn(__fadd_rn(v[b], adv_q[e]), mean);
}

// Taken-action forward Q = V + A(s, a) - mean_i A(s, avail_i): rows [0, B) of the advantage pass are
// the taken actions, rows B + b M + i the available ones.  d V = dq, d A_taken = dq,
// d A_avail[b, i] = -dq[b] / M.  Without available actions (M = 0) the mean is A_taken itself and
// the advantage tower gets no gradient.
__global__ __launch_bounds__(256) void dueling_grad_kernel(const float* __restrict__ dq, int B, int M,
                                                           float* __restrict__ d_adv) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * (1 + M);
  if (e >= total) return;
  if (e < B) {
    d_adv[e] = (M > 0) ? dq[e] : 0.f;
  } else {
    const int b = (int)((e - B) / M);
    d_adv[e] = -(dq[b] / (float)M);
  }
}

// The CQL term on a dueling network (loss_fn_utils.py:52-58 calls get_q_values(state, curr_available
// actions) with no separate available set: Q_all[b, i] = V + A_avail[b, i] - mean_k A_avail[b, k]) next
// to the MSE term's taken-action forward: both read the SAME advantage rows, so one kept pass serves
// both and the gradients add —
//   d V[b]          = dq[b] + sum_i dqa[b, i]
//   d A_taken[b]    = dq[b]
//   d A_avail[b, i] = -dq[b] / M + dqa[b, i] - (sum_k dqa[b, k]) / M
__global__ __launch_bounds__(256) void dueling_cql_grad_kernel(const float* __restrict__ dq,
                                                               const float* __restrict__ dqa, int B,
                                                               int M, float* __restrict__ d_adv,
                                                               float* __restrict__ d_v) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * (1 + M);
  if (e >= total) return;
  const int b = e < B ? (int)e : (int)((e - B) / M);
  float s = 0.f;
  for (int k = 0; k < M; ++k) s += dqa[(int64_t)b * M + k];
  if (e < B) {
    d_adv[e] = dq[e];
    d_v[e] = dq[e] + s;
  } else {
    d_adv[e] = (dqa[e - B] - s / (float)M) - dq[b] / (float)M;
  }
}

// The transpose of rows_bmm for the backward of a multi-head table: df[b, j] (+)= sum_i dqa[b, i] rep[b, i, j]
__global__ __launch_bounds__(256) void rows_bmm_t_kernel(const float* __restrict__ dqa,
                                                         const float* __restrict__ rep,
                                                         int64_t rep_bstride, int B, int Q, int A,
                                                         int accumulate, float* __restrict__ df, int ldf) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * A) return;
  const int b = (int)(e / A), j = (int)(e - (int64_t)b * A);
  const float* r = rep + (int64_t)b * rep_bstride + j;
  float s = 0.f;
  for (int i = 0; i < Q; ++i) s = fmaf(dqa[(int64_t)b * Q + i], r[(int64_t)i * A], s);
  float* dst = df + (int64_t)b * ldf + j;
  *dst = accumulate ? (*dst + s) : s;
}

// dfeat[b, :] (+)= dX[b, :H] + sum_i dX[B + b M + i, :H]   (in row order)
__global__ __launch_bounds__(256) void dueling_feat_grad_kernel(const float* __restrict__ dX, int ldx,
                                                                int B, int M, int H, int accumulate,
                                                                float* __restrict__ dfeat, int ldf) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * H) return;
  const int b = (int)(e / H), k = (int)(e - (int64_t)b * H);
  float s = dX[(int64_t)b * ldx + k];
  for (int i = 0; i < M; ++i) s += dX[((int64_t)B + (int64_t)b * M + i) * ldx + k];
  float* dst = dfeat + (int64_t)b * ldf + k;
  *dst = accumulate ? (*dst + s) : s;
}

// SquareCBExploration.act's probability table (squarecb_exploration.py:59-115), one row per
// context:  gaps = max_a v - v;  p_a = 1 / (A + gamma gaps_a);  the arg-max entry is rewritten to
// 1 - (sum of the row's other entries).  The reference takes that complementary sum over the WHOLE
// (B, A) matrix (:90), which is this rule for B = 1 and an invalid distribution (negative mass,
// Categorical raises) for B > 1; rows are normalised one by one here, so batches of contexts work.
__global__ __launch_bounds__(256) void squarecb_kernel(const float* __restrict__ values, int ldv,
                                                       int B, int A, float gamma, int clamp_values,
                                                       float lb, float ub, float* __restrict__ prob,
                                                       int* __restrict__ argmax_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* v = values + (int64_t)b * ldv;
  float* p = prob + (int64_t)b * A;
  float m = 0.f;
  int mi = 0;
  for (int a = 0; a < A; ++a) {
    float x = v[a];
    if (clamp_values) x = fminf(fmaxf(x, lb), ub);
    const bool take = (a == 0) || ((x > m || x != x) && !(m != m));   // torch.max: first maximum
    m = take ? x : m;
    mi = take ? a : mi;
  }
  float comp = 0.f;
  for (int a = 0; a < A; ++a) {
    float x = v[a];
    if (clamp_values) x = fminf(fmaxf(x, lb), ub);
    const float pa_ = (a == mi) ? 0.f : __fdiv_rn(1.0f, __fadd_rn((float)A, __fmul_rn(gamma, __fsub_rn(m, x))));
    p[a] = pa_;
    comp += pa_;
  }
  p[mi] = 1.0f - comp;
  argmax_out[b] = mi;
}

// DoubleDQN's action choice (double_dqn.py:40-51): first argmax over the unmasked entries of the
// ONLINE network's values, and the representation of that action out of the row's action table.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ q, int ldq,
                                                          const uint8_t* __restrict__ mask, int ldm,
                                                          const float* __restrict__ rep,
                                                          int64_t rep_bstride, int B, int A, int AD,
                                                          int* __restrict__ idx_out,
                                                          float* __restrict__ rep_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* sel = q + (int64_t)b * ldq;
  const uint8_t* mk = mask ? mask + (int64_t)b * ldm : nullptr;
  float m = (mk && mk[0]) ? -INFINITY : sel[0];
  int mi = 0;
  for (int i = 1; i < A; ++i) {
    const float x = (mk && mk[i]) ? -INFINITY : sel[i];
    const bool take = (x > m || x != x) && !(m != m);
    m = take ? x : m;
    mi = take ? i : mi;
  }
  if (idx_out) idx_out[b] = mi;
  if (rep_out) {
    const float* src = rep + (int64_t)b * rep_bstride + (int64_t)mi * AD;
    for (int j = 0; j < AD; ++j) rep_out[(int64_t)b * AD + j] = src[j];
  }
}

unsigned grid_for(int64_t n) { return (unsigned)ceil_div(n, 256); }

}  // namespace

extern "C" int pa_td_target(const float* q_val, int32_t ldv, const float* q_sel, int32_t lds_,
                            const uint8_t* mask, int32_t ldm, const float* reward,
                            const uint8_t* terminated, float gamma, int32_t B, int32_t A,
                            float* next_v, float* y, void* stream) {
  PA_REQUIRE(q_val && B > 0 && A > 0 && (next_v || y), PA_ERR_INVALID, "pa_td_target: bad argument");
  PA_REQUIRE(!y || (reward && terminated), PA_ERR_INVALID,
             "pa_td_target: the Bellman target needs reward and terminated");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (A == 1 && !q_sel && !mask && !next_v) {
    hipLaunchKernelGGL(td_target1_kernel, dim3(grid_for(B)), dim3(256), 0, s, q_val, ldv, reward,
                       terminated, gamma, B, y);
    PA_LAUNCH_CHECK();
    return PA_OK;
  }
  TdTargetArgs a;
  a.q_val = q_val; a.ldv = ldv; a.q_sel = q_sel; a.lds_ = lds_; a.mask = mask; a.ldm = ldm;
  a.reward = reward; a.term = terminated; a.gamma = gamma; a.B = B; a.A = A;
  a.next_v = next_v; a.y = y;
  hipLaunchKernelGGL(td_target_kernel, dim3(grid_for(B)), dim3(256), 0, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_td_head(const float* q, int32_t ldq, const float* y, int32_t B, float grad_scale,
                          float* dq, float* loss_out2, void* stream) {
  PA_REQUIRE(q && y && dq && loss_out2 && B > 0, PA_ERR_INVALID, "pa_td_head: bad argument");
  hipLaunchKernelGGL(td_head_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q,
                     ldq, y, B, grad_scale, dq, loss_out2);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_rows_dot(const float* f, int32_t ldf, const float* rep, int32_t ldr, int32_t B,
                           int32_t A, float* out, void* stream) {
  PA_REQUIRE(f && rep && out && B > 0 && A > 0, PA_ERR_INVALID, "pa_rows_dot: bad argument");
  hipLaunchKernelGGL(rows_dot_kernel, dim3(grid_for(B)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), f, ldf, rep, ldr, B, A, out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_rows_scale(const float* dq, const float* rep, int32_t ldr, int32_t B, int32_t A,
                             float* out, int32_t ldo, void* stream) {
  PA_REQUIRE(dq && rep && out && B > 0 && A > 0, PA_ERR_INVALID, "pa_rows_scale: bad argument");
  hipLaunchKernelGGL(rows_scale_kernel, dim3(grid_for((int64_t)B * A)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dq, rep, ldr, B, A, out, ldo);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_rows_bmm(const float* rep, int64_t rep_bstride, const float* f, int32_t ldf,
                           int32_t B, int32_t Q, int32_t A, float* out, void* stream) {
  PA_REQUIRE(rep && f && out && B > 0 && Q > 0 && A > 0, PA_ERR_INVALID, "pa_rows_bmm: bad argument");
  hipLaunchKernelGGL(rows_bmm_kernel, dim3(grid_for((int64_t)B * Q)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), rep, rep_bstride, f, ldf, B, Q, A, out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_dueling_q(const float* v, const float* adv_q, int32_t Q, const float* adv_mean,
                            int32_t M, int32_t B, float* out, void* stream) {
  PA_REQUIRE(v && adv_q && out && B > 0 && Q > 0 && (!adv_mean || M > 0), PA_ERR_INVALID,
             "pa_dueling_q: bad argument");
  hipLaunchKernelGGL(dueling_q_kernel, dim3(grid_for((int64_t)B * Q)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), v, adv_q, Q, adv_mean, M, B, out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_dueling_grad(const float* dq, int32_t B, int32_t M, float* d_adv_rows,
                               void* stream) {
  PA_REQUIRE(dq && d_adv_rows && B > 0 && M >= 0, PA_ERR_INVALID, "pa_dueling_grad: bad argument");
  hipLaunchKernelGGL(dueling_grad_kernel, dim3(grid_for((int64_t)B * (1 + M))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dq, B, M, d_adv_rows);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_dueling_cql_grad(const float* dq, const float* dq_all, int32_t B, int32_t M,
                                   float* d_adv_rows, float* d_value, void* stream) {
  PA_REQUIRE(dq && dq_all && d_adv_rows && d_value && B > 0 && M > 0, PA_ERR_INVALID,
             "pa_dueling_cql_grad: bad argument");
  hipLaunchKernelGGL(dueling_cql_grad_kernel, dim3(grid_for((int64_t)B * (1 + M))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dq, dq_all, B, M, d_adv_rows, d_value);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_rows_bmm_t(const float* dq_all, const float* rep, int64_t rep_bstride, int32_t B,
                             int32_t Q, int32_t A, int32_t accumulate, float* df, int32_t ldf,
                             void* stream) {
  PA_REQUIRE(dq_all && rep && df && B > 0 && Q > 0 && A > 0, PA_ERR_INVALID, "pa_rows_bmm_t: bad argument");
  hipLaunchKernelGGL(rows_bmm_t_kernel, dim3(grid_for((int64_t)B * A)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dq_all, rep, rep_bstride, B, Q, A, accumulate,
                     df, ldf);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_dueling_feat_grad(const float* dX, int32_t ldx, int32_t B, int32_t M, int32_t H,
                                    int32_t accumulate, float* dfeat, int32_t ldf, void* stream) {
  PA_REQUIRE(dX && dfeat && B > 0 && M >= 0 && H > 0, PA_ERR_INVALID,
             "pa_dueling_feat_grad: bad argument");
  hipLaunchKernelGGL(dueling_feat_grad_kernel, dim3(grid_for((int64_t)B * H)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dX, ldx, B, M, H, accumulate, dfeat, ldf);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_squarecb_probs(const float* values, int32_t ldv, int32_t B, int32_t A, float gamma,
                                 int32_t clamp_values, float reward_lb, float reward_ub, float* prob,
                                 int32_t* argmax_out, void* stream) {
  PA_REQUIRE(values && prob && argmax_out && B > 0 && A > 0, PA_ERR_INVALID,
             "pa_squarecb_probs: bad argument");
  hipLaunchKernelGGL(squarecb_kernel, dim3(grid_for(B)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream),
                     values, ldv, B, A, gamma, clamp_values, reward_lb, reward_ub, prob, argmax_out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_argmax_rows(const float* q, int32_t ldq, const uint8_t* mask, int32_t ldm,
                              const float* rep, int64_t rep_bstride, int32_t B, int32_t A, int32_t AD,
                              int32_t* idx_out, float* rep_out, void* stream) {
  PA_REQUIRE(q && B > 0 && A > 0 && (idx_out || rep_out) && (!rep_out || (rep && AD > 0)),
             PA_ERR_INVALID, "pa_argmax_rows: bad argument");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(grid_for(B)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), q, ldq, mask, ldm, rep, rep_bstride, B, A,
                     AD, idx_out, rep_out);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
