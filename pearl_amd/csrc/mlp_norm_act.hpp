// mlp_block's other forms in the generic MLP engine (round 5): nn.LayerNorm between a hidden Linear
// and its activation, and the hidden activations beside ReLU
// (pearl/neural_networks/common/utils.py:29-56 ActivationType, :75-152 mlp_block).
//
// These are row-local: one wave owns one batch row.  A network that uses them runs layer by layer —
// linear_kernel (bias epilogue) for z = W x + b, then
//   norm_act_fwd_kernel   n = LayerNorm(z) (optional), h = act(n), in place; keeps the normalised
//                         values and 1 / sqrt(var + eps) per row for the backward
//   norm_param_grad_kernel + norm_param_sum_kernel
//                         d gamma = sum_b dn xhat, d beta = sum_b dn   (dn = dh act'(h)), fixed
//                         summation order: row blocks in order, rows of a block in order
//   norm_act_bwd_kernel   dh (the GEMM dZ_{l+1} W_{l+1}, unmasked) -> dz, in place:
//                         dn = dh act'(h);  g = dn gamma;
//                         dz = rstd (g - mean(g) - xhat mean(g xhat))   (torch's layer_norm backward)
// Round 6 — the remaining options of mlp_block, in the same layer-by-layer path:
//   dropout      a keep mask (0 or 1 / (1 - p), the caller's) multiplies the value between LayerNorm
//                and activation (fwd) and the gradient between act' and the LayerNorm backward (bwd):
//                NormActArgs::drop
//   batch norm   AFTER the activation, training mode: column statistics over the batch at hand in a
//                fixed order (col_partial_kernel: 32 row blocks per column, rows of a block in order;
//                bn_stats_kernel adds the blocks in order), y = (h - mean) rstd gamma + beta in place,
//                the activation output kept beside it; running statistics updated as torch does
//                (momentum 0.1, unbiased variance); backward = torch's batch_norm backward
//   residual     out = in + block(in): add_rows_kernel in the forward, and in the backward the saved
//                d out joins the block's input gradient
// act'() is formed from the kept OUTPUT h (relu / leaky_relu: sign; tanh: 1 - h^2; sigmoid:
// h (1 - h); softplus: sigma(z) = 1 - exp(-h)), so nothing but h, xhat and rstd is kept.
#pragma once
#include "common.hpp"

namespace pa {

enum { ACT_RELU = 0, ACT_LEAKY_RELU = 1, ACT_TANH = 2, ACT_SOFTPLUS = 3, ACT_SIGMOID = 4, ACT_COUNT = 5 };

__device__ __forceinline__ float act_forward(int kind, float z) {
  switch (kind) {
    case ACT_LEAKY_RELU: return z > 0.f ? z : __fmul_rn(0.01f, z);         // nn.LeakyReLU(): slope 0.01
    case ACT_TANH: return tanhf(z);
    case ACT_SOFTPLUS: return z > 20.f ? z : log1pf(expf(z));              // nn.Softplus(): beta 1, threshold 20
    case ACT_SIGMOID: return __fdiv_rn(1.f, __fadd_rn(1.f, expf(-z)));
    default: return (z < 0.f) ? 0.f : z;                                   // relu (NaN kept)
  }
}
// d act / d z from the activation's OUTPUT
__device__ __forceinline__ float act_derivative(int kind, float h) {
  switch (kind) {
    case ACT_LEAKY_RELU: return h > 0.f ? 1.f : 0.01f;
    case ACT_TANH: return __fsub_rn(1.f, __fmul_rn(h, h));
    case ACT_SOFTPLUS: return -expm1f(-h);
    case ACT_SIGMOID: return __fmul_rn(h, __fsub_rn(1.f, h));
    default: return h > 0.f ? 1.f : 0.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct NormActArgs {
  float* z; int ldz;            // [B][d] in: Linear output (fwd) / dh (bwd); out: h (fwd) / dz (bwd), in place
  const float* h; int ldh;      // bwd: the kept activation output
  const float* gamma; const float* beta;   // null: no LayerNorm
  float* xhat; float* rstd;     // fwd: written when non-null; bwd: read
  int B, d, act, identity;      // identity: no activation (FlatMlp.identity_layers)
  float eps;
  const float* drop; int ldd;   // dropout keep mask [B][d] (0 or 1 / (1 - p)) or null
};

constexpr int NA_ROWS = 4;      // rows (waves) per workgroup

static __global__ __launch_bounds__(64 * NA_ROWS) void norm_act_fwd_kernel(NormActArgs a) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * NA_ROWS + (threadIdx.x >> 6);
  if (row >= a.B) return;
  float* z = a.z + (int64_t)row * a.ldz;
  float mean = 0.f, rstd = 1.f;
  if (a.gamma) {
    // two passes over the row (it sits in L1 / L2): mean, then the centred second moment — the
    // biased variance nn.LayerNorm uses
    float s = 0.f;
    for (int c = lane; c < a.d; c += 64) s += z[c];
    mean = wave_sum(s) / (float)a.d;
    float q = 0.f;
    for (int c = lane; c < a.d; c += 64) {
      const float t = z[c] - mean;
      q = fmaf(t, t, q);
    }
    const float var = wave_sum(q) / (float)a.d;
    rstd = 1.f / sqrtf(var + a.eps);
    if (a.rstd && lane == 0) a.rstd[row] = rstd;
  }
  for (int c = lane; c < a.d; c += 64) {
    float v = z[c];
    if (a.gamma) {
      const float xh = __fmul_rn(__fsub_rn(v, mean), rstd);
      if (a.xhat) a.xhat[(int64_t)row * a.d + c] = xh;
      v = __fadd_rn(__fmul_rn(xh, a.gamma[c]), a.beta[c]);
    }
    if (a.drop) v = __fmul_rn(v, a.drop[(int64_t)row * a.ldd + c]);
    z[c] = a.identity ? v : act_forward(a.act, v);
  }
}

static __global__ __launch_bounds__(64 * NA_ROWS) void norm_act_bwd_kernel(NormActArgs a) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * NA_ROWS + (threadIdx.x >> 6);
  if (row >= a.B) return;
  float* g = a.z + (int64_t)row * a.ldz;
  const float* h = a.h + (int64_t)row * a.ldh;
  const float* xh = a.xhat ? a.xhat + (int64_t)row * a.d : nullptr;
  const float* dm = a.drop ? a.drop + (int64_t)row * a.ldd : nullptr;
  if (!a.gamma) {
    for (int c = lane; c < a.d; c += 64) {
      float dn = a.identity ? g[c] : __fmul_rn(g[c], act_derivative(a.act, h[c]));
      if (dm) dn = __fmul_rn(dn, dm[c]);
      g[c] = dn;
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < a.d; c += 64) {
    float dn = a.identity ? g[c] : __fmul_rn(g[c], act_derivative(a.act, h[c]));
    if (dm) dn = __fmul_rn(dn, dm[c]);
    const float gg = __fmul_rn(dn, a.gamma[c]);
    s1 += gg;
    s2 = fmaf(gg, xh[c], s2);
  }
  const float m1 = wave_sum(s1) / (float)a.d, m2 = wave_sum(s2) / (float)a.d;
  const float rstd = a.rstd[row];
  for (int c = lane; c < a.d; c += 64) {
    float dn = a.identity ? g[c] : __fmul_rn(g[c], act_derivative(a.act, h[c]));
    if (dm) dn = __fmul_rn(dn, dm[c]);
    const float gg = __fmul_rn(dn, a.gamma[c]);
    g[c] = __fmul_rn(rstd, __fsub_rn(__fsub_rn(gg, m1), __fmul_rn(xh[c], m2)));
  }
}

// d gamma / d beta: column sums over the batch.  Stage 1: block (cx, ry) sums rows
// [ry * rows_per, (ry + 1) * rows_per) of columns [64 cx, 64 cx + 64) in row order -> part[ry][2][d];
// stage 2 adds the row blocks in order.  Must run BEFORE norm_act_bwd_kernel overwrites dh.
constexpr int NP_BLOCKS = 32;
static __global__ __launch_bounds__(64) void norm_param_grad_kernel(NormActArgs a, float* part, int rows_per) {
  const int c = blockIdx.x * 64 + threadIdx.x, ry = blockIdx.y;
  if (c >= a.d) return;
  const int r0 = ry * rows_per, r1 = (r0 + rows_per < a.B) ? r0 + rows_per : a.B;
  float sg = 0.f, sb = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float dh = a.z[(int64_t)r * a.ldz + c];
    float dn = a.identity ? dh : __fmul_rn(dh, act_derivative(a.act, a.h[(int64_t)r * a.ldh + c]));
    if (a.drop) dn = __fmul_rn(dn, a.drop[(int64_t)r * a.ldd + c]);
    sg = fmaf(dn, a.xhat[(int64_t)r * a.d + c], sg);
    sb += dn;
  }
  part[((int64_t)ry * 2 + 0) * a.d + c] = sg;
  part[((int64_t)ry * 2 + 1) * a.d + c] = sb;
}
static __global__ __launch_bounds__(64) void norm_param_sum_kernel(const float* part, int nblocks, int d,
                                                                   float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= d) return;
  float sg = 0.f, sb = 0.f;
  for (int k = 0; k < nblocks; ++k) {
    sg += part[((int64_t)k * 2 + 0) * d + c];
    sb += part[((int64_t)k * 2 + 1) * d + c];
  }
  dgamma[c] = sg;
  dbeta[c] = sb;
}

// ---- BatchNorm1d in training mode (after the activation) -----------------------------------------
struct BnArgs {
  float* y; int ldy;              // [B][d]: in the activation output h, out the normalised y (in place)
  float* hpre;                    // [B][d] kept copy of h (null: forward not kept)
  const float* gamma; const float* beta;
  float* mean; float* rstd;       // [d] batch statistics (scratch when the forward is not kept)
  float* run_mean; float* run_var; long long* nbt;   // running statistics (any may be null)
  float* part;                    // [NP_BLOCKS][2][d] partial sums
  int B, d;
  float eps, momentum;
};
// stage 0: part[ry][0][c] = sum over the block's rows of x;  stage 1: sum of (x - mean[c])^2
static __global__ __launch_bounds__(64) void col_partial_kernel(const float* x, int ld, int B, int d,
                                                                const float* mean, float* part,
                                                                int rows_per, int stage) {
  const int c = blockIdx.x * 64 + threadIdx.x, ry = blockIdx.y;
  if (c >= d) return;
  const int r0 = ry * rows_per, r1 = (r0 + rows_per < B) ? r0 + rows_per : B;
  float s = 0.f;
  if (stage == 0) {
    for (int r = r0; r < r1; ++r) s += x[(int64_t)r * ld + c];
  } else {
    const float m = mean[c];
    for (int r = r0; r < r1; ++r) {
      const float t = x[(int64_t)r * ld + c] - m;
      s = fmaf(t, t, s);
    }
  }
  part[((int64_t)ry * 2 + 0) * d + c] = s;
}
// stage 0: mean = sum / B;  stage 1: var = sum / B (biased, what normalises), rstd, running statistics
static __global__ __launch_bounds__(64) void bn_stats_kernel(BnArgs a, int nblocks, int stage) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.d) return;
  float s = 0.f;
  for (int k = 0; k < nblocks; ++k) s += a.part[((int64_t)k * 2 + 0) * a.d + c];
  if (stage == 0) {
    a.mean[c] = s / (float)a.B;
    return;
  }
  const float var = s / (float)a.B;
  a.rstd[c] = 1.f / sqrtf(var + a.eps);
  if (a.run_mean) a.run_mean[c] = __fadd_rn(__fmul_rn(1.f - a.momentum, a.run_mean[c]), __fmul_rn(a.momentum, a.mean[c]));
  if (a.run_var) {
    const float unbiased = a.B > 1 ? s / (float)(a.B - 1) : var;
    a.run_var[c] = __fadd_rn(__fmul_rn(1.f - a.momentum, a.run_var[c]), __fmul_rn(a.momentum, unbiased));
  }
  if (a.nbt && c == 0) a.nbt[0] += 1;
}
static __global__ __launch_bounds__(256) void bn_apply_fwd_kernel(BnArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.B * a.d) return;
  const int r = (int)(e / a.d), c = (int)(e - (int64_t)r * a.d);
  float* p = a.y + (int64_t)r * a.ldy + c;
  const float h = *p;
  if (a.hpre) a.hpre[e] = h;
  const float xh = __fmul_rn(__fsub_rn(h, a.mean[c]), a.rstd[c]);
  *p = __fadd_rn(__fmul_rn(xh, a.gamma[c]), a.beta[c]);
}
// backward: dy (in a.y, in place -> dh), hpre, mean, rstd, gamma;  part[ry][0] = sum dy xhat, [1] = sum dy
static __global__ __launch_bounds__(64) void bn_param_grad_kernel(BnArgs a, int rows_per) {
  const int c = blockIdx.x * 64 + threadIdx.x, ry = blockIdx.y;
  if (c >= a.d) return;
  const int r0 = ry * rows_per, r1 = (r0 + rows_per < a.B) ? r0 + rows_per : a.B;
  const float m = a.mean[c], rs = a.rstd[c];
  float sg = 0.f, sb = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float dy = a.y[(int64_t)r * a.ldy + c];
    const float xh = __fmul_rn(__fsub_rn(a.hpre[(int64_t)r * a.d + c], m), rs);
    sg = fmaf(dy, xh, sg);
    sb += dy;
  }
  a.part[((int64_t)ry * 2 + 0) * a.d + c] = sg;
  a.part[((int64_t)ry * 2 + 1) * a.d + c] = sb;
}
// dh = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)); dgamma / dbeta (the column sums) given
static __global__ __launch_bounds__(256) void bn_apply_bwd_kernel(BnArgs a, const float* dgamma,
                                                                  const float* dbeta) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.B * a.d) return;
  const int r = (int)(e / a.d), c = (int)(e - (int64_t)r * a.d);
  float* p = a.y + (int64_t)r * a.ldy + c;
  const float xh = __fmul_rn(__fsub_rn(a.hpre[e], a.mean[c]), a.rstd[c]);
  const float invB = 1.f / (float)a.B;
  const float t = __fsub_rn(__fsub_rn(*p, __fmul_rn(dbeta[c], invB)), __fmul_rn(xh, __fmul_rn(dgamma[c], invB)));
  *p = __fmul_rn(__fmul_rn(a.gamma[c], a.rstd[c]), t);
}
// dst[r][:d] (+)= src[r][:d]
static __global__ __launch_bounds__(256) void add_rows_kernel(float* dst, int ldd, const float* src, int lds,
                                                              int B, int d, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * d) return;
  const int r = (int)(e / d), c = (int)(e - (int64_t)r * d);
  float* p = dst + (int64_t)r * ldd + c;
  const float v = src[(int64_t)r * lds + c];
  *p = accumulate ? __fadd_rn(*p, v) : v;
}

}  // namespace pa
