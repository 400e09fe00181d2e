// Row-pass kernels of the generic MLP engine (pa_mlp_*): a whole network — up to
// ROW_MAX_LAYERS Linear(+ReLU) layers of at most 256 output units each — evaluated (or
// back-propagated down to the pre-activation gradients) for 16 batch rows per workgroup in ONE
// launch, the activations of a row tile never leaving LDS, the weights streamed as MFMA
// fragment-major copies from L2.  This is what took DQN's online chain from five launches of
// ~8 us each to one (online_rowpass_kernel); here it is written for any layer list so that every
// learner on the engine (SAC, TD3, DDPG, PPO, IQL, discrete SAC, the bandit, the generic Q
// networks) gets it:  forward 3-4 launches -> 1, input-gradient chain 2-3 launches -> 1, and two
// networks (twin critics; PPO's actor + critic) still share a launch (blockIdx.y).
//
// Reference arithmetic: mlp_block (pearl/neural_networks/common/utils.py:75-152) and its autograd.
// Tiles are the transposed 16x16x4 tiles of online_kernels.hpp: weights = MFMA A operand (unit on
// the lane's low 4 bits), activations = B operand (batch row on the lane's low 4 bits), so a lane
// owns one row and four consecutive units per tile; K runs in groups of 16 (one float4 per operand
// feeds four MFMAs).
#pragma once
#include "online_kernels.hpp"

namespace pa {

constexpr int ROW_MAX_LAYERS = 4;
constexpr int ROW_MAX_OUT = 256;    // units per layer one workgroup covers (8 waves x 2 tiles x 16)
constexpr int ROW_MAX_IN = 512;     // width of the network input (staged once in LDS)

struct RowNetFwd {
  const float* Wf[ROW_MAX_LAYERS];     // fragment-major W_l [d_{l+1} units][d_l]
  const void* Wsp[ROW_MAX_LAYERS];     // the same as bf16x3 split planes (wsp16_index) or null
  const float* bias[ROW_MAX_LAYERS];   // null: no bias (bias-free last layer)
  float* act[ROW_MAX_LAYERS];          // kept hidden outputs [B][d_{l+1}] (null: not kept)
  float* out; int ldo;                 // [B][d_L]
  int dims[ROW_MAX_LAYERS + 1];
  int L;
  int relu;                            // bit l: ReLU after layer l
};
struct RowFwdArgs {
  RowNetFwd net[2];
  const float* x; int ldx;
  int B;
};

__host__ __device__ inline int row_hid_pitch() { return rp_pad(ROW_MAX_OUT); }
inline size_t rowfwd_smem_bytes(int d0) {
  return sizeof(float) * (size_t)RP_ROWS * (rp_pad(d0) + 2 * row_hid_pitch());
}

// One workgroup: 16 rows through every layer.  LDS: the input tile and two hidden tiles used in
// turn.  One barrier per layer.
static __global__ __launch_bounds__(512) void mlp_rowfwd_kernel(RowFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowNetFwd& n = a.net[blockIdx.y];
  const int P0 = rp_pad(n.dims[0]), PH = row_hid_pitch();
  float* xs = smem;
  float* hb[2] = {xs + RP_ROWS * P0, xs + RP_ROWS * P0 + RP_ROWS * PH};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  const int m0 = blockIdx.x * RP_ROWS;
  const int row = m0 + r16;
  const bool rok = row < a.B;
  const int u0 = wave * 32 + 4 * qd;
  const int tile0 = wave * 2;
  {
    const bool vx = is_vec_ok(a.x, a.ldx) && ((n.dims[0] & 3) == 0);
    const int c4 = (P0 - 4) >> 2;
    for (int e = tid; e < RP_ROWS * c4; e += 512) {
      const int r = e / c4, c = (e - r * c4) * 4;
      const bool ok = (m0 + r) < a.B;
      float4 v;
      if (vx) v = ld4_or_zero(a.x, (int64_t)(m0 + r) * a.ldx + c, ok && c < n.dims[0]);
      else v = guarded_load4(a.x, (int64_t)(m0 + r) * a.ldx, ok, c, n.dims[0]);
      lds_st4(xs + r * P0 + c, v);
    }
  }
  const float* in = xs;
  int pin = P0;
  for (int l = 0; l < n.L; ++l) {
    const int K = n.dims[l], N = n.dims[l + 1];
    const int nt = (N + 15) >> 4;
    f32x4v acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n.bias[l]) b = guarded_load4(n.bias[l], 0, true, u0 + 16 * t, N);
      acc[t][0] = b.x; acc[t][1] = b.y; acc[t][2] = b.z; acc[t][3] = b.w;
    }
    __syncthreads();   // the input tile of this layer is complete (x staging / previous epilogue)
    vm_drain();         // (kept-activation stores of the layer before: see vm_drain)
    // (a wave whose unit tiles lie beyond the layer's width has nothing to add to its bias)
    if (tile0 < nt) rows16_gemm<4>(acc, n.Wf[l], wf16_nkg(K), tile0, nt, in + r16 * pin + 4 * qd, lane);
    const bool last = l == n.L - 1;
    const bool relu = (n.relu >> l) & 1;
    float* nxt = hb[l & 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u = u0 + 16 * t;
      float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      if (relu) v = make_float4(relu_keep_nan(v.x), relu_keep_nan(v.y), relu_keep_nan(v.z),
                                relu_keep_nan(v.w));
      if (!last) {
        // units beyond N are exact zeros (zero weights, zero bias): the next layer's padded k groups
        if (u < PH - 4) lds_st4(nxt + r16 * PH + u, v);
        if (rok && n.act[l]) store4_guarded(n.act[l], (int64_t)row * N, u, N, (N & 3) == 0, v);
      } else if (rok) {
        store4_guarded(n.out, (int64_t)row * n.ldo, u, N,
                       is_vec_ok(n.out, n.ldo) && (N & 3) == 0, v);
      }
    }
    in = nxt;
    pin = PH;
  }
}

// ---- backward: pre-activation gradients of every layer (+ the input gradient) ------------------
struct RowNetBwd {
  const float* Wtf[ROW_MAX_LAYERS];    // fragment-major W_l^T [d_l units][d_{l+1}]  (l >= 1; l = 0 for d_x)
  const void* Wtsp[ROW_MAX_LAYERS];    // the same as bf16x3 split planes (wsp16_index) or null
  const float* act[ROW_MAX_LAYERS];    // kept hidden outputs of the forward (ReLU masks)
  float* dz[ROW_MAX_LAYERS];           // dz[l] = gradient w.r.t. the pre-activation of layer l - 1,
                                       // [B][d_l], l = 1 .. L - 1  (weight-gradient operands)
  const float* d_out; int ldd;         // [B][d_L]
  float* d_x; int lddx;                // [B][d_0] or null
  int dims[ROW_MAX_LAYERS + 1];
  int L;
  int relu;
};
struct RowBwdArgs {
  RowNetBwd net[2];
  int B;
};
inline size_t rowbwd_smem_bytes() { return sizeof(float) * (size_t)RP_ROWS * 2 * row_hid_pitch(); }

static __global__ __launch_bounds__(512) void mlp_rowbwd_kernel(RowBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowNetBwd& n = a.net[blockIdx.y];
  const int PH = row_hid_pitch();
  float* hb[2] = {smem, smem + RP_ROWS * PH};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  const int m0 = blockIdx.x * RP_ROWS;
  const int row = m0 + r16;
  const bool rok = row < a.B;
  const int u0 = wave * 32 + 4 * qd;
  const int tile0 = wave * 2;
  {  // d_out tile -> LDS, zero padded to whole k groups
    const int DL = n.dims[n.L];
    const int c4 = (rp_pad(DL) - 4) >> 2;
    for (int e = tid; e < RP_ROWS * c4; e += 512) {
      const int r = e / c4, c = (e - r * c4) * 4;
      const float4 v = guarded_load4(n.d_out, (int64_t)(m0 + r) * n.ldd, (m0 + r) < a.B, c, DL);
      lds_st4(hb[0] + r * PH + c, v);
    }
  }
  const float* in = hb[0];
  int cur = 0;
  for (int l = n.L - 1; l >= 0; --l) {
    if (l == 0 && !n.d_x) break;
    const int K = n.dims[l + 1], N = n.dims[l];   // dIn[row][unit of d_l] = sum_k dZ[row][k] W_l[k][unit]
    const int nt = (N + 15) >> 4;
    const bool mask = l > 0 && ((n.relu >> (l - 1)) & 1);
    float* nxt = hb[cur ^ 1];
    __syncthreads();
    vm_drain();         // (kept-activation stores of the layer before: see vm_drain)
    // layer 0's input may be wider than one pass of 8 waves x 32 units: chunks of 256 units
    for (int c0 = 0; c0 < nt; c0 += 16) {
      f32x4v acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
      if (c0 + tile0 < nt)
        rows16_gemm<4>(acc, n.Wtf[l], wf16_nkg(K), c0 + tile0, nt, in + r16 * PH + 4 * qd, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int u = c0 * 16 + u0 + 16 * t;
        float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if (l > 0) {
          if (mask) {
            const float4 hm = guarded_load4(n.act[l - 1], (int64_t)row * N, rok, u, N);
            v.x = hm.x > 0.f ? v.x : 0.f; v.y = hm.y > 0.f ? v.y : 0.f;
            v.z = hm.z > 0.f ? v.z : 0.f; v.w = hm.w > 0.f ? v.w : 0.f;
          }
          if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (u < PH - 4) lds_st4(nxt + r16 * PH + u, v);
          if (rok) store4_guarded(n.dz[l], (int64_t)row * N, u, N, (N & 3) == 0, v);
        } else if (rok) {
          store4_guarded(n.d_x, (int64_t)row * n.lddx, u, N,
                         is_vec_ok(n.d_x, n.lddx) && (N & 3) == 0, v);
        }
      }
    }
    in = nxt;
    cur ^= 1;
  }
}

// ---- fragment-major copies of every layer of one parameter set ---------------------------------
struct RowPackArgs {
  const float* P;                      // flat parameters
  int64_t woff[ROW_MAX_LAYERS];
  int dims[ROW_MAX_LAYERS + 1];
  int L;
  float* Wf[ROW_MAX_LAYERS];           // [d_{l+1} units][d_l]
  float* Wtf[ROW_MAX_LAYERS];          // [d_l units][d_{l+1}] or null
  void* Wsp[ROW_MAX_LAYERS];           // W_l as bf16x3 split planes (wsp16_index) or null
  void* Wtsp[ROW_MAX_LAYERS];          // W_l^T as split planes ([d_l units][d_{l+1}]) or null
};
static __global__ __launch_bounds__(256) void mlp_rowpack_kernel(RowPackArgs a) {
  const int64_t gsz = (int64_t)gridDim.x * 256;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int l = 0; l < a.L; ++l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    const float* W = a.P + a.woff[l];
    {
      const int nkg = wf16_nkg(K);
      const int64_t total = wf16_floats(N, K) / 4;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int g = (int)(tg % nkg), T = (int)(tg / nkg);
        const int u = T * 16 + (lane & 15), k = g * 16 + 4 * (lane >> 4);
        float4 v;
        v.x = (u < N && k < K) ? W[(int64_t)u * K + k] : 0.f;
        v.y = (u < N && k + 1 < K) ? W[(int64_t)u * K + k + 1] : 0.f;
        v.z = (u < N && k + 2 < K) ? W[(int64_t)u * K + k + 2] : 0.f;
        v.w = (u < N && k + 3 < K) ? W[(int64_t)u * K + k + 3] : 0.f;
        reinterpret_cast<float4*>(a.Wf[l])[e] = v;
      }
    }
    if (a.Wsp[l]) {   // one thread per 16-byte slot of every plane: 8 consecutive k of one unit
      const int nks = wsp16_nks(K);
      const int64_t total = wsp16_bytes(N, K) / 16;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int plane = (int)(tg % 3);
        const int64_t ts = tg / 3;
        const int s = (int)(ts % nks), T = (int)(ts / nks);
        const int u = T * 16 + (lane & 15), k0 = s * 32 + 8 * (lane >> 4);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = (u < N && k0 + j < K) ? W[(int64_t)u * K + k0 + j] : 0.f;
          __bf16 hi, mid, lo;
          split3(x, hi, mid, lo);
          v[j] = plane == 0 ? hi : (plane == 1 ? mid : lo);
        }
        reinterpret_cast<bf16x8*>(a.Wsp[l])[e] = v;
      }
    }
    if (a.Wtsp[l]) {  // the transpose: unit = input index k of W_l, 8 consecutive output units per slot
      const int nks = wsp16_nks(N);
      const int64_t total = wsp16_bytes(K, N) / 16;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int plane = (int)(tg % 3);
        const int64_t ts = tg / 3;
        const int s = (int)(ts % nks), T = (int)(ts / nks);
        const int k = T * 16 + (lane & 15), u0 = s * 32 + 8 * (lane >> 4);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = (k < K && u0 + j < N) ? W[(int64_t)(u0 + j) * K + k] : 0.f;
          __bf16 hi, mid, lo;
          split3(x, hi, mid, lo);
          v[j] = plane == 0 ? hi : (plane == 1 ? mid : lo);
        }
        reinterpret_cast<bf16x8*>(a.Wtsp[l])[e] = v;
      }
    }
    if (a.Wtf[l]) {   // "unit" = input index k of W_l, reduction over its output units
      const int nkg = wf16_nkg(N);
      const int64_t total = wf16_floats(K, N) / 4;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int g = (int)(tg % nkg), T = (int)(tg / nkg);
        const int k = T * 16 + (lane & 15), u = g * 16 + 4 * (lane >> 4);
        float4 v;
        v.x = (k < K && u < N) ? W[(int64_t)u * K + k] : 0.f;
        v.y = (k < K && u + 1 < N) ? W[(int64_t)(u + 1) * K + k] : 0.f;
        v.z = (k < K && u + 2 < N) ? W[(int64_t)(u + 2) * K + k] : 0.f;
        v.w = (k < K && u + 3 < N) ? W[(int64_t)(u + 3) * K + k] : 0.f;
        reinterpret_cast<float4*>(a.Wtf[l])[e] = v;
      }
    }
  }
}

// theta' <- tau theta + (1 - tau) theta' (common/utils.py:214-226) with the target's fragment-major
// copies refreshed in the same pass (no repack launch before the next target forward).
struct RowSoftArgs {
  float* tgt; const float* src;
  int64_t n;
  float tau, one_minus_tau;
  int64_t woff[ROW_MAX_LAYERS];
  int dims[ROW_MAX_LAYERS + 1];
  int L;
  float* Wf[ROW_MAX_LAYERS];           // target packed copies, or all null
};
static __global__ __launch_bounds__(256) void mlp_soft_update_kernel(RowSoftArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float t = __fadd_rn(__fmul_rn(a.tau, a.src[i]), __fmul_rn(a.one_minus_tau, a.tgt[i]));
  a.tgt[i] = t;
  for (int l = 0; l < a.L; ++l) {
    const int64_t e = i - a.woff[l];
    if (a.Wf[l] && e >= 0 && e < (int64_t)a.dims[l + 1] * a.dims[l]) {
      const int row = (int)(e / a.dims[l]), col = (int)(e - (int64_t)row * a.dims[l]);
      a.Wf[l][wf16_index(row, col, wf16_nkg(a.dims[l]))] = t;
      break;
    }
  }
}

}  // namespace pa
