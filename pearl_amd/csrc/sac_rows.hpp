// Continuous soft actor-critic: the row-wise work of one learn_batch in TWO launches.
//
// pearl/policy_learners/sequential_decision_making/soft_actor_critic_continuous.py:131-231 on
// actor_critic_base.py:309-366.  Sequenced launch by launch (sac_step.hip) a step is ~22 launches of
// 3-18 us — each a network pass of sixteen-row tiles that fills 64-128 of the 256 CUs, or a head that
// fills one — and the learner ran at 4.1 M transitions/s (248 us per step of 1024).  Everything
// between two optimizer steps is ROW-LOCAL, so a workgroup can take sixteen rows through all of it
// with the activations in LDS:
//
//   sac_rows_a   grid (tiles, 3)
//     y = 0  actor update rows (:208-231): actor(s) -> (a, log pi) -> both critics at (s, a) ->
//            d(alpha log pi - min q)/da -> head gradient -> the actor's pre-activation gradients.
//            The critics' input gradient needs no per-critic backward pass after the loss: with
//            s2 = [h2 > 0] w3 and G = (s2 W2) [h1 > 0], dq/da = G W1[:, S:], and the twin rule only
//            scales the two row vectors by -w/B (online_kernels.hpp uses the same identity).
//     y = 1, 2  critic c at (s, a_batch) (:178-206), run early: it only depends on parameters the
//            actor update does not touch.  Keeps h1, h2, q, and the UNSCALED pre-activation
//            gradients s2 and G — the Bellman error that scales them comes from sac_rows_b.
//   [the actor's weight gradients + AdamW: weight_grad_kernel, as before]
//   sac_rows_b   grid (tiles)
//     updated actor(s') -> (a', log pi') -> target critics -> y (:155-176); dq_c = (q_c - y) / B;
//     rows of s2 / G scaled by dq_c in place; the last workgroup to finish adds up the critic loss
//     and takes the entropy-coefficient step (:134-151).
//   [the critics' weight gradients + AdamW + soft updates, as before]
//
// Tiles and operand layouts are mlp_rowpass.hpp's (16 rows x 256 units per workgroup, 8 waves x 2
// unit tiles, fragment-major weights).  NGH > 0: every hidden layer is NGH k-groups wide and its
// loops are unrolled through a register ring filled one phase early; NGH = 0: any width <= 256.
#pragma once
#include "mlp_rowpass.hpp"

namespace pa {

struct SacMlp3 {
  const float* W1f; const float* b1;     // [H1 units][K0]
  const float* W2f; const float* b2;     // [H2 units][H1]
  const float* W3f; const float* b3;     // actor head [DO units][H2]; critics: b3 only
  const float* w3;                       // critics: the last layer's row [H2]
  const float* W2tf;                     // [H1 units][H2]   (d h1 = d z2 W2)
  const float* W1tf;                     // [K0 units][H1]   (d x  = d z1 W1)
  const float* W3tf;                     // actor: [H2 units][DO]
  float* act1; float* act2;              // kept ReLU outputs [B][H1], [B][H2]
  float* dz1; float* dz2;                // pre-activation gradients [B][H1], [B][H2]
  int K0, H1, H2, DO;
};

struct SacTicket {
  float* partials;        // [tiles][2]
  unsigned* ticket;       // zero between launches
};

struct SacRowsAArgs {
  SacMlp3 actor, critic[2];
  const float* state; int ld_state;
  const float* action; int ld_action;    // the batch's actions
  const float* noise; int ld_noise;      // [B][A]
  const float* low; const float* high;
  const float* alpha;
  int B, S, A;
  float* d_head;                         // [B][2A]
  float* logp;                           // [B]
  float* xq;                             // [B][S + A]
  float* q[2];                           // [B] critics at (s, a_batch)
  SacTicket tk;
  float* loss_out;                       // actor loss
  long long* prof;
};

struct SacRowsBArgs {
  SacMlp3 actor, target[2];
  float* dz1[2]; float* dz2[2];          // the online critics' unscaled gradients (scaled here)
  int H1c, H2c;
  const float* next_state; int ld_next;
  const float* noise; int ld_noise;
  const float* low; const float* high;
  const float* alpha_in;
  const float* reward; const uint8_t* term; float gamma;
  const float* q[2];
  float* dq[2];                          // [B]
  int B, S, A;
  SacTicket tk;
  float* loss_out;                       // critic loss
  // entropy coefficient (null log_alpha: fixed)
  float* log_alpha; float* am; float* av; float* avmax; float* alpha;
  const float* logp; float target_entropy; AdamScalars ac; float* alpha_loss_out;
  long long* prof;
};

// phase stamps (tools/prof_sac.py): 32 slots per wave, 100 MHz wall clock; null outside the tool
#define SR_STAMP(prof, wg, i)                                                            \
  do {                                                                                   \
    if ((prof) && (threadIdx.x & 63) == 0)                                               \
      (prof)[((int64_t)(wg) * 8 + (threadIdx.x >> 6)) * 32 + (i)] = (long long)wall_clock64(); \
  } while (0)

constexpr int SR_HEADP = 36;     // LDS pitch of 32-wide row vectors (head, head gradient, partials)
constexpr int SR_DHP = 68;       // pitch of the head-gradient tile when it is a GEMM operand

// LDS: xs [16][P0] | hA hB hC [16][PH] | red [8][16][SR_HEADP] | small
__host__ __device__ inline size_t sac_rows_smem_floats(int k0) {
  return (size_t)RP_ROWS * (rp_pad(k0) + 3 * row_hid_pitch()) + 8 * RP_ROWS * SR_HEADP +
         RP_ROWS * SR_DHP + 8 * RP_ROWS + 8 * RP_ROWS;
}

struct SrLane {
  int tid, lane, wave, r16, qd, u0, tile0;
};
__device__ __forceinline__ SrLane sr_lane() {
  SrLane L;
  L.tid = threadIdx.x; L.lane = L.tid & 63; L.wave = L.tid >> 6;
  L.r16 = L.lane & 15; L.qd = L.lane >> 4;
  L.u0 = L.wave * 32 + 4 * L.qd; L.tile0 = L.wave * 2;
  return L;
}

__device__ __forceinline__ void sr_bias(f32x4v (&acc)[2], const float* b, int N, int u0) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 v = guarded_load4(b, 0, true, u0 + 16 * t, N);
    acc[t][0] = v.x; acc[t][1] = v.y; acc[t][2] = v.z; acc[t][3] = v.w;
  }
}
__device__ __forceinline__ void sr_zero(f32x4v (&acc)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
}

template <int NGH>
__device__ __forceinline__ void sr_prefetch(WRing& R, const float* Wf, int tile0, int nt, int lane) {
  if constexpr (NGH > 0) ring_fill<NGH>(R, Wf, tile0, nt, lane);
}
template <int NGH>
__device__ __forceinline__ void sr_gemm(f32x4v (&acc)[2], WRing& R, const float* Wf, int nkg,
                                        int tile0, int nt, const float* actp, int lane) {
  if constexpr (NGH > 0) rows16_gemm_static<NGH>(acc, R, Wf, tile0, nt, actp, lane);
  else rows16_gemm<4>(acc, Wf, nkg, tile0, nt, actp, lane);
}

// relu(acc) -> LDS tile (+ global), returns the ReLU mask (bit 4 t + e)
__device__ __forceinline__ unsigned sr_relu_out(const f32x4v (&acc)[2], float* tile, int PH,
                                                const SrLane& L, float* keep, int64_t row, int N,
                                                bool rok) {
  unsigned m = 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = L.u0 + 16 * t;
    float4 v = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]),
                           relu_keep_nan(acc[t][2]), relu_keep_nan(acc[t][3]));
    m |= (v.x > 0.f ? 1u : 0u) << (4 * t) | (v.y > 0.f ? 2u : 0u) << (4 * t) |
         (v.z > 0.f ? 4u : 0u) << (4 * t) | (v.w > 0.f ? 8u : 0u) << (4 * t);
    *reinterpret_cast<float4*>(tile + L.r16 * PH + u) = v;
    if (keep && rok) store4_guarded(keep, row * N, u, N, (N & 3) == 0, v);
  }
  return m;
}
// acc masked by m -> LDS tile (+ global)
__device__ __forceinline__ void sr_mask_out(const f32x4v (&acc)[2], unsigned m, float* tile, int PH,
                                            const SrLane& L, float* keep, int64_t row, int N,
                                            bool rok) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = L.u0 + 16 * t;
    const unsigned mt = m >> (4 * t);
    float4 v = make_float4((mt & 1u) ? acc[t][0] : 0.f, (mt & 2u) ? acc[t][1] : 0.f,
                           (mt & 4u) ? acc[t][2] : 0.f, (mt & 8u) ? acc[t][3] : 0.f);
    if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tile) *reinterpret_cast<float4*>(tile + L.r16 * PH + u) = v;
    if (keep && rok) store4_guarded(keep, row * N, u, N, (N & 3) == 0, v);
  }
}

// Narrow output (<= 2 unit tiles starting at t_lo), K split over the eight waves: wave w takes
// k-groups w and w + 8 (a hidden layer has at most 16), the partial tiles go to red[wave].
__device__ __forceinline__ void sr_narrow_part(const float* Wf, int nkg, int t_lo, int ntl,
                                               const float* actp, float* red, const SrLane& L) {
  f32x4v acc[2];
  sr_zero(acc);
  const int g0 = L.wave, g1 = L.wave + 8;
  const bool k0 = g0 < nkg, k1 = g1 < nkg;
  const int64_t b0 = ((int64_t)t_lo * nkg) * 256 + L.lane * 4;
  const int64_t b1 = b0 + (int64_t)nkg * 256;
  const float4 w00 = ld4_or_zero(Wf, b0 + (int64_t)g0 * 256, k0 && ntl > 0);
  const float4 w10 = ld4_or_zero(Wf, b1 + (int64_t)g0 * 256, k0 && ntl > 1);
  const float4 w01 = ld4_or_zero(Wf, b0 + (int64_t)g1 * 256, k1 && ntl > 0);
  const float4 w11 = ld4_or_zero(Wf, b1 + (int64_t)g1 * 256, k1 && ntl > 1);
  if (k0) {
    const float4 x4 = *reinterpret_cast<const float4*>(actp + g0 * 16);
    acc[0] = mfma16(w00.x, x4.x, acc[0]); acc[1] = mfma16(w10.x, x4.x, acc[1]);
    acc[0] = mfma16(w00.y, x4.y, acc[0]); acc[1] = mfma16(w10.y, x4.y, acc[1]);
    acc[0] = mfma16(w00.z, x4.z, acc[0]); acc[1] = mfma16(w10.z, x4.z, acc[1]);
    acc[0] = mfma16(w00.w, x4.w, acc[0]); acc[1] = mfma16(w10.w, x4.w, acc[1]);
  }
  if (k1) {
    const float4 x4 = *reinterpret_cast<const float4*>(actp + g1 * 16);
    acc[0] = mfma16(w01.x, x4.x, acc[0]); acc[1] = mfma16(w11.x, x4.x, acc[1]);
    acc[0] = mfma16(w01.y, x4.y, acc[0]); acc[1] = mfma16(w11.y, x4.y, acc[1]);
    acc[0] = mfma16(w01.z, x4.z, acc[0]); acc[1] = mfma16(w11.z, x4.z, acc[1]);
    acc[0] = mfma16(w01.w, x4.w, acc[0]); acc[1] = mfma16(w11.w, x4.w, acc[1]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
    *reinterpret_cast<float4*>(red + (L.wave * RP_ROWS + L.r16) * SR_HEADP + 16 * t + 4 * L.qd) =
        make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
}
__device__ __forceinline__ float sr_narrow_get(const float* red, int r, int c) {
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) s += red[(w * RP_ROWS + r) * SR_HEADP + c];
  return s;
}

// state (or next state) tile -> xs[:, 0:S], zeros up to the pitch
__device__ __forceinline__ void sr_stage(const float* x, int ldx, int S, int m0, int B, float* xs,
                                         int P0, int tid, int col0) {
  const int c4 = (P0 - 4) >> 2;
  const bool vx = is_vec_ok(x, ldx) && ((S & 3) == 0) && ((col0 & 3) == 0);
  for (int e = tid; e < RP_ROWS * c4; e += 512) {
    const int r = e / c4, c = (e - r * c4) * 4;
    const bool ok = (m0 + r) < B;
    if (c < col0) continue;
    float4 v;
    if (vx) v = ld4_or_zero(x, (int64_t)(m0 + r) * ldx + (c - col0), ok && (c - col0) < S);
    else v = guarded_load4(x, (int64_t)(m0 + r) * ldx, ok, c - col0, S);
    *reinterpret_cast<float4*>(xs + r * P0 + c) = v;
  }
}

// One critic on the tile in xs: q, and (WANT_G) the unit gradients  s2 = [h2 > 0] w3 and
// Gm = (s2 W2) [h1 > 0] — Gm stays in hC.  KEEP: h1, h2, s2, Gm also go to the network's kept
// buffers.  Opens with a barrier (xs complete, hA / hB / hC free); the caller puts a barrier
// between this and its first read of hC / qred.  The ring must hold W2f's first k-groups on entry
// (NGH > 0) and holds `Wnext`'s on return.
template <int NGH, bool WANT_G, bool KEEP>
__device__ __forceinline__ void sr_critic(const SacMlp3& n, const float* xs, int P0, float* hA,
                                          float* hB, float* hC, float* qred, WRing& R,
                                          const SrLane& L, int64_t row, bool rok,
                                          const float* Wnext, int nt_next, long long* prof = nullptr,
                                          int wg = 0, int slot = 0) {
  const int PH = row_hid_pitch();
  const int nt1 = (n.H1 + 15) >> 4, nt2 = (n.H2 + 15) >> 4;
  f32x4v acc[2];
  // ---- layer 1
  sr_bias(acc, n.b1, n.H1, L.u0);
  __syncthreads();
  rows16_gemm<4>(acc, n.W1f, wf16_nkg(n.K0), L.tile0, nt1, xs + L.r16 * P0 + 4 * L.qd, L.lane);
  const unsigned m1 = sr_relu_out(acc, hA, PH, L, KEEP ? n.act1 : nullptr, row, n.H1, rok);
  SR_STAMP(prof, wg, slot);
  // ---- layer 2, the head's dot product, s2
  sr_bias(acc, n.b2, n.H2, L.u0);
  float4 w3v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) w3v[t] = guarded_load4(n.w3, 0, true, L.u0 + 16 * t, n.H2);
  __syncthreads();
  sr_gemm<NGH>(acc, R, n.W2f, wf16_nkg(n.H1), L.tile0, nt2, hA + L.r16 * PH + 4 * L.qd, L.lane);
  if (WANT_G) sr_prefetch<NGH>(R, n.W2tf, L.tile0, nt1, L.lane);
  else if (Wnext) sr_prefetch<NGH>(R, Wnext, L.tile0, nt_next, L.lane);
  float qp = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = L.u0 + 16 * t;
    const float4 h = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]),
                                 relu_keep_nan(acc[t][2]), relu_keep_nan(acc[t][3]));
    qp += h.x * w3v[t].x;
    qp += h.y * w3v[t].y;
    qp += h.z * w3v[t].z;
    qp += h.w * w3v[t].w;
    if (KEEP && rok) store4_guarded(n.act2, row * n.H2, u, n.H2, (n.H2 & 3) == 0, h);
    if (WANT_G) {
      const float4 s2 = make_float4(h.x > 0.f ? w3v[t].x : 0.f, h.y > 0.f ? w3v[t].y : 0.f,
                                    h.z > 0.f ? w3v[t].z : 0.f, h.w > 0.f ? w3v[t].w : 0.f);
      *reinterpret_cast<float4*>(hB + L.r16 * PH + u) = s2;
      if (KEEP && rok) store4_guarded(n.dz2, row * n.H2, u, n.H2, (n.H2 & 3) == 0, s2);
    }
  }
  qp += __shfl_xor(qp, 16);
  qp += __shfl_xor(qp, 32);
  if (L.qd == 0) qred[L.wave * RP_ROWS + L.r16] = qp;
  SR_STAMP(prof, wg, slot + 1);
  if (WANT_G) {
    // ---- Gm = (s2 W2) [h1 > 0]
    sr_zero(acc);
    __syncthreads();
    sr_gemm<NGH>(acc, R, n.W2tf, wf16_nkg(n.H2), L.tile0, nt1, hB + L.r16 * PH + 4 * L.qd, L.lane);
    if (Wnext) sr_prefetch<NGH>(R, Wnext, L.tile0, nt_next, L.lane);
    sr_mask_out(acc, m1, hC, PH, L, KEEP ? n.dz1 : nullptr, row, n.H1, rok);
    SR_STAMP(prof, wg, slot + 2);
  }
}
// q of row r from the eight waves' partial dot products (after a barrier)
__device__ __forceinline__ float sr_q(const float* qred, int r, const float* b3) {
  float q = b3[0];
#pragma unroll
  for (int w = 0; w < 8; ++w) q += qred[w * RP_ROWS + r];
  return q;
}

// The actor on the tile in xs[:, 0:S]: head [16][2A] summed into headS (pitch SR_HEADP).  KEEP: the
// hidden activations also go to act1 / act2 and the ReLU masks are returned.  Opens with a
// barrier; closes with the barrier after which headS is complete.  Ring: holds W2f on entry.
template <int NGH, bool KEEP>
__device__ __forceinline__ void sr_actor_fwd(const SacMlp3& n, const float* xs, int P0, float* hA,
                                             float* hB, float* red, float* headS, WRing& R,
                                             const SrLane& L, int64_t row, bool rok, unsigned& m1,
                                             unsigned& m2, const float* Wnext, int nt_next) {
  const int PH = row_hid_pitch();
  const int nt1 = (n.H1 + 15) >> 4, nt2 = (n.H2 + 15) >> 4;
  f32x4v acc[2];
  sr_bias(acc, n.b1, n.H1, L.u0);
  __syncthreads();
  rows16_gemm<4>(acc, n.W1f, wf16_nkg(n.K0), L.tile0, nt1, xs + L.r16 * P0 + 4 * L.qd, L.lane);
  m1 = sr_relu_out(acc, hA, PH, L, KEEP ? n.act1 : nullptr, row, n.H1, rok);
  sr_bias(acc, n.b2, n.H2, L.u0);
  __syncthreads();
  sr_gemm<NGH>(acc, R, n.W2f, wf16_nkg(n.H1), L.tile0, nt2, hA + L.r16 * PH + 4 * L.qd, L.lane);
  if (Wnext) sr_prefetch<NGH>(R, Wnext, L.tile0, nt_next, L.lane);
  m2 = sr_relu_out(acc, hB, PH, L, KEEP ? n.act2 : nullptr, row, n.H2, rok);
  __syncthreads();
  sr_narrow_part(n.W3f, wf16_nkg(n.H2), 0, (n.DO + 15) >> 4, hB + L.r16 * PH + 4 * L.qd, red, L);
  __syncthreads();
  if (L.tid < RP_ROWS * 32) {
    const int r = L.tid >> 5, c = L.tid & 31;
    headS[r * SR_HEADP + c] = c < n.DO ? sr_narrow_get(red, r, c) + n.b3[c] : 0.f;
  }
  __syncthreads();
}

// GaussianActorNetwork.sample_action for the thread's (row r, component j); writes the action into
// xs[r][S + j]; returns the row's log-prob term.  (gauss_sample_kernel's arithmetic.)
struct SrGauss {
  float t, sd, n, eps, bound;
};
__device__ __forceinline__ float sr_sample(const float* headS, int r, int j, int A, float eps,
                                           float lo, float hi, float* xs, int P0, int S,
                                           SrGauss& G) {
  const float mean = headS[r * SR_HEADP + j], raw = headS[r * SR_HEADP + A + j];
  G.t = tanhf(raw);
  const float log_std = -5.0f + 3.5f * (G.t + 1.0f);
  G.sd = expf(log_std);
  const float u = mean + G.sd * eps;
  G.n = tanhf(u);
  G.eps = eps;
  xs[r * P0 + S + j] = (((hi - lo) * (G.n + 1.0f)) / 2.0f) + lo;
  const float var = G.sd * G.sd;
  const float diff = u - mean;
  float l = -(diff * diff) / (2.0f * var) - logf(G.sd) - 0.9189385332046727f;
  G.bound = (hi - lo) / 2.0f;
  l -= logf(G.bound * (1.0f - G.n * G.n) + 1e-6f);
  return l;
}

// sum of 512 per-thread values: fixed order (LDS tree over the first 256 + 256 slots)
__device__ __forceinline__ float sr_block_sum(float v, float* red512) {
  red512[threadIdx.x] = v;
  __syncthreads();
  for (int w = 256; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red512[threadIdx.x] += red512[threadIdx.x + w];
    __syncthreads();
  }
  const float r = red512[0];
  __syncthreads();
  return r;
}

template <int NGH>
__global__ __launch_bounds__(512) void sac_rows_a_kernel(SacRowsAArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SrLane L = sr_lane();
  const int W = a.S + a.A;
  const int P0 = rp_pad(W), PH = row_hid_pitch();
  float* xs = smem;
  float* hA = xs + RP_ROWS * P0;
  float* hB = hA + RP_ROWS * PH;
  float* hC = hB + RP_ROWS * PH;
  float* red = hC + RP_ROWS * PH;                  // [8][16][SR_HEADP]
  float* dhS = red + 8 * RP_ROWS * SR_HEADP;       // [16][SR_DHP]
  float* qred = dhS + RP_ROWS * SR_DHP;            // [8][16]
  float* small = qred + 8 * RP_ROWS;               // [8][16]: q1 q2 logp lossrow ...
  const int m0 = blockIdx.x * RP_ROWS;
  const int64_t row = m0 + L.r16;
  const bool rok = row < a.B;
  WRing R;
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  SR_STAMP(a.prof, wg, 0);

  if (blockIdx.y > 0) {
    // ---------------------------------------------------------------- critic c at (s, a_batch)
    const int c = blockIdx.y - 1;
    const SacMlp3& n = a.critic[c];
    sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
    // xs = state || action, zero padded
    {
      const int c4 = (P0 - 4) >> 2;
      for (int e = L.tid; e < RP_ROWS * c4; e += 512) {
        const int r = e / c4, cc = (e - r * c4) * 4;
        const bool ok = (m0 + r) < a.B;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int col = cc + k;
          v[k] = col < a.S ? ld_or_zero(a.state, (int64_t)(m0 + r) * a.ld_state + col, ok)
                           : ld_or_zero(a.action, (int64_t)(m0 + r) * a.ld_action + (col - a.S),
                                        ok && col < W);
        }
        *reinterpret_cast<float4*>(xs + r * P0 + cc) = make_float4(v[0], v[1], v[2], v[3]);
        if (c == 0 && ok) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (cc + k < W) a.xq[(int64_t)(m0 + r) * W + cc + k] = v[k];
        }
      }
    }
    SR_STAMP(a.prof, wg, 1);
    sr_critic<NGH, true, true>(n, xs, P0, hA, hB, hC, qred, R, L, row, rok, nullptr, 0, a.prof, wg, 4);
    __syncthreads();
    if (L.tid < RP_ROWS && m0 + L.tid < a.B) a.q[c][m0 + L.tid] = sr_q(qred, L.tid, n.b3);
    SR_STAMP(a.prof, wg, 15);
    return;
  }

  // ------------------------------------------------------------------ actor update rows
  const SacMlp3& n = a.actor;
  sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
  // noise of this thread's (row, component), requested before anything else
  const int sr = L.tid / a.A, sj = L.tid - sr * a.A;
  const bool sok = L.tid < RP_ROWS * a.A;
  const bool srok = sok && (m0 + sr) < a.B;
  const float eps = ld_or_zero(a.noise, (int64_t)(m0 + sr) * a.ld_noise + sj, srok);
  const float lo = ld_or_zero(a.low, sj, sok), hi = ld_or_zero(a.high, sj, sok);
  const float alpha = a.alpha[0];
  sr_stage(a.state, a.ld_state, a.S, m0, a.B, xs, P0, L.tid, 0);
  SR_STAMP(a.prof, wg, 1);
  unsigned m1a, m2a;
  float* headS = dhS;   // [16][SR_HEADP]; dhS proper is written only after the head was consumed
  sr_actor_fwd<NGH, true>(n, xs, P0, hA, hB, red, headS, R, L, row, rok, m1a, m2a,
                          a.critic[0].W2f, (a.critic[0].H2 + 15) >> 4);
  SR_STAMP(a.prof, wg, 2);
  // ---- sample: action -> xs[:, S:], log pi
  SrGauss G;
  G.t = G.sd = G.n = G.eps = G.bound = 0.f;
  float* terms = red;                       // [16][16]
  if (sok) terms[sr * 16 + sj] = sr_sample(headS, sr, sj, a.A, eps, lo, hi, xs, P0, a.S, G);
  __syncthreads();
  if (L.tid < RP_ROWS) {
    float lp = 0.f;
    for (int k = 0; k < a.A; ++k) lp += terms[L.tid * 16 + k];
    small[2 * RP_ROWS + L.tid] = lp;
    if (m0 + L.tid < a.B) a.logp[m0 + L.tid] = lp;
  }
  SR_STAMP(a.prof, wg, 3);
  // ---- both critics at (s, a): q_c and gx_c = Gm_c W1_c[:, S:]
  float gx[2];
  gx[0] = gx[1] = 0.f;
  const int t_lo = a.S >> 4;
  const int ntl = ((W + 15) >> 4) - t_lo;       // <= 2 (A <= 16)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const SacMlp3& q = a.critic[c];
    sr_critic<NGH, true, false>(q, xs, P0, hA, hB, hC, qred, R, L, row, rok,
                                c == 0 ? a.critic[1].W2f : n.W2tf,
                                c == 0 ? (a.critic[1].H2 + 15) >> 4 : (n.H1 + 15) >> 4, a.prof, wg,
                                4 + 4 * c);
    __syncthreads();
    if (L.tid < RP_ROWS) small[c * RP_ROWS + L.tid] = sr_q(qred, L.tid, q.b3);
    sr_narrow_part(q.W1tf, wf16_nkg(q.H1), t_lo, ntl, hC + L.r16 * PH + 4 * L.qd, red, L);
    __syncthreads();
    if (sok) gx[c] = sr_narrow_get(red, sr, (a.S & 15) + sj);
    SR_STAMP(a.prof, wg, 7 + 4 * c);
  }
  // ---- twin rule, loss, head gradient (twin_kernel mode 0, gauss_grad_kernel)
  float lossrow = 0.f;
  if (sok) {
    const float q1 = small[sr], q2 = small[RP_ROWS + sr];
    const float w1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
    const float dq1 = -w1 / (float)a.B, dq2 = -(1.f - w1) / (float)a.B;
    const float dla = dq1 * gx[0] + dq2 * gx[1];
    const float coef = alpha / (float)a.B;
    const float one_m_n2 = 1.0f - G.n * G.n;
    const float dlp_du = (2.0f * G.bound * G.n * one_m_n2) / (G.bound * one_m_n2 + 1e-6f);
    const float da_du = G.bound * one_m_n2;
    const float dl_du = coef * dlp_du + dla * da_du;
    const float dl_dls = dl_du * G.eps * G.sd - coef;
    const float d_mu = srok ? dl_du : 0.f;
    const float d_ls = srok ? dl_dls * 3.5f * (1.0f - G.t * G.t) : 0.f;
    // headS and dhS share storage: every read of the head happened before the critics' barriers
    dhS[sr * SR_DHP + sj] = d_mu;
    dhS[sr * SR_DHP + a.A + sj] = d_ls;
    if (srok) {
      a.d_head[(int64_t)(m0 + sr) * 2 * a.A + sj] = d_mu;
      a.d_head[(int64_t)(m0 + sr) * 2 * a.A + a.A + sj] = d_ls;
      if (sj == 0) lossrow = alpha * small[2 * RP_ROWS + sr] - fminf(q1, q2);
    }
  }
  SR_STAMP(a.prof, wg, 12);
  // zero the rest of the head-gradient tile (k padding of the next GEMM)
  for (int e = L.tid; e < RP_ROWS * SR_DHP; e += 512) {
    const int cc = e % SR_DHP;
    if (cc >= 2 * a.A) dhS[e] = 0.f;
  }
  // ---- actor backward: d z2 = (d head W3) [h2 > 0], d z1 = (d z2 W2) [h1 > 0]
  f32x4v acc[2];
  sr_zero(acc);
  __syncthreads();
  rows16_gemm<4>(acc, n.W3tf, wf16_nkg(n.DO), L.tile0, (n.H2 + 15) >> 4,
                 dhS + L.r16 * SR_DHP + 4 * L.qd, L.lane);
  sr_mask_out(acc, m2a, hA, PH, L, n.dz2, row, n.H2, rok);
  SR_STAMP(a.prof, wg, 13);
  sr_zero(acc);
  __syncthreads();
  sr_gemm<NGH>(acc, R, n.W2tf, wf16_nkg(n.H2), L.tile0, (n.H1 + 15) >> 4,
               hA + L.r16 * PH + 4 * L.qd, L.lane);
  sr_mask_out(acc, m1a, nullptr, PH, L, n.dz1, row, n.H1, rok);
  SR_STAMP(a.prof, wg, 14);
  // ---- actor loss: per-tile partial, the last workgroup adds them in tile order
  __syncthreads();
  const float part = sr_block_sum(lossrow, hB);
  __shared__ unsigned last;
  if (L.tid == 0) {
    a.tk.partials[blockIdx.x] = part;
    __threadfence();
    last = (atomicAdd(a.tk.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  SR_STAMP(a.prof, wg, 15);
  if (!last) return;
  __threadfence();
  float p = 0.f;
  for (unsigned k = L.tid; k < gridDim.x; k += 512) p += __builtin_nontemporal_load(a.tk.partials + k);
  const float total = sr_block_sum(p, hB);
  if (L.tid == 0) {
    a.loss_out[0] = total / (float)a.B;
    *a.tk.ticket = 0u;
  }
}

template <int NGH>
__global__ __launch_bounds__(512) void sac_rows_b_kernel(SacRowsBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SrLane L = sr_lane();
  const int W = a.S + a.A;
  const int P0 = rp_pad(W), PH = row_hid_pitch();
  float* xs = smem;
  float* hA = xs + RP_ROWS * P0;
  float* hB = hA + RP_ROWS * PH;
  float* hC = hB + RP_ROWS * PH;
  float* red = hC + RP_ROWS * PH;
  float* headS = red + 8 * RP_ROWS * SR_HEADP;
  float* qred = headS + RP_ROWS * SR_DHP;
  float* small = qred + 8 * RP_ROWS;
  const int m0 = blockIdx.x * RP_ROWS;
  const int64_t row = m0 + L.r16;
  const bool rok = row < a.B;
  WRing R;
  const int wg = blockIdx.x;
  SR_STAMP(a.prof, wg, 0);
  const SacMlp3& n = a.actor;
  sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
  const int sr = L.tid / a.A, sj = L.tid - sr * a.A;
  const bool sok = L.tid < RP_ROWS * a.A;
  const bool srok = sok && (m0 + sr) < a.B;
  const float eps = ld_or_zero(a.noise, (int64_t)(m0 + sr) * a.ld_noise + sj, srok);
  const float lo = ld_or_zero(a.low, sj, sok), hi = ld_or_zero(a.high, sj, sok);
  const float alpha = a.alpha_in[0];
  sr_stage(a.next_state, a.ld_next, a.S, m0, a.B, xs, P0, L.tid, 0);
  SR_STAMP(a.prof, wg, 1);
  unsigned m1a, m2a;
  sr_actor_fwd<NGH, false>(n, xs, P0, hA, hB, red, headS, R, L, row, rok, m1a, m2a, a.target[0].W2f,
                           (a.target[0].H2 + 15) >> 4);
  SR_STAMP(a.prof, wg, 2);
  SrGauss G;
  float* terms = red;
  if (sok) terms[sr * 16 + sj] = sr_sample(headS, sr, sj, a.A, eps, lo, hi, xs, P0, a.S, G);
  __syncthreads();
  if (L.tid < RP_ROWS) {
    float lp = 0.f;
    for (int k = 0; k < a.A; ++k) lp += terms[L.tid * 16 + k];
    small[2 * RP_ROWS + L.tid] = lp;
  }
  SR_STAMP(a.prof, wg, 3);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    sr_critic<NGH, false, false>(a.target[c], xs, P0, hA, hB, hC, qred, R, L, row, rok,
                                 c == 0 ? a.target[1].W2f : nullptr, (a.target[1].H2 + 15) >> 4,
                                 a.prof, wg, 4 + 4 * c);
    __syncthreads();
    if (L.tid < RP_ROWS) small[c * RP_ROWS + L.tid] = sr_q(qred, L.tid, a.target[c].b3);
  }
  __syncthreads();
  // ---- y, the Bellman errors, the critic loss rows (twin_kernel mode 1, mse_head_kernel)
  float lossrow = 0.f;
  if (L.tid < RP_ROWS) {
    const int b = m0 + L.tid;
    float d1 = 0.f, d2 = 0.f;
    if (b < a.B) {
      const float mn = fminf(small[L.tid], small[RP_ROWS + L.tid]);
      const float v = mn - alpha * small[2 * RP_ROWS + L.tid];
      const float live = 1.0f - (a.term[b] ? 1.0f : 0.0f);
      const float y = __fadd_rn(__fmul_rn(__fmul_rn(v, a.gamma), live), a.reward[b]);
      const float e1 = __fsub_rn(a.q[0][b], y), e2 = __fsub_rn(a.q[1][b], y);
      const float gs = 1.0f / (float)a.B;
      d1 = __fmul_rn(gs, e1);
      d2 = __fmul_rn(gs, e2);
      a.dq[0][b] = d1;
      a.dq[1][b] = d2;
      lossrow = e1 * e1 + e2 * e2;
    }
    small[3 * RP_ROWS + L.tid] = d1;
    small[4 * RP_ROWS + L.tid] = d2;
  }
  __syncthreads();
  SR_STAMP(a.prof, wg, 12);
  // ---- the online critics' gradients of these rows: s2, Gm scaled by dq_c
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* dqs = small + (3 + c) * RP_ROWS;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      float* base = which == 0 ? a.dz2[c] : a.dz1[c];
      const int Hn = which == 0 ? a.H2c : a.H1c;
      const int c4 = (Hn + 3) >> 2;
      const bool vec = (Hn & 3) == 0;
      for (int e = L.tid; e < RP_ROWS * c4; e += 512) {
        const int r = e / c4, cc = (e - r * c4) * 4;
        if (m0 + r >= a.B) continue;
        float* p = base + (int64_t)(m0 + r) * Hn + cc;
        const float s = dqs[r];
        if (vec) {
          float4 v = *reinterpret_cast<float4*>(p);
          v.x *= s; v.y *= s; v.z *= s; v.w *= s;
          *reinterpret_cast<float4*>(p) = v;
        } else {
          for (int k = 0; k < 4 && cc + k < Hn; ++k) p[k] *= s;
        }
      }
    }
  }
  SR_STAMP(a.prof, wg, 13);
  // ---- critic loss + entropy coefficient: the last workgroup to finish
  const float part = sr_block_sum(lossrow, hB);
  __shared__ unsigned last;
  if (L.tid == 0) {
    a.tk.partials[blockIdx.x] = part;
    __threadfence();
    last = (atomicAdd(a.tk.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  SR_STAMP(a.prof, wg, 15);
  if (!last) return;
  __threadfence();
  float p = 0.f;
  for (unsigned k = L.tid; k < gridDim.x; k += 512) p += __builtin_nontemporal_load(a.tk.partials + k);
  const float total = sr_block_sum(p, hB);
  if (L.tid == 0) {
    // ((q1 - y)^2 + (q2 - y)^2 summed) / B / 2  ==  (mse1 + mse2) / 2   (critic_utils.py:170-203)
    a.loss_out[0] = (total / (float)a.B) * 0.5f;
    *a.tk.ticket = 0u;
  }
  if (a.log_alpha) {
    // alpha_kernel's arithmetic: 256 strided partial sums, the same tree
    const float ea = expf(a.log_alpha[0]);
    float pa = 0.f;
    if (L.tid < 256)
      for (int b = L.tid; b < a.B; b += 256)
        pa += -ea * (__builtin_nontemporal_load(a.logp + b) + a.target_entropy);
    hB[L.tid] = L.tid < 256 ? pa : 0.f;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if (L.tid < w) hB[L.tid] += hB[L.tid + w];
      __syncthreads();
    }
    if (L.tid == 0) {
      const float g = hB[0] / (float)a.B;
      if (a.alpha_loss_out) a.alpha_loss_out[0] = g;
      AdamState st;
      st.p = a.log_alpha; st.m = a.am; st.v = a.av; st.vmax = a.avmax;
      const float pnew = adam_update(a.ac, st, 0, g);
      a.alpha[0] = expf(pnew);
    }
  }
}

}  // namespace pa
