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
//     rows of s2 / G scaled by dq_c in place.
//   [both critics' weight gradients + AdamW + soft target updates: ONE weight_grad_kernel launch]
//   sac_finish   one workgroup: the two losses from the per-tile partial sums, and the
//     entropy-coefficient step (:134-151).
//
// Tiles and operand layouts are mlp_rowpass.hpp's (16 rows x 256 units per workgroup, 8 waves x 2
// unit tiles, fragment-major weights).  NGH > 0: every hidden layer is NGH k-groups wide and its
// loops are unrolled through a register ring filled one phase early; NGH = 0: any width <= 256.
#pragma once
#include "mlp_rowpass.hpp"
#include "online_f16_kernel.hpp"

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
  const unsigned* um;                    // H2 instantiations: max |w| per row of W2 (bit patterns, [H2]),
                                         // kept by mlp.hip (mlp_ensure_um, the optimizer epilogue)
};

struct SacTicket {
  float* partials;        // [tiles] per-tile loss sums (sac_finish_kernel adds them up)
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
  int actor_rows;                        // 1: role 0 = actor update rows; 0: critic roles only
                                         // (TD3 on a step without an actor update)
  // HEAD 0, split = 1: the second critic at (s, pi(s)) runs in a helper workgroup of the same launch
  // (grid 4 x tiles: actor rows, helper, critic 1 rows, critic 2 rows).  The sampled action travels
  // to the helper and (q2, gx2) travel back as data-tagged words (publish_y / consume_y, the
  // DQN loop's hand-off): xact [tiles][16][16], xres [tiles][16][17], pre-filled with the tag and
  // restored to it by the reader.
  int split;
  float* xact; float* xres;
  int* err; int* err_host;
  long long* prof;
  // split launches inside a native learn loop: the actor's forward on this batch was run by the
  // previous step's sac_rows_b launch (its idle workgroups; SacRowsBArgs::pre_state) — the head
  // [B][2A] is here, the hidden activations in the actor's kept buffers.  null: run it here.
  const float* pre_head;
  const unsigned* pre_mask;              // [tiles][512][2]: the forward's ReLU masks, lane by lane
};
constexpr int SR_XACT = RP_ROWS * 16;
constexpr int SR_XRES = RP_ROWS * 17;
// consume_y with a short sleep (the partner is ~1 us away, not a stream away) + tag restore
__device__ __forceinline__ float sr_take(float* p, int* err, int* err_host) {
  unsigned* q = reinterpret_cast<unsigned*>(p);
  unsigned bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (bits == kYPendingBits) {
    const int limit = (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                          ? 0 : (1 << 23);
    int spins = 0;
    while (bits == kYPendingBits && spins < limit) {
      __builtin_amdgcn_s_sleep(2);
      bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++spins;
    }
    if (bits == kYPendingBits) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (err_host) __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __hip_atomic_store(q, kYPendingBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(float, bits);
}

struct SacRowsBArgs {
  SacMlp3 actor, target[2];
  float* dz1[2]; float* dz2[2];          // the online critics' unscaled gradients (scaled here)
  int H1c, H2c;
  const float* next_state; int ld_next;
  const float* noise; int ld_noise;
  const float* low; const float* high;
  const float* alpha_in;
  const float* reward; const uint8_t* term; float gamma;
  float noise_clip;                      // HEAD 1 (DDPG / TD3): target-policy smoothing clamp
  const float* q[2];
  float* dq[2];                          // [B]
  int B, S, A;
  SacTicket tk;
  // SPLIT instantiation: the second target critic runs in a helper workgroup (grid 2 x tiles); the
  // next action goes over and q2' comes back through the exchange words of SacRowsAArgs
  float* xact; float* xres;
  int* err; int* err_host;
  long long* prof;
  // SPLIT instantiation, optional third role (grid 3 x tiles): the actor's forward on the NEXT
  // step's states — it depends on nothing this launch or the critics' optimizer launch behind it
  // change — with the activations kept and the head written to pre_head (SacRowsAArgs::pre_head)
  const float* pre_state; int ld_pre;
  float* pre_head;
  unsigned* pre_mask;
};

// phase stamps (tools/prof_sac.py): 32 slots per wave, 100 MHz wall clock; null outside the tool
#define SR_STAMP(prof, wg, i)                                                            \
  do {                                                                                   \
    if ((prof) && (threadIdx.x & 63) == 0)                                               \
      (prof)[((int64_t)(wg) * 8 + (threadIdx.x >> 6)) * 32 + (i)] = (long long)wall_clock64(); \
  } while (0)

constexpr int SR_HEADP = 36;     // LDS pitch of 32-wide row vectors (head, head gradient, partials)
constexpr int SR_DHP = 68;       // pitch of the head-gradient tile when it is a GEMM operand
constexpr int SR_CST = 776;      // per network: b1[256] | b2[256] | w3 or b3[256] | critic b3 | pad

// LDS: xs [16][P0] | hA hB hC [16][PH] | red [8][16][SR_HEADP] | dhS | qred | small | cst[3]
__host__ __device__ inline size_t sac_rows_smem_floats(int k0) {
  return (size_t)RP_ROWS * (rp_pad(k0) + 3 * row_hid_pitch()) + 8 * RP_ROWS * SR_HEADP +
         RP_ROWS * SR_DHP + 8 * RP_ROWS + 8 * RP_ROWS + 3 * SR_CST;
}

// ---- H2 instantiations: the 256 x 256 GEMMs on the fp16 matrix pipe at fp32 accuracy ----------------
// online_f16_kernel.hpp's scheme (operands scaled by exact powers of two, split into two fp16 terms,
// three partial products, fp32 accumulators per magnitude class) for every hidden GEMM of the two
// row kernels: the actor's and the critics' layer 2, the critics' G = s2 W2, the actor's
// d z1 = d z2 W2.  First layers, heads and the narrow products stay on the fp32 pipe.  Hidden widths
// are exactly 256.  Extra LDS behind the fp32 layout:
//   usc [3][256]   2^(141 - e_n) per row n of W2, one array per network slot (the cst slots)
//   pA  hi | lo    [16][HF_PITCH] halves: h1, or the actor's scaled d z2 (consumer layout)
//   pB  hi | lo    the critics' scaled s2
//   rmaxw [8][16]  per-wave row maxima
constexpr int SR_USC = 256;
__host__ __device__ inline size_t sac_rows_h2_smem_bytes(int k0) {
  return sac_rows_smem_floats(k0) * sizeof(float) + sizeof(float) * 3 * SR_USC +
         sizeof(_Float16) * 4 * RP_ROWS * HF_PITCH + sizeof(unsigned) * 8 * RP_ROWS;
}
struct SrH2 {
  float* usc;                 // slot 0
  _Float16 *pAh, *pAl, *pBh, *pBl;
  unsigned* rmaxw;
};
__device__ __forceinline__ SrH2 sr_h2_carve(float* smem, int k0) {
  SrH2 X;
  X.usc = smem + sac_rows_smem_floats(k0);
  X.pAh = reinterpret_cast<_Float16*>(X.usc + 3 * SR_USC);
  X.pAl = X.pAh + RP_ROWS * HF_PITCH;
  X.pBh = X.pAl + RP_ROWS * HF_PITCH;
  X.pBl = X.pBh + RP_ROWS * HF_PITCH;
  X.rmaxw = reinterpret_cast<unsigned*>(X.pBl + RP_ROWS * HF_PITCH);
  return X;
}
// row maxima -> unit scales in LDS: request (with the other start-up loads), store later
__device__ __forceinline__ unsigned sr_usc_load(const unsigned* um, int tid) {
  return __builtin_bit_cast(unsigned, ld_or_zero(reinterpret_cast<const float*>(um), tid, tid < SR_USC));
}
__device__ __forceinline__ void sr_usc_store(float* usc, unsigned v, int tid) {
  if (tid < SR_USC) usc[tid] = h2_scale(h2_field(v));
}
__device__ __forceinline__ int sr_field_of(float scale) { return 268 - (int)(__float_as_uint(scale) >> 23); }
__device__ __forceinline__ float sr_inv(float scale) { return __uint_as_float((254u << 23) - __float_as_uint(scale)); }
// this lane's view of a network's unit scales (online_f16_kernel.hpp's HalfFields, read from LDS)
__device__ __forceinline__ HalfFields sr_fields(const float* usc, int wave, int r16, int qd) {
  HalfFields f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f.sA[t] = usc[32 * wave + 16 * t + r16];
    const float4 v = *reinterpret_cast<const float4*>(usc + 32 * wave + 16 * t + 4 * qd);
    f.fC[t][0] = sr_field_of(v.x); f.fC[t][1] = sr_field_of(v.y);
    f.fC[t][2] = sr_field_of(v.z); f.fC[t][3] = sr_field_of(v.w);
  }
  return f;
}
// (the two halves on their own: sA is live across the GEMM, fC only behind it)
__device__ __forceinline__ void sr_fields_A(float (&sA)[2], const float* usc, int wave, int r16) {
  sA[0] = usc[32 * wave + r16];
  sA[1] = usc[32 * wave + 16 + r16];
}
__device__ __forceinline__ void sr_fields_C(int (&fC)[2][4], const float* usc, int wave, int qd) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(usc + 32 * wave + 16 * t + 4 * qd);
    fC[t][0] = sr_field_of(v.x); fC[t][1] = sr_field_of(v.y);
    fC[t][2] = sr_field_of(v.z); fC[t][3] = sr_field_of(v.w);
  }
}
// the row's maximum across the eight waves: per-wave maxima to LDS (the caller's barrier follows)
__device__ __forceinline__ void sr_rowmax_put(unsigned m, unsigned* rmaxw, int wave, int r16, int qd) {
  m = umaxu(m, (unsigned)__shfl_xor((int)m, 16));
  m = umaxu(m, (unsigned)__shfl_xor((int)m, 32));
  if (qd == 0) rmaxw[wave * RP_ROWS + r16] = m;
}
__device__ __forceinline__ int sr_rowmax_field(const unsigned* rmaxw, int r16) {
  unsigned m = rmaxw[r16];
#pragma unroll
  for (int w = 1; w < 8; ++w) m = umaxu(m, rmaxw[w * RP_ROWS + r16]);
  return h2_field(m);
}
// this lane's eight values (units u0 + 16 t + j of its row: k-step `wave` of the consumer) -> planes
__device__ __forceinline__ void sr_planes_put(const float4 (&v)[2], float scale, _Float16* ph, _Float16* pl,
                                              int r16, int qd, int wave) {
  f16x8 hi, lo;
  h2_split8(v[0], v[1], scale, hi, lo);
  const int off = r16 * HF_PITCH + 8 * qd + 32 * wave;
  *reinterpret_cast<f16x8*>(ph + off) = hi;
  *reinterpret_cast<f16x8*>(pl + off) = lo;
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

// Biases and the critics' last-layer rows go through LDS once per workgroup: read from global
// memory right before the GEMM they seed, each cost an exposed round trip (~1 us) per layer.
// Two steps — request into registers, store to LDS — so that a kernel can put ALL its start-up
// requests (input tile, constants of every network) in flight before it waits for the first.
struct SrConsts {
  float v[2];
  float b3;
};
__device__ __forceinline__ void sr_consts_load(SrConsts& c, const SacMlp3& n, bool critic, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + it * 512;
    const int which = e >> 8, u = e & 255;
    const float b1 = ld_or_zero(n.b1, u, which == 0 && u < n.H1);
    const float b2 = ld_or_zero(n.b2, u, which == 1 && u < n.H2);
    const float w3 = ld_or_zero(critic ? n.w3 : n.b3, u, which == 2 && u < (critic ? n.H2 : n.DO));
    c.v[it] = which == 0 ? b1 : (which == 1 ? b2 : w3);     // (e >= 768: all three read as zero)
  }
  c.b3 = ld_or_zero(n.b3, 0, critic && tid == 0);
}
__device__ __forceinline__ void sr_consts_store(float* cst, const SrConsts& c, int tid) {
  cst[tid] = c.v[0];
  if (tid + 512 < 768) cst[tid + 512] = c.v[1];
  if (tid == 0) cst[768] = c.b3;
}
// the input tile: at most SR_XV float4 per thread in registers (wider inputs: a plain loop)
constexpr int SR_XV = 2;
struct SrTile {
  float4 v[SR_XV];
};
__device__ __forceinline__ bool sr_tile_fits(int P0) { return RP_ROWS * ((P0 - 4) >> 2) <= 512 * SR_XV; }
__device__ __forceinline__ void sr_tile_load(SrTile& t, const float* x, int ldx, int S, int m0, int B,
                                             int P0, int tid) {
  const int c4 = (P0 - 4) >> 2;
  const bool vx = is_vec_ok(x, ldx) && ((S & 3) == 0);
#pragma unroll
  for (int it = 0; it < SR_XV; ++it) {
    const int e = tid + it * 512;
    const int r = e / c4, c = (e - r * c4) * 4;
    const bool ok = e < RP_ROWS * c4 && (m0 + r) < B;
    if (vx) t.v[it] = ld4_or_zero(x, (int64_t)(m0 + r) * ldx + c, ok && c < S);
    else t.v[it] = guarded_load4(x, (int64_t)(m0 + r) * ldx, ok, c, S);
  }
}
__device__ __forceinline__ void sr_tile_store(const SrTile& t, float* xs, int P0, int tid) {
  const int c4 = (P0 - 4) >> 2;
#pragma unroll
  for (int it = 0; it < SR_XV; ++it) {
    const int e = tid + it * 512;
    const int r = e / c4, c = (e - r * c4) * 4;
    if (e < RP_ROWS * c4) *reinterpret_cast<float4*>(xs + r * P0 + c) = t.v[it];
  }
}
__device__ __forceinline__ void sr_bias(f32x4v (&acc)[2], const float* cb, int u0) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(cb + u0 + 16 * t);
    acc[t][0] = v.x; acc[t][1] = v.y; acc[t][2] = v.z; acc[t][3] = v.w;
  }
}
__device__ __forceinline__ void sr_zero(f32x4v (&acc)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
}

// ---- weight streams ------------------------------------------------------------------------------
// Hidden layers (NGH k-groups, unrolled): the ring is filled with the first k-groups of the NEXT
// hidden GEMM's stream inside the last iterations of the current one (the slot a k-group was read
// from is refilled at once — rows16_gemm_static_pf with both rings the same registers), so a GEMM
// never starts on a cold ring.  First layers (NG1 <= 8 k-groups): their own small ring, filled a
// whole phase early.  NGH / NG1 = 0: run-time loops, weights requested when the GEMM starts.
template <int NGH>
__device__ __forceinline__ void sr_prefetch(WRing& R, const float* Wf, int tile0, int nt, int lane) {
  if constexpr (NGH > 0) ring_fill<NGH>(R, Wf, tile0, nt, lane);
}
template <int NGH>
__device__ __forceinline__ void sr_gemm(f32x4v (&acc)[2], WRing& R, const float* Wf, int nkg,
                                        int tile0, int nt, const float* actp, int lane,
                                        const float* Wnext, int nt_next) {
  if constexpr (NGH > 0)
    rows16_gemm_static_pf<NGH, NGH>(acc, R, Wf, tile0, nt, actp, lane, R, Wnext, nt_next,
                                    Wnext != nullptr);
  else rows16_gemm<4>(acc, Wf, nkg, tile0, nt, actp, lane);
}
template <int NG1>
__device__ __forceinline__ void sr_l1_fill(WRing& R1, const float* Wf, int tile0, int nt, int lane) {
  if constexpr (NG1 > 0) ring_fill<NG1>(R1, Wf, tile0, nt, lane);
}
template <int NG1>
__device__ __forceinline__ void sr_l1_gemm(f32x4v (&acc)[2], WRing& R1, const float* Wf, int nkg,
                                           int tile0, int nt, const float* actp, int lane) {
  static_assert(NG1 <= RP_PD, "a first layer's k-groups must fit the ring");
  if constexpr (NG1 > 0) rows16_gemm_static<NG1>(acc, R1, Wf, tile0, nt, actp, lane);
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
// k-groups w and w + 8 (a hidden layer has at most 16), the partial tiles go to red[wave].  The
// four weight fragments are requested ahead of time (sr_narrow_load).
struct SrNarrowW {
  float4 w00, w10, w01, w11;
};
__device__ __forceinline__ void sr_narrow_load(SrNarrowW& w, const float* Wf, int nkg, int t_lo,
                                               int ntl, const SrLane& L) {
  const int g0 = L.wave, g1 = L.wave + 8;
  const bool k0 = g0 < nkg, k1 = g1 < nkg;
  const int64_t b0 = ((int64_t)t_lo * nkg) * 256 + L.lane * 4;
  const int64_t b1 = b0 + (int64_t)nkg * 256;
  w.w00 = ld4_or_zero(Wf, b0 + (int64_t)g0 * 256, k0 && ntl > 0);
  w.w10 = ld4_or_zero(Wf, b1 + (int64_t)g0 * 256, k0 && ntl > 1);
  w.w01 = ld4_or_zero(Wf, b0 + (int64_t)g1 * 256, k1 && ntl > 0);
  w.w11 = ld4_or_zero(Wf, b1 + (int64_t)g1 * 256, k1 && ntl > 1);
}
__device__ __forceinline__ void sr_narrow_mma(const SrNarrowW& w, int nkg, const float* actp,
                                              float* red, const SrLane& L) {
  f32x4v acc[2];
  sr_zero(acc);
  const int g0 = L.wave, g1 = L.wave + 8;
  if (g0 < nkg) {
    const float4 x4 = *reinterpret_cast<const float4*>(actp + g0 * 16);
    acc[0] = mfma16(w.w00.x, x4.x, acc[0]); acc[1] = mfma16(w.w10.x, x4.x, acc[1]);
    acc[0] = mfma16(w.w00.y, x4.y, acc[0]); acc[1] = mfma16(w.w10.y, x4.y, acc[1]);
    acc[0] = mfma16(w.w00.z, x4.z, acc[0]); acc[1] = mfma16(w.w10.z, x4.z, acc[1]);
    acc[0] = mfma16(w.w00.w, x4.w, acc[0]); acc[1] = mfma16(w.w10.w, x4.w, acc[1]);
  }
  if (g1 < nkg) {
    const float4 x4 = *reinterpret_cast<const float4*>(actp + g1 * 16);
    acc[0] = mfma16(w.w01.x, x4.x, acc[0]); acc[1] = mfma16(w.w11.x, x4.x, acc[1]);
    acc[0] = mfma16(w.w01.y, x4.y, acc[0]); acc[1] = mfma16(w.w11.y, x4.y, acc[1]);
    acc[0] = mfma16(w.w01.z, x4.z, acc[0]); acc[1] = mfma16(w.w11.z, x4.z, acc[1]);
    acc[0] = mfma16(w.w01.w, x4.w, acc[0]); acc[1] = mfma16(w.w11.w, x4.w, acc[1]);
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

// Wide output, at most two k-groups (the actor's d h2 = d head W3): fragments requested early
struct SrSmallW {
  float4 a0, a1, b0, b1;     // tile0 / tile0 + 1, k-group 0 / 1
};
__device__ __forceinline__ void sr_small_load(SrSmallW& w, const float* Wf, int nkg, int tile0,
                                              int nt, int lane) {
  const int64_t base0 = ((int64_t)tile0 * nkg) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)nkg * 256;
  w.a0 = ld4_or_zero(Wf, base0, tile0 < nt);
  w.b0 = ld4_or_zero(Wf, base1, tile0 + 1 < nt);
  w.a1 = ld4_or_zero(Wf, base0 + 256, tile0 < nt && nkg > 1);
  w.b1 = ld4_or_zero(Wf, base1 + 256, tile0 + 1 < nt && nkg > 1);
}
__device__ __forceinline__ void sr_small_gemm(f32x4v (&acc)[2], const SrSmallW& w, int nkg,
                                              const float* actp) {
  {
    const float4 x4 = *reinterpret_cast<const float4*>(actp);
    acc[0] = mfma16(w.a0.x, x4.x, acc[0]); acc[1] = mfma16(w.b0.x, x4.x, acc[1]);
    acc[0] = mfma16(w.a0.y, x4.y, acc[0]); acc[1] = mfma16(w.b0.y, x4.y, acc[1]);
    acc[0] = mfma16(w.a0.z, x4.z, acc[0]); acc[1] = mfma16(w.b0.z, x4.z, acc[1]);
    acc[0] = mfma16(w.a0.w, x4.w, acc[0]); acc[1] = mfma16(w.b0.w, x4.w, acc[1]);
  }
  if (nkg > 1) {
    const float4 x4 = *reinterpret_cast<const float4*>(actp + 16);
    acc[0] = mfma16(w.a1.x, x4.x, acc[0]); acc[1] = mfma16(w.b1.x, x4.x, acc[1]);
    acc[0] = mfma16(w.a1.y, x4.y, acc[0]); acc[1] = mfma16(w.b1.y, x4.y, acc[1]);
    acc[0] = mfma16(w.a1.z, x4.z, acc[0]); acc[1] = mfma16(w.b1.z, x4.z, acc[1]);
    acc[0] = mfma16(w.a1.w, x4.w, acc[0]); acc[1] = mfma16(w.b1.w, x4.w, acc[1]);
  }
}

// state (or next state) tile -> xs[:, 0:S], zeros up to the pitch
__device__ __forceinline__ void sr_stage(const float* x, int ldx, int S, int m0, int B, float* xs,
                                         int P0, int tid) {
  const int c4 = (P0 - 4) >> 2;
  const bool vx = is_vec_ok(x, ldx) && ((S & 3) == 0);
  for (int e = tid; e < RP_ROWS * c4; e += 512) {
    const int r = e / c4, c = (e - r * c4) * 4;
    const bool ok = (m0 + r) < B;
    float4 v;
    if (vx) v = ld4_or_zero(x, (int64_t)(m0 + r) * ldx + c, ok && c < S);
    else v = guarded_load4(x, (int64_t)(m0 + r) * ldx, ok, c, S);
    *reinterpret_cast<float4*>(xs + r * P0 + c) = v;
  }
}

// What a network pass requests for whoever runs next
struct SrNext {
  const float* W1; int nt1;      // first-layer stream for the small ring (after this pass's layer 1)
  const float* Wh; int nth;      // hidden stream for the ring (inside this pass's last hidden GEMM)
};

// One critic on the tile in xs: q, and (WANT_G) the unit gradients  s2 = [h2 > 0] w3 and
// Gm = (s2 W2) [h1 > 0] — Gm stays in hC.  KEEP: h1, h2, s2, Gm also go to the network's kept
// buffers.  Opens with a barrier (xs complete, hA / hB / hC free); the caller puts a barrier
// between this and its first read of hC / qred.  On entry R1 holds W1f and R holds W2f's first
// k-groups (static instantiations).
template <int NGH, int NG1, bool WANT_G, bool KEEP, bool EARLY1 = true>
__device__ __forceinline__ void sr_critic(const SacMlp3& n, const float* cst, const float* xs,
                                          int P0, float* hA, float* hB, float* hC, float* qred,
                                          WRing& R, WRing& R1, const SrLane& L, int64_t row,
                                          bool rok, const SrNext& nx, long long* prof, int wg,
                                          int slot) {
  const int PH = row_hid_pitch();
  const int nt1 = (n.H1 + 15) >> 4, nt2 = (n.H2 + 15) >> 4;
  f32x4v acc[2];
  // ---- layer 1 (the constants may have been staged by other threads just now: read them
  // after the barrier)
  __syncthreads();
  sr_bias(acc, cst, L.u0);
  sr_l1_gemm<NG1>(acc, R1, n.W1f, wf16_nkg(n.K0), L.tile0, nt1, xs + L.r16 * P0 + 4 * L.qd, L.lane);
  // the next pass's first-layer stream: at once, or (register pressure) after this pass's GEMMs
  if (EARLY1 && nx.W1) sr_l1_fill<NG1>(R1, nx.W1, L.tile0, nx.nt1, L.lane);
  const unsigned m1 = sr_relu_out(acc, hA, PH, L, KEEP ? n.act1 : nullptr, row, n.H1, rok);
  SR_STAMP(prof, wg, slot);
  // ---- layer 2, the head's dot product, s2
  sr_bias(acc, cst + 256, L.u0);
  float4 w3v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) w3v[t] = *reinterpret_cast<const float4*>(cst + 512 + L.u0 + 16 * t);
  __syncthreads();
  sr_gemm<NGH>(acc, R, n.W2f, wf16_nkg(n.H1), L.tile0, nt2, hA + L.r16 * PH + 4 * L.qd, L.lane,
               WANT_G ? n.W2tf : nx.Wh, WANT_G ? nt1 : nx.nth);
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
    if (KEEP && rok && n.act2) store4_guarded(n.act2, row * n.H2, u, n.H2, (n.H2 & 3) == 0, h);
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
    sr_gemm<NGH>(acc, R, n.W2tf, wf16_nkg(n.H2), L.tile0, nt1, hB + L.r16 * PH + 4 * L.qd, L.lane,
                 nx.Wh, nx.nth);
    sr_mask_out(acc, m1, hC, PH, L, KEEP ? n.dz1 : nullptr, row, n.H1, rok);
    SR_STAMP(prof, wg, slot + 2);
  }
  if (!EARLY1 && nx.W1) sr_l1_fill<NG1>(R1, nx.W1, L.tile0, nx.nt1, L.lane);
}
// sr_critic with layer 2 and G on the fp16 matrix pipe (H1 = H2 = 256).  `usc`: this network's unit
// scales (staged before the opening barrier).  h1 lives in the A planes, the scaled s2 in the B
// planes; Gm still goes to hC as fp32 (the narrow product behind it is an fp32 one).
template <int NG1, bool WANT_G, bool KEEP, bool EARLY1 = true>
__device__ __forceinline__ void sr_critic_h2(const SacMlp3& n, const float* cst, const float* usc,
                                             const float* xs, int P0, const SrH2& X, float* hC,
                                             float* qred, WRing& R, WRing& R1, const SrLane& L,
                                             int64_t row, bool rok, const SrNext& nx, long long* prof,
                                             int wg, int slot) {
  const int PH = row_hid_pitch();
  const int plane_off = L.r16 * HF_PITCH + 8 * L.qd;
  f32x4v acc[2];
  __syncthreads();
  sr_bias(acc, cst, L.u0);
  sr_l1_gemm<NG1>(acc, R1, n.W1f, wf16_nkg(n.K0), L.tile0, 16, xs + L.r16 * P0 + 4 * L.qd, L.lane);
  if (EARLY1 && nx.W1) sr_l1_fill<NG1>(R1, nx.W1, L.tile0, nx.nt1, L.lane);
  // ---- h1: kept in registers until its row maximum is known
  float4 h1k[2];
  unsigned m1 = 0, hm = 0u;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = L.u0 + 16 * t;
    h1k[t] = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]), relu_keep_nan(acc[t][2]),
                         relu_keep_nan(acc[t][3]));
    m1 |= (h1k[t].x > 0.f ? 1u : 0u) << (4 * t) | (h1k[t].y > 0.f ? 2u : 0u) << (4 * t) |
          (h1k[t].z > 0.f ? 4u : 0u) << (4 * t) | (h1k[t].w > 0.f ? 8u : 0u) << (4 * t);
    hm = umaxu(hm, umax4(h1k[t]));
    if (KEEP && rok && n.act1) store4_guarded(n.act1, row * n.H1, u, n.H1, true, h1k[t]);
  }
  sr_rowmax_put(hm, X.rmaxw, L.wave, L.r16, L.qd);
  float sA2[2];
  sr_fields_A(sA2, usc, L.wave, L.r16);
  // the backward product's B operand is s2[n] 2^-e_n = [h2 > 0] w3[n] 2^(e_n - 141); its own scale
  // comes from max |w3[n] 2^(e_n - 141)|: every wave forms all 256 (64 lanes x 4)
  int f3 = 0;
  if (WANT_G) {
    const float4 w3a = *reinterpret_cast<const float4*>(cst + 512 + 4 * L.lane);
    const float4 ua = *reinterpret_cast<const float4*>(usc + 4 * L.lane);
    unsigned m = umax4(make_float4(w3a.x * sr_inv(ua.x), w3a.y * sr_inv(ua.y), w3a.z * sr_inv(ua.z),
                                   w3a.w * sr_inv(ua.w)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = umaxu(m, (unsigned)__shfl_xor((int)m, o));
    f3 = h2_field(m);
  }
  SR_STAMP(prof, wg, slot);
  __syncthreads();                                                        // row maxima
  const int fh = sr_rowmax_field(X.rmaxw, L.r16);
  sr_planes_put(h1k, h2_scale(fh), X.pAh, X.pAl, L.r16, L.qd, L.wave);
  __syncthreads();                                                        // h1 planes
  // ---- layer 2, the head's dot product, s2
  int fC2[2][4];
  {
    f32x4v c[2][HF_NACC];
    h2_zero(c);
    rows16_gemm_h2<16, 16, false>(c, R, n.W2f, L.tile0, L.lane, sA2, nullptr, X.pAh + plane_off,
                                  X.pAl + plane_off, R, WANT_G ? n.W2tf : nx.Wh,
                                  WANT_G ? true : nx.Wh != nullptr);
    sr_fields_C(fC2, usc, L.wave, L.qd);
    float4 b2v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) b2v[t] = *reinterpret_cast<const float4*>(cst + 256 + L.u0 + 16 * t);
    h2_finish(acc, c, fh, fC2, b2v);
  }
  float4 w3v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) w3v[t] = *reinterpret_cast<const float4*>(cst + 512 + L.u0 + 16 * t);
  float qp = 0.f;
  float4 z[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = L.u0 + 16 * t;
    const float4 h = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]),
                                 relu_keep_nan(acc[t][2]), relu_keep_nan(acc[t][3]));
    qp += h.x * w3v[t].x;
    qp += h.y * w3v[t].y;
    qp += h.z * w3v[t].z;
    qp += h.w * w3v[t].w;
    if (KEEP && rok && n.act2) store4_guarded(n.act2, row * n.H2, u, n.H2, true, h);
    if (WANT_G) {
      const float4 s2 = make_float4(h.x > 0.f ? w3v[t].x : 0.f, h.y > 0.f ? w3v[t].y : 0.f,
                                    h.z > 0.f ? w3v[t].z : 0.f, h.w > 0.f ? w3v[t].w : 0.f);
      if (KEEP && rok) store4_guarded(n.dz2, row * n.H2, u, n.H2, true, s2);
      auto inv = [](int field) { return __uint_as_float((unsigned)(field - 14) << 23); };
      z[t] = make_float4(s2.x * inv(fC2[t][0]), s2.y * inv(fC2[t][1]), s2.z * inv(fC2[t][2]),
                         s2.w * inv(fC2[t][3]));
    }
  }
  if (WANT_G) sr_planes_put(z, h2_scale(f3), X.pBh, X.pBl, L.r16, L.qd, L.wave);
  qp += __shfl_xor(qp, 16);
  qp += __shfl_xor(qp, 32);
  if (L.qd == 0) qred[L.wave * RP_ROWS + L.r16] = qp;
  SR_STAMP(prof, wg, slot + 1);
  if (WANT_G) {
    // ---- Gm = (s2 W2) [h1 > 0]: the weights' scale depends on the reduction index
    __syncthreads();
    f32x4v c[2][HF_NACC];
    h2_zero(c);
    const float unused[2] = {0.f, 0.f};
    rows16_gemm_h2<16, 16, true>(c, R, n.W2tf, L.tile0, L.lane, unused, usc + 4 * L.qd,
                                 X.pBh + plane_off, X.pBl + plane_off, R, nx.Wh, nx.Wh != nullptr);
    const float4 zero2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int fG[2][4] = {{141, 141, 141, 141}, {141, 141, 141, 141}};
    h2_finish(acc, c, f3, fG, zero2);
    sr_mask_out(acc, m1, hC, PH, L, KEEP ? n.dz1 : nullptr, row, n.H1, rok);
    SR_STAMP(prof, wg, slot + 2);
  }
  if (!EARLY1 && nx.W1) sr_l1_fill<NG1>(R1, nx.W1, L.tile0, nx.nt1, L.lane);
}
// q of row r from the eight waves' partial dot products (after a barrier)
__device__ __forceinline__ float sr_q(const float* qred, int r, const float* cst) {
  float q = cst[768];
#pragma unroll
  for (int w = 0; w < 8; ++w) q += qred[w * RP_ROWS + r];
  return q;
}

// The actor on the tile in xs[:, 0:S]: head [16][2A] summed into headS (pitch SR_HEADP).  KEEP: the
// hidden activations also go to act1 / act2 and the ReLU masks are returned.  Opens with a
// barrier; closes with the barrier after which headS is complete.  On entry R1 holds W1f, R holds
// W2f.
template <int NGH, int NG1, int NG1N, bool KEEP>
__device__ __forceinline__ void sr_actor_fwd(const SacMlp3& n, const float* cst, const float* xs,
                                             int P0, float* hA, float* hB, float* red, float* headS,
                                             WRing& R, WRing& R1, const SrLane& L, int64_t row,
                                             bool rok, unsigned& m1, unsigned& m2,
                                             const SrNext& nx) {
  const int PH = row_hid_pitch();
  const int nt1 = (n.H1 + 15) >> 4, nt2 = (n.H2 + 15) >> 4;
  f32x4v acc[2];
  __syncthreads();
  sr_bias(acc, cst, L.u0);
  sr_l1_gemm<NG1>(acc, R1, n.W1f, wf16_nkg(n.K0), L.tile0, nt1, xs + L.r16 * P0 + 4 * L.qd, L.lane);
  // the head's fragments (needed after layer 2) and, after layer 2, the next pass's first layer
  SrNarrowW hw;
  sr_narrow_load(hw, n.W3f, wf16_nkg(n.H2), 0, (n.DO + 15) >> 4, L);
  m1 = sr_relu_out(acc, hA, PH, L, KEEP ? n.act1 : nullptr, row, n.H1, rok);
  sr_bias(acc, cst + 256, L.u0);
  __syncthreads();
  sr_gemm<NGH>(acc, R, n.W2f, wf16_nkg(n.H1), L.tile0, nt2, hA + L.r16 * PH + 4 * L.qd, L.lane,
               nx.Wh, nx.nth);
  if (nx.W1) sr_l1_fill<NG1N>(R1, nx.W1, L.tile0, nx.nt1, L.lane);
  m2 = sr_relu_out(acc, hB, PH, L, KEEP ? n.act2 : nullptr, row, n.H2, rok);
  __syncthreads();
  sr_narrow_mma(hw, wf16_nkg(n.H2), hB + L.r16 * PH + 4 * L.qd, red, L);
  __syncthreads();
  if (L.tid < RP_ROWS * 32) {
    const int r = L.tid >> 5, c = L.tid & 31;
    headS[r * SR_HEADP + c] = c < n.DO ? sr_narrow_get(red, r, c) + cst[512 + c] : 0.f;
  }
  __syncthreads();
}

// sr_actor_fwd with layer 2 on the fp16 matrix pipe (H1 = H2 = 256); h2 goes to hB as fp32 (the
// head's narrow product is an fp32 one).
template <int NG1, int NG1N, bool KEEP>
__device__ __forceinline__ void sr_actor_fwd_h2(const SacMlp3& n, const float* cst, const float* usc,
                                                const float* xs, int P0, const SrH2& X, float* hB,
                                                float* red, float* headS, WRing& R, WRing& R1,
                                                const SrLane& L, int64_t row, bool rok, unsigned& m1,
                                                unsigned& m2, const SrNext& nx) {
  const int PH = row_hid_pitch();
  const int plane_off = L.r16 * HF_PITCH + 8 * L.qd;
  f32x4v acc[2];
  __syncthreads();
  sr_bias(acc, cst, L.u0);
  sr_l1_gemm<NG1>(acc, R1, n.W1f, wf16_nkg(n.K0), L.tile0, 16, xs + L.r16 * P0 + 4 * L.qd, L.lane);
  SrNarrowW hw;
  sr_narrow_load(hw, n.W3f, wf16_nkg(n.H2), 0, (n.DO + 15) >> 4, L);
  float4 h1k[2];
  unsigned hm = 0u;
  m1 = 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = L.u0 + 16 * t;
    h1k[t] = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]), relu_keep_nan(acc[t][2]),
                         relu_keep_nan(acc[t][3]));
    m1 |= (h1k[t].x > 0.f ? 1u : 0u) << (4 * t) | (h1k[t].y > 0.f ? 2u : 0u) << (4 * t) |
          (h1k[t].z > 0.f ? 4u : 0u) << (4 * t) | (h1k[t].w > 0.f ? 8u : 0u) << (4 * t);
    hm = umaxu(hm, umax4(h1k[t]));
    if (KEEP && rok && n.act1) store4_guarded(n.act1, row * n.H1, u, n.H1, true, h1k[t]);
  }
  sr_rowmax_put(hm, X.rmaxw, L.wave, L.r16, L.qd);
  float sA2[2];
  sr_fields_A(sA2, usc, L.wave, L.r16);
  __syncthreads();                                                        // row maxima
  const int fh = sr_rowmax_field(X.rmaxw, L.r16);
  sr_planes_put(h1k, h2_scale(fh), X.pAh, X.pAl, L.r16, L.qd, L.wave);
  __syncthreads();                                                        // h1 planes
  {
    f32x4v c[2][HF_NACC];
    h2_zero(c);
    rows16_gemm_h2<16, 16, false>(c, R, n.W2f, L.tile0, L.lane, sA2, nullptr, X.pAh + plane_off,
                                  X.pAl + plane_off, R, nx.Wh, nx.Wh != nullptr);
    int fC2[2][4];
    sr_fields_C(fC2, usc, L.wave, L.qd);
    float4 b2v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) b2v[t] = *reinterpret_cast<const float4*>(cst + 256 + L.u0 + 16 * t);
    h2_finish(acc, c, fh, fC2, b2v);
  }
  if (nx.W1) sr_l1_fill<NG1N>(R1, nx.W1, L.tile0, nx.nt1, L.lane);
  m2 = sr_relu_out(acc, hB, PH, L, KEEP ? n.act2 : nullptr, row, n.H2, rok);
  __syncthreads();
  sr_narrow_mma(hw, wf16_nkg(n.H2), hB + L.r16 * PH + 4 * L.qd, red, L);
  __syncthreads();
  if (L.tid < RP_ROWS * 32) {
    const int r = L.tid >> 5, c = L.tid & 31;
    headS[r * SR_HEADP + c] = c < n.DO ? sr_narrow_get(red, r, c) + cst[512 + c] : 0.f;
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

// Per-tile loss partial: rowsum[0..15] (LDS) added in row order.  The launch's total is formed by
// sac_finish_kernel at the end of the step — a ticket + fence per workgroup and a serial sum in the
// last one cost ~2 us on every tile's critical path and ~9 us on the last.
__device__ __forceinline__ void sr_tile_partial(const float* rowsum, float* partials, unsigned tile) {
  if (threadIdx.x == 0) {
    float p = 0.f;
#pragma unroll
    for (int r = 0; r < RP_ROWS; ++r) p += rowsum[r];
    partials[tile] = p;
  }
}

// HEAD 0: tanh-Gaussian policy, twin-critic actor loss (continuous SAC).
// HEAD 1: deterministic tanh policy, actor loss -mean Q1(s, pi(s)) (DDPG / TD3, ddpg.py:106-121):
//         the head is [A] wide, no log-probability, one critic pass, d loss / d q = -1/B.
template <int NGH, int NGA, int NGC, int HEAD, bool SPLIT = false, bool H2 = false>
__global__ __launch_bounds__(512) void sac_rows_a_kernel(SacRowsAArgs a) {
  static_assert(!H2 || (NGH == 16 && HEAD == 0), "H2: hidden 256, continuous SAC");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SrLane L = sr_lane();
  const int W = a.S + a.A;
  const int P0 = rp_pad(W), PH = row_hid_pitch();
  SrH2 X;
  if constexpr (H2) X = sr_h2_carve(smem, W);
  float* xs = smem;
  float* hA = xs + RP_ROWS * P0;
  float* hB = hA + RP_ROWS * PH;
  float* hC = hB + RP_ROWS * PH;
  float* red = hC + RP_ROWS * PH;                  // [8][16][SR_HEADP]
  float* dhS = red + 8 * RP_ROWS * SR_HEADP;       // [16][SR_DHP]
  float* qred = dhS + RP_ROWS * SR_DHP;            // [8][16]
  float* small = qred + 8 * RP_ROWS;               // [8][16]: q1 q2 logp lossrow
  float* cst = small + 8 * RP_ROWS;                // [3][SR_CST]: actor, critic 1, critic 2
  // roles: 0 actor update rows, 1 / 2 critic c at (s, a_batch), 3 (split) critic 2 at (s, pi(s))
  constexpr bool split = HEAD == 0 && SPLIT;
  int tile = (int)blockIdx.x, role = (int)blockIdx.y + (a.actor_rows ? 0 : 1);
  int wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (split) {
    // the long roles first in dispatch order: actor rows, helper, then the two short ones
    const int r4 = (int)blockIdx.x & 3;
    tile = (int)blockIdx.x >> 2;
    role = r4 == 0 ? 0 : (r4 == 1 ? 3 : r4 - 1);
    wg = role * (int)(gridDim.x >> 2) + tile;
  }
  const int m0 = tile * RP_ROWS;
  const int64_t row = m0 + L.r16;
  const bool rok = row < a.B;
  WRing R, R1;
  SR_STAMP(a.prof, wg, 0);

  if constexpr (split) {
    if (role == 3) {
      // ------------------------------------------------- helper: critic 2 at (s, the fresh action)
      const SacMlp3& n = a.critic[1];
      sr_l1_fill<NGC>(R1, n.W1f, L.tile0, (n.H1 + 15) >> 4, L.lane);
      sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
      const int sr = L.tid / a.A, sj = L.tid - sr * a.A;
      const bool sok = L.tid < RP_ROWS * a.A;
      {
        SrTile xt;
        SrConsts kc;
        unsigned uv = 0u;
        const bool fits = sr_tile_fits(P0);
        if (fits) sr_tile_load(xt, a.state, a.ld_state, a.S, m0, a.B, P0, L.tid);
        sr_consts_load(kc, n, true, L.tid);
        if constexpr (H2) uv = sr_usc_load(n.um, L.tid);
        if (fits) sr_tile_store(xt, xs, P0, L.tid);
        else sr_stage(a.state, a.ld_state, a.S, m0, a.B, xs, P0, L.tid);
        sr_consts_store(cst, kc, L.tid);
        if constexpr (H2) sr_usc_store(X.usc, uv, L.tid);
      }
      const int t_lo = a.S >> 4;
      const int ntl = ((W + 15) >> 4) - t_lo;
      SrNarrowW gw;
      sr_narrow_load(gw, n.W1tf, wf16_nkg(n.H1), t_lo, ntl, L);
      SR_STAMP(a.prof, wg, 1);
      __syncthreads();   // the staged zeros of xs[:, S:] are down before the action lands on them
      if (sok) xs[sr * P0 + a.S + sj] = sr_take(a.xact + (int64_t)tile * SR_XACT + sr * 16 + sj, a.err, a.err_host);
      SR_STAMP(a.prof, wg, 3);
      SrNext none;
      none.W1 = nullptr; none.nt1 = 0; none.Wh = nullptr; none.nth = 0;
      if constexpr (H2)
        sr_critic_h2<NGC, true, false>(n, cst, X.usc, xs, P0, X, hC, qred, R, R1, L, row, rok, none,
                                       a.prof, wg, 4);
      else
        sr_critic<NGH, NGC, true, false>(n, cst, xs, P0, hA, hB, hC, qred, R, R1, L, row, rok, none,
                                         a.prof, wg, 4);
      __syncthreads();
      if (L.tid < RP_ROWS)
        publish_y(a.xres + (int64_t)tile * SR_XRES + L.tid * 17 + 16, sr_q(qred, L.tid, cst));
      sr_narrow_mma(gw, wf16_nkg(n.H1), hC + L.r16 * PH + 4 * L.qd, red, L);
      __syncthreads();
      if (sok)
        publish_y(a.xres + (int64_t)tile * SR_XRES + sr * 17 + sj, sr_narrow_get(red, sr, (a.S & 15) + sj));
      SR_STAMP(a.prof, wg, 15);
      return;
    }
  }
  if (role > 0) {
    // ---------------------------------------------------------------- critic c at (s, a_batch)
    const int c = role - 1;
    const SacMlp3& n = a.critic[c];
    sr_l1_fill<NGC>(R1, n.W1f, L.tile0, (n.H1 + 15) >> 4, L.lane);
    sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
    SrConsts kc;
    sr_consts_load(kc, n, true, L.tid);
    unsigned uv = 0u;
    if constexpr (H2) uv = sr_usc_load(n.um, L.tid);
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
          const float vs = ld_or_zero(a.state, (int64_t)(m0 + r) * a.ld_state + col, ok && col < a.S);
          const float va = ld_or_zero(a.action, (int64_t)(m0 + r) * a.ld_action + (col - a.S),
                                      ok && col >= a.S && col < W);
          v[k] = col < a.S ? vs : va;
        }
        *reinterpret_cast<float4*>(xs + r * P0 + cc) = make_float4(v[0], v[1], v[2], v[3]);
        if (c == 0 && ok) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (cc + k < W) a.xq[(int64_t)(m0 + r) * W + cc + k] = v[k];
        }
      }
    }
    sr_consts_store(cst, kc, L.tid);
    if constexpr (H2) sr_usc_store(X.usc, uv, L.tid);
    SR_STAMP(a.prof, wg, 1);
    SrNext none;
    none.W1 = nullptr; none.nt1 = 0; none.Wh = nullptr; none.nth = 0;
    if constexpr (H2)
      sr_critic_h2<NGC, true, true>(n, cst, X.usc, xs, P0, X, hC, qred, R, R1, L, row, rok, none,
                                    a.prof, wg, 4);
    else
      sr_critic<NGH, NGC, true, true>(n, cst, xs, P0, hA, hB, hC, qred, R, R1, L, row, rok, none,
                                      a.prof, wg, 4);
    __syncthreads();
    if (L.tid < RP_ROWS && m0 + L.tid < a.B) a.q[c][m0 + L.tid] = sr_q(qred, L.tid, cst);
    SR_STAMP(a.prof, wg, 15);
    return;
  }

  // ------------------------------------------------------------------ actor update rows
  const SacMlp3& n = a.actor;
  if (!(split && a.pre_head)) {
    sr_l1_fill<NGA>(R1, n.W1f, L.tile0, (n.H1 + 15) >> 4, L.lane);
    sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
  }
  // noise of this thread's (row, component), requested before anything else needs it
  const int sr = L.tid / a.A, sj = L.tid - sr * a.A;
  const bool sok = L.tid < RP_ROWS * a.A;
  const bool srok = sok && (m0 + sr) < a.B;
  const float eps = HEAD == 0 ? ld_or_zero(a.noise, (int64_t)(m0 + sr) * a.ld_noise + sj, srok) : 0.f;
  const float lo = ld_or_zero(a.low, sj, sok), hi = ld_or_zero(a.high, sj, sok);
  const float alpha = HEAD == 0 ? a.alpha[0] : 0.f;
  constexpr int NCRIT = (HEAD == 0 && !split) ? 2 : 1;   // split: the helper runs critic 2
  {
    // every start-up request in flight before the first wait
    SrTile xt;
    SrConsts k0, k1, k2;
    const bool fits = sr_tile_fits(P0);
    if (fits) sr_tile_load(xt, a.state, a.ld_state, a.S, m0, a.B, P0, L.tid);
    sr_consts_load(k0, n, false, L.tid);
    sr_consts_load(k1, a.critic[0], true, L.tid);
    if (NCRIT > 1) sr_consts_load(k2, a.critic[1], true, L.tid);
    unsigned u0v = 0u, u1v = 0u, u2v = 0u;
    if constexpr (H2) {
      u0v = sr_usc_load(n.um, L.tid);
      u1v = sr_usc_load(a.critic[0].um, L.tid);
      if (NCRIT > 1) u2v = sr_usc_load(a.critic[1].um, L.tid);
    }
    if (fits) sr_tile_store(xt, xs, P0, L.tid);
    else sr_stage(a.state, a.ld_state, a.S, m0, a.B, xs, P0, L.tid);
    sr_consts_store(cst, k0, L.tid);
    sr_consts_store(cst + SR_CST, k1, L.tid);
    if (NCRIT > 1) sr_consts_store(cst + 2 * SR_CST, k2, L.tid);
    if constexpr (H2) {
      sr_usc_store(X.usc, u0v, L.tid);
      sr_usc_store(X.usc + SR_USC, u1v, L.tid);
      if (NCRIT > 1) sr_usc_store(X.usc + 2 * SR_USC, u2v, L.tid);
    }
  }
  SR_STAMP(a.prof, wg, 1);
  unsigned m1a, m2a;
  float* headS = dhS;   // [16][SR_HEADP]; dhS proper is written only after the head was consumed
  if (split && a.pre_head) {
    // the forward ran in the previous step's sac_rows_b launch: head from memory, the ReLU masks
    // from the kept activations; the rings take the first critic's streams (what the forward
    // leaves in them).  (Workgroup-uniform.)
    // (the head and the masks first: vector memory returns in issue order, and they are all the
    //  sampling waits for)
    const int hr = L.tid >> 5, hc = L.tid & 31;
    const float hv = ld_or_zero(a.pre_head, (int64_t)(m0 + hr) * n.DO + hc,
                                L.tid < RP_ROWS * 32 && hc < n.DO && (m0 + hr) < a.B);
    const uint2 mk = *reinterpret_cast<const uint2*>(a.pre_mask + ((int64_t)tile * 512 + L.tid) * 2);
    __builtin_amdgcn_sched_barrier(0);
    sr_l1_fill<NGC>(R1, a.critic[0].W1f, L.tile0, (a.critic[0].H1 + 15) >> 4, L.lane);
    sr_prefetch<NGH>(R, a.critic[0].W2f, L.tile0, (a.critic[0].H2 + 15) >> 4, L.lane);
    __builtin_amdgcn_sched_barrier(0);
    if (L.tid < RP_ROWS * 32) headS[hr * SR_HEADP + hc] = hv;
    m1a = mk.x;
    m2a = mk.y;
    __syncthreads();
  } else {
    const SrNext nxa{a.critic[0].W1f, (a.critic[0].H1 + 15) >> 4, a.critic[0].W2f, (a.critic[0].H2 + 15) >> 4};
    if constexpr (H2)
      sr_actor_fwd_h2<NGA, NGC, true>(n, cst, X.usc, xs, P0, X, hB, red, headS, R, R1, L, row, rok, m1a,
                                      m2a, nxa);
    else
      sr_actor_fwd<NGH, NGA, NGC, true>(n, cst, xs, P0, hA, hB, red, headS, R, R1, L, row, rok, m1a, m2a,
                                        nxa);
  }
  SR_STAMP(a.prof, wg, 2);
  // ---- sample: action -> xs[:, S:], log pi
  SrGauss G;
  G.t = G.sd = G.n = G.eps = G.bound = 0.f;
  float* terms = red;                       // [16][16]
  if constexpr (HEAD == 0) {
    if (sok) {
      terms[sr * 16 + sj] = sr_sample(headS, sr, sj, a.A, eps, lo, hi, xs, P0, a.S, G);
      if (split) publish_y(a.xact + (int64_t)tile * SR_XACT + sr * 16 + sj, xs[sr * P0 + a.S + sj]);
    }
    __syncthreads();
    if (L.tid < RP_ROWS) {
      float lp = 0.f;
      for (int k = 0; k < a.A; ++k) lp += terms[L.tid * 16 + k];
      small[2 * RP_ROWS + L.tid] = lp;
      if (m0 + L.tid < a.B) a.logp[m0 + L.tid] = lp;
    }
  } else {
    // VanillaContinuousActorNetwork.sample_action (actor_networks.py:448-485, action_scaling :29-51)
    if (sok) {
      G.t = tanhf(headS[sr * SR_HEADP + sj]);
      xs[sr * P0 + a.S + sj] = (((hi - lo) * (G.t + 1.0f)) / 2.0f) + lo;
    }
  }
  SR_STAMP(a.prof, wg, 3);
  // ---- both critics at (s, a): q_c and gx_c = Gm_c W1_c[:, S:]
  float gx[2];
  gx[0] = gx[1] = 0.f;
  const int t_lo = a.S >> 4;
  const int ntl = ((W + 15) >> 4) - t_lo;       // <= 2 (A <= 16)
  SrSmallW w3t;
#pragma unroll
  for (int c = 0; c < NCRIT; ++c) {
    const SacMlp3& q = a.critic[c];
    constexpr int CN = NCRIT - 1;        // the next critic, when there is one
    const bool lastc = c == NCRIT - 1;
    SrNarrowW gw;
    sr_narrow_load(gw, q.W1tf, wf16_nkg(q.H1), t_lo, ntl, L);
    SrNext nx;
    nx.W1 = lastc ? nullptr : a.critic[CN].W1f; nx.nt1 = (a.critic[CN].H1 + 15) >> 4;
    nx.Wh = lastc ? n.W2tf : a.critic[CN].W2f;
    nx.nth = lastc ? (n.H1 + 15) >> 4 : (a.critic[CN].H2 + 15) >> 4;
    if constexpr (H2)
      sr_critic_h2<NGC, true, false>(q, cst + (1 + c) * SR_CST, X.usc + (1 + c) * SR_USC, xs, P0, X, hC,
                                     qred, R, R1, L, row, rok, nx, a.prof, wg, 4 + 4 * c);
    else
      sr_critic<NGH, NGC, true, false>(q, cst + (1 + c) * SR_CST, xs, P0, hA, hB, hC, qred, R, R1, L,
                                       row, rok, nx, a.prof, wg, 4 + 4 * c);
    if (lastc) sr_small_load(w3t, n.W3tf, wf16_nkg(n.DO), L.tile0, (n.H2 + 15) >> 4, L.lane);
    __syncthreads();
    if (L.tid < RP_ROWS) small[c * RP_ROWS + L.tid] = sr_q(qred, L.tid, cst + (1 + c) * SR_CST);
    sr_narrow_mma(gw, wf16_nkg(q.H1), hC + L.r16 * PH + 4 * L.qd, red, L);
    __syncthreads();
    if (sok) gx[c] = sr_narrow_get(red, sr, (a.S & 15) + sj);
    SR_STAMP(a.prof, wg, 7 + 4 * c);
  }
  if constexpr (split) {
    if (L.tid < RP_ROWS)
      small[RP_ROWS + L.tid] = sr_take(a.xres + (int64_t)tile * SR_XRES + L.tid * 17 + 16, a.err, a.err_host);
    if (sok) gx[1] = sr_take(a.xres + (int64_t)tile * SR_XRES + sr * 17 + sj, a.err, a.err_host);
    __syncthreads();
    SR_STAMP(a.prof, wg, 11);
  }
  // ---- twin rule, loss, head gradient (twin_kernel mode 0, gauss_grad_kernel)
  if constexpr (HEAD == 1) {
    // -mean Q1: d loss / d q = -1/B; d head = (g / 2) (high - low) (1 - tanh^2)  (tanh_action_grad)
    if (sok) {
      const float g = (-1.0f / (float)a.B) * gx[0];
      const float gt = (g / 2.0f) * (hi - lo);
      const float d = srok ? gt * (1.0f - G.t * G.t) : 0.f;
      dhS[sr * SR_DHP + sj] = d;
      if (srok) a.d_head[(int64_t)(m0 + sr) * a.A + sj] = d;
      if (sj == 0) small[3 * RP_ROWS + sr] = srok ? -small[sr] : 0.f;
    }
  } else if (sok) {
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
    }
    if (sj == 0) small[3 * RP_ROWS + sr] = srok ? alpha * small[2 * RP_ROWS + sr] - fminf(q1, q2) : 0.f;
  }
  SR_STAMP(a.prof, wg, 12);
  // zero the rest of the head-gradient tile (k padding of the next GEMM)
  for (int e = L.tid; e < RP_ROWS * SR_DHP; e += 512) {
    const int cc = e % SR_DHP;
    if (cc >= n.DO) dhS[e] = 0.f;
  }
  // ---- actor backward: d z2 = (d head W3) [h2 > 0], d z1 = (d z2 W2) [h1 > 0]
  f32x4v acc[2];
  sr_zero(acc);
  __syncthreads();
  sr_small_gemm(acc, w3t, wf16_nkg(n.DO), dhS + L.r16 * SR_DHP + 4 * L.qd);
  if constexpr (H2) {
    // d z1 = (d z2 W2) [h1 > 0] on the fp16 pipe: the reduction runs over the rows of W2, so their
    // scales ride on d z2 (d z2[n] 2^(e_n - 141), its own scale from the row's maximum) and the
    // weight fragments take a scale per element (sr_critic_h2's G)
    sr_mask_out(acc, m2a, nullptr, PH, L, n.dz2, row, n.H2, rok);
    int fCa[2][4];
    sr_fields_C(fCa, X.usc, L.wave, L.qd);
    float4 z[2];
    unsigned zm = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned mt = m2a >> (4 * t);
      auto inv = [](int field) { return __uint_as_float((unsigned)(field - 14) << 23); };
      z[t] = make_float4((rok && (mt & 1u)) ? acc[t][0] * inv(fCa[t][0]) : 0.f,
                         (rok && (mt & 2u)) ? acc[t][1] * inv(fCa[t][1]) : 0.f,
                         (rok && (mt & 4u)) ? acc[t][2] * inv(fCa[t][2]) : 0.f,
                         (rok && (mt & 8u)) ? acc[t][3] * inv(fCa[t][3]) : 0.f);
      zm = umaxu(zm, umax4(z[t]));
    }
    sr_rowmax_put(zm, X.rmaxw, L.wave, L.r16, L.qd);
    SR_STAMP(a.prof, wg, 13);
    __syncthreads();
    const int fz = sr_rowmax_field(X.rmaxw, L.r16);
    sr_planes_put(z, h2_scale(fz), X.pAh, X.pAl, L.r16, L.qd, L.wave);
    __syncthreads();
    f32x4v c[2][HF_NACC];
    h2_zero(c);
    const float unused[2] = {0.f, 0.f};
    const int plane_off = L.r16 * HF_PITCH + 8 * L.qd;
    WRing none;
    rows16_gemm_h2<16, 0, true>(c, R, n.W2tf, L.tile0, L.lane, unused, X.usc + 4 * L.qd,
                                X.pAh + plane_off, X.pAl + plane_off, none, nullptr, false);
    const float4 zero2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int fG[2][4] = {{141, 141, 141, 141}, {141, 141, 141, 141}};
    h2_finish(acc, c, fz, fG, zero2);
  } else {
    sr_mask_out(acc, m2a, hA, PH, L, n.dz2, row, n.H2, rok);
    SR_STAMP(a.prof, wg, 13);
    sr_zero(acc);
    __syncthreads();
    sr_gemm<NGH>(acc, R, n.W2tf, wf16_nkg(n.H2), L.tile0, (n.H1 + 15) >> 4,
                 hA + L.r16 * PH + 4 * L.qd, L.lane, nullptr, 0);
  }
  sr_mask_out(acc, m1a, nullptr, PH, L, n.dz1, row, n.H1, rok);
  SR_STAMP(a.prof, wg, 14);
  // ---- actor loss: this tile's partial
  sr_tile_partial(small + 3 * RP_ROWS, a.tk.partials, tile);
  SR_STAMP(a.prof, wg, 15);
}

// HEAD 1: `actor` is the TARGET policy, the next action is its tanh-scaled output plus the clamped
// smoothing noise (td3.py:151-175; none for DDPG), and y = min(q1', q2') gamma (1 - term) + r.
template <int NGH, int NGA, int NGC, int HEAD, bool SPLIT = false, bool H2 = false>
__global__ __launch_bounds__(512) void sac_rows_b_kernel(SacRowsBArgs a) {
  static_assert(!H2 || (NGH == 16 && HEAD == 0), "H2: hidden 256, continuous SAC");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SrLane L = sr_lane();
  const int W = a.S + a.A;
  const int P0 = rp_pad(W), PH = row_hid_pitch();
  SrH2 X;
  if constexpr (H2) X = sr_h2_carve(smem, W);
  float* xs = smem;
  float* hA = xs + RP_ROWS * P0;
  float* hB = hA + RP_ROWS * PH;
  float* hC = hB + RP_ROWS * PH;
  float* red = hC + RP_ROWS * PH;
  float* headS = red + 8 * RP_ROWS * SR_HEADP;
  float* qred = headS + RP_ROWS * SR_DHP;
  float* small = qred + 8 * RP_ROWS;
  float* cst = small + 8 * RP_ROWS;
  const int nrole = SPLIT ? (a.pre_state ? 3 : 2) : 1;
  const int tile = (int)blockIdx.x / nrole, brole = (int)blockIdx.x - tile * nrole;
  const int m0 = tile * RP_ROWS;
  const int64_t row = m0 + L.r16;
  const bool rok = row < a.B;
  WRing R, R1;
  const int wg = SPLIT ? brole * (int)(gridDim.x / nrole) + tile : (int)blockIdx.x;
  if constexpr (SPLIT) {
    if (brole == 2) {
      // ------------------------------------------------- the next step's actor forward (pre_state)
      const SacMlp3& n = a.actor;
      sr_l1_fill<NGA>(R1, n.W1f, L.tile0, (n.H1 + 15) >> 4, L.lane);
      sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
      {
        SrTile xt;
        SrConsts kc;
        unsigned uv = 0u;
        const bool fits = sr_tile_fits(P0);
        if (fits) sr_tile_load(xt, a.pre_state, a.ld_pre, a.S, m0, a.B, P0, L.tid);
        sr_consts_load(kc, n, false, L.tid);
        if constexpr (H2) uv = sr_usc_load(n.um, L.tid);
        if (fits) sr_tile_store(xt, xs, P0, L.tid);
        else sr_stage(a.pre_state, a.ld_pre, a.S, m0, a.B, xs, P0, L.tid);
        sr_consts_store(cst, kc, L.tid);
        if constexpr (H2) sr_usc_store(X.usc, uv, L.tid);
      }
      SrNext none;
      none.W1 = nullptr; none.nt1 = 0; none.Wh = nullptr; none.nth = 0;
      unsigned pm1, pm2;
      float* headP = red + 8 * RP_ROWS * SR_HEADP;
      if constexpr (H2)
        sr_actor_fwd_h2<NGA, 0, true>(n, cst, X.usc, xs, P0, X, hB, red, headP, R, R1, L, row, rok, pm1, pm2,
                                      none);
      else
        sr_actor_fwd<NGH, NGA, 0, true>(n, cst, xs, P0, hA, hB, red, headP, R, R1, L, row, rok, pm1, pm2,
                                        none);
      for (int e = L.tid; e < RP_ROWS * n.DO; e += 512) {
        const int r = e / n.DO, c = e - r * n.DO;
        if (m0 + r < a.B) a.pre_head[(int64_t)(m0 + r) * n.DO + c] = headP[r * SR_HEADP + c];
      }
      *reinterpret_cast<uint2*>(a.pre_mask + ((int64_t)tile * 512 + L.tid) * 2) = make_uint2(pm1, pm2);
      return;
    }
  }
  SR_STAMP(a.prof, wg, 0);
  const int sr = L.tid / a.A, sj = L.tid - sr * a.A;
  const bool sok = L.tid < RP_ROWS * a.A;
  if constexpr (SPLIT) {
    if (brole == 1) {
      // ------------------------------------------------- helper: target critic 2 at (s', a')
      const SacMlp3& q = a.target[1];
      sr_l1_fill<NGC>(R1, q.W1f, L.tile0, (q.H1 + 15) >> 4, L.lane);
      sr_prefetch<NGH>(R, q.W2f, L.tile0, (q.H2 + 15) >> 4, L.lane);
      {
        SrTile xt;
        SrConsts kc;
        unsigned uv = 0u;
        const bool fits = sr_tile_fits(P0);
        if (fits) sr_tile_load(xt, a.next_state, a.ld_next, a.S, m0, a.B, P0, L.tid);
        sr_consts_load(kc, q, true, L.tid);
        if constexpr (H2) uv = sr_usc_load(q.um, L.tid);
        if (fits) sr_tile_store(xt, xs, P0, L.tid);
        else sr_stage(a.next_state, a.ld_next, a.S, m0, a.B, xs, P0, L.tid);
        sr_consts_store(cst, kc, L.tid);
        if constexpr (H2) sr_usc_store(X.usc, uv, L.tid);
      }
      SR_STAMP(a.prof, wg, 1);
      __syncthreads();   // the staged zeros of xs[:, S:] are down before the action lands on them
      if (sok) xs[sr * P0 + a.S + sj] = sr_take(a.xact + (int64_t)tile * SR_XACT + sr * 16 + sj, a.err, a.err_host);
      SR_STAMP(a.prof, wg, 3);
      SrNext none;
      none.W1 = nullptr; none.nt1 = 0; none.Wh = nullptr; none.nth = 0;
      if constexpr (H2)
        sr_critic_h2<NGC, false, false>(q, cst, X.usc, xs, P0, X, hC, qred, R, R1, L, row, rok, none,
                                        a.prof, wg, 8);
      else
        sr_critic<NGH, NGC, false, false>(q, cst, xs, P0, hA, hB, hC, qred, R, R1, L, row, rok, none,
                                          a.prof, wg, 8);
      __syncthreads();
      if (L.tid < RP_ROWS)
        publish_y(a.xres + (int64_t)tile * SR_XRES + L.tid * 17 + 16, sr_q(qred, L.tid, cst));
      SR_STAMP(a.prof, wg, 15);
      return;
    }
  }
  const SacMlp3& n = a.actor;
  sr_l1_fill<NGA>(R1, n.W1f, L.tile0, (n.H1 + 15) >> 4, L.lane);
  sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
  const bool srok = sok && (m0 + sr) < a.B;
  const float eps = ld_or_zero(a.noise, (int64_t)(m0 + sr) * a.ld_noise + sj, srok && a.noise != nullptr);
  const float lo = ld_or_zero(a.low, sj, sok), hi = ld_or_zero(a.high, sj, sok);
  const float alpha = HEAD == 0 ? a.alpha_in[0] : 0.f;
  {
    SrTile xt;
    SrConsts k0, k1, k2;
    const bool fits = sr_tile_fits(P0);
    if (fits) sr_tile_load(xt, a.next_state, a.ld_next, a.S, m0, a.B, P0, L.tid);
    sr_consts_load(k0, n, false, L.tid);
    sr_consts_load(k1, a.target[0], true, L.tid);
    if (!SPLIT) sr_consts_load(k2, a.target[1], true, L.tid);
    unsigned u0v = 0u, u1v = 0u, u2v = 0u;
    if constexpr (H2) {
      u0v = sr_usc_load(n.um, L.tid);
      u1v = sr_usc_load(a.target[0].um, L.tid);
      if (!SPLIT) u2v = sr_usc_load(a.target[1].um, L.tid);
    }
    if (fits) sr_tile_store(xt, xs, P0, L.tid);
    else sr_stage(a.next_state, a.ld_next, a.S, m0, a.B, xs, P0, L.tid);
    sr_consts_store(cst, k0, L.tid);
    sr_consts_store(cst + SR_CST, k1, L.tid);
    if (!SPLIT) sr_consts_store(cst + 2 * SR_CST, k2, L.tid);
    if constexpr (H2) {
      sr_usc_store(X.usc, u0v, L.tid);
      sr_usc_store(X.usc + SR_USC, u1v, L.tid);
      if (!SPLIT) sr_usc_store(X.usc + 2 * SR_USC, u2v, L.tid);
    }
  }
  // this tile's Bellman-error inputs
  float qa = 0.f, qb = 0.f, rew = 0.f, live = 0.f;
  if (L.tid < RP_ROWS && m0 + L.tid < a.B) {
    const int b = m0 + L.tid;
    qa = a.q[0][b]; qb = a.q[1][b]; rew = a.reward[b];
    live = 1.0f - (a.term[b] ? 1.0f : 0.0f);
  }
  SR_STAMP(a.prof, wg, 1);
  unsigned m1a, m2a;
  {
    const SrNext nxa{a.target[0].W1f, (a.target[0].H1 + 15) >> 4, a.target[0].W2f, (a.target[0].H2 + 15) >> 4};
    if constexpr (H2)
      sr_actor_fwd_h2<NGA, NGC, false>(n, cst, X.usc, xs, P0, X, hB, red, headS, R, R1, L, row, rok, m1a,
                                       m2a, nxa);
    else
      sr_actor_fwd<NGH, NGA, NGC, false>(n, cst, xs, P0, hA, hB, red, headS, R, R1, L, row, rok, m1a, m2a,
                                         nxa);
  }
  SR_STAMP(a.prof, wg, 2);
  if constexpr (HEAD == 0) {
    SrGauss G;
    float* terms = red;
    if (sok) {
      terms[sr * 16 + sj] = sr_sample(headS, sr, sj, a.A, eps, lo, hi, xs, P0, a.S, G);
      if (SPLIT) publish_y(a.xact + (int64_t)tile * SR_XACT + sr * 16 + sj, xs[sr * P0 + a.S + sj]);
    }
    __syncthreads();
    if (L.tid < RP_ROWS) {
      float lp = 0.f;
      for (int k = 0; k < a.A; ++k) lp += terms[L.tid * 16 + k];
      small[2 * RP_ROWS + L.tid] = lp;
    }
  } else {
    // tanh_action_kernel: a = ((high - low)(tanh z + 1)) / 2 + low; with noise: the draws clamped
    // to [-clip, clip], rescaled by (high - low) / 2, added, the sum clamped to [low, high]
    if (sok) {
      const float t = tanhf(headS[sr * SR_HEADP + sj]);
      float act = (((hi - lo) * (t + 1.0f)) / 2.0f) + lo;
      if (a.noise) {
        float nz = fminf(fmaxf(eps, -a.noise_clip), a.noise_clip);
        nz = (nz * (hi - lo)) / 2.0f;
        act = fminf(fmaxf(act + nz, lo), hi);
      }
      xs[sr * P0 + a.S + sj] = act;
      if (SPLIT) publish_y(a.xact + (int64_t)tile * SR_XACT + sr * 16 + sj, act);
    }
    if (L.tid < RP_ROWS) small[2 * RP_ROWS + L.tid] = 0.f;
  }
  SR_STAMP(a.prof, wg, 3);
  // the rows of s2 / Gm this workgroup will scale: requested now, needed after y
  constexpr int SCALE_IT = 2;     // 16 rows x 256 / 4 floats = 1024 float4 per array
  float4 pre[2][2][SCALE_IT];
  const bool pre_ok = (a.H1c == 256) && (a.H2c == 256);
  constexpr int NT = SPLIT ? 1 : 2;      // target critics run here (SPLIT: the helper runs the second)
#pragma unroll
  for (int c = 0; c < NT; ++c) {
    SrNext nx;
    const bool more = c + 1 < NT;
    nx.W1 = more ? a.target[1].W1f : nullptr; nx.nt1 = (a.target[1].H1 + 15) >> 4;
    nx.Wh = more ? a.target[1].W2f : nullptr; nx.nth = (a.target[1].H2 + 15) >> 4;
    if constexpr (H2)
      sr_critic_h2<NGC, false, false>(a.target[c], cst + (1 + c) * SR_CST, X.usc + (1 + c) * SR_USC, xs,
                                      P0, X, hC, qred, R, R1, L, row, rok, nx, a.prof, wg, 4 + 4 * c);
    else
      sr_critic<NGH, NGC, false, false>(a.target[c], cst + (1 + c) * SR_CST, xs, P0, hA, hB, hC, qred, R,
                                        R1, L, row, rok, nx, a.prof, wg, 4 + 4 * c);
    if (c == NT - 1 && pre_ok) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
          for (int it = 0; it < SCALE_IT; ++it) {
            const int e = L.tid + it * 512;
            const int r = e >> 6, cc = (e & 63) * 4;
            const float* base = which == 0 ? a.dz2[k] : a.dz1[k];
            pre[k][which][it] = ld4_or_zero(base, (int64_t)(m0 + r) * 256 + cc, (m0 + r) < a.B);
          }
    }
    __syncthreads();
    if (L.tid < RP_ROWS) small[c * RP_ROWS + L.tid] = sr_q(qred, L.tid, cst + (1 + c) * SR_CST);
  }
  if constexpr (SPLIT) {
    // (read below by the thread that writes it here)
    if (L.tid < RP_ROWS)
      small[RP_ROWS + L.tid] = sr_take(a.xres + (int64_t)tile * SR_XRES + L.tid * 17 + 16, a.err, a.err_host);
    SR_STAMP(a.prof, wg, 11);
  }
  // ---- y, the Bellman errors, the critic loss rows (twin_kernel mode 1, mse_head_kernel)
  if (L.tid < RP_ROWS) {
    const int b = m0 + L.tid;
    float d1 = 0.f, d2 = 0.f, lr = 0.f;
    if (b < a.B) {
      const float mn = fminf(small[L.tid], small[RP_ROWS + L.tid]);
      const float v = mn - alpha * small[2 * RP_ROWS + L.tid];
      const float y = __fadd_rn(__fmul_rn(__fmul_rn(v, a.gamma), live), rew);
      const float e1 = __fsub_rn(qa, y), e2 = __fsub_rn(qb, y);
      const float gs = 1.0f / (float)a.B;
      d1 = __fmul_rn(gs, e1);
      d2 = __fmul_rn(gs, e2);
      a.dq[0][b] = d1;
      a.dq[1][b] = d2;
      lr = e1 * e1 + e2 * e2;
    }
    small[3 * RP_ROWS + L.tid] = d1;
    small[4 * RP_ROWS + L.tid] = d2;
    small[5 * RP_ROWS + L.tid] = lr;
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
      if (pre_ok) {
#pragma unroll
        for (int it = 0; it < SCALE_IT; ++it) {
          const int e = L.tid + it * 512;
          const int r = e >> 6, cc = (e & 63) * 4;
          if (m0 + r >= a.B) continue;
          float4 v = pre[c][which][it];
          const float s = dqs[r];
          v.x *= s; v.y *= s; v.z *= s; v.w *= s;
          *reinterpret_cast<float4*>(base + (int64_t)(m0 + r) * 256 + cc) = v;
        }
        continue;
      }
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
  // ---- critic loss: this tile's partial
  sr_tile_partial(small + 5 * RP_ROWS, a.tk.partials, tile);
  SR_STAMP(a.prof, wg, 15);
}

// End of the step (one workgroup): the two losses from the per-tile partials, in tile order, and
// the entropy-coefficient step (:134-151) with alpha_kernel's arithmetic.
struct SacFinishArgs {
  const float* part_a; const float* part_b; int tiles; int B;
  float* actor_loss; float* critic_loss;
  float* log_alpha; float* am; float* av; float* avmax; float* alpha;
  const float* logp; float target_entropy; AdamScalars ac; float* alpha_loss_out;
};
static __global__ __launch_bounds__(256) void sac_finish_kernel(SacFinishArgs a) {
  __shared__ float red[256];
  __shared__ float pa_[256], pb_[256];
  const int tid = threadIdx.x;
  float sa = 0.f, sb = 0.f;
  // tiles in order within a thread, threads in order afterwards: a fixed summation order
  for (int base = 0; base < a.tiles; base += 256) {
    pa_[tid] = base + tid < a.tiles ? a.part_a[base + tid] : 0.f;
    pb_[tid] = base + tid < a.tiles ? a.part_b[base + tid] : 0.f;
    __syncthreads();
    if (tid == 0) {
      const int n = a.tiles - base < 256 ? a.tiles - base : 256;
      for (int k = 0; k < n; ++k) {
        sa += pa_[k];
        sb += pb_[k];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (a.actor_loss) a.actor_loss[0] = sa / (float)a.B;
    // ((q1 - y)^2 + (q2 - y)^2 summed) / B / 2  ==  (mse1 + mse2) / 2   (critic_utils.py:170-203)
    a.critic_loss[0] = (sb / (float)a.B) * 0.5f;
  }
  if (!a.log_alpha) return;
  const float ea = expf(a.log_alpha[0]);
  float part = 0.f;
  for (int b = tid; b < a.B; b += 256) part += -ea * (a.logp[b] + a.target_entropy);
  red[tid] = part;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    const float g = red[0] / (float)a.B;
    if (a.alpha_loss_out) a.alpha_loss_out[0] = g;
    AdamState st;
    st.p = a.log_alpha; st.m = a.am; st.v = a.av; st.vmax = a.avmax;
    const float pnew = adam_update(a.ac, st, 0, g);
    a.alpha[0] = expf(pnew);
  }
}

// ---- any [K0, H1, H2, DO] ReLU network forward through the same building blocks ------------------
// pa_mlp_forward / _forward2 for the shapes every actor and critic of the family has (two hidden
// layers <= 256; one output — a critic, whose last layer is a dot product in the layer-2 epilogue —
// or up to 32 — an actor head, K split over the waves).  The generic mlp_rowfwd_kernel spends 19 us
// on such a twin forward at B = 1024: biases fetched from global memory before each GEMM, rolled
// weight loops that expose one memory latency per four k-groups, and a full GEMM pass for a 1- or
// 16-wide last layer; this one 10-12 us.
struct Rows3FwdArgs {
  SacMlp3 net[2];
  const float* x; int ldx;
  int B;
  float* out[2]; int ldo[2];
  long long* prof;
};
template <int NGH, bool CRITIC>
__global__ __launch_bounds__(512) void rows3_fwd_kernel(Rows3FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SrLane L = sr_lane();
  const SacMlp3& n = a.net[blockIdx.y];
  const int P0 = rp_pad(n.K0), PH = row_hid_pitch();
  float* xs = smem;
  float* hA = xs + RP_ROWS * P0;
  float* hB = hA + RP_ROWS * PH;
  float* hC = hB + RP_ROWS * PH;
  float* red = hC + RP_ROWS * PH;
  float* headS = red + 8 * RP_ROWS * SR_HEADP;
  float* qred = headS + RP_ROWS * SR_DHP;
  float* small = qred + 8 * RP_ROWS;
  float* cst = small + 8 * RP_ROWS;
  const int m0 = blockIdx.x * RP_ROWS;
  const int64_t row = m0 + L.r16;
  const bool rok = row < a.B;
  WRing R, R1;
  sr_prefetch<NGH>(R, n.W2f, L.tile0, (n.H2 + 15) >> 4, L.lane);
  {
    SrTile xt;
    SrConsts k0;
    const bool fits = sr_tile_fits(P0);
    if (fits) sr_tile_load(xt, a.x, a.ldx, n.K0, m0, a.B, P0, L.tid);
    sr_consts_load(k0, n, CRITIC, L.tid);
    if (fits) sr_tile_store(xt, xs, P0, L.tid);
    else sr_stage(a.x, a.ldx, n.K0, m0, a.B, xs, P0, L.tid);
    sr_consts_store(cst, k0, L.tid);
  }
  SrNext none;
  none.W1 = nullptr; none.nt1 = 0; none.Wh = nullptr; none.nth = 0;
  if constexpr (CRITIC) {
    sr_critic<NGH, 0, false, true>(n, cst, xs, P0, hA, hB, hC, qred, R, R1, L, row, rok, none,
                                   nullptr, 0, 0);
    __syncthreads();
    if (L.tid < RP_ROWS && m0 + L.tid < a.B)
      a.out[blockIdx.y][(int64_t)(m0 + L.tid) * a.ldo[blockIdx.y]] = sr_q(qred, L.tid, cst);
  } else {
    unsigned m1, m2;
    sr_actor_fwd<NGH, 0, 0, true>(n, cst, xs, P0, hA, hB, red, headS, R, R1, L, row, rok, m1, m2, none);
    for (int e = L.tid; e < RP_ROWS * n.DO; e += 512) {
      const int r = e / n.DO, c = e - r * n.DO;
      if (m0 + r < a.B)
        a.out[blockIdx.y][(int64_t)(m0 + r) * a.ldo[blockIdx.y] + c] = headS[r * SR_HEADP + c];
    }
  }
}

}  // namespace pa
