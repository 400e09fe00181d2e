// Online-network kernels of the DQN learner step (gfx950, fp32 MFMA 16x16x4):
//
//   online_rowpass_kernel   forward, loss and backward-to-pre-activations of the online Q network
//                           for 16 batch rows per workgroup, everything row-local in LDS/registers
//                           (deep_td_learning.py:269-290 forward, :319-320 MSE, :353-354 autograd)
//   repack_online_kernel    MFMA fragment-major copies of the weights (built once; afterwards kept
//                           current by the kernel that updates the weights)
//
// Why row-local: the online chain of one learn_batch is five dependent 1024-row stages of
// 0.07-0.27 GFLOP each.  As separate launches each pays a kernel boundary plus an exposed
// memory latency (5 x ~6-8 us measured); as one launch the activations of a row tile never leave
// the CU and only the weights stream from L2.  16-row tiles (v_mfma_f32_16x16x4_f32) give 64
// workgroups at B = 1024; the matrix pipe time of one workgroup (2 waves per SIMD) is the floor.
//
// Tiles are TRANSPOSED like target_fused_kernel's: the weights are the MFMA A operand
// (i = hidden unit), the activations the B operand (j = batch row):
//   A: lane l holds W[unit = l & 15][k-slot = l >> 4]
//   B: lane l holds X[k-slot = l >> 4][row = l & 15]
//   C: acc[reg] = C[unit = 4 * (l >> 4) + reg][row = l & 15]
// so a lane owns ONE batch row (r16 = l & 15) and four consecutive hidden units per tile: layer
// outputs are float4 stores, the output head is an in-lane fma chain, and the ReLU masks of the
// backward pass are still in the lane's registers when they are needed.
// K is consumed in groups of 16: lane quarter qd = l >> 4 owns k = 16 g + 4 qd + j for the j-th
// MFMA of the group, so ONE float4 per operand feeds four MFMAs.
#pragma once
#include "dqn_kernels.hpp"
#include "target_h2_kernel.hpp"

namespace pa {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four accumulators per output tile: the j-th MFMA of every 16-wide k-group feeds accumulator j
// (k = 16 g + 4 qd + j), they are added pairwise at the end.  The fp32 chain of one accumulator
// was 4 K / 16 steps long (256 for the hidden layers); chains a quarter as long with a pairwise
// tail put Q(s, a) closer to float64 than the reference's own blocked fp32 sums are (bench.py's
// parity block).  `seed` (the bias) opens chain 0, as it opened the single chain before.
// (NACC defaults to 1 — the single chain — for the other users of these loops: sac_rows.hpp,
//  mlp_rowpass.hpp keep their arithmetic.)
// RP_NACC_FWD / RP_NACC_BWD (1, 2 or 4): accumulators per tile of the forward layers / of the
// backward product G = s2 W2 (whose rounding enters gradients, not the Q-values the parity bar is on)
#ifndef RP_NACC_FWD
#define RP_NACC_FWD 4
#endif
#ifndef RP_NACC_BWD
#define RP_NACC_BWD 4
#endif
// Measured on one box, 2000-round learn() of config 2 (round 5, tools/r5_cfgs.txt; us per round | max
// relative error of Q(s, a) against float64 | against the reference's fp32 output; the reference
// itself is 1.61e-5 from float64):
//   FWD/BWD 1/1  36.88 | 2.34e-5 | 1.69e-5      (rounds 1-4)
//           2/1  37.04 | 1.38e-5 | 1.65e-5
//           4/1  37.04 | 8.75e-6 | 1.79e-5
//           4/4  37.14-37.30 | 8.75e-6 | 1.79e-5      (default)
// (4/4 rather than 4/1: the 12-round trajectory fixture dqn_cfg2_shape_small_batch has a layer-2
//  pre-activation within 1e-7 of zero in its second round; the backward product's rounding decides
//  which side of the ReLU it lands on after round 1's step, and 1/1 and 4/4 land on the reference's
//  side — within 2e-7 of its parameters after 12 rounds — while 4/1 and the paired row pass do not:
//  tools/debug_pair_traj.py.  Each variant's gradients are within 1.6e-7 of float64's.)
//   paired row pass (online_pair_kernel.hpp, 4/4)  36.79 | 7.88e-6 | 1.84e-5
template <int NACC>
__device__ __forceinline__ void acc4_seed(f32x4v (&c)[2][NACC], const f32x4v (&acc)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    c[t][0] = acc[t];
#pragma unroll
    for (int j = 1; j < NACC; ++j) c[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NACC>
__device__ __forceinline__ void acc4_sum(f32x4v (&acc)[2], const f32x4v (&c)[2][NACC]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if constexpr (NACC == 4) acc[t] = (c[t][0] + c[t][1]) + (c[t][2] + c[t][3]);
    else if constexpr (NACC == 2) acc[t] = c[t][0] + c[t][1];
    else acc[t] = c[t][0];
  }
}

// ---- fragment-major layout for the 16x16x4 tiles ------------------------------------------
// Wf[(unit_tile * nkg + kgroup) * 64 + lane] = float4{ W[unit][16 g + 4 qd + 0..3] },
// lane = qd * 16 + (unit & 15).  Entries outside the matrix are zero.
__host__ __device__ inline int wf16_nkg(int K) { return (K + 15) >> 4; }
__host__ __device__ inline int64_t wf16_index(int unit, int k, int nkg) {
  return wf16_index_(unit, k, nkg);
}
__host__ __device__ inline int64_t wf16_floats(int units, int K) {
  return (int64_t)((units + 15) >> 4) * wf16_nkg(K) * 256;
}

// All packed copies the learner keeps (device pointers; see pa_dqn in dqn.hip)
struct PackedW {
  float* W1f;    // online W1  [H1 units][S+AD]      (layer 1)
  float* W2f;    // online W2  [H2 units][H1]        (layer 2)
  float* W2tf;   // online W2^T [H1 units][H2]       (dX = dZ2 W2)
  float* tW2f;   // target W2, 32x32x2 fragment-major (target_fused_kernel)
  void* tW2sp;   // target W2 as bf16 split planes (target_split_kernel); null: not kept
  void* tW1sp;   // target W1[:, :S] as bf16 split planes (the tile's own first-layer product); null: not kept
  int sp_S;      // S of tW1sp (multiple of 16)
  void* tW2h; int* tW2hf;   // target W2 as scaled fp16 planes + unit scale fields (target_h2_kernel.hpp); null: not kept
};

struct RepackArgs {
  const float* q; const float* q_target;
  int64_t off_w1, off_w2;
  int IN, H1, H2;
  PackedW pk;
  int do_online, do_target;
  unsigned* umax_out; unsigned* umax_zero;   // do_online: row maxima of W1 / W2 (unit_max_body); null: not kept
};

// thread t0 of gsz cooperating threads (any grid shape: the learn loop's prologue launch runs this
// body in the workgroups behind its sampler blocks)
__device__ __forceinline__ void repack_body(const RepackArgs& a, int64_t t0, int64_t gsz) {
  if (a.do_online) {
    {  // W1f
      const int nkg = wf16_nkg(a.IN);
      const int64_t total = wf16_floats(a.H1, a.IN) / 4;
      const float* W = a.q + a.off_w1;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int g = (int)(tg % nkg), T = (int)(tg / nkg);
        const int n = T * 16 + (lane & 15), k = g * 16 + 4 * (lane >> 4);
        float4 v;
        v.x = (n < a.H1 && k < a.IN) ? W[(int64_t)n * a.IN + k] : 0.f;
        v.y = (n < a.H1 && k + 1 < a.IN) ? W[(int64_t)n * a.IN + k + 1] : 0.f;
        v.z = (n < a.H1 && k + 2 < a.IN) ? W[(int64_t)n * a.IN + k + 2] : 0.f;
        v.w = (n < a.H1 && k + 3 < a.IN) ? W[(int64_t)n * a.IN + k + 3] : 0.f;
        reinterpret_cast<float4*>(a.pk.W1f)[e] = v;
      }
    }
    {  // W2f
      const int nkg = wf16_nkg(a.H1);
      const int64_t total = wf16_floats(a.H2, a.H1) / 4;
      const float* W = a.q + a.off_w2;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int g = (int)(tg % nkg), T = (int)(tg / nkg);
        const int n = T * 16 + (lane & 15), k = g * 16 + 4 * (lane >> 4);
        float4 v;
        v.x = (n < a.H2 && k < a.H1) ? W[(int64_t)n * a.H1 + k] : 0.f;
        v.y = (n < a.H2 && k + 1 < a.H1) ? W[(int64_t)n * a.H1 + k + 1] : 0.f;
        v.z = (n < a.H2 && k + 2 < a.H1) ? W[(int64_t)n * a.H1 + k + 2] : 0.f;
        v.w = (n < a.H2 && k + 3 < a.H1) ? W[(int64_t)n * a.H1 + k + 3] : 0.f;
        reinterpret_cast<float4*>(a.pk.W2f)[e] = v;
      }
    }
    {  // W2tf: "unit" = k of W2 (an H1 unit), reduction index = n (an H2 unit)
      const int nkg = wf16_nkg(a.H2);
      const int64_t total = wf16_floats(a.H1, a.H2) / 4;
      const float* W = a.q + a.off_w2;
      for (int64_t e = t0; e < total; e += gsz) {
        const int lane = (int)(e & 63);
        const int64_t tg = e >> 6;
        const int g = (int)(tg % nkg), T = (int)(tg / nkg);
        const int k = T * 16 + (lane & 15), n = g * 16 + 4 * (lane >> 4);
        float4 v;
        v.x = (k < a.H1 && n < a.H2) ? W[(int64_t)n * a.H1 + k] : 0.f;
        v.y = (k < a.H1 && n + 1 < a.H2) ? W[(int64_t)(n + 1) * a.H1 + k] : 0.f;
        v.z = (k < a.H1 && n + 2 < a.H2) ? W[(int64_t)(n + 2) * a.H1 + k] : 0.f;
        v.w = (k < a.H1 && n + 3 < a.H2) ? W[(int64_t)(n + 3) * a.H1 + k] : 0.f;
        reinterpret_cast<float4*>(a.pk.W2tf)[e] = v;
      }
    }
    if (a.umax_out)
      unit_max_body(a.q, a.off_w1, a.off_w2, a.IN, a.H1, a.H2, a.umax_out, a.umax_zero, t0 >> 6, gsz >> 6,
                    (int)(t0 & 63));
  }
  if (a.do_target) {
    const int nkg = t_nkg(a.H1);
    const int64_t total = w2f_floats(a.H2, a.H1) / 4;
    const float* W = a.q_target + a.off_w2;
    for (int64_t e = t0; e < total; e += gsz) {
      const int lane = (int)(e & 63);
      const int64_t tg = e >> 6;
      const int g = (int)(tg % nkg), t = (int)(tg / nkg);
      const int n = t * 32 + (lane & 31), k = g * 8 + 4 * (lane >> 5);
      float4 v;
      v.x = (n < a.H2 && k < a.H1) ? W[(int64_t)n * a.H1 + k] : 0.f;
      v.y = (n < a.H2 && k + 1 < a.H1) ? W[(int64_t)n * a.H1 + k + 1] : 0.f;
      v.z = (n < a.H2 && k + 2 < a.H1) ? W[(int64_t)n * a.H1 + k + 2] : 0.f;
      v.w = (n < a.H2 && k + 3 < a.H1) ? W[(int64_t)n * a.H1 + k + 3] : 0.f;
      reinterpret_cast<float4*>(a.pk.tW2f)[e] = v;
      // (only whole matrices take the split kernel: H1 = H2 = 256, see target_fast_shape)
      if (a.pk.tW2sp && n < a.H2 && k + 3 < a.H1 && a.H1 == TS_H && a.H2 == TS_H)
        store_w2sp4(a.pk.tW2sp, n, k, v);
    }
    if (a.pk.tW2h && a.H1 == TS_H && a.H2 == TS_H) {
      W2hPack pk;
      pk.W2 = a.q_target + a.off_w2; pk.planes = a.pk.tW2h; pk.fields = a.pk.tW2hf;
      target_w2h_pack<false>(pk, t0 >> 6, gsz >> 6, (int)(t0 & 63));
    }
    if (a.pk.tW1sp && a.H1 == TS_H) {
      // the state columns of the target W1 (row pitch IN) as split planes, four k per thread
      const int S = a.pk.sp_S, ks = S >> 4;
      const float* W1 = a.q_target + a.off_w1;
      const int64_t total1 = (int64_t)a.H1 * (S >> 2);
      for (int64_t e = t0; e < total1; e += gsz) {
        const int n = (int)(e / (S >> 2)), k = (int)(e % (S >> 2)) * 4;
        const float* src = W1 + (int64_t)n * a.IN + k;
        store_wsp4(a.pk.tW1sp, n, k, make_float4(src[0], src[1], src[2], src[3]), ks);
      }
    }
  }
}

static __global__ __launch_bounds__(256) void repack_online_kernel(RepackArgs a) {
  repack_body(a, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// Every wave for itself (no barrier): the word normally holds the value already — the gather ran a
// whole window ago — so this is one L2-bypassing load in front of the operand burst.  Bounded like
// consume_y.
__device__ __forceinline__ void rowpass_wait_x(const int* flag, int value, int* err, int* err_host) {
  int spins = 0;
  while ((__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > kYPollSpins) {
      __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (err_host) __hip_atomic_store(err_host, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

// ---- the row pass ---------------------------------------------------------------------------
struct RowArgs {
  const float* x; int ldx;              // [B][K1] state || rep(action)
  const float* W1f; const float* b1;
  const float* W2f; const float* b2;
  const float* W2tf;
  const float* w3; const float* b3;
  const float* y;                       // [B] Bellman targets; null = forward only (probe)
  int y_tagged;                         // y is produced concurrently by another stream (consume_y)
  int* err;                             // device error word for the bounded wait
  int* err_host;                        // its pinned, device-mapped host twin (written on failure)
  int* signal_flag; int signal_value;   // optional: published by workgroup 0 as soon as it starts
  const int* wait_flag; int wait_value; // optional: x may only be read once *wait_flag >= wait_value (the side
                                        // stream's gather of this window has completed: pa_dqn::pending_wait)
                                        // ("everything before this launch on its stream is done":
                                        // the window hand-off of learn(), see pa_dqn::sig)
  long long* prof;                      // optional phase stamps (tools/prof_chain.py)
  const unsigned* umax;                 // online_rowpass_h2_kernel: max |w| per unit (online_f16_kernel.hpp)
  float* H1a; float* H2a;               // [B][H1], [B][H2] relu outputs (weight-gradient operands)
  float* dZ2; float* dZ1;               // [B][H2], [B][H1] pre-activation gradients
  float* q_out; float* dq_out; float* absd_out;  // [B]; q_out may be null
  const float* q_in;                    // PH == 2: Q(s, a) of the forward launch (its q_out)
  float norm;                           // 2 / (B * world)
  int B, K1, H1, H2;
};

constexpr int RP_ROWS = 16;
__host__ __device__ inline int rp_pad(int K) { return ((K + 63) & ~63) + 4; }  // LDS row pitch
inline size_t rowpass_smem_bytes(int K1, int H1, int H2) {
  return sizeof(float) * ((size_t)RP_ROWS * (rp_pad(K1) + rp_pad(H1) + rp_pad(H2)) + 8 * 16);
}

// acc[t] += sum_k Wf[tile0 + t][k] * act[row][k]; act points at this lane's (row, 4 qd) in LDS.
// nkg is rounded up to a multiple of 4 by the caller's LDS padding (zeros), weights beyond the
// matrix are zero through the out-of-range buffer load.
template <int PD, int NACC = 1>
__device__ __forceinline__ void rows16_gemm(f32x4v (&acc)[2], const float* __restrict__ Wf, int nkg,
                                            int tile0, int ntiles, const float* act, int lane) {
  const bool ok0 = tile0 < ntiles, ok1 = tile0 + 1 < ntiles;
  const int64_t base0 = ((int64_t)tile0 * nkg) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)nkg * 256;
  float4 r0[PD], r1[PD];
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    r0[p] = ld4_or_zero(Wf, base0 + (int64_t)p * 256, ok0 && p < nkg);
    r1[p] = ld4_or_zero(Wf, base1 + (int64_t)p * 256, ok1 && p < nkg);
    // (slot order = issue order: hipcc otherwise issues slot 0's loads LAST, and the loop's first
    //  wait — merged with this entry state — becomes vmcnt(0) in every trip)
    __builtin_amdgcn_sched_barrier(0);
  }
  const int nkgp = (nkg + PD - 1) / PD * PD;
  f32x4v c[2][NACC];
  acc4_seed<NACC>(c, acc);
  for (int g0 = 0; g0 < nkgp; g0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int g = g0 + p;
      // (the slot is refilled behind its MFMAs, into the registers they have just read: refilled
      //  ahead of them, hipcc rotates the ring at the back edge of this rolled loop with v_mov
      //  behind s_waitcnt vmcnt(0) — see rowsN_gemm, mlp_rowstep.hpp)
      if (g < nkg) {
        const float4 w0 = r0[p], w1 = r1[p];
        const float4 x4 = lds_ld4(act + g * 16);
        c[0][0 % NACC] = mfma16(w0.x, x4.x, c[0][0 % NACC]);
        c[1][0 % NACC] = mfma16(w1.x, x4.x, c[1][0 % NACC]);
        c[0][1 % NACC] = mfma16(w0.y, x4.y, c[0][1 % NACC]);
        c[1][1 % NACC] = mfma16(w1.y, x4.y, c[1][1 % NACC]);
        c[0][2 % NACC] = mfma16(w0.z, x4.z, c[0][2 % NACC]);
        c[1][2 % NACC] = mfma16(w1.z, x4.z, c[1][2 % NACC]);
        c[0][3 % NACC] = mfma16(w0.w, x4.w, c[0][3 % NACC]);
        c[1][3 % NACC] = mfma16(w1.w, x4.w, c[1][3 % NACC]);
      }
      __builtin_amdgcn_sched_barrier(0);
      r0[p] = ld4_or_zero(Wf, base0 + (int64_t)(g + PD) * 256, ok0 && (g + PD) < nkg);
      r1[p] = ld4_or_zero(Wf, base1 + (int64_t)(g + PD) * 256, ok1 && (g + PD) < nkg);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  acc4_sum<NACC>(acc, c);
}

// Compile-time k-group count: the whole k loop is unrolled, the weight fragments go through a
// PD-deep register ring with static indices (a rolled loop makes hipcc rotate the ring with
// v_mov behind s_waitcnt vmcnt(0), which exposes one memory latency per iteration), and the ring
// of the NEXT layer is filled before the current layer's epilogue and barrier.
constexpr int RP_PD = 8;
struct WRing {
  float4 r0[RP_PD], r1[RP_PD];
};
template <int NKG>
__device__ __forceinline__ void ring_fill(WRing& R, const float* __restrict__ Wf, int tile0,
                                          int ntiles, int lane) {
  const bool ok0 = tile0 < ntiles, ok1 = tile0 + 1 < ntiles;
  const int64_t base0 = ((int64_t)tile0 * NKG) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)NKG * 256;
#pragma unroll
  for (int p = 0; p < RP_PD; ++p) {
    if (p < NKG) {
      R.r0[p] = ld4_or_zero(Wf, base0 + (int64_t)p * 256, ok0);
      R.r1[p] = ld4_or_zero(Wf, base1 + (int64_t)p * 256, ok1);
    }
  }
}
template <int NKG, int NACC = 1>
__device__ __forceinline__ void rows16_gemm_static(f32x4v (&acc)[2], WRing& R,
                                                   const float* __restrict__ Wf, int tile0,
                                                   int ntiles, const float* act, int lane) {
  const bool ok0 = tile0 < ntiles, ok1 = tile0 + 1 < ntiles;
  const int64_t base0 = ((int64_t)tile0 * NKG) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)NKG * 256;
  f32x4v c[2][NACC];
  acc4_seed<NACC>(c, acc);
#pragma unroll
  for (int g = 0; g < NKG; ++g) {
    const float4 w0 = R.r0[g % RP_PD], w1 = R.r1[g % RP_PD];
    if (g + RP_PD < NKG) {
      R.r0[g % RP_PD] = ld4_or_zero(Wf, base0 + (int64_t)(g + RP_PD) * 256, ok0);
      R.r1[g % RP_PD] = ld4_or_zero(Wf, base1 + (int64_t)(g + RP_PD) * 256, ok1);
    }
    // keep the refill where it is: hipcc's scheduler otherwise sinks it next to its use (8 groups
    // later) to shorten the live range, which turns the prefetch into an exposed latency
    __builtin_amdgcn_sched_barrier(0);
    const float4 x4 = *reinterpret_cast<const float4*>(act + g * 16);
    c[0][0 % NACC] = mfma16(w0.x, x4.x, c[0][0 % NACC]);
    c[1][0 % NACC] = mfma16(w1.x, x4.x, c[1][0 % NACC]);
    c[0][1 % NACC] = mfma16(w0.y, x4.y, c[0][1 % NACC]);
    c[1][1 % NACC] = mfma16(w1.y, x4.y, c[1][1 % NACC]);
    c[0][2 % NACC] = mfma16(w0.z, x4.z, c[0][2 % NACC]);
    c[1][2 % NACC] = mfma16(w1.z, x4.z, c[1][2 % NACC]);
    c[0][3 % NACC] = mfma16(w0.w, x4.w, c[0][3 % NACC]);
    c[1][3 % NACC] = mfma16(w1.w, x4.w, c[1][3 % NACC]);
  }
  acc4_sum<NACC>(acc, c);
}

// The same loop, and in the iterations that have no refill of their own (the last RP_PD k-groups)
// one k-group of the NEXT weight stream is requested into `Rn`: those loads queue behind this
// loop's own operands (vector memory returns in issue order) instead of in front of them, and
// still have the rest of this loop to land.  Slots that do not fit are requested after the loop.
template <int NKG, int NKGN, int NACC = 1>
__device__ __forceinline__ void rows16_gemm_static_pf(f32x4v (&acc)[2], WRing& R,
                                                      const float* __restrict__ Wf, int tile0,
                                                      int ntiles, const float* act, int lane,
                                                      WRing& Rn, const float* __restrict__ Wn,
                                                      int ntiles_n, bool want_next) {
  const bool ok0 = tile0 < ntiles, ok1 = tile0 + 1 < ntiles;
  const int64_t base0 = ((int64_t)tile0 * NKG) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)NKG * 256;
  const bool nk0 = want_next && tile0 < ntiles_n, nk1 = want_next && tile0 + 1 < ntiles_n;
  const int64_t nb0 = ((int64_t)tile0 * NKGN) * 256 + lane * 4;
  const int64_t nb1 = nb0 + (int64_t)NKGN * 256;
  constexpr int FREE0 = NKG > RP_PD ? NKG - RP_PD : 0;        // first iteration without a refill
  constexpr int NSLOT = NKGN < RP_PD ? NKGN : RP_PD;          // slots ring_fill<NKGN> would load
  f32x4v c[2][NACC];
  acc4_seed<NACC>(c, acc);
#pragma unroll
  for (int g = 0; g < NKG; ++g) {
    const float4 w0 = R.r0[g % RP_PD], w1 = R.r1[g % RP_PD];
    if (g + RP_PD < NKG) {
      R.r0[g % RP_PD] = ld4_or_zero(Wf, base0 + (int64_t)(g + RP_PD) * 256, ok0);
      R.r1[g % RP_PD] = ld4_or_zero(Wf, base1 + (int64_t)(g + RP_PD) * 256, ok1);
    } else if (g - FREE0 < NSLOT) {
      Rn.r0[g - FREE0] = ld4_or_zero(Wn, nb0 + (int64_t)(g - FREE0) * 256, nk0);
      Rn.r1[g - FREE0] = ld4_or_zero(Wn, nb1 + (int64_t)(g - FREE0) * 256, nk1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float4 x4 = *reinterpret_cast<const float4*>(act + g * 16);
    c[0][0 % NACC] = mfma16(w0.x, x4.x, c[0][0 % NACC]);
    c[1][0 % NACC] = mfma16(w1.x, x4.x, c[1][0 % NACC]);
    c[0][1 % NACC] = mfma16(w0.y, x4.y, c[0][1 % NACC]);
    c[1][1 % NACC] = mfma16(w1.y, x4.y, c[1][1 % NACC]);
    c[0][2 % NACC] = mfma16(w0.z, x4.z, c[0][2 % NACC]);
    c[1][2 % NACC] = mfma16(w1.z, x4.z, c[1][2 % NACC]);
    c[0][3 % NACC] = mfma16(w0.w, x4.w, c[0][3 % NACC]);
    c[1][3 % NACC] = mfma16(w1.w, x4.w, c[1][3 % NACC]);
  }
  acc4_sum<NACC>(acc, c);
#pragma unroll
  for (int p = NKG - FREE0; p < NSLOT; ++p) {
    Rn.r0[p] = ld4_or_zero(Wn, nb0 + (int64_t)p * 256, nk0);
    Rn.r1[p] = ld4_or_zero(Wn, nb1 + (int64_t)p * 256, nk1);
  }
}

__device__ __forceinline__ void store4_guarded(float* __restrict__ base, int64_t row_off, int col,
                                               int ncols, bool vec_ok, const float4& v) {
  if (vec_ok && col + 3 < ncols) {
    *reinterpret_cast<float4*>(base + row_off + col) = v;
  } else {
    if (col < ncols) base[row_off + col] = v.x;
    if (col + 1 < ncols) base[row_off + col + 1] = v.y;
    if (col + 2 < ncols) base[row_off + col + 2] = v.z;
    if (col + 3 < ncols) base[row_off + col + 3] = v.w;
  }
}

// NG1/NG2/NG3: k-groups of layer 1 (K1), layer 2 (H1) and dX (H2) when known at compile time
// (0 = run-time loop, any shape).
//
// Phase structure (two workgroup barriers; it used to be four plus the x staging):
//   1. x fragments straight from global memory into MFMA B-operand registers (NG1 > 0; every wave
//      reads the whole 16-row tile — 9 KB from L1/L2 — so layer 1 starts after ONE exposed memory
//      latency with no LDS round trip and no barrier), layer 1, h1 -> LDS            [barrier A]
//   2. layer 2; h2, the head's partial dot products and  s2 = [h2 > 0] w3  -> LDS     [barrier B]
//   3. G = s2 W2 on the matrix pipe — the backward GEMM does NOT wait for the loss: with
//      dZ2 = dq (s2) row-wise, dZ1 = (dZ2 W2) [h1 > 0] = dq (s2 W2) [h1 > 0], so the scalar dq of a
//      row multiplies the finished product.  (Same value up to one fp32 rounding per element:
//      torch rounds dq w3 first and then accumulates, this accumulates first and rounds dq G.)
//      Only then: q (cross-wave sum), the Bellman target y — in the overlapped loop this is where
//      the kernel polls for it, one whole GEMM later than before — loss, dZ2, dZ1.
// The y tag of the overlapped loop is restored by the next launch of the chain
// (weight_grad_kernel's loss workgroup), not here: that needed one more barrier.
//
// PH: 0 = the whole pass.  1 / 2 = the same pass as TWO launches cut where it needs the Bellman
// target: 1 = forward (layers 1-2, head; leaves h1, h2, q in HBM and exits — its CUs are free while
// the targets are computed), 2 = backward (rebuilds s2 from the stored h2, G = s2 W2, then loss,
// dZ2, dZ1 exactly as PH 0).  learn() uses the pair for the FIRST round of a target-update window:
// that round's targets cannot exist before the soft update of the previous optimizer launch, and a
// whole pass that sits on its 64 CUs polling for them keeps a quarter of the round's target tiles
// waiting for a second turn.  Same values in the same order: bit-identical to PH 0.
template <int NG1, int NG2, int NG3, int PH = 0>
static __global__ __launch_bounds__(512) void online_rowpass_kernel(RowArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int P1 = rp_pad(a.K1), PH1 = rp_pad(a.H1), PH2 = rp_pad(a.H2);
  float* xs = smem;                          // [16][P1]   (run-time-shape path only)
  float* h1s = xs + RP_ROWS * P1;            // [16][PH1]
  float* d2s = h1s + RP_ROWS * PH1;          // [16][PH2]  s2 = [h2 > 0] w3
  float* qpart = d2s + RP_ROWS * PH2;        // [8][16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  if (a.signal_flag && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(a.signal_flag, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  PA_STAMP(a.prof, blockIdx.x, wave, 0);
  PA_STAMP_CYC(a.prof, blockIdx.x, wave, 14);
  if (a.wait_flag) rowpass_wait_x(a.wait_flag, a.wait_value, a.err, a.err_host);
  const int m0 = blockIdx.x * RP_ROWS;
  const int row = m0 + r16;
  const bool rok = row < a.B;
  const int u0 = wave * 32 + 4 * qd;         // this lane's units: u0 + 16 t + reg, t in {0,1}
  const int tile0 = wave * 2;
  const int nt1 = (a.H1 + 15) >> 4, nt2 = (a.H2 + 15) >> 4;
  const bool v1 = ((a.H1 & 3) == 0), v2 = ((a.H2 & 3) == 0);
  f32x4v acc[2];

  // Every kernel starts on a cold L2 (kernel boundaries write back and invalidate it), and under
  // the load of the co-running target pass one exposed global round trip costs ~2 us here
  // (tools/prof_chain.py).  So EVERYTHING the tile needs before its first barrier is requested in
  // one burst at the top — x, all of this wave's W1 fragments, the head of its W2 stream, biases,
  // w3 — and the W2^T stream of the backward GEMM is requested as soon as layer 1 has released
  // its registers; vector-memory results return in issue order, so layer 1 never waits for more
  // than its own operands.
  WRing R3;   // the backward GEMM's weight stream: requested from inside the layer-2 loop
  float4 h1k[2];  // kept for the ReLU mask of dZ1
  float4 h2k[2];
  float4 w3v[2];
  const float b3v = a.b3[0];
  auto vec4 = [&](const float* p, int col, int n) { return ld4_or_zero(p, col, col < n); };
  float part = 0.f;
  if constexpr (PH == 2) {
    // ---- backward launch: this lane's h1, h2 of the forward launch, s2 = [h2 > 0] w3 -> LDS
    const int64_t r1 = (int64_t)row * a.H1, r2 = (int64_t)row * a.H2;
    if constexpr (NG3 > 0) ring_fill<NG3>(R3, a.W2tf, tile0, nt1, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u = u0 + 16 * t;
      h1k[t] = ld4_or_zero(a.H1a, r1 + u, rok && u < a.H1);
      h2k[t] = ld4_or_zero(a.H2a, r2 + u, rok && u < a.H2);
      w3v[t] = vec4(a.w3, u, a.H2);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u = u0 + 16 * t;
      float4 z;
      z.x = (rok && h2k[t].x > 0.f) ? w3v[t].x : 0.f;
      z.y = (rok && h2k[t].y > 0.f) ? w3v[t].y : 0.f;
      z.z = (rok && h2k[t].z > 0.f) ? w3v[t].z : 0.f;
      z.w = (rok && h2k[t].w > 0.f) ? w3v[t].w : 0.f;
      if (u < PH2 - 4) *reinterpret_cast<float4*>(d2s + r16 * PH2 + u) = z;
    }
  } else {
  WRing R2;
  float4 b2v[2];
  // b1 / b2 / w3 are tensors of the flat parameter buffer: 16-byte aligned, every tensor padded to
  // a multiple of four floats with zeros (param_layout), so a float4 that starts inside a tensor
  // never leaves its slot — one unconditional vector load each, no scalar fallback (the two-path
  // form made hipcc serialise the burst below behind s_waitcnt).
  // ---- layer 1: h1 = relu(W1 x + b1)
  if constexpr (NG1 > 0) {
    // B operand of k-group g: x[row][16 g + 4 qd .. + 3]
    const bool vx = is_vec_ok(a.x, a.ldx) && ((a.K1 & 3) == 0);
    // Issue order = arrival order (vector memory returns in order): the accumulator seed b1 first,
    // then k-group by k-group what that group's eight MFMAs consume, so that layer 1 starts on its
    // first group while the later ones are still in flight.
    float4 xf[NG1], wa[NG1], wb[NG1], b1v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) b1v[t] = vec4(a.b1, u0 + 16 * t, a.H1);
    {
      const bool ok0 = tile0 < nt1, ok1 = tile0 + 1 < nt1;
      const int64_t base0 = ((int64_t)tile0 * NG1) * 256 + lane * 4;
      const int64_t base1 = base0 + (int64_t)NG1 * 256;
#pragma unroll
      for (int g = 0; g < NG1; ++g) {
        const int c = 16 * g + 4 * qd;
        if (vx) xf[g] = ld4_or_zero(a.x, (int64_t)row * a.ldx + c, rok && c < a.K1);
        else xf[g] = guarded_load4(a.x, (int64_t)row * a.ldx, rok, c, a.K1);
        wa[g] = ld4_or_zero(a.W1f, base0 + (int64_t)g * 256, ok0);
        wb[g] = ld4_or_zero(a.W1f, base1 + (int64_t)g * 256, ok1);
      }
    }
    if constexpr (NG2 > 0) ring_fill<NG2>(R2, a.W2f, tile0, nt2, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      b2v[t] = vec4(a.b2, u0 + 16 * t, a.H2);
      w3v[t] = vec4(a.w3, u0 + 16 * t, a.H2);
    }
    PA_STAMP(a.prof, blockIdx.x, wave, 1);
    __builtin_amdgcn_sched_barrier(0);   // the burst above stays above: nothing sinks to its use
#pragma unroll
    for (int t = 0; t < 2; ++t) {        // (seeded here, not at the load: the copy waits for b1)
      acc[t][0] = b1v[t].x; acc[t][1] = b1v[t].y; acc[t][2] = b1v[t].z; acc[t][3] = b1v[t].w;
    }
    constexpr int NACC = RP_NACC_FWD;
    f32x4v c[2][NACC];
    acc4_seed<NACC>(c, acc);
#pragma unroll
    for (int g = 0; g < NG1; ++g) {
      const float4 x4 = xf[g], w0 = wa[g], w1 = wb[g];
      c[0][0 % NACC] = mfma16(w0.x, x4.x, c[0][0 % NACC]);
      c[1][0 % NACC] = mfma16(w1.x, x4.x, c[1][0 % NACC]);
      c[0][1 % NACC] = mfma16(w0.y, x4.y, c[0][1 % NACC]);
      c[1][1 % NACC] = mfma16(w1.y, x4.y, c[1][1 % NACC]);
      c[0][2 % NACC] = mfma16(w0.z, x4.z, c[0][2 % NACC]);
      c[1][2 % NACC] = mfma16(w1.z, x4.z, c[1][2 % NACC]);
      c[0][3 % NACC] = mfma16(w0.w, x4.w, c[0][3 % NACC]);
      c[1][3 % NACC] = mfma16(w1.w, x4.w, c[1][3 % NACC]);
    }
    acc4_sum<NACC>(acc, c);
    PA_STAMP(a.prof, blockIdx.x, wave, 2);
  } else {
    // any shape: stage the x tile in LDS (zero padded to the pitch; the k loop runs over whole groups)
    const bool vx = is_vec_ok(a.x, a.ldx) && ((a.K1 & 3) == 0);
    const int c4 = (P1 - 4) >> 2;  // float4 slots per row
    for (int e = tid; e < RP_ROWS * c4; e += 512) {
      const int r = e / c4, c = (e - r * c4) * 4;
      const bool ok = (m0 + r) < a.B;
      float4 v;
      if (vx) v = ld4_or_zero(a.x, (int64_t)(m0 + r) * a.ldx + c, ok && c < a.K1);
      else v = guarded_load4(a.x, (int64_t)(m0 + r) * a.ldx, ok, c, a.K1);
      *reinterpret_cast<float4*>(xs + r * P1 + c) = v;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 b = vec4(a.b1, u0 + 16 * t, a.H1);
      acc[t][0] = b.x; acc[t][1] = b.y; acc[t][2] = b.z; acc[t][3] = b.w;
    }
    if constexpr (NG2 > 0) ring_fill<NG2>(R2, a.W2f, tile0, nt2, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      b2v[t] = vec4(a.b2, u0 + 16 * t, a.H2);
      w3v[t] = vec4(a.w3, u0 + 16 * t, a.H2);
    }
    PA_STAMP(a.prof, blockIdx.x, wave, 1);
    __syncthreads();
    rows16_gemm<4, RP_NACC_FWD>(acc, a.W1f, wf16_nkg(a.K1), tile0, nt1, xs + r16 * P1 + 4 * qd, lane);
    PA_STAMP(a.prof, blockIdx.x, wave, 2);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = u0 + 16 * t;
    h1k[t] = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]),
                         relu_keep_nan(acc[t][2]), relu_keep_nan(acc[t][3]));
    if (u < PH1 - 4) *reinterpret_cast<float4*>(h1s + r16 * PH1 + u) = h1k[t];
    if (rok && a.H1a) store4_guarded(a.H1a, (int64_t)row * a.H1, u, a.H1, v1, h1k[t]);
  }
  // ---- layer 2: h2 = relu(W2 h1 + b2)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    acc[t][0] = b2v[t].x; acc[t][1] = b2v[t].y; acc[t][2] = b2v[t].z; acc[t][3] = b2v[t].w;
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 3);
  __syncthreads();                                                      // barrier A: h1 tile
  PA_STAMP(a.prof, blockIdx.x, wave, 4);
  if constexpr (NG2 > 0 && NG3 > 0) {
    rows16_gemm_static_pf<NG2, NG3, RP_NACC_FWD>(acc, R2, a.W2f, tile0, nt2, h1s + r16 * PH1 + 4 * qd, lane, R3,
                                    a.W2tf, nt1, PH == 0 && a.y != nullptr);
  } else if constexpr (NG2 > 0) {
    rows16_gemm_static<NG2, RP_NACC_FWD>(acc, R2, a.W2f, tile0, nt2, h1s + r16 * PH1 + 4 * qd, lane);
  } else {
    rows16_gemm<4, RP_NACC_FWD>(acc, a.W2f, wf16_nkg(a.H1), tile0, nt2, h1s + r16 * PH1 + 4 * qd, lane);
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 5);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = u0 + 16 * t;
    h2k[t] = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]),
                         relu_keep_nan(acc[t][2]), relu_keep_nan(acc[t][3]));
    part = fmaf(h2k[t].x, w3v[t].x, part);
    part = fmaf(h2k[t].y, w3v[t].y, part);
    part = fmaf(h2k[t].z, w3v[t].z, part);
    part = fmaf(h2k[t].w, w3v[t].w, part);
    if (rok && a.H2a) store4_guarded(a.H2a, (int64_t)row * a.H2, u, a.H2, v2, h2k[t]);
    if (PH == 0 && a.y) {
      // s2 = [h2 > 0] w3: the B operand of the backward GEMM (rows beyond the batch are zero:
      // their h2 came from a zero x row only if b1/b2 say so, hence the explicit guard)
      float4 z;
      z.x = (rok && h2k[t].x > 0.f) ? w3v[t].x : 0.f;
      z.y = (rok && h2k[t].y > 0.f) ? w3v[t].y : 0.f;
      z.z = (rok && h2k[t].z > 0.f) ? w3v[t].z : 0.f;
      z.w = (rok && h2k[t].w > 0.f) ? w3v[t].w : 0.f;
      if (u < PH2 - 4) *reinterpret_cast<float4*>(d2s + r16 * PH2 + u) = z;
    }
  }
  }  // PH != 2
  // ---- head partials: q = w3 . h2 + b3 (lane quarters, then waves, fixed order)
  if constexpr (PH != 2) {
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (qd == 0) qpart[wave * 16 + r16] = part;
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 6);
  __syncthreads();                                                      // barrier B: s2, qpart
  PA_STAMP(a.prof, blockIdx.x, wave, 7);
  if constexpr (PH == 1) {
    // forward launch: Q(s, a) for the backward launch (and the report), nothing else
    float qf = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) qf += qpart[w * 16 + r16];
    qf += b3v;
    if (wave == 0 && qd == 0 && rok && a.q_out) a.q_out[row] = qf;
    PA_STAMP(a.prof, blockIdx.x, wave, 10);
    return;
  }
  // PH 2 (the backward launch of a window's first round) typically arrives BEFORE its targets and
  // waits for them: there ONE quarter-wave per workgroup polls and hands the values on through LDS
  // (16 polling lanes instead of 512 — agent-scope loads bypass the L2, and the leading target
  // tiles the launch is waiting for share that memory system)
  const bool poller = PH != 2 || (wave == 0 && qd == 0);
  unsigned ybits = kYPendingBits;
  if (a.y && rok && poller) {
    // first look at the Bellman target, in flight while the backward GEMM runs
    ybits = a.y_tagged ? __hip_atomic_load(reinterpret_cast<const unsigned*>(a.y) + row,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : __builtin_bit_cast(unsigned, a.y[row]);
  }
  if (a.y) {
    // ---- G = s2 W2 (scaled by dq below)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
    if constexpr (NG3 > 0) {
      rows16_gemm_static<NG3, RP_NACC_BWD>(acc, R3, a.W2tf, tile0, nt1, d2s + r16 * PH2 + 4 * qd, lane);
    } else {
      rows16_gemm<4, RP_NACC_BWD>(acc, a.W2tf, wf16_nkg(a.H2), tile0, nt1, d2s + r16 * PH2 + 4 * qd, lane);
    }
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 8);
  float q = 0.f;
  if constexpr (PH == 2) {
    q = rok ? a.q_in[row] : 0.f;
  } else {
#pragma unroll
    for (int w = 0; w < 8; ++w) q += qpart[w * 16 + r16];
    q += b3v;
    if (wave == 0 && qd == 0 && rok && a.q_out) a.q_out[row] = q;
  }
  if (!a.y) return;
  // ---- loss, dZ2 = [h2 > 0] * (dq * w3), dZ1 = [h1 > 0] * (dq * G)
  float yv = q;
  if (rok && poller) {
    // (an untagged y can legitimately hold the tag's bit pattern: only the tagged protocol polls)
    if (a.y_tagged && ybits == kYPendingBits) yv = consume_y(a.y + row, a.err, a.err_host);
    else yv = __builtin_bit_cast(float, ybits);
  }
  if constexpr (PH == 2) {
    if (poller) qpart[r16] = yv;      // (the head partials' LDS slots are free in this launch)
    __syncthreads();
    yv = rok ? qpart[r16] : q;
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 9);
  const float d = __fsub_rn(q, yv);
  const float dq = __fmul_rn(a.norm, d);
  if (wave == 0 && qd == 0 && rok) {
    a.dq_out[row] = dq;
    a.absd_out[row] = fabsf(d);
  }
  if (rok) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u = u0 + 16 * t;
      float4 z;
      z.x = (h2k[t].x > 0.f) ? __fmul_rn(dq, w3v[t].x) : 0.f;
      z.y = (h2k[t].y > 0.f) ? __fmul_rn(dq, w3v[t].y) : 0.f;
      z.z = (h2k[t].z > 0.f) ? __fmul_rn(dq, w3v[t].z) : 0.f;
      z.w = (h2k[t].w > 0.f) ? __fmul_rn(dq, w3v[t].w) : 0.f;
      store4_guarded(a.dZ2, (int64_t)row * a.H2, u, a.H2, v2, z);
      float4 g;
      g.x = (h1k[t].x > 0.f) ? __fmul_rn(dq, acc[t][0]) : 0.f;
      g.y = (h1k[t].y > 0.f) ? __fmul_rn(dq, acc[t][1]) : 0.f;
      g.z = (h1k[t].z > 0.f) ? __fmul_rn(dq, acc[t][2]) : 0.f;
      g.w = (h1k[t].w > 0.f) ? __fmul_rn(dq, acc[t][3]) : 0.f;
      store4_guarded(a.dZ1, (int64_t)row * a.H1, u, a.H1, v1, g);
    }
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 10);
  PA_STAMP_CYC(a.prof, blockIdx.x, wave, 15);
}

}  // namespace pa
