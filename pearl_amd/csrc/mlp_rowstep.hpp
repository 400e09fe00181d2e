// One launch for forward + loss head + backward of a network (or of two networks on the same
// input: PPO's actor and critic, ppo.py:152-192; twin critics, critic_utils.py:170-203) for 16 batch
// rows per workgroup: mlp_rowfwd_kernel's layer loop (activations kept for the weight gradients),
// the row-local part of the loss on the output tile while it is still in LDS, and
// mlp_rowbwd_kernel's layer loop on the head's gradient tile.  What it removes from a step is two
// launches and the boundaries between them (3-4 us each on this machine), the
// head's own launch, and the global round trip of the output / output-gradient tiles.
//
// The per-row arithmetic of the heads is that of their stand-alone kernels in mlp.hip
// (ppo_actor_elem_kernel, mse_head_body): the same expressions in the same order, so the
// gradients are bit-identical; the batch sums of the reported losses are taken per 16-row tile and
// then in tile order by the last workgroup to finish (ticket), i.e. grouped differently from the
// stand-alone kernels (equal to rounding).
#pragma once
#include "mlp_rowpass.hpp"

namespace pa {

enum { RS_HEAD_MSE = 1, RS_HEAD_PPO = 2, RS_HEAD_DSAC_ACTOR = 3, RS_HEAD_DSAC_TARGET = 4,
       RS_HEAD_WMSE1 = 5 };   // wmse_kernel with unit weights (neural_linear_bandit.py:176-199)

// Row math of the neural-linear bandit's weighted loss (see wmse_kernel in mlp.hip): output
// activation, unreduced criterion and d loss / d z, in torch's op order.
__device__ __forceinline__ float wloss_act(float z, int out_act) {
  return out_act == PA_OUT_SIGMOID ? 1.0f / (1.0f + expf(-z)) : z;
}
__device__ __forceinline__ float wloss_value(float p, float y, int kind) {
  if (kind == PA_LOSS_MAE) return fabsf(p - y);
  if (kind == PA_LOSS_BCE)
    return (y - 1.0f) * fmaxf(log1pf(-p), -100.0f) - y * fmaxf(logf(p), -100.0f);
  const float d = p - y;
  return d * d;
}
__device__ __forceinline__ float wloss_grad(float p, float y, float w, float wsum, int kind,
                                            int out_act) {
  float dp;
  if (kind == PA_LOSS_MSE) {
    dp = (2.0f * (p - y) * w) / wsum;
  } else {
    const float g = (1.0f / wsum) * w;
    const float d = p - y;
    if (kind == PA_LOSS_MAE) dp = (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f)) * g;
    else dp = (g * d) / fmaxf((1.0f - p) * p, 1e-12f);
  }
  return out_act == PA_OUT_SIGMOID ? (dp * (1.0f - p)) * p : dp;
}

struct RowHead {
  int kind;
  float* d_out; int ldd;        // [B][d_L]: gradient w.r.t. the network output (operand of the
                                // last layer's weight gradient)
  // RS_HEAD_MSE (single_critic_state_value_loss / twin_critic_action_value_loss,
  // critic_utils.py:139-203): d = out - target, d_out = grad_scale d, loss = mean d^2 * loss_scale
  const float* target; float grad_scale; float loss_scale;
  // RS_HEAD_PPO (ProximalPolicyOptimization._actor_loss, ppo.py:152-183), d_L = A <= 32
  const float* arep; int lda;   // [B][A] representation of the taken action
  const float* p_old;           // [B]
  const float* gae;             // [B]
  float eps, ent_scale;
  float* p_rows;                // [B] scratch: chosen-action probability of every row
  // RS_HEAD_DSAC_ACTOR / _TARGET (discrete SoftActorCritic._actor_loss / _get_next_state_expected_
  // values + Bellman target, soft_actor_critic.py:180-287), d_L = A <= 32: dsac_elem_kernel's row math
  const float* q1; const float* q2;      // [B * A], row b * A + j
  const uint8_t* mask;                   // [B, A] 1 = unavailable, or null
  const float* alpha;
  float* h_out;                          // actor: [B] sum_j P_j log(P_j + 1e-8)
  const float* reward; const uint8_t* term; float gamma; float* y;   // target (no backward)
  // RS_HEAD_WMSE1: criterion and output activation (PA_LOSS_*, PA_OUT_*); out_post [B] receives
  // the post-activation predictions when the activation is not linear
  // mean_out (nullable): the batch mean of the post-activation predictions
  int loss_kind, out_act; float* out_post; float* mean_out;
  // lin_x / lin_r (nullable; fp32-forward launches only, L >= 2): the LinUCB operands of the rows'
  // FEATURES (the output of layer L - 2, still in LDS): X[b] = [1 | f_b | 0..] (pitch ldX) and
  // R[b] = [1 | f_b | y_b | 0..] (pitch ldR) — linreg_operands_kernel with unit weights, no launch
  float* lin_x; float* lin_r; int ldX, ldR;
};

struct RowStepArgs {
  RowNetFwd fwd[2];
  RowNetBwd bwd[2];             // (d_out / ldd unused: the head's LDS tile is the operand)
  RowHead head[2];
  const float* x; int ldx;
  int B;
  float* partials;              // [2 networks][gridDim.x][2]
  unsigned* ticket;             // zero on entry, zero again on exit
  float* losses;                // [2]: one per network — or, sum_losses, losses[0] = both
  int sum_losses;
  int split_bwd;                // SPLITF kernel: the backward GEMMs as bf16x3 products too (RowNetBwd::Wtsp)
  long long* prof;              // optional phase stamps [workgroup (x + y gridDim.x)][wave][16] (tools/prof_rowstep.py)
};

#ifndef RS_PD_VALUE
#define RS_PD_VALUE 4
#endif
constexpr int RS_PD = RS_PD_VALUE;   // weight k-groups in flight per wave and tile
#ifndef RS_PD2_VALUE
#define RS_PD2_VALUE 4
#endif
constexpr int RS_PD2 = RS_PD2_VALUE;  // the same for the 32-row form (one workgroup per CU: registers to spare)
constexpr int RS_SCR = 5 * 512;   // floats of head scratch behind the row-pass tiles

__device__ __forceinline__ float block_sum_512(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = 256; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}
// two sums with one set of barriers (red: 1024 floats)
__device__ __forceinline__ float2 block_sum2_512(float v0, float v1, float* red) {
  red[threadIdx.x] = v0;
  red[512 + threadIdx.x] = v1;
  __syncthreads();
  for (int w = 256; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[threadIdx.x] += red[threadIdx.x + w];
      red[512 + threadIdx.x] += red[512 + threadIdx.x + w];
    }
    __syncthreads();
  }
  const float2 r = make_float2(red[0], red[512]);
  __syncthreads();
  return r;
}
// Hand-off of a few words to whichever workgroup finishes last: write-through (agent-scope) stores
// and L2-bypassing loads, ordered by the workgroup barrier's wait for outstanding stores and a
// relaxed ticket — NOT __threadfence(): an agent-scope release writes the whole XCD L2 back, and
// every one of 512 workgroups doing that behind megabytes of freshly stored activations made the
// fused launch slower than the three it replaces (151 vs 143 us per PPO step).
__device__ __forceinline__ void st_through(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_through(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc[t][rt] += sum_k Wf[tile0 + t][k] * act[row + 16 rt][k] for RT row tiles of 16 rows: every
// weight fragment that comes through the CU's L1 feeds 4 RT MFMAs per unit tile instead of 4.
// (PMC on the 16-row form at PPO's 4096 rows: 1.18 M one-KiB wave loads per launch = 35 us of the
// CU's 64 B/clk L1 fill next to 33 us of MFMA, and the two do not overlap well — 69 us.)
// The first PD k-groups of a wave's two unit tiles: issued BEFORE the barrier that completes the
// layer's input tile (weights do not depend on it), so their latency runs under the wait.
template <int PD>
struct RowW {
  float4 r0[PD], r1[PD];
};
template <int PD>
__device__ __forceinline__ void rowsN_fill(RowW<PD>& R, const float* __restrict__ Wf, int nkg, int tile0,
                                           int ntiles, int lane) {
  const bool ok0 = tile0 < ntiles, ok1 = tile0 + 1 < ntiles;
  const int64_t base0 = ((int64_t)tile0 * nkg) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)nkg * 256;
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    R.r0[p] = ld4_or_zero(Wf, base0 + (int64_t)p * 256, ok0 && p < nkg);
    R.r1[p] = ld4_or_zero(Wf, base1 + (int64_t)p * 256, ok1 && p < nkg);
    __builtin_amdgcn_sched_barrier(0);   // (slot order = issue order: see rows16_gemm)
  }
}
template <int PD, int RT>
__device__ __forceinline__ void rowsN_gemm(f32x4v (&acc)[2][RT], RowW<PD>& R, const float* __restrict__ Wf,
                                           int nkg, int tile0, int ntiles, const float* act, int pitch,
                                           int lane) {
  const bool ok0 = tile0 < ntiles, ok1 = tile0 + 1 < ntiles;
  const int64_t base0 = ((int64_t)tile0 * nkg) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)nkg * 256;
  const int nkgp = (nkg + PD - 1) / PD * PD;
  for (int g0 = 0; g0 < nkgp; g0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int g = g0 + p;
      // A ring slot is consumed where it lies and refilled BEHIND its MFMAs (the scheduling barrier
      // keeps the loads there): the slot's registers are the same in every trip of this rolled loop.
      // Refilled ahead of its use, the slot needs a second set of registers, and hipcc rotates the
      // ring at the loop's back edge with v_mov behind s_waitcnt vmcnt(0) — the whole ring drained
      // once per trip.
      const float4 w0 = R.r0[p], w1 = R.r1[p];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 x4 = lds_ld4(act + rt * 16 * pitch + g * 16);
        acc[0][rt] = mfma16(w0.x, x4.x, acc[0][rt]);
        acc[1][rt] = mfma16(w1.x, x4.x, acc[1][rt]);
        acc[0][rt] = mfma16(w0.y, x4.y, acc[0][rt]);
        acc[1][rt] = mfma16(w1.y, x4.y, acc[1][rt]);
        acc[0][rt] = mfma16(w0.z, x4.z, acc[0][rt]);
        acc[1][rt] = mfma16(w1.z, x4.z, acc[1][rt]);
        acc[0][rt] = mfma16(w0.w, x4.w, acc[0][rt]);
        acc[1][rt] = mfma16(w1.w, x4.w, acc[1][rt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      R.r0[p] = ld4_or_zero(Wf, base0 + (int64_t)(g + PD) * 256, ok0 && (g + PD) < nkg);
      R.r1[p] = ld4_or_zero(Wf, base1 + (int64_t)(g + PD) * 256, ok1 && (g + PD) < nkg);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int RT>
inline size_t rowstep_smem_bytes_t(int d0) {
  return sizeof(float) * ((size_t)RT * RP_ROWS * (rp_pad(d0) + 2 * row_hid_pitch()) + RS_SCR);
}

// ---- forward GEMMs on the bf16 matrix pipe at fp32 accuracy (SPLITF instantiation) ---------------
// Launches of 32 rows per workgroup (one workgroup per CU: thousands of rows — PPO's 4096-row
// minibatch) are bound by the fp32 MFMA rate in their 256 x 256 layers (10 us per layer and
// workgroup against 7.8 us of MFMA).  The SPLITF kernel runs every forward layer as a bf16x3
// product (each fp32 operand split exactly into three bf16 terms, six products,
// fp32 accumulation): weights from the engine's split planes (wsp16_index: one coalesced 1 KiB load
// per unit tile, k-step of 32 and plane — kept current by the repack pass and the optimizer
// epilogue), activations as three bf16 planes in LDS, written by the lane that produced the value;
// v_mfma_f32_16x16x32_bf16, 24 per k-step and wave instead of 64 fp32 MFMAs of twice the cycles.
// The backward half is the fp32 kernel's (its tiles alias the planes, dead by then).
constexpr int RS_PP = ROW_MAX_OUT + 8;   // bf16 pitch of a plane row: 528 B = 4 dwords mod 64 banks
constexpr int RS_PLANE_BYTES = 3 * 2 * RP_ROWS * RS_PP * 2;   // three planes of 32 rows
inline size_t rowstep_split_smem_bytes() {
  return (size_t)2 * RS_PLANE_BYTES + sizeof(float) * ((size_t)2 * RP_ROWS * row_hid_pitch() + RS_SCR);
}
// four consecutive units of one row -> the three planes (one ds_write_b64 each)
__device__ __forceinline__ void rs_store_planes4(__bf16* planes, int rows, int row, int col, const float4& v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  bf16x4 q[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 a, b, c;
    split3(x[j], a, b, c);
    q[0][j] = a; q[1][j] = b; q[2][j] = c;
  }
#pragma unroll
  for (int sp = 0; sp < 3; ++sp)
    lds_st_bf16x4(planes + ((size_t)sp * rows + row) * RS_PP + col, q[sp]);
}
#ifndef RS_SPD_VALUE
#define RS_SPD_VALUE 2
#endif
constexpr int RS_SPD = RS_SPD_VALUE;   // k-steps of weight planes in flight per wave
struct RowWS {
  bf16x8 w[RS_SPD][2][3];
};
__device__ __forceinline__ bf16x8 rs_ld_plane(const void* base, int64_t slot, bool ok) {
  // (slot = index of a 16-byte fragment; out-of-range slots read as zeros through the raw buffer load)
  const float4 v = ld4_or_zero(static_cast<const float*>(base), slot * 4, ok);
  return __builtin_bit_cast(bf16x8, v);
}
// (ks0 / kstr: this wave's k-steps are ks0, ks0 + kstr, ... — 0 / 1 everywhere but in a narrow last
//  layer, whose k-steps are dealt out to the eight waves: see the forward loop of the kernel)
__device__ __forceinline__ void rs_fill(RowWS& R, const void* Wsp, int nks, int tile0, int ntiles, int lane,
                                        int ks0 = 0, int kstr = 1) {
#pragma unroll
  for (int p = 0; p < RS_SPD; ++p) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
        R.w[p][t][sp] = rs_ld_plane(Wsp, (((int64_t)(tile0 + t) * nks + ks0 + p * kstr) * 3 + sp) * 64 + lane,
                                    tile0 + t < ntiles && ks0 + p * kstr < nks);
    __builtin_amdgcn_sched_barrier(0);   // (slot order = issue order: see rows16_gemm)
  }
}
// accm / accs[t][rt] += W planes (tiles tile0, tile0 + 1) x activation planes (RT row tiles), all k
template <int RT>
__device__ __forceinline__ void rs_gemm(f32x4v (&accm)[2][RT], f32x4v (&accs)[2][RT], RowWS& R,
                                        const void* Wsp, int nks_all, int tile0, int ntiles,
                                        const __bf16* act, int lane, int ks0 = 0, int kstr = 1) {
  const int r16 = lane & 15, qd = lane >> 4;
  // local k-step i is k-step ks0 + i kstr of the layer; nks = how many this wave has
  const int nks = ks0 < nks_all ? (nks_all - ks0 + kstr - 1) / kstr : 0;
  const int nksp = (nks + RS_SPD - 1) / RS_SPD * RS_SPD;
  for (int s0 = 0; s0 < nksp; s0 += RS_SPD) {
#pragma unroll
    for (int p = 0; p < RS_SPD; ++p) {
      const int s = s0 + p;
      // (the ring is a whole number of k-steps deep: a step past the layer's last one has zero
      //  weights but would read activation columns nobody wrote — 0 x garbage — so it is skipped)
      if (s < nks) {
        bf16x8 b[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int sp = 0; sp < 3; ++sp)
            b[rt][sp] = lds_ld_bf16x8(
                act + ((size_t)sp * RT * RP_ROWS + r16 + 16 * rt) * RS_PP + 32 * (ks0 + s * kstr) + 8 * qd);
        __builtin_amdgcn_sched_barrier(0);   // this k-step's LDS reads are issued ahead of its MFMAs
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            accs[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R.w[p][t][2], b[rt][0], accs[t][rt], 0, 0, 0);
            accs[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R.w[p][t][0], b[rt][2], accs[t][rt], 0, 0, 0);
          }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            accs[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R.w[p][t][1], b[rt][1], accs[t][rt], 0, 0, 0);
            accs[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R.w[p][t][0], b[rt][1], accs[t][rt], 0, 0, 0);
          }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            accs[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R.w[p][t][1], b[rt][0], accs[t][rt], 0, 0, 0);
            accm[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R.w[p][t][0], b[rt][0], accm[t][rt], 0, 0, 0);
          }
      }
      // the slot is refilled BEHIND its MFMAs, into the registers they have just read (see
      // rowsN_gemm: refilled ahead of them, the ring is rotated with v_mov behind vmcnt(0) once per trip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          R.w[p][t][sp] = rs_ld_plane(Wsp, (((int64_t)(tile0 + t) * nks_all + ks0 + (s + RS_SPD) * kstr) * 3 + sp) * 64 + lane,
                                      tile0 + t < ntiles && s + RS_SPD < nks);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// RT row tiles of 16 rows per workgroup (1: up to three workgroups per CU; 2: 32 rows, one
// workgroup per CU, half the weight traffic per row — for launches of more than 256 tiles)
// SPLITF (RT = 2 only): the forward layers as bf16x3 products (see above).  LDS then: two plane
// buffers (the backward's two fp32 tiles alias them), the fp32 output tile, the head scratch.
template <int RT, bool SPLITF = false>
static __global__ __launch_bounds__(512) void mlp_rowstep_kernel(RowStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(!SPLITF || RT == 2, "the split forward is built for 32 rows per workgroup");
  constexpr int ROWS = RP_ROWS * RT;
  const int net = blockIdx.y;
  const RowNetFwd& n = a.fwd[net];
  const RowNetBwd& nb = a.bwd[net];
  const RowHead& hd = a.head[net];
  const int P0 = rp_pad(n.dims[0]), PH = row_hid_pitch();
  float* xs = smem;
  float* hb[2] = {xs + ROWS * P0, xs + ROWS * P0 + ROWS * PH};
  float* scr = xs + ROWS * P0 + 2 * ROWS * PH;   // va | vb | vc | red[2], 512 floats each
  __bf16* pl[2] = {nullptr, nullptr};            // SPLITF: the two activation plane buffers
  float* otile = nullptr;                        // SPLITF: fp32 output tile of the last layer
  if constexpr (SPLITF) {
    unsigned char* base = reinterpret_cast<unsigned char*>(smem);
    pl[0] = reinterpret_cast<__bf16*>(base);
    pl[1] = reinterpret_cast<__bf16*>(base + RS_PLANE_BYTES);
    hb[0] = reinterpret_cast<float*>(base);                       // (backward: planes are dead)
    hb[1] = reinterpret_cast<float*>(base + RS_PLANE_BYTES);
    otile = reinterpret_cast<float*>(base + 2 * RS_PLANE_BYTES);
    scr = otile + ROWS * PH;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  const int m0 = blockIdx.x * ROWS;
  const int u0 = wave * 32 + 4 * qd;
  const int tile0 = wave * 2;
  const int pwg = blockIdx.x + blockIdx.y * gridDim.x;
  PA_STAMP(a.prof, pwg, wave, 0);
  // ReLU masks of this lane's own outputs, 16 bits per layer (bit 8 rt + 4 t + e of layer l at
  // 16 l): the backward's lane owns the same (row, unit) slots, so it needs no look at the kept
  // activations (a 16-dword re-read per lane and layer in front of a barrier before)
  unsigned long long fmask = 0ull;
  if constexpr (SPLITF) {
    // ------------------------------------------------------------ forward, bf16x3 (see rs_gemm)
    // layer 0's first weight fragments are asked for ahead of the input tile (neither depends on
    // the other: one memory round trip under the other's)
    RowWS R;
    const int nt_0 = (n.dims[1] + 15) >> 4, nks_0 = wsp16_nks(n.dims[0]);
    const bool pre0 = !(n.L == 1 && nt_0 <= 2 && nks_0 >= 4);   // (not a k-dealt single layer)
    if (pre0 && tile0 < nt_0) rs_fill(R, n.Wsp[0], nks_0, tile0, nt_0, lane);
    {
      // x tile -> planes 0, split by the staging thread; zero up to the next multiple of 32 columns
      const bool vx = is_vec_ok(a.x, a.ldx) && ((n.dims[0] & 3) == 0);
      const int kp4 = ((n.dims[0] + 31) & ~31) >> 2;
      // (four vectors of the tile requested before the first is split: a thread's trips through this
      //  loop were four to eight DEPENDENT memory round trips — 4.5 us in front of layer 1's barrier)
      for (int e0 = tid; e0 < ROWS * kp4; e0 += 4 * 512) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + j * 512;
          const int r = e / kp4, c = (e - r * kp4) * 4;
          const bool ok = e < ROWS * kp4 && (m0 + r) < a.B;
          if (vx) v[j] = ld4_or_zero(a.x, (int64_t)(m0 + r) * a.ldx + c, ok && c < n.dims[0]);
          else v[j] = guarded_load4(a.x, (int64_t)(m0 + r) * a.ldx, ok, c, n.dims[0]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + j * 512;
          const int r = e / kp4, c = (e - r * kp4) * 4;
          if (e < ROWS * kp4) rs_store_planes4(pl[0], ROWS, r, c, v[j]);
        }
      }
    }
    for (int l = 0; l < n.L; ++l) {
      const int K = n.dims[l], N = n.dims[l + 1];
      const int nt = (N + 15) >> 4, nks = wsp16_nks(K);
      f32x4v accm[2][RT], accs[2][RT];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n.bias[l]) b = guarded_load4(n.bias[l], 0, true, u0 + 16 * t, N);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          accm[t][rt][0] = b.x; accm[t][rt][1] = b.y; accm[t][rt][2] = b.z; accm[t][rt][3] = b.w;
          accs[t][rt][0] = accs[t][rt][1] = accs[t][rt][2] = accs[t][rt][3] = 0.f;
        }
      }
      const bool last = l == n.L - 1;
      // A narrow last layer (one pair of unit tiles: PPO's 16 action logits, a critic's single value)
      // used to be wave 0's alone — all k-steps of the layer behind a two-deep ring, ~3 us of exposed
      // memory round trips with seven waves waiting at the barrier.  Its k-steps are dealt out to the
      // eight waves instead (wave w: k-steps w, w + 8, ...); the partial tiles meet in the plane
      // buffer this layer does not read (the input of the layer before: dead) and wave 0 adds them
      // in wave order — deterministic, rounded differently from one long chain.
      const bool kdeal = last && nt <= 2 && nks >= 4;
      if (l == 0 && pre0) {
      } else if (kdeal) rs_fill(R, n.Wsp[l], nks, 0, nt, lane, wave, 8);
      else if (tile0 < nt) rs_fill(R, n.Wsp[l], nks, tile0, nt, lane);
      __syncthreads();
      vm_drain();   // (stores of the epilogue before + the ring prefetched under the barrier: see vm_drain)
      PA_STAMP(a.prof, pwg, wave, 1 + 2 * l);   // layer l: operands staged
      if (kdeal) {
        // (the bias rides wave 0's partial tile: the other waves' accumulators start from the bias
        //  of their OWN columns, which lie beyond the layer's width — zeros)
        rs_gemm<RT>(accm, accs, R, n.Wsp[l], nks, 0, nt, pl[l & 1], lane, wave, 8);
        float* part = reinterpret_cast<float*>(pl[(l + 1) & 1]);   // [8 waves][2 t][RT][64 lanes] float4
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            lds_st4(part + (((wave * 2 + t) * RT + rt) * 64 + lane) * 4,
                    make_float4(accm[t][rt][0] + accs[t][rt][0], accm[t][rt][1] + accs[t][rt][1],
                                accm[t][rt][2] + accs[t][rt][2], accm[t][rt][3] + accs[t][rt][3]));
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wave == 0) {
              sum = lds_ld4(part + (((0 * 2 + t) * RT + rt) * 64 + lane) * 4);
              for (int w = 1; w < 8; ++w) {
                const float4 q = lds_ld4(part + (((w * 2 + t) * RT + rt) * 64 + lane) * 4);
                sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
              }
            }
            // (waves 1-7 go on with exact zeros: their own 32 columns lie beyond the layer's width)
            accm[t][rt][0] = sum.x; accm[t][rt][1] = sum.y; accm[t][rt][2] = sum.z; accm[t][rt][3] = sum.w;
            accs[t][rt][0] = accs[t][rt][1] = accs[t][rt][2] = accs[t][rt][3] = 0.f;
          }
      } else if (tile0 < nt) {
        rs_gemm<RT>(accm, accs, R, n.Wsp[l], nks, tile0, nt, pl[l & 1], lane);
      }
      PA_STAMP(a.prof, pwg, wave, 2 + 2 * l);   // layer l: GEMM done
      const bool relu = (n.relu >> l) & 1;
      __bf16* nxt = pl[(l + 1) & 1];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int lr = r16 + 16 * rt;
        const int row = m0 + lr;
        const bool rok = row < a.B;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int u = u0 + 16 * t;
          // bias + main products first, the five small classes added once at the end
          float4 v = make_float4(accm[t][rt][0] + accs[t][rt][0], accm[t][rt][1] + accs[t][rt][1],
                                 accm[t][rt][2] + accs[t][rt][2], accm[t][rt][3] + accs[t][rt][3]);
          if (relu) v = make_float4(relu_keep_nan(v.x), relu_keep_nan(v.y), relu_keep_nan(v.z),
                                    relu_keep_nan(v.w));
          fmask |= (unsigned long long)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) |
                                        (v.w > 0.f ? 8u : 0u)) << (16 * l + 8 * rt + 4 * t);
          if (!last) {
            // (every wave writes its 32 columns: units beyond N are exact zeros — zero weights,
            //  zero bias — i.e. the next layer's zero-padded k-steps)
            rs_store_planes4(nxt, ROWS, lr, u, v);
            if (rok && n.act[l]) store4_guarded(n.act[l], (int64_t)row * N, u, N, (N & 3) == 0, v);
          } else {
            if (u < PH - 4) lds_st4(otile + lr * PH + u, v);
            if (rok && n.out)
              store4_guarded(n.out, (int64_t)row * n.ldo, u, N,
                             is_vec_ok(n.out, n.ldo) && (N & 3) == 0, v);
          }
        }
      }
      if (l < 2) PA_STAMP(a.prof, pwg, wave, 13 + l);   // epilogue of layer l done (before the barrier)
    }
  }
  const float* in = xs;
  int pin = P0;
  if constexpr (!SPLITF) {
  // ---------------------------------------------------------------- forward (mlp_rowfwd_kernel)
  {
    const bool vx = is_vec_ok(a.x, a.ldx) && ((n.dims[0] & 3) == 0);
    const int c4 = (P0 - 4) >> 2;
    for (int e0 = tid; e0 < ROWS * c4; e0 += 4 * 512) {   // (four vectors in flight: see the split form)
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 512;
        const int r = e / c4, c = (e - r * c4) * 4;
        const bool ok = e < ROWS * c4 && (m0 + r) < a.B;
        if (vx) v[j] = ld4_or_zero(a.x, (int64_t)(m0 + r) * a.ldx + c, ok && c < n.dims[0]);
        else v[j] = guarded_load4(a.x, (int64_t)(m0 + r) * a.ldx, ok, c, n.dims[0]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 512;
        const int r = e / c4, c = (e - r * c4) * 4;
        if (e < ROWS * c4) lds_st4(xs + r * P0 + c, v[j]);
      }
    }
  }
  for (int l = 0; l < n.L; ++l) {
    const int K = n.dims[l], N = n.dims[l + 1];
    const int nt = (N + 15) >> 4;
    f32x4v acc[2][RT];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n.bias[l]) b = guarded_load4(n.bias[l], 0, true, u0 + 16 * t, N);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[t][rt][0] = b.x; acc[t][rt][1] = b.y; acc[t][rt][2] = b.z; acc[t][rt][3] = b.w;
      }
    }
    // (waves whose unit tiles lie beyond the layer's width skip the loop: a 256 -> 16 head used to
    //  cost every wave a full layer of MFMAs on zero weights — 8.6 us of PPO's 69 us launch)
    constexpr int PDW = RT == 2 ? RS_PD2 : RS_PD;
    RowW<PDW> R;
    if (tile0 < nt) rowsN_fill<PDW>(R, n.Wf[l], wf16_nkg(K), tile0, nt, lane);
    __syncthreads();
    vm_drain();   // (stores of the epilogue before + the ring prefetched under the barrier: see vm_drain)
    PA_STAMP(a.prof, pwg, wave, 1 + 2 * l);   // layer l: operands staged
    if (tile0 < nt)
      rowsN_gemm<PDW, RT>(acc, R, n.Wf[l], wf16_nkg(K), tile0, nt, in + r16 * pin + 4 * qd, pin, lane);
    PA_STAMP(a.prof, pwg, wave, 2 + 2 * l);   // layer l: GEMM done
    const bool last = l == n.L - 1;
    const bool relu = (n.relu >> l) & 1;
    float* nxt = hb[l & 1];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int lr = r16 + 16 * rt;
      const int row = m0 + lr;
      const bool rok = row < a.B;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int u = u0 + 16 * t;
        float4 v = make_float4(acc[t][rt][0], acc[t][rt][1], acc[t][rt][2], acc[t][rt][3]);
        if (relu) v = make_float4(relu_keep_nan(v.x), relu_keep_nan(v.y), relu_keep_nan(v.z),
                                  relu_keep_nan(v.w));
        fmask |= (unsigned long long)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) |
                                      (v.w > 0.f ? 8u : 0u)) << (16 * l + 8 * rt + 4 * t);
        // (the output tile stays in LDS as well: the head reads it from there)
        if (u < PH - 4) lds_st4(nxt + lr * PH + u, v);
        if (!last) {
          if (rok && n.act[l]) store4_guarded(n.act[l], (int64_t)row * N, u, N, (N & 3) == 0, v);
        } else if (rok && n.out) {
          store4_guarded(n.out, (int64_t)row * n.ldo, u, N,
                         is_vec_ok(n.out, n.ldo) && (N & 3) == 0, v);
        }
      }
    }
    in = nxt;
    pin = PH;
  }
  }   // !SPLITF
  __syncthreads();
  PA_STAMP(a.prof, pwg, wave, 9);             // forward done
  if constexpr (!SPLITF) {
    if (hd.kind == RS_HEAD_WMSE1 && hd.lin_x && n.L >= 2) {
      const int D = n.dims[n.L - 1] + 1;
      const int W = hd.ldR > hd.ldX ? hd.ldR : hd.ldX;
      const float* ft = hb[(n.L - 2) & 1];      // [ROWS][PH] features of this tile (post-activation)
      for (int e = tid; e < ROWS * W; e += 512) {
        const int r = e / W, j = e - r * W;
        const int b = m0 + r;
        if (b >= a.B) continue;
        const float x = (j == 0) ? 1.0f : (j < D ? lds_ld(ft + r * PH + (j - 1)) : 0.f);
        if (j < hd.ldX) hd.lin_x[(int64_t)b * hd.ldX + j] = x;
        if (j < hd.ldR) hd.lin_r[(int64_t)b * hd.ldR + j] = j < D ? x : (j == D ? hd.target[b] : 0.f);
      }
    }
  }
  // ---------------------------------------------------------------- head: d_out tile into hb[0]
  const float* ot = SPLITF ? otile : hb[(n.L - 1) & 1];   // [ROWS][PH] network output of this tile
  const int DL = n.dims[n.L];
  float* va = scr;
  float* vb = scr + 512;
  float* red = scr + 3 * 512;
  float dval = 0.f, part0 = 0.f, part1 = 0.f;
  int hr = 0, hj = 0;
  bool hlive = false;
  if (hd.kind == RS_HEAD_PPO) {
    // one thread per (row, action) — ppo_actor_elem_kernel with ROWS rows per workgroup
    const int A = DL;
    hr = tid / A; hj = tid - hr * A;
    const int b = m0 + hr;
    hlive = hr < ROWS && b < a.B;
    const float lo = 1.0f - hd.eps, hi = 1.0f + hd.eps;
    const float z = (hr < ROWS) ? lds_ld(ot + hr * PH + hj) : 0.f;
    const float ar = hlive ? hd.arep[(int64_t)b * hd.lda + hj] : 0.f;
    const float g = hlive ? hd.gae[b] : 0.f;
    const float pold = hlive ? hd.p_old[b] : 1.f;
    const int base = hr * A;
    if (hr < ROWS) va[tid] = z;
    __syncthreads();
    float m = 0.f, s = 0.f, p = 0.f;
    if (hlive) {
      m = va[base];
      for (int k = 1; k < A; ++k) m = fmaxf(m, va[base + k]);
    }
    const float e = expf(z - m);
    if (hr < ROWS) vb[tid] = e;
    __syncthreads();
    if (hlive)
      for (int k = 0; k < A; ++k) s += vb[base + k];
    const float y = hlive ? e / s : 0.f;
    if (hr < ROWS) va[tid] = y * ar;
    __syncthreads();
    if (hlive) {
      for (int k = 0; k < A; ++k) p += va[base + k];
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
      dval = y * (dp * ar - dot);
      if (hj == 0) {
        part0 = -fminf(s1, s2);
        part1 = p;
        st_through(hd.p_rows + b, p);
      }
    }
  } else if (hd.kind == RS_HEAD_DSAC_ACTOR || hd.kind == RS_HEAD_DSAC_TARGET) {
    // one thread per (row, action) — dsac_elem_kernel with ROWS rows per workgroup
    const int A = DL;
    float* vc = scr + 2 * 512;
    hr = tid / A; hj = tid - hr * A;
    const int b = m0 + hr;
    hlive = hr < ROWS && b < a.B;
    const float alpha = hd.alpha[0];
    const float inv_n = 1.0f / ((float)a.B * (float)A);
    const int base = hr * A;
    const float z = (hr < ROWS) ? lds_ld(ot + hr * PH + hj) : 0.f;
    float q = 0.f;
    if (hlive && !(hd.mask && hd.mask[(int64_t)b * A + hj]))
      q = fminf(hd.q1[(int64_t)b * A + hj], hd.q2[(int64_t)b * A + hj]);
    if (hr < ROWS) va[tid] = z;
    __syncthreads();
    float m = 0.f, s = 0.f;
    if (hlive) {
      m = va[base];
      for (int k = 1; k < A; ++k) m = fmaxf(m, va[base + k]);
    }
    const float e = expf(z - m);
    if (hr < ROWS) vb[tid] = e;
    __syncthreads();
    if (hlive)
      for (int k = 0; k < A; ++k) s += vb[base + k];
    const float p = hlive ? e / s : 0.f;
    if (hd.kind == RS_HEAD_DSAC_TARGET) {
      if (hr < ROWS) va[tid] = (q - alpha * logf(p + 1e-8f)) * p;
      __syncthreads();
      if (hlive && hj == 0) {
        float v = 0.f;
        for (int k = 0; k < A; ++k) v += va[base + k];
        const float lv = 1.0f - (hd.term[b] ? 1.0f : 0.0f);
        hd.y[b] = __fadd_rn(__fmul_rn(__fmul_rn(v, hd.gamma), lv), hd.reward[b]);
      }
      hlive = false;   // no gradient tile
    } else {
      const float lp = logf(p + 1e-8f);
      const float f = alpha * lp - q;
      const float g = (f + p * (alpha / (p + 1e-8f))) * inv_n;   // dL/dP_j
      if (hr < ROWS) {
        va[tid] = p * f;
        vc[tid] = g * p;
      }
      __syncthreads();            // (every thread is past its reads of vb: the barrier above the p line)
      if (hr < ROWS) vb[tid] = p * lp;
      float dot = 0.f;
      if (hlive)
        for (int k = 0; k < A; ++k) dot += vc[base + k];
      dval = p * (g - dot);
      __syncthreads();
      if (hlive && hj == 0) {
        float h = 0.f;
        for (int k = 0; k < A; ++k) {
          part0 += va[base + k];
          h += vb[base + k];
        }
        hd.h_out[b] = h;
      }
    }
  } else {   // RS_HEAD_MSE: one thread per row, output column 0
    hr = tid; hj = 0;
    const int b = m0 + hr;
    hlive = hr < ROWS && b < a.B;
    if (hlive && hd.kind == RS_HEAD_WMSE1) {
      // wmse_kernel's expressions with w = 1 and sum w = B
      const float p = wloss_act(lds_ld(ot + hr * PH), hd.out_act);
      if (hd.out_post) hd.out_post[b] = p;
      dval = wloss_grad(p, hd.target[b], 1.0f, (float)a.B, hd.loss_kind, hd.out_act);
      part0 = wloss_value(p, hd.target[b], hd.loss_kind) * 1.0f;
      part1 = p;
    } else if (hlive) {
      const float d = __fsub_rn(lds_ld(ot + hr * PH), hd.target[b]);
      dval = __fmul_rn(hd.grad_scale, d);
      part0 = d * d;
    }
  }
  __syncthreads();   // every read of the output tile is done: hb[0] may be overwritten
  // SPLITF + split_bwd: the backward GEMMs are bf16x3 products as well — d_out goes into plane
  // buffer 0 as three bf16 planes (zero up to the next multiple of 32 columns: whole k-steps)
  bool sbw = false;
  if constexpr (SPLITF) sbw = a.split_bwd != 0;
  if (sbw) {
    if constexpr (SPLITF) {
      const int c8 = ((DL + 31) & ~31) >> 3;
      for (int e = tid; e < 3 * ROWS * c8; e += 512) {
        const int pr = e / c8, c = (e - pr * c8) * 8;      // pr = plane * ROWS + row
        lds_st4(reinterpret_cast<float*>(pl[0] + (size_t)pr * RS_PP + c), make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
  } else {
    const int c4 = (rp_pad(DL) - 4) >> 2;
    for (int e = tid; e < ROWS * c4; e += 512) {
      const int r = e / c4, c = (e - r * c4) * 4;
      lds_st4(hb[0] + r * PH + c, make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
  __syncthreads();
  if (hlive) {
    if (sbw) {
      if constexpr (SPLITF) {
        __bf16 s0, s1, s2;
        split3(dval, s0, s1, s2);
        pl[0][((size_t)0 * ROWS + hr) * RS_PP + hj] = s0;
        pl[0][((size_t)1 * ROWS + hr) * RS_PP + hj] = s1;
        pl[0][((size_t)2 * ROWS + hr) * RS_PP + hj] = s2;
      }
    } else {
      hb[0][hr * PH + hj] = dval;
    }
    hd.d_out[(int64_t)(m0 + hr) * hd.ldd + hj] = dval;
  }
  {
    const float2 s01 = block_sum2_512(part0, part1, red);   // (its barriers also publish the tile)
    if (tid == 0) {
      float* pp = a.partials + ((int64_t)net * gridDim.x + blockIdx.x) * 2;
      st_through(pp, s01.x);
      st_through(pp + 1, s01.y);
    }
  }
  PA_STAMP(a.prof, pwg, wave, 10);            // head done
  // ---------------------------------------------------------------- backward, bf16x3
  // dz_{l} = (dz_{l+1} W_l) o relu': the forward's rs_gemm with W_l^T planes as the weights and the
  // dz planes as the activations (both split exactly: fp32 accuracy, §3.5).  Layers l >= 1 only —
  // the host sets split_bwd when no input gradient is asked for.
  if constexpr (SPLITF) {
    if (sbw) {
      int curp = 0;
      for (int l = nb.L - 1; l >= 1; --l) {
        const int K = nb.dims[l + 1], N = nb.dims[l];
        const int nt = (N + 15) >> 4, nks = wsp16_nks(K);
        const bool mask = (nb.relu >> (l - 1)) & 1;
        __bf16* nxtp = pl[curp ^ 1];
        f32x4v accm[2][RT], accs[2][RT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            accm[t][rt][0] = accm[t][rt][1] = accm[t][rt][2] = accm[t][rt][3] = 0.f;
            accs[t][rt][0] = accs[t][rt][1] = accs[t][rt][2] = accs[t][rt][3] = 0.f;
          }
        RowWS R;
        if (tile0 < nt) rs_fill(R, nb.Wtsp[l], nks, tile0, nt, lane);
        const unsigned lm = (unsigned)(fmask >> (16 * (l - 1))) & 0xffffu;
        __syncthreads();
        vm_drain();   // (stores of the epilogue before + the ring prefetched under the barrier: see vm_drain)
        if (tile0 < nt) rs_gemm<RT>(accm, accs, R, nb.Wtsp[l], nks, tile0, nt, pl[curp], lane);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int lr = r16 + 16 * rt;
          const int row = m0 + lr;
          const bool rok = row < a.B;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int u = u0 + 16 * t;
            float4 v = make_float4(accm[t][rt][0] + accs[t][rt][0], accm[t][rt][1] + accs[t][rt][1],
                                   accm[t][rt][2] + accs[t][rt][2], accm[t][rt][3] + accs[t][rt][3]);
            if (mask) {
              const unsigned mt = lm >> (8 * rt + 4 * t);
              v.x = (mt & 1u) ? v.x : 0.f; v.y = (mt & 2u) ? v.y : 0.f;
              v.z = (mt & 4u) ? v.z : 0.f; v.w = (mt & 8u) ? v.w : 0.f;
            }
            if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            // (units beyond N: zero weights -> exact zeros, the next layer's padded k-steps)
            if (l > 1) rs_store_planes4(nxtp, ROWS, lr, u, v);
            if (rok) store4_guarded(nb.dz[l], (int64_t)row * N, u, N, (N & 3) == 0, v);
          }
        }
        curp ^= 1;
      }
    }
  }
  // ---------------------------------------------------------------- backward (mlp_rowbwd_kernel)
  in = hb[0];
  int cur = 0;
  for (int l = nb.L - 1; l >= 0; --l) {
    if (sbw) break;
    if (l == 0 && !nb.d_x) break;
    const int K = nb.dims[l + 1], N = nb.dims[l];
    const int nt = (N + 15) >> 4;
    const bool mask = l > 0 && ((nb.relu >> (l - 1)) & 1);
    float* nxt = hb[cur ^ 1];
    constexpr int PDW = RT == 2 ? RS_PD2 : RS_PD;
    RowW<PDW> R;
    if (tile0 < nt) rowsN_fill<PDW>(R, nb.Wtf[l], wf16_nkg(K), tile0, nt, lane);
    // the ReLU masks of the first chunk: this lane's own outputs of the forward pass (fmask)
    const unsigned lm = l > 0 ? (unsigned)(fmask >> (16 * (l - 1))) & 0xffffu : 0u;
    __syncthreads();
    vm_drain();   // (stores of the epilogue before + the ring prefetched under the barrier: see vm_drain)
    for (int c0 = 0; c0 < nt; c0 += 16) {
      f32x4v acc[2][RT];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[t][rt][0] = acc[t][rt][1] = acc[t][rt][2] = acc[t][rt][3] = 0.f;
      if (c0 > 0 && c0 + tile0 < nt) rowsN_fill<PDW>(R, nb.Wtf[l], wf16_nkg(K), c0 + tile0, nt, lane);
      if (c0 + tile0 < nt)
        rowsN_gemm<PDW, RT>(acc, R, nb.Wtf[l], wf16_nkg(K), c0 + tile0, nt, in + r16 * PH + 4 * qd, PH,
                            lane);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int lr = r16 + 16 * rt;
        const int row = m0 + lr;
        const bool rok = row < a.B;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int u = c0 * 16 + u0 + 16 * t;
          float4 v = make_float4(acc[t][rt][0], acc[t][rt][1], acc[t][rt][2], acc[t][rt][3]);
          if (l > 0) {
            if (mask) {
              if (c0 == 0) {
                const unsigned mt = lm >> (8 * rt + 4 * t);
                v.x = (mt & 1u) ? v.x : 0.f; v.y = (mt & 2u) ? v.y : 0.f;
                v.z = (mt & 4u) ? v.z : 0.f; v.w = (mt & 8u) ? v.w : 0.f;
              } else {
                const float4 hm = guarded_load4(nb.act[l - 1], (int64_t)row * N, rok, u, N);
                v.x = hm.x > 0.f ? v.x : 0.f; v.y = hm.y > 0.f ? v.y : 0.f;
                v.z = hm.z > 0.f ? v.z : 0.f; v.w = hm.w > 0.f ? v.w : 0.f;
              }
            }
            if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < PH - 4) lds_st4(nxt + lr * PH + u, v);
            if (rok) store4_guarded(nb.dz[l], (int64_t)row * N, u, N, (N & 3) == 0, v);
          } else if (rok) {
            store4_guarded(nb.d_x, (int64_t)row * nb.lddx, u, N,
                           is_vec_ok(nb.d_x, nb.lddx) && (N & 3) == 0, v);
          }
        }
      }
    }
    in = nxt;
    cur ^= 1;
  }
  PA_STAMP(a.prof, pwg, wave, 11);            // backward done
  // ---------------------------------------------------------------- losses: last workgroup
  __shared__ unsigned is_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have landed
  __syncthreads();
  if (tid == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    is_last = (__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
               total - 1u) ? 1u : 0u;
  }
  __syncthreads();
  PA_STAMP(a.prof, pwg, wave, 12);            // ticket taken
  if (!is_last) return;
  float both = 0.f;
  for (int k = 0; k < (int)gridDim.y; ++k) {
    const RowHead& h = a.head[k];
    const float* pp = a.partials + (int64_t)k * gridDim.x * 2;
    float p0 = 0.f, p1 = 0.f;
    for (unsigned i = tid; i < gridDim.x; i += 512) {   // fixed order: strided, then the tree
      p0 += ld_through(pp + 2 * i);
      p1 += ld_through(pp + 2 * i + 1);
    }
    const float2 s01 = block_sum2_512(p0, p1, red);
    const float s0 = s01.x, s1 = s01.y;
    float loss;
    if (h.kind == RS_HEAD_PPO) {
      // - entropy_scale * H(Categorical(p_batch)): the chosen-action probabilities of the whole
      // minibatch as ONE categorical (ppo.py:179-182; a detached scalar)
      float part_e = 0.f;
      const float tiny = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
      for (int i = tid; i < a.B; i += 512) {
        const float pr = ld_through(h.p_rows + i);
        const float pn = pr / s1;
        const float pc = fminf(fmaxf(pn, tiny), 1.0f - tiny);
        float lg = logf(pc);
        lg = fmaxf(lg, -3.4028234663852886e+38f);
        part_e += lg * pn;
      }
      const float ent = -block_sum_512(part_e, red);
      loss = s0 - h.ent_scale * ent;
    } else if (h.kind == RS_HEAD_DSAC_ACTOR) {
      loss = s0 * (1.0f / ((float)a.B * (float)a.fwd[k].dims[a.fwd[k].L]));
    } else if (h.kind == RS_HEAD_WMSE1) {
      loss = s0 / (float)a.B;
      if (tid == 0 && h.mean_out) *h.mean_out = s1 / (float)a.B;
    } else {
      loss = (s0 / (float)a.B) * h.loss_scale;
    }
    both += loss;
    if (tid == 0 && !a.sum_losses && a.losses) a.losses[k] = loss;
  }
  if (tid == 0) {
    if (a.sum_losses && a.losses) a.losses[0] = both;
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace pa
