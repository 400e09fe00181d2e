// The fused target-network tile (target_tile in dqn_kernels.hpp: same inputs, same prologue, same
// epilogue, same C layout) with its layer-2 product — 79 % of a DQN step's FLOPs — on the bf16
// matrix pipe at fp32 accuracy:
//
//   every fp32 operand x is split EXACTLY into three bf16 terms, x = hi + mid + lo
//   (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 significand bits = fp32's 24),
//   and  a b  is evaluated as the six products whose weight is >= 2^-24 of it,
//       a_hi b_hi | a_hi b_mid + a_mid b_hi | a_hi b_lo + a_lo b_hi + a_mid b_mid
//   each exact in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 — one
//   accumulator per magnitude class, so the small terms are never rounded against the large running
//   sum, and the three are added smallest first at the end.  What is dropped (mid lo, lo mid, lo lo)
//   is below 2^-24 |a b|: the result carries the rounding noise of an fp32 dot product and no more
//   (tools/split_mfma_bench.hip measures it against fp64; the parity tests hold the same 1e-5
//   Q-value tolerance they hold for the fp32-MFMA kernel).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s on the chip), the bf16
// form at 16x that; six bf16 products per fp32 product are 2.67x the fp32 matrix rate.
//
// Data: the target W2 as fragment-major split planes (w2sp_index below; kept current by the
// repack pass and by the optimizer epilogue that performs the soft update), the h1 tile as three
// bf16 planes in LDS (the split is done once, by the lane that produced the value).
// One workgroup (8 waves) per CU: 99 KB of LDS for the planes.
#pragma once
#include "dqn_kernels.hpp"

namespace pa {

// 8 bf16 of a split plane through a raw buffer load (like every operand fetch of these kernels: the
// compiler neither predicates nor hoists it — a plain load of the tile-invariant weight stream is
// lifted out of the persistent tile loop, all 48 slots of it, and spills)
__device__ __forceinline__ bf16x8 ld_bf16x8(const void* base, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(base), (int)byte_off, 0, 0);
  return __builtin_bit_cast(bf16x8, v);
}

inline size_t target_split_smem_bytes() {
  return (size_t)3 * T_ROWS * TS_LDP * 2 + sizeof(float) * (8 * 64 + 64) + 16;
}

// Same contract as target_tile<32, true>: the host has checked target_fast_shape (16-byte aligned
// operands, AD <= 16 and a multiple of 4, H1 = H2 = 256).
// U[n][j] = sum_k W1s'[n][k] s'[b0 + j][k] + b1'[n] for the 32 transitions b0 .. b0 + 31 (nb of them
// real) as a bf16x3 product: the states split three ways into LDS planes (rows = transitions; the
// h1 planes' space), one 32-column MFMA tile per wave against the W1s' planes, one accumulator per
// magnitude class.  Returns this lane's 16 values: unit 32 wave + 8 (r >> 2) + 4 (lane >> 5) + (r & 3),
// transition lane & 31.  Deterministic per (unit, transition): the fused tile and u_split_kernel
// produce the same bits.  Ends with every wave still reading nothing: callers barrier before they
// reuse the planes.
template <int KS1>
__device__ __forceinline__ void split_u_columns(const TargetArgs& a, int b0, int nb, __bf16* planes,
                                                float (&u16)[16]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nq0 = wave * 32 + 4 * h;
    constexpr int ks1 = KS1, s4 = 4 * KS1;       // S = 16 KS1 (compile time: every offset an immediate)
    for (int e = tid; e < 32 * s4; e += 512) {
      const int t = e / s4, k = (e - t * s4) * 4;
      const float4 v = ld4_or_zero(a.next_state, (int64_t)(b0 + t) * a.ld_next + k, t < nb);
      const float xv[4] = {v.x, v.y, v.z, v.w};
      bf16x4 p[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __bf16 x0, x1, x2;
        split3(xv[j], x0, x1, x2);
        p[0][j] = x0; p[1][j] = x1; p[2][j] = x2;
      }
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
        *reinterpret_cast<bf16x4*>(planes + (size_t)(sp * T_ROWS + t) * TS_LDP + k) = p[sp];
    }
    // this lane's 16 bias values (added at the end), and the first W1s' k-steps
    float4 b1v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b1v[q] = ld4_or_zero(a.b1, nq0 + 8 * q, true);
    bf16x8 ring1[TS_RD][3];
    const unsigned w1base = (unsigned)((wave * ks1 * 3 * 64 + lane) * 16);
#pragma unroll
    for (int g = 0; g < TS_RD && g < ks1; ++g)
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
        ring1[g][sp] = ld_bf16x8(a.W1sp, w1base + (unsigned)(g * 3 + sp) * 1024u);
    __syncthreads();
    f32x16 cu[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) cu[c][r] = 0.f;
    const __bf16* sp0 = planes + (size_t)l31 * TS_LDP + 8 * h;
#pragma unroll
    for (int g = 0; g < ks1; ++g) {
      bf16x8 wa[3];
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) wa[sp] = ring1[g % TS_RD][sp];
      if (g + TS_RD < ks1) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          ring1[g % TS_RD][sp] = ld_bf16x8(a.W1sp, w1base + (unsigned)((g + TS_RD) * 3 + sp) * 1024u);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        bf16x8 b[3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          b[sp] = *reinterpret_cast<const bf16x8*>(sp0 + (size_t)sp * T_ROWS * TS_LDP + 16 * g);
        cu[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b[2], cu[2], 0, 0, 0);
        cu[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[2], b[0], cu[2], 0, 0, 0);
        cu[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], b[1], cu[2], 0, 0, 0);
        cu[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b[1], cu[1], 0, 0, 0);
        cu[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], b[0], cu[1], 0, 0, 0);
        cu[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b[0], cu[0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float bq4[4] = {b1v[q].x, b1v[q].y, b1v[q].z, b1v[q].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      u16[r] = ((cu[2][r] + cu[1][r]) + cu[0][r]) + bq4[j];
    }
  }
}

// The same product as a launch of its own: U[b][n] for B transitions, 32 per workgroup (the
// persistent remainder of a window reads U instead of forming it: its tiles are throughput work).
template <int KS1>
static __global__ __launch_bounds__(512, 2) void u_split_kernel(TargetArgs a, float* __restrict__ U) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_u[];   // the tile's plane layout
  const int b0 = (int)blockIdx.x * 32;
  const int nb = min(32, a.B - b0);
  float u16[16];
  split_u_columns<KS1>(a, b0, nb, reinterpret_cast<__bf16*>(smem_u), u16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  if (l31 < nb) {
    float* dst = U + (int64_t)(b0 + l31) * a.ldu + wave * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(dst + 8 * q) =
          make_float4(u16[4 * q], u16[4 * q + 1], u16[4 * q + 2], u16[4 * q + 3]);
  }
}

// KS1 > 0: the tile forms U itself from S = 16 KS1 state columns (a.W1sp etc. set); 0: U is read.
template <int KS1>
__device__ __forceinline__ void target_tile_split(const TargetArgs& a, int tile, unsigned char* smem) {
  __bf16* planes = reinterpret_cast<__bf16*>(smem);                        // [3][64][TS_LDP]
  float* qpart = reinterpret_cast<float*>(smem + (size_t)3 * T_ROWS * TS_LDP * 2);   // [8][64]
  float* qv = qpart + 8 * 64;                                              // [64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int b0 = tile * a.bpw;
  const int nb = min(a.bpw, a.B - b0);
  const int nrows = nb * a.A;
  const int nq0 = wave * 32 + 4 * h;     // this lane's hidden units: nq0 + 8*q + j, q,j in 0..3
  PA_STAMP(a.prof, tile, wave, 0);
  const float b3v = a.b3[0];
  unsigned pf_mask = 0, pf_term = 0;
  float pf_reward = 0.f;
  if (tid < nrows && a.mask)
    pf_mask = a.mask[(int64_t)(b0 + tid / a.A) * a.mask_bstride + tid % a.A];
  if (tid < nb && a.y) {
    pf_term = a.term[b0 + tid];
    pf_reward = a.reward[b0 + tid];
  }

  // ---- layer 1 (fp32 MFMA, K = AD <= 16): h1 = relu(U[b] + W1a' rep(b, i)), as in target_tile
  f32x16 acc[2];
  int64_t foff[2];
  bool fok[2];
  int rowb[2];     // tile-local transition of this lane's row (0 for rows past the tile)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int row = tm * 32 + l31;
    const bool rok = row < nrows;
    const int rr = rok ? row : 0;
    rowb[tm] = rr / a.A;
    const int bb = b0 + rowb[tm];
    fok[tm] = rok;
    foff[tm] = (int64_t)bb * a.feat_bstride + (int64_t)(rr % a.A) * a.AD;
    if constexpr (KS1 == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nq0 + 8 * q;
        const float4 u = ld4_or_zero(a.U, (int64_t)bb * a.ldu + n, rok);
        acc[tm][4 * q + 0] = u.x; acc[tm][4 * q + 1] = u.y;
        acc[tm][4 * q + 2] = u.z; acc[tm][4 * q + 3] = u.w;
      }
    }
  }
  if constexpr (KS1 > 0) {
    // ---- U[b] = W1s' s'[b] + b1' for the tile's <= 32 distinct transitions, in the tile: the states
    // split three ways into LDS planes (the h1 planes' space, rows = transitions), a bf16x3 product
    // with the W1s' planes (one 32-column MFMA tile: column j = transition j), then every row of the
    // tile fetches its transition's column with a lane shuffle.  (host: bpw <= 32, S % 16 == 0, S <= 256)
    float u16[16];
    split_u_columns<KS1>(a, b0, nb, planes, u16);
    // U[unit][transition = l31] of this lane -> acc[tm][unit] of the lane that owns row 32 tm + l31
    // (same lane half: a unit's half is a property of the unit)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) acc[tm][r] = __shfl(u16[r], rowb[tm] + 32 * h, 64);
    __syncthreads();   // every wave is done with the state planes: h1 may overwrite them
  }
  const int wcol = wave * 32 + l31;      // hidden unit this lane feeds as the A operand
  const int64_t woff = (int64_t)wcol * a.ldw1;
  float4 fx[2][2], fw[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = 8 * kk + 4 * h;
    fx[kk][0] = ld4_or_zero(a.feat, foff[0] + k, fok[0] && k < a.AD);
    fx[kk][1] = ld4_or_zero(a.feat, foff[1] + k, fok[1] && k < a.AD);
    fw[kk] = ld4_or_zero(a.W1a, woff + k, k < a.AD);
  }
  // ---- layer-2 weights: the first TS_RD k-steps of this wave's split planes
  bf16x8 ring[TS_RD][3];
  const unsigned wbase = (unsigned)((wave * TS_KS * 3 * 64 + lane) * 16);   // bytes; slot stride 1 KiB
#pragma unroll
  for (int g = 0; g < TS_RD; ++g)
#pragma unroll
    for (int s = 0; s < 3; ++s) ring[g][s] = ld_bf16x8(a.W2sp, wbase + (unsigned)(g * 3 + s) * 1024u);
  PA_STAMP(a.prof, tile, wave, 1);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      acc[tm] = mfma32(fw[kk].x, fx[kk][tm].x, acc[tm]);
      acc[tm] = mfma32(fw[kk].y, fx[kk][tm].y, acc[tm]);
      acc[tm] = mfma32(fw[kk].z, fx[kk][tm].z, acc[tm]);
      acc[tm] = mfma32(fw[kk].w, fx[kk][tm].w, acc[tm]);
    }
  // h1 = relu(acc), split three ways by the lane that owns it -> the LDS planes [s][row][k]
  // (rows >= nrows hold relu(0 + 0) = 0 exactly: their U and feat loads returned zeros)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    __bf16* dst = planes + (size_t)(tm * 32 + l31) * TS_LDP + nq0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16x4 p[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __bf16 x0, x1, x2;
        split3(relu_keep_nan(acc[tm][4 * q + j]), x0, x1, x2);
        p[0][j] = x0; p[1][j] = x1; p[2][j] = x2;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
        *reinterpret_cast<bf16x4*>(dst + (size_t)s * T_ROWS * TS_LDP + 8 * q) = p[s];
    }
  }
  PA_STAMP(a.prof, tile, wave, 2);
  __syncthreads();
  PA_STAMP(a.prof, tile, wave, 3);

  // ---- layer 2: C[n][row] = sum_k W2'[n][k] h1[row][k], six bf16 products per k-step, one
  // accumulator per magnitude class
  f32x16 c3[2][3];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) c3[tm][c][r] = 0.f;
  float4 b2v[4], w3v[4];
  {
    const __bf16* bp0 = planes + (size_t)l31 * TS_LDP + 8 * h;
    const __bf16* bp1 = bp0 + (size_t)32 * TS_LDP;
    bf16x8 bq[2][2][3];   // the h1 operands run one k-step ahead of their use
    auto ldb = [&](int g, bf16x8 (&b)[2][3]) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        b[0][s] = *reinterpret_cast<const bf16x8*>(bp0 + (size_t)s * T_ROWS * TS_LDP + 16 * g);
        b[1][s] = *reinterpret_cast<const bf16x8*>(bp1 + (size_t)s * T_ROWS * TS_LDP + 16 * g);
      }
    };
    ldb(0, bq[0]);
#pragma unroll
    for (int g = 0; g < TS_KS; ++g) {
      bf16x8 wa[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) wa[s] = ring[g % TS_RD][s];
      if (g + TS_RD < TS_KS) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
          ring[g % TS_RD][s] = ld_bf16x8(a.W2sp, wbase + (unsigned)((g + TS_RD) * 3 + s) * 1024u);
      } else if (g == TS_KS - 3) {
        // layer-3 constants of this lane's hidden units, into the registers of the drained ring
#pragma unroll
        for (int q = 0; q < 4; ++q) b2v[q] = ld4_or_zero(a.b2, nq0 + 8 * q, true);
      } else if (g == TS_KS - 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w3v[q] = ld4_or_zero(a.w3, nq0 + 8 * q, true);
      }
      if (g + 1 < TS_KS) ldb(g + 1, bq[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // loads stay here, ahead of this k-step's MFMAs
      bf16x8 (&b)[2][3] = bq[g & 1];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        c3[tm][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b[tm][2], c3[tm][2], 0, 0, 0);
        c3[tm][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[2], b[tm][0], c3[tm][2], 0, 0, 0);
        c3[tm][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], b[tm][1], c3[tm][2], 0, 0, 0);
        c3[tm][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b[tm][1], c3[tm][1], 0, 0, 0);
        c3[tm][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], b[tm][0], c3[tm][1], 0, 0, 0);
        c3[tm][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b[tm][0], c3[tm][0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // nothing of the next k-step is hoisted above this one
    }
  }
  PA_STAMP(a.prof, tile, wave, 4);
  // ---- layer 3: in-lane over this lane's 16 hidden units, then the other half, then the waves
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float bq4[4] = {b2v[q].x, b2v[q].y, b2v[q].z, b2v[q].w};
      const float wq4[4] = {w3v[q].x, w3v[q].y, w3v[q].z, w3v[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        const float z = (c3[tm][2][r] + c3[tm][1][r]) + c3[tm][0][r];   // smallest classes first
        sum = fmaf(relu_keep_nan(z + bq4[j]), wq4[j], sum);
      }
    }
    sum += __shfl_xor(sum, 32);
    if (h == 0) qpart[wave * 64 + tm * 32 + l31] = sum;
  }
  PA_STAMP(a.prof, tile, wave, 5);
  __syncthreads();
  PA_STAMP(a.prof, tile, wave, 6);
  if (tid < T_ROWS) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) q += qpart[w * 64 + tid];
    q += b3v;
    if (a.q_all && tid < nrows) a.q_all[(int64_t)b0 * a.A + tid] = q;
    if (tid < nrows && pf_mask) q = -INFINITY;
    qv[tid] = q;
  }
  __syncthreads();
  if (tid < nb) {
    const int bb = b0 + tid;
    float m = qv[tid * a.A];
    int mi = 0;
    for (int i = 1; i < a.A; ++i) {
      const float x = qv[tid * a.A + i];
      const bool take = (x > m || x != x) && !(m != m);  // first maximum; the first NaN wins
      m = take ? x : m;
      mi = take ? i : mi;
    }
    if (a.argmax) {
      a.argmax[bb] = mi;
      if (a.choice_rep) {
        const float* src = a.feat + (int64_t)bb * a.feat_bstride + (int64_t)mi * a.AD;
        for (int j = 0; j < a.AD; ++j) a.choice_rep[(int64_t)bb * a.AD + j] = src[j];
      }
    }
    if (a.next_v) a.next_v[bb] = m;
    if (a.y) {
      // (next_v * gamma * (1 - terminated.float())) + reward, one rounding per op
      const float live = 1.0f - (pf_term ? 1.0f : 0.0f);
      const float t0 = __fmul_rn(m, a.gamma);
      const float t1 = __fmul_rn(t0, live);
      publish_y(a.y + bb, __fadd_rn(t1, pf_reward));
    }
  }
  PA_STAMP(a.prof, tile, wave, 7);
  if (a.prof && (threadIdx.x & 63) == 0) a.prof[((int64_t)tile * 8 + wave) * 16 + 8] = cu_key();
}

// Classic grid (tile = blockIdx.x) or, with a.tile_ctr, persistent work-stealing tiles that stay
// off the CUs reserved for the online chain — the two modes of target_fused_kernel.
template <int KS1>
static __global__ __launch_bounds__(512, 2) void target_split_kernel(TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
  if (a.tile_ctr == nullptr) {
    if ((int)blockIdx.x < a.prio_tiles) __builtin_amdgcn_s_setprio(3);
    target_tile_split<KS1>(a, blockIdx.x, smem_split);
    return;
  }
  __shared__ int next_tile;
  if (a.reserved && a.reserved[cu_key()]) return;
  if (threadIdx.x == 0) {
    next_tile = atomicAdd(a.tile_ctr, 1);
    if (a.dbg_workers && next_tile < a.ntiles) atomicAdd(a.dbg_workers, 1);
  }
  __syncthreads();
  int tile = next_tile;
  while (tile < a.ntiles) {
    int ahead = 0;
    if (threadIdx.x == 0) ahead = atomicAdd(a.tile_ctr, 1);   // in flight under this tile
    target_tile_split<KS1>(a, tile, smem_split);
    if (threadIdx.x == 0) next_tile = ahead;
    __syncthreads();  // publishes next_tile; the planes are reused by the next tile
    tile = next_tile;
  }
}

// ---------------------------------------------------------------------------------------------
// The same tile for 32 rows and FOUR waves (256 threads): two workgroups share a CU (51 KB of LDS
// each), so one workgroup's prologue and epilogue — U / table loads, the K = 16 layer, the three-way
// split, the layer-3 reductions, the barriers: 7 of the 64-row tile's 13 us with one workgroup per
// CU — run beside the other's main loop (+11 % on DoubleDQN's stand-alone
// passes, +7 % on the live launches of the DQN loop; nothing on the isolated rate, which is bound by
// issue — MFMA and VALU share a port — and by the doubled weight traffic per row, not by latency).  Wave w owns unit
// blocks 2 w and 2 w + 1 (64 hidden units) of the one 32-row block.  Same arithmetic in the same
// order per (row, unit): layer 1 and the six-product k loop are per element, and the layer-3 sums
// are kept per 32-unit block and added over the eight blocks in block order exactly as the 64-row
// tile adds its eight waves' partials — the two kernels produce the same bits
// (tests/test_gpu_dqn.py::test_target_split_tile_shapes_are_bitwise_identical).
// Reads U (no fused first layer).  a.bpw = 32 / A, a.ntiles accordingly (launch_target_split32).
// ---------------------------------------------------------------------------------------------
constexpr int TS32_ROWS = 32;
constexpr int TS32_RD = 2;             // k-steps of weights in flight per wave and unit block
inline size_t target_split32_smem_bytes() {
  return (size_t)3 * TS32_ROWS * TS_LDP * 2 + sizeof(float) * (8 * TS32_ROWS + TS32_ROWS) + 16;
}
__device__ __forceinline__ void target_tile_split32(const TargetArgs& a, int tile, unsigned char* smem) {
  __bf16* planes = reinterpret_cast<__bf16*>(smem);                              // [3][32][TS_LDP]
  float* qpart = reinterpret_cast<float*>(smem + (size_t)3 * TS32_ROWS * TS_LDP * 2);   // [8 blocks][32]
  float* qv = qpart + 8 * TS32_ROWS;                                             // [32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int b0 = tile * a.bpw;
  const int nb = min(a.bpw, a.B - b0);
  const int nrows = nb * a.A;
  const float b3v = a.b3[0];
  unsigned pf_mask = 0, pf_term = 0;
  float pf_reward = 0.f;
  if (tid < nrows && a.mask)
    pf_mask = a.mask[(int64_t)(b0 + tid / a.A) * a.mask_bstride + tid % a.A];
  if (tid < nb && a.y) {
    pf_term = a.term[b0 + tid];
    pf_reward = a.reward[b0 + tid];
  }
  // ---- layer 1 (fp32 MFMA, K = AD <= 16): h1 = relu(U[b] + W1a' rep(b, i))
  const bool rok = l31 < nrows;
  const int rr = rok ? l31 : 0;
  const int bb = b0 + rr / a.A;
  const int64_t foff = (int64_t)bb * a.feat_bstride + (int64_t)(rr % a.A) * a.AD;
  f32x16 acc[2];
  int nq0[2];
#pragma unroll
  for (int ub = 0; ub < 2; ++ub) {
    nq0[ub] = (2 * wave + ub) * 32 + 4 * h;    // this lane's hidden units of block ub: nq0 + 8 q + j
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 u = ld4_or_zero(a.U, (int64_t)bb * a.ldu + nq0[ub] + 8 * q, rok);
      acc[ub][4 * q + 0] = u.x; acc[ub][4 * q + 1] = u.y;
      acc[ub][4 * q + 2] = u.z; acc[ub][4 * q + 3] = u.w;
    }
  }
  float4 fx[2], fw[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = 8 * kk + 4 * h;
    fx[kk] = ld4_or_zero(a.feat, foff + k, rok && k < a.AD);
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
      fw[ub][kk] = ld4_or_zero(a.W1a, (int64_t)((2 * wave + ub) * 32 + l31) * a.ldw1 + k, k < a.AD);
  }
  // ---- layer-2 weights: the first k-steps of this wave's two unit blocks
  bf16x8 ring[TS32_RD][2][3];
  unsigned wbase[2];
#pragma unroll
  for (int ub = 0; ub < 2; ++ub) {
    wbase[ub] = (unsigned)(((2 * wave + ub) * TS_KS * 3 * 64 + lane) * 16);   // bytes; slot stride 1 KiB
#pragma unroll
    for (int g = 0; g < TS32_RD; ++g)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        ring[g][ub][s] = ld_bf16x8(a.W2sp, wbase[ub] + (unsigned)(g * 3 + s) * 1024u);
  }
#pragma unroll
  for (int ub = 0; ub < 2; ++ub)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      acc[ub] = mfma32(fw[ub][kk].x, fx[kk].x, acc[ub]);
      acc[ub] = mfma32(fw[ub][kk].y, fx[kk].y, acc[ub]);
      acc[ub] = mfma32(fw[ub][kk].z, fx[kk].z, acc[ub]);
      acc[ub] = mfma32(fw[ub][kk].w, fx[kk].w, acc[ub]);
    }
  // h1 = relu(acc), split three ways by the lane that owns it -> the LDS planes [s][row][k]
#pragma unroll
  for (int ub = 0; ub < 2; ++ub) {
    __bf16* dst = planes + (size_t)l31 * TS_LDP + nq0[ub];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16x4 p[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __bf16 x0, x1, x2;
        split3(relu_keep_nan(acc[ub][4 * q + j]), x0, x1, x2);
        p[0][j] = x0; p[1][j] = x1; p[2][j] = x2;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
        *reinterpret_cast<bf16x4*>(dst + (size_t)s * TS32_ROWS * TS_LDP + 8 * q) = p[s];
    }
  }
  __syncthreads();
  // ---- layer 2: six bf16 products per k-step and unit block, one accumulator per magnitude class
  f32x16 c3[2][3];
#pragma unroll
  for (int ub = 0; ub < 2; ++ub)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) c3[ub][c][r] = 0.f;
  {
    const __bf16* bp0 = planes + (size_t)l31 * TS_LDP + 8 * h;
    bf16x8 bq[2][3];   // the h1 operands run one k-step ahead of their use
    auto ldb = [&](int g, bf16x8 (&b)[3]) {
#pragma unroll
      for (int s = 0; s < 3; ++s)
        b[s] = *reinterpret_cast<const bf16x8*>(bp0 + (size_t)s * TS32_ROWS * TS_LDP + 16 * g);
    };
    ldb(0, bq[0]);
#pragma unroll
    for (int g = 0; g < TS_KS; ++g) {
      bf16x8 wa[2][3];
#pragma unroll
      for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          wa[ub][s] = ring[g % TS32_RD][ub][s];
          if (g + TS32_RD < TS_KS)
            ring[g % TS32_RD][ub][s] = ld_bf16x8(a.W2sp, wbase[ub] + (unsigned)((g + TS32_RD) * 3 + s) * 1024u);
        }
      if (g + 1 < TS_KS) ldb(g + 1, bq[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // loads stay here, ahead of this k-step's MFMAs
      bf16x8 (&b)[3] = bq[g & 1];
#pragma unroll
      for (int ub = 0; ub < 2; ++ub) {
        c3[ub][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ub][0], b[2], c3[ub][2], 0, 0, 0);
        c3[ub][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ub][2], b[0], c3[ub][2], 0, 0, 0);
        c3[ub][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ub][1], b[1], c3[ub][2], 0, 0, 0);
        c3[ub][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ub][0], b[1], c3[ub][1], 0, 0, 0);
        c3[ub][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ub][1], b[0], c3[ub][1], 0, 0, 0);
        c3[ub][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ub][0], b[0], c3[ub][0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // nothing of the next k-step is hoisted above this one
    }
  }
  // ---- layer 3: per 32-unit block — in-lane over this lane's 16 units, then the other half — so that
  // the eight block partials of a row are the eight wave partials of the 64-row tile
#pragma unroll
  for (int ub = 0; ub < 2; ++ub) {
    float4 b2v[4], w3v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b2v[q] = ld4_or_zero(a.b2, nq0[ub] + 8 * q, true);
      w3v[q] = ld4_or_zero(a.w3, nq0[ub] + 8 * q, true);
    }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float bq4[4] = {b2v[q].x, b2v[q].y, b2v[q].z, b2v[q].w};
      const float wq4[4] = {w3v[q].x, w3v[q].y, w3v[q].z, w3v[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        const float z = (c3[ub][2][r] + c3[ub][1][r]) + c3[ub][0][r];   // smallest classes first
        sum = fmaf(relu_keep_nan(z + bq4[j]), wq4[j], sum);
      }
    }
    sum += __shfl_xor(sum, 32);
    if (h == 0) qpart[(2 * wave + ub) * TS32_ROWS + l31] = sum;
  }
  __syncthreads();
  if (tid < TS32_ROWS) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) q += qpart[w * TS32_ROWS + tid];
    q += b3v;
    if (a.q_all && tid < nrows) a.q_all[(int64_t)b0 * a.A + tid] = q;
    if (tid < nrows && pf_mask) q = -INFINITY;
    qv[tid] = q;
  }
  __syncthreads();
  if (tid < nb) {
    const int bt = b0 + tid;
    float m = qv[tid * a.A];
    int mi = 0;
    for (int i = 1; i < a.A; ++i) {
      const float x = qv[tid * a.A + i];
      const bool take = (x > m || x != x) && !(m != m);  // first maximum; the first NaN wins
      m = take ? x : m;
      mi = take ? i : mi;
    }
    if (a.argmax) {
      a.argmax[bt] = mi;
      if (a.choice_rep) {
        const float* src = a.feat + (int64_t)bt * a.feat_bstride + (int64_t)mi * a.AD;
        for (int j = 0; j < a.AD; ++j) a.choice_rep[(int64_t)bt * a.AD + j] = src[j];
      }
    }
    if (a.next_v) a.next_v[bt] = m;
    if (a.y) {
      const float live = 1.0f - (pf_term ? 1.0f : 0.0f);
      const float t0 = __fmul_rn(m, a.gamma);
      const float t1 = __fmul_rn(t0, live);
      publish_y(a.y + bt, __fadd_rn(t1, pf_reward));
    }
  }
}

// Classic grid or persistent work-stealing tiles, as target_split_kernel; two workgroups per CU.
static __global__ __launch_bounds__(256, 2) void target_split32_kernel(TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_split32[];
  if (a.tile_ctr == nullptr) {
    if ((int)blockIdx.x < a.prio_tiles) __builtin_amdgcn_s_setprio(3);
    target_tile_split32(a, blockIdx.x, smem_split32);
    return;
  }
  __shared__ int next_tile32;
  if (a.reserved && a.reserved[cu_key()]) return;
  if (threadIdx.x == 0) next_tile32 = atomicAdd(a.tile_ctr, 1);
  __syncthreads();
  int tile = next_tile32;
  while (tile < a.ntiles) {
    int ahead = 0;
    if (threadIdx.x == 0) ahead = atomicAdd(a.tile_ctr, 1);   // in flight under this tile
    target_tile_split32(a, tile, smem_split32);
    if (threadIdx.x == 0) next_tile32 = ahead;
    __syncthreads();  // publishes next_tile; the planes are reused by the next tile
    tile = next_tile32;
  }
}

}  // namespace pa
