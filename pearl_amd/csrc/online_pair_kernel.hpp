// The online row pass of the DQN learner step on TWO workgroups per 16-row tile (round 5).
//
// online_rowpass_kernel (online_kernels.hpp) runs a 16-row tile of the batch on ONE workgroup:
// 64 workgroups at B = 1024 whose three GEMM phases are bound by the fp32 matrix rate of the 64
// CUs they occupy (8.7 us of v_mfma_f32_16x16x4_f32 in a 17 us launch) while the other CUs the
// chain owns idle.  Rows cannot be split further (a 16-row MFMA tile), but the UNITS can:
//
//   layer 1   h1 = relu(W1 x + b1)      both workgroups of a pair, all 256 units  (K = 144: cheap,
//                                        and layer 2 needs every h1 of the row on both sides)
//   layer 2   h2 = relu(W2 h1 + b2)     half h computes units [128 h, 128 h + 128)       (1/2)
//   head      q_h = w3[half] . h2[half] the two partial dot products are the ONLY thing the pair
//                                        has to exchange before the loss: 16 floats each way,
//                                        as data-tagged words (publish_y / consume_y), in flight
//                                        while the backward GEMM runs
//   G_h = s2[half] W2[half, :]          the backward GEMM, K = the 128 units of the half  (1/2)
//   dZ2[:, half] = dq s2[half]          whole (no partial sums)
//   dZ1_h = [h1 > 0] dq G_h             a PARTIAL of dZ1 = dZ1_0 + dZ1_1: the two are stored
//                                        interleaved per pair of units and the weight-gradient
//                                        kernel adds them as it loads its operand — one 16-byte
//                                        vector per two units (DwProblem::dz_pair): no exchange,
//                                        no extra pass, no extra load instruction.
//
// 200 MFMAs per wave instead of 328, on 128 workgroups.  Each k loop also keeps FOUR accumulators
// (the j-th MFMA of every 16-wide k-group feeds accumulator j; they are added pairwise at the
// end): the 64- / 36-deep fp32 chains of the one-workgroup kernel were the largest single
// contribution to the forward's distance from float64, chains of 16 / 9 with a
// pairwise tail are closer to it than MKL's blocked sums are.
//
// Status (round 5): built, parity-tested, measured — and NOT the default
// (PEARL_AMD_ROWPASS_PAIR=1 selects it).  In-kernel the launch ends after 13.2 us instead of 17.8
// (tools/prof_chain.py), rocprof 16.4 us against 20.2; the weight-gradient launch that adds the two
// dZ1 partials costs 14.9 us against 12.7, a steady-state round 31.3 us against 32.9 — but a
// window's first round (whose backward launch holds 128 CUs while the leading target tiles want all
// 256) grows from 66 to 76 us, and over a 2000-round call the two loops are within a box's noise of
// each other: 27.83 M against 27.65-27.77 M transitions/s on the same box.
//
// Shape: the benchmark's (K1 <= 144 in 9 k-groups, H1 = H2 = 256); everything else keeps
// online_rowpass_kernel.  PH as there: 0 = whole pass, 1 / 2 = forward / backward launches of a
// window-first round (the halves' q partials then travel through plain memory, `qhalf`).
#pragma once
#include "online_kernels.hpp"

namespace pa {

struct PairArgs {
  RowArgs r;
  float* dZ1p;        // [B][H1 / 2]{a(2p), a(2p + 1), b(2p), b(2p + 1)}: the halves' partials of dZ1,
                      // a = half 0's, b = half 1's (DwProblem::dz_pair)
  unsigned* qx;       // [2 halves][qx_rows] tagged exchange words, all kYPendingBits between launches
  int qx_rows;
  float* qhalf;       // [2][B] PH 1 -> PH 2
  int ntiles;
};

constexpr int PR_H = 256;              // H1 = H2
constexpr int PR_P1 = PR_H + 4;        // LDS pitch of an h1 row (floats)
constexpr int PR_P2 = PR_H / 2 + 4;    // ... of an s2 half row
constexpr int PR_NG1 = 9, PR_NG2 = 16, PR_NG3 = 8;
inline size_t pair_smem_bytes() {
  return sizeof(float) * ((size_t)RP_ROWS * (PR_P1 + PR_P2) + 8 * 16 + 2 * 16);
}

__device__ __forceinline__ f32x4v sum4(const f32x4v (&a)[4]) { return (a[0] + a[1]) + (a[2] + a[3]); }

template <int PH>
static __global__ __launch_bounds__(512) void online_rowpass_pair_kernel(PairArgs p) {
  const RowArgs& a = p.r;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* h1s = smem;                        // [16][PR_P1]
  float* d2s = h1s + RP_ROWS * PR_P1;       // [16][PR_P2]  s2 = [h2 > 0] w3 of this half
  float* qpart = d2s + RP_ROWS * PR_P2;     // [8][16]
  float* hand = qpart + 8 * 16;             // [2][16]: the peer's q partial and the Bellman target of each row

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  // workgroups b and b ^ 8 share a tile: under the round-robin placement both sit on XCD b % 8
  const int bid = blockIdx.x;
  const int half = (bid >> 3) & 1;
  const int tile = ((bid >> 4) << 3) + (bid & 7);
  if (tile >= p.ntiles) return;
  if (a.signal_flag && bid == 0 && tid == 0)
    __hip_atomic_store(a.signal_flag, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  PA_STAMP(a.prof, bid, wave, 0);
  PA_STAMP_CYC(a.prof, bid, wave, 14);
  if (a.wait_flag) rowpass_wait_x(a.wait_flag, a.wait_value, a.err, a.err_host);
  const int m0 = tile * RP_ROWS;
  const int row = m0 + r16;
  const bool rok = row < a.B;
  const int u0 = wave * 32 + 4 * qd;                 // layer 1 / G: units u0 + 16 t + reg
  const int tile0 = wave * 2;
  const int T2 = half * 8 + wave;                    // layer 2: this wave's ONE tile of 16 units
  const int uh = T2 * 16 + 4 * qd;                   // ... its units uh + reg
  const float b3v = a.b3[0];
  auto vec4 = [&](const float* q, int col, int n) { return ld4_or_zero(q, col, col < n); };

  float4 h1k[2], h2k, w3v;
  float4 r3a[PR_NG3], r3b[PR_NG3];                   // G's weights: W2^T tiles tile0, tile0 + 1, k-groups of the half
  auto fill_r3 = [&]() {
    const int64_t base0 = ((int64_t)tile0 * 16 + half * 8) * 256 + lane * 4;
    const int64_t base1 = base0 + (int64_t)16 * 256;
#pragma unroll
    for (int g = 0; g < PR_NG3; ++g) {
      r3a[g] = ld4_or_zero(a.W2tf, base0 + (int64_t)g * 256, true);
      r3b[g] = ld4_or_zero(a.W2tf, base1 + (int64_t)g * 256, true);
    }
  };
  float qh = 0.f;   // this half's partial of Q(s, a), every lane of the row
  if constexpr (PH == 2) {
    const int64_t r1 = (int64_t)row * PR_H;
    fill_r3();
#pragma unroll
    for (int t = 0; t < 2; ++t) h1k[t] = ld4_or_zero(a.H1a, r1 + u0 + 16 * t, rok);
    h2k = ld4_or_zero(a.H2a, r1 + uh, rok);
    w3v = vec4(a.w3, uh, PR_H);
    float4 z;
    z.x = (rok && h2k.x > 0.f) ? w3v.x : 0.f;
    z.y = (rok && h2k.y > 0.f) ? w3v.y : 0.f;
    z.z = (rok && h2k.z > 0.f) ? w3v.z : 0.f;
    z.w = (rok && h2k.w > 0.f) ? w3v.w : 0.f;
    *reinterpret_cast<float4*>(d2s + r16 * PR_P2 + 16 * wave + 4 * qd) = z;
  } else {
    // ---- one burst: everything the tile needs before its first barrier (see online_rowpass_kernel)
    const bool vx = is_vec_ok(a.x, a.ldx) && ((a.K1 & 3) == 0);
    float4 xf[PR_NG1], wa[PR_NG1], wb[PR_NG1], b1v[2], w2r[PR_NG2], b2v;
#pragma unroll
    for (int t = 0; t < 2; ++t) b1v[t] = vec4(a.b1, u0 + 16 * t, PR_H);
    {
      const int64_t base0 = ((int64_t)tile0 * PR_NG1) * 256 + lane * 4;
      const int64_t base1 = base0 + (int64_t)PR_NG1 * 256;
#pragma unroll
      for (int g = 0; g < PR_NG1; ++g) {
        const int c = 16 * g + 4 * qd;
        if (vx) xf[g] = ld4_or_zero(a.x, (int64_t)row * a.ldx + c, rok && c < a.K1);
        else xf[g] = guarded_load4(a.x, (int64_t)row * a.ldx, rok, c, a.K1);
        wa[g] = ld4_or_zero(a.W1f, base0 + (int64_t)g * 256, true);
        wb[g] = ld4_or_zero(a.W1f, base1 + (int64_t)g * 256, true);
      }
    }
    {
      const int64_t base = ((int64_t)T2 * PR_NG2) * 256 + lane * 4;
#pragma unroll
      for (int g = 0; g < PR_NG2; ++g) w2r[g] = ld4_or_zero(a.W2f, base + (int64_t)g * 256, true);
    }
    b2v = vec4(a.b2, uh, PR_H);
    w3v = vec4(a.w3, uh, PR_H);
    PA_STAMP(a.prof, bid, wave, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer 1: h1 = relu(W1 x + b1), all 256 units, four accumulators per tile
    f32x4v c1[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      c1[t][0][0] = b1v[t].x; c1[t][0][1] = b1v[t].y; c1[t][0][2] = b1v[t].z; c1[t][0][3] = b1v[t].w;
#pragma unroll
      for (int j = 1; j < 4; ++j) c1[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int g = 0; g < PR_NG1; ++g) {
      const float4 x4 = xf[g], w0 = wa[g], w1 = wb[g];
      c1[0][0] = mfma16(w0.x, x4.x, c1[0][0]);
      c1[1][0] = mfma16(w1.x, x4.x, c1[1][0]);
      c1[0][1] = mfma16(w0.y, x4.y, c1[0][1]);
      c1[1][1] = mfma16(w1.y, x4.y, c1[1][1]);
      c1[0][2] = mfma16(w0.z, x4.z, c1[0][2]);
      c1[1][2] = mfma16(w1.z, x4.z, c1[1][2]);
      c1[0][3] = mfma16(w0.w, x4.w, c1[0][3]);
      c1[1][3] = mfma16(w1.w, x4.w, c1[1][3]);
    }
    PA_STAMP(a.prof, bid, wave, 2);
    if (PH == 0 && a.y) fill_r3();   // layer 1 has released its operand registers
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u = u0 + 16 * t;
      const f32x4v z = sum4(c1[t]);
      h1k[t] = make_float4(relu_keep_nan(z[0]), relu_keep_nan(z[1]), relu_keep_nan(z[2]),
                           relu_keep_nan(z[3]));
      *reinterpret_cast<float4*>(h1s + r16 * PR_P1 + u) = h1k[t];
      // each half stores the h1 columns of "its" four waves: one copy of the tile in HBM
      if (rok && a.H1a && (wave >> 2) == half)
        *reinterpret_cast<float4*>(a.H1a + (int64_t)row * PR_H + u) = h1k[t];
    }
    PA_STAMP(a.prof, bid, wave, 3);
    __syncthreads();                                                      // barrier A: h1 tile
    PA_STAMP(a.prof, bid, wave, 4);
    // ---- layer 2, this half's 128 units: one tile per wave, K = 256
    f32x4v c2[4];
    c2[0][0] = b2v.x; c2[0][1] = b2v.y; c2[0][2] = b2v.z; c2[0][3] = b2v.w;
#pragma unroll
    for (int j = 1; j < 4; ++j) c2[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    {
      const float* act = h1s + r16 * PR_P1 + 4 * qd;
#pragma unroll
      for (int g = 0; g < PR_NG2; ++g) {
        const float4 x4 = *reinterpret_cast<const float4*>(act + g * 16);
        const float4 w = w2r[g];
        c2[0] = mfma16(w.x, x4.x, c2[0]);
        c2[1] = mfma16(w.y, x4.y, c2[1]);
        c2[2] = mfma16(w.z, x4.z, c2[2]);
        c2[3] = mfma16(w.w, x4.w, c2[3]);
      }
    }
    PA_STAMP(a.prof, bid, wave, 5);
    {
      const f32x4v z = sum4(c2);
      h2k = make_float4(relu_keep_nan(z[0]), relu_keep_nan(z[1]), relu_keep_nan(z[2]),
                        relu_keep_nan(z[3]));
    }
    float part = 0.f;
    part = fmaf(h2k.x, w3v.x, part);
    part = fmaf(h2k.y, w3v.y, part);
    part = fmaf(h2k.z, w3v.z, part);
    part = fmaf(h2k.w, w3v.w, part);
    if (rok && a.H2a) *reinterpret_cast<float4*>(a.H2a + (int64_t)row * PR_H + uh) = h2k;
    if (PH == 0 && a.y) {
      float4 z;
      z.x = (rok && h2k.x > 0.f) ? w3v.x : 0.f;
      z.y = (rok && h2k.y > 0.f) ? w3v.y : 0.f;
      z.z = (rok && h2k.z > 0.f) ? w3v.z : 0.f;
      z.w = (rok && h2k.w > 0.f) ? w3v.w : 0.f;
      *reinterpret_cast<float4*>(d2s + r16 * PR_P2 + 16 * wave + 4 * qd) = z;
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (qd == 0) qpart[wave * 16 + r16] = part;
  }
  PA_STAMP(a.prof, bid, wave, 6);
  __syncthreads();                                                        // barrier B: s2, qpart
  PA_STAMP(a.prof, bid, wave, 7);
  if constexpr (PH != 2) {
#pragma unroll
    for (int w = 0; w < 8; ++w) qh += qpart[w * 16 + r16];
  }
  if constexpr (PH == 1) {
    if (wave == 0 && qd == 0 && rok) p.qhalf[(int64_t)half * a.B + row] = qh;
    PA_STAMP(a.prof, bid, wave, 10);
    return;
  }
  // ---- PH 0: hand this half's partial to the other workgroup of the pair: one tagged word per
  // row.  ONE quarter-wave per workgroup talks to memory — it publishes qx[half][row], and after
  // the backward GEMM collects the peer's word (restoring its tag) and the row's Bellman target —
  // and passes both on through LDS: sixteen polling lanes per workgroup, not 512.  (Every wave
  // polling for itself costs the TARGET kernel its memory system whenever the chain runs ahead of
  // it: agent-scope loads bypass the L2; measured, the persistent target launch of a window took
  // 290 us instead of 170 and the loop was slower than with the one-workgroup row pass.)
  unsigned* qx_mine = p.qx + ((int64_t)half * p.qx_rows + row);
  unsigned* qx_peer = p.qx + ((int64_t)(1 - half) * p.qx_rows + row);
  const bool talker = wave == 0 && qd == 0;
  if constexpr (PH == 0) {
    if (talker) publish_y(reinterpret_cast<float*>(qx_mine), qh);
  }
  unsigned ybits = kYPendingBits;
  if (talker && a.y && rok) {
    // first look at the Bellman target, in flight while the backward GEMM runs
    ybits = a.y_tagged ? __hip_atomic_load(reinterpret_cast<const unsigned*>(a.y) + row,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : __builtin_bit_cast(unsigned, a.y[row]);
  }
  f32x4v cg[2][4];
  if (a.y) {
    // ---- G_h = s2[half] W2[half rows, :]  (scaled by dq below): all 256 units, K = 128
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) cg[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const float* act = d2s + r16 * PR_P2 + 4 * qd;
#pragma unroll
    for (int g = 0; g < PR_NG3; ++g) {
      const float4 x4 = *reinterpret_cast<const float4*>(act + g * 16);
      const float4 w0 = r3a[g], w1 = r3b[g];
      cg[0][0] = mfma16(w0.x, x4.x, cg[0][0]);
      cg[1][0] = mfma16(w1.x, x4.x, cg[1][0]);
      cg[0][1] = mfma16(w0.y, x4.y, cg[0][1]);
      cg[1][1] = mfma16(w1.y, x4.y, cg[1][1]);
      cg[0][2] = mfma16(w0.z, x4.z, cg[0][2]);
      cg[1][2] = mfma16(w1.z, x4.z, cg[1][2]);
      cg[0][3] = mfma16(w0.w, x4.w, cg[0][3]);
      cg[1][3] = mfma16(w1.w, x4.w, cg[1][3]);
    }
  }
  PA_STAMP(a.prof, bid, wave, 8);
  if (talker) {
    float qp = 0.f, yv = 0.f;
    if constexpr (PH == 0) {
      qp = consume_y(reinterpret_cast<const float*>(qx_peer), a.err, a.err_host);
      __hip_atomic_store(qx_peer, kYPendingBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      PA_STAMP(a.prof, bid, wave, 11);
    }
    if (a.y && rok) {
      // (an untagged y can legitimately hold the tag's bit pattern: only the tagged protocol polls)
      if (a.y_tagged && ybits == kYPendingBits) yv = consume_y(a.y + row, a.err, a.err_host);
      else yv = __builtin_bit_cast(float, ybits);
    }
    hand[r16] = qp;
    hand[16 + r16] = yv;
  }
  __syncthreads();                                                        // barrier C: peer q, y
  float q;
  if constexpr (PH == 2) {
    q = rok ? (p.qhalf[row] + p.qhalf[(int64_t)a.B + row]) : 0.f;
  } else {
    const float qp = hand[r16];
    q = half == 0 ? (qh + qp) : (qp + qh);     // = q_half0 + q_half1 on both sides
  }
  q += b3v;
  if (half == 0 && wave == 0 && qd == 0 && rok && a.q_out) a.q_out[row] = q;
  if (!a.y) return;
  const float yv = rok ? hand[16 + r16] : q;
  PA_STAMP(a.prof, bid, wave, 9);
  const float d = __fsub_rn(q, yv);
  const float dq = __fmul_rn(a.norm, d);
  if (half == 0 && wave == 0 && qd == 0 && rok) {
    a.dq_out[row] = dq;
    a.absd_out[row] = fabsf(d);
  }
  if (rok) {
    float4 z;
    z.x = (h2k.x > 0.f) ? __fmul_rn(dq, w3v.x) : 0.f;
    z.y = (h2k.y > 0.f) ? __fmul_rn(dq, w3v.y) : 0.f;
    z.z = (h2k.z > 0.f) ? __fmul_rn(dq, w3v.z) : 0.f;
    z.w = (h2k.w > 0.f) ? __fmul_rn(dq, w3v.w) : 0.f;
    *reinterpret_cast<float4*>(a.dZ2 + (int64_t)row * PR_H + uh) = z;
    float* dz1 = p.dZ1p + (int64_t)row * (2 * PR_H) + 2 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4v G = sum4(cg[t]);
      float2 g0, g1;
      g0.x = (h1k[t].x > 0.f) ? __fmul_rn(dq, G[0]) : 0.f;
      g0.y = (h1k[t].y > 0.f) ? __fmul_rn(dq, G[1]) : 0.f;
      g1.x = (h1k[t].z > 0.f) ? __fmul_rn(dq, G[2]) : 0.f;
      g1.y = (h1k[t].w > 0.f) ? __fmul_rn(dq, G[3]) : 0.f;
      const int u = u0 + 16 * t;                       // units u, u + 1 | u + 2, u + 3: pairs u / 2, u / 2 + 1
      *reinterpret_cast<float2*>(dz1 + 2 * u) = g0;
      *reinterpret_cast<float2*>(dz1 + 2 * u + 4) = g1;
    }
  }
  PA_STAMP(a.prof, bid, wave, 10);
  PA_STAMP_CYC(a.prof, bid, wave, 15);
}

// The weight-gradient kernels that add DwProblem::dZb while they load dZ (one workgroup per CU:
// the second raw buffer of the split loop does not fit 128 registers)
static __global__ __launch_bounds__(512) void weight_grad_kernel_pair(DwArgs a) {
  __shared__ float part[4 * DW_TM * DW_TN];
  __shared__ float csum[8 * DW_TM];
  weight_grad_body<4, false, true>(a, part, csum);
}
static __global__ __launch_bounds__(512) void weight_grad_kernel32_pair(DwArgs a) {
  __shared__ float part[4 * 32 * DW_TN];
  __shared__ float csum[8 * 32];
  weight_grad_body<2, false, true>(a, part, csum);
}
static __global__ __launch_bounds__(512) void weight_grad_split_kernel_pair(DwArgs a) {
  __shared__ float part[4 * DW_TM * DW_TN];
  __shared__ float csum[8 * DW_TM];
  weight_grad_body<4, true, true>(a, part, csum);
}
static __global__ __launch_bounds__(512) void weight_grad_split_kernel32_pair(DwArgs a) {
  __shared__ float part[4 * 32 * DW_TN];
  __shared__ float csum[8 * 32];
  weight_grad_body<2, true, true>(a, part, csum);
}

}  // namespace pa
