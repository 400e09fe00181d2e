// Ping-pong version of the fused target-network kernel (see target_tile in dqn_kernels.hpp for the
// arithmetic, which is identical: same MFMA order, same reductions, bit-identical results).
//
// Why: v_mfma_f32_32x32x2_f32 streams from TWO waves per SIMD sustain 154 TFLOP/s on random
// operands, from FOUR only 125 (tools/mfma_bench.hip) — and a tile spends a quarter of its life
// outside its MFMA main loop (operand loads, layer 1, the layer-3 reduction, two barriers).  Two
// independent 8-wave workgroups per CU therefore top out at 74 % of peak: either both are in their
// main loops (four streams per SIMD, the slow regime) or one of them leaves the pipe idle.
//
// Here ONE 16-wave workgroup owns the CU and its two 8-wave teams alternate roles in lock-step:
//
//   phase p:   team A  main loop of tile a      (2 MFMA streams per SIMD, the fast regime)
//              team B  epilogue of its previous tile, then prologue of its next tile
//   phase p+1: roles swapped
//
// so the matrix pipe always sees exactly one team's main loop.  What hides under it is LATENCY, not
// issue: a SIMD that streams fp32 MFMAs from two waves issues nothing else — partner waves get no
// slot at all, whatever their s_setprio (tools/corun_bench.hip: VALU, LDS and global-load partners
// all take exactly their stand-alone time PLUS the MFMA stream's).  So the teams meet at five
// workgroup barriers per phase, B1..B4 inside the main loop and B0 at its end; a barrier is where
// the main-loop team yields the SIMDs, and the other team's work is cut into segments that each
// run at one of them:  layer-3 partials | row sums | max + y, tile index, operand loads ISSUED |
// (loads land under the next k-groups) layer 1, ReLU, h1 -> LDS | weight ring primed.  A phase
// costs the main loop plus the issue time of those segments instead of plus their latency.
//
// Tiles come from a work-stealing counter (a.tile_ctr, one fetch per team and phase, issued a
// phase ahead of its use); workgroups that land on a CU reserved for the online chain exit at
// once (a.reserved).  One workgroup per CU fits (138 KB of LDS).
#pragma once
#include "dqn_kernels.hpp"

namespace pa {

inline size_t target_pp_smem_bytes(int H1) {
  const int H1P = t_nkg(H1) * 8;
  return sizeof(float) * ((size_t)2 * T_ROWS * (H1P + 4) + 2 * 8 * 64 + 2 * 64 + 16 + 2 * T_MAXH);
}

// writers: LDS stores must have landed before the other waves pass the barrier
#define PA_BAR_W() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// main-loop team: nothing of its own to publish, global prefetches stay in flight
#define PA_BAR_N() asm volatile("s_barrier" ::: "memory")

template <int NKG>
static __global__ __launch_bounds__(1024, 4) void target_pp_kernel(TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int H1P = NKG * 8;
  constexpr int PA_ = H1P + 4;
  constexpr int RD = 4;   // k-groups of layer-2 weights in flight per wave (>= 4000 clocks of MFMA ahead)
  // k-groups after which the main-loop team meets B1 .. B4
  constexpr int BG1 = NKG / 16 + (NKG < 16), BG2 = NKG / 8 + (NKG < 16), BG3 = NKG / 4 + (NKG < 16),
                BG4 = (5 * NKG) / 8;
  static_assert(BG1 < BG2 && BG2 < BG3 && BG3 < BG4 && BG4 < NKG, "barrier schedule");
  if (a.reserved && a.reserved[cu_key()]) return;

  const int tid = threadIdx.x, lane = tid & 63, ttid = tid & 511;
  // scalar: role and tile-ownership branches are uniform per wave
  const int team = __builtin_amdgcn_readfirstlane(tid >> 9);
  const int wave = __builtin_amdgcn_readfirstlane(ttid >> 6);
  float* Ah = smem + team * (T_ROWS * PA_);                 // [64][H1P+4] this team's h1 tile
  float* qpart = smem + 2 * T_ROWS * PA_ + team * (8 * 64);  // [8][64]
  float* qv = smem + 2 * T_ROWS * PA_ + 2 * 8 * 64 + team * 64;
  int* ctl = reinterpret_cast<int*>(smem + 2 * T_ROWS * PA_ + 2 * 8 * 64 + 2 * 64);
  // ctl[team]: next tile of the team; ctl[2 + team]: the team still has work in a later phase
  // Layer-3 constants live in LDS for the whole launch: the epilogue segments sit between
  // barriers the MAIN-LOOP team also waits at, so a global round trip there (~2 us when the chain
  // kernels share the memory system) stalls the matrix pipe; an LDS read does not.
  float* b2s = smem + 2 * T_ROWS * PA_ + 2 * 8 * 64 + 2 * 64 + 16;   // [T_MAXH]
  float* w3s = b2s + T_MAXH;                                          // [T_MAXH]
  for (int i = tid; i < T_MAXH; i += 1024) {
    b2s[i] = ld_or_zero(a.b2, i, i < a.H2);
    w3s[i] = ld_or_zero(a.w3, i, i < a.H2);
  }
  const float b3v = a.b3[0];

  const int h = lane >> 5, l31 = lane & 31;
  const int nt1 = H1P >> 5, nt2 = (a.H2 + 31) >> 5;
  const bool l1 = wave < nt1, l2 = wave < nt2;
  const int nq0 = wave * 32 + 4 * h;
  const bool vU = ((reinterpret_cast<uintptr_t>(a.U) & 15) == 0) && ((a.ldu & 3) == 0);
  const bool vfeat = ((reinterpret_cast<uintptr_t>(a.feat) & 15) == 0) && ((a.AD & 3) == 0) &&
                     ((a.feat_bstride & 3) == 0);
  const bool vw1 = ((reinterpret_cast<uintptr_t>(a.W1a) & 15) == 0) && ((a.ldw1 & 3) == 0) &&
                   ((a.AD & 3) == 0);
  const int wcol = wave * 32 + l31;
  const int64_t woff = (int64_t)wcol * a.ldw1;
  const bool wok = l1 && wcol < a.H1;
  // layer-2 weight fragments of this wave: descriptor based at the wave's slice (scalar), lane
  // offset in ONE VGPR, the k-group as the scalar offset — no per-k-group address registers
  // (the main loop has none to spare: 128 VGPRs per wave at four waves per SIMD).  Waves that own
  // no hidden units get an empty descriptor: every load returns zeros.
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.W2f) + (int64_t)wave * NKG * 256, (short)0, l2 ? NKG * 1024 : 0,
      0x00020000);
  const int vlane16 = lane * 16;
  auto ring_load = [&](int g) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w2, vlane16, g * 1024, 0);
    const f32x4_t f = __builtin_bit_cast(f32x4_t, v);
    return make_float4(f[0], f[1], f[2], f[3]);
  };

  // A team's life is  prologue -> main loop -> epilogue -> prologue -> ...; one loop iteration
  // below is an (epilogue + prologue) phase followed by a main-loop phase, so the weight ring
  // lives inside an iteration and only acc / b2v / w3v (main loop -> epilogue) cross the back
  // edge.  Team 1 runs one phase behind team 0: it sits out phase 0.
  f32x16 acc[2];
  int cur = -1;      // tile whose layer-2 result sits in acc (epilogue pending)
  bool exhausted = false;
  // Per-tile scalars of the epilogue (action mask of row ttid, terminal flag and reward of
  // transition ttid), fetched in the tile's PROLOGUE and carried through its main loop, and the
  // team's next tile index, requested one whole epilogue phase before it is needed (a returning
  // device-scope atomic costs 1-3 us under load).
  unsigned pf_mask = 0, pf_term = 0;
  float pf_reward = 0.f;
  int pend = a.ntiles;
  if (ttid == 0) pend = atomicAdd(a.tile_ctr, 1);
  __syncthreads();   // b2s / w3s written (both teams, once)
  if (team == 1) {
    PA_BAR_N();
    PA_BAR_N();
    PA_BAR_N();
    PA_BAR_N();
    // the team already holds its first tile index: it must not leave while that index is a tile
    if (ttid == 0) ctl[3] = (pend < a.ntiles) ? 1 : 0;
    PA_BAR_W();
    if ((ctl[2] | ctl[3]) == 0) return;
  }

  int ph = team;   // phase counter (profiling only)
#define PP_STAMP(k)                                                                          \
  do {                                                                                       \
    if (a.prof && lane == 0 && ph < 8)                                                       \
      a.prof[((int64_t)blockIdx.x * 16 + (tid >> 6)) * 32 + ph * 4 + (k)] = (long long)wall_clock64(); \
  } while (0)
  for (;;) {
    float4 ring[RD];
    int loaded = -1;   // tile whose h1 sits in Ah (main loop pending)
    {
      PP_STAMP(0);
      // ================= epilogue of `cur`, prologue of the next tile =================
      const int nxt = pend;                 // requested during the previous epilogue phase
      pend = a.ntiles;
      if (ttid == 0 && !exhausted && nxt < a.ntiles) pend = atomicAdd(a.tile_ctr, 1);
      if (cur >= 0) {
        // layer 3: in-lane over this lane's 16 hidden units, then the other half, then the waves.
        // Its constants are fetched here, not carried from the main loop: this team has slack,
        // the main loop has no registers to spare.
        float4 b2v[4], w3v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // zeros beyond H2 (waves that own no hidden units)
          b2v[q] = *reinterpret_cast<const float4*>(b2s + nq0 + 8 * q);
          w3v[q] = *reinterpret_cast<const float4*>(w3s + nq0 + 8 * q);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          float sum = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sum = fmaf(relu_keep_nan(acc[tm][4 * q + 0] + b2v[q].x), w3v[q].x, sum);
            sum = fmaf(relu_keep_nan(acc[tm][4 * q + 1] + b2v[q].y), w3v[q].y, sum);
            sum = fmaf(relu_keep_nan(acc[tm][4 * q + 2] + b2v[q].z), w3v[q].z, sum);
            sum = fmaf(relu_keep_nan(acc[tm][4 * q + 3] + b2v[q].w), w3v[q].w, sum);
          }
          sum += __shfl_xor(sum, 32);
          if (h == 0) qpart[wave * 64 + tm * 32 + l31] = sum;
        }
      }
      if (ttid == 0) ctl[team] = nxt;
      PA_BAR_W();  // B1
      PP_STAMP(1);
      const int cb0 = (cur >= 0 ? cur : 0) * a.bpw;
      const int cnb = min(a.bpw, a.B - cb0);
      const int cnrows = cnb * a.A;
      if (cur >= 0 && ttid < T_ROWS) {
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) q += qpart[w * 64 + ttid];
        q += b3v;
        if (ttid < cnrows && pf_mask) q = -INFINITY;
        qv[ttid] = q;
      }
      PA_BAR_W();  // B2
      if (cur >= 0 && ttid < cnb) {
        const int bb = cb0 + ttid;
        float m = qv[ttid * a.A];
        for (int i = 1; i < a.A; ++i) {
          const float x = qv[ttid * a.A + i];
          m = (x > m || x != x) ? x : m;
        }
        if (a.next_v) a.next_v[bb] = m;
        if (a.y) {
          const float live = 1.0f - (pf_term ? 1.0f : 0.0f);
          const float t0 = __fmul_rn(m, a.gamma);
          const float t1 = __fmul_rn(t0, live);
          publish_y(a.y + bb, __fadd_rn(t1, pf_reward));
        }
      }
      cur = -1;
      PP_STAMP(2);
      const int tile = ctl[team];
      if (tile < a.ntiles) {
        // ---- prologue: U + action part of layer 1, ReLU, h1 tile -> LDS; layer-2 ring primed
        const int b0 = tile * a.bpw;
        const int nb = min(a.bpw, a.B - b0);
        const int nrows = nb * a.A;
        int64_t foff[2];
        bool fok[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const int row = tm * 32 + l31;
          const bool rok = l1 && row < nrows;
          const int rr = rok ? row : 0;
          const int bb = b0 + rr / a.A;
          fok[tm] = rok;
          foff[tm] = (int64_t)bb * a.feat_bstride + (int64_t)(rr % a.A) * a.AD;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = nq0 + 8 * q;
            float4 u;
            if (vU) u = ld4_or_zero(a.U, (int64_t)bb * a.ldu + n, rok && n < a.H1);
            else u = guarded_load4(a.U, (int64_t)bb * a.ldu, rok, n, a.H1);
            acc[tm][4 * q + 0] = u.x; acc[tm][4 * q + 1] = u.y;
            acc[tm][4 * q + 2] = u.z; acc[tm][4 * q + 3] = u.w;
          }
        }
        auto l1_loads = [&](int k0, float4 (&x4)[2], float4& w4) {
          const int k = k0 + 4 * h;
          if (vfeat) {
            x4[0] = ld4_or_zero(a.feat, foff[0] + k, fok[0] && k < a.AD);
            x4[1] = ld4_or_zero(a.feat, foff[1] + k, fok[1] && k < a.AD);
          } else {
            x4[0] = guarded_load4(a.feat, foff[0], fok[0], k, a.AD);
            x4[1] = guarded_load4(a.feat, foff[1], fok[1], k, a.AD);
          }
          if (vw1) w4 = ld4_or_zero(a.W1a, woff + k, wok && k < a.AD);
          else w4 = guarded_load4(a.W1a, woff, wok, k, a.AD);
        };
        auto l1_mfma = [&](const float4 (&x4)[2], const float4& w4) {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            acc[tm] = mfma32(w4.x, x4[tm].x, acc[tm]);
            acc[tm] = mfma32(w4.y, x4[tm].y, acc[tm]);
            acc[tm] = mfma32(w4.z, x4[tm].z, acc[tm]);
            acc[tm] = mfma32(w4.w, x4[tm].w, acc[tm]);
          }
        };
        float4 fx[2][2], fw[2];
        l1_loads(0, fx[0], fw[0]);
        l1_loads(8, fx[1], fw[1]);
        // this tile's epilogue scalars (consumed two phases from now)
        pf_mask = 0; pf_term = 0; pf_reward = 0.f;
        if (ttid < nrows && a.mask)
          pf_mask = a.mask[(int64_t)(b0 + ttid / a.A) * a.mask_bstride + ttid % a.A];
        if (ttid < nb && a.y) {
          pf_term = a.term[b0 + ttid];
          pf_reward = a.reward[b0 + ttid];
        }
        PA_BAR_W();  // B3: operand loads are in flight; the main loop runs on
        l1_mfma(fx[0], fw[0]);
        l1_mfma(fx[1], fw[1]);
        for (int k0 = 16; k0 < a.AD; k0 += 8) {
          float4 x4[2], w4;
          l1_loads(k0, x4, w4);
          l1_mfma(x4, w4);
        }
        if (l1) {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            float* dst = Ah + (tm * 32 + l31) * PA_ + nq0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 v;
              v.x = relu_keep_nan(acc[tm][4 * q + 0]); v.y = relu_keep_nan(acc[tm][4 * q + 1]);
              v.z = relu_keep_nan(acc[tm][4 * q + 2]); v.w = relu_keep_nan(acc[tm][4 * q + 3]);
              *reinterpret_cast<float4*>(dst + 8 * q) = v;
            }
          }
        }
        PA_BAR_W();  // B4: h1 tile written
        // layer-2 weight ring, primed only now: the layer-1 operands are dead (register budget:
        // 128 per wave), and the loads land while the other team finishes its main loop
#pragma unroll
        for (int g = 0; g < RD; ++g) ring[g] = ring_load(g);
        loaded = tile;
      } else {
        PA_BAR_W();  // B3
        PA_BAR_W();  // B4
        exhausted = true;
        loaded = -1;
      }
      if (ttid == 0) ctl[2 + team] = (loaded >= 0) ? 1 : 0;
      PP_STAMP(3);
      PA_BAR_W();  // B0
    }
    ++ph;
    if ((ctl[2] | ctl[3]) == 0) break;
    {
      PP_STAMP(0);
      // ================= main loop of `loaded` =================
      if (loaded >= 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
        if (l2) {
          const float* xp0 = Ah + l31 * PA_ + 4 * h;
          const float* xp1 = xp0 + 32 * PA_;
#pragma unroll
          for (int g = 0; g < NKG; ++g) {
            const float4 w4 = ring[g % RD];
            if (g + RD < NKG) ring[g % RD] = ring_load(g + RD);
            const float4 x0 = *reinterpret_cast<const float4*>(xp0 + g * 8);
            const float4 x1 = *reinterpret_cast<const float4*>(xp1 + g * 8);
            acc[0] = mfma32(w4.x, x0.x, acc[0]);
            acc[1] = mfma32(w4.x, x1.x, acc[1]);
            acc[0] = mfma32(w4.y, x0.y, acc[0]);
            acc[1] = mfma32(w4.y, x1.y, acc[1]);
            acc[0] = mfma32(w4.z, x0.z, acc[0]);
            acc[1] = mfma32(w4.z, x1.z, acc[1]);
            acc[0] = mfma32(w4.w, x0.w, acc[0]);
            acc[1] = mfma32(w4.w, x1.w, acc[1]);
            if (g == BG1 - 1 || g == BG2 - 1 || g == BG3 - 1 || g == BG4 - 1) PA_BAR_N();  // B1..B4
          }
        } else {
          PA_BAR_N();
          PA_BAR_N();
          PA_BAR_N();
          PA_BAR_N();
        }
        cur = loaded;
      } else {
        PA_BAR_N();
        PA_BAR_N();
        PA_BAR_N();
        PA_BAR_N();
      }
      if (ttid == 0) ctl[2 + team] = (cur >= 0) ? 1 : 0;
      PP_STAMP(3);
      PA_BAR_W();  // B0
    }
    ++ph;
    if ((ctl[2] | ctl[3]) == 0) break;
  }
#undef PP_STAMP
}

}  // namespace pa
