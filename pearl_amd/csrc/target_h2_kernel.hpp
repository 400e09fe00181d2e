// The fused target-network tile (target_tile_split in target_split_kernel.hpp: same inputs, same
// prologue, same epilogue, same C layout) with its layer-2 product on the FP16 matrix pipe at fp32
// accuracy (round 6): every operand scaled by an exact power of two into fp16's range — an h1 row by
// its own maximum, a row of the target W2 by its maximum — and split into TWO fp16 terms
// (h2_common.hpp), three products  hi hi | hi lo + lo hi  per k-step on v_mfma_f32_32x32x16_f16, the
// hi hi class alternating between two accumulators, the result unscaled with one ldexp.
//
// Against the bf16x3 tile: half the matrix instructions (96 instead of 192 per wave and tile), two
// LDS planes instead of three (68 KB), 2/3 of the weight bytes per launch and XCD, and a split of two
// v_fma_mix instructions per value instead of ~4.  Error relative to sum |a b| 1.0e-7 (bf16x3: 6.5e-8,
// the fp32 MFMA chain 1.8e-7; tools/split_mfma_bench.hip).
//
// The weight planes carry a scale per unit, so they are not kept by the optimizer epilogue (a tile
// of the soft update does not know its rows' maxima): target_w2h_pack rebuilds them — one wave per
// row of W2': maximum, scale, split — wherever the target network may have changed: in every
// rebuild of the packed copies (repack_body) and at the head of every target-update window of
// learn(), inside the launch that waits for the previous window's soft update anyway
// (wait_pack_kernel: resident before the word flips, 1 us of work after it).
#pragma once
#include "dqn_kernels.hpp"
#include "h2_common.hpp"

namespace pa {

// plane slot ((wave * TS_KS + kstep) * 2 + s) * 64 + lane: the 8 fp16 that lane feeds one MFMA as
// its A operand — unit n = 32 wave + (lane & 31), k = 16 kstep + 8 (lane >> 5) + e
__host__ __device__ inline int64_t w2h_bytes() { return (int64_t)8 * TS_KS * 2 * 64 * 16; }

struct W2hPack {
  const float* W2;        // row-major target W2' [256][256]
  void* planes;           // w2h_bytes()
  int* fields;            // [256] scale field per unit (h2_field of the row's maximum)
};

// `coherent`: the rows are read with agent-scope loads (the caller was resident before the weights
// were written: nothing it reads may come from its XCD's L2)
template <bool COHERENT>
__device__ __forceinline__ void target_w2h_pack(const W2hPack& p, int64_t w0, int64_t nw, int lane) {
  for (int64_t n = w0; n < TS_H; n += nw) {
    const float* row = p.W2 + n * TS_H + 4 * lane;
    float x[4];
    if constexpr (COHERENT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = __hip_atomic_load(row + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const float4 v = *reinterpret_cast<const float4*>(row);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    unsigned m = umaxu(umaxu(abs_bits(x[0]), abs_bits(x[1])), umaxu(abs_bits(x[2]), abs_bits(x[3])));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = umaxu(m, (unsigned)__shfl_xor((int)m, o));
    const int f = h2_field(m);
    const float sc = h2_scale(f);
    unsigned hi[2], lo[2];
    h2_pair(x[0], x[1], sc, sc, hi[0], lo[0]);
    h2_pair(x[2], x[3], sc, sc, hi[1], lo[1]);
    // k = 4 lane .. + 3: k-step k >> 4, lane half (k >> 3) & 1, elements (k & 7) .. + 3
    const int k = 4 * lane;
    const int64_t slot = ((int64_t)((int)(n >> 5) * TS_KS + (k >> 4)) * 2) * 64 + (int)(n & 31) + 32 * ((k >> 3) & 1);
    unsigned char* base = static_cast<unsigned char*>(p.planes) + (k & 7) * 2;
    *reinterpret_cast<uint2*>(base + slot * 16) = make_uint2(hi[0], hi[1]);
    *reinterpret_cast<uint2*>(base + (slot + 64) * 16) = make_uint2(lo[0], lo[1]);
    if (lane == 0) p.fields[n] = f;
  }
}

static __global__ __launch_bounds__(256) void target_pack_kernel(W2hPack p) {
  target_w2h_pack<false>(p, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4, threadIdx.x & 63);
}

__device__ __forceinline__ f16x8 ld_f16x8(const void* base, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(base), (int)byte_off, 0, 0);
  return __builtin_bit_cast(f16x8, v);
}

constexpr int TH_NHH = 2;             // accumulators of the hi hi class (k-steps alternate)
inline size_t target_h2_smem_bytes() {
  return (size_t)2 * T_ROWS * TS_LDP * 2 + sizeof(float) * (8 * 64 + 8 * 64 + 64) + 16;
}

// Same contract as target_tile_split<0>: U is read; the host has checked target_fast_shape.
__device__ __forceinline__ void target_tile_h2(const TargetArgs& a, int tile, unsigned char* smem) {
  _Float16* planes = reinterpret_cast<_Float16*>(smem);                               // [2][64][TS_LDP]
  unsigned* rmaxw = reinterpret_cast<unsigned*>(smem + (size_t)2 * T_ROWS * TS_LDP * 2);   // [8][64]
  float* qpart = reinterpret_cast<float*>(rmaxw + 8 * 64);                            // [8][64]
  float* qv = qpart + 8 * 64;                                                         // [64]

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
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int row = tm * 32 + l31;
    const bool rok = row < nrows;
    const int rr = rok ? row : 0;
    const int bb = b0 + rr / a.A;
    fok[tm] = rok;
    foff[tm] = (int64_t)bb * a.feat_bstride + (int64_t)(rr % a.A) * a.AD;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nq0 + 8 * q;
      const float4 u = ld4_or_zero(a.U, (int64_t)bb * a.ldu + n, rok);
      acc[tm][4 * q + 0] = u.x; acc[tm][4 * q + 1] = u.y;
      acc[tm][4 * q + 2] = u.z; acc[tm][4 * q + 3] = u.w;
    }
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
  // ---- layer-2 weights: the first TS_RD k-steps of this wave's planes
  f16x8 ring[TS_RD][2];
  const unsigned wbase = (unsigned)((wave * TS_KS * 2 * 64 + lane) * 16);   // bytes; slot stride 1 KiB
#pragma unroll
  for (int g = 0; g < TS_RD; ++g)
#pragma unroll
    for (int s = 0; s < 2; ++s) ring[g][s] = ld_f16x8(a.W2h, wbase + (unsigned)(g * 2 + s) * 1024u);
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
  // h1 = relu(acc); the row's maximum: in the lane, the other lane half, then the eight waves
  // (rows >= nrows hold relu(0 + 0) = 0 exactly: their U and feat loads returned zeros)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    unsigned m = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[tm][r] = relu_keep_nan(acc[tm][r]);
      m = umaxu(m, abs_bits(acc[tm][r]));
    }
    m = umaxu(m, (unsigned)__shfl_xor((int)m, 32));
    if (h == 0) rmaxw[wave * 64 + tm * 32 + l31] = m;
  }
  __syncthreads();
  int fr[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    unsigned m = rmaxw[tm * 32 + l31];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = umaxu(m, rmaxw[w * 64 + tm * 32 + l31]);
    fr[tm] = h2_field(m);
    const float sc = h2_scale(fr[tm]);
    _Float16* dst = planes + (size_t)(tm * 32 + l31) * TS_LDP + nq0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hi[2], lo[2];
      h2_pair(acc[tm][4 * q + 0], acc[tm][4 * q + 1], sc, sc, hi[0], lo[0]);
      h2_pair(acc[tm][4 * q + 2], acc[tm][4 * q + 3], sc, sc, hi[1], lo[1]);
      *reinterpret_cast<uint2*>(dst + 8 * q) = make_uint2(hi[0], hi[1]);
      *reinterpret_cast<uint2*>(dst + (size_t)T_ROWS * TS_LDP + 8 * q) = make_uint2(lo[0], lo[1]);
    }
  }
  PA_STAMP(a.prof, tile, wave, 2);
  __syncthreads();
  PA_STAMP(a.prof, tile, wave, 3);

  // ---- layer 2: C[n][row] = sum_k W2'[n][k] h1[row][k], three fp16 products per k-step
  f32x16 c3[2][TH_NHH + 1];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int c = 0; c < TH_NHH + 1; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) c3[tm][c][r] = 0.f;
  float4 b2v[4], w3v[4];
  int wf[16];
  {
    const _Float16* bp0 = planes + (size_t)l31 * TS_LDP + 8 * h;
    const _Float16* bp1 = bp0 + (size_t)32 * TS_LDP;
    f16x8 bq[2][2][2];   // the h1 operands run one k-step ahead of their use
    auto ldb = [&](int g, f16x8 (&b)[2][2]) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        b[0][s] = *reinterpret_cast<const f16x8*>(bp0 + (size_t)s * T_ROWS * TS_LDP + 16 * g);
        b[1][s] = *reinterpret_cast<const f16x8*>(bp1 + (size_t)s * T_ROWS * TS_LDP + 16 * g);
      }
    };
    ldb(0, bq[0]);
#pragma unroll
    for (int g = 0; g < TS_KS; ++g) {
      f16x8 wa[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) wa[s] = ring[g % TS_RD][s];
      if (g + TS_RD < TS_KS) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
          ring[g % TS_RD][s] = ld_f16x8(a.W2h, wbase + (unsigned)((g + TS_RD) * 2 + s) * 1024u);
      } else if (g == TS_KS - 4) {
        // layer-3 constants and the unit scales of this lane's hidden units, into the registers of
        // the drained ring
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(a.w2hf), (nq0 + 8 * q) * 4, 0, 0);
          wf[4 * q + 0] = (int)v[0]; wf[4 * q + 1] = (int)v[1]; wf[4 * q + 2] = (int)v[2]; wf[4 * q + 3] = (int)v[3];
        }
      } else if (g == TS_KS - 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b2v[q] = ld4_or_zero(a.b2, nq0 + 8 * q, true);
      } else if (g == TS_KS - 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w3v[q] = ld4_or_zero(a.w3, nq0 + 8 * q, true);
      }
      if (g + 1 < TS_KS) ldb(g + 1, bq[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // loads stay here, ahead of this k-step's MFMAs
      f16x8 (&b)[2][2] = bq[g & 1];
      const int hh = g % TH_NHH;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        c3[tm][TH_NHH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0], b[tm][1], c3[tm][TH_NHH], 0, 0, 0);
        c3[tm][TH_NHH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1], b[tm][0], c3[tm][TH_NHH], 0, 0, 0);
        c3[tm][hh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0], b[tm][0], c3[tm][hh], 0, 0, 0);
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
        float hhs = c3[tm][0][r];
#pragma unroll
        for (int c = 1; c < TH_NHH; ++c) hhs += c3[tm][c][r];
        const float z = ldexpf(hhs + c3[tm][TH_NHH][r], fr[tm] + wf[r] - 282);
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
// off the CUs reserved for the online chain — the two modes of target_split_kernel.
static __global__ __launch_bounds__(512, 2) void target_h2_kernel(TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
  if (a.tile_ctr == nullptr) {
    if ((int)blockIdx.x < a.prio_tiles) __builtin_amdgcn_s_setprio(3);
    target_tile_h2(a, blockIdx.x, smem_h2);
    return;
  }
  __shared__ int next_tile_h2;
  if (a.reserved && a.reserved[cu_key()]) return;
  if (threadIdx.x == 0) {
    next_tile_h2 = atomicAdd(a.tile_ctr, 1);
    if (a.dbg_workers && next_tile_h2 < a.ntiles) atomicAdd(a.dbg_workers, 1);
  }
  __syncthreads();
  int tile = next_tile_h2;
  while (tile < a.ntiles) {
    int ahead = 0;
    if (threadIdx.x == 0) ahead = atomicAdd(a.tile_ctr, 1);   // in flight under this tile
    target_tile_h2(a, tile, smem_h2);
    if (threadIdx.x == 0) next_tile_h2 = ahead;
    __syncthreads();  // publishes next_tile; the planes are reused by the next tile
    tile = next_tile_h2;
  }
}

}  // namespace pa
