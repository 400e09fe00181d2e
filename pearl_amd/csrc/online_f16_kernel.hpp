// The online row pass with its three GEMMs on the fp16 matrix pipe at fp32 accuracy (round 6).
//
// online_rowpass_kernel (online_kernels.hpp) is bound by v_mfma_f32_16x16x4_f32: a workgroup's 16
// rows cost 3.4-4.1 us of matrix time per 256 x 256 layer at the fp32 rate of ONE CU, three times per
// round, on the critical path of the learn loop.  Here every operand is scaled by an exact power of
// two into fp16's range (activations per batch row, weights per output unit) and split into TWO
// fp16 terms, x 2^s = hi + lo with |x 2^s - hi - lo| <= 2^-24 |x 2^s| (2 x 11 significand bits and
// lo's sign), and a product is the three partial products  hi hi | hi lo + lo hi  on
// v_mfma_f32_16x16x32_f16, fp32 accumulation, one accumulator per magnitude class; the dropped
// lo lo term is below 2^-22 of the product.  The result is unscaled with one ldexp.  Error relative
// to sum |a b|: 1.0e-7 (tools/split_mfma_bench.hip: the fp32 MFMA chain 1.8e-7, bf16x3 6.5e-8).
//
// Three matrix instructions of 16 cycles replace eight of 32 per 32 k; the weights are split in
// registers as they stream in (the SAME fragment-major fp32 copies W1f / W2f / W2tf: two k-groups of
// 16 make one k-step of 32, lane quarter qd owning k = 32 s + 4 qd + j and 32 s + 16 + 4 qd + j on
// both operands), the activations once by the lane that produced them.
//
// Scales.  `umax` holds the bit pattern of max |w| per ROW of the online weights: [H1] rows of W1 |
// [H2] rows of W2.  It is maintained by whoever changes those weights: the AdamW epilogue of the
// weight-gradient kernel and the stand-alone AdamW by atomic max into the buffer the NEXT row pass
// reads (two buffers; the one just read is cleared for the round after), repack_body from the
// parameters themselves.  An activation row's scale comes from its own maximum (layer 1: inside
// every wave; hidden layer: across the waves through LDS, one more barrier).  The backward product
// G = s2 W2 reduces over the rows of W2, so there the row scale rides on the OTHER operand:
// G[r][k] = sum_n (s2[r][n] 2^-e_n) (W2[n][k] 2^e_n), the weight fragments taking a per-element
// scale (free: the scale is an operand of the conversion instruction).  A scaled maximum lies in
// [2^14, 2^15): nothing overflows fp16, a term 2^-17 below its row's maximum still has all 22 bits,
// and the absolute floor is 2^-40 of the maximum (fp16 subnormals are not flushed by the matrix
// pipe).  Non-finite rows are not scaled; inf may surface as NaN (hi = inf, lo = inf - inf), as in §3.2.
//
// The split itself is four instructions per pair of values: v_fma_mix{lo,hi}_f16 compute
// hi = f16(x s) and lo = f16(x s - hi) with ONE rounding each straight from the fp32 value (bitwise
// the cvt / sub / cvt sequence, tools/mixtest.hip).
//
// Shape: K1 <= 144 (nine k-groups), H1 = H2 = 256 — the instantiation <9, 16, 16> of the fp32 kernel.
#pragma once
#include "online_kernels.hpp"
#include "h2_common.hpp"

namespace pa {

__device__ __forceinline__ f32x4v mfma16h(const f16x8& a, const f16x8& b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// eight fp32 values (two float4 of one lane's k-step) times exact powers of two -> hi, lo
template <int D = 0>
__device__ __forceinline__ void h2_split8(const float4& a, const float4& b, float s, f16x8& hi, f16x8& lo) {
  unsigned h[4], l[4];
  h2_pair(a.x, a.y, s, s, h[0], l[0]);
  h2_pair(a.z, a.w, s, s, h[1], l[1]);
  h2_pair(b.x, b.y, s, s, h[2], l[2]);
  h2_pair(b.z, b.w, s, s, h[3], l[3]);
  hi = __builtin_bit_cast(f16x8, (u32x4v){h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(f16x8, (u32x4v){l[0], l[1], l[2], l[3]});
}
// ... with a scale per element (sa, sb: the scales of a's and b's four values)
template <int D = 0>
__device__ __forceinline__ void h2_split8v(const float4& a, const float4& b, const float4& sa, const float4& sb,
                                           f16x8& hi, f16x8& lo) {
  unsigned h[4], l[4];
  h2_pair(a.x, a.y, sa.x, sa.y, h[0], l[0]);
  h2_pair(a.z, a.w, sa.z, sa.w, h[1], l[1]);
  h2_pair(b.x, b.y, sb.x, sb.y, h[2], l[2]);
  h2_pair(b.z, b.w, sb.z, sb.w, h[3], l[3]);
  hi = __builtin_bit_cast(f16x8, (u32x4v){h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(f16x8, (u32x4v){l[0], l[1], l[2], l[3]});
}

constexpr int HF_PITCH = 256 + 8;   // halfs per LDS plane row: 528 B = 4 dwords mod 64 banks
// accumulators of the hi hi class per tile (k-steps alternate between them, added pairwise at the
// end): the class carries the whole sum, its chain of roundings is what the Q-values see
#ifndef RP_H2_NHH
#define RP_H2_NHH 2
#endif
constexpr int HF_NACC = RP_H2_NHH + 1;   // + one for the cross terms hi lo + lo hi

struct HalfFields {   // this lane's view of one umax segment
  float sA[2];        // scale of the A-operand unit of tile t: unit 32 wave + 16 t + r16
  int fC[2][4];       // fields of the C-units of tile t: 32 wave + 16 t + 4 qd + reg
};
__device__ __forceinline__ HalfFields half_fields(const unsigned* __restrict__ um, int wave, int r16, int qd) {
  HalfFields f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f.sA[t] = h2_scale(h2_field(um[32 * wave + 16 * t + r16]));
    const uint4 v = *reinterpret_cast<const uint4*>(um + 32 * wave + 16 * t + 4 * qd);
    f.fC[t][0] = h2_field(v.x); f.fC[t][1] = h2_field(v.y);
    f.fC[t][2] = h2_field(v.z); f.fC[t][3] = h2_field(v.w);
  }
  return f;
}

__device__ __forceinline__ void h2_zero(f32x4v (&c)[2][HF_NACC]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < HF_NACC; ++j) c[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
}
// the six matrix instructions of one k-step (two tiles): cross terms first, the tiles interleaved
// so that no instruction waits for the one before it
__device__ __forceinline__ void h2_step(f32x4v (&c)[2][HF_NACC], int s, const f16x8& ah0, const f16x8& al0,
                                        const f16x8& ah1, const f16x8& al1, const f16x8& xh, const f16x8& xl) {
  const int hh = s % RP_H2_NHH;
  c[0][RP_H2_NHH] = mfma16h(ah0, xl, c[0][RP_H2_NHH]);
  c[1][RP_H2_NHH] = mfma16h(ah1, xl, c[1][RP_H2_NHH]);
  c[0][RP_H2_NHH] = mfma16h(al0, xh, c[0][RP_H2_NHH]);
  c[1][RP_H2_NHH] = mfma16h(al1, xh, c[1][RP_H2_NHH]);
  c[0][hh] = mfma16h(ah0, xh, c[0][hh]);
  c[1][hh] = mfma16h(ah1, xh, c[1][hh]);
}

// One 256-deep layer over 8 k-steps (16 k-groups of the fp32 fragment stream).  R holds the first
// RP_PD k-groups on entry; in the iterations that have no refill of their own one k-group of the
// NEXT weight stream is requested into Rn (see rows16_gemm_static_pf).  bh / bl: this lane's
// (row, qd) slot of the hi / lo plane.  VSCALE: the weights take a scale per reduction index, read
// from `sv` (LDS, [256] floats: this lane's quarter reads sv[32 s + 4 qd ..] and sv[32 s + 16 + 4 qd ..]).
template <int NKG, int NKGN, bool VSCALE>
__device__ __forceinline__ void rows16_gemm_h2(f32x4v (&c)[2][HF_NACC], WRing& R, const float* __restrict__ Wf,
                                               int tile0, int lane, const float (&sA)[2], const float* sv,
                                               const _Float16* bh, const _Float16* bl, WRing& Rn,
                                               const float* __restrict__ Wn, bool want_next) {
  static_assert((NKG & 1) == 0, "whole k-steps");
  const int64_t base0 = ((int64_t)tile0 * NKG) * 256 + lane * 4;
  const int64_t base1 = base0 + (int64_t)NKG * 256;
  const int64_t nb0 = ((int64_t)tile0 * NKGN) * 256 + lane * 4;
  const int64_t nb1 = nb0 + (int64_t)NKGN * 256;
  constexpr int FREE0 = NKG > RP_PD ? NKG - RP_PD : 0;
  constexpr int NSLOT = NKGN < RP_PD ? NKGN : RP_PD;
#pragma unroll
  for (int s = 0; s < NKG / 2; ++s) {
    float4 w0[2], w1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int g = 2 * s + e;
      w0[e] = R.r0[g % RP_PD];
      w1[e] = R.r1[g % RP_PD];
      if (g + RP_PD < NKG) {
        R.r0[g % RP_PD] = ld4_or_zero(Wf, base0 + (int64_t)(g + RP_PD) * 256, true);
        R.r1[g % RP_PD] = ld4_or_zero(Wf, base1 + (int64_t)(g + RP_PD) * 256, true);
      } else if (NKGN > 0 && g - FREE0 < NSLOT) {
        Rn.r0[g - FREE0] = ld4_or_zero(Wn, nb0 + (int64_t)(g - FREE0) * 256, want_next);
        Rn.r1[g - FREE0] = ld4_or_zero(Wn, nb1 + (int64_t)(g - FREE0) * 256, want_next);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const f16x8 xh = *reinterpret_cast<const f16x8*>(bh + s * 32);
    const f16x8 xl = *reinterpret_cast<const f16x8*>(bl + s * 32);
    f16x8 ah0, al0, ah1, al1;
    if constexpr (VSCALE) {
      const float4 sa = *reinterpret_cast<const float4*>(sv + s * 32);
      const float4 sb = *reinterpret_cast<const float4*>(sv + s * 32 + 16);
      h2_split8v(w0[0], w0[1], sa, sb, ah0, al0);
      h2_split8v(w1[0], w1[1], sa, sb, ah1, al1);
    } else {
      h2_split8(w0[0], w0[1], sA[0], ah0, al0);
      h2_split8(w1[0], w1[1], sA[1], ah1, al1);
    }
    h2_step(c, s, ah0, al0, ah1, al1, xh, xl);
  }
  if constexpr (NKGN > 0) {
#pragma unroll
    for (int p = NKG - FREE0; p < NSLOT; ++p) {
      Rn.r0[p] = ld4_or_zero(Wn, nb0 + (int64_t)p * 256, want_next);
      Rn.r1[p] = ld4_or_zero(Wn, nb1 + (int64_t)p * 256, want_next);
    }
  }
}

// z[t][reg] = (sum of the classes, smallest first) 2^(fB + fC - 282) + seed
__device__ __forceinline__ void h2_finish(f32x4v (&acc)[2], const f32x4v (&c)[2][HF_NACC], int fB,
                                          const int (&fC)[2][4], const float4 (&seed)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float sd[4] = {seed[t].x, seed[t].y, seed[t].z, seed[t].w};
    f32x4v hh = c[t][0];
#pragma unroll
    for (int j = 1; j < RP_H2_NHH; ++j) hh = hh + c[t][j];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[t][r] = __fadd_rn(ldexpf(__fadd_rn(hh[r], c[t][RP_H2_NHH][r]), fB + fC[t][r] - 282), sd[r]);
  }
}

inline size_t rowpass_h2_smem_bytes() {
  // hi / lo planes of h1 and of s2, the row scales of W2, per-wave row maxima, head partials
  return (size_t)4 * RP_ROWS * HF_PITCH * sizeof(_Float16) + sizeof(float) * HF_UNITS +
         sizeof(unsigned) * 8 * 16 + sizeof(float) * 8 * 16;
}

// RowArgs::umax != null.  PH as in online_rowpass_kernel.
template <int PH = 0>
static __global__ __launch_bounds__(512) void online_rowpass_h2_kernel(RowArgs a) {
  constexpr int NG1 = 9, NG2 = 16, NG3 = 16, NS1 = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* h1h = reinterpret_cast<_Float16*>(smem_raw);     // [16][HF_PITCH]
  _Float16* h1l = h1h + RP_ROWS * HF_PITCH;
  _Float16* s2h = h1l + RP_ROWS * HF_PITCH;
  _Float16* s2l = s2h + RP_ROWS * HF_PITCH;
  float* rsv = reinterpret_cast<float*>(s2l + RP_ROWS * HF_PITCH);           // [256] 2^(141 - e_n) of W2's rows
  unsigned* rmaxw = reinterpret_cast<unsigned*>(rsv + HF_UNITS);             // [8][16]
  float* qpart = reinterpret_cast<float*>(rmaxw + 8 * 16);                   // [8][16]

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
  const int u0 = wave * 32 + 4 * qd;
  const int tile0 = wave * 2;
  const int plane_off = r16 * HF_PITCH + 8 * qd;     // this lane's slot of k-step 0 (consumer view)
  const unsigned* um1 = a.umax;
  const unsigned* um2 = a.umax + HF_UNITS;

  f32x4v acc[2];
  WRing R3;
  float4 h1k[2], h2k[2], w3v[2];
  const float b3v = a.b3[0];
  auto vec4 = [&](const float* p, int col, int n) { return ld4_or_zero(p, col, col < n); };
  float part = 0.f;
  // the backward product's B operand is s2[n] 2^-e_n = [h2 > 0] t[n], t[n] = w3[n] 2^(e_n - 141)
  // (e_n: the row scale of W2 carried over); its own scale comes from max |t|: every wave forms
  // all 256 t (64 lanes x 4)
  const float4 w3all = vec4(a.w3, 4 * lane, a.H2);
  const uint4 um2all = *reinterpret_cast<const uint4*>(um2 + 4 * lane);
  const HalfFields F2 = half_fields(um2, wave, r16, qd);
  if (tid < HF_UNITS) rsv[tid] = h2_scale(h2_field(um2[tid]));
  int f3 = 0;
  auto inv_scale = [](int field) { return __uint_as_float((unsigned)(field - 14) << 23); };   // 2^(field - 141)
  auto head_scale = [&]() {
    const float t0 = w3all.x * inv_scale(h2_field(um2all.x)), t1 = w3all.y * inv_scale(h2_field(um2all.y));
    const float t2 = w3all.z * inv_scale(h2_field(um2all.z)), t3 = w3all.w * inv_scale(h2_field(um2all.w));
    unsigned m = umax4(make_float4(t0, t1, t2, t3));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = umaxu(m, (unsigned)__shfl_xor((int)m, o));
    f3 = h2_field(m);
  };
  // this lane's eight units are k-step `wave` of the backward product
  auto store_s2 = [&]() {
    const float s3 = h2_scale(f3);
    float4 z[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      z[t].x = (rok && h2k[t].x > 0.f) ? w3v[t].x * inv_scale(F2.fC[t][0]) : 0.f;
      z[t].y = (rok && h2k[t].y > 0.f) ? w3v[t].y * inv_scale(F2.fC[t][1]) : 0.f;
      z[t].z = (rok && h2k[t].z > 0.f) ? w3v[t].z * inv_scale(F2.fC[t][2]) : 0.f;
      z[t].w = (rok && h2k[t].w > 0.f) ? w3v[t].w * inv_scale(F2.fC[t][3]) : 0.f;
    }
    f16x8 hi, lo;
    h2_split8(z[0], z[1], s3, hi, lo);
    *reinterpret_cast<f16x8*>(s2h + plane_off + 32 * wave) = hi;
    *reinterpret_cast<f16x8*>(s2l + plane_off + 32 * wave) = lo;
  };
  {
    WRing R2;
    float4 b1v[2], b2v[2];
    float4 xf[2 * NS1], wa[2 * NS1], wb[2 * NS1];
    {
      const bool vx = is_vec_ok(a.x, a.ldx) && ((a.K1 & 3) == 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) b1v[t] = vec4(a.b1, u0 + 16 * t, a.H1);
      const int64_t base0 = ((int64_t)tile0 * NG1) * 256 + lane * 4;
      const int64_t base1 = base0 + (int64_t)NG1 * 256;
#pragma unroll
      for (int g = 0; g < 2 * NS1; ++g) {
        const int c = 16 * g + 4 * qd;
        if (g < NG1) {
          if (vx) xf[g] = ld4_or_zero(a.x, (int64_t)row * a.ldx + c, rok && c < a.K1);
          else xf[g] = guarded_load4(a.x, (int64_t)row * a.ldx, rok, c, a.K1);
          wa[g] = ld4_or_zero(a.W1f, base0 + (int64_t)g * 256, true);
          wb[g] = ld4_or_zero(a.W1f, base1 + (int64_t)g * 256, true);
        } else {
          xf[g] = wa[g] = wb[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    const HalfFields F1 = half_fields(um1, wave, r16, qd);
    ring_fill<NG2>(R2, a.W2f, tile0, 16, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      b2v[t] = vec4(a.b2, u0 + 16 * t, a.H2);
      w3v[t] = vec4(a.w3, u0 + 16 * t, a.H2);
    }
    PA_STAMP(a.prof, blockIdx.x, wave, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer 1: the row's maximum is inside the wave (a lane quarter holds every fourth float4)
    int fx;
    {
      unsigned m = 0u;
#pragma unroll
      for (int g = 0; g < NG1; ++g) m = umaxu(m, umax4(xf[g]));
      m = umaxu(m, (unsigned)__shfl_xor((int)m, 16));
      m = umaxu(m, (unsigned)__shfl_xor((int)m, 32));
      fx = h2_field(m);
    }
    {
      const float sx = h2_scale(fx);
      f32x4v c[2][HF_NACC];
      h2_zero(c);
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        f16x8 xh, xl, ah0, al0, ah1, al1;
        h2_split8(xf[2 * s], xf[2 * s + 1], sx, xh, xl);
        h2_split8(wa[2 * s], wa[2 * s + 1], F1.sA[0], ah0, al0);
        h2_split8(wb[2 * s], wb[2 * s + 1], F1.sA[1], ah1, al1);
        h2_step(c, s, ah0, al0, ah1, al1, xh, xl);
      }
      h2_finish(acc, c, fx, F1.fC, b1v);
    }
    PA_STAMP(a.prof, blockIdx.x, wave, 2);
    unsigned hm = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u = u0 + 16 * t;
      h1k[t] = make_float4(relu_keep_nan(acc[t][0]), relu_keep_nan(acc[t][1]),
                           relu_keep_nan(acc[t][2]), relu_keep_nan(acc[t][3]));
      hm = umaxu(hm, umax4(h1k[t]));
      if (rok && a.H1a) *reinterpret_cast<float4*>(a.H1a + (int64_t)row * a.H1 + u) = h1k[t];
    }
    hm = umaxu(hm, (unsigned)__shfl_xor((int)hm, 16));
    hm = umaxu(hm, (unsigned)__shfl_xor((int)hm, 32));
    if (qd == 0) rmaxw[wave * 16 + r16] = hm;
    head_scale();
    PA_STAMP(a.prof, blockIdx.x, wave, 3);
    __syncthreads();                                                      // barrier A0: row maxima
    int fh;
    {
      unsigned m = rmaxw[r16];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = umaxu(m, rmaxw[w * 16 + r16]);
      fh = h2_field(m);
      f16x8 hi, lo;
      h2_split8(h1k[0], h1k[1], h2_scale(fh), hi, lo);
      *reinterpret_cast<f16x8*>(h1h + plane_off + 32 * wave) = hi;
      *reinterpret_cast<f16x8*>(h1l + plane_off + 32 * wave) = lo;
    }
    __syncthreads();                                                      // barrier A: h1 planes
    PA_STAMP(a.prof, blockIdx.x, wave, 4);
    // ---- layer 2
    {
      f32x4v c[2][HF_NACC];
      h2_zero(c);
      rows16_gemm_h2<NG2, NG3, false>(c, R2, a.W2f, tile0, lane, F2.sA, nullptr, h1h + plane_off,
                                      h1l + plane_off, R3, a.W2tf, a.y != nullptr);
      h2_finish(acc, c, fh, F2.fC, b2v);
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
      if (rok && a.H2a) *reinterpret_cast<float4*>(a.H2a + (int64_t)row * a.H2 + u) = h2k[t];
    }
    if (a.y) store_s2();
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (qd == 0) qpart[wave * 16 + r16] = part;
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 6);
  __syncthreads();                                                        // barrier B: s2, qpart
  PA_STAMP(a.prof, blockIdx.x, wave, 7);
  unsigned ybits = kYPendingBits;
  if (PH == 0 && a.y && rok) {
    // first look at the Bellman target, in flight while the backward GEMM runs
    ybits = a.y_tagged ? __hip_atomic_load(reinterpret_cast<const unsigned*>(a.y) + row,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : __builtin_bit_cast(unsigned, a.y[row]);
  }
  if (a.y) {
    // ---- G = s2 W2 (scaled by dq below): the weights' scale depends on the reduction index
    f32x4v c[2][HF_NACC];
    h2_zero(c);
    WRing none;
    const float unused[2] = {0.f, 0.f};
    rows16_gemm_h2<NG3, 0, true>(c, R3, a.W2tf, tile0, lane, unused, rsv + 4 * qd, s2h + plane_off,
                                 s2l + plane_off, none, nullptr, false);
    const float4 zero2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int fG[2][4] = {{141, 141, 141, 141}, {141, 141, 141, 141}};
    h2_finish(acc, c, f3, fG, zero2);
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 8);
  float q = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) q += qpart[w * 16 + r16];
  q += b3v;
  if (wave == 0 && qd == 0 && rok && a.q_out) a.q_out[row] = q;
  if (!a.y) return;
  if constexpr (PH == 1) {
    // the forward launch of a window's first round: everything that does not need the Bellman
    // target, i.e. also the masked backward factors  s2 = [h2 > 0] w3  and  G [h1 > 0]  (in the
    // places of dZ2 / dZ1); rowpass_scale_kernel multiplies them by dq once the targets exist
    if (rok) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int u = u0 + 16 * t;
        float4 z;
        z.x = (h2k[t].x > 0.f) ? w3v[t].x : 0.f;
        z.y = (h2k[t].y > 0.f) ? w3v[t].y : 0.f;
        z.z = (h2k[t].z > 0.f) ? w3v[t].z : 0.f;
        z.w = (h2k[t].w > 0.f) ? w3v[t].w : 0.f;
        *reinterpret_cast<float4*>(a.dZ2 + (int64_t)row * a.H2 + u) = z;
        float4 g;
        g.x = (h1k[t].x > 0.f) ? acc[t][0] : 0.f;
        g.y = (h1k[t].y > 0.f) ? acc[t][1] : 0.f;
        g.z = (h1k[t].z > 0.f) ? acc[t][2] : 0.f;
        g.w = (h1k[t].w > 0.f) ? acc[t][3] : 0.f;
        *reinterpret_cast<float4*>(a.dZ1 + (int64_t)row * a.H1 + u) = g;
      }
    }
    PA_STAMP(a.prof, blockIdx.x, wave, 10);
    return;
  }
  float yv = q;
  if (rok) {
    if (a.y_tagged && ybits == kYPendingBits) yv = consume_y(a.y + row, a.err, a.err_host);
    else yv = __builtin_bit_cast(float, ybits);
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
      *reinterpret_cast<float4*>(a.dZ2 + (int64_t)row * a.H2 + u) = z;
      float4 g;
      g.x = (h1k[t].x > 0.f) ? __fmul_rn(dq, acc[t][0]) : 0.f;
      g.y = (h1k[t].y > 0.f) ? __fmul_rn(dq, acc[t][1]) : 0.f;
      g.z = (h1k[t].z > 0.f) ? __fmul_rn(dq, acc[t][2]) : 0.f;
      g.w = (h1k[t].w > 0.f) ? __fmul_rn(dq, acc[t][3]) : 0.f;
      *reinterpret_cast<float4*>(a.dZ1 + (int64_t)row * a.H1 + u) = g;
    }
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 10);
  PA_STAMP_CYC(a.prof, blockIdx.x, wave, 15);
}


// The backward launch of a window's first round (after online_rowpass_h2_kernel<1>): dq = norm (q - y)
// per row once its Bellman target exists, dZ2 = dq s2 and dZ1 = dq G [h1 > 0] in place — the values
// online_rowpass_h2_kernel<0> stores (one fp32 rounding per element, zeros where the mask is closed).
// Sixteen rows per workgroup; no weights, 12 registers: it is resident beside the leading target
// tiles (which own their CUs' register files against the full row pass) and ends ~2 us after the
// window's first targets exist, where the backward half of the row pass took 7.6 us from there.
// ONE quarter-wave per workgroup polls (agent-scope loads bypass the L2 and share the memory system
// with the tiles the launch is waiting for).
static __global__ __launch_bounds__(256) void rowpass_scale_kernel(RowArgs a) {
  __shared__ float dqs[RP_ROWS];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * RP_ROWS;
  // this thread's operands first: they do not depend on the target.  Row r of the tile is 128
  // float4 (dZ2 | dZ1, H1 = H2 = 256); thread t owns float4 (t & 127) of rows (t >> 7) + 2 i
  float4 v[8];
  const int c = tid & 127, r0 = tid >> 7;
  float* base = c < 64 ? a.dZ2 : a.dZ1;
  const int cc = (c & 63) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + r0 + 2 * i;
    v[i] = ld4_or_zero(base, (int64_t)row * HF_UNITS + cc, row < a.B);
  }
  if (tid < RP_ROWS) {
    const int row = m0 + tid;
    float dq = 0.f;
    if (row < a.B) {
      const float q = a.q_in[row];
      const float yv = a.y_tagged ? consume_y(a.y + row, a.err, a.err_host) : a.y[row];
      const float d = __fsub_rn(q, yv);
      dq = __fmul_rn(a.norm, d);
      a.dq_out[row] = dq;
      a.absd_out[row] = fabsf(d);
    }
    dqs[tid] = dq;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + r0 + 2 * i;
    if (row >= a.B) continue;
    const float dq = dqs[r0 + 2 * i];
    auto sc = [&](float x) { return x != 0.f ? __fmul_rn(dq, x) : 0.f; };
    *reinterpret_cast<float4*>(base + (int64_t)row * HF_UNITS + cc) =
        make_float4(sc(v[i].x), sc(v[i].y), sc(v[i].z), sc(v[i].w));
  }
}

}  // namespace pa
