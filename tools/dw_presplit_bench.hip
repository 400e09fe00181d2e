// Timing experiment (round 6): what the 4096-row weight-gradient tiles would cost if dZ / X arrived
// as bf16x3 split PLANES written by the row step instead of fp32 rows split by every consuming tile.
//
// Today (dw_mainloop_split, dqn_kernels.hpp): a wave's 32-row step is 16 loads, 255 VALU (216 of
// them the three-way split of 48 fp32 values) and 48 v_mfma_f32_16x16x32_bf16 — issue-bound, a
// 4096-row tile is 28-30 us of main loop.  Candidates, same lanes / same products / same order:
//   A  the current loop (fp32 rows, split8x2 in the consumer)                      [baseline]
//   B  row-major planes [3][B][ld] bf16: 8-byte (dZ) / 4-byte (X) loads per row and plane, the
//      8-rows-per-lane operand formed with v_perm_b32 (72 per step); no column sums
//   B1 B + the column sums db as three extra MFMAs per unit block against a ones fragment
//   C  chunked planes [3][B / 8][ld][8] bf16: one 16-byte load IS the MFMA operand; no VALU at all
//      (what a producer-side transposition would buy on top of B)
// B and C must give A's accumulators bit for bit (same split values, same product order).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dw_presplit_bench.hip -o tools/dw_presplit_bench
//   tools/dw_presplit_bench [workgroups=144]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

constexpr int B = 4096, M = 256, N = 256;   // dW[M units][N cols] = dZ[B][M]^T X[B][N]
constexpr int UPL = 4, TM = 64, TN = 32;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ void split8x2(const f32x2 (&v)[8], bf16x8 (&hi)[2], bf16x8 (&mid)[2],
                                         bf16x8 (&lo)[2]) {
  u32x4 H[2], Mi[2], L[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f32x2 xa = v[2 * p], xb = v[2 * p + 1];
    const unsigned h0 = cvt_pk_bf16(xa[0], xb[0]), h1 = cvt_pk_bf16(xa[1], xb[1]);
    const f32x2 ha = {__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h1 << 16)};
    const f32x2 hb = {__builtin_bit_cast(float, h0 & 0xffff0000u), __builtin_bit_cast(float, h1 & 0xffff0000u)};
    const f32x2 ra = xa - ha, rb = xb - hb;
    const unsigned m0 = cvt_pk_bf16(ra[0], rb[0]), m1 = cvt_pk_bf16(ra[1], rb[1]);
    const f32x2 ma = {__builtin_bit_cast(float, m0 << 16), __builtin_bit_cast(float, m1 << 16)};
    const f32x2 mb = {__builtin_bit_cast(float, m0 & 0xffff0000u), __builtin_bit_cast(float, m1 & 0xffff0000u)};
    const f32x2 sa = ra - ma, sb = rb - mb;
    H[0][p] = h0; H[1][p] = h1;
    Mi[0][p] = m0; Mi[1][p] = m1;
    L[0][p] = cvt_pk_bf16(sa[0], sb[0]); L[1][p] = cvt_pk_bf16(sa[1], sb[1]);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    hi[u] = __builtin_bit_cast(bf16x8, H[u]);
    mid[u] = __builtin_bit_cast(bf16x8, Mi[u]);
    lo[u] = __builtin_bit_cast(bf16x8, L[u]);
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

#define PA_DW_CLASS(A_, X_, ACC_)                                                          \
  _Pragma("unroll") for (int ja = 0; ja < UPL; ++ja)                                        \
  _Pragma("unroll") for (int jx = 0; jx < 2; ++jx)                                          \
      ACC_[ja][jx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[ja], X_[jx], ACC_[ja][jx], 0, 0, 0);
#define PA_SIX_CLASSES                                                                      \
  PA_DW_CLASS(al, xh, small)                                                                \
  PA_DW_CLASS(ah, xl, small)                                                                \
  PA_DW_CLASS(am, xm, small)                                                                \
  PA_DW_CLASS(ah, xm, small)                                                                \
  PA_DW_CLASS(am, xh, small)                                                                \
  PA_DW_CLASS(ah, xh, acc)

struct Args {
  const float* dZ; const float* X;           // fp32 rows
  const __bf16* dZp; const __bf16* Xp;       // row-major planes [3][B][ld]
  const bf16x8* dZc; const bf16x8* Xc;       // chunked planes [3][B / 8][ld]
  float* out;                                // [workgroups][512][32 + 4] accumulators (+ cs)
  int variant;
};

// the epilogue every variant shares: acc + small, written per lane (no cross-wave reduction: the
// comparison is per wave, which is stricter)
__device__ __forceinline__ void finish(const Args& a, f32x4 (&acc)[UPL][2], f32x4 (&small)[UPL][2],
                                       float (&cs)[UPL]) {
  float* o = a.out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 36;
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja)
#pragma unroll
    for (int jx = 0; jx < 2; ++jx)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(ja * 2 + jx) * 4 + r] = acc[ja][jx][r] + small[ja][jx][r];
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja) o[32 + ja] = cs[ja];
}

struct RawA { float a[8][4]; float x[8][2]; };
struct RawB { u32x2 a[3][8]; unsigned x[3][8]; };
struct RawC { bf16x8 a[3][UPL]; bf16x8 x[3][2]; };

template <int V>
__global__ __launch_bounds__(512, 2) void dw_kernel(Args a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int t = blockIdx.x & 31;
  const int i0 = (t >> 3) * TM, j0 = (t & 7) * TN;
  const int ua = i0 + UPL * c, cx = j0 + 2 * c;
  const int nsteps = B / 256, row0 = wave * (B / 8);
  f32x4 acc[UPL][2], small[UPL][2];
  float cs[UPL];
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja) {
    acc[ja][0] = acc[ja][1] = small[ja][0] = small[ja][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    cs[ja] = 0.f;
  }
  if constexpr (V == 0 || V == 4) {
    const __amdgpu_buffer_rsrc_t ra = rsrc(a.dZ, (unsigned)B * M * 4u), rx = rsrc(a.X, (unsigned)B * N * 4u);
    const unsigned va = (unsigned)(8 * q * M + ua) * 4u, vx = (unsigned)(8 * q * N + cx) * 4u;
    auto fetch = [&](int step, RawA& f) {
      const bool live = step < nsteps;
      const unsigned oa = live ? (unsigned)(row0 + 32 * step) * M * 4u : 0x80000000u;
      const unsigned ox = live ? (unsigned)(row0 + 32 * step) * N * 4u : 0x80000000u;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f32x4 f4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(va + oa + (unsigned)r * M * 4u), 0, 0));
        f.a[r][0] = f4[0]; f.a[r][1] = f4[1]; f.a[r][2] = f4[2]; f.a[r][3] = f4[3];
        const f32x2 f2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(vx + ox + (unsigned)r * N * 4u), 0, 0));
        f.x[r][0] = f2[0]; f.x[r][1] = f2[1];
      }
    };
    auto work = [&](const RawA& cur) {
      bf16x8 xh[2], xm[2], xl[2], ah[UPL], am[UPL], al[UPL];
      if constexpr (V == 4) {
        // (timing only: the raw registers as operands, no split)
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
          xh[jx] = __builtin_bit_cast(bf16x8, f32x4{cur.x[0][jx], cur.x[1][jx], cur.x[2][jx], cur.x[3][jx]});
          xm[jx] = __builtin_bit_cast(bf16x8, f32x4{cur.x[4][jx], cur.x[5][jx], cur.x[6][jx], cur.x[7][jx]});
          xl[jx] = __builtin_bit_cast(bf16x8, f32x4{cur.x[1][jx], cur.x[3][jx], cur.x[5][jx], cur.x[7][jx]});
        }
#pragma unroll
        for (int ja = 0; ja < UPL; ++ja) {
          ah[ja] = __builtin_bit_cast(bf16x8, f32x4{cur.a[0][ja], cur.a[1][ja], cur.a[2][ja], cur.a[3][ja]});
          am[ja] = __builtin_bit_cast(bf16x8, f32x4{cur.a[4][ja], cur.a[5][ja], cur.a[6][ja], cur.a[7][ja]});
          al[ja] = __builtin_bit_cast(bf16x8, f32x4{cur.a[1][ja], cur.a[3][ja], cur.a[5][ja], cur.a[7][ja]});
        }
        PA_SIX_CLASSES
        return;
      }
      {
        f32x2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = f32x2{cur.x[r][0], cur.x[r][1]};
        split8x2(v, xh, xm, xl);
      }
#pragma unroll
      for (int jp = 0; jp < UPL; jp += 2) {
        f32x2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          v[r] = f32x2{cur.a[r][jp], cur.a[r][jp + 1]};
          cs[jp] += v[r][0];
          cs[jp + 1] += v[r][1];
        }
        bf16x8 h2[2], m2[2], l2[2];
        split8x2(v, h2, m2, l2);
        ah[jp] = h2[0]; ah[jp + 1] = h2[1];
        am[jp] = m2[0]; am[jp + 1] = m2[1];
        al[jp] = l2[0]; al[jp + 1] = l2[1];
      }
      PA_SIX_CLASSES
    };
    RawA r0, r1;
    fetch(0, r0);
    for (int s = 0; s < nsteps; s += 2) {
      fetch(s + 1, r1);
      __builtin_amdgcn_sched_barrier(0);
      work(r0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s + 2, r0);
      __builtin_amdgcn_sched_barrier(0);
      work(r1);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (V == 1 || V == 2) {
    // row-major planes: plane p of dZ at p * B * M elements
    const __amdgpu_buffer_rsrc_t ra = rsrc(a.dZp, 3u * B * M * 2u), rx = rsrc(a.Xp, 3u * B * N * 2u);
    const unsigned va = (unsigned)(8 * q * M + ua) * 2u, vx = (unsigned)(8 * q * N + cx) * 2u;
    auto fetch = [&](int step, RawB& f) {
      const bool live = step < nsteps;
      const unsigned oa = live ? (unsigned)(row0 + 32 * step) * M * 2u : 0x80000000u;
      const unsigned ox = live ? (unsigned)(row0 + 32 * step) * N * 2u : 0x80000000u;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          f.a[p][r] = __builtin_amdgcn_raw_buffer_load_b64(ra, (int)(va + oa + (unsigned)(p * B + r) * M * 2u), 0, 0);
          f.x[p][r] = __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(vx + ox + (unsigned)(p * B + r) * N * 2u), 0, 0);
        }
    };
    const bf16x8 ones = []() {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)1.0f;
      return o;
    }();
    f32x4 csacc[UPL];
#pragma unroll
    for (int ja = 0; ja < UPL; ++ja) csacc[ja] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto work = [&](const RawB& cur) {
      bf16x8 xs[3][2], as[3][UPL];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x4 x0, x1, a0, a1, a2, a3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          x0[k] = __builtin_amdgcn_perm(cur.x[p][2 * k + 1], cur.x[p][2 * k], 0x05040100u);
          x1[k] = __builtin_amdgcn_perm(cur.x[p][2 * k + 1], cur.x[p][2 * k], 0x07060302u);
          a0[k] = __builtin_amdgcn_perm(cur.a[p][2 * k + 1][0], cur.a[p][2 * k][0], 0x05040100u);
          a1[k] = __builtin_amdgcn_perm(cur.a[p][2 * k + 1][0], cur.a[p][2 * k][0], 0x07060302u);
          a2[k] = __builtin_amdgcn_perm(cur.a[p][2 * k + 1][1], cur.a[p][2 * k][1], 0x05040100u);
          a3[k] = __builtin_amdgcn_perm(cur.a[p][2 * k + 1][1], cur.a[p][2 * k][1], 0x07060302u);
        }
        xs[p][0] = __builtin_bit_cast(bf16x8, x0); xs[p][1] = __builtin_bit_cast(bf16x8, x1);
        as[p][0] = __builtin_bit_cast(bf16x8, a0); as[p][1] = __builtin_bit_cast(bf16x8, a1);
        as[p][2] = __builtin_bit_cast(bf16x8, a2); as[p][3] = __builtin_bit_cast(bf16x8, a3);
      }
      bf16x8(&ah)[UPL] = as[0]; bf16x8(&am)[UPL] = as[1]; bf16x8(&al)[UPL] = as[2];
      bf16x8(&xh)[2] = xs[0]; bf16x8(&xm)[2] = xs[1]; bf16x8(&xl)[2] = xs[2];
      if constexpr (V == 2) {
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
          for (int ja = 0; ja < UPL; ++ja)
            csacc[ja] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[p][ja], ones, csacc[ja], 0, 0, 0);
      }
      PA_SIX_CLASSES
    };
    RawB r0, r1;
    fetch(0, r0);
    for (int s = 0; s < nsteps; s += 2) {
      fetch(s + 1, r1);
      __builtin_amdgcn_sched_barrier(0);
      work(r0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s + 2, r0);
      __builtin_amdgcn_sched_barrier(0);
      work(r1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (V == 2) {
#pragma unroll
      for (int ja = 0; ja < UPL; ++ja) cs[ja] = csacc[ja][0];
    }
  } else {
    // chunked planes: element (p, rowgroup, unit) is one bf16x8
    const __amdgpu_buffer_rsrc_t ra = rsrc(a.dZc, 3u * B * M * 2u), rx = rsrc(a.Xc, 3u * B * N * 2u);
    // (units / columns dealt out so that the 16 lanes of a quarter read 256 contiguous bytes:
    //  lane c's ja-th unit is i0 + 16 ja + c — a relabelling of the accumulators, not of the sums;
    //  with A's labelling (4 c + ja) every 128-byte line is touched by four separate loads: 40.4 us)
    const unsigned va = (unsigned)(q * M + i0 + c) * 16u, vx = (unsigned)(q * N + j0 + c) * 16u;
    auto fetch = [&](int step, RawC& f) {
      const bool live = step < nsteps;
      const unsigned oa = live ? (unsigned)((row0 + 32 * step) / 8) * M * 16u : 0x80000000u;
      const unsigned ox = live ? (unsigned)((row0 + 32 * step) / 8) * N * 16u : 0x80000000u;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int ja = 0; ja < UPL; ++ja)
          f.a[p][ja] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(va + oa + (unsigned)(p * (B / 8) * M + 16 * ja) * 16u), 0, 0));
#pragma unroll
        for (int jx = 0; jx < 2; ++jx)
          f.x[p][jx] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(vx + ox + (unsigned)(p * (B / 8) * N + 16 * jx) * 16u), 0, 0));
      }
    };
    auto work = [&](const RawC& cur) {
      const bf16x8(&ah)[UPL] = cur.a[0]; const bf16x8(&am)[UPL] = cur.a[1]; const bf16x8(&al)[UPL] = cur.a[2];
      const bf16x8(&xh)[2] = cur.x[0]; const bf16x8(&xm)[2] = cur.x[1]; const bf16x8(&xl)[2] = cur.x[2];
      PA_SIX_CLASSES
    };
    RawC r0, r1;
    fetch(0, r0);
    for (int s = 0; s < nsteps; s += 2) {
      fetch(s + 1, r1);
      __builtin_amdgcn_sched_barrier(0);
      work(r0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s + 2, r0);
      __builtin_amdgcn_sched_barrier(0);
      work(r1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  finish(a, acc, small, cs);
}

static void split3_host(float x, unsigned short (&o)[3]) {
  auto bf = [](float v) {
    unsigned u; memcpy(&u, &v, 4);
    const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
  };
  auto up = [](unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
  o[0] = bf(x);
  const float r = x - up(o[0]);
  o[1] = bf(r);
  const float s = r - up(o[1]);
  o[2] = bf(s);
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 144;
  std::vector<float> dz((size_t)B * M), x((size_t)B * N);
  srand(7);
  for (auto& v : dz) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : x) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  std::vector<unsigned short> dzp(3ull * B * M), xp(3ull * B * N), dzc(3ull * B * M), xc(3ull * B * N);
  for (int b = 0; b < B; ++b)
    for (int u = 0; u < M; ++u) {
      unsigned short s[3];
      split3_host(dz[(size_t)b * M + u], s);
      for (int p = 0; p < 3; ++p) {
        dzp[((size_t)p * B + b) * M + u] = s[p];
        dzc[(((size_t)p * (B / 8) + b / 8) * M + u) * 8 + (b & 7)] = s[p];
      }
      split3_host(x[(size_t)b * N + u], s);
      for (int p = 0; p < 3; ++p) {
        xp[((size_t)p * B + b) * N + u] = s[p];
        xc[(((size_t)p * (B / 8) + b / 8) * N + u) * 8 + (b & 7)] = s[p];
      }
    }
  Args a;
  memset(&a, 0, sizeof(a));
  void *d_dz, *d_x, *d_dzp, *d_xp, *d_dzc, *d_xc, *d_out;
  CHECK(hipMalloc(&d_dz, dz.size() * 4)); CHECK(hipMalloc(&d_x, x.size() * 4));
  CHECK(hipMalloc(&d_dzp, dzp.size() * 2)); CHECK(hipMalloc(&d_xp, xp.size() * 2));
  CHECK(hipMalloc(&d_dzc, dzc.size() * 2)); CHECK(hipMalloc(&d_xc, xc.size() * 2));
  const size_t out_n = (size_t)wgs * 512 * 36;
  CHECK(hipMalloc(&d_out, out_n * 4));
  CHECK(hipMemcpy(d_dz, dz.data(), dz.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_dzp, dzp.data(), dzp.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_xp, xp.data(), xp.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_dzc, dzc.data(), dzc.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_xc, xc.data(), xc.size() * 2, hipMemcpyHostToDevice));
  a.dZ = (const float*)d_dz; a.X = (const float*)d_x;
  a.dZp = (const __bf16*)d_dzp; a.Xp = (const __bf16*)d_xp;
  a.dZc = (const bf16x8*)d_dzc; a.Xc = (const bf16x8*)d_xc;
  a.out = (float*)d_out;
  std::vector<float> ref(out_n), got(out_n);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char* names[5] = {"A  fp32 rows, consumer split (current)", "B  row-major planes + v_perm",
                          "B1 row-major planes + v_perm + db by MFMA", "C  chunked planes (no VALU), coalesced labels",
                          "A0 fp32 rows, NO split (timing only: garbage operands)"};
  for (int v = 0; v < 5; ++v) {
    auto launch = [&]() {
      if (v == 0) hipLaunchKernelGGL(dw_kernel<0>, dim3(wgs), dim3(512), 0, 0, a);
      if (v == 1) hipLaunchKernelGGL(dw_kernel<1>, dim3(wgs), dim3(512), 0, 0, a);
      if (v == 2) hipLaunchKernelGGL(dw_kernel<2>, dim3(wgs), dim3(512), 0, 0, a);
      if (v == 3) hipLaunchKernelGGL(dw_kernel<3>, dim3(wgs), dim3(512), 0, 0, a);
      if (v == 4) hipLaunchKernelGGL(dw_kernel<4>, dim3(wgs), dim3(512), 0, 0, a);
    };
    CHECK(hipMemset(d_out, 0, out_n * 4));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(v == 0 ? ref.data() : got.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, badcs = 0;
    double maxcs = 0;
    if (v > 0)
      for (size_t i = 0; i < out_n; ++i) {
        const bool is_cs = (i % 36) >= 32;
        if (!is_cs && memcmp(&ref[i], &got[i], 4) != 0) ++bad;
        if (is_cs && v == 2) {
          const double d = fabs((double)ref[i] - (double)got[i]);
          if (d > maxcs) maxcs = d;
          if (d > 1e-3) ++badcs;
        }
      }
    for (int w = 0; w < 5; ++w) launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipEventRecord(e0, 0));
      launch();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      sum += ms;
    }
    printf("%-46s  %3d workgroups  best %7.2f us  mean %7.2f us  acc mismatches %zu%s", names[v], wgs,
           best * 1e3f, sum / reps * 1e3f, bad, v == 0 ? " (reference)" : "");
    if (v == 2) printf("  db max |diff| %.3g (bad %zu)", maxcs, badcs);
    printf("\n");
  }
  return 0;
}
