// Prototype + microbenchmark of the layer-2 main loop of the target-network tile on
// v_mfma_f32_32x32x16_bf16 with fp32 operands split three ways (x = hi + mid + lo, each a bf16:
// 3 x 8 = 24 significand bits, i.e. the fp32 value exactly), six products per k-step
//   hi*hi | hi*mid + mid*hi | hi*lo + lo*hi + mid*mid
// accumulated in fp32, in one accumulator or in three (one per magnitude class, added at the end).
// The dropped terms (mid*lo, lo*mid, lo*lo) are below 2^-24 of the product.
//
//   hipcc --offload-arch=gfx950 -O3 tools/split_mfma_bench.hip -o tools/split_mfma_bench
//   tools/split_mfma_bench            # correctness vs fp64 on the host, then timings
//
// Tile = 64 rows (batch x action) x 256 hidden units x K = 256, exactly target_tile<32, true> of
// dqn_kernels.hpp: wave w owns units [32 w, 32 w + 32), C[unit = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)]
// [row = 32 tm + (lane & 31)] — the C layout is the same for every 32x32 MFMA on gfx950.
// Operand layout of 32x32x16: lane l supplies A[i = l & 31][k = 8 (l >> 5) + e] and
// B[k = 8 (l >> 5) + e][j = l & 31], e = 0..7 (only the agreement of A's and B's k matters).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 64, H = 256, K = 256, KS = K / 16;   // 16 k-steps
constexpr int LDP = K + 8;                                 // bf16 pitch of an LDS row (528 B)

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

__host__ __device__ inline void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r = x - (float)hi;
  mid = (__bf16)r;
  const float r2 = r - (float)mid;
  lo = (__bf16)r2;
}

// W2 [H units][K] fp32 -> fragment-major split planes: Wsp[((w * KS + g) * 3 + s) * 64 + lane] (16 B)
__global__ void pack_w_kernel(const float* __restrict__ W, bf16x8* __restrict__ Wsp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // (w, g, lane)
  if (e >= 8 * KS * 64) return;
  const int lane = e & 63, g = (e >> 6) % KS, w = e / (64 * KS);
  const int unit = 32 * w + (lane & 31), k0 = 16 * g + 8 * (lane >> 5);
  bf16x8 p[3];
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(W[unit * K + k0 + j], a, b, c);
    p[0][j] = a; p[1][j] = b; p[2][j] = c;
  }
  for (int s = 0; s < 3; ++s) Wsp[((size_t)(w * KS + g) * 3 + s) * 64 + lane] = p[s];
}

// NACC: 1 = one accumulator per row tile, 3 = one per magnitude class.  RD: k-steps of weights in
// flight.  tiles: 64-row tiles per workgroup (the same inputs again: throughput measurement).
template <int NACC, int RD, bool FP32>
__global__ __launch_bounds__(512, 2) void tile_kernel(const float* __restrict__ h1,
                                                      const bf16x8* __restrict__ Wsp,
                                                      const float* __restrict__ Wf32,
                                                      float* __restrict__ out, int tiles,
                                                      int write_out, int zero) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem_raw);   // [3][ROWS][LDP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  float total = 0.f;
  for (int t = 0; t < tiles; ++t) {
    // ---- "prologue": the h1 tile, split, into LDS (the real kernel produces h1 in registers)
    for (int e = tid; e < ROWS * K / 4; e += 512) {
      const int r = e / (K / 4), c = (e % (K / 4)) * 4;
      const float4 v = *reinterpret_cast<const float4*>(h1 + r * K + c);
      const float x[4] = {v.x, v.y, v.z, v.w};
      __bf16 p[3][4];
      for (int j = 0; j < 4; ++j) split3(x[j], p[0][j], p[1][j], p[2][j]);
      for (int s = 0; s < 3; ++s) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 q = {p[s][0], p[s][1], p[s][2], p[s][3]};
        *reinterpret_cast<bf16x4*>(planes + ((size_t)s * ROWS + r) * LDP + c) = q;
      }
    }
    __syncthreads();
    f32x16 acc[2][NACC];
    for (int tm = 0; tm < 2; ++tm)
      for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[tm][c][r] = 0.f;
    if constexpr (FP32) {
      // reference loop: the fp32 MFMA of target_tile (k-group of 8: lane half h owns k = 8 g + 4 h + j)
      // on the un-split fp32 weights, B operand from the hi+mid+lo planes re-assembled (exact)
      for (int g = 0; g < K / 8; ++g) {
        float w4[4], x0[4], x1[4];
        for (int j = 0; j < 4; ++j) {
          const int k = 8 * g + 4 * hh + j;
          w4[j] = Wf32[(32 * wave + l31) * K + k];
          x0[j] = (float)planes[((size_t)0 * ROWS + l31) * LDP + k] + (float)planes[((size_t)1 * ROWS + l31) * LDP + k] +
                  (float)planes[((size_t)2 * ROWS + l31) * LDP + k];
          x1[j] = (float)planes[((size_t)0 * ROWS + 32 + l31) * LDP + k] + (float)planes[((size_t)1 * ROWS + 32 + l31) * LDP + k] +
                  (float)planes[((size_t)2 * ROWS + 32 + l31) * LDP + k];
        }
        for (int j = 0; j < 4; ++j) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[j], x0[j], acc[0][0], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[j], x1[j], acc[1][0], 0, 0, 0);
        }
      }
    } else {
      // ---- main loop: weights stream from the fragment-major planes through a register ring
      bf16x8 ring[RD][3];
      // (`zero` is 0 at run time: it only keeps the compiler from hoisting the whole weight stream
      // out of the tile loop — in the real kernel the tiles differ)
      const bf16x8* wp = Wsp + (size_t)wave * KS * 3 * 64 + lane + (size_t)zero * t;
#pragma unroll
      for (int g = 0; g < RD; ++g)
#pragma unroll
        for (int s = 0; s < 3; ++s) ring[g][s] = wp[(size_t)(g * 3 + s) * 64];
      const __bf16* bp0 = planes + (size_t)l31 * LDP + 8 * hh;          // row tm = 0
      const __bf16* bp1 = bp0 + (size_t)32 * LDP;                       // row tm = 1
      // B operands (the h1 planes in LDS) run ONE k-step ahead of their use, the weights RD steps
      bf16x8 bq[2][2][3];
      auto ldb = [&](int g, bf16x8 (&b)[2][3]) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          b[0][s] = *reinterpret_cast<const bf16x8*>(bp0 + (size_t)s * ROWS * LDP + 16 * g);
          b[1][s] = *reinterpret_cast<const bf16x8*>(bp1 + (size_t)s * ROWS * LDP + 16 * g);
        }
      };
      ldb(0, bq[0]);
#pragma unroll
      for (int g = 0; g < KS; ++g) {
        bf16x8 a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = ring[g % RD][s];
        if (g + RD < KS) {
#pragma unroll
          for (int s = 0; s < 3; ++s) ring[g % RD][s] = wp[(size_t)((g + RD) * 3 + s) * 64];
        }
        if (g + 1 < KS) ldb(g + 1, bq[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 (&b)[2][3] = bq[g & 1];
        // six products, smallest class first inside a k-step (only matters for NACC == 1)
        constexpr int c0 = 0, c1 = NACC == 3 ? 1 : 0, c2 = NACC == 3 ? 2 : 0;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          acc[tm][c2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tm][2], acc[tm][c2], 0, 0, 0);
          acc[tm][c2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[tm][0], acc[tm][c2], 0, 0, 0);
          acc[tm][c2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[tm][1], acc[tm][c2], 0, 0, 0);
          acc[tm][c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tm][1], acc[tm][c1], 0, 0, 0);
          acc[tm][c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[tm][0], acc[tm][c1], 0, 0, 0);
          acc[tm][c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tm][0], acc[tm][c0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // nothing of the next k-step is hoisted above this one
      }
    }
    // ---- "epilogue": combine the classes (smallest first), write or fold
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[tm][NACC - 1][r];
        for (int c = NACC - 2; c >= 0; --c) v += acc[tm][c][r];
        const int unit = 32 * wave + 8 * (r >> 2) + 4 * hh + (r & 3), row = 32 * tm + l31;
        if (write_out && t == 0) out[row * H + unit] = v;
        total += v;
      }
    }
    __syncthreads();
  }
  if (!write_out) out[(size_t)blockIdx.x * 512 + tid] = total;
}

template <int NACC, int RD, bool FP32>
double run(const float* h1, const bf16x8* Wsp, const float* W, float* out, int grid, int tiles,
           std::vector<float>* first) {
  const size_t smem = (size_t)3 * ROWS * LDP * 2;
  CHECK(hipFuncSetAttribute((const void*)tile_kernel<NACC, RD, FP32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  if (first) {
    hipLaunchKernelGGL((tile_kernel<NACC, RD, FP32>), dim3(1), dim3(512), smem, 0, h1, Wsp, W, out, 1, 1, 0);
    CHECK(hipDeviceSynchronize());
    first->resize(ROWS * H);
    CHECK(hipMemcpy(first->data(), out, ROWS * H * 4, hipMemcpyDeviceToHost));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {   // the first repetition brings the clocks up
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((tile_kernel<NACC, RD, FP32>), dim3(grid), dim3(512), smem, 0, h1, Wsp, W, out, tiles, 0, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

// ---------------------------------------------------------------------------------------------
// fp16x2 variant (round 6): every operand scaled by an exact power of two (per h1 row, per weight
// unit) into fp16's range, split into TWO fp16 terms (2 x 11 = 22 significand bits + the sign of
// lo: |x - hi - lo| <= 2^-24 |x|), three products per k-step  hi*hi | hi*lo + lo*hi  on
// v_mfma_f32_32x32x16_f16, the result unscaled with one ldexp.  Half the matrix instructions and
// two thirds of the planes of the bf16x3 form.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// exponent field e of max|x| (clamped) -> the row / unit is scaled by 2^(141 - e): max lands in [2^14, 2^15)
__host__ __device__ inline int scale_field(float mx) {
  unsigned b;
  memcpy(&b, &mx, 4);
  int e = (int)((b >> 23) & 0xff);
  if (e == 255) return 141;               // inf / NaN rows: no scaling, the non-finite value travels
  return e < 15 ? 15 : e;
}
__host__ __device__ inline float pow2_field(int f) {   // 2^(f - 127), f in [1, 254]
  unsigned b = (unsigned)f << 23;
  float r;
  memcpy(&r, &b, 4);
  return r;
}
__host__ __device__ inline void split2h(float xs, _Float16& hi, _Float16& lo) {
  hi = (_Float16)xs;
  lo = (_Float16)(xs - (float)hi);
}

__global__ void pack_wh_kernel(const float* __restrict__ W, f16x8* __restrict__ Wsp, int* __restrict__ wfield) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // (w, g, lane)
  if (e >= 8 * KS * 64) return;
  const int lane = e & 63, g = (e >> 6) % KS, w = e / (64 * KS);
  const int unit = 32 * w + (lane & 31), k0 = 16 * g + 8 * (lane >> 5);
  float mx = 0.f;
  for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(W[unit * K + k]));
  const int f = scale_field(mx);
  const float sc = pow2_field(268 - f);
  if (g == 0 && lane < 32) wfield[unit] = f;
  f16x8 p[2];
  for (int j = 0; j < 8; ++j) {
    _Float16 a, b;
    split2h(W[unit * K + k0 + j] * sc, a, b);
    p[0][j] = a; p[1][j] = b;
  }
  for (int s = 0; s < 2; ++s) Wsp[((size_t)(w * KS + g) * 2 + s) * 64 + lane] = p[s];
}

template <int NACC, int RD>
__global__ __launch_bounds__(512, 2) void tile_h_kernel(const float* __restrict__ h1,
                                                        const f16x8* __restrict__ Wsp,
                                                        const int* __restrict__ wfield,
                                                        float* __restrict__ out, int tiles,
                                                        int write_out, int zero) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* planes = reinterpret_cast<_Float16*>(smem_raw);   // [2][ROWS][LDP]
  __shared__ unsigned rowmax[ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  float total = 0.f;
  for (int t = 0; t < tiles; ++t) {
    if (tid < ROWS) rowmax[tid] = 0u;
    __syncthreads();
    float4 keep[ROWS * K / 4 / 512];
#pragma unroll
    for (int i = 0; i < ROWS * K / 4 / 512; ++i) {
      const int e = tid + 512 * i;
      const int r = e / (K / 4), c = (e % (K / 4)) * 4;
      keep[i] = *reinterpret_cast<const float4*>(h1 + r * K + c);
      const float m = fmaxf(fmaxf(fabsf(keep[i].x), fabsf(keep[i].y)), fmaxf(fabsf(keep[i].z), fabsf(keep[i].w)));
      atomicMax(&rowmax[r], __float_as_uint(m));     // non-negative floats order as unsigned
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ROWS * K / 4 / 512; ++i) {
      const int e = tid + 512 * i;
      const int r = e / (K / 4), c = (e % (K / 4)) * 4;
      const float sc = pow2_field(268 - scale_field(__uint_as_float(rowmax[r])));
      const float x[4] = {keep[i].x * sc, keep[i].y * sc, keep[i].z * sc, keep[i].w * sc};
      f16x4 q0, q1;
      for (int j = 0; j < 4; ++j) { _Float16 a, b; split2h(x[j], a, b); q0[j] = a; q1[j] = b; }
      *reinterpret_cast<f16x4*>(planes + ((size_t)0 * ROWS + r) * LDP + c) = q0;
      *reinterpret_cast<f16x4*>(planes + ((size_t)1 * ROWS + r) * LDP + c) = q1;
    }
    __syncthreads();
    f32x16 acc[2][NACC];
    for (int tm = 0; tm < 2; ++tm)
      for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[tm][c][r] = 0.f;
    f16x8 ring[RD][2];
    const f16x8* wp = Wsp + (size_t)wave * KS * 2 * 64 + lane + (size_t)zero * t;
#pragma unroll
    for (int g = 0; g < RD; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s) ring[g][s] = wp[(size_t)(g * 2 + s) * 64];
    const _Float16* bp0 = planes + (size_t)l31 * LDP + 8 * hh;
    const _Float16* bp1 = bp0 + (size_t)32 * LDP;
    f16x8 bq[2][2][2];
    auto ldb = [&](int g, f16x8 (&b)[2][2]) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        b[0][s] = *reinterpret_cast<const f16x8*>(bp0 + (size_t)s * ROWS * LDP + 16 * g);
        b[1][s] = *reinterpret_cast<const f16x8*>(bp1 + (size_t)s * ROWS * LDP + 16 * g);
      }
    };
    ldb(0, bq[0]);
#pragma unroll
    for (int g = 0; g < KS; ++g) {
      f16x8 a[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) a[s] = ring[g % RD][s];
      if (g + RD < KS) {
#pragma unroll
        for (int s = 0; s < 2; ++s) ring[g % RD][s] = wp[(size_t)((g + RD) * 2 + s) * 64];
      }
      if (g + 1 < KS) ldb(g + 1, bq[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      f16x8 (&b)[2][2] = bq[g & 1];
      constexpr int c0 = 0, c1 = NACC == 2 ? 1 : 0;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        acc[tm][c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[tm][1], acc[tm][c1], 0, 0, 0);
        acc[tm][c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[tm][0], acc[tm][c1], 0, 0, 0);
        acc[tm][c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[tm][0], acc[tm][c0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int row = 32 * tm + l31;
      const int rf = scale_field(__uint_as_float(rowmax[row]));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[tm][NACC - 1][r];
        if (NACC == 2) v += acc[tm][0][r];
        const int unit = 32 * wave + 8 * (r >> 2) + 4 * hh + (r & 3);
        v = ldexpf(v, rf + wfield[unit] - 282);
        if (write_out && t == 0) out[row * H + unit] = v;
        total += v;
      }
    }
    __syncthreads();
  }
  if (!write_out) out[(size_t)blockIdx.x * 512 + tid] = total;
}

template <int NACC, int RD>
double run_h(const float* h1, const f16x8* Wsp, const int* wfield, float* out, int grid, int tiles,
             std::vector<float>* first) {
  const size_t smem = (size_t)2 * ROWS * LDP * 2;
  CHECK(hipFuncSetAttribute((const void*)tile_h_kernel<NACC, RD>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  if (first) {
    hipLaunchKernelGGL((tile_h_kernel<NACC, RD>), dim3(1), dim3(512), smem, 0, h1, Wsp, wfield, out, 1, 1, 0);
    CHECK(hipDeviceSynchronize());
    first->resize(ROWS * H);
    CHECK(hipMemcpy(first->data(), out, ROWS * H * 4, hipMemcpyDeviceToHost));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((tile_h_kernel<NACC, RD>), dim3(grid), dim3(512), smem, 0, h1, Wsp, wfield, out, tiles, 0, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

int main() {
  std::vector<float> h1(ROWS * K), W(H * K);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : h1) { v = rnd() * 1.7f; v = v < 0 ? 0.f : v; }              // ReLU output
  for (auto& v : W) v = rnd() / 16.f;                                         // U(-1/16, 1/16)
  for (int i = 0; i < 64; ++i) h1[i * 5 + 3] *= 1e-3f;                        // mixed magnitudes
  std::vector<double> ref(ROWS * H);
  std::vector<double> mag(ROWS * H);
  for (int r = 0; r < ROWS; ++r)
    for (int u = 0; u < H; ++u) {
      double s = 0, m = 0;
      for (int k = 0; k < K; ++k) { s += (double)h1[r * K + k] * W[u * K + k]; m += fabs((double)h1[r * K + k] * W[u * K + k]); }
      ref[r * H + u] = s; mag[r * H + u] = m;
    }
  float *dh1, *dW, *dout;
  bf16x8* dWsp;
  CHECK(hipMalloc(&dh1, ROWS * K * 4));
  CHECK(hipMalloc(&dW, H * K * 4));
  CHECK(hipMalloc(&dWsp, (size_t)8 * KS * 3 * 64 * 16));
  CHECK(hipMalloc(&dout, (size_t)1024 * 512 * 4));
  CHECK(hipMemcpy(dh1, h1.data(), ROWS * K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW, W.data(), H * K * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pack_w_kernel, dim3((8 * KS * 64 + 255) / 256), dim3(256), 0, 0, dW, dWsp);
  CHECK(hipDeviceSynchronize());
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int tiles = 200;
  const double flop = 2.0 * ROWS * H * K;
  auto report = [&](const char* name, double sec, int grid, const std::vector<float>& got) {
    // error relative to sum |a b| (the scale fp32 accumulation errors live on) and to |result|
    double e_mag = 0, e_rel = 0;
    for (int i = 0; i < ROWS * H; ++i) {
      const double e = fabs((double)got[i] - ref[i]);
      if (e / mag[i] > e_mag) e_mag = e / mag[i];
      if (fabs(ref[i]) > 1e-2 && e / fabs(ref[i]) > e_rel) e_rel = e / fabs(ref[i]);
    }
    printf("{\"variant\": \"%s\", \"grid\": %d, \"tile_us\": %.3f, \"tflops_fp32_equiv\": %.1f, "
           "\"max_err_over_sum_abs\": %.3e, \"max_rel_err\": %.3e}\n",
           name, grid, sec / tiles * 1e6, flop * tiles * grid / sec / 1e12, e_mag, e_rel);
  };
  std::vector<float> got;
  double s;
  s = run<1, 4, true>(dh1, dWsp, dW, dout, ncu, 20, &got);
  {
    double sec = s;  // 20 tiles
    double e_mag = 0, e_rel = 0;
    for (int i = 0; i < ROWS * H; ++i) {
      const double e = fabs((double)got[i] - ref[i]);
      if (e / mag[i] > e_mag) e_mag = e / mag[i];
      if (fabs(ref[i]) > 1e-2 && e / fabs(ref[i]) > e_rel) e_rel = e / fabs(ref[i]);
    }
    printf("{\"variant\": \"fp32 mfma 32x32x2 (naive operand fetch: numerics reference only)\", "
           "\"tile_us\": %.3f, \"max_err_over_sum_abs\": %.3e, \"max_rel_err\": %.3e}\n",
           sec / 20 * 1e6, e_mag, e_rel);
  }
  s = run<3, 4, false>(dh1, dWsp, dW, dout, ncu, tiles, &got);      report("bf16x3 3acc rd4 1wg/cu", s, ncu, got);
  s = run<1, 4, false>(dh1, dWsp, dW, dout, ncu, tiles, &got);      report("bf16x3 1acc rd4 1wg/cu", s, ncu, got);
  s = run<3, 2, false>(dh1, dWsp, dW, dout, ncu, tiles, &got);      report("bf16x3 3acc rd2 1wg/cu", s, ncu, got);
  s = run<3, 4, false>(dh1, dWsp, dW, dout, 2 * ncu, tiles, &got);  report("bf16x3 3acc rd4 2 waves of wgs", s, 2 * ncu, got);
  s = run<3, 4, false>(dh1, dWsp, dW, dout, 192, tiles, &got);      report("bf16x3 3acc rd4 192 wgs", s, 192, got);
  {
    f16x8* dWh;
    int* dwf;
    CHECK(hipMalloc(&dWh, (size_t)8 * KS * 2 * 64 * 16));
    CHECK(hipMalloc(&dwf, H * 4));
    hipLaunchKernelGGL(pack_wh_kernel, dim3((8 * KS * 64 + 255) / 256), dim3(256), 0, 0, dW, dWh, dwf);
    CHECK(hipDeviceSynchronize());
    s = run_h<2, 4>(dh1, dWh, dwf, dout, ncu, tiles, &got);      report("fp16x2 2acc rd4 1wg/cu", s, ncu, got);
    s = run_h<1, 4>(dh1, dWh, dwf, dout, ncu, tiles, &got);      report("fp16x2 1acc rd4 1wg/cu", s, ncu, got);
    s = run_h<2, 2>(dh1, dWh, dwf, dout, ncu, tiles, &got);      report("fp16x2 2acc rd2 1wg/cu", s, ncu, got);
    s = run_h<2, 4>(dh1, dWh, dwf, dout, 2 * ncu, tiles, &got);  report("fp16x2 2acc rd4 2 wgs/cu", s, 2 * ncu, got);
    // the same data with rows / units scaled far outside fp16's range (exact powers of two):
    // the error relative to sum |a b| must not move
    std::vector<float> h1s(h1), Ws(W);
    for (int r = 0; r < ROWS; ++r) { const float f = ldexpf(1.f, (r % 9 - 4) * 15); for (int k = 0; k < K; ++k) h1s[r * K + k] *= f; }
    for (int u = 0; u < H; ++u) { const float f = ldexpf(1.f, (u % 7 - 3) * 12); for (int k = 0; k < K; ++k) Ws[u * K + k] *= f; }
    // one tiny and one moderately small element in every row (lo underflows into fp16 subnormals)
    for (int r = 0; r < ROWS; ++r) { h1s[r * K + 7] *= 1e-6f; h1s[r * K + 11] *= 3e-5f; }
    for (int r = 0; r < ROWS; ++r)
      for (int u = 0; u < H; ++u) {
        double s2 = 0, m = 0;
        for (int k = 0; k < K; ++k) { s2 += (double)h1s[r * K + k] * Ws[u * K + k]; m += fabs((double)h1s[r * K + k] * Ws[u * K + k]); }
        ref[r * H + u] = s2; mag[r * H + u] = m;
      }
    CHECK(hipMemcpy(dh1, h1s.data(), ROWS * K * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dW, Ws.data(), H * K * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_wh_kernel, dim3((8 * KS * 64 + 255) / 256), dim3(256), 0, 0, dW, dWh, dwf);
    hipLaunchKernelGGL(pack_w_kernel, dim3((8 * KS * 64 + 255) / 256), dim3(256), 0, 0, dW, dWsp);
    CHECK(hipDeviceSynchronize());
    s = run_h<2, 4>(dh1, dWh, dwf, dout, ncu, tiles, &got);      report("fp16x2 2acc rd4, rows/units scaled 2^+-60", s, ncu, got);
    s = run<3, 4, false>(dh1, dWsp, dW, dout, ncu, tiles, &got); report("bf16x3 3acc rd4, rows/units scaled 2^+-60", s, ncu, got);
  }
  return 0;
}
