// Device kernels of the DQN learner step (gfx950 / CDNA4, fp32 MFMA).
//
// What is computed (file:line under /root/reference):
//   online Q(s,a)            q_value_networks.py:152-174, common/utils.py:75-152
//   max_a' Q_target(s',a')   deep_q_learning.py:130-167, deep_td_learning.py:386-416
//   Bellman target + MSE     deep_td_learning.py:292-331
//   backward                 autograd of the above (deep_td_learning.py:353-354)
//   AdamW(amsgrad)           deep_td_learning.py:183-185 -> torch/optim/adam.py _single_tensor_adam
//   target soft update       common/utils.py:214-226
//
// All GEMM-shaped work uses v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma
// chain), the only MFMA precision that holds the 1e-5 relative Q-value
// tolerance of BASELINE.json.  Lane maps (cdna_hip_programming.md §3):
//   A operand: lane l holds A[i = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][j = l & 31]
//   C/D:       acc[reg] is C[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l & 31]
// K is consumed in groups of 8: lane half h = l>>5 owns k = 8*g + 4*h + j for
// the j-th MFMA of the group, so one ds_read_b128 feeds four MFMAs.
#pragma once
#include "common.hpp"

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int half) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}
__device__ __forceinline__ float relu_keep_nan(float v) { return (v < 0.f) ? 0.f : v; }

// base[col .. col+3] with zero fill outside [0, ncols) or when !row_ok.
// vec: caller proved 16-byte alignment of base + col for col % 4 == 0.
__device__ __forceinline__ float4 guarded_load4(const float* __restrict__ base, bool row_ok,
                                                int col, int ncols, bool vec) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return r;
  if (vec && col + 3 < ncols) return *reinterpret_cast<const float4*>(base + col);
  if (col < ncols) r.x = base[col];
  if (col + 1 < ncols) r.y = base[col + 1];
  if (col + 2 < ncols) r.z = base[col + 2];
  if (col + 3 < ncols) r.w = base[col + 3];
  return r;
}
__device__ __forceinline__ bool is_vec_ok(const float* p, int ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 3) == 0);
}
__device__ __forceinline__ float f4_get(const float4& v, int j) {
  return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------
// Generic LDS-staged linear layer, C[M,N] = epi(A[M,K] * op(B)), tile 32 x 64.
//   B_KS = false: B is [N][K] (a torch Linear weight; y = x W^T)
//   B_KS = true : B is [K][N] (the same weight used for dX = dY W)
// KW waves share each N half by splitting every 32-deep K chunk between them
// (latency, not throughput, bounds these 1024-row GEMMs); partial tiles are
// summed in a fixed order through LDS, so results are deterministic.
// ---------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2 };

struct GemmArgs {
  const float* A; int lda;
  const float* Bm; int ldb;
  float* C; int ldc;
  const float* bias;
  const float* Hmask; int ldh;
  int M, N, K;
};

constexpr int G_BM = 32, G_BN = 64, G_BK = 32, G_PK = 36 /* K pitch */, G_PN = 68 /* N pitch */;

template <bool B_KS, int EPI, int KW>
__global__ __launch_bounds__(128 * KW) void linear_kernel(GemmArgs g) {
  constexpr int NT = 128 * KW;
  constexpr int A_F4 = G_BM * G_BK / 4;  // 256
  constexpr int B_F4 = G_BN * G_BK / 4;  // 512
  constexpr int NPRE = (A_F4 + B_F4 + NT - 1) / NT;
  constexpr int B_TILE = B_KS ? (G_BK * G_PN) : (G_BN * G_PK);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][32][36]
  float* Bs = As + 2 * G_BM * G_PK;      // [2][B_TILE]
  float* Part = Bs + 2 * B_TILE;         // [KW-1][2][16*64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave & 1, kw = wave >> 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;
  const bool vecA = is_vec_ok(g.A, g.lda), vecB = is_vec_ok(g.Bm, g.ldb);
  const int NK = (g.K + G_BK - 1) / G_BK;

  float4 pre[NPRE];
  auto issue = [&](int kc) {
    const int k0 = kc * G_BK;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int f = tid + q * NT;
      if (f < A_F4) {
        const int r = f >> 3, c = f & 7;
        pre[q] = guarded_load4(g.A + (int64_t)(m0 + r) * g.lda, (m0 + r) < g.M, k0 + c * 4, g.K,
                               vecA);
      } else if (f < A_F4 + B_F4) {
        const int fb = f - A_F4;
        if (!B_KS) {
          const int n = fb >> 3, c = fb & 7;
          pre[q] = guarded_load4(g.Bm + (int64_t)(n0 + n) * g.ldb, (n0 + n) < g.N, k0 + c * 4,
                                 g.K, vecB);
        } else {
          const int kk = fb >> 4, c = fb & 15;
          pre[q] = guarded_load4(g.Bm + (int64_t)(k0 + kk) * g.ldb, (k0 + kk) < g.K, n0 + c * 4,
                                 g.N, vecB);
        }
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int f = tid + q * NT;
      if (f < A_F4) {
        const int r = f >> 3, c = f & 7;
        *reinterpret_cast<float4*>(As + buf * G_BM * G_PK + r * G_PK + c * 4) = pre[q];
      } else if (f < A_F4 + B_F4) {
        const int fb = f - A_F4;
        if (!B_KS) {
          const int n = fb >> 3, c = fb & 7;
          *reinterpret_cast<float4*>(Bs + buf * B_TILE + n * G_PK + c * 4) = pre[q];
        } else {
          const int kk = fb >> 4, c = fb & 15;
          *reinterpret_cast<float4*>(Bs + buf * B_TILE + kk * G_PN + c * 4) = pre[q];
        }
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  issue(0);
  commit(0);
  __syncthreads();
  for (int kc = 0; kc < NK; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < NK) issue(kc + 1);
    const float* as = As + buf * G_BM * G_PK;
    const float* bs = Bs + buf * B_TILE;
#pragma unroll
    for (int kgi = 0; kgi < 4 / KW; ++kgi) {
      const int kg = kgi * KW + kw;
      const float4 a4 = *reinterpret_cast<const float4*>(as + l31 * G_PK + kg * 8 + 4 * h);
      float b[4];
      if (!B_KS) {
        const float4 b4 =
            *reinterpret_cast<const float4*>(bs + (nt * 32 + l31) * G_PK + kg * 8 + 4 * h);
        b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bs[(kg * 8 + 4 * h + j) * G_PN + nt * 32 + l31];
      }
      acc = mfma32(a4.x, b[0], acc);
      acc = mfma32(a4.y, b[1], acc);
      acc = mfma32(a4.z, b[2], acc);
      acc = mfma32(a4.w, b[3], acc);
    }
    if (kc + 1 < NK) commit(buf ^ 1);
    __syncthreads();
  }
  if (KW > 1) {
    if (kw > 0) {
      float* p = Part + ((kw - 1) * 2 + nt) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kw == 0) {
#pragma unroll
      for (int s = 0; s < KW - 1; ++s) {
        const float* p = Part + (s * 2 + nt) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += p[r * 64 + lane];
      }
    }
  }
  if (kw == 0) {
    const int col = n0 + nt * 32 + l31;
    if (col < g.N) {
      float bv = 0.f;
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + acc_row(r, h);
        if (row < g.M) {
          float v = acc[r];
          if (EPI == EPI_BIAS) v += bv;
          if (EPI == EPI_BIAS_RELU) v = relu_keep_nan(v + bv);
          if (EPI == EPI_MASK) v = (g.Hmask[(int64_t)row * g.ldh + col] > 0.f) ? v : 0.f;
          g.C[(int64_t)row * g.ldc + col] = v;
        }
      }
    }
  }
}

template <bool B_KS, int KW>
constexpr size_t linear_smem_bytes() {
  return sizeof(float) * (2 * G_BM * G_PK + 2 * (B_KS ? (G_BK * G_PN) : (G_BN * G_PK)) +
                          (KW > 1 ? (KW - 1) * 2 * 1024 : 0));
}

// ---------------------------------------------------------------------------
// Fused target network: for every (transition b, available action i)
//   h1 = relu(U[b] + W1a' rep(b,i));  h2 = relu(W2' h1 + b2');  q' = w3' . h2 + b3'
// then mask, row max over i, and the Bellman target.  One workgroup owns
// bpw = floor(64 / A) whole transitions (<= 64 MFMA rows) and ALL H2 columns, so
// h1/h2 never leave LDS/registers: the 16384 x 256 x 256 layer-2 product — 79 % of
// a DQN step's FLOPs — runs out of one 64 x H1 LDS tile and a streamed W2'.
// 8 waves = 2 (rows) x 4 (columns); TN1/TN2 = 32-wide column tiles per wave in
// layer 1 / layer 2 (H1 <= 128*TN1, H2 <= 128*TN2).
// ---------------------------------------------------------------------------
struct TargetArgs {
  const float* U; int ldu;                  // [B][H1] = W1s' s' + b1'
  const float* feat; int64_t feat_bstride;  // rep(next_available_actions) [B][A][AD]
  const uint8_t* mask; int64_t mask_bstride;// [B][A], 1 = unavailable; may be null
  const float* W1a; int ldw1;               // W1' + S (action columns), row pitch S+AD
  const float* W2; int ldw2;                // [H2][H1]
  const float* b2; const float* w3; const float* b3;
  const float* reward; const uint8_t* term;
  float gamma;
  float* next_v; float* y;
  int B, A, AD, H1, H2, bpw;
};

constexpr int T_ROWS = 64, T_ADC = 16, T_ADP = 20;

template <int TN1, int TN2>
constexpr size_t target_smem_bytes() {
  return sizeof(float) * (T_ROWS * (128 * TN1 + 4) + 2 * (128 * TN2) * G_PK + 4 * 64 + 64);
}

template <int TN1, int TN2>
__global__ __launch_bounds__(512) void target_fused_kernel(TargetArgs a) {
  constexpr int H1P = 128 * TN1, H2P = 128 * TN2, PA_ = H1P + 4;
  constexpr int W2_F4 = H2P * 8;          // float4 per 32-deep chunk of W2'
  constexpr int NPRE = W2_F4 / 512;       // 2 * TN2
  static_assert((T_ROWS + H1P) * T_ADP <= 2 * H2P * G_PK, "layer-1 staging must fit in Bw");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ah = smem;                        // [64][H1P+4]   h1 tile (layer-2 A operand)
  float* Bw = Ah + T_ROWS * PA_;           // [2][H2P][36]  streamed W2' chunks
  float* qpart = Bw + 2 * H2P * G_PK;      // [4][64]
  float* qv = qpart + 4 * 64;              // [64]
  float* featS = Bw;                       // [64][20]   (aliases Bw during layer 1)
  float* W1aS = Bw + T_ROWS * T_ADP;       // [H1P][20]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int h = lane >> 5, l31 = lane & 31;
  const int b0 = blockIdx.x * a.bpw;
  const int nb = min(a.bpw, a.B - b0);
  const int nrows = nb * a.A;
  const bool vecW2 = is_vec_ok(a.W2, a.ldw2);
  const int NK = (a.H1 + G_BK - 1) / G_BK;

  float4 pre[NPRE];
  auto issue = [&](int kc) {
    const int k0 = kc * G_BK;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int f = tid + q * 512;
      const int n = f >> 3, c = f & 7;
      pre[q] = guarded_load4(a.W2 + (int64_t)n * a.ldw2, n < a.H2, k0 + c * 4, a.H1, vecW2);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int f = tid + q * 512;
      const int n = f >> 3, c = f & 7;
      *reinterpret_cast<float4*>(Bw + buf * H2P * G_PK + n * G_PK + c * 4) = pre[q];
    }
  };
  issue(0);  // in flight during layer 1

  // ---- layer 1: acc1 = U[b(row)] + rep(row) . W1a'^T
  f32x16 acc1[TN1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + acc_row(r, h);
    const bool rok = row < nrows;
    const int bb = rok ? (b0 + row / a.A) : 0;
#pragma unroll
    for (int t = 0; t < TN1; ++t) {
      const int col = (wn * TN1 + t) * 32 + l31;
      acc1[t][r] = (rok && col < a.H1) ? a.U[(int64_t)bb * a.ldu + col] : 0.f;
    }
  }
  for (int c0 = 0; c0 < a.AD; c0 += T_ADC) {
    for (int e = tid; e < T_ROWS * T_ADC; e += 512) {
      const int r = e >> 4, j = e & 15;
      float v = 0.f;
      if (r < nrows && c0 + j < a.AD) {
        const int bb = b0 + r / a.A, i = r % a.A;
        v = a.feat[(int64_t)bb * a.feat_bstride + (int64_t)i * a.AD + c0 + j];
      }
      featS[r * T_ADP + j] = v;
    }
    for (int e = tid; e < H1P * T_ADC; e += 512) {
      const int n = e >> 4, j = e & 15;
      float v = 0.f;
      if (n < a.H1 && c0 + j < a.AD) v = a.W1a[(int64_t)n * a.ldw1 + c0 + j];
      W1aS[n * T_ADP + j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kg = 0; kg < T_ADC / 8; ++kg) {
      const float4 a4 =
          *reinterpret_cast<const float4*>(featS + (wm * 32 + l31) * T_ADP + kg * 8 + 4 * h);
#pragma unroll
      for (int t = 0; t < TN1; ++t) {
        const float4 b4 = *reinterpret_cast<const float4*>(
            W1aS + ((wn * TN1 + t) * 32 + l31) * T_ADP + kg * 8 + 4 * h);
        acc1[t] = mfma32(a4.x, b4.x, acc1[t]);
        acc1[t] = mfma32(a4.y, b4.y, acc1[t]);
        acc1[t] = mfma32(a4.z, b4.z, acc1[t]);
        acc1[t] = mfma32(a4.w, b4.w, acc1[t]);
      }
    }
    __syncthreads();
  }
  // h1 = relu(acc1) -> LDS A tile (columns >= H1 and rows >= nrows are exact zeros)
#pragma unroll
  for (int t = 0; t < TN1; ++t) {
    const int col = (wn * TN1 + t) * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + acc_row(r, h);
      Ah[row * PA_ + col] = relu_keep_nan(acc1[t][r]);
    }
  }
  commit(0);  // staging region is dead: every wave passed the barrier above
  __syncthreads();

  // ---- layer 2: acc2 = h1 . W2'^T, W2' streamed in 32-deep chunks
  f32x16 acc2[TN2];
#pragma unroll
  for (int t = 0; t < TN2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

  for (int kc = 0; kc < NK; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < NK) issue(kc + 1);
    const float* as = Ah + (wm * 32 + l31) * PA_ + kc * G_BK + 4 * h;
    const float* bs = Bw + buf * H2P * G_PK + (wn * TN2 * 32 + l31) * G_PK + 4 * h;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const float4 a4 = *reinterpret_cast<const float4*>(as + kg * 8);
      float4 b4[TN2];
#pragma unroll
      for (int t = 0; t < TN2; ++t)
        b4[t] = *reinterpret_cast<const float4*>(bs + t * 32 * G_PK + kg * 8);
#pragma unroll
      for (int t = 0; t < TN2; ++t) acc2[t] = mfma32(a4.x, b4[t].x, acc2[t]);
#pragma unroll
      for (int t = 0; t < TN2; ++t) acc2[t] = mfma32(a4.y, b4[t].y, acc2[t]);
#pragma unroll
      for (int t = 0; t < TN2; ++t) acc2[t] = mfma32(a4.z, b4[t].z, acc2[t]);
#pragma unroll
      for (int t = 0; t < TN2; ++t) acc2[t] = mfma32(a4.w, b4[t].w, acc2[t]);
    }
    if (kc + 1 < NK) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- layer 3 + mask + max + Bellman target
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
#pragma unroll
  for (int t = 0; t < TN2; ++t) {
    const int col = (wn * TN2 + t) * 32 + l31;
    const bool cok = col < a.H2;
    const float bv = cok ? a.b2[col] : 0.f;
    const float wv = cok ? a.w3[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += relu_keep_nan(acc2[t][r] + bv) * wv;
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += __shfl_xor(v[r], off);
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) qpart[wn * 64 + wm * 32 + acc_row(r, h)] = v[r];
  }
  __syncthreads();
  if (tid < T_ROWS) {
    float q = ((qpart[tid] + qpart[64 + tid]) + qpart[128 + tid]) + qpart[192 + tid];
    q += a.b3[0];
    if (tid < nrows && a.mask) {
      const int bb = b0 + tid / a.A, i = tid % a.A;
      if (a.mask[(int64_t)bb * a.mask_bstride + i]) q = -INFINITY;
    }
    qv[tid] = q;
  }
  __syncthreads();
  if (tid < nb) {
    const int bb = b0 + tid;
    float m = qv[tid * a.A];
    for (int i = 1; i < a.A; ++i) {
      const float x = qv[tid * a.A + i];
      m = (x > m || x != x) ? x : m;
    }
    if (a.next_v) a.next_v[bb] = m;
    if (a.y) {
      // (next_v * gamma * (1 - terminated.float())) + reward, one rounding per op
      const float live = 1.0f - (a.term[bb] ? 1.0f : 0.0f);
      const float t0 = __fmul_rn(m, a.gamma);
      const float t1 = __fmul_rn(t0, live);
      a.y[bb] = __fadd_rn(t1, a.reward[bb]);
    }
  }
}

// ---------------------------------------------------------------------------
// Output head + loss + first backward stage (one wave per transition row):
//   q = w3 . h2 + b3;  d = q - y;  dq = (2 / (B * world)) * d
//   dZ2[b][n] = h2[b][n] > 0 ? dq * w3[n] : 0
// and per-workgroup slabs of dW3 = sum_b dq h2[b], db3 = sum_b dq, sum_b |d|.
// ---------------------------------------------------------------------------
struct HeadArgs {
  const float* H2a; int ldh;  // [B][H2] relu output of layer 2
  const float* w3; const float* b3;
  const float* y;             // Bellman target (may be null in probe mode)
  float* q_out;               // may be null
  float* dZ2; int ldz;        // may be null (probe mode)
  float* slab;                // [gridDim.x][H2 + 2]: dW3 | db3 | sum|d|
  float norm;                 // 2 / (B * world)
  int B, H2;
};
constexpr int HEAD_ROWS = 16;

__global__ __launch_bounds__(256) void head_loss_kernel(HeadArgs a) {
  __shared__ float dq_s[HEAD_ROWS];
  __shared__ float ad_s[HEAD_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * HEAD_ROWS;
  const float b3 = a.b3[0];
  for (int rr = wave; rr < HEAD_ROWS; rr += 4) {
    const int b = r0 + rr;
    float dq = 0.f, ad = 0.f;
    if (b < a.B) {
      const float* hrow = a.H2a + (int64_t)b * a.ldh;
      float p = 0.f;
      for (int n = lane; n < a.H2; n += 64) p = fmaf(hrow[n], a.w3[n], p);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off);
      const float q = p + b3;
      if (a.q_out && lane == 0) a.q_out[b] = q;
      if (a.y) {
        const float d = __fsub_rn(q, a.y[b]);
        dq = __fmul_rn(a.norm, d);
        ad = fabsf(d);
        if (a.dZ2) {
          float* zrow = a.dZ2 + (int64_t)b * a.ldz;
          for (int n = lane; n < a.H2; n += 64)
            zrow[n] = (hrow[n] > 0.f) ? __fmul_rn(dq, a.w3[n]) : 0.f;
        }
      }
    }
    if (lane == 0) {
      dq_s[rr] = dq;
      ad_s[rr] = ad;
    }
  }
  __syncthreads();
  if (!a.slab) return;
  float* slab = a.slab + (int64_t)blockIdx.x * (a.H2 + 2);
  for (int n = tid; n < a.H2; n += 256) {
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < HEAD_ROWS; ++rr) {
      const int b = r0 + rr;
      if (b < a.B) s = fmaf(dq_s[rr], a.H2a[(int64_t)b * a.ldh + n], s);
    }
    slab[n] = s;
  }
  if (tid == 0) {
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int rr = 0; rr < HEAD_ROWS; ++rr) {
      s += dq_s[rr];
      t += ad_s[rr];
    }
    slab[a.H2] = s;
    slab[a.H2 + 1] = t;
  }
}

// ---------------------------------------------------------------------------
// Weight gradients: dW[i][j] = sum_b dZ[b][i] X[b][j], db[i] = sum_b dZ[b][i].
// One 32 x 32 output tile per workgroup; its 8 waves split the batch (the K
// dimension) and add their partial tiles in a fixed order through LDS.  Both
// operands are row-contiguous across lanes, so they go global -> VGPR -> MFMA
// with no LDS staging.  The last block folds the head kernel's slabs into
// dW3 / db3 and the reported loss.
// ---------------------------------------------------------------------------
struct DwProblem {
  const float* dZ; int ldz;   // [B][M]
  const float* X; int ldx;    // [B][N]
  float* dW; int ldw;         // [M][N]
  float* db;                  // [M]
  int M, N, tiles_n, tile0;
};
struct DwArgs {
  DwProblem p[2];
  int B, total_tiles;
  const float* slab; int nslab; int H2;
  float* dW3; float* db3; float* loss_out; float inv_B;
};

__global__ __launch_bounds__(512) void weight_grad_kernel(DwArgs a) {
  __shared__ float part[8 * 1024];
  __shared__ float csum[8 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  if ((int)blockIdx.x >= a.total_tiles) {
    // slab fold (deterministic order)
    for (int n = tid; n < a.H2 + 2; n += 512) {
      float s = 0.f;
      for (int w = 0; w < a.nslab; ++w) s += a.slab[(int64_t)w * (a.H2 + 2) + n];
      if (n < a.H2) a.dW3[n] = s;
      else if (n == a.H2) a.db3[0] = s;
      else if (a.loss_out) a.loss_out[0] = s * a.inv_B;
    }
    return;
  }
  const DwProblem& P = ((int)blockIdx.x >= a.p[1].tile0 && a.p[1].tiles_n > 0) ? a.p[1] : a.p[0];
  const int t = blockIdx.x - P.tile0;
  const int i0 = (t / P.tiles_n) * 32, j0 = (t % P.tiles_n) * 32;
  const int per = ((a.B + 63) / 64) * 8;  // rows per wave, multiple of 8
  const int bs = wave * per;
  const int be = min(a.B, bs + per);
  const bool iok = (i0 + l31) < P.M, jok = (j0 + l31) < P.N;
  const float* zp = P.dZ + i0 + l31;
  const float* xp = P.X + j0 + l31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float cs = 0.f;
#pragma unroll 2
  for (int kb = bs; kb < be; kb += 8) {
    float av[4], xv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bb = kb + 4 * h + j;
      const bool ok = bb < be;
      av[j] = (ok && iok) ? zp[(int64_t)bb * P.ldz] : 0.f;
      xv[j] = (ok && jok) ? xp[(int64_t)bb * P.ldx] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc = mfma32(av[j], xv[j], acc);
      cs += av[j];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave * 1024 + r * 64 + lane] = acc[r];
  cs += __shfl_xor(cs, 32);
  if (h == 0) csum[wave * 32 + l31] = cs;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * 512;
    float s = part[e];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += part[w * 1024 + e];
    const int reg = e >> 6, ln = e & 63;
    const int row = i0 + acc_row(reg, ln >> 5), col = j0 + (ln & 31);
    if (row < P.M && col < P.N) P.dW[(int64_t)row * P.ldw + col] = s;
  }
  if (j0 == 0 && tid < 32 && (i0 + tid) < P.M) {
    float s = csum[tid];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += csum[w * 32 + tid];
    P.db[i0 + tid] = s;
  }
}

// ---------------------------------------------------------------------------
// AdamW(amsgrad=True), the op order of torch/optim/adam.py::_single_tensor_adam
// (param.mul_, exp_avg.lerp_, exp_avg_sq.mul_().addcmul_, maximum, sqrt/div/add,
// addcdiv_), one rounding per op as ATen's CPU kernels do.
// ---------------------------------------------------------------------------
struct AdamArgs {
  float* p; const float* g; float* m; float* v; float* vmax;
  int64_t n;
  float decay;       // 1 - lr * weight_decay
  float w1;          // 1 - beta1
  float beta2;
  float omb2;        // 1 - beta2
  float bc2_sqrt;    // sqrt(1 - beta2^t)
  float neg_step;    // -lr / (1 - beta1^t)
  float eps;
  int amsgrad;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float g = a.g[i];
  float p = __fmul_rn(a.p[i], a.decay);
  float m = a.m[i];
  m = __fadd_rn(m, __fmul_rn(a.w1, __fsub_rn(g, m)));
  float v = __fmul_rn(a.v[i], a.beta2);
  v = __fadd_rn(v, __fmul_rn(__fmul_rn(a.omb2, g), g));
  float dn;
  if (a.amsgrad) {
    float vm = a.vmax[i];
    vm = (v > vm || v != v) ? v : vm;
    a.vmax[i] = vm;
    dn = vm;
  } else {
    dn = v;
  }
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(dn), a.bc2_sqrt), a.eps);
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(a.neg_step, m), denom));
  a.p[i] = p;
  a.m[i] = m;
  a.v[i] = v;
}

// theta' <- tau * theta + (1 - tau) * theta'   (common/utils.py:214-226)
__global__ __launch_bounds__(256) void soft_update_kernel(float* __restrict__ tgt,
                                                          const float* __restrict__ src,
                                                          int64_t n, float tau,
                                                          float one_minus_tau) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  tgt[i] = __fadd_rn(__fmul_rn(tau, src[i]), __fmul_rn(one_minus_tau, tgt[i]));
}

// x[b] = state[b] || action_rep[b]   (q_value_networks.py:166-168 torch.cat)
__global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ state,
                                                     const float* __restrict__ arep,
                                                     float* __restrict__ x, int B, int S, int AD) {
  const int W = S + AD;
  const int64_t total = (int64_t)B * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / W;
    const int j = (int)(e - b * W);
    x[e] = (j < S) ? state[b * S + j] : arep[b * AD + (j - S)];
  }
}

}  // namespace pa
