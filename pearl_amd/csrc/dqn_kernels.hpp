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
#include <type_traits>

#include "common.hpp"

namespace pa {

// Phase stamps for kernel tuning: lane 0 of every wave stores the 100 MHz wall clock
// (s_memrealtime) at slot i.  `prof` is null outside tools/prof_chain.py.
#define PA_STAMP(prof, wg, wave, i)                                                     \
  do {                                                                                  \
    if ((prof) && (threadIdx.x & 63) == 0)                                              \
      (prof)[((int64_t)(wg) * 8 + (wave)) * 16 + (i)] = (long long)wall_clock64();      \
  } while (0)

// Same slot layout, but the SHADER clock (s_memtime): with the wall-clock stamps of the same two
// points it gives the effective shader frequency of the launch (tools/prof_chain.py).
#define PA_STAMP_CYC(prof, wg, wave, i)                                                 \
  do {                                                                                  \
    if ((prof) && (threadIdx.x & 63) == 0)                                              \
      (prof)[((int64_t)(wg) * 8 + (wave)) * 16 + (i)] = (long long)clock64();           \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int half) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}
__device__ __forceinline__ float relu_keep_nan(float v) { return (v < 0.f) ? 0.f : v; }

// Guarded loads without branches: raw buffer loads through a wave-uniform resource descriptor;
// an out-of-range byte offset makes the hardware return 0.  (hipcc turns "cond ? load : 0" — even
// a clamped-address load followed by a select — into an exec-masked branch per load and waits for
// each in turn, which serialises what should be one batch of outstanding loads;
// cdna_hip_programming.md §5 trap (c), §5.5 T8.)  `p` must be wave-uniform (a kernel argument);
// byte offsets must stay below 2 GiB.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kBufOob = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, 0x7FFFFFF0, 0x00020000);
}
// Descriptor that covers exactly `bytes`: any byte offset at or beyond it (carried in the VGPR
// offset; the hardware does not range-check the scalar offset) reads as zero.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc_n(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ld_or_zero(const float* __restrict__ p, int64_t off, bool ok) {
  const unsigned bo = ok ? (unsigned)off * 4u : kBufOob;
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(buf_rsrc(p), (int)bo, 0, 0));
}
__device__ __forceinline__ float4 ld4_or_zero(const float* __restrict__ p, int64_t off, bool ok) {
  const unsigned bo = ok ? (unsigned)off * 4u : kBufOob;
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(p), (int)bo, 0, 0);
  // whole-vector bit_cast: element-wise __builtin_bit_cast(float, v.x) is narrowed by hipcc 7.2
  // to ONE dword load splatted over all four lanes (observed in the ISA)
  const f32x4_t f = __builtin_bit_cast(f32x4_t, v);
  return make_float4(f[0], f[1], f[2], f[3]);
}
// base[col .. col+3] of one row with zero fill outside [0, ncols) or when !row_ok; no alignment
// or multiple-of-4 assumption (four dword loads).
__device__ __forceinline__ float4 guarded_load4(const float* __restrict__ p, int64_t row_off,
                                                bool row_ok, int col, int ncols) {
  float4 r;
  r.x = ld_or_zero(p, row_off + col, row_ok && col < ncols);
  r.y = ld_or_zero(p, row_off + col + 1, row_ok && col + 1 < ncols);
  r.z = ld_or_zero(p, row_off + col + 2, row_ok && col + 2 < ncols);
  r.w = ld_or_zero(p, row_off + col + 3, row_ok && col + 3 < ncols);
  return r;
}
__device__ __forceinline__ bool is_vec_ok(const float* p, int ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 3) == 0);
}
__device__ __forceinline__ float f4_get(const float4& v, int j) {
  return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------
// LDS-staged linear layer, C[M,N] = epi(A[M,K] * op(B)), tile 32 x 64, up to two
// independent problems per launch (blockIdx.z).
//   B_KS = false: B is [N][K] (a torch Linear weight; y = x W^T)
//   B_KS = true : B is [K][N] (the same weight used for dX = dY W)
// These are 1024-row GEMMs of 0.07-0.27 GFLOP: latency, not throughput, bounds
// them.  So a workgroup issues EVERY global load of its (32 + 64) x K operand
// panel up front (K <= 256 per pass; one exposed memory latency instead of one
// per K chunk), stages it once in LDS, and KW waves per N half split the K range;
// partial tiles are summed in a fixed order through LDS (deterministic).
// ---------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_NONE = 3 };

struct GemmArgs {
  const float* A; int lda;
  const float* Bm; int ldb;
  float* C; int ldc;
  const float* bias;
  const float* Hmask; int ldh;
  int M, N, K;
  int epi;
};
struct LinArgs {
  GemmArgs p[2];
};

constexpr int G_BM = 32, G_BN = 64, G_BK = 32, G_PK = 36 /* K pitch of 32-deep chunks */,
              G_PN = 68 /* N pitch */, G_KCAP = 256 /* K per staging pass */;

__host__ __device__ inline int lin_kpad(int K) {
  const int k = K < G_KCAP ? K : G_KCAP;
  return (k + 31) & ~31;
}

template <bool B_KS, int KW>
static __global__ __launch_bounds__(128 * KW) void linear_kernel(LinArgs args) {
  constexpr int NT = 128 * KW;
  constexpr int SLOTS = (G_BM + G_BN) * (G_KCAP / 4);  // float4 slots of one full panel
  constexpr int NPRE = (SLOTS + NT - 1) / NT;
  constexpr int A_SLOTS = G_BM * (G_KCAP / 4);
  const GemmArgs& g = args.p[blockIdx.z];
  const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;
  if (m0 >= g.M || n0 >= g.N) return;  // grid covers the larger of the two problems
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KP = lin_kpad(g.K);        // staged K (multiple of 32)
  const int P = KP + 4;                // K pitch (== 4 mod 32: conflict-free b128 reads)
  float* As = smem;                                   // [32][P]
  float* Bs = As + G_BM * P;                          // KC: [64][P]   KS: [KP][68]
  float* Part = Bs + (B_KS ? KP * G_PN : G_BN * P);   // [KW-1][2][16*64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave & 1, kw = wave >> 1;
  const int h = lane >> 5, l31 = lane & 31;
  // wave-uniform: whole float4s are in or out of range, so the load phase needs no branches
  const bool fastA = is_vec_ok(g.A, g.lda) && ((g.K & 3) == 0);
  const bool fastB = is_vec_ok(g.Bm, g.ldb) && (((B_KS ? g.N : g.K) & 3) == 0);
  // epilogue operands are fetched now, not after the MFMA chain
  const int ecol = n0 + nt * 32 + l31;
  float bv = 0.f;
  if (g.epi == EPI_BIAS || g.epi == EPI_BIAS_RELU) bv = ld_or_zero(g.bias, ecol, ecol < g.N);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int kb = 0; kb < g.K; kb += G_KCAP) {
    const int klen = min(G_KCAP, g.K - kb);
    const int q4 = ((klen + 31) & ~31) >> 2;  // float4 per staged row
    float4 pre[NPRE];
    if (fastA && fastB) {
      // every float4 is entirely inside or entirely outside the operand: clamp + select
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const int f = tid + q * NT;
        if (f < A_SLOTS) {
          const int r = f >> 6, c = f & 63;
          pre[q] = ld4_or_zero(g.A, (int64_t)(m0 + r) * g.lda + kb + c * 4,
                               (m0 + r) < g.M && c * 4 < klen);
        } else {
          const int fb = f - A_SLOTS;
          if (!B_KS) {
            const int n = fb >> 6, c = fb & 63;
            pre[q] = ld4_or_zero(g.Bm, (int64_t)(n0 + n) * g.ldb + kb + c * 4,
                                 f < SLOTS && (n0 + n) < g.N && c * 4 < klen);
          } else {
            const int kk = fb >> 4, c = fb & 15;
            pre[q] = ld4_or_zero(g.Bm, (int64_t)(kb + kk) * g.ldb + n0 + c * 4,
                                 f < SLOTS && kk < klen && (n0 + c * 4) < g.N);
          }
        }
      }
    } else {
      // generic shapes (unaligned rows, K or N not a multiple of 4): element-wise guards
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const int f = tid + q * NT;
        pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < A_SLOTS) {
          const int r = f >> 6, c = f & 63;
          if (c < q4)
            pre[q] = guarded_load4(g.A, (int64_t)(m0 + r) * g.lda + kb, (m0 + r) < g.M, c * 4, klen);
        } else if (f < SLOTS) {
          const int fb = f - A_SLOTS;
          if (!B_KS) {
            const int n = fb >> 6, c = fb & 63;
            if (c < q4)
              pre[q] = guarded_load4(g.Bm, (int64_t)(n0 + n) * g.ldb + kb, (n0 + n) < g.N, c * 4,
                                     klen);
          } else {
            const int kk = fb >> 4, c = fb & 15;
            if (kk < q4 * 4)
              pre[q] = guarded_load4(g.Bm, (int64_t)(kb + kk) * g.ldb + n0, kk < klen, c * 4,
                                     g.N - n0);
          }
        }
      }
    }
    if (kb > 0) __syncthreads();  // previous pass is done reading the panel
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int f = tid + q * NT;
      if (f < A_SLOTS) {
        const int r = f >> 6, c = f & 63;
        if (c < q4) *reinterpret_cast<float4*>(As + r * P + c * 4) = pre[q];
      } else if (f < SLOTS) {
        const int fb = f - A_SLOTS;
        if (!B_KS) {
          const int n = fb >> 6, c = fb & 63;
          if (c < q4) *reinterpret_cast<float4*>(Bs + n * P + c * 4) = pre[q];
        } else {
          const int kk = fb >> 4, c = fb & 15;
          if (kk < q4 * 4) *reinterpret_cast<float4*>(Bs + kk * G_PN + c * 4) = pre[q];
        }
      }
    }
    __syncthreads();
    const int nkg = q4 >> 1;  // groups of 8 k
    const float* ap = As + l31 * P + 4 * h;
#pragma unroll 2
    for (int kg = kw; kg < nkg; kg += KW) {
      const float4 a4 = *reinterpret_cast<const float4*>(ap + kg * 8);
      float b[4];
      if (!B_KS) {
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + (nt * 32 + l31) * P + kg * 8 + 4 * h);
        b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[(kg * 8 + 4 * h + j) * G_PN + nt * 32 + l31];
      }
      acc = mfma32(a4.x, b[0], acc);
      acc = mfma32(a4.y, b[1], acc);
      acc = mfma32(a4.z, b[2], acc);
      acc = mfma32(a4.w, b[3], acc);
    }
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
    const int col = ecol;
    float hm[16];
    if (g.epi == EPI_MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + acc_row(r, h);
        hm[r] = ld_or_zero(g.Hmask, (int64_t)row * g.ldh + col, row < g.M && col < g.N);
      }
    }
    if (col < g.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + acc_row(r, h);
        if (row < g.M) {
          float v = acc[r];
          if (g.epi == EPI_BIAS) v += bv;
          else if (g.epi == EPI_BIAS_RELU) v = relu_keep_nan(v + bv);
          else if (g.epi == EPI_MASK) v = (hm[r] > 0.f) ? v : 0.f;
          g.C[(int64_t)row * g.ldc + col] = v;
        }
      }
    }
  }
}

template <bool B_KS, int KW>
inline size_t linear_smem_bytes(int K) {
  const int KP = lin_kpad(K), P = KP + 4;
  return sizeof(float) * ((size_t)G_BM * P + (B_KS ? (size_t)KP * G_PN : (size_t)G_BN * P) +
                          (KW > 1 ? (KW - 1) * 2 * 1024 : 0));
}

// ---------------------------------------------------------------------------
// Fused target network: for every (transition b, available action i)
//   h1 = relu(U[b] + W1a' rep(b,i));  h2 = relu(W2' h1 + b2');  q' = w3' . h2 + b3'
// then mask, row max over i, and the Bellman target.  One workgroup owns
// bpw = floor(64 / A) whole transitions (<= 64 MFMA rows) and ALL H2 columns, so
// h1/h2 never leave LDS/registers: the 16384 x 256 x 256 layer-2 product — 79 % of
// a DQN step's FLOPs — runs out of one 64 x H1 LDS tile.
//
// Wave w of the 8 owns hidden units [32w, 32w+32) of h1 and of h2 for all 64 rows (two
// 32 x 32 accumulators).  Its weight operand, W2'[32 units][all k], is read from a
// FRAGMENT-MAJOR copy of W2' (W2f[tile][kgroup][lane] = the float4 that lane feeds
// to the four MFMAs of that k-group): one fully coalesced 1 KiB buffer load per
// k-group straight into VGPRs — no LDS staging of weights and NO barrier in the
// main loop, so the eight waves run decoupled and the matrix pipes stay fed.
// (PMC on the LDS-staged version: 53 % MFMA busy, 33 % of wave time in
// s_waitcnt/s_barrier.)  W2f is refreshed whenever the target net changes
// (repack_w2_kernel, or in place by the fused soft update of adamw_kernel).
// ---------------------------------------------------------------------------
struct TargetArgs {
  const float* U; int ldu;                  // [B][H1] = W1s' s' + b1'
  const float* feat; int64_t feat_bstride;  // rep(next_available_actions) [B][A][AD]
  const uint8_t* mask; int64_t mask_bstride;// [B][A], 1 = unavailable; may be null
  const float* W1a; int ldw1;               // W1' + S (action columns), row pitch S+AD
  const float* W2f;                         // fragment-major W2' (see w2f_index)
  const void* W2sp;                         // optional: W2' as bf16 split planes (w2sp_index):
                                            // target_split_kernel instead of the fp32-MFMA kernels
  const void* W2h; const int* w2hf;         // optional: W2' as scaled fp16 hi / lo planes + the scale field of
                                            // every unit (target_h2_kernel.hpp): target_h2_kernel
  // optional (target_split_kernel only): the first-layer state product computed in the tile —
  // U[b] = W1s' s'[b] + b1' as a bf16x3 product of the tile's distinct states — instead of read
  // from `U`: no first-layer GEMM launch in front of the target pass, no [rows][H1] round trip
  const void* W1sp;                         // W1'[:, :S] as split planes (wsp_index, ks = S / 16)
  const float* next_state; int ld_next;     // s' [B][S]
  const float* b1;                          // b1' [H1]
  int S;                                    // state width (multiple of 16, <= 256) when W1sp is set
  const float* b2; const float* w3; const float* b3;
  const float* reward; const uint8_t* term;
  float gamma;
  float* next_v; float* y;
  int B, A, AD, H1, H2, bpw;
  // persistent mode (learn() with the overlapped online chain): workgroups pull 64-row tiles from
  // a counter, and workgroups that land on a compute unit reserved for the chain exit at once
  int* tile_ctr;             // null: classic grid, tile = blockIdx.x
  int ntiles;
  const uint8_t* reserved;   // [kCuKeys] 1 = this CU belongs to the online chain; may be null
  long long* prof;           // optional phase stamps [tile][wave][16] (tools/prof_chain.py)
  int* argmax;               // optional [B]: index of the FIRST row maximum (torch.max(1)[1];
                             // Double DQN's action choice, double_dqn.py:47)
  float* choice_rep;         // with argmax: [B][AD] = feat row of that action (double_dqn.py:48-51)
  float* q_all;              // optional [B * A]: every (transition, action) value before masking
                             // (TwinCritic.get_q_values on an action set, discrete SAC)
  int rows_hint;             // 32: this pass prefers the 32-row, four-wave tile (two workgroups per CU:
                             // stand-alone passes — Double DQN, all-actions values); 0: the 64-row tile
  int* dbg_workers;          // optional: += 1 per workgroup of a persistent launch that takes at least one tile
  int prio_tiles;            // classic grid: tiles below this index run at raised wave priority (the
                             // first round of a window, whose targets the online chain waits for,
                             // shares every CU with a later round's tile)
};

// (XCC_ID, HW_ID.se_id|sh_id|cu_id) of the compute unit the calling wave runs on
constexpr int kCuKeys = 4096;
__device__ __forceinline__ unsigned cu_key() {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  return ((xcc & 0xFu) << 8) | ((hw >> 8) & 0xFFu);
}
// One record per workgroup; the spin keeps the slot busy so the dispatcher spreads the grid over
// every CU.  Host side: cu_partition() in dqn.hip.
static __global__ __launch_bounds__(512) void cu_census_kernel(unsigned* keys, int spin) {
  if (threadIdx.x == 0) keys[blockIdx.x] = cu_key();
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

constexpr int T_ROWS = 64;

// Bellman targets travel from the target-network stream to the online chain of learn() as
// data-tagged 4-byte granules (MI355X_MICROARCH.md, hand-off recipe R2): the buffer is pre-filled
// with kYPendingBits (a NaN payload no arithmetic produces), the producer publishes each value with
// ONE write-through (agent-scope) store, the consumer polls with L1-bypassing loads until the word
// differs.  No flag, no fence: the value is its own tag.
constexpr unsigned kYPendingBits = 0xFFC0DE5Au;
__device__ __forceinline__ void publish_y(float* p, float v) {
  unsigned bits = __builtin_bit_cast(unsigned, v);
  if (bits == kYPendingBits) bits ^= 1u;  // still a NaN; keeps the tag unambiguous
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), bits, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// spins: ~0.6 us each (about a second in all; the longest legitimate wait seen is 3 ms, a first-use
// code-object load on the side stream); the bound exists so that a broken producer cannot hang the GPU
constexpr int kYPollSpins = 1 << 21;
// `err` is a device word (read on the polling path: host memory there costs a PCIe round trip per
// row pass — measured: 43 -> 51 us per round); a failure is ALSO written to `err_host`, a pinned
// device-mapped host word, so that the call needs no device-to-host copy to report it.
__device__ __forceinline__ float consume_y(const float* p, int* err, int* err_host = nullptr) {
  const unsigned* q = reinterpret_cast<const unsigned*>(p);
  unsigned bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (bits == kYPendingBits) {
    const int limit = (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                          ? 0 : kYPollSpins;
    int spins = 0;
    while (bits == kYPendingBits && spins < limit) {
      __builtin_amdgcn_s_sleep(16);
      bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++spins;
    }
    if (bits == kYPendingBits) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (err_host) __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  return __builtin_bit_cast(float, bits);
}
constexpr int T_MAXH = 256;  // hidden widths the kernel is built for (8 waves x 32 columns)

// k-groups of layer 2, padded to the kernel instantiations (8 / 16 / 32 <-> H1 <= 64 / 128 / 256)
__host__ __device__ inline int t_nkg(int H1) { return H1 <= 64 ? 8 : (H1 <= 128 ? 16 : 32); }
// float index of W2'[n][k] inside the fragment-major copy; nkg = t_nkg(H1)
__host__ __device__ inline int64_t w2f_index(int n, int k, int nkg) {
  const int t = n >> 5, l31 = n & 31, g = k >> 3, h = (k >> 2) & 1, j = k & 3;
  return ((((int64_t)t * nkg + g) * 64) + h * 32 + l31) * 4 + j;
}
__host__ __device__ inline int64_t w2f_floats(int H2, int H1) {
  return (int64_t)((H2 + 31) / 32) * t_nkg(H1) * 256;
}

// ---- the target W2 as bf16 split planes (target_split_kernel.hpp: the layer-2 product on the bf16
// matrix pipe at fp32 accuracy) ------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// LDS access through a pointer the compiler can no longer trace back to the __shared__ array (tile
// buffers picked by a run-time layer index: the generic engine's row kernels).  Such an access is
// compiled as a FLAT instruction, which counts on vmcnt AND lgkmcnt: every wait for an LDS operand
// then also waits for every global load in flight — the weight ring of the GEMM loop is drained
// in front of each k-step (s_waitcnt vmcnt(0) lgkmcnt(0) ahead of every MFMA group in the ISA of
// mlp_rowstep_kernel up to round 6).  The explicit address-space cast makes it ds_read / ds_write.
// (concrete typedefs: hipcc drops the attribute from a dependent type in a template)
// Every global access of this wave has completed (s_waitcnt vmcnt(0) as an instruction the
// compiler's wait-count pass SEES, unlike inline asm).  Used in front of a weight-ring loop whose
// wave still has global STORES in flight (the previous layer's kept activations): gfx9 counts loads
// and stores on one counter and may complete them out of order with respect to each other, so with
// a store pending anywhere on the way into the loop hipcc turns the loop's first partial wait of
// every trip into vmcnt(0) — the ring drained once per trip.  A workgroup barrier does not wait for
// stores on this target; after it they have all but landed, and the loop's waits become partial.
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }
typedef __attribute__((address_space(3))) f32x4_t lds_f32x4_t;
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8_t;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
typedef __attribute__((address_space(3))) __bf16 lds_bf16_t;
__device__ __forceinline__ float4 lds_ld4(const float* p) {
  const f32x4_t v = *(const lds_f32x4_t*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_st4(float* p, const float4& v) {
  const f32x4_t q = {v.x, v.y, v.z, v.w};
  *(lds_f32x4_t*)p = q;
}
__device__ __forceinline__ float lds_ld(const float* p) { return *(const lds_f32_t*)p; }
__device__ __forceinline__ void lds_st(float* p, float v) { *(lds_f32_t*)p = v; }
__device__ __forceinline__ bf16x8 lds_ld_bf16x8(const __bf16* p) { return *(const lds_bf16x8_t*)p; }
__device__ __forceinline__ void lds_st_bf16x4(__bf16* p, const bf16x4& v) { *(lds_bf16x4_t*)p = v; }
__device__ __forceinline__ void lds_st_bf16(__bf16* p, __bf16 v) { *(lds_bf16_t*)p = v; }

constexpr int TS_H = 256;            // H1 = H2 = 256 (the FAST shape of target_tile)
constexpr int TS_KS = TS_H / 16;     // k-steps of v_mfma_f32_32x32x16_bf16
constexpr int TS_LDP = TS_H + 8;     // bf16 pitch of a plane row: 528 B = 4 dwords mod 64 banks
constexpr int TS_RD = 4;             // k-steps of weights in flight per wave

// Split planes of a weight matrix W[n][k] with ks = K / 16 k-steps (8 waves x 32 units = 256 rows):
//   slot ((wave * ks + kstep) * 3 + s) * 64 + lane holds the 8 bf16 that lane feeds one MFMA as its
//   A operand: unit n = 32 wave + (lane & 31), k = 16 kstep + 8 (lane >> 5) + e
__host__ __device__ inline int64_t wsp_bytes(int ks) { return (int64_t)8 * ks * 3 * 64 * 16; }
__host__ __device__ inline int64_t wsp_index(int n, int k, int s, int ks) {
  const int w = n >> 5, lane = (n & 31) + 32 * ((k >> 3) & 1), g = k >> 4, e = k & 7;
  return ((((int64_t)(w * ks + g) * 3 + s) * 64) + lane) * 8 + e;
}
__host__ __device__ inline int64_t w2sp_bytes() { return wsp_bytes(TS_KS); }
__host__ __device__ inline int64_t w2sp_index(int n, int k, int s) { return wsp_index(n, k, s, TS_KS); }

__host__ __device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;                        // round to nearest even
  const float r = x - (float)hi;         // exact
  mid = (__bf16)r;
  lo = (__bf16)(r - (float)mid);         // exact difference; the last conversion is exact too
}

// four consecutive k (k % 4 == 0) of one unit: one 8-byte store per plane
__device__ __forceinline__ void store_wsp4(void* base, int n, int k, const float4& v, int ks) {
  __bf16* p = static_cast<__bf16*>(base);
  const float x[4] = {v.x, v.y, v.z, v.w};
  bf16x4 q[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 a, b, c;
    split3(x[j], a, b, c);
    q[0][j] = a; q[1][j] = b; q[2][j] = c;
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) *reinterpret_cast<bf16x4*>(p + wsp_index(n, k, s, ks)) = q[s];
}
__device__ __forceinline__ void store_wsp1(void* base, int n, int k, float v, int ks) {
  __bf16* p = static_cast<__bf16*>(base);
  __bf16 a, b, c;
  split3(v, a, b, c);
  p[wsp_index(n, k, 0, ks)] = a;
  p[wsp_index(n, k, 1, ks)] = b;
  p[wsp_index(n, k, 2, ks)] = c;
}
// Split planes of a weight matrix W[n][k] as the A operand of v_mfma_f32_16x16x32_bf16, for the
// generic MLP engine's fused row step (mlp_rowstep.hpp): tiles of 16 units, k-steps of 32;
//   slot ((tile * nks + kstep) * 3 + plane) * 64 + lane,  lane = 16 ((k >> 3) & 3) + (n & 15),
// holds the 8 bf16  k = 32 kstep + 8 (lane >> 4) + e  of unit n = 16 tile + (lane & 15): one
// coalesced 1 KiB load per (tile, k-step, plane).  Zero beyond N / K.
__host__ __device__ inline int wsp16_nks(int K) { return (K + 31) >> 5; }
__host__ __device__ inline int64_t wsp16_bytes(int N, int K) {
  return (int64_t)((N + 15) >> 4) * wsp16_nks(K) * 3 * 64 * 16;
}
__host__ __device__ inline int64_t wsp16_index(int n, int k, int plane, int nks) {
  return (((((int64_t)(n >> 4) * nks + (k >> 5)) * 3 + plane) * 64) + ((k >> 3) & 3) * 16 + (n & 15)) * 8 +
         (k & 7);
}
// four consecutive k (k % 4 == 0) of one unit: one 8-byte store per plane
__device__ __forceinline__ void store_wsp16_4(void* base, int n, int k, const float4& v, int nks) {
  __bf16* p = static_cast<__bf16*>(base);
  const float x[4] = {v.x, v.y, v.z, v.w};
  bf16x4 q[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 a, b, c;
    split3(x[j], a, b, c);
    q[0][j] = a; q[1][j] = b; q[2][j] = c;
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) *reinterpret_cast<bf16x4*>(p + wsp16_index(n, k, s, nks)) = q[s];
}
__device__ __forceinline__ void store_wsp16_1(void* base, int n, int k, float v, int nks) {
  __bf16* p = static_cast<__bf16*>(base);
  __bf16 a, b, c;
  split3(v, a, b, c);
  p[wsp16_index(n, k, 0, nks)] = a;
  p[wsp16_index(n, k, 1, nks)] = b;
  p[wsp16_index(n, k, 2, nks)] = c;
}
__device__ __forceinline__ void store_w2sp4(void* base, int n, int k, const float4& v) {
  store_wsp4(base, n, k, v, TS_KS);
}
__device__ __forceinline__ void store_w2sp1(void* base, int n, int k, float v) {
  store_wsp1(base, n, k, v, TS_KS);
}

static __global__ __launch_bounds__(256) void repack_w2_kernel(const float* __restrict__ W2, int H2,
                                                        int H1, float* __restrict__ W2f,
                                                        void* W2sp = nullptr) {
  const int nkg = t_nkg(H1);
  const int64_t total = w2f_floats(H2, H1) / 4;  // float4 slots
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    const int64_t tg = e >> 6;
    const int g = (int)(tg % nkg), t = (int)(tg / nkg);
    const int n = t * 32 + (lane & 31), k = g * 8 + 4 * (lane >> 5);
    float4 v;
    v.x = (n < H2 && k < H1) ? W2[(int64_t)n * H1 + k] : 0.f;
    v.y = (n < H2 && k + 1 < H1) ? W2[(int64_t)n * H1 + k + 1] : 0.f;
    v.z = (n < H2 && k + 2 < H1) ? W2[(int64_t)n * H1 + k + 2] : 0.f;
    v.w = (n < H2 && k + 3 < H1) ? W2[(int64_t)n * H1 + k + 3] : 0.f;
    reinterpret_cast<float4*>(W2f)[e] = v;
    // the bf16 split planes of the same matrix (target_split_kernel), whole 256 x 256 matrices only
    if (W2sp && H1 == TS_H && H2 == TS_H) store_w2sp4(W2sp, n, k, v);
  }
}

// two matrices of one shape in one launch (blockIdx.y): the twin critics' all-actions passes
static __global__ __launch_bounds__(256) void repack_w2_pair_kernel(const float* __restrict__ W2a,
                                                             const float* __restrict__ W2b, int H2,
                                                             int H1, float* __restrict__ W2fa,
                                                             float* __restrict__ W2fb, void* spa,
                                                             void* spb) {
  const float* W2 = blockIdx.y ? W2b : W2a;
  float* W2f = blockIdx.y ? W2fb : W2fa;
  void* W2sp = blockIdx.y ? spb : spa;
  const int nkg = t_nkg(H1);
  const int64_t total = w2f_floats(H2, H1) / 4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    const int64_t tg = e >> 6;
    const int g = (int)(tg % nkg), t = (int)(tg / nkg);
    const int n = t * 32 + (lane & 31), k = g * 8 + 4 * (lane >> 5);
    float4 v;
    v.x = (n < H2 && k < H1) ? W2[(int64_t)n * H1 + k] : 0.f;
    v.y = (n < H2 && k + 1 < H1) ? W2[(int64_t)n * H1 + k + 1] : 0.f;
    v.z = (n < H2 && k + 2 < H1) ? W2[(int64_t)n * H1 + k + 2] : 0.f;
    v.w = (n < H2 && k + 3 < H1) ? W2[(int64_t)n * H1 + k + 3] : 0.f;
    reinterpret_cast<float4*>(W2f)[e] = v;
    if (W2sp && H1 == TS_H && H2 == TS_H) store_w2sp4(W2sp, n, k, v);
  }
}

inline size_t target_smem_bytes(int H1) {
  const int H1P = t_nkg(H1) * 8;
  return sizeof(float) * ((size_t)T_ROWS * (H1P + 4) + 8 * 64 + 64);
}

// Waves per SIMD the kernel is compiled for: 4 (= two co-resident 8-wave workgroups per CU, <= 128
// VGPRs) for the widest instantiation, so that one workgroup's prologue / epilogue overlaps the
// other's MFMA stream when a launch covers many 64-row tiles (a window of learn() rounds).
// FAST: the host has checked that every operand is 16-byte aligned with pitches that are multiples
// of 4, AD <= 16, H1 = 8 NKG and H2 = 256 (all eight waves own hidden units): the generic guards
// fold away.  A prologue / epilogue instruction is serial time on this machine.
template <int NKG, bool FAST>
__device__ __forceinline__ void target_tile(const TargetArgs& a, int tile, float* smem) {
  constexpr int H1P = NKG * 8;           // padded layer-2 K (64 / 128 / 256)
  constexpr int PA_ = H1P + 4;           // == 4 mod 32: conflict-free b128 reads AND writes by row
  constexpr int RD = 8;                  // W2' fragment prefetch ring depth (k-groups in flight)
  float* Ah = smem;                      // [64][H1P+4]   h1 tile (layer-2 B operand)
  float* qpart = Ah + T_ROWS * PA_;      // [8][64]
  float* qv = qpart + 8 * 64;            // [64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int b0 = tile * a.bpw;
  const int nb = min(a.bpw, a.B - b0);
  const int nrows = nb * a.A;
  const int nt1 = H1P >> 5, nt2 = (a.H2 + 31) >> 5;  // 32-wide tiles of h1 (padded) / h2 columns
  const bool l1 = FAST || wave < nt1, l2 = FAST || wave < nt2;

  // TRANSPOSED tiles: the weights are the MFMA A operand (i = hidden unit n), the activations the
  // B operand (j = batch row), so an accumulator register holds
  //   C[n = 32*wave + acc_row(reg, h)][row = 32*tm + l31]
  // i.e. a lane owns ONE batch row and 16 hidden units: the layer-1 output goes to LDS as four
  // ds_write_b128 per tile, and layer 3 (a dot product over hidden units) is an in-lane fma chain
  // instead of a cross-lane reduction.
  const int nq0 = wave * 32 + 4 * h;     // this lane's hidden units: nq0 + 8*q + j, q,j in 0..3
  PA_STAMP(a.prof, tile, wave, 0);
  // scalars of the epilogue, requested with the first operands: fetched where they are used they
  // are one more exposed global round trip each at the end of every tile
  const float b3v = a.b3[0];
  unsigned pf_mask = 0, pf_term = 0;
  float pf_reward = 0.f;
  if (tid < nrows && a.mask)
    pf_mask = a.mask[(int64_t)(b0 + tid / a.A) * a.mask_bstride + tid % a.A];
  if (tid < nb && a.y) {
    pf_term = a.term[b0 + tid];
    pf_reward = a.reward[b0 + tid];
  }

  // ---- layer 1 operands first (vmcnt retires in order: layer 1 never waits for the W2' stream)
  f32x16 acc[2];
  const bool vU = FAST || (((reinterpret_cast<uintptr_t>(a.U) & 15) == 0) && ((a.ldu & 3) == 0));
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
      if (vU) u = ld4_or_zero(a.U, (int64_t)bb * a.ldu + n, rok && n < a.H1);  // H1 % 4 == 0 here
      else u = guarded_load4(a.U, (int64_t)bb * a.ldu, rok, n, a.H1);
      acc[tm][4 * q + 0] = u.x; acc[tm][4 * q + 1] = u.y;
      acc[tm][4 * q + 2] = u.z; acc[tm][4 * q + 3] = u.w;
    }
  }
  const bool vfeat = FAST || (((reinterpret_cast<uintptr_t>(a.feat) & 15) == 0) &&
                              ((a.AD & 3) == 0) && ((a.feat_bstride & 3) == 0));
  const bool vw1 = FAST || (((reinterpret_cast<uintptr_t>(a.W1a) & 15) == 0) &&
                            ((a.ldw1 & 3) == 0) && ((a.AD & 3) == 0));
  const int wcol = wave * 32 + l31;      // hidden unit this lane feeds as the A operand
  const int64_t woff = (int64_t)wcol * a.ldw1;
  const bool wok = l1 && wcol < a.H1;
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
  float4 fx[2][2], fw[2];  // the first two k-groups (AD <= 16 covers one-hot over 16 actions)
  l1_loads(0, fx[0], fw[0]);
  l1_loads(8, fx[1], fw[1]);

  // ---- layer-2 A fragments: W2f[wave][g][lane], one coalesced 1 KiB load per k-group, RD
  // k-groups in flight
  float4 ring[RD];
  const int64_t wbase = ((int64_t)wave * NKG) * 256 + lane * 4;
#pragma unroll
  for (int g = 0; g < RD; ++g) ring[g] = ld4_or_zero(a.W2f, wbase + (int64_t)g * 256, l2);
  // layer-3 constants of this lane's hidden units: fetched while the last RD k-groups run (they
  // take over the registers of the drained prefetch ring)
  float4 b2v[4], w3v[4];
  const bool v2 = FAST || (((reinterpret_cast<uintptr_t>(a.b2) & 15) == 0) &&
                           ((reinterpret_cast<uintptr_t>(a.w3) & 15) == 0) && ((a.H2 & 3) == 0));
  auto l3_load = [&](const float* p, int q) {
    const int n = nq0 + 8 * q;
    return v2 ? ld4_or_zero(p, n, n < a.H2) : guarded_load4(p, 0, true, n, a.H2);
  };

  PA_STAMP(a.prof, tile, wave, 1);
  l1_mfma(fx[0], fw[0]);
  l1_mfma(fx[1], fw[1]);
  if constexpr (!FAST) {
    for (int k0 = 16; k0 < a.AD; k0 += 8) {  // wider action representations (rare)
      float4 x4[2], w4;
      l1_loads(k0, x4, w4);
      l1_mfma(x4, w4);
    }
  }
  if (l1) {
    // h1 = relu(acc) -> LDS tile [row][k] (hidden units >= H1 and rows >= nrows are exact zeros)
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
  PA_STAMP(a.prof, tile, wave, 2);
  __syncthreads();  // the only workgroup barrier before the epilogue
  PA_STAMP(a.prof, tile, wave, 3);

  // ---- layer 2: acc[n][row] = sum_k W2'[n][k] h1[row][k]
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
      if (g + RD < NKG) ring[g % RD] = ld4_or_zero(a.W2f, wbase + (int64_t)(g + RD) * 256, true);
      else if (g + RD - NKG < 4) b2v[g + RD - NKG] = l3_load(a.b2, g + RD - NKG);
      else w3v[g + RD - NKG - 4] = l3_load(a.w3, g + RD - NKG - 4);
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
    }
  }

  PA_STAMP(a.prof, tile, wave, 4);
  if (!l2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = w3v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- layer 3: in-lane over this lane's 16 hidden units, then the other half, then the waves
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

template <int NKG, bool FAST>
static __global__ __launch_bounds__(512, 4) void target_fused_kernel(TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (a.tile_ctr == nullptr) {
    if ((int)blockIdx.x < a.prio_tiles) __builtin_amdgcn_s_setprio(3);
    target_tile<NKG, FAST>(a, blockIdx.x, smem);
    return;
  }
  __shared__ int next_tile;
  if (a.reserved && a.reserved[cu_key()]) return;
  // (Tried and measured without effect on this kernel: delaying every second workgroup of a CU by
  // half a tile, and s_setprio 3 outside the MFMA main loop.)
  // The index of the NEXT tile is requested while the current one is computed (a returning
  // device-scope atomic costs 1-3 us under load; exposed, it would stretch every tile).
  if (threadIdx.x == 0) next_tile = atomicAdd(a.tile_ctr, 1);
  __syncthreads();
  int tile = next_tile;
  while (tile < a.ntiles) {
    int ahead = 0;
    if (threadIdx.x == 0) ahead = atomicAdd(a.tile_ctr, 1);
    target_tile<NKG, FAST>(a, tile, smem);
    if (threadIdx.x == 0) next_tile = ahead;
    __syncthreads();  // publishes next_tile; LDS is reused by the next tile
    tile = next_tile;
  }
}

// ---------------------------------------------------------------------------
// Output head + loss + first backward stage, one wave per transition row:
//   q = w3 . h2 + b3;  d = q - y;  dq = (2 / (B * world)) * d
//   dZ2[b][n] = h2[b][n] > 0 ? dq * w3[n] : 0
// dq[b] and |d|[b] go to small vectors; their batch reductions ride the weight
// gradient kernel (dW3 = dq^T h2 is its third problem) and the AdamW kernel.
// ---------------------------------------------------------------------------
struct HeadArgs {
  const float* H2a; int ldh;  // [B][H2] relu output of layer 2
  const float* w3; const float* b3;
  const float* y;             // Bellman target (null in probe mode)
  float* q_out;               // may be null
  float* dq_out;              // [B] (null in probe mode)
  float* absd_out;            // [B]
  float* dZ2; int ldz;        // may be null (probe mode)
  float norm;                 // 2 / (B * world)
  int B, H2;
};

static __global__ __launch_bounds__(256) void head_loss_kernel(HeadArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const bool fast = is_vec_ok(a.H2a, a.ldh) && is_vec_ok(a.w3, 4) && ((a.H2 & 3) == 0);
  float p = 0.f;
  // H2 <= 256 = 64 lanes x 4 columns
  const int c = lane * 4;
  float4 hv, wv;
  if (fast) {
    hv = ld4_or_zero(a.H2a, (int64_t)b * a.ldh + c, c < a.H2);
    wv = ld4_or_zero(a.w3, c, c < a.H2);
  } else {
    hv = guarded_load4(a.H2a, (int64_t)b * a.ldh, true, c, a.H2);
    wv = guarded_load4(a.w3, 0, true, c, a.H2);
  }
  p = fmaf(hv.x, wv.x, p); p = fmaf(hv.y, wv.y, p); p = fmaf(hv.z, wv.z, p); p = fmaf(hv.w, wv.w, p);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off);
  const float q = p + a.b3[0];
  if (a.q_out && lane == 0) a.q_out[b] = q;
  if (!a.y) return;
  const float d = __fsub_rn(q, a.y[b]);
  const float dq = __fmul_rn(a.norm, d);
  if (lane == 0) {
    a.dq_out[b] = dq;
    a.absd_out[b] = fabsf(d);
  }
  if (a.dZ2) {
    float* zrow = a.dZ2 + (int64_t)b * a.ldz;
    float4 z;
    z.x = (hv.x > 0.f) ? __fmul_rn(dq, wv.x) : 0.f;
    z.y = (hv.y > 0.f) ? __fmul_rn(dq, wv.y) : 0.f;
    z.z = (hv.z > 0.f) ? __fmul_rn(dq, wv.z) : 0.f;
    z.w = (hv.w > 0.f) ? __fmul_rn(dq, wv.w) : 0.f;
    if (is_vec_ok(a.dZ2, a.ldz) && c + 3 < a.H2) {
      *reinterpret_cast<float4*>(zrow + c) = z;
    } else {
      if (c < a.H2) zrow[c] = z.x;
      if (c + 1 < a.H2) zrow[c + 1] = z.y;
      if (c + 2 < a.H2) zrow[c + 2] = z.z;
      if (c + 3 < a.H2) zrow[c + 3] = z.w;
    }
  }
}

// ---- per-row maxima of the online weights (the scales of online_f16_kernel.hpp) -----------------
// umax buffer: [H1] rows of W1 | [H2] rows of W2, bit patterns of max |w| (unsigned order = float order)
constexpr int HF_UNITS = 256;
constexpr int HF_UMAX = 2 * HF_UNITS;
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned umaxu(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umax4(const float4& v) {
  return umaxu(umaxu(abs_bits(v.x), abs_bits(v.y)), umaxu(abs_bits(v.z), abs_bits(v.w)));
}
__device__ __forceinline__ void umax_atomic(unsigned* p, unsigned m) {
  (void)__hip_atomic_fetch_max(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// From the row-major parameters, one wave per row (any grid: wave w0 of nw cooperating waves).
// `out`: the buffer the next row pass reads; `zero`: the other buffer of the pair, cleared for the
// atomic maxima of the next optimizer launch.
__device__ __forceinline__ void unit_max_body(const float* __restrict__ q, int64_t off_w1, int64_t off_w2,
                                              int IN, int H1, int H2, unsigned* __restrict__ out,
                                              unsigned* __restrict__ zero, int64_t w0, int64_t nw, int lane) {
  for (int64_t unit = w0; unit < H1 + H2; unit += nw) {
    const float* W = unit < H1 ? q + off_w1 + unit * IN : q + off_w2 + (unit - H1) * H1;
    const int K = unit < H1 ? IN : H1;
    unsigned m = 0u;
    for (int k = lane; k < K; k += 64) m = umaxu(m, abs_bits(W[k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = umaxu(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) {
      out[unit] = m;
      if (zero) zero[unit] = 0u;
    }
  }
}

// AdamW(amsgrad=True), the op order of torch/optim/adam.py::_single_tensor_adam
// (param.mul_, exp_avg.lerp_, exp_avg_sq.mul_().addcmul_, maximum, sqrt/div/add,
// addcdiv_), one rounding per op as ATen's CPU kernels do.
struct AdamScalars {
  float decay;       // 1 - lr * weight_decay
  float w1;          // 1 - beta1
  float beta2;
  float omb2;        // 1 - beta2
  float bc2_sqrt;    // sqrt(1 - beta2^t)
  float neg_step;    // -lr / (1 - beta1^t)
  float eps;
  int amsgrad;
};
struct AdamState {
  float* p; float* m; float* v; float* vmax;
};
// The same update on values already in registers (vm is ignored unless amsgrad).
__device__ __forceinline__ void adam_math(const AdamScalars& c, float g, float& p, float& m,
                                          float& v, float& vm) {
  p = __fmul_rn(p, c.decay);
  m = __fadd_rn(m, __fmul_rn(c.w1, __fsub_rn(g, m)));
  v = __fmul_rn(v, c.beta2);
  v = __fadd_rn(v, __fmul_rn(__fmul_rn(c.omb2, g), g));
  float dn = v;
  if (c.amsgrad) {
    vm = (v > vm || v != v) ? v : vm;
    dn = vm;
  }
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(dn), c.bc2_sqrt), c.eps);
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(c.neg_step, m), denom));
}
__device__ __forceinline__ float adam_update(const AdamScalars& c, const AdamState& st, int64_t i,
                                             float g) {
  float p = __fmul_rn(st.p[i], c.decay);
  float m = st.m[i];
  m = __fadd_rn(m, __fmul_rn(c.w1, __fsub_rn(g, m)));
  float v = __fmul_rn(st.v[i], c.beta2);
  v = __fadd_rn(v, __fmul_rn(__fmul_rn(c.omb2, g), g));
  float dn;
  if (c.amsgrad) {
    float vm = st.vmax[i];
    vm = (v > vm || v != v) ? v : vm;
    st.vmax[i] = vm;
    dn = vm;
  } else {
    dn = v;
  }
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(dn), c.bc2_sqrt), c.eps);
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(c.neg_step, m), denom));
  st.p[i] = p;
  st.m[i] = m;
  st.v[i] = v;
  return p;
}

// Fused optimizer tail of the weight-gradient kernel (single-GPU learn path): the workgroup that
// finishes a 32 x 32 tile of dW applies AdamW to those parameters at once, refreshes the MFMA
// fragment-major copies the next step's kernels read, and — when the next step's forward() opens
// with a soft target update (deep_td_learning.py:283-284) — performs that update too.
struct AdamFuse {
  int enabled;
  AdamScalars c;
  AdamState st;
  const float* grad_base;            // flat gradient buffer (parameter index = dW - grad_base)
  float* W1f; float* W2f; float* W2tf;  // online packed copies (16x16x4 fragment-major)
  int nkg_w1, nkg_w2, nkg_w2t;
  int soft_next; float* tgt; float tau, one_minus_tau;
  float* tW2f; int nkg_t;            // target W2, 32x32x2 fragment-major
  void* tW2sp;                       // target W2 as bf16 split planes (null: not kept)
  void* tW1sp; int sp_S;             // target W1[:, :sp_S] as split planes (null: not kept)
  float* oW2f; void* oW2sp;          // Double DQN: the ONLINE W2 in the target tile's layouts (w2f_index,
                                     // split planes) for the argmax pass; null: not kept
  const float* absd; int nabs; float inv_B; float* loss_out;  // mean |Q - target| of this step
  // overlapped learn loop: the Bellman targets of this round have been consumed by the row pass
  // (the previous launch on this stream); the loss workgroup hands the words back to the producer
  // by restoring the pending tag (see consume_y / publish_y)
  unsigned* y_restore; int n_restore;
  // optional device word: non-zero = an in-launch hand-off upstream of these gradients expired
  // (sac_rows.hpp helper workgroups), the operands hold pending tags -> form the gradients but do
  // NOT step the optimizer or touch the target network with them
  const int* guard;
  // online_f16_kernel.hpp's row scales: the new max |w| of every row of W1 / W2 goes (atomic max) into
  // umax_acc — the buffer the NEXT row pass reads, zero on entry — and umax_clear, the buffer the last
  // row pass read, is zeroed for the launch after this one.  null: not kept.
  unsigned* umax_acc; unsigned* umax_clear;
};

struct DwProblem {
  const float* dZ; int ldz;   // [B][M]
  int dz_pair;                // 1: dZ holds TWO partials of the operand, interleaved per pair of units —
                              // [B][M / 2]{a(2p), a(2p + 1), b(2p), b(2p + 1)}, ldz = the physical pitch (2 M) —
                              // and the operand is a + b, added as it is loaded: ONE 16-byte vector per two
                              // units (online_rowpass_pair_kernel's two halves of dZ1).  Honoured by the
                              // *_pair kernels only (run_weight_grad, dqn.hip); M even, 16-byte aligned rows
  const float* X; int ldx;    // [B][N]
  float* dW; int ldw;         // [M][N]
  float* db;                  // [M]
  int M, N, tiles_n, tile0;
  int kind;                   // 0: W2, 1: W1, 2: w3 (which packed copies a parameter feeds);
                              // 3: a layer of the generic engine, packed copies below
  float* pkf; int nkgf;       // kind 3: fragment-major W [M units][N]  (wf16 layout) or null
  float* pktf; int nkgtf;     // kind 3: fragment-major W^T [N units][M] or null
  float* pkf_t;               // kind 3 + soft update: the TARGET's fragment-major W (pkf layout) or null
  void* pks; int nks;         // kind 3: W as bf16x3 split planes for v_mfma_f32_16x16x32_bf16 (wsp16_index) or null
  void* pkts; int nkts;       // kind 3: W^T the same way ([N units][M]) or null
  int bias_frozen;            // the bias slot is not a parameter (bias-free layer): no AdamW on db
  int net;                    // kind 3: which network's optimizer state (0: ad, 1: net2)
  int raw;                    // a plain X^T dZ product riding an optimizer launch: dW / db stored, no AdamW
                              // (the bandit's LinUCB moment update next to its network's gradients)
  // kind 3, fp16x2 row kernels of the generic engine (sac_rows.hpp, H2 instantiations): max |w| per ROW
  // of this layer after the step, as AdamFuse::umax_acc / umax_clear keep it for the DQN networks —
  // atomic max into um_acc (zero on entry: the buffer the NEXT row launch reads), um_clear (the buffer
  // the last one read) zeroed for the launch after; *_t: the same for the soft-updated target.  M
  // entries each; null: not kept.  Host contract: every element of such a problem is on the float4
  // epilogue path (N % 4 == 0, aligned rows).
  unsigned* um_acc; unsigned* um_clear;
  unsigned* um_acc_t; unsigned* um_clear_t;
};
// A second network in the same launch (twin critics: one launch instead of two half-empty ones).
// Same AdamW hyper-parameters and step as `ad` (one optimizer), its own flat buffers.
struct DwNet2 {
  AdamState st;
  const float* grad_base;
  float* tgt;
};
// Optional extra workgroup of a weight-gradient launch: end-of-step scalar work that would
// otherwise be a launch of its own.  kind 1 = continuous SAC's step tail (sac_rows.hpp): the two
// losses from per-tile partial sums, in tile order, and the entropy-coefficient AdamW step
// (soft_actor_critic_continuous.py:134-151) with alpha_kernel's arithmetic.
struct TailJob {
  int kind;
  const float* part_a; const float* part_b; int tiles; int B;
  float* actor_loss; float* critic_loss;
  float* log_alpha; float* am; float* av; float* avmax; float* alpha;
  const float* logp; float target_entropy; AdamScalars ac; float* alpha_loss_out;
};
constexpr int DW_MAX_PROB = 6;
struct DwArgs {
  DwProblem p[DW_MAX_PROB];
  int nprob, B, total_tiles;
  AdamFuse ad;
  DwNet2 net2;
  TailJob tail;
  long long* prof;   // optional phase stamps (tools/prof_chain.py): [workgroup][wave][16]
  // Large batches: `ksplit` workgroups share a tile, each reducing its slice of the batch; they
  // leave their partial tile in `kscratch` (write-through stores) and take a ticket, the last one
  // adds the partials in slice order (deterministic) and runs the epilogue.  ksplit = 1: off.
  int ksplit;
  // rows (output units) per tile: 0 / 64 = DW_TM, or 32 — the same kernel with two units per lane
  // instead of four: twice the workgroups for launches that have the CUs for them (the DQN chain
  // once the target pass needs fewer: 105 workgroups, 3.5 us of MFMA each instead of 7)
  int tm;
  // 1: weight_grad_split_kernel — the main loop on the bf16 matrix pipe at fp32 accuracy (every
  // operand split exactly three ways, six products, dw_mainloop_split below): launches whose tiles
  // are MFMA-bound (thousands of batch rows per workgroup).  The host sets it only when every
  // matrix problem of the launch has 16-byte-aligned dZ rows and 8-byte-aligned X rows.
  int split;
  float* kscratch;       // [total_tiles][ksplit][DW_TM * DW_TN + DW_TM]
  unsigned* ktickets;    // [total_tiles], zero between launches
  // 1: consecutive tiles (the column tiles of one block of units, which read the same dZ panel) run
  // on ONE XCD — workgroup b goes to XCD b mod 8, so tile = (b mod 8) * (tiles / 8) + b / 8.  Set by
  // the host for launches whose operand panels are re-read from memory (thousands of batch rows)
  // and whose tile count is a multiple of 8.
  int xcd_order;
};

// index of fragment-major slots (defined here, used by online_kernels.hpp as well)
__host__ __device__ inline int64_t wf16_index_(int unit, int k, int nkg) {
  return ((((int64_t)(unit >> 4) * nkg + (k >> 4)) * 64) + ((k >> 2) & 3) * 16 + (unit & 15)) * 4 +
         (k & 3);
}

__device__ __forceinline__ float adam_fused_weight(const AdamFuse& f, int kind, int64_t i, int row,
                                                   int col, float g) {
  const float p = adam_update(f.c, f.st, i, g);
  if (kind == 0) {         // W2[n = row][k = col]
    f.W2f[wf16_index_(row, col, f.nkg_w2)] = p;
    f.W2tf[wf16_index_(col, row, f.nkg_w2t)] = p;
    if (f.oW2f) {
      f.oW2f[w2f_index(row, col, f.nkg_t)] = p;
      if (f.oW2sp) store_w2sp1(f.oW2sp, row, col, p);
    }
  } else if (kind == 1) {  // W1[n = row][k = col]
    f.W1f[wf16_index_(row, col, f.nkg_w1)] = p;
  }
  if (f.soft_next) {  // update_target_network (common/utils.py:214-226)
    const float t = __fadd_rn(__fmul_rn(f.tau, p), __fmul_rn(f.one_minus_tau, f.tgt[i]));
    f.tgt[i] = t;
    if (kind == 0) {
      f.tW2f[w2f_index(row, col, f.nkg_t)] = t;
      if (f.tW2sp) store_w2sp1(f.tW2sp, row, col, t);
    } else if (kind == 1 && f.tW1sp && col < f.sp_S) {
      store_wsp1(f.tW1sp, row, col, t, f.sp_S >> 4);
    }
  }
  return p;
}
// kind 3: refresh the generic engine's fragment-major copies of one weight element
__device__ __forceinline__ void pack_generic(const DwProblem& P, int row, int col, float p) {
  if (P.pkf) P.pkf[wf16_index_(row, col, P.nkgf)] = p;
  if (P.pktf) P.pktf[wf16_index_(col, row, P.nkgtf)] = p;
  if (P.pks) store_wsp16_1(P.pks, row, col, p, P.nks);
  if (P.pkts) store_wsp16_1(P.pkts, col, row, p, P.nkts);
}
// kind 3, one scalar parameter: AdamW, packed copies, and (soft) the target with its packed copy
__device__ __forceinline__ void adam_generic_weight(const AdamFuse& f, const AdamState& st, float* tgt,
                                                    const DwProblem& P, int64_t i, int row, int col,
                                                    float g) {
  const float p = adam_update(f.c, st, i, g);
  pack_generic(P, row, col, p);
  if (f.soft_next && tgt) {
    const float t = __fadd_rn(__fmul_rn(f.tau, p), __fmul_rn(f.one_minus_tau, tgt[i]));
    tgt[i] = t;
    if (P.pkf_t) P.pkf_t[wf16_index_(row, col, P.nkgf)] = t;
  }
}
__device__ __forceinline__ void adam_generic_bias(const AdamFuse& f, const AdamState& st, float* tgt,
                                                  int64_t i, float g) {
  const float p = adam_update(f.c, st, i, g);
  if (f.soft_next && tgt)
    tgt[i] = __fadd_rn(__fmul_rn(f.tau, p), __fmul_rn(f.one_minus_tau, tgt[i]));
}
__device__ __forceinline__ void adam_fused_bias(const AdamFuse& f, int64_t i, float g) {
  const float p = adam_update(f.c, f.st, i, g);
  if (f.soft_next)
    f.tgt[i] = __fadd_rn(__fmul_rn(f.tau, p), __fmul_rn(f.one_minus_tau, f.tgt[i]));
}

// ---------------------------------------------------------------------------
// Weight gradients: dW[i][j] = sum_b dZ[b][i] X[b][j], db[i] = sum_b dZ[b][i], for
// up to three problems per launch (dW2/db2, dW1/db1, dW3/db3 with dZ = dq[B][1]).
//
// One 64 x 32 output tile per workgroup, v_mfma_f32_16x16x4_f32; its 8 waves split the batch
// (the K dimension) and add their partial tiles in a fixed order through LDS.  Both operands are
// row-contiguous across lanes and go global -> VGPR -> MFMA with no LDS staging:
//   lane (c = lane & 15, q = lane >> 4), step s (batch rows b0 .. b0 + 3):
//     a4 = dZ[b0 + q][i0 + 4c .. 4c + 3]   one 16-byte load, A operand of 4 MFMAs (unit 4c + ja)
//     x2 = X [b0 + q][j0 + 2c .. 2c + 1]   one  8-byte load, B operand of 2 MFMAs (col 2c + jx)
//     acc[ja][jx] += A_ja (16 units x 4 rows) * B_jx (4 rows x 16 cols)         8 MFMAs
// i.e. 256 MFMA cycles per pair of wide loads.  (The first version fed 32x32x2 MFMAs from dword
// loads, one per operand per MFMA: the CU's address pipe, ~15 cycles per wave-load whatever its
// width, was the bound — 6.4 us of issue per workgroup against 3.4 us of MFMA.)  Loads run
// RING steps ahead of their use through a register ring with static indices.  The buffer
// descriptors cover exactly B rows, so rows past the batch read as zero by range check, and the
// column guards are loop-invariant lane offsets.  One extra workgroup (blockIdx == total_tiles)
// folds |Q - target| into the reported loss when ad.loss_out is set.
// ---------------------------------------------------------------------------
constexpr int DW_TM = 64, DW_TN = 32, DW_RING = 8;
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));

template <int UPL, bool PAIR = false>
struct DwFrag {
  float a[UPL];   // UPL consecutive units of one batch row (one 16- or 8-byte load)
  float b[PAIR ? UPL : 1];   // PAIR: the same vector of the second partial (added at use)
  float2 x;
};

// Byte offsets: a lane whose vector lies outside the operand carries kBufOob (2^31), a step past
// the wave's slice adds kDwDead (2^30); the descriptors cover < 2^30 bytes, so either one (or
// both: 3 * 2^30) fails the range check and the load returns zeros.
constexpr unsigned kDwDead = 0x40000000u;

template <bool FAST, int UPL, bool PAIR = false>
__device__ __forceinline__ void dw_fetch(const __amdgpu_buffer_rsrc_t& ra,
                                         const __amdgpu_buffer_rsrc_t& rx, const unsigned (&va)[UPL],
                                         const unsigned (&vx)[2], unsigned ba, unsigned bx,
                                         DwFrag<UPL, PAIR>& f) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if constexpr (FAST) {
    if constexpr (PAIR) {
      // interleaved partials: {a0, a1, b0, b1} per pair of units (va[0] is the pair's byte offset)
#pragma unroll
      for (int pr = 0; pr < UPL / 2; ++pr) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(va[0] + ba + 16u * pr), 0, 0);
        const f32x4_t f4 = __builtin_bit_cast(f32x4_t, v);
        f.a[2 * pr] = f4[0]; f.a[2 * pr + 1] = f4[1];
        f.b[2 * pr] = f4[2]; f.b[2 * pr + 1] = f4[3];
      }
    } else if constexpr (UPL == 4) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(va[0] + ba), 0, 0);
      const f32x4_t f4 = __builtin_bit_cast(f32x4_t, v);
      f.a[0] = f4[0]; f.a[1] = f4[1]; f.a[2] = f4[2]; f.a[3] = f4[3];
    } else {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(ra, (int)(va[0] + ba), 0, 0);
      const f32x2 f2 = __builtin_bit_cast(f32x2, v);
      f.a[0] = f2[0]; f.a[1] = f2[1];
    }
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(vx[0] + bx), 0, 0);
    const f32x2 f2 = __builtin_bit_cast(f32x2, w);
    f.x = make_float2(f2[0], f2[1]);
  } else {
    float g[2];
#pragma unroll
    for (int k = 0; k < UPL; ++k)
      f.a[k] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(va[k] + ba), 0, 0));
    if constexpr (PAIR) {   // (never taken: pair operands are whole aligned vectors; host contract)
#pragma unroll
      for (int k = 0; k < UPL; ++k) f.b[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      g[k] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(vx[k] + bx), 0, 0));
    f.x = make_float2(g[0], g[1]);
  }
}

// `after_prologue` runs between the ring's first fetches and the main loop: loads that are only
// needed later (the optimizer state) go there, so that they queue BEHIND the first operands —
// vector-memory results return in issue order.
template <bool FAST, int UPL, bool PAIR, typename Hook>
__device__ __forceinline__ void dw_mainloop(const __amdgpu_buffer_rsrc_t& ra,
                                            const __amdgpu_buffer_rsrc_t& rx,
                                            const unsigned (&va)[UPL], const unsigned (&vx)[2],
                                            unsigned oa, unsigned ox, unsigned sa, unsigned sx,
                                            int nsteps, dw_f32x4 (&acc)[UPL][2], float (&cs)[UPL],
                                            Hook after_prologue) {
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja) {
    acc[ja][0] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    acc[ja][1] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    cs[ja] = 0.f;
  }
  DwFrag<UPL, PAIR> ring[DW_RING];
#pragma unroll
  for (int p = 0; p < DW_RING; ++p) {
    const bool live = p < nsteps;
    dw_fetch<FAST, UPL, PAIR>(ra, rx, va, vx, live ? oa + (unsigned)p * sa : kDwDead,
                              live ? ox + (unsigned)p * sx : kDwDead, ring[p]);
  }
  after_prologue();
  for (int s0 = 0; s0 < nsteps; s0 += DW_RING) {
#pragma unroll
    for (int p = 0; p < DW_RING; ++p) {
      const DwFrag<UPL, PAIR> f = ring[p];
      const int sn = s0 + p + DW_RING;
      const bool live = sn < nsteps;  // steps past the slice must not read the next wave's rows
      dw_fetch<FAST, UPL, PAIR>(ra, rx, va, vx, live ? oa + (unsigned)sn * sa : kDwDead,
                                live ? ox + (unsigned)sn * sx : kDwDead, ring[p]);
      __builtin_amdgcn_sched_barrier(0);  // keep the refill here, DW_RING steps ahead of its use
      const float xv[2] = {f.x.x, f.x.y};
#pragma unroll
      for (int ja = 0; ja < UPL; ++ja) {
        float av = f.a[ja];
        if constexpr (PAIR) av = __fadd_rn(av, f.b[ja]);
        acc[ja][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv[0], acc[ja][0], 0, 0, 0);
        acc[ja][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv[1], acc[ja][1], 0, 0, 0);
        cs[ja] += av;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The same partial tile on the bf16 matrix pipe at fp32 accuracy (DwArgs::split; DESIGN.md §3.5 is
// the arithmetic): v_mfma_f32_16x16x32_bf16 takes 32 batch rows per instruction, a lane holding 8
// CONSECUTIVE rows (k = 8 (lane >> 4) + e) of one unit / one column.  With operands that are
// row-major in the batch that is eight wide loads per operand and step — row 8 q + r of the step,
// r = 0 .. 7, the same 16- / 8-byte vectors as the fp32 loop — and the eight values of a unit are
// already in the lane that needs them: split each exactly into hi + mid + lo (split3), pack the
// planes into the 8-element fragments, six products per (unit, column) position:
//     small += a_lo x_hi + a_hi x_lo + a_mid x_mid + a_hi x_mid + a_mid x_hi ;  main += a_hi x_hi
// (two accumulators: the five small classes are never rounded against the running main sum; they
// are added once at the end).  A and B share the row <-> (lane group, element) map, so the sum over
// rows does not depend on how the hardware orders k inside an instruction.  Per 32-row step a wave
// issues 16 loads, ~260 VALU (the splits) and 48 MFMAs of 16 passes: 2.7x the fp32 loop's 64 MFMAs
// of 32.  Accumulator layout = the fp32 loop's: everything after the main loop is shared.
// ---------------------------------------------------------------------------
template <int UPL, bool PAIR = false>
struct DwRaw32 {
  float a[8][UPL];
  float b[PAIR ? 8 : 1][UPL];   // PAIR: the second partial's rows (added when the step is worked on)
  float x[8][2];
};
template <int UPL, bool PAIR = false>
__device__ __forceinline__ void dw_fetch32(const __amdgpu_buffer_rsrc_t& ra,
                                           const __amdgpu_buffer_rsrc_t& rx, unsigned va, unsigned vx,
                                           unsigned ba, unsigned bx, unsigned lda4, unsigned ldx4,
                                           DwRaw32<UPL, PAIR>& f) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  static_assert(UPL == 4 || UPL == 2, "16- / 8-byte loads of dZ");
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if constexpr (PAIR) {
      // interleaved partials: {a0, a1, b0, b1} per pair of units (va is the pair's byte offset)
#pragma unroll
      for (int pr = 0; pr < UPL / 2; ++pr) {
        const u32x4 v =
            __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(va + ba + (unsigned)r * lda4 + 16u * pr), 0, 0);
        const f32x4_t f4 = __builtin_bit_cast(f32x4_t, v);
        f.a[r][2 * pr] = f4[0]; f.a[r][2 * pr + 1] = f4[1];
        f.b[r][2 * pr] = f4[2]; f.b[r][2 * pr + 1] = f4[3];
      }
    } else if constexpr (UPL == 4) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(va + ba + (unsigned)r * lda4), 0, 0);
      const f32x4_t f4 = __builtin_bit_cast(f32x4_t, v);
      f.a[r][0] = f4[0]; f.a[r][1] = f4[1]; f.a[r][2] = f4[2]; f.a[r][3] = f4[3];
    } else {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(ra, (int)(va + ba + (unsigned)r * lda4), 0, 0);
      const f32x2 f2a = __builtin_bit_cast(f32x2, v);
      f.a[r][0] = f2a[0]; f.a[r][1] = f2a[1];
    }
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(vx + bx + (unsigned)r * ldx4), 0, 0);
    const f32x2 f2 = __builtin_bit_cast(f32x2, w);
    f.x[r][0] = f2[0]; f.x[r][1] = f2[1];
  }
}
// v_cvt_pk_bf16_f32: two fp32 -> two bf16 (round to nearest even) in one dword, a in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
// split3 of 8 rows x 2 ADJACENT units (the two floats of one 8-byte piece of every row: adjacent
// registers, so the residuals are packed-fp32 subtractions): 4.5 VALU per element —
//   per (2 rows x 2 units): 2 cvt_pk, 2 lshl + 2 and (the bf16 back as fp32), 2 v_pk_add -> r;
//   the same -> s; 2 cvt_pk.  Element e of a fragment = row e (k = 8 (lane >> 4) + e).
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8x2(const dw_f32x2 (&v)[8], bf16x8 (&hi)[2], bf16x8 (&mid)[2],
                                         bf16x8 (&lo)[2]) {
  u32x4 H[2], M[2], L[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const dw_f32x2 xa = v[2 * p], xb = v[2 * p + 1];          // rows 2p, 2p + 1: (unit 0, unit 1)
    const unsigned h0 = cvt_pk_bf16(xa[0], xb[0]), h1 = cvt_pk_bf16(xa[1], xb[1]);
    const dw_f32x2 ha = {__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h1 << 16)};
    const dw_f32x2 hb = {__builtin_bit_cast(float, h0 & 0xffff0000u),
                         __builtin_bit_cast(float, h1 & 0xffff0000u)};
    const dw_f32x2 ra = xa - ha, rb = xb - hb;                // exact
    const unsigned m0 = cvt_pk_bf16(ra[0], rb[0]), m1 = cvt_pk_bf16(ra[1], rb[1]);
    const dw_f32x2 ma = {__builtin_bit_cast(float, m0 << 16), __builtin_bit_cast(float, m1 << 16)};
    const dw_f32x2 mb = {__builtin_bit_cast(float, m0 & 0xffff0000u),
                         __builtin_bit_cast(float, m1 & 0xffff0000u)};
    const dw_f32x2 sa = ra - ma, sb = rb - mb;                // exact; bf16(s) is exact too
    H[0][p] = h0; H[1][p] = h1;
    M[0][p] = m0; M[1][p] = m1;
    L[0][p] = cvt_pk_bf16(sa[0], sb[0]); L[1][p] = cvt_pk_bf16(sa[1], sb[1]);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    hi[u] = __builtin_bit_cast(bf16x8, H[u]);
    mid[u] = __builtin_bit_cast(bf16x8, M[u]);
    lo[u] = __builtin_bit_cast(bf16x8, L[u]);
  }
}
// va / vx: byte offset of this lane's vector in row 8 q of a step (kBufOob: outside the operand);
// oa / ox: byte offset of the wave's first row.
template <int UPL, bool PAIR, typename Hook>
__device__ __forceinline__ void dw_mainloop_split(const __amdgpu_buffer_rsrc_t& ra,
                                                  const __amdgpu_buffer_rsrc_t& rx, unsigned va,
                                                  unsigned vx, unsigned oa, unsigned ox,
                                                  unsigned lda4, unsigned ldx4, int nsteps,
                                                  dw_f32x4 (&acc)[UPL][2], float (&cs)[UPL],
                                                  Hook after_prologue) {
  static_assert(UPL == 4 || UPL == 2, "pairs of adjacent units per lane");
  dw_f32x4 small[UPL][2];
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja) {
    acc[ja][0] = acc[ja][1] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    small[ja][0] = small[ja][1] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    cs[ja] = 0.f;
  }
  const unsigned sa32 = 32u * lda4, sx32 = 32u * ldx4;
  // one 32-row step: split the raw rows three ways, then the six product classes, each over the
  // eight (unit block, column block) positions — consecutive MFMAs write different accumulators
  // (no back-to-back dependent chain)
  auto work = [&](const DwRaw32<UPL, PAIR>& cur) {
    bf16x8 xh[2], xm[2], xl[2], ah[UPL], am[UPL], al[UPL];
    {
      dw_f32x2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = dw_f32x2{cur.x[r][0], cur.x[r][1]};
      split8x2(v, xh, xm, xl);
    }
#pragma unroll
    for (int jp = 0; jp < UPL; jp += 2) {
      dw_f32x2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r] = dw_f32x2{cur.a[r][jp], cur.a[r][jp + 1]};
        if constexpr (PAIR) v[r] += dw_f32x2{cur.b[r][jp], cur.b[r][jp + 1]};
        cs[jp] += v[r][0];
        cs[jp + 1] += v[r][1];
      }
      bf16x8 h2[2], m2[2], l2[2];
      split8x2(v, h2, m2, l2);
      ah[jp] = h2[0]; ah[jp + 1] = h2[1];
      am[jp] = m2[0]; am[jp + 1] = m2[1];
      al[jp] = l2[0]; al[jp + 1] = l2[1];
    }
#define PA_DW_CLASS(A_, X_, ACC_)                                                          \
  _Pragma("unroll") for (int ja = 0; ja < UPL; ++ja)                                        \
  _Pragma("unroll") for (int jx = 0; jx < 2; ++jx)                                          \
      ACC_[ja][jx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[ja], X_[jx], ACC_[ja][jx], 0, 0, 0);
    PA_DW_CLASS(al, xh, small)
    PA_DW_CLASS(ah, xl, small)
    PA_DW_CLASS(am, xm, small)
    PA_DW_CLASS(ah, xm, small)
    PA_DW_CLASS(am, xh, small)
    PA_DW_CLASS(ah, xh, acc)
#undef PA_DW_CLASS
  };
  // two raw buffers in ping-pong (nsteps is even: the host-side slice is a multiple of 512 rows):
  // the next step's sixteen loads are in flight under this step's splits and MFMAs, and nothing
  // is copied between them
  DwRaw32<UPL, PAIR> ra0, ra1;
  auto fetch = [&](int step, DwRaw32<UPL, PAIR>& f) {
    const bool live = step < nsteps;    // steps past the slice must not read the next wave's rows
    dw_fetch32<UPL, PAIR>(ra, rx, va, vx, live ? oa + (unsigned)step * sa32 : kDwDead,
                          live ? ox + (unsigned)step * sx32 : kDwDead, lda4, ldx4, f);
  };
  fetch(0, ra0);
  after_prologue();
  for (int s = 0; s < nsteps; s += 2) {
    fetch(s + 1, ra1);
    __builtin_amdgcn_sched_barrier(0);
    work(ra0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(s + 2, ra0);
    __builtin_amdgcn_sched_barrier(0);
    work(ra1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja)
#pragma unroll
    for (int jx = 0; jx < 2; ++jx)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ja][jx][r] += small[ja][jx][r];
}

__device__ __forceinline__ void sac_step_tail(const TailJob& t, float* lds, int tid) {
  float sa = 0.f, sb = 0.f;
  for (int base = 0; base < t.tiles; base += 256) {
    if (tid < 256) {
      lds[tid] = base + tid < t.tiles ? t.part_a[base + tid] : 0.f;
      lds[256 + tid] = base + tid < t.tiles ? t.part_b[base + tid] : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
      const int n = t.tiles - base < 256 ? t.tiles - base : 256;
      for (int k = 0; k < n; ++k) {
        sa += lds[k];
        sb += lds[256 + k];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (t.actor_loss) t.actor_loss[0] = sa / (float)t.B;
    // ((q1 - y)^2 + (q2 - y)^2 summed) / B / 2  ==  (mse1 + mse2) / 2   (critic_utils.py:170-203)
    t.critic_loss[0] = (sb / (float)t.B) * 0.5f;
  }
  if (!t.log_alpha) return;
  const float ea = expf(t.log_alpha[0]);
  float part = 0.f;
  if (tid < 256)
    for (int b = tid; b < t.B; b += 256) part += -ea * (t.logp[b] + t.target_entropy);
  if (tid < 256) lds[tid] = part;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) lds[tid] += lds[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    const float g = lds[0] / (float)t.B;
    if (t.alpha_loss_out) t.alpha_loss_out[0] = g;
    AdamState st;
    st.p = t.log_alpha; st.m = t.am; st.v = t.av; st.vmax = t.avmax;
    const float pnew = adam_update(t.ac, st, 0, g);
    t.alpha[0] = expf(pnew);
  }
}

// UPL = units per lane: 4 -> 64-row tiles (one 16-byte load of dZ per step), 2 -> 32-row tiles
// (8-byte loads); everything below is written for TM = 16 UPL.
template <int UPL, bool SPLIT = false, bool PAIR = false>
__device__ __forceinline__ void weight_grad_body(const DwArgs& a, float* part, float* csum) {
  constexpr int TM = 16 * UPL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row offsets stay in SGPRs
  const int KSP = a.ksplit > 1 ? a.ksplit : 1;
  int wg_tile = (int)blockIdx.x / KSP;
  const int kslice = (int)blockIdx.x % KSP;
  if (a.xcd_order && wg_tile < a.total_tiles) wg_tile = (wg_tile & 7) * (a.total_tiles >> 3) + (wg_tile >> 3);
  if (wg_tile >= a.total_tiles) {
    if (a.tail.kind == 1) {
      sac_step_tail(a.tail, part, tid);
      return;
    }
    // mean |Q - target| of this step (deep_td_learning.py:358-359), fixed summation order
    float s = 0.f;
    for (int i = tid; i < a.ad.nabs; i += 512) s += a.ad.absd[i];
    part[tid] = s;
    __syncthreads();
    for (int w = 256; w >= 1; w >>= 1) {
      if (tid < w) part[tid] += part[tid + w];
      __syncthreads();
    }
    if (tid == 0) a.ad.loss_out[0] = part[0] * a.ad.inv_B;
    if (a.ad.umax_clear)
      for (int i = tid; i < HF_UMAX; i += 512) a.ad.umax_clear[i] = 0u;
    if (a.ad.y_restore)
      for (int i = tid; i < a.ad.n_restore; i += 512)
        __hip_atomic_store(a.ad.y_restore + i, kYPendingBits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 0);
  PA_STAMP_CYC(a.prof, blockIdx.x, wave, 14);
  // (a local, not a write to the by-value argument: that would copy the whole struct to scratch)
  const bool adam_all = a.ad.enabled && !(a.ad.guard && __hip_atomic_load(a.ad.guard, __ATOMIC_RELAXED,
                                                                          __HIP_MEMORY_SCOPE_AGENT) != 0);
  const int c = lane & 15, q = lane >> 4;
  int pi = 0;
#pragma unroll
  for (int k = 1; k < DW_MAX_PROB; ++k)
    if (a.nprob > k && wg_tile >= a.p[k].tile0) pi = k;
  const DwProblem& P = a.p[pi];
  const bool adam_on = adam_all && !P.raw;
  // the optimizer state this problem's parameters live in (wave-uniform selects)
  AdamState st = a.ad.st;
  const float* gbase = a.ad.grad_base;
  float* tgt = a.ad.tgt;
  if (P.net) {
    st = a.net2.st;
    gbase = a.net2.grad_base;
    tgt = a.net2.tgt;
  }
  const int t = wg_tile - P.tile0;
  const int i0 = (t / P.tiles_n) * TM, j0 = (t % P.tiles_n) * DW_TN;
  if (P.M == 1 && kslice > 0) return;  // the GEMV path below is not split
  if (P.M == 1) {
    // Single-output layers (dW3 = dq^T h2, db3 = sum dq): a matrix tile would be 63/64 padding.
    // 32 columns x 16 row groups per workgroup, sequential fma per thread, row groups summed in
    // a fixed order through LDS.
    const int col = j0 + (tid & 31), rg = tid >> 5;
    const __amdgpu_buffer_rsrc_t rz = buf_rsrc_n(P.dZ, (unsigned)a.B * (unsigned)P.ldz * 4u);
    const __amdgpu_buffer_rsrc_t rxx = buf_rsrc_n(P.X, (unsigned)a.B * (unsigned)P.ldx * 4u);
    const unsigned vcol = (col < P.N) ? (unsigned)col * 4u : kBufOob;
    float accv = 0.f, sdz = 0.f;
#pragma unroll 8
    for (int b = rg; b < a.B; b += 16) {
      const float dz = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rz, (int)((unsigned)(b * P.ldz) * 4u), 0, 0));
      const float xv = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (int)(vcol + (unsigned)(b * P.ldx) * 4u), 0, 0));
      accv = fmaf(dz, xv, accv);
      sdz += dz;
    }
    part[rg * 32 + (tid & 31)] = accv;
    if ((tid & 31) == 0) csum[rg] = sdz;
    __syncthreads();
    if (tid < 32 && col < P.N) {
      float g = part[tid];
#pragma unroll
      for (int w = 1; w < 16; ++w) g += part[w * 32 + tid];
      float* dst = P.dW + col;
      *dst = g;
      if (adam_on) {
        if (P.kind == 3) adam_generic_weight(a.ad, st, tgt, P, dst - gbase, 0, col, g);
        else adam_fused_weight(a.ad, P.kind, dst - a.ad.grad_base, 0, col, g);
      }
    }
    if (j0 == 0 && tid == 0) {
      float g = csum[0];
#pragma unroll
      for (int w = 1; w < 16; ++w) g += csum[w];
      P.db[0] = g;
      if (adam_on && !P.bias_frozen) {
        if (P.kind == 3) adam_generic_bias(a.ad, st, tgt, P.db - gbase, g);
        else adam_fused_bias(a.ad, P.db - a.ad.grad_base, g);
      }
    }
    return;
  }
  // wave-uniform: whole 16- / 8-byte vectors are inside or outside the operand
  const bool fa = ((P.ldz & (UPL - 1)) == 0) && ((P.M & (UPL - 1)) == 0) &&
                  ((reinterpret_cast<uintptr_t>(P.dZ) & (4 * UPL - 1)) == 0);
  const bool fx = ((P.ldx & 1) == 0) && ((P.N & 1) == 0) &&
                  ((reinterpret_cast<uintptr_t>(P.X) & 7) == 0);
  // byte sizes stay below 2^30 (max_batch * ld * 4; checked by the host)
  const __amdgpu_buffer_rsrc_t ra = buf_rsrc_n(P.dZ, (unsigned)a.B * (unsigned)P.ldz * 4u);
  const __amdgpu_buffer_rsrc_t rx = buf_rsrc_n(P.X, (unsigned)a.B * (unsigned)P.ldx * 4u);
  const int ua = i0 + UPL * c, cx = j0 + 2 * c;  // first unit / column of this lane's vectors
  // PAIR kernels: a problem whose dZ holds two interleaved partials (wave-uniform, DwProblem::dz_pair)
  // adds them as it loads; its element offsets are those of the pairs: twice the unit index
  const bool pairp = PAIR && P.dz_pair != 0;
  const int uoff = pairp ? 2 * ua : ua;
  unsigned va[UPL], vx[2];                       // per-component byte offsets of row q (kBufOob: none)
#pragma unroll
  for (int e = 0; e < UPL; ++e)
    va[e] = (ua + e < P.M) ? (unsigned)(q * P.ldz + uoff + e) * 4u : kBufOob;
#pragma unroll
  for (int e = 0; e < 2; ++e)
    vx[e] = (cx + e < P.N) ? (unsigned)(q * P.ldx + cx + e) * 4u : kBufOob;
  const unsigned sa = (unsigned)P.ldz * 16u, sx = (unsigned)P.ldx * 16u;  // bytes per 4-row step
  // this workgroup's slice of the batch (all of it unless ksplit > 1), this wave's share of that:
  // nsteps 4-row steps
  // (SPLIT: 32-row steps, so a wave's share is a multiple of 32 rows; rows past the batch read as
  //  zeros by the descriptors' range check, as above)
  constexpr int SLICE_Q = SPLIT ? 512 : 32;   // (SPLIT: an even number of 32-row steps per wave)
  const int bslice = ((a.B + KSP - 1) / KSP + SLICE_Q - 1) / SLICE_Q * SLICE_Q;
  const int nsteps = SPLIT ? bslice / 256 : bslice / 32, row0 = kslice * bslice + wave * (bslice / 8);
  const unsigned oa = (unsigned)row0 * (unsigned)P.ldz * 4u, ox = (unsigned)row0 * (unsigned)P.ldx * 4u;
  // Epilogue assignment, fixed now so that the optimizer state can be fetched under the main
  // loop: thread -> tile row tid >> 3, columns 4 (tid & 7) .. + 3.  evec: the four elements are one
  // aligned float4 of dW (and of every flat optimizer buffer, which share its offsets).
  // (32-row tiles: the upper half of the workgroup has no epilogue element)
  const bool eact = tid < TM * 8;
  const int erl = tid >> 3, ecg = tid & 7;
  const int eja = erl % UPL, ereg = (erl / UPL) & 3, eq = erl / (4 * UPL);
  const int erow = i0 + erl, ecol = j0 + 4 * ecg;
  const bool evec = eact && erow < P.M && ecol + 3 < P.N && ((P.ldw & 3) == 0) &&
                    ((reinterpret_cast<uintptr_t>(P.dW) & 15) == 0);
  const int64_t eflat = (P.dW + (int64_t)erow * P.ldw + ecol) - gbase;
  float4 p4, m4, v4, x4;
  p4 = m4 = v4 = x4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto prefetch_state = [&]() {
    if (evec && adam_on) {
      p4 = *reinterpret_cast<const float4*>(st.p + eflat);
      m4 = *reinterpret_cast<const float4*>(st.m + eflat);
      v4 = *reinterpret_cast<const float4*>(st.v + eflat);
      if (a.ad.c.amsgrad) x4 = *reinterpret_cast<const float4*>(st.vmax + eflat);
    }
  };
  dw_f32x4 acc[UPL][2];
  float cs[UPL];
  PA_STAMP(a.prof, blockIdx.x, wave, 1);
  auto run_loop = [&](auto pair_tag) {
    constexpr bool PR = decltype(pair_tag)::value;
    if constexpr (SPLIT) {
      // (host contract: fa && fx for every matrix problem of a split launch)
      const unsigned lda4 = (unsigned)P.ldz * 4u, ldx4 = (unsigned)P.ldx * 4u;
      const unsigned va8 = (ua < P.M) ? (unsigned)(8 * q * P.ldz + uoff) * 4u : kBufOob;
      const unsigned vx8 = (cx < P.N) ? (unsigned)(8 * q * P.ldx + cx) * 4u : kBufOob;
      dw_mainloop_split<UPL, PR>(ra, rx, va8, vx8, oa, ox, lda4, ldx4, nsteps, acc, cs, prefetch_state);
    } else {
      if (fa && fx) dw_mainloop<true, UPL, PR>(ra, rx, va, vx, oa, ox, sa, sx, nsteps, acc, cs, prefetch_state);
      else dw_mainloop<false, UPL, PR>(ra, rx, va, vx, oa, ox, sa, sx, nsteps, acc, cs, prefetch_state);
    }
  };
  if constexpr (PAIR) {
    if (pairp) run_loop(std::true_type{});
    else run_loop(std::false_type{});
  } else {
    run_loop(std::false_type{});
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 2);
  // ---- partial tiles: (waves 4..7 -> LDS, waves 0..3 add), then (waves 0..3 -> LDS, all sum)
  // element id of acc[ja][jx][reg] on `lane`: ((ja * 2 + jx) * 4 + reg) * 64 + lane
  if (wave >= 4) {
    float* dst = part + (wave - 4) * (TM * DW_TN) + lane;
#pragma unroll
    for (int ja = 0; ja < UPL; ++ja)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[((ja * 2 + jx) * 4 + r) * 64] = acc[ja][jx][r];
  }
#pragma unroll
  for (int ja = 0; ja < UPL; ++ja) {
    cs[ja] += __shfl_xor(cs[ja], 16);
    cs[ja] += __shfl_xor(cs[ja], 32);
    if (q == 0) csum[wave * TM + UPL * c + ja] = cs[ja];
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 3);
  __syncthreads();
  if (wave < 4) {
    float* slot = part + wave * (TM * DW_TN) + lane;
#pragma unroll
    for (int ja = 0; ja < UPL; ++ja)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ja][jx][r] += slot[((ja * 2 + jx) * 4 + r) * 64];
#pragma unroll
    for (int ja = 0; ja < UPL; ++ja)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx)
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[((ja * 2 + jx) * 4 + r) * 64] = acc[ja][jx][r];
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 4);
  __syncthreads();
  PA_STAMP(a.prof, blockIdx.x, wave, 5);
  {
    // thread -> one row, four consecutive columns of the tile (see the prefetch above)
    float g4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // (threads without an epilogue element read slot 0: in range, unused)
      const int idx = eact ? ((eja * 2 + (e & 1)) * 4 + ereg) * 64 + eq * 16 + 2 * ecg + (e >> 1) : 0;
      float sum = part[idx];
#pragma unroll
      for (int w = 1; w < 4; ++w) sum += part[w * (TM * DW_TN) + idx];
      g4[e] = sum;
    }
    if (KSP > 1) {
      // ---- split-K: publish this slice's partial tile, last arriver adds them up in slice order
      __shared__ unsigned is_last;
      float* mine = a.kscratch + ((int64_t)wg_tile * KSP + kslice) * (TM * DW_TN + TM);
      {
        // one 16-byte write-through store per thread (dword write-through stores are one fabric
        // write each: ~6x the time per byte, MI355X_MICROARCH.md)
        const f32x4_t v = {g4[0], g4[1], g4[2], g4[3]};
        if (eact)
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(mine + tid * 4), "v"(v) : "memory");
      }
      if (tid < TM) {
        float sdb = csum[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) sdb += csum[w * TM + tid];
        __hip_atomic_store(mine + TM * DW_TN + tid, sdb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores are out
      __syncthreads();
      if (tid == 0) is_last = (atomicAdd(a.ktickets + wg_tile, 1u) == (unsigned)KSP - 1u) ? 1u : 0u;
      __syncthreads();
      if (!is_last) return;
      const float* base = a.kscratch + (int64_t)wg_tile * KSP * (TM * DW_TN + TM);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sum = 0.f;
        for (int k = 0; k < KSP && eact; ++k)
          sum += __hip_atomic_load(base + (int64_t)k * (TM * DW_TN + TM) + tid * 4 + e,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g4[e] = sum;
      }
      if (tid < TM) {
        float sdb = 0.f;
        for (int k = 0; k < KSP; ++k)
          sdb += __hip_atomic_load(base + (int64_t)k * (TM * DW_TN + TM) + TM * DW_TN + tid,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        csum[tid] = sdb;   // read back by the bias epilogue below (waves 1.. hold zeros there)
#pragma unroll
        for (int w = 1; w < 8; ++w) csum[w * TM + tid] = 0.f;
      }
      if (tid == 0) a.ktickets[wg_tile] = 0u;
      __syncthreads();
    }
    unsigned um_m = 0u;   // max |new weight| of this thread's elements (row scales of the fp16 row pass)
    unsigned um_t = 0u;   // ... of the soft-updated target's (kind 3)
    if (evec) {
      *reinterpret_cast<float4*>(P.dW + (int64_t)erow * P.ldw + ecol) =
          make_float4(g4[0], g4[1], g4[2], g4[3]);
      if (adam_on) {
        float pv[4] = {p4.x, p4.y, p4.z, p4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
        float vv[4] = {v4.x, v4.y, v4.z, v4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) adam_math(a.ad.c, g4[e], pv[e], mv[e], vv[e], xv[e]);
        const float4 pn = make_float4(pv[0], pv[1], pv[2], pv[3]);
        um_m = umax4(pn);
        *reinterpret_cast<float4*>(st.p + eflat) = pn;
        *reinterpret_cast<float4*>(st.m + eflat) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4*>(st.v + eflat) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (a.ad.c.amsgrad)
          *reinterpret_cast<float4*>(st.vmax + eflat) = make_float4(xv[0], xv[1], xv[2], xv[3]);
        // fragment-major copies: four consecutive k of one unit are one float4 slot
        if (P.kind == 0) {
          *reinterpret_cast<float4*>(a.ad.W2f + wf16_index_(erow, ecol, a.ad.nkg_w2)) = pn;
#pragma unroll
          for (int e = 0; e < 4; ++e) a.ad.W2tf[wf16_index_(ecol + e, erow, a.ad.nkg_w2t)] = pv[e];
          if (a.ad.oW2f) {   // Double DQN's argmax pass reads the online W2 through the target tile
            *reinterpret_cast<float4*>(a.ad.oW2f + w2f_index(erow, ecol, a.ad.nkg_t)) = pn;
            if (a.ad.oW2sp) store_w2sp4(a.ad.oW2sp, erow, ecol, pn);
          }
        } else if (P.kind == 1) {
          *reinterpret_cast<float4*>(a.ad.W1f + wf16_index_(erow, ecol, a.ad.nkg_w1)) = pn;
        } else if (P.kind == 3) {
          if (P.pkf) *reinterpret_cast<float4*>(P.pkf + wf16_index_(erow, ecol, P.nkgf)) = pn;
          if (P.pks) store_wsp16_4(P.pks, erow, ecol, pn, P.nks);
          if (P.pktf) {
#pragma unroll
            for (int e = 0; e < 4; ++e) P.pktf[wf16_index_(ecol + e, erow, P.nkgtf)] = pv[e];
          }
          if (P.pkts) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_wsp16_1(P.pkts, ecol + e, erow, pv[e], P.nkts);
          }
        }
        if (a.ad.soft_next && tgt) {  // update_target_network (common/utils.py:214-226)
          const float4 t4 = *reinterpret_cast<const float4*>(tgt + eflat);
          float4 tn;
          tn.x = __fadd_rn(__fmul_rn(a.ad.tau, pv[0]), __fmul_rn(a.ad.one_minus_tau, t4.x));
          tn.y = __fadd_rn(__fmul_rn(a.ad.tau, pv[1]), __fmul_rn(a.ad.one_minus_tau, t4.y));
          tn.z = __fadd_rn(__fmul_rn(a.ad.tau, pv[2]), __fmul_rn(a.ad.one_minus_tau, t4.z));
          tn.w = __fadd_rn(__fmul_rn(a.ad.tau, pv[3]), __fmul_rn(a.ad.one_minus_tau, t4.w));
          *reinterpret_cast<float4*>(tgt + eflat) = tn;
          if (P.kind == 0) {
            *reinterpret_cast<float4*>(a.ad.tW2f + w2f_index(erow, ecol, a.ad.nkg_t)) = tn;
            if (a.ad.tW2sp) store_w2sp4(a.ad.tW2sp, erow, ecol, tn);
          } else if (P.kind == 1 && a.ad.tW1sp) {
            // (sp_S is a multiple of 16 and ecol of 4: a float4 is inside or outside the state part)
            if (ecol < a.ad.sp_S) store_wsp4(a.ad.tW1sp, erow, ecol, tn, a.ad.sp_S >> 4);
          } else if (P.kind == 3 && P.pkf_t)
            *reinterpret_cast<float4*>(P.pkf_t + wf16_index_(erow, ecol, P.nkgf)) = tn;
          um_t = umax4(tn);
        } else if (P.um_acc_t && tgt) {
          um_t = umax4(*reinterpret_cast<const float4*>(tgt + eflat));   // no soft update: as it is
        }
      } else if (P.um_acc) {
        // (a tripped guard: the parameters stay as they are, and so do their maxima)
        um_m = umax4(*reinterpret_cast<const float4*>(st.p + eflat));
        if (P.um_acc_t && tgt) um_t = umax4(*reinterpret_cast<const float4*>(tgt + eflat));
      }
    } else if (eact && erow < P.M) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ecol + e < P.N) {
          float* dst = P.dW + (int64_t)erow * P.ldw + ecol + e;
          *dst = g4[e];
          if (adam_on) {
            if (P.kind == 3)
              adam_generic_weight(a.ad, st, tgt, P, dst - gbase, erow, ecol + e, g4[e]);
            else
              um_m = umaxu(um_m, abs_bits(adam_fused_weight(a.ad, P.kind, dst - a.ad.grad_base, erow,
                                                            ecol + e, g4[e])));
          }
        }
      }
    }
    if (a.ad.umax_acc && adam_on && P.kind <= 1) {   // (workgroup-uniform)
      // the eight threads of a tile row are adjacent lanes
      um_m = umaxu(um_m, (unsigned)__shfl_xor((int)um_m, 1));
      um_m = umaxu(um_m, (unsigned)__shfl_xor((int)um_m, 2));
      um_m = umaxu(um_m, (unsigned)__shfl_xor((int)um_m, 4));
      if (eact && ecg == 0 && erow < P.M) umax_atomic(a.ad.umax_acc + (P.kind == 0 ? HF_UNITS : 0) + erow, um_m);
    }
    if (P.um_acc && !P.raw) {   // (workgroup-uniform)
      um_m = umaxu(um_m, (unsigned)__shfl_xor((int)um_m, 1));
      um_m = umaxu(um_m, (unsigned)__shfl_xor((int)um_m, 2));
      um_m = umaxu(um_m, (unsigned)__shfl_xor((int)um_m, 4));
      um_t = umaxu(um_t, (unsigned)__shfl_xor((int)um_t, 1));
      um_t = umaxu(um_t, (unsigned)__shfl_xor((int)um_t, 2));
      um_t = umaxu(um_t, (unsigned)__shfl_xor((int)um_t, 4));
      if (eact && ecg == 0 && erow < P.M) {
        umax_atomic(P.um_acc + erow, um_m);
        if (P.um_acc_t) umax_atomic(P.um_acc_t + erow, um_t);
        if (j0 == 0) {
          if (P.um_clear) P.um_clear[erow] = 0u;
          if (P.um_clear_t) P.um_clear_t[erow] = 0u;
        }
      }
    }
  }
  if (j0 == 0 && tid < TM && (i0 + tid) < P.M) {
    float s = csum[tid];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += csum[w * TM + tid];
    P.db[i0 + tid] = s;
    if (adam_on && !P.bias_frozen) {
      if (P.kind == 3) adam_generic_bias(a.ad, st, tgt, (P.db + i0 + tid) - gbase, s);
      else adam_fused_bias(a.ad, (P.db + i0 + tid) - a.ad.grad_base, s);
    }
  }
  PA_STAMP(a.prof, blockIdx.x, wave, 6);
  PA_STAMP_CYC(a.prof, blockIdx.x, wave, 15);
}


static __global__ __launch_bounds__(512, 2) void weight_grad_kernel(DwArgs a) {
  __shared__ float part[4 * DW_TM * DW_TN];  // 32 KB: four partial tiles
  __shared__ float csum[8 * DW_TM];
  weight_grad_body<4>(a, part, csum);
}
// DwArgs::tm == 32 (two kernels, not one with a branch: the merged body allocated 208 registers
// against 158 / 103 for the two on their own)
static __global__ __launch_bounds__(512, 2) void weight_grad_kernel32(DwArgs a) {
  __shared__ float part[4 * 32 * DW_TN];
  __shared__ float csum[8 * 32];
  weight_grad_body<2>(a, part, csum);
}
// DwArgs::split == 1: 64-row tiles, main loop on v_mfma_f32_16x16x32_bf16 (dw_mainloop_split)
static __global__ __launch_bounds__(512, 2) void weight_grad_split_kernel(DwArgs a) {
  __shared__ float part[4 * DW_TM * DW_TN];
  __shared__ float csum[8 * DW_TM];
  weight_grad_body<4, true>(a, part, csum);
}
// ... and 32-row tiles (tm == 32): the DQN chain's 113-workgroup launch
static __global__ __launch_bounds__(512, 2) void weight_grad_split_kernel32(DwArgs a) {
  __shared__ float part[4 * 32 * DW_TN];
  __shared__ float csum[8 * 32];
  weight_grad_body<2, true>(a, part, csum);
}

// ---------------------------------------------------------------------------
// Stand-alone AdamW on a flat gradient buffer (data-parallel path: after the all-reduce).
// ---------------------------------------------------------------------------
struct AdamArgs {
  AdamState st; const float* g;
  int64_t n;
  AdamScalars c;
  const float* absd; int nabs; float inv_B; float* loss_out;  // loss_out[0] = mean |Q - target|
};

static __global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  __shared__ float red[256];
  if (blockIdx.x == 0 && a.loss_out) {
    // mean |Q - target| of this step (deep_td_learning.py:358-359), fixed summation order
    float s = 0.f;
    for (int i = threadIdx.x; i < a.nabs; i += 256) s += a.absd[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) a.loss_out[0] = red[0] * a.inv_B;
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  (void)adam_update(a.c, a.st, i, a.g[i]);
}

// Two networks' stand-alone AdamW in one launch (blockIdx.y picks the network): the data-parallel
// step of a pair of networks (PPO's actor + critic) after ONE all-reduce of both gradient buffers.
static __global__ __launch_bounds__(256) void adamw2_kernel(AdamArgs a0, AdamArgs a1) {
  const AdamArgs& a = blockIdx.y == 0 ? a0 : a1;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  (void)adam_update(a.c, a.st, i, a.g[i]);
}

// AdamW on the flat DQN gradient buffer AFTER a data-parallel all-reduce, with the same optimizer
// tail as the fused weight-gradient kernel: fragment-major copies refreshed, optional soft update
// of the target network for the next step.  The parameter index is decoded back to (tensor, row,
// col) from the flat layout W1 | b1 | W2 | b2 | W3 | b3.
struct AdamDqnArgs {
  AdamFuse f;
  const float* g;
  int64_t n;
  int64_t off[6];
  int IN, H1, H2;
};
static __global__ __launch_bounds__(256) void adamw_dqn_kernel(AdamDqnArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int uid = -1;        // slot of this element's row in the umax buffer (weights of W1 / W2 only)
  unsigned um = 0u;
  if (i < a.n) {
    const float g = a.g[i];
    if (i >= a.off[0] && i < a.off[0] + (int64_t)a.H1 * a.IN) {
      const int64_t e = i - a.off[0];
      uid = (int)(e / a.IN);
      um = abs_bits(adam_fused_weight(a.f, 1, i, uid, (int)(e % a.IN), g));
    } else if (i >= a.off[2] && i < a.off[2] + (int64_t)a.H2 * a.H1) {
      const int64_t e = i - a.off[2];
      const int row = (int)(e / a.H1);
      uid = HF_UNITS + row;
      um = abs_bits(adam_fused_weight(a.f, 0, i, row, (int)(e % a.H1), g));
    } else if (i >= a.off[4] && i < a.off[4] + a.H2) {
      (void)adam_fused_weight(a.f, 2, i, 0, (int)(i - a.off[4]), g);
    } else {
      adam_fused_bias(a.f, i, g);   // biases (and the zero alignment gaps)
    }
  }
  if (a.f.umax_acc) {   // (grid-uniform)
    // one atomic per row and wave: the lanes of each row present in the wave reduce among themselves
    const int lane = threadIdx.x & 63;
    unsigned long long rem = __ballot(uid >= 0);
    while (rem) {
      const int leader = __ffsll((long long)rem) - 1;
      const int u = __shfl(uid, leader);
      const bool mine = uid == u;
      unsigned v = mine ? um : 0u;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v = umaxu(v, (unsigned)__shfl_xor((int)v, o));
      if (lane == leader) umax_atomic(a.f.umax_acc + u, v);
      rem &= ~__ballot(mine);
    }
    if (i < HF_UMAX) a.f.umax_clear[i] = 0u;
  }
}

// theta' <- tau * theta + (1 - tau) * theta'   (common/utils.py:214-226)
static __global__ __launch_bounds__(256) void soft_update_kernel(float* __restrict__ tgt,
                                                          const float* __restrict__ src,
                                                          int64_t n, float tau,
                                                          float one_minus_tau) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  tgt[i] = __fadd_rn(__fmul_rn(tau, src[i]), __fmul_rn(one_minus_tau, tgt[i]));
}

// x[b] = state[b] || action_rep[b]   (q_value_networks.py:166-168 torch.cat)
static __global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ state,
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
