// Philox4x32-10 and the deterministic without-replacement index sampler of the replay arena
// (device code shared by arena.hip's stand-alone launch and the learn loop's prologue launch in
// dqn.hip, which draws the index lists of a call and rebuilds the packed weight copies in one grid).
#pragma once
#include "common.hpp"

namespace pa {

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                      uint32_t c3, uint32_t k0, uint32_t k1,
                                                      uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Unbiased integer in [0, n) from the four words of one Philox block (Lemire's
// multiply-shift with rejection; falls back to the last word after 4 rejections).
__host__ __device__ __forceinline__ uint32_t bounded_draw(const uint32_t w[4], uint32_t n) {
  const uint32_t thresh = (uint32_t)(0u - n) % n;
  uint64_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m = (uint64_t)w[i] * n;
    if ((uint32_t)m >= thresh) break;
  }
  return (uint32_t)(m >> 32);
}

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_MAX_B = 8192;
constexpr unsigned long long EMPTY_ENTRY = ~0ull;

// Deterministic uniform sample of B distinct values from [0, n).
//   round t: every unresolved position i proposes v = draw(philox(i, t, offset; seed)).
//   A proposal is accepted iff v was not accepted in an earlier round and i is the
//   lowest position proposing v in this round.  Losers redraw in round t+1.
// The hash table (open addressing, LDS) stores (v << 32 | t << 16 | i); min() on that
// word implements "earlier round, then lower position" without ordering races.
// The rule is symmetric in the values, so the resulting set is uniform over B-subsets.
// Block r of the grid draws the sample of round r (Philox counter offset + r) into
// idx_out[r * B ...]: one launch covers every round of a learn() call.
struct SampleArgs {
  int64_t* idx_out;
  uint32_t n;
  int B, hs;
  uint32_t seed_lo, seed_hi;
  uint64_t offset0;
};
// One workgroup (SAMPLE_THREADS threads) draws the sample of round `round`; `table` is the
// workgroup's dynamic LDS region (hs 8-byte entries + 16 bytes).
__device__ __forceinline__ void sample_indices_block(const SampleArgs& a, int round,
                                                     unsigned long long* table) {
  int64_t* __restrict__ idx_out = a.idx_out;
  const uint32_t n = a.n, seed_lo = a.seed_lo, seed_hi = a.seed_hi;
  const int B = a.B, hs = a.hs;
  const uint64_t offset = a.offset0 + (uint64_t)round;
  const uint32_t off_lo = (uint32_t)offset, off_hi = (uint32_t)(offset >> 32);
  idx_out += (int64_t)round * B;
  // all LDS lives in the dynamic region so its base stays 16-byte aligned
  int* pending_p = reinterpret_cast<int*>(table + hs);
  const int tid = threadIdx.x;
  for (int i = tid; i < hs; i += SAMPLE_THREADS) table[i] = EMPTY_ENTRY;
  constexpr int PER = SAMPLE_MAX_B / SAMPLE_THREADS;
  uint32_t val[PER];
  bool done[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    done[q] = (tid + q * SAMPLE_THREADS) >= B;
    val[q] = 0;
  }
  const uint32_t mask = (uint32_t)hs - 1;
  __syncthreads();
  for (uint32_t t = 0; t < 65535u; ++t) {
    // propose
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (done[q]) continue;
      const uint32_t i = tid + q * SAMPLE_THREADS;
      uint32_t w[4];
      philox4x32_10(i, t, off_lo, off_hi, seed_lo, seed_hi, w);
      const uint32_t v = bounded_draw(w, n);
      val[q] = v;
      const unsigned long long entry = ((unsigned long long)v << 32) | (t << 16) | i;
      uint32_t h = (v * 0x9E3779B1u) & mask;
      for (int probe = 0; probe < hs; ++probe) {  // the table is never more than half full
        const unsigned long long old = atomicCAS(&table[h], EMPTY_ENTRY, entry);
        if (old == EMPTY_ENTRY) break;
        if ((uint32_t)(old >> 32) == v) {
          atomicMin(&table[h], entry);
          break;
        }
        h = (h + 1) & mask;
      }
    }
    if (tid == 0) *pending_p = 0;
    __syncthreads();
    // resolve
    bool any = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (done[q]) continue;
      const uint32_t i = tid + q * SAMPLE_THREADS;
      const uint32_t v = val[q];
      uint32_t h = (v * 0x9E3779B1u) & mask;
      for (int probe = 0; probe < hs && (uint32_t)(table[h] >> 32) != v; ++probe)
        h = (h + 1) & mask;
      const unsigned long long e = table[h];
      if ((uint32_t)(e & 0xFFFFu) == i && (uint32_t)((e >> 16) & 0xFFFFu) == t) {
        done[q] = true;
        idx_out[i] = (int64_t)v;
      } else {
        any = true;
      }
    }
    if (any) atomicOr(pending_p, 1);
    __syncthreads();
    const int p = *pending_p;
    __syncthreads();
    if (!p) break;
  }
}


// host side (arena.hip): argument checks with the reference's error text, and the launch arguments
int sample_check(int64_t population, int32_t B);
SampleArgs sample_args(int64_t population, uint64_t seed, uint64_t offset, int32_t B,
                       int64_t* idx_out_dev);

inline size_t sample_smem_bytes(int hs) { return (size_t)hs * sizeof(unsigned long long) + 16; }
inline int sample_table_slots(int B) {
  int hs = 1024;
  while (hs < 2 * B) hs <<= 1;
  return hs;
}

}  // namespace pa
