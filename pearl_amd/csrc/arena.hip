// Replay arena: structure-of-arrays ring in HBM + pinned staging ring for push,
// gather (parity mode, caller-drawn indices) and Philox sample (fast mode).
//
// Reference behaviour being replaced (file:line under /root/reference):
//   pearl/replay_buffers/tensor_based_replay_buffer.py:55-133   push
//   pearl/replay_buffers/tensor_based_replay_buffer.py:179-251  create_action_tensor_and_mask
//   pearl/replay_buffers/tensor_based_replay_buffer.py:253-282  sample
//   pearl/replay_buffers/tensor_based_replay_buffer.py:290-400  _create_transition_batch
//   pearl/replay_buffers/basic_replay_buffer.py:21-48           _store_transition
//   pearl/action_representation_modules/one_hot_action_representation_module.py:27-34
//
// All of it is byte movement: HBM-bound, coalesced row copies, no MFMA.
#include <stdarg.h>

#include <atomic>
#include <map>
#include <tuple>
#include <mutex>
#include <utility>

#include <new>

#include "common.hpp"
#include "sampler.hpp"

namespace pa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace pa

extern "C" const char* pa_last_error(void) { return pa::g_err; }
extern "C" int pa_abi_version(void) { return PA_ABI_VERSION; }
extern "C" int pa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

namespace pa {

// --------------------------------------------------------------------------
// device kernels
// --------------------------------------------------------------------------

struct StageLayout {
  int64_t row_bytes;
  int32_t off_state, off_action, off_reward, off_term, off_trunc, off_next_state, off_curr_avail,
      off_curr_mask, off_next_avail, off_next_mask, off_cost;
  int32_t state_bytes, action_bytes, reward_bytes, avail_bytes, mask_bytes;
  int32_t has_next_state, has_cost, has_avail;
};

__device__ __forceinline__ void copy_bytes_wave(uint8_t* __restrict__ dst,
                                                const uint8_t* __restrict__ src, int nbytes,
                                                int lane) {
  // nbytes and both pointers are multiples of 4 for the float columns; generic tail otherwise.
  if (((nbytes | (int)(uintptr_t)dst | (int)(uintptr_t)src) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = lane; i < (nbytes >> 4); i += 64) d4[i] = s4[i];
  } else if (((nbytes | (int)(uintptr_t)dst | (int)(uintptr_t)src) & 3) == 0) {
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
    for (int i = lane; i < (nbytes >> 2); i += 64) d1[i] = s1[i];
  } else {
    for (int i = lane; i < nbytes; i += 64) dst[i] = src[i];
  }
}

// Unpack `n` staged AoS rows into the SoA columns at slots (slot0 + r) % capacity.
__global__ __launch_bounds__(256) void scatter_rows_kernel(ArenaCols c, StageLayout L,
                                                           const uint8_t* __restrict__ stage,
                                                           int64_t n, int64_t slot0,
                                                           int64_t capacity) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  int64_t slot = slot0 + r;
  if (slot >= capacity) slot -= capacity;
  const uint8_t* row = stage + r * L.row_bytes;
  copy_bytes_wave(reinterpret_cast<uint8_t*>(c.state) + slot * L.state_bytes, row + L.off_state,
                  L.state_bytes, lane);
  if (L.has_next_state)
    copy_bytes_wave(reinterpret_cast<uint8_t*>(c.next_state) + slot * L.state_bytes,
                    row + L.off_next_state, L.state_bytes, lane);
  copy_bytes_wave(c.action + slot * L.action_bytes, row + L.off_action, L.action_bytes, lane);
  copy_bytes_wave(c.reward + slot * L.reward_bytes, row + L.off_reward, L.reward_bytes, lane);
  if (lane == 0) {
    c.terminated[slot] = row[L.off_term];
    c.truncated[slot] = row[L.off_trunc];
  }
  if (L.has_cost && lane == 1)
    c.cost[slot] = *reinterpret_cast<const float*>(row + L.off_cost);
  if (L.has_avail) {
    copy_bytes_wave(reinterpret_cast<uint8_t*>(c.curr_avail) + slot * L.avail_bytes,
                    row + L.off_curr_avail, L.avail_bytes, lane);
    copy_bytes_wave(reinterpret_cast<uint8_t*>(c.next_avail) + slot * L.avail_bytes,
                    row + L.off_next_avail, L.avail_bytes, lane);
    copy_bytes_wave(c.curr_mask + slot * L.mask_bytes, row + L.off_curr_mask, L.mask_bytes, lane);
    copy_bytes_wave(c.next_mask + slot * L.mask_bytes, row + L.off_next_mask, L.mask_bytes, lane);
  }
}

// Device-resident column ingest: src columns [n, ...] -> arena slots.
struct DevCols {
  const float* state;
  const uint8_t* action;
  const uint8_t* reward;
  const uint8_t* terminated;
  const uint8_t* truncated;
  const float* next_state;
  const float* curr_avail;
  const uint8_t* curr_mask;
  const float* next_avail;
  const uint8_t* next_mask;
  const float* cost;
  int32_t avail_bcast;
};

__global__ __launch_bounds__(256) void scatter_cols_kernel(ArenaCols c, StageLayout L, DevCols s,
                                                           int64_t n, int64_t slot0,
                                                           int64_t capacity) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  int64_t slot = (slot0 + r) % capacity;
  copy_bytes_wave(reinterpret_cast<uint8_t*>(c.state) + slot * L.state_bytes,
                  reinterpret_cast<const uint8_t*>(s.state) + r * L.state_bytes, L.state_bytes,
                  lane);
  if (L.has_next_state)
    copy_bytes_wave(reinterpret_cast<uint8_t*>(c.next_state) + slot * L.state_bytes,
                    reinterpret_cast<const uint8_t*>(s.next_state) + r * L.state_bytes,
                    L.state_bytes, lane);
  copy_bytes_wave(c.action + slot * L.action_bytes, s.action + r * L.action_bytes, L.action_bytes,
                  lane);
  copy_bytes_wave(c.reward + slot * L.reward_bytes, s.reward + r * L.reward_bytes, L.reward_bytes,
                  lane);
  if (lane == 0) {
    c.terminated[slot] = s.terminated[r] ? 1 : 0;
    c.truncated[slot] = s.truncated[r] ? 1 : 0;
  }
  if (L.has_cost && lane == 1) c.cost[slot] = s.cost[r];
  if (L.has_avail) {
    const int64_t rr = s.avail_bcast ? 0 : r;
    copy_bytes_wave(reinterpret_cast<uint8_t*>(c.curr_avail) + slot * L.avail_bytes,
                    reinterpret_cast<const uint8_t*>(s.curr_avail) + rr * L.avail_bytes,
                    L.avail_bytes, lane);
    copy_bytes_wave(reinterpret_cast<uint8_t*>(c.next_avail) + slot * L.avail_bytes,
                    reinterpret_cast<const uint8_t*>(s.next_avail) + rr * L.avail_bytes,
                    L.avail_bytes, lane);
    copy_bytes_wave(c.curr_mask + slot * L.mask_bytes, s.curr_mask + rr * L.mask_bytes,
                    L.mask_bytes, lane);
    copy_bytes_wave(c.next_mask + slot * L.mask_bytes, s.next_mask + rr * L.mask_bytes,
                    L.mask_bytes, lane);
  }
}

__device__ __forceinline__ int64_t load_index_value(const uint8_t* p, int dtype) {
  // x.long() of the one-hot module (one_hot_action_representation_module.py:30)
  switch (dtype) {
    case PA_I64: return *reinterpret_cast<const int64_t*>(p);
    case PA_I32: return *reinterpret_cast<const int32_t*>(p);
    case PA_U8: return *p;
    case PA_F64: return (int64_t)(*reinterpret_cast<const double*>(p));
    default: return (int64_t)(*reinterpret_cast<const float*>(p));
  }
}

struct GatherArgs {
  ArenaCols c;
  pa_batch_out o;
  const int64_t* idx;  // logical indices (0 = oldest)
  int64_t head, capacity;
  int32_t B, S, A, avail_dim, action_elems, action_dtype, reward_dtype;
  int32_t state_bytes, action_bytes, reward_bytes, avail_bytes, mask_bytes;
  // optional: published by the first wave as soon as the launch starts ("everything before this
  // launch on its stream is done": the call-start hand-off of pa_dqn_learn, one launch less)
  int* signal_flag; int signal_value;
};

// One wave per sampled transition: coalesced 16-byte copies of the state rows,
// the small columns ride on the first lanes.  Also emits the learner-side views
// (x = state || rep(action), rep(next_available_actions), float reward) so that
// preprocess_batch (policy_learner.py:197-218) costs no extra pass.
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs g) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g.signal_flag && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(g.signal_flag, g.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (b >= g.B) return;
  int64_t slot = g.head + g.idx[b];
  if (slot >= g.capacity) slot -= g.capacity;
  const uint8_t* st = reinterpret_cast<const uint8_t*>(g.c.state) + slot * g.state_bytes;
  // The two big rows of a transition travel TOGETHER: lanes 0-31 move the state row (into
  // o.state and / or the state part of x), lanes 32-63 the next_state row, 16 bytes per lane and
  // pass — one load instruction where there were two half-empty ones, twice the bytes in flight per
  // wave.  (x rows are 16-byte aligned when (S + rep_dim) % 4 == 0.)
  const int R0 = g.o.rep_dim;
  const bool pair = g.o.next_state && (g.o.state || g.o.x) && (g.state_bytes & 15) == 0 &&
                    (!g.o.x || ((g.S + R0) & 3) == 0);
  bool x_state_done = false;
  if (pair) {
    const int half = lane >> 5, l = lane & 31;
    const float4* src = reinterpret_cast<const float4*>(
        half ? reinterpret_cast<const uint8_t*>(g.c.next_state) + slot * g.state_bytes : st);
    float4* d0 = half ? reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(g.o.next_state) +
                                                  (int64_t)b * g.state_bytes)
                      : (g.o.state ? reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(g.o.state) +
                                                               (int64_t)b * g.state_bytes)
                                   : nullptr);
    float4* d1 = (!half && g.o.x) ? reinterpret_cast<float4*>(g.o.x + (int64_t)b * (g.S + R0)) : nullptr;
    for (int i = l; i < (g.state_bytes >> 4); i += 32) {
      const float4 v = src[i];
      if (d0) d0[i] = v;
      if (d1) d1[i] = v;
    }
    x_state_done = true;
  } else {
    if (g.o.state)
      copy_bytes_wave(reinterpret_cast<uint8_t*>(g.o.state) + (int64_t)b * g.state_bytes, st,
                      g.state_bytes, lane);
    if (g.o.next_state)
      copy_bytes_wave(reinterpret_cast<uint8_t*>(g.o.next_state) + (int64_t)b * g.state_bytes,
                      reinterpret_cast<const uint8_t*>(g.c.next_state) + slot * g.state_bytes,
                      g.state_bytes, lane);
  }
  const uint8_t* act = g.c.action + slot * g.action_bytes;
  if (g.o.action)
    copy_bytes_wave(reinterpret_cast<uint8_t*>(g.o.action) + (int64_t)b * g.action_bytes, act,
                    g.action_bytes, lane);
  const uint8_t* rw = g.c.reward + slot * g.reward_bytes;
  if (g.o.reward)
    copy_bytes_wave(reinterpret_cast<uint8_t*>(g.o.reward) + (int64_t)b * g.reward_bytes, rw,
                    g.reward_bytes, lane);
  if (lane == 0) {
    if (g.o.terminated) g.o.terminated[b] = g.c.terminated[slot];
    if (g.o.truncated) g.o.truncated[b] = g.c.truncated[slot];
    if (g.o.cost) g.o.cost[b] = g.c.cost[slot];
    if (g.o.reward_f32) {
      float r;
      if (g.reward_dtype == PA_F32) r = *reinterpret_cast<const float*>(rw);
      else if (g.reward_dtype == PA_F64) r = (float)*reinterpret_cast<const double*>(rw);
      else r = (float)load_index_value(rw, g.reward_dtype);
      g.o.reward_f32[b] = r;
    }
  }
  if (g.A > 0) {
    if (g.o.curr_avail)
      copy_bytes_wave(reinterpret_cast<uint8_t*>(g.o.curr_avail) + (int64_t)b * g.avail_bytes,
                      reinterpret_cast<const uint8_t*>(g.c.curr_avail) + slot * g.avail_bytes,
                      g.avail_bytes, lane);
    if (g.o.next_avail)
      copy_bytes_wave(reinterpret_cast<uint8_t*>(g.o.next_avail) + (int64_t)b * g.avail_bytes,
                      reinterpret_cast<const uint8_t*>(g.c.next_avail) + slot * g.avail_bytes,
                      g.avail_bytes, lane);
    if (g.o.curr_mask)
      copy_bytes_wave(g.o.curr_mask + (int64_t)b * g.mask_bytes,
                      g.c.curr_mask + slot * g.mask_bytes, g.mask_bytes, lane);
    if (g.o.next_mask)
      copy_bytes_wave(g.o.next_mask + (int64_t)b * g.mask_bytes,
                      g.c.next_mask + slot * g.mask_bytes, g.mask_bytes, lane);
  }
  // ---- fused preprocess views
  const int R = g.o.rep_dim;
  if (g.o.x) {
    float* xrow = g.o.x + (int64_t)b * (g.S + R);
    const float* srow = reinterpret_cast<const float*>(st);
    if (!x_state_done)
      for (int i = lane; i < g.S; i += 64) xrow[i] = srow[i];
    if (g.o.rep_onehot) {
      const int64_t a = load_index_value(act, g.action_dtype);
      for (int j = lane; j < R; j += 64) xrow[g.S + j] = (j == a) ? 1.0f : 0.0f;
    } else {
      // identity representation: the action itself, as float
      for (int j = lane; j < R; j += 64) {
        float v;
        const uint8_t* p = act + (int64_t)j * (g.action_bytes / g.action_elems);
        if (g.action_dtype == PA_F32) v = *reinterpret_cast<const float*>(p);
        else if (g.action_dtype == PA_F64) v = (float)*reinterpret_cast<const double*>(p);
        else v = (float)load_index_value(p, g.action_dtype);
        xrow[g.S + j] = v;
      }
    }
  }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float* rep_out = which ? g.o.curr_avail_rep : g.o.next_avail_rep;
    if (!rep_out || g.A <= 0) continue;
    const float* na = (which ? g.c.curr_avail : g.c.next_avail) + slot * (int64_t)(g.A * g.avail_dim);
    float* orow = rep_out + (int64_t)b * g.A * R;
    if (g.o.rep_onehot) {
      for (int e = lane; e < g.A * R; e += 64) {
        const int i = e / R, j = e - i * R;
        const int64_t a = (int64_t)na[i * g.avail_dim];
        orow[e] = (j == a) ? 1.0f : 0.0f;
      }
    } else {
      for (int e = lane; e < g.A * R; e += 64) orow[e] = na[e];
    }
  }
}

// ---- the same gather for launches that are HBM-bound (tens of thousands of rows) -----------------
// gather_kernel keeps ONE transition's bytes in flight per wave (its 1 KiB state || next_state load,
// then a dependent chain of small loads): right for the learn loop's 10 240-row window gathers, which
// are launch-bound, and 0.31-0.49 of the achievable 6.3 TB/s once a launch moves hundreds of MB
// (profiles/r06_d_gather_hbm_bound.txt).  Here a wave takes RPW = 4 transitions: their four indices by
// one load (lane r reads index r), their four state || next_state rows by four loads issued back to
// back before any store, and every small column by a 16-lane group per transition — all four rows'
// action / reward / flag / table loads leave in the same instructions.  Same bytes, same values.
template <int GW>
__device__ __forceinline__ void copy_bytes_group(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                                 int nbytes, int l) {
  if (((nbytes | (int)(uintptr_t)dst | (int)(uintptr_t)src) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = l; i < (nbytes >> 4); i += GW) d4[i] = s4[i];
  } else if (((nbytes | (int)(uintptr_t)dst | (int)(uintptr_t)src) & 3) == 0) {
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
    for (int i = l; i < (nbytes >> 2); i += GW) d1[i] = s1[i];
  } else {
    for (int i = l; i < nbytes; i += GW) dst[i] = src[i];
  }
}

template <int RPW>
__global__ __launch_bounds__(256) void gather_multi_kernel(GatherArgs g) {
  constexpr int GW = 64 / RPW;
  const int lane = threadIdx.x & 63;
  const int64_t b0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (g.signal_flag && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(g.signal_flag, g.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (b0 >= g.B) return;
  long long mine = -1;
  if (lane < RPW && b0 + lane < g.B) {
    mine = g.head + g.idx[b0 + lane];
    if (mine >= g.capacity) mine -= g.capacity;
  }
  long long slot[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) slot[r] = __shfl(mine, r);
  const int R0 = g.o.rep_dim;
  // ---- the big rows (the host takes this kernel only when gather_kernel's `pair` condition holds)
  {
    const int half = lane >> 5, l = lane & 31;
    const int n16 = g.state_bytes >> 4;
    for (int i = l; i < n16; i += 32) {
      // (rows past B — only in the launch's last wave — re-read row 0 of the wave and store nothing:
      //  no branch around a load; four named values, not an array: the compiler parked an array in LDS)
      const uint8_t* base = half ? reinterpret_cast<const uint8_t*>(g.c.next_state)
                                 : reinterpret_cast<const uint8_t*>(g.c.state);
      auto ld = [&](int r) {
        const long long sr = slot[r] < 0 ? slot[0] : slot[r];
        return reinterpret_cast<const float4*>(base + sr * g.state_bytes)[i];
      };
      auto st = [&](int r, const float4& v) {
        const int64_t b = b0 + r;
        if (slot[r] < 0) return;
        if (half) {
          reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(g.o.next_state) + b * g.state_bytes)[i] = v;
        } else {
          if (g.o.state)
            reinterpret_cast<float4*>(reinterpret_cast<uint8_t*>(g.o.state) + b * g.state_bytes)[i] = v;
          if (g.o.x) reinterpret_cast<float4*>(g.o.x + b * (g.S + R0))[i] = v;
        }
      };
      static_assert(RPW == 4, "four named rows");
      const float4 v0 = ld(0), v1 = ld(1), v2 = ld(2), v3 = ld(3);
      st(0, v0); st(1, v1); st(2, v2); st(3, v3);
    }
  }
  // ---- everything else: one 16-lane group per transition
  const int grp = lane / GW, l = lane - grp * GW;
  const int64_t b = b0 + grp;
  long long sl = -1;
#pragma unroll
  for (int r = 0; r < RPW; ++r) sl = (grp == r) ? slot[r] : sl;
  if (b >= g.B || sl < 0) return;
  const uint8_t* act = g.c.action + sl * g.action_bytes;
  if (g.o.action)
    copy_bytes_group<GW>(reinterpret_cast<uint8_t*>(g.o.action) + b * g.action_bytes, act, g.action_bytes, l);
  const uint8_t* rw = g.c.reward + sl * g.reward_bytes;
  if (g.o.reward)
    copy_bytes_group<GW>(reinterpret_cast<uint8_t*>(g.o.reward) + b * g.reward_bytes, rw, g.reward_bytes, l);
  if (l == 0) {
    if (g.o.terminated) g.o.terminated[b] = g.c.terminated[sl];
    if (g.o.truncated) g.o.truncated[b] = g.c.truncated[sl];
    if (g.o.cost) g.o.cost[b] = g.c.cost[sl];
    if (g.o.reward_f32) {
      float r;
      if (g.reward_dtype == PA_F32) r = *reinterpret_cast<const float*>(rw);
      else if (g.reward_dtype == PA_F64) r = (float)*reinterpret_cast<const double*>(rw);
      else r = (float)load_index_value(rw, g.reward_dtype);
      g.o.reward_f32[b] = r;
    }
  }
  if (g.A > 0) {
    if (g.o.curr_avail)
      copy_bytes_group<GW>(reinterpret_cast<uint8_t*>(g.o.curr_avail) + b * g.avail_bytes,
                           reinterpret_cast<const uint8_t*>(g.c.curr_avail) + sl * g.avail_bytes,
                           g.avail_bytes, l);
    if (g.o.next_avail)
      copy_bytes_group<GW>(reinterpret_cast<uint8_t*>(g.o.next_avail) + b * g.avail_bytes,
                           reinterpret_cast<const uint8_t*>(g.c.next_avail) + sl * g.avail_bytes,
                           g.avail_bytes, l);
    if (g.o.curr_mask)
      copy_bytes_group<GW>(g.o.curr_mask + b * g.mask_bytes, g.c.curr_mask + sl * g.mask_bytes,
                           g.mask_bytes, l);
    if (g.o.next_mask)
      copy_bytes_group<GW>(g.o.next_mask + b * g.mask_bytes, g.c.next_mask + sl * g.mask_bytes,
                           g.mask_bytes, l);
  }
  const int R = g.o.rep_dim;
  if (g.o.x) {
    float* xrow = g.o.x + b * (g.S + R);
    if (g.o.rep_onehot) {
      const int64_t a = load_index_value(act, g.action_dtype);
      for (int j = l; j < R; j += GW) xrow[g.S + j] = (j == a) ? 1.0f : 0.0f;
    } else {
      for (int j = l; j < R; j += GW) {
        float v;
        const uint8_t* p = act + (int64_t)j * (g.action_bytes / g.action_elems);
        if (g.action_dtype == PA_F32) v = *reinterpret_cast<const float*>(p);
        else if (g.action_dtype == PA_F64) v = (float)*reinterpret_cast<const double*>(p);
        else v = (float)load_index_value(p, g.action_dtype);
        xrow[g.S + j] = v;
      }
    }
  }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float* rep_out = which ? g.o.curr_avail_rep : g.o.next_avail_rep;
    if (!rep_out || g.A <= 0) continue;
    const float* na = (which ? g.c.curr_avail : g.c.next_avail) + sl * (int64_t)(g.A * g.avail_dim);
    float* orow = rep_out + b * g.A * R;
    if (g.o.rep_onehot) {
      for (int e = l; e < g.A * R; e += GW) {
        const int i = e / R, j = e - i * R;
        const int64_t a = (int64_t)na[i * g.avail_dim];
        orow[e] = (j == a) ? 1.0f : 0.0f;
      }
    } else {
      for (int e = l; e < g.A * R; e += GW) orow[e] = na[e];
    }
  }
}

// --------------------------------------------------------------------------
// Philox4x32-10 and the without-replacement index sampler: sampler.hpp
// Block r of the grid draws the sample of round r (Philox counter offset + r) into
// idx_out[r * B ...]: one launch covers every round of a learn() call.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(SAMPLE_THREADS) void sample_indices_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long table[];
  sample_indices_block(a, (int)blockIdx.x, table);
}

__global__ void one_hot_kernel(const uint8_t* __restrict__ idx, int dtype, int esize, int64_t n,
                               int ncls, float* __restrict__ out) {
  const int64_t total = n * ncls;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / ncls;
    const int j = (int)(e - r * ncls);
    const int64_t a = load_index_value(idx + r * esize, dtype);
    out[e] = (a == j) ? 1.0f : 0.0f;
  }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
static StageLayout make_layout(const pa_arena* a) {
  StageLayout L;
  memset(&L, 0, sizeof(L));
  const pa_arena_desc& d = a->d;
  L.state_bytes = d.state_dim * 4;
  L.action_bytes = d.action_elems * a->action_size;
  L.reward_bytes = a->reward_size;
  L.avail_bytes = d.max_actions * d.avail_dim * 4;
  L.mask_bytes = d.max_actions;
  L.has_next_state = d.has_next_state;
  L.has_cost = d.has_cost;
  L.has_avail = d.max_actions > 0;
  L.row_bytes = a->row_bytes;
  L.off_state = (int32_t)a->off_state;
  L.off_action = (int32_t)a->off_action;
  L.off_reward = (int32_t)a->off_reward;
  L.off_term = (int32_t)a->off_term;
  L.off_trunc = (int32_t)a->off_trunc;
  L.off_next_state = (int32_t)a->off_next_state;
  L.off_curr_avail = (int32_t)a->off_curr_avail;
  L.off_curr_mask = (int32_t)a->off_curr_mask;
  L.off_next_avail = (int32_t)a->off_next_avail;
  L.off_next_mask = (int32_t)a->off_next_mask;
  L.off_cost = (int32_t)a->off_cost;
  return L;
}

static int flush_impl(pa_arena* a, hipStream_t s) {
  if (a->staged == 0) return PA_OK;
  PA_HIP(hipSetDevice(a->d.device));
  // The pinned ring is reused: the previous H2D copy must have left it.
  if (a->stage_busy) {
    PA_HIP(hipEventSynchronize(a->stage_done));
    a->stage_busy = false;
  }
  PA_HIP(hipMemcpyAsync(a->stage_dev, a->stage_host, (size_t)(a->staged * a->row_bytes),
                        hipMemcpyHostToDevice, s));
  const StageLayout L = make_layout(a);
  const unsigned grid = (unsigned)ceil_div(a->staged, 4);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, s, a->c, L, a->stage_dev,
                     a->staged, a->stage_slot0, a->d.capacity);
  PA_LAUNCH_CHECK();
  PA_HIP(hipEventRecord(a->stage_done, s));
  a->stage_busy = true;
  a->ingest_stream = s;
  a->has_ingest = true;
  // The host may overwrite the pinned ring only after the copy is done; we wait
  // lazily (next push that needs the ring) via stage_done.
  a->staged = 0;
  return PA_OK;
}

int arena_wait_ingest(pa_arena* a, hipStream_t s) {
  if (a->has_ingest && a->ingest_stream != s) PA_HIP(hipStreamWaitEvent(s, a->stage_done, 0));
  return PA_OK;
}

// Reserve the next ring slot(s) exactly like deque(maxlen=capacity).append.
static void ring_advance(pa_arena* a, int64_t n) {
  const int64_t cap = a->d.capacity;
  if (a->size + n <= cap) {
    a->size += n;
  } else {
    const int64_t evict = a->size + n - cap;
    a->head = (a->head + evict) % cap;
    a->size = cap;
  }
}

static int wait_ring_free(pa_arena* a) {
  if (a->stage_busy) {
    PA_HIP(hipEventSynchronize(a->stage_done));
    a->stage_busy = false;
  }
  return PA_OK;
}

int arena_gather_device(pa_arena* a, const int64_t* idx_dev, int32_t B, const pa_batch_out* out,
                        hipStream_t s, int* signal_flag, int signal_value) {
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.signal_flag = signal_flag;
  g.signal_value = signal_value;
  g.c = a->c;
  g.o = *out;
  g.idx = idx_dev;
  g.head = a->head;
  g.capacity = a->d.capacity;
  g.B = B;
  g.S = a->d.state_dim;
  g.A = a->d.max_actions;
  g.avail_dim = a->d.avail_dim;
  g.action_elems = a->d.action_elems;
  g.action_dtype = a->d.action_dtype;
  g.reward_dtype = a->d.reward_dtype;
  g.state_bytes = a->d.state_dim * 4;
  g.action_bytes = a->d.action_elems * a->action_size;
  g.reward_bytes = a->reward_size;
  g.avail_bytes = a->d.max_actions * a->d.avail_dim * 4;
  g.mask_bytes = a->d.max_actions;
  if (!a->d.has_next_state) g.o.next_state = nullptr;
  if (!a->d.has_cost) g.o.cost = nullptr;
  if (out->x || out->next_avail_rep || out->curr_avail_rep) {
    PA_REQUIRE(out->rep_dim > 0, PA_ERR_INVALID, "gather: rep_dim must be > 0 for fused views");
    if (out->rep_onehot) {
      PA_REQUIRE(a->d.action_elems == 1, PA_ERR_UNSUPPORTED,
                 "one-hot representation needs scalar actions (action_elems=%d)",
                 a->d.action_elems);
      PA_REQUIRE(!(out->next_avail_rep || out->curr_avail_rep) || a->d.avail_dim == 1, PA_ERR_UNSUPPORTED,
                 "one-hot representation needs avail_dim == 1 (got %d)", a->d.avail_dim);
    } else {
      PA_REQUIRE(out->rep_dim == a->d.action_elems, PA_ERR_INVALID,
                 "identity representation: rep_dim %d != action_elems %d", out->rep_dim,
                 a->d.action_elems);
      PA_REQUIRE(!(out->next_avail_rep || out->curr_avail_rep) || a->d.avail_dim == out->rep_dim, PA_ERR_INVALID,
                 "identity representation: rep_dim %d != avail_dim %d", out->rep_dim,
                 a->d.avail_dim);
    }
  }
  if (B == 0) return PA_OK;
  // Launches that move enough bytes to be HBM-bound take four transitions per wave
  // (gather_multi_kernel); PEARL_AMD_GATHER_MULTI_MINB: the row count from which (default 32768,
  // 0 = never).  Needs gather_kernel's `pair` condition: both big rows wanted, whole 16-byte vectors.
  static const int64_t multi_minb = []() {
    const char* v = getenv("PEARL_AMD_GATHER_MULTI_MINB");
    return v && *v ? (int64_t)atoll(v) : (int64_t)32768;
  }();
  const bool pair = g.o.next_state && (g.o.state || g.o.x) && (g.state_bytes & 15) == 0 &&
                    (!g.o.x || ((g.S + g.o.rep_dim) & 3) == 0);
  if (multi_minb > 0 && B >= multi_minb && pair) {
    // (eight per wave was measured too: 2.4 / 2.7 TB/s against 3.2 / 4.1 — 127 VGPRs and spills)
    hipLaunchKernelGGL(gather_multi_kernel<4>, dim3((unsigned)ceil_div(B, 16)), dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, g);
  }
  PA_LAUNCH_CHECK();
  return PA_OK;
}

int sample_check(int64_t population, int32_t B) {
  PA_REQUIRE(B >= 0 && B <= SAMPLE_MAX_B, PA_ERR_UNSUPPORTED,
             "device sampler supports batch sizes up to %d (got %d)", SAMPLE_MAX_B, B);
  PA_REQUIRE(population > 0 && population < 0xFFFFFFFFll, PA_ERR_UNSUPPORTED,
             "device sampler supports populations below 2^32 (got %lld)", (long long)population);
  PA_REQUIRE((int64_t)B <= population, PA_ERR_VALUE,
             "Can't get a batch of size %d from a replay buffer with only %lld elements", B,
             (long long)population);
  return PA_OK;
}
SampleArgs sample_args(int64_t population, uint64_t seed, uint64_t offset, int32_t B,
                       int64_t* idx_out_dev) {
  SampleArgs sa;
  sa.idx_out = idx_out_dev;
  sa.n = (uint32_t)population;
  sa.B = B;
  sa.hs = sample_table_slots(B);
  sa.seed_lo = (uint32_t)seed;
  sa.seed_hi = (uint32_t)(seed >> 32);
  sa.offset0 = offset;
  return sa;
}

int sample_indices_launch(int64_t population, uint64_t seed, uint64_t offset, int32_t B,
                          int32_t rounds, int64_t* idx_out_dev, hipStream_t s) {
  {
    int rc = sample_check(population, B);
    if (rc != PA_OK) return rc;
  }
  if (B == 0 || rounds <= 0) return PA_OK;
  const SampleArgs sa = sample_args(population, seed, offset, B, idx_out_dev);
  hipLaunchKernelGGL(sample_indices_kernel, dim3((unsigned)rounds), dim3(SAMPLE_THREADS),
                     sample_smem_bytes(sa.hs), s, sa);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

int arena_sample(pa_arena* a, uint64_t seed, uint64_t offset, int32_t B, const pa_batch_out* out,
                 int64_t* idx_out_dev, hipStream_t s) {
  int rc = sample_indices_launch(a->size, seed, offset, B, 1, idx_out_dev, s);
  if (rc != PA_OK) return rc;
  return arena_gather_device(a, idx_out_dev, B, out, s);
}

}  // namespace pa

using namespace pa;

namespace pa {
// One process drives one GPU, for the life of the process: launch-time kernel attributes
// (max dynamic LDS, set once per kernel behind `static configured` guards) and the library's scratch
// buffers / tickets (row-step partials, split-K slabs, the solve's work space) are per-process
// statics that live on the device of the first handle.  The binding is therefore PERMANENT once
// taken — releasing it when the last handle is destroyed (as round 3 did) let a later handle bind
// another device and hand its kernels scratch pointers of the first one.  A process
// that wants another GPU is another process (torchrun's model).  Atomic: handles may be created
// from several threads.
static std::atomic<int> g_bound_device{-1};
static std::atomic<int> g_live_handles{0};
int bind_process_device(int device) {
  int expected = -1;
  if (!g_bound_device.compare_exchange_strong(expected, device) && expected != device) {
    set_error("pearl_amd: one process drives one GPU — this process is bound to HIP device %d "
              "(its first handle) and cannot create a handle on device %d; launch one process per "
              "device",
              expected, device);
    return PA_ERR_UNSUPPORTED;
  }
  g_live_handles.fetch_add(1);
  return PA_OK;
}
void release_process_device() { g_live_handles.fetch_sub(1); }

static std::atomic<int> g_dw_split_mode{-1};
int dw_split_mode() {
  int m = g_dw_split_mode.load();
  if (m < 0) {
    const char* v = getenv("PEARL_AMD_DW_SPLIT");      // 0: never, 1 (default): from PEARL_AMD_DW_MINB rows on,
    m = (v && v[0] == '0') ? 0 : ((v && v[0] == '2') ? 2 : 1);   // 2: every launch with 64-row tiles (tuning)
  }
  return m;
}
void set_dw_split_mode(int mode) { g_dw_split_mode.store(mode); }

static std::atomic<int> g_target_rows{0};
int target_rows_mode() {
  int m = g_target_rows.load();
  if (m == 0) {
    const char* v = getenv("PEARL_AMD_TARGET_ROWS");
    const int e = v ? atoi(v) : 0;
    m = (e == 32 || e == 64) ? e : -1;       // -1: per pass (TargetArgs::rows_hint)
  }
  return m < 0 ? 0 : m;
}
void set_target_rows_mode(int rows) { g_target_rows.store(rows); }

namespace {
struct ScratchBuf { float* p = nullptr; size_t floats = 0; };
std::mutex g_scratch_mu;
// keyed by (device, slot, stream): the null stream's handle is the same value on every device
// (ADVICE r5), and a buffer must live on the device whose launches use it.  Entries of destroyed
// streams stay until the process ends (a few KB each; a recycled handle value on the same device
// simply reuses the buffer, which only that stream's launches touch).
std::map<std::tuple<int, int, hipStream_t>, ScratchBuf> g_scratch;
}  // namespace
float* stream_scratch(int slot, hipStream_t s, size_t floats) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("stream_scratch: hipGetDevice failed"); return nullptr; }
  ScratchBuf& b = g_scratch[std::make_tuple(dev, slot, s)];
  if (floats <= b.floats) return b.p;
  if (b.p) {
    // only this stream's launches use the old buffer
    if (hipStreamSynchronize(s) != hipSuccess) { set_error("stream_scratch: stream sync failed"); return nullptr; }
    (void)hipFree(b.p);
    b.p = nullptr;
    b.floats = 0;
  }
  const size_t n = floats * 2 + 64;
  if (hipMalloc((void**)&b.p, n * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    set_error("stream_scratch: hipMalloc of %zu bytes failed", n * sizeof(float));
    b.p = nullptr;
    return nullptr;
  }
  // zeroed on the launch stream: ordered in front of the launch that asked for it
  if (hipMemsetAsync(b.p, 0, n * sizeof(float), s) != hipSuccess) {
    set_error("stream_scratch: memset failed");
    (void)hipFree(b.p);
    b.p = nullptr;
    return nullptr;
  }
  b.floats = n;
  return b.p;
}
}  // namespace pa

extern "C" int pa_arena_create(pa_arena** out, const pa_arena_desc* desc) {
  PA_REQUIRE(out && desc, PA_ERR_INVALID, "pa_arena_create: null argument");
  PA_REQUIRE(desc->capacity > 0, PA_ERR_INVALID, "capacity must be positive");
  PA_REQUIRE(desc->state_dim > 0, PA_ERR_INVALID, "state_dim must be positive");
  PA_REQUIRE(desc->action_elems > 0, PA_ERR_INVALID, "action_elems must be positive");
  PA_REQUIRE(dtype_size(desc->action_dtype) > 0 && dtype_size(desc->reward_dtype) > 0,
             PA_ERR_INVALID, "unknown action/reward dtype");
  PA_REQUIRE(desc->max_actions >= 0 && (desc->max_actions == 0 || desc->avail_dim > 0),
             PA_ERR_INVALID, "max_actions/avail_dim inconsistent");
  int ndev = pa_device_count();
  PA_REQUIRE(desc->device >= 0 && desc->device < ndev, PA_ERR_HIP,
             "HIP device %d not available (%d visible): the replay arena lives in HBM and has "
             "no CPU fallback",
             desc->device, ndev);
  PA_HIP(hipSetDevice(desc->device));
  pa_arena* a = new (std::nothrow) pa_arena();
  PA_REQUIRE(a, PA_ERR_NOMEM, "out of host memory");
  memset(a, 0, sizeof(*a));
  {
    // (after the handle exists: every later failure goes through pa_arena_destroy, which releases)
    int rc_dev = bind_process_device(desc->device);
    if (rc_dev != PA_OK) {
      delete a;
      return rc_dev;
    }
  }
  a->d = *desc;
  a->action_size = dtype_size(desc->action_dtype);
  a->reward_size = dtype_size(desc->reward_dtype);
  const int64_t N = desc->capacity;
  const int64_t S4 = (int64_t)desc->state_dim * 4;
  const int64_t AV = (int64_t)desc->max_actions * desc->avail_dim * 4;
#define PA_ALLOC(ptr, bytes)                                                 \
  do {                                                                       \
    void* _p = nullptr;                                                      \
    hipError_t _e = hipMalloc(&_p, (size_t)((bytes) > 0 ? (bytes) : 16));    \
    if (_e != hipSuccess) {                                                  \
      set_error("hipMalloc(%lld bytes) failed: %s", (long long)(bytes),      \
                hipGetErrorString(_e));                                      \
      pa_arena_destroy(a);                                                   \
      return PA_ERR_NOMEM;                                                   \
    }                                                                        \
    (ptr) = reinterpret_cast<decltype(ptr)>(_p);                             \
  } while (0)
  PA_ALLOC(a->c.state, N * S4);
  if (desc->has_next_state) PA_ALLOC(a->c.next_state, N * S4);
  PA_ALLOC(a->c.action, N * desc->action_elems * a->action_size);
  PA_ALLOC(a->c.reward, N * a->reward_size);
  PA_ALLOC(a->c.terminated, N);
  PA_ALLOC(a->c.truncated, N);
  if (desc->has_cost) PA_ALLOC(a->c.cost, N * 4);
  if (desc->max_actions > 0) {
    PA_ALLOC(a->c.curr_avail, N * AV);
    PA_ALLOC(a->c.next_avail, N * AV);
    PA_ALLOC(a->c.curr_mask, N * desc->max_actions);
    PA_ALLOC(a->c.next_mask, N * desc->max_actions);
  }
  // staged row layout: 16-byte aligned members so the unpack kernel vectorises
  int64_t off = 0;
  auto place = [&](int64_t bytes) {
    int64_t o = off;
    off = round_up(off + bytes, 16);
    return o;
  };
  a->off_state = place(S4);
  a->off_next_state = place(desc->has_next_state ? S4 : 0);
  a->off_action = place((int64_t)desc->action_elems * a->action_size);
  a->off_reward = place(a->reward_size);
  a->off_cost = place(4);
  a->off_term = place(1);
  a->off_trunc = a->off_term + 1;
  a->off_curr_avail = place(AV);
  a->off_next_avail = place(AV);
  a->off_curr_mask = place(desc->max_actions);
  a->off_next_mask = place(desc->max_actions);
  a->row_bytes = off;
  a->stage_rows = desc->staging_rows > 0 ? desc->staging_rows : 4096;
  if (a->stage_rows > N) a->stage_rows = N;
  {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)(a->stage_rows * a->row_bytes), hipHostMallocDefault);
    if (e != hipSuccess) {
      set_error("hipHostMalloc(staging) failed: %s", hipGetErrorString(e));
      pa_arena_destroy(a);
      return PA_ERR_NOMEM;
    }
    a->stage_host = reinterpret_cast<uint8_t*>(p);
  }
  PA_ALLOC(a->stage_dev, a->stage_rows * a->row_bytes);
#undef PA_ALLOC
  if (hipEventCreateWithFlags(&a->stage_done, hipEventDisableTiming) != hipSuccess) {
    set_error("hipEventCreate failed");
    pa_arena_destroy(a);
    return PA_ERR_HIP;
  }
  if (desc->max_actions > 0) {
    a->sh_next_avail = static_cast<float*>(calloc((size_t)desc->max_actions * desc->avail_dim, 4));
    a->sh_next_mask = static_cast<uint8_t*>(calloc((size_t)desc->max_actions, 1));
    if (!a->sh_next_avail || !a->sh_next_mask) {
      set_error("out of host memory");
      pa_arena_destroy(a);
      return PA_ERR_NOMEM;
    }
  }
  *out = a;
  return PA_OK;
}

extern "C" int pa_arena_destroy(pa_arena* a) {
  if (!a) return PA_OK;
  (void)hipSetDevice(a->d.device);
  (void)hipDeviceSynchronize();
  void* ptrs[] = {a->c.state, a->c.next_state, a->c.action, a->c.reward, a->c.terminated,
                  a->c.truncated, a->c.cost, a->c.curr_avail, a->c.next_avail, a->c.curr_mask,
                  a->c.next_mask, a->stage_dev};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (a->stage_host) (void)hipHostFree(a->stage_host);
  if (a->stage_done) (void)hipEventDestroy(a->stage_done);
  free(a->sh_next_avail);
  free(a->sh_next_mask);
  delete a;
  release_process_device();
  return PA_OK;
}

extern "C" int64_t pa_arena_len(const pa_arena* a) { return a ? a->size : 0; }
extern "C" int64_t pa_arena_capacity(const pa_arena* a) { return a ? a->d.capacity : 0; }
extern "C" int64_t pa_arena_head(const pa_arena* a) { return a ? a->head : 0; }

extern "C" int pa_arena_clear(pa_arena* a) {
  PA_REQUIRE(a, PA_ERR_INVALID, "null arena");
  a->head = 0;
  a->size = 0;
  a->staged = 0;
  a->shared_next = 0;
  return PA_OK;
}

// Track whether every stored row shares one next-action table (host pointers).
static void note_next_table(pa_arena* a, const float* next_avail, const uint8_t* next_mask) {
  const pa_arena_desc& d = a->d;
  if (d.max_actions <= 0 || a->shared_next == 2) return;
  const size_t av = (size_t)d.max_actions * d.avail_dim * 4, mk = (size_t)d.max_actions;
  if (a->shared_next == 0) {
    memcpy(a->sh_next_avail, next_avail, av);
    memcpy(a->sh_next_mask, next_mask, mk);
    a->shared_next = 1;
    // a PROCESS-wide generation: a buffer destroyed and re-created at the same heap address with
    // another static action table must not hit a learner's (arena pointer, generation) cache of
    // the old one
    static std::atomic<int> next_gen{0};
    a->shared_gen = next_gen.fetch_add(1) + 1;
  } else if (memcmp(a->sh_next_avail, next_avail, av) != 0 ||
             memcmp(a->sh_next_mask, next_mask, mk) != 0) {
    a->shared_next = 2;
  }
}

extern "C" int32_t pa_arena_shared_next_table(const pa_arena* a) {
  return (a && a->shared_next == 1) ? 1 : 0;
}

extern "C" int pa_arena_push(pa_arena* a, const pa_transition* t) {
  PA_REQUIRE(a && t, PA_ERR_INVALID, "pa_arena_push: null argument");
  PA_REQUIRE(t->state && t->action && t->reward, PA_ERR_INVALID,
             "pa_arena_push: state/action/reward are mandatory");
  const pa_arena_desc& d = a->d;
  PA_REQUIRE(!d.has_next_state || t->next_state, PA_ERR_INVALID, "next_state missing");
  PA_REQUIRE(d.max_actions == 0 || (t->curr_avail && t->curr_mask && t->next_avail && t->next_mask),
             PA_ERR_INVALID, "available-action tables missing");
  if (a->staged == a->stage_rows) {
    int rc = flush_impl(a, nullptr);
    if (rc != PA_OK) return rc;
  }
  if (a->staged == 0) {
    int rc = wait_ring_free(a);
    if (rc != PA_OK) return rc;
    a->stage_slot0 = (a->head + a->size) % d.capacity;
  }
  uint8_t* row = a->stage_host + a->staged * a->row_bytes;
  memcpy(row + a->off_state, t->state, (size_t)d.state_dim * 4);
  if (d.has_next_state) memcpy(row + a->off_next_state, t->next_state, (size_t)d.state_dim * 4);
  memcpy(row + a->off_action, t->action, (size_t)d.action_elems * a->action_size);
  memcpy(row + a->off_reward, t->reward, (size_t)a->reward_size);
  row[a->off_term] = t->terminated ? 1 : 0;
  row[a->off_trunc] = t->truncated ? 1 : 0;
  if (d.has_cost) {
    float c = t->cost ? *t->cost : 0.0f;
    memcpy(row + a->off_cost, &c, 4);
  }
  if (d.max_actions > 0) {
    const size_t av = (size_t)d.max_actions * d.avail_dim * 4;
    memcpy(row + a->off_curr_avail, t->curr_avail, av);
    memcpy(row + a->off_next_avail, t->next_avail, av);
    memcpy(row + a->off_curr_mask, t->curr_mask, (size_t)d.max_actions);
    memcpy(row + a->off_next_mask, t->next_mask, (size_t)d.max_actions);
    note_next_table(a, t->next_avail, t->next_mask);
  }
  a->staged += 1;
  ring_advance(a, 1);
  return PA_OK;
}

extern "C" int pa_arena_push_many(pa_arena* a, int64_t n, const pa_columns* cols) {
  PA_REQUIRE(a && cols && n >= 0, PA_ERR_INVALID, "pa_arena_push_many: bad argument");
  const pa_arena_desc& d = a->d;
  const size_t S4 = (size_t)d.state_dim * 4;
  const size_t AB = (size_t)d.action_elems * a->action_size;
  const size_t AV = (size_t)d.max_actions * d.avail_dim * 4;
  const uint8_t* act = reinterpret_cast<const uint8_t*>(cols->action);
  const uint8_t* rew = reinterpret_cast<const uint8_t*>(cols->reward);
  for (int64_t r = 0; r < n; ++r) {
    pa_transition t;
    memset(&t, 0, sizeof(t));
    const int64_t rr = cols->avail_bcast ? 0 : r;
    t.state = cols->state + r * d.state_dim;
    t.action = act + r * AB;
    t.reward = rew + r * a->reward_size;
    t.terminated = cols->terminated[r];
    t.truncated = cols->truncated[r];
    if (d.has_next_state) t.next_state = cols->next_state + r * d.state_dim;
    if (d.has_cost && cols->cost) t.cost = cols->cost + r;
    if (d.max_actions > 0) {
      t.curr_avail = cols->curr_avail + rr * (AV / 4);
      t.next_avail = cols->next_avail + rr * (AV / 4);
      t.curr_mask = cols->curr_mask + rr * d.max_actions;
      t.next_mask = cols->next_mask + rr * d.max_actions;
    }
    (void)S4;
    int rc = pa_arena_push(a, &t);
    if (rc != PA_OK) return rc;
  }
  return PA_OK;
}

extern "C" int pa_arena_push_many_device(pa_arena* a, int64_t n, const pa_columns* cols,
                                         void* stream) {
  PA_REQUIRE(a && cols && n >= 0, PA_ERR_INVALID, "pa_arena_push_many_device: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(a->d.device));
  int rc = flush_impl(a, s);
  if (rc != PA_OK) return rc;
  const pa_arena_desc& d = a->d;
  if (n == 0) return PA_OK;
  DevCols sc;
  memset(&sc, 0, sizeof(sc));
  // only the last `capacity` rows survive a FIFO of that capacity
  int64_t skip = n > d.capacity ? n - d.capacity : 0;
  const int64_t m = n - skip;
  sc.state = cols->state + skip * d.state_dim;
  sc.action = reinterpret_cast<const uint8_t*>(cols->action) + skip * d.action_elems * a->action_size;
  sc.reward = reinterpret_cast<const uint8_t*>(cols->reward) + skip * a->reward_size;
  sc.terminated = cols->terminated + skip;
  sc.truncated = cols->truncated + skip;
  if (d.has_next_state) sc.next_state = cols->next_state + skip * d.state_dim;
  if (d.has_cost) sc.cost = cols->cost + skip;
  sc.avail_bcast = cols->avail_bcast;
  if (d.max_actions > 0) {
    const int64_t k = cols->avail_bcast ? 0 : skip;
    sc.curr_avail = cols->curr_avail + k * d.max_actions * d.avail_dim;
    sc.next_avail = cols->next_avail + k * d.max_actions * d.avail_dim;
    sc.curr_mask = cols->curr_mask + k * d.max_actions;
    sc.next_mask = cols->next_mask + k * d.max_actions;
  }
  if (d.max_actions > 0 && a->shared_next != 2) {
    if (cols->avail_bcast) {
      // one table for the whole ingest: fetch its 80-odd bytes and compare / adopt (a bulk
      // ingest can afford one small synchronous copy)
      const size_t av = (size_t)d.max_actions * d.avail_dim * 4;
      float* tab = static_cast<float*>(malloc(av));
      uint8_t* msk = static_cast<uint8_t*>(malloc((size_t)d.max_actions));
      bool ok = tab && msk;
      // (on the CALLER's stream, then a wait for it: the tables were uploaded on that stream, and
      //  a null-stream copy is not ordered behind work of a non-blocking stream — ADVICE r3)
      if (ok) ok = hipMemcpyAsync(tab, cols->next_avail, av, hipMemcpyDeviceToHost, s) == hipSuccess &&
                   hipMemcpyAsync(msk, cols->next_mask, (size_t)d.max_actions, hipMemcpyDeviceToHost,
                                  s) == hipSuccess &&
                   hipStreamSynchronize(s) == hipSuccess;
      if (ok) note_next_table(a, tab, msk);
      else a->shared_next = 2;
      free(tab);
      free(msk);
    } else {
      a->shared_next = 2;   // per-row tables on the device: not inspected
    }
  }
  // account for the skipped rows exactly as successive appends would
  ring_advance(a, skip);
  const int64_t slot0 = (a->head + a->size) % d.capacity;
  const StageLayout L = make_layout(a);
  hipLaunchKernelGGL(scatter_cols_kernel, dim3((unsigned)ceil_div(m, 4)), dim3(256), 0, s, a->c, L,
                     sc, m, slot0, d.capacity);
  PA_LAUNCH_CHECK();
  PA_HIP(hipEventRecord(a->stage_done, s));
  a->ingest_stream = s;
  a->has_ingest = true;
  ring_advance(a, m);
  return PA_OK;
}

extern "C" int pa_arena_flush(pa_arena* a, void* stream) {
  PA_REQUIRE(a, PA_ERR_INVALID, "null arena");
  PA_HIP(hipSetDevice(a->d.device));
  int rc = flush_impl(a, reinterpret_cast<hipStream_t>(stream));
  if (rc != PA_OK) return rc;
  return arena_wait_ingest(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_arena_gather_device(pa_arena* a, const int64_t* logical_idx_dev, int32_t B,
                                      const pa_batch_out* out, void* stream) {
  PA_REQUIRE(a && out && (logical_idx_dev || B == 0), PA_ERR_INVALID, "pa_arena_gather: null argument");
  PA_REQUIRE(B >= 0, PA_ERR_INVALID, "negative batch size");
  PA_REQUIRE((int64_t)B <= a->size, PA_ERR_VALUE,
             "Can't get a batch of size %d from a replay buffer with only %lld elements", B,
             (long long)a->size);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(a->d.device));
  int rc = flush_impl(a, s);
  if (rc != PA_OK) return rc;
  rc = arena_wait_ingest(a, s);
  if (rc != PA_OK) return rc;
  return arena_gather_device(a, logical_idx_dev, B, out, s);
}

extern "C" int pa_arena_gather(pa_arena* a, const int64_t* logical_idx_host, int32_t B,
                               const pa_batch_out* out, int64_t* idx_dev_scratch, void* stream) {
  PA_REQUIRE(a && out, PA_ERR_INVALID, "pa_arena_gather: null argument");
  PA_REQUIRE(B >= 0, PA_ERR_INVALID, "negative batch size");
  PA_REQUIRE((int64_t)B <= a->size, PA_ERR_VALUE,
             "Can't get a batch of size %d from a replay buffer with only %lld elements", B,
             (long long)a->size);
  if (B == 0) return PA_OK;
  PA_REQUIRE(logical_idx_host && idx_dev_scratch, PA_ERR_INVALID, "null index buffers");
  for (int32_t i = 0; i < B; ++i)
    PA_REQUIRE(logical_idx_host[i] >= 0 && logical_idx_host[i] < a->size, PA_ERR_INVALID,
               "index %lld out of range [0, %lld)", (long long)logical_idx_host[i],
               (long long)a->size);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(a->d.device));
  // pageable source: the copy is staged by the runtime before the call returns
  PA_HIP(hipMemcpyAsync(idx_dev_scratch, logical_idx_host, (size_t)B * 8, hipMemcpyHostToDevice, s));
  return pa_arena_gather_device(a, idx_dev_scratch, B, out, stream);
}

extern "C" int pa_arena_sample(pa_arena* a, uint64_t seed, uint64_t offset, int32_t B,
                               const pa_batch_out* out, int64_t* idx_out_dev, void* stream) {
  PA_REQUIRE(a && out && idx_out_dev, PA_ERR_INVALID, "pa_arena_sample: null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(a->d.device));
  int rc = flush_impl(a, s);
  if (rc != PA_OK) return rc;
  rc = arena_wait_ingest(a, s);
  if (rc != PA_OK) return rc;
  return arena_sample(a, seed, offset, B, out, idx_out_dev, s);
}

extern "C" int pa_sample_indices(int64_t population, uint64_t seed, uint64_t offset, int32_t B,
                                 int64_t* idx_out_dev, int32_t device, void* stream) {
  PA_REQUIRE(idx_out_dev || B == 0, PA_ERR_INVALID, "null output");
  PA_HIP(hipSetDevice(device));
  return sample_indices_launch(population, seed, offset, B, 1, idx_out_dev,
                               reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_sample_indices_rounds(int64_t population, uint64_t seed, uint64_t offset0,
                                        int32_t B, int32_t rounds, int64_t* idx_out_dev,
                                        int32_t device, void* stream) {
  PA_REQUIRE(idx_out_dev || B == 0 || rounds == 0, PA_ERR_INVALID, "null output");
  PA_REQUIRE(rounds >= 0, PA_ERR_INVALID, "negative rounds");
  PA_HIP(hipSetDevice(device));
  return sample_indices_launch(population, seed, offset0, B, rounds, idx_out_dev,
                               reinterpret_cast<hipStream_t>(stream));
}

// out[b] = src[idx[b]] for rows of row_bytes bytes (extra per-transition columns kept next to the
// arena in logical order, e.g. PPO's gae / lam_return / action_probs, ppo.py:47-82)
static __global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ src,
                                                                 int row_bytes,
                                                                 const int64_t* __restrict__ idx,
                                                                 int B, uint8_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  copy_bytes_wave(out + (int64_t)b * row_bytes, src + idx[b] * row_bytes, row_bytes, lane);
}

extern "C" int pa_gather_rows(const void* src_dev, int32_t row_bytes, const int64_t* idx_dev,
                              int32_t B, void* out_dev, void* stream) {
  PA_REQUIRE(src_dev && idx_dev && out_dev && row_bytes > 0 && B >= 0, PA_ERR_INVALID,
             "pa_gather_rows: bad argument");
  if (B == 0) return PA_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint8_t*>(src_dev),
                     row_bytes, idx_dev, B, reinterpret_cast<uint8_t*>(out_dev));
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// out[p][b] = src[p][idx[b]] for `planes` float vectors of one length laid out `plane_stride`
// floats apart: the three per-transition columns of PPOTransition in ONE launch (a launch each
// cost 4.5 us apiece in a 190 us PPO step)
static __global__ __launch_bounds__(256) void gather_planes_kernel(const float* __restrict__ src,
                                                                   int64_t plane_stride, int planes,
                                                                   const int64_t* __restrict__ idx,
                                                                   int B, float* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int64_t i = idx[b];
  for (int p = 0; p < planes; ++p) out[(int64_t)p * B + b] = src[p * plane_stride + i];
}

extern "C" int pa_gather_planes(const float* src_dev, int64_t plane_stride, int32_t planes,
                                const int64_t* idx_dev, int32_t B, float* out_dev, void* stream) {
  PA_REQUIRE(src_dev && idx_dev && out_dev && planes > 0 && plane_stride >= 0 && B >= 0,
             PA_ERR_INVALID, "pa_gather_planes: bad argument");
  if (B == 0) return PA_OK;
  hipLaunchKernelGGL(gather_planes_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src_dev, plane_stride, planes, idx_dev, B,
                     out_dev);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

extern "C" int pa_one_hot(const void* idx_dev, int32_t idx_dtype, int64_t n, int32_t num_classes,
                          float* out_dev, void* stream) {
  PA_REQUIRE(num_classes > 0 && n >= 0, PA_ERR_INVALID, "bad one-hot shape");
  const int es = dtype_size(idx_dtype);
  PA_REQUIRE(es > 0, PA_ERR_INVALID, "bad index dtype");
  if (n == 0) return PA_OK;
  PA_REQUIRE(idx_dev && out_dev, PA_ERR_INVALID, "null pointer");
  const int64_t total = n * num_classes;
  unsigned grid = (unsigned)(ceil_div(total, 256) > 2048 ? 2048 : ceil_div(total, 256));
  hipLaunchKernelGGL(one_hot_kernel, dim3(grid), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint8_t*>(idx_dev), idx_dtype, es, n, num_classes,
                     out_dev);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
