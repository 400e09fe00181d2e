// Host orchestration of the DQN learner step: workspaces, launch sequence,
// the fused learn() loop (all rounds' index lists drawn by one launch, then a
// single-stream chain of kernels per step with no events), and
// HIP-event kernel timers for bench.py's roofline block.
//
// Reference call stack being replaced:
//   PolicyLearner.learn            pearl/policy_learners/policy_learner.py:162-195
//   DeepTDLearning.learn_batch     .../sequential_decision_making/deep_td_learning.py:333-360
//   DeepTDLearning.forward / loss  deep_td_learning.py:269-331
//   DeepQLearning.get_next_state_values  deep_q_learning.py:130-167
#include <math.h>
#include <stdlib.h>

#include <deque>
#include <new>
#include <string>
#include <vector>

#include "online_kernels.hpp"
#include "online_pair_kernel.hpp"
#include "online_f16_kernel.hpp"
#include "target_pp_kernel.hpp"
#include "host_launch.hpp"
#include "sampler.hpp"

using namespace pa;

namespace {

struct Timer {
  std::string name;
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  int64_t units = 0;  // transitions covered by the timed launches
};

constexpr size_t kMaxTimedPairs = 8192;
constexpr int kTileCtrs = 1024;

}  // namespace

struct pa_dqn {
  pa_dqn_desc d;
  pa_dqn_buffers bufs;
  bool bound;
  int64_t P;         // flat parameter count (with alignment gaps)
  int64_t off[6];    // W1,b1,W2,b2,W3,b3
  int IN;            // S + AD
  // workspaces (HBM)
  float *H1a, *H2a, *dZ2, *dZ1, *y, *nextv, *qbuf, *dq, *absd, *xpack, *loss_scratch;
  float* w2f;  // fragment-major copy of the target net's W2 (target_fused_kernel's weight operand)
  void* w2sp;  // the same matrix as bf16 split planes (target_split_kernel), H1 = H2 = 256 only
  void* w2sp_online;  // Double DQN: split planes of the ONLINE W2 (rebuilt with w2f_online every round)
  void* w1sp;  // the target W1's state columns as split planes: the split tile forms U itself
  int fuse_u;  // PEARL_AMD_FUSE_U (default 0): the split tiles form U themselves (no first-layer GEMM
               // launch in front of the leading pieces).  Bit-identical, all tests green, measured
               // SLOWER as built: the in-tile product puts a second exposed memory latency, two
               // barriers and 64 lane shuffles in front of the tile's own operand loads (leading
               // 1-round piece 26 us against 8 + 15.5 with the GEMM launch; 27.2 vs 28.6 M
               // transitions/s over 2000 rounds) — kept as a switch for the version that issues the
               // tile's loads ahead of the product.
  int use_split;  // PEARL_AMD_TARGET_SPLIT (default 1): the bf16x3 kernel where its shape applies
  int use_h2t;    // PEARL_AMD_TARGET_H2 (default 1): the fp16x2 tile (target_h2_kernel.hpp) for plain DQN's passes
  void* w2h;      // target W2 as scaled fp16 hi / lo planes
  int* w2hf;      // ... and the scale field of every unit
  int dw_tm;      // PEARL_AMD_DW_TM: rows per weight-gradient tile, 64 or 32 (0 = by the CU partition)
  int rp_split;   // PEARL_AMD_ROWPASS_SPLIT (default 1): window-first row pass as forward + backward
  // Double DQN only (desc.double_q): the same copy of the ONLINE W2, the chosen next actions, and
  // the row-per-transition value pass through the target network
  float* w2f_online;
  int* choice;
  float* choice_rep;   // [max_batch][AD] representation of the chosen next action
  float *W1f, *W2f16, *W2tf;  // fragment-major copies of the online weights (online_rowpass_kernel)
  // the row pass on the fp16 matrix pipe (online_f16_kernel.hpp): max |w| per unit of the online weights,
  // two buffers of [H1 | H2 | H1] words (the optimizer launch fills the one the next row pass reads)
  int rp_h2;           // PEARL_AMD_ROWPASS_H2 (default 1; 0 = the fp32-MFMA row pass)
  unsigned* umax;
  int umax_cur;        // which buffer holds the maxima of the parameters as they are now
  // the row pass on two workgroups per 16-row tile (online_pair_kernel.hpp; PEARL_AMD_ROWPASS_PAIR=1,
  // the benchmark's shape only; off by default — DESIGN.md §3.10 has the measurements): the halves'
  // interleaved partials of dZ1, the pair's tagged exchange words, the halves' q partials of a
  // forward / backward launch pair
  int pair;            // 0: off
  int pair_lds;        // dynamic LDS bytes asked for per workgroup (> 80 KB: one workgroup per CU)
  bool pair_live;      // the last row pass wrote dZ1 as two interleaved partials (dZ1p)
  bool qx_clean;       // every exchange word holds kYPendingBits
  float* dZ1p;
  unsigned* qx;
  int qx_rows;
  float* qhalf;
  // fused learn(): gathered batch (single stream, no events) + index lists of all rounds
  // learn(): the target-network side of a window (gather -> U -> Bellman targets y) runs on a
  // low-priority side stream into one of TWO buffer sets, while the main stream runs the
  // sequential online chain; x is only touched by the main stream and needs one copy.
  struct BatchBuf {
    float* next_state;
    float* next_avail_rep;
    uint8_t* next_mask;
    float* reward;
    uint8_t* term;
  } bb[2];
  float* bb_x;       // 2 x [wrows][IN] state || rep(action) of a window (double-buffered like bb)
  int bb_A;          // A the batch buffers were sized for
  float* Uw[2];      // [wrows][H1] per buffer set
  float* yw[2];      // [wrows] Bellman targets, data-tagged (kYPending = not yet produced)
  hipStream_t side;  // target-network stream of learn()
  hipEvent_t ev_start, ev_tail, ev_chain[2];
  hipEvent_t ev_gather[2];  // side stream: window inputs (incl. x) gathered into buffer set p
  // Cross-stream hand-offs inside learn() go through a device word, not through events: a queue
  // that is blocked on another queue's event wakes up 12-17 us after the event fires on this
  // stack (measured: the target pass of every window started that long after the soft update it
  // waits for), a one-wave kernel spinning on a word written by a one-wave kernel takes ~3 us.
  int* sig;          // [4] generation word (monotonic, never reset)
  int sig_gen;       // last generation handed out
  int pending_signal;  // generation the NEXT row-pass launch publishes when it starts (0: none)
  int pending_wait;    // generation of sig[1] the NEXT row-pass launch waits for before it reads x (0: none):
                       // the side stream's gather of this window's x has completed (published by the
                       // wait_flag_kernel launch behind that gather).  Replaces a hipStreamWaitEvent on the
                       // learner stream, which costs ~6 us of idle stream per window even when the event
                       // completed long ago (rocprof: gap in front of every window's first row pass)
  int use_flags;     // PEARL_AMD_FLAG_HOP (default 1); 0 = events as before
  int lead_persist;  // PEARL_AMD_LEAD_PERSIST: leading target pieces keep off the chain's CUs (1: all of them;
                     // 2, the default since round 6: all but a window's first — the first weight-gradient
                     // launch of a window no longer queues for slots behind the second piece's grid)
  // the fragment-major copies (online + target) match the flat parameters: true after a learn()
  // whose every round refreshed them in its optimizer epilogue; cleared by anything else that
  // writes parameters (bind, step, apply, update_target, pa_dqn_invalidate)
  bool packed_ok;
  bool online_tgt_ok;   // Double DQN: w2f_online / w2sp_online hold the current online W2 (kept by the
                        // optimizer epilogue of the previous round of this call / step)
  int* err_dev;      // device error word (a bounded wait expired)
  int* err_host;     // pinned, device-mapped twin: a kernel that sets err_dev sets it too (no copy)
  int overlap;       // 0: single-stream learn loop (PEARL_AMD_OVERLAP=0 or timing level >= 2)
  int split_first;   // rounds of a window whose target pass is issued first, as its own launch
  bool y_clean;      // both yw buffers hold kYPendingBits everywhere (tagged hand-off invariant)
  // CU partition of the overlapped loop: the persistent target kernel stays off `n_reserved` CUs
  uint8_t* reserved_dev;   // [kCuKeys] or null (no partition)
  int n_reserved, ncu;
  // the arena's shared next-action table as the target kernel reads it: rep(table) [A][AD] and the
  // mask [A], rebuilt on the host when (arena, generation, representation) change
  float* sh_rep;           // device
  uint8_t* sh_mask;        // device
  uint8_t* sh_stage;       // pinned host staging for both
  const pa_arena* sh_arena;
  int sh_gen, sh_onehot, sh_A, sh_enable;
  int* tile_ctr;           // [kTileCtrs] work-stealing counters, one per persistent launch
  int* dbg_workers;        // PEARL_AMD_DEBUG_WORKERS: [2] device, participating workgroups of persistent launches
  int64_t dbg_launches;
  int ctr_next;
  int pingpong;            // PEARL_AMD_PINGPONG: 1 (default) target_pp_kernel for the persistent
                           // launches of learn(), 2 for every launch, 0 never
  long long *prof_row, *prof_dw;  // phase-stamp buffers (pa_debug_set_prof), normally null
  long long* prof_tgt;            // target kernel stamps [tile][8][16] (pa_debug_set_prof_target)
  int prof_tgt_tiles;
  int prof_round, cur_round;      // stamp only this round of a learn() call (-1: every launch)
  int wcap;          // rounds whose target-network pass is batched into one launch (learn())
  int64_t wrows;     // wcap * max_batch: rows of the window-sized workspaces (U, y, batch buffers)
  int64_t* idx_all;  // [idx_cap] logical indices, round-major
  int64_t idx_cap;
  int64_t tick;      // steps seen while timing (sparse sampling of the level-1 timer)
  // timing
  int timing;  // 0 off, 1 dominant kernel only, 2 every stage
  int timing_chain;  // level | 4: also the online chain (row pass, weight gradients) of ONE mid-window
                     // round of every sampled window (roofline.chain of the bench line)
  bool chain_sample; // the chain launches being enqueued belong to such a round
  std::deque<Timer> timers;  // deque: references stay valid while nested timers are added
};

namespace {

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }

void param_layout(int S, int AD, int H1, int H2, int64_t off[6], int64_t* total) {
  int64_t o = 0;
  off[0] = o; o = align4(o + (int64_t)H1 * (S + AD));
  off[1] = o; o = align4(o + H1);
  off[2] = o; o = align4(o + (int64_t)H2 * H1);
  off[3] = o; o = align4(o + H2);
  off[4] = o; o = align4(o + H2);
  off[5] = o; o = align4(o + 1);
  *total = o;
}

Timer* find_timer(pa_dqn* h, const char* name) {
  for (auto& t : h->timers)
    if (t.name == name) return &t;
  h->timers.push_back(Timer());
  h->timers.back().name = name;
  return &h->timers.back();
}

struct ScopedTimer {
  pa_dqn* h;
  Timer* t;
  hipStream_t s;
  bool active;
  // every: record only on every `every`-th step (a hipEventRecord costs ~6 us of GPU idle on
  // this stack, so the dominant kernel is sampled, not bracketed on every launch)
  ScopedTimer(pa_dqn* h_, const char* name, hipStream_t s_, int level = 2, int every = 1,
              int64_t units = 0)
      : h(h_), t(nullptr), s(s_), active(false) {
    if (h->timing < level) return;
    if (every > 1 && h->timing == level && (h->tick % every) != 0) return;
    t = find_timer(h, name);
    if (t->used + 2 > 2 * kMaxTimedPairs) return;
    while (t->ev.size() < t->used + 2) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      t->ev.push_back(e);
    }
    (void)hipEventRecord(t->ev[t->used], s);
    t->units += units;
    active = true;
  }
  ~ScopedTimer() {
    if (!active) return;
    (void)hipEventRecord(t->ev[t->used + 1], s);
    t->used += 2;
  }
};

// One 16-wave workgroup per CU, two teams alternating main loop and epilogue / prologue
// (target_pp_kernel.hpp).  Always work-stealing: a.tile_ctr must point at a zeroed counter.
template <int NKG>
int launch_target_pp_t(const TargetArgs& a, int ncu, hipStream_t s) {
  static bool configured = false;
  const size_t smem = target_pp_smem_bytes(a.H1);
  if (!configured) {
    int rc = set_max_smem(target_pp_kernel<NKG>, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  // every CU hosts one workgroup; twice as many are offered so that the ones which land on a
  // reserved CU (and exit) do not leave other CUs empty
  // (with a partition the full offer is needed whatever the tile count: a short grid could
  // land on reserved CUs only and process nothing)
  int grid = 2 * ncu;
  if (!a.reserved && grid > a.ntiles) grid = a.ntiles;
  hipLaunchKernelGGL(target_pp_kernel<NKG>, dim3((unsigned)grid), dim3(1024), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

int launch_target_pp(const TargetArgs& a, int ncu, hipStream_t s) {
  switch (t_nkg(a.H1)) {
    case 8: return launch_target_pp_t<8>(a, ncu, s);
    case 16: return launch_target_pp_t<16>(a, ncu, s);
    default: return launch_target_pp_t<32>(a, ncu, s);
  }
}


// ---- cross-stream hand-off through a device word (see pa_dqn::sig) -------------------------
static __global__ void signal_kernel(int* flag, int value) {
  if (threadIdx.x == 0)
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Bounded like consume_y: a producer that never runs must not hang the GPU (err word, pa_dqn_check).
// The bound is wall-clock time (the constant 100 MHz counter), and generous: in a data-parallel
// run the producer — the learner stream — can legitimately sit in a collective for seconds (RCCL's
// first-call set-up, a rank that is still filling its arena).
// `done_flag` (optional): published when the launch STARTS — "everything enqueued on this stream
// before me has completed" (the window's gather of x, for the learner stream's first row pass).
static __global__ void wait_flag_kernel(const int* flag, int value, int* err, int* err_host,
                                        int* done_flag = nullptr, int done_value = 0) {
  if (threadIdx.x != 0) return;
  if (done_flag) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  constexpr long long kLimitTicks = 120LL * 100000000LL;     // 120 s
  const long long t0 = (long long)wall_clock64();
  int spins = 0;
  while ((__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if ((++spins & 1023) == 0 && (long long)wall_clock64() - t0 > kLimitTicks) {
      __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (err_host) __hip_atomic_store(err_host, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}
// The same wait at the head of a target-update window, by a launch that then rebuilds the fp16 planes of
// the target W2 (target_h2_kernel.hpp: the soft update that the word announces changed their scales):
// 64 workgroups x 4 waves, one wave per row of W2'; every wave waits for itself.  Resident before the
// word flips, ~1 us of work after it — a pack launch of its own would sit on the window's critical path.
static __global__ __launch_bounds__(256) void wait_pack_kernel(const int* flag, int value, int* err, int* err_host,
                                                               int* done_flag, int done_value, W2hPack p) {
  if (done_flag && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(done_flag, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  constexpr long long kLimitTicks = 120LL * 100000000LL;     // 120 s
  const long long t0 = (long long)wall_clock64();
  int spins = 0;
  while ((__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if ((++spins & 1023) == 0 && (long long)wall_clock64() - t0 > kLimitTicks) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err_host) __hip_atomic_store(err_host, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  target_w2h_pack<true>(p, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4, threadIdx.x & 63);
}
W2hPack w2h_pack_args(const pa_dqn* h) {
  W2hPack p;
  p.W2 = h->bufs.q_target + h->off[2]; p.planes = h->w2h; p.fields = h->w2hf;
  return p;
}
// Everything enqueued on `to` after this call runs after everything enqueued on `from` before it.
int stream_hop(pa_dqn* h, hipStream_t from, hipStream_t to, hipEvent_t fallback) {
  if (!h->use_flags) {
    PA_HIP(hipEventRecord(fallback, from));
    PA_HIP(hipStreamWaitEvent(to, fallback, 0));
    return PA_OK;
  }
  const int gen = ++h->sig_gen;
  hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, from, h->sig, gen);
  PA_LAUNCH_CHECK();
  hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, to, h->sig, gen, h->err_dev, h->err_host);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// The first launch of a learn() call: workgroups [0, rounds) draw the index lists of all rounds
// (sampler.hpp), the workgroups behind them rebuild EVERY fragment-major weight copy from the
// row-major parameters.  The two halves are independent, so the rebuild — 6 us as a launch of its
// own — hides under the sampler's ~11 us, and learn() no longer has to guess whether somebody wrote
// the parameters since the last call (torch's version counters miss writes through `.data`, which
// is the reference's own update_target_network idiom, common/utils.py:214-226).
constexpr int kPrologueRepackWgs = 48;
static __global__ __launch_bounds__(SAMPLE_THREADS) void learn_prologue_kernel(SampleArgs sa, int rounds,
                                                                              RepackArgs ra, int* zero,
                                                                              int nzero) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long table[];
  if ((int)blockIdx.x < rounds) {
    sample_indices_block(sa, (int)blockIdx.x, table);
    return;
  }
  // (the call's work-stealing counters and error word: a memset launch of their own otherwise)
  if ((int)blockIdx.x == rounds)
    for (int i = threadIdx.x; i < nzero; i += SAMPLE_THREADS) zero[i] = 0;
  repack_body(ra, (int64_t)((int)blockIdx.x - rounds) * SAMPLE_THREADS + threadIdx.x,
              (int64_t)kPrologueRepackWgs * SAMPLE_THREADS);
}

struct NetPtrs {
  const float *W1, *b1, *W2, *b2, *W3, *b3;
};
NetPtrs net_ptrs(const pa_dqn* h, const float* base) {
  NetPtrs n;
  n.W1 = base + h->off[0]; n.b1 = base + h->off[1]; n.W2 = base + h->off[2];
  n.b2 = base + h->off[3]; n.W3 = base + h->off[4]; n.b3 = base + h->off[5];
  return n;
}

int check_batch(const pa_dqn* h, const pa_dqn_batch* b) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "learner has no bound parameter buffers");
  PA_REQUIRE(b, PA_ERR_INVALID, "null batch");
  PA_REQUIRE(b->B > 0 && b->B <= h->d.max_batch, PA_ERR_INVALID,
             "batch size %d outside (0, max_batch=%d]", b->B, h->d.max_batch);
  PA_REQUIRE(b->A > 0 && b->A <= h->d.max_actions && b->A <= T_ROWS, PA_ERR_UNSUPPORTED,
             "available-action count %d outside (0, min(max_actions=%d, %d)]", b->A,
             h->d.max_actions, T_ROWS);
  PA_REQUIRE(b->x || (b->state && b->action_rep), PA_ERR_INVALID,
             "batch needs x or (state, action_rep)");
  PA_REQUIRE(b->next_state && b->reward && b->terminated, PA_ERR_INVALID,
             "batch needs next_state, reward, terminated");
  if (h->d.double_q == 2)
    PA_REQUIRE(b->next_action_rep, PA_ERR_INVALID, "a SARSA batch needs next_action_rep");
  else
    PA_REQUIRE(b->next_avail_rep, PA_ERR_INVALID, "batch needs next_avail_rep");
  return PA_OK;
}

// x operand of the online net: either supplied, or packed from (state, action_rep).
int resolve_x(pa_dqn* h, const pa_dqn_batch* b, const float** x, hipStream_t s) {
  if (b->x) {
    *x = b->x;
    return PA_OK;
  }
  const int64_t total = (int64_t)b->B * h->IN;
  unsigned grid = (unsigned)(ceil_div(total, 256) > 1024 ? 1024 : ceil_div(total, 256));
  hipLaunchKernelGGL(pack_x_kernel, dim3(grid), dim3(256), 0, s, b->state, b->action_rep,
                     h->xpack, b->B, h->d.state_dim, h->d.action_dim);
  PA_LAUNCH_CHECK();
  *x = h->xpack;
  return PA_OK;
}

GemmArgs target_l1_problem(pa_dqn* h, const float* next_state, int rows, float* U,
                           const float* params = nullptr) {
  // U = s' W1s'^T + b1'   (state columns of the first layer; the target net's unless `params`)
  const pa_dqn_desc& d = h->d;
  const NetPtrs t = net_ptrs(h, params ? params : h->bufs.q_target);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = next_state; g.lda = d.state_dim;
  g.Bm = t.W1; g.ldb = h->IN;
  g.C = U; g.ldc = d.hidden1;
  g.bias = t.b1;
  g.M = rows; g.N = d.hidden1; g.K = d.state_dim;
  g.epi = EPI_BIAS;
  return g;
}

// The split tile forms U = W1s' s' + b1' itself (target_split_kernel.hpp) when the pass runs on the
// TARGET parameters, the planes exist, and a 64-row tile holds at most 32 transitions (A >= 2):
// no first-layer GEMM launch, no [rows][H1] round trip through HBM.
bool fuse_u_ok(const pa_dqn* h, int A) {
  return h->fuse_u && h->use_split && h->w2sp && h->w1sp && A >= 2 && T_ROWS / A <= 32 &&
         target_split_fusable_S(h->d.state_dim);
}

// max_a' Q_target(s', a') and the Bellman target  (deep_q_learning.py:130-167,
// deep_td_learning.py:313-317) for b->B transitions (a whole window of rounds inside learn());
// U (= W1s' s' + b1' of the same rows) must already be computed.
// The operands of one target pass; `fused_u` (out): the split tile forms U itself.
TargetArgs make_target_args(const pa_dqn* h, const pa_dqn_batch* b, const float* U, float* next_v,
                            float* y, int* argmax, bool* split_out, bool* fused_u) {
  const pa_dqn_desc& d = h->d;
  const NetPtrs t = net_ptrs(h, argmax ? h->bufs.q : h->bufs.q_target);
  TargetArgs a;
  memset(&a, 0, sizeof(a));
  a.U = U; a.ldu = d.hidden1;
  a.feat = b->next_avail_rep;
  a.feat_bstride = b->next_avail_bcast ? 0 : (int64_t)b->A * d.action_dim;
  a.mask = b->next_mask;
  a.mask_bstride = b->next_avail_bcast ? 0 : b->A;
  a.W1a = t.W1 + d.state_dim; a.ldw1 = h->IN;
  a.W2f = argmax ? h->w2f_online : h->w2f;
  // (Double DQN's argmax pass runs on the ONLINE parameters: their planes are rebuilt per round)
  a.W2sp = !h->use_split ? nullptr : (argmax ? h->w2sp_online : h->w2sp);
  if (h->use_h2t && h->w2h && !argmax) { a.W2h = h->w2h; a.w2hf = h->w2hf; }
  a.argmax = argmax;
  a.choice_rep = argmax ? h->choice_rep : nullptr;
  a.b2 = t.b2; a.w3 = t.W3; a.b3 = t.b3;
  a.reward = b->reward; a.term = b->terminated;
  a.gamma = d.discount;
  a.next_v = next_v; a.y = y;
  a.B = b->B; a.A = b->A; a.AD = d.action_dim; a.H1 = d.hidden1; a.H2 = d.hidden2;
  a.bpw = T_ROWS / b->A;
  a.ntiles = (int)ceil_div(b->B, a.bpw);
  const bool split = a.W2sp && t_nkg(a.H1) == 32 && target_fast_shape(a, 32);
  const bool fuse = split && !argmax && fuse_u_ok(h, b->A) && b->next_state;
  if (fuse) {
    a.W1sp = h->w1sp;
    a.next_state = b->next_state; a.ld_next = d.state_dim;
    a.b1 = t.b1;
    a.S = d.state_dim;
  }
  if (split_out) *split_out = split;
  if (fused_u) *fused_u = fuse;
  return a;
}
// Will run_target_fused_u on this batch form U in the tile?  (callers skip the first-layer GEMM)
bool target_fuses_u(const pa_dqn* h, const pa_dqn_batch* b) {
  bool fuse = false;
  (void)make_target_args(h, b, h->Uw[0], nullptr, nullptr, nullptr, nullptr, &fuse);
  return fuse;
}

// U of a batch with the split tile's own arithmetic, as a launch (u_split_kernel): for target passes
// that are throughput work (the persistent remainder of a window) and read U like the fp32 tiles do
int run_u_split(pa_dqn* h, const pa_dqn_batch* b, float* U, hipStream_t s) {
  bool fuse = false;
  TargetArgs a = make_target_args(h, b, U, nullptr, nullptr, nullptr, nullptr, &fuse);
  PA_REQUIRE(fuse, PA_ERR_INVALID, "run_u_split: batch does not qualify");
  ScopedTimer tm(h, "target_l1", s, 2, 1, b->B);
  return launch_u_split(a, U, s);
}

// read_u: U has been computed by run_u_split (or, for batches that never fuse, by the fp32 GEMM):
// the tile reads it instead of forming it
int run_target_fused_u(pa_dqn* h, const pa_dqn_batch* b, const float* U, float* next_v, float* y,
                       hipStream_t s, bool persistent = false, int* argmax = nullptr,
                       bool sample_timer = true, bool no_pingpong = false,
                       int prio_first_rows = 0, bool read_u = false, int rows_hint = 0) {
  // argmax != null: the pass runs on the ONLINE parameters and only reports each row's first
  // maximum (Double DQN's action choice); always the classic grid
  // level 1: only the launches the caller marks (learn(): the last, largest piece of every 4th
  // window of a call, the first window included, so that even a 3-window call is sampled);
  // level 2: every launch
  ScopedTimer tm(h, "target", s, (sample_timer || h->timing >= 2) ? 1 : 2, 1, b->B);
  bool split = false;
  TargetArgs a = make_target_args(h, b, U, next_v, y, argmax, &split, nullptr);
  if (read_u) a.W1sp = nullptr;
  a.rows_hint = rows_hint;
  a.prof = (h->prof_tgt && a.ntiles <= h->prof_tgt_tiles) ? h->prof_tgt : nullptr;
  if (prio_first_rows > 0) a.prio_tiles = (int)ceil_div(prio_first_rows, a.bpw);
  const bool pp = !argmax && !no_pingpong && !split &&
                  (h->pingpong == 2 || (h->pingpong == 1 && persistent));
  if (persistent || pp) {
    if (h->ctr_next >= kTileCtrs) {  // ordered after every earlier launch on this stream
      PA_HIP(hipMemsetAsync(h->tile_ctr, 0, kTileCtrs * sizeof(int), s));
      h->ctr_next = 0;
    }
    a.tile_ctr = h->tile_ctr + h->ctr_next++;
    a.reserved = persistent ? h->reserved_dev : nullptr;
    if (a.reserved && h->dbg_workers) {
      a.dbg_workers = h->dbg_workers;
      h->dbg_launches += 1;
    }
  }
  if (pp) return launch_target_pp(a, h->ncu, s);
  return launch_target(a, s);
}
int run_target_fused(pa_dqn* h, const pa_dqn_batch* b, float* next_v, float* y, hipStream_t s) {
  h->y_clean = false;  // stand-alone paths use yw[0] as plain scratch
  return run_target_fused_u(h, b, h->Uw[0], next_v, y, s);
}

PackedW packed(pa_dqn* h) {
  PackedW pk;
  pk.W1f = h->W1f; pk.W2f = h->W2f16; pk.W2tf = h->W2tf; pk.tW2f = h->w2f;
  pk.tW2sp = h->w2sp;
  pk.tW1sp = h->w1sp; pk.sp_S = h->d.state_dim;
  pk.tW2h = h->use_h2t ? h->w2h : nullptr; pk.tW2hf = h->w2hf;
  return pk;
}

// The fp16 row pass's row maxima (online_f16_kernel.hpp): a rebuild writes the buffer the next row
// pass reads and clears the other; an optimizer launch accumulates into the other buffer, clears the
// one just read, and the two swap.
void repack_umax(pa_dqn* h, RepackArgs& a) {
  if (!h->umax) return;
  a.umax_out = h->umax + (size_t)h->umax_cur * HF_UMAX;
  a.umax_zero = h->umax + (size_t)(h->umax_cur ^ 1) * HF_UMAX;
}
void optimizer_umax(pa_dqn* h, AdamFuse& f) {
  if (!h->umax) return;
  f.umax_acc = h->umax + (size_t)(h->umax_cur ^ 1) * HF_UMAX;
  f.umax_clear = h->umax + (size_t)h->umax_cur * HF_UMAX;
  h->umax_cur ^= 1;
}

// Rebuild the fragment-major weight copies from the row-major parameters.  Needed whenever the
// parameters may have been changed by someone else (checkpoint load, stand-alone soft update,
// the data-parallel AdamW); inside the fused learn loop the optimizer tail keeps them current.
int run_repack(pa_dqn* h, bool online, bool target, hipStream_t s) {
  const pa_dqn_desc& d = h->d;
  RepackArgs a;
  memset(&a, 0, sizeof(a));
  a.q = h->bufs.q; a.q_target = h->bufs.q_target;
  a.off_w1 = h->off[0]; a.off_w2 = h->off[2];
  a.IN = h->IN; a.H1 = d.hidden1; a.H2 = d.hidden2;
  a.pk = packed(h);
  a.do_online = online; a.do_target = target;
  if (online) repack_umax(h, a);
  hipLaunchKernelGGL(repack_online_kernel, dim3(128), dim3(256), 0, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// Double DQN's next-state values and Bellman targets (double_dqn.py:29-57):
//   a'_b = argmax over the available next actions of Q_ONLINE(s'_b, .)   (masked -> -inf, first max)
//   v_b  = Q_TARGET(s'_b, a'_b);   y_b = v_b gamma (1 - term_b) + r_b
// Both passes are target_fused_kernel: first on the ONLINE parameters over all A actions, reporting
// each row's first maximum and the representation of that action (its fragment-major W2 copy is
// rebuilt here: the online net moves every round); then on the target parameters with ONE action
// per transition — the chosen one — whose "row max" is the value and whose epilogue writes y.
// The two first-layer state products (same s', two parameter sets) share one launch.
// value_stream (learn() with two streams): the value pass runs THERE, released by the device word
// the next row-pass launch on `s` publishes when it starts (= the argmax pass has completed), and
// hands y over as data-tagged words — so it runs beside the row pass's forward half instead of in
// front of it (8.6 us of a 70 us round).  The caller launches that row pass next, with tagged y.
int run_double_targets(pa_dqn* h, const pa_dqn_batch* b, float* next_v, float* y, hipStream_t s,
                       hipStream_t value_stream = nullptr) {
  const pa_dqn_desc& d = h->d;
  PA_REQUIRE(h->w2f_online && b->B <= d.max_batch, PA_ERR_INVALID,
             "double-Q pass: learner was not created with double_q, or batch above max_batch");
  h->y_clean = false;
  int rc;
  if (!h->online_tgt_ok) {
    RepackArgs a;
    memset(&a, 0, sizeof(a));
    a.q = h->bufs.q; a.q_target = h->bufs.q;     // "target" slot <- the ONLINE W2
    a.off_w1 = h->off[0]; a.off_w2 = h->off[2];
    a.IN = h->IN; a.H1 = d.hidden1; a.H2 = d.hidden2;
    a.pk = packed(h);
    a.pk.tW2f = h->w2f_online;
    a.pk.tW2sp = h->w2sp_online;                 // (never h->w2sp: those stay the TARGET network's)
    a.pk.tW1sp = nullptr;
    a.pk.tW2h = nullptr;
    a.do_online = 0; a.do_target = 1;
    hipLaunchKernelGGL(repack_online_kernel, dim3(128), dim3(256), 0, s, a);
    PA_LAUNCH_CHECK();
  }
  {
    ScopedTimer tm(h, "target_l1", s, 2, 1, b->B);
    GemmArgs g[2] = {target_l1_problem(h, b->next_state, b->B, h->Uw[0], h->bufs.q),
                     target_l1_problem(h, b->next_state, b->B, h->Uw[1])};
    rc = launch_linear<false>(g, 2, s);
    if (rc != PA_OK) return rc;
  }
  rc = run_target_fused_u(h, b, h->Uw[0], nullptr, nullptr, s, false, h->choice, true, false, 0, false, 32);
  if (rc != PA_OK) return rc;
  pa_dqn_batch one = *b;
  one.A = 1;
  one.next_avail_rep = h->choice_rep;
  one.next_avail_bcast = 0;
  one.next_mask = nullptr;
  hipStream_t vs = s;
  if (value_stream) {
    const int gen = ++h->sig_gen;
    h->pending_signal = gen;
    hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, value_stream, h->sig, gen, h->err_dev,
                       h->err_host);
    PA_LAUNCH_CHECK();
    vs = value_stream;
  }
  return run_target_fused_u(h, &one, h->Uw[1], next_v, y, vs, false, nullptr, true, false, 0, false, 32);
}

// max_a' Q_target(s', a') (DeepQLearning) or Q_target(s', argmax_a' Q(s', a')) (DoubleDQN) and the
// Bellman targets of one batch, U computed here.
int run_next_values(pa_dqn* h, const pa_dqn_batch* b, float* next_v, float* y, hipStream_t s) {
  if (h->d.double_q == 2) {
    // DeepSARSA (deep_sarsa.py:59-78): Q_target(s', a') for the committed next action — the fused
    // kernel with ONE action per transition
    h->y_clean = false;
    GemmArgs g = target_l1_problem(h, b->next_state, b->B, h->Uw[0]);
    int rc = launch_linear<false>(&g, 1, s);
    if (rc != PA_OK) return rc;
    pa_dqn_batch one = *b;
    one.A = 1;
    one.next_avail_rep = b->next_action_rep;
    one.next_avail_bcast = 0;
    one.next_mask = nullptr;
    return run_target_fused_u(h, &one, h->Uw[0], next_v, y, s);
  }
  if (h->d.double_q) return run_double_targets(h, b, next_v, y, s);
  if (!target_fuses_u(h, b)) {       // (the split tile forms U itself otherwise)
    ScopedTimer tm(h, "target_l1", s);
    GemmArgs g = target_l1_problem(h, b->next_state, b->B, h->Uw[0]);
    int rc = launch_linear<false>(&g, 1, s);
    if (rc != PA_OK) return rc;
  }
  return run_target_fused(h, b, next_v, y, s);
}

// Forward (+ loss + backward to dZ2 / dZ1 when y is given) of the online network.
// phase: 0 = one launch; 1 / 2 = the forward / backward halves as two launches (PH of
// online_rowpass_kernel; only for the <9, 16, 16> shape, see rowpass_can_split)
int run_rowpass(pa_dqn* h, const float* x, int B, const float* y, bool y_tagged, float* q_out,
                int world, hipStream_t s, int phase = 0) {
  const pa_dqn_desc& d = h->d;
  const NetPtrs q = net_ptrs(h, h->bufs.q);
  ScopedTimer tm(h, "rowpass", s, h->chain_sample ? 1 : 2, 1, B);
  const size_t smem = rowpass_smem_bytes(h->IN, d.hidden1, d.hidden2);
  RowArgs a;
  memset(&a, 0, sizeof(a));
  const bool use_h2 = h->umax && !(h->pair && h->qx);
  if (use_h2) a.umax = h->umax + (size_t)h->umax_cur * HF_UMAX;
  a.x = x; a.ldx = h->IN;
  a.W1f = h->W1f; a.b1 = q.b1;
  a.W2f = h->W2f16; a.b2 = q.b2;
  a.W2tf = h->W2tf;
  a.w3 = q.W3; a.b3 = q.b3;
  a.y = y;
  a.y_tagged = y_tagged ? 1 : 0;
  a.err = h->err_dev;
  a.err_host = h->err_host;
  if (h->pending_signal && phase != 2) {
    a.signal_flag = h->sig;
    a.signal_value = h->pending_signal;
    h->pending_signal = 0;
  }
  if (h->pending_wait && phase != 2) {
    a.wait_flag = h->sig + 1;
    a.wait_value = h->pending_wait;
    h->pending_wait = 0;
  }
  a.prof = (h->prof_round < 0 || h->prof_round == h->cur_round) ? h->prof_row : nullptr;
  a.H1a = y ? h->H1a : nullptr; a.H2a = y ? h->H2a : nullptr;
  a.dZ2 = h->dZ2; a.dZ1 = h->dZ1;
  a.q_out = phase == 2 ? nullptr : q_out; a.dq_out = h->dq; a.absd_out = h->absd;
  a.q_in = phase == 2 ? q_out : nullptr;
  a.norm = (float)(2.0 / ((double)B * (double)world));
  a.B = B; a.K1 = h->IN; a.H1 = d.hidden1; a.H2 = d.hidden2;
  const dim3 grid((unsigned)ceil_div(B, RP_ROWS));
  const int g1 = wf16_nkg(h->IN), g2 = wf16_nkg(d.hidden1), g3 = wf16_nkg(d.hidden2);
  if (h->pair && h->qx) {
    // two workgroups per 16-row tile (online_pair_kernel.hpp): the benchmark's shape
    PairArgs pa_;
    memset(&pa_, 0, sizeof(pa_));
    pa_.r = a;
    pa_.dZ1p = h->dZ1p;
    pa_.qx = h->qx; pa_.qx_rows = h->qx_rows;
    pa_.qhalf = h->qhalf;
    pa_.ntiles = (int)ceil_div(B, RP_ROWS);
    if (phase == 2) { pa_.r.q_out = q_out; pa_.r.q_in = nullptr; }   // the backward launch reports Q(s, a)
    if (!h->qx_clean) {
      PA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->qx), (int)kYPendingBits,
                               (size_t)2 * h->qx_rows, s));
      h->qx_clean = true;
    }
    size_t smem_p = pair_smem_bytes();
    if ((size_t)h->pair_lds > smem_p) smem_p = (size_t)h->pair_lds;
    const dim3 pgrid((unsigned)(16 * ceil_div(pa_.ntiles, 8)));
#define PA_PAIR(PH_)                                                                          \
  do {                                                                                        \
    static size_t configured = 0;                                                             \
    if (smem_p > configured) {                                                                \
      int rc = set_max_smem(online_rowpass_pair_kernel<PH_>, smem_p);                          \
      if (rc != PA_OK) return rc;                                                             \
      configured = smem_p;                                                                    \
    }                                                                                         \
    hipLaunchKernelGGL((online_rowpass_pair_kernel<PH_>), pgrid, dim3(512), smem_p, s, pa_);   \
  } while (0)
    if (phase == 1) PA_PAIR(1);
    else if (phase == 2) PA_PAIR(2);
    else PA_PAIR(0);
#undef PA_PAIR
    PA_LAUNCH_CHECK();
    if (y && phase != 1) h->pair_live = true;
    return PA_OK;
  }
  if (y && phase != 1) h->pair_live = false;
  if (use_h2) {
    const size_t smem2 = rowpass_h2_smem_bytes();
    if (phase == 1) hipLaunchKernelGGL((online_rowpass_h2_kernel<1>), grid, dim3(512), smem2, s, a);
    else if (phase == 2) hipLaunchKernelGGL(rowpass_scale_kernel, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((online_rowpass_h2_kernel<0>), grid, dim3(512), smem2, s, a);
    PA_LAUNCH_CHECK();
    return PA_OK;
  }
  // fully unrolled instantiations for the shapes that matter; anything else takes the run-time loops
#define PA_ROWPASS(N1, N2, N3)                                                               \
  do {                                                                                       \
    static size_t configured = 0;                                                            \
    if (smem > configured) {                                                                 \
      int rc = set_max_smem(online_rowpass_kernel<N1, N2, N3>, smem);                        \
      if (rc != PA_OK) return rc;                                                            \
      configured = smem;                                                                     \
    }                                                                                        \
    hipLaunchKernelGGL((online_rowpass_kernel<N1, N2, N3>), grid, dim3(512), smem, s, a);    \
  } while (0)
#define PA_ROWPASS_PH(PH_)                                                                   \
  do {                                                                                       \
    static size_t configured = 0;                                                            \
    if (smem > configured) {                                                                 \
      int rc = set_max_smem(online_rowpass_kernel<9, 16, 16, PH_>, smem);                    \
      if (rc != PA_OK) return rc;                                                            \
      configured = smem;                                                                     \
    }                                                                                        \
    hipLaunchKernelGGL((online_rowpass_kernel<9, 16, 16, PH_>), grid, dim3(512), smem, s, a); \
  } while (0)
  if (phase != 0) {
    PA_REQUIRE(g1 == 9 && g2 == 16 && g3 == 16 && y && q_out, PA_ERR_INVALID,
               "split row pass: shape or arguments not covered");
    if (phase == 1) PA_ROWPASS_PH(1);
    else PA_ROWPASS_PH(2);
  } else if (g2 == 16 && g3 == 16 && g1 == 9) PA_ROWPASS(9, 16, 16);
  else if (g2 == 16 && g3 == 16) PA_ROWPASS(0, 16, 16);
  else if (g2 == 4 && g3 == 4 && g1 == 1) PA_ROWPASS(1, 4, 4);
  else PA_ROWPASS(0, 0, 0);
#undef PA_ROWPASS
#undef PA_ROWPASS_PH
  PA_LAUNCH_CHECK();
  return PA_OK;
}

AdamScalars adam_scalars(const pa_dqn_desc& d, int64_t step) {
  // python-float (double) scalars exactly as _single_tensor_adam forms them
  const double bc1 = 1.0 - pow(d.beta1, (double)step);
  const double bc2 = 1.0 - pow(d.beta2, (double)step);
  AdamScalars c;
  c.decay = (float)(1.0 - d.lr * d.weight_decay);
  c.w1 = (float)(1.0 - d.beta1);
  c.beta2 = (float)d.beta2;
  c.omb2 = (float)(1.0 - d.beta2);
  c.bc2_sqrt = (float)sqrt(bc2);
  c.neg_step = (float)(-(d.lr / bc1));
  c.eps = (float)d.eps;
  c.amsgrad = d.amsgrad;
  return c;
}
AdamState adam_state(pa_dqn* h) {
  AdamState st;
  st.p = h->bufs.q; st.m = h->bufs.exp_avg; st.v = h->bufs.exp_avg_sq;
  st.vmax = h->bufs.max_exp_avg_sq;
  return st;
}

// Weight gradients of all three layers; with fuse_adam the workgroup that finishes a tile also
// applies AdamW to it (+ the next step's soft target update when soft_next) and the extra
// workgroup folds |Q - target| into loss_out.
int run_weight_grad(pa_dqn* h, const float* x, int B, bool fuse_adam, int64_t adam_step,
                    float* loss_out, int soft_next, hipStream_t s, const float* y_tagged = nullptr) {
  const pa_dqn_desc& d = h->d;
  float* G = h->bufs.grad;
  ScopedTimer tm(h, "bwd_dw", s, h->chain_sample ? 1 : 2, 1, B);
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.nprob = 3;
  // 32-row tiles (twice the workgroups, half the MFMA time each) when the chain has the CUs for
  // them: the partition of the overlapped loop reserves at least one CU per workgroup
  int TM = DW_TM;
  {
    const int tiles32 = (int)(ceil_div(d.hidden2, 32) * ceil_div(d.hidden1, DW_TN) +
                              ceil_div(d.hidden1, 32) * ceil_div(h->IN, DW_TN) +
                              ceil_div(d.hidden2, DW_TN)) + 1;
    if (h->dw_tm == 32 || (h->dw_tm == 0 && h->reserved_dev && h->n_reserved >= tiles32)) TM = 32;
  }
  a.tm = TM;
  // dW2 = dZ2^T H1a, db2
  a.p[0].dZ = h->dZ2; a.p[0].ldz = d.hidden2;
  a.p[0].X = h->H1a; a.p[0].ldx = d.hidden1;
  a.p[0].dW = G + h->off[2]; a.p[0].ldw = d.hidden1;
  a.p[0].db = G + h->off[3];
  a.p[0].M = d.hidden2; a.p[0].N = d.hidden1;
  a.p[0].tiles_n = (int)ceil_div(d.hidden1, DW_TN);
  a.p[0].tile0 = 0;
  a.p[0].kind = 0;
  int t0 = (int)ceil_div(d.hidden2, TM) * a.p[0].tiles_n;
  // dW1 = dZ1^T x, db1
  a.p[1].dZ = h->dZ1; a.p[1].ldz = d.hidden1;
  if (h->pair_live) {   // the two halves' partials, interleaved per pair of units (online_pair_kernel.hpp)
    a.p[1].dZ = h->dZ1p; a.p[1].ldz = 2 * d.hidden1;
    a.p[1].dz_pair = 1;
  }
  a.p[1].X = x; a.p[1].ldx = h->IN;
  a.p[1].dW = G + h->off[0]; a.p[1].ldw = h->IN;
  a.p[1].db = G + h->off[1];
  a.p[1].M = d.hidden1; a.p[1].N = h->IN;
  a.p[1].tiles_n = (int)ceil_div(h->IN, DW_TN);
  a.p[1].tile0 = t0;
  a.p[1].kind = 1;
  t0 += (int)ceil_div(d.hidden1, TM) * a.p[1].tiles_n;
  // dW3 = dq^T H2a, db3 = sum dq   (dq is a [B][1] "dZ")
  a.p[2].dZ = h->dq; a.p[2].ldz = 1;
  a.p[2].X = h->H2a; a.p[2].ldx = d.hidden2;
  a.p[2].dW = G + h->off[4]; a.p[2].ldw = d.hidden2;
  a.p[2].db = G + h->off[5];
  a.p[2].M = 1; a.p[2].N = d.hidden2;
  a.p[2].tiles_n = (int)ceil_div(d.hidden2, DW_TN);
  a.p[2].tile0 = t0;
  a.p[2].kind = 2;
  t0 += a.p[2].tiles_n;
  a.total_tiles = t0;
  // TIMING EXPERIMENT ONLY (results are wrong: the left-out layers never step): what a launch costs
  // that carries only the tiles the next round's layer 1 waits for (1: W1), or only the ones whose
  // operands exist before dZ1 does (2: W2 + w3) — the two halves of a weight-gradient launch split
  // by dependency (profiles/r06_b_dw_by_dependency.txt)
  static const int dw_only = env_int("PEARL_AMD_DEBUG_DW_ONLY", 0);
  if (dw_only == 1) {
    a.p[0] = a.p[1];
    a.p[0].tile0 = 0;
    a.nprob = 1;
    a.total_tiles = (int)ceil_div(d.hidden1, TM) * a.p[0].tiles_n;
  } else if (dw_only == 2) {
    a.p[1] = a.p[2];
    a.p[1].tile0 = (int)ceil_div(d.hidden2, TM) * a.p[0].tiles_n;
    a.nprob = 2;
    a.total_tiles = a.p[1].tile0 + a.p[1].tiles_n;
  }
  a.B = B;
  a.prof = (h->prof_round < 0 || h->prof_round == h->cur_round) ? h->prof_dw : nullptr;
  a.ad.absd = h->absd; a.ad.nabs = B; a.ad.inv_B = (float)(1.0 / (double)B);
  a.ad.loss_out = loss_out;
  a.ad.y_restore = reinterpret_cast<unsigned*>(const_cast<float*>(y_tagged));
  a.ad.n_restore = y_tagged ? B : 0;
  if (fuse_adam) {
    PA_REQUIRE(adam_step >= 1, PA_ERR_INVALID, "adam step must be >= 1 (got %lld)",
               (long long)adam_step);
    a.ad.enabled = 1;
    a.ad.c = adam_scalars(d, adam_step);
    a.ad.st = adam_state(h);
    a.ad.grad_base = G;
    a.ad.W1f = h->W1f; a.ad.W2f = h->W2f16; a.ad.W2tf = h->W2tf;
    a.ad.nkg_w1 = wf16_nkg(h->IN); a.ad.nkg_w2 = wf16_nkg(d.hidden1);
    a.ad.nkg_w2t = wf16_nkg(d.hidden2);
    a.ad.soft_next = soft_next;
    a.ad.tgt = h->bufs.q_target; a.ad.tau = d.tau;
    a.ad.one_minus_tau = (float)(1.0 - (double)d.tau);
    a.ad.tW2f = h->w2f; a.ad.nkg_t = t_nkg(d.hidden1);
    a.ad.tW2sp = h->w2sp;
    a.ad.tW1sp = h->w1sp; a.ad.sp_S = d.state_dim;
    optimizer_umax(h, a.ad);
    static const bool keep_online = env_int("PEARL_AMD_DDQN_KEEP_ONLINE", 1) != 0;
    if (d.double_q == 1 && h->w2f_online && keep_online) {
      // the argmax pass of the next round reads these: no repack launch in front of it
      a.ad.oW2f = h->w2f_online;
      a.ad.oW2sp = h->use_split ? h->w2sp_online : nullptr;
      h->online_tgt_ok = true;
    }
  }
  if (h->pair_live) {
    static const DwKernelFn kPairKernels[4] = {weight_grad_split_kernel32_pair, weight_grad_kernel32_pair,
                                               weight_grad_split_kernel_pair, weight_grad_kernel_pair};
    return launch_weight_grad(a, loss_out != nullptr, s, kPairKernels);
  }
  return launch_weight_grad(a, loss_out != nullptr, s);
}

// AdamW on bufs.grad after the data-parallel all-reduce: same optimizer tail as the fused
// weight-gradient kernel (fragment-major copies, optional soft update for the next step).
int run_adamw(pa_dqn* h, int64_t step, int soft_next, hipStream_t s) {
  const pa_dqn_desc& d = h->d;
  PA_REQUIRE(step >= 1, PA_ERR_INVALID, "adam step must be >= 1 (got %lld)", (long long)step);
  ScopedTimer tm(h, "adamw", s);
  h->online_tgt_ok = false;   // (this kernel does not keep Double DQN's online copies: repack next round)
  AdamDqnArgs a;
  memset(&a, 0, sizeof(a));
  a.f.enabled = 1;
  a.f.c = adam_scalars(d, step);
  a.f.st = adam_state(h);
  a.f.grad_base = h->bufs.grad;
  a.f.W1f = h->W1f; a.f.W2f = h->W2f16; a.f.W2tf = h->W2tf;
  a.f.nkg_w1 = wf16_nkg(h->IN); a.f.nkg_w2 = wf16_nkg(d.hidden1); a.f.nkg_w2t = wf16_nkg(d.hidden2);
  a.f.soft_next = soft_next;
  a.f.tgt = h->bufs.q_target; a.f.tau = d.tau; a.f.one_minus_tau = (float)(1.0 - (double)d.tau);
  a.f.tW2f = h->w2f; a.f.nkg_t = t_nkg(d.hidden1);
  a.f.tW2sp = h->w2sp;
  a.f.tW1sp = h->w1sp; a.f.sp_S = d.state_dim;
  optimizer_umax(h, a.f);
  a.g = h->bufs.grad;
  a.n = h->P;
  for (int i = 0; i < 6; ++i) a.off[i] = h->off[i];
  a.IN = h->IN; a.H1 = d.hidden1; a.H2 = d.hidden2;
  hipLaunchKernelGGL(adamw_dqn_kernel, dim3((unsigned)ceil_div(h->P, 256)), dim3(256), 0, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

int run_soft_update(pa_dqn* h, hipStream_t s) {
  ScopedTimer tm(h, "soft_update", s);
  hipLaunchKernelGGL(soft_update_kernel, dim3((unsigned)ceil_div(h->P, 256)), dim3(256), 0, s,
                     h->bufs.q_target, h->bufs.q, h->P, h->d.tau, (float)(1.0 - (double)h->d.tau));
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// Everything of one learn_batch that depends on the ONLINE parameters, given the Bellman targets
// y[B] of the batch: row pass (forward, loss, dZ2, dZ1) -> weight gradients (+ AdamW).
// soft_next: fuse the NEXT step's soft target update into this step's optimizer tail.
bool rowpass_can_split(const pa_dqn* h) {
  return h->rp_split && wf16_nkg(h->IN) == 9 && wf16_nkg(h->d.hidden1) == 16 &&
         wf16_nkg(h->d.hidden2) == 16;
}

// split_rowpass: the row pass as forward + backward launches (the first round of a window, whose
// targets are still being computed: the forward's CUs go back to the target tiles meanwhile)
// (the two halves of online_chain, for callers that enqueue something between them)
int chain_front(pa_dqn* h, const float* x, int B, const float* y, bool y_tagged, int grad_world,
                hipStream_t s, bool split_rowpass) {
  const int world = grad_world < 0 ? -grad_world : grad_world;
  return run_rowpass(h, x, B, y, y_tagged, h->qbuf, world, s,
                     (split_rowpass && rowpass_can_split(h)) ? 1 : 0);
}
int chain_back(pa_dqn* h, const float* x, int B, const float* y, bool y_tagged, int64_t adam_step,
               int grad_world, float* loss_out, int soft_next, hipStream_t s, bool split_rowpass) {
  const int world = grad_world < 0 ? -grad_world : grad_world;
  if (split_rowpass && rowpass_can_split(h)) {
    int rc = run_rowpass(h, x, B, y, y_tagged, h->qbuf, world, s, 2);
    if (rc != PA_OK) return rc;
  }
  float* lo = loss_out ? loss_out : h->loss_scratch;
  return run_weight_grad(h, x, B, grad_world == 1, adam_step, lo, soft_next, s,
                         y_tagged ? y : nullptr);
}
int online_chain(pa_dqn* h, const float* x, int B, const float* y, bool y_tagged, int64_t adam_step,
                 int grad_world, float* loss_out, int soft_next, hipStream_t s,
                 bool split_rowpass = false) {
  // grad_world < 0: data-parallel split requested explicitly (|grad_world| ranks, AdamW later)
  int rc = chain_front(h, x, B, y, y_tagged, grad_world, s, split_rowpass);
  if (rc != PA_OK) return rc;
  return chain_back(h, x, B, y, y_tagged, adam_step, grad_world, loss_out, soft_next, s, split_rowpass);
}

// One stand-alone learn_batch (pa_dqn_step).  do_target_update: soft update BEFORE the forward
// (deep_td_learning.py:283-284).  The fragment-major weight copies are rebuilt every time: the
// caller may have loaded a checkpoint or stepped the optimizer itself between calls.
int step_impl(pa_dqn* h, const pa_dqn_batch* batch, int do_target_update, int64_t adam_step,
              int grad_world, float* loss_out, hipStream_t s) {
  int rc = check_batch(h, batch);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(grad_world >= 1, PA_ERR_INVALID, "grad_world must be >= 1");
  if (h->timing) h->tick++;
  if (do_target_update) {
    rc = run_soft_update(h, s);
    if (rc != PA_OK) return rc;
  }
  rc = run_repack(h, true, true, s);
  if (rc != PA_OK) return rc;
  const float* x = nullptr;
  rc = resolve_x(h, batch, &x, s);
  if (rc != PA_OK) return rc;
  rc = run_next_values(h, batch, h->nextv, h->yw[0], s);
  if (rc != PA_OK) return rc;
  return online_chain(h, x, batch->B, h->yw[0], false, adam_step, grad_world, loss_out, 0, s);
}

void free_batchbufs(pa_dqn* h) {
  for (int p = 0; p < 2; ++p) {
    void* ptrs[] = {h->bb[p].next_state, h->bb[p].next_avail_rep, h->bb[p].next_mask,
                    h->bb[p].reward, h->bb[p].term};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
  }
  if (h->bb_x) (void)hipFree(h->bb_x);
  h->bb_x = nullptr;
  memset(h->bb, 0, sizeof(h->bb));
  h->bb_A = 0;
}

int ensure_batchbufs(pa_dqn* h, int A) {
  if (h->bb_A >= A && h->bb_x) return PA_OK;
  PA_HIP(hipDeviceSynchronize());
  free_batchbufs(h);
  const pa_dqn_desc& d = h->d;
  const int64_t B = h->wrows;
  PA_HIP(hipMalloc((void**)&h->bb_x, (size_t)(2 * B * h->IN * 4)));
  for (int p = 0; p < 2; ++p) {
    PA_HIP(hipMalloc((void**)&h->bb[p].next_state, (size_t)(B * d.state_dim * 4)));
    PA_HIP(hipMalloc((void**)&h->bb[p].next_avail_rep, (size_t)(B * A * d.action_dim * 4)));
    PA_HIP(hipMalloc((void**)&h->bb[p].next_mask, (size_t)(B * A)));
    PA_HIP(hipMalloc((void**)&h->bb[p].reward, (size_t)(B * 4)));
    PA_HIP(hipMalloc((void**)&h->bb[p].term, (size_t)B));
  }
  h->bb_A = A;
  return PA_OK;
}


// Choose the compute units the online chain keeps for itself: a census kernel reports the
// (XCC, SE, SH, CU) key of every CU; `want` of them, spread evenly over XCDs and shader engines,
// are marked in the table the persistent target kernel consults.  Any surprise (fewer keys than
// CUs, odd topology) leaves the table null: the loop then runs without a partition.
int cu_partition(pa_dqn* h, int want, hipStream_t s) {
  h->reserved_dev = nullptr;
  h->n_reserved = 0;
  if (want <= 0) return PA_OK;
  const int G = 8 * h->ncu;
  unsigned* keys_dev = nullptr;
  PA_HIP(hipMalloc((void**)&keys_dev, (size_t)G * 4));
  hipLaunchKernelGGL(cu_census_kernel, dim3((unsigned)G), dim3(512), 0, s, keys_dev, 100000);
  std::vector<unsigned> keys((size_t)G);
  hipError_t e = hipMemcpyAsync(keys.data(), keys_dev, (size_t)G * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(keys_dev);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return PA_OK;
  }
  std::vector<uint8_t> present(kCuKeys, 0), table(kCuKeys, 0);
  int distinct = 0;
  for (unsigned k : keys)
    if (k < (unsigned)kCuKeys && !present[k]) { present[k] = 1; ++distinct; }
  if (distinct != h->ncu || want >= h->ncu) return PA_OK;
  // round-robin over (xcc, se) groups so that every XCD / shader engine gives up the same share
  std::vector<std::vector<int>> groups;
  for (int g = 0; g < kCuKeys / 32; ++g) {  // key = xcc[11:8] se[7:5] sh[4] cu[3:0]
    std::vector<int> cus;
    for (int c = 0; c < 32; ++c)
      if (present[g * 32 + c]) cus.push_back(g * 32 + c);
    if (!cus.empty()) groups.push_back(cus);
  }
  int taken = 0;
  for (size_t depth = 0; taken < want; ++depth) {
    bool any = false;
    for (auto& cus : groups) {
      if (taken >= want) break;
      if (depth < cus.size()) {
        table[cus[cus.size() - 1 - depth]] = 1;
        ++taken;
        any = true;
      }
    }
    if (!any) break;
  }
  PA_HIP(hipMalloc((void**)&h->reserved_dev, kCuKeys));
  PA_HIP(hipMemcpy(h->reserved_dev, table.data(), kCuKeys, hipMemcpyHostToDevice));
  h->n_reserved = taken;
  return PA_OK;
}

// Side stream + events of the overlapped learn loop (created on first use).
int ensure_side(pa_dqn* h) {
  if (h->side) return PA_OK;
  // The target-network kernels saturate every CU they may run on (two 8-wave workgroups and
  // 137 KB of LDS per CU), and the hardware does not preempt: a chain kernel launched while they
  // run would wait for workgroup slots for most of the window (measured: a 14 us weight-gradient
  // launch stretched to 125 us).  So the side stream is confined to `side_cus` of the 256 CUs with
  // a CU mask (bit i = XCD i % 8: every XCD keeps (256 - side_cus) / 8 CUs free), and the chain
  // always finds idle CUs.  PEARL_AMD_SIDE_CUS=0: no mask, lowest stream priority instead.
  // (A CU-masked side stream, PEARL_AMD_SIDE_CUS=n, was the first attempt: on this stack kernels
  // of a masked queue and of the main queue no longer overlap at all and every launch gains ~7 us.
  // The partition is therefore done inside the target kernel: cu_partition + persistent tiles.)
  const int side_cus = env_int("PEARL_AMD_SIDE_CUS", 0);
  hipDeviceProp_t prop;
  PA_HIP(hipGetDeviceProperties(&prop, h->d.device));
  const int ncu = prop.multiProcessorCount;
  h->ncu = ncu;
  if (side_cus > 0 && side_cus < ncu) {
    uint32_t mask[16];
    memset(mask, 0, sizeof(mask));
    for (int i = 0; i < side_cus && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
    PA_HIP(hipExtStreamCreateWithCUMask(&h->side, (uint32_t)((ncu + 31) / 32), mask));
  } else {
    int lo = 0, hi = 0;
    PA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least urgent
    PA_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo));
  }
  PA_HIP(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&h->ev_chain[0], hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&h->ev_chain[1], hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&h->ev_gather[0], hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&h->ev_gather[1], hipEventDisableTiming));
  // with the bf16x3 target kernel the target side needs far fewer CUs: the chain keeps one per
  // workgroup of its widest launch (112 weight-gradient tiles of 32 rows + the loss workgroup;
  // measured: 112 reserved -> 25.3 M transitions/s with 64-row tiles, 128 -> 27.8 M, 144 -> 25.0 M,
  // where the target side becomes the bound)
  return cu_partition(h, env_int("PEARL_AMD_RESERVED_CUS", (h->use_split && h->w2sp) ? 128 : 64),
                      h->side);
}


// rep(shared next-action table) [A][AD] + mask [A] on the device, for TargetArgs.feat with stride 0.
// one-hot: rep[i][j] = (j == (long)table[i][0])  (one_hot_action_representation_module.py:27-34 on the
// padded table of tensor_based_replay_buffer.py:179-251); identity: rep = table.
int ensure_shared_table(pa_dqn* h, const pa_arena* arena, int rep_onehot, hipStream_t s) {
  const pa_dqn_desc& d = h->d;
  const int A = arena->d.max_actions, AD = d.action_dim, avd = arena->d.avail_dim;
  if (h->sh_arena == arena && h->sh_gen == arena->shared_gen && h->sh_onehot == rep_onehot &&
      h->sh_A == A)
    return PA_OK;
  const size_t rep_bytes = (size_t)d.max_actions * AD * 4, mask_bytes = (size_t)d.max_actions;
  if (!h->sh_rep) {
    PA_HIP(hipMalloc((void**)&h->sh_rep, rep_bytes));
    PA_HIP(hipMalloc((void**)&h->sh_mask, (mask_bytes + 15) & ~size_t(15)));
    PA_HIP(hipHostMalloc((void**)&h->sh_stage, rep_bytes + mask_bytes, hipHostMallocDefault));
  } else {
    PA_HIP(hipStreamSynchronize(s));   // an earlier copy out of the staging buffer may be in flight
    if (h->side) PA_HIP(hipStreamSynchronize(h->side));
  }
  float* rep = reinterpret_cast<float*>(h->sh_stage);
  for (int i = 0; i < A; ++i)
    for (int j = 0; j < AD; ++j) {
      if (rep_onehot) rep[i * AD + j] = ((long long)arena->sh_next_avail[(size_t)i * avd] == j) ? 1.f : 0.f;
      else rep[i * AD + j] = arena->sh_next_avail[(size_t)i * avd + j];
    }
  memcpy(h->sh_stage + rep_bytes, arena->sh_next_mask, (size_t)A);
  PA_HIP(hipMemcpyAsync(h->sh_rep, rep, (size_t)A * AD * 4, hipMemcpyHostToDevice, s));
  PA_HIP(hipMemcpyAsync(h->sh_mask, h->sh_stage + rep_bytes, (size_t)A, hipMemcpyHostToDevice, s));
  h->sh_arena = arena; h->sh_gen = arena->shared_gen; h->sh_onehot = rep_onehot; h->sh_A = A;
  return PA_OK;
}

int ensure_idx(pa_dqn* h, int64_t n) {
  if (h->idx_cap >= n) return PA_OK;
  // grow geometrically with a floor of 256 rounds: a reallocation is a device synchronisation
  // plus hipFree + hipMalloc (hundreds of microseconds), and it used to land inside the first
  // learn() call that asked for more rounds than the one before it
  const int64_t floor_n = (int64_t)256 * h->d.max_batch;
  if (n < floor_n) n = floor_n;
  if (n < 2 * h->idx_cap) n = 2 * h->idx_cap;
  if (h->idx_all) {
    PA_HIP(hipDeviceSynchronize());
    (void)hipFree(h->idx_all);
    h->idx_all = nullptr;
    h->idx_cap = 0;
  }
  PA_HIP(hipMalloc((void**)&h->idx_all, (size_t)(n * 8)));
  h->idx_cap = n;
  return PA_OK;
}

}  // namespace

extern "C" int64_t pa_dqn_param_count(int32_t S, int32_t AD, int32_t H1, int32_t H2) {
  int64_t off[6], total;
  param_layout(S, AD, H1, H2, off, &total);
  return total;
}

extern "C" int pa_dqn_param_offsets(int32_t S, int32_t AD, int32_t H1, int32_t H2,
                                    int64_t* offsets6) {
  PA_REQUIRE(offsets6, PA_ERR_INVALID, "null output");
  int64_t total;
  param_layout(S, AD, H1, H2, offsets6, &total);
  return PA_OK;
}

extern "C" int pa_dqn_create(pa_dqn** out, const pa_dqn_desc* desc) {
  PA_REQUIRE(out && desc, PA_ERR_INVALID, "pa_dqn_create: null argument");
  PA_REQUIRE(desc->state_dim > 0 && desc->action_dim > 0 && desc->hidden1 > 0 && desc->hidden2 > 0,
             PA_ERR_INVALID, "dimensions must be positive");
  PA_REQUIRE(desc->hidden1 <= 256 && desc->hidden2 <= 256, PA_ERR_UNSUPPORTED,
             "hidden sizes above 256 are not built (got [%d, %d]); the fused target kernel keeps "
             "a 64 x H1 tile and all H2 columns on one CU",
             desc->hidden1, desc->hidden2);
  PA_REQUIRE(desc->max_batch > 0 && desc->max_actions > 0, PA_ERR_INVALID,
             "max_batch / max_actions must be positive");
  const int ndev = pa_device_count();
  PA_REQUIRE(desc->device >= 0 && desc->device < ndev, PA_ERR_HIP,
             "HIP device %d not available (%d visible): the learner step is HIP-only and has no "
             "CPU fallback",
             desc->device, ndev);
  PA_HIP(hipSetDevice(desc->device));
  pa_dqn* h = new (std::nothrow) pa_dqn();
  PA_REQUIRE(h, PA_ERR_NOMEM, "out of host memory");
  {
    int rc_dev = bind_process_device(desc->device);
    if (rc_dev != PA_OK) {
      delete h;
      return rc_dev;
    }
  }
  h->d = *desc;
  h->bound = false;
  memset(&h->bufs, 0, sizeof(h->bufs));
  h->IN = desc->state_dim + desc->action_dim;
  param_layout(desc->state_dim, desc->action_dim, desc->hidden1, desc->hidden2, h->off, &h->P);
  h->timing = 0;
  h->timing_chain = 0;
  h->chain_sample = false;
  h->bb_A = 0;
  memset(&h->bb, 0, sizeof(h->bb));
  h->idx_all = nullptr;
  h->idx_cap = 0;
  h->tick = 0;
  h->H1a = h->H2a = h->dZ2 = h->dZ1 = h->nextv = h->qbuf = h->dq = h->absd =
      h->xpack = h->loss_scratch = h->w2f = h->W1f = h->W2f16 = h->W2tf = nullptr;
  h->w2f_online = h->choice_rep = nullptr;
  h->w2sp = nullptr;
  h->w2sp_online = nullptr;
  h->w1sp = nullptr;
  h->fuse_u = env_int("PEARL_AMD_FUSE_U", 0);
  h->use_split = env_int("PEARL_AMD_TARGET_SPLIT", 1);
  h->dw_tm = env_int("PEARL_AMD_DW_TM", 0);
  h->rp_split = env_int("PEARL_AMD_ROWPASS_SPLIT", 1);
  h->rp_h2 = env_int("PEARL_AMD_ROWPASS_H2", 1);
  h->use_h2t = env_int("PEARL_AMD_TARGET_H2", 1);
  h->pair = env_int("PEARL_AMD_ROWPASS_PAIR", 0);
  h->pair_lds = env_int("PEARL_AMD_PAIR_LDS", 82 * 1024);
  h->pair_live = false;
  h->qx_clean = false;
  h->dZ1p = nullptr; h->qx = nullptr; h->qhalf = nullptr; h->qx_rows = 0;
  h->choice = nullptr;
  h->Uw[0] = h->Uw[1] = h->yw[0] = h->yw[1] = nullptr;
  h->bb_x = nullptr;
  h->side = nullptr;
  h->ev_start = h->ev_tail = h->ev_chain[0] = h->ev_chain[1] = nullptr;
  h->ev_gather[0] = h->ev_gather[1] = nullptr;
  h->sig = nullptr;
  h->sig_gen = 0;
  h->pending_signal = 0;
  h->pending_wait = 0;
  h->use_flags = env_int("PEARL_AMD_FLAG_HOP", 1);
  h->lead_persist = env_int("PEARL_AMD_LEAD_PERSIST", 2);
  h->err_dev = nullptr;
  h->err_host = nullptr;
  h->packed_ok = false;
  h->online_tgt_ok = false;
  h->overlap = env_int("PEARL_AMD_OVERLAP", 1);
  // 12 = "1 round, then 2 rounds, then the rest": measured best with the bf16x3 target kernel
  // (20-round call 23.2 M transitions/s against 22.0 M with one leading piece of 3 rounds)
  h->split_first = env_int("PEARL_AMD_SPLIT_FIRST", 12);
  h->y_clean = false;
  h->reserved_dev = nullptr;
  h->n_reserved = 0;
  h->ncu = 0;
  h->tile_ctr = nullptr;
  h->dbg_workers = nullptr;
  h->dbg_launches = 0;
  h->ctr_next = 0;
  h->sh_rep = nullptr; h->sh_mask = nullptr; h->sh_stage = nullptr;
  h->sh_arena = nullptr; h->sh_gen = -1; h->sh_onehot = -1; h->sh_A = 0;
  h->sh_enable = env_int("PEARL_AMD_SHARED_TABLE", 1);
  h->prof_row = h->prof_dw = nullptr;
  h->prof_tgt = nullptr;
  h->prof_tgt_tiles = 0;
  h->prof_round = h->cur_round = -1;
  const int64_t B = desc->max_batch;
  // learn() evaluates the target network for a whole window of rounds in one launch (the target
  // parameters only change every target_update_freq rounds): up to 16 rounds / 16384 transitions
  h->wcap = (int)(16384 / B);
  h->wcap = h->wcap < 1 ? 1 : (h->wcap > 16 ? 16 : h->wcap);
  h->wrows = (int64_t)h->wcap * B;
#define PA_WS(ptr, floats)                                                       \
  do {                                                                           \
    hipError_t _e = hipMalloc((void**)&(ptr), (size_t)((floats) * 4));           \
    if (_e != hipSuccess) {                                                      \
      set_error("hipMalloc(workspace) failed: %s", hipGetErrorString(_e));       \
      pa_dqn_destroy(h);                                                         \
      return PA_ERR_NOMEM;                                                       \
    }                                                                            \
  } while (0)
  PA_WS(h->Uw[0], h->wrows * desc->hidden1);
  PA_WS(h->Uw[1], h->wrows * desc->hidden1);
  PA_WS(h->H1a, B * desc->hidden1);
  PA_WS(h->H2a, B * desc->hidden2);
  PA_WS(h->dZ2, B * desc->hidden2);
  PA_WS(h->dZ1, B * desc->hidden1);
  PA_WS(h->yw[0], h->wrows);
  PA_WS(h->yw[1], h->wrows);
  PA_WS(h->nextv, h->wrows);
  // one allocation: [kTileCtrs] work-stealing counters | 4-int error word (one memset per call)
  PA_WS(h->tile_ctr, kTileCtrs + 8);
  h->err_dev = h->tile_ctr + kTileCtrs;
  h->sig = h->tile_ctr + kTileCtrs + 4;     // outside the per-call memset
  {
    hipDeviceProp_t prop;
    if (hipMemset(h->tile_ctr, 0, (kTileCtrs + 8) * sizeof(int)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess ||
        hipGetDeviceProperties(&prop, desc->device) != hipSuccess) {
      set_error("pa_dqn_create: device query failed");
      pa_dqn_destroy(h);
      return PA_ERR_HIP;
    }
    h->ncu = prop.multiProcessorCount;
  }
  h->pingpong = env_int("PEARL_AMD_PINGPONG", 1);
  if (env_int("PEARL_AMD_DEBUG_WORKERS", 0)) {
    float* g = nullptr;
    PA_WS(g, 4);
    h->dbg_workers = reinterpret_cast<int*>(g);
    (void)hipMemset(h->dbg_workers, 0, 16);
  }
  if (hipHostMalloc((void**)&h->err_host, 16, hipHostMallocDefault) != hipSuccess) {
    set_error("pa_dqn_create: error-word allocation failed");
    pa_dqn_destroy(h);
    return PA_ERR_NOMEM;
  }
  h->err_host[0] = 0;
  PA_WS(h->qbuf, B);
  PA_WS(h->dq, B);
  PA_WS(h->absd, B);
  PA_WS(h->xpack, B * h->IN);
  PA_WS(h->loss_scratch, 4);
  PA_WS(h->w2f, w2f_floats(desc->hidden2, desc->hidden1));
  if (desc->hidden1 == TS_H && desc->hidden2 == TS_H) {
    float* sp = nullptr;
    PA_WS(sp, w2sp_bytes() / 4);
    h->w2sp = sp;
    if ((desc->state_dim & 15) == 0 && desc->state_dim <= TS_H) {
      float* sp1 = nullptr;
      PA_WS(sp1, wsp_bytes(desc->state_dim >> 4) / 4);
      h->w1sp = sp1;
    }
  }
  // (not beside PEARL_AMD_FUSE_U: that experiment's tiles form U with the bf16x3 arithmetic)
  if (h->use_h2t && h->use_split && !h->fuse_u && desc->hidden1 == TS_H && desc->hidden2 == TS_H && desc->double_q == 0) {
    float* w = nullptr;
    PA_WS(w, w2h_bytes() / 4);
    h->w2h = w;
    PA_WS(w, TS_H);
    h->w2hf = reinterpret_cast<int*>(w);
  } else {
    h->use_h2t = 0;
  }
  PA_WS(h->W1f, wf16_floats(desc->hidden1, h->IN));
  PA_WS(h->W2f16, wf16_floats(desc->hidden2, desc->hidden1));
  PA_WS(h->W2tf, wf16_floats(desc->hidden1, desc->hidden2));
  if (h->rp_h2 && desc->hidden1 == HF_UNITS && desc->hidden2 == HF_UNITS && wf16_nkg(h->IN) == 9) {
    float* w = nullptr;
    PA_WS(w, 2 * HF_UMAX);
    h->umax = reinterpret_cast<unsigned*>(w);
    PA_HIP(hipMemset(h->umax, 0, sizeof(unsigned) * 2 * HF_UMAX));
  }
  if (h->pair && desc->hidden1 == PR_H && desc->hidden2 == PR_H && wf16_nkg(h->IN) == PR_NG1) {
    float* w = nullptr;
    h->qx_rows = (int)round_up(B, RP_ROWS);
    PA_WS(h->dZ1p, 2 * B * desc->hidden1);
    PA_WS(w, (int64_t)2 * h->qx_rows);
    h->qx = reinterpret_cast<unsigned*>(w);
    PA_WS(h->qhalf, 2 * B);
  } else {
    h->pair = 0;
  }
  if (desc->double_q == 1) {
    if (h->w2sp) {
      float* sp = nullptr;
      PA_WS(sp, w2sp_bytes() / 4);
      h->w2sp_online = sp;
    }
    PA_WS(h->w2f_online, w2f_floats(desc->hidden2, desc->hidden1));
    PA_WS(h->choice, B);
    PA_WS(h->choice_rep, B * desc->action_dim);
  }
#undef PA_WS
  *out = h;
  return PA_OK;
}

extern "C" int pa_dqn_destroy(pa_dqn* h) {
  if (!h) return PA_OK;
  (void)hipSetDevice(h->d.device);
  (void)hipDeviceSynchronize();
  void* ptrs[] = {h->Uw[0], h->Uw[1], h->H1a, h->H2a, h->dZ2, h->dZ1, h->yw[0], h->yw[1], h->nextv,
                  h->qbuf, h->dq, h->absd, h->xpack, h->loss_scratch, h->idx_all, h->w2f, h->W1f,
                  h->W2f16, h->W2tf, h->reserved_dev, h->tile_ctr, h->w2f_online, h->w2sp, h->w2sp_online, h->w1sp,
                  h->choice, h->choice_rep, h->dZ1p, h->qx, h->qhalf, h->dbg_workers, h->umax, h->w2h, h->w2hf};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->err_host) (void)hipHostFree(h->err_host);
  if (h->sh_stage) (void)hipHostFree(h->sh_stage);
  if (h->sh_rep) (void)hipFree(h->sh_rep);
  if (h->sh_mask) (void)hipFree(h->sh_mask);
  free_batchbufs(h);
  if (h->side) (void)hipStreamDestroy(h->side);
  hipEvent_t evs[] = {h->ev_start, h->ev_tail, h->ev_chain[0], h->ev_chain[1], h->ev_gather[0],
                      h->ev_gather[1]};
  for (hipEvent_t e : evs)
    if (e) (void)hipEventDestroy(e);
  for (auto& t : h->timers)
    for (auto e : t.ev) (void)hipEventDestroy(e);
  delete h;
  release_process_device();
  return PA_OK;
}

extern "C" int pa_dqn_bind(pa_dqn* h, const pa_dqn_buffers* bufs) {
  PA_REQUIRE(h && bufs, PA_ERR_INVALID, "pa_dqn_bind: null argument");
  PA_REQUIRE(bufs->q && bufs->q_target && bufs->grad && bufs->exp_avg && bufs->exp_avg_sq,
             PA_ERR_INVALID, "pa_dqn_bind: null buffer");
  PA_REQUIRE(!h->d.amsgrad || bufs->max_exp_avg_sq, PA_ERR_INVALID,
             "amsgrad needs max_exp_avg_sq");
  const float* all[] = {bufs->q, bufs->q_target, bufs->grad, bufs->exp_avg, bufs->exp_avg_sq};
  for (const float* p : all)
    PA_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, PA_ERR_INVALID,
               "flat buffers must be 16-byte aligned");
  h->bufs = *bufs;
  h->bound = true;
  h->packed_ok = false;
  h->online_tgt_ok = false;
  return PA_OK;
}

// The flat parameter buffers were written from outside the library (torch in-place ops): derived
// copies are rebuilt by the next call that needs them.
extern "C" int pa_dqn_invalidate(pa_dqn* h) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  h->packed_ok = false;
  h->online_tgt_ok = false;
  return PA_OK;
}

extern "C" int pa_dqn_qvalues(pa_dqn* h, const pa_dqn_batch* batch, float* q_out,
                              float* next_v_out, float* target_out, void* stream) {
  int rc = check_batch(h, batch);
  if (rc != PA_OK) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(h->d.device));
  rc = run_repack(h, q_out != nullptr, next_v_out || target_out, s);
  if (rc != PA_OK) return rc;
  if (next_v_out || target_out) {
    rc = run_next_values(h, batch, next_v_out, target_out, s);
    if (rc != PA_OK) return rc;
  }
  if (q_out) {
    const float* x = nullptr;
    rc = resolve_x(h, batch, &x, s);
    if (rc != PA_OK) return rc;
    rc = run_rowpass(h, x, batch->B, nullptr, false, q_out, 1, s);
    if (rc != PA_OK) return rc;
  }
  return PA_OK;
}

extern "C" int pa_dqn_update_target(pa_dqn* h, void* stream) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "learner has no bound parameter buffers");
  PA_HIP(hipSetDevice(h->d.device));
  h->packed_ok = false;
  h->online_tgt_ok = false;
  return run_soft_update(h, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_dqn_step(pa_dqn* h, const pa_dqn_batch* batch, int32_t do_target_update,
                           int64_t adam_step, int32_t grad_world, float* mean_abs_td_out,
                           void* stream) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  PA_HIP(hipSetDevice(h->d.device));
  h->packed_ok = false;
  h->online_tgt_ok = false;
  return step_impl(h, batch, do_target_update, adam_step, grad_world, mean_abs_td_out,
                   reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_dqn_apply(pa_dqn* h, int64_t adam_step, void* stream) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "learner has no bound parameter buffers");
  PA_HIP(hipSetDevice(h->d.device));
  h->packed_ok = false;
  h->online_tgt_ok = false;
  return run_adamw(h, adam_step, 0, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pa_dqn_learn(pa_dqn* h, pa_arena* arena, const pa_learn_args* args, void* stream) {
  PA_REQUIRE(h && h->bound, PA_ERR_INVALID, "learner has no bound parameter buffers");
  PA_REQUIRE(arena && args, PA_ERR_INVALID, "pa_dqn_learn: null argument");
  const pa_dqn_desc& d = h->d;
  const int B = args->batch_size;
  const int A = arena->d.max_actions;
  PA_REQUIRE(args->rounds >= 0, PA_ERR_INVALID, "negative rounds");
  PA_REQUIRE(B > 0 && B <= d.max_batch, PA_ERR_INVALID, "batch size %d outside (0, %d]", B,
             d.max_batch);
  PA_REQUIRE((int64_t)B <= arena->size, PA_ERR_VALUE,
             "Can't get a batch of size %d from a replay buffer with only %lld elements", B,
             (long long)arena->size);
  PA_REQUIRE(A > 0 && A <= d.max_actions && A <= T_ROWS, PA_ERR_UNSUPPORTED,
             "arena max_actions %d outside (0, min(%d, %d)]", A, d.max_actions, T_ROWS);
  PA_REQUIRE(arena->d.state_dim == d.state_dim, PA_ERR_INVALID, "state_dim mismatch: %d vs %d",
             arena->d.state_dim, d.state_dim);
  PA_REQUIRE(arena->d.has_next_state, PA_ERR_INVALID, "arena stores no next_state");
  PA_REQUIRE(arena->d.device == d.device, PA_ERR_INVALID, "arena and learner on different devices");
  PA_REQUIRE(args->target_update_freq > 0, PA_ERR_INVALID, "target_update_freq must be positive");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PA_HIP(hipSetDevice(d.device));
  int rc = ensure_batchbufs(h, A);
  if (rc != PA_OK) return rc;
  rc = pa_arena_flush(arena, stream);
  if (rc != PA_OK) return rc;
  if (args->rounds == 0) return PA_OK;
  const int R = args->rounds;
  rc = ensure_idx(h, (int64_t)R * B);
  if (rc != PA_OK) return rc;
  const bool dp = args->allreduce_start != nullptr;
  const int world = dp ? (args->grad_world > 0 ? args->grad_world : 1) : 1;
  // Two streams: the target-network side of a window (gather of the target inputs, U, Bellman
  // targets y for every round of the window) runs on the low-priority side stream `t`; the main
  // stream runs the only truly sequential part, the per-round online chain.  The chain needs
  // nothing from `t` except y, which travels as data-tagged 4-byte granules (write-through
  // stores in target_fused_kernel, polled L1-bypassing loads in online_rowpass_kernel): round j
  // starts as soon as ITS targets exist, while the targets of rounds j+1.. are still being
  // computed on the CUs the chain leaves idle.  Stream-level ordering (events) is only needed
  // once per window: the next window's target pass must see the soft update that the last chain
  // of this window performs.  Bit-identical to the single-stream loop (same kernels, same data).
  // Double DQN chooses the next action with the ONLINE network, which moves every round: no
  // window batching and nothing to overlap — each round's targets need the previous round's step.
  PA_REQUIRE(d.double_q != 2, PA_ERR_UNSUPPORTED,
             "pa_dqn_learn: a SARSA learner steps through pa_dqn_step (its batches carry the "
             "committed next action, which the arena does not store)");
  const bool dbl = d.double_q != 0;
  const bool overlap = h->overlap && h->timing < 2 && !dbl;
  // Double DQN: the rounds stay sequential, but the INPUTS of a window of rounds — x, next states,
  // rewards, terminals, tables: none depends on the parameters — are still gathered by one launch
  // (PEARL_AMD_DDQN_WINDOW=0: one gather per round, as before; same values either way)
  const bool ddqn_window = env_int("PEARL_AMD_DDQN_WINDOW", 1) != 0;
  const int wcap = (dbl && !ddqn_window) ? 1 : h->wcap;
  // ... and the value pass Q_target(s', a*) of a round runs on the side stream beside the row
  // pass's forward half (data-tagged y, as in the DQN loop; PEARL_AMD_DDQN_OVERLAP=0: in front of it)
  const bool dbl2 = dbl && h->overlap && h->timing < 2 && h->use_flags &&
                    env_int("PEARL_AMD_DDQN_OVERLAP", 1) != 0;
  rc = ensure_side(h);
  if (rc != PA_OK) return rc;
  // fresh work-stealing counters for this call's persistent target launches, and a clean error
  // word (a bounded wait that expired in an earlier call must not poison this one): zeroed by the
  // prologue launch below (device sampler) or one memset (host index lists)
  if (args->idx_host) PA_HIP(hipMemsetAsync(h->tile_ctr, 0, (kTileCtrs + 4) * sizeof(int), s));
  h->ctr_next = 0;
  if (h->err_host[0]) h->qx_clean = false;   // a bounded wait expired: exchange words may hold anything
  h->err_host[0] = 0;
  // Static action space: every stored row carries the same padded next-action table, so the
  // target pass reads ONE [A, AD] table with stride 0 and the window gather neither reads the
  // per-row tables nor writes (rows, A, AD) one-hot rows (1 120 B less per transition on the
  // gather, 1 040 B less read per transition by the target kernel).  Same values: bit-identical.
  const bool shared_tab = h->sh_enable && arena->shared_next == 1 &&
                          (args->rep_onehot ? arena->d.avail_dim == 1 : arena->d.avail_dim == d.action_dim);
  if (shared_tab) {
    rc = ensure_shared_table(h, arena, args->rep_onehot, s);   // ordered before the hop to `t` below
    if (rc != PA_OK) return rc;
  }
  hipStream_t t = overlap ? h->side : s;
  int start_gen = 0;      // generation the first x gather publishes (call-start hand-off), 0: none
  ScopedTimer tm_all(h, "learn", s);
  // PolicyLearner.learn pre-increments _training_steps (policy_learner.py:183);
  // forward() soft-updates when (_training_steps + 1) % freq == 0 (:283-284).
  auto due = [&](int r) { return ((args->training_steps0 + r + 2) % args->target_update_freq) == 0; };
  // the first round's soft update runs stand-alone; later ones ride the previous optimizer launch
  if (due(0)) {
    rc = run_soft_update(h, s);
    if (rc != PA_OK) return rc;
  }
  // ---- the index lists of EVERY round in one go (they do not depend on the parameters), and
  // every packed weight copy rebuilt from the parameters as they are NOW (learn_prologue_kernel)
  RepackArgs rpk;
  memset(&rpk, 0, sizeof(rpk));
  rpk.q = h->bufs.q; rpk.q_target = h->bufs.q_target;
  rpk.off_w1 = h->off[0]; rpk.off_w2 = h->off[2];
  rpk.IN = h->IN; rpk.H1 = d.hidden1; rpk.H2 = d.hidden2;
  rpk.pk = packed(h);
  rpk.do_online = 1; rpk.do_target = 1;
  repack_umax(h, rpk);
  if (args->idx_host) {
    for (int64_t i = 0; i < (int64_t)R * B; ++i)
      PA_REQUIRE(args->idx_host[i] >= 0 && args->idx_host[i] < arena->size, PA_ERR_INVALID,
                 "index %lld out of range [0, %lld)", (long long)args->idx_host[i],
                 (long long)arena->size);
    PA_HIP(hipMemcpyAsync(h->idx_all, args->idx_host, (size_t)R * B * 8, hipMemcpyHostToDevice, s));
    rc = run_repack(h, true, true, s);
    if (rc != PA_OK) return rc;
  } else {
    ScopedTimer tm(h, "sample", s);
    rc = sample_check(arena->size, B);
    if (rc != PA_OK) return rc;
    const SampleArgs sa = sample_args(arena->size, args->seed, args->offset0, B, h->idx_all);
    static size_t configured = 0;
    const size_t smem = sample_smem_bytes(sa.hs);
    if (smem > configured) {
      rc = set_max_smem(learn_prologue_kernel, smem);
      if (rc != PA_OK) return rc;
      configured = smem;
    }
    hipLaunchKernelGGL(learn_prologue_kernel, dim3((unsigned)(R + kPrologueRepackWgs)),
                       dim3(SAMPLE_THREADS), smem, s, sa, R, rpk, h->tile_ctr, kTileCtrs + 4);
    PA_LAUNCH_CHECK();
  }
  h->packed_ok = false;
  h->online_tgt_ok = false;   // until this call has completed its last round
  if (overlap) {
    // Tagged hand-off invariant: every word of both y buffers is kYPendingBits whenever no target
    // pass is in flight.  The consumer (online_rowpass_kernel) restores the tag after reading, so
    // back-to-back learn() calls need no refill; anything else that wrote y (pa_dqn_step,
    // pa_dqn_qvalues, the single-stream loop) marks the buffers dirty.
    if (!h->y_clean) {
      for (int p = 0; p < 2; ++p)
        PA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->yw[p]), (int)kYPendingBits,
                                 (size_t)h->wrows, s));
      h->y_clean = true;
    }
    // the side stream starts behind everything enqueued on `s` so far (index lists, repack): with
    // the device-word hand-off the word is published by the first launch of the main stream's
    // head, the gather of x (no signal launch of its own)
    if (h->use_flags) {
      start_gen = ++h->sig_gen;
      hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, t, h->sig, start_gen, h->err_dev, h->err_host,
                         (int*)nullptr, 0);
      PA_LAUNCH_CHECK();
    } else {
      rc = stream_hop(h, s, t, h->ev_start);
      if (rc != PA_OK) return rc;
    }
  } else if (dbl2) {
    if (!h->y_clean) {
      for (int p = 0; p < 2; ++p)
        PA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->yw[p]), (int)kYPendingBits,
                                 (size_t)h->wrows, s));
      h->y_clean = true;
    }
  } else {
    h->y_clean = false;
  }
  // The target network is constant between two soft updates, and the index lists of all rounds
  // are already known: gather and run the (dominant) target-network pass for a whole WINDOW of
  // rounds at once.
  bool head_emitted = false, front_emitted = false;   // first window's main-stream head (see emit_head)
  int r = 0, k = 0;
  while (r < R) {
    int w = 1;
    while (r + w < R && w < wcap && !due(r + w)) ++w;
    const int rows = w * B;
    const int p = k & 1;
    const pa_dqn::BatchBuf& bb = h->bb[p];
    if (h->timing) h->tick++;
    // level-1 timers: every 4th window of a long call, every window of a short one (its target
    // launches but the first — that one opens the window's critical path — and its gather), so that
    // even the driver's 20-round call carries four or more sampled launches
    const bool short_call = R < 100;
    const bool sample_w = short_call || (k % 4) == 0;
    // main-stream head of the call's first window (see the hook in the piece loop below): x of the
    // whole window, then the front half of round 0's chain
    float* xwin = h->bb_x + (overlap ? (int64_t)p * h->wrows * h->IN : 0);
    static const bool no_chain_dbg = env_int("PEARL_AMD_DEBUG_NO_CHAIN", 0) != 0;
    const int gw_chain = dp ? -world : 1;
    auto emit_gather_x = [&]() -> int {
      pa_batch_out o;
      memset(&o, 0, sizeof(o));
      o.x = xwin;
      o.rep_dim = d.action_dim;
      o.rep_onehot = args->rep_onehot;
      ScopedTimer tm(h, "gather_x", s, 2, 1, rows);
      const int gen = start_gen;
      start_gen = 0;
      return arena_gather_device(arena, h->idx_all + (int64_t)r * B, rows, &o, s,
                                 gen ? h->sig : nullptr, gen);
    };
    // (round 6, profiles/r06_c_first_window.txt) two orders / forms of the call's first chain launch:
    // PEARL_AMD_HEAD_EARLY (default 1): the head goes out right behind the gather of x, BEFORE the side
    // stream's gather / U / first target piece (the host is the pacemaker of a call's first 50 us:
    // each launch costs it ~5 us, and the forward half needs nothing from the side stream) — a 1-round
    // call 142.9 -> 135.9 us, a 20-round call 845 -> 835 us; PEARL_AMD_FIRST_UNSPLIT=1: the call's very
    // first row pass as ONE launch that waits for its targets inside (tagged y) instead of forward +
    // backward launches — measured, no gain (141.5 / 842 us), off
    static const bool head_early = env_int("PEARL_AMD_HEAD_EARLY", 1) != 0;
    static const bool first_unsplit = env_int("PEARL_AMD_FIRST_UNSPLIT", 0) != 0;
    auto emit_head = [&]() -> int {
      head_emitted = true;
      if (no_chain_dbg) return PA_OK;
      h->cur_round = r;
      front_emitted = true;
      return chain_front(h, xwin, B, h->yw[p], true, gw_chain, s, !first_unsplit);
    };
    // (x of the call's first window: it needs nothing from the side stream, so it goes out first —
    // one launch ahead of the side stream's critical gather -> U -> first target piece)
    if (overlap && k == 0) {
      rc = emit_gather_x();
      if (rc != PA_OK) return rc;
      if (head_early && !dbl) {
        rc = emit_head();
        if (rc != PA_OK) return rc;
      }
    }
    // ---- side stream: target inputs of the window
    {
      pa_batch_out o;
      memset(&o, 0, sizeof(o));
      o.next_state = bb.next_state;
      o.next_avail_rep = shared_tab ? nullptr : bb.next_avail_rep;
      o.next_mask = shared_tab ? nullptr : bb.next_mask;
      o.reward_f32 = bb.reward;
      o.terminated = bb.term;
      o.rep_dim = d.action_dim;
      o.rep_onehot = args->rep_onehot;
      // one gather per window: the chain's x rides the same launch (it used to be a second
      // gather on the main stream, at the head of every window's bubble)
      // (not for the first window of a call: there the main stream gathers x itself, so that
      // its first row pass needs no cross-stream wait and is resident before the target grid)
      if (!overlap || k > 0) o.x = h->bb_x + (overlap ? (int64_t)p * h->wrows * h->IN : 0);
      // level 1 samples the gather like the target launches (every 4th window; every window of a
      // short call): bench.py's HBM roofline of the sample + gather kernel.  "gather_nox": the
      // launch of a call's first window, which leaves x to the main stream.
      // (not the first window's of an overlapped call: that gather opens the call's critical path,
      // and an event record costs ~6 us of idle stream)
      ScopedTimer tm(h, o.x ? "gather" : "gather_nox", t, (sample_w && o.x) ? 1 : 2, 1, rows);
      rc = arena_gather_device(arena, h->idx_all + (int64_t)r * B, rows, &o, t);
      if (rc != PA_OK) return rc;
      if (overlap && k > 0) PA_HIP(hipEventRecord(h->ev_gather[p], t));
    }
    // the previous window's last optimizer launch (soft update of the target net) must be done
    if (overlap && k > 0) {
      if (h->use_flags) {
        // The word is published by the FIRST row-pass launch of this window, on the main stream:
        // it starts when the previous window's last optimizer launch (soft update) has completed,
        // it is resident on the chain's CUs before the target grid below takes every free slot,
        // and it costs no launch of its own.
        const int gen = ++h->sig_gen;
        h->pending_signal = gen;
        // (PEARL_AMD_X_FLAG=1: measured, no difference — 27.60 against 27.55 M transitions/s over 2000 rounds
        //  on one box, 20-round calls within noise — so the event of rounds 1-4 stays the default)
        static const bool x_by_flag = env_int("PEARL_AMD_X_FLAG", 0) != 0;
        h->pending_wait = x_by_flag ? gen : 0;
        if (h->use_h2t && h->w2h)
          hipLaunchKernelGGL(wait_pack_kernel, dim3(64), dim3(256), 0, t, h->sig, gen, h->err_dev, h->err_host,
                             x_by_flag ? h->sig + 1 : nullptr, gen, w2h_pack_args(h));
        else
          hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, t, h->sig, gen, h->err_dev, h->err_host,
                             x_by_flag ? h->sig + 1 : nullptr, gen);
        PA_LAUNCH_CHECK();
      } else {
        PA_HIP(hipStreamWaitEvent(t, h->ev_chain[(k - 1) & 1], 0));
        if (h->use_h2t && h->w2h) {
          hipLaunchKernelGGL(target_pack_kernel, dim3(64), dim3(256), 0, t, w2h_pack_args(h));
          PA_LAUNCH_CHECK();
        }
      }
    } else if (!overlap && k > 0 && h->use_h2t && h->w2h) {
      // single stream: the previous window's last optimizer launch (soft update) is simply earlier on `s`
      hipLaunchKernelGGL(target_pack_kernel, dim3(64), dim3(256), 0, t, w2h_pack_args(h));
      PA_LAUNCH_CHECK();
    }
    // U and the Bellman targets; the first round of the window as its own pair of launches, so
    // the chain can start after one round's worth of target work instead of the whole window's
    // (the first piece runs as a classic grid on every CU — the chain is idle then anyway — the
    // rest as persistent tiles that stay off the chain's CUs)
    // Piece schedule of the overlapped loop (split_first = 12 means "1 round, then 2 rounds, then
    // the rest"): the first pieces are classic grids on every CU — the chain is still waiting for
    // its first targets — the last one runs as persistent tiles that stay off the chain's CUs.
    // Small leading pieces shorten the wait of round 0 (one round = one workgroup per CU, ~20 us),
    // and the second piece is sized so that it is done when round 0's chain is.
    int sched[4] = {w, 0, 0, 0}, npieces = 1;
    if (overlap && h->split_first > 0) {
      int digits[3], nd = 0;
      for (int v = h->split_first; v > 0 && nd < 3; v /= 10) digits[nd++] = v % 10;
      int used = 0;
      npieces = 0;
      for (int i = nd - 1; i >= 0; --i)
        if (digits[i] > 0 && used + digits[i] < w) { sched[npieces++] = digits[i]; used += digits[i]; }
      sched[npieces++] = w - used;
    }
    const bool persist = env_int("PEARL_AMD_PERSIST", overlap ? 1 : 0) != 0;
    int pc = 0;
    for (int j0 = 0; j0 < w; ++pc) {
      const int nj = sched[pc];
      const int64_t row0 = (int64_t)j0 * B;
      const int prow = nj * B;
      pa_dqn_batch b;
      memset(&b, 0, sizeof(b));
      b.B = prow; b.A = A;
      b.reward = bb.reward + row0;
      b.terminated = bb.term + row0;
      b.next_state = bb.next_state + row0 * d.state_dim;
      if (shared_tab) {
        b.next_avail_rep = h->sh_rep;
        b.next_mask = h->sh_mask;
        b.next_avail_bcast = 1;
      } else {
        b.next_avail_rep = bb.next_avail_rep + row0 * A * d.action_dim;
        b.next_mask = bb.next_mask + row0 * A;
      }
      float* Up = h->Uw[p] + row0 * d.hidden1;
      if (dbl) {   // per round, inside the chain loop below: the argmax pass needs the previous round's step
        j0 += nj;
        continue;
      }
      // U = s' W1s'^T + b1': ONE launch for all leading (classic-grid) pieces together — a launch
      // per piece put a 7-20 us GEMM between every two target launches on this stream — and one
      // for the persistent remainder
      // (batches that qualify form U with the split tile's arithmetic: in the tile for the leading,
      // latency-bound pieces, as one u_split_kernel launch for a long persistent remainder — the
      // same bits either way)
      const bool fuses = target_fuses_u(h, &b);
      const bool read_u = fuses && pc == npieces - 1 && nj >= 4;
      if (read_u) {
        rc = run_u_split(h, &b, Up, t);
        if (rc != PA_OK) return rc;
      }
      if ((pc == 0 || pc == npieces - 1) && !fuses) {
        int cover = nj;
        if (pc == 0 && npieces > 1) {
          cover = 0;
          for (int q = 0; q < npieces - 1; ++q) cover += sched[q];
        }
        ScopedTimer tm(h, "target_l1", t, 2, 1, cover * B);
        GemmArgs g = target_l1_problem(h, b.next_state, cover * B, Up);
        // PEARL_AMD_U_EXCLUSIVE=1 (experiment): the remainder's U launch — a classic grid over every
        // CU — asks for more LDS than a CU with a resident row-pass workgroup has left (163 840 -
        // 35 840 B), so it stays off the chain's CUs while a row pass runs
        static const int u_excl = env_int("PEARL_AMD_U_EXCLUSIVE", 0);
        const size_t pad = (u_excl && persist && pc == npieces - 1 && npieces > 1) ? (size_t)129024 : 0;
        rc = launch_linear<false>(&g, 1, t, pad);
        if (rc != PA_OK) return rc;
      }
      // Leading pieces: a classic grid takes every CU, the chain's too — the row pass that is
      // already resident keeps its CUs, but the weight-gradient launch behind it then queues for
      // slots (measured: 31-52 us instead of 17).  lead_persist runs them as work-stealing tiles of
      // the two-workgroups-per-CU kernel that stay off the reserved CUs, like the remainder.
      const bool last = pc == npieces - 1;
      const bool lead_p = !last && persist && (h->lead_persist == 1 || (h->lead_persist == 2 && pc >= 1));
      static const int prio = env_int("PEARL_AMD_PRIO_FIRST", 1);
      // PEARL_AMD_LEAD_ROWS=32: the leading (latency-bound, 16- / 32-tile) pieces on the 32-row,
      // four-wave tile — twice the workgroups, half the rows each, bitwise the 64-row tile
      // (1: only the window's first piece)
      static const int lead_rows = env_int("PEARL_AMD_LEAD_ROWS", 0);
      const int hint = (!last && (lead_rows == 32 || (lead_rows == 1 && pc == 0))) ? 32 : 0;
      rc = run_target_fused_u(h, &b, Up, nullptr, h->yw[p] + row0, t, (persist && last) || lead_p,
                              nullptr, sample_w && (last || (short_call && pc >= 1)), lead_p,
                              (prio && pc == 0 && !last) ? B : 0, read_u, hint);
      if (rc != PA_OK) return rc;
      j0 += nj;
      // The call's first window: the host is the pacemaker here (nothing is queued ahead), and the
      // chain's own head — gather of x, forward of round 0: 25 us of device time — must not wait
      // behind the ~45 us of host time the remaining side-stream launches of the window take to
      // enqueue (rocprof: x was gathered 45 us after the first targets existed).  So: first
      // target piece, THEN the main stream's head, then the rest of the window's target work.
      if (overlap && k == 0 && pc == 0 && !head_emitted && !no_chain_dbg) {
        rc = emit_head();
        if (rc != PA_OK) return rc;
      }
    }
    // ---- main stream: the per-round chains.  x of the window was gathered by the side stream
    // (long ago for every window but the first): the event is normally already complete
    if (overlap && k > 0) {
      // (with the device-word hand-off the window's first row pass polls sig[1] itself: pending_wait)
      if (!h->pending_wait) PA_HIP(hipStreamWaitEvent(s, h->ev_gather[p], 0));
    } else if (overlap && !head_emitted) {
      rc = emit_head();      // (no leading piece ran the hook: a one-piece window)
      if (rc != PA_OK) return rc;
    }
    // diagnostics only: the target side of the loop with the chain left out, to tell the target
    // kernel's own speed on its share of the chip from co-run interference
    const bool no_chain = no_chain_dbg;
    if (no_chain) h->y_clean = false;
    if (no_chain && h->pending_signal) {
      hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, s, h->sig, h->pending_signal);
      h->pending_signal = 0;
    }
    for (int j = 0; j < w && !no_chain; ++j) {
      const int round = r + j;
      h->cur_round = round;
      const int soft_next = (round + 1 < R) ? due(round + 1) : 0;
      const float* xj = xwin + (int64_t)j * B * h->IN;
      const float* yj = h->yw[p] + (int64_t)j * B;
      float* lo = args->losses_out ? args->losses_out + round : nullptr;
      const bool split_rp = overlap && j == 0 && !(k == 0 && first_unsplit);
      if (dbl) {
        // Q_target(s', argmax_a Q_online(s', a)) and the Bellman targets of THIS round, with the
        // online parameters as the previous round left them
        const int64_t row0 = (int64_t)j * B;
        pa_dqn_batch b;
        memset(&b, 0, sizeof(b));
        b.B = B; b.A = A;
        b.reward = bb.reward + row0;
        b.terminated = bb.term + row0;
        b.next_state = bb.next_state + row0 * d.state_dim;
        if (shared_tab) {
          b.next_avail_rep = h->sh_rep;
          b.next_mask = h->sh_mask;
          b.next_avail_bcast = 1;
        } else {
          b.next_avail_rep = bb.next_avail_rep + row0 * A * d.action_dim;
          b.next_mask = bb.next_mask + row0 * A;
        }
        rc = run_double_targets(h, &b, nullptr, h->yw[p] + row0, s, dbl2 ? h->side : nullptr);
        if (rc != PA_OK) return rc;
      }
      const bool tagged = overlap || dbl2;
      // roofline.chain: the two chain launches of one round in the middle of a sampled window
      // (four event records, ~6 us of idle each: one round in forty carries them)
      h->chain_sample = h->timing_chain && h->timing == 1 && sample_w && !split_rp && !dbl && !dp &&
                        j == (w > 1 ? w / 2 : 0) && !(k == 0 && j == 0 && front_emitted);
      struct ChainSampleOff { pa_dqn* h; ~ChainSampleOff() { h->chain_sample = false; } } cs_off{h};
      if (!(k == 0 && j == 0 && front_emitted)) {
        rc = chain_front(h, xj, B, yj, tagged, gw_chain, s, split_rp);
        if (rc != PA_OK) return rc;
      }
      if (!dp) {
        rc = chain_back(h, xj, B, yj, tagged, args->adam_step0 + round + 1, 1, lo, soft_next, s, split_rp);
        if (rc != PA_OK) return rc;
        continue;
      }
      // data parallel: local gradients (pre-scaled by 1/world) -> SUM all-reduce -> AdamW.  The
      // exchange hides behind the target work of the side stream.
      rc = chain_back(h, xj, B, yj, tagged, args->adam_step0 + round + 1, -world, lo, 0, s, split_rp);
      if (rc != PA_OK) return rc;
      {
        // "allreduce": the gradient exchange as the learner stream sees it — from the point the
        // stream reaches allreduce_start to the point its wait on the exchange is over — on one
        // mid-window round of every sampled window (level 1; bench.py's comm.exchange_us)
        ScopedTimer tm_ar(h, "allreduce", s, (sample_w && j == (w > 1 ? w / 2 : 0)) ? 1 : 2, 1, B);
        PA_REQUIRE(args->allreduce_start(args->allreduce_ctx, h->bufs.grad, h->P, stream) == 0,
                   PA_ERR_HIP, "allreduce_start hook failed");
        if (args->allreduce_wait)
          PA_REQUIRE(args->allreduce_wait(args->allreduce_ctx, stream) == 0, PA_ERR_HIP,
                     "allreduce_wait hook failed");
      }
      rc = run_adamw(h, args->adam_step0 + round + 1, soft_next, s);
      if (rc != PA_OK) return rc;
    }
    if (overlap) {
      // the next window's target pass needs this window's last optimizer launch (soft update):
      // with flags the next window's first row pass says so (above)
      if (!h->use_flags) PA_HIP(hipEventRecord(h->ev_chain[p], s));
    }
    r += w;
    ++k;
  }
  // single-process rounds end in weight_grad_kernel's fused optimizer epilogue, which refreshes
  // every packed copy (target included, on soft-update rounds)
  h->packed_ok = !dp && !dbl;
  // (Double DQN with the tagged hand-off: every round's targets were consumed and their tags
  //  restored by that round's weight-gradient launch, whatever run_double_targets noted)
  if (dbl2) h->y_clean = true;
  if (overlap || dbl2) {
    // everything the side stream did is ordered before whatever the caller enqueues next
    PA_HIP(hipEventRecord(h->ev_tail, dbl2 ? h->side : t));
    PA_HIP(hipStreamWaitEvent(s, h->ev_tail, 0));
  }
  return PA_OK;
}

extern "C" int pa_dqn_set_overlap(pa_dqn* h, int32_t on) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  h->overlap = on ? 1 : 0;
  return PA_OK;
}

extern "C" int pa_dqn_check(pa_dqn* h) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  PA_REQUIRE(!h->err_host || h->err_host[0] == 0, PA_ERR_HIP,
             "pa_dqn_learn: a bounded wait for the target-network stream expired (code %d); the "
             "results of that call are invalid",
             h->err_host[0]);
  return PA_OK;
}

extern "C" int pa_dqn_enable_timing(pa_dqn* h, int32_t on) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  h->timing = on < 0 ? 0 : (on & 3);
  h->timing_chain = on < 0 ? 0 : ((on >> 2) & 1);
  h->chain_sample = false;
  for (auto& t : h->timers) { t.used = 0; t.units = 0; }
  if (h->timing >= 1) {
    // the level-1 timer's events exist before the timed call starts (hipEventCreate inside it cost
    // tens of microseconds of a 20-round learn())
    for (const char* name : {"target", "gather", "gather_nox"}) {
      Timer* t = find_timer(h, name);
      while (t->ev.size() < 64) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        t->ev.push_back(e);
      }
    }
  }
  return PA_OK;
}

extern "C" int pa_dqn_get_timing(pa_dqn* h, const char* name, double* avg_ms, int64_t* count) {
  PA_REQUIRE(h && name && avg_ms && count, PA_ERR_INVALID, "null argument");
  *avg_ms = 0.0;
  *count = 0;
  for (auto& t : h->timers) {
    if (t.name != name) continue;
    double total = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < t.used; i += 2) {
      PA_HIP(hipEventSynchronize(t.ev[i + 1]));
      float ms = 0.f;
      PA_HIP(hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
      total += ms;
      ++n;
    }
    *count = n;
    *avg_ms = n ? total / (double)n : 0.0;
    return PA_OK;
  }
  return PA_OK;
}

extern "C" int pa_dqn_get_timing_units(pa_dqn* h, const char* name, int64_t* units) {
  PA_REQUIRE(h && name && units, PA_ERR_INVALID, "null argument");
  *units = 0;
  for (auto& t : h->timers)
    if (t.name == name) *units = t.units;
  return PA_OK;
}

// ---- diagnostics ---------------------------------------------------------------
extern "C" int pa_debug_set_prof_target(pa_dqn* h, long long* stamps, int32_t max_tiles) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  h->prof_tgt = stamps;
  h->prof_tgt_tiles = max_tiles;
  return PA_OK;
}
extern "C" int pa_debug_set_prof(pa_dqn* h, long long* rowpass_stamps, long long* dw_stamps,
                                 int32_t round) {
  PA_REQUIRE(h, PA_ERR_INVALID, "null learner");
  h->prof_row = rowpass_stamps;
  h->prof_dw = dw_stamps;
  h->prof_round = round;
  return PA_OK;
}

extern "C" int pa_debug_linear(const float* A, int32_t lda, const float* B, int32_t ldb, float* C,
                               int32_t ldc, const float* bias, const float* hmask, int32_t ldh,
                               int32_t M, int32_t N, int32_t K, int32_t b_is_kn, int32_t epi,
                               void* stream) {
  PA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, PA_ERR_INVALID, "pa_debug_linear: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.Bm = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.Hmask = hmask; g.ldh = ldh; g.M = M; g.N = N; g.K = K; g.epi = epi;
  PA_REQUIRE(epi >= 0 && epi <= 3, PA_ERR_INVALID, "bad epilogue %d", epi);
  PA_REQUIRE(epi == EPI_NONE || (epi == EPI_MASK ? hmask != nullptr : bias != nullptr), PA_ERR_INVALID,
             "epilogue %d needs %s", epi, epi == EPI_MASK ? "hmask" : "bias");
  return b_is_kn ? launch_linear<true>(&g, 1, s) : launch_linear<false>(&g, 1, s);
}

// Copy one of the learner's workspaces of the LAST step out (tools/debug_pair_traj.py): "H1a", "H2a",
// "dZ2", "dZ1" (two interleaved partials when the paired row pass wrote it: 2 x the floats), "dq", "q"
extern "C" int pa_debug_workspace(pa_dqn* h, const char* name, float* out_dev, int64_t n, int32_t* paired_out) {
  PA_REQUIRE(h && name && out_dev && n > 0, PA_ERR_INVALID, "pa_debug_workspace: bad argument");
  const std::string k(name);
  const float* src = k == "H1a" ? h->H1a : k == "H2a" ? h->H2a : k == "dZ2" ? h->dZ2 :
                     k == "dZ1" ? (h->pair_live ? h->dZ1p : h->dZ1) : k == "dq" ? h->dq :
                     k == "q" ? h->qbuf : nullptr;
  PA_REQUIRE(src, PA_ERR_INVALID, "pa_debug_workspace: unknown workspace '%s'", name);
  if (paired_out) *paired_out = (k == "dZ1" && h->pair_live) ? 1 : 0;
  PA_HIP(hipMemcpy(out_dev, src, (size_t)n * 4, hipMemcpyDeviceToDevice));
  return PA_OK;
}

// PEARL_AMD_DEBUG_WORKERS=1: workgroups that took tiles in the persistent target launches so far / those launches
extern "C" int pa_debug_target_workers(pa_dqn* h, int64_t* workers_out, int64_t* launches_out) {
  PA_REQUIRE(h && workers_out && launches_out, PA_ERR_INVALID, "pa_debug_target_workers: null argument");
  int w = 0;
  if (h->dbg_workers) PA_HIP(hipMemcpy(&w, h->dbg_workers, 4, hipMemcpyDeviceToHost));
  *workers_out = w;
  *launches_out = h->dbg_launches;
  return PA_OK;
}

extern "C" int pa_debug_set_target_rows(int32_t rows) {
  PA_REQUIRE(rows == 0 || rows == 32 || rows == 64, PA_ERR_INVALID,
             "pa_debug_set_target_rows: 0 (environment / default), 32 or 64");
  set_target_rows_mode(rows);
  return PA_OK;
}

extern "C" int pa_debug_set_dw_split(int32_t mode) {
  PA_REQUIRE(mode >= -1 && mode <= 2, PA_ERR_INVALID, "pa_debug_set_dw_split: mode is -1, 0, 1 or 2");
  set_dw_split_mode(mode);
  return PA_OK;
}

extern "C" int pa_debug_weight_grad(const float* dZ, int32_t ldz, const float* X, int32_t ldx,
                                    float* dW, int32_t ldw, float* db, int32_t M, int32_t N,
                                    int32_t Bn, void* stream) {
  PA_REQUIRE(dZ && X && dW && db && M > 0 && N > 0 && Bn > 0, PA_ERR_INVALID,
             "pa_debug_weight_grad: bad argument");
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.p[0].dZ = dZ; a.p[0].ldz = ldz; a.p[0].X = X; a.p[0].ldx = ldx;
  a.p[0].dW = dW; a.p[0].ldw = ldw; a.p[0].db = db; a.p[0].M = M; a.p[0].N = N;
  a.p[0].tiles_n = (int)ceil_div(N, DW_TN); a.p[0].tile0 = 0;
  a.total_tiles = (int)ceil_div(M, DW_TM) * a.p[0].tiles_n;
  a.nprob = 1;
  a.B = Bn;
  return launch_weight_grad(a, false, reinterpret_cast<hipStream_t>(stream));
}
