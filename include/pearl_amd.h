/*
 * pearl_amd.h — C ABI of the MI355X-native replay/learner core for Pearl.
 *
 * This is the drop-in boundary.  The reference has no FFI: the
 * hot path sits behind two Python ABCs,
 *     pearl/replay_buffers/replay_buffer.py:18-91       (ReplayBuffer)
 *     pearl/policy_learners/policy_learner.py:40-229    (PolicyLearner)
 * so every entry point below cites the reference *method* it replaces.  The
 * Python host side (pearl_amd/) mirrors those ABCs and binds this library with
 * ctypes; INTEGRATION.md shows the stub a Pearl maintainer would add.
 *
 * Conventions
 *   - plain C types only; every device pointer is a raw HBM address owned by the
 *     caller (torch owns batches, parameters and optimizer state; the arena is
 *     owned by the library);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all
 *     work is enqueued on it, nothing synchronises unless documented;
 *   - return 0 on success, a negative pa_status otherwise; never throws.
 *     pa_last_error() returns a thread-local human-readable message;
 *   - handles are opaque, not re-entrant, one learner = one handle.
 */
#ifndef PEARL_AMD_H
#define PEARL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 1

typedef enum pa_status {
  PA_OK = 0,
  PA_ERR_INVALID = -1,     /* bad argument (shim raises AssertionError/TypeError)   */
  PA_ERR_VALUE = -2,       /* the reference's ValueError (batch_size > len(buffer)) */
  PA_ERR_HIP = -3,         /* a HIP runtime call failed                             */
  PA_ERR_UNSUPPORTED = -4, /* shape outside what the kernels were built for         */
  PA_ERR_NOMEM = -5
} pa_status;

typedef enum pa_dtype {
  PA_F32 = 0,
  PA_I64 = 1,
  PA_I32 = 2,
  PA_U8 = 3, /* torch.bool / torch.uint8 */
  PA_F64 = 4
} pa_dtype;

const char* pa_last_error(void);
int pa_abi_version(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails). */
int pa_device_count(void);

/* ------------------------------------------------------------------------ */
/* Replay arena: structure-of-arrays ring in HBM.                            */
/* Replaces the deque of per-transition CPU tensors of                       */
/* pearl/replay_buffers/tensor_based_replay_buffer.py:25-36 (+ basic_replay_ */
/* buffer.py:21-48).  Logical index i (0 = oldest) lives in slot             */
/* (head + i) % capacity, the same FIFO eviction as deque(maxlen=capacity).  */
/* ------------------------------------------------------------------------ */
typedef struct pa_arena pa_arena;

typedef struct pa_arena_desc {
  int64_t capacity;      /* transitions                                                    */
  int32_t device;        /* HIP device ordinal                                             */
  int32_t state_dim;     /* floats per state row (states are stored as float32, :353)       */
  int32_t action_elems;  /* elements of one stored action (1 for tensor([a]))               */
  int32_t action_dtype;  /* pa_dtype of the stored action (dtype is preserved, :354)        */
  int32_t reward_dtype;  /* PA_F32 for python floats, PA_I64 for python ints (:165-166)     */
  int32_t max_actions;   /* A: rows of the padded available-action table; 0 = none/continuous */
  int32_t avail_dim;     /* floats per available-action row (action_dim, :236-239)          */
  int32_t has_next_state;
  int32_t has_cost;
  int64_t staging_rows;  /* pinned host staging ring, rows (0 = default 4096)               */
} pa_arena_desc;

/* One transition as host pointers (what push() tensorises, :55-133).  NULL is
 * allowed for the optional members the descriptor disabled. */
typedef struct pa_transition {
  const float* state;       /* [state_dim]                                         */
  const void* action;       /* [action_elems] of action_dtype                      */
  const void* reward;       /* one element of reward_dtype                         */
  const float* next_state;  /* [state_dim]                                         */
  const float* curr_avail;  /* [max_actions * avail_dim], zero padded (:236-245)   */
  const uint8_t* curr_mask; /* [max_actions], 1 = unavailable (:247-251)           */
  const float* next_avail;
  const uint8_t* next_mask;
  const float* cost;        /* one float                                           */
  uint8_t terminated;
  uint8_t truncated;
} pa_transition;

/* n transitions as host column pointers (batched ingest). */
typedef struct pa_columns {
  const float* state;       /* [n, state_dim]   */
  const void* action;       /* [n, action_elems] */
  const void* reward;       /* [n]              */
  const uint8_t* terminated;/* [n]              */
  const uint8_t* truncated; /* [n]              */
  const float* next_state;  /* [n, state_dim]   */
  const float* curr_avail;  /* [n, A, avail_dim] or NULL with *_bcast=1 -> [A, avail_dim] */
  const uint8_t* curr_mask; /* [n, A]                                                      */
  const float* next_avail;
  const uint8_t* next_mask;
  const float* cost;        /* [n]              */
  int32_t avail_bcast;      /* 1: the four avail/mask pointers hold ONE row shared by all n */
} pa_columns;

/* Device pointers the gather writes (torch-allocated by the caller).  Any
 * member may be NULL = not wanted.  Shapes are the TransitionBatch contract of
 * pearl/replay_buffers/transition.py:89-132. */
typedef struct pa_batch_out {
  float* state;             /* [B, state_dim]                                            */
  void* action;             /* [B, action_elems] action_dtype                            */
  void* reward;             /* [B] reward_dtype                                          */
  uint8_t* terminated;      /* [B]                                                       */
  uint8_t* truncated;       /* [B]                                                       */
  float* next_state;        /* [B, state_dim]                                            */
  float* curr_avail;        /* [B, A, avail_dim]                                         */
  uint8_t* curr_mask;       /* [B, A]                                                    */
  float* next_avail;        /* [B, A, avail_dim]                                         */
  uint8_t* next_mask;       /* [B, A]                                                    */
  float* cost;              /* [B]                                                       */
  /* Fused learner-side views (policy_learner.py:197-218 preprocess_batch folded
   * into the gather): */
  float* x;                 /* [B, state_dim + rep_dim]: state || rep(action)            */
  float* next_avail_rep;    /* [B, A, rep_dim]: rep(next available actions)              */
  float* reward_f32;        /* [B] reward converted to float32                           */
  int32_t rep_dim;          /* representation width                                      */
  int32_t rep_onehot;       /* 1: rep = one-hot(index) (one_hot_action_representation_module.py:27-34); 0: identity */
  float* curr_avail_rep;    /* [B, A, rep_dim]: rep(current available actions) (discrete SAC's
                               actor update reads the critics on every available action)       */
} pa_batch_out;

int pa_arena_create(pa_arena** out, const pa_arena_desc* desc);
int pa_arena_destroy(pa_arena* a);
/* TensorBasedReplayBuffer.push (:55-133) after tensorisation.  Host-side only:
 * the row is packed into the pinned staging ring; it reaches HBM at the next
 * pa_arena_flush (called implicitly by gather/sample). */
int pa_arena_push(pa_arena* a, const pa_transition* t);
int pa_arena_push_many(pa_arena* a, int64_t n, const pa_columns* cols);
/* Device-resident ingest: columns already in HBM (same layout as pa_columns). */
int pa_arena_push_many_device(pa_arena* a, int64_t n, const pa_columns* cols, void* stream);
int pa_arena_flush(pa_arena* a, void* stream);
/* len(replay_buffer) (:284-285) */
int64_t pa_arena_len(const pa_arena* a);
int64_t pa_arena_capacity(const pa_arena* a);
/* slot of logical index 0 */
int64_t pa_arena_head(const pa_arena* a);
/* ReplayBuffer.clear (:287-288) */
int pa_arena_clear(pa_arena* a);
/* 1 when every row stored so far carries the SAME padded next-action table and mask — what
 * create_action_tensor_and_mask (tensor_based_replay_buffer.py:179-251) produces for a static
 * action space; tracked on push / push_many, reset by clear.  pa_dqn_learn then feeds the target
 * pass one broadcast table (stride 0) and the window gather skips the (B, A, rep) rows. */
int32_t pa_arena_shared_next_table(const pa_arena* a);
/* Parity mode of sample() (:253-282): caller supplies the B logical indices it
 * drew with random.sample(range(len), B) (host pointer, int64).  Returns
 * PA_ERR_VALUE if B > len.  idx_dev_scratch: device int64[B] scratch. */
int pa_arena_gather(pa_arena* a, const int64_t* logical_idx_host, int32_t B,
                    const pa_batch_out* out, int64_t* idx_dev_scratch, void* stream);
/* Same with indices already on the device. */
int pa_arena_gather_device(pa_arena* a, const int64_t* logical_idx_dev, int32_t B,
                           const pa_batch_out* out, void* stream);
/* Fast mode: uniform sampling WITHOUT replacement on the device (Philox4x32-10
 * keyed by (seed, offset), rejection of duplicates in LDS).  Statistically, not
 * bitwise, equivalent to random.sample.  idx_out_dev (int64[B], may be NULL)
 * receives the logical indices that were drawn. */
int pa_arena_sample(pa_arena* a, uint64_t seed, uint64_t offset, int32_t B,
                    const pa_batch_out* out, int64_t* idx_out_dev, void* stream);
/* Only the index draw of pa_arena_sample (for tests of the sampler). */
int pa_sample_indices(int64_t population, uint64_t seed, uint64_t offset, int32_t B,
                      int64_t* idx_out_dev, int32_t device, void* stream);
/* The index lists of `rounds` successive sample(B) calls in ONE launch (one workgroup per list):
 * list r = pa_sample_indices(population, seed, offset0 + r, B) at idx_out_dev + r * B.  What
 * PolicyLearner.learn (policy_learner.py:162-195) needs for its whole loop; pa_dqn_learn draws its
 * lists the same way. */
int pa_sample_indices_rounds(int64_t population, uint64_t seed, uint64_t offset0, int32_t B,
                             int32_t rounds, int64_t* idx_out_dev, int32_t device, void* stream);

/* out[b] = src[idx[b]] for rows of row_bytes bytes: per-transition columns kept beside the arena in
 * logical order (PPOTransition's gae / lam_return / action_probs, ppo.py:47-82). */
int pa_gather_rows(const void* src_dev, int32_t row_bytes, const int64_t* idx_dev, int32_t B,
                   void* out_dev, void* stream);
/* out[p][b] = src[p * plane_stride + idx[b]]: `planes` float columns of equal length gathered with
 * one index list in one launch (PPOTransition's gae / lam_return / action_probs, ppo.py:47-82). */
int pa_gather_planes(const float* src_dev, int64_t plane_stride, int32_t planes,
                     const int64_t* idx_dev, int32_t B, float* out_dev, void* stream);

/* OneHotActionTensorRepresentationModule.forward
 * (one_hot_action_representation_module.py:27-34): idx[n] -> out[n, num_classes]. */
int pa_one_hot(const void* idx_dev, int32_t idx_dtype, int64_t n, int32_t num_classes,
               float* out_dev, void* stream);

/* ------------------------------------------------------------------------ */
/* DQN learner: replaces DeepTDLearning.learn_batch / DeepQLearning           */
/* .get_next_state_values for a VanillaQValueNetwork with two hidden layers.  */
/* (deep_td_learning.py:269-360, deep_q_learning.py:130-167,                  */
/*  q_value_networks.py:152-174, common/utils.py:75-152, :214-226)            */
/* ------------------------------------------------------------------------ */
typedef struct pa_dqn pa_dqn;

typedef struct pa_dqn_desc {
  int32_t device;
  int32_t state_dim;   /* S                                                     */
  int32_t action_dim;  /* AD: width of the action representation                */
  int32_t hidden1;     /* H1                                                    */
  int32_t hidden2;     /* H2                                                    */
  int32_t max_batch;   /* largest B a step will see (workspace sizing)          */
  int32_t max_actions; /* largest A a step will see                             */
  float discount;      /* gamma (deep_td_learning.py:313-317)                   */
  float tau;           /* soft_update_tau (common/utils.py:214-226)             */
  double lr, beta1, beta2, eps, weight_decay; /* optim.AdamW defaults (:183-185) */
  int32_t amsgrad;
  int32_t double_q;    /* how the next state is valued.  0: DeepQLearning (deep_q_learning.py:
                        * 130-167), max over the available next actions of Q_target.  1: DoubleDQN
                        * (double_dqn.py:29-57), the ONLINE net's argmax valued by the target net.
                        * 2: DeepSARSA (deep_sarsa.py:59-78), Q_target(s', next_action) for the
                        * batch's committed next action (pa_dqn_batch.next_action_rep); stand-alone
                        * steps only (pa_dqn_step / pa_dqn_qvalues), not pa_dqn_learn             */
} pa_dqn_desc;

/* Parameter storage, flat fp32.  Layout (floats), every tensor offset rounded
 * up to 4: W1[H1, S+AD] | b1[H1] | W2[H2, H1] | b2[H2] | W3[1, H2] | b3[1].
 * All five buffers are caller-owned (torch) so that state_dict()/compare()
 * stay truthful (SURVEY.md §5 checkpoint row). */
typedef struct pa_dqn_buffers {
  float* q;        /* online parameters                                    */
  float* q_target; /* target parameters                                    */
  float* grad;     /* gradient of the mean-squared Bellman error           */
  float* exp_avg;
  float* exp_avg_sq;
  float* max_exp_avg_sq;
} pa_dqn_buffers;

/* One preprocessed batch (after policy_learner.py:197-218). */
typedef struct pa_dqn_batch {
  int32_t B;
  int32_t A;                    /* available-action rows per transition                    */
  const float* x;               /* [B, S+AD] state || rep(action); or NULL and use:        */
  const float* state;           /* [B, S]                                                   */
  const float* action_rep;      /* [B, AD]                                                  */
  const float* reward;          /* [B] float32                                              */
  const uint8_t* terminated;    /* [B]                                                      */
  const float* next_state;      /* [B, S]                                                   */
  const float* next_avail_rep;  /* [B, A, AD], or [A, AD] when next_avail_bcast             */
  const uint8_t* next_mask;     /* [B, A] (1 = unavailable), or [A] when bcast, or NULL     */
  int32_t next_avail_bcast;
  const float* next_action_rep; /* [B, AD] rep of the committed next action: DeepSARSA only
                                 * (pa_dqn_desc.double_q == 2), NULL otherwise               */
} pa_dqn_batch;

int64_t pa_dqn_param_count(int32_t S, int32_t AD, int32_t H1, int32_t H2);
/* offsets[6] of W1,b1,W2,b2,W3,b3 inside the flat buffer */
int pa_dqn_param_offsets(int32_t S, int32_t AD, int32_t H1, int32_t H2, int64_t* offsets6);

int pa_dqn_create(pa_dqn** out, const pa_dqn_desc* desc);
int pa_dqn_destroy(pa_dqn* h);
int pa_dqn_bind(pa_dqn* h, const pa_dqn_buffers* bufs);
/* The bound flat buffers were written from outside the library (e.g. torch in-place ops): derived
 * copies (fragment-major weights) are rebuilt by the next call that needs them. */
int pa_dqn_invalidate(pa_dqn* h);

/* Parity probe: Q(s,a) of the online net (q_out[B]), max_a' Q_target(s',a')
 * (next_v_out[B]) and the Bellman target (target_out[B]).  Any output may be NULL. */
int pa_dqn_qvalues(pa_dqn* h, const pa_dqn_batch* batch, float* q_out, float* next_v_out,
                   float* target_out, void* stream);

/* update_target_network (common/utils.py:214-226): theta' <- tau*theta + (1-tau)*theta'. */
int pa_dqn_update_target(pa_dqn* h, void* stream);

/* DeepTDLearning.learn_batch (:333-360) for one batch: optional target soft
 * update first (forward(), :283-284), forward, loss, backward, AdamW step.
 * adam_step is the 1-based optimizer step this call performs.  grad_world > 1:
 * stop after the backward (gradients in bufs.grad, scaled by 1/grad_world) so
 * the caller can all-reduce them; then call pa_dqn_apply.
 * mean_abs_td_out: device float; receives mean |Q - target| (:358-359). */
int pa_dqn_step(pa_dqn* h, const pa_dqn_batch* batch, int32_t do_target_update,
                int64_t adam_step, int32_t grad_world, float* mean_abs_td_out, void* stream);
/* AdamW(amsgrad) on bufs.grad (torch/optim/adam.py _single_tensor_adam). */
int pa_dqn_apply(pa_dqn* h, int64_t adam_step, void* stream);

/* PolicyLearner.learn (policy_learner.py:162-195) fused: `rounds` iterations of
 * device-side sample -> gather(+one-hot) -> learn_batch on the arena, all on
 * `stream`.  The index lists of every round are drawn by one launch; because the
 * target network only changes every target_update_freq rounds
 * (deep_td_learning.py:283-284), the gather and the target-network pass
 * (deep_q_learning.py:130-167) of a whole window of rounds run as ONE launch
 * each (bit-identical to doing them round by round); only the online chain
 * (forward, loss, backward, AdamW) is issued per round.  training_steps0 /
 * adam_step0: counters before the call.  losses_out: device float[rounds]. */
typedef struct pa_learn_args {
  int32_t rounds;
  int32_t batch_size;
  int32_t rep_onehot;          /* action representation: 1 one-hot, 0 identity          */
  int32_t target_update_freq;
  int64_t training_steps0;
  int64_t adam_step0;
  uint64_t seed;               /* Philox key                                            */
  uint64_t offset0;            /* Philox counter base (advance by `rounds` per call)    */
  float* losses_out;
  const int64_t* idx_host;     /* optional parity mode: [rounds, batch_size] logical indices */
  /* Data parallelism (no reference counterpart).  grad_world > 1: gradients are
   * pre-scaled by 1/grad_world and, every round, the library calls
   *   allreduce_start(ctx, grad, n, stream)  — enqueue a SUM all-reduce of grad[n] ordered after
   *                                            the work already on `stream` (RCCL over xGMI)
   *   allreduce_wait(ctx, stream)            — make `stream` wait for that all-reduce
   * with the NEXT round's target-network pass enqueued in between, so the exchange overlaps
   * compute; AdamW runs after the wait.  The library itself links no communication library: the
   * host passes the hooks (torch.distributed in pearl_amd, ncclAllReduce in a native host). */
  int32_t grad_world;
  int (*allreduce_start)(void* ctx, float* grad, int64_t n, void* stream);
  int (*allreduce_wait)(void* ctx, void* stream);
  void* allreduce_ctx;
} pa_learn_args;
int pa_dqn_learn(pa_dqn* h, pa_arena* arena, const pa_learn_args* args, void* stream);
/* pa_dqn_learn overlaps the target-network pass of a window (on an internal low-priority stream)
 * with the per-round online chains on `stream`; the Bellman targets cross between the two as
 * data-tagged words that the chain polls with a BOUNDED wait.  After synchronising `stream`, call
 * pa_dqn_check: PA_ERR_HIP if such a wait expired during the last pa_dqn_learn (results invalid),
 * PA_OK otherwise.  PEARL_AMD_OVERLAP=0 in the environment selects the single-stream loop. */
int pa_dqn_check(pa_dqn* h);
/* Select the overlapped (1, default) or the single-stream (0) learn loop for later pa_dqn_learn
 * calls; both produce bit-identical results.  bench.py uses the single-stream loop to time the
 * target kernel with the chip to itself. */
int pa_dqn_set_overlap(pa_dqn* h, int32_t on);

/* Native all-reduce hooks for pa_learn_args, backed by RCCL (ncclAllReduce over xGMI) resolved at
 * run time with dlopen (the library has no link-time dependency on it).  One communicator per
 * process / GPU: rank 0 calls pa_comm_unique_id, the host broadcasts the 128 bytes to the other
 * ranks (torch.distributed in pearl_amd), every rank calls pa_comm_create.  Pass
 * pa_comm_allreduce_start / pa_comm_allreduce_wait and the pa_comm* as allreduce_ctx. */
typedef struct pa_comm pa_comm;
int pa_comm_available(void);
int pa_comm_unique_id(void* id128_out);
int pa_comm_create(pa_comm** out, int32_t device, int32_t world, int32_t rank, const void* id128);
int pa_comm_destroy(pa_comm* c);
/* the rank count / rank RCCL itself reports for the communicator (ncclCommCount, ncclCommUserRank) */
int pa_comm_info(pa_comm* c, int32_t* ranks_out, int32_t* rank_out);
/* The same hooks on a ONE-SHOT PEER-TO-PEER exchange instead of RCCL (the
 * per-round message — 413 KB for the DQN of BASELINE config 2 — is latency-bound, and xGMI is point
 * to point): every rank owns one device buffer its peers map through hipIpc; a round is "copy my
 * gradient into my slot, publish a round counter, wait for the peers' counters, add the G slots in
 * rank order" — two launches on the learner stream, no ring, bitwise-identical sums on every rank.
 * Bring-up: every rank calls pa_comm_create_p2p (world <= 8, messages of up to max_floats floats),
 * exchanges pa_comm_p2p_handle()'s 64 bytes with its peers (torch.distributed in pearl_amd/_comm.py)
 * and maps them with pa_comm_p2p_open; pa_comm_allreduce_start / _wait / pa_comm_info / _destroy
 * then work as above.  The wait for a peer's round counter is bounded (PEARL_AMD_P2P_TIMEOUT_S,
 * default 30 s) so that a dead peer cannot hang the GPU; RCCL would block instead.  A wait that
 * expires never yields a sum: the round's gradient is overwritten with NaN, a STICKY error is
 * raised, and every later exchange on the communicator is refused (the ranks are no longer in
 * lock-step).  pa_comm_p2p_check / pa_comm_check (any communicator; PA_OK for RCCL ones) report it —
 * one read of a pinned host word, meant to follow the host sync that ends a learn() call.
 * pa_comm_max_floats: the largest message one exchange takes (0 = unlimited); callers chunk. */
int pa_comm_create_p2p(pa_comm** out, int32_t device, int32_t world, int32_t rank, int64_t max_floats);
int pa_comm_p2p_handle(pa_comm* c, void* handle64_out);
int pa_comm_p2p_open(pa_comm* c, int32_t peer, const void* handle64);
int pa_comm_p2p_check(pa_comm* c);
int pa_comm_check(pa_comm* c);
int64_t pa_comm_max_floats(pa_comm* c);
int pa_comm_allreduce_start(void* comm, float* buf, int64_t n, void* stream);
int pa_comm_allreduce_wait(void* comm, void* stream);

/* Kernel timing of the last pa_dqn_learn / pa_dqn_step when timing is enabled
 * (HIP events on the launch stream; used by bench.py's roofline block). */
/* level 0: off; 1: only the dominant kernel ("target"); 2: every stage.  Resets the counters.
 * level | 4 (with level 1): also "rowpass" and "bwd_dw" — the online chain's two launches — of one
 * mid-window round of every sampled window of the overlapped loop (bench.py's roofline.chain). */
int pa_dqn_enable_timing(pa_dqn* h, int32_t level);
/* names: "allreduce" (data-parallel loop: allreduce_start .. allreduce_wait on the learner stream, one
 * mid-window round per sampled window at level 1),
 * "target" (every 4th launch at level 1), "target_l1", "l1_dual", "online_l1", "gather",
 * "sample", "online_l2", "head", "bwd_dx", "bwd_dw", "adamw", "soft_update", "learn"
 * -> average milliseconds per timed launch and the number of timed launches */
int pa_dqn_get_timing(pa_dqn* h, const char* name, double* avg_ms, int64_t* count);
/* transitions covered by the timed launches of `name` (a window launch covers several rounds) */
int pa_dqn_get_timing_units(pa_dqn* h, const char* name, int64_t* units);

/* ------------------------------------------------------------------------ */
/* Generic fully-connected network: mlp_block(Linear+ReLU ..., Linear)       */
/* (pearl/neural_networks/common/utils.py:75-152) over flat caller-owned     */
/* fp32 buffers, W_l[d_{l+1}, d_l] | b_l[d_{l+1}] per layer, every tensor     */
/* offset rounded up to 4 floats.  Used by VanillaValueNetwork               */
/* (common/value_networks.py:35-59), VanillaActorNetwork / GaussianActor-    */
/* Network (actor_networks.py:107-176, :488-629; fc_mu and fc_std are ONE    */
/* last layer of 2A rows) and the VanillaQValueNetwork critics of TwinCritic */
/* (twin_critic.py:22-91).                                                   */
/* ------------------------------------------------------------------------ */
typedef struct pa_mlp pa_mlp;
#define PA_MLP_MAX_LAYERS 8
typedef struct pa_mlp_desc {
  int32_t device;
  int32_t n_layers;                      /* Linear layers                                 */
  int32_t dims[PA_MLP_MAX_LAYERS + 1];   /* d_0 (input) ... d_L (output)                   */
  int32_t max_batch;
  double lr, beta1, beta2, eps, weight_decay; /* optim.AdamW (actor_critic_base.py:159-167) */
  int32_t amsgrad;
  int32_t no_last_bias; /* 1: the last layer is nn.Linear(bias=False) (linear_layer_e2e,
                           neural_linear_regression.py:84-86); its bias slot stays zero */
  int32_t identity_layers; /* bit l set: hidden layer l has NO ReLU (the output layer of
                              NeuralLinearRegression._nn_layers, :65-75); the last layer never has */
  /* mlp_block's other forms (common/utils.py:75-152; round 5).  Networks that use either run layer
   * by layer (GEMM launches + row-local normalisation / activation kernels, mlp_norm_act.hpp), not
   * through the fused row-pass kernels; the fused multi-network steps refuse them
   * (PA_ERR_UNSUPPORTED). */
  int32_t hidden_act;      /* ActivationType of the hidden layers (utils.py:29-56): 0 relu, 1 leaky_relu
                              (slope 0.01), 2 tanh, 3 softplus (beta 1, threshold 20), 4 sigmoid */
  int32_t layer_norm;      /* bit l set: nn.LayerNorm(d_{l+1}) (eps 1e-5, affine) between hidden Linear l
                              and its activation (utils.py:110-113; mlp_block sets it for every hidden
                              layer, the bandit's trunk for all but its activation-free output layer).
                              The weights / biases are parameters: they follow the W / b block of the
                              flat buffers (pa_mlp_norm_offsets) */
  /* round 6: the remaining options of mlp_block (utils.py:113-126, :142-144), layer by layer as well */
  int32_t batch_norm;      /* bit l set: nn.BatchNorm1d(d_{l+1}) (eps 1e-5, momentum 0.1, affine, TRAINING
                              mode: statistics of the batch at hand — the reference never switches its
                              networks to eval) AFTER hidden layer l's activation (utils.py:119-121).  gamma /
                              beta are parameters behind the LayerNorm block (pa_mlp_bn_offsets); the
                              running statistics are the caller's buffers (pa_mlp_bind_batch_norm) */
  int32_t dropout;         /* bit l set: nn.Dropout between hidden layer l's (LayerNorm and) activation
                              (utils.py:114-116).  The keep mask of every forward is the caller's
                              (pa_mlp_set_dropout: [B][d_{l+1}] floats, 0 or 1 / (1 - p)) */
  int32_t residual;        /* bit l set (l = 0 .. n_layers - 1): layer l's block is wrapped in
                              ResidualWrapper — out = in + block(in), d_l == d_{l+1}
                              (utils.py:122-131, :142-150; residual_wrapper.py:13-29) */
} pa_mlp_desc;
typedef struct pa_mlp_buffers {
  float* p;
  float* p_target;       /* NULL if the network has no target copy */
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* max_exp_avg_sq;
} pa_mlp_buffers;
int64_t pa_mlp_param_count(const pa_mlp_desc* d);
/* offsets[2 * n_layers]: W_0, b_0, W_1, b_1, ... */
int pa_mlp_param_offsets(const pa_mlp_desc* d, int64_t* offsets);
/* layer_norm != 0: offsets[2 * (n_layers - 1)]: gamma_l, beta_l of hidden layer l's LayerNorm (each
 * d_{l+1} floats) at [2 l], [2 l + 1]; -1 for hidden layers without one */
int pa_mlp_norm_offsets(const pa_mlp_desc* d, int64_t* offsets);
/* batch_norm != 0: offsets[2 * (n_layers - 1)]: weight_l, bias_l of hidden layer l's BatchNorm1d (each
 * d_{l+1} floats); -1 for hidden layers without one */
int pa_mlp_bn_offsets(const pa_mlp_desc* d, int64_t* offsets);
/* The running statistics of hidden layer `layer`'s BatchNorm1d of the online (use_target = 0) or
 * target network: running_mean / running_var [d_{l+1}] floats and num_batches_tracked (one int64),
 * updated by every forward as torch's training-mode batch_norm does (momentum 0.1, unbiased
 * variance); any may be NULL (then not updated). */
int pa_mlp_bind_batch_norm(pa_mlp* h, int32_t use_target, int32_t layer, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked);
/* The dropout keep mask of hidden layer `layer` for the NEXT forward(s): [B][d_{l+1}] floats, 0 or
 * 1 / (1 - p); the kept forward's mask must stay alive until its backward.  NULL: no dropout
 * (evaluation). */
int pa_mlp_set_dropout(pa_mlp* h, int32_t layer, const float* mask, int32_t ldm);
int pa_mlp_create(pa_mlp** out, const pa_mlp_desc* desc);
int pa_mlp_destroy(pa_mlp* h);
int pa_mlp_bind(pa_mlp* h, const pa_mlp_buffers* bufs);
/* The caller (or anyone else) may have written the bound parameter buffers since the last call:
 * cached derived copies (the MFMA fragment-major weights of the row-pass kernels) are rebuilt on
 * next use.  pa_mlp_bind / pa_mlp_adam / pa_mlp_soft_update invalidate on their own. */
int pa_mlp_invalidate(pa_mlp* h);
/* nn.Sequential forward; keep = 1 retains the hidden activations for pa_mlp_backward. */
int pa_mlp_forward(pa_mlp* h, int32_t use_target, const float* x, int32_t ldx, int32_t B,
                   float* out, int32_t ldo, int32_t keep, void* stream);
/* kept output (after ReLU) of hidden layer `layer` of the last keep = 1 forward */
int pa_mlp_copy_activation(pa_mlp* h, int32_t layer, int32_t B, float* out, int32_t ldo, void* stream);
/* the kept activation in place: device pointer and row pitch (floats) of hidden layer `layer`'s
 * output [kept batch][d]; valid until the next forward of this network (AdamW does not touch it) */
int pa_mlp_activation(pa_mlp* h, int32_t layer, float** ptr_out, int32_t* ld_out);
/* autograd of the kept forward: d_x[B, d_0] (nullable) and the weight gradients dW/db into
 * bufs.grad — want_dw 0: none; 1: now; 2: DEFERRED to the next pa_mlp_adam on this network, which
 * then runs dW, AdamW and the refresh of the row-pass kernels' packed weights as ONE launch per
 * three layers (x and d_out must stay valid until then), or to pa_mlp_flush_grads (weight gradients
 * only: a data-parallel step all-reduces bufs.grad before AdamW). */
int pa_mlp_backward(pa_mlp* h, const float* x, int32_t ldx, int32_t B, const float* d_out,
                    int32_t ldd, int32_t want_dw, float* d_x, int32_t lddx, void* stream);
/* Q(s_b, a_i) of a [S + AD, H1, H2, 1] ReLU critic for every action of an action set,
 * q_out[b * A + i] (TwinCritic.get_q_values on (B, A, AD) actions, twin_critic.py:75-91,
 * q_value_networks.py:152-174), through DQN's fused all-actions kernel: no (B A, S + AD) expansion
 * and no hidden activations in HBM.  rep: [rows, A, AD] (rep_bstride = A * AD) or one shared
 * [A, AD] table (rep_bstride = 0).  Needs H1, H2 <= 256, A <= 64, rows <= max_batch;
 * PA_ERR_UNSUPPORTED otherwise (callers then expand the input and use pa_mlp_forward). */
int pa_mlp_q_all(pa_mlp* h, int32_t use_target, const float* state, int32_t ld_state,
                 const float* rep, int64_t rep_bstride, int32_t rows, int32_t A, int32_t AD,
                 float* q_out, void* stream);
/* pa_mlp_q_all for the two critics of a twin (one shape, one input) with the shareable launches
 * shared: one repack launch, one first-layer GEMM launch with two problems.  Same values as two
 * pa_mlp_q_all calls (twin_critic.py:75-91). */
int pa_mlp_q_all2(pa_mlp* h1, pa_mlp* h2, int32_t use_target, const float* state, int32_t ld_state,
                  const float* rep, int64_t rep_bstride, int32_t rows, int32_t A, int32_t AD,
                  float* q1_out, float* q2_out, void* stream);

/* Two networks of the same depth on the same input in lock-step (TwinCritic, twin_critic.py:22-91;
 * PPO's actor and critic): the same arithmetic as two pa_mlp_forward / pa_mlp_backward calls, with
 * every layer of both networks in one launch.  Layer widths may differ.  d_x1 / d_x2 are both
 * given or both NULL. */
int pa_mlp_forward2(pa_mlp* h1, pa_mlp* h2, int32_t use_target, const float* x, int32_t ldx,
                    int32_t B, float* out1, int32_t ldo1, float* out2, int32_t ldo2, int32_t keep,
                    void* stream);
int pa_mlp_backward2(pa_mlp* h1, pa_mlp* h2, const float* x, int32_t ldx, int32_t B,
                     const float* d_out1, int32_t ldd1, const float* d_out2, int32_t ldd2,
                     int32_t want_dw, float* d_x1, float* d_x2, int32_t lddx, void* stream);
int pa_mlp_flush_grads(pa_mlp* h, void* stream);
int pa_mlp_adam(pa_mlp* h, int64_t step, void* stream);
/* Twin networks with one optimizer configuration (twin critics), both with deferred weight
 * gradients: dW + AdamW of both in one launch and, with soft_tau >= 0, their soft target updates
 * in the same epilogue.  PA_ERR_UNSUPPORTED when the pair does not qualify. */
int pa_mlp_adam2(pa_mlp* a, pa_mlp* b, int64_t step, float soft_tau, void* stream);
/* Data-parallel step of two networks that share a batch (PPO's actor and critic, ppo.py:152-199,
 * under torch.distributed): pa_mlp_flush_grads2 forms BOTH networks' deferred weight gradients in
 * one launch without stepping the optimizer; the caller all-reduces the gradient buffers (ONE
 * message when they are adjacent in memory: FlatMlp.join_grads); pa_mlp_adamw2 then applies
 * AdamW(amsgrad) step_a / step_b to both in one launch.  No reference counterpart. */
int pa_mlp_flush_grads2(pa_mlp* a, pa_mlp* b, void* stream);
int pa_mlp_adamw2(pa_mlp* a, pa_mlp* b, int64_t step_a, int64_t step_b, void* stream);
/* update_target_network (common/utils.py:214-226) */
int pa_mlp_soft_update(pa_mlp* h, float tau, void* stream);

/* Call after synchronising the stream a fused actor-critic step / learn loop ran on
 * (pa_sac_step, pa_sac_learn, pa_ddpg_step, pa_ddpg_learn): PA_ERR_HIP if a bounded in-launch
 * workgroup hand-off of that learner expired.  The step then skipped its AdamW / soft-update
 * epilogues (parameters intact) and its reported losses are invalid.  `actor`: the learner's
 * actor network handle.  No reference counterpart (a robustness hook of the fused kernels that
 * replace soft_actor_critic_continuous.py:131-231 / ddpg.py:105-156). */
int pa_ac_check(pa_mlp* actor);

/* ------------------------------------------------------------------------ */
/* One ContinuousSoftActorCritic.learn_batch as one call                      */
/* (soft_actor_critic_continuous.py:131-231, actor_critic_base.py:309-366):   */
/* actor update, critic update, critic-target soft update, entropy step, in   */
/* the reference's order — the same launches, sequenced in C (sac_step.hip).   */
/* The actor is the [S, ..., 2A] network whose last layer stacks fc_mu | fc_std;*/
/* the critics are [S + A, ..., 1].  Single process: a data-parallel step      */
/* all-reduces between backward and AdamW and stays with the per-launch API.   */
/* ------------------------------------------------------------------------ */
typedef struct pa_sac_step_args {
  pa_mlp* actor; pa_mlp* critic1; pa_mlp* critic2;
  const float* state; int32_t ld_state;        /* [B, S] */
  const float* action; int32_t ld_action;      /* [B, A] the batch's actions */
  const float* reward;                         /* [B] */
  const uint8_t* terminated;                   /* [B] */
  const float* next_state; int32_t ld_next_state;
  const float* noise_actor;                    /* [B, A] standard-normal draws of the actor update */
  const float* noise_critic;                   /* [B, A] ... of the Bellman target's next action */
  const float* low; const float* high;         /* [A] action box */
  float* alpha;                                /* device scalar: entropy coefficient */
  float* log_alpha;                            /* NULL: fixed coefficient (no autotune) */
  float* alpha_m; float* alpha_v; float* alpha_vmax;
  float target_entropy;
  double alpha_lr, alpha_beta1, alpha_beta2, alpha_eps, alpha_weight_decay;
  int32_t alpha_amsgrad;
  int64_t alpha_step;                          /* AdamW step number of this entropy update */
  int32_t B, S, A;
  float gamma, tau;
  int64_t actor_step, critic_step;             /* AdamW step numbers (1-based) of this update */
  float* scratch;                              /* pa_sac_scratch_floats(B, S, A) floats, 16-byte aligned */
  float* losses;                               /* [3] actor loss, critic loss, entropy loss */
  float* log_prob_out;                         /* optional [B]: log-prob of the actor update's sample */
} pa_sac_step_args;
int64_t pa_sac_scratch_floats(int32_t B, int32_t S, int32_t A);
int pa_sac_step(const pa_sac_step_args* args, void* stream);
/* ------------------------------------------------------------------------ */
/* One DeepDeterministicPolicyGradient / TD3 learn_batch as one call          */
/* (ddpg.py:106-147, td3.py:105-175 on actor_critic_base.py:309-366): the same */
/* launches the per-stage path issues, sequenced in C.  The actor is            */
/* [S, ..., A] (tanh + action scaling applied by pa_tanh_action), the critics    */
/* [S + A, ..., 1], every network with its target copy bound.  Single process.   */
/* ------------------------------------------------------------------------ */
typedef struct pa_ddpg_step_args {
  pa_mlp* actor; pa_mlp* critic1; pa_mlp* critic2;
  const float* state; int32_t ld_state;
  const float* action; int32_t ld_action;
  const float* reward;
  const uint8_t* terminated;
  const float* next_state; int32_t ld_next_state;
  const float* target_noise;     /* [B, A] N(0, sigma^2) draws of TD3's target smoothing, or NULL */
  float noise_clip;
  const float* low; const float* high;
  const float* zeros;            /* at least B device floats of 0 */
  int32_t B, S, A;
  float gamma;
  int32_t do_actor;              /* update the actor this step (TD3: every actor_update_freq-th) */
  int32_t do_targets;            /* soft-update the critic targets and the actor target */
  float critic_tau, actor_tau;
  int64_t actor_step, critic_step;   /* AdamW step numbers (1-based) of this update */
  float* scratch;                /* pa_ddpg_scratch_floats(B, S, A) floats, 16-byte aligned */
  float* losses;                 /* [2] actor loss (written when do_actor), critic loss */
} pa_ddpg_step_args;
int64_t pa_ddpg_scratch_floats(int32_t B, int32_t S, int32_t A);
int pa_ddpg_step(const pa_ddpg_step_args* args, void* stream);

/* ------------------------------------------------------------------------ */
/* PolicyLearner.learn (policy_learner.py:190-231) of the learners above as    */
/* one call: training_rounds x (gather the round's presampled index list from  */
/* the arena into one batch of workspace; the step).  `step0` describes round  */
/* 0 and reads the workspace (state / action / reward / terminated /           */
/* next_state pointers equal `batch`'s); per round the call advances the       */
/* AdamW step numbers, the noise and the losses pointers.  Actions and rewards */
/* in the arena must be float32.                                               */
/* ------------------------------------------------------------------------ */
typedef struct pa_ac_loop_args {
  int32_t rounds;
  const int64_t* idx_lists;      /* device [rounds][B] logical indices (pa_sample_indices_rounds) */
  pa_batch_out batch;            /* device workspace of one batch (G batches: gather_rounds) */
  const float* noise;            /* device [rounds][noise_stride]: SAC [2][B][A] standard normal
                                    (actor update, Bellman target); TD3 [B][A] N(0, sigma^2); NULL: none */
  int64_t noise_stride;
  float* losses;                 /* device [rounds][losses_stride]; per round the step's `losses` */
  int32_t losses_stride;
  int32_t actor_update_freq;     /* TD3: actor step + target updates on rounds where               */
  int64_t training_step0;        /* (training_step0 + r + 1) % actor_update_freq == 0; <= 1: all   */
  int32_t gather_rounds;         /* G > 1: `batch` holds G x B rows and ONE gather launch fills the  */
                                 /* batches of G consecutive rounds (G x B <= rows in the arena);    */
                                 /* round r steps on rows [(r % G) B, (r % G + 1) B)                 */
} pa_ac_loop_args;
int pa_sac_learn(const pa_sac_step_args* step0, pa_arena* arena, const pa_ac_loop_args* loop,
                 void* stream);
int pa_ddpg_learn(const pa_ddpg_step_args* step0, pa_arena* arena, const pa_ac_loop_args* loop,
                  void* stream);

/* Discrete SoftActorCritic.learn_batch as ONE call (soft_actor_critic.py:153-287 on
 * actor_critic_base.py:309-366): twin all-actions pass (online) -> actor row step + AdamW -> twin
 * all-actions pass (target, next states) -> Bellman targets under the updated policy -> twin critics'
 * row step on xq = state || rep(action) -> weight gradients + AdamW (+ soft target update, tau >= 0)
 * -> entropy coefficient (log_alpha non-NULL).  The same launches as the per-stage entry points
 * above issue, in the same order: bit-identical.  PA_ERR_UNSUPPORTED when a network is outside the
 * fused row steps' shapes.  scratch: pa_dsac_scratch_floats(B, A) floats.
 * curr_rep / next_rep: rep(available actions) per row [B][A][AD] (bstride A * AD) or ONE [A][AD]
 * table (bstride 0); masks [B][A] (1 = unavailable) or NULL; h_out [B] (nullable): sum_a P log(P + 1e-8). */
typedef struct pa_dsac_step_args {
  pa_mlp* actor; pa_mlp* critic1; pa_mlp* critic2;
  int32_t B, S, A, AD;
  const float* state; int32_t ld_state;
  const float* next_state; int32_t ld_next_state;
  const float* xq; int32_t ld_xq;
  const float* reward; const uint8_t* terminated;
  const float* curr_rep; int64_t curr_rep_bstride;
  const float* next_rep; int64_t next_rep_bstride;
  const uint8_t* curr_mask; const uint8_t* next_mask;
  float gamma, tau;                     /* tau < 0: no soft update of the targets in this step */
  float* alpha;                         /* device scalar: the entropy coefficient */
  float* log_alpha; float* alpha_m; float* alpha_v;   /* autotune (log_alpha NULL: fixed) */
  float target_entropy;
  double alpha_lr, alpha_beta1, alpha_beta2, alpha_eps, alpha_weight_decay;
  int64_t alpha_step, actor_step, critic_step;        /* 1-based optimizer steps of THIS call */
  float* scratch; float* losses;        /* losses [3]: actor, critic, entropy-coefficient */
  float* h_out;
} pa_dsac_step_args;
int64_t pa_dsac_scratch_floats(int32_t B, int32_t A);
int pa_dsac_step(const pa_dsac_step_args* args, void* stream);
/* SoftActorCritic.learn's rounds as one call: per group of loop->gather_rounds rounds one gather
 * launch writes state, next_state, x, reward_f32, terminated, both masks and both rep(available
 * actions) views of `loop->batch` (the workspace `step0` points into, dense rows), then pa_dsac_step
 * per round; losses [rounds][losses_stride >= 3].  loop->noise is unused. */
int pa_dsac_learn(const pa_dsac_step_args* step0, pa_arena* arena, const pa_ac_loop_args* loop,
                  void* stream);

/* ImplicitQLearning.learn_batch as ONE call (implicit_q_learning.py:159-269): target critics at
 * (s, a) -> V(s') -> V(s) -> expectile value head + advantage weights -> y = r + gamma V(s') -> twin
 * critics' row step -> actor forward + policy-extraction head -> backward of value and actor ->
 * AdamW of value, actor, critics (+ soft target update, tau >= 0).  The launches of the per-stage
 * entry points above in the same order: bit-identical.  actor_kind 0: VanillaContinuousActorNetwork
 * (tanh-squashed, pa_tanh_action / pa_awr_head(0) / pa_tanh_action_grad), 1: GaussianActorNetwork
 * (pa_gauss_awr_head; the actor outputs 2A), 2: softmax actor (pa_awr_head(1)).  action [B][A]: the
 * representation of the taken action; xq [B][S + A] = state || action or NULL (formed in scratch).
 * pick_value / pick_actor (0 / 1): which target critic the value loss regresses to and which weighs
 * the policy extraction — the reference's two host draws (:189-190, :205-206).  zeros: B + 1 device
 * zeros.  losses [3]: value, critic, actor.  PA_ERR_UNSUPPORTED when the critics are outside the
 * fused row step's shapes. */
typedef struct pa_iql_step_args {
  pa_mlp* actor; pa_mlp* value; pa_mlp* critic1; pa_mlp* critic2;
  int32_t B, S, A, actor_kind;
  const float* state; int32_t ld_state;
  const float* next_state; int32_t ld_next_state;
  const float* action; int32_t ld_action;
  const float* xq; int32_t ld_xq;
  const float* reward; const uint8_t* terminated;
  const float* low; const float* high;
  int32_t pick_value, pick_actor;
  float expectile, temperature, adv_clamp, gamma, tau;
  int64_t actor_step, value_step, critic_step;
  const float* zeros;
  float* scratch;                       /* pa_iql_scratch_floats(B, S, A, actor output width) */
  float* losses;
} pa_iql_step_args;
int64_t pa_iql_scratch_floats(int32_t B, int32_t S, int32_t A, int32_t head_width);
int pa_iql_step(const pa_iql_step_args* args, void* stream);
/* ImplicitQLearning.learn's rounds as one call: grouped gathers (state, next_state, x, reward_f32,
 * terminated of loop->batch; action = x + S), pa_iql_step per round; picks: host int32 [rounds][2]. */
int pa_iql_learn(const pa_iql_step_args* step0, pa_arena* arena, const pa_ac_loop_args* loop,
                 const int32_t* picks, void* stream);

/* ProximalPolicyOptimization.learn's training rounds as ONE call (policy_learner.py:190-231 around
 * ppo.py:152-192; after preprocess_replay_buffer).  Per group of gather_rounds rounds: one arena
 * gather of x = state || one-hot(action) rows and one pa_gather_planes of the three per-transition
 * columns; per round: pa_ppo_rowstep, then pa_mlp_adam2 (or two pa_mlp_adam when the optimizers
 * differ).  The arena's actions are indices (one element per transition). */
typedef struct pa_ppo_learn_args {
  pa_mlp* actor; pa_mlp* critic;
  int32_t B, S, A;
  int32_t rounds, gather_rounds;       /* gather_rounds x B <= rows in the arena */
  const int64_t* idx_lists;            /* device [rounds][B] logical indices */
  const float* planes; int64_t plane_stride;  /* device [3][plane_stride]: gae, lam_return,
                                                 action_probs in logical (rollout) order */
  float* x;                            /* workspace [gather_rounds * B][S + A] */
  float* planes_ws;                    /* workspace [3][gather_rounds * B] */
  float epsilon, entropy_scale, value_grad_scale;
  float* d_logits; float* d_value;     /* scratch [B][A], [B] */
  float* losses; int64_t losses_stride;/* [rounds][losses_stride >= 2]: actor loss, critic loss */
  int64_t actor_step, critic_step;     /* AdamW step of the first round (1-based) */
} pa_ppo_learn_args;
int pa_ppo_learn(const pa_ppo_learn_args* args, pa_arena* arena, void* stream);

/* tuning aid (tools/prof_sac.py): in-kernel phase stamps of the two fused row kernels */
int pa_debug_sac_prof(long long* rows_a, long long* rows_b);
/* Fused row steps of 32 rows per workgroup (launches of more than 256 row tiles: PPO's 4096-row
 * minibatch) run their forward GEMMs on the bf16 matrix pipe at fp32 accuracy (bf16x3 split operands,
 * mlp_rowstep.hpp) when every layer input is at most 256 wide — and, when no input gradient is asked
 * for, their backward GEMMs as well (W^T planes; PEARL_AMD_ROWSTEP_SPLIT_BWD=0: forward only).
 * mode -1: that default; 0: fp32 MFMA everywhere (also PEARL_AMD_ROWSTEP_SPLIT=0); 1: bf16x3 forward,
 * fp32-MFMA backward.  pa_rowstep_last_split: what the most recent fused row step ran — 0: fp32,
 * 1: bf16x3 forward, 2: bf16x3 forward and backward (bench lines report the pipe). */
int pa_debug_set_rowstep_split(int32_t mode);
int pa_rowstep_last_split(void);
/* the same for the fused row step (mlp_rowstep.hpp): [workgroup][8][16] ticks of the next launches */
int pa_debug_rowstep_prof(long long* stamps);
int pa_debug_mlp_dw_prof(long long* stamps);   /* weight_grad_kernel launches of the MLP engine */
/* HIP-event timing of the two fused row launches (first 64 steps after enabling): bench lines */
int pa_sac_timing(int32_t enable);
int pa_sac_timing_read(double* rows_a_us, double* rows_b_us, int64_t* steps);
/* the same for the MLP engine (PPO, bandit, twin-critic steps): the first 64 fused row-step launches
 * (which = 0) and weight-gradient (+ AdamW) launches (which = 1) after enabling */
int pa_mlp_timing(int32_t enable);
int pa_mlp_timing_read(int32_t which, double* avg_us, int64_t* launches);

/* VanillaActorNetwork.get_action_prob (actor_networks.py:155-176): softmax(logits) . action_rep.
 * probs_out [B, A] may be NULL. */
int pa_softmax_action_prob(const float* logits, int32_t ldl, const float* action_rep, int32_t lda,
                           int32_t B, int32_t A, float* probs_out, float* action_prob_out,
                           void* stream);
/* ProximalPolicyOptimization._actor_loss (ppo.py:152-183) and its gradient w.r.t. the logits:
 * -sum min(r g, clamp(r, 1-eps, 1+eps) g) - entropy_scale * H(Categorical(p_batch)). */
int pa_ppo_actor_loss(const float* logits, int32_t ldl, const float* action_rep, int32_t lda,
                      const float* p_old, const float* gae, int32_t B, int32_t A, float epsilon,
                      float entropy_scale, float* d_logits, int32_t ldd, float* loss_out,
                      void* stream);
/* The actor head above and the critic's MSE head of the same minibatch (ppo.py:152-192,
 * critic_utils.py:139-168) in one launch; losses[0] = actor loss, losses[1] = critic loss;
 * d_value = (2 / B)(value - value_target).  Same values as pa_ppo_actor_loss + pa_mse_head. */
int pa_ppo_heads(const float* logits, int32_t ldl, const float* action_rep, int32_t lda,
                 const float* p_old, const float* gae, int32_t B, int32_t A, float epsilon,
                 float entropy_scale, float* d_logits, int32_t ldd, const float* value, int32_t ldv,
                 const float* value_target, float* d_value, float* losses, void* stream);

/* ---- fused row steps (pearl_amd/csrc/mlp_rowstep.hpp): forward (activations kept) + the
 * row-local part of the loss + backward down to every pre-activation gradient, ONE launch for two
 * networks on the same input.  They leave both networks' weight gradients pending exactly like
 * pa_mlp_backward2(want_dw = 2): follow with pa_mlp_adam2, or pa_mlp_flush_grads2 -> all-reduce ->
 * pa_mlp_adamw2.  Gradients are bit-identical to pa_mlp_forward2 -> heads -> pa_mlp_backward2; the
 * reported losses are summed per 16-row tile and then in tile order (equal to rounding). */
/* 1 when the two networks qualify: every layer <= 256 wide, same depth and input width;
 * ppo_actions > 0: h1 is an actor with that many (<= 32) outputs and h2 a one-output critic;
 * ppo_actions == 0: two one-output critics.  0 also under PEARL_AMD_ROWSTEP=0. */
int pa_rowstep_supported(const pa_mlp* h1, const pa_mlp* h2, int32_t ppo_actions);
/* ProximalPolicyOptimization's actor and critic steps of one minibatch (ppo.py:152-192,
 * actor_critic_base.py:309-366; critic_utils.py:139-167): what pa_mlp_forward2(keep) ->
 * pa_ppo_heads -> pa_mlp_backward2(want_dw = 2) compute.  d_value = value_grad_scale (v - target)
 * (2 / B, or 2 / (B world) under data parallelism); losses[0] = actor, losses[1] = critic.
 * logits_out / value_out may be NULL. */
int pa_ppo_rowstep(pa_mlp* actor, pa_mlp* critic, const float* x, int32_t ldx, int32_t B,
                   const float* action_rep, int32_t lda, const float* p_old, const float* gae,
                   float epsilon, float entropy_scale, const float* value_target,
                   float value_grad_scale, float* logits_out, int32_t ldl, float* value_out,
                   int32_t ldv, float* d_logits, int32_t ldd, float* d_value, float* losses,
                   void* stream);
/* Twin critics against one target (twin_critic_action_value_loss, critic_utils.py:170-203):
 * d_q_i = grad_scale (q_i - target), loss_out[0] = loss_scale (mse_1 + mse_2).  q*_out may be NULL. */
int pa_mse_rowstep2(pa_mlp* c1, pa_mlp* c2, const float* x, int32_t ldx, int32_t B,
                    const float* target, float grad_scale, float loss_scale, float* q1_out,
                    float* q2_out, float* d_q1, float* d_q2, float* loss_out, void* stream);
/* Discrete SoftActorCritic (soft_actor_critic.py:180-287) on the same fused launch.
 * pa_dsac_actor_rowstep: actor forward (kept) -> pa_dsac_actor_head's row math -> backward; the
 * weight gradients stay pending for pa_mlp_adam.  pa_dsac_target_rowstep: actor forward on the next
 * states -> pa_dsac_target's row math -> y; nothing kept.  pa_rowstep_supported(actor, NULL, A). */
int pa_dsac_actor_rowstep(pa_mlp* actor, const float* x, int32_t ldx, int32_t B, const float* q1,
                          const float* q2, const uint8_t* mask, const float* alpha, float* d_logits,
                          int32_t ldd, float* h_out, float* loss_out, void* stream);
int pa_dsac_target_rowstep(pa_mlp* actor, const float* next_state, int32_t ldx, int32_t B,
                           const float* q1, const float* q2, const uint8_t* mask, const float* alpha,
                           const float* reward, const uint8_t* term, float gamma, float* y,
                           void* stream);
/* The neural-linear bandit's network step with unit weights (neural_linear_bandit.py:176-199) on the
 * same launch: d_pred = 2 (pred - y) / B, loss_out[0] = mean (pred - y)^2; the kept activations
 * serve pa_mlp_copy_activation as after pa_mlp_forward(keep); pred_out [B] may be NULL.
 * pa_rowstep_supported(net, NULL, 0). */
int pa_wmse_rowstep(pa_mlp* net, const float* x, int32_t ldx, int32_t B, const float* y,
                    float* pred_out, float* d_pred, float* loss_out, void* stream);
/* The same launch for every criterion / output activation NeuralLinearBandit accepts
 * (loss_type, output_activation_name: neural_linear_bandit.py:68-72, :176-199;
 * neural_networks/common/utils.py:60-72 LossType): pred = act(z), d_pred = d mean(criterion) / d z.
 * pred_pre_out [B] (the network output z) and pred_out [B] (act(z)) may be NULL; with a linear
 * activation they are the same values. */
enum { PA_LOSS_MSE = 0, PA_LOSS_MAE = 1, PA_LOSS_BCE = 2 };   /* nn.functional.mse_loss / l1_loss /
                                                                binary_cross_entropy, reduction none */
enum { PA_OUT_LINEAR = 0, PA_OUT_SIGMOID = 1 };
int pa_wloss_rowstep(pa_mlp* net, const float* x, int32_t ldx, int32_t B, const float* y,
                     int32_t loss_kind, int32_t out_act, float* pred_pre_out, float* pred_out,
                     float* d_pred, float* loss_out, void* stream);
/* nn.MSELoss head (critic_utils.py:139-203): d_pred = grad_scale * (pred - target);
 * loss_out (=|+=) mean((pred - target)^2) * loss_scale. */
int pa_mse_head(const float* pred, int32_t ldp, const float* target, int32_t B, float grad_scale,
                float loss_scale, int32_t accumulate, float* d_pred, float* loss_out, void* stream);
/* preprocess_replay_buffer's GAE / lambda-return recurrence (ppo.py:271-293), logical order
 * (index 0 = oldest transition); parallel over episodes, sequential fp32 inside one. */
int pa_ppo_gae(const float* reward, const uint8_t* terminated, const uint8_t* truncated,
               const float* values, const float* next_value_last, float gamma, float lam, int64_t N,
               float* gae_out, float* lam_return_out, void* stream);
/* GaussianActorNetwork.sample_action(get_log_prob=True) (actor_networks.py:551-591) from the
 * network head [B, 2A] = mean | raw log_std and caller-supplied standard-normal noise [B, A]. */
int pa_gauss_sample(const float* head, int32_t ldh, const float* noise, int32_t ldn,
                    const float* low, const float* high, int32_t B, int32_t A, float* action,
                    int32_t lda, float* log_prob, void* stream);
/* d/d head of mean(alpha * log_prob - q) given dL/d action through the critic(s). */
int pa_gauss_actor_grad(const float* head, int32_t ldh, const float* noise, int32_t ldn,
                        const float* low, const float* high, const float* dl_daction,
                        const float* dl_daction2, int32_t ldda, const float* alpha, int32_t B,
                        int32_t A, float* d_head, int32_t lddh, void* stream);
/* Discrete SoftActorCritic (soft_actor_critic.py:180-287).
 * pa_expand_state_actions: x[b * A + i] = state[b] || rep[b, i] — the (B, A, S + AD) input of
 *   TwinCritic.get_q_values on an action set (rep_bstride = A * AD, or 0 for one shared table).
 * pa_dsac_actor_head: P = softmax(logits); loss = mean over (B, A) of P (alpha log(P + 1e-8) - q),
 *   q = min(q1, q2) with masked (unavailable) actions at 0 (:254-287); d_logits = its gradient
 *   through the softmax; h_out[b] = sum_j P log(P + 1e-8) (minus the row's entropy: the input of
 *   the entropy-coefficient step, :134-151, through pa_sac_alpha_step).
 * pa_dsac_target: y = (sum_j (q_j - alpha log(P_j + 1e-8)) P_j) gamma (1 - term) + reward
 *   (:180-252), q and P of the NEXT state. */
int pa_expand_state_actions(const float* state, int32_t ld_state, const float* rep,
                            int64_t rep_bstride, int32_t B, int32_t A, int32_t S, int32_t AD,
                            float* x_out, void* stream);
int pa_dsac_actor_head(const float* logits, int32_t ldl, const float* q1, const float* q2,
                       const uint8_t* mask, const float* alpha, int32_t B, int32_t A,
                       float* d_logits, int32_t ldd, float* loss_out, float* h_out, void* stream);
int pa_dsac_target(const float* logits, int32_t ldl, const float* q1, const float* q2,
                   const uint8_t* mask, const float* alpha, const float* reward,
                   const uint8_t* terminated, float gamma, int32_t B, int32_t A, float* y,
                   void* stream);

/* DeepTDLearning(is_conservative=True) (deep_td_learning.py:292-331, loss_fn_utils.py:17-72).
 * EXPERIMENTAL: written against the pinned oracle, not yet validated on the GPU; the Python learner
 * only reaches it with PEARL_AMD_EXPERIMENTAL_CQL=1.  Head of the (B + B A)-row pass: q_rows[0, B)
 * = Q(s_b, a_b), q_rows[B + b A + i] = Q(s_b, available action i) (padded, unmasked).  dq_rows
 * gets 2 (q - y) / B on the first B rows and alpha (softmax_i / B - count[b, i] / (B AD)) on the
 * rest, count = how often i appears in long(action[b, :]) (the reference's gather index);
 * loss_out[0] = mean |q - y| (the reported loss), loss_out[1] = mse + alpha cql. */
int pa_cql_head(const float* q_rows, const float* y, const float* action, int32_t lda, int32_t B,
                int32_t A, int32_t AD, float alpha, float* dq_rows, float* loss_out, void* stream);

/* ImplicitQLearning (implicit_q_learning.py:159-285).
 * pa_iql_value_head: expectile regression of V(s) (v, pitch ldv) towards a target critic's Q(s, a)
 *   (tq_value): loss = mean(w d^2), d = tq - v, w = expectile if d > 0 else 1 - expectile (:186-196,
 *   :271-285), dv = its gradient; adv_out = min(exp((tq_actor - v) temperature), adv_clamp), the
 *   detached advantage weights of the policy extraction (:203-215).
 * pa_awr_head: advantage-weighted regression (:197-246).  mode 0 (deterministic actor): x = the
 *   predicted actions, loss = mean_b(adv_b mean_j (x - action)^2).  mode 1 (softmax actor): x = the
 *   logits, action one-hot, loss = -mean_b(adv_b log softmax(x)[argmax action]).  dx = d loss / d x. */
int pa_iql_value_head(const float* tq_value, const float* tq_actor, const float* v, int32_t ldv,
                      float expectile, float temperature, float adv_clamp, int32_t B, float* dv,
                      float* adv_out, float* loss_out, void* stream);
int pa_awr_head(int32_t mode, const float* x, int32_t ldx, const float* action, int32_t lda,
                const float* adv, int32_t B, int32_t A, float* dx, int32_t lddx, float* loss_out,
                void* stream);
/* IQL policy extraction with a GaussianActorNetwork (implicit_q_learning.py:231-243 through
 * GaussianActorNetwork.get_log_probability, actor_networks.py:593-629): from the network head
 * [B, 2A] = mean | raw log_std and the dataset actions, log_prob[b] = log pi(a_b | s_b),
 * loss_out[0] = -mean_b(adv_b log_prob_b) and d_head = its gradient. */
int pa_gauss_awr_head(const float* head, int32_t ldh, const float* action, int32_t lda,
                      const float* low, const float* high, const float* adv, int32_t B, int32_t A,
                      float* d_head, int32_t lddh, float* log_prob, float* loss_out, void* stream);

/* ------------------------------------------------------------------------ */
/* Row-local heads of the generic TD learner (qheads.hip): DeepQLearning /   */
/* DoubleDQN / DeepSARSA over Q-network architectures the fused pa_dqn_*      */
/* kernels do not cover — VanillaQValueNetwork of any depth,                  */
/* VanillaQValueMultiHeadNetwork (q_value_networks.py:185-249) and            */
/* DuelingQValueNetwork (:352-508) — around the pa_mlp engine.                */
/* ------------------------------------------------------------------------ */
/* get_next_state_values + Bellman target from per-action values [B, A]:
 *   q_sel == NULL: v = max over unmasked i of q_val[b, i]   (deep_q_learning.py:130-167)
 *   q_sel != NULL: i* = first argmax over unmasked i of q_sel, v = q_val[b, i*]  (double_dqn.py:29-57)
 *   A == 1, no mask: v = q_val[b]   (DeepSARSA, deep_sarsa.py:59-97)
 * y = v gamma (1 - terminated) + reward, one rounding per op (deep_td_learning.py:313-317). */
int pa_td_target(const float* q_val, int32_t ldv, const float* q_sel, int32_t lds,
                 const uint8_t* mask, int32_t ldm, const float* reward, const uint8_t* terminated,
                 float gamma, int32_t B, int32_t A, float* next_v, float* y, void* stream);
/* DoubleDQN's action choice (double_dqn.py:40-51): idx_out[b] = first argmax over the unmasked
 * entries of q[b, :], rep_out[b, :AD] = rep[b, idx, :] (rep_bstride = A AD, or 0 for one shared
 * table).  Either output may be NULL. */
int pa_argmax_rows(const float* q, int32_t ldq, const uint8_t* mask, int32_t ldm, const float* rep,
                   int64_t rep_bstride, int32_t B, int32_t A, int32_t AD, int32_t* idx_out,
                   float* rep_out, void* stream);
/* dq = grad_scale (q - y) (MSELoss(mean): grad_scale = 2 / (B world)); loss_out2[0] = mean |q - y|
 * (the reported loss, deep_td_learning.py:358-359), loss_out2[1] = mean (q - y)^2. */
int pa_td_head(const float* q, int32_t ldq, const float* y, int32_t B, float grad_scale, float* dq,
               float* loss_out2, void* stream);
/* Multi-head network: Q(s_b, a_b) = rep[b, :] . f[b, :] (torch.bmm with a one-hot action,
 * q_value_networks.py:232-238); its gradient d f = dq[b] rep[b, :]; and over an action set
 * q[b, i] = rep[b, i, :] . f[b, :] (rep_bstride = Q A, or 0 for one shared [Q, A] table). */
int pa_rows_dot(const float* f, int32_t ldf, const float* rep, int32_t ldr, int32_t B, int32_t A,
                float* out, void* stream);
int pa_rows_scale(const float* dq, const float* rep, int32_t ldr, int32_t B, int32_t A, float* out,
                  int32_t ldo, void* stream);
int pa_rows_bmm(const float* rep, int64_t rep_bstride, const float* f, int32_t ldf, int32_t B,
                int32_t Q, int32_t A, float* out, void* stream);
/* Dueling network (q_value_networks.py:474-506): out[b, i] = (v[b] + adv_q[b, i]) - mean_j adv_mean[b, j]
 * (adv_mean NULL: the mean runs over the Q query actions themselves).  pa_dueling_grad: the
 * gradient for the (B + B M)-row advantage pass of the taken-action forward — rows [0, B) get dq,
 * rows B + b M + i get -dq[b] / M (M = 0: zeros; the value tower's gradient is dq itself).
 * pa_dueling_feat_grad: dfeat[b, :H] (+)= dX[b, :H] + sum_i dX[B + b M + i, :H]. */
int pa_dueling_q(const float* v, const float* adv_q, int32_t Q, const float* adv_mean, int32_t M,
                 int32_t B, float* out, void* stream);
int pa_dueling_grad(const float* dq, int32_t B, int32_t M, float* d_adv_rows, void* stream);
int pa_dueling_feat_grad(const float* dX, int32_t ldx, int32_t B, int32_t M, int32_t H,
                         int32_t accumulate, float* dfeat, int32_t ldf, void* stream);
/* The CQL term (utils/functional_utils/learning/loss_fn_utils.py:17-72) on the Q networks that
 * evaluate every action from one pass.  Dueling: the all-actions table Q_all[b, i] = V + A_avail[b, i]
 * - mean_k A_avail[b, k] reads the taken-action forward's own advantage rows, so ONE kept pass serves
 * the MSE term and the table; pa_dueling_cql_grad adds their gradients: d_value[b] = dq[b] +
 * sum_i dq_all[b, i]; d_adv rows [0, B) = dq; rows B + b M + i = -dq[b] / M + dq_all[b, i] -
 * (sum_k dq_all[b, k]) / M.  Multi-head: the table is rep[b, i, :] . f[b, :] (pa_rows_bmm);
 * pa_rows_bmm_t is its transpose, df[b, j] (+)= sum_i dq_all[b, i] rep[b, i, j]. */
int pa_dueling_cql_grad(const float* dq, const float* dq_all, int32_t B, int32_t M, float* d_adv_rows,
                        float* d_value, void* stream);
int pa_rows_bmm_t(const float* dq_all, const float* rep, int64_t rep_bstride, int32_t B, int32_t Q,
                  int32_t A, int32_t accumulate, float* df, int32_t ldf, void* stream);

/* SquareCBExploration.act's probability table (squarecb_exploration.py:59-115), one row per
 * context: p_a = 1 / (A + gamma (max_a v - v_a)), the arg-max entry rewritten to 1 - (sum of the
 * row's other entries).  Equal to the reference for B = 1, the only batch size its whole-matrix
 * complementary sum (:90) yields a distribution for.  argmax_out[B] = torch.max's index per row.
 * Sampling stays with the caller (torch's generator: a seeded run draws the reference's actions). */
int pa_squarecb_probs(const float* values, int32_t ldv, int32_t B, int32_t A, float gamma,
                      int32_t clamp_values, float reward_lb, float reward_ub, float* prob,
                      int32_t* argmax_out, void* stream);

/* Deterministic policies (DDPG ddpg.py:106-156, TD3 td3.py:106-201).
 * pa_tanh_action: VanillaContinuousActorNetwork.sample_action (actor_networks.py:448-485) from the
 * actor's pre-tanh outputs: a = ((high - low) (tanh(z) + 1)) / 2 + low.  With `noise` ([B, A]
 * N(0, sigma^2) draws; TD3's target policy smoothing, td3.py:151-175): clamp(noise, -clip, clip)
 * (high - low) / 2 is added and the sum clamped to [low, high].  `action` may point into a
 * [B, S + A] critic input (lda = S + A).
 * pa_tanh_action_grad: d_head = dL/da . da/dz of the un-noised action.
 * pa_neg_mean_head: DDPG's actor objective, loss = -mean(q), dq = -1/B. */
int pa_tanh_action(const float* head, int32_t ldh, const float* noise, int32_t ldn,
                   const float* low, const float* high, float noise_clip, int32_t B, int32_t A,
                   float* action, int32_t lda, void* stream);
int pa_tanh_action_grad(const float* head, int32_t ldh, const float* low, const float* high,
                        const float* dl_daction, int32_t ldda, int32_t B, int32_t A,
                        float* d_head, int32_t lddh, void* stream);
int pa_neg_mean_head(const float* q, int32_t ldq, int32_t B, float* dq, float* loss_out,
                     void* stream);

/* ContinuousSoftActorCritic twin-critic plumbing (soft_actor_critic_continuous.py:155-231).
 * mode 0: loss_out = mean(alpha*log_prob - min(q1,q2)); out1/out2 = dL/dq1, dL/dq2.
 * mode 1: out1 = (min(q1,q2) - alpha*log_prob) * gamma * (1 - terminated) + reward. */
int pa_sac_twin(int32_t mode, const float* q1, const float* q2, const float* log_prob,
                const float* alpha, const float* reward, const uint8_t* terminated, float gamma,
                int32_t B, float* out1, float* out2, float* loss_out, void* stream);
/* entropy autotune (soft_actor_critic_continuous.py:134-151): AdamW step `step` on the scalar
 * log_alpha with gradient mean(-exp(log_alpha) * (log_prob + target_entropy)); alpha_out = exp. */
int pa_sac_alpha_step(float* log_alpha, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                      float* alpha_out, const float* log_prob, int32_t B, float target_entropy,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      int32_t amsgrad, int64_t step, float* loss_out, void* stream);
/* NeuralLinearBandit.learn_batch loss (neural_linear_bandit.py:176-199): weighted MSE
 * sum w (pred - y)^2 / sum w and its gradient; w may be NULL (ones); wsum_out may be NULL. */
int pa_weighted_mse_head(const float* pred, int32_t ldp, const float* y, const float* w, int32_t B,
                         float* d_pred, float* loss_out, float* wsum_out, void* stream);
/* The general form (neural_linear_bandit.py:176-199 with LossType MSE / MAE / CROSS_ENTROPY and a
 * linear or sigmoid output activation): pred holds the network output z, pred_out [B] (may be
 * NULL) receives act(z), d_pred = d loss / d z with torch's backward formulas
 * (mse_loss / l1_loss / binary_cross_entropy with its -100 log clamp and 1e-12 denominator clamp;
 * sigmoid), loss_out = sum w criterion / sum w. */
int pa_weighted_loss_head(const float* pred, int32_t ldp, const float* y, const float* w, int32_t B,
                          int32_t loss_kind, int32_t out_act, float* pred_out, float* d_pred,
                          float* loss_out, float* wsum_out, void* stream);
/* LinearRegression.learn_batch (neural_networks/contextual_bandit/linear_regression.py:192-219)
 * in three steps so that the host can all-reduce the packed delta in between (:207-210):
 *   pa_linreg_delta: delta[D][D+1] = [1|f]^T [ [1|f] w | y w ], delta[D*(D+1)] = sum w   (D = d+1)
 *                    scratch: x_scratch[B*D + D], r_scratch[B*(D+1)] floats
 *   pa_linreg_apply: A += (dA + dA^T)/2, b += db, sum_weight += dsw
 *   pa_linreg_solve: inv_A = inv(A + lambda I) (:154-170, fp64 Gauss-Jordan, work[D*2D] doubles),
 *                    coefs = inv_A b (:252-259); *singular_out = 1 if a pivot vanished
 *   pa_linreg_sigma: sqrt([1|f] inv_A [1|f]^T) per row (:261-270) */
int pa_linreg_delta(const float* features, int32_t ldf, const float* y, const float* w, int32_t B,
                    int32_t d, float* x_scratch, float* r_scratch, float* delta_out, void* stream);
/* pa_linreg_delta with vector-aligned scratch rows (x_scratch[B*Dp + D], Dp = D rounded up to 4;
 * r_scratch[B*Rp], Rp = D + 1 rounded up to 2; both 16-byte aligned): the same delta, and batches of
 * >= 2048 contexts then take the bf16x3 weight-gradient loop for X^T R as well */
int pa_linreg_delta2(const float* features, int32_t ldf, const float* y, const float* w, int32_t B,
                     int32_t d, float* x_scratch, float* r_scratch, float* delta_out, void* stream);
int pa_linreg_apply(const float* delta, int32_t d, float* A, float* b, float* sum_weight,
                    void* stream);
/* the same, also writing the updated A [D][D] and b [D] into a snapshot pair (both or none): the
 * operands of a solve that runs off the learner's stream while the next step updates A and b */
int pa_linreg_apply2(const float* delta, int32_t d, float* A, float* b, float* sum_weight,
                     float* A_snap, float* b_snap, void* stream);
int pa_linreg_solve(const float* A, const float* b, float l2_reg_lambda, int32_t d, double* work,
                    float* inv_A_out, float* coefs_out, int32_t* singular_out, void* stream);
/* force_pinv (linear_regression.py:138-157): inv_A = torch.linalg.pinv(A + lambda I, hermitian=True) —
 * eigenvalues by a one-sided Jacobi iteration in fp64, those at or below (d + 1) * eps(float32) of the
 * largest dropped (torch's default rtol) — and coefs = inv_A b; *rank_out = the number kept.  For the
 * unregularised, possibly singular regression; with lambda > 0 pa_linreg_solve computes the same
 * matrix faster.  d + 1 <= 72. */
int pa_linreg_pinv(const float* A, const float* b, float l2_reg_lambda, int32_t d, float* inv_A_out,
                   float* coefs_out, int32_t* rank_out, void* stream);
/* One NeuralLinearBandit.learn_batch on unit weights in a single process, as ONE call
 * (pearl/policy_learners/contextual_bandits/neural_linear_bandit.py:139-214): pa_wloss_rowstep, the
 * LinUCB operands from the kept features (pa_linreg_delta2's), ONE weight-gradient launch forming
 * the network's gradients with AdamW step `adam_step` AND the moment update [delta_A | delta_b],
 * then pa_linreg_apply2 (the batch's weight sum read from delta_A[0][0]).  Three launches (the row
 * step writes the operands itself unless it takes the bf16x3 forward), plus the solve's two on the
 * side stream when one is given.
 *   pred [B]: act(network output); d_pred [B]: scratch that must stay untouched until the call's
 *   launches have run; scalars [2]: the loss, then the batch mean of pred;
 *   x_scratch / r_scratch / delta: as pa_linreg_delta2 (delta [D*(D+1)], D = d + 1);
 *   A [D][D], b [D], sum_weight [1], A_snap / b_snap (both or none): as pa_linreg_apply2.
 * PA_ERR_UNSUPPORTED when pa_rowstep_supported(net, NULL, 0) is 0 (callers then issue the calls
 * above one by one). */
typedef struct pa_bandit_step_args {
  pa_mlp* net;
  const float* x; int32_t ldx; int32_t B;
  const float* y;
  int32_t loss_kind, out_act;   /* PA_LOSS_*, PA_OUT_* */
  int64_t adam_step;            /* 1-based */
  float* pred; float* d_pred; float* scalars;
  int32_t d;                    /* feature width of the regression = the trunk's output width */
  float* x_scratch; float* r_scratch; float* delta;
  float* A; float* b; float* sum_weight; float* A_snap; float* b_snap;
  /* optional (side_stream null: none): pa_linreg_solve of the snapshot on a second stream, so that
   * the serial fp64 solve runs beside the next step.  ev_slot_free (nullable hipEvent_t): `stream`
   * waits for it before the snapshot is overwritten (the solve that last read this snapshot pair);
   * ev_ready / ev_done (hipEvent_t, created by the caller): recorded on `stream` after the update
   * and on side_stream after the solve.  Needs A_snap / b_snap. */
  void* side_stream; void* ev_slot_free; void* ev_ready; void* ev_done;
  float l2_reg_lambda; double* work; float* inv_A; float* coefs; int32_t* singular;
} pa_bandit_step_args;
int pa_bandit_step(const pa_bandit_step_args* args, void* stream);
int pa_linreg_sigma(const float* features, int32_t ldf, const float* inv_A, int32_t B, int32_t d,
                    float* sigma_out, void* stream);
/* out[B, nl + nr] = left[B, nl] || right[B, nr]   (q_value_networks.py:166-168 torch.cat) */
int pa_concat_cols(const float* left, int32_t ldl, const float* right, int32_t ldr, float* out,
                   int32_t B, int32_t nl, int32_t nr, void* stream);

/* ------------------------------------------------------------------------ */
/* Diagnostics: single-kernel entry points so the GPU test-suite can localise */
/* a parity failure to one kernel (no reference counterpart).                 */
/* ------------------------------------------------------------------------ */
/* C[M,N] = epi(A[M,K] * op(B)).  b_is_kn = 0: B is [N,K] (y = x W^T); 1: B is [K,N].
 * epi: 0 = +bias, 1 = relu(+bias), 2 = * (hmask > 0), 3 = none. */
/* Kernel tuning aid (tools/prof_chain.py): device buffers of int64[workgroups][8 waves][16] that
 * online_rowpass_kernel / weight_grad_kernel fill with 100 MHz wall-clock stamps at their phase
 * boundaries, on every launch (round < 0) or only in round `round` of a pa_dqn_learn call.  NULL
 * (the default) disables the stamps. */
int pa_debug_set_prof(pa_dqn* h, long long* rowpass_stamps, long long* dw_stamps, int32_t round);
/* The same for target_fused_kernel: int64[max_tiles][8][16]; launches with more tiles than
 * max_tiles are not stamped. */
int pa_debug_set_prof_target(pa_dqn* h, long long* stamps, int32_t max_tiles);
int pa_debug_linear(const float* A, int32_t lda, const float* B, int32_t ldb, float* C,
                    int32_t ldc, const float* bias, const float* hmask, int32_t ldh, int32_t M,
                    int32_t N, int32_t K, int32_t b_is_kn, int32_t epi, void* stream);
/* Which main loop the weight-gradient launches take: -1 = by environment (PEARL_AMD_DW_SPLIT,
 * default on), 0 = fp32 MFMA everywhere, 1 = the bf16x3 split loop (v_mfma_f32_16x16x32_bf16 on
 * exactly split operands, fp32 accuracy) for batches of >= 2048 rows with aligned operands,
 * 2 = the same below the batch threshold (tests). */
int pa_debug_set_dw_split(int32_t mode);
/* Tile shape of the bf16x3 target kernel: 32 = 32-row tiles on four waves, two workgroups per CU (one
 * workgroup's prologue / epilogue runs beside the other's main loop); 64 = the 64-row, eight-wave
 * tile; 0 = the default: per pass (32 for Double DQN's stand-alone passes, 64 for the DQN window
 * loop), or what PEARL_AMD_TARGET_ROWS says.  Bitwise-identical results either way. */
int pa_debug_set_target_rows(int32_t rows);
/* PEARL_AMD_DEBUG_WORKERS=1: how many workgroups took tiles in the persistent target launches of learn() so far */
/* diagnostics: copy a workspace of the last learner step ("H1a", "H2a", "dZ2", "dZ1", "dq", "q") to out_dev */
int pa_debug_workspace(pa_dqn* h, const char* name, float* out_dev, int64_t n, int32_t* paired_out);
int pa_debug_target_workers(pa_dqn* h, int64_t* workers_out, int64_t* launches_out);
/* dW[M,N] = dZ[Bn,M]^T X[Bn,N], db[M] = column sums of dZ. */
int pa_debug_weight_grad(const float* dZ, int32_t ldz, const float* X, int32_t ldx, float* dW,
                         int32_t ldw, float* db, int32_t M, int32_t N, int32_t Bn, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PEARL_AMD_H */
