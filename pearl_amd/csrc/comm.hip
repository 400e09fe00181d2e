// RCCL all-reduce hooks for the data-parallel learn loop.
//
// pa_dqn_learn itself links no communication library: it calls the two hooks of pa_learn_args.
// This file provides native hooks backed by RCCL (ncclAllReduce over xGMI), resolved at run time
// with dlopen so that the library loads on hosts without RCCL and shares the copy PyTorch already
// mapped when there is one.  One communicator per process (one process per GPU); the exchange
// runs on its own HIP stream, ordered against the learner stream with events, so the target-network
// pass the learn loop enqueues between start and wait overlaps the all-reduce.
#include <dlfcn.h>
#include <stdlib.h>

#include <new>

#include "common.hpp"

using namespace pa;

namespace {

struct NcclId { char bytes[128]; };   // ncclUniqueId
typedef void* NcclComm;
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(NcclComm*, int, NcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_destroy)(NcclComm);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(NcclComm, int*);

struct Rccl {
  void* lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
  fn_count count = nullptr, user_rank = nullptr;   // optional
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {   // prefer a copy that is already mapped (PyTorch's)
    r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (r.lib) break;
  }
  for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) return nullptr;
  r.get_id = (fn_get_id)dlsym(r.lib, "ncclGetUniqueId");
  r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
  r.allreduce = (fn_allreduce)dlsym(r.lib, "ncclAllReduce");
  r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
  r.errstr = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
  r.count = (fn_count)dlsym(r.lib, "ncclCommCount");
  r.user_rank = (fn_count)dlsym(r.lib, "ncclCommUserRank");
  if (!r.get_id || !r.init_rank || !r.allreduce || !r.destroy) {
    r.lib = nullptr;
    return nullptr;
  }
  return &r;
}

}  // namespace

// One-shot peer-to-peer exchange (pa_comm_create_p2p): every rank owns one device allocation
//   [2 slots][max_floats] floats | flag word (the latest round published)
// that its peers map through hipIpc.  SURVEY.md §8(e): the per-round message is 413 KB — latency,
// not bandwidth — so every rank READS the other G - 1 buffers directly over its own xGMI links
// (point to point: seven links, seven peers) and adds them up itself, instead of a ring's 2 (G - 1)
// dependent hops.
constexpr int kP2PMaxWorld = 8;
struct P2P {
  float* mine = nullptr;            // this rank's allocation
  float* peer[kP2PMaxWorld] = {};   // peer[r]: rank r's allocation as mapped here (peer[rank] = mine)
  int64_t max_floats = 0;
  unsigned round = 0;               // rounds published so far (host copy)
  int* err_host = nullptr;          // pinned, device-mapped: non-zero = a bounded wait expired.  STICKY:
                                    // once set the communicator is poisoned — the ranks are no longer in
                                    // lock-step, the two-slot reuse argument is void — and every later
                                    // exchange is refused (p2p_allreduce, pa_comm_allreduce_wait, pa_comm_p2p_check)
  long long timeout_ticks = 0;      // bound of the peer-flag wait in 100 MHz wall-clock ticks
  int opened = 0;
};

struct pa_comm {
  NcclComm comm;
  int device, world, rank;
  hipStream_t stream;      // the exchange stream (PEARL_AMD_COMM_INLINE=0 only)
  hipEvent_t ready, done;  // learner stream -> exchange stream -> learner stream
  int inline_mode;         // 1 (default): the collective is enqueued on the learner stream itself
  P2P* p2p;                // non-null: the one-shot peer-to-peer exchange instead of RCCL
};

#define PA_NCCL(expr)                                                                      \
  do {                                                                                     \
    int _e = (expr);                                                                       \
    if (_e != 0) {                                                                         \
      Rccl* _r = rccl();                                                                   \
      set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                         \
                (_r && _r->errstr) ? _r->errstr(_e) : "rccl error");                       \
      return PA_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

extern "C" int pa_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int pa_comm_unique_id(void* id128_out) {
  PA_REQUIRE(id128_out, PA_ERR_INVALID, "null output");
  Rccl* r = rccl();
  PA_REQUIRE(r, PA_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
  PA_NCCL(r->get_id(reinterpret_cast<NcclId*>(id128_out)));
  return PA_OK;
}

extern "C" int pa_comm_create(pa_comm** out, int32_t device, int32_t world, int32_t rank,
                              const void* id128) {
  PA_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, PA_ERR_INVALID,
             "pa_comm_create: bad argument");
  Rccl* r = rccl();
  PA_REQUIRE(r, PA_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
  PA_HIP(hipSetDevice(device));
  pa_comm* c = new (std::nothrow) pa_comm();
  PA_REQUIRE(c, PA_ERR_NOMEM, "out of host memory");
  memset(c, 0, sizeof(*c));
  c->device = device; c->world = world; c->rank = rank;
  {
    const char* v = getenv("PEARL_AMD_COMM_INLINE");
    c->inline_mode = (v && *v) ? atoi(v) : 1;
  }
  NcclId id;
  memcpy(&id, id128, sizeof(id));
  int e = r->init_rank(&c->comm, world, id, rank);
  if (e != 0) {
    set_error("ncclCommInitRank failed: %s", r->errstr ? r->errstr(e) : "rccl error");
    delete c;
    return PA_ERR_HIP;
  }
  PA_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PA_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  *out = c;
  return PA_OK;
}

// ---------------------------------------------------------------------------------------------
// One-shot P2P all-reduce (deterministic: every rank adds the G buffers in rank order, so all
// ranks hold bitwise-identical sums).  Two launches on the caller's stream per exchange:
//   p2p_publish_kernel   grad -> this rank's slot (round & 1), 16-byte stores.  The end of a kernel
//                        is a system-scope release: the slot is visible to the peers once the
//                        launch has completed.
//   p2p_reduce_kernel    thread 0 of block 0 publishes flag = round (system scope); every block
//                        waits (bounded) until every peer's flag has reached `round`, acquires at
//                        system scope, and writes grad[j] = sum_r slot_r[j], r = 0 .. G - 1.
// Two slots are enough: a rank rewrites slot (round & 1) two rounds later, after its own reduce of
// round + 1 — which waited for every peer's flag >= round + 1, and a peer publishes round + 1 only
// after its reduce of `round` has finished reading.
// ---------------------------------------------------------------------------------------------
namespace {
struct P2PArgs {
  float* grad; long long n;
  float* mine_slot;
  const float* slot[kP2PMaxWorld];        // every rank's slot of this round (own included)
  unsigned* flag_mine;
  const unsigned* flag[kP2PMaxWorld];
  int world, rank;
  unsigned round;
  int* err;                  // pinned host word (what the host reads): written on expiry only
  int* err_dev;              // its device twin (what the kernels read: a host word costs a PCIe round
                             // trip per block and launch — measured: 22.6 -> 18.3 M transitions/s at 1 rank)
  long long timeout_ticks;   // wall_clock64() ticks (100 MHz) a block waits for a peer's flag
};
__global__ __launch_bounds__(256) void p2p_publish_kernel(P2PArgs a) {
  const long long n4 = a.n >> 2;
  const float4* src = reinterpret_cast<const float4*>(a.grad);
  float4* dst = reinterpret_cast<float4*>(a.mine_slot);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
    dst[i] = src[i];
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) a.mine_slot[i] = a.grad[i];
}
// A wait that expires NEVER produces a sum: the block stores NaN over its share of grad (the local,
// unreduced gradient must not step the optimizer as if it were the group's: RCCL would have blocked
// here) and raises the sticky error word; blocks that start after the word is up do the same at once.
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PArgs a) {
  __shared__ int dead;
  if (threadIdx.x == 0)
    dead = __hip_atomic_load(a.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (the publish launch has completed: its stores are released at system scope already; the
    //  fence keeps this store behind anything else this thread has issued)
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __hip_atomic_store(a.flag_mine, a.round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!dead && threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank) {
    const unsigned* f = a.flag[threadIdx.x];
    const long long t0 = wall_clock64();
    int spins = 0;
    // rounds compare modulo 2^32 (a peer is at most one round ahead)
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.round) < 0) {
      __builtin_amdgcn_s_sleep(8);
      // every 256 polls: the clock, and the error word another block / an earlier round may have raised
      if ((++spins & 255) == 0 &&
          (wall_clock64() - t0 > a.timeout_ticks ||
           __hip_atomic_load(a.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        // a peer lags by more than the bound (default 30 s; a rank that checkpoints or evaluates
        // lags by far less) or is gone: report, do not hang the GPU, and do NOT use its slot
        // (a plain store: PCIe atomics to pinned host memory are not a given; when several waits
        //  expire the word names one of the late peers)
        if (__hip_atomic_load(a.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __hip_atomic_store(a.err_dev, 1 + (int)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a.err, 1 + (int)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        dead = 1;
        break;
      }
    }
  }
  __syncthreads();
  const long long n4 = a.n >> 2;
  if (dead) {
    const float nanv = __builtin_nanf("");
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
      reinterpret_cast<float4*>(a.grad)[i] = make_float4(nanv, nanv, nanv, nanv);
    if (blockIdx.x == 0)
      for (long long i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) a.grad[i] = nanv;
    return;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: nothing below reads a stale line
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 acc = reinterpret_cast<const float4*>(a.slot[0])[i];
    for (int r = 1; r < a.world; ++r) {
      const float4 v = reinterpret_cast<const float4*>(a.slot[r])[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(a.grad)[i] = acc;
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) {
      float acc = a.slot[0][i];
      for (int r = 1; r < a.world; ++r) acc += a.slot[r][i];
      a.grad[i] = acc;
    }
}
constexpr int kP2PFlagFloats = 128;  // the flag word gets a 256-byte line of its own; the next line holds
                                     // the device twin of the error word (P2PArgs::err_dev)
}  // namespace

// A communicator whose all-reduce is the one-shot P2P exchange above.  Bring-up: every rank creates
// one, hands pa_comm_p2p_handle()'s 64 bytes to every other rank (torch.distributed carries them),
// and opens its peers' with pa_comm_p2p_open(); pa_comm_allreduce_start / _wait then work as for
// the RCCL communicator (messages of up to max_floats floats, 16-byte aligned).
extern "C" int pa_comm_create_p2p(pa_comm** out, int32_t device, int32_t world, int32_t rank,
                                  int64_t max_floats) {
  PA_REQUIRE(out && world >= 1 && world <= kP2PMaxWorld && rank >= 0 && rank < world && max_floats > 0,
             PA_ERR_INVALID, "pa_comm_create_p2p: bad argument (1 <= world <= 8)");
  PA_HIP(hipSetDevice(device));
  pa_comm* c = new (std::nothrow) pa_comm();
  PA_REQUIRE(c, PA_ERR_NOMEM, "out of host memory");
  memset(c, 0, sizeof(*c));
  c->device = device; c->world = world; c->rank = rank;
  c->inline_mode = 1;
  c->p2p = new (std::nothrow) P2P();
  PA_REQUIRE(c->p2p, PA_ERR_NOMEM, "out of host memory");
  P2P* x = c->p2p;
  x->max_floats = (max_floats + 63) / 64 * 64;
  const size_t bytes = (size_t)(2 * x->max_floats + kP2PFlagFloats) * sizeof(float);
  // Fine-grained device memory when the runtime hands it out (peer reads over xGMI are then
  // coherent without relying on cache maintenance at all); ordinary coarse-grained memory otherwise
  // or under PEARL_AMD_P2P_COARSE=1 — the protocol's system-scope release (end of the publish
  // launch) / acquire (after the flag wait) pair is what makes that correct.
  {
    const char* v = getenv("PEARL_AMD_P2P_COARSE");
    const bool coarse = v && *v == '1';
    if (coarse || hipExtMallocWithFlags((void**)&x->mine, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      x->mine = nullptr;
      PA_HIP(hipMalloc((void**)&x->mine, bytes));
    }
  }
  PA_HIP(hipMemset(x->mine, 0, bytes));
  PA_HIP(hipDeviceSynchronize());
  PA_HIP(hipHostMalloc((void**)&x->err_host, 16, hipHostMallocMapped));
  x->err_host[0] = 0;
  {
    // RCCL would block for as long as a peer takes; this exchange bounds the wait so that a dead
    // peer cannot hang the GPU — generously (PEARL_AMD_P2P_TIMEOUT_S, default 30 s)
    const char* v = getenv("PEARL_AMD_P2P_TIMEOUT_S");
    double sec = (v && *v) ? atof(v) : 30.0;
    if (!(sec > 0.0)) sec = 30.0;
    x->timeout_ticks = (long long)(sec * 1e8);
  }
  x->peer[rank] = x->mine;
  x->opened = 1;
  *out = c;
  return PA_OK;
}
extern "C" int pa_comm_p2p_handle(pa_comm* c, void* handle64_out) {
  PA_REQUIRE(c && c->p2p && handle64_out, PA_ERR_INVALID, "pa_comm_p2p_handle: not a P2P communicator");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipIpcGetMemHandle(&h, c->p2p->mine));
  memcpy(handle64_out, &h, sizeof(h));
  return PA_OK;
}
extern "C" int pa_comm_p2p_open(pa_comm* c, int32_t peer, const void* handle64) {
  PA_REQUIRE(c && c->p2p && handle64 && peer >= 0 && peer < c->world && peer != c->rank,
             PA_ERR_INVALID, "pa_comm_p2p_open: bad argument");
  PA_REQUIRE(!c->p2p->peer[peer], PA_ERR_INVALID, "pa_comm_p2p_open: peer %d is open already", peer);
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  PA_HIP(hipSetDevice(c->device));
  void* p = nullptr;
  PA_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  c->p2p->peer[peer] = static_cast<float*>(p);
  c->p2p->opened += 1;
  return PA_OK;
}
namespace {
bool p2p_poisoned(pa_comm* c) {
  const int e = __atomic_load_n(c->p2p->err_host, __ATOMIC_RELAXED);
  if (e == 0) return false;
  set_error("P2P exchange: the wait for rank %d's gradient expired (PEARL_AMD_P2P_TIMEOUT_S); that "
            "round's gradient was overwritten with NaN and the communicator is poisoned — the ranks "
            "are no longer in lock-step",
            e - 1);
  return true;
}
}  // namespace
// non-zero status when a bounded wait for a peer expired since the communicator was created
// (sticky).  Cheap — one read of a pinned host word — and meant for the hot path: the learners call
// it after the host sync that ends every learn() / step.
extern "C" int pa_comm_p2p_check(pa_comm* c) {
  PA_REQUIRE(c && c->p2p, PA_ERR_INVALID, "pa_comm_p2p_check: not a P2P communicator");
  return p2p_poisoned(c) ? PA_ERR_HIP : PA_OK;
}
// any communicator: PA_OK for RCCL ones (RCCL blocks instead of expiring), the sticky P2P status otherwise
extern "C" int pa_comm_check(pa_comm* c) {
  PA_REQUIRE(c, PA_ERR_INVALID, "pa_comm_check: null communicator");
  return (c->p2p && p2p_poisoned(c)) ? PA_ERR_HIP : PA_OK;
}
// the largest message (floats) one exchange takes: the P2P slot size; 0 = unlimited (RCCL)
extern "C" int64_t pa_comm_max_floats(pa_comm* c) { return (c && c->p2p) ? c->p2p->max_floats : 0; }

namespace {
int p2p_allreduce(pa_comm* c, float* buf, int64_t n, hipStream_t s) {
  P2P* x = c->p2p;
  if (p2p_poisoned(c)) return 1;      // (set_error inside)
  if (x->opened != c->world) {
    set_error("P2P exchange: %d of %d peer buffers are mapped", x->opened, c->world);
    return 1;
  }
  if (n > x->max_floats || (reinterpret_cast<uintptr_t>(buf) & 15)) {
    set_error("P2P exchange: a message of %lld floats at %p does not fit (slots of %lld floats, 16-byte "
              "aligned; PEARL_AMD_P2P_FLOATS) — callers chunk or fall back (pearl_amd/_comm.py)",
              (long long)n, (void*)buf, (long long)x->max_floats);
    return 1;
  }
  x->round += 1;
  const int slot = (int)(x->round & 1u);
  P2PArgs a;
  memset(&a, 0, sizeof(a));
  a.grad = buf; a.n = n;
  a.mine_slot = x->mine + (int64_t)slot * x->max_floats;
  a.flag_mine = reinterpret_cast<unsigned*>(x->mine + 2 * x->max_floats);
  for (int r = 0; r < c->world; ++r) {
    a.slot[r] = x->peer[r] + (int64_t)slot * x->max_floats;
    a.flag[r] = reinterpret_cast<const unsigned*>(x->peer[r] + 2 * x->max_floats);
  }
  a.world = c->world; a.rank = c->rank; a.round = x->round;
  a.err = x->err_host;
  a.err_dev = reinterpret_cast<int*>(x->mine + 2 * x->max_floats + 64);
  a.timeout_ticks = x->timeout_ticks;
  unsigned grid = (unsigned)((n / 4 + 255) / 256);
  if (grid > 208) grid = 208;   // one wave of blocks: every block polls the peers' flags once
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(p2p_publish_kernel, dim3(grid), dim3(256), 0, s, a);
  hipLaunchKernelGGL(p2p_reduce_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace

extern "C" int pa_comm_destroy(pa_comm* c) {
  if (!c) return PA_OK;
  (void)hipSetDevice(c->device);
  if (c->p2p) {
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
      if (r != c->rank && c->p2p->peer[r]) (void)hipIpcCloseMemHandle(c->p2p->peer[r]);
    if (c->p2p->mine) (void)hipFree(c->p2p->mine);
    if (c->p2p->err_host) (void)hipHostFree(c->p2p->err_host);
    delete c->p2p;
    delete c;
    return PA_OK;
  }
  (void)hipStreamSynchronize(c->stream);
  Rccl* r = rccl();
  if (r && c->comm) (void)r->destroy(c->comm);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return PA_OK;
}

// pa_learn_args.allreduce_start: SUM-all-reduce buf[n] on the exchange stream, after everything
// already enqueued on `stream`.
extern "C" int pa_comm_allreduce_start(void* ctx, float* buf, int64_t n, void* stream) {
  pa_comm* c = reinterpret_cast<pa_comm*>(ctx);
  if (c && c->p2p) {
    if (!buf || n <= 0) return 1;
    return p2p_allreduce(c, buf, n, reinterpret_cast<hipStream_t>(stream));
  }
  Rccl* r = rccl();
  if (!c || !r || !buf || n <= 0) return 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (c->inline_mode) {
    // The exchange sits on the online chain's critical path anyway (AdamW needs the reduced
    // gradient, the next forward needs AdamW), while the target-network work it could overlap
    // with already runs on pa_dqn_learn's side stream.  Enqueuing the collective directly on the
    // learner stream saves two event hops (~10 us each on this stack) per round.
    return r->allreduce(buf, buf, (size_t)n, /*ncclFloat*/ 7, /*ncclSum*/ 0, c->comm, s) == 0 ? 0 : 1;
  }
  if (hipEventRecord(c->ready, s) != hipSuccess) return 1;
  if (hipStreamWaitEvent(c->stream, c->ready, 0) != hipSuccess) return 1;
  if (r->allreduce(buf, buf, (size_t)n, /*ncclFloat*/ 7, /*ncclSum*/ 0, c->comm, c->stream) != 0)
    return 1;
  if (hipEventRecord(c->done, c->stream) != hipSuccess) return 1;
  return 0;
}

// pa_learn_args.allreduce_wait: `stream` waits for the exchange started last.
extern "C" int pa_comm_allreduce_wait(void* ctx, void* stream) {
  pa_comm* c = reinterpret_cast<pa_comm*>(ctx);
  if (!c) return 1;
  // (P2P: the exchange is asynchronous, so this sees an expiry of an EARLIER round at the latest
  //  one round later; the learners' end-of-call pa_comm_check catches the last one)
  if (c->p2p && p2p_poisoned(c)) return 1;
  if (c->inline_mode) return 0;
  return hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->done, 0) == hipSuccess ? 0 : 1;
}

// What the communicator itself reports (ncclCommCount / ncclCommUserRank): the rank count RCCL
// observed, as opposed to the one the caller asked for.  -1 where the library has no such query.
extern "C" int pa_comm_info(pa_comm* c, int32_t* ranks_out, int32_t* rank_out) {
  PA_REQUIRE(c, PA_ERR_INVALID, "pa_comm_info: null communicator");
  if (c->p2p) {   // the buffers actually mapped (own + opened peers), not the world that was asked for
    if (ranks_out) *ranks_out = c->p2p->opened;
    if (rank_out) *rank_out = c->rank;
    return PA_OK;
  }
  Rccl* r = rccl();
  PA_REQUIRE(r, PA_ERR_UNSUPPORTED, "RCCL is not available");
  int n = -1, me = -1;
  if (r->count) PA_NCCL(r->count(c->comm, &n));
  if (r->user_rank) PA_NCCL(r->user_rank(c->comm, &me));
  if (ranks_out) *ranks_out = n;
  if (rank_out) *rank_out = me;
  return PA_OK;
}
