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

struct pa_comm {
  NcclComm comm;
  int device, world, rank;
  hipStream_t stream;      // the exchange stream (PEARL_AMD_COMM_INLINE=0 only)
  hipEvent_t ready, done;  // learner stream -> exchange stream -> learner stream
  int inline_mode;         // 1 (default): the collective is enqueued on the learner stream itself
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

extern "C" int pa_comm_destroy(pa_comm* c) {
  if (!c) return PA_OK;
  (void)hipSetDevice(c->device);
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
  if (c->inline_mode) return 0;
  return hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->done, 0) == hipSuccess ? 0 : 1;
}

// What the communicator itself reports (ncclCommCount / ncclCommUserRank): the rank count RCCL
// observed, as opposed to the one the caller asked for.  -1 where the library has no such query.
extern "C" int pa_comm_info(pa_comm* c, int32_t* ranks_out, int32_t* rank_out) {
  PA_REQUIRE(c, PA_ERR_INVALID, "pa_comm_info: null communicator");
  Rccl* r = rccl();
  PA_REQUIRE(r, PA_ERR_UNSUPPORTED, "RCCL is not available");
  int n = -1, me = -1;
  if (r->count) PA_NCCL(r->count(c->comm, &n));
  if (r->user_rank) PA_NCCL(r->user_rank(c->comm, &me));
  if (ranks_out) *ranks_out = n;
  if (rank_out) *rank_out = me;
  return PA_OK;
}
