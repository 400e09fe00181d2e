// One ContinuousSoftActorCritic.learn_batch as ONE C-ABI call
// (pearl/policy_learners/sequential_decision_making/soft_actor_critic_continuous.py:131-231 on top of
// actor_critic_base.py:309-366): actor update -> critic update -> critic-target soft update ->
// entropy-coefficient step, in the reference's order, every arithmetic step the same pa_* launch the
// Python-sequenced path issues (bit-identical results).
//
// Why it exists: the step is ~22 short launches.  Sequenced from Python — a ctypes call, a torch
// allocation or two and a handful of attribute lookups per launch — the host needed 330 us per step
// while the kernels summed to ~250 us (tools/host_bound.py): the learner was host-bound.  Here the
// host side of a launch is a C++ function call; the caller passes its batch, the two noise draws
// and one scratch buffer.
#include <stdint.h>
#include <string.h>

#include <stdlib.h>

#include "common.hpp"
#include "host_launch.hpp"
#include "mlp_internal.hpp"
#include "sac_rows.hpp"

using namespace pa;

#define PA_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != PA_OK) return _rc; \
  } while (0)


namespace {
// scratch carve-up (floats); every piece 16-byte aligned
struct SacScratch {
  float *xa, *head, *logp, *q1, *q2, *dq1, *dq2, *dx1, *dx2, *d_head;
  float *xn, *head_n, *nlogp, *nq1, *nq2, *y, *xq, *qa, *qb, *dqa, *dqb;
};
int64_t a4(int64_t x) { return (x + 3) & ~int64_t(3); }
int64_t carve(SacScratch* s, float* base, int64_t B, int64_t S, int64_t A) {
  int64_t o = 0;
  auto take = [&](float** p, int64_t n) {
    if (p && base) *p = base + o;
    o += a4(n);
  };
  take(s ? &s->xa : nullptr, B * (S + A));
  take(s ? &s->head : nullptr, B * 2 * A);
  take(s ? &s->logp : nullptr, B);
  take(s ? &s->q1 : nullptr, B);
  take(s ? &s->q2 : nullptr, B);
  take(s ? &s->dq1 : nullptr, B);
  take(s ? &s->dq2 : nullptr, B);
  take(s ? &s->dx1 : nullptr, B * (S + A));
  take(s ? &s->dx2 : nullptr, B * (S + A));
  take(s ? &s->d_head : nullptr, B * 2 * A);
  take(s ? &s->xn : nullptr, B * (S + A));
  take(s ? &s->head_n : nullptr, B * 2 * A);
  take(s ? &s->nlogp : nullptr, B);
  take(s ? &s->nq1 : nullptr, B);
  take(s ? &s->nq2 : nullptr, B);
  take(s ? &s->y : nullptr, B);
  take(s ? &s->xq : nullptr, B * (S + A));
  take(s ? &s->qa : nullptr, B);
  take(s ? &s->qb : nullptr, B);
  take(s ? &s->dqa : nullptr, B);
  take(s ? &s->dqb : nullptr, B);
  return o;
}
}  // namespace

int64_t carve_fused_size(int64_t B, int64_t S, int64_t A);
extern "C" int64_t pa_sac_scratch_floats(int32_t B, int32_t S, int32_t A) {
  const int64_t a = carve(nullptr, nullptr, B, S, A), b = carve_fused_size(B, S, A);
  return a > b ? a : b;
}


namespace {

// ---- the fused form (sac_rows.hpp): two row launches + the weight-gradient launches --------------
bool fused_enabled() {   // read per call: tests compare the two forms in one process
  const char* v = getenv("PEARL_AMD_SAC_FUSED");
  return !(v && *v == '0');
}

bool relu3(const pa_mlp* h) {
  return h->bound && h->row_ok && h->L == 3 && h->d.identity_layers == 0 && !h->d.no_last_bias &&
         h->bufs.grad && h->bufs.exp_avg;
}

bool fused_ok(const pa_sac_step_args* a) {
  if (!fused_enabled()) return false;
  const pa_mlp *ac = a->actor, *c1 = a->critic1, *c2 = a->critic2;
  if (!relu3(ac) || !relu3(c1) || !relu3(c2)) return false;
  if (!c1->bufs.p_target || !c2->bufs.p_target) return false;
  if (a->A < 1 || a->A > 16 || a->S + a->A > ROW_MAX_IN) return false;
  if (ac->d.dims[0] != a->S || ac->d.dims[3] != 2 * a->A) return false;
  for (const pa_mlp* c : {c1, c2}) {
    if (c->d.dims[0] != a->S + a->A || c->d.dims[3] != 1) return false;
    if (c->d.dims[1] != c1->d.dims[1] || c->d.dims[2] != c1->d.dims[2]) return false;
  }
  const int B = a->B;
  return B <= ac->d.max_batch && B <= c1->d.max_batch && B <= c2->d.max_batch;
}

void fill_net(const pa_mlp* h, bool target, SacMlp3& n) {
  const float* P = target ? h->bufs.p_target : h->bufs.p;
  float* const* wf = target ? h->wf_t : h->wf;
  n.W1f = wf[0]; n.b1 = P + h->boff[0];
  n.W2f = wf[1]; n.b2 = P + h->boff[1];
  n.W3f = wf[2]; n.b3 = P + h->boff[2];
  n.w3 = P + h->woff[2];
  n.W1tf = h->wtf[0]; n.W2tf = h->wtf[1]; n.W3tf = h->wtf[2];
  n.act1 = h->act[0]; n.act2 = h->act[1];
  n.dz1 = h->dz[1]; n.dz2 = h->dz[2];
  n.K0 = h->d.dims[0]; n.H1 = h->d.dims[1]; n.H2 = h->d.dims[2]; n.DO = h->d.dims[3];
  n.um = mlp_um(h, target);    // (null unless something asked for them: mlp_ensure_um)
}
// the fp16x2 instantiations of the row kernels (sac_rows.hpp, H2): hidden layers exactly 256 wide
bool h2_enabled() {
  const char* v = getenv("PEARL_AMD_SAC_H2");      // read per call: tests compare the forms
  return !(v && v[0] == '0');
}
bool h2_shape(const pa_mlp* ac, const pa_mlp* c1, const pa_mlp* c2) {
  for (const pa_mlp* h : {ac, c1, c2})
    if (h->d.dims[1] != 256 || h->d.dims[2] != 256 || (h->woff[1] & 3) != 0) return false;
  return true;
}

// hidden widths all one k-group count -> the unrolled instantiation
int static_groups(const pa_mlp* ac, const pa_mlp* c) {
  const int g = wf16_nkg(ac->d.dims[1]);
  if (g != 16) return 0;
  for (const pa_mlp* h : {ac, c})
    if (wf16_nkg(h->d.dims[1]) != g || wf16_nkg(h->d.dims[2]) != g) return 0;
  return g;
}

struct FusedScratch {
  float *d_head, *logp, *xq, *q1, *q2, *dq1, *dq2;
};
int64_t carve_fused(FusedScratch* s, float* base, int64_t B, int64_t S, int64_t A) {
  int64_t o = 0;
  auto take = [&](float** p, int64_t n) {
    if (s && base) *p = base + o;
    o += a4(n);
  };
  take(s ? &s->d_head : nullptr, B * 2 * A);
  take(s ? &s->logp : nullptr, B);
  take(s ? &s->xq : nullptr, B * (S + A));
  take(s ? &s->q1 : nullptr, B);
  take(s ? &s->q2 : nullptr, B);
  take(s ? &s->dq1 : nullptr, B);
  take(s ? &s->dq2 : nullptr, B);
  return o;
}

// per-tile partial loss sums of the two row kernels: one buffer per stream (stream_scratch,
// common.hpp), grown on demand — two learners on two streams do not share them
int tickets(int tiles, SacTicket* ta, SacTicket* tb, hipStream_t s) {
  const size_t half = (size_t)2 * tiles;
  float* buf = stream_scratch(SCR_SAC_TICKETS, s, 2 * half);
  if (!buf) return PA_ERR_NOMEM;
  ta->partials = buf;
  tb->partials = buf + half;
  return PA_OK;
}

// The split launch's exchange words (sac_rows.hpp, SacRowsAArgs::split): one buffer set PER LEARNER
// (owned by its actor network, released with it — two learners on two streams never share words),
// pre-filled with the pending tag; readers restore the tag, so a step leaves it as it found it.
struct Exchange {
  float* xact = nullptr;
  float* xres = nullptr;
  int* err = nullptr;
  int* err_host = nullptr;
  int cap = 0;
  // the actor's forward on the NEXT step's states, run by this step's sac_rows_b launch
  // (SacRowsBArgs::pre_state): valid for exactly the step that follows inside one pa_sac_learn call
  float* pre_head = nullptr;
  int64_t pre_cap = 0;
  bool pre_valid = false;
  const float* pre_state = nullptr;
  int pre_ld = 0, pre_B = 0;
};
// set by pa_sac_learn around a step that is followed by another one of the same call: the states the
// next step will read (same batch size)
struct NextHint {
  const float* state = nullptr;
  int ld = 0;
};
thread_local NextHint g_next_hint;
bool pre_enabled() {
  const char* v = getenv("PEARL_AMD_SAC_PRE");     // read per call: tests compare the forms
  return !(v && v[0] == '0');
}
void exchange_free(void* p) {
  Exchange* x = static_cast<Exchange*>(p);
  if (!x) return;
  if (x->xact) (void)hipFree(x->xact);
  if (x->pre_head) (void)hipFree(x->pre_head);
  if (x->err) (void)hipFree(x->err);
  if (x->err_host) (void)hipHostFree(x->err_host);
  delete x;
}
int exchange(pa_mlp* owner, int tiles, hipStream_t s, Exchange** out) {
  Exchange* x = static_cast<Exchange*>(owner->aux);
  if (!x) {
    x = new (std::nothrow) Exchange();
    PA_REQUIRE(x, PA_ERR_NOMEM, "out of host memory");
    owner->aux = x;
    owner->aux_free = exchange_free;
  }
  if (!x->err) {
    PA_HIP(hipMalloc((void**)&x->err, 16));
    PA_HIP(hipMemset(x->err, 0, 16));
    PA_HIP(hipDeviceSynchronize());
    PA_HIP(hipHostMalloc((void**)&x->err_host, 16, hipHostMallocDefault));
    x->err_host[0] = 0;
  }
  if (tiles > x->cap) {
    if (x->xact) {
      PA_HIP(hipDeviceSynchronize());
      (void)hipFree(x->xact);
      x->xact = nullptr;
    }
    const int n = tiles * 2;
    const size_t words = (size_t)n * (SR_XACT + SR_XRES);
    PA_HIP(hipMalloc((void**)&x->xact, words * sizeof(float)));
    PA_HIP(hipMemsetD32Async((hipDeviceptr_t)x->xact, (int)kYPendingBits, words, s));
    x->xres = x->xact + (size_t)n * SR_XACT;
    x->cap = n;
  }
  *out = x;
  return PA_OK;
}
// Workgroups of a split launch wait for each other, so ALL of them must be resident at once: the
// bound is what THIS device (a partition, a CU-masked queue's parent) offers — its compute-unit
// count, one row workgroup per CU (their LDS and 8 waves x >128 VGPRs admit no second) — not a
// constant.  PEARL_AMD_SPLIT_MAX_WGS lowers it for processes that share the GPU with other work.
int resident_row_wgs(int device) {
  static int cached_dev = -1, cached = 0;
  if (cached_dev != device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    cached = prop.multiProcessorCount;
    const char* v = getenv("PEARL_AMD_SPLIT_MAX_WGS");
    if (v && *v && atoi(v) < cached) cached = atoi(v);
    cached_dev = device;
  }
  return cached;
}
bool split_enabled() {
  const char* v = getenv("PEARL_AMD_SAC_SPLIT");     // read per call: tests compare the forms
  return !(v && v[0] == '0');
}

template <int NGH, int NGA, int NGC, int HEAD = 0>
int launch_rows(const SacRowsAArgs* ra, const SacRowsBArgs* rb, int W, hipStream_t s, bool h2 = false) {
  if constexpr (NGH == 16 && HEAD == 0) {
    if (h2) {   // split launches only (fused_step)
      static size_t configured_h2 = 0;
      const size_t smem2 = sac_rows_h2_smem_bytes(W);
      if (smem2 > configured_h2) {
        int rc = set_max_smem(sac_rows_a_kernel<NGH, NGA, NGC, 0, true, true>, smem2);
        if (rc != PA_OK) return rc;
        rc = set_max_smem(sac_rows_b_kernel<NGH, NGA, NGC, 0, true, true>, smem2);
        if (rc != PA_OK) return rc;
        configured_h2 = smem2;
      }
      if (ra) {
        const unsigned tiles = (unsigned)ceil_div(ra->B, RP_ROWS);
        hipLaunchKernelGGL((sac_rows_a_kernel<NGH, NGA, NGC, 0, true, true>), dim3(4 * tiles), dim3(512),
                           smem2, s, *ra);
      } else {
        const unsigned tiles = (unsigned)ceil_div(rb->B, RP_ROWS);
        hipLaunchKernelGGL((sac_rows_b_kernel<NGH, NGA, NGC, 0, true, true>),
                           dim3((rb->pre_state ? 3 : 2) * tiles), dim3(512), smem2, s, *rb);
      }
      PA_LAUNCH_CHECK();
      return PA_OK;
    }
  }
  static size_t configured = 0;
  const size_t smem = sac_rows_smem_floats(W) * sizeof(float);
  if (smem > configured) {
    int rc = set_max_smem(sac_rows_a_kernel<NGH, NGA, NGC, HEAD>, smem);
    if (rc != PA_OK) return rc;
    rc = set_max_smem(sac_rows_b_kernel<NGH, NGA, NGC, HEAD>, smem);
    if (rc != PA_OK) return rc;
    configured = smem;
  }
  if (ra) {
    const unsigned tiles = (unsigned)ceil_div(ra->B, RP_ROWS);
    if constexpr (HEAD == 0) {
      if (ra->split) {
        static size_t configured_split = 0;
        if (smem > configured_split) {
          const int rc = set_max_smem(sac_rows_a_kernel<NGH, NGA, NGC, 0, true>, smem);
          if (rc != PA_OK) return rc;
          configured_split = smem;
        }
        hipLaunchKernelGGL((sac_rows_a_kernel<NGH, NGA, NGC, 0, true>), dim3(4 * tiles), dim3(512),
                           smem, s, *ra);
        PA_LAUNCH_CHECK();
        return PA_OK;
      }
    }
    hipLaunchKernelGGL((sac_rows_a_kernel<NGH, NGA, NGC, HEAD>), dim3(tiles, ra->actor_rows ? 3 : 2),
                       dim3(512), smem, s, *ra);
  } else {
    const unsigned tiles = (unsigned)ceil_div(rb->B, RP_ROWS);
    if (rb->xact) {
      static size_t configured_split_b = 0;
      if (smem > configured_split_b) {
        const int rc = set_max_smem(sac_rows_b_kernel<NGH, NGA, NGC, HEAD, true>, smem);
        if (rc != PA_OK) return rc;
        configured_split_b = smem;
      }
      hipLaunchKernelGGL((sac_rows_b_kernel<NGH, NGA, NGC, HEAD, true>),
                         dim3((rb->pre_state ? 3 : 2) * tiles), dim3(512), smem, s, *rb);
    } else {
      hipLaunchKernelGGL((sac_rows_b_kernel<NGH, NGA, NGC, HEAD>), dim3(tiles), dim3(512), smem, s, *rb);
    }
  }
  PA_LAUNCH_CHECK();
  return PA_OK;
}

AdamScalars scalar_adam(double lr, double b1, double b2, double eps, double wd, int amsgrad,
                        int64_t step) {
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  AdamScalars c;
  c.decay = (float)(1.0 - lr * wd);
  c.w1 = (float)(1.0 - b1);
  c.beta2 = (float)b2;
  c.omb2 = (float)(1.0 - b2);
  c.bc2_sqrt = (float)sqrt(bc2);
  c.neg_step = (float)(-(lr / bc1));
  c.eps = (float)eps;
  c.amsgrad = amsgrad;
  return c;
}

long long* g_prof_a = nullptr;
long long* g_prof_b = nullptr;

// Live kernel timing for the bench line (bench_algos.py): HIP events on the learner's stream around
// the two row launches of the first kTimedSteps steps after pa_sac_timing(1).
constexpr int kTimedSteps = 64;
struct RowTimers {
  bool on = false;
  int n = 0;
  hipEvent_t ev[kTimedSteps][4];
  bool made = false;
} g_tm;

int fused_step(const pa_sac_step_args* a, hipStream_t s) {
  pa_mlp *ac = a->actor, *c1 = a->critic1, *c2 = a->critic2;
  const int B = a->B, S = a->S, A = a->A, W = S + A;
  PA_HIP(hipSetDevice(ac->d.device));
  FusedScratch w;
  memset(&w, 0, sizeof(w));
  carve_fused(&w, a->scratch, B, S, A);
  const int tiles = (int)ceil_div(B, RP_ROWS);
  SacTicket ta, tb;
  PA_TRY(tickets(tiles, &ta, &tb, s));
  PA_TRY(mlp_ensure_packed(ac, false, s));
  PA_TRY(mlp_ensure_packed(c1, false, s));
  PA_TRY(mlp_ensure_packed(c2, false, s));
  PA_TRY(mlp_ensure_packed(c1, true, s));
  PA_TRY(mlp_ensure_packed(c2, true, s));
  const int ngh = static_groups(ac, c1);
  // the fp16x2 kernels run with the split launches (decided below, same conditions): their row
  // maxima are made current here and kept so by this step's optimizer launches
  const bool want_split = split_enabled() && 4 * tiles <= resident_row_wgs(ac->d.device) && A <= 16;
  const bool h2 = want_split && ngh == 16 && h2_enabled() && h2_shape(ac, c1, c2);
  if (h2) {
    PA_TRY(mlp_ensure_um(ac, false, s));
    PA_TRY(mlp_ensure_um(c1, false, s));
    PA_TRY(mlp_ensure_um(c2, false, s));
    PA_TRY(mlp_ensure_um(c1, true, s));
    PA_TRY(mlp_ensure_um(c2, true, s));
  } else {
    // nobody reads them: the optimizer launches need not keep them
    ac->um_ok[0] = c1->um_ok[0] = c2->um_ok[0] = c1->um_ok[1] = c2->um_ok[1] = false;
  }
  // ---------------------------------------------------------------- rows A
  SacRowsAArgs ra;
  memset(&ra, 0, sizeof(ra));
  fill_net(ac, false, ra.actor);
  fill_net(c1, false, ra.critic[0]);
  fill_net(c2, false, ra.critic[1]);
  ra.state = a->state; ra.ld_state = a->ld_state;
  ra.action = a->action; ra.ld_action = a->ld_action;
  ra.noise = a->noise_actor; ra.ld_noise = A;
  ra.low = a->low; ra.high = a->high; ra.alpha = a->alpha;
  ra.B = B; ra.S = S; ra.A = A;
  float* logp = a->log_prob_out ? a->log_prob_out : w.logp;   // no copy afterwards
  ra.d_head = w.d_head; ra.logp = logp; ra.xq = w.xq;
  ra.q[0] = w.q1; ra.q[1] = w.q2;
  ra.tk = ta;
  ra.actor_rows = 1;
  ra.prof = g_prof_a;
  // the second critic of the actor loss in a helper workgroup, while every workgroup of the launch
  // is resident at once (a waiting workgroup must never keep its partner off the chip)
  Exchange* xch = nullptr;
  ac->adam_guard = c1->adam_guard = c2->adam_guard = nullptr;
  if (want_split) {
    PA_TRY(exchange(ac, tiles, s, &xch));
    PA_REQUIRE(xch->err_host[0] == 0, PA_ERR_HIP,
               "an earlier SAC step's workgroup hand-off expired (code %d): the parameters were "
               "left untouched by that step; re-create the learner", xch->err_host[0]);
    ra.split = 1;
    ra.xact = xch->xact; ra.xres = xch->xres;
    ra.err = xch->err; ra.err_host = xch->err_host;
    // an expired hand-off leaves pending tags (NaNs) in the gradients: the optimizer launches of
    // this step then skip AdamW and the soft update (AdamFuse::guard)
    ac->adam_guard = c1->adam_guard = c2->adam_guard = xch->err;
    // the previous step of this learn loop ran the actor's forward on exactly these states
    if (xch->pre_valid && xch->pre_state == a->state && xch->pre_ld == a->ld_state && xch->pre_B == B) {
      ra.pre_head = xch->pre_head;
      ra.pre_mask = reinterpret_cast<const unsigned*>(xch->pre_head + a4((int64_t)B * 2 * A));
    }
    xch->pre_valid = false;
  }
  // instantiations: every loop unrolled (hidden 256, S = 49..64, S + A = 65..80: the benchmark
  // shape), hidden layers unrolled only, all run-time
  const int form = ngh != 16 ? 0 : (wf16_nkg(S) == 4 && wf16_nkg(W) == 5 ? 2 : 1);
  const bool timed = g_tm.on && g_tm.n < kTimedSteps;
  if (timed) PA_HIP(hipEventRecord(g_tm.ev[g_tm.n][0], s));
  PA_TRY((form == 2   ? launch_rows<16, 4, 5>(&ra, nullptr, W, s, h2)
          : form == 1 ? launch_rows<16, 0, 0>(&ra, nullptr, W, s, h2)
                      : launch_rows<0, 0, 0>(&ra, nullptr, W, s)));
  if (timed) PA_HIP(hipEventRecord(g_tm.ev[g_tm.n][1], s));
  // ---------------------------------------------------------------- actor: dW + AdamW
  {
    const float* dzs[3] = {ac->dz[1], ac->dz[2], w.d_head};
    const int ldzs[3] = {ac->d.dims[1], ac->d.dims[2], 2 * A};
    mlp_set_pending(ac, a->state, a->ld_state, B, dzs, ldzs);
    PA_TRY(pa_mlp_adam(ac, a->actor_step, s));
  }
  // ---------------------------------------------------------------- rows B
  SacRowsBArgs rb;
  memset(&rb, 0, sizeof(rb));
  fill_net(ac, false, rb.actor);
  fill_net(c1, true, rb.target[0]);
  fill_net(c2, true, rb.target[1]);
  rb.dz1[0] = c1->dz[1]; rb.dz2[0] = c1->dz[2];
  rb.dz1[1] = c2->dz[1]; rb.dz2[1] = c2->dz[2];
  rb.H1c = c1->d.dims[1]; rb.H2c = c1->d.dims[2];
  rb.next_state = a->next_state; rb.ld_next = a->ld_next_state;
  rb.noise = a->noise_critic; rb.ld_noise = A;
  rb.low = a->low; rb.high = a->high; rb.alpha_in = a->alpha;
  rb.reward = a->reward; rb.term = a->terminated; rb.gamma = a->gamma;
  rb.q[0] = w.q1; rb.q[1] = w.q2;
  rb.dq[0] = w.dq1; rb.dq[1] = w.dq2;
  rb.B = B; rb.S = S; rb.A = A;
  rb.tk = tb;
  rb.prof = g_prof_b;
  bool pre_launched = false;
  if (ra.split) {
    rb.xact = xch->xact; rb.xres = xch->xres;
    rb.err = xch->err; rb.err_host = xch->err_host;
    // a step follows inside this learn call: its actor forward rides this launch's idle workgroups
    if (g_next_hint.state && pre_enabled() && 3 * tiles <= resident_row_wgs(ac->d.device)) {
      const int64_t head_floats = a4((int64_t)B * 2 * A);
      const int64_t need = head_floats + (int64_t)tiles * 512 * 2;   // + the ReLU masks, lane by lane
      if (need > xch->pre_cap) {
        if (xch->pre_head) {
          PA_HIP(hipDeviceSynchronize());
          (void)hipFree(xch->pre_head);
          xch->pre_head = nullptr;
          xch->pre_cap = 0;
        }
        PA_HIP(hipMalloc((void**)&xch->pre_head, (size_t)need * sizeof(float)));
        xch->pre_cap = need;
      }
      rb.pre_state = g_next_hint.state; rb.ld_pre = g_next_hint.ld;
      rb.pre_head = xch->pre_head;
      rb.pre_mask = reinterpret_cast<unsigned*>(xch->pre_head + head_floats);
      pre_launched = true;
    }
  }
  if (timed) PA_HIP(hipEventRecord(g_tm.ev[g_tm.n][2], s));
  PA_TRY((form == 2   ? launch_rows<16, 4, 5>(nullptr, &rb, W, s, h2)
          : form == 1 ? launch_rows<16, 0, 0>(nullptr, &rb, W, s, h2)
                      : launch_rows<0, 0, 0>(nullptr, &rb, W, s)));
  if (timed) {
    PA_HIP(hipEventRecord(g_tm.ev[g_tm.n][3], s));
    ++g_tm.n;
  }
  if (pre_launched) {
    xch->pre_valid = true;
    xch->pre_state = rb.pre_state; xch->pre_ld = rb.ld_pre; xch->pre_B = B;
  }
  // ---------------------------------------------------------------- critics: dW + AdamW, targets
  pa_mlp* cs[2] = {c1, c2};
  float* dqs[2] = {w.dq1, w.dq2};
  for (int i = 0; i < 2; ++i) {
    const float* dzs[3] = {cs[i]->dz[1], cs[i]->dz[2], dqs[i]};
    const int ldzs[3] = {cs[i]->d.dims[1], cs[i]->d.dims[2], 1};
    mlp_set_pending(cs[i], w.xq, W, B, dzs, ldzs);
  }
  // ---------------------------------------------------------------- losses, entropy coefficient
  TailJob tj;
  memset(&tj, 0, sizeof(tj));
  tj.kind = 1;
  tj.part_a = ta.partials; tj.part_b = tb.partials; tj.tiles = tiles; tj.B = B;
  tj.actor_loss = a->losses + 0; tj.critic_loss = a->losses + 1;
  if (a->log_alpha) {
    tj.log_alpha = a->log_alpha; tj.am = a->alpha_m; tj.av = a->alpha_v; tj.avmax = a->alpha_vmax;
    tj.alpha = a->alpha;
    tj.logp = logp; tj.target_entropy = a->target_entropy;
    tj.ac = scalar_adam(a->alpha_lr, a->alpha_beta1, a->alpha_beta2, a->alpha_eps,
                        a->alpha_weight_decay, a->alpha_amsgrad, a->alpha_step);
    tj.alpha_loss_out = a->losses + 2;
  }
  if (mlp_pair_fusable(c1, c2, true)) {
    // both critics' weight gradients, AdamW, soft target updates AND the step's scalar tail (one
    // extra workgroup): one launch
    const int rc_pair = mlp_adam_pair(c1, c2, a->critic_step, a->tau, s, &tj);
    ac->adam_guard = c1->adam_guard = c2->adam_guard = nullptr;   // (see the end of this function)
    return rc_pair;
  }
  PA_TRY(pa_mlp_adam(c1, a->critic_step, s));
  PA_TRY(pa_mlp_adam(c2, a->critic_step, s));
  PA_TRY(pa_mlp_soft_update(c1, a->tau, s));
  PA_TRY(pa_mlp_soft_update(c2, a->tau, s));
  SacFinishArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.part_a = tj.part_a; fa.part_b = tj.part_b; fa.tiles = tiles; fa.B = B;
  fa.actor_loss = tj.actor_loss; fa.critic_loss = tj.critic_loss;
  fa.log_alpha = tj.log_alpha; fa.am = tj.am; fa.av = tj.av; fa.avmax = tj.avmax; fa.alpha = tj.alpha;
  fa.logp = tj.logp; fa.target_entropy = tj.target_entropy; fa.ac = tj.ac;
  fa.alpha_loss_out = tj.alpha_loss_out;
  hipLaunchKernelGGL(sac_finish_kernel, dim3(1), dim3(256), 0, s, fa);
  PA_LAUNCH_CHECK();
  // the launches above hold the guard by value; the networks must not keep a pointer into
  // another network's exchange (they may be destroyed in any order)
  ac->adam_guard = c1->adam_guard = c2->adam_guard = nullptr;
  return PA_OK;
}

}  // namespace

int64_t carve_fused_size(int64_t B, int64_t S, int64_t A) { return carve_fused(nullptr, nullptr, B, S, A); }

extern "C" int pa_sac_step(const pa_sac_step_args* a, void* stream) {
  PA_REQUIRE(a && a->actor && a->critic1 && a->critic2 && a->state && a->action && a->reward &&
                 a->terminated && a->next_state && a->noise_actor && a->noise_critic && a->low &&
                 a->high && a->alpha && a->scratch && a->losses && a->B > 0 && a->S > 0 && a->A > 0,
             PA_ERR_INVALID, "pa_sac_step: bad argument");
  if (fused_ok(a)) return fused_step(a, reinterpret_cast<hipStream_t>(stream));
  const int B = a->B, S = a->S, A = a->A, W = S + A;
  SacScratch w;
  memset(&w, 0, sizeof(w));
  carve(&w, a->scratch, B, S, A);
  // ---------------------------------------------------------------- actor update (:208-231)
  // xa = [state | sampled action]: the state columns now, the action columns by the sampling head
  PA_HIP(hipMemcpy2DAsync(w.xa, (size_t)W * 4, a->state, (size_t)a->ld_state * 4, (size_t)S * 4,
                          (size_t)B, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
  PA_TRY(pa_mlp_forward(a->actor, 0, a->state, a->ld_state, B, w.head, 2 * A, 1, stream));
  PA_TRY(pa_gauss_sample(w.head, 2 * A, a->noise_actor, A, a->low, a->high, B, A, w.xa + S, W, w.logp,
                         stream));
  PA_TRY(pa_mlp_forward2(a->critic1, a->critic2, 0, w.xa, W, B, w.q1, 1, w.q2, 1, 1, stream));
  PA_TRY(pa_sac_twin(0, w.q1, w.q2, w.logp, a->alpha, nullptr, nullptr, 0.f, B, w.dq1, w.dq2,
                     a->losses + 0, stream));
  // only the critics' INPUT gradient matters here (the reference forms and discards their
  // parameter gradients, actor_critic_base.py:342-348)
  PA_TRY(pa_mlp_backward2(a->critic1, a->critic2, w.xa, W, B, w.dq1, 1, w.dq2, 1, 0, w.dx1, w.dx2, W,
                          stream));
  PA_TRY(pa_gauss_actor_grad(w.head, 2 * A, a->noise_actor, A, a->low, a->high, w.dx1 + S, w.dx2 + S,
                             W, a->alpha, B, A, w.d_head, 2 * A, stream));
  PA_TRY(pa_mlp_backward(a->actor, a->state, a->ld_state, B, w.d_head, 2 * A, 2, nullptr, 0, stream));
  PA_TRY(pa_mlp_adam(a->actor, a->actor_step, stream));
  // ---------------------------------------------------------------- critic update (:155-206)
  PA_HIP(hipMemcpy2DAsync(w.xn, (size_t)W * 4, a->next_state, (size_t)a->ld_next_state * 4,
                          (size_t)S * 4, (size_t)B, hipMemcpyDeviceToDevice,
                          reinterpret_cast<hipStream_t>(stream)));
  PA_TRY(pa_mlp_forward(a->actor, 0, a->next_state, a->ld_next_state, B, w.head_n, 2 * A, 0, stream));
  PA_TRY(pa_gauss_sample(w.head_n, 2 * A, a->noise_critic, A, a->low, a->high, B, A, w.xn + S, W,
                         w.nlogp, stream));
  PA_TRY(pa_mlp_forward2(a->critic1, a->critic2, 1, w.xn, W, B, w.nq1, 1, w.nq2, 1, 0, stream));
  PA_TRY(pa_sac_twin(1, w.nq1, w.nq2, w.nlogp, a->alpha, a->reward, a->terminated, a->gamma, B, w.y,
                     nullptr, nullptr, stream));
  PA_TRY(pa_concat_cols(a->state, a->ld_state, a->action, a->ld_action, w.xq, B, S, A, stream));
  PA_TRY(pa_mlp_forward2(a->critic1, a->critic2, 0, w.xq, W, B, w.qa, 1, w.qb, 1, 1, stream));
  PA_TRY(pa_mse_head(w.qa, 1, w.y, B, 1.0f / (float)B, 0.5f, 0, w.dqa, a->losses + 1, stream));
  PA_TRY(pa_mse_head(w.qb, 1, w.y, B, 1.0f / (float)B, 0.5f, 1, w.dqb, a->losses + 1, stream));
  PA_TRY(pa_mlp_backward2(a->critic1, a->critic2, w.xq, W, B, w.dqa, 1, w.dqb, 1, 2, nullptr, nullptr,
                          0, stream));
  // ---------------------------------------------------------------- AdamW, targets
  if (mlp_pair_fusable(a->critic1, a->critic2, true)) {
    PA_TRY(pa_mlp_adam2(a->critic1, a->critic2, a->critic_step, a->tau, stream));
  } else {
    PA_TRY(pa_mlp_adam(a->critic1, a->critic_step, stream));
    PA_TRY(pa_mlp_adam(a->critic2, a->critic_step, stream));
    PA_TRY(pa_mlp_soft_update(a->critic1, a->tau, stream));
    PA_TRY(pa_mlp_soft_update(a->critic2, a->tau, stream));
  }
  // ---------------------------------------------------------------- entropy coefficient
  if (a->log_alpha) {
    PA_TRY(pa_sac_alpha_step(a->log_alpha, a->alpha_m, a->alpha_v, a->alpha_vmax, a->alpha, w.logp, B,
                             a->target_entropy, a->alpha_lr, a->alpha_beta1, a->alpha_beta2,
                             a->alpha_eps, a->alpha_weight_decay, a->alpha_amsgrad, a->alpha_step,
                             a->losses + 2, stream));
  }
  if (a->log_prob_out)
    PA_HIP(hipMemcpyAsync(a->log_prob_out, w.logp, (size_t)B * 4, hipMemcpyDeviceToDevice,
                          reinterpret_cast<hipStream_t>(stream)));
  return PA_OK;
}

// tools/prof_sac.py: device buffers for the row kernels' phase stamps
// ([3 * tiles][8 waves][32] and [tiles][8][32] long long; null = off)
extern "C" int pa_debug_sac_prof(long long* rows_a, long long* rows_b) {
  g_prof_a = rows_a;
  g_prof_b = rows_b;
  return PA_OK;
}

// HIP-event timing of the two fused row launches (bench_algos.py's roofline object)
extern "C" int pa_sac_timing(int32_t enable) {
  if (enable && !g_tm.made) {
    for (int i = 0; i < kTimedSteps; ++i)
      for (int k = 0; k < 4; ++k) PA_HIP(hipEventCreate(&g_tm.ev[i][k]));
    g_tm.made = true;
  }
  g_tm.on = enable != 0;
  g_tm.n = 0;
  return PA_OK;
}
// average launch durations (us) over the steps timed so far; synchronises with their events
extern "C" int pa_sac_timing_read(double* rows_a_us, double* rows_b_us, int64_t* steps) {
  PA_REQUIRE(rows_a_us && rows_b_us && steps, PA_ERR_INVALID, "null output");
  double sa = 0.0, sb = 0.0;
  for (int i = 0; i < g_tm.n; ++i) {
    float ms = 0.f;
    PA_HIP(hipEventSynchronize(g_tm.ev[i][3]));
    PA_HIP(hipEventElapsedTime(&ms, g_tm.ev[i][0], g_tm.ev[i][1]));
    sa += ms;
    PA_HIP(hipEventElapsedTime(&ms, g_tm.ev[i][2], g_tm.ev[i][3]));
    sb += ms;
  }
  *steps = g_tm.n;
  *rows_a_us = g_tm.n ? 1e3 * sa / g_tm.n : 0.0;
  *rows_b_us = g_tm.n ? 1e3 * sb / g_tm.n : 0.0;
  return PA_OK;
}

// =============================================================================================
// DDPG / TD3: one learn_batch as one call
// (pearl/policy_learners/sequential_decision_making/ddpg.py:106-147, td3.py:105-175 on
// actor_critic_base.py:309-366).  The same launches the per-stage Python path issues, in its order
// (bit-identical results): that path needed ~185 us of host time per step for ~160 us of kernels.
// =============================================================================================
namespace {
struct DdpgScratch {
  float *xa, *head, *q1, *dq, *dx, *d_head, *xn, *head_n, *nq1, *nq2, *y, *xq, *qa, *qb, *dqa, *dqb;
};
int64_t carve_ddpg(DdpgScratch* s, float* base, int64_t B, int64_t S, int64_t A) {
  int64_t o = 0;
  auto take = [&](float** p, int64_t n) {
    if (s && base) *p = base + o;
    o += a4(n);
  };
  take(s ? &s->xa : nullptr, B * (S + A));
  take(s ? &s->head : nullptr, B * A);
  take(s ? &s->q1 : nullptr, B);
  take(s ? &s->dq : nullptr, B);
  take(s ? &s->dx : nullptr, B * (S + A));
  take(s ? &s->d_head : nullptr, B * A);
  take(s ? &s->xn : nullptr, B * (S + A));
  take(s ? &s->head_n : nullptr, B * A);
  take(s ? &s->nq1 : nullptr, B);
  take(s ? &s->nq2 : nullptr, B);
  take(s ? &s->y : nullptr, B);
  take(s ? &s->xq : nullptr, B * (S + A));
  take(s ? &s->qa : nullptr, B);
  take(s ? &s->qb : nullptr, B);
  take(s ? &s->dqa : nullptr, B);
  take(s ? &s->dqb : nullptr, B);
  return o;
}
}  // namespace

namespace {
// ---- the fused form: sac_rows.hpp with HEAD = 1 -------------------------------------------------
bool ddpg_fused_ok(const pa_ddpg_step_args* a) {
  const char* v = getenv("PEARL_AMD_DDPG_FUSED");     // read per call: tests compare the forms
  if (v && *v == '0') return false;
  const pa_mlp *ac = a->actor, *c1 = a->critic1, *c2 = a->critic2;
  if (!relu3(ac) || !relu3(c1) || !relu3(c2)) return false;
  if (!ac->bufs.p_target || !c1->bufs.p_target || !c2->bufs.p_target) return false;
  if (a->A < 1 || a->A > 16 || a->S + a->A > ROW_MAX_IN) return false;
  if (ac->d.dims[0] != a->S || ac->d.dims[3] != a->A) return false;
  for (const pa_mlp* c : {c1, c2}) {
    if (c->d.dims[0] != a->S + a->A || c->d.dims[3] != 1) return false;
    if (c->d.dims[1] != c1->d.dims[1] || c->d.dims[2] != c1->d.dims[2]) return false;
  }
  return a->B <= ac->d.max_batch && a->B <= c1->d.max_batch && a->B <= c2->d.max_batch;
}

struct DdpgFusedScratch {
  float *d_head, *xq, *q1, *q2, *dq1, *dq2, *logp_unused;
};
int64_t carve_ddpg_fused(DdpgFusedScratch* s, float* base, int64_t B, int64_t S, int64_t A) {
  int64_t o = 0;
  auto take = [&](float** p, int64_t n) {
    if (s && base) *p = base + o;
    o += a4(n);
  };
  take(s ? &s->d_head : nullptr, B * A);
  take(s ? &s->xq : nullptr, B * (S + A));
  take(s ? &s->q1 : nullptr, B);
  take(s ? &s->q2 : nullptr, B);
  take(s ? &s->dq1 : nullptr, B);
  take(s ? &s->dq2 : nullptr, B);
  take(s ? &s->logp_unused : nullptr, B);
  return o;
}

template <int NGH, int NGA, int NGC>
int launch_rows_ddpg(const SacRowsAArgs* ra, const SacRowsBArgs* rb, int W, hipStream_t s) {
  return launch_rows<NGH, NGA, NGC, 1>(ra, rb, W, s);
}

int ddpg_fused_step(const pa_ddpg_step_args* a, hipStream_t s) {
  pa_mlp *ac = a->actor, *c1 = a->critic1, *c2 = a->critic2;
  const int B = a->B, S = a->S, A = a->A, W = S + A;
  PA_HIP(hipSetDevice(ac->d.device));
  DdpgFusedScratch w;
  memset(&w, 0, sizeof(w));
  carve_ddpg_fused(&w, a->scratch, B, S, A);
  const int tiles = (int)ceil_div(B, RP_ROWS);
  SacTicket ta, tb;
  PA_TRY(tickets(tiles, &ta, &tb, s));
  PA_TRY(mlp_ensure_packed(ac, false, s));
  PA_TRY(mlp_ensure_packed(ac, true, s));
  PA_TRY(mlp_ensure_packed(c1, false, s));
  PA_TRY(mlp_ensure_packed(c2, false, s));
  PA_TRY(mlp_ensure_packed(c1, true, s));
  PA_TRY(mlp_ensure_packed(c2, true, s));
  const int ngh = static_groups(ac, c1);
  const int form = ngh != 16 ? 0 : (wf16_nkg(S) == 4 && wf16_nkg(W) == 5 ? 2 : 1);
  // ---------------------------------------------------------------- rows A
  SacRowsAArgs ra;
  memset(&ra, 0, sizeof(ra));
  fill_net(ac, false, ra.actor);
  fill_net(c1, false, ra.critic[0]);
  fill_net(c2, false, ra.critic[1]);
  ra.state = a->state; ra.ld_state = a->ld_state;
  ra.action = a->action; ra.ld_action = a->ld_action;
  ra.low = a->low; ra.high = a->high;
  ra.B = B; ra.S = S; ra.A = A;
  ra.d_head = w.d_head; ra.logp = w.logp_unused; ra.xq = w.xq;
  ra.q[0] = w.q1; ra.q[1] = w.q2;
  ra.tk = ta;
  ra.actor_rows = a->do_actor ? 1 : 0;
  PA_TRY((form == 2   ? launch_rows_ddpg<16, 4, 5>(&ra, nullptr, W, s)
          : form == 1 ? launch_rows_ddpg<16, 0, 0>(&ra, nullptr, W, s)
                      : launch_rows_ddpg<0, 0, 0>(&ra, nullptr, W, s)));
  // ---------------------------------------------------------------- actor: dW + AdamW
  if (a->do_actor) {
    const float* dzs[3] = {ac->dz[1], ac->dz[2], w.d_head};
    const int ldzs[3] = {ac->d.dims[1], ac->d.dims[2], A};
    mlp_set_pending(ac, a->state, a->ld_state, B, dzs, ldzs);
    PA_TRY(pa_mlp_adam(ac, a->actor_step, s));
  }
  // ---------------------------------------------------------------- rows B (TARGET policy)
  SacRowsBArgs rb;
  memset(&rb, 0, sizeof(rb));
  fill_net(ac, true, rb.actor);
  fill_net(c1, true, rb.target[0]);
  fill_net(c2, true, rb.target[1]);
  rb.dz1[0] = c1->dz[1]; rb.dz2[0] = c1->dz[2];
  rb.dz1[1] = c2->dz[1]; rb.dz2[1] = c2->dz[2];
  rb.H1c = c1->d.dims[1]; rb.H2c = c1->d.dims[2];
  rb.next_state = a->next_state; rb.ld_next = a->ld_next_state;
  rb.noise = a->target_noise; rb.ld_noise = A; rb.noise_clip = a->noise_clip;
  rb.low = a->low; rb.high = a->high;
  rb.reward = a->reward; rb.term = a->terminated; rb.gamma = a->gamma;
  rb.q[0] = w.q1; rb.q[1] = w.q2;
  rb.dq[0] = w.dq1; rb.dq[1] = w.dq2;
  rb.B = B; rb.S = S; rb.A = A;
  rb.tk = tb;
  // the second target critic in a helper workgroup (every workgroup of the launch resident at once)
  ac->adam_guard = c1->adam_guard = c2->adam_guard = nullptr;
  if (split_enabled() && 2 * tiles <= resident_row_wgs(ac->d.device) && A <= 16) {
    Exchange* xch = nullptr;
    PA_TRY(exchange(ac, tiles, s, &xch));
    PA_REQUIRE(xch->err_host[0] == 0, PA_ERR_HIP,
               "an earlier step's workgroup hand-off expired (code %d): the parameters were left "
               "untouched by that step; re-create the learner", xch->err_host[0]);
    rb.xact = xch->xact; rb.xres = xch->xres;
    rb.err = xch->err; rb.err_host = xch->err_host;
    c1->adam_guard = c2->adam_guard = xch->err;
  }
  PA_TRY((form == 2   ? launch_rows_ddpg<16, 4, 5>(nullptr, &rb, W, s)
          : form == 1 ? launch_rows_ddpg<16, 0, 0>(nullptr, &rb, W, s)
                      : launch_rows_ddpg<0, 0, 0>(nullptr, &rb, W, s)));
  // ---------------------------------------------------------------- critics: dW + AdamW, targets
  pa_mlp* cs[2] = {c1, c2};
  float* dqs[2] = {w.dq1, w.dq2};
  for (int i = 0; i < 2; ++i) {
    const float* dzs[3] = {cs[i]->dz[1], cs[i]->dz[2], dqs[i]};
    const int ldzs[3] = {cs[i]->d.dims[1], cs[i]->d.dims[2], 1};
    mlp_set_pending(cs[i], w.xq, W, B, dzs, ldzs);
  }
  const bool soft = a->do_targets != 0;
  TailJob tj;
  memset(&tj, 0, sizeof(tj));
  tj.kind = 1;
  tj.part_a = ta.partials; tj.part_b = tb.partials; tj.tiles = tiles; tj.B = B;
  tj.actor_loss = a->do_actor ? a->losses + 0 : nullptr;
  tj.critic_loss = a->losses + 1;
  if (mlp_pair_fusable(c1, c2, soft)) {
    PA_TRY(mlp_adam_pair(c1, c2, a->critic_step, soft ? a->critic_tau : -1.f, s, &tj));
  } else {
    PA_TRY(pa_mlp_adam(c1, a->critic_step, s));
    PA_TRY(pa_mlp_adam(c2, a->critic_step, s));
    if (soft) {
      PA_TRY(pa_mlp_soft_update(c1, a->critic_tau, s));
      PA_TRY(pa_mlp_soft_update(c2, a->critic_tau, s));
    }
    SacFinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.part_a = tj.part_a; fa.part_b = tj.part_b; fa.tiles = tiles; fa.B = B;
    fa.actor_loss = tj.actor_loss; fa.critic_loss = tj.critic_loss;
    hipLaunchKernelGGL(sac_finish_kernel, dim3(1), dim3(256), 0, s, fa);
    PA_LAUNCH_CHECK();
  }
  if (soft) PA_TRY(pa_mlp_soft_update(ac, a->actor_tau, s));
  // the launches above hold the guard by value; the networks must not keep a pointer into
  // another network's exchange (they may be destroyed in any order)
  ac->adam_guard = c1->adam_guard = c2->adam_guard = nullptr;
  return PA_OK;
}
}  // namespace

// After the stream the fused steps ran on has been synchronised: did a workgroup hand-off of a
// split launch expire?  (The affected step skipped its optimizer launches' AdamW — AdamFuse::guard
// — so the parameters are intact, but its report is not.)  `actor` = the learner's actor network.
extern "C" int pa_ac_check(pa_mlp* actor) {
  PA_REQUIRE(actor, PA_ERR_INVALID, "null network");
  const Exchange* x = static_cast<const Exchange*>(actor->aux);
  PA_REQUIRE(!x || !x->err_host || x->err_host[0] == 0, PA_ERR_HIP,
             "a fused actor-critic step's workgroup hand-off expired (code %d): that step did not "
             "update the parameters and its losses are invalid (is the GPU shared with other work? "
             "PEARL_AMD_SAC_SPLIT=0 selects the unsplit kernels)",
             x->err_host[0]);
  return PA_OK;
}

extern "C" int64_t pa_ddpg_scratch_floats(int32_t B, int32_t S, int32_t A) {
  const int64_t x = carve_ddpg(nullptr, nullptr, B, S, A), y = carve_ddpg_fused(nullptr, nullptr, B, S, A);
  return x > y ? x : y;
}

extern "C" int pa_ddpg_step(const pa_ddpg_step_args* a, void* stream) {
  PA_REQUIRE(a && a->actor && a->critic1 && a->critic2 && a->state && a->action && a->reward &&
                 a->terminated && a->next_state && a->low && a->high && a->zeros && a->scratch &&
                 a->losses && a->B > 0 && a->S > 0 && a->A > 0,
             PA_ERR_INVALID, "pa_ddpg_step: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (ddpg_fused_ok(a)) return ddpg_fused_step(a, s);
  const int B = a->B, S = a->S, A = a->A, W = S + A;
  DdpgScratch w;
  memset(&w, 0, sizeof(w));
  carve_ddpg(&w, a->scratch, B, S, A);
  // ---------------------------------------------------------------- actor update (ddpg.py:106-121)
  if (a->do_actor) {
    PA_HIP(hipMemcpy2DAsync(w.xa, (size_t)W * 4, a->state, (size_t)a->ld_state * 4, (size_t)S * 4,
                            (size_t)B, hipMemcpyDeviceToDevice, s));
    PA_TRY(pa_mlp_forward(a->actor, 0, a->state, a->ld_state, B, w.head, A, 1, stream));
    PA_TRY(pa_tanh_action(w.head, A, nullptr, 0, a->low, a->high, 0.f, B, A, w.xa + S, W, stream));
    PA_TRY(pa_mlp_forward(a->critic1, 0, w.xa, W, B, w.q1, 1, 1, stream));
    PA_TRY(pa_neg_mean_head(w.q1, 1, B, w.dq, a->losses + 0, stream));
    // only critic 1's INPUT gradient matters (actor_critic_base.py:342-348)
    PA_TRY(pa_mlp_backward(a->critic1, w.xa, W, B, w.dq, 1, 0, w.dx, W, stream));
    PA_TRY(pa_tanh_action_grad(w.head, A, a->low, a->high, w.dx + S, W, B, A, w.d_head, A, stream));
    PA_TRY(pa_mlp_backward(a->actor, a->state, a->ld_state, B, w.d_head, A, 2, nullptr, 0, stream));
    PA_TRY(pa_mlp_adam(a->actor, a->actor_step, stream));
  }
  // ---------------------------------------------------------------- critic update (ddpg.py:123-147)
  PA_HIP(hipMemcpy2DAsync(w.xn, (size_t)W * 4, a->next_state, (size_t)a->ld_next_state * 4,
                          (size_t)S * 4, (size_t)B, hipMemcpyDeviceToDevice, s));
  PA_TRY(pa_mlp_forward(a->actor, 1, a->next_state, a->ld_next_state, B, w.head_n, A, 0, stream));
  PA_TRY(pa_tanh_action(w.head_n, A, a->target_noise, A, a->low, a->high, a->noise_clip, B, A,
                        w.xn + S, W, stream));
  PA_TRY(pa_mlp_forward2(a->critic1, a->critic2, 1, w.xn, W, B, w.nq1, 1, w.nq2, 1, 0, stream));
  // pa_sac_twin(mode 1) with alpha = 0, log_prob = 0:  y = min(q1', q2') gamma (1 - term) + r
  PA_TRY(pa_sac_twin(1, w.nq1, w.nq2, a->zeros, a->zeros, a->reward, a->terminated, a->gamma, B, w.y,
                     nullptr, nullptr, stream));
  PA_TRY(pa_concat_cols(a->state, a->ld_state, a->action, a->ld_action, w.xq, B, S, A, stream));
  PA_TRY(pa_mlp_forward2(a->critic1, a->critic2, 0, w.xq, W, B, w.qa, 1, w.qb, 1, 1, stream));
  PA_TRY(pa_mse_head(w.qa, 1, w.y, B, 1.0f / (float)B, 0.5f, 0, w.dqa, a->losses + 1, stream));
  PA_TRY(pa_mse_head(w.qb, 1, w.y, B, 1.0f / (float)B, 0.5f, 1, w.dqb, a->losses + 1, stream));
  PA_TRY(pa_mlp_backward2(a->critic1, a->critic2, w.xq, W, B, w.dqa, 1, w.dqb, 1, 2, nullptr, nullptr,
                          0, stream));
  const bool soft = a->do_targets != 0;
  if (mlp_pair_fusable(a->critic1, a->critic2, soft)) {
    PA_TRY(pa_mlp_adam2(a->critic1, a->critic2, a->critic_step, soft ? a->critic_tau : -1.f, stream));
  } else {
    PA_TRY(pa_mlp_adam(a->critic1, a->critic_step, stream));
    PA_TRY(pa_mlp_adam(a->critic2, a->critic_step, stream));
    if (soft) {
      PA_TRY(pa_mlp_soft_update(a->critic1, a->critic_tau, stream));
      PA_TRY(pa_mlp_soft_update(a->critic2, a->critic_tau, stream));
    }
  }
  if (soft) PA_TRY(pa_mlp_soft_update(a->actor, a->actor_tau, stream));
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// PolicyLearner.learn() for the twin-critic continuous-control learners (policy_learner.py:190-231:
// training_rounds x (sample, preprocess, learn_batch)) as ONE call: per round the arena's gather
// of the presampled index list into one batch of workspace, then the step.  Nothing here that the
// per-round calls do not do; what goes away is ~80 us of interpreter per round, which after the
// fused row kernels was more than the device needs for the step.
// ------------------------------------------------------------------------------------------------
namespace {
int loop_args_ok(const pa_ac_loop_args* lp, pa_arena* arena, int B) {
  PA_REQUIRE(lp && arena, PA_ERR_INVALID, "null loop arguments");
  PA_REQUIRE(lp->rounds >= 0 && lp->idx_lists && lp->losses && lp->losses_stride >= 2, PA_ERR_INVALID,
             "bad loop arguments");
  PA_REQUIRE(lp->batch.state && lp->batch.action && lp->batch.reward && lp->batch.terminated &&
                 lp->batch.next_state,
             PA_ERR_INVALID, "the loop's batch workspace needs state, action, reward, terminated, next_state");
  PA_REQUIRE(B > 0, PA_ERR_INVALID, "bad batch size");
  return PA_OK;
}
}  // namespace

extern "C" int pa_sac_learn(const pa_sac_step_args* step0, pa_arena* arena, const pa_ac_loop_args* lp,
                            void* stream) {
  PA_REQUIRE(step0, PA_ERR_INVALID, "null step arguments");
  PA_TRY(loop_args_ok(lp, arena, step0->B));
  PA_REQUIRE(lp->noise && lp->noise_stride >= 2ll * step0->B * step0->A, PA_ERR_INVALID,
             "pa_sac_learn: noise is [rounds][2][B][A]");
  PA_REQUIRE(step0->state == lp->batch.state && step0->next_state == lp->batch.next_state &&
                 (const void*)step0->action == lp->batch.action &&
                 (const void*)step0->reward == lp->batch.reward &&
                 step0->terminated == lp->batch.terminated,
             PA_ERR_INVALID, "pa_sac_learn: the step reads the loop's batch workspace");
  pa_sac_step_args a = *step0;
  const int64_t BA = (int64_t)a.B * a.A;
  const int G = lp->gather_rounds > 1 ? lp->gather_rounds : 1;
  for (int r = 0; r < lp->rounds; ++r) {
    // the batches of G rounds in ONE gather launch (the lists are contiguous), then G steps on
    // consecutive slices of the workspace
    const int slot = r % G;
    if (slot == 0) {
      const int n = lp->rounds - r < G ? lp->rounds - r : G;
      PA_TRY(pa_arena_gather_device(arena, lp->idx_lists + (int64_t)r * a.B, n * a.B, &lp->batch, stream));
    }
    const int64_t row0 = (int64_t)slot * a.B;
    a.state = step0->state + row0 * step0->ld_state;
    a.next_state = step0->next_state + row0 * step0->ld_next_state;
    a.action = step0->action + row0 * step0->ld_action;
    a.reward = step0->reward + row0;
    a.terminated = step0->terminated + row0;
    a.noise_actor = lp->noise + (int64_t)r * lp->noise_stride;
    a.noise_critic = a.noise_actor + BA;
    a.actor_step = step0->actor_step + r;
    a.critic_step = step0->critic_step + r;
    a.alpha_step = step0->alpha_step + r;
    a.losses = lp->losses + (int64_t)r * lp->losses_stride;
    // the next step's states, when they are already gathered (same group): its actor forward rides
    // this step's second row launch (fused_step)
    const bool next_here = r + 1 < lp->rounds && slot + 1 < G;
    g_next_hint.state = next_here ? a.state + (int64_t)a.B * step0->ld_state : nullptr;
    g_next_hint.ld = step0->ld_state;
    const int rc = pa_sac_step(&a, stream);
    g_next_hint.state = nullptr;
    if (rc != PA_OK) return rc;
  }
  return PA_OK;
}

extern "C" int pa_ddpg_learn(const pa_ddpg_step_args* step0, pa_arena* arena,
                             const pa_ac_loop_args* lp, void* stream) {
  PA_REQUIRE(step0, PA_ERR_INVALID, "null step arguments");
  PA_TRY(loop_args_ok(lp, arena, step0->B));
  PA_REQUIRE(!lp->noise || lp->noise_stride >= (int64_t)step0->B * step0->A, PA_ERR_INVALID,
             "pa_ddpg_learn: noise is [rounds][B][A]");
  PA_REQUIRE(step0->state == lp->batch.state && step0->next_state == lp->batch.next_state &&
                 (const void*)step0->action == lp->batch.action &&
                 (const void*)step0->reward == lp->batch.reward &&
                 step0->terminated == lp->batch.terminated,
             PA_ERR_INVALID, "pa_ddpg_learn: the step reads the loop's batch workspace");
  pa_ddpg_step_args a = *step0;
  int64_t actor_steps = 0;
  const int freq = lp->actor_update_freq > 1 ? lp->actor_update_freq : 1;
  const int G = lp->gather_rounds > 1 ? lp->gather_rounds : 1;
  for (int r = 0; r < lp->rounds; ++r) {
    const int slot = r % G;
    if (slot == 0) {
      const int n = lp->rounds - r < G ? lp->rounds - r : G;
      PA_TRY(pa_arena_gather_device(arena, lp->idx_lists + (int64_t)r * a.B, n * a.B, &lp->batch, stream));
    }
    const int64_t row0 = (int64_t)slot * a.B;
    a.state = step0->state + row0 * step0->ld_state;
    a.next_state = step0->next_state + row0 * step0->ld_next_state;
    a.action = step0->action + row0 * step0->ld_action;
    a.reward = step0->reward + row0;
    a.terminated = step0->terminated + row0;
    // TD3 (td3.py:106-141): the actor step and both target updates on every freq-th training step
    const bool due = ((lp->training_step0 + r + 1) % freq) == 0;
    a.do_actor = due;
    a.do_targets = due;
    a.target_noise = lp->noise ? lp->noise + (int64_t)r * lp->noise_stride : nullptr;
    a.actor_step = step0->actor_step + actor_steps;
    a.critic_step = step0->critic_step + r;
    a.losses = lp->losses + (int64_t)r * lp->losses_stride;
    PA_TRY(pa_ddpg_step(&a, stream));
    if (due) ++actor_steps;
  }
  return PA_OK;
}

// ProximalPolicyOptimization.learn's training rounds (actor_critic_base.py / policy_learner.py:190-231
// around ppo.py:152-192) sequenced in C: per group of `gather_rounds` rounds ONE gather launch writes
// state || one-hot(action) rows (the learner-side view of the gather kernel: preprocess_batch's
// action representation costs no launch) and one more gathers the three per-transition columns
// (gae, lam_return, action_probs); per round the fused row step of actor + critic
// (pa_ppo_rowstep) and ONE weight-gradient + AdamW launch for both (pa_mlp_adam2; two pa_mlp_adam
// launches when the two optimizers differ).  The same launches, on the same index lists, as the
// per-round Python loop — which spent 112 us of host time per 107 us round (tools/host_bound.py).
extern "C" int pa_ppo_learn(const pa_ppo_learn_args* g, pa_arena* arena, void* stream) {
  PA_REQUIRE(g && arena && g->actor && g->critic && g->idx_lists && g->planes && g->x &&
                 g->planes_ws && g->d_logits && g->d_value && g->losses,
             PA_ERR_INVALID, "pa_ppo_learn: null argument");
  PA_REQUIRE(g->B > 0 && g->S > 0 && g->A > 0 && g->rounds >= 0 && g->losses_stride >= 2 &&
                 g->actor_step >= 1 && g->critic_step >= 1 && g->plane_stride > 0,
             PA_ERR_INVALID, "pa_ppo_learn: bad sizes");
  PA_REQUIRE(g->actor->d.dims[0] == g->S && g->critic->d.dims[0] == g->S &&
                 g->actor->d.dims[g->actor->L] == g->A,
             PA_ERR_INVALID, "pa_ppo_learn: S / A are not the networks' input / output widths");
  PA_REQUIRE(pa_rowstep_supported(g->actor, g->critic, g->A), PA_ERR_UNSUPPORTED,
             "pa_ppo_learn: the networks are outside the fused row step's shapes");
  const int G = g->gather_rounds > 1 ? g->gather_rounds : 1;
  const int ldx = g->S + g->A;
  pa_batch_out out;
  memset(&out, 0, sizeof(out));
  out.x = g->x;
  out.rep_dim = g->A;
  out.rep_onehot = 1;
  for (int r = 0; r < g->rounds; ++r) {
    const int slot = r % G;
    if (slot == 0) {
      const int n = g->rounds - r < G ? g->rounds - r : G;
      const int64_t* idx = g->idx_lists + (int64_t)r * g->B;
      PA_TRY(pa_arena_gather_device(arena, idx, n * g->B, &out, stream));
      // (pitch of the gathered planes = the rows gathered by THIS launch)
      PA_TRY(pa_gather_planes(g->planes, g->plane_stride, 3, idx, n * g->B, g->planes_ws, stream));
    }
    const int nrows = (g->rounds - (r - slot) < G ? g->rounds - (r - slot) : G) * g->B;
    const int64_t row0 = (int64_t)slot * g->B;
    const float* x = g->x + row0 * ldx;
    const float* gae = g->planes_ws + row0;
    const float* lam_return = g->planes_ws + nrows + row0;
    const float* p_old = g->planes_ws + 2ll * nrows + row0;
    PA_TRY(pa_ppo_rowstep(g->actor, g->critic, x, ldx, g->B, x + g->S, ldx, p_old, gae, g->epsilon,
                          g->entropy_scale, lam_return, g->value_grad_scale, nullptr, 0, nullptr, 0,
                          g->d_logits, g->A, g->d_value, g->losses + (int64_t)r * g->losses_stride,
                          stream));
    if (g->actor_step == g->critic_step && mlp_pair_fusable(g->actor, g->critic, false)) {
      PA_TRY(pa_mlp_adam2(g->actor, g->critic, g->actor_step + r, -1.f, stream));
    } else {
      PA_TRY(pa_mlp_adam(g->actor, g->actor_step + r, stream));
      PA_TRY(pa_mlp_adam(g->critic, g->critic_step + r, stream));
    }
  }
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// Discrete SoftActorCritic.learn_batch as ONE call (soft_actor_critic.py:153-287 on top of
// actor_critic_base.py:309-366): the launches the per-stage Python path issues, in its order —
//   twin all-actions pass (online)   pa_mlp_q_all2            min Q(s, .) for the policy loss
//   actor row step + AdamW           pa_dsac_actor_rowstep, pa_mlp_adam
//   twin all-actions pass (target)   pa_mlp_q_all2            on the next states
//   Bellman targets                  pa_dsac_target_rowstep   (actor forward at s' inside)
//   twin critics row step            pa_mse_rowstep2          on x = state || rep(action)
//   weight gradients + AdamW (+ soft target update)           pa_mlp_adam2, or the single-network calls
//   entropy coefficient              pa_sac_alpha_step        (autotune)
// — bit-identical to that path.  Why: its interpreter cost (231 us per step) equals the device time
// of the step (tools/host_bound.py dsac); this is a C++ function call per launch.
extern "C" int64_t pa_dsac_scratch_floats(int32_t B, int32_t A) {
  return 5ll * B * A + 4ll * B;
}

extern "C" int pa_dsac_step(const pa_dsac_step_args* g, void* stream) {
  PA_REQUIRE(g && g->actor && g->critic1 && g->critic2 && g->state && g->next_state && g->xq &&
                 g->reward && g->terminated && g->curr_rep && g->next_rep && g->alpha && g->scratch &&
                 g->losses,
             PA_ERR_INVALID, "pa_dsac_step: null argument");
  PA_REQUIRE(g->B > 0 && g->A > 0 && g->AD > 0 && g->actor_step >= 1 && g->critic_step >= 1,
             PA_ERR_INVALID, "pa_dsac_step: bad sizes / steps");
  PA_REQUIRE(g->actor->d.dims[g->actor->L] == g->A, PA_ERR_INVALID,
             "pa_dsac_step: the actor outputs one logit per available-action slot");
  PA_REQUIRE(pa_rowstep_supported(g->actor, nullptr, g->A) && pa_rowstep_supported(g->critic1, g->critic2, 0),
             PA_ERR_UNSUPPORTED, "pa_dsac_step: networks outside the fused row steps' shapes");
  const int B = g->B, A = g->A;
  const int64_t BA = (int64_t)B * A;
  float* q1 = g->scratch;
  float* q2 = q1 + BA;
  float* nq1 = q2 + BA;
  float* nq2 = nq1 + BA;
  float* d_logits = nq2 + BA;
  float* h = g->h_out ? g->h_out : d_logits + BA;
  float* y = d_logits + BA + B;
  float* dq1 = y + B;
  float* dq2 = dq1 + B;
  // ---- actor update (:153-208)
  PA_TRY(pa_mlp_q_all2(g->critic1, g->critic2, 0, g->state, g->ld_state, g->curr_rep,
                       g->curr_rep_bstride, B, A, g->AD, q1, q2, stream));
  PA_TRY(pa_dsac_actor_rowstep(g->actor, g->state, g->ld_state, B, q1, q2, g->curr_mask, g->alpha,
                               d_logits, A, h, g->losses + 0, stream));
  PA_TRY(pa_mlp_adam(g->actor, g->actor_step, stream));
  // ---- critic update (:210-287): targets under the UPDATED policy, then the twin critics
  PA_TRY(pa_mlp_q_all2(g->critic1, g->critic2, 1, g->next_state, g->ld_next_state, g->next_rep,
                       g->next_rep_bstride, B, A, g->AD, nq1, nq2, stream));
  PA_TRY(pa_dsac_target_rowstep(g->actor, g->next_state, g->ld_next_state, B, nq1, nq2, g->next_mask,
                                g->alpha, g->reward, g->terminated, g->gamma, y, stream));
  PA_TRY(pa_mse_rowstep2(g->critic1, g->critic2, g->xq, g->ld_xq, B, y, 1.0f / (float)B, 0.5f, nullptr,
                         nullptr, dq1, dq2, g->losses + 1, stream));
  if (mlp_pair_fusable(g->critic1, g->critic2, g->tau >= 0.f)) {
    PA_TRY(pa_mlp_adam2(g->critic1, g->critic2, g->critic_step, g->tau, stream));
  } else {
    PA_TRY(pa_mlp_adam(g->critic1, g->critic_step, stream));
    PA_TRY(pa_mlp_adam(g->critic2, g->critic_step, stream));
    if (g->tau >= 0.f) {
      PA_TRY(pa_mlp_soft_update(g->critic1, g->tau, stream));
      PA_TRY(pa_mlp_soft_update(g->critic2, g->tau, stream));
    }
  }
  // ---- entropy coefficient (:134-151 of the continuous learner's form; h = sum_a P log(P + 1e-8))
  if (g->log_alpha) {
    PA_REQUIRE(g->alpha_m && g->alpha_v && g->alpha_step >= 1, PA_ERR_INVALID,
               "pa_dsac_step: the entropy optimizer's state is missing");
    PA_TRY(pa_sac_alpha_step(g->log_alpha, g->alpha_m, g->alpha_v, nullptr, g->alpha, h, B,
                             g->target_entropy, g->alpha_lr, g->alpha_beta1, g->alpha_beta2,
                             g->alpha_eps, g->alpha_weight_decay, 0, g->alpha_step, g->losses + 2,
                             stream));
  }
  return PA_OK;
}

// SoftActorCritic.learn's rounds in one call: per group of gather_rounds rounds ONE gather launch
// writes everything a step reads — state, next_state, x = state || rep(action), float reward,
// terminated, the availability masks and rep(available actions) of both states (the learner-side
// views of the gather kernel: no one-hot / concat launches) — then pa_dsac_step on consecutive
// slices.  `step0` points at the first slice of that workspace.
extern "C" int pa_dsac_learn(const pa_dsac_step_args* step0, pa_arena* arena, const pa_ac_loop_args* lp,
                             void* stream) {
  PA_REQUIRE(step0 && lp && arena && lp->idx_lists && lp->losses && lp->losses_stride >= 3 &&
                 lp->rounds >= 0,
             PA_ERR_INVALID, "pa_dsac_learn: bad loop arguments");
  const pa_batch_out& o = lp->batch;
  PA_REQUIRE(o.state == step0->state && o.next_state == step0->next_state && o.x == step0->xq &&
                 o.reward_f32 == step0->reward && o.terminated == step0->terminated &&
                 o.curr_avail_rep == step0->curr_rep && o.next_avail_rep == step0->next_rep &&
                 o.curr_mask == step0->curr_mask && o.next_mask == step0->next_mask &&
                 o.rep_dim == step0->AD,
             PA_ERR_INVALID, "pa_dsac_learn: the step reads the loop's batch workspace");
  PA_REQUIRE(step0->curr_rep_bstride == (int64_t)step0->A * step0->AD &&
                 step0->next_rep_bstride == (int64_t)step0->A * step0->AD &&
                 step0->ld_state == step0->S && step0->ld_next_state == step0->S &&
                 step0->ld_xq == step0->S + step0->AD,
             PA_ERR_INVALID, "pa_dsac_learn: workspace rows are dense");
  pa_dsac_step_args a = *step0;
  const int B = a.B, G = lp->gather_rounds > 1 ? lp->gather_rounds : 1;
  const int64_t AAD = (int64_t)a.A * a.AD;
  for (int r = 0; r < lp->rounds; ++r) {
    const int slot = r % G;
    if (slot == 0) {
      const int n = lp->rounds - r < G ? lp->rounds - r : G;
      PA_TRY(pa_arena_gather_device(arena, lp->idx_lists + (int64_t)r * B, n * B, &lp->batch, stream));
    }
    const int64_t row0 = (int64_t)slot * B;
    a.state = step0->state + row0 * a.S;
    a.next_state = step0->next_state + row0 * a.S;
    a.xq = step0->xq + row0 * (a.S + a.AD);
    a.reward = step0->reward + row0;
    a.terminated = step0->terminated + row0;
    a.curr_rep = step0->curr_rep + row0 * AAD;
    a.next_rep = step0->next_rep + row0 * AAD;
    a.curr_mask = step0->curr_mask ? step0->curr_mask + row0 * a.A : nullptr;
    a.next_mask = step0->next_mask ? step0->next_mask + row0 * a.A : nullptr;
    a.actor_step = step0->actor_step + r;
    a.critic_step = step0->critic_step + r;
    a.alpha_step = step0->alpha_step + r;
    a.losses = lp->losses + (int64_t)r * lp->losses_stride;
    PA_TRY(pa_dsac_step(&a, stream));
  }
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// ImplicitQLearning.learn_batch as ONE call (implicit_q_learning.py:159-269): the launches of the
// per-stage Python path in its order — target critics at (s, a), V(s'), V(s) (kept), the expectile
// value head + advantage weights, y = r + gamma V(s'), the twin critics' row step, the actor's
// forward and policy-extraction head (tanh-squashed / Gaussian / softmax), the two backward passes,
// then the optimizer steps in the reference's order (value, actor, critics + soft target update).
// Bit-identical to that path; the two host draws that pick the target critic of the value loss and
// of the advantage weights stay with the caller (pick_value, pick_actor).
extern "C" int64_t pa_iql_scratch_floats(int32_t B, int32_t S, int32_t A, int32_t head_width) {
  return (int64_t)B * (S + A) + 10ll * B + 2ll * B * head_width + 2ll * B * A;
}

extern "C" int pa_iql_step(const pa_iql_step_args* g, void* stream) {
  PA_REQUIRE(g && g->actor && g->value && g->critic1 && g->critic2 && g->state && g->next_state &&
                 g->action && g->reward && g->terminated && g->zeros && g->scratch && g->losses,
             PA_ERR_INVALID, "pa_iql_step: null argument");
  PA_REQUIRE(g->B > 0 && g->S > 0 && g->A > 0 && g->actor_kind >= 0 && g->actor_kind <= 2 &&
                 (g->pick_value | g->pick_actor) >= 0 && g->pick_value <= 1 && g->pick_actor <= 1 &&
                 g->actor_step >= 1 && g->value_step >= 1 && g->critic_step >= 1,
             PA_ERR_INVALID, "pa_iql_step: bad sizes / picks / steps");
  PA_REQUIRE(g->actor_kind == 2 || (g->low && g->high), PA_ERR_INVALID,
             "pa_iql_step: continuous actors need the action bounds");
  const int B = g->B, S = g->S, A = g->A;
  const int HW = g->actor->d.dims[g->actor->L];
  PA_REQUIRE(HW == (g->actor_kind == 1 ? 2 * A : A), PA_ERR_INVALID,
             "pa_iql_step: the actor's output width does not fit its kind");
  PA_REQUIRE(pa_rowstep_supported(g->critic1, g->critic2, 0), PA_ERR_UNSUPPORTED,
             "pa_iql_step: the critics are outside the fused row step's shapes");
  float* p = g->scratch;
  float* xq_own = p; p += (int64_t)B * (S + A);
  float* tq[2]; tq[0] = p; p += B; tq[1] = p; p += B;
  float* vn = p; p += B;
  float* v = p; p += B;
  float* dv = p; p += B;
  float* adv = p; p += B;
  float* y = p; p += B;
  float* dq1 = p; p += B;
  float* dq2 = p; p += B;
  float* logp = p; p += B;
  float* head = p; p += (int64_t)B * HW;
  float* d_head = p; p += (int64_t)B * HW;
  float* pred = p; p += (int64_t)B * A;
  float* d_pred = p;
  const float* xq = g->xq;
  int ld_xq = g->ld_xq;
  if (!xq) {
    PA_TRY(pa_concat_cols(g->state, g->ld_state, g->action, g->ld_action, xq_own, B, S, A, stream));
    xq = xq_own;
    ld_xq = S + A;
  }
  PA_TRY(pa_mlp_forward2(g->critic1, g->critic2, 1, xq, ld_xq, B, tq[0], 1, tq[1], 1, 0, stream));
  // V(s') first: the engine keeps ONE forward's activations per network for the backward pass
  PA_TRY(pa_mlp_forward(g->value, 0, g->next_state, g->ld_next_state, B, vn, 1, 0, stream));
  PA_TRY(pa_mlp_forward(g->value, 0, g->state, g->ld_state, B, v, 1, 1, stream));
  PA_TRY(pa_iql_value_head(tq[g->pick_value], tq[g->pick_actor], v, 1, g->expectile, g->temperature,
                           g->adv_clamp, B, dv, adv, g->losses + 0, stream));
  // y = V(s') gamma (1 - term) + r: pa_sac_twin(mode 1) with q1 = q2 = V(s'), alpha = 0
  PA_TRY(pa_sac_twin(1, vn, vn, g->zeros, g->zeros + B, g->reward, g->terminated, g->gamma, B, y,
                     nullptr, nullptr, stream));
  PA_TRY(pa_mse_rowstep2(g->critic1, g->critic2, xq, ld_xq, B, y, 1.0f / (float)B, 0.5f, nullptr,
                         nullptr, dq1, dq2, g->losses + 1, stream));
  PA_TRY(pa_mlp_forward(g->actor, 0, g->state, g->ld_state, B, head, HW, 1, stream));
  if (g->actor_kind == 0) {
    PA_TRY(pa_tanh_action(head, HW, nullptr, 0, g->low, g->high, 0.0f, B, A, pred, A, stream));
    PA_TRY(pa_awr_head(0, pred, A, g->action, g->ld_action, adv, B, A, d_pred, A, g->losses + 2, stream));
    PA_TRY(pa_tanh_action_grad(head, HW, g->low, g->high, d_pred, A, B, A, d_head, HW, stream));
  } else if (g->actor_kind == 1) {
    PA_TRY(pa_gauss_awr_head(head, HW, g->action, g->ld_action, g->low, g->high, adv, B, A, d_head, HW,
                             logp, g->losses + 2, stream));
  } else {
    PA_TRY(pa_awr_head(1, head, HW, g->action, g->ld_action, adv, B, A, d_head, HW, g->losses + 2, stream));
  }
  // one backward each (weight gradients deferred to the optimizer launches), then the steps in the
  // reference's order (:171-176) and the critics' soft target update
  PA_TRY(pa_mlp_backward(g->value, g->state, g->ld_state, B, dv, 1, 2, nullptr, 0, stream));
  PA_TRY(pa_mlp_backward(g->actor, g->state, g->ld_state, B, d_head, HW, 2, nullptr, 0, stream));
  PA_TRY(pa_mlp_adam(g->value, g->value_step, stream));
  PA_TRY(pa_mlp_adam(g->actor, g->actor_step, stream));
  if (mlp_pair_fusable(g->critic1, g->critic2, g->tau >= 0.f)) {
    PA_TRY(pa_mlp_adam2(g->critic1, g->critic2, g->critic_step, g->tau, stream));
  } else {
    PA_TRY(pa_mlp_adam(g->critic1, g->critic_step, stream));
    PA_TRY(pa_mlp_adam(g->critic2, g->critic_step, stream));
    if (g->tau >= 0.f) {
      PA_TRY(pa_mlp_soft_update(g->critic1, g->tau, stream));
      PA_TRY(pa_mlp_soft_update(g->critic2, g->tau, stream));
    }
  }
  return PA_OK;
}

// ImplicitQLearning.learn's rounds in one call: per group of gather_rounds rounds one gather launch
// (state, next_state, x = state || rep(action), float reward, terminated), then pa_iql_step per
// round on consecutive slices; picks [rounds][2] (host): the caller's draws, value loss first.
extern "C" int pa_iql_learn(const pa_iql_step_args* step0, pa_arena* arena, const pa_ac_loop_args* lp,
                            const int32_t* picks, void* stream) {
  PA_REQUIRE(step0 && lp && arena && picks && lp->idx_lists && lp->losses && lp->losses_stride >= 3 &&
                 lp->rounds >= 0,
             PA_ERR_INVALID, "pa_iql_learn: bad loop arguments");
  const pa_batch_out& o = lp->batch;
  PA_REQUIRE(o.state == step0->state && o.next_state == step0->next_state && o.x && o.x == step0->xq &&
                 o.reward_f32 == step0->reward && o.terminated == step0->terminated &&
                 o.rep_dim == step0->A && step0->action == step0->xq + step0->S,
             PA_ERR_INVALID, "pa_iql_learn: the step reads the loop's batch workspace");
  PA_REQUIRE(step0->ld_state == step0->S && step0->ld_next_state == step0->S &&
                 step0->ld_xq == step0->S + step0->A && step0->ld_action == step0->S + step0->A,
             PA_ERR_INVALID, "pa_iql_learn: workspace rows are dense");
  pa_iql_step_args a = *step0;
  const int B = a.B, G = lp->gather_rounds > 1 ? lp->gather_rounds : 1;
  for (int r = 0; r < lp->rounds; ++r) {
    const int slot = r % G;
    if (slot == 0) {
      const int n = lp->rounds - r < G ? lp->rounds - r : G;
      PA_TRY(pa_arena_gather_device(arena, lp->idx_lists + (int64_t)r * B, n * B, &lp->batch, stream));
    }
    const int64_t row0 = (int64_t)slot * B;
    a.state = step0->state + row0 * a.S;
    a.next_state = step0->next_state + row0 * a.S;
    a.xq = step0->xq + row0 * (a.S + a.A);
    a.action = a.xq + a.S;
    a.reward = step0->reward + row0;
    a.terminated = step0->terminated + row0;
    a.pick_value = picks[2 * r];
    a.pick_actor = picks[2 * r + 1];
    a.actor_step = step0->actor_step + r;
    a.value_step = step0->value_step + r;
    a.critic_step = step0->critic_step + r;
    a.losses = lp->losses + (int64_t)r * lp->losses_stride;
    PA_TRY(pa_iql_step(&a, stream));
  }
  return PA_OK;
}
