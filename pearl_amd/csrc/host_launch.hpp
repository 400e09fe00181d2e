// Host-side launch helpers shared by dqn.hip and mlp.hip.
#pragma once
#include <stdlib.h>
#include "dqn_kernels.hpp"
#include "target_split_kernel.hpp"
#include "target_h2_kernel.hpp"

namespace pa {
namespace {

template <typename K>
int set_max_smem(K kernel, size_t bytes) {
  PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PA_OK;
}

// One launch, one or two independent problems (blockIdx.z).
// min_smem: request at least this much dynamic LDS per workgroup (unused by the kernel) — a launch
// that must not share a CU with another kernel's resident workgroups asks for more than they leave.
template <bool B_KS>
int launch_linear(const GemmArgs* probs, int nprob, hipStream_t s, size_t min_smem = 0) {
  constexpr int KW = 4;
  static size_t configured = 0;
  auto kern = linear_kernel<B_KS, KW>;
  LinArgs a;
  memset(&a, 0, sizeof(a));
  size_t smem = 0;
  int gx = 0, gy = 0;
  for (int i = 0; i < nprob; ++i) {
    a.p[i] = probs[i];
    const size_t b = linear_smem_bytes<B_KS, KW>(probs[i].K);
    smem = b > smem ? b : smem;
    gx = (int)ceil_div(probs[i].N, G_BN) > gx ? (int)ceil_div(probs[i].N, G_BN) : gx;
    gy = (int)ceil_div(probs[i].M, G_BM) > gy ? (int)ceil_div(probs[i].M, G_BM) : gy;
  }
  smem = min_smem > smem ? min_smem : smem;
  if (smem > configured) {
    int rc = set_max_smem(kern, smem);
    if (rc != PA_OK) return rc;
    configured = smem;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy, (unsigned)nprob), dim3(128 * KW), smem,
                     s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// every operand 16-byte aligned with pitches that are multiples of 4, AD <= 16, all eight waves of
// a workgroup own hidden units in both layers: the kernels' FAST instantiation applies
inline bool target_fast_shape(const TargetArgs& a, int nkg) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return al(a.U) && al(a.feat) && al(a.W1a) && al(a.b2) && al(a.w3) && (a.ldu & 3) == 0 &&
         (a.AD & 3) == 0 && a.AD <= 16 && (a.feat_bstride & 3) == 0 && (a.ldw1 & 3) == 0 &&
         a.H1 == nkg * 8 && a.H2 == 256 && nkg * 8 >= 256;
}

template <int NKG, bool FAST>
int launch_target_t(const TargetArgs& a, hipStream_t s) {
  static bool configured = false;
  const size_t smem = target_smem_bytes(a.H1);
  if (!configured) {
    int rc = set_max_smem(target_fused_kernel<NKG, FAST>, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  // persistent mode: three workgroups per CU are offered; two fit (LDS), the surplus and the ones
  // on reserved CUs exit immediately
  const unsigned grid = a.tile_ctr ? (unsigned)(3 * a.ntiles < 3 * 256 ? 3 * a.ntiles : 3 * 256)
                                   : (unsigned)ceil_div(a.B, a.bpw);
  hipLaunchKernelGGL((target_fused_kernel<NKG, FAST>), dim3(grid), dim3(512), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// target_split_kernel: one workgroup per CU (99 KB of LDS); persistent mode offers two per CU so
// that the ones landing on reserved CUs (and exiting) leave no other CU empty
template <int KS1>
int launch_target_split_t(const TargetArgs& a, hipStream_t s) {
  static bool configured = false;
  const size_t smem = target_split_smem_bytes();
  if (!configured) {
    int rc = set_max_smem(target_split_kernel<KS1>, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  unsigned grid = (unsigned)ceil_div(a.B, a.bpw);
  // PEARL_AMD_PERSIST_OFFER: workgroups offered to a persistent launch beside a CU partition.  The
  // hardware places each exactly once: a non-reserved CU that is full when its turn comes (a chain
  // workgroup with a large LDS / register footprint sits there) gets no target workgroup for the
  // whole launch, and the offers that land on reserved CUs are used up within microseconds.  Measured
  // (round 5, PEARL_AMD_DEBUG_WORKERS): with 512 offers 107 of the 128 non-reserved CUs took tiles in
  // an average window launch (75 beside the paired row pass's 82 KB workgroups), with 2048: 123-128.
  static const unsigned offer = []() {
    const char* v = getenv("PEARL_AMD_PERSIST_OFFER");
    const int n = v ? atoi(v) : 2048;
    return (unsigned)(n >= 256 ? n : 2048);
  }();
  if (a.tile_ctr) grid = a.reserved ? offer : (unsigned)(a.ntiles < 256 ? a.ntiles : 256);
  hipLaunchKernelGGL(target_split_kernel<KS1>, dim3(grid), dim3(512), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
// U = W1s' s' + b1' for a.B transitions with the split tile's own arithmetic (bit-identical to what
// the fused tile forms), as one launch: 32 transitions per workgroup
template <int KS1>
int launch_u_split_t(const TargetArgs& a, float* U, hipStream_t s) {
  static bool configured = false;
  const size_t smem = target_split_smem_bytes();
  if (!configured) {
    int rc = set_max_smem(u_split_kernel<KS1>, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  hipLaunchKernelGGL(u_split_kernel<KS1>, dim3((unsigned)ceil_div(a.B, 32)), dim3(512), smem, s, a, U);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
inline int launch_u_split(const TargetArgs& a, float* U, hipStream_t s) {
  switch (a.S) {
    case 64: return launch_u_split_t<4>(a, U, s);
    case 128: return launch_u_split_t<8>(a, U, s);
    case 256: return launch_u_split_t<16>(a, U, s);
    default:
      set_error("u_split_kernel: not built for S = %d", a.S);
      return PA_ERR_UNSUPPORTED;
  }
}
// target_split32_kernel: 32-row tiles on four waves, two workgroups per CU (target_split_kernel.hpp).
// Takes the 64-row tile's arguments and re-derives the tile geometry.  Tiles that read U, A <= 32.
// Default: the passes that ask for it (TargetArgs::rows_hint — Double DQN's two passes, the
// all-actions passes of pa_mlp_q_all: +11 % on DoubleDQN); the DQN window loop keeps the 64-row tile
// (its persistent launches share the chip with the online chain: 27.5 M against 26.4 M in steady
// state, same box).  PEARL_AMD_TARGET_ROWS=32|64 / pa_debug_set_target_rows: either one everywhere.
inline bool target_split32_ok(const TargetArgs& a) {
  const int mode = target_rows_mode();   // 0: per pass (a.rows_hint), 32 / 64: everywhere
  const bool want = mode == 32 || (mode == 0 && a.rows_hint == 32);
  return want && !a.W1sp && a.A <= TS32_ROWS && a.prof == nullptr;
}
inline int launch_target_split32(const TargetArgs& a64, hipStream_t s) {
  static bool configured = false;
  const size_t smem = target_split32_smem_bytes();
  if (!configured) {
    int rc = set_max_smem(target_split32_kernel, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  TargetArgs a = a64;
  const int prio_rows = a64.prio_tiles * a64.bpw;
  a.bpw = TS32_ROWS / a.A;
  a.ntiles = (int)ceil_div(a.B, a.bpw);
  a.prio_tiles = prio_rows > 0 ? (int)ceil_div(prio_rows, a.bpw) : 0;
  a.prof = nullptr;
  unsigned grid = (unsigned)a.ntiles;
  // persistent mode: two workgroups per CU are resident; offer enough that the ones landing on
  // reserved CUs (and exiting) leave no other CU short
  if (a.tile_ctr) grid = a.reserved ? 1024u : (unsigned)(a.ntiles < 512 ? a.ntiles : 512);
  hipLaunchKernelGGL(target_split32_kernel, dim3(grid), dim3(256), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}
// state widths the tile can form U from itself (compile-time k-step counts)
inline bool target_split_fusable_S(int S) { return S == 64 || S == 128 || S == 256; }
inline int launch_target_split(const TargetArgs& a, hipStream_t s) {
  if (target_split32_ok(a)) return launch_target_split32(a, s);
  if (!a.W1sp) return launch_target_split_t<0>(a, s);
  switch (a.S) {
    case 64: return launch_target_split_t<4>(a, s);
    case 128: return launch_target_split_t<8>(a, s);
    case 256: return launch_target_split_t<16>(a, s);
    default:
      set_error("target_split_kernel: fused first layer not built for S = %d", a.S);
      return PA_ERR_UNSUPPORTED;
  }
}

// target_h2_kernel: the 64-row tile on the fp16 matrix pipe; grids as launch_target_split_t
inline int launch_target_h2(const TargetArgs& a, hipStream_t s) {
  static bool configured = false;
  const size_t smem = target_h2_smem_bytes();
  if (!configured) {
    int rc = set_max_smem(target_h2_kernel, smem);
    if (rc != PA_OK) return rc;
    configured = true;
  }
  unsigned grid = (unsigned)ceil_div(a.B, a.bpw);
  static const unsigned offer = []() {
    const char* v = getenv("PEARL_AMD_PERSIST_OFFER");
    const int n = v ? atoi(v) : 2048;
    return (unsigned)(n >= 256 ? n : 2048);
  }();
  if (a.tile_ctr) grid = a.reserved ? offer : (unsigned)(a.ntiles < 256 ? a.ntiles : 256);
  hipLaunchKernelGGL(target_h2_kernel, dim3(grid), dim3(512), smem, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

// classic grid (or, with a.tile_ctr, persistent tiles) of target_fused_kernel for the shape at hand
inline int launch_target(const TargetArgs& a, hipStream_t s) {
  // (a forced tile shape — PEARL_AMD_TARGET_ROWS, rows_hint — means the bf16x3 kernels: their 32- and
  //  64-row forms are the pair the tile-shape tests compare)
  if (a.W2h && !a.W1sp && t_nkg(a.H1) == 32 && target_fast_shape(a, 32) && target_rows_mode() == 0 &&
      a.rows_hint == 0)
    return launch_target_h2(a, s);
  if (a.W2sp && t_nkg(a.H1) == 32 && target_fast_shape(a, 32)) return launch_target_split(a, s);
  switch (t_nkg(a.H1)) {
    case 8: return launch_target_t<8, false>(a, s);
    case 16: return launch_target_t<16, false>(a, s);
    default:
      return target_fast_shape(a, 32) ? launch_target_t<32, true>(a, s)
                                      : launch_target_t<32, false>(a, s);
  }
}

// weight_grad_kernel with the split-K policy: batches of 2048 rows and more are cut into slices of
// >= 512 rows per workgroup (a 64 x 32 tile over 4096 rows is 27 us of MFMA on one CU).  The scratch
// for the partial tiles is one buffer per stream (stream_scratch, common.hpp), grown on demand.
// `variants`: optional replacements for the four kernels, in the order {32-row split, 32-row fp32,
// 64-row split, 64-row fp32} (dqn.hip: the *_pair kernels that honour DwProblem::dZb)
typedef void (*DwKernelFn)(DwArgs);
inline int launch_weight_grad(DwArgs& a, bool loss_wg, hipStream_t s, const DwKernelFn* variants = nullptr) {
  // the number of slices is the one that fills the chip once: total_tiles * ks <= 256 workgroups
  // (72 tiles x 4 slices = 288 left 32 CUs with two workgroups each and everyone waiting for them:
  // 33.7 us per PPO network at B = 4096 against 14.5 us for the same tiles at B = 1024)
  int ks = 1;
  static const int min_b = []() {
    const char* v = getenv("PEARL_AMD_DW_MINB");
    const int n = v ? atoi(v) : 2048;
    return n > 0 ? n : 2048;
  }();
  static const int slice_rows = []() {
    const char* v = getenv("PEARL_AMD_DW_SLICE");
    const int n = v ? atoi(v) : 512;
    return n >= 64 ? n : 512;
  }();
  if (a.B >= min_b) {
    ks = a.B / slice_rows;
    static const int slots = []() {
      const char* v = getenv("PEARL_AMD_DW_SLOTS");
      const int n = v ? atoi(v) : 256;
      return n > 0 ? n : 256;
    }();
    const int fit = a.total_tiles > 0 ? slots / a.total_tiles : 1;
    if (ks > fit) ks = fit;
    if (ks > 8) ks = 8;
    if (ks < 1) ks = 1;
  }
  {
    // TIMING EXPERIMENT ONLY: a forced slice count for the DQN chain's launch (see
    // PEARL_AMD_DEBUG_DW_ONLY in dqn.hip) — whether the CUs a dependency split frees buy anything
    static const int force_ks = []() {
      const char* v = getenv("PEARL_AMD_DEBUG_DW_KS");
      return v ? atoi(v) : 0;
    }();
    if (force_ks >= 1 && force_ks <= 8 && a.B == 1024) ks = force_ks;
  }
  a.ksplit = ks;
  {
    static const int xcd_env = []() {
      const char* v = getenv("PEARL_AMD_DW_XCD_ORDER");
      return v ? atoi(v) : 0;
    }();
    a.xcd_order = (xcd_env != 0 && ks == 1 && a.B >= min_b && (a.total_tiles & 7) == 0) ? 1 : 0;
  }
  a.kscratch = nullptr;
  a.ktickets = nullptr;
  if (ks > 1) {
    // partial tiles and tickets: per stream (stream_scratch); the tickets are zero between launches
    // (the last arriver of a tile resets its own)
    const size_t need = (size_t)a.total_tiles * ks * (DW_TM * DW_TN + DW_TM);
    a.kscratch = stream_scratch(SCR_DW_PARTIALS, s, need);
    a.ktickets = reinterpret_cast<unsigned*>(stream_scratch(SCR_DW_TICKETS, s, (size_t)a.total_tiles));
    if (!a.kscratch || !a.ktickets) return PA_ERR_NOMEM;
  }
  const unsigned grid = (unsigned)(a.total_tiles * ks) + (loss_wg ? 1u : 0u);
  // Batches whose tiles are MFMA-bound (>= PEARL_AMD_DW_MINB rows: PPO's and the bandit's 4096) run
  // the main loop on the bf16 matrix pipe at fp32 accuracy (weight_grad_split_kernel: 2.7x the fp32
  // matrix rate) when every matrix problem has whole, aligned 16- / 8-byte operand vectors.
  // PEARL_AMD_DW_SPLIT=0: the fp32-MFMA kernel everywhere.
  const int split_mode = dw_split_mode();   // pa_debug_set_dw_split / PEARL_AMD_DW_SPLIT
  a.split = 0;
  // (32-row tiles — the DQN chain's launch — from PEARL_AMD_DW_MINB32 = 1024 rows on: 0.7 us of a
  //  36 us round; smaller batches keep the fp32 loop, whose sums the tile-shape
  //  and loop-equivalence tests compare bitwise)
  static const int min_b32 = []() {
    const char* v = getenv("PEARL_AMD_DW_MINB32");
    const int n = v ? atoi(v) : 1024;
    return n > 0 ? n : 1024;
  }();
  if (split_mode > 0 && ((a.tm != 32 && a.B >= min_b) || (a.tm == 32 && a.B >= min_b32) || split_mode == 2)) {
    bool ok = true;
    const unsigned za = a.tm == 32 ? 7u : 15u;   // dZ vectors: two / four units
    for (int k = 0; k < a.nprob; ++k) {
      const DwProblem& P = a.p[k];
      if (P.M == 1) continue;   // the GEMV path
      // (row pitches of whole vectors: a vector that straddles M or N reads pad / neighbour
      //  columns of its own row — inside the allocation — into accumulator rows / columns beyond
      //  the problem, which the epilogue never stores)
      ok = ok && (P.ldz & (za >> 2)) == 0 && (reinterpret_cast<uintptr_t>(P.dZ) & za) == 0 &&
           (P.ldx & 1) == 0 && (reinterpret_cast<uintptr_t>(P.X) & 7) == 0;
    }
    a.split = ok ? 1 : 0;
  }
  if (variants) {
    const DwKernelFn k = variants[(a.tm == 32 ? 0 : 2) + (a.split ? 0 : 1)];
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, s, a);
  } else if (a.tm == 32 && a.split) hipLaunchKernelGGL(weight_grad_split_kernel32, dim3(grid), dim3(512), 0, s, a);
  else if (a.tm == 32) hipLaunchKernelGGL(weight_grad_kernel32, dim3(grid), dim3(512), 0, s, a);
  else if (a.split) hipLaunchKernelGGL(weight_grad_split_kernel, dim3(grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(weight_grad_kernel, dim3(grid), dim3(512), 0, s, a);
  PA_LAUNCH_CHECK();
  return PA_OK;
}

}  // namespace
}  // namespace pa
