// host_common.hpp -- host-side declarations shared by the translation units of libautompc_hip.so:
// the handle and plan structs, error plumbing, upload/download helpers, the (W, NT, MT) kernel
// dispatch and the heavy launchers' declarations.
//
//   api.cpp                       the C ABI (include/autompc_hip.h), host logic, the small kernels
//   api_model.cpp                 MLP staging: weight packing, ampc_set_mlp, ampc_set_mlp_dev
//   launch_mlp.cpp   -DAMPC_T=..  MLP forward / Jacobian launchers         } one unit per precision:
//   launch_mppi.cpp  -DAMPC_T=..  MPPI rollout / update launcher           } each holds the explicit
//   launch_ilqr.cpp  -DAMPC_T=..  iLQR sweep / line-search launcher        } instantiation for AMPC_T
#pragma once
#include "autompc_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mlp_kernels.hpp"
#include "linear_kernels.hpp"
#include "ilqr_kernels.hpp"
#include "ilqr_ls4.hpp"
#include "ilqr_lsw.hpp"
#include "ilqr_wide.hpp"
#include "mppi_kernels.hpp"
#include "mppi_rollout4.hpp"
#include "rng_kernels.hpp"
#include "sindy_kernels.hpp"
#include "score_kernels.hpp"
#include "shapes.hpp"

using namespace ampc;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
extern thread_local std::string g_err;     // defined in api.cpp (in jit_plugin.cpp for a shape plugin)
inline int fail(const std::string& msg) {
  g_err = msg;
  return -1;
}
#define HIP_OK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
  } while (0)
#define REQUIRE(cond, msg) \
  do {                     \
    if (!(cond)) return fail(msg); \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n ? n : 16);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// function-local device scratch: freed on every exit path (HIP_OK / REQUIRE return early)
struct ScopedBuf : DevBuf {
  ScopedBuf() = default;
  ScopedBuf(const ScopedBuf&) = delete;
  ScopedBuf& operator=(const ScopedBuf&) = delete;
  ~ScopedBuf() { release(); }
};

static constexpr size_t kLdsLimit = 160 * 1024;

inline int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
}

// ---------------------------------------------------------------------------------------------
// shape plugin: the launchers of ONE model shape, compiled at run time (shapes.hpp, jit_host.hpp)
// ---------------------------------------------------------------------------------------------
struct ampc_mppi_plan;
struct ampc_ilqr_plan;
struct JitPlugin {
  void* dl = nullptr;
  int (*mppi_solve)(ampc_mppi_plan*) = nullptr;          // mppi_solve_impl<T>
  int (*ilqr_iter)(ampc_ilqr_plan*, int) = nullptr;      // ilqr_launch_iter<T>
  int (*ilqr_refresh)(ampc_ilqr_plan*) = nullptr;        // ilqr_refresh_jacobians<T>
  const char* (*last_error)() = nullptr;
};
// result of a plugin call -> this library's error slot
inline int jit_result(const JitPlugin* j, int rc) {
  if (rc != 0) g_err = std::string("shape plugin: ") + (j->last_error ? j->last_error() : "error");
  return rc;
}

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct ampc_handle {
  int device = 0;
  int n_cus = 256;      // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int precision = AMPC_F64;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  size_t esz() const { return precision == AMPC_F64 ? 8 : 4; }
  // Plans hold raw pointers to their handle.  Language bindings with garbage collection may
  // destroy a handle before the plans built on it, so the handle is reference counted: it is
  // actually freed when ampc_destroy has been called AND the last plan is gone.
  int refs = 0;
  bool dead = false;
  // ampc_handle_set_jit: may this handle START (and wait for) the run-time build of kernels specialised for its
  // model's shape?  A plugin that is already loaded or cached on disk is used either way.
  bool jit_build = true;

  // model (host copy, double) ---------------------------------------------------------------
  bool has_mlp = false;
  bool has_sindy = false;         // SINDy feature-library dynamics instead of an MLP
  bool has_lin = false;           // wide linear model (65..256 states): linear_kernels.hpp
  int l_nxp = 0, l_kp = 0;
  DevBuf lin_buf;                 // fragments of [A | B], then the plain row-major copy
  bool has_model() const { return has_mlp || has_sindy || has_lin; }
  int s_nfeat = 0, s_continuous = 0, s_strict = 1, s_ntrig = 0, s_npow = 0, s_ntab = 0, s_nmon = 0, s_npool = 0;
  double s_dt = 0.0;
  DevBuf sindy_int, sindy_flt;    // kind|a0|a1 (int), par|xi (T)
  int nx = 0, nu = 0, n_hidden = 0, act = 0;
  int hidden[kMaxHidden] = {0, 0, 0, 0};
  int hpad = 0, nt = 0, nw = 4, k1p = 0, nxp = 0;  // nw = waves per workgroup (4 or 8)
  std::vector<std::vector<double>> W, b;
  std::vector<double> norm;
  DevBuf model_buf;   // all packed arrays, contiguous
  const void* wout_plain = nullptr;  // [nx][hpad] view into model_buf (Jacobian chain)
  MlpDev<double> md{};
  MlpDev<float> mf{};

  // cost blocks / bounds ----------------------------------------------------------------------
  int n_costs = 0, obs_dim = 0, cost_stride = 0, cost_diag = 0, cost_affine = 0;
  DevBuf cost_buf;
  // indicator terms of the MPPI stage cost (ampc_set_indicator_costs): [n_ind][ind_stride(obs_dim)], compute precision
  int n_ind = 0;
  DevBuf ind_buf;
  bool has_bounds = false;
  std::vector<double> lo, hi;
  DevBuf bounds_buf;  // lo/scale, hi/scale, scale   (MPPI units)
  DevBuf ubounds_buf; // lo, hi                     (iLQR clips in physical units)

  // scratch for the batched model calls -----------------------------------------------------------
  DevBuf s_states, s_ctrls, s_out, s_dz, s_jx, s_ju;
};

// Start (or find) the shape plugin of the staged model (api.cpp; never blocks).
void ampc_internal_jit_kick(ampc_handle* h);

// Plans hold references on their handles (api.cpp).
void handle_release(ampc_handle* h);

template <typename T> inline MlpDev<T>& model_of(ampc_handle* h);
template <> inline MlpDev<double>& model_of<double>(ampc_handle* h) { return h->md; }
template <> inline MlpDev<float>& model_of<float>(ampc_handle* h) { return h->mf; }

// ---- several controller models of one shape in a plan (ampc_*_plan_set_models) ----------------------------
inline int check_same_shape(const ampc_handle* h, const ampc_handle* m, const char* who) {
  const std::string w(who);
  REQUIRE(m != nullptr, w + ": NULL model handle");
  REQUIRE(m->has_mlp && h->has_mlp, w + ": per-problem models are MLP models");
  REQUIRE(m->device == h->device && m->precision == h->precision, w + ": models must share the plan's device and precision");
  bool same = m->nx == h->nx && m->nu == h->nu && m->n_hidden == h->n_hidden && m->act == h->act &&
              m->hpad == h->hpad && m->nw == h->nw && m->nt == h->nt;
  for (int l = 0; l < kMaxHidden; ++l) same = same && m->hidden[l] == h->hidden[l];
  REQUIRE(same, w + ": every model must have the shape (dimensions, hidden layers, activation) of the plan's model");
  return 0;
}

// device table of the models' buffer offsets (mlp_tile.hpp: shift_model); takes a reference on every handle
template <typename T>
inline int build_model_table(ampc_handle* h, int n, ampc_handle* const* ms, DevBuf* tab, std::vector<ampc_handle*>* keep) {
  std::vector<long long> host(n);
  const MlpDev<T>& m0 = model_of<T>(h);
  for (int i = 0; i < n; ++i) {
    const MlpDev<T>& mi = model_of<T>(ms[i]);
    host[i] = (long long)((const char*)mi.wbase - (const char*)m0.wbase);
    // same shape => same packing: every array sits at the same offset of its model's buffer
    bool same = true;
    for (int l = 0; l <= h->n_hidden; ++l)
      same = same && (mi.w[l] - mi.wbase) == (m0.w[l] - m0.wbase) && (mi.b[l] - mi.wbase) == (m0.b[l] - m0.wbase) &&
             (mi.w4[l] - mi.wbase) == (m0.w4[l] - m0.wbase);
    REQUIRE(same && ((const T*)ms[i]->wout_plain - mi.wbase) == ((const T*)h->wout_plain - m0.wbase),
            "internal: models of one shape are packed differently");
  }
  HIP_OK(tab->reserve(host.size() * sizeof(long long)));
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(tab->p, host.data(), host.size() * sizeof(long long), hipMemcpyHostToDevice));
  for (ampc_handle* old : *keep) handle_release(old);
  keep->assign(ms, ms + n);
  for (ampc_handle* mh : *keep) mh->refs++;
  return 0;
}


static const char* const kNeedStatic =
    ": several controller models in one plan run on the kernels specialised for the model's shape; they are "
    "not available for this plan (AMPC_STATIC=0 / AMPC_JIT=0, no hipcc for the run-time build of an unregistered "
    "shape, a tile geometry without a specialised instantiation, or a cost with indicator terms)";


// LDS bytes the kernels need on top of their own regions for the staged feature program
template <typename T> static size_t sindy_stage_bytes(const ampc_handle* h) {
  if (h->s_ntab == 0) return 0;
  const size_t b = sindy_prog_elems(h->nx, h->s_nfeat, h->s_ntrig, h->s_npow, h->s_nmon, h->s_npool, sizeof(T)) * sizeof(T);
  return b <= (size_t)kSindyStageBytes ? b + 2 * sizeof(T) : 0;
}

template <typename T> static LinDev<T> lin_of(const ampc_handle* h) {
  LinDev<T> m;
  m.nx = h->nx; m.nu = h->nu; m.nxp = h->l_nxp; m.kp = h->l_kp; m.ntile = h->l_nxp / 16; m.ksn = h->l_kp / 4;
  m.wf = (const T*)h->lin_buf.p;
  m.plain = m.wf + (size_t)m.ntile * m.ksn * 64;
  m.ldj = round_up(h->nx + h->nu, 16);
  m.jp = m.plain + (size_t)round_up(h->nx * (h->nx + h->nu), 4);
  return m;
}

template <typename T> static SindyDev<T> sindy_of(const ampc_handle* h) {
  SindyDev<T> m;
  m.nx = h->nx; m.nu = h->nu; m.n_feat = h->s_nfeat; m.continuous = h->s_continuous;
  m.strict = h->s_strict; m.dt = (T)h->s_dt;
  const int* ip = (const int*)h->sindy_int.p;
  const int nf = h->s_nfeat;
  m.kind = ip; m.a0 = ip + nf; m.a1 = ip + 2 * nf;
  m.fx = ip + 3 * nf; m.fy = ip + 4 * nf; m.tvar = ip + 5 * nf; m.pvar = ip + 6 * nf;
  m.moff = ip + 7 * nf; m.mcnt = ip + 8 * nf; m.mpool = ip + 9 * nf;
  m.n_mon = h->s_nmon; m.n_pool = h->s_npool;
  const T* fp = (const T*)h->sindy_flt.p;
  m.par = fp; m.xi = fp + nf;
  m.tpar = fp + (size_t)nf * (h->nx + 1);
  m.ppar = m.tpar + nf;
  m.n_trig = h->s_ntrig; m.n_pow = h->s_npow; m.n_tab = h->s_ntab;
  m.stage = sindy_stage_bytes<T>(h) > 0;
  return m;
}

template <typename T>
static hipError_t upload_converted(void* dst, const double* src, size_t n, hipStream_t s) {
  if (sizeof(T) == 8) return hipMemcpyAsync(dst, src, n * 8, hipMemcpyHostToDevice, s);
  std::vector<float> tmp(n);
  for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
  hipError_t e = hipMemcpyAsync(dst, tmp.data(), n * 4, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);  // tmp goes out of scope
}
template <typename T>
static hipError_t download_converted(double* dst, const void* src, size_t n, hipStream_t s) {
  if (sizeof(T) == 8) {
    hipError_t e = hipMemcpyAsync(dst, src, n * 8, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
  }
  std::vector<float> tmp(n);
  hipError_t e = hipMemcpyAsync(tmp.data(), src, n * 4, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  for (size_t i = 0; i < n; ++i) dst[i] = (double)tmp[i];
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// kernel dispatch on (NT, MT)
// ---------------------------------------------------------------------------------------------
template <typename K> static hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  // (the attribute sticks to the function: asked again only when a larger amount is needed -- a hot control
  //  step launches the same kernel with the same size thousands of times a second)
  static thread_local const void* last_fn[4] = {nullptr, nullptr, nullptr, nullptr};
  static thread_local size_t last_bytes[4] = {0, 0, 0, 0};
  static thread_local int last_dev[4] = {-1, -1, -1, -1};
  const void* fn = reinterpret_cast<const void*>(kernel);
  int dev = 0;
  (void)hipGetDevice(&dev);
  int slot = -1;
  for (int i = 0; i < 4; ++i)
    if (last_fn[i] == fn && last_dev[i] == dev) { slot = i; break; }
  if (slot >= 0 && last_bytes[slot] >= bytes) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  if (slot < 0) {
    static thread_local int next = 0;
    slot = next;
    next = (next + 1) % 4;
  }
  last_fn[slot] = fn; last_bytes[slot] = bytes; last_dev[slot] = dev;
  return hipSuccess;
}

// (W, NT) is one of (4,1) (8,1) (4,3) (8,2); MT is 1, 2 or 4.  The body sees W, NT, MT and WD
// (wide outputs: model states 33..64).
#define AMPC_CASE(WV, NTV, MTV, WDV, ...) \
  case ((WV) * 100 + (NTV) * 10 + (MTV)) * 2 + (WDV): { constexpr int W = WV, NT = NTV, MT = MTV; constexpr bool WD = WDV != 0; (void)WD; __VA_ARGS__; } break;
#define AMPC_DISPATCH_W(WIDEV, WV, NTV, MTV, ...)                            \
  do {                                                                       \
    switch (((WV) * 100 + (NTV) * 10 + (MTV)) * 2 + ((WIDEV) ? 1 : 0)) {     \
      AMPC_CASE(4, 1, 1, 0, __VA_ARGS__) AMPC_CASE(4, 1, 2, 0, __VA_ARGS__) AMPC_CASE(4, 1, 4, 0, __VA_ARGS__) \
      AMPC_CASE(8, 1, 1, 0, __VA_ARGS__) AMPC_CASE(8, 1, 2, 0, __VA_ARGS__) AMPC_CASE(8, 1, 4, 0, __VA_ARGS__) \
      AMPC_CASE(4, 3, 1, 0, __VA_ARGS__) AMPC_CASE(4, 3, 2, 0, __VA_ARGS__) AMPC_CASE(4, 3, 4, 0, __VA_ARGS__) \
      AMPC_CASE(8, 2, 1, 0, __VA_ARGS__) AMPC_CASE(8, 2, 2, 0, __VA_ARGS__) AMPC_CASE(8, 2, 4, 0, __VA_ARGS__) \
      AMPC_CASE(4, 1, 1, 1, __VA_ARGS__) AMPC_CASE(4, 1, 2, 1, __VA_ARGS__) AMPC_CASE(4, 1, 4, 1, __VA_ARGS__) \
      AMPC_CASE(8, 1, 1, 1, __VA_ARGS__) AMPC_CASE(8, 1, 2, 1, __VA_ARGS__) AMPC_CASE(8, 1, 4, 1, __VA_ARGS__) \
      AMPC_CASE(4, 3, 1, 1, __VA_ARGS__) AMPC_CASE(4, 3, 2, 1, __VA_ARGS__) AMPC_CASE(4, 3, 4, 1, __VA_ARGS__) \
      AMPC_CASE(8, 2, 1, 1, __VA_ARGS__) AMPC_CASE(8, 2, 2, 1, __VA_ARGS__) AMPC_CASE(8, 2, 4, 1, __VA_ARGS__) \
      default: return fail("internal: unsupported (W, NT, MT, wide) combination");  \
    }                                                                        \
  } while (0)
// wide <=> the staged model has more than two 16-column output tiles
#define AMPC_DISPATCH(H, MTV, ...) AMPC_DISPATCH_W((H)->nxp > 32, (H)->nw, (H)->nt, MTV, __VA_ARGS__)

// Largest tile (MT) that fits LDS, preferring enough workgroups to cover the 256 CUs twice.
// LDS map for a tile: separate partials region when it fits the 160 KB, else aliased onto `act`.
template <typename T>
static TileLds tile_lds_for(const ampc_handle* h, const MlpDev<T>& m, int M, size_t extra_elems) {
  // richest map first: ping-pong activations + separate partials, then drop one at a time
  TileLds L = make_tile_lds(m, M, h->nw, true, env_int("AMPC_PINGPONG", 1) != 0);
  if (((size_t)L.extra + extra_elems) * sizeof(T) > kLdsLimit) L = make_tile_lds(m, M, h->nw, true, false);
  if (((size_t)L.extra + extra_elems) * sizeof(T) > kLdsLimit) L = make_tile_lds(m, M, h->nw, false, false);
  return L;
}

template <typename T>
static int choose_mt(const ampc_handle* h, const MlpDev<T>& m, long long total_rows,
                     size_t extra_elems, int forced_mt = 0) {
  const int forced = forced_mt ? forced_mt : env_int("AMPC_MT", 0);
  int best = 1;
  for (int mt : {1, 2, 4}) {
    TileLds L = tile_lds_for<T>(h, m, 16 * mt, extra_elems);
    const size_t bytes = ((size_t)L.extra + extra_elems) * sizeof(T);
    if (bytes > kLdsLimit) break;
    if (forced == mt) return mt;
    if (mt == 4 && h->nt == 3 && sizeof(T) == 8) break;   // 12 f64 accumulator tiles per wave spill
    if (h->hpad == 256) {
      // 256-wide networks, measured on 2 / 4 / 8 config-3 solves per launch (round 6, profiles/r06_tile_height.log):
      //   f32: 16-row tiles at every size (0.79-0.80 of the f32 peak against 0.77 / 0.73 at 32 / 64 rows) -- the
      //        16-row kernel holds 116 VGPRs, so TWO workgroups share a CU and one tile's serial chain (layer 0,
      //        output reduction, state update, barriers) runs behind the other's MFMAs;
      //   f64: 32-row tiles once they give a workgroup per CU (0.84 against 0.82 at 16 rows and 0.80 at 64).
      if (sizeof(T) == 4) { if (mt == 1) best = 1; }
      else if (mt == 1 || (mt == 2 && total_rows / 32 >= h->n_cus)) best = mt;
      continue;
    }
    if (mt == 1 || total_rows / (16 * mt) >= 512) best = mt;
  }
  return best;
}

// ---------------------------------------------------------------------------------------------
// MPPI plan
// ---------------------------------------------------------------------------------------------
struct ampc_mppi_plan {
  ampc_handle* h = nullptr;
  int B = 0, term_mode = 0, mt = 1, n_tiles = 0, max_h = 0;
  int forced_mt = 0;    // ampc_mppi_plan_set_geometry: tile height fixed by the caller (0 = automatic)
  bool forced_quad = false;   // ... to the four-row kernel (tile_rows = 4)
  bool quad = false;    // the four-row rollout kernel runs (mppi_rollout4.hpp); mt is 0 then
  bool eps_inline = false;    // the plan's noise is Philox(eps_seed, eps_stream), formed inside the four-row rollout
  uint64_t eps_seed = 0, eps_stream = 0;
  // Noise one solve ahead (device Philox noise, sixteen-row plans): callers draw stream s, s + 1, s + 2, ...
  // (MPPI.run(), bench.py), so once that pattern has been seen a solve's combine launch also forms the
  // noise of stream + 1 into eps_next (extra workgroups of the same launch), and the next
  // ampc_mppi_generate_eps(seed, stream + 1) swaps the buffers instead of launching the generator: one
  // launch and its gap (~10 us) less per solve.  Anything that changes what the generator would produce
  // (problems, noise ids, an uploaded buffer) drops the speculation.
  DevBuf eps_next;
  bool ahead_on = false, ahead_valid = false, eps_from_generator = false;
  uint64_t ahead_seed = 0, ahead_stream = 0;
  // several controller models of the handle's shape (ampc_mppi_plan_set_models): device table of their
  // descriptors, MppiProblem::model names a problem's entry; the handles are kept alive by the plan
  std::vector<ampc_handle*> models;
  std::vector<int> model_idx;     // [B]
  DevBuf mlp_tab;                 // [n_models] byte offsets of the models' buffers from the plan model's
  DevBuf tile_order;              // [n_tiles] workgroup -> tile, XCD-aware by model (MppiArgs::tile_order)
  bool use_tile_order = false;
  std::vector<int> tile_prob_host;   // [n_tiles] tile -> problem, as uploaded
  int lift_n = 0;             // ampc_mppi_plan_set_state_lift: basis functions of the controller model's lift
  DevBuf lift_prog;           // [lift_n][2] (kind, parameter) in compute precision
  uint64_t step_offset = 0;   // ampc_mppi_plan_set_step_offset: index of the next closed loop's first control step
  void* pin_x0 = nullptr;     // ampc_mppi_run: pinned staging of x0 in / controls out (compute precision)
  void* pin_u = nullptr;
  // ... mapped into the device's address space: the rollout reads x0 and the update writes u straight from / to
  // host memory, followed by a per-problem sequence word the host polls (MppiArgs::done_flag)
  void* pin_x0_dev = nullptr;
  void* pin_u_dev = nullptr;
  unsigned long long* pin_flag = nullptr;       // [B]
  unsigned long long* pin_flag_dev = nullptr;
  unsigned long long run_seq = 0;
  bool host_io = false;       // the launch being assembled is such a one-call control step (make_args)
  bool u_in_pin = false;      // the last solve was one: its controls are in pin_u, not in u_out (ampc_mppi_download)
  // numpy-stream mode: the NEXT call's normals are drawn right behind this call's update, from the generator
  // state this call hands back (lg_pre_*); a next call that presents exactly that state finds its noise in place
  bool lg_pre = false;
  hipEvent_t lg_pre_done = nullptr;   // the pre-drawn call's draw kernel has finished
  hipStream_t lg_draw = nullptr;      // ... it runs on this stream, next to the current solve, into eps_pre
  DevBuf eps_pre;                     // (swapped with eps when the next call presents the predicted generator state)
  DevBuf cl_obs, cl_ctl, cl_next, cl_sim;   // scratch of the device-resident closed loop (trajectory rows, next states)
  bool lg_pre_inflight = false;       // a pre-draw has been enqueued and not been waited for yet
  int lg_pre_pos = 0, lg_pre_has_gauss = 0;
  double lg_pre_cached = 0.0;
  std::vector<uint32_t> lg_pre_key;
  long long lg_pre_draw[6] = {0, 0, 0, 0, 0, 0};   // the LegacyDraw of the pre-drawn call (api.cpp)
  int static_shape = -1;  // >= 0: id of the registered shape whose specialised kernel runs (shapes.hpp)
  const JitPlugin* jit = nullptr;   // the shape's kernels live in a run-time compiled plugin (id 0 there)
  int static_lv = 0;      // which LDS map variant (StaticShape LV) the plan's tile uses
  int tile_m = 16;      // samples per rollout workgroup (16*mt for the MLP tile, 64 for SINDy)
  std::vector<int> N, H, cost_idx, a_off;
  std::vector<unsigned> noise_id;   // per problem: key of its device noise stream (default: index)
  std::vector<double> sigma, lmda;
  std::vector<long long> eps_off, epso_off, cost_off;
  long long sum_n = 0, sum_hnu = 0, sum_nhnu = 0;
  DevBuf probs, tile_prob, x0, act[2], eps, eps_out, costs, term_last, u_out, tile_stat, tile_part, tile_done;
  bool fused_combine = false;   // the four-row rollout's last workgroup finishes the update (no combine launch)
  // numpy legacy-stream generation (ampc_mppi_legacy_normal).  The raw MT19937 stream of the NEXT
  // call is generated speculatively on a side stream (it only depends on the generator state this
  // call leaves behind) and used if the next call indeed starts from that state.
  DevBuf lg_key[2], lg_stream[2], lg_cnt, lg_scale, lg_xraw, lg_poly[4], lg_win, lg_logtab;
  void* lg_pin = nullptr;         // pinned results of a draw, written by the draw kernel (last attempt, cached value, total, final stream block, status)
  void* lg_pin_dev = nullptr;     //   ... its device address
  unsigned lg_epoch = 0;          // call counter that tags the draw kernel's look-back words in lg_cnt
  bool lg_scale_set = false;      // sqrt(sigma_b) uploaded (the sigmas of a plan never change)
  // The raw MT19937 stream is generated AHEAD of the draws, several calls' worth per buffer, on a
  // side stream (api.cpp: legacy_enqueue / legacy_speculate / legacy_finish):
  hipStream_t lg_side = nullptr;
  hipEvent_t lg_evs[2] = {nullptr, nullptr};   // buffer b completely generated
  hipEvent_t lg_drawn = nullptr;   // main-stream generation finished (a call the run-ahead missed)
  int lg_blocks[2] = {0, 0};      // blocks of 624 words held by lg_stream[b]
  int lg_cur = 0;                 // buffer the generator's state currently lies in
  int lg_blk0 = 0;                // ... and the block of it that is the generator's key
  bool lg_spec = false;           // lg_cur / lg_blk0 / lg_spec_key / lg_spec_pos describe the state the
  int lg_spec_pos = 0;            //   previous call left: a call presenting exactly it takes its words
  std::vector<uint32_t> lg_spec_key;   // from the buffer
  bool lg_next = false;           // lg_stream[1 - lg_cur] holds (or is receiving) the continuation
  int lg_next_from = 0;           //   of lg_stream[lg_cur] from this block on
  int lg_hits = 0;                // consecutive calls served from the run-ahead
  long long lg_redraws = 0;       // draws repeated because a look-back wait of the draw kernel expired (ampc_mppi_plan_info)
  int lds_eps = -1, lds_red = 0;   // fused softmin update (tile partials) when the noise fits LDS
  bool keep_eps_out = true;        // materialise the clipped noise in HBM (download / non-fused)
  int cur = 0;          // act[cur] is the input of the next solve
  bool costs_final = true;
  bool solved = false;
  size_t lds_bytes = 0;
  TileLds L{};
  int lds_aseq = 0, lds_cost = 0;
  // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
  bool timing = false;
  int timing_stride = 1;        // every n-th solve is bracketed by events (three event records cost ~11 us of a solve)
  long long timing_count = 0;
  std::vector<hipEvent_t> ev;   // 3 per solve: before rollout, after rollout, after update
  size_t ev_used = 0;
};

template <typename T> static MppiArgs<T> make_args(ampc_mppi_plan* p) {
  ampc_handle* h = p->h;
  MppiArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.mlp = model_of<T>(h);
  a.lds = p->L;
  a.lds_aseq = p->lds_aseq;
  a.lds_cost = p->lds_cost;
  a.obs_dim = h->obs_dim;
  a.cost_stride = h->cost_stride;
  a.term_mode = p->term_mode;
  a.max_h = p->max_h;
  a.cost_diag = h->cost_diag;
  a.cost_affine = h->cost_affine;
  a.lds_eps = p->lds_eps;
  a.lds_red = p->lds_red;
  a.write_eps_out = (p->keep_eps_out || p->lds_eps < 0) ? 1 : 0;
  a.eps_inline = (p->eps_inline && p->quad) ? 1 : 0;
  a.eps_seed = p->eps_seed; a.eps_stream = p->eps_stream;
  a.hnu_stride = p->max_h * h->nu;
  a.tile_stat = (T*)p->tile_stat.p;
  a.tile_part = (T*)p->tile_part.p;
  a.tile_done = (int*)p->tile_done.p;
  a.fused_combine = p->fused_combine ? 1 : 0;
  a.model_delta = p->models.empty() ? nullptr : (const long long*)p->mlp_tab.p;
  a.tile_order = p->use_tile_order ? (const int*)p->tile_order.p : nullptr;
  a.ind_tab = (const T*)h->ind_buf.p;
  a.n_ind = h->n_ind;
  a.costs_par = (const T*)h->cost_buf.p;
  a.bounds = (const T*)h->bounds_buf.p;
  a.probs = (const MppiProblem<T>*)p->probs.p;
  a.tile_prob = (const int*)p->tile_prob.p;
  a.x0 = (const T*)p->x0.p;
  a.act_in = (const T*)p->act[p->cur].p;
  a.act_out = (T*)p->act[p->cur ^ 1].p;
  a.eps = (const T*)p->eps.p;
  a.eps_out = (T*)p->eps_out.p;
  a.costs = (T*)p->costs.p;
  a.term_last = (T*)p->term_last.p;
  a.u_out = (T*)p->u_out.p;
  a.done_flag = nullptr;
  a.done_seq = 0;
  if (p->host_io) {           // one-call control step: x0 / u / completion word in mapped host memory
    a.x0 = (const T*)p->pin_x0_dev;
    a.u_out = (T*)p->pin_u_dev;
    a.done_flag = p->pin_flag_dev;
    a.done_seq = p->run_seq;
  }
  return a;
}

// ---------------------------------------------------------------------------------------------
// iLQR plan
// ---------------------------------------------------------------------------------------------
struct ampc_ilqr_plan {
  ampc_handle* h = nullptr;
  int B = 0, H = 0, ls_n = 10, bounded = 0, term_goal = 0;
  int static_shape = -1;     // >= 0: registered shape whose specialised kernels run (shapes.hpp)
  const JitPlugin* jit = nullptr;   // ... in a run-time compiled plugin (shape id 0 there)
  double dt = 0, u_threshold = 1e-3, ls_discount = 0.2, ls_cost_threshold = 0.3;
  std::vector<int> cost_idx;
  DevBuf d_cost_idx, states, ctrls, jx, ju, Ks, ks, ls_states, ls_ctrls, obj, flags, dz, ric;
  // flags layout (ints): converged[B] active[B] iters[B] status[B] refresh[B] ls_rows[B] ls_count[B] ls_pass[B] ls_need[B]
  int use_ls4 = 1, use_mfma_sweep = 1, par_passes = 1;   // (AMPC_LS4_PAR = 0: passes one after the other)
  int ls_split = 0;          // (AMPC_LS4_SPLIT = 1: large batches search in two launches, ilqr_ls4.hpp; measured: no gain)
  // Line-search kernel when the passes are NOT side by side (many problems per launch, in lock-step):
  // four-row passes (ilqr_ls4.hpp: a launch lasts as many passes as its slowest search needs) or all step
  // sizes in one twelve-row pass (ilqr_lsw.hpp: 2.4 four-row passes' time whatever the searches need).
  // ls_rb: 0 = chosen per poll from what the slots' last searches needed (ls_need: twelve rows as soon as
  // some search needed a third pass), 1 / 3 = AMPC_LS4_RB forces one.  Same results either way.
  int ls_rb = 0, ls_rb_now = 1;
  // kernel choices, fixed at plan build (AMPC_LS4 / AMPC_RICCATI = 0: the general kernels)
  TileLds L{};
  int lds_work = 0, lds_xn = 0;
  size_t lds_bytes = 0;
  int last_iterations = 0;
  long long last_ls_rows = 0;   // candidate rows the line searches of the last solve rolled out (all problems)
  // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg):
  // events bracket the four launches of an iteration: sweep | line search | forward | Jacobians
  bool timing = false;
  int timing_stride = 1;        // queue: every n-th iteration is bracketed (five event records cost ~15 us of an iteration)
  std::vector<hipEvent_t> ev;   // 5 per timed iteration
  size_t ev_used = 0;
  hipEvent_t* ev_cur = nullptr; // the running iteration's five events (null: not timed)
  // timed iterations that did real work: the polled loop queues up to 2 * kPoll iterations past
  // convergence (no-ops); only the first max_b iters[b] iterations of a solve count in the averages
  std::vector<unsigned char> ev_live;   // one flag per timed iteration (ev_used / 5 of them)
  int last_effective = 0;               // iterations of the last solve in which a problem was active
  int active_hint = 0;                  // problems known to be still active (from the last poll; never
                                        // below the true count): few enough and the passes of a line
                                        // search run side by side on the idle CUs
  // convergence polling: the `active` flags of one batch of iterations are copied to pinned host
  // memory behind that batch and read while the NEXT batch is already queued
  int* poll_host = nullptr;     // [2][B + 2] pinned (queue mode: + the queue's two counters)
  hipEvent_t poll_ev[2] = {nullptr, nullptr};
  // continuous batching (ampc_ilqr_solve_queue): P problems stream through the B slots
  bool queue_on = false;        // kernels read the per-slot mode and the per-problem iteration cap
  int queue_max_iter = 0;
  DevBuf q_ctl;                 // ints: [0] next, [1] harvested, then slot_prob[B], slot_mode[B]
  // problems / episodes of different horizons in one plan (ampc_ilqr_solve_queue_var, _closed_loop_var): every
  // array keeps the plan's horizon H as its stride, a slot's loops run to slot_h[slot] <= H
  bool var_h = false;
  DevBuf slot_h;                // [B] ints
  // several controller models of the handle's shape (ampc_ilqr_plan_set_models): device table of their
  // descriptors; a queue problem / episode names its entry (ampc_ilqr_*_var), the slot carries it
  std::vector<ampc_handle*> models;
  DevBuf mlp_tab, slot_model;   // [n_models] byte offsets of the models' buffers, [B] ints
  DevBuf slot_of;               // [B] workgroup -> slot, the slots with work first (queues with more slots than CUs)
  bool compact_on = false;      //   ... in use by the running queue
  bool var_model = false;       // the running queue / closed loop uses per-slot models
  DevBuf q_x0, q_u, q_cost, q_states, q_ctrls, q_Ks, q_ks, q_obj, q_flags;
  DevBuf vj;                    // wide linear models: the sweep's VJ scratch [B][nxp][ldj] (ilqr_wide.hpp)
  DevBuf c_ints, c_iters, c_stage, c_obs, c_ctl;   // ampc_ilqr_closed_loop: chain bookkeeping, staged rows, trajectories
  long long last_queue_launches = 0;   // iterations launched by the last queue solve
};

template <typename T> static IlqrArgs<T> make_ilqr_args(ampc_ilqr_plan* p, int mode) {
  ampc_handle* h = p->h;
  IlqrArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.mlp = model_of<T>(h);
  if (h->has_sindy) a.sindy = sindy_of<T>(h);
  if (h->has_lin) { a.lin = lin_of<T>(h); a.vj = (T*)p->vj.p; }
  a.lds_xn = p->lds_xn;
  a.lds = p->L;
  a.lds_work = p->lds_work;
  a.H = p->H; a.obs_dim = h->obs_dim; a.cost_stride = h->cost_stride; a.bounded = p->bounded;
  a.ls_n = p->ls_n; a.mode = mode; a.cost_diag = h->cost_diag; a.cost_affine = h->cost_affine; a.term_goal = p->term_goal;
  a.dt = (T)p->dt; a.u_threshold = (T)p->u_threshold; a.ls_cost_threshold = (T)p->ls_cost_threshold;
  for (int j = 0; j < kIlqrMaxLs; ++j) a.alphas[j] = (T)std::pow(p->ls_discount, (double)j);
  a.costs_par = (const T*)h->cost_buf.p;
  a.cost_idx = (const int*)p->d_cost_idx.p;
  a.ubounds = (const T*)h->ubounds_buf.p;
  a.states = (T*)p->states.p; a.ctrls = (T*)p->ctrls.p;
  a.jx = (const T*)p->jx.p; a.ju = (const T*)p->ju.p;
  a.Ks = (T*)p->Ks.p; a.ks = (T*)p->ks.p;
  a.ls_states = (T*)p->ls_states.p; a.ls_ctrls = (T*)p->ls_ctrls.p;
  a.obj = (T*)p->obj.p;
  int* f = (int*)p->flags.p;
  a.converged = f; a.active = f + p->B; a.iters = f + 2 * p->B; a.status = f + 3 * p->B;
  a.refresh = f + 4 * p->B; a.ls_rows = f + 5 * p->B; a.ls_count = f + 6 * p->B; a.ls_pass = f + 7 * p->B; a.ls_need = f + 8 * p->B;
  a.ric = (T*)p->ric.p;
  if (p->queue_on) {
    a.slot_mode = (int*)p->q_ctl.p + 2 + p->B;
    a.slot_of = p->compact_on ? (const int*)p->slot_of.p : nullptr;
    a.max_iter = p->queue_max_iter;
    if (p->var_h) a.slot_h = (const int*)p->slot_h.p;
    if (p->var_model) { a.model_delta = (const long long*)p->mlp_tab.p; a.slot_model = (const int*)p->slot_model.p; }
  }
  return a;
}

// id of the registered static shape (shapes.hpp) the staged MLP matches, or -1
template <typename T> inline int static_shape_of(const ampc_handle* h, const MlpDev<T>& m) {
  if (!h->has_mlp) return -1;
#define AMPC_SHAPE_MATCH(ID, NX, NU, NO, NH, HPAD)                                                  \
  if (m.nx == NX && m.nu == NU && h->obs_dim == NO && m.n_hidden == NH && m.hpad == HPAD &&         \
      m.tail4 == (StaticShape<NX, NU, NO, NH, HPAD>::template tail4<T> ? 1 : 0))                    \
    return ID;
  AMPC_STATIC_SHAPES(AMPC_SHAPE_MATCH)
#undef AMPC_SHAPE_MATCH
  return -1;
}

// Run AMPC_SD_BODY (a macro the caller defines; it sees `SH`, `W`, `NT`) for the registered shape
// `SID`, with the activation folded when RELU:
//   #define AMPC_SD_BODY  { auto k = kernel<T, NT, .., W, SH>; launch(k); }
//   AMPC_STATIC_DISPATCH(sid, relu);
//   #undef AMPC_SD_BODY
#define AMPC_SD_CASE(ID, NX, NU, NO, NH, HPAD, RELU)                              \
  case (ID) * 2 + (RELU): {                                                       \
    using SH = StaticShape<NX, NU, NO, NH, HPAD, (RELU) ? 0 : -1>;                 \
    constexpr int W = (HPAD % 128 == 0) ? 8 : 4, NT = HPAD / (16 * W);            \
    (void)W; (void)NT;                                                            \
    AMPC_SD_BODY                                                                  \
  } break;
#define AMPC_SD_ONE(ID, NX, NU, NO, NH, HPAD) \
  AMPC_SD_CASE(ID, NX, NU, NO, NH, HPAD, 0) AMPC_SD_CASE(ID, NX, NU, NO, NH, HPAD, 1)
#define AMPC_STATIC_DISPATCH(SID, RELU)                                  \
  switch ((SID) * 2 + ((RELU) ? 1 : 0)) {                                \
    AMPC_STATIC_SHAPES(AMPC_SD_ONE)                                      \
    default: return fail("internal: unknown static shape");              \
  }

// LDS map variant (StaticShape LV) of a tile map produced by tile_lds_for, or -1 if it is none of
// the three maps the shape implies
template <typename T> inline int lds_variant_of(const MlpDev<T>& m, const TileLds& L, int M, int W) {
  for (int lv = 0; lv < 3; ++lv) {
    const TileLds S = tile_lds_dims((int)sizeof(T), m.hpad, m.k1p, m.nxp, m.n_hidden, M, W, lv < 2, lv == 0);
    if (std::memcmp(&S, &L, sizeof(TileLds)) == 0) return lv;
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------
// heavy launchers: defined in launch_*.cpp, explicitly instantiated there for double and float
// ---------------------------------------------------------------------------------------------
template <typename T> int pred_impl(ampc_handle* h, const double* states, const double* ctrls, double* out,
                                    double* jx, double* ju, int n);
template <typename T> int surrogate_step(ampc_handle* h, ampc_handle* sur, const void* x, const void* u,
                                         void* x_next, int B);
// Workgroups (slots) a per-slot launch of the running iteration needs: with the slots that have work listed first
// (compact_on) and an upper bound of their number known from the polls, the grid ends there.
inline int ilqr_grid_slots(const ampc_ilqr_plan* p) {
  return (p->compact_on && p->active_hint > 0 && p->active_hint < p->B) ? p->active_hint : p->B;
}
template <typename T> int ilqr_refresh_jacobians(ampc_ilqr_plan* p);
template <typename T> int mppi_solve_impl(ampc_mppi_plan* p);
template <typename T> int ilqr_launch_iter(ampc_ilqr_plan* p, int mode);
