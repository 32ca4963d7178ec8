// api_ilqr.cpp -- iLQR plans: batch solves, continuous batching (queue), device-resident iLQR episodes
// Part of the C ABI of libautompc_hip.so (include/autompc_hip.h); see api.cpp for the map of the translation units.
// Built with hipcc for gfx950 only.
#include "host_common.hpp"
#include <mutex>
#include "jit_host.hpp"                // shape plugins compiled at run time

extern template int pred_impl<double>(ampc_handle*, const double*, const double*, double*, double*, double*, int);
extern template int pred_impl<float>(ampc_handle*, const double*, const double*, double*, double*, double*, int);
extern template int surrogate_step<double>(ampc_handle*, ampc_handle*, const void*, const void*, void*, int);
extern template int surrogate_step<float>(ampc_handle*, ampc_handle*, const void*, const void*, void*, int);
extern template int ilqr_refresh_jacobians<double>(ampc_ilqr_plan*);
extern template int ilqr_refresh_jacobians<float>(ampc_ilqr_plan*);
extern template int mppi_solve_impl<double>(ampc_mppi_plan*);
extern template int mppi_solve_impl<float>(ampc_mppi_plan*);
extern template int ilqr_launch_iter<double>(ampc_ilqr_plan*, int);
extern template int ilqr_launch_iter<float>(ampc_ilqr_plan*, int);

// Which line-search kernel the next launches of a many-problem plan take (ampc_ilqr_plan::ls_rb): the
// four-row kernel's launch lasts as many passes as its slowest search, the twelve-row kernel's about 2.4
// passes' time whatever the searches need -- twelve rows once some active slot's last search needed a
// third pass.  (A speed heuristic only: both kernels give the same results bit for bit.)
static int ls_rb_from_poll(const int* active, const int* need, int B) {
  int most = 0;
  for (int b = 0; b < B; ++b)
    if (active[b] != 0 && need[b] > most) most = need[b];
  return most >= 3 ? 3 : 1;
}

template <typename T> static int ilqr_plan_build(ampc_ilqr_plan* p) {
  ampc_handle* h = p->h;
  const MlpDev<T>& m = model_of<T>(h);
  const int nx = h->nx, nu = h->nu, B = p->B, H = p->H;
  const IlqrWork wk = make_ilqr_work(nx, nu, h->cost_stride, h->has_lin);
  if (h->has_sindy) {
    std::memset(&p->L, 0, sizeof(p->L));
    p->L.xu = 0;
    p->L.xu_stride = nx + nu + 1;
    p->lds_xn = round_up(16 * p->L.xu_stride, 4);
    p->L.extra = round_up(p->lds_xn + 16 * nx + 16 * h->s_ntab, 4);   // xnext + table scratch
  } else if (h->has_lin) {         // line search: [x | u] rows for lin_tile, next states, compact work map
    std::memset(&p->L, 0, sizeof(p->L));
    p->L.xu = 0;
    p->L.xu_stride = lin_xs(h->l_kp, (int)sizeof(T));
    p->lds_xn = round_up(16 * p->L.xu_stride, 4);
    p->L.extra = round_up(p->lds_xn + 16 * nx, 4);
  } else {
    p->L = tile_lds_for<T>(h, m, 16, (size_t)wk.total + 8);
  }
  p->lds_work = p->L.extra;
  p->lds_bytes = ((size_t)p->lds_work + wk.total) * sizeof(T);
  p->use_ls4 = env_int("AMPC_LS4", 1) != 0;
  p->use_mfma_sweep = env_int("AMPC_RICCATI", 1) != 0;
  p->par_passes = env_int("AMPC_LS4_PAR", 1) != 0;
  p->ls_split = env_int("AMPC_LS4_SPLIT", 0) != 0;
  { const int rb = env_int("AMPC_LS4_RB", 0); p->ls_rb = (rb == 1 || rb == 3) ? rb : 0; }
  p->static_shape = -1;
  p->jit = nullptr;
  if (!h->has_sindy && !h->has_lin && env_int("AMPC_STATIC", 1) != 0) {
    int sid = static_shape_of<T>(h, m);
    if (sid < 0 && (p->jit = jit::get<T>(h)) != nullptr) sid = 0;      // run-time compiled shape
    if (sid >= 0) {
      const TileLds S = tile_lds_dims((int)sizeof(T), m.hpad, m.k1p, m.nxp, m.n_hidden, 16, h->nw, true, true);
      if (std::memcmp(&S, &p->L, sizeof(TileLds)) == 0) p->static_shape = sid;
    }
    if (p->static_shape < 0) p->jit = nullptr;
  }
  REQUIRE(p->lds_bytes <= kLdsLimit, "ilqr plan: model does not fit the 160 KB LDS");
  REQUIRE((size_t)wk.total * sizeof(T) <= kLdsLimit,
          "ilqr plan: the Riccati workspace for this state dimension does not fit the 160 KB LDS");
  if (h->has_lin)
    REQUIRE((size_t)make_wide_lds(nx, nu, h->obs_dim).total * sizeof(T) <= kLdsLimit,
            "ilqr plan: the sweep's workspace for this state / observation dimension does not fit the 160 KB LDS");
  const size_t e = sizeof(T);
  HIP_OK(p->d_cost_idx.reserve(B * sizeof(int)));
  HIP_OK(hipMemcpy(p->d_cost_idx.p, p->cost_idx.data(), B * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(p->states.reserve((size_t)B * (H + 1) * nx * e));
  HIP_OK(p->ctrls.reserve((size_t)B * H * nu * e));
  if (!h->has_lin) {              // (a linear model's Jacobians are the constant [A | B]: LinDev::jp)
    HIP_OK(p->jx.reserve((size_t)B * H * nx * nx * e));
    HIP_OK(p->ju.reserve((size_t)B * H * nx * nu * e));
  } else {
    HIP_OK(p->vj.reserve((size_t)B * h->l_nxp * round_up(nx + nu, 16) * e));
  }
  HIP_OK(p->Ks.reserve((size_t)B * H * nu * nx * e));
  HIP_OK(p->ks.reserve((size_t)B * H * nu * e));
  HIP_OK(hipMemset(p->Ks.p, 0, (size_t)B * H * nu * nx * e));
  HIP_OK(hipMemset(p->ks.p, 0, (size_t)B * H * nu * e));
  HIP_OK(p->ls_states.reserve((size_t)B * p->ls_n * (H + 1) * nx * e));
  HIP_OK(p->ls_ctrls.reserve((size_t)B * p->ls_n * H * nu * e));
  HIP_OK(p->obj.reserve((size_t)B * e));
  HIP_OK(p->flags.reserve((size_t)9 * B * sizeof(int)));
  HIP_OK(hipMemset(p->flags.p, 0, (size_t)9 * B * sizeof(int)));
  const int rows = B * H;
  const int n_pad = round_up(rows, 64);
  if (!h->has_sindy && !h->has_lin) HIP_OK(p->dz.reserve((size_t)m.n_hidden * n_pad * m.hpad * e));
  HIP_OK(p->ric.reserve((size_t)kRicStride * p->B * e));
  return 0;
}

extern "C" int ampc_ilqr_plan_create(ampc_handle* h, int B, int horizon, double dt,
                                     const int* cost_index, int clip_to_bounds,
                                     ampc_ilqr_plan** out) {
  REQUIRE(h && out, "ampc_ilqr_plan_create: NULL argument");
  if (h->has_lin) {      // wide linear models: ilqr_wide.hpp (V [nx][nx] in LDS, per-thread Quu solves)
    REQUIRE(h->nx <= kLinMaxIlqrNx, "ampc_ilqr_plan_create: iLQR on wide linear models takes up to 128 model states (the "
                                    "value function's Hessian lives in LDS); larger ones run MPPI and the closed loop");
    REQUIRE(h->nu == 1 || h->nu == 2 || h->nu == 3 || h->nu == 4 || h->nu == 6 || h->nu == 8,
            "ampc_ilqr_plan_create: iLQR on wide linear models is built for 1, 2, 3, 4, 6 or 8 controls");
  } else {
    REQUIRE(h->nx + h->nu + 1 <= 64,
            "ampc_ilqr_plan_create: state dim + ctrl dim must be <= 63 (one wave holds the augmented Quu system)");
  }
  REQUIRE(h->has_model() && h->n_costs > 0, "ampc_ilqr_plan_create: model and cost must be set first");
  REQUIRE(h->n_ind == 0, "ampc_ilqr_plan_create: the handle's cost has indicator terms (threshold / box): they have no "
                         "gradient or Hessian, iLQR takes sums of quadratic costs only");
  REQUIRE(B >= 1 && horizon >= 1, "ampc_ilqr_plan_create: B >= 1 and horizon >= 1 required");
  REQUIRE(!clip_to_bounds || h->has_bounds, "ampc_ilqr_plan_create: bounds requested but not set");
  HIP_OK(hipSetDevice(h->device));
  ampc_ilqr_plan* p = new ampc_ilqr_plan();
  h->refs++;
  p->h = h; p->B = B; p->H = horizon; p->dt = dt; p->bounded = clip_to_bounds ? 1 : 0;
  for (int b = 0; b < B; ++b) {
    const int ci = cost_index ? cost_index[b] : 0;
    if (ci < 0 || ci >= h->n_costs) { h->refs--; delete p; return fail("ampc_ilqr_plan_create: bad cost_index"); }
    p->cost_idx.push_back(ci);
  }
  int rc = h->precision == AMPC_F64 ? ilqr_plan_build<double>(p) : ilqr_plan_build<float>(p);
  if (rc) { ampc_ilqr_plan_destroy(p); return rc; }
  *out = p;
  return 0;
}

extern "C" int ampc_ilqr_plan_set_constants(ampc_ilqr_plan* p, double u_threshold, int ls_max_iter, double ls_discount,
                                            double ls_cost_threshold) {
  REQUIRE(p, "ampc_ilqr_plan_set_constants: NULL plan");
  REQUIRE(ls_max_iter >= 1 && ls_max_iter <= kIlqrMaxLs, "ampc_ilqr_plan_set_constants: 1..16 line-search step sizes");
  REQUIRE(u_threshold >= 0.0 && ls_discount > 0.0, "ampc_ilqr_plan_set_constants: u_threshold >= 0 and ls_discount > 0");
  ampc_handle* h = p->h;
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));            // the candidate buffers below may be re-allocated
  p->u_threshold = u_threshold;
  p->ls_discount = ls_discount;
  p->ls_cost_threshold = ls_cost_threshold;
  p->ls_n = ls_max_iter;
  const size_t e = h->esz();
  HIP_OK(p->ls_states.reserve((size_t)p->B * p->ls_n * (p->H + 1) * h->nx * e));
  HIP_OK(p->ls_ctrls.reserve((size_t)p->B * p->ls_n * p->H * h->nu * e));
  return 0;
}

extern "C" int ampc_ilqr_plan_set_timing(ampc_ilqr_plan* p, int enable) {
  REQUIRE(p, "ampc_ilqr_plan_set_timing: NULL plan");
  p->timing = enable != 0;
  p->timing_stride = enable > 1 ? enable : 1;      // (queue: enable = n > 1 brackets every n-th iteration)
  p->ev_used = 0;
  return 0;
}

extern "C" int ampc_ilqr_plan_timing(ampc_ilqr_plan* p, double* kernel_ms, int* iterations) {
  REQUIRE(p && kernel_ms, "ampc_ilqr_plan_timing: NULL argument");
  HIP_OK(hipSetDevice(p->h->device));
  HIP_OK(hipStreamSynchronize(p->h->stream));
  double acc[4] = {0, 0, 0, 0};
  const size_t n = p->ev_used / 5;
  size_t live = 0;
  for (size_t i = 0; i < n; ++i) {
    if (i < p->ev_live.size() && !p->ev_live[i]) continue;     // queued past convergence: a no-op
    ++live;
    for (int k = 0; k < 4; ++k) {
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, p->ev[5 * i + k], p->ev[5 * i + k + 1]));
      acc[k] += ms;
    }
  }
  for (int k = 0; k < 4; ++k) kernel_ms[k] = live ? acc[k] / live : 0.0;
  if (iterations) *iterations = (int)live;
  p->ev_used = 0;
  p->ev_live.clear();
  return 0;
}

extern "C" int ampc_ilqr_plan_stats(ampc_ilqr_plan* p, long long* iterations, long long* candidate_rows) {
  REQUIRE(p, "ampc_ilqr_plan_stats: NULL plan");
  if (iterations) *iterations = p->last_effective;     // performed (launched: last_iterations)
  if (candidate_rows) *candidate_rows = p->last_ls_rows;
  return 0;
}

extern "C" int ampc_ilqr_plan_set_terminal_goal(ampc_ilqr_plan* p, int use_goal) {
  REQUIRE(p, "ampc_ilqr_plan_set_terminal_goal: NULL plan");
  p->term_goal = use_goal ? 1 : 0;
  return 0;
}

extern "C" int ampc_ilqr_plan_destroy(ampc_ilqr_plan* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->h->device);
  (void)hipStreamSynchronize(p->h->stream);
  DevBuf* bufs[] = {&p->d_cost_idx, &p->states, &p->ctrls, &p->jx, &p->ju, &p->Ks, &p->ks,
                    &p->ls_states, &p->ls_ctrls, &p->obj, &p->flags, &p->dz, &p->ric,
                    &p->q_ctl, &p->q_x0, &p->q_u, &p->q_cost, &p->q_states, &p->q_ctrls, &p->q_Ks, &p->q_ks,
                    &p->q_obj, &p->q_flags, &p->c_ints, &p->c_iters, &p->c_stage, &p->c_obs, &p->c_ctl, &p->slot_h, &p->vj};
  for (DevBuf* b : bufs) b->release();
  p->mlp_tab.release(); p->slot_model.release(); p->slot_of.release();
  for (ampc_handle* mh : p->models) handle_release(mh);
  for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
  if (p->poll_host) (void)hipHostFree(p->poll_host);
  for (hipEvent_t e : p->poll_ev) if (e) (void)hipEventDestroy(e);
  ampc_handle* h = p->h;
  delete p;
  handle_release(h);
  return 0;
}

template <typename T>
static int ilqr_solve_impl(ampc_ilqr_plan* p, const double* x0, const double* uguess, int max_iter,
                           double* states, double* ctrls, double* Ks, double* ks, int* converged,
                           int* iters, int* status, double* objective) {
  ampc_handle* h = p->h;
  const int nx = h->nx, nu = h->nu, B = p->B, H = p->H;
  // states[:, 0, :] = x0 ; ctrls = uguess
  std::vector<double> st((size_t)B * (H + 1) * nx, 0.0);
  for (int b = 0; b < B; ++b) std::memcpy(&st[(size_t)b * (H + 1) * nx], x0 + (size_t)b * nx, nx * 8);
  HIP_OK(upload_converted<T>(p->states.p, st.data(), st.size(), h->stream));
  HIP_OK(upload_converted<T>(p->ctrls.p, uguess, (size_t)B * H * nu, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (int rc = ilqr_launch_iter<T>(p, 0)) return rc;        // rollout of the guess + objective
  if (int rc = ilqr_refresh_jacobians<T>(p)) return rc;
  std::vector<int> flags(7 * B);
  HIP_OK(hipMemsetAsync((int*)p->flags.p + 5 * B, 0, (size_t)4 * B * sizeof(int), h->stream));   // ls_rows, ls_count, ls_pass, ls_need
  p->ls_rb_now = 1;
  if (!p->poll_host) HIP_OK(hipHostMalloc((void**)&p->poll_host, (size_t)4 * B * sizeof(int), hipHostMallocDefault));
  if (!p->poll_ev[0])
    for (int i = 0; i < 2; ++i) HIP_OK(hipEventCreateWithFlags(&p->poll_ev[i], hipEventDisableTiming));
  // Iterations are queued in batches of kPoll; the `active` flags of a batch are copied out behind
  // it and inspected only after the NEXT batch has been queued, so the stream never drains while the
  // host decides.  Iterations queued past convergence are no-ops (retired problems exit at once).
  constexpr int kPoll = 4;
  int it = 0, batch = 0, pending = -1;     // pending: batch whose flags are in flight
  bool done = false;
  p->active_hint = B;
  // (ev_cur points into p->ev: never leave it set behind an early return)
  struct EvGuard { ampc_ilqr_plan* p; ~EvGuard() { p->ev_cur = nullptr; } } ev_guard{p};
  const size_t ev_first = p->ev_used / 5;  // this solve's first timed iteration
  while (it < max_iter && !done) {
    const int n = std::min(kPoll, max_iter - it);
    for (int k = 0; k < n; ++k) {
      if (p->timing) {
        if (p->ev_used + 5 > p->ev.size())
          for (int i = 0; i < 5; ++i) {
            hipEvent_t x;
            HIP_OK(hipEventCreate(&x));
            p->ev.push_back(x);
          }
        p->ev_cur = &p->ev[p->ev_used];
        p->ev_used += 5;
      }
      int rc = ilqr_launch_iter<T>(p, 1);                     // backward sweep + line search + accept
      if (rc == 0) rc = ilqr_refresh_jacobians<T>(p);
      p->ev_cur = nullptr;
      if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
    }
    it += n;
    const int slot = batch & 1;
    HIP_OK(hipMemcpyAsync(p->poll_host + (size_t)slot * 2 * B, (const int*)p->flags.p + B, (size_t)B * sizeof(int),
                          hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(p->poll_host + (size_t)slot * 2 * B + B, (const int*)p->flags.p + 8 * B,
                          (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipEventRecord(p->poll_ev[slot], h->stream));
    if (pending >= 0) {
      const int ps = pending & 1;
      HIP_OK(hipEventSynchronize(p->poll_ev[ps]));
      const int* pa = p->poll_host + (size_t)ps * 2 * B;
      int live = 0;
      for (int b = 0; b < B; ++b) live += pa[b] != 0;
      p->ls_rb_now = ls_rb_from_poll(pa, pa + B, B);
      p->active_hint = live;          // (as of two batches ago: an upper bound of the current count)
      if (live == 0) done = true;
    }
    pending = batch++;
  }
  p->last_iterations = it;
  HIP_OK(hipMemcpyAsync(flags.data(), p->flags.p, flags.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (converged) std::memcpy(converged, flags.data(), B * sizeof(int));
  if (iters) std::memcpy(iters, flags.data() + 2 * B, B * sizeof(int));
  if (status) std::memcpy(status, flags.data() + 3 * B, B * sizeof(int));
  p->last_ls_rows = 0;
  for (int b = 0; b < B; ++b) p->last_ls_rows += flags[5 * B + b];
  p->last_effective = 0;
  for (int b = 0; b < B; ++b) p->last_effective = std::max(p->last_effective, flags[2 * B + b]);
  if (p->timing) {           // iterations past the last one any problem performed were no-ops
    p->ev_live.resize(p->ev_used / 5, 1);
    for (size_t i = ev_first + (size_t)p->last_effective; i < p->ev_live.size(); ++i) p->ev_live[i] = 0;
  }
  if (states) HIP_OK(download_converted<T>(states, p->states.p, (size_t)B * (H + 1) * nx, h->stream));
  if (ctrls) HIP_OK(download_converted<T>(ctrls, p->ctrls.p, (size_t)B * H * nu, h->stream));
  if (Ks) HIP_OK(download_converted<T>(Ks, p->Ks.p, (size_t)B * H * nu * nx, h->stream));
  if (ks) HIP_OK(download_converted<T>(ks, p->ks.p, (size_t)B * H * nu, h->stream));
  if (objective) HIP_OK(download_converted<T>(objective, p->obj.p, (size_t)B, h->stream));
  return 0;
}

extern "C" int ampc_ilqr_solve(ampc_ilqr_plan* p, const double* x0, const double* uguess,
                               int max_iter, double* states, double* ctrls, double* Ks, double* ks,
                               int* converged, int* iters, int* status, double* objective) {
  REQUIRE(p && x0 && uguess, "ampc_ilqr_solve: NULL argument");
  REQUIRE(max_iter >= 0, "ampc_ilqr_solve: max_iter < 0");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? ilqr_solve_impl<double>(p, x0, uguess, max_iter, states, ctrls, Ks, ks, converged, iters, status, objective)
             : ilqr_solve_impl<float>(p, x0, uguess, max_iter, states, ctrls, Ks, ks, converged, iters, status, objective);
}

// ---------------------------------------------------------------------------------------------
// More slots than CUs (a finite batch admitted at once, or a wide evaluator plan): the kernels of an iteration
// take their slot through ilqr_slot_of -- the slots with work first -- rebuilt by one small launch per iteration.
// MLP models only (the feature-library and wide-linear kernels keep workgroup b = slot b).
static bool ilqr_wants_compaction(const ampc_ilqr_plan* p) {
  const int force = env_int("AMPC_ILQR_COMPACT", -1);
  if (force == 0 || !p->h->has_mlp || p->h->has_sindy || p->h->has_lin) return false;
  return force == 1 || p->B > p->h->n_cus;
}
template <typename T> static int ilqr_compact(ampc_ilqr_plan* p) {
  if (!p->compact_on) return 0;
  IlqrArgs<T> a = make_ilqr_args<T>(p, 1);
  hipLaunchKernelGGL(ilqr_compact_kernel<T>, dim3(1), dim3(256), 0, p->h->stream, a, p->B, (int*)p->slot_of.p);
  HIP_OK(hipGetLastError());
  return 0;
}

// Continuous batching: P problems through the plan's B slots (ilqr_queue_refill_kernel)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int ilqr_solve_queue_impl(ampc_ilqr_plan* p, int P, const double* x0, const double* uguess,
                                 const int* cost_index, const int* horizon, const int* model_index, int max_iter,
                                 double* states, double* ctrls, double* Ks, double* ks, int* converged, int* iters,
                                 int* status, double* objective) {
  ampc_handle* h = p->h;
  const int nx = h->nx, nu = h->nu, B = p->B, H = p->H;
  const size_t e = sizeof(T);
  HIP_OK(p->q_ctl.reserve((size_t)(2 + 2 * B) * sizeof(int)));
  HIP_OK(p->q_x0.reserve((size_t)P * nx * e));
  HIP_OK(p->q_u.reserve((size_t)P * H * nu * e));
  HIP_OK(p->q_cost.reserve((size_t)3 * P * sizeof(int)));               // cost block [P], horizon [P], model [P]
  HIP_OK(p->q_states.reserve((size_t)P * (H + 1) * nx * e));
  HIP_OK(p->q_ctrls.reserve((size_t)P * H * nu * e));
  HIP_OK(p->q_Ks.reserve((size_t)P * H * nu * nx * e));
  HIP_OK(p->q_ks.reserve((size_t)P * H * nu * e));
  HIP_OK(p->q_obj.reserve((size_t)P * e));
  HIP_OK(p->q_flags.reserve((size_t)4 * P * sizeof(int)));
  std::vector<int> ctl(2 + 2 * B, 0), cost(P, 0);
  for (int b = 0; b < B; ++b) { ctl[2 + b] = -1; ctl[2 + B + b] = 1; }       // no problem; mode "iterate"
  if (cost_index) std::memcpy(cost.data(), cost_index, (size_t)P * sizeof(int));
  HIP_OK(hipMemcpyAsync(p->q_ctl.p, ctl.data(), ctl.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(p->q_cost.p, cost.data(), (size_t)P * sizeof(int), hipMemcpyHostToDevice, h->stream));
  std::vector<int> slot_h0;
  if (horizon) {
    slot_h0.assign(B, H);
    HIP_OK(p->slot_h.reserve((size_t)B * sizeof(int)));
    HIP_OK(hipMemcpyAsync(p->slot_h.p, slot_h0.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipMemcpyAsync((int*)p->q_cost.p + P, horizon, (size_t)P * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  if (model_index) {
    HIP_OK(p->slot_model.reserve((size_t)B * sizeof(int)));
    HIP_OK(hipMemsetAsync(p->slot_model.p, 0, (size_t)B * sizeof(int), h->stream));
    HIP_OK(hipMemcpyAsync((int*)p->q_cost.p + 2 * P, model_index, (size_t)P * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  HIP_OK(upload_converted<T>(p->q_x0.p, x0, (size_t)P * nx, h->stream));
  if (uguess) HIP_OK(upload_converted<T>(p->q_u.p, uguess, (size_t)P * H * nu, h->stream));
  else HIP_OK(hipMemsetAsync(p->q_u.p, 0, (size_t)P * H * nu * e, h->stream));
  HIP_OK(hipMemsetAsync(p->flags.p, 0, (size_t)9 * B * sizeof(int), h->stream));     // every slot idle
  HIP_OK(hipMemsetAsync(p->states.p, 0, (size_t)B * (H + 1) * nx * e, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  const int npoll = 2 * B + 2;               // active[B], ls_need[B], the queue's two counters
  p->ls_rb_now = 1;
  if (p->poll_host) { (void)hipHostFree(p->poll_host); p->poll_host = nullptr; }
  HIP_OK(hipHostMalloc((void**)&p->poll_host, (size_t)2 * npoll * sizeof(int), hipHostMallocDefault));
  if (!p->poll_ev[0])
    for (int i = 0; i < 2; ++i) HIP_OK(hipEventCreateWithFlags(&p->poll_ev[i], hipEventDisableTiming));
  struct Guard {
    ampc_ilqr_plan* p;
    // (every exit path: no copy into the pinned poll buffer may still be in flight when it is freed, and
    //  the slots' cost blocks are the plan's own again, as ampc_ilqr_solve expects them)
    ~Guard() {
      (void)hipStreamSynchronize(p->h->stream);
      (void)hipMemcpy(p->d_cost_idx.p, p->cost_idx.data(), (size_t)p->B * sizeof(int), hipMemcpyHostToDevice);
      p->queue_on = false; p->var_h = false; p->var_model = false; p->compact_on = false; p->ev_cur = nullptr;
      (void)hipHostFree(p->poll_host); p->poll_host = nullptr;
    }
  } guard{p};
  p->queue_on = true;
  p->var_h = horizon != nullptr;
  p->var_model = model_index != nullptr;
  p->queue_max_iter = max_iter;
  p->active_hint = B;
  p->compact_on = ilqr_wants_compaction(p);
  if (p->compact_on) HIP_OK(p->slot_of.reserve((size_t)(B + 1) * sizeof(int)));
  IlqrQueue<T> q;
  q.P = P; q.B = B; q.H = H; q.nx = nx; q.nu = nu;
  q.horizon = horizon ? (const int*)p->q_cost.p + P : nullptr; q.slot_h = (int*)p->slot_h.p;
  q.model = model_index ? (const int*)p->q_cost.p + 2 * P : nullptr; q.slot_model = (int*)p->slot_model.p;
  q.ctl = (int*)p->q_ctl.p; q.slot_prob = q.ctl + 2;
  q.x0 = (const T*)p->q_x0.p; q.uguess = (const T*)p->q_u.p; q.cost = (const int*)p->q_cost.p;
  q.cost_idx = (int*)p->d_cost_idx.p;
  q.out_states = (T*)p->q_states.p; q.out_ctrls = (T*)p->q_ctrls.p; q.out_Ks = (T*)p->q_Ks.p;
  q.out_ks = (T*)p->q_ks.p; q.out_obj = (T*)p->q_obj.p; q.out_flags = (int*)p->q_flags.p;
  // An iteration = refill + sweep + line search (or the guess's rollout, per slot) + Jacobian refresh.
  // The queue's counters and the slots' `active` flags are copied out behind every batch of kPoll
  // iterations and read after the NEXT batch has been queued (the stream never drains); everything
  // is done when P problems have been harvested.  Upper bound on the iterations: every problem takes
  // at most max_iter + 1 slot-iterations (+1: the rollout of its guess).
  constexpr int kPoll = 4;
  const long long bound = ((long long)(P + B - 1) / B) * (max_iter + 2LL) + (long long)P + 4 * kPoll;
  long long it = 0;
  int batch = 0, pending = -1;
  bool done = false;
  const size_t ev_first = p->ev_used / 5;
  int poll_now = kPoll;                      // (2 once every problem has been handed out and the grids follow the
  while (!done && it < bound) {             //  count of slots with work, compact_on: the count is then 4 iterations old)
    for (int k = 0; k < poll_now; ++k) {
      IlqrArgs<T> a = make_ilqr_args<T>(p, 1);
      hipLaunchKernelGGL(ilqr_queue_refill_kernel<T>, dim3(B), dim3(256), 0, h->stream, a, q);
      HIP_OK(hipGetLastError());
      if (int rc = ilqr_compact<T>(p)) return rc;
      if (p->timing && (it + k) % p->timing_stride == 0) {
        if (p->ev_used + 5 > p->ev.size())
          for (int i = 0; i < 5; ++i) {
            hipEvent_t x;
            HIP_OK(hipEventCreate(&x));
            p->ev.push_back(x);
          }
        p->ev_cur = &p->ev[p->ev_used];
        p->ev_used += 5;
      }
      int rc = ilqr_launch_iter<T>(p, 1);
      if (rc == 0) rc = ilqr_refresh_jacobians<T>(p);
      p->ev_cur = nullptr;
      if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
    }
    it += poll_now;
    const int slot = batch & 1;
    int* ph = p->poll_host + (size_t)slot * npoll;
    HIP_OK(hipMemcpyAsync(ph, (const int*)p->flags.p + B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(ph + B, (const int*)p->flags.p + 8 * B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(ph + 2 * B, p->q_ctl.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipEventRecord(p->poll_ev[slot], h->stream));
    if (pending >= 0) {
      const int* pp = p->poll_host + (size_t)(pending & 1) * npoll;
      HIP_OK(hipEventSynchronize(p->poll_ev[pending & 1]));
      int live = 0;
      for (int b = 0; b < B; ++b) live += pp[b] != 0;
      p->ls_rb_now = ls_rb_from_poll(pp, pp + B, B);
      // while the queue still holds problems every slot is (about to be) busy
      p->active_hint = pp[2 * B] < P ? B : std::max(live, 1);
      if (p->compact_on && pp[2 * B] >= P) poll_now = 2;
      if (pp[2 * B + 1] >= P) done = true;
    }
    pending = batch++;
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  p->last_queue_launches = it;
  p->last_iterations = (int)std::min<long long>(it, 1 << 30);
  int fin[2] = {0, 0};
  HIP_OK(hipMemcpy(fin, p->q_ctl.p, sizeof(fin), hipMemcpyDeviceToHost));
  if (fin[1] < P) return fail("ampc_ilqr_solve_queue: internal: the queue did not drain");
  std::vector<int> fl((size_t)4 * P);
  HIP_OK(hipMemcpy(fl.data(), p->q_flags.p, fl.size() * sizeof(int), hipMemcpyDeviceToHost));
  p->last_ls_rows = 0;
  p->last_effective = (int)std::min<long long>(it, 1 << 30);
  for (int j = 0; j < P; ++j) {
    if (converged) converged[j] = fl[4 * j];
    if (iters) iters[j] = fl[4 * j + 1];
    if (status) status[j] = fl[4 * j + 2];
    p->last_ls_rows += fl[4 * j + 3];
  }
  if (p->timing) p->ev_live.resize(p->ev_used / 5, 1);
  (void)ev_first;
  if (states) HIP_OK(download_converted<T>(states, p->q_states.p, (size_t)P * (H + 1) * nx, h->stream));
  if (ctrls) HIP_OK(download_converted<T>(ctrls, p->q_ctrls.p, (size_t)P * H * nu, h->stream));
  if (Ks) HIP_OK(download_converted<T>(Ks, p->q_Ks.p, (size_t)P * H * nu * nx, h->stream));
  if (ks) HIP_OK(download_converted<T>(ks, p->q_ks.p, (size_t)P * H * nu, h->stream));
  if (objective) HIP_OK(download_converted<T>(objective, p->q_obj.p, (size_t)P, h->stream));
  return 0;      // (Guard: the slots' cost blocks as given at plan creation)
}

static int check_horizons(const ampc_ilqr_plan* p, int n, const int* horizon, const char* who) {
  if (horizon)
    for (int j = 0; j < n; ++j)
      REQUIRE(horizon[j] >= 1 && horizon[j] <= p->H, std::string(who) + ": horizons must lie in [1, the plan's horizon]");
  return 0;
}

static int check_models(const ampc_ilqr_plan* p, int n, const int* model_index, const char* who) {
  if (model_index) {
    REQUIRE(!p->models.empty(), std::string(who) + ": model_index given but the plan has no model table (ampc_ilqr_plan_set_models)");
    for (int j = 0; j < n; ++j)
      REQUIRE(model_index[j] >= 0 && model_index[j] < (int)p->models.size(), std::string(who) + ": bad model_index");
  }
  return 0;
}

extern "C" int ampc_ilqr_plan_set_models(ampc_ilqr_plan* p, int n_models, ampc_handle* const* models) {
  REQUIRE(p, "ampc_ilqr_plan_set_models: NULL plan");
  HIP_OK(hipSetDevice(p->h->device));
  if (n_models == 0) {
    HIP_OK(hipStreamSynchronize(p->h->stream));
    for (ampc_handle* mh : p->models) handle_release(mh);
    p->models.clear();
    return 0;
  }
  REQUIRE(n_models >= 1 && models, "ampc_ilqr_plan_set_models: NULL argument");
  for (int i = 0; i < n_models; ++i)
    if (int rc = check_same_shape(p->h, models[i], "ampc_ilqr_plan_set_models")) return rc;
  if (p->static_shape < 0) {
    const bool ready = jit::eligible(p->h) && (p->h->precision == AMPC_F64 ? jit::get<double>(p->h, true) != nullptr
                                                                           : jit::get<float>(p->h, true) != nullptr);
    if (ready) {
      HIP_OK(hipStreamSynchronize(p->h->stream));
      if (int rc = p->h->precision == AMPC_F64 ? ilqr_plan_build<double>(p) : ilqr_plan_build<float>(p)) return rc;
    }
    REQUIRE(p->static_shape >= 0, std::string("ampc_ilqr_plan_set_models") + kNeedStatic);
  }
  return p->h->precision == AMPC_F64 ? build_model_table<double>(p->h, n_models, models, &p->mlp_tab, &p->models)
                                     : build_model_table<float>(p->h, n_models, models, &p->mlp_tab, &p->models);
}

extern "C" int ampc_ilqr_solve_queue_var(ampc_ilqr_plan* p, int n_problems, const double* x0, const double* uguess,
                                         const int* cost_index, const int* horizon, const int* model_index,
                                         int max_iter, double* states, double* ctrls, double* Ks, double* ks,
                                         int* converged, int* iters, int* status, double* objective) {
  REQUIRE(p && x0, "ampc_ilqr_solve_queue_var: NULL argument");
  REQUIRE(n_problems >= 1, "ampc_ilqr_solve_queue_var: n_problems < 1");
  REQUIRE(max_iter >= 1, "ampc_ilqr_solve_queue_var: max_iter < 1");
  if (cost_index)
    for (int j = 0; j < n_problems; ++j)
      REQUIRE(cost_index[j] >= 0 && cost_index[j] < p->h->n_costs, "ampc_ilqr_solve_queue_var: bad cost_index");
  if (int rc = check_horizons(p, n_problems, horizon, "ampc_ilqr_solve_queue_var")) return rc;
  if (int rc = check_models(p, n_problems, model_index, "ampc_ilqr_solve_queue_var")) return rc;
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? ilqr_solve_queue_impl<double>(p, n_problems, x0, uguess, cost_index, horizon, model_index, max_iter, states,
                                             ctrls, Ks, ks, converged, iters, status, objective)
             : ilqr_solve_queue_impl<float>(p, n_problems, x0, uguess, cost_index, horizon, model_index, max_iter, states,
                                            ctrls, Ks, ks, converged, iters, status, objective);
}

extern "C" int ampc_ilqr_solve_queue(ampc_ilqr_plan* p, int n_problems, const double* x0, const double* uguess,
                                     const int* cost_index, int max_iter, double* states, double* ctrls,
                                     double* Ks, double* ks, int* converged, int* iters, int* status,
                                     double* objective) {
  REQUIRE(p && x0, "ampc_ilqr_solve_queue: NULL argument");
  REQUIRE(n_problems >= 1, "ampc_ilqr_solve_queue: n_problems < 1");
  REQUIRE(max_iter >= 1, "ampc_ilqr_solve_queue: max_iter < 1");
  if (cost_index)
    for (int j = 0; j < n_problems; ++j)
      REQUIRE(cost_index[j] >= 0 && cost_index[j] < p->h->n_costs, "ampc_ilqr_solve_queue: bad cost_index");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? ilqr_solve_queue_impl<double>(p, n_problems, x0, uguess, cost_index, nullptr, nullptr, max_iter, states, ctrls,
                                             Ks, ks, converged, iters, status, objective)
             : ilqr_solve_queue_impl<float>(p, n_problems, x0, uguess, cost_index, nullptr, nullptr, max_iter, states, ctrls,
                                            Ks, ks, converged, iters, status, objective);
}

// ---------------------------------------------------------------------------------------------
// Device-resident closed loops of iLQR controllers (ilqr_chain_*_kernel)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int ilqr_closed_loop_impl(ampc_ilqr_plan* p, ampc_handle* sur, int C, const double* init_obs,
                                 const int* cost_index, const int* horizon, const int* model_index, int n_steps,
                                 int max_iter, double* traj_obs,
                                 double* traj_ctrls, int* failed, int* steps_done, long long* iterations) {
  ampc_handle* h = p->h;
  const int nx = h->nx, nu = h->nu, B = p->B, H = p->H, T1 = n_steps + 1;
  const size_t e = sizeof(T);
  // ints: ctl[2] | slot_mode[B] (where make_ilqr_args expects it: q_ctl + 2 + B) ...
  HIP_OK(p->q_ctl.reserve((size_t)(2 + 2 * B) * sizeof(int)));
  HIP_OK(p->c_ints.reserve((size_t)(2 * B + 5 * C) * sizeof(int)));      // need[B] slot_chain[B] chain_t[C] chain_fail[C] cost[C] horizon[C] model[C]
  HIP_OK(p->c_iters.reserve((size_t)C * sizeof(long long)));
  HIP_OK(p->c_stage.reserve((size_t)B * (2 * nx + nu) * e));
  HIP_OK(p->c_obs.reserve((size_t)C * T1 * nx * e));
  HIP_OK(p->c_ctl.reserve((size_t)C * T1 * nu * e));
  HIP_OK(p->q_x0.reserve((size_t)C * nx * e));
  std::vector<int> ctl(2 + 2 * B, 0), ci(2 * B + 5 * C, 0), slot_h0(B, H);
  for (int b = 0; b < B; ++b) { ctl[2 + b] = -1; ctl[2 + B + b] = 1; ci[B + b] = -1; }
  if (cost_index) std::memcpy(ci.data() + 2 * B + 2 * C, cost_index, (size_t)C * sizeof(int));
  if (horizon) {
    std::memcpy(ci.data() + 2 * B + 3 * C, horizon, (size_t)C * sizeof(int));
    HIP_OK(p->slot_h.reserve((size_t)B * sizeof(int)));
    HIP_OK(hipMemcpyAsync(p->slot_h.p, slot_h0.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  if (model_index) {
    std::memcpy(ci.data() + 2 * B + 4 * C, model_index, (size_t)C * sizeof(int));
    HIP_OK(p->slot_model.reserve((size_t)B * sizeof(int)));
    HIP_OK(hipMemsetAsync(p->slot_model.p, 0, (size_t)B * sizeof(int), h->stream));
  }
  HIP_OK(hipMemcpyAsync(p->q_ctl.p, ctl.data(), ctl.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(p->c_ints.p, ci.data(), ci.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemsetAsync(p->c_iters.p, 0, (size_t)C * sizeof(long long), h->stream));
  HIP_OK(hipMemsetAsync(p->c_obs.p, 0, (size_t)C * T1 * nx * e, h->stream));
  HIP_OK(hipMemsetAsync(p->c_ctl.p, 0, (size_t)C * T1 * nu * e, h->stream));
  HIP_OK(hipMemsetAsync(p->c_stage.p, 0, (size_t)B * (2 * nx + nu) * e, h->stream));
  HIP_OK(upload_converted<T>(p->q_x0.p, init_obs, (size_t)C * nx, h->stream));
  HIP_OK(hipMemsetAsync(p->flags.p, 0, (size_t)9 * B * sizeof(int), h->stream));
  HIP_OK(hipMemsetAsync(p->states.p, 0, (size_t)B * (H + 1) * nx * e, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (p->poll_host) { (void)hipHostFree(p->poll_host); p->poll_host = nullptr; }
  const int npoll = 2 * B + 3;               // active[B], ls_need[B], the chains' two counters, slots with work
  HIP_OK(hipHostMalloc((void**)&p->poll_host, (size_t)2 * npoll * sizeof(int), hipHostMallocDefault));
  if (!p->poll_ev[0])
    for (int i = 0; i < 2; ++i) HIP_OK(hipEventCreateWithFlags(&p->poll_ev[i], hipEventDisableTiming));
  struct Guard {
    ampc_ilqr_plan* p;
    ~Guard() {
      (void)hipStreamSynchronize(p->h->stream);
      (void)hipMemcpy(p->d_cost_idx.p, p->cost_idx.data(), (size_t)p->B * sizeof(int), hipMemcpyHostToDevice);
      p->queue_on = false; p->var_h = false; p->var_model = false; p->compact_on = false; p->ev_cur = nullptr;
      (void)hipHostFree(p->poll_host); p->poll_host = nullptr;
    }
  } guard{p};
  p->queue_on = true;
  p->var_h = horizon != nullptr;
  p->var_model = model_index != nullptr;
  p->queue_max_iter = max_iter;
  p->active_hint = B;
  p->ls_rb_now = 1;
  p->compact_on = ilqr_wants_compaction(p);
  if (p->compact_on) HIP_OK(p->slot_of.reserve((size_t)(B + 1) * sizeof(int)));
  IlqrChains<T> q;
  q.C = C; q.B = B; q.H = H; q.nx = nx; q.nu = nu; q.n_steps = n_steps; q.max_iter = max_iter;
  q.ctl = (int*)p->q_ctl.p;
  int* ints = (int*)p->c_ints.p;
  q.need = ints; q.slot_chain = ints + B; q.chain_t = ints + 2 * B; q.chain_fail = ints + 2 * B + C;
  q.cost = ints + 2 * B + 2 * C;
  q.horizon = horizon ? ints + 2 * B + 3 * C : nullptr; q.slot_h = (int*)p->slot_h.p;
  q.model = model_index ? ints + 2 * B + 4 * C : nullptr; q.slot_model = (int*)p->slot_model.p;
  q.chain_iters = (long long*)p->c_iters.p;
  q.x0 = (const T*)p->q_x0.p; q.cost_idx = (int*)p->d_cost_idx.p;
  q.stage_x = (T*)p->c_stage.p; q.stage_u = q.stage_x + (size_t)B * nx; q.stage_next = q.stage_u + (size_t)B * nu;
  q.traj_obs = (T*)p->c_obs.p; q.traj_ctrls = (T*)p->c_ctl.p;
  // Upper bound on the plan iterations: every control step of every chain takes at most max_iter + 2.
  constexpr int kPoll = 4;
  const long long bound = ((long long)(C + B - 1) / B) * n_steps * (max_iter + 2LL) + (long long)C + 4 * kPoll;
  long long it = 0;
  int batch = 0, pending = -1;
  bool done = false;
  int poll_now = kPoll;                      // (2 once every chain has been handed out: the grids follow the count of
  while (!done && it < bound) {             //  slots whose chain is unfinished, compact_on)
    for (int k = 0; k < poll_now; ++k) {
      IlqrArgs<T> a = make_ilqr_args<T>(p, 1);
      hipLaunchKernelGGL(ilqr_chain_pre_kernel<T>, dim3(B), dim3(64), 0, h->stream, a, q);
      HIP_OK(hipGetLastError());
      if (int rc = surrogate_step<T>(h, sur, q.stage_x, q.stage_u, q.stage_next, B)) return rc;
      hipLaunchKernelGGL(ilqr_chain_post_kernel<T>, dim3(B), dim3(256), 0, h->stream, a, q);
      HIP_OK(hipGetLastError());
      if (int rc = ilqr_compact<T>(p)) return rc;
      int rc = ilqr_launch_iter<T>(p, 1);
      if (rc == 0) rc = ilqr_refresh_jacobians<T>(p);
      if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
    }
    it += poll_now;
    const int slot = batch & 1;
    int* ph = p->poll_host + (size_t)slot * npoll;
    HIP_OK(hipMemcpyAsync(ph, (const int*)p->flags.p + B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(ph + B, (const int*)p->flags.p + 8 * B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(ph + 2 * B, p->q_ctl.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (p->compact_on)
      HIP_OK(hipMemcpyAsync(ph + 2 * B + 2, (const int*)p->slot_of.p + B, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipEventRecord(p->poll_ev[slot], h->stream));
    if (pending >= 0) {
      const int* pp = p->poll_host + (size_t)(pending & 1) * npoll;
      HIP_OK(hipEventSynchronize(p->poll_ev[pending & 1]));
      p->ls_rb_now = ls_rb_from_poll(pp, pp + B, B);
      // every chain handed out: the slots with work (an unfinished chain: solving, or its next control step just
      // loaded -- counted by ilqr_compact_kernel) can only become fewer
      if (p->compact_on && pp[2 * B] >= C) {
        p->active_hint = std::min(B, std::max(pp[2 * B + 2], 1));
        poll_now = 2;
      }
      if (pp[2 * B + 1] >= C) done = true;
    }
    pending = batch++;
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  p->last_queue_launches = it;
  p->last_iterations = (int)std::min<long long>(it, 1 << 30);
  int fin[2] = {0, 0};
  HIP_OK(hipMemcpy(fin, p->q_ctl.p, sizeof(fin), hipMemcpyDeviceToHost));
  if (fin[1] < C) return fail("ampc_ilqr_closed_loop: internal: the chains did not finish");
  std::vector<int> back((size_t)2 * B + 3 * C);
  HIP_OK(hipMemcpy(back.data(), p->c_ints.p, back.size() * sizeof(int), hipMemcpyDeviceToHost));
  if (steps_done) std::memcpy(steps_done, back.data() + 2 * B, (size_t)C * sizeof(int));
  if (failed) std::memcpy(failed, back.data() + 2 * B + C, (size_t)C * sizeof(int));
  if (iterations) HIP_OK(hipMemcpy(iterations, p->c_iters.p, (size_t)C * sizeof(long long), hipMemcpyDeviceToHost));
  if (traj_obs) HIP_OK(download_converted<T>(traj_obs, p->c_obs.p, (size_t)C * T1 * nx, h->stream));
  if (traj_ctrls) HIP_OK(download_converted<T>(traj_ctrls, p->c_ctl.p, (size_t)C * T1 * nu, h->stream));
  return 0;
}

extern "C" int ampc_ilqr_closed_loop(ampc_ilqr_plan* p, ampc_handle* surrogate, int n_chains, const double* init_obs,
                                     const int* cost_index, int n_steps, int max_iter, double* traj_obs,
                                     double* traj_ctrls, int* failed, int* steps_done, long long* iterations) {
  return ampc_ilqr_closed_loop_var(p, surrogate, n_chains, init_obs, cost_index, nullptr, nullptr, n_steps, max_iter,
                                   traj_obs, traj_ctrls, failed, steps_done, iterations);
}

extern "C" int ampc_ilqr_closed_loop_var(ampc_ilqr_plan* p, ampc_handle* surrogate, int n_chains, const double* init_obs,
                                         const int* cost_index, const int* horizon, const int* model_index,
                                         int n_steps, int max_iter, double* traj_obs, double* traj_ctrls, int* failed,
                                         int* steps_done, long long* iterations) {
  REQUIRE(p && init_obs, "ampc_ilqr_closed_loop: NULL argument");
  if (int rc = check_horizons(p, n_chains, horizon, "ampc_ilqr_closed_loop_var")) return rc;
  if (int rc = check_models(p, n_chains, model_index, "ampc_ilqr_closed_loop_var")) return rc;
  REQUIRE(n_chains >= 1 && n_steps >= 1 && max_iter >= 1, "ampc_ilqr_closed_loop: n_chains, n_steps, max_iter must be >= 1");
  ampc_handle* sur = surrogate ? surrogate : p->h;
  REQUIRE(sur->has_model() && sur->nx == p->h->nx && sur->nu == p->h->nu,
          "ampc_ilqr_closed_loop: surrogate model must have the controller model's dimensions");
  REQUIRE(sur->precision == p->h->precision && sur->device == p->h->device,
          "ampc_ilqr_closed_loop: surrogate must share the plan's device and precision");
  if (cost_index)
    for (int j = 0; j < n_chains; ++j)
      REQUIRE(cost_index[j] >= 0 && cost_index[j] < p->h->n_costs, "ampc_ilqr_closed_loop: bad cost_index");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? ilqr_closed_loop_impl<double>(p, sur, n_chains, init_obs, cost_index, horizon, model_index, n_steps, max_iter,
                                             traj_obs, traj_ctrls, failed, steps_done, iterations)
             : ilqr_closed_loop_impl<float>(p, sur, n_chains, init_obs, cost_index, horizon, model_index, n_steps, max_iter,
                                            traj_obs, traj_ctrls, failed, steps_done, iterations);
}
