// api_mppi.cpp -- MPPI plans, noise (Philox / numpy's legacy stream on the device), one-call control steps,
// the device-resident MPPI closed loop and trajectory scoring
// Part of the C ABI of libautompc_hip.so (include/autompc_hip.h); see api.cpp for the map of the translation units.
// Built with hipcc for gfx950 only.
#include "host_common.hpp"
#include "legacy_rng_kernels.hpp"      // (non-template kernels: this translation unit only)
#include <link.h>                      // dl_iterate_phdr: the C library's log() tables (host_log_mode)
#include <atomic>
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

// Anything that rewrites a plan's noise buffer (or rebuilds the plan) drops a pre-drawn next call (mppi_run_impl);
// a draw that is still running is waited for (it writes buffers the caller is about to reuse).
static inline void legacy_predraw_drop(ampc_mppi_plan* p) {
  p->lg_pre = false; p->u_in_pin = false;
  if (p->lg_pre_inflight) { (void)hipEventSynchronize(p->lg_pre_done); p->lg_pre_inflight = false; }
}

// Workgroup -> tile order of a plan whose problems run on several models: workgroups are dealt round-robin
// to the eight XCDs (blockIdx % 8), each with its own 4 MB L2, and the rollout streams its model's weights
// from L2 at every time step -- eight 2 x 256 f64 models (610 KB each) in flight on every XCD do not fit.
// The tiles (in the plan's longest-horizon-first order) are poured model by model into eight queues of
// equal length, queue x feeding the workgroups of XCD x: an XCD then sees one or two models.
static int build_tile_order(ampc_mppi_plan* p) {
  p->use_tile_order = false;
  int n_models = 0;
  for (int m : p->model_idx) n_models = std::max(n_models, m + 1);
  if (p->models.empty() || n_models < 2 || p->quad || p->h->has_sindy || p->h->has_lin) return 0;
  // MEASURED (tools/models_rate.py, 64 candidates x 8 models of 2 x 256, 199 control steps): the plan's plain
  // longest-horizon-first order 35 076 solves/s -- the one-model rate, 35 086: eight models' weights (4.9 MB)
  // stream from L2 / MALL without loss -- against 26 558 with this order, which gives up part of the
  // longest-first dispatch.  Kept behind the switch for batches with many more models; off by default.
  if (env_int("AMPC_MODEL_XCD", 0) == 0) return 0;
  // (a tile's run time is proportional to its horizon: the queues are filled to equal WORK, not equal length)
  const int n = p->n_tiles, nx = 8;
  double total = 0.0;
  for (int t = 0; t < n; ++t) total += p->H[p->tile_prob_host[t]];
  const double cap = total / nx;
  std::vector<std::vector<int>> queue(nx);
  int x = 0;
  double filled = 0.0;
  for (int m = 0; m < n_models; ++m)
    for (int t = 0; t < n; ++t)
      if (p->model_idx[p->tile_prob_host[t]] == m) {
        const double wt = p->H[p->tile_prob_host[t]];
        if (x < nx - 1 && filled + 0.5 * wt > cap * (x + 1)) ++x;
        queue[x].push_back(t);
        filled += wt;
      }
  // workgroup i runs on XCD i % 8: deal the queues round-robin; a queue that runs dry (they differ in
  // length, not in work) hands its turns to the longest remaining one
  std::vector<int> order;
  order.reserve(n);
  std::vector<size_t> pos(nx, 0);
  while ((int)order.size() < n)
    for (int q = 0; q < nx && (int)order.size() < n; ++q) {
      int src = q;
      if (pos[src] >= queue[src].size()) {
        size_t best = 0;
        for (int k = 0; k < nx; ++k)
          if (queue[k].size() - pos[k] > best) { best = queue[k].size() - pos[k]; src = k; }
      }
      order.push_back(queue[src][pos[src]++]);
    }
  HIP_OK(p->tile_order.reserve(order.size() * sizeof(int)));
  HIP_OK(hipMemcpy(p->tile_order.p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
  p->use_tile_order = true;
  return 0;
}

template <typename T> static int plan_build(ampc_mppi_plan* p) {
  ampc_handle* h = p->h;
  const MlpDev<T>& m = model_of<T>(h);
  const int nu = h->nu, nx = h->nx;
  const size_t extra = (size_t)p->max_h * nu + h->cost_stride + 3 * nu + 8;
  // Four-row tiles (mppi_rollout4.hpp) when sixteen-row tiles would leave most of the chip idle:
  // a rollout's time is its per-step latency, and a four-row step costs a quarter of the matrix
  // pipe.  AMPC_QUAD: -1 automatic, 0 never, 1 whenever the shape is supported.
  p->quad = false;
  p->eps_inline = false;
  if (sizeof(T) == 8 && h->has_mlp && q4_supported(m.hpad, m.n_hidden, m.nxp, m.k1p)) {
    long long tiles16 = 0;
    for (int b = 0; b < p->B; ++b) tiles16 += (p->N[b] + 15) / 16;
    const int mode = env_int("AMPC_QUAD", -1);
    const bool fits = make_q4_lds(nu, m.k1p, m.nxp, m.hpad, m.n_hidden, h->cost_stride, p->max_h).total *
                          sizeof(T) <= kLdsLimit;
    p->quad = fits && (p->forced_quad || (p->forced_mt == 0 && env_int("AMPC_MT", 0) == 0 &&
                                           (mode == 1 || (mode < 0 && tiles16 * 2 <= h->n_cus))));
  }
  REQUIRE(!p->forced_quad || p->quad,
          "ampc_mppi_plan_set_geometry: four-row tiles need an f64 MLP of hidden width <= 64 (<= 128 with at "
          "most two hidden layers) and at most 32 states");
  if (p->quad) {
    p->mt = 0;
  } else if (h->has_sindy) {
    p->mt = 4;
  } else if (h->has_lin) {
    REQUIRE(p->forced_mt <= 1, "ampc_mppi_plan_set_geometry: wide linear models roll out in 16-row tiles");
    p->mt = 1;
  } else {
    p->mt = choose_mt<T>(h, m, p->sum_n, extra, p->forced_mt);
    // a caller that fixed the tile height relies on it (summation orders, hence bit-identical
    // scores across batches, depend on it): refuse instead of quietly picking another one
    REQUIRE(p->forced_mt == 0 || p->mt == p->forced_mt,
            "ampc_mppi_plan_set_geometry: the requested tile_rows does not fit the 160 KB LDS for this "
            "model / horizon");
    // Problems of different horizons (tuning candidates): 64-row tiles are fewer, coarser work
    // items for the longest-first tile order to balance and leave no LDS for the fused update;
    // 32-row tiles measured 3 % faster on c5.  (AMPC_MT / set_geometry still override.)
    bool mixed = false;
    for (int b = 1; b < p->B; ++b) mixed = mixed || p->H[b] != p->H[0];
    if (mixed && p->mt > 2 && env_int("AMPC_MT", 0) == 0 && p->forced_mt == 0) p->mt = 2;
  }
  const int M = p->quad ? 4 : 16 * p->mt;
  p->tile_m = M;
  if (p->quad) {
    std::memset(&p->L, 0, sizeof(p->L));          // (the kernel derives its own map: make_q4_lds)
  } else if (h->has_sindy) {
    std::memset(&p->L, 0, sizeof(p->L));
    p->L.extra = (2 * nx + nu + h->s_ntab) * 64;   // per-thread columns: [x|u], next x, value table
  } else if (h->has_lin) {
    std::memset(&p->L, 0, sizeof(p->L));
    p->L.xu_stride = lin_xs(h->l_kp, (int)sizeof(T));
    p->L.extra = lin_lds_base(h->l_kp, (int)sizeof(T));         // [x | u] twice (ping-pong)
  } else {
    p->L = tile_lds_for<T>(h, m, M, extra);
  }
  // behind the tile map, fixed-size regions first (their offsets are compile-time constants of a
  // shape-specialised kernel): cost block + bounds, then the shifted sequence [max_h][nu]
  p->lds_cost = p->L.extra;
  p->lds_aseq = round_up(p->lds_cost + h->cost_stride + 3 * nu, 4);
  p->lds_bytes = ((size_t)p->lds_aseq + (size_t)p->max_h * nu) * sizeof(T);
  REQUIRE(p->lds_bytes <= kLdsLimit, "mppi plan: model + horizon do not fit the 160 KB LDS");
  if (p->quad) {
    p->lds_bytes = (size_t)make_q4_lds(nu, m.k1p, m.nxp, m.hpad, m.n_hidden, h->cost_stride, p->max_h).total * sizeof(T);
    p->lds_eps = env_int("AMPC_FUSED_UPDATE", 1) != 0 ? 0 : -1;       // (a flag here: the map has the region)
    p->lds_red = 0;
  } else
  {  // fused update: keep the tile's clipped noise [max_h][M][nu] (+ 2M reduction slots) in LDS
    const int e0 = round_up(p->lds_aseq + p->max_h * nu, 4);
    const size_t bytes = ((size_t)e0 + (size_t)p->max_h * M * nu + 2 * M) * sizeof(T);
    if (!h->has_sindy && bytes <= kLdsLimit && env_int("AMPC_FUSED_UPDATE", 1) != 0) {
      p->lds_eps = e0;
      p->lds_red = e0 + p->max_h * M * nu;
      p->lds_bytes = bytes;
    }
  }
  // shape-specialised kernel: registered shape, 16- or 32-row tile, and the LDS map the shape
  // implies (ping-pong activations + separate partials -- tile_lds_for picked it iff it fits)
  p->static_shape = -1;
  p->jit = nullptr;
  // (indicator cost terms live in the run-time-shape kernels only: mppi_kernels.hpp)
  const bool ext = h->n_ind > 0;
  if (ext) {
  } else if (p->quad && env_int("AMPC_STATIC", 1) != 0) {          // (the four-row kernel has one LDS map)
    int sid = static_shape_of<T>(h, m);
    if (sid < 0 && (p->jit = jit::get<T>(h)) != nullptr) sid = 0;
    p->static_shape = sid;
  } else if (!h->has_sindy && !h->has_lin && !p->quad && p->mt <= 2 && env_int("AMPC_STATIC", 1) != 0) {
    int sid = static_shape_of<T>(h, m);
    if (sid < 0 && (p->jit = jit::get<T>(h)) != nullptr) sid = 0;      // run-time compiled shape
    const int lv = sid >= 0 ? lds_variant_of<T>(m, p->L, M, h->nw) : -1;
    // instantiated: 16-row tiles with the richest map, 32-row tiles with any of the three
    if (lv == 0 || (lv > 0 && p->mt == 2)) { p->static_shape = sid; p->static_lv = lv; }
    else p->jit = nullptr;
  }
  std::vector<MppiProblem<T>> pr(p->B);
  std::vector<int> tile_prob;
  int tile = 0;
  // Workgroups are dispatched in blockIdx order and a tile's run time is proportional to its
  // horizon: hand out the longest-horizon problems first (LPT) so heterogeneous candidate batches
  // do not end on a tail of 30-step tiles.
  std::vector<int> order(p->B);
  for (int b = 0; b < p->B; ++b) order[b] = b;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p->H[a] > p->H[b]; });
  for (int oi = 0; oi < p->B; ++oi) {
    const int b = order[oi];
    MppiProblem<T>& q = pr[b];
    std::memset(&q, 0, sizeof(q));
    q.N = p->N[b]; q.H = p->H[b]; q.tile0 = tile; q.cost_idx = p->cost_idx[b];
    q.lam_over_sigma = (T)(p->lmda[b] / p->sigma[b]);
    q.neg_inv_lambda = (T)(-1.0 / p->lmda[b]);
    q.sqrt_sigma = (T)std::sqrt(p->sigma[b]);
    q.eps_off = p->eps_off[b]; q.epso_off = p->epso_off[b]; q.cost_off = p->cost_off[b];
    q.a_off = p->a_off[b];
    q.noise_id = p->noise_id[b];
    q.model = p->model_idx.empty() ? 0 : p->model_idx[b];
    const int nt = (q.N + M - 1) / M;
    for (int t = 0; t < nt; ++t) tile_prob.push_back(b);
    tile += nt;
  }
  p->n_tiles = tile;
  p->tile_prob_host = tile_prob;
  if (int rc = build_tile_order(p)) return rc;
  HIP_OK(p->probs.reserve(pr.size() * sizeof(MppiProblem<T>)));
  HIP_OK(hipMemcpy(p->probs.p, pr.data(), pr.size() * sizeof(MppiProblem<T>), hipMemcpyHostToDevice));
  HIP_OK(p->tile_prob.reserve(tile_prob.size() * sizeof(int)));
  HIP_OK(hipMemcpy(p->tile_prob.p, tile_prob.data(), tile_prob.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(p->x0.reserve((size_t)p->B * nx * sizeof(T)));
  for (int i = 0; i < 2; ++i) {
    HIP_OK(p->act[i].reserve((size_t)p->sum_hnu * sizeof(T)));
    HIP_OK(hipMemset(p->act[i].p, 0, (size_t)p->sum_hnu * sizeof(T)));
  }
  legacy_predraw_drop(p);
  HIP_OK(p->eps.reserve((size_t)p->sum_nhnu * sizeof(T)));
  HIP_OK(hipMemset(p->eps.p, 0, (size_t)p->sum_nhnu * sizeof(T)));
  HIP_OK(p->eps_out.reserve((size_t)p->sum_nhnu * sizeof(T)));
  HIP_OK(p->costs.reserve((size_t)p->sum_n * sizeof(T)));
  HIP_OK(p->term_last.reserve((size_t)p->B * sizeof(T)));
  HIP_OK(p->u_out.reserve((size_t)p->B * nu * sizeof(T)));
  HIP_OK(p->tile_stat.reserve((size_t)p->n_tiles * 2 * sizeof(T)));
  HIP_OK(p->tile_part.reserve((size_t)p->n_tiles * p->max_h * nu * sizeof(T)));
  HIP_OK(p->tile_done.reserve((size_t)p->B * sizeof(int)));
  HIP_OK(hipMemset(p->tile_done.p, 0, (size_t)p->B * sizeof(int)));
  // AMPC_FUSED_COMBINE = 1 (experiment, off: measured slower): four-row plans finish the update inside the rollout launch
  {
    int max_tiles = 0;
    for (int b = 0; b < p->B; ++b) max_tiles = std::max(max_tiles, (p->N[b] + M - 1) / M);
    p->fused_combine = p->quad && p->lds_eps >= 0 && env_int("AMPC_FUSED_COMBINE", 0) != 0 &&
                       (size_t)(2 * max_tiles + 256) * sizeof(T) <= p->lds_bytes;
  }
  HIP_OK(hipMemset(p->x0.p, 0, (size_t)p->B * nx * sizeof(T)));
  return 0;
}

extern "C" int ampc_mppi_plan_create(ampc_handle* h, int B, const int* num_path,
                                     const int* horizon, const double* sigma, const double* lmda,
                                     const int* cost_index, int term_mode, ampc_mppi_plan** out) {
  REQUIRE(h && num_path && horizon && sigma && lmda && out, "ampc_mppi_plan_create: NULL argument");
  REQUIRE(h->has_model() && h->n_costs > 0 && h->has_bounds,
          "ampc_mppi_plan_create: model, cost and control bounds must be set first");
  REQUIRE(B >= 1, "ampc_mppi_plan_create: B < 1");
  REQUIRE(term_mode == 0 || term_mode == 1, "ampc_mppi_plan_create: bad term_mode");
  for (int j = 0; j < h->nu; ++j)
    REQUIRE(std::isfinite(h->lo[j]) && std::isfinite(h->hi[j]) && h->hi[j] != 0.0,
            "MPPI requires finite, non-zero upper control bounds (ctrl_scale = umax)");
  HIP_OK(hipSetDevice(h->device));
  ampc_mppi_plan* p = new ampc_mppi_plan();
  p->h = h; p->B = B; p->term_mode = term_mode;
  h->refs++;
  const int nu = h->nu;
  for (int b = 0; b < B; ++b) {
    if (!(num_path[b] >= 1 && horizon[b] >= 2 && sigma[b] > 0 && lmda[b] > 0) ||
        (cost_index && (cost_index[b] < 0 || cost_index[b] >= h->n_costs))) {
      h->refs--;
      delete p;
      return fail("ampc_mppi_plan_create: need num_path>=1, horizon>=2, sigma>0, lmda>0, valid cost_index");
    }
    p->N.push_back(num_path[b]); p->H.push_back(horizon[b]);
    p->sigma.push_back(sigma[b]); p->lmda.push_back(lmda[b]);
    p->cost_idx.push_back(cost_index ? cost_index[b] : 0);
    p->noise_id.push_back((unsigned)b);
    p->a_off.push_back((int)p->sum_hnu);
    p->eps_off.push_back(p->sum_nhnu); p->epso_off.push_back(p->sum_nhnu);
    p->cost_off.push_back(p->sum_n);
    p->sum_n += num_path[b];
    p->sum_hnu += (long long)horizon[b] * nu;
    p->sum_nhnu += (long long)num_path[b] * horizon[b] * nu;
    p->max_h = horizon[b] > p->max_h ? horizon[b] : p->max_h;
  }
  int rc = h->precision == AMPC_F64 ? plan_build<double>(p) : plan_build<float>(p);
  if (rc) {
    ampc_mppi_plan_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}

extern "C" int ampc_mppi_plan_destroy(ampc_mppi_plan* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->h->device);
  (void)hipStreamSynchronize(p->h->stream);
  DevBuf* bufs[] = {&p->lift_prog, &p->probs, &p->tile_prob, &p->x0, &p->act[0], &p->act[1], &p->eps, &p->eps_next,
                    &p->eps_out, &p->costs, &p->term_last, &p->u_out, &p->tile_stat, &p->tile_part, &p->tile_done,
                    &p->lg_key[0], &p->lg_key[1], &p->lg_stream[0], &p->lg_stream[1], &p->lg_cnt,
                    &p->lg_scale, &p->lg_xraw, &p->lg_poly[0], &p->lg_poly[1], &p->lg_poly[2], &p->lg_poly[3],
                    &p->lg_win, &p->lg_logtab};
  // (a pre-drawn next call / the raw-stream run-ahead may still be running on their own streams: they write the
  //  pinned results and the buffers freed below)
  if (p->lg_draw) { (void)hipStreamSynchronize(p->lg_draw); (void)hipStreamDestroy(p->lg_draw); }
  if (p->lg_side) { (void)hipStreamSynchronize(p->lg_side); (void)hipStreamDestroy(p->lg_side); }
  if (p->lg_pre_done) (void)hipEventDestroy(p->lg_pre_done);
  p->eps_pre.release();
  p->cl_obs.release(); p->cl_ctl.release(); p->cl_next.release(); p->cl_sim.release();
  if (p->lg_pin) (void)hipHostFree(p->lg_pin);
  if (p->pin_x0) (void)hipHostFree(p->pin_x0);
  if (p->pin_u) (void)hipHostFree(p->pin_u);
  if (p->pin_flag) (void)hipHostFree(p->pin_flag);
  for (hipEvent_t e : p->lg_evs) if (e) (void)hipEventDestroy(e);
  if (p->lg_drawn) (void)hipEventDestroy(p->lg_drawn);
  for (DevBuf* b : bufs) b->release();
  p->mlp_tab.release(); p->tile_order.release();
  for (ampc_handle* mh : p->models) handle_release(mh);
  for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
  ampc_handle* h = p->h;
  delete p;
  handle_release(h);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// several controller models of one shape in a plan (tuning candidates that carry their own model)
// ---------------------------------------------------------------------------------------------
template <typename T> static int mppi_set_noise_ids_impl(ampc_mppi_plan* p);
// Several models in a plan run on the shape-specialised kernels (mlp_tile.hpp: plan_model): a registered
// shape, or the shape's run-time compiled plugin -- waited for here if it is still building.
template <typename T> static int plan_build(ampc_mppi_plan* p);
template <typename T> static int mppi_require_static(ampc_mppi_plan* p, const char* who) {
  if (p->static_shape >= 0) return 0;
  if (p->h->n_ind == 0 && jit::eligible(p->h) && jit::get<T>(p->h, true) != nullptr) {
    HIP_OK(hipStreamSynchronize(p->h->stream));
    if (int rc = plan_build<T>(p)) return rc;          // (the plugin is ready now: the rebuilt plan takes it)
  }
  REQUIRE(p->static_shape >= 0, std::string(who) + kNeedStatic);
  return 0;
}

extern "C" int ampc_mppi_plan_set_models(ampc_mppi_plan* p, int n_models, ampc_handle* const* models,
                                         const int* model_index) {
  REQUIRE(p, "ampc_mppi_plan_set_models: NULL plan");
  HIP_OK(hipSetDevice(p->h->device));
  if (n_models == 0) {                               // back to the handle's own model
    HIP_OK(hipStreamSynchronize(p->h->stream));
    for (ampc_handle* mh : p->models) handle_release(mh);
    p->models.clear();
    p->model_idx.clear();
    p->use_tile_order = false;
    return 0;
  }
  REQUIRE(n_models >= 1 && models && model_index, "ampc_mppi_plan_set_models: NULL argument");
  for (int i = 0; i < n_models; ++i)
    if (int rc = check_same_shape(p->h, models[i], "ampc_mppi_plan_set_models")) return rc;
  for (int b = 0; b < p->B; ++b)
    REQUIRE(model_index[b] >= 0 && model_index[b] < n_models, "ampc_mppi_plan_set_models: bad model_index");
  if (int rc = p->h->precision == AMPC_F64 ? mppi_require_static<double>(p, "ampc_mppi_plan_set_models")
                                           : mppi_require_static<float>(p, "ampc_mppi_plan_set_models")) return rc;
  if (int rc = p->h->precision == AMPC_F64 ? build_model_table<double>(p->h, n_models, models, &p->mlp_tab, &p->models)
                                           : build_model_table<float>(p->h, n_models, models, &p->mlp_tab, &p->models))
    return rc;
  p->model_idx.assign(model_index, model_index + p->B);
  HIP_OK(hipStreamSynchronize(p->h->stream));
  if (int rc = build_tile_order(p)) return rc;
  // the problems' descriptors carry the entry
  return p->h->precision == AMPC_F64 ? mppi_set_noise_ids_impl<double>(p) : mppi_set_noise_ids_impl<float>(p);
}

template <typename T>
static int mppi_upload_impl(ampc_mppi_plan* p, const double* x0, const double* act_seq,
                            const double* eps) {
  ampc_handle* h = p->h;
  if (x0) HIP_OK(upload_converted<T>(p->x0.p, x0, (size_t)p->B * h->nx, h->stream));
  if (act_seq) HIP_OK(upload_converted<T>(p->act[p->cur].p, act_seq, (size_t)p->sum_hnu, h->stream));
  if (eps) {
    legacy_predraw_drop(p);
    HIP_OK(upload_converted<T>(p->eps.p, eps, (size_t)p->sum_nhnu, h->stream));
    p->eps_inline = false;
    p->eps_from_generator = false; p->ahead_valid = false; p->ahead_on = false;
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int ampc_mppi_upload(ampc_mppi_plan* p, const double* x0, const double* act_seq,
                                const double* eps) {
  REQUIRE(p, "ampc_mppi_upload: NULL plan");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64 ? mppi_upload_impl<double>(p, x0, act_seq, eps)
                                     : mppi_upload_impl<float>(p, x0, act_seq, eps);
}

template <typename T> static int mppi_generate_impl(ampc_mppi_plan* p, uint64_t seed, uint64_t stream) {
  ampc_handle* h = p->h;
  const int nu = h->nu;
  if (p->lg_pre || p->lg_pre_inflight) legacy_predraw_drop(p);     // (a pre-drawn numpy-stream call is void)
  // The four-row rollout (small problems: a solve is a few tens of microseconds, a launch is five)
  // forms this noise in its own prologue -- the same values, element by element -- so nothing is
  // launched here; the buffer keeps whatever it held (AMPC_INLINE_NOISE = 0: always generate it).
  p->eps_inline = p->quad && env_int("AMPC_INLINE_NOISE", 1) != 0;
  // streams drawn in sequence (s, s + 1, ...): from now on a solve's combine launch forms the next one's
  // noise as well (eps_next, host_common.hpp), and a draw that was predicted is a swap of two pointers
  const bool ahead_allowed = env_int("AMPC_NOISE_AHEAD", 1) != 0;
  if (!ahead_allowed) p->ahead_on = false;
  if (ahead_allowed && p->eps_from_generator && seed == p->eps_seed && stream == p->eps_stream + 1) p->ahead_on = true;
  const bool hit = p->ahead_valid && p->ahead_seed == seed && p->ahead_stream == stream;
  p->ahead_valid = false;
  p->eps_seed = seed; p->eps_stream = stream;
  p->eps_from_generator = true;
  if (p->eps_inline) return 0;
  if (hit) {
    std::swap(p->eps, p->eps_next);
    return 0;
  }
  long long max_pairs = 0;
  for (int b = 0; b < p->B; ++b) {
    const long long pairs = ((long long)p->N[b] * p->H[b] * nu + 1) / 2;
    max_pairs = pairs > max_pairs ? pairs : max_pairs;
  }
  // one launch for the whole plan: problems over grid.y (and grid.z beyond 65535 of them)
  const unsigned gy = (unsigned)std::min(p->B, 65535), gz = (unsigned)((p->B + 65534) / 65535);
  hipLaunchKernelGGL(philox_normal_batch_kernel<T>, dim3((unsigned)((max_pairs + 255) / 256), gy, gz),
                     dim3(256), 0, h->stream, (T*)p->eps.p, (const MppiProblem<T>*)p->probs.p, p->B, nu,
                     seed, stream);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int ampc_mppi_generate_eps(ampc_mppi_plan* p, uint64_t seed, uint64_t stream) {
  REQUIRE(p, "ampc_mppi_generate_eps: NULL plan");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64 ? mppi_generate_impl<double>(p, seed, stream)
                                     : mppi_generate_impl<float>(p, seed, stream);
}
// numpy's legacy normal stream generated on the device (legacy_rng_kernels.hpp)
// Jump polynomial tables (tools/mt_jump.py), host copies, ascending in segment length.  Short
// segments make a short chain (a short-horizon plan's run-ahead is ready within a control step or
// two), long segments need fewer jump evaluations per block (config 3's run-ahead: 24 k blocks).
struct MtJumpTable {
  int jump_blocks = 0;
  std::vector<uint32_t> polys;                  // [n][624]
  long long cap() const { return (long long)(polys.size() / kMtN) * jump_blocks; }   // blocks it covers
};
constexpr int kMtMaxTables = 4;
static std::vector<MtJumpTable> g_mt_tables;
static constexpr int kMtHead = 34;            // blocks 0..33 come from the sequential head kernel

extern "C" int ampc_set_mt_jump_table(const uint32_t* polys, int n_polys, int jump_blocks) {
  REQUIRE(polys && n_polys >= 1 && jump_blocks >= kMtHead, "ampc_set_mt_jump_table: bad table");
  MtJumpTable t;
  t.jump_blocks = jump_blocks;
  t.polys.assign(polys, polys + (size_t)n_polys * kMtN);
  for (auto& old : g_mt_tables)
    if (old.jump_blocks == jump_blocks) { old = std::move(t); return 0; }
  REQUIRE((int)g_mt_tables.size() < kMtMaxTables, "ampc_set_mt_jump_table: too many tables");
  g_mt_tables.push_back(std::move(t));
  std::sort(g_mt_tables.begin(), g_mt_tables.end(),
            [](const MtJumpTable& x, const MtJumpTable& y) { return x.jump_blocks < y.jump_blocks; });
  return 0;
}

// most blocks any installed table covers (0: none installed)
static long long mt_jump_cap() {
  long long c = 0;
  for (const auto& t : g_mt_tables) c = std::max(c, t.cap());
  return c;
}

// raw MT19937 stream of `nblocks` blocks from `key` into `stream` (enqueued on `st`): the head
// sequentially, the rest block-parallel with the shortest segments whose table covers it
static int launch_mt_stream(ampc_mppi_plan* p, hipStream_t st, const uint32_t* d_key, int nblocks,
                            uint32_t* stream) {
  int ti = -1;
  for (int i = 0; i < (int)g_mt_tables.size() && ti < 0; ++i)
    if (g_mt_tables[i].cap() >= nblocks) ti = i;
  // block-parallel only where it pays: the head kernel (34 blocks, 22 us), the jump kernel (>= 23 us)
  // and a segment cost more than generating up to ~128 blocks in one sequential kernel (0.66 us each)
  const bool par = ti >= 0 && nblocks > std::max(kMtHead, env_int("AMPC_MT_PAR_MIN", 128));
  if (!par) {
    hipLaunchKernelGGL(mt19937_stream_kernel, dim3(1), dim3(256), 0, st, d_key, nblocks, stream, (uint32_t*)nullptr);
    return 0;
  }
  const MtJumpTable& t = g_mt_tables[ti];
  const int nseg = (nblocks - 1 + t.jump_blocks - 1) / t.jump_blocks;
  HIP_OK(p->lg_xraw.reserve((size_t)(kMtHead - 1) * kMtN * sizeof(uint32_t)));
  DevBuf& dpoly = p->lg_poly[ti];
  if (dpoly.bytes != t.polys.size() * sizeof(uint32_t)) {
    HIP_OK(dpoly.reserve(t.polys.size() * sizeof(uint32_t)));
    HIP_OK(hipMemcpy(dpoly.p, t.polys.data(), t.polys.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    dpoly.bytes = t.polys.size() * sizeof(uint32_t);
  }
  hipLaunchKernelGGL(mt19937_stream_kernel, dim3(1), dim3(256), 0, st, d_key, kMtHead, stream, (uint32_t*)p->lg_xraw.p);
  if (nseg > 1) {
    HIP_OK(p->lg_win.reserve((size_t)(nseg - 1) * kMtN * sizeof(uint32_t)));
    HIP_OK(hipMemsetAsync(p->lg_win.p, 0, (size_t)(nseg - 1) * kMtN * sizeof(uint32_t), st));
    hipLaunchKernelGGL(mt19937_jump_kernel, dim3(nseg - 1, kMtSlices), dim3(256), 0, st,
                       (const uint32_t*)p->lg_xraw.p, (const uint32_t*)dpoly.p, (uint32_t*)p->lg_win.p);
  }
  hipLaunchKernelGGL(mt19937_segment_kernel, dim3(nseg), dim3(256), 0, st, (const uint32_t*)p->lg_xraw.p,
                     (const uint32_t*)p->lg_win.p, kMtHead, t.jump_blocks, nblocks, stream);
  return 0;
}

// ---- the host C library's log(), proven reproducible (glibc_log.hpp) ---------------------------
// numpy's legacy_gauss calls log() of the C library this process runs on.  The device path may
// claim numpy's normals only if it evaluates the SAME function: find the library's __log_data in the
// loaded libm (by the bit patterns of ln2hi / ln2lo), then check both restated builds against
// log() itself on 2 * 10^5 arguments of the kind the polar method produces.  1 / 2 = that build
// reproduces log() bit for bit (and the table is kept); 0 = neither does (another libm): the
// device falls back to its own log() and the Python layer keeps the host draw as the default.
static double g_log_table[kLogTableDoubles];

struct LogLocate { const double* found = nullptr; };

static bool log_table_plausible(const double* t) {
  if (t[7] != -0.5) return false;                        // poly1[0]
  for (int i = 0; i < kLogN; ++i) {
    const double invc = t[18 + 2 * i], logc = t[19 + 2 * i];
    if (!(invc > 0.7 && invc < 1.5) || std::fabs(logc + std::log(invc)) > 1e-9) return false;
  }
  return true;
}

static int log_locate_cb(struct dl_phdr_info* info, size_t, void* data) {
  LogLocate* ctx = (LogLocate*)data;
  if (!info->dlpi_name || !std::strstr(info->dlpi_name, "libm")) return 0;
  const double pat[2] = {0x1.62e42fefa3800p-1, 0x1.ef35793c76730p-45};     // ln2hi, ln2lo
  const size_t need = (size_t)kLogTableDoubles * sizeof(double);
  for (int s = 0; s < info->dlpi_phnum; ++s) {
    const ElfW(Phdr)& ph = info->dlpi_phdr[s];
    if (ph.p_type != PT_LOAD || !(ph.p_flags & PF_R) || (ph.p_flags & PF_W)) continue;
    const char* base = (const char*)(info->dlpi_addr + ph.p_vaddr);
    for (size_t off = 0; off + need <= ph.p_memsz; off += 8) {
      if (std::memcmp(base + off, pat, sizeof(pat)) != 0) continue;
      if (log_table_plausible((const double*)(base + off))) {
        ctx->found = (const double*)(base + off);
        return 1;
      }
    }
  }
  return 0;
}

static int host_log_mode() {
  static std::once_flag once;
  static int mode = 0;
  std::call_once(once, [] {
    // test hook: 0 forces the device's own log(); 1 / 2 force that build's restatement onto the
    // device whatever the host's log() is (the device code of the build this host does not run
    // is checked against the CPU restatement of the same build, tests/test_gpu_legacy_noise.py)
    const int forced = env_int("AMPC_LEGACY_LOG", -1);
    if (forced == 0) return;
    LogLocate ctx;
    dl_iterate_phdr(log_locate_cb, &ctx);
    if (!ctx.found) return;
    std::memcpy(g_log_table, ctx.found, sizeof(g_log_table));
    if (forced == 1 || forced == 2) { mode = forced; return; }
    bool ok1 = true, ok2 = true;
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (int j = 0; j < 200000 && (ok1 || ok2); ++j) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const double u = (double)(s >> 11) * 0x1p-53;
      double x = (j & 3) == 3 ? 0.9375 + 0.13 * u : (u > 0 ? u : 0.5);
      if ((j & 63) == 5) x = std::ldexp(x, -(int)(s & 127));
      volatile double xv = x;                              // (no constant folding of the reference)
      const double ref = std::log(xv);
      if (ok1 && log_bits(glibc_log<1>(x, g_log_table)) != log_bits(ref)) ok1 = false;
      if (ok2 && log_bits(glibc_log<2>(x, g_log_table)) != log_bits(ref)) ok2 = false;
    }
    mode = ok1 ? 1 : (ok2 ? 2 : 0);
  });
  return mode;
}

extern "C" int ampc_legacy_log_mode(void) { return host_log_mode(); }

// What the enqueue phase of a legacy draw leaves for the phases after it.
struct LegacyDraw {
  bool trivial = false;      // the single value asked for was the cached one: nothing was drawn
  int pos = 0, shift = 0;
  long long n = 0, n_pairs = 0, n_att = 0;
};

constexpr size_t kLegacyStatusSlot = 3 + kMtN * sizeof(uint32_t) / sizeof(long long);   // (in long longs) 1: a look-back wait expired
constexpr size_t kLegacyGather = (kLegacyStatusSlot + 1) * sizeof(long long);

// blocks a draw of n_att attempts can touch, counted from the block holding the generator's key
static int legacy_blocks_for(int start_pos, long long n_att) {
  return (int)(((long long)start_pos + 4 * n_att) / kMtN) + 2;
}

// A legacy draw runs in three phases so that a caller can put a whole control step between them
// (ampc_mppi_run_legacy) and synchronise ONCE:
//   legacy_enqueue    main stream: the normals into the plan's noise buffer; what the host needs
//                     afterwards (last attempt, cached value, the stream block the generator ends
//                     in) gathered and copied to pinned memory
//   legacy_speculate  side stream: keeps the raw MT19937 stream generated AHEAD of the draws
//   legacy_finish     after the main stream has been synchronised: the generator state to hand back.
//
// Run-ahead.  Generating the raw stream is a latency-bound chain (sequential head, jump
// polynomials, segments; ~125 us for one config-3 draw) and, run next to a rollout, it and the
// rollout slow each other down (they share the CUs' LDS pipelines: measured 5x on the chain, +40 us
// on the rollout).  So it is not done per call: a buffer holds the stream of SEVERAL calls
// (AMPC_LEGACY_AHEAD, 8), and while the calls consume buffer c the side stream fills buffer 1 - c
// with the continuation from block lg_next_from of c -- placed one call's worth before the end of c,
// so that a call starting before that block still fits into c and a call starting at or after it
// switches buffers.  The chain then has several control steps of time to finish and its cost is
// paid once per several calls.  A call whose generator state is not the one the previous call left
// (someone else drew from numpy's generator in between) generates its own words on the main
// stream and the run-ahead starts again from there.
// What legacy_finish returns when a look-back wait of the draw kernel expired (the results are not to be used: draw
// again -- the generator state the caller passed in has not been touched).
constexpr int kLegacyExpired = 2;
// ... and legacy_enqueue for a pre-draw it cannot serve from the run-ahead (nothing was launched).
constexpr int kLegacySkip = 3;

// A pre-drawn next call (mppi_run_impl) may still be running on its own stream: it writes the second noise buffer,
// the look-back words and the pinned results.  Anything else that is about to touch those waits for it first.
static int legacy_predraw_wait(ampc_mppi_plan* p) {
  if (p->lg_pre_inflight) {
    HIP_OK(hipEventSynchronize(p->lg_pre_done));
    p->lg_pre_inflight = false;
  }
  return 0;
}

// pre = true: the draw of the NEXT control step, made while the current solve runs -- on a stream of its own
// (p->lg_draw), into the plan's second noise buffer (p->eps_pre), and only if its words are already in the
// run-ahead buffer (kLegacySkip otherwise).  The draw kernel's workgroups (4 waves, 115 VGPRs) fit next to a
// rollout workgroup on every CU (8 waves of 172 VGPRs), so the scan runs inside the rollout's shadow.
template <typename T>
static int legacy_enqueue(ampc_mppi_plan* p, const uint32_t* key, int pos, int has_gauss, double cached,
                          LegacyDraw* d, bool retry = false, bool pre = false) {
  ampc_handle* h = p->h;
  if (!pre) {
    if (int rc = legacy_predraw_wait(p)) return rc;
    p->eps_inline = false;          // (this draw fills the plan's noise buffer)
    p->eps_from_generator = false; p->ahead_valid = false; p->ahead_on = false;
  }
  const long long n = p->sum_nhnu;
  const int shift = has_gauss ? 1 : 0;
  const long long n_pairs = (n - shift + 1) / 2;
  d->pos = pos; d->shift = shift; d->n = n; d->n_pairs = n_pairs;
  if (n_pairs == 0 && pre) return kLegacySkip;
  if (n_pairs == 0) {          // the single value asked for is the cached one
    HIP_OK(p->lg_scale.reserve(sizeof(double)));
    const double sc = std::sqrt(p->sigma[0]);
    HIP_OK(hipMemcpyAsync(p->lg_scale.p, &sc, sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(legacy_first_value_kernel<T>, dim3(1), dim3(64), 0, h->stream, cached,
                       (const double*)p->lg_scale.p, (T*)p->eps.p);
    HIP_OK(hipStreamSynchronize(h->stream));          // (sc is on the stack)
    d->trivial = true;
    return 0;
  }
  // attempts to evaluate: acceptance probability pi/4; the number needed for n_pairs acceptances
  // has mean n_pairs / p and standard deviation sqrt(n_pairs (1 - p)) / p.  16 standard deviations
  // + 64 of slack (the count is checked afterwards); words past the last consumed one cost
  // generation time only, so the slack is kept small: a short horizon's draw then fits a few blocks.
  const double kAcc = 0.7853981633974483;
  const long long n_att = (long long)(((double)n_pairs + 16.0 * std::sqrt((double)n_pairs * (1.0 - kAcc))) / kAcc) + 64;
  d->n_att = n_att;
  const int nblocks = legacy_blocks_for(pos, n_att);
  const int n_wg = (int)((n_att + kPolarPerWg - 1) / kPolarPerWg);
  REQUIRE(n_wg <= 65536 * 16, "legacy normal: too many values for one call");
  if (!p->lg_side) {
    HIP_OK(hipStreamCreateWithFlags(&p->lg_side, hipStreamNonBlocking));
    for (hipEvent_t& e : p->lg_evs) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&p->lg_drawn, hipEventDisableTiming));
  }
  if (pre) {
    // served from the run-ahead or not at all (the same test as below, without its side effects on a miss)
    bool ok = p->lg_spec && p->lg_spec_pos == pos && std::memcmp(p->lg_spec_key.data(), key, kMtN * sizeof(uint32_t)) == 0;
    if (ok) {
      const bool sw = p->lg_next && p->lg_blk0 >= p->lg_next_from;
      const int cur = sw ? 1 - p->lg_cur : p->lg_cur, blk0 = sw ? p->lg_blk0 - p->lg_next_from : p->lg_blk0;
      ok = p->lg_blocks[cur] - blk0 >= nblocks;
    }
    if (!ok) return kLegacySkip;
    if (!p->lg_draw) {
      HIP_OK(hipStreamCreateWithFlags(&p->lg_draw, hipStreamNonBlocking));
      HIP_OK(hipEventCreateWithFlags(&p->lg_pre_done, hipEventDisableTiming));
    }
    HIP_OK(p->eps_pre.reserve((size_t)p->sum_nhnu * sizeof(T)));
  }
  const hipStream_t st = pre ? p->lg_draw : h->stream;
  // look-back words of the draw kernel: cleared when (re)allocated and when the 24-bit epoch wraps
  if ((size_t)n_wg * sizeof(unsigned long long) > p->lg_cnt.bytes || (++p->lg_epoch & 0xffffffu) == 0) {
    HIP_OK(p->lg_cnt.reserve((size_t)n_wg * sizeof(unsigned long long)));
    HIP_OK(hipMemsetAsync(p->lg_cnt.p, 0, p->lg_cnt.bytes, st));
    p->lg_epoch = 1;
  }
  if (!p->lg_pin) {
    HIP_OK(hipHostMalloc(&p->lg_pin, kLegacyGather, hipHostMallocDefault));
    HIP_OK(hipHostGetDevicePointer(&p->lg_pin_dev, p->lg_pin, 0));
  }
  ((long long*)p->lg_pin)[kLegacyStatusSlot] = 0;      // (the previous draw's results have been read: legacy_finish)
  if (!p->lg_scale_set) {            // sqrt(sigma_b): once per plan
    HIP_OK(p->lg_scale.reserve((size_t)p->B * sizeof(double)));
    std::vector<double> sc(p->B);
    for (int b = 0; b < p->B; ++b) sc[b] = std::sqrt(p->sigma[b]);
    HIP_OK(hipMemcpy(p->lg_scale.p, sc.data(), sc.size() * sizeof(double), hipMemcpyHostToDevice));
    p->lg_scale_set = true;
  }
  // where the words come from: the run-ahead if this call starts where the previous one ended
  bool hit = p->lg_spec && p->lg_spec_pos == pos &&
             std::memcmp(p->lg_spec_key.data(), key, kMtN * sizeof(uint32_t)) == 0;
  if (hit && p->lg_next && p->lg_blk0 >= p->lg_next_from) {     // past the continuation's start: switch
    p->lg_blk0 -= p->lg_next_from;
    p->lg_cur = 1 - p->lg_cur;
    p->lg_next = false;
  }
  if (hit && p->lg_blocks[p->lg_cur] - p->lg_blk0 < nblocks) hit = false;    // (buffer too short)
  if (hit) {
    HIP_OK(hipStreamWaitEvent(st, p->lg_evs[p->lg_cur], 0));
    ++p->lg_hits;
  } else {
    REQUIRE(!pre, "internal: a pre-draw missed the run-ahead after the check");
    HIP_OK(hipStreamSynchronize(p->lg_side));         // (a stale run-ahead may still be running)
    p->lg_cur = 0; p->lg_blk0 = 0; p->lg_next = false; p->lg_hits = 0;
    HIP_OK(p->lg_key[0].reserve(kMtN * sizeof(uint32_t)));
    HIP_OK(p->lg_stream[0].reserve((size_t)nblocks * kMtN * sizeof(uint32_t)));
    HIP_OK(hipMemcpyAsync(p->lg_key[0].p, key, kMtN * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));          // (the source is caller memory)
    if (int rc = launch_mt_stream(p, h->stream, (const uint32_t*)p->lg_key[0].p, nblocks,
                                  (uint32_t*)p->lg_stream[0].p)) return rc;
    HIP_OK(hipEventRecord(p->lg_drawn, h->stream));
    HIP_OK(hipStreamWaitEvent(p->lg_side, p->lg_drawn, 0));      // (the side stream reads this buffer)
    p->lg_blocks[0] = nblocks;
  }
  p->lg_spec = false;
  const uint32_t* stream = (const uint32_t*)p->lg_stream[p->lg_cur].p + (size_t)p->lg_blk0 * kMtN;
  const uint32_t* u = stream + pos;                 // the generator's next output
  const int logv = host_log_mode();
  if (logv && p->lg_logtab.bytes == 0) {
    HIP_OK(p->lg_logtab.reserve(sizeof(g_log_table)));
    HIP_OK(hipMemcpyAsync(p->lg_logtab.p, g_log_table, sizeof(g_log_table), hipMemcpyHostToDevice, st));
  }
  // ONE launch: attempts, the scan of the accept flags, the normals, and what the host needs afterwards (last
  // attempt, cached value, pair count, the stream block the generator ends in) straight into pinned memory
  auto draw = logv == 1 ? polar_draw_kernel<T, 1> : logv == 2 ? polar_draw_kernel<T, 2> : polar_draw_kernel<T, 0>;
  hipLaunchKernelGGL(draw, dim3(std::min(n_wg, kPolarMaxWgs)), dim3(256), 0, st, u, (int)n_att,
                     (unsigned long long*)p->lg_cnt.p, p->lg_epoch & 0xffffffu, n, shift, cached,
                     (const MppiProblem<T>*)p->probs.p, (const double*)p->lg_scale.p, p->B,
                     (T*)(pre ? p->eps_pre.p : p->eps.p), stream, pos,
                     (long long*)p->lg_pin_dev, (const double*)p->lg_logtab.p, n_wg,
                     // (a second attempt always waits the full bound; AMPC_POLAR_SPIN_LIMIT is the tests' hook)
                     retry ? kPolarSpinLimit : std::max(1, env_int("AMPC_POLAR_SPIN_LIMIT", kPolarSpinLimit)));
  HIP_OK(hipGetLastError());
  return 0;
}

static int legacy_speculate(ampc_mppi_plan* p, const LegacyDraw& d) {
  if (d.trivial || p->lg_next || env_int("AMPC_LEGACY_SPECULATE", 1) == 0) return 0;
  // the most blocks one call can touch (a position of 624 at entry)
  const int per_call = legacy_blocks_for(kMtN, d.n_att);
  // Continuation of the current buffer from one call's worth before its end.  Its length: `ahead`
  // calls (2 right after a miss -- the very next call waits for it), at least two calls' worth so
  // that every buffer serves at least one call, at most what the jump table covers.
  const int c = p->lg_cur, nb = 1 - c;
  const int from = std::max(0, p->lg_blocks[c] - per_call);
  const int ahead = p->lg_hits == 0 ? 2 : std::max(2, env_int("AMPC_LEGACY_AHEAD", 8));
  long long sb = (long long)ahead * per_call;
  const long long cap = mt_jump_cap();
  if (sb > cap) sb = std::max<long long>(cap, 2LL * per_call);
  HIP_OK(p->lg_key[nb].reserve(kMtN * sizeof(uint32_t)));
  HIP_OK(p->lg_stream[nb].reserve((size_t)sb * kMtN * sizeof(uint32_t)));
  hipLaunchKernelGGL(mt19937_key_of_block_kernel, dim3(1), dim3(256), 0, p->lg_side,
                     (const uint32_t*)p->lg_stream[c].p + (size_t)from * kMtN, (uint32_t*)p->lg_key[nb].p);
  if (int rc = launch_mt_stream(p, p->lg_side, (const uint32_t*)p->lg_key[nb].p, (int)sb,
                                (uint32_t*)p->lg_stream[nb].p)) return rc;
  HIP_OK(hipEventRecord(p->lg_evs[nb], p->lg_side));
  p->lg_blocks[nb] = (int)sb;
  p->lg_next = true;
  p->lg_next_from = from;
  return 0;
}

// The main stream has been synchronised: p->lg_pin holds the gathered results.
static int legacy_finish(ampc_mppi_plan* p, const LegacyDraw& d, const uint32_t* key, uint32_t* key_out,
                         int* pos_out, int* has_gauss_out, double* cached_out) {
  if (d.trivial) {
    if (key_out != key) std::memcpy(key_out, key, kMtN * sizeof(uint32_t));
    *pos_out = d.pos; *has_gauss_out = 0; *cached_out = 0.0;
    return 0;
  }
  const long long* gl = (const long long*)p->lg_pin;
  const long long fin[2] = {gl[0], gl[1]};
  const int total = (int)gl[2];
  if (gl[kLegacyStatusSlot] != 0) {
    g_err = "legacy normal: the draw kernel gave up waiting for a workgroup's pair count";
    return kLegacyExpired;
  }
  REQUIRE(total >= d.n_pairs && fin[0] >= 0, "legacy normal: not enough accepted pairs in the generated stream");
  // generator state after the last consumed word
  const long long idx = (long long)d.pos + 4 * (fin[0] + 1);
  int po = (int)(idx % kMtN);
  long long blk = idx / kMtN;
  if (po == 0 && idx > 0) { po = kMtN; blk -= 1; }  // randomkit regenerates lazily: pos == 624
  const uint32_t* last = (const uint32_t*)(gl + 3);
  for (int i = 0; i < kMtN; ++i) key_out[i] = mt_untemper(last[i]);
  *pos_out = po;
  const bool odd = ((d.n - d.shift) & 1) != 0;
  *has_gauss_out = odd ? 1 : 0;
  double cg = 0.0;
  if (odd) std::memcpy(&cg, &fin[1], sizeof(double));
  *cached_out = cg;
  // what the next call has to present to take its words from the run-ahead
  p->lg_spec_key.assign(key_out, key_out + kMtN);
  p->lg_spec_pos = po;
  p->lg_blk0 += (int)blk;
  p->lg_spec = true;
  return 0;
}

template <typename T>
static int legacy_normal_impl(ampc_mppi_plan* p, const uint32_t* key, int pos, int has_gauss, double cached,
                              uint32_t* key_out, int* pos_out, int* has_gauss_out, double* cached_out) {
  legacy_predraw_drop(p);           // (a pre-drawn next call of ampc_mppi_run_legacy: drawn again here, same values)
  for (int attempt = 0;; ++attempt) {
    LegacyDraw d;
    if (int rc = legacy_enqueue<T>(p, key, pos, has_gauss, cached, &d, attempt > 0)) return rc;
    if (int rc = legacy_speculate(p, d)) return rc;
    HIP_OK(hipStreamSynchronize(p->h->stream));
    const int rc = legacy_finish(p, d, key, key_out, pos_out, has_gauss_out, cached_out);
    if (rc != kLegacyExpired || attempt > 0) return rc == kLegacyExpired ? -1 : rc;
    ++p->lg_redraws;                // (a look-back wait expired under contention: the same draw once more)
  }
}

extern "C" int ampc_mppi_plan_legacy_redraws(const ampc_mppi_plan* p, long long* count) {
  REQUIRE(p && count, "ampc_mppi_plan_legacy_redraws: NULL argument");
  *count = p->lg_redraws;
  return 0;
}

extern "C" int ampc_mppi_legacy_normal(ampc_mppi_plan* p, const uint32_t* key, int pos, int has_gauss,
                                       double cached, uint32_t* key_out, int* pos_out,
                                       int* has_gauss_out, double* cached_out) {
  REQUIRE(p && key && key_out && pos_out && has_gauss_out && cached_out, "ampc_mppi_legacy_normal: NULL argument");
  REQUIRE(pos >= 0 && pos <= kMtN, "ampc_mppi_legacy_normal: generator position must be in 0..624");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? legacy_normal_impl<double>(p, key, pos, has_gauss, cached, key_out, pos_out, has_gauss_out, cached_out)
             : legacy_normal_impl<float>(p, key, pos, has_gauss, cached, key_out, pos_out, has_gauss_out, cached_out);
}

extern "C" int ampc_mppi_plan_set_geometry(ampc_mppi_plan* p, int tile_rows, int horizon_cap) {
  REQUIRE(p, "ampc_mppi_plan_set_geometry: NULL plan");
  REQUIRE(tile_rows == 0 || tile_rows == 4 || tile_rows == 16 || tile_rows == 32 || tile_rows == 64,
          "ampc_mppi_plan_set_geometry: tile_rows must be 0 (automatic), 4, 16, 32 or 64");
  REQUIRE(horizon_cap >= 0, "ampc_mppi_plan_set_geometry: horizon_cap < 0");
  HIP_OK(hipSetDevice(p->h->device));
  HIP_OK(hipStreamSynchronize(p->h->stream));
  p->forced_mt = tile_rows / 16;
  p->forced_quad = tile_rows == 4;
  for (int hb : p->H) p->max_h = hb > p->max_h ? hb : p->max_h;
  if (horizon_cap > p->max_h) p->max_h = horizon_cap;
  p->lds_eps = -1;
  p->lds_red = 0;
  // noise drawn by the device generator before the rebuild: the rebuild may change HOW the plan holds it
  // (formed inside the four-row rollout, or in the noise buffer, which a rebuild resets) -- draw it again
  // for the new geometry, the same (seed, stream): the values the caller asked for
  const bool redraw = p->eps_from_generator;
  const uint64_t seed = p->eps_seed, stream = p->eps_stream;
  p->eps_from_generator = false; p->ahead_valid = false; p->ahead_on = false;
  if (int rc = p->h->precision == AMPC_F64 ? plan_build<double>(p) : plan_build<float>(p)) return rc;
  if (!p->models.empty()) {
    // a model table was set before: the rebuilt plan must still be on the shape-specialised kernels (the only
    // ones that take the per-problem model offset) -- refuse here rather than roll out on one model silently
    if (int rc = p->h->precision == AMPC_F64 ? mppi_require_static<double>(p, "ampc_mppi_plan_set_geometry")
                                             : mppi_require_static<float>(p, "ampc_mppi_plan_set_geometry")) return rc;
    if (int rc = build_tile_order(p)) return rc;
  }
  return redraw ? ampc_mppi_generate_eps(p, seed, stream) : 0;
}

extern "C" int ampc_mppi_plan_set_step_offset(ampc_mppi_plan* p, uint64_t first_step) {
  REQUIRE(p, "ampc_mppi_plan_set_step_offset: NULL plan");
  REQUIRE(first_step < (1ull << 56), "ampc_mppi_plan_set_step_offset: first_step must be < 2^56");
  p->step_offset = first_step;
  return 0;
}

template <typename T> static int mppi_set_noise_ids_impl(ampc_mppi_plan* p) {
  std::vector<MppiProblem<T>> pr(p->B);
  HIP_OK(hipStreamSynchronize(p->h->stream));
  HIP_OK(hipMemcpy(pr.data(), p->probs.p, pr.size() * sizeof(MppiProblem<T>), hipMemcpyDeviceToHost));
  for (int b = 0; b < p->B; ++b) {
    pr[b].noise_id = p->noise_id[b];
    pr[b].model = p->model_idx.empty() ? 0 : p->model_idx[b];
  }
  HIP_OK(hipMemcpy(p->probs.p, pr.data(), pr.size() * sizeof(MppiProblem<T>), hipMemcpyHostToDevice));
  p->ahead_valid = false;          // (noise formed ahead carried the old ids)
  // noise already drawn by the device generator is keyed by the ids: draw it again, so that the plan's
  // noise is Philox(seed, stream, NEW id) whether it is formed inside the rollout or held in the buffer
  if (p->eps_from_generator) {
    const uint64_t seed = p->eps_seed, stream = p->eps_stream;
    p->eps_from_generator = false; p->ahead_on = false;
    return mppi_generate_impl<T>(p, seed, stream);
  }
  return 0;
}

extern "C" int ampc_mppi_plan_set_noise_ids(ampc_mppi_plan* p, const uint32_t* ids) {
  REQUIRE(p && ids, "ampc_mppi_plan_set_noise_ids: NULL argument");
  HIP_OK(hipSetDevice(p->h->device));
  p->noise_id.assign(ids, ids + p->B);
  return p->h->precision == AMPC_F64 ? mppi_set_noise_ids_impl<double>(p)
                                     : mppi_set_noise_ids_impl<float>(p);
}

extern "C" int ampc_mppi_solve(ampc_mppi_plan* p) {
  REQUIRE(p, "ampc_mppi_solve: NULL plan");
  HIP_OK(hipSetDevice(p->h->device));
  p->u_in_pin = false;
  return p->h->precision == AMPC_F64 ? mppi_solve_impl<double>(p) : mppi_solve_impl<float>(p);
}

template <typename T>
static int mppi_download_impl(ampc_mppi_plan* p, double* act_seq, double* u, double* costs,
                              double* eps_out) {
  ampc_handle* h = p->h;
  if (costs && !p->costs_final && p->term_mode == 0) {
    MppiArgs<T> a = make_args<T>(p);
    int maxn = 0;
    for (int n : p->N) maxn = n > maxn ? n : maxn;
    hipLaunchKernelGGL(mppi_finalize_costs_kernel<T>, dim3((maxn + 255) / 256, p->B), dim3(256), 0,
                       h->stream, a);
    HIP_OK(hipGetLastError());
    p->costs_final = true;
  }
  if (act_seq) HIP_OK(download_converted<T>(act_seq, p->act[p->cur].p, (size_t)p->sum_hnu, h->stream));
  if (u && p->u_in_pin) {          // the last solve was a one-call control step: its controls went to host memory
    HIP_OK(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < (size_t)p->B * h->nu; ++i) u[i] = (double)((const T*)p->pin_u)[i];
  } else if (u) {
    HIP_OK(download_converted<T>(u, p->u_out.p, (size_t)p->B * h->nu, h->stream));
  }
  if (costs) HIP_OK(download_converted<T>(costs, p->costs.p, (size_t)p->sum_n, h->stream));
  if (eps_out) HIP_OK(download_converted<T>(eps_out, p->eps_out.p, (size_t)p->sum_nhnu, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int ampc_mppi_download(ampc_mppi_plan* p, double* act_seq, double* u, double* costs,
                                  double* eps_out) {
  REQUIRE(p, "ampc_mppi_download: NULL plan");
  REQUIRE(p->solved || (!u && !costs && !eps_out), "ampc_mppi_download: nothing solved yet");
  REQUIRE(!eps_out || p->keep_eps_out || p->lds_eps < 0,
          "ampc_mppi_download: eps_out was not kept (ampc_mppi_plan_set_outputs(plan, 0))");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64 ? mppi_download_impl<double>(p, act_seq, u, costs, eps_out)
                                     : mppi_download_impl<float>(p, act_seq, u, costs, eps_out);
}

// MPPI.run() in one call: x0 (and optionally a new warm start) in, noise, solve, controls out, ONE
// host synchronisation; x0 and the controls travel through pinned staging buffers.
struct LegacyState {          // numpy's legacy generator state, in and out (ampc_mppi_run_legacy)
  const uint32_t* key; int pos, has_gauss; double cached;
  uint32_t* key_out; int* pos_out; int* has_gauss_out; double* cached_out;
};

static inline void draw_pack(const LegacyDraw& d, long long* o) {
  o[0] = d.trivial ? 1 : 0; o[1] = d.pos; o[2] = d.shift; o[3] = d.n; o[4] = d.n_pairs; o[5] = d.n_att;
}
static inline LegacyDraw draw_unpack(const long long* o) {
  LegacyDraw d;
  d.trivial = o[0] != 0; d.pos = (int)o[1]; d.shift = (int)o[2]; d.n = o[3]; d.n_pairs = o[4]; d.n_att = o[5];
  return d;
}

// One control step in one call (MPPI.run, mppi.py:154-168).  The host's part of it is kept off the stream:
//   * x0 is written into host memory the rollout reads directly (mapped), no copy packet;
//   * the update writes u into mapped host memory and raises a per-problem sequence word behind it
//     (MppiArgs::done_flag); the host polls that word -- no copy packet, no hipStreamSynchronize;
//   * numpy-stream mode: the NEXT call's normals are drawn from the generator state this call returns -- known
//     before the solve on a pre-drawn call -- on a second stream into a second noise buffer, inside the running
//     rollout's shadow; a next call presenting exactly that state (nobody drew from numpy's generator in between)
//     swaps the buffers and launches only the rollout (+ update) -- otherwise it draws as before, results identical.
// AMPC_RUN_MAPPED=0 / AMPC_LEGACY_PREDRAW=0 restore the copy / in-call-draw behaviour (same results).
template <typename T>
static int mppi_run_impl(ampc_mppi_plan* p, const double* x0, const double* act_seq, int noise,
                         uint64_t seed, uint64_t stream, double* u, const LegacyState* lg = nullptr,
                         bool redo = false) {
  ampc_handle* h = p->h;
  const size_t nx0 = (size_t)p->B * h->nx, nuo = (size_t)p->B * h->nu;
  if (!p->pin_x0) {
    // (the runtime's default mapping, NOT hipHostMallocCoherent: with the uncached coherent mapping every
    //  workgroup's read of x0 crosses PCIe -- measured c3 3127 -> 1711, c2 23.5 k -> 9.6 k calls/s.  The default is
    //  sufficient: the host writes x0 before the launch (whose acquire is system scope), and the completion word is a
    //  system-scope release store at the very end of the last kernel, at worst visible when that kernel retires)
    HIP_OK(hipHostMalloc(&p->pin_x0, nx0 * sizeof(T), hipHostMallocMapped));
    HIP_OK(hipHostMalloc(&p->pin_u, nuo * sizeof(T), hipHostMallocMapped));
    HIP_OK(hipHostMalloc((void**)&p->pin_flag, (size_t)p->B * sizeof(unsigned long long), hipHostMallocMapped));
    HIP_OK(hipHostGetDevicePointer(&p->pin_x0_dev, p->pin_x0, 0));
    HIP_OK(hipHostGetDevicePointer(&p->pin_u_dev, p->pin_u, 0));
    HIP_OK(hipHostGetDevicePointer((void**)&p->pin_flag_dev, p->pin_flag, 0));
    std::memset(p->pin_flag, 0, (size_t)p->B * sizeof(unsigned long long));
  }
  const bool mapped = env_int("AMPC_RUN_MAPPED", 1) != 0;
  const bool predraw = lg && env_int("AMPC_LEGACY_PREDRAW", 1) != 0;
  LegacyDraw draw;
  bool pre_hit = false, finished = false;
  if (lg) {
    if (p->lg_pre) {
      pre_hit = p->lg_pre_pos == lg->pos && p->lg_pre_has_gauss == lg->has_gauss &&
                (lg->has_gauss == 0 || std::memcmp(&p->lg_pre_cached, &lg->cached, sizeof(double)) == 0) &&
                std::memcmp(p->lg_pre_key.data(), lg->key, kMtN * sizeof(uint32_t)) == 0;
      p->lg_pre = false;
    }
    if (pre_hit) {
      // this call's normals were drawn behind the previous call's update: what the draw left for the host is (or
      // will in a moment be) in pinned memory -- the generator state to hand back is known BEFORE the solve
      draw = draw_unpack(p->lg_pre_draw);
      if (int rcw = legacy_predraw_wait(p)) return rcw;
      std::swap(p->eps, p->eps_pre);         // the pre-drawn buffer is this call's noise
      p->eps_inline = false; p->eps_from_generator = false; p->ahead_valid = false; p->ahead_on = false;
      const int rc = legacy_finish(p, draw, lg->key, lg->key_out, lg->pos_out, lg->has_gauss_out, lg->cached_out);
      if (rc == kLegacyExpired) { pre_hit = false; ++p->lg_redraws; std::swap(p->eps, p->eps_pre); }   // void: draw in this call
      else if (rc) return rc;
      else finished = true;
    }
    if (!pre_hit)
      if (int rc = legacy_enqueue<T>(p, lg->key, lg->pos, lg->has_gauss, lg->cached, &draw, redo)) return rc;
  } else {
    legacy_predraw_drop(p);
  }
  T* px = (T*)p->pin_x0;
  for (size_t i = 0; i < nx0; ++i) px[i] = (T)x0[i];
  if (!mapped) HIP_OK(hipMemcpyAsync(p->x0.p, px, nx0 * sizeof(T), hipMemcpyHostToDevice, h->stream));
  if (act_seq) HIP_OK(upload_converted<T>(p->act[p->cur].p, act_seq, (size_t)p->sum_hnu, h->stream));
  if (noise == 1)
    if (int rc = mppi_generate_impl<T>(p, seed, stream)) return rc;
  const unsigned long long seq = ++p->run_seq;
  p->host_io = mapped;
  const int rc_solve = mppi_solve_impl<T>(p);
  p->host_io = false;
  if (rc_solve) return rc_solve;
  if (!mapped) HIP_OK(hipMemcpyAsync(p->pin_u, p->u_out.p, nuo * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  // The NEXT call's draw, from the generator state this call hands back: on its own stream, into the second noise
  // buffer, NEXT TO the solve that has just been launched (legacy_enqueue, pre) -- only once calls follow each other
  // on the generator (this call took its words from the run-ahead or from a pre-drawn buffer): a caller who draws
  // from numpy's generator between calls never pays for a wasted draw.
  auto enqueue_next = [&]() -> int {
    if (!predraw || draw.trivial || !(pre_hit || p->lg_hits > 0)) return 0;
    LegacyDraw nd;
    const int rce = legacy_enqueue<T>(p, lg->key_out, *lg->pos_out, *lg->has_gauss_out, *lg->cached_out, &nd, false, true);
    if (rce == kLegacySkip) return 0;          // (its words are not in the run-ahead yet: the next call draws itself)
    if (rce) return rce;
    HIP_OK(hipEventRecord(p->lg_pre_done, p->lg_draw));
    p->lg_pre_inflight = true;
    if (int rc = legacy_speculate(p, nd)) return rc;
    p->lg_pre_key.assign(lg->key_out, lg->key_out + kMtN);
    p->lg_pre_pos = *lg->pos_out; p->lg_pre_has_gauss = *lg->has_gauss_out; p->lg_pre_cached = *lg->cached_out;
    draw_pack(nd, p->lg_pre_draw);
    p->lg_pre = true;
    return 0;
  };
  // (the solve is on its way: the next call's raw stream goes to the side stream behind it)
  if (lg && !pre_hit)
    if (int rc = legacy_speculate(p, draw)) return rc;
  if (finished)                       // ... and so does the next call's draw, while the host would only be waiting
    if (int rc = enqueue_next()) return rc;
  if (mapped) {
    // poll the problems' completion words; every so often ask the stream whether it has stopped (an error, or --
    // never expected -- a finished stream whose words did not arrive)
    volatile const unsigned long long* f = p->pin_flag;
    for (unsigned spins = 1;; ++spins) {
      bool all = true;
      for (int b = 0; b < p->B; ++b) all = all && f[b] == seq;
      if (all) break;
      if ((spins & 0xfffu) == 0) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) {
          bool now = true;
          for (int b = 0; b < p->B; ++b) now = now && f[b] == seq;
          REQUIRE(now, "ampc_mppi_run: the stream finished but the solve's completion word did not arrive");
          break;
        }
        if (q != hipErrorNotReady) return fail(std::string("ampc_mppi_run: ") + hipGetErrorString(q));
      }
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    HIP_OK(hipStreamSynchronize(h->stream));
  }
  if (lg && !finished) {
    const int rc = legacy_finish(p, draw, lg->key, lg->key_out, lg->pos_out, lg->has_gauss_out, lg->cached_out);
    if (rc == kLegacyExpired && !redo) {
      // a look-back wait of this call's draw expired (contention): the solve ran on void noise.  The generator state
      // is still the caller's and the solve read act[cur ^ 1 now] without modifying it: the whole step once more.
      ++p->lg_redraws;
      p->cur ^= 1;
      return mppi_run_impl<T>(p, x0, nullptr, noise, seed, stream, u, lg, true);
    }
    if (rc) return rc == kLegacyExpired ? -1 : rc;
    if (int rc2 = enqueue_next()) return rc2;
  }
  const T* pu = (const T*)p->pin_u;
  for (size_t i = 0; i < nuo; ++i) u[i] = (double)pu[i];
  p->u_in_pin = mapped;
  return 0;
}

extern "C" int ampc_mppi_run_legacy(ampc_mppi_plan* p, const double* x0, const double* act_seq,
                                    const uint32_t* key, int pos, int has_gauss, double cached,
                                    uint32_t* key_out, int* pos_out, int* has_gauss_out, double* cached_out,
                                    double* u) {
  REQUIRE(p && x0 && u && key && key_out && pos_out && has_gauss_out && cached_out,
          "ampc_mppi_run_legacy: NULL argument");
  REQUIRE(pos >= 0 && pos <= kMtN, "ampc_mppi_run_legacy: pos must be in [0, 624]");
  REQUIRE(p->sum_nhnu > 0, "ampc_mppi_run_legacy: empty plan");
  HIP_OK(hipSetDevice(p->h->device));
  const LegacyState lg{key, pos, has_gauss, cached, key_out, pos_out, has_gauss_out, cached_out};
  return p->h->precision == AMPC_F64 ? mppi_run_impl<double>(p, x0, act_seq, 0, 0, 0, u, &lg)
                                     : mppi_run_impl<float>(p, x0, act_seq, 0, 0, 0, u, &lg);
}

extern "C" int ampc_mppi_run(ampc_mppi_plan* p, const double* x0, const double* act_seq, int noise,
                             uint64_t seed, uint64_t stream, double* u) {
  REQUIRE(p && x0 && u, "ampc_mppi_run: NULL argument");
  REQUIRE(noise == 0 || noise == 1, "ampc_mppi_run: noise must be 0 (resident) or 1 (Philox)");
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64 ? mppi_run_impl<double>(p, x0, act_seq, noise, seed, stream, u)
                                     : mppi_run_impl<float>(p, x0, act_seq, noise, seed, stream, u);
}

extern "C" int ampc_mppi_set_x0_dev(ampc_mppi_plan* p, const void* x0_dev) {
  REQUIRE(p && x0_dev, "ampc_mppi_set_x0_dev: NULL argument");
  HIP_OK(hipSetDevice(p->h->device));
  HIP_OK(hipMemcpyAsync(p->x0.p, x0_dev, (size_t)p->B * p->h->nx * p->h->esz(),
                        hipMemcpyDeviceToDevice, p->h->stream));
  return 0;
}

extern "C" int ampc_mppi_plan_info(const ampc_mppi_plan* p, int* n_workgroups, int* samples_per_wg,
                                   double* flops, double* bytes) {
  REQUIRE(p, "ampc_mppi_plan_info: NULL plan");
  const ampc_handle* h = p->h;
  if (n_workgroups) *n_workgroups = p->n_tiles;
  if (samples_per_wg) *samples_per_wg = p->tile_m;
  // Algorithmic work: per sample-step 2*sum(in*out) MLP flops plus the quadratic
  // stage cost; bytes = noise in + clipped noise out + costs + weights once.
  double macs = h->has_sindy ? (double)h->s_nfeat * h->nx : (h->has_lin ? (double)h->nx * (h->nx + h->nu) : 0.0);
  for (int l = 0; h->has_mlp && l <= h->n_hidden; ++l) {
    const int in = l == 0 ? h->nx + h->nu : h->hidden[l - 1];
    const int out = l == h->n_hidden ? h->nx : h->hidden[l];
    macs += (double)in * out;
  }
  const int no = h->obs_dim, nu = h->nu;
  const double fcost = 2.0 * (no * no + no) + 2.0 * nu * nu + 2.0 * nu;
  double f = 0, by = 0;
  for (int b = 0; b < p->B; ++b) {
    f += (double)p->N[b] * p->H[b] * (2.0 * macs + fcost);
    const bool writes_eps = p->keep_eps_out || p->lds_eps < 0;
    by += (writes_eps ? 2.0 : 1.0) * h->esz() * (double)p->N[b] * p->H[b] * nu   // noise in (+ out)
          + (double)h->esz() * p->N[b];                                            // costs
    if (p->lds_eps >= 0)   // fused update: per-tile partial sums instead of re-reading the noise
      by += (double)h->esz() * ((p->N[b] + p->tile_m - 1) / p->tile_m) * (p->H[b] * nu + 2);
  }
  by += (double)h->esz() * (macs + 0);
  if (flops) *flops = f;
  if (bytes) *bytes = by;
  return 0;
}

extern "C" int ampc_mppi_plan_set_timing(ampc_mppi_plan* p, int enable) {
  REQUIRE(p, "ampc_mppi_plan_set_timing: NULL plan");
  p->timing = enable != 0;
  p->timing_stride = enable > 1 ? enable : 1;      // (enable = n > 1: every n-th solve is bracketed)
  p->timing_count = 0;
  p->ev_used = 0;
  return 0;
}

extern "C" int ampc_mppi_plan_timing(ampc_mppi_plan* p, double* rollout_ms, double* update_ms,
                                     int* count) {
  REQUIRE(p, "ampc_mppi_plan_timing: NULL plan");
  HIP_OK(hipSetDevice(p->h->device));
  HIP_OK(hipStreamSynchronize(p->h->stream));
  double r = 0, u = 0;
  const size_t n = p->ev_used / 3;
  for (size_t i = 0; i < n; ++i) {
    float a = 0, b = 0;
    HIP_OK(hipEventElapsedTime(&a, p->ev[3 * i], p->ev[3 * i + 1]));
    HIP_OK(hipEventElapsedTime(&b, p->ev[3 * i + 1], p->ev[3 * i + 2]));
    r += a;
    u += b;
  }
  if (rollout_ms) *rollout_ms = n ? r / n : 0.0;
  if (update_ms) *update_ms = n ? u / n : 0.0;
  if (count) *count = (int)n;
  p->ev_used = 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Trajectory scoring (Cost.__call__, cost.py:27-41) on device-resident trajectories
// ---------------------------------------------------------------------------------------------
struct ScoreSpec {
  int n_terms = 0;
  const int* kinds = nullptr;
  const double* params = nullptr;
};

static int score_spec_check(const ScoreSpec& sp, int no, int nu, std::vector<int>* offs, int* total) {
  REQUIRE(sp.n_terms >= 1 && sp.kinds && sp.params, "score: empty cost specification");
  int o = 0;
  offs->clear();
  for (int k = 0; k < sp.n_terms; ++k) {
    const int sz = score_term_size(sp.kinds[k], no, nu);
    REQUIRE(sz >= 0, "score: unknown cost term kind (0 quad, 1 threshold, 2 box)");
    if (sp.kinds[k] == SCORE_THRESHOLD) {
      const double lo = sp.params[o + no], hi = sp.params[o + no + 1];
      REQUIRE(lo >= 0 && hi <= no && lo == (double)(int)lo && hi == (double)(int)hi,
              "score: threshold term obs_range must lie inside [0, obs_dim]");
    }
    offs->push_back(o);
    o += sz;
  }
  *total = o;
  return 0;
}

// d_obs [B][T1][nx], d_ctl [B][T1][nu] in compute precision on h's device -> scores [B] (host)
template <typename T>
static int score_device(ampc_handle* h, const void* d_obs, const void* d_ctl, int B, int T1, int nx,
                        int nu, int no, const ScoreSpec& sp, double* scores) {
  std::vector<int> offs;
  int total = 0;
  if (int rc = score_spec_check(sp, no, nu, &offs, &total)) return rc;
  ScopedBuf d_int, d_par, d_out;
  HIP_OK(d_int.reserve((size_t)2 * sp.n_terms * sizeof(int)));
  HIP_OK(d_par.reserve((size_t)total * sizeof(T)));
  HIP_OK(d_out.reserve((size_t)B * sizeof(T)));
  std::vector<int> ints(sp.kinds, sp.kinds + sp.n_terms);
  ints.insert(ints.end(), offs.begin(), offs.end());
  HIP_OK(hipMemcpyAsync(d_int.p, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_OK(upload_converted<T>(d_par.p, sp.params, (size_t)total, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  hipLaunchKernelGGL(score_trajectories_kernel<T>, dim3(B), dim3(kWG), 0, h->stream, (const T*)d_obs,
                     (const T*)d_ctl, T1, nx, nu, no, sp.n_terms, (const int*)d_int.p,
                     (const int*)d_int.p + sp.n_terms, (const T*)d_par.p, (T*)d_out.p);
  HIP_OK(hipGetLastError());
  HIP_OK(download_converted<T>(scores, d_out.p, (size_t)B, h->stream));
  return 0;
}

template <typename T>
static int score_host_impl(ampc_handle* h, int B, int T1, int nx, int nu, int no, const double* obs,
                           const double* ctrls, const ScoreSpec& sp, double* scores) {
  ScopedBuf d_obs, d_ctl;
  HIP_OK(d_obs.reserve((size_t)B * T1 * nx * sizeof(T)));
  HIP_OK(d_ctl.reserve((size_t)B * T1 * nu * sizeof(T)));
  HIP_OK(upload_converted<T>(d_obs.p, obs, (size_t)B * T1 * nx, h->stream));
  HIP_OK(upload_converted<T>(d_ctl.p, ctrls, (size_t)B * T1 * nu, h->stream));
  int rc = score_device<T>(h, d_obs.p, d_ctl.p, B, T1, nx, nu, no, sp, scores);
  (void)hipStreamSynchronize(h->stream);
  return rc;
}

extern "C" int ampc_score_trajectories(ampc_handle* h, int n_traj, int n_rows, int state_dim,
                                       int obs_dim, int ctrl_dim, const double* obs,
                                       const double* ctrls, int n_terms, const int* kinds,
                                       const double* params, double* scores) {
  REQUIRE(h && obs && ctrls && scores, "ampc_score_trajectories: NULL argument");
  REQUIRE(n_traj >= 1 && n_rows >= 1, "ampc_score_trajectories: empty batch");
  REQUIRE(obs_dim >= 1 && obs_dim <= state_dim && ctrl_dim >= 1,
          "ampc_score_trajectories: need 1 <= obs_dim <= state_dim and ctrl_dim >= 1");
  HIP_OK(hipSetDevice(h->device));
  ScoreSpec sp;
  sp.n_terms = n_terms; sp.kinds = kinds; sp.params = params;
  return h->precision == AMPC_F64
             ? score_host_impl<double>(h, n_traj, n_rows, state_dim, ctrl_dim, obs_dim, obs, ctrls, sp, scores)
             : score_host_impl<float>(h, n_traj, n_rows, state_dim, ctrl_dim, obs_dim, obs, ctrls, sp, scores);
}

// ---------------------------------------------------------------------------------------------
// Closed loop on a surrogate model, device resident (simulate(), utils/simulation.py:11-64)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int closed_loop_impl(ampc_mppi_plan* p, ampc_handle* sur, const double* init_obs, int n_steps,
                            uint64_t seed, const double* eps_all, double* traj_obs,
                            double* traj_ctrls, const ScoreSpec* score = nullptr,
                            double* scores = nullptr) {
  ampc_handle* h = p->h;
  const int nx = h->nx, nu = h->nu, B = p->B, T1 = n_steps + 1;
  legacy_predraw_drop(p);
  // With a state lift (Koopman controller model) the simulation model's state is carried separately:
  // simulate() advances simstate = sim_model.pred(simstate, u) and hands the controller only the
  // observation simstate[:obs_dim], from which update_state re-lifts (simulation.py:52-58,
  // koopman.py:166-168).  snx = width of the carried state = width of the recorded rows.
  const bool lift = p->lift_n > 0;
  const int snx = lift ? sur->nx : nx;
  // (scratch of the loop, kept by the plan: an evaluator that runs an episode in segments -- a user termination
  //  condition -- calls this every few control steps, and every hipFree synchronises the whole device)
  DevBuf &d_obs = p->cl_obs, &d_ctl = p->cl_ctl, &d_next = p->cl_next, &d_sim = p->cl_sim;
  HIP_OK(d_obs.reserve((size_t)B * T1 * snx * sizeof(T)));
  HIP_OK(d_ctl.reserve((size_t)B * T1 * nu * sizeof(T)));
  HIP_OK(d_next.reserve((size_t)B * snx * sizeof(T)));
  if (lift) HIP_OK(d_sim.reserve((size_t)B * snx * sizeof(T)));
  void* state = lift ? d_sim.p : p->x0.p;             // what the surrogate advances
  HIP_OK(hipMemsetAsync(d_ctl.p, 0, (size_t)B * T1 * nu * sizeof(T), h->stream));
  HIP_OK(upload_converted<T>(state, init_obs, (size_t)B * snx, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  // traj_obs[:, 0, :] = init_obs
  HIP_OK(hipMemcpy2DAsync(d_obs.p, (size_t)T1 * snx * sizeof(T), state, (size_t)snx * sizeof(T),
                          (size_t)snx * sizeof(T), B, hipMemcpyDeviceToDevice, h->stream));
  int rc = 0;
  for (int s = 0; s < n_steps && rc == 0; ++s) {
    if (lift) {
      const int n = B * nx;
      hipLaunchKernelGGL(state_lift_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const T*)d_sim.p,
                         (T*)p->x0.p, (const T*)p->lift_prog.p, B, snx, h->obs_dim, p->lift_n);
      HIP_OK(hipGetLastError());
    }
    if (eps_all) {
      rc = mppi_upload_impl<T>(p, nullptr, nullptr, eps_all + (size_t)s * p->sum_nhnu);
    } else {
      rc = mppi_generate_impl<T>(p, seed, p->step_offset + (uint64_t)s);
    }
    if (rc) break;
    rc = mppi_solve_impl<T>(p);
    if (rc) break;
    // x_next = surrogate.pred(x, u)
    rc = surrogate_step<T>(h, sur, state, p->u_out.p, d_next.p, B);
    if (rc) break;
    const int n = B * (snx > nu ? snx : nu);
    hipLaunchKernelGGL(closed_loop_record_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, h->stream,
                       (const T*)d_next.p, (const T*)p->u_out.p, (T*)state, (T*)d_obs.p,
                       (T*)d_ctl.p, B, snx, nu, T1, s);
    HIP_OK(hipGetLastError());
  }
  if (rc == 0) {
    if (traj_obs) rc = download_converted<T>(traj_obs, d_obs.p, (size_t)B * T1 * snx, h->stream) == hipSuccess ? 0 : fail("closed loop: download failed");
    if (rc == 0 && traj_ctrls) rc = download_converted<T>(traj_ctrls, d_ctl.p, (size_t)B * T1 * nu, h->stream) == hipSuccess ? 0 : fail("closed loop: download failed");
  }
  if (rc == 0 && score && scores)
    rc = score_device<T>(h, d_obs.p, d_ctl.p, B, T1, snx, nu, h->obs_dim, *score, scores);
  (void)hipStreamSynchronize(h->stream);
  p->step_offset = 0;      // one-shot: ampc_mppi_plan_set_step_offset names the NEXT closed loop's first step
  return rc;
}

// Which surrogate a plan's closed loop accepts: the controller model's dimensions -- or, with a state
// lift, any model with the same controls whose state starts with the observation.
static int closed_loop_check(const ampc_mppi_plan* p, const ampc_handle* sur, const char* who) {
  const bool ok = sur->has_model() && sur->nu == p->h->nu &&
                  (p->lift_n > 0 ? sur->nx >= p->h->obs_dim : sur->nx == p->h->nx);
  if (!ok) return fail(std::string(who) + ": surrogate model must have the controller model's dimensions (with a "
                                          "state lift: its controls, and a state that starts with the observation)");
  if (sur->precision != p->h->precision || sur->device != p->h->device)
    return fail(std::string(who) + ": surrogate must share the plan's device and precision");
  return 0;
}

extern "C" int ampc_mppi_plan_set_state_lift(ampc_mppi_plan* p, int n_basis, const int* kinds, const double* params) {
  REQUIRE(p, "ampc_mppi_plan_set_state_lift: NULL plan");
  if (n_basis <= 0) { p->lift_n = 0; return 0; }
  REQUIRE(kinds && params, "ampc_mppi_plan_set_state_lift: NULL argument");
  REQUIRE(p->h->obs_dim >= 1 && n_basis * p->h->obs_dim == p->h->nx,
          "ampc_mppi_plan_set_state_lift: n_basis * obs_dim must be the model's state dimension");
  std::vector<double> prog(2 * (size_t)n_basis);
  for (int k = 0; k < n_basis; ++k) {
    REQUIRE(kinds[k] >= 0 && kinds[k] <= 3, "ampc_mppi_plan_set_state_lift: kind must be 0 identity, 1 power, 2 sin, 3 cos");
    REQUIRE(kinds[k] != 1 || (params[k] >= 0 && params[k] <= 64 && params[k] == std::floor(params[k])),
            "ampc_mppi_plan_set_state_lift: powers must be integers in 0..64");
    prog[2 * k] = kinds[k]; prog[2 * k + 1] = params[k];
  }
  HIP_OK(hipSetDevice(p->h->device));
  HIP_OK(p->lift_prog.reserve(prog.size() * p->h->esz()));
  if (p->h->precision == AMPC_F64) HIP_OK(upload_converted<double>(p->lift_prog.p, prog.data(), prog.size(), p->h->stream));
  else HIP_OK(upload_converted<float>(p->lift_prog.p, prog.data(), prog.size(), p->h->stream));
  HIP_OK(hipStreamSynchronize(p->h->stream));
  p->lift_n = n_basis;
  return 0;
}

extern "C" int ampc_mppi_closed_loop(ampc_mppi_plan* p, ampc_handle* surrogate,
                                     const double* init_obs, int n_steps, uint64_t seed,
                                     const double* eps_all, double* traj_obs, double* traj_ctrls) {
  REQUIRE(p && init_obs, "ampc_mppi_closed_loop: NULL argument");
  REQUIRE(n_steps >= 1, "ampc_mppi_closed_loop: n_steps < 1");
  ampc_handle* sur = surrogate ? surrogate : p->h;
  if (int rc = closed_loop_check(p, sur, "ampc_mppi_closed_loop")) return rc;
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? closed_loop_impl<double>(p, sur, init_obs, n_steps, seed, eps_all, traj_obs, traj_ctrls)
             : closed_loop_impl<float>(p, sur, init_obs, n_steps, seed, eps_all, traj_obs, traj_ctrls);
}

extern "C" int ampc_mppi_closed_loop_scored(ampc_mppi_plan* p, ampc_handle* surrogate,
                                            const double* init_obs, int n_steps, uint64_t seed,
                                            const double* eps_all, int n_terms, const int* kinds,
                                            const double* params, double* scores, double* traj_obs,
                                            double* traj_ctrls) {
  REQUIRE(p && init_obs && scores, "ampc_mppi_closed_loop_scored: NULL argument");
  REQUIRE(n_steps >= 1, "ampc_mppi_closed_loop_scored: n_steps < 1");
  ampc_handle* sur = surrogate ? surrogate : p->h;
  if (int rc = closed_loop_check(p, sur, "ampc_mppi_closed_loop_scored")) return rc;
  ScoreSpec sp;
  sp.n_terms = n_terms; sp.kinds = kinds; sp.params = params;
  std::vector<int> offs;
  int total = 0;
  if (int rc = score_spec_check(sp, p->h->obs_dim, p->h->nu, &offs, &total)) return rc;
  HIP_OK(hipSetDevice(p->h->device));
  return p->h->precision == AMPC_F64
             ? closed_loop_impl<double>(p, sur, init_obs, n_steps, seed, eps_all, traj_obs, traj_ctrls, &sp, scores)
             : closed_loop_impl<float>(p, sur, init_obs, n_steps, seed, eps_all, traj_obs, traj_ctrls, &sp, scores);
}

extern "C" int ampc_mppi_plan_set_outputs(ampc_mppi_plan* p, int keep_eps_out) {
  REQUIRE(p, "ampc_mppi_plan_set_outputs: NULL plan");
  p->keep_eps_out = keep_eps_out != 0;
  return 0;
}
