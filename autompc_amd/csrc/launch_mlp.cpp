// launch_mlp.cpp -- Model.pred_batch / pred_diff_batch, the surrogate step and the iLQR Jacobian refresh
// Compiled once per precision (-DAMPC_T=double|float, csrc/build.py); the explicit instantiations
// at the end are what api.cpp links against.
#include "host_common.hpp"

#ifndef AMPC_T
#error "compile with -DAMPC_T=double or -DAMPC_T=float"
#endif

#ifndef AMPC_JIT_PLUGIN       // (a shape plugin only carries the launchers that have static variants)
// ---------------------------------------------------------------------------------------------
// Model.pred_batch / pred_diff_batch
// ---------------------------------------------------------------------------------------------
template <typename T>
int pred_impl(ampc_handle* h, const double* states, const double* ctrls, double* out,
                     double* jx, double* ju, int n) {
  const MlpDev<T>& m = model_of<T>(h);
  const int nx = h->nx, nu = h->nu;
  const bool deriv = jx != nullptr;
  HIP_OK(h->s_states.reserve((size_t)n * nx * sizeof(T)));
  HIP_OK(h->s_ctrls.reserve((size_t)n * nu * sizeof(T)));
  HIP_OK(h->s_out.reserve((size_t)n * nx * sizeof(T)));
  HIP_OK(upload_converted<T>(h->s_states.p, states, (size_t)n * nx, h->stream));
  HIP_OK(upload_converted<T>(h->s_ctrls.p, ctrls, (size_t)n * nu, h->stream));
  const int mt = choose_mt<T>(h, m, n, 0);
  const int M = 16 * mt;
  const int tiles = (n + M - 1) / M;
  const int n_pad = tiles * M;
  TileLds L = tile_lds_for<T>(h, m, M, 0);
  const size_t lds_bytes = (size_t)L.extra * sizeof(T);
  if (deriv) HIP_OK(h->s_dz.reserve((size_t)m.n_hidden * n_pad * m.hpad * sizeof(T)));
  T* dz = (T*)h->s_dz.p;
  const RowMap rm{n, 0, 0, nullptr};
  AMPC_DISPATCH(h, mt, {
    if (deriv) {
      auto k = mlp_forward_kernel<T, NT, MT, W, true, DynShape, WD>;
      HIP_OK(allow_lds(k, lds_bytes));
      hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * W), lds_bytes, h->stream, m, L,
                         (const T*)h->s_states.p, (const T*)h->s_ctrls.p, (T*)h->s_out.p, dz, n,
                         n_pad, rm);
    } else {
      auto k = mlp_forward_kernel<T, NT, MT, W, false, DynShape, WD>;
      HIP_OK(allow_lds(k, lds_bytes));
      hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * W), lds_bytes, h->stream, m, L,
                         (const T*)h->s_states.p, (const T*)h->s_ctrls.p, (T*)h->s_out.p,
                         (T*)nullptr, n, n_pad, rm);
    }
  });
  HIP_OK(hipGetLastError());
  if (deriv) {
    HIP_OK(h->s_jx.reserve((size_t)n * nx * nx * sizeof(T)));
    HIP_OK(h->s_ju.reserve((size_t)n * nx * nu * sizeof(T)));
    const int jmt = 1;
    const int JM = 16 * jmt;
    const int jtiles = ((n + JM - 1) / JM) * nx;       // (sample block, output index) tiles
    const int kinp = 16 * ((m.kin + 15) / 16);
    const size_t jl = (size_t)JM * imax(m.hpad + 2, h->nw * kinp) * sizeof(T);
    AMPC_DISPATCH(h, jmt, {
      auto k = mlp_jacobian_kernel<T, NT, MT, W, DynShape, WD>;
      HIP_OK(allow_lds(k, jl));
      hipLaunchKernelGGL(k, dim3(jtiles), dim3(64 * W), jl, h->stream, m,
                         (const T*)h->wout_plain, (const T*)dz, n, n_pad, (T*)h->s_jx.p,
                         (T*)h->s_ju.p, rm);
    });
    HIP_OK(hipGetLastError());
    HIP_OK(download_converted<T>(jx, h->s_jx.p, (size_t)n * nx * nx, h->stream));
    HIP_OK(download_converted<T>(ju, h->s_ju.p, (size_t)n * nx * nu, h->stream));
  }
  HIP_OK(download_converted<T>(out, h->s_out.p, (size_t)n * nx, h->stream));
  return 0;
}

#endif  // AMPC_JIT_PLUGIN

// Jacobians of every (problem, t) row of the nominal trajectories whose problem asked for it.
template <typename T> int ilqr_refresh_jacobians(ampc_ilqr_plan* p) {
#ifndef AMPC_JIT_PLUGIN
  if (p->jit) return jit_result(p->jit, p->jit->ilqr_refresh(p));
#endif
  ampc_handle* h = p->h;
  const MlpDev<T>& m = model_of<T>(h);
  const int nx = h->nx, nu = h->nu;
  // slots that carry different controller models (ampc_ilqr_plan_set_models): groups padded to whole
  // 16-row tiles, so that a tile's rows share one model
  const bool per_slot = p->queue_on && p->var_model;
  const int gpad = per_slot ? round_up(p->H, 16) : 0;
  const int rows = ilqr_grid_slots(p) * (per_slot ? gpad : p->H);
  const int n_pad = round_up(rows, 64);
  const RowMap rm{p->H, (long long)(p->H + 1) * nx, (long long)p->H * nu, (const int*)p->flags.p + 4 * p->B,
                  (p->queue_on && p->var_h) ? (const int*)p->slot_h.p : nullptr,
                  gpad, per_slot ? (const int*)p->slot_model.p : nullptr, per_slot ? (const long long*)p->mlp_tab.p : nullptr,
                  p->B, (p->queue_on && p->compact_on) ? (const int*)p->slot_of.p : nullptr};
  if (per_slot) HIP_OK(p->dz.reserve((size_t)m.n_hidden * n_pad * m.hpad * sizeof(T)));
#ifndef AMPC_JIT_PLUGIN
  if (h->has_lin) {               // a linear model's Jacobians are constant ([A | B], LinDev::jp): nothing to refresh
    if (p->ev_cur) { HIP_OK(hipEventRecord(p->ev_cur[3], h->stream)); HIP_OK(hipEventRecord(p->ev_cur[4], h->stream)); }
    return 0;
  }
  if (h->has_sindy) {
    hipLaunchKernelGGL(sindy_jacobian_kernel<T>, dim3((rows + 63) / 64), dim3(64), 0, h->stream,
                       sindy_of<T>(h), (const T*)p->states.p, (const T*)p->ctrls.p, (T*)p->jx.p,
                       (T*)p->ju.p, rows, rm);
    HIP_OK(hipGetLastError());
    if (p->ev_cur) { HIP_OK(hipEventRecord(p->ev_cur[3], h->stream)); HIP_OK(hipEventRecord(p->ev_cur[4], h->stream)); }
    return 0;
  }
#endif
  {
    const int mt = 1, M = 16, tiles = (rows + M - 1) / M;
    TileLds L = tile_lds_for<T>(h, m, M, 0);
    const size_t lb = (size_t)L.extra * sizeof(T);
    const TileLds S = tile_lds_dims((int)sizeof(T), m.hpad, m.k1p, m.nxp, m.n_hidden, M, h->nw, true, true);
    if (p->static_shape >= 0 && std::memcmp(&S, &L, sizeof(TileLds)) == 0) {
#define AMPC_SD_BODY { auto k = mlp_forward_kernel<T, NT, 1, W, true, SH>; HIP_OK(allow_lds(k, lb));          \
      hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * W), lb, h->stream, m, L, (const T*)p->states.p,       \
                         (const T*)p->ctrls.p, (T*)nullptr, (T*)p->dz.p, rows, n_pad, rm); }
      AMPC_STATIC_DISPATCH(p->static_shape, h->act == 0);
#undef AMPC_SD_BODY
    } else {
#ifdef AMPC_JIT_PLUGIN
      (void)mt;
      return fail("shape plugin entered without its static shape");
#else
      AMPC_DISPATCH(h, mt, {
        auto k = mlp_forward_kernel<T, NT, MT, W, true, DynShape, WD>;
        HIP_OK(allow_lds(k, lb));
        hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * W), lb, h->stream, m, L, (const T*)p->states.p,
                           (const T*)p->ctrls.p, (T*)nullptr, (T*)p->dz.p, rows, n_pad, rm);
      });
#endif
    }
  }
  if (p->ev_cur) HIP_OK(hipEventRecord(p->ev_cur[3], h->stream));
  {
    // 16-row tiles: their 33 KB of LDS lets four workgroups share a CU, so one tile's set-up
    // (global loads of dz / W_out) and reduction overlap another's MFMAs; measured 4 % faster on
    // c4 than 32- or 64-row tiles (AMPC_JMT overrides for experiments).
    const int kinp = 16 * ((m.kin + 15) / 16);
    int jmt = env_int("AMPC_JMT", 0);
    if (jmt == 0 || per_slot) jmt = 1;          // (per-slot models: groups are padded to 16-row tiles)
    while (jmt > 1 && (size_t)16 * jmt * imax(m.hpad + 2, h->nw * kinp) * sizeof(T) > kLdsLimit) jmt /= 2;
    const int JM = 16 * jmt, jtiles = ((rows + JM - 1) / JM) * nx;   // (sample block, output) tiles
    const size_t jl = (size_t)JM * imax(m.hpad + 2, h->nw * kinp) * sizeof(T);
    if (p->static_shape >= 0 && jmt == 1) {
#define AMPC_SD_BODY { auto k = mlp_jacobian_kernel<T, NT, 1, W, SH>; HIP_OK(allow_lds(k, jl));               \
      hipLaunchKernelGGL(k, dim3(jtiles), dim3(64 * W), jl, h->stream, m, (const T*)h->wout_plain,       \
                         (const T*)p->dz.p, rows, n_pad, (T*)p->jx.p, (T*)p->ju.p, rm); }
      AMPC_STATIC_DISPATCH(p->static_shape, 0);       // (the chain multiplies stored derivatives)
#undef AMPC_SD_BODY
    } else {
#ifdef AMPC_JIT_PLUGIN
      return fail("shape plugin: Jacobian tile height other than 16 (AMPC_JMT) needs the library's kernels");
#else
      AMPC_DISPATCH(h, jmt, {
        auto k = mlp_jacobian_kernel<T, NT, MT, W, DynShape, WD>;
        HIP_OK(allow_lds(k, jl));
        hipLaunchKernelGGL(k, dim3(jtiles), dim3(64 * W), jl, h->stream, m, (const T*)h->wout_plain,
                           (const T*)p->dz.p, rows, n_pad, (T*)p->jx.p, (T*)p->ju.p, rm);
      });
#endif
    }
  }
  HIP_OK(hipGetLastError());
  if (p->ev_cur) HIP_OK(hipEventRecord(p->ev_cur[4], h->stream));
  return 0;
}

#ifndef AMPC_JIT_PLUGIN

// x_next[b] = surrogate.pred(x[b], u[b]) for B rows, all device pointers, enqueued on h's stream.
template <typename T>
int surrogate_step(ampc_handle* h, ampc_handle* sur, const void* x, const void* u, void* x_next, int B) {
  if (sur->has_lin) {
    const LinDev<T> lm = lin_of<T>(sur);
    const size_t lb = (size_t)16 * lin_xs(lm.kp, (int)sizeof(T)) * sizeof(T);
    HIP_OK(allow_lds(linear_forward_kernel<T>, lb));
    hipLaunchKernelGGL(linear_forward_kernel<T>, dim3((B + 15) / 16), dim3(64 * kLinW), lb, h->stream, lm,
                       (const T*)x, (const T*)u, (T*)x_next, B);
    HIP_OK(hipGetLastError());
    return 0;
  }
  if (sur->has_sindy) {
    const SindyDev<T> sd = sindy_of<T>(sur);
    const size_t lb = (size_t)(2 * sur->nx + sur->nu + sur->s_ntab) * 64 * sizeof(T) + sindy_stage_bytes<T>(sur);
    HIP_OK(allow_lds(sindy_forward_kernel<T>, lb));
    hipLaunchKernelGGL(sindy_forward_kernel<T>, dim3((B + 63) / 64), dim3(64), lb, h->stream, sd,
                       (const T*)x, (const T*)u, (T*)x_next, B);
    HIP_OK(hipGetLastError());
    return 0;
  }
  const MlpDev<T>& sm = model_of<T>(sur);
  const int SM = 16, stiles = (B + SM - 1) / SM;
  TileLds SL = tile_lds_for<T>(sur, sm, SM, 0);
  const size_t slds = (size_t)SL.extra * sizeof(T);
  const RowMap rm{B, 0, 0, nullptr};
  AMPC_DISPATCH(sur, 1, {
    auto k = mlp_forward_kernel<T, NT, MT, W, false, DynShape, WD>;
    HIP_OK(allow_lds(k, slds));
    hipLaunchKernelGGL(k, dim3(stiles), dim3(64 * W), slds, h->stream, sm, SL, (const T*)x,
                       (const T*)u, (T*)x_next, (T*)nullptr, B, stiles * SM, rm);
  });
  HIP_OK(hipGetLastError());
  return 0;
}

template int pred_impl<AMPC_T>(ampc_handle*, const double*, const double*, double*, double*, double*, int);
template int surrogate_step<AMPC_T>(ampc_handle*, ampc_handle*, const void*, const void*, void*, int);
#endif  // AMPC_JIT_PLUGIN
template int ilqr_refresh_jacobians<AMPC_T>(ampc_ilqr_plan*);
