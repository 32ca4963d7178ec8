// launch_ilqr.cpp -- one iLQR iteration: backward Riccati sweep + batched line search
// Compiled once per precision (-DAMPC_T=double|float, csrc/build.py); the explicit instantiations
// at the end are what api.cpp links against.
#include "host_common.hpp"

#ifndef AMPC_T
#error "compile with -DAMPC_T=double or -DAMPC_T=float"
#endif

template <typename T> int ilqr_launch_iter(ampc_ilqr_plan* p, int mode) {
#ifndef AMPC_JIT_PLUGIN
  if (p->var_model && p->static_shape < 0 && !p->jit)
    return fail("iLQR plan: per-slot controller models need the shape-specialised kernels (the run-time-shape kernels "
                "take one model per plan); call ampc_ilqr_plan_set_models on a plan of a registered / compiled shape");
  if (p->jit) return jit_result(p->jit, p->jit->ilqr_iter(p, mode));
#endif
  ampc_handle* h = p->h;
  IlqrArgs<T> a = make_ilqr_args<T>(p, mode);
  const int Bg = ilqr_grid_slots(p);        // (slots with work first: the grid ends where they end)
  hipEvent_t* e = mode == 1 ? p->ev_cur : nullptr;
  if (e) HIP_OK(hipEventRecord(e[0], h->stream));
#ifndef AMPC_JIT_PLUGIN
  if (h->has_lin) {     // wide linear models: ilqr_wide.hpp + the sixteen-row line search on the K-tiled linear step
    if (mode == 1) {
      const size_t wb = (size_t)make_wide_lds(h->nx, h->nu, h->obs_dim).total * sizeof(T);
#define AMPC_WIDE_NU(NUV)                                                                          \
      case NUV: { auto rk = ilqr_riccati_wide_kernel<T, NUV>; HIP_OK(allow_lds(rk, wb));              \
        hipLaunchKernelGGL(rk, dim3(Bg), dim3(kRicThreads), wb, h->stream, a); } break;
      switch (h->nu) {
        AMPC_WIDE_NU(1) AMPC_WIDE_NU(2) AMPC_WIDE_NU(3) AMPC_WIDE_NU(4) AMPC_WIDE_NU(6) AMPC_WIDE_NU(8)
        default: return fail("internal: control dimension of a wide linear iLQR plan");
      }
#undef AMPC_WIDE_NU
      HIP_OK(hipGetLastError());
    }
    if (e) HIP_OK(hipEventRecord(e[1], h->stream));
    auto k = ilqr_iter_kernel<T, 1, 8, 2>;
    HIP_OK(allow_lds(k, p->lds_bytes));
    hipLaunchKernelGGL(k, dim3(Bg), dim3(64 * 8), p->lds_bytes, h->stream, a);
    HIP_OK(hipGetLastError());
    if (e) HIP_OK(hipEventRecord(e[2], h->stream));
    return 0;
  }
#endif
  if (mode == 1) {      // backward sweep first: gains + expected reduction for the line search
    const IlqrWork wk = make_ilqr_work(h->nx, h->nu, h->cost_stride);
    const size_t rb = (size_t)wk.total * sizeof(T);
    // latency-optimised sweep (MFMA products, per-lane LU): model states <= 32 and the control
    // dimensions it is instantiated for; everything else takes the general kernel
    const size_t mb = (size_t)make_ric_lds(h->nx, h->nu, h->obs_dim).total * sizeof(T);
    const bool mfma_sweep = p->use_mfma_sweep != 0;
#define AMPC_RIC_NU(NUV, SHT)                                                                        \
    case NUV: { auto rk = ilqr_riccati_mfma_kernel<T, NUV, SHT>; HIP_OK(allow_lds(rk, mb));          \
      hipLaunchKernelGGL(rk, dim3(Bg), dim3(kRicThreads), mb, h->stream, a); done = true; break; }
    bool done = false;
    // (the control dimensions the latency-optimised sweep is validated for; a shape-specialised
    //  build takes the same decision as the run-time-shape one, so both give identical results)
    const bool nu_ok = h->nu == 1 || h->nu == 2 || h->nu == 3 || h->nu == 4 || h->nu == 6 || h->nu == 8;
    if (mfma_sweep && h->nx <= 32 && nu_ok) {
      if (p->static_shape >= 0) {
#define AMPC_SD_BODY { auto rk = ilqr_riccati_mfma_kernel<T, SH::nu, SH>; HIP_OK(allow_lds(rk, mb));   \
        hipLaunchKernelGGL(rk, dim3(Bg), dim3(kRicThreads), mb, h->stream, a); done = true; }
        AMPC_STATIC_DISPATCH(p->static_shape, 0);    // (the sweep never evaluates the activation)
#undef AMPC_SD_BODY
      } else {
#ifndef AMPC_JIT_PLUGIN
        switch (h->nu) {
          AMPC_RIC_NU(1, DynShape) AMPC_RIC_NU(2, DynShape) AMPC_RIC_NU(3, DynShape)
          AMPC_RIC_NU(4, DynShape) AMPC_RIC_NU(6, DynShape) AMPC_RIC_NU(8, DynShape)
          default: break;
        }
#endif
      }
    }
#undef AMPC_RIC_NU
    if (done) {
    } else if (p->static_shape >= 0) {
#define AMPC_SD_BODY { auto rk = ilqr_riccati_kernel<T, false, SH>; HIP_OK(allow_lds(rk, rb));   \
      hipLaunchKernelGGL(rk, dim3(Bg), dim3(kRicThreads), rb, h->stream, a); }
      AMPC_STATIC_DISPATCH(p->static_shape, 0);
#undef AMPC_SD_BODY
    }
#ifdef AMPC_JIT_PLUGIN
    else return fail("shape plugin entered without its static shape");
#else
    else if (h->nx > 32) {
      auto rk = ilqr_riccati_kernel<T, true>;
      HIP_OK(allow_lds(rk, rb));
      hipLaunchKernelGGL(rk, dim3(Bg), dim3(kRicThreads), rb, h->stream, a);
    } else {
      auto rk = ilqr_riccati_kernel<T, false>;
      HIP_OK(allow_lds(rk, rb));
      hipLaunchKernelGGL(rk, dim3(Bg), dim3(kRicThreads), rb, h->stream, a);
    }
#endif
    HIP_OK(hipGetLastError());
  }
  if (e) HIP_OK(hipEventRecord(e[1], h->stream));
#ifndef AMPC_JIT_PLUGIN
  if (h->has_sindy) {
    auto k = ilqr_iter_kernel<T, 1, 4, 1>;
    HIP_OK(allow_lds(k, p->lds_bytes));
    hipLaunchKernelGGL(k, dim3(Bg), dim3(256), p->lds_bytes, h->stream, a);
    HIP_OK(hipGetLastError());
    if (e) HIP_OK(hipEventRecord(e[2], h->stream));
    return 0;
  }
#endif
  // f64 MLP models with <= 32 states: candidates four at a time on 4x4x4 MFMA tiles (ilqr_ls4.hpp)
  if constexpr (sizeof(T) == 8) {
    // one hidden -> hidden layer: that layer partly resident on chip (registers + LDS)
    const bool res = h->n_hidden == 2;
    const bool ls4_ok = p->use_ls4 && !h->has_sindy && h->nx <= 32;
    size_t lb = ls4_ok
        ? (size_t)make_ls4_lds(h->nu, h->k1p, h->nxp, h->hpad, h->n_hidden, res, h->cost_stride).total * sizeof(T)
        : 0;
    // (a wide first layer with many controls can push the resident LDS copies past 160 KB: those
    //  shapes take the general kernel below)
    if (lb > 0 && lb <= kLdsLimit) {
      // few problems: the passes of a line search side by side on otherwise idle CUs
      int npass = (p->ls_n + 3) / 4;
      // (retired problems' workgroups exit at once, so what has to fit is the ACTIVE problems' passes)
      const int live = (p->active_hint > 0 && p->active_hint < p->B) ? p->active_hint : p->B;
      a.par_passes = (mode == 1 && p->par_passes && live * npass <= h->n_cus) ? 1 : 0;
      // many problems: the line search in two launches -- pass 0 for everybody, then the remaining
      // passes side by side for the problems it left undecided (ilqr_ls4.hpp)
      const bool split = mode == 1 && !a.par_passes && p->ls_split && npass > 1;
      // many problems in lock-step: all step sizes in one pass of a twelve-row tile (ilqr_lsw.hpp) once
      // some search of the batch needs a third four-row pass (the launch lasts as long as its slowest
      // search); the host picks from the slots' last searches (ls_need), see ampc_ilqr_plan::ls_rb.
      // (A whole-batch rollout of the guess, mode 0 without per-slot modes, has one row per problem.)
      int rb = 1;
      const bool wide = p->ls_rb == 3 || (p->ls_rb == 0 && p->ls_rb_now == 3);
      if (!a.par_passes && !split && wide && npass > 1 && (mode == 1 || a.slot_mode)) {
        const size_t lb3 = (size_t)make_ls4_lds(h->nu, h->k1p, h->nxp, h->hpad, h->n_hidden, res, h->cost_stride, 3)
                               .total * sizeof(T);
        if (lb3 <= kLdsLimit) { rb = 3; lb = lb3; npass = (p->ls_n + 11) / 12; }
      }
      auto launch_ls = [&](const dim3 grid, const IlqrArgs<T>& a) -> int {
        if (p->static_shape >= 0) {
          // (the activation is a compile-time constant for relu AND tanh here: with the run-time
          //  switch the epilogue keeps all five activations' temporaries alive and the kernel falls
          //  off its register budget -- spills inside the time loop, 2x slower)
#define AMPC_LS4_SHAPE(ID, NX, NU, NO, NH, HPAD, ACTV)                                            \
          case (ID) * 8 + ((ACTV) < 0 ? 7 : (ACTV)): {                                               \
            using SH = StaticShape<NX, NU, NO, NH, HPAD, ACTV>;                                      \
            auto k = rb == 3 ? ilqr_lsw_kernel<HPAD / 64, NH == 2, SH, 3>                            \
                             : ilqr_ls4_kernel<HPAD / 64, NH == 2, SH>; HIP_OK(allow_lds(k, lb));       \
            hipLaunchKernelGGL(k, grid, dim3(64 * kLs4W), lb, h->stream, a); } break;
#define AMPC_LS4_ONE(ID, NX, NU, NO, NH, HPAD)                                                    \
          AMPC_LS4_SHAPE(ID, NX, NU, NO, NH, HPAD, 0) AMPC_LS4_SHAPE(ID, NX, NU, NO, NH, HPAD, 1)    \
          AMPC_LS4_SHAPE(ID, NX, NU, NO, NH, HPAD, -1)
          switch (p->static_shape * 8 + ((h->act == 0 || h->act == 1) ? h->act : 7)) {
            AMPC_STATIC_SHAPES(AMPC_LS4_ONE)
            default: return fail("internal: unknown static shape");
          }
#undef AMPC_LS4_ONE
#undef AMPC_LS4_SHAPE
        } else {
#ifdef AMPC_JIT_PLUGIN
          return fail("shape plugin entered without its static shape");
#else
#define AMPC_LS4_CASE(NTV, RESV)                                                               \
          case (NTV) * 2 + (RESV): { auto k = rb == 3 ? ilqr_lsw_kernel<NTV, (RESV) != 0, DynShape, 3>  \
                                                      : ilqr_ls4_kernel<NTV, (RESV) != 0, DynShape>;    \
            HIP_OK(allow_lds(k, lb));                                                            \
            hipLaunchKernelGGL(k, grid, dim3(64 * kLs4W), lb, h->stream, a); } break;
          switch ((h->hpad / 64) * 2 + (res ? 1 : 0)) {
            AMPC_LS4_CASE(1, 0) AMPC_LS4_CASE(1, 1) AMPC_LS4_CASE(2, 0) AMPC_LS4_CASE(2, 1)
            AMPC_LS4_CASE(3, 0) AMPC_LS4_CASE(3, 1) AMPC_LS4_CASE(4, 0) AMPC_LS4_CASE(4, 1)
            default: return fail("internal: unsupported hidden width for the four-row line search");
          }
#undef AMPC_LS4_CASE
#endif
        }
        return 0;
      };
      a.ls_split = split ? 1 : 0;
      if (int rc = launch_ls(dim3(Bg, a.par_passes ? npass : 1), a)) return rc;
      if (split) {
        HIP_OK(hipGetLastError());
        a.ls_split = 2;
        if (int rc = launch_ls(dim3(Bg, npass - 1), a)) return rc;
      }
      HIP_OK(hipGetLastError());
      if (e) HIP_OK(hipEventRecord(e[2], h->stream));
      return 0;
    }
  }
  if (p->static_shape >= 0) {
#define AMPC_SD_BODY { auto k = ilqr_iter_kernel<T, NT, W, 0, SH>; HIP_OK(allow_lds(k, p->lds_bytes));   \
      hipLaunchKernelGGL(k, dim3(Bg), dim3(64 * W), p->lds_bytes, h->stream, a); }
    AMPC_STATIC_DISPATCH(p->static_shape, h->act == 0);
#undef AMPC_SD_BODY
  } else {
#ifdef AMPC_JIT_PLUGIN
    return fail("shape plugin entered without its static shape");
#else
    AMPC_DISPATCH(h, 1, {
      auto k = ilqr_iter_kernel<T, NT, W, 0, DynShape, WD>;
      HIP_OK(allow_lds(k, p->lds_bytes));
      hipLaunchKernelGGL(k, dim3(Bg), dim3(64 * W), p->lds_bytes, h->stream, a);
    });
#endif
  }
  HIP_OK(hipGetLastError());
  if (e) HIP_OK(hipEventRecord(e[2], h->stream));
  return 0;
}

AMPC_PROBE_HOST_ILQR      // (timing-experiment builds only: read-back of the marks, probe.hpp)

template int ilqr_launch_iter<AMPC_T>(ampc_ilqr_plan*, int);
