// launch_mppi.cpp -- one MPPI solve: rollout kernel + softmin update
// Compiled once per precision (-DAMPC_T=double|float, csrc/build.py); the explicit instantiations
// at the end are what api.cpp links against.
#include "host_common.hpp"

#ifndef AMPC_T
#error "compile with -DAMPC_T=double or -DAMPC_T=float"
#endif

template <typename T> int mppi_solve_impl(ampc_mppi_plan* p) {
#ifndef AMPC_JIT_PLUGIN
  // indicator cost terms (set on the handle, possibly after the plan was built) live in the run-time-shape
  // kernels only (mppi_kernels.hpp): same tiles, same LDS map, same arithmetic
  // (a plan that holds a table of controller models -- ampc_mppi_plan_set_models -- runs the shape-specialised
  //  kernels only: the run-time-shape kernels do not take the per-problem model offset and would roll every
  //  problem out on the plan handle's model without saying so)
  if (!p->models.empty() && p->h->n_ind > 0)
    return fail("ampc_mppi_solve: the plan holds a table of controller models and its handle has indicator cost terms; "
                "indicator terms run on the run-time-shape kernels, which take one model per plan");
  if ((p->static_shape >= 0 || p->jit) && p->h->n_ind > 0) { p->static_shape = -1; p->jit = nullptr; }
  if (!p->models.empty() && p->static_shape < 0 && !p->jit)
    return fail("ampc_mppi_solve: the plan holds a table of controller models but is no longer on the shape-specialised "
                "kernels (its geometry was rebuilt?): call ampc_mppi_plan_set_models again after ampc_mppi_plan_set_geometry");
  if (p->jit) return jit_result(p->jit, p->jit->mppi_solve(p));     // same function, compiled for the shape
#endif
  ampc_handle* h = p->h;
  MppiArgs<T> a = make_args<T>(p);
  hipEvent_t* e = nullptr;
  if (p->timing && (p->timing_count++ % p->timing_stride) == 0) {
    if (p->ev_used + 3 > p->ev.size()) {
      for (int i = 0; i < 3; ++i) {
        hipEvent_t x;
        HIP_OK(hipEventCreate(&x));
        p->ev.push_back(x);
      }
    }
    e = &p->ev[p->ev_used];
    p->ev_used += 3;
    HIP_OK(hipEventRecord(e[0], h->stream));
  }
  if constexpr (sizeof(T) == 8) {
    if (p->quad) {              // four-row tiles (mppi_rollout4.hpp): small problems, f64
      void (*k)(MppiArgs<double>) = nullptr;
      if (p->static_shape >= 0) {
        switch (p->static_shape * 2 + (h->act == 0 ? 1 : 0)) {
#define AMPC_SHAPE_QUAD_ONE(ID, NX, NU, NO, NH, HPAD, RELU)                                              \
          case ID * 2 + RELU:                                                                            \
            if constexpr (q4_supported(HPAD, NH, (NX + 15) / 16 * 16, (NX + NU + 7) / 8 * 8))            \
              k = mppi_rollout4_kernel<HPAD / 64, NH, StaticShape<NX, NU, NO, NH, HPAD, RELU ? 0 : -1, 0>>; \
            break;
#define AMPC_SHAPE_QUAD(ID, NX, NU, NO, NH, HPAD)     \
          AMPC_SHAPE_QUAD_ONE(ID, NX, NU, NO, NH, HPAD, 0) \
          AMPC_SHAPE_QUAD_ONE(ID, NX, NU, NO, NH, HPAD, 1)
          AMPC_STATIC_SHAPES(AMPC_SHAPE_QUAD)
#undef AMPC_SHAPE_QUAD
#undef AMPC_SHAPE_QUAD_ONE
          default: break;
        }
        if (!k) return fail("internal: four-row rollout for an unsupported static shape");
      } else {
#ifdef AMPC_JIT_PLUGIN
        return fail("shape plugin entered without its static shape");
#else
        switch (h->hpad / 64 * 10 + h->n_hidden) {
          case 11: k = mppi_rollout4_kernel<1, 1>; break;
          case 12: k = mppi_rollout4_kernel<1, 2>; break;
          case 13: k = mppi_rollout4_kernel<1, 3>; break;
          case 14: k = mppi_rollout4_kernel<1, 4>; break;
          case 21: k = h->nxp <= 16 ? mppi_rollout4_kernel<2, 1, DynShape, true> : mppi_rollout4_kernel<2, 1>; break;
          case 22: k = h->nxp <= 16 ? mppi_rollout4_kernel<2, 2, DynShape, true> : mppi_rollout4_kernel<2, 2>; break;
          default: return fail("internal: four-row rollout for an unsupported shape");
        }
#endif
      }
      HIP_OK(allow_lds(k, p->lds_bytes));
      hipLaunchKernelGGL(k, dim3(p->n_tiles), dim3(256), p->lds_bytes, h->stream, a);
    }
  }
  if (p->quad) {
    if (sizeof(T) != 8) return fail("internal: four-row rollout in f32");      // (launched above)
  } else
#ifndef AMPC_JIT_PLUGIN
  if (h->has_lin) {               // wide linear model (linear_kernels.hpp)
    HIP_OK(allow_lds(linear_rollout_kernel<T>, p->lds_bytes));
    hipLaunchKernelGGL(linear_rollout_kernel<T>, dim3(p->n_tiles), dim3(64 * kLinW), p->lds_bytes, h->stream, a,
                       lin_of<T>(h));
  } else if (h->has_sindy) {
    const SindyDev<T> sm = sindy_of<T>(h);
    // features spread over sixteen lanes per sample (sindy_kernels.hpp) when the library is staged in LDS
    // and has a product table and at most eight states; AMPC_SINDY_FP = 0: one thread per sample
    const bool fp_allowed = env_int("AMPC_SINDY_FP", 1) != 0;
    // lanes per sample: all 64 while that still leaves the chip waves to spare, else 32 / 16
    const long long samples = (long long)p->n_tiles * 64;
    int G = samples <= 2048 ? 64 : (samples <= 8192 ? 32 : 16);
    { const int g = env_int("AMPC_SINDY_G", 0); if (g == 16 || g == 32 || g == 64) G = g; }
    const int hcap = (p->max_h * h->nu + 3) / 4 * 4;
    const size_t lbf = ((size_t)(64 / G) * (h->nx + h->nu + h->s_ntab + 2 * hcap) + h->cost_stride + 3 * h->nu + 2) * sizeof(T) +
                       sindy_stage_bytes<T>(h);
    if (fp_allowed && sm.stage && sm.n_tab > 0 && h->nx <= 8 && lbf <= kLdsLimit) {
      const dim3 grid((unsigned)(p->n_tiles * G));
      auto launch = [&](auto kernel) -> int {
        HIP_OK(allow_lds(kernel, lbf));
        hipLaunchKernelGGL(kernel, grid, dim3(64), lbf, h->stream, a, sm, hcap);
        return 0;
      };
      const bool ind = h->n_ind > 0;
      int rc;
      if (G == 64) rc = ind ? launch(mppi_rollout_sindy_fp_kernel<T, 64, true>) : launch(mppi_rollout_sindy_fp_kernel<T, 64>);
      else if (G == 32) rc = ind ? launch(mppi_rollout_sindy_fp_kernel<T, 32, true>) : launch(mppi_rollout_sindy_fp_kernel<T, 32>);
      else rc = ind ? launch(mppi_rollout_sindy_fp_kernel<T, 16, true>) : launch(mppi_rollout_sindy_fp_kernel<T, 16>);
      if (rc) return rc;
    } else {
      const size_t lb = ((size_t)(2 * h->nx + h->nu + h->s_ntab) * 64 + h->cost_stride + 3 * h->nu + 2) * sizeof(T) +
                        sindy_stage_bytes<T>(h);
      HIP_OK(allow_lds(mppi_rollout_sindy_kernel<T>, lb));
      hipLaunchKernelGGL(mppi_rollout_sindy_kernel<T>, dim3(p->n_tiles), dim3(64), lb, h->stream, a, sm);
    }
  } else
#endif
  if (p->static_shape >= 0) {
    // shape-specialised instantiation (dimensions, strides and LDS offsets are immediates)
    // (relu -- the reference's default, mlp.py:126-128 -- gets its own instantiation; the other
    // activations share one with a run-time switch.  (tile rows, LDS map): (16, 0) (32, 0) (32, 1) (32, 2))
    const int geo = p->mt == 1 ? 0 : 1 + p->static_lv;
    switch ((p->static_shape * 4 + geo) * 2 + (h->act == 0 ? 1 : 0)) {
#define AMPC_SHAPE_LAUNCH_ONE(ID, NX, NU, NO, NH, HPAD, GEO, MTV, LVV, RELU)                            \
      case (ID * 4 + GEO) * 2 + RELU: {                                                               \
        constexpr int W = (HPAD % 128 == 0) ? 8 : 4, NT = HPAD / (16 * W);                             \
        auto k = mppi_rollout_kernel<T, NT, MTV, W, StaticShape<NX, NU, NO, NH, HPAD, RELU ? 0 : -1, LVV>>; \
        HIP_OK(allow_lds(k, p->lds_bytes));                                                           \
        hipLaunchKernelGGL(k, dim3(p->n_tiles), dim3(64 * W), p->lds_bytes, h->stream, a);             \
      } break;
#define AMPC_SHAPE_LAUNCH_GEO(ID, NX, NU, NO, NH, HPAD, GEO, MTV, LVV)   \
      AMPC_SHAPE_LAUNCH_ONE(ID, NX, NU, NO, NH, HPAD, GEO, MTV, LVV, 0)  \
      AMPC_SHAPE_LAUNCH_ONE(ID, NX, NU, NO, NH, HPAD, GEO, MTV, LVV, 1)
#define AMPC_SHAPE_LAUNCH(ID, NX, NU, NO, NH, HPAD)             \
      AMPC_SHAPE_LAUNCH_GEO(ID, NX, NU, NO, NH, HPAD, 0, 1, 0)  \
      AMPC_SHAPE_LAUNCH_GEO(ID, NX, NU, NO, NH, HPAD, 1, 2, 0)  \
      AMPC_SHAPE_LAUNCH_GEO(ID, NX, NU, NO, NH, HPAD, 2, 2, 1)  \
      AMPC_SHAPE_LAUNCH_GEO(ID, NX, NU, NO, NH, HPAD, 3, 2, 2)
      AMPC_STATIC_SHAPES(AMPC_SHAPE_LAUNCH)
#undef AMPC_SHAPE_LAUNCH
#undef AMPC_SHAPE_LAUNCH_GEO
#undef AMPC_SHAPE_LAUNCH_ONE
      default: return fail("internal: unknown static shape");
    }
  } else {
#ifdef AMPC_JIT_PLUGIN
    return fail("shape plugin entered without its static shape");
#else
    AMPC_DISPATCH(h, p->mt, {
      auto k = mppi_rollout_kernel<T, NT, MT, W, DynShape, WD>;
      HIP_OK(allow_lds(k, p->lds_bytes));
      hipLaunchKernelGGL(k, dim3(p->n_tiles), dim3(64 * W), p->lds_bytes, h->stream, a);
    });
#endif
  }
  if (e) HIP_OK(hipEventRecord(e[1], h->stream));
  // Noise one solve ahead (host_common.hpp): extra blocks of the update / combine launch form the next stream
  // index's noise; gx = blocks along x of that launch.
  auto noise_ahead = [&](unsigned& gx) {
    NoiseAhead<T> ahead{nullptr, 0, 0};
    gx = (unsigned)p->max_h;
    p->ahead_valid = false;
    if (p->ahead_on && p->eps_from_generator && !p->eps_inline && p->B <= 65535) {
      long long max_pairs = 0;
      for (int b = 0; b < p->B; ++b) {
        const long long pairs = ((long long)p->N[b] * p->H[b] * h->nu + 1) / 2;
        max_pairs = pairs > max_pairs ? pairs : max_pairs;
      }
      if (p->eps_next.reserve((size_t)p->sum_nhnu * sizeof(T)) == hipSuccess) {
        ahead.eps = (T*)p->eps_next.p;
        ahead.seed = p->eps_seed;
        ahead.stream = p->eps_stream + 1;
        gx += (unsigned)((max_pairs + kWG - 1) / kWG);
        p->ahead_valid = true;
        p->ahead_seed = ahead.seed;
        p->ahead_stream = ahead.stream;
      }
    }
    return ahead;
  };
  if (p->fused_combine && p->quad) {
    // (the rollout's last workgroup per problem finished the update: mppi_kernels.hpp)
    p->ahead_valid = false;
  } else if (p->lds_eps >= 0) {
    unsigned gx = 0;
    const NoiseAhead<T> ahead = noise_ahead(gx);
    hipLaunchKernelGGL(mppi_combine_kernel<T>, dim3(gx, p->B), dim3(kWG), 0, h->stream, a, p->tile_m, ahead);
  } else {
    int maxn = 0;
    for (int n : p->N) maxn = n > maxn ? n : maxn;
    const size_t ub = ((size_t)(maxn <= kUpdateMaxN ? maxn : 0) + kWaves + kWG) * sizeof(T);
    auto uk = mppi_update_kernel<T>;
    HIP_OK(allow_lds(uk, ub));
    unsigned gx = 0;
    const NoiseAhead<T> ahead = noise_ahead(gx);
    hipLaunchKernelGGL(uk, dim3(gx, p->B), dim3(kWG), ub, h->stream, a, ahead);
  }
  if (e) HIP_OK(hipEventRecord(e[2], h->stream));
  HIP_OK(hipGetLastError());
  p->cur ^= 1;
  p->costs_final = false;
  p->solved = true;
  return 0;
}

AMPC_PROBE_HOST_MPPI      // (timing-experiment builds only: read-back of the marks, probe.hpp)

template int mppi_solve_impl<AMPC_T>(ampc_mppi_plan*);
