// jit_plugin.cpp -- entry points of a run-time compiled SHAPE PLUGIN (shapes.hpp, jit_host.hpp).
//
// A plugin is launch_mppi.cpp + launch_mlp.cpp + launch_ilqr.cpp + this file, compiled with
//   -DAMPC_JIT_PLUGIN -DAMPC_T=<double|float> -DAMPC_JIT_NX= _NU= _NO= _NH= _HPAD=  -fvisibility=hidden
// so that the shape registry holds exactly the staged model's shape (id 0) and every
// StaticShape kernel the library has for a registered shape exists for it: the MPPI rollout
// (16- / 32-row tiles, every LDS map, relu / other activations), the forward + Jacobian chain of
// the iLQR refresh, both backward sweeps, the four-row line search and the 16-row line search.
// The launchers are the library's own functions (same sources, same structs); the library calls
// them through the four symbols below and never links against the plugin.
#include "host_common.hpp"

#ifndef AMPC_JIT_PLUGIN
#error "jit_plugin.cpp is only built as part of a shape plugin"
#endif

thread_local std::string g_err;

#define AMPC_EXPORT extern "C" __attribute__((visibility("default")))

AMPC_EXPORT int ampc_jit_mppi_solve(ampc_mppi_plan* p) { return mppi_solve_impl<AMPC_T>(p); }
AMPC_EXPORT int ampc_jit_ilqr_iter(ampc_ilqr_plan* p, int mode) { return ilqr_launch_iter<AMPC_T>(p, mode); }
AMPC_EXPORT int ampc_jit_ilqr_refresh(ampc_ilqr_plan* p) { return ilqr_refresh_jacobians<AMPC_T>(p); }
AMPC_EXPORT const char* ampc_jit_last_error(void) { return g_err.c_str(); }
// what the plugin was compiled for: {nx, nu, obs_dim, n_hidden, hpad, sizeof(T), tail4, sizeof(plan structs)}
AMPC_EXPORT void ampc_jit_info(int* out) {
  using SH = StaticShape<AMPC_JIT_NX, AMPC_JIT_NU, AMPC_JIT_NO, AMPC_JIT_NH, AMPC_JIT_HPAD>;
  out[0] = AMPC_JIT_NX; out[1] = AMPC_JIT_NU; out[2] = AMPC_JIT_NO; out[3] = AMPC_JIT_NH;
  out[4] = AMPC_JIT_HPAD; out[5] = (int)sizeof(AMPC_T); out[6] = SH::template tail4<AMPC_T> ? 1 : 0;
  out[7] = (int)(sizeof(ampc_mppi_plan) + sizeof(ampc_ilqr_plan) + sizeof(ampc_handle));
}
