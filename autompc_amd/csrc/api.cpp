// api.cpp -- the C ABI of libautompc_hip.so (include/autompc_hip.h), core part: errors, version, handles, the
// shape plugins' status calls.  The rest of the ABI by family:
//   api_model.cpp   models and what a handle holds: ampc_set_mlp[_dev], ampc_set_linear, ampc_set_sindy, costs,
//                   bounds, ampc_mlp_pred_* / ampc_sindy_pred_*
//   api_mppi.cpp    MPPI plans, noise (Philox, numpy's legacy stream), ampc_mppi_run*, the MPPI closed loop,
//                   trajectory scores
//   api_ilqr.cpp    iLQR plans, batch / queue solves, the iLQR closed loop
// The heavy kernel families are launched through the templates declared in host_common.hpp (launch_*.cpp).
// Built with hipcc for gfx950 only.
#include "host_common.hpp"
#include <mutex>
#include "jit_host.hpp"                // shape plugins compiled at run time

thread_local std::string g_err;

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

extern "C" const char* ampc_last_error(void) { return g_err.c_str(); }
extern "C" int ampc_version(void) { return 107; }   // 1.07: round 6 (ampc_set_mlp_dev, ampc_ilqr_plan_set_constants); 1.06: round 5 (ampc_ilqr_*_var); 1.04: round 3 (ampc_set_sindy monomial pair list; ampc_mppi_run_legacy)
extern "C" int ampc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Start (or find) the shape plugin of the staged model as soon as model and observation dimension
// are both known; never blocks.
void ampc_internal_jit_kick(ampc_handle* h) {
  if (!h->has_mlp || h->obs_dim < 1) return;
  if (h->precision == AMPC_F64) {
    if (static_shape_of<double>(h, h->md) < 0) (void)jit::get<double>(h);
  } else {
    if (static_shape_of<float>(h, h->mf) < 0) (void)jit::get<float>(h);
  }
}

extern "C" int ampc_jit_status(ampc_handle* h, char* msg, int msg_len) {
  if (!h) return 0;
  std::string m;
  const int st = h->precision == AMPC_F64 ? jit::status<double>(h, &m) : jit::status<float>(h, &m);
  if (msg && msg_len > 0) { std::strncpy(msg, m.c_str(), (size_t)msg_len - 1); msg[msg_len - 1] = 0; }
  return st;
}

extern "C" int ampc_plan_kernel_kind(const ampc_mppi_plan* mppi, const ampc_ilqr_plan* ilqr) {
  if (mppi && mppi->quad && mppi->static_shape < 0) return 3;     // (specialised four-row kernels report 1 / 2)
  const int sid = mppi ? mppi->static_shape : (ilqr ? ilqr->static_shape : -1);
  const JitPlugin* j = mppi ? mppi->jit : (ilqr ? ilqr->jit : nullptr);
  return sid < 0 ? 0 : (j ? 2 : 1);
}

extern "C" int ampc_handle_set_jit(ampc_handle* h, int build) {
  REQUIRE(h, "ampc_handle_set_jit: NULL handle");
  h->jit_build = build != 0;
  return 0;
}

extern "C" int ampc_jit_wait(ampc_handle* h) {
  REQUIRE(h, "ampc_jit_wait: NULL handle");
  ampc_internal_jit_kick(h);
  if (!jit::eligible(h)) return 0;
  const bool reg = h->precision == AMPC_F64 ? static_shape_of<double>(h, h->md) >= 0
                                            : static_shape_of<float>(h, h->mf) >= 0;
  if (reg) return 0;
  const JitPlugin* j = h->precision == AMPC_F64 ? jit::get<double>(h, true) : jit::get<float>(h, true);
  if (j) return 0;
  std::string m;
  (void)(h->precision == AMPC_F64 ? jit::status<double>(h, &m) : jit::status<float>(h, &m));
  return fail("ampc_jit_wait: " + m);
}

extern "C" int ampc_create(int device, int precision, void* stream, ampc_handle** out) {
  REQUIRE(out != nullptr, "ampc_create: out is NULL");
  REQUIRE(precision == AMPC_F64 || precision == AMPC_F32, "ampc_create: bad precision");
  int n = 0;
  HIP_OK(hipGetDeviceCount(&n));
  REQUIRE(device >= 0 && device < n, "ampc_create: no such HIP device");
  HIP_OK(hipSetDevice(device));
  ampc_handle* h = new ampc_handle();
  h->device = device;
  h->precision = precision;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cus = cus;
  }
  if (stream) {
    h->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete h;
      return fail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    h->own_stream = true;
  }
  *out = h;
  return 0;
}

static void handle_free(ampc_handle* h);

extern "C" int ampc_destroy(ampc_handle* h) {
  if (!h) return 0;
  if (h->refs > 0) {
    h->dead = true;
    return 0;
  }
  handle_free(h);
  return 0;
}

void handle_release(ampc_handle* h) {
  if (--h->refs == 0 && h->dead) handle_free(h);
}

static void handle_free(ampc_handle* h) {
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  DevBuf* bufs[] = {&h->model_buf, &h->lin_buf, &h->sindy_int, &h->sindy_flt, &h->cost_buf, &h->bounds_buf, &h->ubounds_buf, &h->ind_buf, &h->s_states,
                    &h->s_ctrls,   &h->s_out,      &h->s_dz,     &h->s_jx,       &h->s_ju};
  for (DevBuf* b : bufs) b->release();
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int ampc_synchronize(ampc_handle* h) {
  REQUIRE(h, "ampc_synchronize: NULL handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int ampc_precision(const ampc_handle* h) { return h ? h->precision : -1; }
