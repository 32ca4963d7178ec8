// probe.hpp -- the timing experiments' switchboard (tools/variants.sh, tools/ab_variants.py,
// tools/wavetime.py, tools/phasetime*.py).  This is the ONE place where the build flags of an
// experiment (-DAMPC_X_<NAME>) are read: they become the compile-time constants of `Probe` and the
// AMPC_MARK / AMPC_PROBE_* hooks below, which is all the kernel headers see.  In the product build
// none of the flags is defined: every constant is false (the `if constexpr` branches they guard
// are not instantiated) and every hook expands to nothing.
//
//   no_mfma      MFMAs replaced by one scalar FMA (everything-but-MFMA time)
//   no_hid / no_l0 / no_out   one layer's MFMAs skipped
//   no_load      weight fragments synthesised instead of loaded
//   no_sched     without the hand-placed sched_group_barrier pattern
//   vmem_first   all weight loads of a sub-group up front
//   valu_pad(64) 64 dummy int32 / f64 VALU instructions per layer call (VALU cost calibration)
//   no_res0      layer-0 fragments not register-resident
//   f32_ahead / f32_fine   f32 hidden layer: A-fragment reads further ahead / request pattern (0 = f64's, 1 = product)
//   sg8          whole-group weight streaming in f64
//   side_late    the caller's side work after the hidden layers instead of inside them
//   no_cost      dense stage cost skipped
//   lsw_global   twelve-row line search: per-step traffic as global_load / global_store, not raw buffer
//   wave_time    per-wave s_memtime marks of one rollout step, kept in registers
//   phase_time   per-phase s_memtime marks of workgroup 7 in a device array
#pragma once
#include <hip/hip_runtime.h>

// ric_occ: the backward sweep compiled for FOUR waves per SIMD (<= 128 VGPRs: two sweep workgroups per CU) instead of
//          two -- an attribute, so it cannot be an `if constexpr` constant like the others
#ifdef AMPC_X_RIC_OCC
#define AMPC_PROBE_RIC_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define AMPC_PROBE_RIC_OCC
#endif

namespace ampc {

#define AMPC_PROBE_FLAG(name, macro) static constexpr bool name = macro
struct Probe {
#ifdef AMPC_X_NOMFMA
  AMPC_PROBE_FLAG(no_mfma, true);
#else
  AMPC_PROBE_FLAG(no_mfma, false);
#endif
#ifdef AMPC_X_NOHID
  AMPC_PROBE_FLAG(no_hid, true);
#else
  AMPC_PROBE_FLAG(no_hid, false);
#endif
#ifdef AMPC_X_NOL0
  AMPC_PROBE_FLAG(no_l0, true);
#else
  AMPC_PROBE_FLAG(no_l0, false);
#endif
#ifdef AMPC_X_NOOUT
  AMPC_PROBE_FLAG(no_out, true);
#else
  AMPC_PROBE_FLAG(no_out, false);
#endif
#ifdef AMPC_X_NOLOAD
  AMPC_PROBE_FLAG(no_load, true);
#else
  AMPC_PROBE_FLAG(no_load, false);
#endif
#ifdef AMPC_X_NOSCHED
  AMPC_PROBE_FLAG(no_sched, true);
#else
  AMPC_PROBE_FLAG(no_sched, false);
#endif
#ifdef AMPC_X_VMEMFIRST
  AMPC_PROBE_FLAG(vmem_first, true);
#else
  AMPC_PROBE_FLAG(vmem_first, false);
#endif
#ifdef AMPC_X_VALUPAD
  AMPC_PROBE_FLAG(valu_pad, true);
#else
  AMPC_PROBE_FLAG(valu_pad, false);
#endif
#ifdef AMPC_X_VALUPAD64
  AMPC_PROBE_FLAG(valu_pad64, true);
#else
  AMPC_PROBE_FLAG(valu_pad64, false);
#endif
#ifdef AMPC_X_NORES0
  AMPC_PROBE_FLAG(no_res0, true);
#else
  AMPC_PROBE_FLAG(no_res0, false);
#endif
#ifdef AMPC_X_F32AHEAD     // f32 hidden layer: k-steps of A-fragment LDS reads issued ahead of the MFMAs (product: 2)
  static constexpr int f32_ahead = AMPC_X_F32AHEAD;
#else
  static constexpr int f32_ahead = 0;
#endif
#ifdef AMPC_X_F32FINE      // f32 hidden layer: request pattern 0 (per pair of k-steps, the f64 pattern), 1 (product), 2, 3
  static constexpr int f32_fine = AMPC_X_F32FINE;
#else
  static constexpr int f32_fine = 1;
#endif
#ifdef AMPC_X_NOJACPIPE     // Jacobian chain without the hidden layer's hand-placed issue pattern (round-4 code)
  AMPC_PROBE_FLAG(jac_pipe, false);
#else
  AMPC_PROBE_FLAG(jac_pipe, true);
#endif
#ifdef AMPC_X_SG8
  AMPC_PROBE_FLAG(sg8, true);
#else
  AMPC_PROBE_FLAG(sg8, false);
#endif
#ifdef AMPC_X_SIDELATE
  AMPC_PROBE_FLAG(side_late, true);
#else
  AMPC_PROBE_FLAG(side_late, false);
#endif
#ifdef AMPC_X_NOCOST
  AMPC_PROBE_FLAG(no_cost, true);
#else
  AMPC_PROBE_FLAG(no_cost, false);
#endif
#ifdef AMPC_X_LSWGLOBAL
  AMPC_PROBE_FLAG(lsw_global, true);
#else
  AMPC_PROBE_FLAG(lsw_global, false);
#endif
#ifdef AMPC_X_WAVETIME
  AMPC_PROBE_FLAG(wave_time, true);
#else
  AMPC_PROBE_FLAG(wave_time, false);
#endif
#if defined(AMPC_X_PHASETIME) && !defined(AMPC_X_WAVETIME)
  AMPC_PROBE_FLAG(phase_time, true);
#else
  AMPC_PROBE_FLAG(phase_time, false);
#endif
};
#undef AMPC_PROBE_FLAG

// Registers a tile keeps for the wave_time experiment (empty otherwise).
template <bool ON> struct ProbeWave {};
template <> struct ProbeWave<true> {
  long long m[16];
  bool on = false;
};

#if defined(AMPC_X_WAVETIME)
// Low-perturbation timeline (tools/wavetime.py): every wave of workgroup 7 keeps its own s_memtime
// marks of ONE time step in registers (TileNet::probe) and dumps them at kernel end.
__device__ long long g_wave_marks[8 * 16];
#define AMPC_PROBE_LOCALS(pw) auto& _xm = (pw).m; const bool _xon = (pw).on; (void)_xm; (void)_xon
#define AMPC_MARK(idx) do { if (_xon) _xm[idx] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define AMPC_MARK_ALWAYS(idx) do { } while (0)
#define AMPC_PROBE_KERNEL_BEGIN(pw) auto& _xm = (pw).m; _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) _xm[i_] = 0
#define AMPC_PROBE_STEP(pw, cond) (pw).on = (blockIdx.x == 7 && (cond)); const bool _xon = (pw).on
#define AMPC_PROBE_KERNEL_END()                                                                  \
  do {                                                                                           \
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0)                                              \
      for (int i_ = 0; i_ < 16; ++i_) g_wave_marks[(threadIdx.x >> 6) * 16 + i_] = _xm[i_];      \
  } while (0)
#elif defined(AMPC_X_PHASETIME)
__device__ long long g_phase_marks[128];   // (64..: per-wave register marks of the twelve-row line search)
#define AMPC_PROBE_LOCALS(pw) do { } while (0)
#define AMPC_MARK(idx)                                                              \
  do {                                                                              \
    if (blockIdx.x == 7 && threadIdx.x == 0 && g_phase_marks[63] == 1)              \
      g_phase_marks[idx] = (long long)__builtin_amdgcn_s_memtime();                 \
  } while (0)
// unconditional variant for coarse, once-per-kernel marks
#define AMPC_MARK_ALWAYS(idx)                                                       \
  do {                                                                              \
    if (blockIdx.x == 7 && threadIdx.x == 0)                                        \
      g_phase_marks[idx] = (long long)__builtin_amdgcn_s_memtime();                 \
  } while (0)
#define AMPC_PROBE_KERNEL_BEGIN(pw) do { } while (0)
#define AMPC_PROBE_STEP(pw, cond) \
  do { if (blockIdx.x == 7 && threadIdx.x == 0) g_phase_marks[63] = (cond) ? 1 : 0; } while (0)
#define AMPC_PROBE_KERNEL_END() do { } while (0)
#else
#define AMPC_PROBE_LOCALS(pw) do { } while (0)
#define AMPC_MARK(idx) do { } while (0)
#define AMPC_MARK_ALWAYS(idx) do { } while (0)
#define AMPC_PROBE_KERNEL_BEGIN(pw) do { } while (0)
#define AMPC_PROBE_STEP(pw, cond) do { } while (0)
#define AMPC_PROBE_KERNEL_END() do { } while (0)
#endif

// phase marks of the iLQR kernels (tools/phasetime_ilqr.py): only in the phase_time build
#if defined(AMPC_X_PHASETIME) && !defined(AMPC_X_WAVETIME)
// low-perturbation variant (ilqr_lsw.hpp): every wave keeps its marks of one time step in registers
#define AMPC_LSW_MARK(pm, on, i) do { if (on) (pm)[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define AMPC_LSW_DUMP(pm, w, n)                                                                   \
  do {                                                                                            \
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0)                                               \
      for (int i_ = 0; i_ < (n); ++i_) g_phase_marks[64 + (w) * 16 + i_] = (pm)[i_];              \
  } while (0)
#define AMPC_IMARK(idx) AMPC_MARK(idx)
#define AMPC_IMARK_ALWAYS(idx) AMPC_MARK_ALWAYS(idx)
#define AMPC_IPROBE_STEP(cond) AMPC_PROBE_STEP(0, cond)
#else
#define AMPC_LSW_MARK(pm, on, i) do { } while (0)
#define AMPC_LSW_DUMP(pm, w, n) do { } while (0)
#define AMPC_IMARK(idx) do { } while (0)
#define AMPC_IMARK_ALWAYS(idx) do { } while (0)
#define AMPC_IPROBE_STEP(cond) do { } while (0)
#endif

#if defined(AMPC_X_PHASETIME) && !defined(AMPC_X_WAVETIME)
#define AMPC_NMARKS 128
#else
#define AMPC_NMARKS 64
#endif
// host-side read-back entry points of the experiment builds (expanded in launch_*.cpp)
#if defined(AMPC_X_WAVETIME) && defined(AMPC_T_IS_F64)
#define AMPC_PROBE_HOST_MPPI                                                                        \
  extern "C" int ampc_x_wave_marks(long long* out) {                                                \
    HIP_OK(hipDeviceSynchronize());                                                                 \
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ampc::g_wave_marks), 8 * 16 * sizeof(long long)));   \
    return 0;                                                                                       \
  }
#elif defined(AMPC_X_WAVETIME)      /* the f32 translation unit: its own copy of the marks */
#define AMPC_PROBE_HOST_MPPI                                                                        \
  extern "C" int ampc_x_wave_marks_f32(long long* out) {                                            \
    HIP_OK(hipDeviceSynchronize());                                                                 \
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ampc::g_wave_marks), 8 * 16 * sizeof(long long)));   \
    return 0;                                                                                       \
  }
#elif defined(AMPC_X_PHASETIME) && defined(AMPC_T_IS_F64)
#define AMPC_PROBE_HOST_MPPI                                                                        \
  extern "C" int ampc_x_phase_marks(long long* out) {                                               \
    HIP_OK(hipDeviceSynchronize());                                                                 \
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ampc::g_phase_marks), AMPC_NMARKS * sizeof(long long)));      \
    return 0;                                                                                       \
  }
#else
#define AMPC_PROBE_HOST_MPPI
#endif
#if defined(AMPC_X_PHASETIME) && !defined(AMPC_X_WAVETIME) && defined(AMPC_T_IS_F64)
#define AMPC_PROBE_HOST_ILQR                                                                        \
  extern "C" int ampc_x_phase_marks_ilqr(long long* out) {                                          \
    HIP_OK(hipDeviceSynchronize());                                                                 \
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ampc::g_phase_marks), AMPC_NMARKS * sizeof(long long)));      \
    return 0;                                                                                       \
  }
#else
#define AMPC_PROBE_HOST_ILQR
#endif

}  // namespace ampc
