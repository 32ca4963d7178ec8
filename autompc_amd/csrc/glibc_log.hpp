// glibc_log.hpp -- the host C library's double-precision log(), reproduced bit for bit.
//
// numpy's legacy normal generator -- what the reference's MPPI draws its noise from
// (autompc/control/mppi.py:16-24, :126 -> np.random.normal -> legacy_gauss) -- computes
// f = sqrt(-2.0 * log(r2) / r2) with the C library's log().  The device's own log() rounds
// differently from glibc's for 2.6 % of the arguments (last bit), so the normals generated on the
// device (legacy_rng_kernels.hpp) were not numpy's.  This header restates glibc's algorithm
// (glibc >= 2.28, sysdeps/ieee754/dbl-64/e_log.c: 128-entry table reduction + degree-5 polynomial,
// degree-11 polynomial with a split leading term near 1) operation by operation, in the two builds
// x86-64 glibc selects between at load time (sysdeps/x86_64/fpu/multiarch/e_log.c):
//   VARIANT 1  __log_fma        (FMA + AVX2 CPUs; r = fma(z, invc, -1), contracted polynomials --
//                               the fused operations are the ones in the library's machine code)
//   VARIANT 2  __log_sse2/_avx  (r = (z - chi - clo) * invc, every operation rounded on its own)
// The tables are the library's data (__log_data): the host side finds them in the loaded libm
// and proves the restatement against log() itself before the device path may use it
// (host_log_mode() in api.cpp).  The same function body is compiled for the host (the proof) and
// for the device (the product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace ampc {

constexpr int kLogN = 128;
constexpr int kLogTableDoubles = 2 + 5 + 11 + 2 * kLogN + 2 * kLogN;   // layout of __log_data

// A value the optimiser cannot look through: keeps a product from being fused into the addition
// that consumes it (hipcc's device back end contracts fmul + fadd even under fp contract(off)).
__host__ __device__ __forceinline__ double log_opaque(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  asm volatile("" : "+x"(v));
#endif
  return v;
}

__host__ __device__ __forceinline__ uint64_t log_bits(double f) {
  uint64_t u;
  memcpy(&u, &f, 8);
  return u;
}
__host__ __device__ __forceinline__ double log_from_bits(uint64_t u) {
  double f;
  memcpy(&f, &u, 8);
  return f;
}

// t: kLogTableDoubles doubles laid out as __log_data: ln2hi, ln2lo, A[5], B[11], {invc, logc}[128],
// {chi, clo}[128].
template <int VARIANT>
__host__ __device__ inline double glibc_log(double x, const double* __restrict__ t) {
#pragma clang fp contract(off)
  const double ln2hi = t[0], ln2lo = t[1];
  const double* A = t + 2;
  const double* B = t + 7;
  const double* T = t + 18;
  const double* T2 = t + 18 + 2 * kLogN;
  uint64_t ix = log_bits(x);
  const uint32_t top = (uint32_t)(ix >> 48);
  const uint64_t LO = 0x3fee000000000000ull;                 // 1 - 0x1p-4
  const uint64_t HI = 0x3ff1090000000000ull;                 // 1 + 0x1.09p-4
  if (ix - LO < HI - LO) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;
    const double r2 = log_opaque(r * r), r3 = log_opaque(r * r2);
    if (VARIANT == 1) {
      double a = __builtin_fma(r, B[2], B[1]);
      double b = __builtin_fma(r, B[5], B[4]);
      double c = __builtin_fma(r, B[8], B[7]);
      a = __builtin_fma(r2, B[3], a);
      b = __builtin_fma(r2, B[6], b);
      c = __builtin_fma(r2, B[9], c);
      c = __builtin_fma(r3, B[10], c);
      const double d = __builtin_fma(c, r3, b);
      const double e = __builtin_fma(d, r3, a);
      const double tt = __builtin_fma(r, 0x1p27, r);
      const double rhi = __builtin_fma(-0x1p27, r, tt);
      const double rlo = r - rhi;
      const double rhi2 = log_opaque(rhi * rhi);
      const double hi = __builtin_fma(rhi2, B[0], r);
      double lo = __builtin_fma(rhi2, B[0], r - hi);
      lo = __builtin_fma(log_opaque(B[0] * rlo), r + rhi, lo);
      const double y = __builtin_fma(e, r3, lo);
      return y + hi;
    }
    const double p7 = B[7] + log_opaque(r * B[8]) + log_opaque(r2 * B[9]) + log_opaque(r3 * B[10]);
    const double p4 = B[4] + log_opaque(r * B[5]) + log_opaque(r2 * B[6]) + log_opaque(r3 * p7);
    const double p1 = B[1] + log_opaque(r * B[2]) + log_opaque(r2 * B[3]) + log_opaque(r3 * p4);
    double y = log_opaque(r3 * p1);
    double w = log_opaque(r * 0x1p27);
    const double rhi = r + w - w;
    const double rlo = r - rhi;
    w = log_opaque(log_opaque(rhi * rhi) * B[0]);
    const double hi = r + w;
    double lo = r - hi + w;
    lo += log_opaque(log_opaque(B[0] * rlo) * (rhi + r));
    y += lo;
    y += hi;
    return y;
  }
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
    if (ix * 2 == 0) return -__builtin_inf();                 // log(+-0)
    if (ix == 0x7ff0000000000000ull) return x;                // log(inf)
    if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __builtin_nan("");   // x < 0, NaN
    ix = log_bits(x * 0x1p52);                                // subnormal: normalise
    ix -= 52ull << 52;
  }
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> (52 - 7)) % kLogN);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double invc = T[2 * i], logc = T[2 * i + 1];
  const double z = log_from_bits(iz);
  const double kd = (double)k;
  if (VARIANT == 1) {
    const double r = __builtin_fma(z, invc, -1.0);
    const double w = __builtin_fma(kd, ln2hi, logc);
    const double p12 = __builtin_fma(r, A[2], A[1]);
    const double hi = r + w;
    const double r2 = log_opaque(r * r);
    double lo = (w - hi) + r;
    lo = __builtin_fma(kd, ln2lo, lo);
    const double r3 = log_opaque(r * r2);
    const double p34 = __builtin_fma(r, A[4], A[3]);
    const double q = __builtin_fma(r2, A[0], lo);
    const double p = __builtin_fma(p34, r2, p12);
    const double y = __builtin_fma(r3, p, q);
    return y + hi;
  }
  const double r = log_opaque((z - T2[2 * i] - T2[2 * i + 1]) * invc);
  const double w = log_opaque(kd * ln2hi) + logc;
  const double hi = w + r;
  const double lo = w - hi + r + log_opaque(kd * ln2lo);
  const double r2 = log_opaque(r * r);
  const double p34 = A[3] + log_opaque(r * A[4]);
  const double p = A[1] + log_opaque(r * A[2]) + log_opaque(r2 * p34);
  return lo + log_opaque(r2 * A[0]) + log_opaque(log_opaque(r * r2) * p) + hi;
}

}  // namespace ampc
