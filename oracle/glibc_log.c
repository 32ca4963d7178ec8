/*
 * glibc_log.c -- CPU restatement of glibc's double-precision log() (oracle, TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
 * oracle/; the product never does).
 *
 * Why it is on the path.  The reference draws MPPI's noise with np.random.normal from numpy's
 * global legacy RandomState (autompc/control/mppi.py:16-24, :126).  numpy's legacy_gauss
 * (numpy/random/src/legacy/legacy-distributions.c) evaluates f = sqrt(-2.0 * log(r2) / r2) with
 * the C library's log() -- a third-party dependency that is not under /root/reference: glibc
 * (this image: Ubuntu GLIBC 2.35-0ubuntu3.11; the algorithm is unchanged since glibc 2.28).  To
 * reproduce numpy's normals bit for bit on the device, log() has to be reproduced bit for bit.
 *
 * What is restated: glibc 2.35 sysdeps/ieee754/dbl-64/e_log.c (__log; from ARM's optimized
 * routines): N = 128 table-driven reduction log(x) = log1p(z/c - 1) + log(c) + k ln2 with a
 * degree-5 polynomial, and a degree-11 polynomial with a split leading term near 1.  On x86-64
 * the library carries two builds of it behind an IFUNC (sysdeps/x86_64/fpu/multiarch/e_log.c):
 *   variant 1  __log_fma   (CPUs with FMA + AVX2; compiled -mfma -mavx2, so __FP_FAST_FMA is
 *              defined: r = fma(z, invc, -1) and gcc contracts the polynomial evaluation).  The
 *              fused operations below are the ones in the library's machine code
 *              (objdump -d libm.so.6, the function the IFUNC resolver returns first).
 *   variant 2  __log_sse2 / __log_avx  (no FMA: r = (z - chi - clo) * invc from the second
 *              table, every operation rounded separately in source order).
 * The tables (__log_data: ln2hi, ln2lo, poly[5], poly1[11], tab[128]{invc, logc},
 * tab2[128]{chi, clo}) are data of the library, not restated: glibc_log_locate() finds them in the
 * loaded libm by the bit patterns of ln2hi / ln2lo and the caller validates the result against
 * log() itself (glibc_log_probe).
 *
 * Pinned by tests/test_glibc_log.py: variant = probe(), restatement == the host's log() on 10^7
 * arguments of the kind legacy_gauss produces (and on the near-1 branch).
 */
#define _GNU_SOURCE
#include <link.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#define LOG_N 128
#define LOG_TABLE_DOUBLES (2 + 5 + 11 + 2 * LOG_N + 2 * LOG_N)

static inline uint64_t asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* t: LOG_TABLE_DOUBLES doubles laid out as __log_data.  variant 1 = FMA build, 2 = plain build.
 * Compiled with -ffp-contract=off: every fused operation is written out. */
double glibc_log_restated(double x, const double* t, int variant) {
  const double ln2hi = t[0], ln2lo = t[1];
  const double* A = t + 2;
  const double* B = t + 7;
  const double* T = t + 18;
  const double* T2 = t + 18 + 2 * LOG_N;
  uint64_t ix = asuint64(x);
  uint32_t top = (uint32_t)(ix >> 48);
  const uint64_t LO = asuint64(1.0 - 0x1p-4), HI = asuint64(1.0 + 0x1.09p-4);
  if (ix - LO < HI - LO) {
    if (ix == asuint64(1.0)) return 0;
    double r = x - 1.0, r2 = r * r, r3 = r * r2, y, hi, lo;
    if (variant == 1) {
      double a = __builtin_fma(r, B[2], B[1]);
      double b = __builtin_fma(r, B[5], B[4]);
      double c = __builtin_fma(r, B[8], B[7]);
      a = __builtin_fma(r2, B[3], a);
      b = __builtin_fma(r2, B[6], b);
      c = __builtin_fma(r2, B[9], c);
      c = __builtin_fma(r3, B[10], c);
      double d = __builtin_fma(c, r3, b);
      double e = __builtin_fma(d, r3, a);
      double tt = __builtin_fma(r, 0x1p27, r);
      double rhi = __builtin_fma(-0x1p27, r, tt);
      double rlo = r - rhi;
      double rhi2 = rhi * rhi;
      hi = __builtin_fma(rhi2, B[0], r);
      lo = __builtin_fma(rhi2, B[0], r - hi);
      lo = __builtin_fma(B[0] * rlo, r + rhi, lo);
      y = __builtin_fma(e, r3, lo);
      return y + hi;
    }
    y = r3 * (B[1] + r * B[2] + r2 * B[3] +
              r3 * (B[4] + r * B[5] + r2 * B[6] + r3 * (B[7] + r * B[8] + r2 * B[9] + r3 * B[10])));
    double w = r * 0x1p27;
    double rhi = r + w - w;
    double rlo = r - rhi;
    w = rhi * rhi * B[0];
    hi = r + w;
    lo = r - hi + w;
    lo += B[0] * rlo * (rhi + r);
    y += lo;
    y += hi;
    return y;
  }
  if (top - 0x0010 >= 0x7ff0 - 0x0010) {
    if (ix * 2 == 0) return -INFINITY;            /* log(+-0) */
    if (ix == asuint64(INFINITY)) return x;
    if ((top & 0x8000) || (top & 0x7ff0) == 0x7ff0) return (x - x) / (x - x);   /* x < 0, NaN */
    ix = asuint64(x * 0x1p52);                    /* subnormal: normalise */
    ix -= 52ULL << 52;
  }
  uint64_t tmp = ix - 0x3fe6000000000000ULL;
  int i = (int)((tmp >> (52 - 7)) % LOG_N);
  int k = (int)((int64_t)tmp >> 52);
  uint64_t iz = ix - (tmp & (0xfffULL << 52));
  double invc = T[2 * i], logc = T[2 * i + 1];
  double z = asdouble(iz);
  double kd = (double)k;
  if (variant == 1) {
    double r = __builtin_fma(z, invc, -1.0);
    double w = __builtin_fma(kd, ln2hi, logc);
    double p12 = __builtin_fma(r, A[2], A[1]);
    double hi = r + w;
    double r2 = r * r;
    double lo = (w - hi) + r;
    lo = __builtin_fma(kd, ln2lo, lo);
    double r3 = r * r2;
    double p34 = __builtin_fma(r, A[4], A[3]);
    double q = __builtin_fma(r2, A[0], lo);
    double p = __builtin_fma(p34, r2, p12);
    double y = __builtin_fma(r3, p, q);
    return y + hi;
  }
  double r = (z - T2[2 * i] - T2[2 * i + 1]) * invc;
  double w = kd * ln2hi + logc;
  double hi = w + r;
  double lo = w - hi + r + kd * ln2lo;
  double r2 = r * r;
  return lo + r2 * A[0] + r * r2 * (A[1] + r * A[2] + r2 * (A[3] + r * A[4])) + hi;
}

/* --- locating __log_data in the loaded C math library ------------------------------------------ */
struct locate_ctx { const double* found; };

static int plausible(const double* t) {
  if (t[7] != -0.5) return 0;                                /* poly1[0] = B[0] */
  for (int i = 0; i < LOG_N; ++i) {
    double invc = t[18 + 2 * i], logc = t[19 + 2 * i];
    if (!(invc > 0.7 && invc < 1.5)) return 0;
    if (fabs(logc + log(invc)) > 1e-9) return 0;
  }
  return 1;
}

static int locate_cb(struct dl_phdr_info* info, size_t size, void* data) {
  (void)size;
  struct locate_ctx* ctx = (struct locate_ctx*)data;
  if (!info->dlpi_name || !strstr(info->dlpi_name, "libm")) return 0;
  const double pat[2] = {0x1.62e42fefa3800p-1, 0x1.ef35793c76730p-45};   /* ln2hi, ln2lo */
  for (int s = 0; s < info->dlpi_phnum; ++s) {
    const ElfW(Phdr)* ph = &info->dlpi_phdr[s];
    if (ph->p_type != PT_LOAD || !(ph->p_flags & PF_R) || (ph->p_flags & PF_W)) continue;
    const char* base = (const char*)(info->dlpi_addr + ph->p_vaddr);
    size_t len = ph->p_memsz, need = LOG_TABLE_DOUBLES * sizeof(double);
    for (size_t off = 0; off + need <= len; off += 8) {
      if (memcmp(base + off, pat, sizeof(pat)) != 0) continue;
      if (plausible((const double*)(base + off))) { ctx->found = (const double*)(base + off); return 1; }
    }
  }
  return 0;
}

/* Copies the LOG_TABLE_DOUBLES doubles of __log_data to out; returns 0, or -1 if not found. */
int glibc_log_locate(double* out) {
  struct locate_ctx ctx = {0};
  dl_iterate_phdr(locate_cb, &ctx);
  if (!ctx.found) return -1;
  memcpy(out, ctx.found, LOG_TABLE_DOUBLES * sizeof(double));
  return 0;
}

/* Which build of log() this process runs: 1 (FMA), 2 (plain), 0 (neither reproduces it).  n
 * arguments from a 64-bit LCG: r2-like values in (0, 1) and values around 1. */
int glibc_log_probe(const double* t, long n) {
  int ok1 = 1, ok2 = 1;
  uint64_t s = 0x9e3779b97f4a7c15ULL;
  for (long j = 0; j < n && (ok1 || ok2); ++j) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    double u = (double)(s >> 11) * 0x1p-53;
    double x = (j & 3) == 3 ? 0.9375 + 0.13 * u : (u > 0 ? u : 0.5);
    if ((j & 63) == 5) x = ldexp(x, -(int)(s & 127));
    double ref = log(x);
    if (ok1 && asuint64(glibc_log_restated(x, t, 1)) != asuint64(ref)) ok1 = 0;
    if (ok2 && asuint64(glibc_log_restated(x, t, 2)) != asuint64(ref)) ok2 = 0;
  }
  return ok1 ? 1 : (ok2 ? 2 : 0);
}

/* Vector form for the tests: out[i] = restated log(x[i]); returns the number of i where it differs
 * (bitwise) from the host's log(). */
long glibc_log_compare(const double* x, long n, const double* t, int variant, double* out) {
  long bad = 0;
  for (long i = 0; i < n; ++i) {
    double y = glibc_log_restated(x[i], t, variant);
    if (out) out[i] = y;
    double ref = log(x[i]);
    if (asuint64(y) != asuint64(ref) && !(y != y && ref != ref)) ++bad;
  }
  return bad;
}
