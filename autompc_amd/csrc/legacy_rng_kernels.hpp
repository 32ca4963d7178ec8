// legacy_rng_kernels.hpp -- numpy's LEGACY normal stream on the device.
//
// The reference draws MPPI's noise with np.random.normal(scale=sqrt(sigma), size=(N, H, nu)) from
// numpy's global legacy RandomState (autompc/control/mppi.py:16-24, :126): MT19937 -> 53-bit
// uniform doubles (rk_double) -> Marsaglia's polar method with a cached second value
// (legacy_gauss) -> loc + scale * g.  In parity mode the host makes that draw (737 k values per
// config-3 solve: ~6 ms of one CPU core, 95 % of a drop-in MPPI.run()).  These kernels reproduce
// the same stream from the same generator state:
//   mt19937_stream_kernel   the raw tempered 32-bit stream, block after block of 624 words.  The
//                           recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) only exposes 227
//                           independent elements per dependent step, so this part is latency
//                           bound (one workgroup); ampc_mppi_legacy_normal overlaps it with the
//                           previous solve by generating the next call's stream speculatively.
//   polar_count / polar_scatter   attempt a consumes words 4a..4a+3: x1, x2, r2 and the
//                           accept / reject decision are exact IEEE operations (no fused
//                           multiply-add), hence bit-identical to numpy's; an exclusive scan of
//                           the accept flags gives every accepted pair its output position.
// What is NOT guaranteed bit-identical is log(): numpy calls the host libm, the device its own
// implementation (both < 1 ulp); a normal can differ from numpy's in its last bit where the two
// disagree.  The MT19937 state handed back (and therefore every later host draw) is exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mppi_kernels.hpp"

namespace ampc {

constexpr int kMtN = 624, kMtM = 397;

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// out[b][i], b = 0 .. nblocks-1: block 0 is the tempered CURRENT key (its words from `pos` on are
// the generator's next outputs), block b the b-th regeneration (randomkit's rk_gen).
// One workgroup of 256 threads; three dependent phases per block, separated by LDS-only barriers
// (__syncthreads() would also wait for the block's global stores -- measured 1300 cycles per block
// against ~450 with the stores left in flight); (elements 0..226 depend on the
// previous block only, 227..453 on the first phase, 454..623 on the second).
__global__ __launch_bounds__(256) void mt19937_stream_kernel(const uint32_t* __restrict__ key_in,
                                                             int nblocks, uint32_t* __restrict__ out) {
  __shared__ uint32_t st[2][kMtN];
  const int tid = threadIdx.x;
  for (int i = tid; i < kMtN; i += 256) {
    const uint32_t v = key_in[i];
    st[0][i] = v;
    out[i] = mt_temper(v);
  }
  __syncthreads();
  int cur = 0;
  for (int b = 1; b < nblocks; ++b) {
    const uint32_t* o = st[cur];
    uint32_t* n = st[cur ^ 1];
    if (tid < kMtN - kMtM) n[tid] = mt_twist(o[tid], o[tid + 1], o[tid + kMtM]);          // 0..226
    lds_barrier();
    if (tid < kMtN - kMtM) {                                                                 // 227..453
      const int i = tid + (kMtN - kMtM);
      n[i] = mt_twist(o[i], o[i + 1], n[i - (kMtN - kMtM)]);
    }
    lds_barrier();
    if (tid < kMtN - 2 * (kMtN - kMtM)) {                                                    // 454..623
      const int i = tid + 2 * (kMtN - kMtM);
      n[i] = (i < kMtN - 1) ? mt_twist(o[i], o[i + 1], n[i - (kMtN - kMtM)])
                            : mt_twist(o[kMtN - 1], n[0], n[kMtM - 1]);
    }
    lds_barrier();
    uint32_t* ob = out + (size_t)b * kMtN;
    for (int i = tid; i < kMtN; i += 256) ob[i] = mt_temper(n[i]);
    cur ^= 1;
  }
}

// hipcc contracts a * b + c into a fused multiply-add (the AMDGPU backend fuses even under
// `#pragma clang fp contract(off)`, and __dmul_rn / __dadd_rn are plain operators to it); the host
// compiles numpy's legacy_gauss without FMA.  A product that must be rounded before it is added is
// therefore passed through an empty asm statement, which hides the multiply from the combiner.
__device__ __forceinline__ double rounded(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

// rk_double from two consecutive words
__device__ __forceinline__ double mt_double(uint32_t w0, uint32_t w1) {
#pragma clang fp contract(off)
  const double a = (double)(w0 >> 5), b = (double)(w1 >> 6);
  return __dadd_rn(__dmul_rn(a, 67108864.0), b) / 9007199254740992.0;
}
// one attempt of the polar method (legacy_gauss): true if accepted
__device__ __forceinline__ bool polar_attempt(const uint32_t* __restrict__ u, double* x1, double* x2,
                                              double* r2) {
#pragma clang fp contract(off)
  *x1 = __dadd_rn(__dmul_rn(2.0, mt_double(u[0], u[1])), -1.0);
  *x2 = __dadd_rn(__dmul_rn(2.0, mt_double(u[2], u[3])), -1.0);
  *r2 = rounded(*x1 * *x1) + rounded(*x2 * *x2);                       // two rounded products, then the sum
  return !(*r2 >= 1.0 || *r2 == 0.0);
}

constexpr int kPolarPerWg = 1024;    // attempts per workgroup (256 threads x 4)

// accepted attempts of every workgroup's slice
__global__ __launch_bounds__(256) void polar_count_kernel(const uint32_t* __restrict__ u, int n_attempts,
                                                          int* __restrict__ wg_count) {
  __shared__ int red[4];
  const int tid = threadIdx.x;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = blockIdx.x * kPolarPerWg + k * 256 + tid;
    double x1, x2, r2;
    if (a < n_attempts && polar_attempt(u + 4 * (size_t)a, &x1, &x2, &r2)) ++c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((tid & 63) == 0) red[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) wg_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the workgroup counts (one workgroup; n_wg <= 4096)
__global__ __launch_bounds__(256) void polar_scan_kernel(int* __restrict__ wg_count, int n_wg,
                                                         int* __restrict__ total) {
  __shared__ int part[256];
  const int tid = threadIdx.x;
  const int per = (n_wg + 255) / 256;
  int s = 0;
  for (int i = tid * per; i < n_wg && i < (tid + 1) * per; ++i) s += wg_count[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
    *total = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int i = tid * per; i < n_wg && i < (tid + 1) * per; ++i) {
    const int v = wg_count[i];
    wg_count[i] = run;
    run += v;
  }
}

// Output positions: the k-th accepted pair yields normals 2k (f*x2, what legacy_gauss returns) and
// 2k+1 (f*x1, the value it caches); with a cached value at entry (`shift` = 1) everything moves up
// by one.  Element e of the stream belongs to the problem whose [eps_off, eps_off + N*H*nu)
// contains it and is scaled by that problem's sqrt(sigma) (loc + scale * g with loc = 0).
// fin[0] = index of the attempt that produced the last needed pair, fin[1..2] = bits of the
// unscaled value left in the cache (valid when an odd number of values was consumed).
template <typename T>
__global__ __launch_bounds__(256) void polar_scatter_kernel(const uint32_t* __restrict__ u, int n_attempts,
                                                            const int* __restrict__ wg_offset,
                                                            long long n_values, int shift,
                                                            const MppiProblem<T>* __restrict__ probs,
                                                            const double* __restrict__ scale,
                                                            int n_probs, T* __restrict__ eps,
                                                            long long* __restrict__ fin) {
#pragma clang fp contract(off)
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long n_pairs = (n_values - shift + 1) / 2;     // pairs that have to be produced
  int base = wg_offset[blockIdx.x];
  for (int k = 0; k < 4; ++k) {
    const int a = blockIdx.x * kPolarPerWg + k * 256 + tid;
    double x1 = 0, x2 = 0, r2 = 1;
    const bool acc = a < n_attempts && polar_attempt(u + 4 * (size_t)a, &x1, &x2, &r2);
    // exclusive scan of the flags over the workgroup
    const unsigned long long m = __ballot(acc);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += wsum[w];
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const long long kpair = (long long)base + woff + before;
    if (acc && kpair < n_pairs) {
      const double f = sqrt(__dmul_rn(-2.0, log(r2)) / r2);
      const double g0 = __dmul_rn(f, x2), g1 = __dmul_rn(f, x1);
      const long long e0 = shift + 2 * kpair;
      for (int j = 0; j < 2; ++j) {
        const long long e = e0 + j;
        const double g = j ? g1 : g0;
        if (e < n_values) {
          int b = 0;
          while (b + 1 < n_probs && e >= probs[b + 1].eps_off) ++b;     // eps_off ascending in b
          eps[e] = (T)__dmul_rn(scale[b], g);       // loc + scale * g with loc = 0, in double as numpy
        } else {
          fin[1] = __double_as_longlong(g);                              // stays in the cache
        }
      }
      if (kpair == n_pairs - 1) fin[0] = a;
    }
    base += tot;
  }
}

// the cached value at entry is the first normal of the call
template <typename T>
__global__ void legacy_first_value_kernel(double cached, const double* __restrict__ scale,
                                          T* __restrict__ eps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) eps[0] = (T)__dmul_rn(scale[0], cached);
}

}  // namespace ampc
