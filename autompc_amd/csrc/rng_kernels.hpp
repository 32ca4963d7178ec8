// rng_kernels.hpp -- counter-based normal noise on the device (Philox4x32-10 + Box-Muller).
//
// "Fast mode" replacement for the reference's host-side draw
//   np.random.normal(scale=sqrt(sigma), size=(N, H, nu))     (autompc/control/mppi.py:21-24, :126)
// The reference uses numpy's legacy global MT19937 stream; that stream is reproduced only by
// generating the noise on the host with numpy ("parity mode": ampc_mppi_upload).  This kernel is
// statistically equivalent (same N(0, sigma) marginals, independent across samples/steps/dims)
// but not bit-identical, and results obtained with it are compared distributionally only.
#pragma once
#include <hip/hip_runtime.h>

#include "mppi_kernels.hpp"
#include <stdint.h>

namespace ampc {

struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

// Philox4x32 with 10 rounds (Salmon et al., SC'11): counter c, key k.
__host__ __device__ inline Philox4 philox4x32_10(Philox4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(M0, c.v[0]), lo0 = M0 * c.v[0];
    const uint32_t hi1 = mulhi32(M1, c.v[2]), lo1 = M1 * c.v[2];
    Philox4 n;
    n.v[0] = hi1 ^ c.v[1] ^ k0;
    n.v[1] = lo1;
    n.v[2] = hi0 ^ c.v[3] ^ k1;
    n.v[3] = lo0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// One Philox block -> two uniforms in (0,1) with 53 (f64) bits -> two standard normals.
// Counter = (pair index [40 bits], stream [56 bits], noise id [32 bits]); key = seed.  `stream` is
// the caller's step counter, `id` the problem's noise id: (seed, stream, id, element) names a value
// independently of which plan / GPU / position in the batch the problem occupies.
template <typename T>
__device__ __forceinline__ void philox_normal_pair(T* __restrict__ out, long long count, T scale,
                                                   uint64_t seed, uint64_t stream, uint32_t id,
                                                   long long pair) {
  const long long e0 = 2 * pair;
  if (e0 >= count) return;
  Philox4 c;
  c.v[0] = (uint32_t)pair;
  c.v[1] = (uint32_t)(((uint64_t)pair >> 32) & 0xffu) | (uint32_t)((stream >> 32) << 8);
  c.v[2] = (uint32_t)stream;
  c.v[3] = id;
  const Philox4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = ((uint64_t)r.v[0] << 32) | r.v[1];
  const uint64_t b = ((uint64_t)r.v[2] << 32) | r.v[3];
  const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double s, cth;
  sincospi(2.0 * u2, &s, &cth);
  out[e0] = (T)(rad * cth) * scale;
  if (e0 + 1 < count) out[e0 + 1] = (T)(rad * s) * scale;
}

// One element of the same stream (value e of a problem's draw): what philox_normal_pair stores at
// out[e], recomputed by whoever needs it -- the four-row rollout forms its noise in its prologue
// instead of reading a buffer another launch filled (mppi_rollout4.hpp).  Bit-identical to the pair
// form: the same block, the same radius, the cosine (even e) or sine (odd e) branch.
template <typename T>
__device__ __forceinline__ T philox_normal_elem(T scale, uint64_t seed, uint64_t stream, uint32_t id, long long e) {
  const long long pair = e >> 1;
  Philox4 c;
  c.v[0] = (uint32_t)pair;
  c.v[1] = (uint32_t)(((uint64_t)pair >> 32) & 0xffu) | (uint32_t)((stream >> 32) << 8);
  c.v[2] = (uint32_t)stream;
  c.v[3] = id;
  const Philox4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = ((uint64_t)r.v[0] << 32) | r.v[1];
  const uint64_t b = ((uint64_t)r.v[2] << 32) | r.v[3];
  const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double s, cth;
  sincospi(2.0 * u2, &s, &cth);
  return (T)(rad * ((e & 1) ? s : cth)) * scale;
}

// The noise of every problem of a plan in one launch: grid.y * grid.z covers the problems; problem
// b draws N_b*H_b*nu values of std sqrt(sigma_b) keyed by (seed, stream, noise_id_b).
template <typename T>
__global__ void philox_normal_batch_kernel(T* __restrict__ eps, const MppiProblem<T>* __restrict__ probs,
                                           int n_probs, int nu, uint64_t seed, uint64_t stream) {
  const int b = blockIdx.z * gridDim.y + blockIdx.y;
  if (b >= n_probs) return;
  const MppiProblem<T> pr = probs[b];
  philox_normal_pair<T>(eps + pr.eps_off, (long long)pr.N * pr.H * nu, pr.sqrt_sigma, seed, stream,
                        pr.noise_id, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

}  // namespace ampc
