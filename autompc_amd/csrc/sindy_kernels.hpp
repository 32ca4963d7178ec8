// sindy_kernels.hpp -- SINDy feature-library dynamics on gfx950 (scalar path, one thread per sample).
//
// x' = Theta([x,u]) Xi'            (discrete)      reference: autompc/sysid/sindy.py:173-179
// x' = x + dt Theta([x,u]) Xi'     (continuous)
// Theta is the reference's CustomLibrary: identity, sin/cos(f v), the four trig interaction
// terms v_a sin(f v_b) / v_a cos(f v_b) in both argument orders, powers v^d
// (sindy.py:134-152, basis_funcs.py:8-126).  The model is tiny (CartPole: 5 variables, 55
// features, 4x55 coefficients): this is VALU / latency-bound plumbing for BASELINE config 1, not
// an MFMA workload, so one thread owns one sample and keeps its state in LDS columns
// ([i][lane]: conflict-free).  Pinned by tests/golden/sindy_*.npz: outputs of the reference's own
// pred_batch / pred_diff_batch over a stand-in for the absent pysindy package that restates only
// CustomLibrary's feature enumeration (tests/golden/gen_golden.py, oracle/sindy.py).
// Polynomial cross terms (basis_funcs.py:27-93, sindy.py:143-145) are monomial features: a product
// of up to 10 integer powers of distinct variables, described by (variable, exponent) pairs.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_kernels.hpp"
#include "mppi_kernels.hpp"

namespace ampc {

enum { SF_ID = 0, SF_SIN = 1, SF_COS = 2, SF_XSIN = 3, SF_XCOS = 4, SF_POW = 5, SF_MONO = 6 };

template <typename T> struct SindyDev {
  int nx, nu, n_feat, continuous, strict;   // strict: reproduce the reference's Jacobian quirks
  T dt;
  const int* kind;      // [n_feat]
  const int* a0;        // [n_feat] first variable (the multiplier for interaction terms)
  const int* a1;        // [n_feat] second variable (argument of sin/cos for interaction terms)
  const T* par;         // [n_feat] frequency / exponent
  const T* xi;          // [nx][n_feat]
  // Product form (n_tab > 0).  Per step a thread first fills a small table -- sin / cos of the
  // distinct (variable, frequency) arguments the library uses (CartPole: 5 for 40 trigonometric
  // features), the distinct powers, and a constant 1 -- and then every feature is the product of
  // two entries: fx[k] >= 0 selects variable fx[k], fx[k] < 0 table entry -fx[k]-1; fy[k] is a
  // table entry (the constant for single-factor features).  No data-dependent branches, so the
  // feature loop unrolls and its loads overlap.  n_tab == 0: features are evaluated directly from
  // (kind, a0, a1, par) -- libraries whose table would not fit.
  int n_trig, n_pow, n_tab;   // n_tab = 2 n_trig + n_pow + n_mon + 1
  // Monomial features (kind SF_MONO: a0 = first (variable, exponent) pair in mpool, a1 = number of
  // pairs).  In product form every monomial is a table entry behind the powers.
  int n_mon, n_pool;
  const int* moff;      // [n_mon] first pair of table monomial j
  const int* mcnt;      // [n_mon] its number of pairs
  const int* mpool;     // [n_pool][2] (variable, exponent >= 1)
  const int* fx;        // [n_feat]
  const int* fy;        // [n_feat]
  const int* tvar;      // [n_trig] variable index of a trig argument
  const T* tpar;        // [n_trig] its frequency
  const int* pvar;      // [n_pow] variable index of a power
  const T* ppar;        // [n_pow] its exponent
  int stage;            // 1: the kernels copy the program to LDS first (sindy_stage)
};
constexpr int kSindyMaxTab = 160;   // table entries kept per thread (in LDS columns)
constexpr int kSindyStageBytes = 48 * 1024;   // programs up to this size are copied to LDS

// Elements of T the staged program occupies (floats first, then the int arrays, 8-byte aligned).
__host__ __device__ inline size_t sindy_prog_elems(int nx, int n_feat, int n_trig, int n_pow, int n_mon,
                                                   int n_pool, size_t tsize) {
  const size_t flt = (size_t)n_feat * nx + n_trig + n_pow;
  const size_t ints = (size_t)2 * n_feat + n_trig + n_pow + 2 * (size_t)n_mon + 2 * (size_t)n_pool;
  return flt + (ints * sizeof(int) + tsize - 1) / tsize + 2;
}

// Copy the feature program (descriptors, coefficients, trig table) from global memory into the
// workgroup's LDS and return a descriptor that points there.  Every per-feature operand of
// sindy_step is wave-uniform; from global memory each is a dependent scalar-load round trip per
// feature, from LDS a broadcast read (measured 5x on the CartPole library).  Must be called by
// all threads; ends with a barrier.  g.stage == 0 (program too large): returns g unchanged.
// ALWAYS: the caller is only launched for staged programs (g.stage == 1) -- without the early return every
// pointer of the result derives from the LDS array, so the compiler addresses the program with ds_read
// instead of flat loads (a pointer that MAY be global is a flat pointer: ~5 x the latency per operand, and
// every read counts against vmcnt as well).
template <typename T, bool ALWAYS = false>
__device__ __forceinline__ SindyDev<T> sindy_stage(const SindyDev<T>& g, T* area, int tid, int nthr);

template <typename T, bool ALWAYS>
__device__ __forceinline__ SindyDev<T> sindy_stage(const SindyDev<T>& g, T* area, int tid, int nthr) {
  if constexpr (!ALWAYS) {
    if (!g.stage) return g;
  }
  const int nf = g.n_feat, nt = g.n_trig, np = g.n_pow, nx = g.nx;
  T* xi = area;
  T* tpar = xi + (size_t)nx * nf;
  T* ppar = tpar + nt;
  int* fx = reinterpret_cast<int*>(ppar + np + 1);
  int* fy = fx + nf; int* tvar = fy + nf; int* pvar = tvar + nt;
  int* moff = pvar + np; int* mcnt = moff + g.n_mon; int* mpool = mcnt + g.n_mon;
  for (int j = tid; j < g.n_mon; j += nthr) { moff[j] = g.moff[j]; mcnt[j] = g.mcnt[j]; }
  for (int j = tid; j < 2 * g.n_pool; j += nthr) mpool[j] = g.mpool[j];
  for (int k = tid; k < nf; k += nthr) { fx[k] = g.fx[k]; fy[k] = g.fy[k]; }
  for (int e = tid; e < nx * nf; e += nthr) xi[e] = g.xi[e];
  for (int j = tid; j < nt; j += nthr) { tpar[j] = g.tpar[j]; tvar[j] = g.tvar[j]; }
  for (int j = tid; j < np; j += nthr) { ppar[j] = g.ppar[j]; pvar[j] = g.pvar[j]; }
  __syncthreads();
  SindyDev<T> s = g;
  s.xi = xi; s.tpar = tpar; s.ppar = ppar; s.fx = fx; s.fy = fy; s.tvar = tvar; s.pvar = pvar;
  s.moff = moff; s.mcnt = mcnt; s.mpool = mpool;
  return s;
}

// x ** e for a small integer e >= 0 by repeated multiplication (the reference multiplies
// arg ** exp factors into a running product, basis_funcs.py:40-44; <= 1 ulp per factor either way)
template <typename T> __device__ __forceinline__ T sindy_ipow(T x, int e) {
  T r = T(1);
  for (int i = 0; i < e; ++i) r *= x;
  return r;
}

// prod_j v[var_j] ** exp_j over `cnt` pairs starting at pair `off`; skip >= 0: pair `skip`
// contributes exp * v ** (exp - 1) instead (the partial derivative, basis_funcs.py:74-84)
template <typename T>
__device__ __forceinline__ T sindy_monomial(const int* __restrict__ pool, int off, int cnt, const T* v,
                                            int vs, int skip = -1) {
  T val = T(1);
  for (int j = 0; j < cnt; ++j) {
    const int var = pool[2 * (off + j)], e = pool[2 * (off + j) + 1];
    const T x = v[var * vs];
    val *= (j == skip) ? T(e) * sindy_ipow<T>(x, e - 1) : sindy_ipow<T>(x, e);
  }
  return val;
}

template <typename T>
__device__ __forceinline__ T sindy_feature(int kind, T va, T vb, T par) {
  switch (kind) {
    case SF_ID: return va;
    case SF_SIN: return sin(par * va);
    case SF_COS: return cos(par * va);
    case SF_XSIN: return va * sin(par * vb);
    case SF_XCOS: return va * cos(par * vb);
    default: return pow(va, par);
  }
}

// v: this thread's variables, element i at v[i * vs]; out: next state at out[i * os];
// tr: this thread's table scratch (n_tab values, element j at tr[j * ts]).
template <typename T>
__device__ __forceinline__ void sindy_step(const SindyDev<T>& m, const T* v, int vs, T* out, int os,
                                           T* tr, int ts) {
  constexpr int NR = 8;
  if (m.n_tab > 0) {
    // pass 1: the table
    for (int j = 0; j < m.n_trig; ++j) {
      const T arg = m.tpar[j] * v[m.tvar[j] * vs];
      tr[(2 * j) * ts] = sin(arg);
      tr[(2 * j + 1) * ts] = cos(arg);
    }
    for (int j = 0; j < m.n_pow; ++j) tr[(2 * m.n_trig + j) * ts] = pow(v[m.pvar[j] * vs], m.ppar[j]);
    for (int j = 0; j < m.n_mon; ++j)
      tr[(2 * m.n_trig + m.n_pow + j) * ts] = sindy_monomial<T>(m.mpool, m.moff[j], m.mcnt[j], v, vs);
    tr[(m.n_tab - 1) * ts] = T(1);
    // pass 2: features as products of two entries
    auto feature = [&](int k) -> T {
      const int ix = m.fx[k], iy = m.fy[k];
      const T xv = v[(ix >= 0 ? ix : 0) * vs];
      const T xt = tr[(ix >= 0 ? 0 : -ix - 1) * ts];
      return (ix >= 0 ? xv : xt) * tr[iy * ts];
    };
    if (m.nx <= NR) {
      // few outputs (CartPole: 4): accumulate in registers; through `out` every term would be an
      // LDS read-modify-write in a dependent chain.  Coefficient loads are unguarded (row index
      // clamped) so that they are all in flight together; rows >= nx are never stored.
      T acc[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) acc[i] = T(0);
      const int last = m.nx - 1;
#pragma unroll 4
      for (int k = 0; k < m.n_feat; ++k) {
        T c[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) c[i] = m.xi[(i < last ? i : last) * m.n_feat + k];
        const T f = feature(k);
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[i] += c[i] * f;
      }
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (i < m.nx) out[i * os] = m.continuous ? v[i * vs] + m.dt * acc[i] : acc[i];
      return;
    }
    for (int i = 0; i < m.nx; ++i) out[i * os] = T(0);
    for (int k = 0; k < m.n_feat; ++k) {
      const T f = feature(k);
      for (int i = 0; i < m.nx; ++i) out[i * os] += m.xi[i * m.n_feat + k] * f;
    }
  } else {
    for (int i = 0; i < m.nx; ++i) out[i * os] = T(0);
    for (int k = 0; k < m.n_feat; ++k) {
      const T f = m.kind[k] == SF_MONO
                      ? sindy_monomial<T>(m.mpool, m.a0[k], m.a1[k], v, vs)
                      : sindy_feature<T>(m.kind[k], v[m.a0[k] * vs], v[m.a1[k] * vs], m.par[k]);
      for (int i = 0; i < m.nx; ++i) out[i * os] += m.xi[i * m.n_feat + k] * f;
    }
  }
  if (m.continuous)
    for (int i = 0; i < m.nx; ++i) out[i * os] = v[i * vs] + m.dt * out[i * os];
}

template <typename T>
__global__ void sindy_forward_kernel(const SindyDev<T> mg, const T* __restrict__ states,
                                     const T* __restrict__ ctrls, T* __restrict__ out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  const SindyDev<T> m = sindy_stage<T>(mg, lds + (2 * mg.nx + mg.nu + mg.n_tab) * blockDim.x,
                                       threadIdx.x, blockDim.x);
  const int lane = threadIdx.x, nv = m.nx + m.nu, bs = blockDim.x;
  const int r = blockIdx.x * bs + lane;
  T* v = lds + lane;                       // [nv][bs]
  T* o = lds + nv * bs + lane;             // [nx][bs]
  T* tr = lds + (nv + m.nx) * bs + lane;   // [n_tab][bs]
  if (r < n) {
    for (int i = 0; i < m.nx; ++i) v[i * bs] = states[(size_t)r * m.nx + i];
    for (int j = 0; j < m.nu; ++j) v[(m.nx + j) * bs] = ctrls[(size_t)r * m.nu + j];
    sindy_step<T>(m, v, bs, o, bs, tr, bs);
    for (int i = 0; i < m.nx; ++i) out[(size_t)r * m.nx + i] = o[i * bs];
  }
}

// Jacobian of the step wrt [x, u] (sindy.py:189-244).  strict: interaction features are counted
// twice and the polynomial gradient omits the exponent factor, as the reference computes them.
template <typename T>
__global__ void sindy_jacobian_kernel(const SindyDev<T> m, const T* __restrict__ states,
                                      const T* __restrict__ ctrls, T* __restrict__ jx,
                                      T* __restrict__ ju, int n, const RowMap rm) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (!rm.plain() && !rm.live(r)) return;
  const int nx = m.nx, nu = m.nu;
  states += (size_t)(r / rm.grp) * rm.s_stride + (size_t)(r % rm.grp) * nx - (size_t)r * nx;
  ctrls += (size_t)(r / rm.grp) * rm.c_stride + (size_t)(r % rm.grp) * nu - (size_t)r * nu;
  T* Jx = jx + (size_t)r * nx * nx;
  T* Ju = ju + (size_t)r * nx * nu;
  for (int i = 0; i < nx * nx; ++i) Jx[i] = T(0);
  for (int i = 0; i < nx * nu; ++i) Ju[i] = T(0);
  auto var = [&](int c) { return c < nx ? states[(size_t)r * nx + c] : ctrls[(size_t)r * nu + (c - nx)]; };
  auto add = [&](int i, int c, T g) {
    if (c < nx) Jx[i * nx + c] += g;
    else Ju[i * nu + (c - nx)] += g;
  };
  const T twice = m.strict ? T(2) : T(1);
  for (int k = 0; k < m.n_feat; ++k) {
    const int kind = m.kind[k], c0 = m.a0[k], c1 = m.a1[k];
    if (kind == SF_MONO) {
      // d/dv_i of prod_j v_j ** e_j (basis_funcs.py:74-84).  The reference's name lookup finds a
      // cross-term feature through exactly one argument order, so nothing is counted twice here.
      for (int f = 0; f < c1; ++f) {
        T g = T(1);
        for (int j = 0; j < c1; ++j) {
          const int vj = m.mpool[2 * (c0 + j)], e = m.mpool[2 * (c0 + j) + 1];
          g *= (j == f) ? T(e) * sindy_ipow<T>(var(vj), e - 1) : sindy_ipow<T>(var(vj), e);
        }
        const int vf = m.mpool[2 * (c0 + f)];
        for (int i = 0; i < nx; ++i) add(i, vf, m.xi[i * m.n_feat + k] * g);
      }
      continue;
    }
    const T va = var(c0), vb = var(c1), par = m.par[k];
    T g0 = T(0), g1 = T(0);
    switch (kind) {
      case SF_ID: g0 = T(1); break;
      case SF_SIN: g0 = par * cos(par * va); break;
      case SF_COS: g0 = -par * sin(par * va); break;
      case SF_XSIN: g0 = twice * sin(par * vb); g1 = twice * va * par * cos(par * vb); break;
      case SF_XCOS: g0 = twice * cos(par * vb); g1 = -twice * va * par * sin(par * vb); break;
      default: g0 = (m.strict ? T(1) : par) * pow(va, par - T(1)); break;
    }
    for (int i = 0; i < nx; ++i) {
      const T c = m.xi[i * m.n_feat + k];
      add(i, c0, c * g0);
      if (kind == SF_XSIN || kind == SF_XCOS) add(i, c1, c * g1);
    }
  }
  if (m.continuous) {
    for (int i = 0; i < nx; ++i) {
      for (int c = 0; c < nx; ++c) Jx[i * nx + c] = (i == c ? T(1) : T(0)) + m.dt * Jx[i * nx + c];
      for (int c = 0; c < nu; ++c) Ju[i * nu + c] *= m.dt;
    }
  }
}

// MPPI rollout with SINDy dynamics: one thread per sample, one wave per workgroup.  Same
// semantics as mppi_rollout_kernel (mppi.py:120-152); costs / term_last / eps_out go to HBM and
// mppi_update_kernel finishes the solve.
template <typename T>
__global__ __launch_bounds__(64) void mppi_rollout_sindy_kernel(const MppiArgs<T> args,
                                                                const SindyDev<T> mg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int BS = 64;
  const SindyDev<T> m = sindy_stage<T>(
      mg, lds + (2 * mg.nx + mg.nu + mg.n_tab) * BS + args.cost_stride + 3 * mg.nu + 1, threadIdx.x, BS);
  const int lane = threadIdx.x, nx = m.nx, nu = m.nu, nv = nx + nu, no = args.obs_dim;
  const int p = args.tile_prob[blockIdx.x];
  const MppiProblem<T> pr = args.probs[p];
  const int n = (blockIdx.x - pr.tile0) * BS + lane;
  const int H = pr.H, N = pr.N;
  const bool valid = n < N;
  T* v = lds + lane;                        // [nv][BS]  x | u
  T* o = lds + nv * BS + lane;              // [nx][BS]  next state
  T* tr = lds + (nv + nx) * BS + lane;      // [n_tab][BS]  per-thread table
  T* cpar = lds + (nv + nx + m.n_tab) * BS;   // Q R F goal | lo hi scale (shared)
  for (int i = lane; i < args.cost_stride; i += BS)
    cpar[i] = args.costs_par[(size_t)pr.cost_idx * args.cost_stride + i];
  for (int i = lane; i < 3 * nu; i += BS) cpar[args.cost_stride + i] = args.bounds[i];
  __syncthreads();
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;
  const T* lin = goal + no; const T* lint = lin + no;       // affine part (all zero for one QuadCost)
  const T* blo = cpar + args.cost_stride; const T* bhi = blo + nu; const T* bsc = bhi + nu;
  for (int i = 0; i < nx; ++i) v[i * BS] = args.x0[p * nx + i];
  const T* eps_row = args.eps + pr.eps_off + (size_t)(valid ? n : 0) * H * nu;
  T* epso = args.eps_out + pr.epso_off;
  T c = T(0), ca = T(0);
  for (int t = 0; t < H; ++t) {
    const int ts = (t + 1 < H) ? t + 1 : H - 1;               // a[:-1] = a[1:]; a[-1] = a[-2]
    for (int j = 0; j < nu; ++j) {
      const T a = args.act_in[pr.a_off + ts * nu + j];
      T A = (valid ? eps_row[t * nu + j] : T(0)) + a;
      A = A < blo[j] ? blo[j] : A;
      A = A > bhi[j] ? bhi[j] : A;
      const T ec = A - a;
      if (valid) epso[((size_t)t * N + n) * nu + j] = ec;
      ca += A * ec;
      v[(nx + j) * BS] = A * bsc[j];
    }
    for (int i = 0; i < no; ++i) {
      T s = T(0);
      for (int j = 0; j < no; ++j) s += Qm[i * no + j] * (v[j * BS] - goal[j]);
      c += (v[i * BS] - goal[i]) * (s + lin[i]);
    }
    c += lint[no];
    if (args.n_ind) c += indicator_all<T>(args.ind_tab, args.n_ind, v, BS, no);
    for (int i = 0; i < nu; ++i) {
      T s = T(0);
      for (int j = 0; j < nu; ++j) s += Rm[i * nu + j] * v[(nx + j) * BS];
      c += v[(nx + i) * BS] * s;
    }
    sindy_step<T>(m, v, BS, o, BS, tr, BS);
    for (int i = 0; i < nx; ++i) v[i * BS] = o[i * BS];
  }
  T term = T(0);
  for (int i = 0; i < no; ++i) {
    T s = T(0);
    for (int j = 0; j < no; ++j) s += Fm[i * no + j] * (v[j * BS] - goal[j]);
    term += (v[i * BS] - goal[i]) * (s + lint[i]);
  }
  term += lint[no + 1];
  c += pr.lam_over_sigma * ca;
  if (args.term_mode == 1) c += term;
  if (valid) {
    args.costs[pr.cost_off + n] = c;
    if (n == N - 1) args.term_last[p] = term;
  }
}


// The same rollout with the FEATURES spread over lanes (round 4; BASELINE config 1).  One thread per sample
// leaves the machine idle on the small problems this model family is used for (CartPole: 256 samples = four
// waves) and makes a time step a chain of ~55 dependent LDS round trips (36 k cycles).  Here G = 16 lanes
// share a sample: lane g forms the table entries j = g, g + G, ... (one sin / cos pair each instead of five
// in a row), then the features k = g, g + G, ... and their nx products with the coefficient column, and an
// xor-butterfly over the G lanes leaves the sums in every lane; state, controls and cost are kept
// redundantly by all lanes of the group (the same arithmetic, so they agree bit for bit) and only lane 0
// stores.  Block b of the launch takes samples [64 (b / G) + 4 (b % G), + 4) -- sub-block b % G of the plan's
// 64-sample tile b / G -- so the plan's tiling, offsets and update kernel are untouched.  A sample's sum over
// the features is formed in a different order than mppi_rollout_sindy_kernel forms it (G partial sums, then
// the butterfly): results agree to rounding (tests/test_gpu_sindy.py pins both to the reference's goldens).
// Shapes it does not take (nx > 8, a table beyond kSindyMaxTab, programs that are not staged) keep the kernel
// above.
// Sum over the G lanes of a group, left in every lane: quad, half-row and row stages are DPP moves (no LDS
// round trip), the 16- and 32-lane stages shuffles.
template <int CTRL, typename T> __device__ __forceinline__ T sindy_dpp(T v) {
  if constexpr (sizeof(T) == 8) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  } else {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
  }
}
template <int G, typename T> __device__ __forceinline__ T sindy_group_sum(T v) {
  v += sindy_dpp<0xb1>(v);                         // quad_perm [1,0,3,2]
  v += sindy_dpp<0x4e>(v);                         // quad_perm [2,3,0,1]
  v += sindy_dpp<0x141>(v);                        // row_half_mirror
  v += sindy_dpp<0x140>(v);                        // row_mirror: all 16 lanes of a row hold its sum
  if constexpr (G >= 32) v += __shfl_xor(v, 16);
  if constexpr (G >= 64) v += __shfl_xor(v, 32);
  return v;
}

// (G lanes per sample: 16, 32 or 64 -- the fewer samples a plan has, the more lanes each can use; hcap =
//  max_h * nu rounded up: a sample's noise row and the shifted action sequence are read into LDS once,
//  ahead of the time loop, instead of a global-memory round trip per step on the chain)
// IND: the handle's cost has indicator terms (a separate instantiation: the never-taken branch cost the c1 kernel
//  10 % through scheduling alone, 0.037 -> 0.041 ms)
template <typename T, int G, bool IND = false>
__global__ __launch_bounds__(64) void mppi_rollout_sindy_fp_kernel(const MppiArgs<T> args, const SindyDev<T> mg,
                                                                   const int hcap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int BS = 64, SPB = BS / G, NR = 8;
  const int lane = threadIdx.x, nx = mg.nx, nu = mg.nu, nv = nx + nu, no = args.obs_dim, ntab = mg.n_tab;
  // LDS: per sample v [nv], table [ntab], noise row [hcap], shifted actions [hcap]; then the shared cost
  // block / bounds; then the staged program
  const int per_s = nv + ntab + 2 * hcap;
  T* cpar = lds + SPB * per_s;
  const SindyDev<T> m = sindy_stage<T, true>(mg, cpar + args.cost_stride + 3 * nu + 1, lane, BS);
  const int tile = blockIdx.x / G, sub = blockIdx.x - tile * G;
  const int p = args.tile_prob[tile];
  const MppiProblem<T> pr = args.probs[p];
  const int sl = lane / G, g = lane - sl * G;               // sample of the block, lane of the group
  const int n = (tile - pr.tile0) * BS + sub * SPB + sl;
  const int H = pr.H, N = pr.N;
  const bool valid = n < N;
  T* v = lds + sl * per_s;                                   // this sample's x | u
  T* tr = v + nv;                                            // ... and its table
  T* er = tr + ntab;                                         // ... its noise row [H][nu]
  T* ar = er + hcap;                                         // ... and a[min(t + 1, H - 1)][nu]
  for (int i = lane; i < args.cost_stride; i += BS)
    cpar[i] = args.costs_par[(size_t)pr.cost_idx * args.cost_stride + i];
  for (int i = lane; i < 3 * nu; i += BS) cpar[args.cost_stride + i] = args.bounds[i];
  for (int i = g; i < nx; i += G) v[i] = args.x0[p * nx + i];
  if (g == 0) tr[ntab - 1] = T(1);
  {
    const T* eps_row = args.eps + pr.eps_off + (size_t)(n < pr.N ? n : 0) * pr.H * nu;
    for (int e = g; e < pr.H * nu; e += G) {
      const int t = e / nu, j = e - t * nu, ts = (t + 1 < pr.H) ? t + 1 : pr.H - 1;   // a[:-1] = a[1:]; a[-1] = a[-2]
      er[e] = n < pr.N ? eps_row[e] : T(0);
      ar[e] = args.act_in[pr.a_off + ts * nu + j];
    }
  }
  __syncthreads();
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;
  const T* lin = goal + no; const T* lint = lin + no;
  const T* blo = cpar + args.cost_stride; const T* bhi = blo + nu; const T* bsc = bhi + nu;
  T* epso = args.eps_out + pr.epso_off;
  // What a lane needs in every step does not change over the steps: it is read ONCE, here -- its trig
  // argument (variable, frequency), its first KF features (two table indices, nx coefficients), the diagonal
  // of the cost block.  The state lives in registers of every lane (all lanes of a group hold the sums after
  // the group reduction); the LDS copy of x | u exists for the data-dependent reads only (a trig argument, a
  // feature's factors).  Per step that leaves: noise + action, the trig argument, the feature's two factors
  // and the two cross-row reduction stages as LDS round trips (the loop used to have sixty).
  const bool cdiag = args.cost_diag != 0;
  constexpr int KF = 2;
  T xreg[NR], qd[NR], gl[NR], ln[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int ix = i < no ? i : 0;
    xreg[i] = v[i < nx ? i : 0];
    qd[i] = Qm[ix * no + ix]; gl[i] = goal[ix]; ln[i] = lin[ix];
  }
  const T lc = lint[no];
  const bool t0 = g < m.n_trig;
  const int tv0 = t0 ? m.tvar[g] : 0;
  const T tp0 = t0 ? m.tpar[g] : T(0);
  int fxr[KF], fyr[KF];
  bool fk[KF];
  T xir[KF][NR];
#pragma unroll
  for (int q = 0; q < KF; ++q) {
    const int k = g + q * G;
    fk[q] = k < m.n_feat;
    const int kk = fk[q] ? k : 0;
    fxr[q] = m.fx[kk]; fyr[q] = m.fy[kk];
#pragma unroll
    for (int i = 0; i < NR; ++i) xir[q][i] = (fk[q] && i < nx) ? m.xi[(i < nx ? i : 0) * m.n_feat + kk] : T(0);
  }
  T c = T(0), ca = T(0);
  for (int t = 0; t < H; ++t) {
    for (int j = 0; j < nu; ++j) {
      const T a = ar[t * nu + j];
      T A = er[t * nu + j] + a;
      A = A < blo[j] ? blo[j] : A;
      A = A > bhi[j] ? bhi[j] : A;
      const T ec = A - a;
      if (valid && g == 0) epso[((size_t)t * N + n) * nu + j] = ec;
      ca += A * ec;
      const T uj = A * bsc[j];
      if (g == 0) v[nx + j] = uj;
      if (cdiag) c += uj * (Rm[j * nu + j] * uj);
    }
    // (the group's lanes run in one wave: LDS accesses of a wave complete in order, no barrier needed)
    // stage cost; diagonal blocks: the off-diagonal terms the dense loops add are exact zeros, skipped
    if (cdiag) {
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (i < no) {
          const T d = xreg[i] - gl[i];
          c += d * (qd[i] * d + ln[i]);
        }
    } else {
      for (int i = 0; i < no; ++i) {
        T s = T(0);
        for (int j = 0; j < no; ++j) s += Qm[i * no + j] * (v[j] - goal[j]);
        c += (v[i] - goal[i]) * (s + lin[i]);
      }
      for (int i = 0; i < nu; ++i) {
        T s = T(0);
        for (int j = 0; j < nu; ++j) s += Rm[i * nu + j] * v[nx + j];
        c += v[nx + i] * s;
      }
    }
    c += lc;
    if constexpr (IND) {                   // indicator terms (threshold / box, mlp_tile.hpp), from the register copy
      for (int k = 0; k < args.n_ind; ++k) {
        const T* tk = args.ind_tab + (size_t)k * ind_stride(no);
        const int kind = (int)tk[0];
        bool viol = false, nan_in = false;
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (i < no) {
            viol = viol || ind_entry<T>(kind, xreg[i], tk[2 + i], tk[2 + no + i]);
            nan_in = nan_in || ind_void<T>(kind, xreg[i], tk[2 + no + i]);
          }
        if (viol && !nan_in) c += T(1);
      }
    }
    // ---- table entries of this lane (the first one from hoisted operands)
    if (t0) {
      const T arg = tp0 * v[tv0];
      T sv, cv;
      if constexpr (sizeof(T) == 8) sincos(arg, &sv, &cv);    // (one argument reduction for the pair)
      else sincosf(arg, &sv, &cv);
      tr[2 * g] = sv;
      tr[2 * g + 1] = cv;
    }
    for (int j = g + G; j < m.n_trig; j += G) {
      const T arg = m.tpar[j] * v[m.tvar[j]];
      T sv, cv;
      if constexpr (sizeof(T) == 8) sincos(arg, &sv, &cv);
      else sincosf(arg, &sv, &cv);
      tr[2 * j] = sv;
      tr[2 * j + 1] = cv;
    }
    for (int j = g; j < m.n_pow; j += G) tr[2 * m.n_trig + j] = pow(v[m.pvar[j]], m.ppar[j]);
    for (int j = g; j < m.n_mon; j += G)
      tr[2 * m.n_trig + m.n_pow + j] = sindy_monomial<T>(m.mpool, m.moff[j], m.mcnt[j], v, 1);
    // ---- features of this lane, times their coefficient column
    T acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = T(0);
#pragma unroll
    for (int q = 0; q < KF; ++q) {
      // (lanes without a q-th feature read entry 0 and multiply by zero coefficients: no branch around the reads)
      const int ix = fxr[q];
      const T fa = v[ix >= 0 ? ix : 0], fb = tr[ix >= 0 ? 0 : -ix - 1], fc = tr[fyr[q]];
      const T f = (ix >= 0 ? fa : fb) * fc;
#pragma unroll
      for (int i = 0; i < NR; ++i) acc[i] += xir[q][i] * f;
    }
    for (int k = g + KF * G; k < m.n_feat; k += G) {
      const int ix = m.fx[k], iy = m.fy[k];
      const T f = (ix >= 0 ? v[ix] : tr[-ix - 1]) * tr[iy];
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (i < nx) acc[i] += m.xi[i * m.n_feat + k] * f;
    }
    // (no branch per sum: side by side their shuffle stages overlap; sums past nx are sums of zeros)
    if (nx <= NR / 2) {
#pragma unroll
      for (int i = 0; i < NR / 2; ++i) acc[i] = sindy_group_sum<G>(acc[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) acc[i] = sindy_group_sum<G>(acc[i]);
    }
    // ---- next state: in every lane's registers; lane i of the group refreshes the LDS copy
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i < nx) {
        xreg[i] = m.continuous ? xreg[i] + m.dt * acc[i] : acc[i];
        if (g == i % G) v[i] = xreg[i];
      }
  }
  T term = T(0);
  for (int i = 0; i < no; ++i) {
    T s = T(0);
    for (int j = 0; j < no; ++j) s += Fm[i * no + j] * (v[j] - goal[j]);
    term += (v[i] - goal[i]) * (s + lint[i]);
  }
  term += lint[no + 1];
  c += pr.lam_over_sigma * ca;
  if (args.term_mode == 1) c += term;
  if (valid && g == 0) {
    args.costs[pr.cost_off + n] = c;
    if (n == N - 1) args.term_last[p] = term;
  }
}

}  // namespace ampc
