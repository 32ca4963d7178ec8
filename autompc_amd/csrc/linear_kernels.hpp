// linear_kernels.hpp -- wide linear dynamics  x' = A x + B u  (65 .. 256 model states) on gfx950.
//
// autompc.sysid.ARX keeps `history` (1..10, arx.py:27,37-45) past observations and controls in its
// state: HalfCheetah (18 observations, 6 controls) at history 3 is already 66 states, at history 10
// it is 235; Koopman lifts (koopman.py:105-122) grow the same way.  Up to 64 states a linear model
// rides the MLP tile as an identity-activation network (ampc_set_linear, api.cpp); beyond that the
// tile's output layer (four 16-column tiles) and first layer (K <= 80) are too narrow, and a linear
// step needs none of the tile's layer machinery anyway: it is ONE product of the [16 samples] x K
// operand [x | u] with M' (M = [A | B], nx x K).
//
//   linear_forward_kernel   pred_batch / surrogate step (model.py:109-130, arx.py:151-154)
//   linear_rollout_kernel   the MPPI rollout (mppi.py:120-152), same contract as mppi_rollout_kernel:
//                           clipped actions, stage / action / terminal costs, fused softmin partials
//   linear_jacobian_kernel  pred_diff_batch: the Jacobians of a linear model are A and B (arx.py:156-164)
//
// One workgroup = 16 samples x W waves.  M is packed on the host in MFMA fragment order
// wf[tile nt][k-step ks][lane] = M[16 nt + (lane & 15)][4 ks + (lane >> 4)] (zero padded), so a wave's
// B operand for one k-step is one coalesced 64-element run; wave w owns output column tiles
// w, w + W, ...; the A operand [16][K] sits in LDS and is read once per (tile, k-step).  The state
// ping-pongs between two LDS buffers, so a time step needs ONE workgroup barrier.  Per step and
// workgroup: 2 * 16 * nx * K flops against nx * K weights streamed from L2 -- at 16 rows the f64 MFMA
// time (64 cycles per tile and k-step on one of four SIMDs) exceeds the stream time by 2x: MFMA-bound.
#pragma once
#include "mppi_kernels.hpp"

namespace ampc {

template <typename T> struct LinDev {
  int nx, nu, nxp, kp, ntile, ksn;     // nxp = 16 * ntile >= nx;  kp = 4 * ksn >= nx + nu
  const T* wf;                         // [ntile][ksn][64] fragments of M = [A | B]
  const T* plain;                      // [nx][nx + nu] row-major
  const T* jp;                         // [nxp][ldj] the same, zero padded to MFMA tiles (ldj = nx + nu rounded up to 16):
  int ldj;                             // the constant Jacobian of the wide iLQR sweep (ilqr_wide.hpp)
};

constexpr int kLinW = 8;                                     // waves per workgroup
constexpr int kLinMaxIlqrNx = 128;                           // iLQR on wide linear models: V [nx][nx] lives in LDS (ilqr_wide.hpp)
__host__ __device__ constexpr int lin_xs(int kp, int esz) {  // LDS row stride of [x | u]: odd in 8-byte units
  return esz == 8 ? (kp | 1) : ((kp + 2) | 2);
}
__host__ __device__ constexpr int lin_lds_base(int kp, int esz) { return (2 * 16 * lin_xs(kp, esz) + 3) / 4 * 4; }

// acc[r] = sum_k xu[row][k] M[16 nt + col][k] for this wave's tile nt: rows acc_row<T>(lane >> 4, r), column lane & 15
template <typename T>
__device__ __forceinline__ typename Acc<T>::type lin_tile(const LinDev<T>& m, const T* __restrict__ xu, int xs,
                                                          int nt, int lane) {
  using acc_t = typename Acc<T>::type;
  acc_t acc = {T(0), T(0), T(0), T(0)};
  const T* ap = xu + (lane & 15) * xs + (lane >> 4);
  const T* bp = m.wf + ((size_t)nt * m.ksn) * 64 + lane;
  int ks = 0;
  for (; ks + 4 <= m.ksn; ks += 4) {             // four fragments in flight per wave
    const T b0 = bp[(size_t)ks * 64], b1 = bp[(size_t)(ks + 1) * 64], b2 = bp[(size_t)(ks + 2) * 64],
            b3 = bp[(size_t)(ks + 3) * 64];
    const T a0 = ap[4 * ks], a1 = ap[4 * ks + 4], a2 = ap[4 * ks + 8], a3 = ap[4 * ks + 12];
    acc = mfma16(a0, b0, acc);
    acc = mfma16(a1, b1, acc);
    acc = mfma16(a2, b2, acc);
    acc = mfma16(a3, b3, acc);
  }
  for (; ks < m.ksn; ++ks) acc = mfma16(ap[4 * ks], bp[(size_t)ks * 64], acc);
  return acc;
}

template <typename T>
__global__ __launch_bounds__(64 * kLinW) void linear_forward_kernel(const LinDev<T> m, const T* __restrict__ states,
                                                                    const T* __restrict__ ctrls,
                                                                    T* __restrict__ out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xu = reinterpret_cast<T*>(smem_raw);
  constexpr int NTHR = 64 * kLinW;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nx = m.nx, nu = m.nu, xs = lin_xs(m.kp, (int)sizeof(T)), first = blockIdx.x * 16;
  for (int i = tid; i < 16 * xs; i += NTHR) {
    const int row = i / xs, col = i - row * xs, gr = first + row;
    T v = T(0);
    if (gr < n) {
      if (col < nx) v = states[(size_t)gr * nx + col];
      else if (col < nx + nu) v = ctrls[(size_t)gr * nu + (col - nx)];
    }
    xu[i] = v;
  }
  __syncthreads();
  for (int nt = w; nt < m.ntile; nt += kLinW) {
    const typename Acc<T>::type acc = lin_tile<T>(m, xu, xs, nt, lane);
    const int col = 16 * nt + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = first + acc_row<T>(lane >> 4, r);
      if (col < nx && gr < n) out[(size_t)gr * nx + col] = acc[r];
    }
  }
}

// jx[r] = A, ju[r] = B for every row (Model.pred_diff_batch of a linear model)
template <typename T>
__global__ void linear_jacobian_kernel(const LinDev<T> m, T* __restrict__ jx, T* __restrict__ ju, int n) {
  const int nx = m.nx, nu = m.nu, k = nx + nu;
  const size_t total = (size_t)n * nx * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / ((size_t)nx * k);
    const int e = (int)(i - r * nx * k), a = e / k, c = e - a * k;
    const T v = m.plain[e];
    if (c < nx) jx[(r * nx + a) * nx + c] = v;
    else ju[(r * nx + a) * nu + (c - nx)] = v;
  }
}

// MPPI rollout on a wide linear model.  LDS: [x | u] twice ([16][xs] each), then -- at the offsets
// plan_build computes (args.lds_cost, lds_aseq, lds_eps, lds_red) -- cost block + bounds, shifted
// sequence, the tile's clipped noise and the reduction scratch.
template <typename T>
__global__ __launch_bounds__(64 * kLinW) void linear_rollout_kernel(const MppiArgs<T> args, const LinDev<T> m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int M = 16, NTHR = 64 * kLinW, TPS = NTHR / M;      // 32 threads per sample, one wave half each
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nx = m.nx, nu = m.nu, no = args.obs_dim, xs = lin_xs(m.kp, (int)sizeof(T));
  const bool diag = args.cost_diag != 0, affine = args.cost_affine != 0;
  const int cost_stride = args.cost_stride;
  const int p = args.tile_prob[blockIdx.x];
  const MppiProblem<T> pr = args.probs[p];
  const int first = (blockIdx.x - pr.tile0) * M;
  const int H = pr.H, N = pr.N;
  T* xb[2] = {lds, lds + 16 * xs};
  T* aseq = lds + args.lds_aseq;
  T* cpar = lds + args.lds_cost;
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu; const T* goal = Fm + no * no;
  const T* lin = goal + no; const T* lint = lin + no;
  const T* blo = cpar + cost_stride; const T* bhi = blo + nu; const T* bsc = bhi + nu;

  for (int i = tid; i < 2 * 16 * xs; i += NTHR) lds[i] = T(0);
  for (int i = tid; i < cost_stride; i += NTHR) cpar[i] = args.costs_par[(size_t)pr.cost_idx * cost_stride + i];
  for (int i = tid; i < 3 * nu; i += NTHR) cpar[cost_stride + i] = args.bounds[i];
  for (int i = tid; i < H * nu; i += NTHR) {
    const int t = i / nu, j = i - t * nu;
    const int ts = (t + 1 < H) ? t + 1 : H - 1;                 // a[:-1] = a[1:]; a[-1] = a[-2]
    aseq[i] = args.act_in[pr.a_off + ts * nu + j];
  }
  const int ms = tid / TPS, r = tid % TPS;                      // sample in tile, helper index
  const int n = first + ms;
  const bool valid = n < N;
  const T* eps_row = args.eps + pr.eps_off + (size_t)(valid ? n : 0) * H * nu;
  T* epso = args.eps_out + pr.epso_off;
  __syncthreads();
  for (int i = tid; i < M * nx; i += NTHR) {
    const int row = i / nx, col = i - row * nx;
    xb[0][row * xs + col] = args.x0[p * nx + col];
  }
  T c_part = T(0), ca_part = T(0);
  // actions of step t into buffer `dst`: A = clip(eps + a), eps <- A - a, u = A * scale (mppi.py:134-139)
  auto actions = [&](int t, T* dst) {
    if (r < nu) {
      const int j = r;
      const T a = aseq[t * nu + j];
      T A = (valid ? eps_row[t * nu + j] : T(0)) + a;
      A = A < blo[j] ? blo[j] : A;              // by comparison: a NaN poisons the sample as in the reference
      A = A > bhi[j] ? bhi[j] : A;
      const T ec = A - a;
      if (valid && args.write_eps_out) epso[((size_t)t * N + n) * nu + j] = ec;
      if (args.lds_eps >= 0) lds[args.lds_eps + (t * M + ms) * nu + j] = ec;
      ca_part += A * ec;
      const T u = A * bsc[j];
      dst[ms * xs + nx + j] = u;
      if (diag) c_part += Rm[j * nu + j] * u * u;
    }
  };
  actions(0, xb[0]);
  __syncthreads();
  if (affine && r == 0) c_part += T(H) * lint[no];
  for (int t = 0; t < H; ++t) {
    T* cur = xb[t & 1];
    T* nxt = xb[(t + 1) & 1];
    // ---- stage cost of (x_t, u_t): TPS partials per sample, reduced once after the loop
    c_part += quad_rows<T>(Qm, cur + ms * xs, goal, no, r, TPS, diag);
    if (args.n_ind) c_part += indicator_rows<T>(args.ind_tab, args.n_ind, cur + ms * xs, 1, no, r, TPS);
    if (!diag) c_part += quad_rows<T>(Rm, cur + ms * xs + nx, nullptr, nu, r, TPS, false);
    if (affine) c_part += affine_rows<T>(lin, cur + ms * xs, goal, no, r, TPS, T(0));
    // ---- dynamics: x_{t+1} = M [x_t ; u_t] into the other buffer; the next step's actions beside it
    for (int nt = w; nt < m.ntile; nt += kLinW) {
      const typename Acc<T>::type acc = lin_tile<T>(m, cur, xs, nt, lane);
      const int col = 16 * nt + (lane & 15);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (col < nx) nxt[acc_row<T>(lane >> 4, rr) * xs + col] = acc[rr];
    }
    if (t + 1 < H) actions(t + 1, nxt);
    lds_barrier();
  }
  // ---- epilogue: terminal cost, reduce the TPS partials, write ---------------------------------
  const T* xe = xb[H & 1] + ms * xs;
  T term = quad_rows<T>(Fm, xe, goal, no, r, TPS, diag);
  if (affine) term += affine_rows<T>(lint, xe, goal, no, r, TPS, lint[no + 1]);
  T c = c_part + pr.lam_over_sigma * ca_part;
  if (args.term_mode == 1) c += term;
#pragma unroll
  for (int off = TPS / 2; off > 0; off >>= 1) {
    c += __shfl_xor(c, off);
    term += __shfl_xor(term, off);
  }
  if (r == 0 && valid) {
    args.costs[pr.cost_off + n] = c;
    if (n == N - 1) args.term_last[p] = term;
  }
  if (args.lds_eps < 0) return;
  // ---- fused softmin update, tile part (as mppi_rollout_kernel; mppi_combine_kernel finishes)
  T* cred = lds + args.lds_red;
  T* sred = cred + M;
  if (r == 0) cred[ms] = valid ? c : T(INFINITY);
  __syncthreads();
  T mw = cred[0];
  for (int i = 1; i < M; ++i) mw = cred[i] < mw ? cred[i] : mw;
  const bool dead_tile = !(mw < T(INFINITY));
  if (tid < M)
    sred[tid] = (first + tid < N && !dead_tile) ? exp(pr.neg_inv_lambda * (cred[tid] - mw)) : T(0);
  __syncthreads();
  const T* el = lds + args.lds_eps;
  T* tp = args.tile_part + (size_t)blockIdx.x * args.hnu_stride;
  for (int e = tid; e < H * nu; e += NTHR) {
    const int t = e / nu, j = e - t * nu;
    T s = T(0);
    for (int i = 0; i < M; ++i) s += sred[i] * el[(t * M + i) * nu + j];
    tp[e] = s;
  }
  if (tid == 0) {
    T ss = T(0);
    for (int i = 0; i < M; ++i) ss += sred[i];
    args.tile_stat[2 * blockIdx.x] = mw;
    args.tile_stat[2 * blockIdx.x + 1] = ss;
  }
}

}  // namespace ampc
