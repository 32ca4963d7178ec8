// ilqr_kernels.hpp -- iLQR on gfx950: one workgroup per problem, batched over problems.
//
// What one solve computes (reference: autompc/control/ilqr.py:100-265):
//   rollout the guess, obj = dt*sum(stage) + terminal
//   repeat <= max_iter:
//     backward Riccati sweep t = H-1..0:  Qt = Ct + J'VJ, qt = ct + J'v,
//         K = -solve(Quu, Qux), k = -solve(Quu, qu)   (unregularised, partial-pivot LU)
//         V <- Qxx + Qxu K + K'Qux + K'Quu K ;  v <- qx + Qxu k + K'(qu + Quu k)
//     forward: all ls_n step sizes alpha_j = discount^j rolled out together
//         u = alpha k_i + ubar_i + K_i (x - xbar_i), clipped when bounded
//     accept the first alpha with (obj-new)/(-(alpha lin + alpha^2 quad/2)) > 0.3, else best-so-far;
//     refresh Jacobians of the accepted trajectory; stop on ||du|| < 1e-3 or line-search failure
//
// Decomposition.  One iteration is four launches.  The backward sweep: ilqr_riccati_mfma_kernel
// (model states <= 32 and the usual control dimensions: MFMA products from LDS, per-lane LDL' / LU
// solve) or ilqr_riccati_kernel (everything else).  The line search + acceptance: ilqr_ls4_kernel
// (ilqr_ls4.hpp: f64 MLP models, candidates four at a time on 4x4x4 MFMA tiles, weights resident
// on chip) or ilqr_iter_kernel (this file: all step sizes as the rows of one 16-row tile of
// mlp_tile.hpp; f32, SINDy, wide states).  One problem per workgroup, entirely from LDS.  Then
// mlp_forward_kernel<DERIV> + mlp_jacobian_kernel over all (problem, t) rows at once -- the
// Jacobian refresh is the only phase with H-fold parallelism, so it gets the whole chip.
// Problems that have converged / failed keep their workgroup slot but exit immediately.
//
// Kept reference quirks: terminal gradient / Hessian ignore the goal (cost.py:195,208-211); when
// the line search neither succeeds nor trips the failure test the LAST candidate evaluated
// becomes the nominal trajectory and the Jacobians stay stale (ilqr.py:208-255).
#pragma once
#include "mlp_tile.hpp"
#include "sindy_kernels.hpp"
#include "linear_kernels.hpp"

namespace ampc {

// (AMPC_IMARK / AMPC_IPROBE_STEP: phase marks of the timing experiments, nothing in the product -- probe.hpp)

constexpr int kRicThreads = 512;  // workgroup of the backward-sweep kernel
constexpr int kIlqrMaxLs = 16;   // line-search candidates live in the 16 rows of one MFMA tile
constexpr int kRicStride = 4 + kIlqrMaxLs;   // per problem: sweep sums [4], candidate objectives [16]

template <typename T> struct IlqrArgs {
  MlpDev<T> mlp;
  SindyDev<T> sindy;             // used instead of the MLP tile when the kernel is built with DYN = 1
  int lds_xn;                    // SINDy: [16][nx] next-state scratch (elements)
  TileLds lds;
  int lds_work;                  // start of the Riccati / line-search scratch (elements)
  int H, obs_dim, cost_stride, bounded, ls_n, mode;   // mode 0: initial rollout, 1: iteration
  int cost_diag;                 // 1: Q, R, F of every cost block are diagonal -> O(n) objective
  int cost_affine;               // 1: some cost block has an affine part (mlp_tile.hpp: cost_block_stride)
  int max_iter;                  // > 0: a problem is retired once it has done this many iterations (the
                                 //      queue's per-problem cap; 0: the host loop counts)
  int ls_split;                  // four-row line search in two launches (ilqr_ls4.hpp): 0 no, 1 first launch
                                 // (pass 0 for every problem), 2 second (the other passes side by side, for
  int* ls_pass;                  // the problems the first left undecided: ls_pass[p] = 1)
  int* ls_need;                  // [B] four-row passes the slot's LAST line search needed (1..): what the host
                                 // picks the line-search kernel of the next launches from (ilqr_lsw.hpp)
  const long long* model_delta;  // controller models of the plan's shape (ampc_ilqr_plan_set_models; byte offsets of
  const int* slot_model;         // their buffers, mlp_tile.hpp) and the entry each slot's problem uses, or nullptr
  const int* slot_h;             // [B] per-slot horizon <= H (ampc_ilqr_*_var: problems of different horizons share
                                 // a plan; every array keeps the stride of H), nullptr: H for every problem
  int* slot_mode;                // queue mode (ampc_ilqr_solve_queue): per slot 0 = roll out the guess of the
                                 // problem just loaded, 1 = iterate; nullptr: `mode` for every problem
  const int* slot_of;            // queue with more slots than CUs: workgroup -> slot, the slots with work FIRST
                                 // (ilqr_compact_kernel, rebuilt every iteration); nullptr: workgroup b = slot b
  int term_goal;                 // 0: terminal gradient (F+F')x_N as the reference computes it
                                 //    (cost.py:195, goal ignored); 1: (F+F')(x_N - goal)
  T dt, u_threshold, ls_cost_threshold;
  T alphas[kIlqrMaxLs];          // step sizes discount**j, computed on the host like the reference
  const T* costs_par;            // [n_costs][cost_stride]: Q R F goal lin lint c0 c1
  const int* cost_idx;           // [B]
  const T* ubounds;              // lo[nu] hi[nu]
  T* states;                     // [B][H+1][nx]  nominal trajectory
  T* ctrls;                      // [B][H][nu]
  const T* jx;                   // [B*H][nx][nx]
  const T* ju;                   // [B*H][nx][nu]
  T* Ks;                         // [B][H][nu][nx]
  T* ks;                         // [B][H][nu]
  T* ls_states;                  // [B][ls_n][H+1][nx]
  T* ls_ctrls;                   // [B][ls_n][H][nu]
  T* obj;                        // [B]
  int* converged;                // [B]
  int* active;                   // [B]
  int* iters;                    // [B]
  int* status;                   // [B] 0 ok, 1 singular Quu, 2 no line-search candidate
  int* refresh;                  // [B] Jacobians must be recomputed for this problem
  int* ls_rows;                  // [B] candidate rows rolled out by the line search since the solve began
  T* ric;                        // [B][kRicStride] sweep -> line search: lin, quad, |k|, singular flag;
                                 //    then the candidates' objectives when their passes run in parallel
  int* ls_count;                 // [B] passes of the running line search that have finished (parallel passes)
  int par_passes;                // line-search passes of one problem on separate workgroups (grid.y)
  // wide linear models (65..128 states; ilqr_wide.hpp, ilqr_iter_kernel<.., DYN = 2>): the model and the
  // sweep's per-problem VJ scratch [B][nxp][ldj] -- behind everything the MLP kernels read
  LinDev<T> lin;
  T* vj;
};

// Scratch map inside the work region (offsets in elements of T), nx/nu/n known at run time.
struct IlqrWork {
  int V, v, J, VJ, Qt, qt, K, k, Wk, wq, lu, rhs, xbar, ubar, cpar, lo, hi, scal, lsobj, piv, total;
};
// (compact: the map of a kernel that only runs the line search -- no Riccati matrices; wide linear models)
__host__ __device__ inline IlqrWork make_ilqr_work(int nx, int nu, int cost_stride, bool compact = false) {
  const int n = nx + nu;
  IlqrWork w;
  int o = 0;
  w.V = o; o += compact ? 0 : nx * nx;
  w.v = o; o += compact ? 0 : nx;
  w.J = o; o += compact ? 0 : nx * n;
  w.VJ = o; o += compact ? 0 : nx * n;
  w.Qt = o; o += compact ? 0 : n * n;
  w.qt = o; o += compact ? 0 : n;
  w.K = o; o += nu * nx;
  w.k = o; o += nu;
  w.Wk = o; o += compact ? 0 : nu * nx;
  w.wq = o; o += nu;
  w.lu = o; o += nu * nu;
  w.rhs = o; o += compact ? 0 : nu * (nx + 1);
  w.xbar = o; o += nx;
  w.ubar = o; o += nu;
  w.cpar = o; o += cost_stride;
  w.lo = o; o += nu;
  w.hi = o; o += nu;
  w.scal = o; o += 16;
  w.lsobj = o; o += kIlqrMaxLs;
  w.piv = o; o += (nu + 1) / 2 + 8;   // int pivots stored in this slot
  w.total = o;
  return w;
}

template <typename T> __device__ __forceinline__ T block_sum_any(T v, T* scratch, int nwaves) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  T o = T(0);
  for (int w = 0; w < nwaves; ++w) o += scratch[w];
  __syncthreads();
  return o;
}

// Value of `v` in lane `lane` (wave-uniform index), broadcast to the whole wave.
__device__ __forceinline__ float readlane_t(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double readlane_t(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Quu [K | k] = -[Qux | qu] for one Riccati step, by ONE wave, entirely in registers.
// Gauss-Jordan with partial pivoting (the pivot sequence of numpy.linalg.solve / LAPACK gesv) on
// the augmented matrix [Quu | Qux | qu]: lane j owns column j (nu + nx + 1 <= 49 columns) and
// keeps its nu entries in col[].  Column c's entries reach the other lanes by readlane, rows are
// swapped by uniform-branch register moves; there is no LDS round trip inside the elimination.
// NU > 0: the control dimension as a compile-time constant (fully unrolled, static register
// indices); NU == 0: any nu <= kMaxNu at run time.
// Reads Qt [n][n], qt [n]; writes Km [nu][nx], kv [nu].  Returns 1 when a pivot is exactly zero.
template <typename T, int NU>
__device__ __attribute__((noinline)) int quu_solve(const T* __restrict__ Qt, const T* __restrict__ qt,
                                         T* __restrict__ Km, T* __restrict__ kv, int lane, int nx,
                                         int nu_rt) {
  constexpr int NB = NU > 0 ? NU : kMaxNu;
  const int nu = NU > 0 ? NU : nu_rt;
  const int n = nx + nu, nc = nu + nx + 1;
  T col[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    col[i] = T(0);
    if (i < nu && lane < nc)
      col[i] = lane < nu ? Qt[(nx + i) * n + nx + lane]
                         : (lane < nu + nx ? Qt[(nx + i) * n + (lane - nu)] : qt[nx + i]);
  }
  int sing = 0;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (c >= nu || sing) break;
    // pivot row: first maximum of |Aug[i][c]|, i >= c (lane c holds that column)
    T best = fabs(col[c]);
    int pr = c;
#pragma unroll
    for (int i = c + 1; i < NB; ++i) {
      if (i >= nu) break;
      const T a = fabs(col[i]);
      if (a > best) { best = a; pr = i; }
    }
    pr = __builtin_amdgcn_readlane(pr, c);
#pragma unroll
    for (int i = c + 1; i < NB; ++i)
      if (pr == i) { const T tmp = col[c]; col[c] = col[i]; col[i] = tmp; }
    const T d = readlane_t(col[c], c);
    if (d == T(0)) { sing = 1; break; }
    const T rd = T(1) / d;      // LAPACK's getf2 scales by the reciprocal pivot as well
    const T rowc = col[c];
#pragma unroll
    for (int i = 0; i < NB; ++i) {                 // eliminate column c from every other row
      if (i >= nu) break;
      if (i == c) continue;
      const T f = readlane_t(col[i], c) * rd;
      if (lane > c) col[i] -= f * rowc;
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (i >= nu) break;
    const T val = -col[i] / readlane_t(col[i], i);
    if (lane >= nu && lane < nu + nx) Km[i * nx + (lane - nu)] = val;
    else if (lane == nu + nx) kv[i] = val;
  }
  return sing;
}

// Backward Riccati sweep of one problem per workgroup (ilqr.py:159-187).  A kernel of its own:
// it needs none of the MLP tile's registers or LDS, so it is compiled once per precision, keeps
// the Quu solve in registers without spilling, and leaves K_t, k_t (global) and the expected-
// reduction sums `ric[p] = {lin, quad, |k|, singular}` for the line-search kernel.
// The slot a per-slot workgroup works on.  With more slots than CUs (a finite batch admitted at once: no drain)
// the slots that still have work are scattered over the grid: their workgroups -- each the size of a CU in the
// line search -- would land on the XCDs unevenly (workgroup b goes to XCD b % 8) and queue behind each other
// while other CUs idle, and every launch would cost what the full grid costs.  args.slot_of lists the slots with
// work first, in slot order, then the idle ones (whose workgroups exit at once, as before).
template <typename T> __device__ __forceinline__ int ilqr_slot(const IlqrArgs<T>& args) {
  const int b = blockIdx.x;
  return args.slot_of ? __builtin_amdgcn_readfirstlane(args.slot_of[b]) : b;
}

template <typename T, bool WIDE = false, typename SH = DynShape>   // WIDE: model states above 32 (longer register staging)
__global__ __launch_bounds__(kRicThreads) void ilqr_riccati_kernel(const IlqrArgs<T> args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Wr = reinterpret_cast<T*>(smem_raw);
  constexpr int NTHR = kRicThreads;
  const MlpDev<T> mlp = SH::template fold<T>(args.mlp);
  const int tid = threadIdx.x, p = ilqr_slot(args);
  const int nx = mlp.nx, nu = mlp.nu, n = nx + nu, no = SH::kStatic ? SH::no : args.obs_dim;
  const int HS = args.H, H = args.slot_h ? args.slot_h[p] : HS;      // array stride, this slot's horizon
  if (args.active[p] == 0) return;
  const int cost_stride = SH::kStatic ? cost_block_stride(SH::no, SH::nu) : args.cost_stride;
  const IlqrWork wk = make_ilqr_work(nx, nu, cost_stride);
  T* V = Wr + wk.V; T* v = Wr + wk.v; T* Jm = Wr + wk.J; T* VJ = Wr + wk.VJ;
  T* Qt = Wr + wk.Qt; T* qt = Wr + wk.qt; T* Km = Wr + wk.K; T* kv = Wr + wk.k;
  T* Wk = Wr + wk.Wk; T* wq = Wr + wk.wq;
  T* xbar = Wr + wk.xbar; T* ubar = Wr + wk.ubar; T* cpar = Wr + wk.cpar; T* scal = Wr + wk.scal;
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;
  const T* clin = goal + no; const T* clint = clin + no;     // affine part of the stage / terminal cost
  for (int i = tid; i < cost_stride; i += NTHR)
    cpar[i] = args.costs_par[(size_t)args.cost_idx[p] * cost_stride + i];
  __syncthreads();
  const T* st = args.states + (size_t)p * (HS + 1) * nx;
  const T* ct = args.ctrls + (size_t)p * HS * nu;
  T* Kg = args.Ks + (size_t)p * HS * nu * nx;
  T* kg = args.ks + (size_t)p * HS * nu;
  T lin = T(0), quad = T(0), ksn2 = T(0);   // meaningful in thread 0 only
  const T dt = args.dt;
  AMPC_IMARK_ALWAYS(30);
  for (int idx = tid; idx < nx * nx; idx += NTHR) {
    const int a = idx / nx, b = idx - a * nx;
    V[idx] = (a < no && b < no) ? Fm[a * no + b] + Fm[b * no + a] : T(0);
  }
  for (int a = tid; a < nx; a += NTHR) {
    T s = T(0);
    if (a < no)
      for (int b = 0; b < no; ++b)
        s += (Fm[a * no + b] + Fm[b * no + a]) * (st[(size_t)H * nx + b] - (args.term_goal ? goal[b] : T(0)));
    // term_goal == 0: no goal subtraction -- the reference's terminal gradient quirk, term by term in
    // a sum (sum_cost.py:49-54 over cost.py:195), so the affine part only enters with term_goal
    if (a < no && args.term_goal) s += clint[a];
    v[a] = s;
  }
  if (tid == 0) scal[8] = T(0);
  __syncthreads();
  // J_t = [jx | ju], xbar_t, ubar_t are staged into LDS one step ahead, through registers: the
  // global loads for step t-1 are issued at the top of step t and land in LDS at its end, so
  // their latency is covered by the step's arithmetic (barriers in the loop are LDS-only).
  constexpr int JR = ((WIDE ? 64 * 80 : 32 * 48) + NTHR - 1) / NTHR;   // predicated on nx * n
  T jreg[JR];
  T xreg = T(0), ureg = T(0);
  auto fetch_step = [&](int t) {
    const T* jxp = args.jx + ((size_t)p * HS + t) * nx * nx;
    const T* jup = args.ju + ((size_t)p * HS + t) * nx * nu;
#pragma unroll
    for (int k = 0; k < JR; ++k) {
      const int idx = tid + k * NTHR;
      if (idx < nx * n) {
        const int a = idx / n, c = idx - a * n;
        jreg[k] = c < nx ? jxp[a * nx + c] : jup[a * nu + (c - nx)];
      }
    }
    if (tid < nx) xreg = st[(size_t)t * nx + tid];
    if (tid < nu) ureg = ct[(size_t)t * nu + tid];
  };
  auto commit_step = [&]() {
#pragma unroll
    for (int k = 0; k < JR; ++k) {
      const int idx = tid + k * NTHR;
      if (idx < nx * n) Jm[idx] = jreg[k];
    }
    if (tid < nx) xbar[tid] = xreg;
    if (tid < nu) ubar[tid] = ureg;
  };
  fetch_step(H - 1);
  commit_step();
  __syncthreads();
  for (int t = H - 1; t >= 0; --t) {
    AMPC_IPROBE_STEP(t == H / 2);
    AMPC_IMARK(20);
    if (t > 0) fetch_step(t - 1);
    for (int idx = tid; idx < nx * n; idx += NTHR) {       // VJ = V J
      const int a = idx / n, c = idx - a * n;
      T s = T(0);
#pragma unroll 8
      for (int b = 0; b < nx; ++b) s += V[a * nx + b] * Jm[b * n + c];
      VJ[idx] = s;
    }
    lds_barrier();
    AMPC_IMARK(21);
    // Qt = Ct + J' V J is symmetric (V is, up to rounding): only the upper triangle is computed
    // and mirrored.  Rows r and n-1-r are folded into one row of n+1 tasks, so the triangle is
    // ceil(n/2) x (n+1) tasks -- one round for n <= 30 (for odd n the middle row is done twice,
    // with identical results).
    const int fold_rows = (n + 1) / 2;
    for (int idx = tid; idx < fold_rows * (n + 1); idx += NTHR) {
      const int r = idx / (n + 1), j = idx - r * (n + 1);
      const int c = (j - 1 >= r) ? r : n - 1 - r;
      const int d = (j - 1 >= r) ? j - 1 : n - 1 - j;
      T s = T(0);
#pragma unroll 8
      for (int a = 0; a < nx; ++a) s += Jm[a * n + c] * VJ[a * n + d];
      T cc = T(0);
      if (c < no && d < no) cc = (Qm[c * no + d] + Qm[d * no + c]) * dt;
      else if (c >= nx && d >= nx) cc = (Rm[(c - nx) * nu + (d - nx)] + Rm[(d - nx) * nu + (c - nx)]) * dt;
      Qt[c * n + d] = cc + s;
      Qt[d * n + c] = cc + s;
    }
    for (int c = NTHR - 1 - tid; c < n; c += NTHR) {       // qt = ct + J' v  (threads from the far end)
      T s = T(0);
#pragma unroll 8
      for (int a = 0; a < nx; ++a) s += Jm[a * n + c] * v[a];
      T cc = T(0);
      if (c < no) {
        for (int b = 0; b < no; ++b) cc += (Qm[c * no + b] + Qm[b * no + c]) * (xbar[b] - goal[b]);
        cc += clin[c];
      } else if (c >= nx) {
        const int cj = c - nx;
        for (int j = 0; j < nu; ++j) cc += (Rm[cj * nu + j] + Rm[j * nu + cj]) * ubar[j];
      }
      qt[c] = cc * dt + s;
    }
    lds_barrier();
    AMPC_IMARK(22);
    // ---- Quu [K | k] = -[Qux | qu]: Gauss-Jordan with partial pivoting (the pivot sequence of
    // numpy.linalg.solve / LAPACK gesv) on the augmented matrix [Quu | Qux | qu], by wave 0,
    // entirely in registers: lane j owns column j (nc = nu + nx + 1 <= 49 columns), its nu
    // entries are col[0..nu).  Column c's entries reach the other lanes by readlane; rows are
    // swapped by uniform-branch register moves.  No LDS round trips inside the elimination.
    if (tid < 64) {
      int sing;
      switch (nu) {     // the usual control dimensions get fully unrolled solvers
        case 1: sing = quu_solve<T, 1>(Qt, qt, Km, kv, tid, nx, nu); break;
        case 2: sing = quu_solve<T, 2>(Qt, qt, Km, kv, tid, nx, nu); break;
        case 3: sing = quu_solve<T, 3>(Qt, qt, Km, kv, tid, nx, nu); break;
        case 4: sing = quu_solve<T, 4>(Qt, qt, Km, kv, tid, nx, nu); break;
        case 6: sing = quu_solve<T, 6>(Qt, qt, Km, kv, tid, nx, nu); break;
        case 8: sing = quu_solve<T, 8>(Qt, qt, Km, kv, tid, nx, nu); break;
        default: sing = quu_solve<T, 0>(Qt, qt, Km, kv, tid, nx, nu); break;
      }
      if (sing && tid == 0) { args.status[p] = 1; scal[8] = T(1); }
    }
    AMPC_IMARK(23);
    lds_barrier();
    AMPC_IMARK(24);
    // three independent pieces on disjoint waves: sums on wave 0, wq on wave 1, Wk on waves 2..
    for (int idx = tid - 128; idx < nu * nx; idx += NTHR - 128) {   // Wk = Quu K ; store K
      if (idx < 0) break;
      const int i = idx / nx, b = idx - i * nx;
      T s = T(0);
      for (int j = 0; j < nu; ++j) s += Qt[(nx + i) * n + nx + j] * Km[j * nx + b];
      Wk[idx] = s;
      Kg[(size_t)t * nu * nx + idx] = Km[idx];
    }
    for (int i = tid - 64; i >= 0 && i < nu; i += NTHR) {  // wq = qu + Quu k ; store k
      T s = qt[nx + i];
      for (int j = 0; j < nu; ++j) s += Qt[(nx + i) * n + nx + j] * kv[j];
      wq[i] = s;
      kg[(size_t)t * nu + i] = kv[i];
    }
    if (tid < 64) {                                        // lin += qu.k ; quad += k'Quu k ; |k|^2
      T l = T(0), qd = T(0), k2 = T(0);
      if (tid < nu) {
        const T ki = kv[tid];
        T s = T(0);
        for (int j = 0; j < nu; ++j) s += Qt[(nx + tid) * n + nx + j] * kv[j];
        l = qt[nx + tid] * ki;
        qd = ki * s;
        k2 = ki * ki;
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {              // nu <= 16
        l += __shfl_xor(l, off);
        qd += __shfl_xor(qd, off);
        k2 += __shfl_xor(k2, off);
      }
      if (tid == 0) { lin += l; quad += qd; ksn2 += k2; }
    }
    lds_barrier();
    AMPC_IMARK(25);
    for (int idx = tid; idx < nx * nx; idx += NTHR) {      // V <- Qxx + Qxu K + K'Qux + K'Quu K
      const int a = idx / nx, b = idx - a * nx;
      T s = Qt[a * n + b];
#pragma unroll 4
      for (int j = 0; j < nu; ++j)
        s += Qt[a * n + nx + j] * Km[j * nx + b] + Km[j * nx + a] * Qt[(nx + j) * n + b] +
             Km[j * nx + a] * Wk[j * nx + b];
      V[idx] = s;
    }
    for (int a = NTHR - 1 - tid; a < nx; a += NTHR) {      // v <- qx + Qxu k + K'(qu + Quu k)
      T s = qt[a];
      for (int j = 0; j < nu; ++j) s += Qt[a * n + nx + j] * kv[j] + Km[j * nx + a] * wq[j];
      v[a] = s;
    }
    if (t > 0) commit_step();                              // J, xbar, ubar are not read in this phase
    lds_barrier();
    AMPC_IMARK(26);
  }
  AMPC_IMARK_ALWAYS(33);
  if (tid == 0) {
    T* out = args.ric + (size_t)p * kRicStride;
    out[0] = lin; out[1] = quad; out[2] = sqrt(ksn2); out[3] = scal[8];
    if (scal[8] != T(0)) {        // singular Quu: the reference raises LinAlgError here
      args.active[p] = 0; args.refresh[p] = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward sweep, latency-optimised variant (model states <= 32, nu in {1, 2, 3, 4, 6, 8}).
//
// The sweep is a chain of H dependent steps, each a chain of three small matrix products and one
// nu x nu solve: what bounds it is the latency of that chain, not arithmetic.  Here the products
// run on MFMA 16x16x4 tiles straight from LDS (one ds_read per operand per MFMA instead of two per
// scalar FMA), and the solve is done by every lane on a private register copy of Quu with its own
// right-hand-side column, so that the factorisation has no cross-lane traffic at all.
//
//   A  VJ  = V J                                   (nx x n;  column n of VJ holds v)
//   B  Qt  = Ct + J' [VJ | v]                      (n x (n+1); column n is qt)
//   C  LU with partial pivoting of Quu in every lane (the pivot sequence of LAPACK gesv);
//      lane b solves Quu x = Qux[:, b], lane nx solves Quu x = qu;  K = -x, k = -x;
//      Z = Qux + Quu K, z = qu + Quu k                      -> SB = [K k ; Z z]
//   D  V   = Qxx + [Qxu | K'] [K ; Z],  v = qx + [Qxu | K'] [k ; z]
//
// (D is the reference's Qxx + Qxu K + K'Qux + K'Quu K with the last two terms factored.)
// LDS matrices are zero padded to MFMA tile multiples once; only valid entries are ever rewritten.
struct RicLds {
  int V, J, VJ, Qt, SB, CQ, CR, cq, xbar, ubar, goal, Fm, scal, total;
  int ldV, ldJ, ldQ, ldB, nxp4, nxp16, nr16, np16, nb16, nu4;
};
__host__ __device__ constexpr int ric_ld(int cols16) { return cols16 % 32 == 0 ? cols16 + 16 : cols16; }
__host__ __device__ constexpr RicLds make_ric_lds(int nx, int nu, int no) {
  const int n = nx + nu;
  RicLds r{};
  r.nxp4 = (nx + 3) / 4 * 4; r.nxp16 = (nx + 15) / 16 * 16; r.nr16 = (n + 15) / 16 * 16;
  r.np16 = (n + 1 + 15) / 16 * 16; r.nb16 = (nx + 1 + 15) / 16 * 16; r.nu4 = (nu + 3) / 4 * 4;
  r.ldV = r.nxp4 | 1; r.ldJ = ric_ld(r.np16); r.ldQ = r.np16 + 1; r.ldB = ric_ld(r.nb16);
  int o = 0;
  r.V = o; o += r.nxp16 * r.ldV + 4;
  r.J = o; o += r.nxp4 * r.ldJ;
  r.VJ = o; o += r.nxp4 * r.ldJ;
  r.Qt = o; o += r.nr16 * r.ldQ + 8;
  r.SB = o; o += 2 * r.nu4 * r.ldB;
  r.CQ = o; o += no * no;
  r.CR = o; o += nu * nu;
  r.Fm = o; o += no * no;
  r.cq = o; o += 2 * n;
  r.xbar = o; o += nx;
  r.ubar = o; o += nu;
  r.goal = o; o += 3 * no;           // goal | lin | lint
  r.scal = o; o += 4;
  r.total = (o + 3) / 4 * 4;
  return r;
}

// Quu x = rhs on a private register copy, for symmetric positive definite Quu (the usual case):
// LDL' on the lower triangle, no pivot search.  Returns 0 when some pivot is not safely positive;
// the caller then repeats the solve with lu_solve_lane.  x: rhs in, solution out.
template <typename T, int NU>
__device__ __forceinline__ int ldl_solve_lane(const T (&A0)[NU][NU], T (&x)[NU]) {
  T L[NU][NU], rd[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i][j] = A0[i][j];
  T dmax = T(0), dmin = L[0][0];
#pragma unroll
  for (int i = 0; i < NU; ++i) dmax = fmax(dmax, fabs(A0[i][i]));
#pragma unroll
  for (int c = 0; c < NU; ++c) {
    const T d = L[c][c];
    dmin = fmin(dmin, d);
    rd[c] = fast_rcp(d);
    T lc[NU];                                       // column c of the unit lower factor
#pragma unroll
    for (int i = c + 1; i < NU; ++i) lc[i] = L[i][c] * rd[c];
#pragma unroll
    for (int i = c + 1; i < NU; ++i) {
#pragma unroll
      for (int j = c + 1; j <= i; ++j) L[i][j] -= lc[i] * L[j][c];   // L[j][c] is still d_c * l_jc
      x[i] -= lc[i] * x[c];
    }
#pragma unroll
    for (int i = c + 1; i < NU; ++i) L[i][c] = lc[i];
  }
#pragma unroll
  for (int c = NU - 1; c >= 0; --c) {
    T s = x[c] * rd[c];
#pragma unroll
    for (int j = c + 1; j < NU; ++j) s -= L[j][c] * x[j];
    x[c] = s;
  }
  return dmin > dmax * T(sizeof(T) == 8 ? 1e-10 : 1e-4);
}

// General case: LU with partial pivoting (the pivot sequence of numpy.linalg.solve / LAPACK gesv).
// Every lane holds the same matrix, so the pivot row is wave-uniform.  Returns 1 on a zero pivot.
template <typename T, int NU>
__device__ __forceinline__ int lu_solve_lane(const T (&A0)[NU][NU], T (&x)[NU]) {
  T A[NU][NU], rd[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i)
#pragma unroll
    for (int j = 0; j < NU; ++j) A[i][j] = A0[i][j];
  int sing = 0;
#pragma unroll
  for (int c = 0; c < NU; ++c) {
    T best = fabs(A[c][c]);
    int pr = c;
#pragma unroll
    for (int i = c + 1; i < NU; ++i) {
      const T a = fabs(A[i][c]);
      if (a > best) { best = a; pr = i; }
    }
    pr = __builtin_amdgcn_readfirstlane(pr);
#pragma unroll
    for (int i = c + 1; i < NU; ++i)
      if (pr == i) {
#pragma unroll
        for (int j = c; j < NU; ++j) { const T tmp = A[c][j]; A[c][j] = A[i][j]; A[i][j] = tmp; }
        const T tmp = x[c]; x[c] = x[i]; x[i] = tmp;
      }
    const T d = A[c][c];
    if (d == T(0)) sing = 1;
    rd[c] = T(1) / d;
#pragma unroll
    for (int i = c + 1; i < NU; ++i) {
      const T f = A[i][c] * rd[c];
#pragma unroll
      for (int j = c + 1; j < NU; ++j) A[i][j] -= f * A[c][j];
      x[i] -= f * x[c];
    }
  }
#pragma unroll
  for (int c = NU - 1; c >= 0; --c) {
    T s = x[c];
#pragma unroll
    for (int j = c + 1; j < NU; ++j) s -= A[c][j] * x[j];
    x[c] = s * rd[c];
  }
  return sing;
}

template <typename T, int NU, typename SH = DynShape>
__global__ __launch_bounds__(kRicThreads) AMPC_PROBE_RIC_OCC void ilqr_riccati_mfma_kernel(const IlqrArgs<T> args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Wr = reinterpret_cast<T*>(smem_raw);
  using acc_t = typename Acc<T>::type;
  constexpr int NTHR = kRicThreads, NW = NTHR / 64, SIDE0 = NTHR / 2;   // side work: threads >= SIDE0
  constexpr int TPW = 2;                                // tiles per wave: n + 1 <= 48 -> at most 9 tiles
  constexpr int KX = 8, KD = 2 * ((NU + 3) / 4);        // k-steps over the state / the stacked controls
  constexpr int nu = NU, nu4 = (NU + 3) / 4 * 4;
  const MlpDev<T> mlp = SH::template fold<T>(args.mlp);
  const int tid = threadIdx.x, p = ilqr_slot(args), lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int nx = mlp.nx, n = nx + nu, no = SH::kStatic ? SH::no : args.obs_dim;
  const int HS = args.H, H = args.slot_h ? args.slot_h[p] : HS;      // array stride, this slot's horizon
  if (args.active[p] == 0) return;
  const int cost_stride = SH::kStatic ? cost_block_stride(SH::no, SH::nu) : args.cost_stride;
  const RicLds R = make_ric_lds(nx, nu, no);
  T* V = Wr + R.V; T* Jm = Wr + R.J; T* VJ = Wr + R.VJ; T* Qt = Wr + R.Qt; T* SB = Wr + R.SB;
  T* CQ = Wr + R.CQ; T* CR = Wr + R.CR; T* Fs = Wr + R.Fm; T* cq = Wr + R.cq;
  T* xbar = Wr + R.xbar; T* ubar = Wr + R.ubar; T* goal = Wr + R.goal; T* scal = Wr + R.scal;
  const int ldV = R.ldV, ldJ = R.ldJ, ldQ = R.ldQ, ldB = R.ldB;
  const T* cpar = args.costs_par + (size_t)args.cost_idx[p] * cost_stride;   // Q R F goal
  const T* st = args.states + (size_t)p * (HS + 1) * nx;
  const T* ct = args.ctrls + (size_t)p * HS * nu;
  T* Kg = args.Ks + (size_t)p * HS * nu * nx;
  T* kg = args.ks + (size_t)p * HS * nu;
  const T dt = args.dt;
  for (int i = tid; i < R.total; i += NTHR) Wr[i] = T(0);
  __syncthreads();
  for (int i = tid; i < no * no; i += NTHR) {           // symmetrised cost Hessians (cost.py:181-211)
    const int a = i / no, b = i - a * no;
    CQ[i] = cpar[a * no + b] + cpar[b * no + a];
    Fs[i] = cpar[no * no + nu * nu + a * no + b] + cpar[no * no + nu * nu + b * no + a];
  }
  for (int i = tid; i < nu * nu; i += NTHR) {
    const int a = i / nu, b = i - a * nu;
    CR[i] = cpar[no * no + a * nu + b] + cpar[no * no + b * nu + a];
  }
  for (int i = tid; i < 3 * no; i += NTHR) goal[i] = cpar[2 * no * no + nu * nu + i];   // goal | lin | lint
  const T* clin = goal + no; const T* clint = clin + no;
  __syncthreads();
  for (int i = tid; i < no * no; i += NTHR) {           // V_H = F + F' on the observed block
    const int a = i / no, b = i - a * no;
    V[a * ldV + b] = Fs[i];
  }
  for (int a = tid; a < no; a += NTHR) {                // v_H (term_goal == 0: the reference's quirk)
    T s = T(0);
    for (int b = 0; b < no; ++b) s += Fs[a * no + b] * (st[(size_t)H * nx + b] - (args.term_goal ? goal[b] : T(0)));
    if (args.term_goal) s += clint[a];
    VJ[a * ldJ + n] = s;
  }
  // J_t = [jx | ju], xbar_t, ubar_t: loaded one step ahead into registers by the side threads
  constexpr int SIDE = NTHR - SIDE0;
  constexpr int JR = (32 * 40 + SIDE - 1) / SIDE;
  const int sid = tid - SIDE0;
  T jreg[JR];
  T xreg = T(0), ureg = T(0);
  auto fetch_step = [&](int t) {
    const T* jxp = args.jx + ((size_t)p * HS + t) * nx * nx;
    const T* jup = args.ju + ((size_t)p * HS + t) * nx * nu;
#pragma unroll
    for (int k = 0; k < JR; ++k) {
      const int idx = sid + k * SIDE;
      if (idx < nx * n) {
        const int a = idx / n, c = idx - a * n;
        jreg[k] = c < nx ? jxp[a * nx + c] : jup[a * nu + (c - nx)];
      }
    }
    if (sid < nx) xreg = st[(size_t)t * nx + sid];
    if (sid < nu) ureg = ct[(size_t)t * nu + sid];
  };
  auto commit_step = [&]() {
#pragma unroll
    for (int k = 0; k < JR; ++k) {
      const int idx = sid + k * SIDE;
      if (idx < nx * n) {
        const int a = idx / n, c = idx - a * n;
        Jm[a * ldJ + c] = jreg[k];
      }
    }
    if (sid < nx) xbar[sid] = xreg;
    if (sid < nu) ubar[sid] = ureg;
  };
  // gradient of the stage cost about (xbar, ubar), times dt: eight threads per entry, into cq[buf]
  auto stage_gradient = [&](int buf) {
    for (int base = 0; base < 8 * n; base += SIDE) {
      const int idx = base + sid, c = idx >> 3, part = idx & 7;
      T cc = T(0);
      if (c < no) {
        for (int b = part; b < no; b += 8) cc += CQ[c * no + b] * (xbar[b] - goal[b]);
      } else if (c >= nx && c < n) {
        for (int j = part; j < nu; j += 8) cc += CR[(c - nx) * nu + j] * ubar[j];
      }
      cc += __shfl_xor(cc, 1);
      cc += __shfl_xor(cc, 2);
      cc += __shfl_xor(cc, 4);
      if (part == 0 && c < n) cq[buf * n + c] = (c < no ? cc + clin[c] : cc) * dt;
    }
  };
  if (sid >= 0) { fetch_step(H - 1); commit_step(); }
  __syncthreads();
  if (sid >= 0) stage_gradient((H - 1) & 1);
  const int kx = R.nxp4 / 4;                            // k-steps over the state dimension
  const int mtx = R.nxp16 / 16, mtn = R.nr16 / 16, ntn = R.np16 / 16, ntb = R.nb16 / 16;
  // loop-invariant part of phase B's epilogue: the Hessian of the stage cost at this lane's entries
  T ccB[TPW][4];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int tile = w + ti * NW, mt = tile / ntn, nt = tile - mt * ntn, d = 16 * nt + i16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * mt + acc_row<T>(q, r);
      T cc = T(0);
      if (tile < mtn * ntn) {
        if (c < no && d < no) cc = CQ[c * no + d] * dt;
        else if (c >= nx && c < n && d >= nx && d < n) cc = CR[(c - nx) * nu + (d - nx)] * dt;
      }
      ccB[ti][r] = cc;
    }
  }
  __syncthreads();
  T lin = T(0), quad = T(0), ksn2 = T(0);               // meaningful in thread nx only
  int sing_any = 0;
  AMPC_IMARK_ALWAYS(30);
  for (int t = H - 1; t >= 0; --t) {
    AMPC_IPROBE_STEP(t == H / 2);
    AMPC_IMARK(20);
    // ---- A: VJ = V J on the tile waves; side threads issue the next step's loads
    if (sid >= 0 && t > 0) fetch_step(t - 1);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int tile = w + ti * NW;
      if (tile < mtx * ntn) {
        const int mt = tile / ntn, nt = tile - mt * ntn;
        const T* a = V + (16 * mt + i16) * ldV + q;
        const T* b = Jm + q * ldJ + 16 * nt + i16;
        T av[KX], bv[KX];
#pragma unroll
        for (int ks = 0; ks < KX; ++ks)
          if (ks < kx) { av[ks] = a[4 * ks]; bv[ks] = b[4 * ks * ldJ]; }
        __builtin_amdgcn_sched_barrier(0);
        acc_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KX; ++ks)
          if (ks < kx) acc = mfma16(av[ks], bv[ks], acc);
        const int col = 16 * nt + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + acc_row<T>(q, r);
          if (row < R.nxp4 && col < n) VJ[row * ldJ + col] = acc[r];
        }
      }
    }
    AMPC_IMARK(10);
    lds_barrier();
    AMPC_IMARK(21);
    // ---- B: Qt = Ct + J' [VJ | v]
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int tile = w + ti * NW;
      if (tile < mtn * ntn) {
        const int mt = tile / ntn, nt = tile - mt * ntn;
        const T* a = Jm + q * ldJ + 16 * mt + i16;
        const T* b = VJ + q * ldJ + 16 * nt + i16;
        const int d = 16 * nt + i16;
        T av[KX], bv[KX], cqv[4];
#pragma unroll
        for (int ks = 0; ks < KX; ++ks)
          if (ks < kx) { av[ks] = a[4 * ks * ldJ]; bv[ks] = b[4 * ks * ldJ]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mt + acc_row<T>(q, r);
          cqv[r] = cq[(t & 1) * n + (c < n ? c : 0)];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KX; ++ks)
          if (ks < kx) acc = mfma16(av[ks], bv[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mt + acc_row<T>(q, r);
          if (c < n && d <= n) Qt[c * ldQ + d] = (d == n ? cqv[r] : ccB[ti][r]) + acc[r];
        }
      }
    }
    AMPC_IMARK(11);
    lds_barrier();
    AMPC_IMARK(22);
    // ---- C: the nu x nu solves, one right-hand side per lane; side threads commit step t-1
    if (sid >= 0 && t > 0) commit_step();
    if (tid < 64) {
      const int colr = tid < nx ? tid : n;             // this lane's right-hand-side column of Qt
      const bool valid = tid <= nx;
      T A0[NU][NU], x[NU], x0[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
#pragma unroll
        for (int j = 0; j < NU; ++j) A0[i][j] = Qt[(nx + i) * ldQ + nx + j];
        x0[i] = Qt[(nx + i) * ldQ + colr];
        x[i] = x0[i];
      }
      int ok = ldl_solve_lane<T, NU>(A0, x);
      ok = __builtin_amdgcn_readfirstlane(ok);         // identical in every lane
      if (!ok) {
#pragma unroll
        for (int i = 0; i < NU; ++i) x[i] = x0[i];
        sing_any |= lu_solve_lane<T, NU>(A0, x);
      }
      T l = T(0), qd = T(0), k2 = T(0);
#pragma unroll
      for (int i = 0; i < NU; ++i) x[i] = -x[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NU; ++j) s += A0[i][j] * x[j];
        l += x0[i] * x[i];
        qd += x[i] * s;
        k2 += x[i] * x[i];
        if (valid) {
          SB[i * ldB + tid] = x[i];
          SB[(nu4 + i) * ldB + tid] = x0[i] + s;
        }
      }
      if (tid == nx) { lin += l; quad += qd; ksn2 += k2; }
    }
    AMPC_IMARK(23);
    lds_barrier();
    AMPC_IMARK(24);
    // ---- D: V = Qxx + [Qxu | K'] [K ; Z] (column nx: v); side threads store K_t, k_t and
    //         prepare the next step's stage-cost gradient
    if (sid >= 0) {
      for (int i = sid; i < nu * nx; i += SIDE) {
        const int j = i / nx, b = i - j * nx;
        Kg[(size_t)t * nu * nx + i] = SB[j * ldB + b];
      }
      if (sid < nu) kg[(size_t)t * nu + sid] = SB[sid * ldB + nx];
      if (t > 0) stage_gradient((t - 1) & 1);
    }
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int tile = w + ti * NW;
      if (tile < mtx * ntb) {
        const int mt = tile / ntb, nt = tile - mt * ntb;
        const int b = 16 * nt + i16;
        T av[KD], bv[KD], q0[4];
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
          av[ks] = ks < KD / 2 ? Qt[(16 * mt + i16) * ldQ + nx + 4 * ks + q]
                               : SB[(4 * ks - nu4 + q) * ldB + 16 * mt + i16];
          bv[ks] = SB[(4 * ks + q) * ldB + 16 * nt + i16];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * mt + acc_row<T>(q, r);
          q0[r] = Qt[a * ldQ + (b < nx ? b : n)];       // rows >= nx: finite padding, discarded
        }
        __builtin_amdgcn_sched_barrier(0);
        acc_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) acc = mfma16(av[ks], bv[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * mt + acc_row<T>(q, r);
          if (a < nx) {
            if (b < nx) V[a * ldV + b] = q0[r] + acc[r];
            else if (b == nx) VJ[a * ldJ + n] = q0[r] + acc[r];
          }
        }
      }
    }
    AMPC_IMARK(25);
    lds_barrier();
    AMPC_IMARK(26);
  }
  AMPC_IMARK_ALWAYS(33);
  if (tid < 64 && sing_any) scal[0] = T(1);
  __syncthreads();
  if (tid == nx) {
    T* out = args.ric + (size_t)p * kRicStride;
    const T sg = scal[0];
    out[0] = lin; out[1] = quad; out[2] = sqrt(ksn2); out[3] = sg;
    if (sg != T(0)) {             // singular Quu: the reference raises LinAlgError here
      args.status[p] = 1; args.active[p] = 0; args.refresh[p] = 0;
    }
  }
}

// DYN = 0: MLP dynamics through the MFMA tile;  DYN = 1: SINDy feature-library dynamics, one
// thread per line-search candidate (the model is tiny; see sindy_kernels.hpp);  DYN = 2: wide linear
// models (65..128 states), the K-tiled step of linear_kernels.hpp on the 16 candidate rows.
template <typename T, int NT, int W, int DYN = 0, typename SH = DynShape, bool WIDE = false>
__global__ __launch_bounds__(64 * W) void ilqr_iter_kernel(const IlqrArgs<T> args) {
  const int mode = args.slot_mode ? args.slot_mode[ilqr_slot(args)] : args.mode;   // (queue: per slot)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  // nothing resident: the kernel is launched once per iteration and runs only H steps, so filling
  // resident fragments does not pay (measured with a static shape: 0.74 ms resident vs 0.58 ms)
  using Net = TileNet<T, NT, 1, W, false, 2, SH, WIDE>;
  constexpr int M = 16, NTHR = 64 * W, TPS = NTHR / M;
  const int tid = threadIdx.x, p = ilqr_slot(args);
  const MlpDev<T> mlp = plan_model<SH, T>(args.mlp, [&] {               // (per-slot models: mlp_tile.hpp)
    return model_delta_of(args.model_delta, args.model_delta ? args.slot_model[p] : 0); });
  const TileLds L = SH::template fold_lds<T, M, W>(args.lds);
  const int nx = mlp.nx, nu = mlp.nu, n = nx + nu, no = SH::kStatic ? SH::no : args.obs_dim;
  const int HS = args.H, H = args.slot_h ? args.slot_h[p] : HS;      // array stride, this slot's horizon
  const int xs_ = L.xu_stride;
  const int cost_stride = SH::kStatic ? cost_block_stride(SH::no, SH::nu) : args.cost_stride;
  const IlqrWork wk = make_ilqr_work(nx, nu, cost_stride, DYN == 2);
  T* Wr = lds + (SH::kStatic ? L.extra : args.lds_work);
  T* V = Wr + wk.V; T* v = Wr + wk.v; T* Jm = Wr + wk.J; T* VJ = Wr + wk.VJ;
  T* Qt = Wr + wk.Qt; T* qt = Wr + wk.qt; T* Km = Wr + wk.K; T* kv = Wr + wk.k;
  T* Wk = Wr + wk.Wk; T* wq = Wr + wk.wq; T* lu = Wr + wk.lu; T* rhs = Wr + wk.rhs;
  T* xbar = Wr + wk.xbar; T* ubar = Wr + wk.ubar; T* cpar = Wr + wk.cpar;
  T* blo = Wr + wk.lo; T* bhi = Wr + wk.hi; T* scal = Wr + wk.scal; T* lsobj = Wr + wk.lsobj;
  int* piv = reinterpret_cast<int*>(Wr + wk.piv);
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;
  const T* clin = goal + no; const T* clint = clin + no;     // affine part of the stage / terminal cost
  T* xu = lds + L.xu;

  if (mode == 1 && args.active[p] == 0) {
    if (tid == 0) args.refresh[p] = 0;
    return;
  }

  Net net;
  if constexpr (DYN == 0) {
    tile_load_constants<T, W>(mlp, L, lds, M);   // (net.init: after the Riccati sweep, which
  } else {                                       //  wants the registers for itself)
    for (int i = tid; i < M * L.xu_stride; i += NTHR) lds[L.xu + i] = T(0);
  }
  for (int i = tid; i < cost_stride; i += NTHR)
    cpar[i] = args.costs_par[(size_t)args.cost_idx[p] * cost_stride + i];
  for (int i = tid; i < nu; i += NTHR) {
    blo[i] = args.bounded ? args.ubounds[i] : T(0);
    bhi[i] = args.bounded ? args.ubounds[nu + i] : T(0);
  }
  __syncthreads();

  T* st = args.states + (size_t)p * (HS + 1) * nx;
  T* ct = args.ctrls + (size_t)p * HS * nu;
  T* Kg = args.Ks + (size_t)p * HS * nu * nx;
  T* kg = args.ks + (size_t)p * HS * nu;
  T lin = T(0), quad = T(0), ksn2 = T(0);   // meaningful in thread 0 only

  // (backward Riccati sweep: ilqr_riccati_kernel above, launched just before this kernel)
  if (mode == 1) {
    const T* rin = args.ric + (size_t)p * kRicStride;
    if (rin[3] != T(0)) return;     // singular Quu: the sweep already retired this problem
    if (tid == 0) { scal[0] = rin[0]; scal[1] = rin[1]; scal[2] = rin[2]; }
    __syncthreads();
  }

  // =========================== forward rollout(s) (ilqr.py:141-149, 196-205) ===================
  AMPC_IMARK_ALWAYS(31);
  if constexpr (DYN == 0) net.init(mlp);
  const int rows = mode == 0 ? 1 : args.ls_n;
  const int m = tid / TPS, r = tid % TPS;        // row-in-tile, helper index (same wave)
  T obj_part = T(0);
  const T alpha = args.alphas[m];
  for (int i = tid; i < M * nx; i += NTHR) {
    const int row = i / nx, col = i - row * nx;
    xu[row * xs_ + col] = st[col];
  }
  __syncthreads();
  T* lss = args.ls_states + (size_t)p * args.ls_n * (HS + 1) * nx;
  T* lsc = args.ls_ctrls + (size_t)p * args.ls_n * HS * nu;
  // K_t, k_t, ubar_t, xbar_t reach LDS one step ahead, through registers (as in the sweep): the
  // global loads for step t+1 are issued at the top of step t and committed at its end; the
  // barriers inside the loop are LDS-only, so neither these loads nor the line-search stores
  // (lss / lsc) stall a step.
  // nu <= 16; nx <= 32 with the MLP tile, <= 64 on the feature-library path (predicated on nu * nx)
  constexpr int KR = (16 * (DYN == 2 ? kLinMaxIlqrNx : (DYN == 1 || WIDE) ? 64 : 32) + NTHR - 1) / NTHR;
  T kreg[KR];
  T kvr = T(0), ubr = T(0), xbr = T(0);
  auto fetch_ls = [&](int t) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int idx = tid + k * NTHR;
      if (idx < nu * nx) kreg[k] = Kg[(size_t)t * nu * nx + idx];
    }
    if (tid < nu) { kvr = kg[(size_t)t * nu + tid]; ubr = ct[(size_t)t * nu + tid]; }
    if (tid < nx) xbr = st[(size_t)t * nx + tid];
  };
  auto commit_ls = [&]() {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int idx = tid + k * NTHR;
      if (idx < nu * nx) Km[idx] = kreg[k];
    }
    if (tid < nu) { kv[tid] = kvr; ubar[tid] = ubr; }
    if (tid < nx) xbar[tid] = xbr;
  };
  if (mode == 1) {
    fetch_ls(0);
    commit_ls();
    __syncthreads();
  }
  const bool cdiag = args.cost_diag != 0, caff = args.cost_affine != 0;
  for (int t = 0; t < H; ++t) {
    AMPC_IPROBE_STEP(mode == 1 && t == H / 2);
    AMPC_IMARK(40);
    if (mode == 1 && t + 1 < H) fetch_ls(t + 1);
    // controls for this step
    for (int a = r; a < nu; a += TPS) {
      T u;
      if (mode == 0) {
        u = ct[(size_t)t * nu + a];
      } else {
        // four partial sums over interleaved b: the LDS reads of a whole group are in flight
        // together instead of one dependent round trip per term
        T f0 = T(0), f1 = T(0), f2 = T(0), f3 = T(0);
        int b = 0;
        for (; b + 4 <= nx; b += 4) {
          const T k0 = Km[a * nx + b], k1 = Km[a * nx + b + 1], k2 = Km[a * nx + b + 2], k3 = Km[a * nx + b + 3];
          const T d0 = xu[m * xs_ + b] - xbar[b], d1 = xu[m * xs_ + b + 1] - xbar[b + 1];
          const T d2 = xu[m * xs_ + b + 2] - xbar[b + 2], d3 = xu[m * xs_ + b + 3] - xbar[b + 3];
          f0 += k0 * d0; f1 += k1 * d1; f2 += k2 * d2; f3 += k3 * d3;
        }
        for (; b < nx; ++b) f0 += Km[a * nx + b] * (xu[m * xs_ + b] - xbar[b]);
        const T fb = (f0 + f1) + (f2 + f3);
        u = alpha * kv[a] + ubar[a] + fb;
        if (args.bounded) { u = u < blo[a] ? blo[a] : u; u = u > bhi[a] ? bhi[a] : u; }
        if (m < rows) lsc[((size_t)m * HS + t) * nu + a] = u;
      }
      xu[m * xs_ + nx + a] = u;
    }
    if (mode == 1 && m < rows)
      for (int a = r; a < nx; a += TPS) lss[((size_t)m * (HS + 1) + t) * nx + a] = xu[m * xs_ + a];
    AMPC_IMARK(41);
    lds_barrier();
    AMPC_IMARK(42);
    // objective: dt * (stage costs)
    obj_part += args.dt * (quad_rows<T>(Qm, xu + m * xs_, goal, no, r, TPS, cdiag) +
                           quad_rows<T>(Rm, xu + m * xs_ + nx, nullptr, nu, r, TPS, cdiag));
    if (caff) obj_part += args.dt * affine_rows<T>(clin, xu + m * xs_, goal, no, r, TPS, clint[no]);
    AMPC_IMARK(43);
    if constexpr (DYN == 0) {
      net.run(mlp, L, lds);
      AMPC_IMARK(44);
      for (int a = r; a < nx; a += TPS) {
        const T xn = xu[m * xs_ + a] + Net::output(mlp, L, lds, m, a);
        xu[m * xs_ + a] = xn;
        if (mode == 0 && m == 0) st[(size_t)(t + 1) * nx + a] = xn;
      }
    } else if constexpr (DYN == 2) {
      // x_{t+1} = M [x_t ; u_t] for the 16 rows: wave w takes output column tiles w, w + W, ...
      T* xnext = lds + args.lds_xn;
      const int lane = tid & 63, wv = tid >> 6;
      for (int nt = wv; nt < args.lin.ntile; nt += W) {
        const typename Acc<T>::type acc = lin_tile<T>(args.lin, xu, xs_, nt, lane);
        const int col = 16 * nt + (lane & 15);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (col < nx) xnext[acc_row<T>(lane >> 4, rr) * nx + col] = acc[rr];
      }
      __syncthreads();
      for (int a = r; a < nx; a += TPS) {
        const T xn = xnext[m * nx + a];
        xu[m * xs_ + a] = xn;
        if (mode == 0 && m == 0) st[(size_t)(t + 1) * nx + a] = xn;
      }
    } else {
      T* xnext = lds + args.lds_xn;
      if (r == 0)
        sindy_step<T>(args.sindy, xu + m * xs_, 1, xnext + m * nx, 1,
                      xnext + M * nx + m * args.sindy.n_tab, 1);   // table scratch after xnext
      __syncthreads();
      for (int a = r; a < nx; a += TPS) {
        const T xn = xnext[m * nx + a];
        xu[m * xs_ + a] = xn;
        if (mode == 0 && m == 0) st[(size_t)(t + 1) * nx + a] = xn;
      }
    }
    // every thread passed the barrier above after its last read of K, k, ubar, xbar
    AMPC_IMARK(45);
    if (mode == 1 && t + 1 < H) commit_ls();
    lds_barrier();
    AMPC_IMARK(46);
  }
  AMPC_IMARK_ALWAYS(32);
  if (mode == 1 && m < rows)
    for (int a = r; a < nx; a += TPS) lss[((size_t)m * (HS + 1) + H) * nx + a] = xu[m * xs_ + a];
  obj_part += quad_rows<T>(Fm, xu + m * xs_, goal, no, r, TPS, cdiag);
  if (caff) obj_part += affine_rows<T>(clint, xu + m * xs_, goal, no, r, TPS, clint[no + 1]);
#pragma unroll
  for (int off = TPS / 2; off > 0; off >>= 1) obj_part += __shfl_xor(obj_part, off);
  if (r == 0) lsobj[m] = obj_part;
  __syncthreads();

  if (mode == 0) {
    if (tid == 0) {
      args.obj[p] = lsobj[0];
      args.active[p] = 1; args.converged[p] = 0; args.iters[p] = 0; args.status[p] = 0;
      args.refresh[p] = 1; args.ls_rows[p] = 0;
    }
    return;
  }

  // =========================== acceptance (ilqr.py:207-261) ====================================
  if (tid == 0) {
    const T obj = args.obj[p];
    const T lin_ = scal[0], quad_ = scal[1], ksn = scal[2];
    T best_obj = INFINITY;
    int best = -1, last = 0;
    for (int j = 0; j < rows; ++j) {
      last = j;
      const T a = args.alphas[j];
      const T new_obj = lsobj[j];
      const T expect = a * lin_ + a * a * quad_ / T(2);
      const T ratio = (obj - new_obj) / (-expect);
      if (ratio > args.ls_cost_threshold) { best_obj = new_obj; best = j; break; }
      if (new_obj < best_obj) { best_obj = new_obj; best = j; }
      if (ksn < args.u_threshold) break;
    }
    const bool success = (best_obj < obj) || (ksn < args.u_threshold);
    int sel = success ? best : last;
    int fail = 0;
    if (best < 0) { fail = 1; sel = 0; if (success) args.status[p] = 2; }
    const T new_obj = lsobj[sel];
    if (!success && new_obj > obj + T(1e-3)) fail = 1;
    piv[0] = sel; piv[1] = fail; piv[2] = success ? 1 : 0;
    scal[3] = new_obj;
    args.iters[p] += 1;
    args.ls_rows[p] += M;
  }
  __syncthreads();
  const int sel = piv[0], fail = piv[1], success = piv[2];
  if (fail) {
    if (tid == 0) { args.active[p] = 0; args.refresh[p] = 0; }
    return;
  }
  // ||new_ctrls - ctrls||, then swap in the selected candidate
  T du2 = T(0);
  for (int i = tid; i < H * nu; i += NTHR) {
    const T d = lsc[(size_t)sel * HS * nu + i] - ct[i];
    du2 += d * d;
  }
  du2 = block_sum_any(du2, lsobj, W);
  for (int i = tid; i < H * nu; i += NTHR) ct[i] = lsc[(size_t)sel * HS * nu + i];
  for (int i = tid; i < (H + 1) * nx; i += NTHR) st[i] = lss[(size_t)sel * (HS + 1) * nx + i];
  if (tid == 0) {
    const bool conv = sqrt(du2) < args.u_threshold;
    args.obj[p] = scal[3];
    args.refresh[p] = success;
    if (conv) { args.converged[p] = 1; args.active[p] = 0; }
    else if (args.max_iter > 0 && args.iters[p] >= args.max_iter) args.active[p] = 0;
    if (args.active[p] == 0) args.refresh[p] = 0;       // retired: nobody reads its Jacobians again
  }
}

// ---- continuous batching (ampc_ilqr_solve_queue) ---------------------------------------------------
// P problems stream through the plan's B slots.  Once per iteration, ahead of the sweep, every idle
// slot (active == 0) hands the finished problem's results to its output row and takes the next
// unsolved problem from the queue: x0 / guess / cost block into the slot, slot_mode = 0 -- the
// line-search launch of this iteration rolls the guess out instead (what ampc_ilqr_solve does before
// its first iteration), the Jacobian launch refreshes it, and from the next iteration on the slot
// iterates with everybody else.  No host round trip; a problem's arithmetic is the one-problem solve's.
template <typename T> struct IlqrQueue {
  int P, B, H, nx, nu;       // H: the plan's horizon = the stride of every [..][H][..] array below
  const int* horizon;        // [P] per-problem horizon <= H (rows past it: zero in the outputs), nullptr: H
  int* slot_h;               // [B] the plan's per-slot horizon (read by every kernel), with `horizon`
  const int* model;          // [P] per-problem controller model (entry of IlqrArgs::model_delta), nullptr: the plan's
  int* slot_model;           // [B] the plan's per-slot model, with `model`
  int* ctl;                  // [0] next problem to hand out, [1] problems harvested
  int* slot_prob;            // [B] problem in the slot, -1: none
  const T* x0;               // [P][nx]
  const T* uguess;           // [P][H][nu]
  const int* cost;           // [P] cost block of the problem
  int* cost_idx;             // [B] the plan's per-slot cost block (read by every kernel)
  T* out_states; T* out_ctrls; T* out_Ks; T* out_ks; T* out_obj;      // [P] rows
  int* out_flags;            // [P][4] converged, iters, status, candidate rows
};

// A slot's mode is IMMUTABLE within a line-search launch: the passes of a search may run as separate
// workgroups of that launch (grid.y, ilqr_ls4.hpp), every one of which reads slot_mode[p] at entry -- were
// the workgroup that rolls out a fresh guess to flip the mode to "iterate" itself, a pass workgroup
// dispatched after it had finished would take the slot for an iterating one and search along the
// PREVIOUS problem's gains.  So the rollout only raises active[p]; the flip happens here, in the first
// kernel of the next iteration (one workgroup per slot, ahead of the sweep).  Returns true for such a slot
// (it is running: nothing to harvest or refill).
template <typename T> __device__ __forceinline__ bool ilqr_slot_started(const IlqrArgs<T>& args, int p, int tid) {
  if (args.slot_mode[p] != 0 || args.active[p] == 0) return false;
  __syncthreads();                                     // (every thread has read the mode before it changes)
  if (tid == 0) args.slot_mode[p] = 1;
  return true;
}

// slot_of for the iteration being queued (one workgroup, after the refill / chain kernels): slots that have work --
// solving, or loaded and about to roll out their guess -- first and in slot order, the idle ones behind them.
template <typename T>
__global__ __launch_bounds__(256) void ilqr_compact_kernel(const IlqrArgs<T> args, int B, int* __restrict__ slot_of) {
  __shared__ int cnt[256];
  __shared__ int total_s;
  const int tid = threadIdx.x;
  const int per = (B + 255) / 256, lo = tid * per < B ? tid * per : B, hi = lo + per < B ? lo + per : B;
  auto has_work = [&](int p) { return args.active[p] != 0 || args.slot_mode[p] == 0; };
  int c = 0;
  for (int p = lo; p < hi; ++p) c += has_work(p) ? 1 : 0;
  cnt[tid] = c;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int v = cnt[i]; cnt[i] = run; run += v; }
    total_s = run;
  }
  __syncthreads();
  if (tid == 0) slot_of[B] = total_s;                   // (read by the refresh kernels: row groups past it have no live row)
  int w = cnt[tid], d = total_s + (lo - cnt[tid]);      // next position among the slots with work / the idle ones
  for (int p = lo; p < hi; ++p) {
    if (has_work(p)) slot_of[w++] = p;
    else slot_of[d++] = p;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ilqr_queue_refill_kernel(const IlqrArgs<T> args, const IlqrQueue<T> q) {
  __shared__ int next_s;
  const int p = blockIdx.x, tid = threadIdx.x;
  if (ilqr_slot_started(args, p, tid)) return;
  if (args.active[p] != 0 || args.slot_mode[p] == 0) return;      // running, or loaded and not yet started
  const int H = q.H, nx = q.nx, nu = q.nu;
  T* st = args.states + (size_t)p * (H + 1) * nx;
  T* ct = args.ctrls + (size_t)p * H * nu;
  const int j = q.slot_prob[p];
  if (j >= 0) {                                        // harvest (rows past the problem's own horizon: zeros)
    const int hj = q.horizon ? q.horizon[j] : H;
    for (int i = tid; i < (H + 1) * nx; i += 256) q.out_states[(size_t)j * (H + 1) * nx + i] = i < (hj + 1) * nx ? st[i] : T(0);
    for (int i = tid; i < H * nu; i += 256) q.out_ctrls[(size_t)j * H * nu + i] = i < hj * nu ? ct[i] : T(0);
    for (int i = tid; i < H * nu * nx; i += 256)
      q.out_Ks[(size_t)j * H * nu * nx + i] = i < hj * nu * nx ? args.Ks[(size_t)p * H * nu * nx + i] : T(0);
    for (int i = tid; i < H * nu; i += 256) q.out_ks[(size_t)j * H * nu + i] = i < hj * nu ? args.ks[(size_t)p * H * nu + i] : T(0);
    if (tid == 0) {
      q.out_obj[j] = args.obj[p];
      q.out_flags[4 * j] = args.converged[p]; q.out_flags[4 * j + 1] = args.iters[p];
      q.out_flags[4 * j + 2] = args.status[p]; q.out_flags[4 * j + 3] = args.ls_rows[p];
    }
  }
  if (tid == 0) {
    int n = q.ctl[0] < q.P ? atomicAdd(&q.ctl[0], 1) : q.P;
    if (n >= q.P) n = -1;
    next_s = n;
    q.slot_prob[p] = n;
    if (n >= 0) {
      q.cost_idx[p] = q.cost[n]; args.slot_mode[p] = 0; args.refresh[p] = 0;
      if (q.horizon) q.slot_h[p] = q.horizon[n];
      if (q.model) q.slot_model[p] = q.model[n];
    }
    if (j >= 0) { __threadfence(); atomicAdd(&q.ctl[1], 1); }
  }
  __syncthreads();
  const int n = next_s;
  if (n < 0) return;
  for (int i = tid; i < nx; i += 256) st[i] = q.x0[(size_t)n * nx + i];
  for (int i = tid; i < H * nu; i += 256) ct[i] = q.uguess[(size_t)n * H * nu + i];
}

// ---- device-resident closed loops of iLQR controllers (ampc_ilqr_closed_loop) ------------------------------
// simulate() with an IterativeLQR controller (utils/simulation.py:44-63, ilqr.py:267-295): every control
// step is a full solve from a zero guess, u = ubar_0 (+ K_0 (x - xbar_0) with x = xbar_0), then
// x <- surrogate.pred(x, u).  C such episodes ("chains") stream through the plan's B slots: a slot keeps
// its chain -- solve, surrogate step, next solve -- until the episode ends, then takes the next chain.
// Per iteration of the plan:   pre (this kernel)  ->  surrogate step over the B staged rows  ->  post.
template <typename T> struct IlqrChains {
  int C, B, H, nx, nu, n_steps, max_iter;     // H: the plan's horizon (array stride)
  const int* horizon;        // [C] per-chain iLQR horizon <= H, nullptr: H
  int* slot_h;               // [B] the plan's per-slot horizon, with `horizon`
  const int* model;          // [C] per-chain controller model, nullptr: the plan's
  int* slot_model;           // [B]
  int* ctl;                  // [0] next chain to hand out, [1] chains finished
  int* slot_chain;           // [B] chain in the slot, -1: none
  int* need;                 // [B] 0 nothing, 1 step the surrogate and continue, 2 chain failed, 3 wants a chain
  int* chain_t;              // [C] control steps done
  int* chain_fail;           // [C] status of the solve that ended the chain early (1: singular Quu), else 0
  long long* chain_iters;    // [C] iLQR iterations over the episode
  const T* x0;               // [C][nx]
  const int* cost;           // [C]
  int* cost_idx;             // [B]
  T* stage_x; T* stage_u; T* stage_next;     // [B][nx], [B][nu], [B][nx]
  T* traj_obs;               // [C][n_steps+1][nx]
  T* traj_ctrls;             // [C][n_steps+1][nu]  (row n_steps stays zero, as simulate() appends it)
};

template <typename T>
__global__ __launch_bounds__(64) void ilqr_chain_pre_kernel(const IlqrArgs<T> args, const IlqrChains<T> q) {
  const int p = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) q.need[p] = 0;
  if (ilqr_slot_started(args, p, tid)) return;
  if (args.active[p] != 0 || args.slot_mode[p] == 0) return;        // solving, or loaded and not yet started
  const int c = q.slot_chain[p];
  if (c < 0) { if (tid == 0) q.need[p] = 3; return; }
  if (tid == 0) q.chain_iters[c] += args.iters[p];
  if (args.status[p] == 1) {                    // singular Quu -- the reference's LinAlgError: the episode ends
    if (tid == 0) { q.chain_fail[c] = 1; q.need[p] = 2; }
    return;
  }
  const T* st = args.states + (size_t)p * (q.H + 1) * q.nx;
  const T* ct = args.ctrls + (size_t)p * q.H * q.nu;
  for (int i = tid; i < q.nx; i += 64) q.stage_x[(size_t)p * q.nx + i] = st[i];
  for (int i = tid; i < q.nu; i += 64) q.stage_u[(size_t)p * q.nu + i] = ct[i];
  if (tid == 0) q.need[p] = 1;
}

template <typename T>
__global__ __launch_bounds__(256) void ilqr_chain_post_kernel(const IlqrArgs<T> args, const IlqrChains<T> q) {
  __shared__ int next_s;
  const int p = blockIdx.x, tid = threadIdx.x;
  const int need = q.need[p];
  if (need == 0) return;
  const int H = q.H, nx = q.nx, nu = q.nu, T1 = q.n_steps + 1;
  T* st = args.states + (size_t)p * (H + 1) * nx;
  T* ct = args.ctrls + (size_t)p * H * nu;
  int c = q.slot_chain[p];
  bool finished = need == 2;
  if (need == 1) {
    const int t = q.chain_t[c];
    for (int i = tid; i < nu; i += 256) q.traj_ctrls[((size_t)c * T1 + t) * nu + i] = q.stage_u[(size_t)p * nu + i];
    for (int i = tid; i < nx; i += 256) {
      const T v = q.stage_next[(size_t)p * nx + i];
      q.traj_obs[((size_t)c * T1 + t + 1) * nx + i] = v;
      st[i] = v;                                  // the next solve of this chain starts here ...
    }
    __syncthreads();
    if (tid == 0) q.chain_t[c] = t + 1;
    if (t + 1 < q.n_steps) {                      // ... from a zero guess (ilqr.py:280-281)
      for (int i = tid; i < H * nu; i += 256) ct[i] = T(0);
      if (tid == 0) { args.slot_mode[p] = 0; args.refresh[p] = 0; }
      return;
    }
    finished = true;
  }
  if (tid == 0) {
    if (finished) { __threadfence(); atomicAdd(&q.ctl[1], 1); }
    int n = q.ctl[0] < q.C ? atomicAdd(&q.ctl[0], 1) : q.C;
    if (n >= q.C) n = -1;
    next_s = n;
    q.slot_chain[p] = n;
    if (n >= 0) {
      q.cost_idx[p] = q.cost[n]; args.slot_mode[p] = 0; args.refresh[p] = 0; q.chain_t[n] = 0;
      if (q.horizon) q.slot_h[p] = q.horizon[n];
      if (q.model) q.slot_model[p] = q.model[n];
    }
  }
  __syncthreads();
  const int n = next_s;
  if (n < 0) return;
  for (int i = tid; i < nx; i += 256) {
    const T v = q.x0[(size_t)n * nx + i];
    st[i] = v;
    q.traj_obs[(size_t)n * T1 * nx + i] = v;
  }
  for (int i = tid; i < H * nu; i += 256) ct[i] = T(0);
}

}  // namespace ampc
