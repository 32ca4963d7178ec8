// mppi_kernels.hpp -- MPPI solve on gfx950: persistent rollout kernel + softmin update kernel.
//
// What one solve computes (reference: autompc/control/mppi.py:110-152):
//   a      <- shift(a)                                   a[t] = a_old[min(t+1, H-1)]
//   for t: A = clip(eps_t + a_t, lo, hi); eps_t <- A - a_t; u = A * umax
//          c  += (x-g)'Q(x-g) + u'Ru ;  ca += (lmda/sigma) * sum_j A_j eps_tj ;  x <- f(x, u)
//   c += terminal (last particle's, or per particle) + ca
//   w = softmax(-(c - min c)/lmda) ;  a += sum_n w_n eps[:, n, :]
//
// Kernel 1 (rollout): one workgroup owns M = 16*MT samples of one problem for ALL H time steps
// (the time loop is serial, parallelism is over samples / problems).  State, running costs and
// activations never leave the CU; per step the only HBM traffic is the sample's noise row in and
// the clipped noise out.  MLP weights stream from L2 in MFMA-fragment order (mlp_tile.hpp).
// Kernel 2 (update): one workgroup per (problem, time step): wave-shuffle min / sum reductions
// for the softmin weights, then the weighted noise sum for that step.
#pragma once
#include "mlp_tile.hpp"
#include <stdint.h>

namespace ampc {

constexpr int kWG = 256;      // threads of the update / combine / finalize kernels
constexpr int kWaves = 4;

template <typename T> struct MppiProblem {
  int N, H;            // samples, horizon
  int tile0;           // first rollout workgroup of this problem
  int cost_idx;        // cost block
  T lam_over_sigma;    // lmda / sigma   (mppi.py:143)
  T neg_inv_lambda;    // -1 / lmda      (mppi.py:115)
  T sqrt_sigma;        // noise std      (mppi.py:18)
  long long eps_off;   // into eps      [N][H][nu]
  long long epso_off;  // into eps_out  [H][N][nu]
  long long cost_off;  // into costs    [N]
  int a_off;           // into act_seq  [H][nu]
  unsigned noise_id;   // keys the problem's device noise stream (default: index in the plan; the
                       // candidate evaluator sets the candidate's GLOBAL index, so a candidate's
                       // noise does not depend on how the batch is sharded over GPUs)
  int model;           // entry of MppiArgs::model_delta (ampc_mppi_plan_set_models), 0 without a table
};

template <typename T> struct MppiArgs {
  MlpDev<T> mlp;
  TileLds lds;
  int lds_aseq, lds_cost;       // extra LDS regions: shifted act sequence, cost block + bounds
  int obs_dim, cost_stride;     // cost block = Q R F goal lin lint c0 c1 (mlp_tile.hpp: cost_block_stride)
  int term_mode, max_h;
  int cost_diag;                // 1: every Q, R, F is diagonal -> O(n) stage cost
  int cost_affine;              // 1: some block has a non-zero affine part (sum of quadratics with
                                //    different goals, sum_cost.py:49-54)
  int lds_eps;                  // >= 0: clipped noise of the tile is kept in LDS ([max_h][M][nu]) and
                                //       the softmin update is fused (tile partials + combine kernel)
  int lds_red;                  // small reduction scratch: [M] costs, [M] weights
  int write_eps_out;            // materialise the clipped noise in HBM (needed only for download)
  int eps_inline;               // four-row rollout: 1 = the noise is Philox(eps_seed, eps_stream, noise id),
  unsigned long long eps_seed, eps_stream;   // formed in the kernel's prologue (no generator launch, no buffer)
  int hnu_stride;               // max_h * nu: row length of tile_part
  const T* costs_par;           // [n_costs][cost_stride]
  const T* bounds;              // lo[nu] hi[nu] scale[nu]   (lo, hi already divided by scale)
  const MppiProblem<T>* probs;
  const int* tile_prob;         // [n_tiles] -> problem
  const T* x0;                  // [B][nx]
  const T* act_in;              // [sum H*nu]
  T* act_out;                   // [sum H*nu]
  const T* eps;                 // noise in
  T* eps_out;                   // clipped noise out
  T* costs;                     // per-sample cost (terminal scalar of reference mode NOT included)
  T* term_last;                 // [B] terminal cost of the last particle (reference mode)
  T* u_out;                     // [B][nu] first action * scale (written by the update kernel)
  T* tile_stat;                 // [n_tiles][2]  fused update: tile min cost, tile weight sum
  T* tile_part;                 // [n_tiles][hnu_stride]  fused update: sum_m S_m eps[t][m][j]
  int* tile_done;               // [B] tickets: the four-row rollout's last workgroup of a problem finishes the
                                // softmin update itself (fused_combine; no combine launch)
  int fused_combine;
  // (behind everything the shape-specialised kernels read: their argument layout is the tuned one)
  int n_ind;                    // indicator terms of the stage cost (threshold / box; mlp_tile.hpp), shared by every
  const T* ind_tab;             // cost block: [n_ind][ind_stride(obs_dim)]; 0: none
  const long long* model_delta; // per-problem controller models of the plan's shape: byte offsets of their buffers
                                // from the plan model's (MppiProblem::model names the entry; mlp_tile.hpp), or nullptr
  const int* tile_order;        // workgroup -> tile (nullptr: the identity).  With several models in a plan the
                                // tiles of one model are dealt to ONE XCD (workgroups go round-robin over the eight
                                // XCDs, each with its own 4 MB L2): an XCD then streams one or two models' weights
                                // from its L2 instead of all of them
  // One-call control steps (ampc_mppi_run[_legacy]): x0 and u_out then point into host memory mapped into the
  // device's address space, and problem p's first action is followed by done_seq in done_flag[p] (host memory as
  // well, system-scope release) -- the host polls that word instead of waiting for copy packets and a stream
  // synchronisation.  nullptr everywhere else.
  unsigned long long* done_flag;
  unsigned long long done_seq;
};

// First action of problem p, control j (lanes j < nu of ONE wave call this together, mppi.py:166).
template <typename T>
__device__ __forceinline__ void publish_u(const MppiArgs<T>& args, int p, int nu, int j, T value) {
  args.u_out[p * nu + j] = value;
  if (args.done_flag) {
    __threadfence_system();            // the wave's stores to host memory are through before the flag goes up
    if (j == 0) __hip_atomic_store(args.done_flag + p, args.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <typename T> __device__ __forceinline__ T block_min(T v, T* scratch);
template <typename T> __device__ __forceinline__ T block_sum(T v, T* scratch);
template <typename T> __device__ __forceinline__ T wave_sum(T v);


// ---- finishing the softmin update inside the rollout launch (small problems) -------------------------
// A solve of a small problem is a few tens of microseconds; the dependent combine launch is seven of
// them.  So the LAST workgroup of a problem to publish its tile partials finishes the update itself.
// Cross-workgroup visibility WITHOUT agent-scope fences (a release fence is an L2 write-back on this
// eight-L2 part: measured 28 -> 61 us per c2 rollout in round 3):
//   * every published value is stored with an agent-scope atomic store: the AMDGPU memory model lowers it
//     to a store with sc1 = 1, which is written THROUGH the XCD's L2 to memory;
//   * each thread then waits for its own stores (s_waitcnt vmcnt(0): a write-through store is
//     acknowledged once it has left the L2 towards memory), the workgroup meets at a barrier, and one
//     thread takes a ticket with an agent-scope atomic add (performed at the memory side, where all
//     XCDs' atomics meet): a ticket value of tiles - 1 therefore happens after every other workgroup's
//     stores have been written through;
//   * the winner reads the published values with agent-scope atomic loads (sc1 = 1: they bypass its own
//     L2, which may hold stale lines of these addresses from the previous solve).
// No other data crosses workgroups.  tools/stress_fused_combine.py repeats one solve 2*10^5 times and
// checks that the result never changes (0 differences on c2 and the ARX workload).
// MEASURED (round 4): correct, but SLOWER than the combine launch it replaces -- every hop of the
// protocol (write-through acknowledgement, ticket, two rounds of bypassing loads) is a round trip to
// memory, not to an L2: the c2 rollout goes from 27.7 to 46.6 us where the combine launch costs 7.4.
// Off by default (AMPC_FUSED_COMBINE=1 enables it); kept as the measured answer to "fuse the combine".
template <typename T> __device__ __forceinline__ void st_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ T ld_agent(const T* p) {
  return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by every thread of a workgroup after its tile's partials were published with st_agent.
// Returns for all but the problem's last workgroup; that one computes a[t] += sum_w P_w[t] / sum_w s_w
// with every tile rescaled to the global minimum (mppi.py:110-118) -- tiles in index order, so the
// result does not depend on which workgroup happens to be last.  scratch: 2 * tiles + NTHR values of LDS.
template <typename T, int NTHR>
__device__ __forceinline__ void finish_update_if_last(const MppiArgs<T>& args, const MppiProblem<T>& pr, int p,
                                                      int tile_m, T* scratch) {
  __shared__ int ticket_s;
  const int tid = threadIdx.x, nu = args.mlp.nu, H = pr.H;
  const int tiles = (pr.N + tile_m - 1) / tile_m;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's published values are through
  __syncthreads();
  if (tid == 0) ticket_s = __hip_atomic_fetch_add(&args.tile_done[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket_s != tiles - 1) return;
  if (tid == 0) __hip_atomic_store(&args.tile_done[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const T* st = args.tile_stat + 2 * (size_t)pr.tile0;
  T* sc = scratch;                       // [tiles] tile minima, then tile weights
  T* ss = scratch + tiles;               // [tiles] tile weight sums
  T* part = ss + tiles;                  // [NTHR] partial sums of the second phase
  for (int i = tid; i < tiles; i += NTHR) { sc[i] = ld_agent(st + 2 * i); ss[i] = ld_agent(st + 2 * i + 1); }
  __syncthreads();
  T vmin = T(INFINITY);
  for (int i = 0; i < tiles; ++i) vmin = sc[i] < vmin ? sc[i] : vmin;
  __syncthreads();
  for (int i = tid; i < tiles; i += NTHR)
    sc[i] = (sc[i] < T(INFINITY)) ? exp(pr.neg_inv_lambda * (sc[i] - vmin)) : T(0);   // all-inf tile: weight 0
  __syncthreads();
  T ssum = T(0);
  for (int i = 0; i < tiles; ++i) ssum += sc[i] * ss[i];
  // sum_w scale_w P_w[t][j]: the tiles of an element are dealt to NG thread groups (all their loads in
  // flight together), the groups' partial sums are added in group order -- a fixed order either way
  const T* tp = args.tile_part + (size_t)pr.tile0 * args.hnu_stride;
  const int E = H * nu;
  for (int e0 = 0; e0 < E; e0 += NTHR) {
    const int ne = (E - e0) < NTHR ? (E - e0) : NTHR;        // elements of this round
    const int NG = NTHR / ne;                                // thread groups per element
    const int g = tid / ne, e = e0 + tid - g * ne;
    T acc = T(0);
    if (g < NG)
      for (int i = g; i < tiles; i += NG) acc += sc[i] * ld_agent(tp + (size_t)i * args.hnu_stride + e);
    __syncthreads();
    if (g < NG) part[tid] = acc;
    __syncthreads();
    if (tid < ne) {
      T tot = T(0);
      for (int k = 0; k < NG; ++k) tot += part[k * ne + tid];
      const int ee = e0 + tid, t = ee / nu, j = ee - t * nu;
      const int ts = (t + 1 < H) ? t + 1 : H - 1;
      const T a_new = args.act_in[pr.a_off + ts * nu + j] + tot / ssum;
      args.act_out[pr.a_off + ee] = a_new;
      if (t == 0) publish_u(args, p, nu, j, a_new * args.bounds[2 * nu + j]);
    }
  }
}

// SH: DynShape (everything read from `args` at run time) or a StaticShape (shapes.hpp) whose
// dimensions, strides and LDS offsets are compile-time constants -- see mlp_tile.hpp.
template <typename T, int NT, int MT, int W, typename SH = DynShape, bool WIDE = false>
__global__ __launch_bounds__(64 * W) void mppi_rollout_kernel(const MppiArgs<T> args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  using Net = TileNet<T, NT, MT, W, false, 0, SH, WIDE>;
  constexpr int M = 16 * MT, NTHR = 64 * W;
  constexpr int TPS = NTHR / M;                 // threads per sample (32..4), all in one wave
  constexpr int EPT = (16 + TPS - 1) / TPS;     // noise elements per thread (nu <= 16)
  // Indicator terms of the stage cost (MppiArgs::n_ind) are compiled into the run-time-shape instantiation
  // only: the shape-specialised kernels are the tuned hot path (a branch and a ballot more in their time loop
  // cost 2.5 % through scheduling alone), and a plan whose cost has such terms runs the run-time-shape kernels
  // (plan_build).  Per-problem models cost one scalar add per pointer: every instantiation takes them.
  // (the descriptor is initialised ONCE, as a constant: a struct that is assigned again and indexed with a
  //  run-time layer number is kept in scratch memory, and every weight pointer becomes a per-lane value)
  constexpr bool EXT = !SH::kStatic;
  // (wave-uniform by construction; said explicitly for the loaded index)
  const int tile = args.tile_order ? __builtin_amdgcn_readfirstlane(args.tile_order[blockIdx.x]) : (int)blockIdx.x;
  const int p = args.tile_prob[tile];
  const MppiProblem<T> pr = args.probs[p];
  const MlpDev<T> mlp = plan_model<SH, T>(args.mlp, [&] { return model_delta_of(args.model_delta, pr.model); });
  const TileLds L = SH::template fold_lds<T, M, W>(args.lds);
  const int tid = threadIdx.x;
  const int nx = mlp.nx, nu = mlp.nu, no = SH::kStatic ? SH::no : args.obs_dim;
  const int xs_ = L.xu_stride;
  const bool diag = args.cost_diag != 0, affine = args.cost_affine != 0;
  // cost block stride and the fixed-position LDS regions behind the tile map (plan_build lays
  // them out in this order: cost block + bounds, shifted sequence, [clipped noise, reduction])
  const int cost_stride = SH::kStatic ? cost_block_stride(SH::no, SH::nu) : args.cost_stride;
  const int lds_cost = SH::kStatic ? L.extra : args.lds_cost;
  const int lds_aseq = SH::kStatic ? round_up(L.extra + cost_stride + 3 * SH::nu, 4) : args.lds_aseq;

  const int first = (tile - pr.tile0) * M;
  const int H = pr.H, N = pr.N;

  // The two waves that share a SIMD (w and w + 4: a workgroup's waves are dealt to the SIMDs
  // cyclically) get different issue priorities, so they drift half a phase apart and one's
  // epilogue / LDS traffic overlaps the other's MFMAs (measured +0.7 % f64, +1 % f32).
  if (W == 8) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
  }
  Net net;
  net.init(mlp);

  T* aseq = lds + lds_aseq;              // [H][nu] shifted warm start
  T* cpar = lds + lds_cost;              // Q R F goal | lo hi scale
  const T* Qm = cpar;
  const T* Rm = Qm + no * no;
  const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;
  const T* lin = goal + no;                 // affine part of the stage cost, about `goal`
  const T* lint = lin + no;                 // ... of the terminal cost
  const T* blo = cpar + cost_stride;
  const T* bhi = blo + nu;
  const T* bsc = bhi + nu;
  T* xu = lds + L.xu;

  // ---- prologue: constants, shifted sequence, initial state ------------------------------
  tile_load_constants<T, W>(mlp, L, lds, M);
  for (int i = tid; i < cost_stride; i += NTHR)
    cpar[i] = args.costs_par[(size_t)pr.cost_idx * cost_stride + i];
  for (int i = tid; i < 3 * nu; i += NTHR) cpar[cost_stride + i] = args.bounds[i];
  for (int i = tid; i < H * nu; i += NTHR) {
    const int t = i / nu, j = i - t * nu;
    const int ts = (t + 1 < H) ? t + 1 : H - 1;  // a[:-1] = a[1:]; a[-1] = a[-2]
    aseq[i] = args.act_in[pr.a_off + ts * nu + j];
  }

  const int m = tid / TPS, r = tid % TPS;   // sample-in-tile, helper index
  const int n = first + m;
  const bool valid = n < N;
  const T* eps_row = args.eps + pr.eps_off + (size_t)(valid ? n : 0) * H * nu;
  T* epso = args.eps_out + pr.epso_off;

  T c_part = T(0), ca_part = T(0);
  T e_next[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int j = r + e * TPS;
    e_next[e] = (valid && j < nu) ? eps_row[j] : T(0);
  }
  __syncthreads();
  for (int i = tid; i < M * nx; i += NTHR) {
    const int row = i / nx, col = i - row * nx;
    xu[row * xs_ + col] = args.x0[p * nx + col];
  }

  // actions of step t: A = clip(eps + a), eps <- A - a, u = A * scale  (mppi.py:134-139)
  // (the thread's bounds / scale / R weight are loop invariants kept in registers: every LDS
  // read and every VALU instruction of this lambda sits between the MFMAs of a time step)
  T lo_r[EPT], hi_r[EPT], sc_r[EPT], rd_r[EPT];
  auto actions = [&](int t) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int j = r + e * TPS;
      if (j < nu) {
        const T a = aseq[t * nu + j];
        T A = e_next[e] + a;
        // np.minimum(hi, np.maximum(lo, .)) (mppi.py:135) -- by comparison, so that a NaN warm
        // start or noise value poisons the sample's cost as it does there (fmin/fmax drop NaNs)
        A = A < lo_r[e] ? lo_r[e] : A;
        A = A > hi_r[e] ? hi_r[e] : A;
        const T ec = A - a;
        if (valid && args.write_eps_out) epso[((size_t)t * N + n) * nu + j] = ec;
        if (args.lds_eps >= 0) lds[args.lds_eps + (t * M + m) * nu + j] = ec;
        ca_part += A * ec;
        const T u = A * sc_r[e];
        xu[m * xs_ + nx + j] = u;
        if (diag) c_part += rd_r[e] * u * u;          // diagonal R: the term is thread-local
        if (t + 1 < H) e_next[e] = valid ? eps_row[(t + 1) * nu + j] : T(0);
      }
    }
  };
#pragma unroll
  for (int e = 0; e < EPT; ++e) {          // (cpar was published by the barrier above)
    const int j = r + e * TPS;
    lo_r[e] = j < nu ? blo[j] : T(0);
    hi_r[e] = j < nu ? bhi[j] : T(0);
    sc_r[e] = j < nu ? bsc[j] : T(0);
    rd_r[e] = j < nu ? Rm[j * nu + j] : T(0);
  }
  actions(0);
  __syncthreads();
  // Diagonal costs are accumulated where the values are produced (actions / state update), so
  // the time loop has no separate cost phase; the stage cost of x_0 is added here.
  if (diag) c_part += quad_rows<T>(Qm, xu + m * xs_, goal, no, r, TPS, true);
  // affine part: lin'(x_t - goal) is added where the quadratic part is; the constant c0 is charged
  // for the H stage costs at once
  if (affine) {
    if (diag) c_part += affine_rows<T>(lin, xu + m * xs_, goal, no, r, TPS, T(0));
    if (r == 0) c_part += T(H) * lint[no];
  }

  AMPC_PROBE_KERNEL_BEGIN(net.probe);
  // the (at most ceil(nx / TPS)) state columns this thread updates: goal and diagonal Q weight
  constexpr int XPT = ((WIDE ? 64 : 32) + TPS - 1) / TPS;
  T qd_r[XPT], gl_r[XPT], ql_r[XPT];
#pragma unroll
  for (int e = 0; e < XPT; ++e) {
    const int i = r + e * TPS;
    qd_r[e] = (diag && i < no) ? Qm[i * no + i] : T(0);
    gl_r[e] = i < no ? goal[i] : T(0);
    ql_r[e] = (diag && i < no) ? lin[i] : T(0);       // (zero without an affine part)
  }
  for (int t = 0; t < H; ++t) {
    AMPC_PROBE_STEP(net.probe, t == 5);
    AMPC_MARK(0);
    // ---- stage cost of (x_t, u_t): partial per thread, reduced once after the loop ------------
    if constexpr (EXT)
      if (args.n_ind) c_part += indicator_rows<T>(args.ind_tab, args.n_ind, xu + m * xs_, 1, no, r, TPS);   // (x_t)
    if (!diag && !Probe::no_cost) {
      c_part += quad_rows<T>(Qm, xu + m * xs_, goal, no, r, TPS, false);
      c_part += quad_rows<T>(Rm, xu + m * xs_ + nx, nullptr, nu, r, TPS, false);
      if (affine) c_part += affine_rows<T>(lin, xu + m * xs_, goal, no, r, TPS, T(0));
    }
    AMPC_MARK(1);
    // ---- dynamics: x <- x + net'([x,u]) -------------------------------------------------------
    // The next step's actions do not depend on this step's output: they are formed while the
    // output layer's MFMAs drain.  (The control columns of xu were last read by layer 0, several
    // barriers ago; the dense-cost path reads them above, before run.)
    net.run_side(mlp, L, lds, [&] { if (t + 1 < H) actions(t + 1); });
#pragma unroll
    for (int e = 0; e < XPT; ++e) {
      const int i = r + e * TPS;
      if (i < nx) {
        const T xn = xu[m * xs_ + i] + Net::output(mlp, L, lds, m, i);
        xu[m * xs_ + i] = xn;
        if (diag && i < no && t + 1 < H) {     // stage cost of x_{t+1} (x_H only pays the terminal cost)
          const T d = xn - gl_r[e];
          c_part += (qd_r[e] * d + ql_r[e]) * d;     // (x-g) q (x-g) + lin (x-g): two FMAs either way
        }
      }
    }
    AMPC_MARK(12);
    AMPC_MARK(10);
    lds_barrier();
    AMPC_MARK(11);
  }

  AMPC_PROBE_KERNEL_END();
  // ---- epilogue: terminal cost, reduce the TPS partials, write ---------------------------------
  T term = quad_rows<T>(Fm, xu + m * xs_, goal, no, r, TPS, diag);
  if (affine) term += affine_rows<T>(lint, xu + m * xs_, goal, no, r, TPS, lint[no + 1]);
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

  // ---- fused softmin update, tile part (mppi.py:110-118): with the tile's own minimum m_w as the
  // reference point, S_m = exp(-(c_m - m_w)/lmda); the tile publishes m_w, sum_m S_m and
  // sum_m S_m eps[t][m][j].  mppi_combine_kernel rescales the tiles to the global minimum.
  T* cred = lds + args.lds_red;
  T* sred = cred + M;
  if (r == 0) cred[m] = valid ? c : T(INFINITY);
  __syncthreads();
  T mw = cred[0];
  for (int i = 1; i < M; ++i) mw = cred[i] < mw ? cred[i] : mw;
  // A tile whose every sample has cost +inf (a diverged rollout: the cost overflows before the
  // state does) has mw = inf and would form exp(-(inf - inf)/lmda) = NaN.  Such samples carry
  // weight 0 in the reference's softmin (mppi.py:113-116) as long as one finite cost exists
  // anywhere: the tile publishes zero sums and the combine kernel skips it.
  const bool dead_tile = !(mw < T(INFINITY));
  if (tid < M)
    sred[tid] = (first + tid < N && !dead_tile) ? exp(pr.neg_inv_lambda * (cred[tid] - mw)) : T(0);
  __syncthreads();
  const T* el = lds + args.lds_eps;
  T* tp = args.tile_part + (size_t)tile * args.hnu_stride;
  for (int e = tid; e < H * nu; e += NTHR) {
    const int t = e / nu, j = e - t * nu;
    T s = T(0);
    for (int i = 0; i < M; ++i) s += sred[i] * el[(t * M + i) * nu + j];
    tp[e] = s;
  }
  if (tid == 0) {
    T ss = T(0);
    for (int i = 0; i < M; ++i) ss += sred[i];
    args.tile_stat[2 * tile] = mw;
    args.tile_stat[2 * tile + 1] = ss;
  }
}

// Second half of the fused update.  grid = (max_h, B): block (t, p) rescales every tile's partial
// sums for step t from the tile minimum to the global minimum and finishes
// a[t] += (sum_tiles scale_w P_w[t]) / (sum_tiles scale_w s_w).  Each thread owns a strided set of
// tiles (its nu values of a tile row are contiguous), then one shuffle/LDS reduction per block.
constexpr int kMaxNu = 16;

// Noise one solve ahead (ampc_mppi_plan::eps_next): blocks x >= max_h of the same launch form problem p's
// Philox noise of the NEXT stream index -- the generator kernel's values (philox_normal_batch_kernel), thread
// for thread -- while blocks x < max_h finish this solve's update.
// (rng_kernels.hpp, which includes this file for MppiProblem)
template <typename T>
__device__ __forceinline__ void philox_normal_pair(T* __restrict__ out, long long count, T scale, uint64_t seed,
                                                   uint64_t stream, uint32_t id, long long pair);
template <typename T> struct NoiseAhead {
  T* eps;                        // nullptr: nothing to generate
  unsigned long long seed, stream;
};

template <typename T>
__global__ __launch_bounds__(kWG) void mppi_combine_kernel(const MppiArgs<T> args, int tile_m,
                                                           const NoiseAhead<T> ahead) {
  __shared__ T scratch[kWaves];
  __shared__ T red[kWaves][kMaxNu];
  const int p = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const MppiProblem<T> pr = args.probs[p];
  if (t >= args.max_h) {
    if (ahead.eps)
      philox_normal_pair<T>(ahead.eps + pr.eps_off, (long long)pr.N * pr.H * args.mlp.nu, pr.sqrt_sigma, ahead.seed,
                            ahead.stream, pr.noise_id, (long long)(t - args.max_h) * kWG + tid);
    return;
  }
  if (t >= pr.H) return;
  const int nu = args.mlp.nu, H = pr.H;
  const int tiles = (pr.N + tile_m - 1) / tile_m;
  const T* st = args.tile_stat + 2 * (size_t)pr.tile0;
  T vmin = st[0];
  for (int i = tid; i < tiles; i += kWG) vmin = st[2 * i] < vmin ? st[2 * i] : vmin;
  vmin = block_min(vmin, scratch);
  const T* tp = args.tile_part + (size_t)pr.tile0 * args.hnu_stride + t * nu;
  T ssum = T(0);
  T acc[kMaxNu];
#pragma unroll
  for (int j = 0; j < kMaxNu; ++j) acc[j] = T(0);
  for (int i = tid; i < tiles; i += kWG) {
    if (!(st[2 * i] < T(INFINITY))) continue;      // all-inf tile: weight 0 (its sums are zero too)
    const T sc = exp(pr.neg_inv_lambda * (st[2 * i] - vmin));
    ssum += sc * st[2 * i + 1];
    const T* row = tp + (size_t)i * args.hnu_stride;
#pragma unroll
    for (int j = 0; j < kMaxNu; ++j)
      if (j < nu) acc[j] += sc * row[j];
  }
  ssum = block_sum(ssum, scratch);
#pragma unroll
  for (int j = 0; j < kMaxNu; ++j)
    if (j < nu) {
      const T s = wave_sum(acc[j]);
      if ((tid & 63) == 0) red[tid >> 6][j] = s;
    }
  __syncthreads();
  if (tid < nu) {
    T s = red[0][tid];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[w][tid];
    const int ts = (t + 1 < H) ? t + 1 : H - 1;
    const T a_new = args.act_in[pr.a_off + ts * nu + tid] + s / ssum;
    args.act_out[pr.a_off + t * nu + tid] = a_new;
    if (t == 0) publish_u(args, p, nu, tid, a_new * args.bounds[2 * nu + tid]);
  }
}

// ---- block-wide reductions built on wave shuffles ---------------------------------------------
template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { T o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <typename T> __device__ __forceinline__ T block_min(T v, T* scratch) {
  v = wave_min(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  T o = scratch[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) o = scratch[w] < o ? scratch[w] : o;
  return o;
}
template <typename T> __device__ __forceinline__ T block_sum(T v, T* scratch) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  T o = scratch[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) o += scratch[w];
  return o;
}

// grid = (max_h, B).  Block (t, p) recomputes the softmin normaliser (N reads, L2 resident),
// keeps the unnormalised weights S_n = exp(-(c_n - min c)/lmda) in LDS, and produces
// act_out[p][t][:] = a_shift[t] + (sum_n S_n eps[t][n][:]) / sum_n S_n.  The [N][nu] noise slab of
// step t is read as one flat, fully coalesced stream: only the first (256/nu)*nu threads take part,
// so a thread's control index j = tid % nu is fixed and its partial sum is a scalar.
// Block t == 0 also writes the control to apply.
constexpr int kUpdateMaxN = 8192;   // weights kept in LDS up to this many samples (64 KB f64)

template <typename T>
__global__ __launch_bounds__(kWG) void mppi_update_kernel(const MppiArgs<T> args, const NoiseAhead<T> ahead) {
  extern __shared__ __attribute__((aligned(16))) unsigned char upd_smem[];
  T* wts = reinterpret_cast<T*>(upd_smem);          // [min(N, kUpdateMaxN)] weights, then scratch
  const int p = blockIdx.y, t = blockIdx.x;
  const MppiProblem<T> pr = args.probs[p];
  if (t >= args.max_h) {                            // (noise one solve ahead, as in mppi_combine_kernel)
    if (ahead.eps)
      philox_normal_pair<T>(ahead.eps + pr.eps_off, (long long)pr.N * pr.H * args.mlp.nu, pr.sqrt_sigma, ahead.seed,
                            ahead.stream, pr.noise_id, (long long)(t - args.max_h) * kWG + threadIdx.x);
    return;
  }
  if (t >= pr.H) return;
  const int N = pr.N, nu = args.mlp.nu, tid = threadIdx.x;
  const bool cached = N <= kUpdateMaxN;
  T* scratch = wts + (cached ? N : 0);              // kWaves reduction slots + kWG partials
  T* part = scratch + kWaves;
  const T* c = args.costs + pr.cost_off;
  // this thread's first U noise values are requested BEFORE the cost reductions: their global-memory
  // round trip overlaps the costs' instead of following the two block reductions
  const int active = (kWG / nu) * nu;               // threads with a fixed j = tid % nu
  const T* e = args.eps_out + pr.epso_off + (size_t)t * N * nu;
  constexpr int U = 8;                               // independent loads in flight per thread
  const int total = N * nu;
  T pre[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int idx = tid + k * active;
    pre[k] = (tid < active && idx < total) ? e[idx] : T(0);
  }

  T vmin = c[0];
  for (int n = tid; n < N; n += kWG) vmin = c[n] < vmin ? c[n] : vmin;
  vmin = block_min(vmin, scratch);
  T ssum = T(0);
  for (int n = tid; n < N; n += kWG) {
    const T s = exp(pr.neg_inv_lambda * (c[n] - vmin));
    if (cached) wts[n] = s;
    ssum += s;
  }
  ssum = block_sum(ssum, scratch);   // (barriers inside also publish wts)

  T acc = T(0);
  if (tid < active) {
    int n = tid / nu;
    const int dn = active / nu;
    int i = tid;
    // (the prefetched values first, in index order -- the same order of additions as the loops below)
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (i < total) {
        const T wgt = cached ? wts[n] : exp(pr.neg_inv_lambda * (c[n] - vmin));
        acc += wgt * pre[k];
        i += active; n += dn;
      }
    }
    for (; i + (U - 1) * active < total; i += U * active, n += U * dn) {
      T ev[U];
#pragma unroll
      for (int k = 0; k < U; ++k) ev[k] = e[i + k * active];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int nn = n + k * dn;
        const T wgt = cached ? wts[nn] : exp(pr.neg_inv_lambda * (c[nn] - vmin));
        acc += wgt * ev[k];
      }
    }
    for (; i < total; i += active, n += dn) {
      const T wgt = cached ? wts[n] : exp(pr.neg_inv_lambda * (c[n] - vmin));
      acc += wgt * e[i];
    }
  }
  part[tid] = tid < active ? acc : T(0);
  __syncthreads();
  if (tid < nu) {
    T s = T(0);
    for (int i = tid; i < active; i += nu) s += part[i];
    const int ts = (t + 1 < pr.H) ? t + 1 : pr.H - 1;
    const T a_new = args.act_in[pr.a_off + ts * nu + tid] + s / ssum;
    args.act_out[pr.a_off + t * nu + tid] = a_new;
    if (t == 0) publish_u(args, p, nu, tid, a_new * args.bounds[2 * nu + tid]);
  }
}

// Adds the reference-mode terminal scalar into the stored per-sample costs (only needed when the
// caller downloads them; the softmin weights are invariant to a constant shift).
template <typename T>
__global__ void mppi_finalize_costs_kernel(const MppiArgs<T> args) {
  const int p = blockIdx.y;
  const MppiProblem<T> pr = args.probs[p];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < pr.N) args.costs[pr.cost_off + n] += args.term_last[p];
}

// Closed-loop bookkeeping after one control step of every problem: append the applied control and
// the surrogate's next observation to the trajectory buffers and make it the next solve's x0.
//   traj_obs [B][T+1][nx], traj_ctrls [B][T+1][nu] (row T of ctrls stays zero, as simulate()
//   appends a zero control row, utils/simulation.py:59-61)
template <typename T>
__global__ void closed_loop_record_kernel(const T* __restrict__ x_next, const T* __restrict__ u,
                                          T* __restrict__ x0, T* __restrict__ traj_obs,
                                          T* __restrict__ traj_ctrls, int B, int nx, int nu, int T1,
                                          int step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * nx) {
    const int p = i / nx, c = i - p * nx;
    const T v = x_next[i];
    x0[i] = v;
    traj_obs[((size_t)p * T1 + step + 1) * nx + c] = v;
  }
  if (i < B * nu) {
    const int p = i / nu, c = i - p * nu;
    traj_ctrls[((size_t)p * T1 + step) * nu + c] = u[i];
  }
}

// Controller state of a lifted model from the simulated observation (ampc_mppi_plan_set_state_lift):
// Koopman.update_state (koopman.py:166-168) rebuilds the model state from EVERY new observation,
//   x0 = [f_0(o_0 .. o_{no-1}), f_1(o_0 ..), ...]   (basis-major, koopman.py:105-122)
// with f_k one of  0 identity  1 o ** p (integer p)  2 sin(p o)  3 cos(p o);  prog[k] = (kind, p).
// sim [B][snx]: the simulation model's state, whose first `no` entries are the observation.
template <typename T>
__global__ void state_lift_kernel(const T* __restrict__ sim, T* __restrict__ x0, const T* __restrict__ prog,
                                  int B, int snx, int no, int n_basis) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_basis * no) return;
  const int b = i / (n_basis * no), e = i - b * n_basis * no, f = e / no, j = e - f * no;
  const T o = sim[(size_t)b * snx + j];
  const int kind = (int)prog[2 * f];
  const T par = prog[2 * f + 1];
  T v = o;
  if (kind == 1) {
    // o ** p as the host computes it (Koopman's basis, koopman.py:105-122: numpy power = x * x for p = 2,
    // the C library's pow() above): the product is carried as a double-double and rounded ONCE, which is
    // pow()'s result except where the exact value lies within ~1e-16 relative of a rounding boundary --
    // a chain of rounded multiplications would be off by an ulp or two for p >= 3
    const int pw = (int)par;
    double hi = pw >= 1 ? (double)o : 1.0, lo = 0.0;
    for (int k = 1; k < pw; ++k) {
      const double ph = hi * (double)o;
      const double pe = fma(hi, (double)o, -ph) + lo * (double)o;
      hi = ph + pe;
      lo = pe - (hi - ph);
    }
    v = (T)hi;
  } else if (kind == 2) {
    v = sin(par * o);
  } else if (kind == 3) {
    v = cos(par * o);
  }
  x0[(size_t)b * (n_basis * no) + e] = v;
}

}  // namespace ampc
