// mppi_rollout4.hpp -- MPPI rollout on FOUR-row MFMA tiles (f64, gfx950): the small-problem kernel.
//
// mppi_rollout_kernel gives a workgroup 16*MT samples: a v_mfma_f64_16x16x4 occupies a SIMD's
// matrix pipe for 64 cycles whether its 16 rows are used or not, and a rollout is a chain of H
// dependent network evaluations, so a solve with few samples (BASELINE config 2: 1024 samples =
// 64 sixteen-row tiles on 256 CUs) runs on a quarter of the chip at the full per-step latency of
// a 16-row tile (~3.2 k cycles per step for a 2 x 64 network, 1.35 k of them matrix pipe).  This
// kernel evaluates FOUR samples per workgroup on v_mfma_f64_4x4x4_4b (16 cycles, the same flop
// rate): config 2 becomes 256 workgroups, one per CU, each with a quarter of the matrix-pipe time
// per step.  The host picks it when the sixteen-row tiles would leave most CUs idle (plan_build).
//
// What it computes is mppi_rollout_kernel's contract exactly (mppi.py:110-152; see mppi_kernels.hpp):
// clipped actions, stage + terminal costs, the per-tile softmin partials of the fused update.
//
// Layout.  Four waves; wave w owns sample row w between the network evaluations (actions, state
// update, costs: 64 lanes per sample) and hidden columns [16 NT w, 16 NT (w+1)) inside them --
// the N-split four-wave weight packing the four-row line search uses (MlpDev::w4, ilqr_ls4.hpp;
// operand layout there).  All weight fragments of the wave stay in registers for the whole
// rollout (NT = 1: 12 + 16 (NH-1) + 8 values per lane), activations go through LDS ([4][hpad+1]).
#pragma once
#include "mppi_kernels.hpp"
#include "rng_kernels.hpp"

namespace ampc {

struct Q4Lds {
  int xu, xs, act0, act1, as, part, bias, cpar, aseq, eps, useq, red, total;
};
__host__ __device__ constexpr Q4Lds make_q4_lds(int nu, int k1p, int nxp, int hpad, int n_hidden,
                                                int cost_stride, int max_h) {
  Q4Lds L{};
  int o = 0;
  L.xs = k1p + 1; L.as = hpad + 1;
  L.xu = o; o += 4 * 4 * L.xs;                   // (one [4][xs] copy per wave when FULL, see the kernel)
  L.act0 = o; o += 4 * L.as;
  L.act1 = o; o += 4 * L.as;
  L.part = o; o += 4 * 4 * nxp;
  L.bias = o; o += n_hidden * hpad + nxp;
  L.cpar = o; o += cost_stride + 3 * nu;
  L.aseq = o; o += max_h * nu;
  L.eps = o; o += max_h * 4 * nu;
  L.useq = o; o += max_h * 4 * nu;
  L.red = o; o += 8;
  L.total = (o + 3) / 4 * 4;
  return L;
}

// Shapes the kernel is instantiated for: hpad = 64 NT; every fragment register-resident.
__host__ __device__ constexpr bool q4_supported(int hpad, int n_hidden, int nxp, int k1p) {
  return ((hpad == 64 && n_hidden >= 1 && n_hidden <= 4) || (hpad == 128 && n_hidden >= 1 && n_hidden <= 2)) &&
         nxp <= 32 && k1p <= 48;
}

// SH: DynShape, or a StaticShape whose dimensions and activation are compile-time constants.  The
// run-time-shape loop body is dominated by what the shape decides (predicated first-layer k-steps,
// the activation switch, index arithmetic): 3.7 k cycles per step against 1.7 k specialised.
// ONE: the network has one 16-column output tile (nx <= 16) -- decides the output-layer scheme below,
// so it is a template parameter of the run-time-shape kernel too: both versions of a shape then sum
// in the same order and give the same bits.
template <int NT, int NH, typename SH = DynShape, bool ONE = (SH::kStatic && SH::nxp <= 16)>
__global__ __launch_bounds__(256) void mppi_rollout4_kernel(const MppiArgs<double> args) {
  using T = double;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int W = 4, ROWS = 4, NTHR = 256, HP = 64 * NT, KSH = HP / 4, KSW = KSH / W, KS0MAX = 12;
  constexpr bool FULL = KSH <= 16 || (ONE && KSH <= 32);     // output-layer scheme, see below
  // FULL: every wave ends a step with the new state of all four rows in registers and keeps its OWN
  // copy of the first layer's operand [x | u] in LDS -- nothing crosses waves between the output
  // layer and the next first layer, so the barrier there goes (it stays for an odd number of hidden
  // layers, where the first layer would overwrite the activations a slower wave is still reading).
  constexpr bool BAR_A = !FULL || (NH & 1);
  constexpr bool EXT = !SH::kStatic;           // (indicator cost terms: run-time shapes only, as in mppi_rollout_kernel)
  const int p = args.tile_prob[blockIdx.x];
  const MppiProblem<T> pr = args.probs[p];
  const MlpDev<T> mlp = plan_model<SH, T>(args.mlp, [&] { return model_delta_of(args.model_delta, pr.model); });
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = mlp.nx, nu = mlp.nu, no = SH::kStatic ? SH::no : args.obs_dim;
  const int nxp = mlp.nxp, tiles = nxp / 16, ks0 = mlp.k1p / 4;
  const int cost_stride = SH::kStatic ? cost_block_stride(SH::no, SH::nu) : args.cost_stride;
  const bool diag = args.cost_diag != 0, affine = args.cost_affine != 0;
  const Q4Lds L = make_q4_lds(nu, mlp.k1p, nxp, HP, NH, cost_stride, args.max_h);
  const int xs = L.xs, as = L.as;
  T* xu = lds + L.xu + (FULL ? w * 4 * xs : 0); T* part = lds + L.part; T* bias = lds + L.bias; T* cpar = lds + L.cpar;
  T* aseq = lds + L.aseq; T* el = lds + L.eps;
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu; const T* goal = Fm + no * no;
  const T* lin = goal + no; const T* lint = lin + no;      // affine part (stage / terminal), c0 c1 behind
  const T* blo = cpar + cost_stride; const T* bhi = blo + nu; const T* bsc = bhi + nu;

  const int first = (blockIdx.x - pr.tile0) * ROWS;
  const int H = pr.H, N = pr.N;

  // ---- resident weight fragments (MlpDev::w4: [wave][k-step][chunk][lane][cw], api.cpp build_model)
  constexpr int CW = NT % 2 == 0 ? 2 : 1, CH = NT / CW;
  auto frag = [&](const T* base, int kstep, T (&b)[NT]) {      // one k-step of this wave's stream
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < CW; ++e) b[c * CW + e] = base[(((size_t)kstep * CH + c) * 64 + lane) * CW + e];
  };
  T w0[KS0MAX][NT];
  {
    const T* b0 = mlp.W4(0) + (size_t)w * ks0 * 64 * NT;
#pragma unroll
    for (int ks = 0; ks < KS0MAX; ++ks) {
      if (ks < ks0) frag(b0, ks, w0[ks]);
      else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w0[ks][nt] = T(0);
      }
    }
  }
  T wh[NH > 1 ? NH - 1 : 1][KSH][NT];
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    const T* bl = mlp.W4(l) + (size_t)w * KSH * 64 * NT;
#pragma unroll
    for (int ks = 0; ks < KSH; ++ks) frag(bl, ks, wh[l - 1][ks]);
  }
  // Output layer.  FULL: every wave evaluates the whole layer (all KSH k-steps) for the four rows --
  // redundant matrix-pipe work, but the wave that owns a row then holds its outputs in registers and
  // updates the state directly: no partial sums through LDS, one barrier less per step.  Otherwise
  // (long K and two output tiles) the k-steps are split over the waves as in ilqr_ls4_kernel.
  constexpr int KSO = FULL ? KSH : KSW;
  T wout[KSO][2];
  {
    const T* wl = mlp.W4(NH) + ((size_t)(FULL ? 0 : w * KSW) * 64 + lane) * tiles;
#pragma unroll
    for (int ks = 0; ks < KSO; ++ks) {
      wout[ks][0] = wl[(size_t)ks * 64 * tiles];
      wout[ks][1] = tiles > 1 ? wl[(size_t)ks * 64 * tiles + 1] : T(0);
    }
  }

  const int m = w, r = lane;                   // sample row of this wave, helper index within it
  const int n = first + m;
  const bool valid = n < N;
  const T* eps_row = args.eps + pr.eps_off + (size_t)(valid ? n : 0) * H * nu;
  // The row's noise: read from the plan's buffer (numpy's stream, uploaded or generated there), or --
  // device Philox noise -- formed right here: value e of the problem's draw is a pure function of
  // (seed, stream, noise id, e) (rng_kernels.hpp), so the generator's launch and the HBM round trip of
  // the buffer go (the stream position is the same element index the generator kernel would have used).
  const bool inl = args.eps_inline != 0;
  auto noise = [&](int i) -> T {
    if (!valid) return T(0);
    if (inl) return philox_normal_elem<T>(pr.sqrt_sigma, args.eps_seed, args.eps_stream, pr.noise_id,
                                          (long long)n * H * nu + i);
    return eps_row[i];
  };
  const T e_first = lane < H * nu ? noise(lane) : T(0);   // (in flight during the staging below)

  // ---- prologue: constants, shifted sequence, initial state ---------------------------------
  for (int l = 0; l < NH; ++l)
    for (int i = tid; i < HP; i += NTHR) bias[l * HP + i] = mlp.B(l)[i];
  for (int i = tid; i < nxp; i += NTHR) bias[NH * HP + i] = mlp.B(NH)[i];
  for (int i = tid; i < cost_stride; i += NTHR)
    cpar[i] = args.costs_par[(size_t)pr.cost_idx * cost_stride + i];
  for (int i = tid; i < 3 * nu; i += NTHR) cpar[cost_stride + i] = args.bounds[i];
  for (int i = tid; i < H * nu; i += NTHR) {
    const int t = i / nu, j = i - t * nu;
    const int ts = (t + 1 < H) ? t + 1 : H - 1;  // a[:-1] = a[1:]; a[-1] = a[-2]
    aseq[i] = args.act_in[pr.a_off + ts * nu + j];
  }
  for (int i = FULL ? lane : tid; i < ROWS * xs; i += FULL ? 64 : NTHR) {
    const int row = i / xs, col = i - row * xs;
    xu[i] = col < nx ? args.x0[p * nx + col] : T(0);
  }

  T* epso = args.eps_out + pr.epso_off;
  const bool keep = args.lds_eps >= 0;

  T c_part = T(0), ca_part = T(0);
  T* ul = lds + L.useq;
  __syncthreads();
  // ---- all actions of the row, before the time loop: A = clip(eps + a), eps <- A - a, u = A * scale
  // (mppi.py:134-139) do not depend on the state, so the serial chain of the rollout is left with
  // the network alone.  One (t, j) element per lane; the sample's noise row is read once, coalesced
  // (a load issued one step ahead would not be ahead enough: a step here is shorter than an HBM
  // round trip).
  for (int i = lane; i < H * nu; i += 64) {
    const int t = i / nu, j = i - t * nu;
    const T a = aseq[i];
    T A = (i < 64 ? e_first : noise(i)) + a;
    A = A < blo[j] ? blo[j] : A;                 // by comparison: a NaN poisons the cost as in numpy
    A = A > bhi[j] ? bhi[j] : A;
    const T ec = A - a;
    if (valid && args.write_eps_out) epso[((size_t)t * N + n) * nu + j] = ec;
    el[(t * ROWS + m) * nu + j] = ec;
    ca_part += A * ec;
    const T u = A * bsc[j];
    ul[(t * ROWS + m) * nu + j] = u;
    if (diag) c_part += Rm[j * nu + j] * u * u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's own LDS writes, read below)
  if (!diag)
    for (int t = lane; t < H; t += 64) c_part += quad_rows<T>(Rm, ul + (t * ROWS + m) * nu, nullptr, nu, 0, 1, false);
  // where lane r's element of a step's controls [4][nu] goes in [x | u] (FULL: all four rows per wave)
  const int cdst = (r / nu) * xs + nx + (r - (r / nu) * nu);
  if (FULL) {
    __syncthreads();                             // (the other rows' controls come from their waves)
    if (r < ROWS * nu) xu[cdst] = ul[r];
  } else if (r < nu) {
    xu[m * xs + nx + r] = ul[m * nu + r];
  }
  if (diag) c_part += quad_rows<T>(Qm, xu + m * xs, goal, no, r, 64, true);     // stage cost of x_0
  if (affine) {                                  // lin'(x_0 - goal); the constant for all H stage costs
    if (diag) c_part += affine_rows<T>(lin, xu + m * xs, goal, no, r, 64, T(0));
    if (r == 0) c_part += T(H) * lint[no];
  }

  const int arow = lane & 3, ak = lane >> 4, drow = lane >> 4, dcol = lane & 15;
  // per-lane constants of the state update: diagonal Q weight, goal, output bias of the lane's columns
  const T qd_r = (diag && r < no) ? Qm[r * no + r] : T(0), gl_r = r < no ? goal[r] : T(0);
  const T qd0 = (diag && dcol < no) ? Qm[dcol * no + dcol] : T(0), gl0 = dcol < no ? goal[dcol] : T(0);
  const T qd1 = (diag && 16 + dcol < no) ? Qm[(16 + dcol) * no + 16 + dcol] : T(0);
  const T ql_r = (diag && r < no) ? lin[r] : T(0), ql0 = (diag && dcol < no) ? lin[dcol] : T(0);
  const T ql1 = (diag && 16 + dcol < no) ? lin[16 + dcol] : T(0);
  const T gl1 = 16 + dcol < no ? goal[16 + dcol] : T(0);
  const T ob0 = bias[NH * HP + dcol], ob1 = tiles > 1 ? bias[NH * HP + 16 + dcol] : T(0);
  T xr0 = dcol < nx ? xu[drow * xs + dcol] : T(0), xr1 = 16 + dcol < nx ? xu[drow * xs + 16 + dcol] : T(0);
  ProbeWave<Probe::wave_time> probe;             // (timing-experiment builds only, tools/wavetime4.py)
  AMPC_PROBE_KERNEL_BEGIN(probe);
  // A layer's k-steps go round NA independent accumulators: a v_mfma_f64_4x4x4 occupies the pipe
  // for 16 cycles but its result is only back after ~4x that, so ONE accumulator chain runs at the
  // latency of the sixteen-row instruction (measured: the same 1.35 us per step as that kernel).
  constexpr int NA = 4;
  auto fold = [](const T (&a)[NA]) { return (a[0] + a[1]) + (a[2] + a[3]); };
  for (int t = 0; t < H; ++t) {
    AMPC_PROBE_STEP(probe, t == 5);
    AMPC_MARK(0);
    if (BAR_A) lds_barrier();                    // state and controls of step t are in xu
    AMPC_MARK(1);
    if constexpr (EXT)
      if (args.n_ind) c_part += indicator_rows<T>(args.ind_tab, args.n_ind, xu + m * xs, 1, no, r, 64);   // (x_t)
    if (!diag) {
      c_part += quad_rows<T>(Qm, xu + m * xs, goal, no, r, 64, false);
      if (affine) c_part += affine_rows<T>(lin, xu + m * xs, goal, no, r, 64, T(0));
    }
    // ---- layer 0
    T* ain = lds + L.act0;
    {
      T acc[NT][NA];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[nt][a] = T(0);
      const T* ap = xu + arow * xs + ak;
      T av0[KS0MAX];
#pragma unroll
      for (int ks = 0; ks < KS0MAX; ++ks)
        if (ks < ks0) av0[ks] = ap[4 * ks];
#pragma unroll
      for (int ks = 0; ks < KS0MAX; ++ks)
        if (ks < ks0) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][ks % NA] = mfma4(av0[ks], w0[ks][nt], acc[nt][ks % NA]);
        }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * (NT * w + nt) + dcol;
        ain[drow * as + col] = act_apply<T>(mlp.act, fold(acc[nt]) + bias[col]);
      }
    }
    AMPC_MARK(2);
    lds_barrier();
    AMPC_MARK(3);
    // ---- hidden -> hidden layers
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      T* aout = lds + ((l & 1) ? L.act1 : L.act0);
      T acc[NT][NA];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[nt][a] = T(0);
      const T* ap = ain + arow * as + ak;
      T av[KSH];
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) av[ks] = ap[4 * ks];
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][ks % NA] = mfma4(av[ks], wh[l - 1][ks][nt], acc[nt][ks % NA]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * (NT * w + nt) + dcol;
        aout[drow * as + col] = act_apply<T>(mlp.act, fold(acc[nt]) + bias[l * HP + col]);
      }
      ain = aout;
      AMPC_MARK(4);
      lds_barrier();
      AMPC_MARK(5);
    }
    // ---- output layer and x <- x + net([x, u]) on row w; the controls of step t + 1 move into xu
    // (its control columns were last read by layer 0, two barriers ago)
    if constexpr (FULL) {
      T o0[NA], o1[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) o0[a] = o1[a] = T(0);
      const T un = (t + 1 < H && r < ROWS * nu) ? ul[(t + 1) * ROWS * nu + r] : T(0);   // (in flight under the MFMAs)
      const T* ap = ain + arow * as + ak;
      T avo[KSO];
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) avo[ks] = ap[4 * ks];
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) {
        o0[ks % NA] = mfma4(avo[ks], wout[ks][0], o0[ks % NA]);
        if (tiles > 1) o1[ks % NA] = mfma4(avo[ks], wout[ks][1], o1[ks % NA]);
      }
      AMPC_MARK(6);
      if (t + 1 < H && r < ROWS * nu) xu[cdst] = un;
      AMPC_MARK(7);
      AMPC_MARK(8);
      // every lane keeps its (row drow, columns dcol / 16 + dcol) of the state in registers; the
      // costs of row w are accumulated by the lanes holding row w
      if (dcol < nx) {
        xr0 += fold(o0) + ob0;
        xu[drow * xs + dcol] = xr0;
        if (diag && drow == w && dcol < no && t + 1 < H) { const T d = xr0 - gl0; c_part += (qd0 * d + ql0) * d; }
      }
      if (tiles > 1 && 16 + dcol < nx) {
        xr1 += fold(o1) + ob1;
        xu[drow * xs + 16 + dcol] = xr1;
        if (diag && drow == w && 16 + dcol < no && t + 1 < H) { const T d = xr1 - gl1; c_part += (qd1 * d + ql1) * d; }
      }
    } else {
      T o0[NA], o1[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) o0[a] = o1[a] = T(0);
      const T* ap = ain + arow * as + 4 * (w * KSW) + ak;
      T avo[KSO];
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) avo[ks] = ap[4 * ks];
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) {
        o0[ks % NA] = mfma4(avo[ks], wout[ks][0], o0[ks % NA]);
        if (tiles > 1) o1[ks % NA] = mfma4(avo[ks], wout[ks][1], o1[ks % NA]);
      }
      part[(w * ROWS + drow) * nxp + dcol] = fold(o0);
      if (tiles > 1) part[(w * ROWS + drow) * nxp + 16 + dcol] = fold(o1);
      AMPC_MARK(6);
      if (t + 1 < H && r < nu) xu[m * xs + nx + r] = ul[((t + 1) * ROWS + m) * nu + r];
      AMPC_MARK(7);
      lds_barrier();
      AMPC_MARK(8);
      if (r < nx) {
        T s = bias[NH * HP + r];
#pragma unroll
        for (int ww = 0; ww < W; ++ww) s += part[(ww * ROWS + m) * nxp + r];
        const T xn = xu[m * xs + r] + s;
        xu[m * xs + r] = xn;
        if (diag && r < no && t + 1 < H) {       // stage cost of x_{t+1} (x_H only pays the terminal cost)
          const T d = xn - gl_r;
          c_part += (qd_r * d + ql_r) * d;
        }
      }
    }
    AMPC_MARK(9);
  }
  AMPC_PROBE_KERNEL_END();
  __syncthreads();

  // ---- epilogue: terminal cost, reduce the 64 partials of the row, write -----------------------
  T term = quad_rows<T>(Fm, xu + m * xs, goal, no, r, 64, diag);
  if (affine) term += affine_rows<T>(lint, xu + m * xs, goal, no, r, 64, lint[no + 1]);
  T c = c_part + pr.lam_over_sigma * ca_part;
  if (args.term_mode == 1) c += term;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    c += __shfl_xor(c, off);
    term += __shfl_xor(term, off);
  }
  if (r == 0 && valid) {
    args.costs[pr.cost_off + n] = c;
    if (n == N - 1) args.term_last[p] = term;
  }
  if (!keep) return;

  // ---- fused softmin update, tile part (as mppi_rollout_kernel; mppi_combine_kernel finishes)
  T* cred = lds + L.red;
  T* sred = cred + ROWS;
  if (r == 0) cred[m] = valid ? c : T(INFINITY);
  __syncthreads();
  T mw = cred[0];
  for (int i = 1; i < ROWS; ++i) mw = cred[i] < mw ? cred[i] : mw;
  const bool dead_tile = !(mw < T(INFINITY));
  if (tid < ROWS)
    sred[tid] = (first + tid < N && !dead_tile) ? exp(pr.neg_inv_lambda * (cred[tid] - mw)) : T(0);
  __syncthreads();
  T* tp = args.tile_part + (size_t)blockIdx.x * args.hnu_stride;
  const bool fuse = args.fused_combine != 0;
  for (int e = tid; e < H * nu; e += NTHR) {
    const int t = e / nu, j = e - t * nu;
    T s = T(0);
    for (int i = 0; i < ROWS; ++i) s += sred[i] * el[(t * ROWS + i) * nu + j];
    if (fuse) st_agent(tp + e, s);
    else tp[e] = s;
  }
  if (tid == 0) {
    T ss = T(0);
    for (int i = 0; i < ROWS; ++i) ss += sred[i];
    if (fuse) { st_agent(args.tile_stat + 2 * blockIdx.x, mw); st_agent(args.tile_stat + 2 * blockIdx.x + 1, ss); }
    else { args.tile_stat[2 * blockIdx.x] = mw; args.tile_stat[2 * blockIdx.x + 1] = ss; }
  }
  // the problem's last workgroup finishes the softmin update (mppi_kernels.hpp: no combine launch)
  if (fuse) finish_update_if_last<T, NTHR>(args, pr, p, ROWS, lds);
}

}  // namespace ampc
