// ilqr_ls4.hpp -- iLQR forward pass / line search on four-row MFMA tiles (f64, gfx950).
//
// The line search of one problem is a chain of H dependent MLP evaluations, so its time is the
// latency of one evaluation times H.  ilqr_iter_kernel evaluates all (<= 16) step sizes as the rows
// of one 16-row tile: 64 cycles of matrix pipe per v_mfma_f64_16x16x4, i.e. ~20 k cycles per time
// step for a 2 x 256 network on the one CU a problem occupies -- whether the tile holds ten
// candidates or one.  The reference accepts the FIRST step size that passes its test
// (ilqr.py:207-233) and that is almost always one of the first few, so this kernel evaluates the
// candidates four at a time on v_mfma_f64_4x4x4_4b (16 cycles, the same flop rate: a quarter of the
// matrix-pipe time per step) and only rolls out the next four when none of them was accepted.  The
// decisions, and therefore the results, are those of the reference's sequential loop.
//
// Operand layout (lane l of a wave; probed, see mfma4 in mlp_tile.hpp): A = act[row l%4][k = 4ks +
// l/16] (the same for the four blocks), B = W[k = 4ks + l/16][col 16g + l%16] -- exactly the
// fragment the 16x16x4 MFMA takes, so the packed weights of mlp_tile.hpp are used unchanged -- and
// D = out[row l/16][col 16g + l%16], one value per lane.
//
// First-layer and output-layer fragments stay in registers for the kernel's lifetime; hidden ->
// hidden layers are streamed from L2 through a ring of NB groups of G k-steps that runs NB-1 groups
// ahead of the MFMAs, across layer and time-step boundaries.
#pragma once
#include "ilqr_kernels.hpp"

namespace ampc {

struct Ls4Lds {
  int xu, xs, act0, act1, as, part, bias, Km, kv, ubar, xbar, cpar, blo, bhi, scal, lsobj, piv, total;
};
__host__ __device__ constexpr Ls4Lds make_ls4_lds(int nx, int nu, int k1p, int nxp, int hpad, int n_hidden,
                                                  int W, int cost_stride) {
  Ls4Lds L{};
  int o = 0;
  L.xs = k1p + 1; L.as = hpad + 1;
  L.xu = o; o += 4 * L.xs;
  L.act0 = o; o += 4 * L.as;
  L.act1 = o; o += 4 * L.as;
  L.part = o; o += W * 4 * nxp;
  L.bias = o; o += n_hidden * hpad + nxp;
  L.Km = o; o += nu * nx;
  L.kv = o; o += nu;
  L.ubar = o; o += nu;
  L.xbar = o; o += nx;
  L.cpar = o; o += cost_stride;
  L.blo = o; o += nu;
  L.bhi = o; o += nu;
  L.scal = o; o += 8;
  L.lsobj = o; o += 2 * kIlqrMaxLs;
  L.piv = o; o += 8;
  L.total = (o + 3) / 4 * 4;
  return L;
}

template <int NT, int W, typename SH = DynShape>
__global__ __launch_bounds__(64 * W) void ilqr_ls4_kernel(const IlqrArgs<double> args) {
  using T = double;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int NTHR = 64 * W, ROWS = 4, TPS = NTHR / ROWS;
  constexpr int HP = 16 * NT * W, KSH = HP / 4, KSW = KSH / W, KS0MAX = 12;
  constexpr int G = (KSH % 32 == 0) ? 8 : 4, NB = 4, NGH = KSH / G, D = NB - 1;
  static_assert(NGH % NB == 0 && D < NGH, "ring phase must repeat per layer");
  static_assert(W <= 8 && TPS % 64 == 0, "objective partials: one slot per wave");
  constexpr int NG8 = KSH / 8;
  constexpr bool OWNPACK = (NT == 2) && ((NG8 & (NG8 - 1)) == 0);   // api.cpp: own_first_packing()
  static_assert(!OWNPACK || G == 8, "rotated streams rotate by whole groups of 8 k-steps");
  const MlpDev<T> mlp = SH::template fold<T>(args.mlp);
  const int tid = threadIdx.x, p = blockIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = mlp.nx, nu = mlp.nu, no = SH::kStatic ? SH::no : args.obs_dim, H = args.H;
  const int Lh = mlp.n_hidden, nxp = mlp.nxp, tiles = nxp / 16, ks0 = mlp.k1p / 4;
  const int cost_stride = SH::kStatic ? round_up(2 * SH::no * SH::no + SH::nu * SH::nu + SH::no, 4) : args.cost_stride;
  const Ls4Lds L = make_ls4_lds(nx, nu, mlp.k1p, nxp, HP, Lh, W, cost_stride);
  T* xu = lds + L.xu; T* part = lds + L.part; T* bias = lds + L.bias;
  T* Km = lds + L.Km; T* kv = lds + L.kv; T* ubar = lds + L.ubar; T* xbar = lds + L.xbar;
  T* cpar = lds + L.cpar; T* blo = lds + L.blo; T* bhi = lds + L.bhi; T* scal = lds + L.scal;
  T* lsobj = lds + L.lsobj; int* piv = reinterpret_cast<int*>(lds + L.piv);
  const int xs = L.xs, as = L.as;
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;

  if (args.mode == 1 && args.active[p] == 0) {
    if (tid == 0) args.refresh[p] = 0;
    return;
  }
  if (args.mode == 1 && args.ric[(size_t)p * 4 + 3] != T(0)) return;   // singular Quu: retired by the sweep

  for (int l = 0; l < Lh; ++l)
    for (int i = tid; i < HP; i += NTHR) bias[l * HP + i] = mlp.b[l][i];
  for (int i = tid; i < nxp; i += NTHR) bias[Lh * HP + i] = mlp.b[Lh][i];
  for (int i = tid; i < 4 * xs; i += NTHR) xu[i] = T(0);
  for (int i = tid; i < cost_stride; i += NTHR)
    cpar[i] = args.costs_par[(size_t)args.cost_idx[p] * cost_stride + i];
  for (int i = tid; i < nu; i += NTHR) {
    blo[i] = args.bounded ? args.ubounds[i] : T(0);
    bhi[i] = args.bounded ? args.ubounds[nu + i] : T(0);
  }
  if (tid == 0 && args.mode == 1) {
    const T* rin = args.ric + (size_t)p * 4;
    scal[0] = rin[0]; scal[1] = rin[1]; scal[2] = rin[2];
  }

  // ---- resident fragments + the hidden-layer ring ------------------------------------------------
  const rsrc_t wr = weight_rsrc(mlp.wbase);
  const unsigned lo = (unsigned)lane * NT;
  T w0[KS0MAX][NT];
  {
    const unsigned s0 = (unsigned)(mlp.w[0] - mlp.wbase) + (unsigned)w * (unsigned)ks0 * 64u * NT;
#pragma unroll
    for (int ks = 0; ks < KS0MAX; ++ks) {
      if (ks < ks0) load_frag<T, NT>(wr, s0 + (unsigned)ks * 64u * NT, lo, w0[ks]);
      else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w0[ks][nt] = T(0);
      }
    }
  }
  T wout[KSW][2];
  {
    const T* wl = mlp.w[Lh] + ((size_t)w * KSW * 64 + lane) * tiles;
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      wout[ks][0] = wl[(size_t)ks * 64 * tiles];
      wout[ks][1] = tiles > 1 ? wl[(size_t)ks * 64 * tiles + 1] : T(0);
    }
  }
  T ring[NB][G][NT];
  auto slice_h = [&](int l) {
    return (unsigned)(mlp.w[l] - mlp.wbase) + (unsigned)w * (unsigned)KSH * 64u * NT;
  };
  if (Lh > 1) {
    const unsigned s1 = slice_h(1);
#pragma unroll
    for (int g = 0; g < D; ++g)
#pragma unroll
      for (int kk = 0; kk < G; ++kk) load_frag<T, NT>(wr, s1 + (unsigned)(g * G + kk) * 64u * NT, lo, ring[g][kk]);
  }
  // this lane's hidden biases live in LDS; its A-operand row / k offsets
  const int arow = lane & 3, ak = lane >> 4, drow = lane >> 4, dcol = lane & 15;

  const T* st = args.states + (size_t)p * (H + 1) * nx;
  T* stw = args.states + (size_t)p * (H + 1) * nx;
  T* ctw = args.ctrls + (size_t)p * H * nu;
  const T* Kg = args.Ks + (size_t)p * H * nu * nx;
  const T* kg = args.ks + (size_t)p * H * nu;
  T* lss = args.ls_states + (size_t)p * args.ls_n * (H + 1) * nx;
  T* lsc = args.ls_ctrls + (size_t)p * args.ls_n * H * nu;
  const int rows = args.mode == 0 ? 1 : args.ls_n;
  const int m = tid / TPS, r = tid % TPS;          // candidate row of this thread, helper index
  const bool cdiag = args.cost_diag != 0;
  // control law: `parts` threads per (row, control), interleaved over the state index
  const int parts = (8 * nu <= TPS) ? 8 : 4;
  const int ca = r / parts, cpart = r - ca * parts;

  constexpr int KR = (16 * 32 + NTHR - 1) / NTHR;
  T kreg[KR];
  T kvr = T(0), ubr = T(0), xbr = T(0);
  auto fetch_ls = [&](int t) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int idx = tid + k * NTHR;
      if (idx < nu * nx) kreg[k] = Kg[(size_t)t * nu * nx + idx];
    }
    if (tid < nu) { kvr = kg[(size_t)t * nu + tid]; ubr = ctw[(size_t)t * nu + tid]; }
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

  // acceptance state of the reference's sequential loop (thread 0)
  T best_obj = INFINITY;
  int best = -1, last = 0, decided = 0;
  __syncthreads();

  const int npass = (rows + ROWS - 1) / ROWS;
  for (int pass = 0; pass < npass; ++pass) {
    const int j = ROWS * pass + m;                 // this thread's candidate
    const bool live = j < rows;
    const T alpha = args.alphas[j < kIlqrMaxLs ? j : 0];
    T obj_part = T(0);
    for (int i = tid; i < ROWS * nx; i += NTHR) {
      const int row = i / nx, col = i - row * nx;
      xu[row * xs + col] = st[col];
    }
    if (args.mode == 1) { fetch_ls(0); commit_ls(); }
    __syncthreads();
    for (int t = 0; t < H; ++t) {
#ifdef AMPC_X_PHASETIME
      if (blockIdx.x == 7 && threadIdx.x == 0) g_phase_marks[63] = (args.mode == 1 && pass == 0 && t == H / 2) ? 1 : 0;
#endif
      AMPC_IMARK(40);
      if (args.mode == 1 && t + 1 < H) fetch_ls(t + 1);
      // ---- controls of this step (ilqr.py:196-205)
      if (args.mode == 0) {
        for (int a = r; a < nu; a += TPS) xu[m * xs + nx + a] = ctw[(size_t)t * nu + a];
      } else {
        T f = T(0);
        if (ca < nu)
          for (int b = cpart; b < nx; b += parts) f += Km[ca * nx + b] * (xu[m * xs + b] - xbar[b]);
        f += __shfl_xor(f, 1);
        f += __shfl_xor(f, 2);
        if (parts == 8) f += __shfl_xor(f, 4);
        if (ca < nu && cpart == 0) {
          T u = alpha * kv[ca] + ubar[ca] + f;
          if (args.bounded) { u = u < blo[ca] ? blo[ca] : u; u = u > bhi[ca] ? bhi[ca] : u; }
          if (live) lsc[((size_t)j * H + t) * nu + ca] = u;
          xu[m * xs + nx + ca] = u;
        }
        if (live)
          for (int a = r; a < nx; a += TPS) lss[((size_t)j * (H + 1) + t) * nx + a] = xu[m * xs + a];
      }
      AMPC_IMARK(41);
      lds_barrier();
      AMPC_IMARK(42);
      // ---- objective: dt * (stage costs)
      obj_part += args.dt * (quad_rows<T>(Qm, xu + m * xs, goal, no, r, TPS, cdiag) +
                             quad_rows<T>(Rm, xu + m * xs + nx, nullptr, nu, r, TPS, cdiag));
      AMPC_IMARK(43);
      // ---- layer 0
      T* ain = lds + L.act0;
      {
        T acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = T(0);
        const T* ap = xu + arow * xs + ak;
#pragma unroll
        for (int ks = 0; ks < KS0MAX; ++ks)
          if (ks < ks0) {
            const T a = ap[4 * ks];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(a, w0[ks][nt], acc[nt]);
          }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = 16 * (NT * w + nt) + dcol;
          ain[drow * as + col] = act_apply<T>(mlp.act, acc[nt] + bias[col]);
        }
      }
      AMPC_IMARK(44);
      lds_barrier();
      AMPC_IMARK(45);
      // ---- hidden -> hidden layers, streamed
      for (int l = 1; l < Lh; ++l) {
        T* aout = lds + ((l & 1) ? L.act1 : L.act0);
        const unsigned sl = slice_h(l), sn = slice_h(l + 1 < Lh ? l + 1 : 1);
        T acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = T(0);
        const T* ap = ain + arow * as + ak;
#pragma unroll
        for (int g = 0; g < NGH; ++g) {
          const int gn = g + D;                     // the group fetched now: D ahead, into the slot group g-1 left
#pragma unroll
          for (int kk = 0; kk < G; ++kk) {
            if (gn < NGH) load_frag<T, NT>(wr, sl + (unsigned)(gn * G + kk) * 64u * NT, lo, ring[gn % NB][kk]);
            else load_frag<T, NT>(wr, sn + (unsigned)((gn - NGH) * G + kk) * 64u * NT, lo, ring[gn % NB][kk]);
          }
          const int kg0 = OWNPACK ? 8 * ((g + w) & (NG8 - 1)) : G * g;   // first k-step of this group
#pragma unroll
          for (int kk = 0; kk < G; ++kk) {
            const T a = ap[4 * (kg0 + kk)];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(a, ring[g % NB][kk][nt], acc[nt]);
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = 16 * (NT * w + nt) + dcol;
          aout[drow * as + col] = act_apply<T>(mlp.act, acc[nt] + bias[l * HP + col]);
        }
        ain = aout;
        lds_barrier();
      }
      AMPC_IMARK(46);
      // ---- output layer: this wave's k range, partial sums to LDS
      {
        T o0 = T(0), o1 = T(0);
        const T* ap = ain + arow * as + 4 * (w * KSW) + ak;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
          const T a = ap[4 * ks];
          o0 = mfma4(a, wout[ks][0], o0);
          if (tiles > 1) o1 = mfma4(a, wout[ks][1], o1);
        }
        part[(w * ROWS + drow) * nxp + dcol] = o0;
        if (tiles > 1) part[(w * ROWS + drow) * nxp + 16 + dcol] = o1;
      }
      AMPC_IMARK(47);
      lds_barrier();
      AMPC_IMARK(48);
      for (int a = r; a < nx; a += TPS) {
        T s = bias[Lh * HP + a];
#pragma unroll
        for (int ww = 0; ww < W; ++ww) s += part[(ww * ROWS + m) * nxp + a];
        const T xn = xu[m * xs + a] + s;
        xu[m * xs + a] = xn;
        if (args.mode == 0 && m == 0) stw[(size_t)(t + 1) * nx + a] = xn;
      }
      if (args.mode == 1 && t + 1 < H) commit_ls();
      AMPC_IMARK(49);
      lds_barrier();
      AMPC_IMARK(50);
    }
    if (args.mode == 1 && live)
      for (int a = r; a < nx; a += TPS) lss[((size_t)j * (H + 1) + H) * nx + a] = xu[m * xs + a];
    obj_part += quad_rows<T>(Fm, xu + m * xs, goal, no, r, TPS, cdiag);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) obj_part += __shfl_xor(obj_part, off);
    if (lane == 0) lsobj[kIlqrMaxLs + w] = obj_part;      // one partial per wave; TPS / 64 waves per row
    __syncthreads();
    if (tid < ROWS) {
      T s = T(0);
      for (int ww = 0; ww < TPS / 64; ++ww) s += lsobj[kIlqrMaxLs + tid * (TPS / 64) + ww];
      lsobj[ROWS * pass + tid] = s;
    }
    __syncthreads();

    if (args.mode == 0) {
      if (tid == 0) {
        args.obj[p] = lsobj[0];
        args.active[p] = 1; args.converged[p] = 0; args.iters[p] = 0; args.status[p] = 0;
        args.refresh[p] = 1;
      }
      return;
    }
    // ---- the reference's acceptance loop over the candidates rolled out so far (ilqr.py:207-233)
    if (tid == 0) {
      const T obj = args.obj[p];
      const T lin_ = scal[0], quad_ = scal[1], ksn = scal[2];
      for (int jj = ROWS * pass; jj < rows && jj < ROWS * (pass + 1); ++jj) {
        last = jj;
        const T a = args.alphas[jj];
        const T new_obj = lsobj[jj];
        const T expect = a * lin_ + a * a * quad_ / T(2);
        const T ratio = (obj - new_obj) / (-expect);
        if (ratio > args.ls_cost_threshold) { best_obj = new_obj; best = jj; decided = 1; break; }
        if (new_obj < best_obj) { best_obj = new_obj; best = jj; }
        if (ksn < args.u_threshold) { decided = 1; break; }
      }
      piv[3] = decided;
    }
    __syncthreads();
    if (piv[3]) break;
  }

  // =========================== acceptance (ilqr.py:234-261) ====================================
  if (tid == 0) {
    const T obj = args.obj[p];
    const T ksn = scal[2];
    const bool success = (best_obj < obj) || (ksn < args.u_threshold);
    int sel = success ? best : last;
    int fail = 0;
    if (best < 0) { fail = 1; sel = 0; if (success) args.status[p] = 2; }
    const T new_obj = lsobj[sel];
    if (!success && new_obj > obj + T(1e-3)) fail = 1;
    piv[0] = sel; piv[1] = fail; piv[2] = success ? 1 : 0;
    scal[3] = new_obj;
    args.iters[p] += 1;
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
    const T d = lsc[(size_t)sel * H * nu + i] - ctw[i];
    du2 += d * d;
  }
  du2 = block_sum_any(du2, lsobj + kIlqrMaxLs, W);
  for (int i = tid; i < H * nu; i += NTHR) ctw[i] = lsc[(size_t)sel * H * nu + i];
  for (int i = tid; i < (H + 1) * nx; i += NTHR) stw[i] = lss[(size_t)sel * (H + 1) * nx + i];
  if (tid == 0) {
    const bool conv = sqrt(du2) < args.u_threshold;
    args.obj[p] = scal[3];
    args.refresh[p] = success;
    if (conv) { args.converged[p] = 1; args.active[p] = 0; }
  }
}

}  // namespace ampc
