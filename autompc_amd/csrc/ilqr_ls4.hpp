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
// l/16] (the same for the four blocks), B = W[k = 4ks + l/16][col 16g + l%16] -- the lane <-> (k, col)
// map of the fragment the 16x16x4 MFMA takes, so the packers of api.cpp serve both -- and
// D = out[row l/16][col 16g + l%16], one value per lane.
//
// Weights.  A CU can fetch 64 B per clock from L2: streaming the 512 KB of a 256 x 256 f64 layer
// costs 8 k cycles per step, twice the matrix-pipe time of four rows.  The kernel runs H steps on
// the same weights, so it keeps what fits on chip.  The workgroup is FOUR waves, one per SIMD (its
// own N-split packing, MlpDev::w4: wave w owns hidden columns [16 NT w, 16 NT (w+1)), NT = hpad/64):
// a wave then has the SIMD's whole 512-entry register file, and the per-wave working set is paid
// once per SIMD instead of twice.  Output-layer fragments stay in registers, first-layer fragments
// in registers or (NT >= 3) LDS, and with RES (networks with one hidden -> hidden layer) that layer
// is split, per round of KSH/8 k-steps, into RPR register-resident, LPR LDS-resident and SPR
// streamed k-steps.  Streamed k-steps go through a ring of NB groups that runs NB-1 rounds ahead of
// the MFMAs, across time-step boundaries.  Without RES every hidden -> hidden layer is streamed.
#pragma once
#include "ilqr_kernels.hpp"

namespace ampc {

constexpr int kLs4W = 4;   // waves per workgroup

// Sum over aligned groups of 4 or 8 consecutive lanes with DPP moves (no LDS round trips, unlike the
// ds_bpermute a generic shuffle lowers to): row_half_mirror pairs lane i with 7-i, the two
// quad_perm steps finish the sum inside each quad.  Every lane ends up with its group's total.
template <int CTRL> __device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum(double v, bool eight) {
  if (eight) v = dpp_add<0x141>(v);              // row_half_mirror
  v = dpp_add<0xb1>(v);                          // quad_perm [1,0,3,2]
  return dpp_add<0x4e>(v);                       // quad_perm [2,3,0,1]
}

// One k-step of a wave's MlpDev::w4 stream: NT values per lane, stored [chunk][lane][CW] (CW = 2 for
// even NT, else 1).  so = element offset of the k-step from MlpDev::wbase (wave-uniform).
template <int NT>
__device__ __forceinline__ void load_frag4(rsrc_t r, unsigned so, int lane, double (&b)[NT]) {
  constexpr int CW = NT % 2 == 0 ? 2 : 1, CH = NT / CW;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    double t[CW];
    load_frag<double, CW>(r, so + (unsigned)c * 64u * CW, (unsigned)lane * CW, t);
#pragma unroll
    for (int e = 0; e < CW; ++e) b[c * CW + e] = t[e];
  }
}
// the same fragment from / to an LDS copy with the same [chunk][lane][CW] layout (p = start of the k-step)
template <int NT>
__device__ __forceinline__ void lds_frag4(const double* p, int lane, double (&b)[NT]) {
  constexpr int CW = NT % 2 == 0 ? 2 : 1, CH = NT / CW;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const vec_t<double, CW> v = *reinterpret_cast<const vec_t<double, CW>*>(p + (c * 64 + lane) * CW);
#pragma unroll
    for (int e = 0; e < CW; ++e) b[c * CW + e] = v[e];
  }
}
template <int NT>
__device__ __forceinline__ void lds_put4(double* p, int lane, const double (&b)[NT]) {
  constexpr int CW = NT % 2 == 0 ? 2 : 1, CH = NT / CW;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < CW; ++e) p[(c * 64 + lane) * CW + e] = b[c * CW + e];
}

struct Ls4Lds {
  int xu, xs, act0, act1, as, part, bias, cpar, blo, bhi, scal, lsobj, piv, hres, w0, total;
};
// hidden-layer residency split (k-steps per round of KSH/8 = 2 NT): registers, LDS, streamed
// (rb = row blocks of the tile: 1 here, 3 in ilqr_lsw.hpp, whose activations of 4 rb rows take the LDS
//  the resident k-steps had)
// (rb > 1, NT = 4: two k-steps per round -- the twelve-row tile's own working set needs the registers, and
//  its MFMA time hides six streamed k-steps per round)
__host__ __device__ constexpr int ls4_rpr(int NT, int rb = 1) {
  return NT <= 2 ? 2 * NT : (NT == 4 ? (rb > 1 ? 1 : 3) : 4);
}
__host__ __device__ constexpr int ls4_lpr(int NT, int rb = 1) { return NT == 4 && rb == 1 ? 1 : 0; }
// first-layer fragments in LDS instead of registers
__host__ __device__ constexpr bool ls4_w0_lds(int NT) { return NT >= 3; }
__host__ __device__ constexpr Ls4Lds make_ls4_lds(int nu, int k1p, int nxp, int hpad, int n_hidden, bool res,
                                                  int cost_stride, int rb = 1) {
  const int NT = hpad / 64, W = kLs4W, rows = 4 * rb;
  Ls4Lds L{};
  int o = 0;
  // Activation row stride.  The A operand of a k-step is act[row l%4][4 ks + l/16]: four rows x two k
  // offsets per 32-lane group of a ds_read_b64 (64 banks of 4 B).  The four-row kernel's reads pair up
  // into ds_read2_b64 (16-lane groups, one k offset: any odd stride is conflict-free); the twelve-row
  // kernel's rows 4 r + l%4 are too far apart for that, it issues ds_read_b64 -- with stride hpad + 1
  // rows (a, k+1) and (a+1, k) share a bank (2-way: 37 % of its LDS cycles were conflict cycles,
  // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE), with hpad + 2 the eight addresses take eight bank pairs.
  L.xs = k1p + 1; L.as = rb > 1 ? hpad + 2 : hpad + 1;
  L.xu = o; o += rows * L.xs;
  L.act0 = o; o += rows * L.as;
  L.act1 = o; o += rows * L.as;
  L.part = o; o += W * rows * nxp;
  L.bias = o; o += n_hidden * hpad + nxp;
  L.cpar = o; o += cost_stride;
  L.blo = o; o += nu;
  L.bhi = o; o += nu;
  L.scal = o; o += 8;
  L.lsobj = o; o += 2 * kIlqrMaxLs;
  L.piv = o; o += 8;
  o = (o + 3) / 4 * 4;                               // 32-byte aligned fragment reads
  L.hres = o; o += res ? W * 8 * ls4_lpr(NT, rb) * 64 * NT : 0;
  L.w0 = o; o += ls4_w0_lds(NT) ? W * (k1p / 4) * 64 * NT : 0;
  L.total = (o + 3) / 4 * 4;
  return L;
}

template <int NT, bool RES, typename SH = DynShape>
__global__ __launch_bounds__(64 * kLs4W) void ilqr_ls4_kernel(const IlqrArgs<double> args) {
  const int slot = ilqr_slot(args);
  const int mode = args.slot_mode ? args.slot_mode[slot] : args.mode;   // (queue: per slot)
  if (mode == 0 && blockIdx.y > 0) return;       // (side-by-side passes: a slot rolling out its guess has one)
  // second launch of a split line search (below): only problems the first launch left undecided
  if (args.ls_split == 2 && args.ls_pass[slot] == 0) return;
  using T = double;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int W = kLs4W, NTHR = 64 * W, ROWS = 4;
  static_assert(W == ROWS, "wave w owns candidate row w");
  constexpr int HP = 16 * NT * W, KSH = HP / 4, KSW = KSH / W, KS0MAX = 12;
  // RES: the layer in rounds of PPR k-steps -- SPR streamed, RPR in registers, LPR in LDS
  constexpr int ROUNDS = 8, PPR = KSH / ROUNDS;
  constexpr int RPR = RES ? ls4_rpr(NT) : 0, LPR = RES ? ls4_lpr(NT) : 0, SPR = RES ? PPR - RPR - LPR : 0;
  static_assert(SPR >= 0, "residency split");
  // streamed groups: RES: SPR k-steps per round; otherwise G k-steps, KSH / G groups per layer
  constexpr int G = RES ? (SPR > 0 ? SPR : 1) : (NT <= 2 && KSH % 32 == 0 ? 8 : 4);
  constexpr int NGH = RES ? ROUNDS : KSH / G, NB = 4, D = NB - 1;
  static_assert(NGH % NB == 0 && D < NGH, "ring phase must repeat per layer");
  constexpr bool STREAM = !RES || SPR > 0;
  constexpr bool W0LDS = ls4_w0_lds(NT);
  const int tid = threadIdx.x, p = slot, lane = tid & 63;
  const MlpDev<T> mlp = plan_model<SH, T>(args.mlp, [&] {               // (per-slot models: mlp_tile.hpp)
    return model_delta_of(args.model_delta, args.model_delta ? args.slot_model[p] : 0); });
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = mlp.nx, nu = mlp.nu, no = SH::kStatic ? SH::no : args.obs_dim;
  const int HS = args.H, H = args.slot_h ? args.slot_h[p] : HS;      // array stride, this slot's horizon
  const int Lh = RES ? 2 : mlp.n_hidden, nxp = mlp.nxp, tiles = nxp / 16, ks0 = mlp.k1p / 4;
  const int cost_stride = SH::kStatic ? cost_block_stride(SH::no, SH::nu) : args.cost_stride;
  const Ls4Lds L = make_ls4_lds(nu, mlp.k1p, nxp, HP, Lh, RES, cost_stride);
  T* xu = lds + L.xu; T* part = lds + L.part; T* bias = lds + L.bias;
  T* cpar = lds + L.cpar; T* blo = lds + L.blo; T* bhi = lds + L.bhi; T* scal = lds + L.scal;
  T* lsobj = lds + L.lsobj; int* piv = reinterpret_cast<int*>(lds + L.piv);
  const int xs = L.xs, as = L.as;
  const T* Qm = cpar; const T* Rm = Qm + no * no; const T* Fm = Rm + nu * nu;
  const T* goal = Fm + no * no;
  const T* clin = goal + no; const T* clint = clin + no;     // affine part of the stage / terminal cost

  if (mode == 1 && args.active[p] == 0) {
    if (tid == 0) args.refresh[p] = 0;
    return;
  }
  if (mode == 1 && args.ric[(size_t)p * kRicStride + 3] != T(0)) return;   // singular Quu: retired by the sweep

  for (int l = 0; l < Lh; ++l)
    for (int i = tid; i < HP; i += NTHR) bias[l * HP + i] = mlp.B(l)[i];
  for (int i = tid; i < nxp; i += NTHR) bias[Lh * HP + i] = mlp.B(Lh)[i];
  for (int i = tid; i < 4 * xs; i += NTHR) xu[i] = T(0);
  for (int i = tid; i < cost_stride; i += NTHR)
    cpar[i] = args.costs_par[(size_t)args.cost_idx[p] * cost_stride + i];
  for (int i = tid; i < nu; i += NTHR) {
    blo[i] = args.bounded ? args.ubounds[i] : T(0);
    bhi[i] = args.bounded ? args.ubounds[nu + i] : T(0);
  }
  if (tid == 0 && mode == 1) {
    const T* rin = args.ric + (size_t)p * kRicStride;
    scal[0] = rin[0]; scal[1] = rin[1]; scal[2] = rin[2];
  }

  // ---- resident fragments + the hidden-layer ring ------------------------------------------------
  const rsrc_t wr = weight_rsrc(mlp.WB());
  T w0[W0LDS ? 1 : KS0MAX][NT];
  T* w0l = lds + L.w0 + (size_t)w * ks0 * 64 * NT;               // this wave's fragments, k-step stride 64 * NT
  {
    const unsigned s0 = (unsigned)(mlp.W4(0) - mlp.WB()) + (unsigned)w * (unsigned)ks0 * 64u * NT;
#pragma unroll
    for (int ks = 0; ks < KS0MAX; ++ks) {
      if constexpr (W0LDS) {
        if (ks < ks0) {
          T tmp[NT];
          load_frag4<NT>(wr, s0 + (unsigned)ks * 64u * NT, lane, tmp);
          lds_put4<NT>(w0l + ks * 64 * NT, lane, tmp);
        }
      } else {
        if (ks < ks0) load_frag4<NT>(wr, s0 + (unsigned)ks * 64u * NT, lane, w0[ks]);
        else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) w0[ks][nt] = T(0);
        }
      }
    }
  }
  T wout[KSW][2];
  {
    const T* wl = mlp.W4(Lh) + ((size_t)w * KSW * 64 + lane) * tiles;
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      wout[ks][0] = wl[(size_t)ks * 64 * tiles];
      wout[ks][1] = tiles > 1 ? wl[(size_t)ks * 64 * tiles + 1] : T(0);
    }
  }
  auto slice_h = [&](int l) {
    return (unsigned)(mlp.W4(l) - mlp.WB()) + (unsigned)w * (unsigned)KSH * 64u * NT;
  };
  // position (k-step index in this wave's packed stream) of streamed k-step kk of group g
  auto spos = [&](int g, int kk) { return RES ? g * PPR + kk : g * G + kk; };
  T ring[STREAM ? NB : 1][G][NT];
  T res[RES ? ROUNDS : 1][RPR > 0 ? RPR : 1][NT];
  T* hres = lds + L.hres + (size_t)w * ROUNDS * (LPR > 0 ? LPR : 1) * 64 * NT;
  if (Lh > 1) {
    const unsigned s1 = slice_h(1);
    if constexpr (STREAM) {
#pragma unroll
      for (int g = 0; g < D; ++g)
#pragma unroll
        for (int kk = 0; kk < G; ++kk) load_frag4<NT>(wr, s1 + (unsigned)spos(g, kk) * 64u * NT, lane, ring[g][kk]);
    }
    if constexpr (RES) {
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
        for (int i = 0; i < RPR; ++i) load_frag4<NT>(wr, s1 + (unsigned)(rd * PPR + SPR + i) * 64u * NT, lane, res[rd][i]);
#pragma unroll
        for (int i = 0; i < LPR; ++i) {
          T tmp[NT];
          load_frag4<NT>(wr, s1 + (unsigned)(rd * PPR + SPR + RPR + i) * 64u * NT, lane, tmp);
          lds_put4<NT>(hres + (rd * LPR + i) * 64 * NT, lane, tmp);
        }
      }
    }
  }
  const int arow = lane & 3, ak = lane >> 4, drow = lane >> 4, dcol = lane & 15;

  const T* st = args.states + (size_t)p * (HS + 1) * nx;
  T* stw = args.states + (size_t)p * (HS + 1) * nx;
  T* ctw = args.ctrls + (size_t)p * HS * nu;
  const T* Kg = args.Ks + (size_t)p * HS * nu * nx;
  const T* kg = args.ks + (size_t)p * HS * nu;
  T* lss = args.ls_states + (size_t)p * args.ls_n * (HS + 1) * nx;
  T* lsc = args.ls_ctrls + (size_t)p * args.ls_n * HS * nu;
  const int rows = mode == 0 ? 1 : args.ls_n;
  const bool cdiag = args.cost_diag != 0, caff = args.cost_affine != 0;
  // Wave w owns row w of the tile between time steps: it adds the network output to its state,
  // evaluates the control law (ilqr.py:196-205) -- no workgroup barrier in between -- and
  // accumulates the row's stage cost.  Control law: `parts` lanes per control, interleaved over
  // the state index; each lane keeps its entries of K_t, xbar_t (and k_t, ubar_t) in registers,
  // loaded one step ahead.
  const int parts = nu <= 8 ? 8 : 4;
  const int ca = lane / parts, cpart = lane - ca * parts;
  constexpr int KPL = 8;                             // entries per lane: nx <= 32 = 4 * 8
  T kreg[KPL], xbreg[KPL], kvr = T(0), ubr = T(0);
  auto fetch_law = [&](int t) {
    if (mode == 1) {
#pragma unroll
      for (int i = 0; i < KPL; ++i) {
        const int b = cpart + parts * i;
        if (ca < nu && b < nx) { kreg[i] = Kg[((size_t)t * nu + ca) * nx + b]; xbreg[i] = st[(size_t)t * nx + b]; }
      }
      if (ca < nu) kvr = kg[(size_t)t * nu + ca];
    }
    if (ca < nu) ubr = ctw[(size_t)t * nu + ca];    // mode 0: the control itself
  };

  // acceptance state of the reference's sequential loop (thread 0)
  T best_obj = INFINITY;
  int best = -1, last = 0, decided = 0;
  __syncthreads();

  // Passes: one after the other in this workgroup (stopping at the first accepted candidate), or --
  // small batches, idle CUs -- each pass on its own workgroup (blockIdx.y), all at once; the last
  // workgroup to finish then runs the reference's acceptance loop over all of them.
  const int npass = (rows + ROWS - 1) / ROWS;
  // Split line search (args.ls_split; many problems per launch): the problems of a launch run in
  // lock-step, and most line searches are decided by their first four step sizes -- a launch that ran
  // all three passes back to back would keep every decided problem's CU idle while the few undecided
  // ones work through passes two and three.  So launch 1 rolls out pass 0 for everybody and runs the
  // acceptance loop over it; a problem it leaves undecided parks its four objectives in `ric` and raises
  // ls_pass[p]; launch 2 (grid.y = npass - 1, everybody else exits at once) rolls out the REMAINING
  // passes side by side on the now idle CUs, and the last of them to arrive runs the reference's
  // acceptance loop over all candidates.  Same decisions, same arithmetic as the passes in sequence.
  const int split = mode == 1 ? args.ls_split : 0;
  const bool par = mode == 1 && (args.par_passes != 0 || split == 2);
  const int par_first = split == 2 ? 1 : 0;                   // first pass the side-by-side launch covers
  T* gstate = args.ric + (size_t)p * kRicStride + 4;          // [16] candidate objectives
  const int pass0 = par ? (int)blockIdx.y + par_first : 0;
  const int pass1 = par ? pass0 + 1 : (split == 1 ? 1 : npass);
  for (int pass = pass0; pass < pass1; ++pass) {
    const int jw = ROWS * pass + w;                  // the candidate this wave's row carries
    const bool livew = jw < rows;
    const T alpha = args.alphas[jw < kIlqrMaxLs ? jw : 0];
    T obj_part = T(0);
    if (lane < nx) xu[w * xs + lane] = st[lane];
    fetch_law(0);
    for (int t = 0; t <= H; ++t) {
      AMPC_IPROBE_STEP(mode == 1 && pass == 0 && t == H / 2);
      AMPC_IMARK(40);
      // ---- between steps, on row w: x_t = x_{t-1} + net output, then u_t
      if (t > 0 && lane < nx) {
        T s = bias[Lh * HP + lane];
#pragma unroll
        for (int ww = 0; ww < W; ++ww) s += part[(ww * ROWS + w) * nxp + lane];
        const T xn = xu[w * xs + lane] + s;
        xu[w * xs + lane] = xn;
        if (mode == 0 && w == 0) stw[(size_t)t * nx + lane] = xn;
      }
      if (mode == 1 && livew && lane < nx) lss[((size_t)jw * (HS + 1) + t) * nx + lane] = xu[w * xs + lane];
      if (t == H) break;
      {
        T u;
        if (mode == 0) {
          u = ubr;
        } else {
          T f = T(0);
#pragma unroll
          for (int i = 0; i < KPL; ++i) {
            const int b = cpart + parts * i;
            if (ca < nu && b < nx) f += kreg[i] * (xu[w * xs + b] - xbreg[i]);
          }
          f = group_sum(f, parts == 8);
          u = alpha * kvr + ubr + f;
          if (args.bounded && ca < nu) { u = u < blo[ca] ? blo[ca] : u; u = u > bhi[ca] ? bhi[ca] : u; }
        }
        if (ca < nu && cpart == 0) {
          if (mode == 1 && livew) lsc[((size_t)jw * HS + t) * nu + ca] = u;
          xu[w * xs + nx + ca] = u;
        }
        if (t + 1 < H) fetch_law(t + 1);
      }
      AMPC_IMARK(41);
      lds_barrier();
      AMPC_IMARK(42);
      // ---- layer 0
      T* ain = lds + L.act0;
      {
        T acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = T(0);
        const T* ap = xu + arow * xs + ak;
        T av0[KS0MAX];
#pragma unroll
        for (int ks = 0; ks < KS0MAX; ++ks)
          if (ks < ks0) av0[ks] = ap[4 * ks];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS0MAX; ++ks)
          if (ks < ks0) {
            const T a = av0[ks];
            T bv[NT];
            if constexpr (W0LDS) {
              lds_frag4<NT>(w0l + ks * 64 * NT, lane, bv);
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) bv[nt] = w0[ks][nt];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(a, bv[nt], acc[nt]);
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
      // ---- hidden -> hidden layers
      for (int l = 1; l < Lh; ++l) {
        T* aout = lds + ((l & 1) ? L.act1 : L.act0);
        const unsigned sl = slice_h(l), sn = slice_h(l + 1 < Lh ? l + 1 : 1);
        T acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = T(0);
        const T* ap = ain + arow * as + ak;
        constexpr int APG = RES ? PPR : G;            // A operands (k-steps) per group
        T av[2][APG];                                 // one group ahead of the MFMAs
#pragma unroll
        for (int kk = 0; kk < APG; ++kk) av[0][kk] = ap[4 * kk];
#pragma unroll
        for (int g = 0; g < NGH; ++g) {
          if (g + 1 < NGH) {
#pragma unroll
            for (int kk = 0; kk < APG; ++kk) av[(g + 1) & 1][kk] = ap[4 * (APG * (g + 1) + kk)];
          }
          const int gn = g + D;                       // the streamed group fetched now: D ahead
          if constexpr (STREAM) {
#pragma unroll
            for (int kk = 0; kk < G; ++kk) {
              if (gn < NGH) load_frag4<NT>(wr, sl + (unsigned)spos(gn, kk) * 64u * NT, lane, ring[gn % NB][kk]);
              else load_frag4<NT>(wr, sn + (unsigned)spos(gn - NGH, kk) * 64u * NT, lane, ring[gn % NB][kk]);
            }
          }
          if constexpr (RES) {
            // round g = k-steps [PPR g, PPR (g+1)): streamed, then register-resident, then LDS-resident
            T lv[LPR > 0 ? LPR : 1][NT];
#pragma unroll
            for (int i = 0; i < LPR; ++i) lds_frag4<NT>(hres + (g * LPR + i) * 64 * NT, lane, lv[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < PPR; ++kk) {
              const T a = av[g & 1][kk];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const T bv = kk < SPR ? ring[g % NB][kk < SPR ? kk : 0][nt]
                           : kk < SPR + RPR ? res[g][kk - SPR < RPR && kk >= SPR ? kk - SPR : 0][nt]
                                            : lv[kk >= SPR + RPR ? kk - SPR - RPR : 0][nt];
                acc[nt] = mfma4(a, bv, acc[nt]);
              }
            }
          } else {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < G; ++kk) {
              const T a = av[g & 1][kk];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(a, ring[g % NB][kk][nt], acc[nt]);
            }
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
        T avo[KSW];
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) avo[ks] = ap[4 * ks];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
          const T a = avo[ks];
          o0 = mfma4(a, wout[ks][0], o0);
          if (tiles > 1) o1 = mfma4(a, wout[ks][1], o1);
        }
        part[(w * ROWS + drow) * nxp + dcol] = o0;
        if (tiles > 1) part[(w * ROWS + drow) * nxp + 16 + dcol] = o1;
      }
      AMPC_IMARK(47);
      lds_barrier();
      AMPC_IMARK(48);
    }
    // ---- objective of row w (ilqr.py:141-149, 206): dt * stage costs + terminal cost, from the stored
    // trajectory, one time step per lane -- kept off the serial chain of the rollout above
    __syncthreads();                                 // (orders this workgroup's trajectory stores)
    {
      const T* xsrc = mode == 0 ? stw : lss + (size_t)jw * (HS + 1) * nx;
      const T* usrc = mode == 0 ? ctw : lsc + (size_t)jw * HS * nu;
      if (livew)
        for (int t = lane; t <= H; t += 64) {
          const T* xt = xsrc + (size_t)t * nx;
          if (t < H) obj_part += args.dt * (quad_rows<T>(Qm, xt, goal, no, 0, 1, cdiag) +
                                            quad_rows<T>(Rm, usrc + (size_t)t * nu, nullptr, nu, 0, 1, cdiag));
          else obj_part += quad_rows<T>(Fm, xt, goal, no, 0, 1, cdiag);
          if (caff) {
            if (t < H) obj_part += args.dt * affine_rows<T>(clin, xt, goal, no, 0, 1, clint[no]);
            else obj_part += affine_rows<T>(clint, xt, goal, no, 0, 1, clint[no + 1]);
          }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) obj_part += __shfl_xor(obj_part, off);
    if (lane == 0) lsobj[ROWS * pass + w] = obj_part;
    __syncthreads();

    if (mode == 0) {
      if (tid == 0) {
        args.obj[p] = lsobj[0];
        args.active[p] = 1; args.converged[p] = 0; args.iters[p] = 0; args.status[p] = 0;
        args.refresh[p] = 1; args.ls_rows[p] = 0; args.ls_count[p] = 0;
      }
      return;
    }
    if (par) {
      // publish this pass's objectives; the last pass to arrive takes over
      T* gobj = args.ric + (size_t)p * kRicStride + 4;
      if (tid < ROWS) gobj[ROWS * pass + tid] = lsobj[ROWS * pass + tid];
      __threadfence();
      __syncthreads();
      if (tid == 0) piv[4] = atomicAdd(&args.ls_count[p], 1);
      __syncthreads();
      if (piv[4] != npass - par_first - 1) return;
      __threadfence();
      if (tid == 0) { args.ls_count[p] = 0; if (split == 2) args.ls_pass[p] = 0; }
      if (tid < kIlqrMaxLs) lsobj[tid] = tid < rows ? __builtin_nontemporal_load(gobj + tid) : T(0);
      __syncthreads();
    }
    // ---- the reference's acceptance loop over the candidates rolled out so far (ilqr.py:207-233)
    if (tid == 0) {
      const T obj = args.obj[p];
      const T lin_ = scal[0], quad_ = scal[1], ksn = scal[2];
      for (int jj = par ? 0 : ROWS * pass; jj < rows && (par || jj < ROWS * (pass + 1)); ++jj) {
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
    if (split == 1 && npass > 1) {               // undecided, more step sizes to try: the second launch
      if (tid < ROWS) gstate[tid] = lsobj[tid];
      if (tid == 0) { args.ls_pass[p] = 1; args.refresh[p] = 0; }
      return;
    }
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
    args.ls_rows[p] += par ? ROWS * npass : ROWS * (last / ROWS + 1);
    args.ls_need[p] = last / 4 + 1;
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
    const T d = lsc[(size_t)sel * HS * nu + i] - ctw[i];
    du2 += d * d;
  }
  du2 = block_sum_any(du2, lsobj + kIlqrMaxLs, W);
  for (int i = tid; i < H * nu; i += NTHR) ctw[i] = lsc[(size_t)sel * HS * nu + i];
  for (int i = tid; i < (H + 1) * nx; i += NTHR) stw[i] = lss[(size_t)sel * (HS + 1) * nx + i];
  if (tid == 0) {
    const bool conv = sqrt(du2) < args.u_threshold;
    args.obj[p] = scal[3];
    args.refresh[p] = success;
    if (conv) { args.converged[p] = 1; args.active[p] = 0; }
    else if (args.max_iter > 0 && args.iters[p] >= args.max_iter) args.active[p] = 0;
    if (args.active[p] == 0) args.refresh[p] = 0;       // retired: nobody reads its Jacobians again
  }
}

}  // namespace ampc
