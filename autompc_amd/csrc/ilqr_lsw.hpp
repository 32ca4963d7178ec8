// ilqr_lsw.hpp -- iLQR line search with ALL step sizes in one pass: 4 RB candidate rows per tile (f64, gfx950).
//
// ilqr_ls4_kernel (ilqr_ls4.hpp) rolls the candidates out four at a time and stops at the first pass that
// decides the search -- the fastest way to finish ONE search.  When many problems share a launch they run
// in lock-step, and the launch lasts as long as its slowest search: three four-row passes, because among a
// few hundred slots SOME search always needs all ten step sizes (on the converging HalfCheetah set 31 % of
// the searches do, 40 % need two passes, 29 % one; oracle trace, tools/ls_pass_histogram.py).  Here the tile
// carries 4 RB rows (RB = 3: twelve rows, ten of them candidates) -- RB v_mfma_f64_4x4x4_4b per weight
// fragment, one per block of four rows -- so every search of the launch is decided after ONE pass: the
// weight stream and the serial chain of a time step (state update, control law, four barriers) are paid
// once instead of three times, and three times the matrix-pipe work per streamed byte makes the hidden
// layer MFMA-bound (12 independent accumulators per wave also cover the ~90-cycle dependent latency of the
// instruction, which four do not: tools/mfma44_rate1.cpp, 17.4 against 22.8 cycles per MFMA at one wave
// per SIMD).  The streamed share of the hidden layer hides under the MFMAs, so the LDS the resident
// k-steps had goes to the activations of the extra rows (ls4_lpr), and the ring runs one round ahead.
//
// A row's arithmetic does not depend on what else is in the tile, and the acceptance loop below is the
// four-row kernel's (the reference's, ilqr.py:207-261) over all candidates at once: results are those
// of ilqr_ls4_kernel bit for bit (tests/test_gpu_ilqr.py, test_gpu_ilqr_queue.py run both).
// Layout, operand maps and the weight packing (MlpDev::w4) are ilqr_ls4.hpp's.
//
// The phase between two time steps is a chain of LDS round trips, so it is written without divergent
// branches: every lane reads (addresses clamped into range, wave-uniform bases + 32-bit lane offsets),
// entries that do not exist enter the sums as exact zeros, only the final stores are predicated -- the
// reads of all rows and entries go out together, one round trip per stage.
#pragma once
#include "ilqr_ls4.hpp"

namespace ampc {

// Per-step global traffic of the time loop (K_t, xbar_t, k_t, ubar_t in; x_t, u_t out) as raw buffer
// accesses: resource in scalar registers, the time step in the scalar offset, the lane's share in ONE
// 32-bit vector offset -- no 64-bit address pair per access to keep (or spill) across the loop.
// (the resource is rebuilt from the pointer, made scalar with readfirstlane, at every use: a resource the
//  register allocator has parked in vector registers would be applied lane group by lane group)
template <typename T> __device__ __forceinline__ rsrc_t uni_rsrc(const T* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return weight_rsrc(reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo));
}
template <typename T> __device__ __forceinline__ double ld_buf(const T* base, unsigned voff, unsigned soff) {
  if constexpr (Probe::lsw_global)
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + soff + voff);
  else return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(uni_rsrc(base), voff, soff, 0));
}
template <typename T> __device__ __forceinline__ void st_buf(T* base, unsigned voff, unsigned soff, double v) {
  if constexpr (Probe::lsw_global) *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + soff + voff) = v;
  else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), uni_rsrc(base), voff, soff, 0);
}

template <int NT, bool RES, typename SH, int RB>
__global__ __launch_bounds__(64 * kLs4W) void ilqr_lsw_kernel(const IlqrArgs<double> args) {
  const int slot = ilqr_slot(args);
  const int mode = args.slot_mode ? args.slot_mode[slot] : args.mode;   // (queue: per slot)
  using T = double;
  // (phase_time build: this wave's marks -- 0..7 one time step, 8 kernel entry, 9 / 10 around the time loop
  //  of pass 0, 11 objectives done)
  [[maybe_unused]] long long pm[Probe::phase_time ? 12 : 1] = {};
  [[maybe_unused]] const bool pm_k = Probe::phase_time && blockIdx.x == 7;
  AMPC_LSW_MARK(pm, pm_k, 8);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int W = kLs4W, NTHR = 64 * W, ROWS = 4 * RB;
  static_assert(W == 4 && RB >= 2 && RB <= 4, "wave w owns candidate rows w, 4 + w, ...");
  constexpr int HP = 16 * NT * W, KSH = HP / 4, KSW = KSH / W, KS0MAX = 12;
  // RES: the layer in rounds of PPR k-steps -- SPR streamed, RPR in registers, LPR in LDS
  constexpr int ROUNDS = 8, PPR = KSH / ROUNDS;
  constexpr int RPR = RES ? ls4_rpr(NT, RB) : 0, LPR = RES ? ls4_lpr(NT, RB) : 0, SPR = RES ? PPR - RPR - LPR : 0;
  static_assert(SPR >= 0, "residency split");
  // streamed groups: RES: SPR k-steps per round; otherwise G k-steps, KSH / G groups per layer
  constexpr int G = RES ? (SPR > 0 ? SPR : 1) : (NT <= 2 && KSH % 32 == 0 ? 8 : 4);
  // RES: a round lasts RB times as long as in the four-row kernel, so one round of look-ahead covers the
  // L2 latency, and the ring is ONE group of registers that rolls: the fragment of streamed k-step kk is
  // consumed at k-step kk of a round, and the request for the next round's goes out one k-step later into
  // the same registers (the MFMAs that read them have been issued; the data lands hundreds of cycles on).
  constexpr int NGH = RES ? ROUNDS : KSH / G, NB = RES ? 1 : 4, D = RES ? 1 : NB - 1;
  static_assert(NGH % NB == 0 && D < NGH, "ring phase must repeat per layer");
  static_assert(!RES || G < PPR, "the rolling ring needs a k-step behind the last streamed one");
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
  const Ls4Lds L = make_ls4_lds(nu, mlp.k1p, nxp, HP, Lh, RES, cost_stride, RB);
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
  for (int i = tid; i < ROWS * xs; i += NTHR) xu[i] = T(0);
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
  // Wave w owns rows w, 4 + w, ... of the tile between time steps: it adds the network output to their
  // states and evaluates the control law (ilqr.py:196-205) -- no workgroup barrier in between.
  // Control law: `parts` lanes per control, interleaved over the state index; each lane keeps its
  // entries of K_t, xbar_t (and k_t, ubar_t) in registers, loaded one step ahead (see the file comment
  // for the branch-free form).
  const int parts = nu <= 8 ? 8 : 4;
  const int ca = lane / parts, cpart = lane - ca * parts;
  const int cac = ca < nu ? ca : nu - 1;             // (lanes past the controls: results unused)
  constexpr int KPL = 8;                             // entries per lane: nx <= 32 = 4 * 8
  T kreg[KPL], xbreg[KPL], kvr = T(0), ubr = T(0);
  const unsigned kofs = (unsigned)(cac * nx + cpart);
  auto fetch_law = [&](int t) {
    if (mode == 1) {
      const unsigned tk = (unsigned)(t * nu * nx) * 8u, tx = (unsigned)(t * nx) * 8u;   // (scalar offsets)
#pragma unroll
      for (int i = 0; i < KPL; ++i)
        if (parts * i < nx) {                          // (uniform; static shapes: compile time)
          // (entries past nx: a valid address, the value is masked where it is USED -- a select here
          //  would wait for the load one step early)
          const unsigned bo = cpart + parts * i < nx ? (unsigned)(parts * i) * 8u : 0u;
          kreg[i] = ld_buf(Kg, kofs * 8u + bo, tk);
          xbreg[i] = ld_buf(st, (unsigned)cpart * 8u + bo, tx);
        }
      kvr = ld_buf(kg, (unsigned)cac * 8u, (unsigned)(t * nu) * 8u);
    }
    ubr = ld_buf(ctw, (unsigned)cac * 8u, (unsigned)(t * nu) * 8u);     // mode 0: the control itself
  };
  // state update: lane (xr, xj) = (lane / 32, lane % 32) takes state xj of rows 4 (2 q + xr) + w
  constexpr int XQ = (RB + 1) / 2;
  const int xr = lane >> 5, xj = lane & 31, xjc = xj < nx ? xj : 0;
  T blo_r = T(0), bhi_r = T(0);
  if (args.bounded) { blo_r = args.ubounds[cac]; bhi_r = args.ubounds[nu + cac]; }

  // acceptance state of the reference's sequential loop (thread 0)
  T best_obj = INFINITY;
  int best = -1, last = 0, decided = 0;
  __syncthreads();

  // Passes (more than twelve step sizes only): one after the other, stopping at the first that decides
  const int npass = (rows + ROWS - 1) / ROWS;
  for (int pass = 0; pass < npass; ++pass) {
    // the candidates this wave's rows carry: tile row 4 r + w = candidate ROWS pass + 4 r + w
    int jw[RB];
    bool livew[RB];
    T alpha[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      jw[r] = ROWS * pass + 4 * r + w;
      livew[r] = jw[r] < rows;
      alpha[r] = args.alphas[jw[r] < kIlqrMaxLs ? jw[r] : 0];
    }
    if (lane < nx) {
#pragma unroll
      for (int r = 0; r < RB; ++r) xu[(4 * r + w) * xs + lane] = st[lane];
    }
    fetch_law(0);
    AMPC_LSW_MARK(pm, pm_k && pass == 0, 9);
    for (int t = 0; t <= H; ++t) {
      [[maybe_unused]] const bool pm_on = Probe::phase_time && blockIdx.x == 7 && mode == 1 && pass == 0 && t == H / 2;
      AMPC_LSW_MARK(pm, pm_on, 0);
      // ---- between steps, on row w: x_t = x_{t-1} + net output, then u_t
      // x_t = x_{t-1} + net output (all reads first: one LDS round trip), kept in registers for the
      // trajectory store, which goes out AFTER the control law: the law's wait for its operands (loaded a
      // step ago) is a vmcnt wait, and a store issued just before it would put its acknowledgement
      // latency on the chain
      T xn[XQ];
      {
        T xo[XQ], pv[XQ][W + 1];
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
          const int r = 2 * q + xr < RB ? 2 * q + xr : RB - 1, row = 4 * r + w;
          xo[q] = xu[row * xs + xjc];
          if (t > 0) {
            pv[q][W] = bias[Lh * HP + xjc];
#pragma unroll
            for (int ww = 0; ww < W; ++ww) pv[q][ww] = part[(ww * ROWS + row) * nxp + xjc];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
          xn[q] = xo[q];
          if (t > 0) {
            T sacc = pv[q][W];
#pragma unroll
            for (int ww = 0; ww < W; ++ww) sacc += pv[q][ww];
            xn[q] = xo[q] + sacc;
            const int r = 2 * q + xr;
            if (r < RB && xj < nx) xu[(4 * r + w) * xs + xj] = xn[q];
          }
        }
      }
      auto store_x = [&]() {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
          const int r = 2 * q + xr;
          if (r < RB && xj < nx) {
            const int row = 4 * r + w, jl = ROWS * pass + row;
            if (mode == 0) { if (t > 0 && row == 0) st_buf(stw, (unsigned)xj * 8u, (unsigned)(t * nx) * 8u, xn[q]); }
            else if (jl < rows) st_buf(lss, (unsigned)(jl * (HS + 1) * nx + xj) * 8u, (unsigned)(t * nx) * 8u, xn[q]);
          }
        }
      };
      if (t == H) { store_x(); break; }
      {
        T u[RB];
        if (mode == 0) {
#pragma unroll
          for (int r = 0; r < RB; ++r) u[r] = ubr;
        } else {
          T xv[RB][KPL], f[RB];
#pragma unroll
          for (int i = 0; i < KPL; ++i)
            if (parts * i < nx) {
              const int b = cpart + parts * i < nx ? cpart + parts * i : cpart;
#pragma unroll
              for (int r = 0; r < RB; ++r) xv[r][i] = xu[(4 * r + w) * xs + b];
            }
#pragma unroll
          for (int r = 0; r < RB; ++r) f[r] = T(0);
#pragma unroll
          for (int i = 0; i < KPL; ++i)
            if (parts * i < nx) {
              const bool ok = cpart + parts * i < nx;
              const T kq = ok ? kreg[i] : T(0), xq = ok ? xbreg[i] : T(0);
#pragma unroll
              for (int r = 0; r < RB; ++r) f[r] += kq * (xv[r][i] - xq);
            }
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            f[r] = group_sum(f[r], parts == 8);
            u[r] = alpha[r] * kvr + ubr + f[r];
            if (args.bounded) { u[r] = u[r] < blo_r ? blo_r : u[r]; u[r] = u[r] > bhi_r ? bhi_r : u[r]; }
          }
        }
        if (ca < nu && cpart == 0) {
#pragma unroll
          for (int r = 0; r < RB; ++r) xu[(4 * r + w) * xs + nx + ca] = u[r];
        }
        store_x();
        if (mode == 1 && ca < nu && cpart == 0) {
#pragma unroll
          for (int r = 0; r < RB; ++r)
            if (livew[r]) st_buf(lsc, (unsigned)ca * 8u, (unsigned)((jw[r] * HS + t) * nu) * 8u, u[r]);
        }
        if (t + 1 < H) fetch_law(t + 1);
      }
      AMPC_LSW_MARK(pm, pm_on, 1);
      lds_barrier();
      AMPC_LSW_MARK(pm, pm_on, 2);
      // ---- layer 0
      T* ain = lds + L.act0;
      {
        T acc[RB][NT];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[r][nt] = T(0);
        const T* ap = xu + arow * xs + ak;
        // A operands and (first layer in LDS) B fragments one k-step ahead, requested in the MFMAs' shadows
        T av0[2][RB], bn[2][NT];
#pragma unroll
        for (int r = 0; r < RB; ++r) av0[0][r] = ap[4 * r * xs];
        if constexpr (W0LDS) lds_frag4<NT>(w0l, lane, bn[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS0MAX; ++ks)
          if (ks < ks0) {
            if (ks + 1 < ks0) {
#pragma unroll
              for (int r = 0; r < RB; ++r) av0[(ks + 1) & 1][r] = ap[4 * r * xs + 4 * (ks + 1)];
              if constexpr (W0LDS) lds_frag4<NT>(w0l + (ks + 1) * 64 * NT, lane, bn[(ks + 1) & 1]);
            }
            T bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = W0LDS ? bn[ks & 1][nt] : w0[W0LDS ? 0 : ks][nt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int r = 0; r < RB; ++r) acc[r][nt] = mfma4(av0[ks & 1][r], bv[nt], acc[r][nt]);
#pragma unroll
            for (int i = 0; i < RB + 2; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, RB * NT - RB - 2, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = 16 * (NT * w + nt) + dcol;
          const T bc = bias[col];
#pragma unroll
          for (int r = 0; r < RB; ++r) ain[(4 * r + drow) * as + col] = act_apply<T>(mlp.act, acc[r][nt] + bc);
        }
      }
      AMPC_LSW_MARK(pm, pm_on, 3);
      lds_barrier();
      AMPC_LSW_MARK(pm, pm_on, 4);
      // ---- hidden -> hidden layers
      for (int l = 1; l < Lh; ++l) {
        T* aout = lds + ((l & 1) ? L.act1 : L.act0);
        unsigned sl = slice_h(l), sn = slice_h(l + 1 < Lh ? l + 1 : 1);
        // (the stream's scalar offsets are base + constant: left alone, the compiler forms all of them once,
        //  outside the time loop, parks them in vector lanes and pays a v_readlane + hazard nop per load;
        //  an opaque base keeps them as one s_add each, next to the load)
        sl = (unsigned)__builtin_amdgcn_readfirstlane((int)sl);   // (wave-uniform by construction; a layer index
        sn = (unsigned)__builtin_amdgcn_readfirstlane((int)sn);   //  that is a run-time loop variable hides it)
        asm volatile("" : "+s"(sl), "+s"(sn));
        T acc[RB][NT];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[r][nt] = T(0);
        const T* ap = ain + arow * as + ak;
        // A k-step's RB x NT MFMAs (>= 128 cycles) cover an LDS read: the A operands run one k-step ahead
        // of the MFMAs, and fragment kk of the streamed group D rounds ahead is requested at k-step kk,
        // under its MFMAs, not in a clump at the start of the round.
        T av[2][RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) av[0][r] = ap[4 * r * as];
#pragma unroll
        for (int g = 0; g < NGH; ++g) {
          const int gn = g + D;
          constexpr int KPG = RES ? PPR : G;          // k-steps per group
          T lv[LPR > 0 ? LPR : 1][NT];
#pragma unroll
          for (int i = 0; i < LPR; ++i) lds_frag4<NT>(hres + (g * LPR + i) * 64 * NT, lane, lv[i]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kk = 0; kk < KPG; ++kk) {
            const int kg = KPG * g + kk;              // k-step of the layer
            if constexpr (STREAM) {
              constexpr int LAG = RES ? 1 : 0;        // (rolling ring: one k-step behind the consumer)
              if (kk >= LAG && kk - LAG < G) {
                const int ks_ = kk - LAG;
                if (gn < NGH) load_frag4<NT>(wr, sl + (unsigned)spos(gn, ks_) * 64u * NT, lane, ring[gn % NB][ks_]);
                else load_frag4<NT>(wr, sn + (unsigned)spos(gn - NGH, ks_) * 64u * NT, lane, ring[gn % NB][ks_]);
              }
            }
            if (kg + 1 < KSH) {
#pragma unroll
              for (int r = 0; r < RB; ++r) av[(kg + 1) & 1][r] = ap[4 * r * as + 4 * (kg + 1)];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              // RES: round g = k-steps [PPR g, PPR (g+1)): streamed, then register-resident, then LDS-resident
              const T bv = !RES ? ring[g % NB][kk < G ? kk : 0][nt]
                         : kk < SPR ? ring[g % NB][kk < SPR ? kk : 0][nt]
                         : kk < SPR + RPR ? res[g][kk - SPR < RPR && kk >= SPR ? kk - SPR : 0][nt]
                                          : lv[kk >= SPR + RPR ? kk - SPR - RPR : 0][nt];
#pragma unroll
              for (int r = 0; r < RB; ++r) acc[r][nt] = mfma4(av[kg & 1][r], bv, acc[r][nt]);
            }
            // One wave per SIMD issues in order: whatever stands BEHIND a run of MFMAs waits until the last
            // of them has been issued, and then the matrix pipe idles while it is issued.  So every request
            // of this k-step (A operands of the next, the streamed fragment, its scalar offset) goes out in
            // the 12-cycle shadow of one MFMA: MFMA, request, MFMA, request, ...
#pragma unroll
            for (int i = 0; i < RB; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU (v_readlane of a spilled offset)
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, RB * NT - RB - 4, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = 16 * (NT * w + nt) + dcol;
          const T bc = bias[l * HP + col];
#pragma unroll
          for (int r = 0; r < RB; ++r) aout[(4 * r + drow) * as + col] = act_apply<T>(mlp.act, acc[r][nt] + bc);
        }
        ain = aout;
        lds_barrier();
      }
      AMPC_LSW_MARK(pm, pm_on, 5);
      // ---- output layer: this wave's k range, partial sums to LDS
      {
        T o0[RB], o1[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) { o0[r] = T(0); o1[r] = T(0); }
        const T* ap = ain + arow * as + 4 * (w * KSW) + ak;
        T avo[2][RB];                                  // A operands one k-step ahead, in the MFMAs' shadows
#pragma unroll
        for (int r = 0; r < RB; ++r) avo[0][r] = ap[4 * r * as];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
          if (ks + 1 < KSW) {
#pragma unroll
            for (int r = 0; r < RB; ++r) avo[(ks + 1) & 1][r] = ap[4 * r * as + 4 * (ks + 1)];
          }
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const T a = avo[ks & 1][r];
            o0[r] = mfma4(a, wout[ks][0], o0[r]);
            if (tiles > 1) o1[r] = mfma4(a, wout[ks][1], o1[r]);
          }
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * RB - RB, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          part[(w * ROWS + 4 * r + drow) * nxp + dcol] = o0[r];
          if (tiles > 1) part[(w * ROWS + 4 * r + drow) * nxp + 16 + dcol] = o1[r];
        }
      }
      AMPC_LSW_MARK(pm, pm_on, 6);
      lds_barrier();
      AMPC_LSW_MARK(pm, pm_on, 7);
    }
    AMPC_LSW_MARK(pm, pm_k && pass == 0, 10);
    // ---- objectives of this wave's rows (ilqr.py:141-149, 206): dt * stage costs + terminal cost, from the stored
    // trajectory, one time step per lane -- kept off the serial chain of the rollout above
    __syncthreads();                                 // (orders this workgroup's trajectory stores)
    {
      // (the rows' sums side by side, time step outermost: three independent chains of loads and FMAs)
      T obj_part[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) obj_part[r] = T(0);
      for (int t = lane; t <= H; t += 64) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          if (!livew[r]) continue;
          const T* xsrc = mode == 0 ? stw : lss + (size_t)jw[r] * (HS + 1) * nx;
          const T* usrc = mode == 0 ? ctw : lsc + (size_t)jw[r] * HS * nu;
          const T* xt = xsrc + (size_t)t * nx;
          if (t < H) obj_part[r] += args.dt * (quad_rows<T>(Qm, xt, goal, no, 0, 1, cdiag) +
                                               quad_rows<T>(Rm, usrc + (size_t)t * nu, nullptr, nu, 0, 1, cdiag));
          else obj_part[r] += quad_rows<T>(Fm, xt, goal, no, 0, 1, cdiag);
          if (caff) {
            if (t < H) obj_part[r] += args.dt * affine_rows<T>(clin, xt, goal, no, 0, 1, clint[no]);
            else obj_part[r] += affine_rows<T>(clint, xt, goal, no, 0, 1, clint[no + 1]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        T o = obj_part[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) o += __shfl_xor(o, off);
        if (lane == 0) lsobj[ROWS * pass + 4 * r + w] = o;
      }
    }
    __syncthreads();
    AMPC_LSW_MARK(pm, pm_k && pass == 0, 11);
    if (pass == 0) AMPC_LSW_DUMP(pm, w, 12);

    if (mode == 0) {
      if (tid == 0) {
        args.obj[p] = lsobj[0];
        args.active[p] = 1; args.converged[p] = 0; args.iters[p] = 0; args.status[p] = 0;
        args.refresh[p] = 1; args.ls_rows[p] = 0; args.ls_count[p] = 0;
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
    { const int ro = ROWS * (last / ROWS + 1); args.ls_rows[p] += ro < rows ? ro : rows; }   // candidates rolled out
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
