// ilqr_wide.hpp -- iLQR backward sweep on wide LINEAR models (65 .. 128 model states), gfx950.
//
// The reference's iLQR runs on any Model (ilqr.py:144-147, 228-235); its ARX model with the DEFAULT
// history 4 (arx.py:27,37-45) on the 18-observation / 6-control HalfCheetah already has 91 states, Koopman
// lifts grow the same way (koopman.py:105-122).  Up to 64 states a linear model rides the MLP kernels
// (ampc_set_linear); beyond that the Riccati matrices of a problem no longer fit the layouts of
// ilqr_kernels.hpp (V, J, VJ, Qt in LDS: four n x n matrices).  A linear model needs less: its Jacobian
// J = [A | B] is ONE constant matrix for every time step and every problem (arx.py:156-164), so it stays
// in global memory (zero padded to MFMA tiles once, LinDev::jp: a few hundred KB, L2 resident), there is
// no Jacobian refresh, and a step of the sweep (ilqr.py:159-187) is
//
//   A  VJ = V J                      V [ns x ns] from LDS, J from L2, VJ to a per-problem global scratch
//      qt = ct + J' v                (matrix-vector, one thread per entry)
//   B  Qt = Ct + J' VJ               upper-triangular 16 x 16 tiles only (Qt is symmetric up to rounding);
//                                    Qxx lands in V's storage (V is dead once VJ is complete), Qux / Quu
//                                    in small LDS blocks
//   C  Quu [K | k] = -[Qux | qu]     every lane solves on a private register copy of Quu for its own
//                                    right-hand side (LDL', LU with partial pivoting when Quu is not safely
//                                    positive definite: ilqr_kernels.hpp), Z = Qux + Quu K
//   D  V = Qxx + [Qxu | K'] [K ; Z]  rank-2 nu update from LDS;  v = qx + Qxu k + K' (qu + Quu k)
//
// on 16x16x4 MFMA tiles, eight waves per problem.  Per step 2 ns^2 n + ns n^2 flops (ns = 91, nu = 6:
// 2.5 MFLOP, ~8 us on the one CU a problem occupies); problems are independent workgroups.  The line
// search is ilqr_iter_kernel<.., DYN = 2> (ilqr_kernels.hpp) with the K-tiled linear step of
// linear_kernels.hpp.  State, flags, queue / per-slot-horizon protocol: exactly the other sweeps'.
#pragma once
#include "ilqr_kernels.hpp"
#include "linear_kernels.hpp"

namespace ampc {


struct WideLds {
  int V, ldV, vv, qt, K, Z, ldB, Qux, Quu, goal, xbar, ubar, scal, total;
};
__host__ __device__ constexpr WideLds make_wide_lds(int ns, int nu, int no) {
  WideLds L{};
  const int nsp = (ns + 15) / 16 * 16, n = ns + nu, nu4 = (nu + 3) / 4 * 4;
  int o = 0;
  L.ldV = nsp + 1;
  L.V = o; o += nsp * L.ldV;
  L.vv = o; o += nsp;
  L.qt = o; o += (n + 3) / 4 * 4;
  L.ldB = nsp + 1;
  L.K = o; o += nu4 * L.ldB;
  L.Z = o; o += nu4 * L.ldB;
  L.Qux = o; o += nu4 * L.ldB;
  L.Quu = o; o += nu4 * nu4;
  L.goal = o; o += 3 * no;             // goal | lin | lint
  L.xbar = o; o += nsp;
  L.ubar = o; o += nu4;
  L.scal = o; o += 8;
  L.total = (o + 3) / 4 * 4;
  return L;
}

template <typename T, int NU>
__global__ __launch_bounds__(kRicThreads) void ilqr_riccati_wide_kernel(const IlqrArgs<T> args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Wr = reinterpret_cast<T*>(smem_raw);
  using acc_t = typename Acc<T>::type;
  constexpr int NTHR = kRicThreads, NW = NTHR / 64, nu = NU;
  const LinDev<T> lin = args.lin;
  const int tid = threadIdx.x, p = blockIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int ns = lin.nx, n = ns + nu, no = args.obs_dim, nsp = lin.nxp, ldJ = lin.ldj;
  const int HS = args.H, H = args.slot_h ? args.slot_h[p] : HS;      // array stride, this slot's horizon
  if (args.active[p] == 0) return;
  const WideLds L = make_wide_lds(ns, nu, no);
  T* V = Wr + L.V; T* vv = Wr + L.vv; T* qt = Wr + L.qt; T* Km = Wr + L.K; T* Zm = Wr + L.Z;
  T* Qux = Wr + L.Qux; T* Quu = Wr + L.Quu; T* goal = Wr + L.goal;
  T* xbar = Wr + L.xbar; T* ubar = Wr + L.ubar; T* scal = Wr + L.scal;
  const int ldV = L.ldV, ldB = L.ldB, nu4 = (nu + 3) / 4 * 4;
  const T* cpar = args.costs_par + (size_t)args.cost_idx[p] * args.cost_stride;      // Q R F goal lin lint
  const T* st = args.states + (size_t)p * (HS + 1) * ns;
  const T* ct = args.ctrls + (size_t)p * HS * nu;
  T* Kg = args.Ks + (size_t)p * HS * nu * ns;
  T* kg = args.ks + (size_t)p * HS * nu;
  T* vj = args.vj + (size_t)p * nsp * ldJ;            // this problem's VJ scratch [nsp][ldJ]
  const T* Jp = lin.jp;                               // [nsp][ldJ], zero padded
  const T dt = args.dt;
  for (int i = tid; i < L.total; i += NTHR) Wr[i] = T(0);
  __syncthreads();
  // symmetrised cost Hessians Q + Q', R + R' (cost.py:181-211), read from the cost block where they are
  // needed (L2 resident; at 128 states the value function's Hessian leaves no LDS for them)
  const T* Rp = cpar + no * no;
  auto CQ = [&](int a, int b) { return cpar[a * no + b] + cpar[b * no + a]; };
  auto CR = [&](int a, int b) { return Rp[a * nu + b] + Rp[b * nu + a]; };
  for (int i = tid; i < 3 * no; i += NTHR) goal[i] = cpar[2 * no * no + nu * nu + i];
  const T* clin = goal + no; const T* clint = clin + no;
  const T* Fp = cpar + no * no + nu * nu;
  for (int i = tid; i < no * no; i += NTHR) {         // V_H = F + F' on the observed block
    const int a = i / no, b = i - a * no;
    V[a * ldV + b] = Fp[a * no + b] + Fp[b * no + a];
  }
  __syncthreads();
  for (int a = tid; a < no; a += NTHR) {              // v_H (term_goal == 0: the reference's goal-less gradient)
    T s = T(0);
    for (int b = 0; b < no; ++b) s += V[a * ldV + b] * (st[(size_t)H * ns + b] - (args.term_goal ? goal[b] : T(0)));
    if (args.term_goal) s += clint[a];
    vv[a] = s;
  }
  for (int i = tid; i < ns; i += NTHR) xbar[i] = st[(size_t)(H - 1) * ns + i];
  for (int i = tid; i < nu; i += NTHR) ubar[i] = ct[(size_t)(H - 1) * nu + i];
  __syncthreads();
  const int ksn = nsp / 4, mts = nsp / 16, ntn = ldJ / 16;
  T lin_s = T(0), quad_s = T(0), ksn2 = T(0);         // meaningful in thread ns only
  int sing_any = 0;
  for (int t = H - 1; t >= 0; --t) {
    // ---- A: VJ = V J (to the global scratch);  qt = ct + J' v
    for (int tile = w; tile < mts * ntn; tile += NW) {
      const int mt = tile / ntn, nt = tile - mt * ntn;
      const T* a = V + (16 * mt + i16) * ldV + q;
      const T* b = Jp + (size_t)q * ldJ + 16 * nt + i16;
      acc_t acc = {0, 0, 0, 0};
      int ks = 0;
      for (; ks + 4 <= ksn; ks += 4) {                // four fragments in flight
        const T b0 = b[(size_t)(4 * ks) * ldJ], b1 = b[(size_t)(4 * ks + 4) * ldJ], b2 = b[(size_t)(4 * ks + 8) * ldJ],
                b3 = b[(size_t)(4 * ks + 12) * ldJ];
        const T a0 = a[4 * ks], a1 = a[4 * ks + 4], a2 = a[4 * ks + 8], a3 = a[4 * ks + 12];
        acc = mfma16(a0, b0, acc); acc = mfma16(a1, b1, acc); acc = mfma16(a2, b2, acc); acc = mfma16(a3, b3, acc);
      }
      for (; ks < ksn; ++ks) acc = mfma16(a[4 * ks], b[(size_t)(4 * ks) * ldJ], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) vj[(size_t)(16 * mt + acc_row<T>(q, r)) * ldJ + 16 * nt + i16] = acc[r];
    }
    for (int c = tid; c < n; c += NTHR) {
      T s = T(0);
      for (int a = 0; a < ns; ++a) s += Jp[(size_t)a * ldJ + c] * vv[a];
      T cc = T(0);
      if (c < no) {
        for (int b = 0; b < no; ++b) cc += CQ(c, b) * (xbar[b] - goal[b]);
        cc += clin[c];
      } else if (c >= ns) {
        for (int j = 0; j < nu; ++j) cc += CR(c - ns, j) * ubar[j];
      }
      qt[c] = cc * dt + s;
    }
    __syncthreads();                                  // (VJ: global stores visible to the workgroup)
    // ---- B: Qt = Ct + J' VJ, upper-triangular tiles; Qxx -> V, Qux, Quu
    for (int tile = w; tile < ntn * ntn; tile += NW) {
      const int mt = tile / ntn, nt = tile - mt * ntn;
      if (nt < mt || 16 * mt >= n || 16 * nt >= n) continue;
      const T* a = Jp + (size_t)q * ldJ + 16 * mt + i16;
      const T* b = vj + (size_t)q * ldJ + 16 * nt + i16;
      acc_t acc = {0, 0, 0, 0};
      int ks = 0;
      for (; ks + 4 <= ksn; ks += 4) {
        const size_t o0 = (size_t)(4 * ks) * ldJ, o1 = o0 + 4 * (size_t)ldJ, o2 = o1 + 4 * (size_t)ldJ, o3 = o2 + 4 * (size_t)ldJ;
        const T a0 = a[o0], a1 = a[o1], a2 = a[o2], a3 = a[o3];
        const T b0 = b[o0], b1 = b[o1], b2 = b[o2], b3 = b[o3];
        acc = mfma16(a0, b0, acc); acc = mfma16(a1, b1, acc); acc = mfma16(a2, b2, acc); acc = mfma16(a3, b3, acc);
      }
      for (; ks < ksn; ++ks) acc = mfma16(a[(size_t)(4 * ks) * ldJ], b[(size_t)(4 * ks) * ldJ], acc);
      const int s = 16 * nt + i16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * mt + acc_row<T>(q, r);
        if (c >= n || s >= n || s < c) continue;      // (diagonal tiles: the upper half, mirrored below)
        T val = acc[r];
        if (c < no && s < no) val += CQ(c, s) * dt;
        else if (c >= ns && s >= ns) val += CR(c - ns, s - ns) * dt;
        if (s < ns) { V[c * ldV + s] = val; V[s * ldV + c] = val; }              // Qxx (c <= s < ns)
        else if (c < ns) Qux[(s - ns) * ldB + c] = val;                           // Qxu[c][s-ns] = Qux[s-ns][c]
        else { Quu[(c - ns) * nu4 + (s - ns)] = val; Quu[(s - ns) * nu4 + (c - ns)] = val; }
      }
    }
    __syncthreads();
    // ---- C: the nu x nu solves, one right-hand side per thread (columns 0..ns-1: Qux, column ns: qu)
    if (tid <= ns) {
      T A0[NU][NU], x[NU], x0[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
#pragma unroll
        for (int j = 0; j < NU; ++j) A0[i][j] = Quu[i * nu4 + j];
        x0[i] = tid < ns ? Qux[i * ldB + tid] : qt[ns + i];
        x[i] = x0[i];
      }
      int ok = ldl_solve_lane<T, NU>(A0, x);
      if (!ok) {                                       // (identical in every thread: Quu is shared)
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
        Km[i * ldB + tid] = x[i];                      // column ns: k
        Zm[i * ldB + tid] = x0[i] + s;                 // column ns: z = qu + Quu k
      }
      if (tid == ns) { lin_s += l; quad_s += qd; ksn2 += k2; }
    }
    __syncthreads();
    // ---- D: V = Qxx + Qxu K + K' Z (Z = Qux + Quu K);  v = qx + Qxu k + K' z;  store K_t, k_t
    for (int idx = tid; idx < ns * ns; idx += NTHR) {
      const int a = idx / ns, b = idx - a * ns;
      T s = V[a * ldV + b];
#pragma unroll
      for (int j = 0; j < NU; ++j) s += Qux[j * ldB + a] * Km[j * ldB + b] + Km[j * ldB + a] * Zm[j * ldB + b];
      V[a * ldV + b] = s;                              // (every thread reads and writes only its own entries of V)
    }
    for (int a = tid; a < ns; a += NTHR) {
      T s = qt[a];
#pragma unroll
      for (int j = 0; j < NU; ++j) s += Qux[j * ldB + a] * Km[j * ldB + ns] + Km[j * ldB + a] * Zm[j * ldB + ns];
      vv[a] = s;
    }
    for (int i = tid; i < nu * ns; i += NTHR) {
      const int j = i / ns, b = i - j * ns;
      Kg[(size_t)t * nu * ns + i] = Km[j * ldB + b];
    }
    if (tid < nu) kg[(size_t)t * nu + tid] = Km[tid * ldB + ns];
    if (t > 0) {
      for (int i = tid; i < ns; i += NTHR) xbar[i] = st[(size_t)(t - 1) * ns + i];
      for (int i = tid; i < nu; i += NTHR) ubar[i] = ct[(size_t)(t - 1) * nu + i];
    }
    __syncthreads();
  }
  if (sing_any) scal[0] = T(1);
  __syncthreads();
  if (tid == ns) {
    T* out = args.ric + (size_t)p * kRicStride;
    const T sg = scal[0];
    out[0] = lin_s; out[1] = quad_s; out[2] = sqrt(ksn2); out[3] = sg;
    if (sg != T(0)) {             // singular Quu: the reference raises LinAlgError here
      args.status[p] = 1; args.active[p] = 0; args.refresh[p] = 0;
    }
  }
}

}  // namespace ampc
