// mlp_kernels.hpp -- batched MLP dynamics step and its analytic Jacobian (gfx950).
//
// pred_batch      (reference: autompc/sysid/mlp.py:229-236)   -> mlp_forward_kernel
// pred_diff_batch (reference: autompc/sysid/mlp.py:281-305)   -> mlp_forward_kernel<DERIV> +
//                                                                mlp_jacobian_kernel
// The reference obtains d(net)/d(input) by running autograd over an nx-fold repeated batch.
// Here it is the chain  J = W_out' D_L W_L ... D_1 W_1' (+ I on the state block) on the folded
// weights (mlp_tile.hpp), evaluated left to right on MFMA with the rows (sample s, output i) of
// all samples flattened into one tall matrix so the 16-row MFMA tiles carry no padding:
// G_L[(s,i)][k] = W_out'[i][k] d_L[s][k], then per layer G_{l-1} = (G_l W_l) * d_{l-1}[s],
// finally J = G_1 W_1' + [c == i].
#pragma once
#include "mlp_tile.hpp"

namespace ampc {

// Row r of a batched model call lives at states + (r / grp) * s_stride + (r % grp) * nx (and the
// same with c_stride / nu for controls).  A plain [n][nx] batch is grp = n, strides 0.  iLQR uses
// grp = H with the [B][H+1][nx] / [B][H][nu] trajectory layout.  mask (optional) is per group:
// groups with mask[g] == 0 are skipped by the Jacobian kernel (their Jacobians stay untouched).
// glen (optional) is per group too: only the first glen[g] rows of group g exist (iLQR slots whose
// horizon is shorter than the plan's, ampc_ilqr_solve_queue_var); the others are skipped like masked ones.
// grp_pad (optional, a multiple of the tile height): the kernels' row index runs over groups PADDED to
// grp_pad rows -- row v is step v % grp_pad of group v / grp_pad and exists for steps < grp -- so that no
// tile straddles two groups; with it grp_model[g] names the group's entry of the table model_delta
// (iLQR slots that carry different controller models, ampc_ilqr_plan_set_models; mlp_tile.hpp).  Outputs (jx, ju, out)
// are indexed by the UNPADDED row g * grp + step either way; the stored derivatives by the kernels' row.
struct RowMap {
  int grp;
  long long s_stride, c_stride;
  const int* mask;
  const int* glen;
  int grp_pad;
  const int* grp_model;
  const long long* model_delta;
  int n_groups;          // groups (slots) of the plan: perm has n_groups + 1 entries
  const int* perm;       // group g of the row index space is slot perm[g] (iLQR queue with more slots than CUs: the
                         // slots with work first, ilqr_compact_kernel -- tiles without a live row bunch at the end
                         // of the grid instead of being interleaved with the live ones); nullptr: identity
  // (perm[number of groups] = how many of them have work: a tile that starts past those has no live row)
  __device__ __forceinline__ bool past_work(int first_row, int n_groups) const {
    return perm != nullptr && first_row / gp() >= perm[n_groups];
  }
  __device__ __forceinline__ bool plain() const { return mask == nullptr && glen == nullptr && grp_pad == 0 && perm == nullptr; }
  __device__ __forceinline__ int gp() const { return grp_pad > 0 ? grp_pad : grp; }
  __device__ __forceinline__ int slot(int g) const { return perm != nullptr ? perm[g] : g; }
  __device__ __forceinline__ int group(int v) const { return slot(v / gp()); }      // the slot row v belongs to
  __device__ __forceinline__ int step(int v) const { return v % gp(); }
  __device__ __forceinline__ long long out_row(int v) const { return (long long)group(v) * grp + step(v); }
  __device__ __forceinline__ bool live(int v) const {
    const int t = v % gp(), g = group(v);
    return t < grp && (mask == nullptr || mask[g] != 0) && (glen == nullptr || t < glen[g]);
  }
  // some row of [first, last] is live
  __device__ __forceinline__ bool any_live(int first, int last) const {
    const int G = gp();
    for (int vg = first / G; vg <= last / G; ++vg) {
      const int g = slot(vg);
      if (mask != nullptr && mask[g] == 0) continue;
      const int len = glen != nullptr ? (glen[g] < grp ? glen[g] : grp) : grp;
      const int lo = first > vg * G ? first : vg * G;
      const int hi = vg * G + len - 1;
      if (lo <= (hi < last ? hi : last)) return true;
    }
    return false;
  }
  // byte offset of the model the tile starting at row `first` runs on
  __device__ __forceinline__ long long delta(int first) const {
    return model_delta_of(model_delta, model_delta != nullptr ? grp_model[group(first)] : 0);
  }
};

template <typename T, int NT, int MT, int W, bool DERIV, typename SH = DynShape, bool WIDE = false>
__global__ __launch_bounds__(64 * W) void mlp_forward_kernel(const MlpDev<T> mlp_in, const TileLds L_in,
                                                             const T* __restrict__ states,
                                                             const T* __restrict__ ctrls,
                                                             T* __restrict__ out,
                                                             T* __restrict__ dz, int n, int n_pad,
                                                             const RowMap rm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  using Net = TileNet<T, NT, MT, W, DERIV, 0, SH, WIDE>;
  constexpr int M = 16 * MT, NTHR = 64 * W;
  const int first = blockIdx.x * M;
  const MlpDev<T> mlp = plan_model<SH, T>(mlp_in, [&] { return rm.delta(first); });   // (per-group models)
  const TileLds L = SH::template fold_lds<T, M, W>(L_in);
  const int tid = threadIdx.x, nx = mlp.nx, nu = mlp.nu;
  T* xu = lds + L.xu;
  if (!rm.plain() && first < n) {          // (iLQR refresh) every row of this tile is masked out
    if (rm.past_work(first, rm.n_groups) || !rm.any_live(first, first + M - 1 < n ? first + M - 1 : n - 1)) return;
  }
  Net net;
  net.init(mlp);
  tile_load_constants<T, W>(mlp, L, lds, M);
  __syncthreads();
  for (int i = tid; i < M * nx; i += NTHR) {
    const int row = i / nx, col = i - row * nx;
    const int gr = first + row;
    xu[row * L.xu_stride + col] =
        (gr < n && ((rm.glen == nullptr && rm.grp_pad == 0) || rm.live(gr)))    // (rows past a slot's horizon: never written)
            ? states[(size_t)rm.group(gr) * rm.s_stride + (size_t)rm.step(gr) * nx + col] : T(0);
  }
  for (int i = tid; i < M * nu; i += NTHR) {
    const int row = i / nu, col = i - row * nu;
    const int gr = first + row;
    xu[row * L.xu_stride + nx + col] =
        (gr < n && ((rm.glen == nullptr && rm.grp_pad == 0) || rm.live(gr)))
            ? ctrls[(size_t)rm.group(gr) * rm.c_stride + (size_t)rm.step(gr) * nu + col] : T(0);
  }
  __syncthreads();
  // dz layout: [layer][n_pad][hpad]; this tile's rows start at `first`
  net.run(mlp, L, lds, DERIV ? dz + (size_t)first * mlp.hpad : nullptr, (size_t)n_pad * mlp.hpad);
  for (int i = tid; i < M * nx; i += NTHR) {
    const int row = i / nx, col = i - row * nx;
    if (out != nullptr && first + row < n && (rm.grp_pad == 0 || rm.step(first + row) < rm.grp))
      out[(size_t)rm.out_row(first + row) * nx + col] = xu[row * L.xu_stride + col] + Net::output(mlp, L, lds, row, col);
  }
}

// K-split small-N stage used by the Jacobian's last product: acc[mt][n] over this wave's k range.
template <typename T, int MT, int KSW, int NMAX>
__device__ __forceinline__ void ksplit_mma(const T* __restrict__ arow, int as,
                                           const T* __restrict__ wl, int n_tiles,
                                           typename Acc<T>::type (&acc)[MT][NMAX]) {
#pragma unroll
  for (int ks = 0; ks < KSW; ++ks) {
    T a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt * 16 * as + 4 * ks];
#pragma unroll
    for (int nn = 0; nn < NMAX; ++nn)
      if (nn < n_tiles) {
        const T b = wl[(size_t)ks * 64 * n_tiles + nn];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nn] = mfma16(a[mt], b, acc[mt][nn]);
      }
  }
}

// Tile = 16*MT samples x ONE output index i (blockIdx -> tile: see the XCD-aware map below).  Every row of a
// tile then shares the W_out row (a broadcast) and maps to its sample without a division.
// wout_plain: folded output weights [nx][hpad] (zero padded); dz: [layer][n_pad][hpad] from
// mlp_forward_kernel<DERIV>.  jx[n][nx][nx], ju[n][nx][nu].
// (16-row, 8-wave tiles ask for >= 6 waves per SIMD, i.e. <= 80 VGPRs: three workgroups per CU
// overlap one tile's global loads with the others' MFMAs; measured +4 % on c4 over the default 88)
template <typename T, int NT, int MT, int W, typename SH = DynShape, bool WIDE = false>
__global__ __launch_bounds__(64 * W, (MT == 1 && W == 8) ? 6 : 1) void mlp_jacobian_kernel(const MlpDev<T> mlp_in,
                                                              const T* __restrict__ wout_plain_in,
                                                              const T* __restrict__ dz, int n,
                                                              int n_pad, T* __restrict__ jx,
                                                              T* __restrict__ ju, const RowMap rm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* G = reinterpret_cast<T*>(smem_raw);
  using acc_t = typename Acc<T>::type;
  using Net = TileNet<T, NT, MT, W, false>;
  constexpr int M = 16 * MT, NTHR = 64 * W;
  constexpr int NIMAX = WIDE ? 5 : 3;  // kin <= 48 (80 when WIDE)
  constexpr int KSH = Net::KSH, KSW = Net::KSW, GH = Net::GH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int nx = SH::template fold<T>(mlp_in).nx;       // (every model of a plan has the plan's shape)
  constexpr int HP = Net::HP;                      // == hpad for this instantiation
  const int gs = HP + (sizeof(T) == 8 ? 1 : 2);    // same padding rule as TileLds::act_stride
  // blockIdx -> (sample block, output index), XCD-aware: workgroups are dealt round-robin to the 8
  // XCDs (blockIdx % 8), each with its own L2.  The nx tiles of one sample block read the same
  // stored activation derivatives, so they are given block indices of one residue mod 8 -- one
  // XCD, one HBM fetch of that data instead of up to eight.  (Sample blocks beyond the last
  // multiple of 8 keep the plain order.)
  int i_out, sblk;
  {
    const int nblk = gridDim.x / nx, full = nblk / 8 * 8, b = blockIdx.x;
    if (b < full * nx) {
      const int xcd = b & 7, q8 = b >> 3;          // q8 counts this XCD's tiles
      i_out = q8 % nx;
      sblk = (q8 / nx) * 8 + xcd;
    } else {
      const int r = b - full * nx;
      i_out = r % nx;
      sblk = full + r / nx;
    }
  }
  const int s0 = sblk * M;                         // the tile's first sample
  const MlpDev<T> mlp = plan_model<SH, T>(mlp_in, [&] { return rm.delta(s0 < n ? s0 : 0); });
  const int nu = mlp.nu, hpad = mlp.hpad, Lh = mlp.n_hidden;
  // (the folded output weights lie in the model's buffer as well: the same byte offset applies)
  const T* wout_plain = reinterpret_cast<const T*>(reinterpret_cast<const char*>(wout_plain_in) + mlp.delta);
  const size_t lstride = (size_t)n_pad * hpad;
  if (!rm.plain()) {                   // every row this tile touches is masked out: nothing to refresh
    if (s0 >= n || rm.past_work(s0, rm.n_groups) || !rm.any_live(s0, s0 + M - 1 < n ? s0 + M - 1 : n - 1)) return;
  }

  // G_L[s][k] = W_out'[i][k] * d_L[s][k]
  for (int e = tid; e < M * HP; e += NTHR) {
    const int row = e / HP, k = e % HP;
    const int sidx = s0 + row;
    T v = T(0);
    if (sidx < n)
      v = wout_plain[i_out * hpad + k] * dz[(size_t)(Lh - 1) * lstride + (size_t)sidx * hpad + k];
    G[row * gs + k] = v;
  }
  __syncthreads();

  for (int l = Lh - 1; l >= 1; --l) {  // hidden->hidden layer l (torch index), uses wj[l]
    acc_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{0, 0, 0, 0};
    const rsrc_t wr = weight_rsrc(mlp.WB());
    const unsigned wl = (unsigned)(mlp.WJ(l) - mlp.WB()) + (unsigned)w * KSH * 64u * NT;   // uniform
    T first_group[GH][NT];
    load_group<T, NT, GH>(wr, wl, (unsigned)lane * NT, 0, first_group);
    // (round 5: the rollout's issue pattern -- fragment reads one k-step pair ahead, weight loads spread between
    // MFMA clusters -- also pays here despite three workgroups per CU: 0.369 -> 0.350 ms per c4 launch.  Fetching
    // the activation derivatives before the MFMAs instead of after them measured slower: 8 more live VGPRs.)
    layer_mma_static<T, NT, MT, KSH, GH, Probe::jac_pipe>(G, gs, wr, wl, lane, first_group, acc);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * (NT * w + nt) + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + acc_row<T>(q, r);
          T d = T(0);
          if (s0 + row < n) d = dz[(size_t)(l - 1) * lstride + (size_t)(s0 + row) * hpad + col];
          G[row * gs + col] = acc[mt][nt][r] * d;
        }
      }
    __syncthreads();
  }

  // last product: J_net = G_1 W_1'  (K = hpad split over waves, N = kin padded to 16*ni)
  const int ni = (mlp.kin + 15) / 16;
  const int kinp = 16 * ni;
  acc_t oacc[MT][NIMAX];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nn = 0; nn < NIMAX; ++nn) oacc[mt][nn] = acc_t{0, 0, 0, 0};
  ksplit_mma<T, MT, KSW, NIMAX>(G + i16 * gs + q + 4 * w * KSW, gs,
                                mlp.WJ(0) + ((size_t)w * KSW * 64 + lane) * ni, ni, oacc);
  __syncthreads();
  T* part = G + w * M * kinp;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nn = 0; nn < NIMAX; ++nn)
      if (nn < ni) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          part[(16 * mt + acc_row<T>(q, r)) * kinp + 16 * nn + i16] = oacc[mt][nn][r];
      }
  __syncthreads();
  for (int e = tid; e < M * mlp.kin; e += NTHR) {
    const int row = e / mlp.kin, c = e - row * mlp.kin;
    const int s = s0 + row, i = i_out;
    if (s >= n) continue;
    if (!rm.plain() && !rm.live(s)) continue;
    T v = T(0);
#pragma unroll
    for (int ww = 0; ww < W; ++ww) v += G[ww * M * kinp + row * kinp + c];
    const size_t so = (size_t)rm.out_row(s);
    if (c < nx) jx[(so * nx + i) * nx + c] = v + (c == i ? T(1) : T(0));
    else ju[(so * nx + i) * nu + (c - nx)] = v;
  }
}

}  // namespace ampc
