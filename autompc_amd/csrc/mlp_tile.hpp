// mlp_tile.hpp -- fused MLP dynamics step on one workgroup-resident tile of samples (gfx950).
//
// A workgroup is W wave64 (W = 8: two waves per SIMD, or W = 4 for narrow nets).  A tile is
// M = 16*MT samples.  Hidden width is padded to HPAD = 16*NT*W so each wave owns NT 16-column
// MFMA tiles of every hidden layer (N-split); the small output layer is K-split across the W
// waves and reduced through LDS.  Arithmetic is exact f64 (v_mfma_f64_16x16x4_f64) or exact f32
// (v_mfma_f32_16x16x4_f32): k-ordered fma chains, no reduced-precision inputs.
//
// Operand sources
//   A (activations)  LDS, row-major [M][K+pad]: pad = 2 (f32) / 1 (f64) keeps the 16-row fragment
//                    reads (ds_read2_b32 / ds_read2_b64) free of bank conflicts.
//   B (weights)      global memory, pre-packed on the host in exact fragment order, so each wave
//                    reads one contiguous, fully coalesced run per k-step.  Weights are re-read
//                    every time step but stay L2-resident (<= 640 KB per model).  Fragments are
//                    double-buffered in registers in groups of G k-steps, and the FIRST group of
//                    every layer is requested before the barrier that precedes that layer, so
//                    the L2 round trip overlaps the previous layer's epilogue.
//
// The math (reference: autompc/sysid/mlp.py:20-30, :55-59, :229-236):
//   x' = x + dy_mean + dy_std * net(([x,u] - xu_mean) / xu_std)
// The affine normalisers are folded into the first and last layer on the host (double
// precision): W1' = W1 diag(1/xu_std), b1' = b1 - W1' xu_mean, W3' = diag(dy_std) W3,
// b3' = dy_std*b3 + dy_mean, so the kernel computes x' = x + net'([x,u]) with no per-step
// normalisation work; the Jacobian chain on the folded weights is already the scaled Jacobian.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace ampc {

constexpr int kMaxHidden = 4;  // hidden layers supported (reference config space: 1..4)
constexpr int kMaxWaves = 8;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Acc;
template <> struct Acc<double> { using type = d4; };
template <> struct Acc<float> { using type = f4; };

// AMPC_X_* macros are timing experiments only (tools/variants.sh); never defined in the product.
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
#ifdef AMPC_X_NOMFMA
  c[0] += a * b;
  return c;
#else
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
#ifdef AMPC_X_NOMFMA
  c[0] += a * b;
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products (blocks); 16 cycles against the 64 of
// the 16x16x4.  Lane layout (probed on gfx950, tools/mfma44_probe.cpp), block = (lane/4)%4:
//   A[i][k]: i = lane%4, k = lane/16      B[k][j]: k = lane/16, j = lane%4
//   D[i][j]: i = lane/16, j = lane%4
// With rows 4*block + i the A operand is act[row = lane%16][k = lane/16] -- exactly the fragment
// the 16x16x4 MFMA takes -- so the same register feeds both; B is shared by the four blocks and
// D gives 16 rows x 4 columns, one value per lane.
__device__ __forceinline__ double mfma4(double a, double b, double c) {
#ifdef AMPC_X_NOMFMA
  return c + a * b;
#else
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ float mfma4(float, float, float c) { return c; }   // f64 only (tail4 is never set for f32)
// row / column (relative to 16) of the value mfma4 leaves in this lane
__device__ __forceinline__ int tail_row(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int tail_col(int lane) { return lane & 3; }

// Row of accumulator register r held by lane-quad q (= lane >> 4); column is lane & 15.
//   f64 16x16x4: row = q + 4 r      f32 16x16x4: row = 4 q + r
template <typename T> __device__ __forceinline__ int acc_row(int q, int r);
template <> __device__ __forceinline__ int acc_row<double>(int q, int r) { return q + 4 * r; }
template <> __device__ __forceinline__ int acc_row<float>(int q, int r) { return 4 * q + r; }

// ---- activations (torch semantics: ReLU, Tanh, Sigmoid, SELU; mlp.py:44-51) -------------
// kind 4 = identity: a linear model x' = A x + B u (ARX arx.py:151-154, Koopman
// koopman.py:170-173) staged as a one-hidden-layer linear network.
template <typename T> __device__ __forceinline__ T act_apply(int kind, T z) {
  switch (kind) {
    case 0: return z > T(0) ? z : T(0);
    case 1: return tanh(z);
    case 2: return T(1) / (T(1) + exp(-z));
    case 4: return z;
    default: {
      const T alpha = T(1.6732632423543772848170429916717);
      const T scale = T(1.0507009873554804934193349852946);
      return scale * (z > T(0) ? z : alpha * expm1(z));
    }
  }
}
// derivative expressed from z (pre-activation)
template <typename T> __device__ __forceinline__ T act_deriv(int kind, T z) {
  switch (kind) {
    case 0: return z > T(0) ? T(1) : T(0);
    case 1: { T t = tanh(z); return T(1) - t * t; }
    case 2: { T s = T(1) / (T(1) + exp(-z)); return s * (T(1) - s); }
    case 4: return T(1);
    default: {
      const T alpha = T(1.6732632423543772848170429916717);
      const T scale = T(1.0507009873554804934193349852946);
      return z > T(0) ? scale : scale * alpha * exp(z);
    }
  }
}

// ---- device-side model descriptor ------------------------------------------------------------
template <typename T> struct MlpDev {
  int nx, nu, kin;      // state dim, ctrl dim, nx+nu
  int k1p;              // kin zero-padded to a multiple of 8, <= 48 (first-layer MFMA K)
  int n_hidden;         // hidden layers
  int hpad;             // 16*NT*W
  int nxp;              // nx rounded up to a multiple of 16
  int act;              // activation kind
  const T* w[kMaxHidden + 1];   // packed fragments, layer 0..n_hidden (last = output layer)
  const T* b[kMaxHidden + 1];   // padded biases (normalisers folded in)
  const T* wj[kMaxHidden + 1];  // packed fragments for the Jacobian chain (transposed use)
  // f64, 16 < nx <= 20: output columns 16..19 come from v_mfma_f64_4x4x4_4b (see mfma4 below)
  // instead of a second, mostly empty 16-column tile.  wt [W][KSW][64]: lane l of k-step ks holds
  // W_out'[k = 4 ks + l/16][col = 16 + l%4].
  const T* wt;
  int tail4;
};

// LDS carve-up shared by all kernels that run the tile (offsets in elements of T).
struct TileLds {
  int act;      // [M][hpad+pad]
  int act2;     // second activation buffer (ping-pong across layers); == act when LDS is tight
  int part;     // [W][M][nxp] output-layer partials; aliases `act` when LDS is tight
  int part_alias;
  int xu;       // [M][k1p+2]   raw state | control | zero pad: the first layer's A operand
  int bias;     // n_hidden*hpad + nxp
  int extra;    // kernel-specific region starts here
  int act_stride, xu_stride;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope
// release fence that hipcc lowers to s_waitcnt vmcnt(0): it would drain the weight prefetches
// this kernel deliberately keeps in flight across phase boundaries.  Every cross-wave hand-off in
// the tile goes through LDS, so waiting for this wave's LDS operations (lgkmcnt) is sufficient;
// the asm memory clobbers keep the compiler from moving LDS accesses across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ inline int round_up(int a, int m) { return (a + m - 1) / m * m; }

template <typename T>
__host__ inline TileLds make_tile_lds(const MlpDev<T>& m, int M, int W, bool separate_partials = true,
                                      bool double_act = true) {
  TileLds L;
  int o = 0;
  // Row padding for conflict-free A-fragment reads (16 rows x consecutive k per access):
  //   f32  ds_read(2)_b32 banks = dword mod 32 over 32-lane halves  -> stride = 2 (mod 4)
  //   f64  hipcc pairs k-steps into ds_read2_b64 (16-lane groups, banks = dword mod 32)
  //        -> odd stride in 8-byte units (measured: SQ_LDS_BANK_CONFLICT 40% -> ~0 of LDS cycles)
  const int pad = sizeof(T) == 8 ? 1 : 2;
  L.act_stride = m.hpad + pad;
  L.xu_stride = m.k1p + pad;
  L.part_alias = separate_partials ? 0 : 1;
  if (separate_partials) {
    L.act = o; o += M * L.act_stride;
    L.act2 = L.act;
    if (double_act && m.n_hidden > 1) { L.act2 = o; o += M * L.act_stride; }
    L.part = o; o += W * M * m.nxp;
  } else {
    L.act = o; L.act2 = o; L.part = o; o += M * imax(L.act_stride, W * m.nxp);
  }
  L.xu = o; o += M * L.xu_stride;
  L.bias = o; o += m.n_hidden * m.hpad + m.nxp;
  L.extra = round_up(o, 4);
  return L;
}

// Stage biases into LDS and zero the first-layer operand (call once, then __syncthreads()).
template <typename T, int W>
__device__ __forceinline__ void tile_load_constants(const MlpDev<T>& m, const TileLds& L, T* lds,
                                                    int M) {
  const int tid = threadIdx.x;
  for (int l = 0; l < m.n_hidden; ++l)
    for (int i = tid; i < m.hpad; i += 64 * W) lds[L.bias + l * m.hpad + i] = m.b[l][i];
  for (int i = tid; i < m.nxp; i += 64 * W) lds[L.bias + m.n_hidden * m.hpad + i] = m.b[m.n_hidden][i];
  for (int i = tid; i < M * L.xu_stride; i += 64 * W) lds[L.xu + i] = T(0);
}

// ---- weight fragment loads -------------------------------------------------------------------
// A lane's NT consecutive fragment values for one k-step, as the widest aligned vector load.
template <typename T, int N> using vec_t = T __attribute__((ext_vector_type(N)));

template <typename T, int NT>
__device__ __forceinline__ void load_frag(const T* __restrict__ p, T (&b)[NT]) {
#ifdef AMPC_X_NOLOAD
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b[nt] = T((threadIdx.x + nt) & 7) * T(1e-3);
  return;
#endif
  if constexpr (NT == 4) {
    const vec_t<T, 4> v = *reinterpret_cast<const vec_t<T, 4>*>(p);
    b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
  } else if constexpr (NT == 2) {
    const vec_t<T, 2> v = *reinterpret_cast<const vec_t<T, 2>*>(p);
    b[0] = v[0]; b[1] = v[1];
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = p[nt];
  }
}

// Fetch group `g` (G k-steps) of a wave's N-split fragment stream.  wl = layer base + this
// wave's slice + lane*NT.
template <typename T, int NT, int G>
__device__ __forceinline__ void load_group(const T* __restrict__ wl, int g, T (&b)[G][NT]) {
#pragma unroll
  for (int kk = 0; kk < G; ++kk) load_frag<T, NT>(wl + (size_t)(g * G + kk) * 64 * NT, b[kk]);
}

// One N-split layer with compile-time k extent KS: acc[mt][nt] += A[16mt.., :] * Wpacked.
// Fully unrolled; group 0 arrives pre-loaded in `first`, later groups are double-buffered so the
// fetch of group g+1 is in flight while group g's MFMAs issue.
// OWN: the fragment stream is packed starting at k-group `rot` (the group whose activations this
// wave itself produced, see TileNet::run); group 0 of the stream is consumed BEFORE the tile-wide
// barrier that publishes the other waves' activations, which is passed inside this function.
// SG: streaming granularity after the pre-loaded first group.  The double buffer holds two
// sub-groups of SG k-steps; tall f64 tiles (MT >= 2) have enough MFMA work per k-step to cover an
// L2 round trip with SG = 4, which halves the buffer's registers (64 VGPRs).
template <typename T, int NT, int MT, int KS, int G, bool PIPE = false, bool OWN = false, int SG = G>
__device__ __forceinline__ void layer_mma_static(const T* __restrict__ A, int a_stride,
                                                 const T* __restrict__ wl, int lane,
                                                 const T (&first)[G][NT],
                                                 typename Acc<T>::type (&acc)[MT][NT], int rot = 0) {
  static_assert(KS % G == 0 && G % SG == 0, "group sizes must divide the k extent");
  constexpr int NG = KS / G;       // groups (the unit of the rotated k order)
  constexpr int NS = KS / SG;      // sub-groups (the unit of the weight stream)
  constexpr int FS = G / SG;       // sub-groups that arrive pre-loaded in `first`
  static_assert(!OWN || (NG & (NG - 1)) == 0, "rotated k order needs a power-of-two group count");
  const int i = lane & 15, q = lane >> 4;
  const T* arow = A + i * a_stride + q;
  T b[2][SG][NT];
  if constexpr (FS == 1) {       // whole-group streaming: the pre-loaded group IS buffer 0
#pragma unroll
    for (int kk = 0; kk < SG; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[0][kk][nt] = first[kk][nt];
  }
#pragma unroll
  for (int sgi = 0; sgi < NS; ++sgi) {
    if (sgi + 1 >= FS && sgi + 1 < NS) load_group<T, NT, SG>(wl, sgi + 1, b[(sgi + 1) & 1]);
    const int g = sgi / FS;
    const T* ag = (OWN ? arow + 4 * G * ((g + rot) & (NG - 1)) : arow + 4 * G * g) + 4 * SG * (sgi % FS);
#pragma unroll
    for (int kk = 0; kk < SG; ++kk) {
      T a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = ag[mt * 16 * a_stride + 4 * kk];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const T bv = (FS > 1 && sgi < FS) ? first[sgi * SG + kk][nt] : b[sgi & 1][kk][nt];
          acc[mt][nt] = mfma16(a[mt], bv, acc[mt][nt]);
        }
    }
#ifndef AMPC_X_NOSCHED
    // (8-wave tiles only; measured neutral-to-negative with one wave per SIMD)
    // Issue order for this sub-group: LDS fragment reads run one k-step pair AHEAD of the MFMAs
    // that consume them, weight loads for the next sub-group are spread between MFMA clusters.
    //   masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read
    if (PIPE && KS >= 8) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
#pragma unroll
      for (int i = 0; i < SG / 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
    }
#endif
    if (OWN && sgi == FS - 1) lds_barrier();
  }
}

#ifdef AMPC_X_PHASETIME
__device__ long long g_phase_marks[64];
#define AMPC_MARK(idx)                                                              \
  do {                                                                              \
    if (blockIdx.x == 7 && threadIdx.x == 0 && g_phase_marks[63] == 1)              \
      g_phase_marks[idx] = (long long)__builtin_amdgcn_s_memtime();                 \
  } while (0)
// unconditional variant for coarse, once-per-kernel marks
#define AMPC_MARK_ALWAYS(idx)                                                       \
  do {                                                                              \
    if (blockIdx.x == 7 && threadIdx.x == 0)                                        \
      g_phase_marks[idx] = (long long)__builtin_amdgcn_s_memtime();                 \
  } while (0)
#else
#define AMPC_MARK(idx) do { } while (0)
#define AMPC_MARK_ALWAYS(idx) do { } while (0)
#endif

// ---- the fused network on one tile -------------------------------------------------------------
// LEAN: what stays resident in registers between calls, for callers that need registers
// themselves: 0 output fragments + hidden biases, 1 output fragments only, 2 nothing.
template <typename T, int NT, int MT, int W, bool DERIV = false, int LEAN = 0>
struct TileNet {
  using acc_t = typename Acc<T>::type;
  static constexpr int M = 16 * MT;
  static constexpr int HP = 16 * NT * W;      // padded hidden width
  static constexpr int KSH = HP / 4;          // k-steps of a hidden->hidden layer
  static constexpr int KSW = KSH / W;         // output-layer k-steps per wave (= 4*NT)
  static constexpr int KS0MAX = 12;           // first-layer k-steps (k1p/4 is 2, 4, .. 12)
  static constexpr int GH = 8;                // hidden-layer group
  static constexpr int NOMAX = 2;             // nx <= 32
  // A wave's own output columns [16 NT w, 16 NT (w+1)) are one whole k-group of the next hidden
  // layer: that layer starts on them before the barrier (see run()).  Host packing must agree
  // (own_first_packing() in api.cpp).
  static constexpr bool OWN = (16 * NT == 4 * GH) && (((KSH / GH) & (KSH / GH - 1)) == 0);

  // Layer 0's fragments requested ahead of time (prefetch0): the whole layer in f32 (its few
  // MFMAs cannot cover an in-loop L2 round trip; measured +3 %), only the first two k-steps in
  // f64, where the extra 40 live VGPRs cost more than the latency they hide (measured -4 %).
  static constexpr bool FULL0 = sizeof(T) == 4;
  T pf0[FULL0 ? KS0MAX : 2][NT];

  __device__ __forceinline__ static const T* slice0(const MlpDev<T>& m, int w, int lane) {
    return m.w[0] + ((size_t)w * (m.k1p / 4) * 64 + lane) * NT;
  }
  __device__ __forceinline__ static const T* slice_h(const MlpDev<T>& m, int l, int w, int lane) {
    return m.w[l] + ((size_t)w * KSH * 64 + lane) * NT;
  }
  template <int KS> __device__ __forceinline__ const T (&first0() const)[KS][NT] {
    return reinterpret_cast<const T(&)[KS][NT]>(pf0);
  }
  template <int KS> __device__ __forceinline__ T (&first0())[KS][NT] {
    return reinterpret_cast<T(&)[KS][NT]>(pf0);
  }

  // This wave's output-layer fragments (K-split: k-steps [w*KSW, (w+1)*KSW), up to two 16-column
  // tiles).  They are the same for every call, so they stay in registers for the kernel's
  // lifetime: fetching them per call put a 64 KB-per-CU burst on L2 right before the output
  // MFMAs needed them (measured ~1 us exposed per rollout step).
  // (Not in the 64-row f64 tile: its accumulators leave no room, the copy would spill.)
  static constexpr bool RESIDENT_OUT = LEAN < 2 && !(MT == 4 && sizeof(T) == 8);
  T wout[KSW][NOMAX];            // dead (never written or read) when not resident

  __device__ __forceinline__ static void load_out(const MlpDev<T>& m, int w, int lane,
                                                  T (&dst)[KSW][NOMAX]) {
    if (m.nxp == 16) {
      const T* wl = m.w[m.n_hidden] + ((size_t)w * KSW * 64 + lane);
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) { dst[ks][0] = wl[ks * 64]; dst[ks][1] = T(0); }
    } else {
      const T* wl = m.w[m.n_hidden] + ((size_t)w * KSW * 64 + lane) * 2;
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) load_frag<T, 2>(wl + ks * 128, dst[ks]);
      if (m.tail4) {          // second slot: the 4x4x4 tail fragment instead of tile 1
        const T* wt = m.wt + ((size_t)w * KSW * 64 + lane);
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) dst[ks][1] = wt[ks * 64];
      }
    }
  }

  // Hidden-layer biases of this lane's columns, resident: accumulators start from them, so the
  // epilogue has neither an LDS read nor an add between the last MFMA and the activation.
  // (16-row tiles only: the taller tiles are register-bound and keep reading the bias from LDS.)
  static constexpr bool RESIDENT_BIAS = LEAN < 1 && (MT == 1);
  T bias_r[kMaxHidden][NT];
  // Prefetch buffer: the first group of the next hidden layer (and, when they are not resident,
  // the output-layer fragments).  A member, not a local of run(): the first group of hidden layer 1
  // for the NEXT call is requested at the end of a call, together with layer 0's fragments, so it
  // has the caller's whole inter-call phase to arrive (measured +1 % f64, +3 % f32 on c3).
  T pfn[GH][NT];
  bool pfn_ready = false;

  // Once per kernel, before the first run(): resident biases / output weights + the first prefetch.
  __device__ __forceinline__ void init(const MlpDev<T>& m) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int l = 0; l < kMaxHidden; ++l)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bias_r[l][nt] = (RESIDENT_BIAS && l < m.n_hidden) ? m.b[l][16 * (NT * w + nt) + (lane & 15)] : T(0);
    if constexpr (RESIDENT_OUT) load_out(m, w, lane, wout);
    prefetch0(m);
  }

  // Request layer 0's weights.  Call before the barrier/phase that precedes run(); the loads
  // complete while other work proceeds.
  __device__ __forceinline__ void prefetch0(const MlpDev<T>& m) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T* wl = slice0(m, w, lane);
    if constexpr (!FULL0) {
      load_group<T, NT, 2>(wl, 0, first0<2>());
      return;
    }
    switch (m.k1p) {
      case 8: load_group<T, NT, 2>(wl, 0, first0<2>()); break;
      case 16: load_group<T, NT, 4>(wl, 0, first0<4>()); break;
      case 24: load_group<T, NT, 6>(wl, 0, first0<6>()); break;
      case 32: load_group<T, NT, 8>(wl, 0, first0<8>()); break;
      case 40: load_group<T, NT, 10>(wl, 0, first0<10>()); break;
      default: load_group<T, NT, 12>(wl, 0, first0<12>()); break;
    }
  }

  // On entry lds[L.xu] holds [x | u | 0] for the tile's M rows, init() has been called and every
  // thread has passed a barrier after the last write to lds[L.xu].  On exit
  // lds[L.part + (w*M + row)*nxp + col] holds wave w's partial of the output layer (bias NOT
  // added), pf0 has been re-requested for the next call, and a barrier has been passed.
  // If DERIV, act'(z) of hidden layer l is written to dz[l*dz_layer_stride + row*hpad + col].
  __device__ __forceinline__ void run(const MlpDev<T>& m, const TileLds& L, T* lds,
                                      T* __restrict__ dz = nullptr, size_t dz_layer_stride = 0) {
    run_side(m, L, lds, [] {}, dz, dz_layer_stride);
  }

  // As run(); `side()` is executed by every thread after the output layer's MFMAs have been
  // issued and before their results are read: work placed there (the caller's bookkeeping for the
  // next step -- anything that does not depend on this call's output and does not touch the
  // activation / partials buffers) runs while the matrix pipe drains instead of after it.
  template <typename Side>
  __device__ __forceinline__ void run_side(const MlpDev<T>& m, const TileLds& L, T* lds, Side&& side,
                                           T* __restrict__ dz = nullptr, size_t dz_layer_stride = 0) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    T* act = lds + L.act;            // buffer the next layer reads
    T* act_other = lds + L.act2;     // buffer the next epilogue may write (== act if single-buffered)
    const bool pingpong = L.act2 != L.act;
    const int as = L.act_stride;
    const int no = m.nxp / 16;
    static_assert(GH * NT == KSW * NOMAX, "prefetch buffer shapes must coincide");   // pfn holds either
    auto prefetch_next = [&](int l_next) {
      if (l_next < m.n_hidden) {
        load_group<T, NT, GH>(slice_h(m, l_next, w, lane), 0, pfn);
      } else if constexpr (!RESIDENT_OUT) {
        // not resident: the output fragments ([KSW][2]) reuse the hidden prefetch buffer
        // ([GH][NT], the same 8*NT values) -- one register range for whatever comes next
        T* flat = &pfn[0][0];
        if (m.nxp == 16) {
          const T* wl = m.w[m.n_hidden] + ((size_t)w * KSW * 64 + lane);
#pragma unroll
          for (int ks = 0; ks < KSW; ++ks) { flat[2 * ks] = wl[ks * 64]; flat[2 * ks + 1] = T(0); }
        } else {
          const T* wl = m.w[m.n_hidden] + ((size_t)w * KSW * 64 + lane) * 2;
#pragma unroll
          for (int ks = 0; ks < KSW; ++ks) {
            T two[2];
            load_frag<T, 2>(wl + ks * 128, two);
            flat[2 * ks] = two[0];
            flat[2 * ks + 1] = two[1];
          }
          if (m.tail4) {
            const T* wt = m.wt + ((size_t)w * KSW * 64 + lane);
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) flat[2 * ks + 1] = wt[ks * 64];
          }
        }
      }
    };
    auto wo = [&](int ks, int n) -> T {
      if constexpr (RESIDENT_OUT) return wout[ks][n];
      else return (&pfn[0][0])[2 * ks + n];
    };
    // bias + activation + store of one layer's accumulators (activation kind hoisted out of
    // the element loops: one uniform branch per layer instead of one per element)
    auto epilogue_k = [&](int l, acc_t (&acc)[MT][NT], T* dst, auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
      const T* bias = lds + L.bias + l * m.hpad;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = 16 * (NT * w + nt) + i;
          const T bc = RESIDENT_BIAS ? T(0) : bias[col];   // resident: the accumulator started from it
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + acc_row<T>(q, r);
            const T z = RESIDENT_BIAS ? acc[mt][nt][r] : acc[mt][nt][r] + bc;
            dst[row * as + col] = act_apply<T>(KIND, z);
            if (DERIV) dz[(size_t)l * dz_layer_stride + (size_t)row * m.hpad + col] = act_deriv<T>(KIND, z);
          }
        }
    };
    auto epilogue = [&](int l, acc_t (&acc)[MT][NT], T* dst) {
      switch (m.act) {
        case 0: epilogue_k(l, acc, dst, std::integral_constant<int, 0>{}); break;
        case 1: epilogue_k(l, acc, dst, std::integral_constant<int, 1>{}); break;
        case 2: epilogue_k(l, acc, dst, std::integral_constant<int, 2>{}); break;
        case 4: epilogue_k(l, acc, dst, std::integral_constant<int, 4>{}); break;
        default: epilogue_k(l, acc, dst, std::integral_constant<int, 3>{}); break;
      }
    };

    // ---- layer 0: K = k1p (8, 16, .. 48), A = [x | u] ----------------------------------------
    {
      acc_t acc[MT][NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{bias_r[0][nt], bias_r[0][nt], bias_r[0][nt], bias_r[0][nt]};
      const T* wl = slice0(m, w, lane);
      const T* A = lds + L.xu;
      // first group of hidden layer 1: in flight under layer 0's MFMAs (64-row tiles have no
      // registers to spare for that and fetch it after the MFMAs instead)
      if (!pfn_ready) prefetch_next(1);   // first call only; later calls were served at the previous call's end
      if constexpr (FULL0) {
        switch (m.k1p) {   // one fully unrolled variant per padded input width
          case 8: layer_mma_static<T, NT, MT, 2, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
          case 16: layer_mma_static<T, NT, MT, 4, 4>(A, L.xu_stride, wl, lane, first0<4>(), acc); break;
          case 24: layer_mma_static<T, NT, MT, 6, 6>(A, L.xu_stride, wl, lane, first0<6>(), acc); break;
          case 32: layer_mma_static<T, NT, MT, 8, 8>(A, L.xu_stride, wl, lane, first0<8>(), acc); break;
          case 40: layer_mma_static<T, NT, MT, 10, 10>(A, L.xu_stride, wl, lane, first0<10>(), acc); break;
          default: layer_mma_static<T, NT, MT, 12, 12>(A, L.xu_stride, wl, lane, first0<12>(), acc); break;
        }
      } else {
        switch (m.k1p) {
          case 8: layer_mma_static<T, NT, MT, 2, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
          case 16: layer_mma_static<T, NT, MT, 4, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
          case 24: layer_mma_static<T, NT, MT, 6, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
          case 32: layer_mma_static<T, NT, MT, 8, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
          case 40: layer_mma_static<T, NT, MT, 10, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
          default: layer_mma_static<T, NT, MT, 12, 2>(A, L.xu_stride, wl, lane, first0<2>(), acc); break;
        }
      }
      AMPC_MARK(2);
      epilogue(0, acc, act);
    }
    // Barrier placement.  A layer's epilogue leaves wave w's columns in LDS.  The output layer is
    // K-split so that wave w consumes exactly those columns: no barrier before it.  A hidden
    // layer needs every wave's columns, but (OWN) starts with its own group and takes the
    // barrier after it, inside layer_mma_static, so barrier skew is covered by MFMA work.
    if (!OWN && m.n_hidden > 1) lds_barrier();
    AMPC_MARK(3);

    // ---- hidden -> hidden layers ---------------------------------------------------------------
    for (int l = 1; l < m.n_hidden; ++l) {
      acc_t acc[MT][NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          T b = bias_r[0][nt];                  // bias_r[l][nt] with l a loop variable: selected
#pragma unroll                                  // by compares so the array stays in registers
          for (int k = 1; k < kMaxHidden; ++k) b = (l == k) ? bias_r[k][nt] : b;
          acc[mt][nt] = acc_t{b, b, b, b};
        }
      // f64 streams the weights in half-groups (32-64 VGPRs less: the 64-row tile stops spilling,
      // +2 %, and the 16-row tile has room for the early prefetch); f32 keeps whole groups
      // (half-groups measured -4 % there)
      layer_mma_static<T, NT, MT, KSH, GH, (W == 8), OWN, (sizeof(T) == 8 ? GH / 2 : GH)>(
          act, as, slice_h(m, l, w, lane), lane, pfn, acc, w);
      AMPC_MARK(4);
      prefetch_next(l + 1);
      // single buffer: every wave must finish reading act before it is overwritten;
      // ping-pong: the epilogue writes the other buffer, no barrier needed here
      if (!pingpong) lds_barrier();
      AMPC_MARK(5);
      epilogue(l, acc, act_other);
      if (!OWN && l + 1 < m.n_hidden) lds_barrier();
      AMPC_MARK(6);
      { T* tmp = act; act = act_other; act_other = tmp; }
    }

    // ---- output layer: K-split, wave w owns k-steps [w*KSW, (w+1)*KSW) -----------------------
    acc_t oacc[MT][NOMAX];
    T tacc[MT];                      // tail4: columns 16..19, one value per lane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      tacc[mt] = T(0);
#pragma unroll
      for (int n = 0; n < NOMAX; ++n) oacc[mt][n] = acc_t{0, 0, 0, 0};
    }
    {
      const T* arow = act + i * as + q + 4 * w * KSW;
      if (no == 1) {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            oacc[mt][0] = mfma16(arow[mt * 16 * as + 4 * ks], wo(ks, 0), oacc[mt][0]);
      } else if (m.tail4) {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const T a = arow[mt * 16 * as + 4 * ks];
            oacc[mt][0] = mfma16(a, wo(ks, 0), oacc[mt][0]);
            tacc[mt] = mfma4(a, wo(ks, 1), tacc[mt]);
          }
      } else {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const T a = arow[mt * 16 * as + 4 * ks];
            oacc[mt][0] = mfma16(a, wo(ks, 0), oacc[mt][0]);
            oacc[mt][1] = mfma16(a, wo(ks, 1), oacc[mt][1]);
          }
      }
    }
    AMPC_MARK(7);
    prefetch0(m);     // next call's first group: overlaps the reduction and the caller's work
    if (m.n_hidden > 1) { prefetch_next(1); pfn_ready = true; }
    side();
    if (L.part_alias) lds_barrier();  // partials reuse `act`: every wave must be done reading it
    AMPC_MARK(8);
    T* part = lds + L.part + w * M * m.nxp;
    const int nfull = m.tail4 ? 1 : no;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int n = 0; n < NOMAX; ++n)
        if (n < nfull) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + acc_row<T>(q, r);
            part[row * m.nxp + 16 * n + i] = oacc[mt][n][r];
          }
        }
      if (m.tail4) part[(16 * mt + tail_row(lane)) * m.nxp + 16 + tail_col(lane)] = tacc[mt];
    }
    AMPC_MARK(13);
    lds_barrier();
    AMPC_MARK(9);
  }

  // y[row][col] (folded output: already the state increment) from the partials left by run().
  __device__ __forceinline__ static T output(const MlpDev<T>& m, const TileLds& L, const T* lds,
                                             int row, int col) {
    const T* p = lds + L.part + row * m.nxp + col;
    T y = lds[L.bias + m.n_hidden * m.hpad + col];
#pragma unroll
    for (int w = 0; w < W; ++w) y += p[w * M * m.nxp];
    return y;
  }
};

// x' M x for the rows this thread owns (rows r, r+TPS, ...); d[j] = v[j] - g[j].
template <typename T>
__device__ __forceinline__ T quad_rows(const T* __restrict__ Mx, const T* __restrict__ v,
                                       const T* __restrict__ g, int n, int r, int tps, bool diag) {
  T acc = T(0);
  if (diag) {
    for (int i = r; i < n; i += tps) {
      const T d = v[i] - (g ? g[i] : T(0));
      acc += Mx[i * n + i] * d * d;
    }
  } else {
    for (int i = r; i < n; i += tps) {
      T s = T(0);
      for (int j = 0; j < n; ++j) s += Mx[i * n + j] * (v[j] - (g ? g[j] : T(0)));
      acc += (v[i] - (g ? g[i] : T(0))) * s;
    }
  }
  return acc;
}

}  // namespace ampc
