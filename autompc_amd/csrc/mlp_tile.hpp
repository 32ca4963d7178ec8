// mlp_tile.hpp -- fused MLP dynamics step on one workgroup-resident tile of samples (gfx950).
//
// One workgroup = 256 threads = 4 wave64, one wave per SIMD.  A tile is M = 16*MT samples.
// Hidden width is padded to HPAD = 64*NT so each wave owns NT 16-column MFMA tiles of every
// hidden layer (N-split); the small output layer is K-split across the four waves and reduced
// through LDS.  Arithmetic is exact f64 (v_mfma_f64_16x16x4_f64) or exact f32
// (v_mfma_f32_16x16x4_f32) -- both are k-ordered fma chains, no reduced-precision inputs.
//
// Operand sources:
//   A (activations)  LDS, row-major [M][K+2]: the +2 pad makes the 16-row x 2-column access of
//                    each 32-lane half conflict-free for ds_read_b64 (f64) / ds_read_b32 (f32).
//   B (weights)      global memory, pre-packed on the host in exact fragment order so every wave
//                    reads one contiguous, fully coalesced run per k-step; weights are re-read
//                    every time step but stay L2-resident (<= 640 KB per model).
//
// What the math is (reference: autompc/sysid/mlp.py:20-30, :55-59, :229-236):
//   xin = ([x,u] - xu_mean) / xu_std ; h = act(W h + b) per hidden layer ; y = W_out h + b_out
//   x' = x + (y * dy_std + dy_mean)
#pragma once
#include <hip/hip_runtime.h>

namespace ampc {

constexpr int kWG = 256;      // threads per workgroup
constexpr int kWaves = 4;     // waves per workgroup
constexpr int kMaxHidden = 4; // hidden layers supported (reference config space: 1..4)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Acc;
template <> struct Acc<double> { using type = d4; };
template <> struct Acc<float> { using type = f4; };

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Row of accumulator register r held by lane-quad q (= lane >> 4); column is lane & 15.
//   f64 16x16x4: row = q + 4 r      f32 16x16x4: row = 4 q + r
template <typename T> __device__ __forceinline__ int acc_row(int q, int r);
template <> __device__ __forceinline__ int acc_row<double>(int q, int r) { return q + 4 * r; }
template <> __device__ __forceinline__ int acc_row<float>(int q, int r) { return 4 * q + r; }

// ---- activations (torch semantics: ReLU, Tanh, Sigmoid, SELU; mlp.py:44-51) -------------
template <typename T> __device__ __forceinline__ T act_apply(int kind, T z) {
  switch (kind) {
    case 0: return z > T(0) ? z : T(0);
    case 1: return tanh(z);
    case 2: return T(1) / (T(1) + exp(-z));
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
  int k1p;              // kin zero-padded to 16, 32 or 48 (first-layer MFMA K)
  int n_hidden;         // hidden layers
  int hpad;             // 64*NT
  int nxp;              // nx rounded up to a multiple of 16
  int act;              // activation kind
  const T* w[kMaxHidden + 1];   // packed fragments, layer 0..n_hidden (last = output layer)
  const T* b[kMaxHidden + 1];   // padded biases
  const T* wj[kMaxHidden + 1];  // packed fragments for the Jacobian chain (transposed use)
  const T* norm;                // xu_mean[kin] xu_std[kin] dy_mean[nx] dy_std[nx]
};

// LDS carve-up shared by all kernels that run the tile (offsets in elements of T).
struct TileLds {
  int act;      // [M][hpad+2]  (also reused for the output-layer partials [4][M][nxp])
  int xin;      // [M][k1p+2]
  int xs;       // [M][nx]   current state
  int us;       // [M][nu]   current scaled control
  int norm;     // xu_mean, xu_std, dy_mean, dy_std
  int bias;     // n_hidden*hpad + nxp
  int extra;    // kernel-specific region starts here
  int act_stride, xin_stride;
};

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ inline int round_up(int a, int m) { return (a + m - 1) / m * m; }

template <typename T>
__host__ inline TileLds make_tile_lds(const MlpDev<T>& m, int M) {
  TileLds L;
  int o = 0;
  L.act_stride = m.hpad + 2;
  L.xin_stride = m.k1p + 2;
  L.act = o; o += M * imax(L.act_stride, kWaves * m.nxp);
  L.xin = o; o += M * L.xin_stride;
  L.xs = o; o += M * m.nx;
  L.us = o; o += M * m.nu;
  L.norm = o; o += 2 * m.kin + 2 * m.nx;
  L.bias = o; o += m.n_hidden * m.hpad + m.nxp;
  L.extra = round_up(o, 4);
  return L;
}

// Stage normalisers and biases into LDS (call once per kernel, then __syncthreads()).
template <typename T>
__device__ __forceinline__ void tile_load_constants(const MlpDev<T>& m, const TileLds& L, T* lds) {
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * m.kin + 2 * m.nx; i += kWG) lds[L.norm + i] = m.norm[i];
  for (int l = 0; l < m.n_hidden; ++l)
    for (int i = tid; i < m.hpad; i += kWG) lds[L.bias + l * m.hpad + i] = m.b[l][i];
  for (int i = tid; i < m.nxp; i += kWG) lds[L.bias + m.n_hidden * m.hpad + i] = m.b[m.n_hidden][i];
}

// ---- weight fragment loads -------------------------------------------------------------------
// A lane's NT consecutive fragment values for one k-step, as the widest aligned vector load.
template <typename T, int N> using vec_t = T __attribute__((ext_vector_type(N)));

template <typename T, int NT>
__device__ __forceinline__ void load_frag(const T* __restrict__ p, T (&b)[NT]) {
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

// One N-split layer with compile-time k extent KS (hidden layers, K = HPAD): fully unrolled,
// weight fragments double-buffered in groups of G k-steps so the L2 fetch of group g+1 is in
// flight while group g's MFMAs issue.
template <typename T, int NT, int MT, int KS, int G>
__device__ __forceinline__ void layer_mma_static(const T* __restrict__ A, int a_stride,
                                                 const T* __restrict__ wp, int lane,
                                                 typename Acc<T>::type (&acc)[MT][NT]) {
  static_assert(KS % G == 0, "group size must divide the k extent");
  constexpr int NG = KS / G;
  const int i = lane & 15, q = lane >> 4;
  const T* arow = A + i * a_stride + q;
  const T* wl = wp + lane * NT;
  T b[2][G][NT];
#pragma unroll
  for (int kk = 0; kk < G; ++kk) load_frag<T, NT>(wl + kk * 64 * NT, b[0][kk]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int kk = 0; kk < G; ++kk)
        load_frag<T, NT>(wl + ((g + 1) * G + kk) * 64 * NT, b[(g + 1) & 1][kk]);
    }
#pragma unroll
    for (int kk = 0; kk < G; ++kk) {
      const int ks = g * G + kk;
      T a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt * 16 * a_stride + 4 * ks];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(a[mt], b[g & 1][kk][nt], acc[mt][nt]);
    }
  }
}

// Full network on the tile.  On entry lds[L.xin] holds the normalised inputs [M][k1p] (columns
// kin..k1p-1 zero) and every thread has passed a barrier after writing them.  On exit
// lds[L.act + (w*M + row)*nxp + col] holds wave w's partial of the output layer (bias NOT added)
// and a barrier has been passed, i.e. y[row][col] = bias + sum_w partial.
// If DERIV, act'(z) of hidden layer l is also written to dz[l][row*hpad + col] (global scratch).
template <typename T, int NT, int MT, bool DERIV = false>
__device__ __forceinline__ void tile_network(const MlpDev<T>& m, const TileLds& L, T* lds,
                                             T* __restrict__ dz = nullptr, int dz_layer_stride = 0) {
  using acc_t = typename Acc<T>::type;
  constexpr int M = 16 * MT;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  T* act = lds + L.act;
  const int as = L.act_stride;
  const int ks_h = m.hpad / 4;

  for (int l = 0; l < m.n_hidden; ++l) {
    acc_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{0, 0, 0, 0};
    if (l == 0) {
      // first layer: K = k1p is 16, 32 or 48 (kin zero-padded) -> three fully unrolled variants
      const T* w0 = m.w[0] + (size_t)w * (m.k1p / 4) * 64 * NT;
      if (m.k1p == 16) layer_mma_static<T, NT, MT, 4, 4>(lds + L.xin, L.xin_stride, w0, lane, acc);
      else if (m.k1p == 32) layer_mma_static<T, NT, MT, 8, 8>(lds + L.xin, L.xin_stride, w0, lane, acc);
      else layer_mma_static<T, NT, MT, 12, 4>(lds + L.xin, L.xin_stride, w0, lane, acc);
    } else {
      layer_mma_static<T, NT, MT, 16 * NT, 8>(act, as, m.w[l] + (size_t)w * ks_h * 64 * NT, lane, acc);
      __syncthreads();  // every wave finished reading act before it is overwritten
    }
    const T* bias = lds + L.bias + l * m.hpad;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * (NT * w + nt) + i;
        const T bc = bias[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + acc_row<T>(q, r);
          const T z = acc[mt][nt][r] + bc;
          act[row * as + col] = act_apply<T>(m.act, z);
          if (DERIV) dz[(size_t)l * dz_layer_stride + row * m.hpad + col] = act_deriv<T>(m.act, z);
        }
      }
    __syncthreads();
  }

  // output layer: K-split, wave w owns k-steps [w*KSW, (w+1)*KSW)
  constexpr int NOMAX = 2;  // nx <= 32
  constexpr int KSW = 4 * NT;
  const int no = m.nxp / 16;
  acc_t oacc[MT][NOMAX];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int n = 0; n < NOMAX; ++n) oacc[mt][n] = acc_t{0, 0, 0, 0};
  {
    const T* arow = act + i * as + q + 4 * w * KSW;
    if (no == 1) {
      const T* wl = m.w[m.n_hidden] + ((size_t)w * KSW * 64 + lane);
      T b[KSW];
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) b[ks] = wl[ks * 64];
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          oacc[mt][0] = mfma16(arow[mt * 16 * as + 4 * ks], b[ks], oacc[mt][0]);
    } else {
      const T* wl = m.w[m.n_hidden] + ((size_t)w * KSW * 64 + lane) * 2;
      T b[KSW][2];
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) load_frag<T, 2>(wl + ks * 128, b[ks]);
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const T a = arow[mt * 16 * as + 4 * ks];
          oacc[mt][0] = mfma16(a, b[ks][0], oacc[mt][0]);
          oacc[mt][1] = mfma16(a, b[ks][1], oacc[mt][1]);
        }
    }
  }
  __syncthreads();  // act fully consumed; reuse it for the partials
  T* part = act + w * M * m.nxp;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int n = 0; n < NOMAX; ++n)
      if (n < no) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + acc_row<T>(q, r);
          part[row * m.nxp + 16 * n + i] = oacc[mt][n][r];
        }
      }
  __syncthreads();
}

// y[row][col] from the partials left by tile_network.
template <typename T, int MT>
__device__ __forceinline__ T tile_output(const MlpDev<T>& m, const TileLds& L, const T* lds, int row,
                                         int col) {
  constexpr int M = 16 * MT;
  const T* p = lds + L.act + row * m.nxp + col;
  T y = lds[L.bias + m.n_hidden * m.hpad + col];
#pragma unroll
  for (int w = 0; w < kWaves; ++w) y += p[w * M * m.nxp];
  return y;
}

}  // namespace ampc
