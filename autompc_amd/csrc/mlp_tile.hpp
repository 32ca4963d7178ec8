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

#include "probe.hpp"

namespace ampc {

constexpr int kMaxHidden = 4;  // hidden layers supported (reference config space: 1..4)
constexpr int kMaxWaves = 8;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Acc;
template <> struct Acc<double> { using type = d4; };
template <> struct Acc<float> { using type = f4; };

// (Probe::* are timing-experiment switches, all false in the product build: probe.hpp)
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
  if constexpr (Probe::no_mfma) { c[0] += a * b; return c; }
  else return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  if constexpr (Probe::no_mfma) { c[0] += a * b; return c; }
  else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// timing experiments: the MFMAs of one layer class replaced by a single FMA (dependencies kept)
template <bool REAL, typename T, typename A>
__device__ __forceinline__ A mfma16_x(T a, T b, A c) {
  if constexpr (REAL) return mfma16(a, b, c);
  else { c[0] += a * b; return c; }
}
constexpr bool kRealHid = !Probe::no_hid, kRealL0 = !Probe::no_l0, kRealOut = !Probe::no_out;

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products (blocks); 16 cycles against the 64 of
// the 16x16x4.  Lane layout (probed on gfx950, tools/mfma44_probe.cpp), block = (lane/4)%4:
//   A[i][k]: i = lane%4, k = lane/16      B[k][j]: k = lane/16, j = lane%4
//   D[i][j]: i = lane/16, j = lane%4
// With rows 4*block + i the A operand is act[row = lane%16][k = lane/16] -- exactly the fragment
// the 16x16x4 MFMA takes -- so the same register feeds both; B is shared by the four blocks and
// D gives 16 rows x 4 columns, one value per lane.
__device__ __forceinline__ double mfma4(double a, double b, double c) {
  if constexpr (Probe::no_mfma) return c + a * b;
  else return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float mfma4(float, float, float c) { return c; }   // f64 only (tail4 is never set for f32)
// row / column (relative to 16) of the value mfma4 leaves in this lane
__device__ __forceinline__ int tail_row(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int tail_col(int lane) { return lane & 3; }

// Row of accumulator register r held by lane-quad q (= lane >> 4); column is lane & 15.
//   f64 16x16x4: row = q + 4 r      f32 16x16x4: row = 4 q + r
// Physical LDS row of logical tile row `row` in the ACTIVATION buffers.  f32: bits 2 and 3 of the
// row index are swapped.  An f32 accumulator lane (i, q) holds rows 4q .. 4q+3 of column i, so one
// ds_write_b32 of a wave stores rows {r, 4+r} (lanes 0-31) and {8+r, 12+r} (lanes 32-63) x 16
// columns; with the row stride = 2 (mod 32) dwords that the A-fragment reads need (16 rows x 2
// consecutive k per 32-lane group -> banks 2i + q), rows r and 4+r start 8 banks apart and half of
// every store group lands on a busy bank (SQ_LDS_BANK_CONFLICT 17.5 % of LDS cycles in round 2).
// No plain stride serves both patterns (reads need stride = 2 mod 4, stores 4 * stride = 16 mod 32);
// with the swap the store rows are 16 banks apart (physical r and 8+r) and the 16 rows of a read
// still cover the 16 even banks.  f64 (ds_write_b64 serves one accumulator row per 16-lane group)
// has no such conflict and keeps the identity.
template <typename T> __device__ __forceinline__ constexpr int act_row(int row) {
  return sizeof(T) == 4 ? ((row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1)) : row;
}

template <typename T> __device__ __forceinline__ int acc_row(int q, int r);
template <> __device__ __forceinline__ int acc_row<double>(int q, int r) { return q + 4 * r; }
template <> __device__ __forceinline__ int acc_row<float>(int q, int r) { return 4 * q + r; }

// 1 / d to within an ulp or two: hardware estimate + Newton steps (the full IEEE division sequence
// is twice as long).
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  return fma(fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}

// expm1(y) in f64 for |y| <= 700, ~25 instructions: y = n ln2 + r (Cody-Waite), E = expm1(r) from a
// degree-14 polynomial on |r| <= 0.347, expm1(y) = 2^n E + (2^n - 1).  The three smooth activations
// are built on it -- a few ulp each (checked against long-double references over millions of
// points, incl. subnormal and saturating arguments) at a quarter of the library routines'
// instruction count, and with few enough temporaries that the run-time activation switch no longer
// decides a kernel's register budget.
__device__ __forceinline__ double expm1_core(double y) {
  const double n = __builtin_rint(y * 1.4426950408889634074);
  double r = fma(-n, 6.93147180369123816490e-01, y);               // ln2 = hi + lo
  r = fma(-n, 1.90821492927058770002e-10, r);
  double q = 1.1470745597729725e-11;                               // 1/14!
  q = fma(q, r, 1.6059043836821613e-10);                           // 1/13!
  q = fma(q, r, 2.0876756987868099e-09);                           // 1/12!
  q = fma(q, r, 2.5052108385441719e-08);                           // 1/11!
  q = fma(q, r, 2.7557319223985891e-07);                           // 1/10!
  q = fma(q, r, 2.7557319223985893e-06);                           // 1/9!
  q = fma(q, r, 2.4801587301587302e-05);                           // 1/8!
  q = fma(q, r, 1.9841269841269841e-04);                           // 1/7!
  q = fma(q, r, 1.3888888888888889e-03);                           // 1/6!
  q = fma(q, r, 8.3333333333333332e-03);                           // 1/5!
  q = fma(q, r, 4.1666666666666664e-02);                           // 1/4!
  q = fma(q, r, 1.6666666666666666e-01);                           // 1/3!
  q = fma(q, r, 0.5);                                              // 1/2!
  const double E = fma(r * r, q, r);
  const double s = ldexp(1.0, (int)n);
  return fma(s, E, s - 1.0);
}
// num / den with one correction step (den > 0, no overflow at the call sites)
__device__ __forceinline__ double fast_div(double num, double den) {
  const double rc = fast_rcp(den);
  const double t = num * rc;
  return fma(fma(-den, t, num), rc, t);
}
// tanh|x| = M / (M + 2), M = expm1(2|x|): one formula for all magnitudes, no cancellation (M >= 0)
__device__ __forceinline__ double fast_tanh(double x) {
  const double a = fmin(fabs(x), 20.0);                            // tanh(20) rounds to 1
  const double M = expm1_core(a + a);
  const double t = fast_div(M, M + 2.0);
  return x != x ? x : copysign(t, x);
}
// sigmoid(x) = 1 / (1 + e^-x) = 1 / (2 + expm1(-x))
__device__ __forceinline__ double fast_sigmoid(double x) {
  const double M = expm1_core(-fmax(fmin(x, 700.0), -700.0));
  const double t = fast_div(1.0, M + 2.0);
  return x != x ? x : t;
}
// expm1 for the negative branch of SELU (saturates at -1 below -40)
__device__ __forceinline__ double fast_expm1_neg(double x) { return expm1_core(fmax(x, -40.0)); }
__device__ __forceinline__ double tanh_t(double x) { return fast_tanh(x); }
__device__ __forceinline__ float tanh_t(float x) { return tanhf(x); }
__device__ __forceinline__ double sigmoid_t(double x) { return fast_sigmoid(x); }
__device__ __forceinline__ float sigmoid_t(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ double expm1n_t(double x) { return fast_expm1_neg(x); }
__device__ __forceinline__ float expm1n_t(float x) { return expm1f(x); }

// ---- activations (torch semantics: ReLU, Tanh, Sigmoid, SELU; mlp.py:44-51) -------------
// kind 4 = identity: a linear model x' = A x + B u (ARX arx.py:151-154, Koopman
// koopman.py:170-173) staged as a one-hidden-layer linear network.
template <typename T> __device__ __forceinline__ T act_apply(int kind, T z) {
  switch (kind) {
    case 0: return z > T(0) ? z : T(0);
    case 1: return tanh_t(z);
    case 2: return sigmoid_t(z);
    case 4: return z;
    default: {
      const T alpha = T(1.6732632423543772848170429916717);
      const T scale = T(1.0507009873554804934193349852946);
      return scale * (z > T(0) ? z : alpha * expm1n_t(z));
    }
  }
}
// derivative expressed from z (pre-activation)
template <typename T> __device__ __forceinline__ T act_deriv(int kind, T z) {
  switch (kind) {
    case 0: return z > T(0) ? T(1) : T(0);
    case 1: { T t = tanh_t(z); return T(1) - t * t; }
    case 2: { T s = sigmoid_t(z); return s * (T(1) - s); }
    case 4: return T(1);
    default: {
      const T alpha = T(1.6732632423543772848170429916717);
      const T scale = T(1.0507009873554804934193349852946);
      return z > T(0) ? scale : scale * alpha * (expm1n_t(z) + T(1));
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
  const T* wbase;               // start of the packed model buffer (every array below lies in it)
  const T* w[kMaxHidden + 1];   // packed fragments, layer 0..n_hidden (last = output layer)
  const T* b[kMaxHidden + 1];   // padded biases (normalisers folded in)
  const T* wj[kMaxHidden + 1];  // packed fragments for the Jacobian chain (transposed use)
  // f64, 16 < nx <= 20: output columns 16..19 come from v_mfma_f64_4x4x4_4b (see mfma4 below)
  // instead of a second, mostly empty 16-column tile.  wt [W][KSW][64]: lane l of k-step ks holds
  // W_out'[k = 4 ks + l/16][col = 16 + l%4].
  const T* wt;
  int tail4;
  // the forward weights once more, packed for a FOUR-wave workgroup whatever (W, NT) the tile uses
  // (N-split hidden layers with hpad/64 column tiles per wave, no rotation, chunked so that every
  // fragment load is coalesced -- see build_model; K-split output layer): the four-row line-search
  // kernel (ilqr_ls4.hpp).
  const T* w4[kMaxHidden + 1];
  // Byte offset of the model a kernel runs on from the one staged here (0 on the host side; set per
  // problem / slot by the kernels of a plan that holds several models of one shape, see model_delta_of
  // below).  The kernels take every pointer through the accessors, which add it: the pointer ARRAYS are
  // never rewritten -- a descriptor whose arrays are modified and then indexed with a run-time layer
  // number is kept in scratch memory, and every weight pointer read back from there is a per-lane value.
  long long delta;
  __device__ __forceinline__ const T* sh(const T* p) const {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + delta);
  }
  // (entry l of a pointer array by a chain of selects over CONSTANT indices: with no run-time index into
  //  the descriptor left, it is split into scalar registers whether or not a kernel has modified it)
  __device__ __forceinline__ const T* pick(const T* const (&a)[kMaxHidden + 1], int l) const {
    static_assert(kMaxHidden == 4, "one select per entry, written out (a loop would be a run-time index until unrolled)");
    const T* p = a[0];
    p = l == 1 ? a[1] : p;
    p = l == 2 ? a[2] : p;
    p = l == 3 ? a[3] : p;
    p = l == 4 ? a[4] : p;
    return sh(p);
  }
  __device__ __forceinline__ const T* WB() const { return sh(wbase); }
  __device__ __forceinline__ const T* W(int l) const { return pick(w, l); }
  __device__ __forceinline__ const T* B(int l) const { return pick(b, l); }
  __device__ __forceinline__ const T* WJ(int l) const { return pick(wj, l); }
  __device__ __forceinline__ const T* W4(int l) const { return pick(w4, l); }
  __device__ __forceinline__ const T* WT() const { return sh(wt); }
};

// Several controller models of ONE shape in a plan (tuning candidates that carry their own model,
// pipeline.py:138-145).  Models of one shape are packed identically (build_model: every array at the same
// offset of the model's buffer), so model k is the plan handle's descriptor with every pointer moved by
// ONE byte offset, delta[k] = buffer of model k - buffer of the plan's model: the plan holds the table of
// deltas, a problem / slot names its entry.  One wave-uniform 64-bit value per workgroup; the descriptor
// itself stays the kernel argument (scalar registers, fields fetched where they are used).
__device__ __forceinline__ long long model_delta_of(const long long* __restrict__ tab, int idx) {
  if (tab == nullptr) return 0;
  const long long d = tab[idx];
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)d);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)d >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ MlpDev<T> shift_model(MlpDev<T> m, long long d) {
  m.delta = d;
  return m;
}
// The descriptor a kernel works with: folded for the shape and -- SHAPE-SPECIALISED instantiations only --
// moved to the problem's / slot's model.  The run-time-shape instantiations read the kernel argument in
// place (their layer loops have run-time trip counts: a modified copy of the descriptor ends up in scratch
// memory there, every pointer a per-lane value, the kernel 40 % longer), so a plan that holds several
// models needs the shape-specialised kernels -- a registered shape or its run-time compiled plugin; the
// host waits for the plugin (api.cpp: require_static_for_models).
template <typename SH, typename T, typename F>
__device__ __forceinline__ MlpDev<T> plan_model(const MlpDev<T>& in, F delta) {
  if constexpr (SH::kStatic) return shift_model(SH::template fold<T>(in), delta());
  else return SH::template fold<T>(in);
}

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

// Layout of one cost block (ampc_set_affine_quad_costs; one block per controller / tuning candidate):
//   Q[no*no] R[nu*nu] F[no*no] goal[no] lin[no] lint[no] c0 c1
// stage cost   (x-goal)'Q(x-goal) + lin'(x-goal) + c0 + u'Ru      (dt-scaled in iLQR, not in MPPI)
// terminal     (x-goal)'F(x-goal) + lint'(x-goal) + c1
// A sum of quadratic terms with DIFFERENT goals (SumCost._sum_results, sum_cost.py:49-54, e.g.
// QuadCostFactory + GaussRegFactory, gauss_reg_factory.py:37-45) is such a form about the first
// term's goal; for a single QuadCost / a same-goal sum lin = lint = c0 = c1 = 0.
__host__ __device__ constexpr int cost_block_stride(int no, int nu) {
  return (2 * no * no + nu * nu + 3 * no + 2 + 3) / 4 * 4;
}
__host__ __device__ constexpr int cost_off_goal(int no, int nu) { return 2 * no * no + nu * nu; }
__host__ __device__ constexpr int cost_off_lin(int no, int nu) { return 2 * no * no + nu * nu + no; }
__host__ __device__ constexpr int cost_off_lint(int no, int nu) { return 2 * no * no + nu * nu + 2 * no; }
__host__ __device__ constexpr int cost_off_c(int no, int nu) { return 2 * no * no + nu * nu + 3 * no; }

// (constexpr on plain integers: the shape-specialised kernels evaluate it at compile time, the host
// at plan build -- both must agree, see StaticShape below)
// Row stride of the output-layer partials [W][M][.]: nxp in f64; nxp + 4 in f32, where a wave's
// ds_write_b32 stores accumulator rows r and 4 + r of 16 columns per 32-lane group -- with a stride
// of 32 dwords both rows hit the same 16 banks (the last third of round 2's f32 store conflicts),
// with 36 they are 16 banks apart.
__host__ __device__ constexpr int part_stride(int esz, int nxp) { return esz == 4 ? nxp + 4 : nxp; }

__host__ __device__ constexpr TileLds tile_lds_dims(int esz, int hpad, int k1p, int nxp, int n_hidden,
                                                    int M, int W, bool separate_partials,
                                                    bool double_act) {
  TileLds L{};
  int o = 0;
  // Row padding for conflict-free A-fragment reads (16 rows x consecutive k per access):
  //   f32  ds_read(2)_b32 banks = dword mod 32 over 32-lane halves  -> stride = 2 (mod 4)
  //   f64  hipcc pairs k-steps into ds_read2_b64 (16-lane groups, banks = dword mod 32)
  //        -> odd stride in 8-byte units (measured: SQ_LDS_BANK_CONFLICT 40% -> ~0 of LDS cycles)
  const int pad = esz == 8 ? 1 : 2;
  L.act_stride = hpad + pad;
  L.xu_stride = k1p + pad;
  L.part_alias = separate_partials ? 0 : 1;
  if (separate_partials) {
    L.act = o; o += M * L.act_stride;
    L.act2 = L.act;
    if (double_act && n_hidden > 1) { L.act2 = o; o += M * L.act_stride; }
    L.part = o; o += W * M * part_stride(esz, nxp);
  } else {
    const int a = L.act_stride, b = W * part_stride(esz, nxp);
    L.act = o; L.act2 = o; L.part = o; o += M * (a > b ? a : b);
  }
  L.xu = o; o += M * L.xu_stride;
  L.bias = o; o += n_hidden * hpad + nxp;
  L.extra = (o + 3) / 4 * 4;
  return L;
}

template <typename T>
__host__ inline TileLds make_tile_lds(const MlpDev<T>& m, int M, int W, bool separate_partials = true,
                                      bool double_act = true) {
  return tile_lds_dims((int)sizeof(T), m.hpad, m.k1p, m.nxp, m.n_hidden, M, W, separate_partials, double_act);
}

// ---- compile-time shapes ----------------------------------------------------------------------
// Every dimension the kernels read from the model descriptor and the LDS map at run time is a
// wave-uniform value that must stay live across the whole time loop.  The rollout kernel needs
// ~190 of them against the 102 SGPRs a wave has; hipcc parks the excess in VGPR lanes and every
// later use is a v_readlane -- a VALU instruction, ~8 cycles of matrix-pipe time each on gfx950
// (VALU and MFMA do not overlap).  A kernel instantiated with a StaticShape folds those values to
// immediates instead.  DynShape keeps the fully general run-time path; the host picks the static
// instantiation when the staged model matches a registered shape (shapes.hpp) AND its LDS map is
// the one the shape implies (richest map: ping-pong activations + separate partials).
struct DynShape {
  static constexpr bool kStatic = false;
  static constexpr int nx = 0, nu = 0, no = 0, k1p = 0, nxp = 0, n_hidden = 0, hpad = 0;
  template <typename T> __device__ __forceinline__ static MlpDev<T> fold(const MlpDev<T>& m) { return m; }
  template <typename T, int M, int W>
  __device__ __forceinline__ static TileLds fold_lds(const TileLds& L) { return L; }
};

// ACT >= 0 also fixes the activation (the run-time switch keeps all five epilogues in the loop
// body: ~4x the code of the relu-only loop); ACT = -1 leaves it a run-time value.
// LV selects the LDS map the tile uses (tile_lds_for picks the richest that fits the 160 KB):
//   0 ping-pong activations + separate partials, 1 one activation buffer + separate partials,
//   2 one buffer shared by activations and partials.
template <int NX, int NU, int NO, int NH, int HPAD, int ACT = -1, int LV = 0> struct StaticShape {
  static constexpr bool kStatic = true;
  static constexpr int nx = NX, nu = NU, no = NO, n_hidden = NH, hpad = HPAD;
  static constexpr int k1p = (NX + NU + 7) / 8 * 8;
  static constexpr int nxp = (NX + 15) / 16 * 16;
  template <typename T> static constexpr bool tail4 = sizeof(T) == 8 && NX > 16 && NX <= 20;
  template <typename T> __device__ __forceinline__ static MlpDev<T> fold(const MlpDev<T>& in) {
    MlpDev<T> m = in;
    m.nx = NX; m.nu = NU; m.kin = NX + NU; m.k1p = k1p; m.n_hidden = NH; m.hpad = HPAD; m.nxp = nxp;
    m.tail4 = tail4<T> ? 1 : 0;
    if (ACT >= 0) m.act = ACT;
    return m;
  }
  template <typename T, int M, int W> static constexpr TileLds lds_map() {
    return tile_lds_dims((int)sizeof(T), HPAD, k1p, nxp, NH, M, W, LV < 2, LV == 0);
  }
  template <typename T, int M, int W>
  __device__ __forceinline__ static TileLds fold_lds(const TileLds&) { return lds_map<T, M, W>(); }
};

// Stage biases into LDS and zero the first-layer operand (call once, then __syncthreads()).
template <typename T, int W>
__device__ __forceinline__ void tile_load_constants(const MlpDev<T>& m, const TileLds& L, T* lds,
                                                    int M) {       // (callers pass folded m, L)
  const int tid = threadIdx.x;
  for (int l = 0; l < m.n_hidden; ++l)
    for (int i = tid; i < m.hpad; i += 64 * W) lds[L.bias + l * m.hpad + i] = m.B(l)[i];
  for (int i = tid; i < m.nxp; i += 64 * W) lds[L.bias + m.n_hidden * m.hpad + i] = m.B(m.n_hidden)[i];
  for (int i = tid; i < M * L.xu_stride; i += 64 * W) lds[L.xu + i] = T(0);
}

// ---- weight fragment loads -------------------------------------------------------------------
// A lane's NT consecutive fragment values for one k-step, as the widest aligned vector load.
template <typename T, int N> using vec_t = T __attribute__((ext_vector_type(N)));

// Weight fragments are fetched with BUFFER loads: one resource descriptor for the whole packed
// model (MlpDev::wbase), the stream position as a wave-uniform element offset `so` (SGPR soffset +
// immediate) and the lane's 32-bit element offset `lo` (VGPR voffset):
//   buffer_load_dwordx4 v, v_lo, s[rsrc:rsrc+3], s_so offen offset:imm
// All of a stream's address arithmetic then runs on the scalar unit.  This matters because VALU
// instructions are NOT hidden behind MFMAs on gfx950: each costs ~8 cycles of matrix-pipe time
// (calibrated by padding the hidden layer with dummy v_add_u32 / v_add_f64), and the per-lane 64-bit
// pointer form cost ~5 VALU instructions per 8 MFMAs.
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ rsrc_t weight_rsrc(const T* base) {
  // raw buffer (stride 0), no bounds clamp, gfx9 untyped-dword format word
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, -1, 0x00020000);
}

template <typename T, int NT>
__device__ __forceinline__ void load_frag(rsrc_t r, unsigned so, unsigned lo, T (&b)[NT]) {
  if constexpr (Probe::no_load) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = T((threadIdx.x + nt) & 7) * T(1e-3);
    return;
  }
  const unsigned vo = lo * (unsigned)sizeof(T), sb = so * (unsigned)sizeof(T);
  constexpr int BYTES = NT * (int)sizeof(T);
  if constexpr (BYTES == 16) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, sb, 0);
    struct P { T e[NT]; };
    const P p = __builtin_bit_cast(P, v);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = p.e[nt];
  } else if constexpr (BYTES == 8) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo, sb, 0);
    struct P { T e[NT]; };
    const P p = __builtin_bit_cast(P, v);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = p.e[nt];
  } else if constexpr (BYTES == 12) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, vo, sb, 0);   // (a 3-vector occupies 16 bytes)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = __builtin_bit_cast(T, (unsigned int)v[nt]);
  } else if constexpr (BYTES == 4) {
    b[0] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, vo, sb, 0));
  } else {                       // e.g. NT = 3 in f64: one element per load
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (sizeof(T) == 8)
        b[nt] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, vo + nt * 8u, sb, 0));
      else
        b[nt] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, vo + nt * 4u, sb, 0));
    }
  }
}

// Fetch group `g` (G k-steps) of a wave's N-split fragment stream.  so = element offset of the
// wave's slice from MlpDev::wbase (uniform); lo = lane*NT.
template <typename T, int NT, int G>
__device__ __forceinline__ void load_group(rsrc_t r, unsigned so, unsigned lo, int g, T (&b)[G][NT]) {
#pragma unroll
  for (int kk = 0; kk < G; ++kk) load_frag<T, NT>(r, so + (unsigned)(g * G + kk) * 64u * NT, lo, b[kk]);
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
struct NoSide { __device__ __forceinline__ void operator()() const {} };

// `mid` is invoked once, right after the barrier an OWN layer takes behind its first group (never
// for !OWN): caller work placed there issues between this layer's MFMAs instead of on the serial
// chain at the end of a step.
// APERM: A is an activation buffer (rows stored at act_row<T>()); false for the [x | u] operand.
template <typename T, int NT, int MT, int KS, int G, bool PIPE = false, bool OWN = false, int SG = G,
          bool APERM = false, typename Mid = NoSide>
__device__ __forceinline__ void layer_mma_static(const T* __restrict__ A, int a_stride,
                                                 rsrc_t wr, unsigned wl, int lane,
                                                 const T (&first)[G][NT],
                                                 typename Acc<T>::type (&acc)[MT][NT], int rot = 0,
                                                 Mid&& mid = Mid()) {
  static_assert(KS % G == 0 && G % SG == 0, "group sizes must divide the k extent");
  constexpr int NG = KS / G;       // groups (the unit of the rotated k order)
  constexpr int NS = KS / SG;      // sub-groups (the unit of the weight stream)
  constexpr int FS = G / SG;       // sub-groups that arrive pre-loaded in `first`
  static_assert(!OWN || (NG & (NG - 1)) == 0, "rotated k order needs a power-of-two group count");
  const int i = lane & 15, q = lane >> 4;
  const T* arow = A + (APERM ? act_row<T>(i) : i) * a_stride + q;
  T b[2][SG][NT];
  if constexpr (FS == 1) {       // whole-group streaming: the pre-loaded group IS buffer 0
#pragma unroll
    for (int kk = 0; kk < SG; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[0][kk][nt] = first[kk][nt];
  }
#pragma unroll
  for (int sgi = 0; sgi < NS; ++sgi) {
    if (sgi + 1 >= FS && sgi + 1 < NS) load_group<T, NT, SG>(wr, wl, (unsigned)lane * NT, sgi + 1, b[(sgi + 1) & 1]);
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
          acc[mt][nt] = mfma16_x<(KS < 16 ? kRealL0 : kRealHid)>(a[mt], bv, acc[mt][nt]);
        }
    }
    // (8-wave tiles only; measured neutral-to-negative with one wave per SIMD)
    // Issue order for this sub-group: LDS fragment reads run one k-step pair AHEAD of the MFMAs
    // that consume them, weight loads for the next sub-group are spread between MFMA clusters.
    //   masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read
    if constexpr (PIPE && KS >= 8 && !Probe::no_sched) {
      if constexpr (Probe::vmem_first) {
        // all of the next sub-group's weight loads up front: the full sub-group of MFMAs covers them
        __builtin_amdgcn_sched_group_barrier(0x020, SG, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
#pragma unroll
        for (int i = 0; i < SG / 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * NT, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
        }
      } else if constexpr (sizeof(T) == 4 && (Probe::f32_ahead > 0 || Probe::f32_fine > 0)) {
        // f32 (round 5): one fragment read and one weight load behind EVERY k-step's MFMAs instead of two reads and
        // two loads behind every pair -- a 32-cycle f32 MFMA covers half of what a 64-cycle f64 one does, and the
        // pair pattern left the second k-step's requests waiting behind four MFMAs: c3 f32 rollout 0.1716 ->
        // 0.1650 ms, 0.693 -> 0.72 of the f32 peak.  Reads further ahead (4, 6, 8 k-steps), a request behind every
        // half of a k-step's MFMAs, or the weight load before the fragment read measured the same or worse
        // (profiles/r05_f32_sched_variants.log).  f64 keeps the pair pattern: its hidden layer already runs at the
        // issue rate.
        constexpr int AH = Probe::f32_ahead > 0 ? Probe::f32_ahead : 2;
        __builtin_amdgcn_sched_group_barrier(0x100, AH * MT, 0);
        if constexpr (Probe::f32_fine == 1) {
#pragma unroll
          for (int i = 0; i < SG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            if (i + AH < SG) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        } else if constexpr (Probe::f32_fine == 2) {      // a request behind every half of a k-step's MFMAs
          constexpr int H1 = (MT * NT + 1) / 2, H2 = MT * NT - H1;
#pragma unroll
          for (int i = 0; i < SG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, H1, 0);
            if (i + AH < SG) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
            if (H2 > 0) __builtin_amdgcn_sched_group_barrier(0x008, H2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        } else if constexpr (Probe::f32_fine == 3) {      // weight load first, fragment read second
#pragma unroll
          for (int i = 0; i < SG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (i + AH < SG) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < SG / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * NT, 0);
            if (2 * i + AH < SG) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
            if (2 * i + AH + 1 < SG) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          }
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
#pragma unroll
        for (int i = 0; i < SG / 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * NT, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
      }
    }
    if (OWN && sgi == FS - 1) {
      lds_barrier();
      mid();
    }
    if constexpr (Probe::valu_pad) {      // calibration: 64 extra int32 VALU instructions per layer call
      if (KS >= 16 && sgi >= 2 && sgi < 10) {
        int padv = lane;
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv));
        asm volatile("" :: "v"(padv));
      }
    }
    if constexpr (Probe::valu_pad64) {    // calibration: 64 extra f64 VALU instructions per layer call
      if (KS >= 16 && sgi >= 2 && sgi < 10) {
        double padd = (double)lane;
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(padd));
        asm volatile("" :: "v"(padd));
      }
    }
  }
}


// ---- the fused network on one tile -------------------------------------------------------------
// LEAN: what stays resident in registers between calls, for callers that need registers
// themselves: 0 output fragments + hidden biases, 1 output fragments only, 2 nothing.
// WIDE: up to four 16-column output tiles (model states 33..64: long-history ARX, large Koopman
// lifts -- arx.py:146-162, koopman.py:166-181) and first layers of up to 20 k-steps (nx + nu <= 80).
// Instantiated for the 64-wide tile only (W = 4, NT = 1: such models are linear, staged with a hidden
// width equal to the state dimension); the output fragments are always resident there.
template <typename T, int NT, int MT, int W, bool DERIV = false, int LEAN = 0, typename SH = DynShape,
          bool WIDE = false>
struct TileNet {
  using acc_t = typename Acc<T>::type;
  static constexpr int M = 16 * MT;
  static constexpr int HP = 16 * NT * W;      // padded hidden width
  static constexpr int KSH = HP / 4;          // k-steps of a hidden->hidden layer
  static constexpr int KSW = KSH / W;         // output-layer k-steps per wave (= 4*NT)
  static constexpr int KS0MAX = 12;           // first-layer k-steps (k1p/4 is 2, 4, .. 12)
  static constexpr int GH = 8;                // hidden-layer group
  static constexpr int NOMAX = WIDE ? 4 : 2;  // nx <= 32 (64 when WIDE)
  // A wave's own output columns [16 NT w, 16 NT (w+1)) are one whole k-group of the next hidden
  // layer: that layer starts on them before the barrier (see run()).  Host packing must agree
  // (own_first_packing() in api.cpp).
  static constexpr bool OWN = (16 * NT == 4 * GH) && (((KSH / GH) & (KSH / GH - 1)) == 0);

  // Layer 0's fragments are RESIDENT in registers for the kernel's lifetime whenever the layer is
  // at most KS0RES k-steps (all widths in f32; k1p <= 24, i.e. nx + nu <= 24, in f64 -- HalfCheetah
  // is 23): layer 0 has only 2..12 MFMAs per tile to cover an L2 round trip with, and its loads sat
  // on the serial chain of a time step (state update -> layer 0 -> layer 1) behind in-order vmcnt
  // waits.  Wider f64 first layers keep the streamed scheme (first two k-steps prefetched, the
  // rest double-buffered in the loop): 12 k-steps would cost 48 VGPRs.
  static constexpr int KS0RES = sizeof(T) == 4 ? KS0MAX : 6;
  T pf0[KS0RES][NT];
  // layer 0 resident?  (wave-uniform; a function of the model only)
  __device__ __forceinline__ static bool resident0(const MlpDev<T>& m) {
    // (f64 tiles of 32 / 64 rows have no registers to spare: they keep the streamed scheme)
    return !Probe::no_res0 && LEAN < 2 && (sizeof(T) == 4 || MT == 1) && m.k1p <= 4 * KS0RES;
  }

  // wave-uniform element offsets (from MlpDev::wbase) of this wave's fragment streams; the lane's
  // own offset is lane*NT elements
  __device__ __forceinline__ static unsigned slice0(const MlpDev<T>& m, int w) {
    return (unsigned)(m.W(0) - m.WB()) + (unsigned)w * (unsigned)(m.k1p / 4) * 64u * NT;
  }
  __device__ __forceinline__ static unsigned slice_h(const MlpDev<T>& m, int l, int w) {
    return (unsigned)(m.W(l) - m.WB()) + (unsigned)w * (unsigned)KSH * 64u * NT;
  }
  rsrc_t wr;                      // buffer resource of the packed model (set by init())
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
  static constexpr bool RESIDENT_OUT = WIDE || (LEAN < 2 && !(MT == 4 && sizeof(T) == 8));
  T wout[KSW][NOMAX];            // dead (never written or read) when not resident

  __device__ __forceinline__ static void load_out(const MlpDev<T>& m, int w, int lane,
                                                  T (&dst)[KSW][NOMAX]) {
    if constexpr (WIDE) {              // packed [w][ks][lane][tile], tiles = nxp / 16 (1..4)
      const int tiles = m.nxp / 16;
      const T* wl = m.W(m.n_hidden) + ((size_t)w * KSW * 64 + lane) * tiles;
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
        for (int n = 0; n < NOMAX; ++n) dst[ks][n] = n < tiles ? wl[(size_t)ks * 64 * tiles + n] : T(0);
    } else if (m.nxp == 16) {
      const T* wl = m.W(m.n_hidden) + ((size_t)w * KSW * 64 + lane);
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) { dst[ks][0] = wl[ks * 64]; dst[ks][1] = T(0); }
    } else {
      const T* wl = m.W(m.n_hidden) + ((size_t)w * KSW * 64 + lane) * 2;
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) { const vec_t<T, 2> v = *reinterpret_cast<const vec_t<T, 2>*>(wl + ks * 128); dst[ks][0] = v[0]; dst[ks][1] = v[1]; }
      if (m.tail4) {          // second slot: the 4x4x4 tail fragment instead of tile 1
        const T* wt = m.WT() + ((size_t)w * KSW * 64 + lane);
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) dst[ks][1] = wt[ks * 64];
      }
    }
  }

  // Hidden-layer biases of this lane's columns, resident: the epilogue adds them without an LDS
  // read.  (16-row tiles only: the taller tiles are register-bound and read the bias from LDS.)
  static constexpr bool RESIDENT_BIAS = LEAN < 1 && (MT == 1);
  // (the first kResBias hidden layers -- the reference's default network has two; deeper layers'
  // accumulators are seeded from the LDS copy of the bias instead, one read per column tile)
  static constexpr int kResBias = 2;
  T bias_r[kResBias][NT];
  // Prefetch buffer: the first group of the next hidden layer (and, when they are not resident,
  // the output-layer fragments).  A member, not a local of run(): the first group of hidden layer 1
  // for the NEXT call is requested at the end of a call, together with layer 0's fragments, so it
  // has the caller's whole inter-call phase to arrive (measured +1 % f64, +3 % f32 on c3).
  T pfn[GH][NT];
  ProbeWave<Probe::wave_time> probe;    // (empty in the product build)

  // Once per kernel, before the first run(): resident biases / output weights + the first prefetch.
  __device__ __forceinline__ void init(const MlpDev<T>& m_in) {
    const MlpDev<T> m = SH::template fold<T>(m_in);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int l = 0; l < kResBias; ++l)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bias_r[l][nt] = (RESIDENT_BIAS && l < m.n_hidden) ? m.B(l)[16 * (NT * w + nt) + (lane & 15)] : T(0);
    wr = weight_rsrc(m.WB());
    if constexpr (RESIDENT_OUT) load_out(m, w, lane, wout);
    load0_all(m);
    // first group of hidden layer 1 for the FIRST call (later calls request it at the end of the
    // previous one): requested here so that run() has a single source for it -- a "first call"
    // branch in run() makes the prefetch buffer a phi of two register sets, i.e. 16 v_mov per call
    if (m.n_hidden > 1) load_group<T, NT, GH>(wr, slice_h(m, 1, w), (unsigned)lane * NT, 0, pfn);
  }

  // layer 0's fragments into pf0, once per kernel (init()).  Written as a fully unrolled, predicated
  // loop: a switch over the k extent gets merged by the compiler into a loop with a run-time
  // index, which forces the whole TileNet object into scratch memory.
  __device__ __forceinline__ void load0_all(const MlpDev<T>& m_in) {
    const MlpDev<T> m = SH::template fold<T>(m_in);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wl = slice0(m, w);
    // resident: the whole layer; streamed: its first two k-steps (what prefetch0 requests) -- ONE
    // code path for both, so that no two branches store to different elements of pf0
    const int ks0 = resident0(m) ? m.k1p / 4 : 2;
#pragma unroll
    for (int ks = 0; ks < KS0RES; ++ks) {
      T tmp[NT];                  // (values, not references, cross the branch: a store through a
#pragma unroll                    //  phi of two pf0 addresses would pin pf0 to the stack)
      for (int nt = 0; nt < NT; ++nt) tmp[nt] = T(0);
      if (ks < ks0) load_frag<T, NT>(wr, wl + (unsigned)ks * 64u * NT, (unsigned)lane * NT, tmp);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pf0[ks][nt] = tmp[nt];
    }
  }

  // Layer 0 from the resident fragments: KS k-steps, operands straight from pf0 (static indices:
  // no reinterpreted views of the member array, which the compiler answers with a stack copy).
  template <int KS>
  __device__ __forceinline__ void layer0_resident(const T* __restrict__ A, int a_stride, int lane,
                                                  acc_t (&acc)[MT][NT]) const {
    static_assert(KS <= KS0RES, "layer 0 wider than the resident buffer");
    const T* arow = A + (lane & 15) * a_stride + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      T a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt * 16 * a_stride + 4 * ks];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16_x<kRealL0>(a[mt], pf0[ks][nt], acc[mt][nt]);
    }
  }

  // Request layer 0's weights.  Call before the barrier/phase that precedes run(); the loads
  // complete while other work proceeds.
  __device__ __forceinline__ void prefetch0(const MlpDev<T>& m_in) {
    const MlpDev<T> m = SH::template fold<T>(m_in);
    if (resident0(m)) return;     // resident: nothing to request
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    load_group<T, NT, 2>(wr, slice0(m, w), (unsigned)lane * NT, 0, first0<2>());
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

  // As run(); `side()` is executed once by every thread at a point where it costs least: work
  // placed there is the caller's bookkeeping for the next step -- anything that does not depend
  // on this call's output, does not touch the activation / partials buffers, and may overwrite
  // the CONTROL columns of lds[L.xu] (layer 0 has consumed them).  With two or more hidden
  // layers in the own-group-first scheme it runs right after the barrier inside the second
  // hidden layer, i.e. between that layer's MFMAs (every wave has finished reading lds[L.xu]
  // by then); otherwise after the output layer's MFMAs have been issued, while the pipe drains.
  // (Measured on c3 f64, one 16-row tile per CU: at the end of the step the same work sat on
  // the serial chain for ~1.3 k cycles with the matrix pipe idle.)
  template <typename Side>
  __device__ __forceinline__ void run_side(const MlpDev<T>& m_in, const TileLds& L_in, T* lds, Side&& side,
                                           T* __restrict__ dz = nullptr, size_t dz_layer_stride = 0) {
    const MlpDev<T> m = SH::template fold<T>(m_in);
    const TileLds L = SH::template fold_lds<T, M, W>(L_in);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    AMPC_PROBE_LOCALS(probe);
    T* act = lds + L.act;            // buffer the next layer reads
    T* act_other = lds + L.act2;     // buffer the next epilogue may write (== act if single-buffered)
    const bool pingpong = L.act2 != L.act;
    const int as = L.act_stride;
    const int no = m.nxp / 16;
    static_assert(WIDE || GH * NT == KSW * NOMAX, "prefetch buffer shapes must coincide");   // pfn holds either
    auto prefetch_next = [&](int l_next) {
      if (l_next < m.n_hidden) {
        load_group<T, NT, GH>(wr, slice_h(m, l_next, w), (unsigned)lane * NT, 0, pfn);
      } else if constexpr (!RESIDENT_OUT) {
        // not resident: the output fragments ([KSW][2]) reuse the hidden prefetch buffer
        // ([GH][NT], the same 8*NT values) -- one register range for whatever comes next
        T* flat = &pfn[0][0];
        if (m.nxp == 16) {
          const T* wl = m.W(m.n_hidden) + ((size_t)w * KSW * 64 + lane);
#pragma unroll
          for (int ks = 0; ks < KSW; ++ks) { flat[2 * ks] = wl[ks * 64]; flat[2 * ks + 1] = T(0); }
        } else {
          const T* wl = m.W(m.n_hidden) + ((size_t)w * KSW * 64 + lane) * 2;
#pragma unroll
          for (int ks = 0; ks < KSW; ++ks) {
            T two[2];
            const vec_t<T, 2> v2 = *reinterpret_cast<const vec_t<T, 2>*>(wl + ks * 128);
            two[0] = v2[0]; two[1] = v2[1];
            flat[2 * ks] = two[0];
            flat[2 * ks + 1] = two[1];
          }
          if (m.tail4) {
            const T* wt = m.WT() + ((size_t)w * KSW * 64 + lane);
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
    // Accumulators start from ZERO (an inline constant of the first MFMA: no register moves) and the
    // bias is added here, from registers (RESIDENT_BIAS) or LDS.  The sum is a canonical value, so
    // relu is ONE v_max (on a raw MFMA result hipcc first emits a quieting v_max z, z).  All rows
    // of a lane's values are addressed from one base pointer with constant offsets (immediates of
    // the ds_write when the strides are compile-time, i.e. for a StaticShape): every VALU
    // instruction here costs matrix-pipe time (VALU does not overlap MFMA on gfx950).
    auto epilogue_k = [&](int l, acc_t (&acc)[MT][NT], T* dst, auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
      constexpr int RS = sizeof(T) == 8 ? 4 : 1;                    // acc_row(q, r) = acc_row(q, 0) + RS*r
      const T* bias = lds + L.bias + l * m.hpad + 16 * NT * w + i;
      T* d0 = dst + act_row<T>(acc_row<T>(q, 0)) * as + 16 * NT * w + i;   // (+ ro: bits 0-1 and >= 4 only)
      T* z0 = DERIV ? dz + (size_t)l * dz_layer_stride + (size_t)acc_row<T>(q, 0) * m.hpad + 16 * NT * w + i : nullptr;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          T bc;
          if (RESIDENT_BIAS && l < kResBias) bc = (l == 0) ? bias_r[0][nt] : bias_r[kResBias - 1][nt];
          else bc = bias[16 * nt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ro = 16 * mt + RS * r;
            const T z = acc[mt][nt][r] + bc;
            d0[ro * as + 16 * nt] = act_apply<T>(KIND, z);
            if (DERIV) z0[(size_t)ro * m.hpad + 16 * nt] = act_deriv<T>(KIND, z);
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

    bool side_done = false;
    // ---- layer 0: K = k1p (8, 16, .. 48), A = [x | u] ----------------------------------------
    {
      acc_t acc[MT][NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc_t{0, 0, 0, 0};
      const unsigned wl = slice0(m, w);
      const T* A = lds + L.xu;
      // a single hidden layer: what follows layer 0 is the output layer, whose fragments (when
      // they are not resident) travel in the prefetch buffer and are requested here, per call
      if (m.n_hidden == 1) prefetch_next(1);
      // first group of hidden layer 1: in flight under layer 0's MFMAs (64-row tiles have no
      // registers to spare for that and fetch it after the MFMAs instead)
      if (resident0(m)) {
        switch (m.k1p) {   // one fully unrolled variant per padded input width, no loads
          case 8: layer0_resident<2>(A, L.xu_stride, lane, acc); break;
          case 16: layer0_resident<4>(A, L.xu_stride, lane, acc); break;
          case 24: layer0_resident<6>(A, L.xu_stride, lane, acc); break;
          default:
            if constexpr (KS0RES >= 12) {
              switch (m.k1p) {
                case 32: layer0_resident<8>(A, L.xu_stride, lane, acc); break;
                case 40: layer0_resident<10>(A, L.xu_stride, lane, acc); break;
                default: layer0_resident<12>(A, L.xu_stride, lane, acc); break;
              }
            }
            break;
        }
      } else {
        switch (m.k1p) {
          case 8: layer_mma_static<T, NT, MT, 2, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
          case 16: layer_mma_static<T, NT, MT, 4, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
          case 24: layer_mma_static<T, NT, MT, 6, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
          case 32: layer_mma_static<T, NT, MT, 8, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
          case 40: layer_mma_static<T, NT, MT, 10, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
          case 48: layer_mma_static<T, NT, MT, 12, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
          default:
            if constexpr (WIDE) {
              switch (m.k1p) {
                case 56: layer_mma_static<T, NT, MT, 14, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
                case 64: layer_mma_static<T, NT, MT, 16, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
                case 72: layer_mma_static<T, NT, MT, 18, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
                default: layer_mma_static<T, NT, MT, 20, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc); break;
              }
            } else {
              layer_mma_static<T, NT, MT, 12, 2>(A, L.xu_stride, wr, wl, lane, first0<2>(), acc);
            }
            break;
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
          acc[mt][nt] = acc_t{0, 0, 0, 0};
        }
      // f64 streams the weights in half-groups (32-64 VGPRs less: the 64-row tile stops spilling,
      // +2 %, and the 16-row tile has room for the early prefetch); f32 keeps whole groups
      // (half-groups measured -4 % there)
      constexpr int SGH = (sizeof(T) == 8 && !Probe::sg8) ? GH / 2 : GH;
      layer_mma_static<T, NT, MT, KSH, GH, (W == 8), OWN, SGH, true>(
          act, as, wr, slice_h(m, l, w), lane, pfn, acc, w, [&] {
            if (!Probe::side_late && l == 1) { side(); side_done = true; }
          });
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
      const T* arow = act + act_row<T>(i) * as + q + 4 * w * KSW;
      if constexpr (WIDE) {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const T a = arow[mt * 16 * as + 4 * ks];
#pragma unroll
            for (int n = 0; n < NOMAX; ++n)
              if (n < no) oacc[mt][n] = mfma16_x<kRealOut>(a, wo(ks, n), oacc[mt][n]);
          }
      } else if (no == 1) {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            oacc[mt][0] = mfma16_x<kRealOut>(arow[mt * 16 * as + 4 * ks], wo(ks, 0), oacc[mt][0]);
      } else if (m.tail4) {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const T a = arow[mt * 16 * as + 4 * ks];
            oacc[mt][0] = mfma16_x<kRealOut>(a, wo(ks, 0), oacc[mt][0]);
            tacc[mt] = kRealOut ? mfma4(a, wo(ks, 1), tacc[mt]) : tacc[mt] + a * wo(ks, 1);
          }
      } else {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const T a = arow[mt * 16 * as + 4 * ks];
            oacc[mt][0] = mfma16_x<kRealOut>(a, wo(ks, 0), oacc[mt][0]);
            oacc[mt][1] = mfma16_x<kRealOut>(a, wo(ks, 1), oacc[mt][1]);
          }
      }
    }
    AMPC_MARK(7);
    prefetch0(m);     // next call's first group: overlaps the reduction and the caller's work
    if (m.n_hidden > 1) prefetch_next(1);
    // With ONE hidden layer no barrier has been passed since layer 0 read [x | u], and the side work
    // (the rollout's next actions) writes the control columns of that operand: every wave must be
    // past its layer-0 reads first.  (Deeper networks call side() inside hidden layer 1, behind the
    // barrier that follows layer 0.)
    if (!side_done) {
      if (m.n_hidden == 1) lds_barrier();
      side();
    }
    if (L.part_alias) lds_barrier();  // partials reuse `act`: every wave must be done reading it
    AMPC_MARK(8);
    const int ps = part_stride((int)sizeof(T), m.nxp);
    T* part = lds + L.part + w * M * ps;
    const int nfull = m.tail4 ? 1 : no;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int n = 0; n < NOMAX; ++n)
        if (n < nfull) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + acc_row<T>(q, r);
            part[row * ps + 16 * n + i] = oacc[mt][n][r];
          }
        }
      if (m.tail4) part[(16 * mt + tail_row(lane)) * ps + 16 + tail_col(lane)] = tacc[mt];
    }
    AMPC_MARK(13);
    lds_barrier();
    AMPC_MARK(9);
  }

  // y[row][col] (folded output: already the state increment) from the partials left by run().
  __device__ __forceinline__ static T output(const MlpDev<T>& m_in, const TileLds& L_in, const T* lds,
                                             int row, int col) {
    const MlpDev<T> m = SH::template fold<T>(m_in);
    const TileLds L = SH::template fold_lds<T, M, W>(L_in);
    const int ps = part_stride((int)sizeof(T), m.nxp);
    const T* p = lds + L.part + row * ps + col;
    T y = lds[L.bias + m.n_hidden * m.hpad + col];
#pragma unroll
    for (int w = 0; w < W; ++w) y += p[w * M * ps];
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

// The affine part of a cost block over the same rows: sum_i lin_i (v_i - g_i), plus the constant on
// the helper thread r == 0.
template <typename T>
__device__ __forceinline__ T affine_rows(const T* __restrict__ lin, const T* __restrict__ v,
                                         const T* __restrict__ g, int n, int r, int tps, T c) {
  T acc = r == 0 ? c : T(0);
  for (int i = r; i < n; i += tps) acc += lin[i] * (v[i] - g[i]);
  return acc;
}

// ---- indicator terms of an MPPI stage cost ---------------------------------------------------------
// The reference's MPPI charges whatever Cost the task holds (mppi.py:73-82), e.g. QuadCost + ThresholdCost
// or a bare BoxThresholdCost (thresh_cost.py:27-32, 73-77): 1 per time step whose observation violates the
// term, no control or terminal part (thresh_cost.py:34-38, 79-83).  Device table, kIndStride(no) values
// per term:   kind | 0 | a[no] | b[no]
//   kind 1  threshold: a = goal, b_i = threshold inside the term's observation range, +inf outside;
//                      violated where  x_i - a_i > b_i  or  a_i - x_i > b_i   (|x - g|_inf > threshold)
//   kind 2  box:       a = lower, b = upper limits;  violated where  x_i < a_i  or  x_i > b_i
// NaN observations, as in the reference: a box term compares entry by entry (a NaN entry violates nothing, the other
// entries still count, thresh_cost.py:73-77); a threshold term is `norm(diff, inf) > thr` (thresh_cost.py:27-32) --
// numpy's maximum is NaN as soon as ONE in-range entry is, and NaN > thr is false: the term is not charged at all.
constexpr int kMaxInd = 8;
__host__ __device__ constexpr int ind_stride(int no) { return 2 * no + 2; }
template <typename T> __device__ __forceinline__ bool ind_entry(int kind, T x, T a, T b) {
  if (kind == 1) { const T d = x - a; return d > b || -d > b; }
  return x < a || x > b;
}
// a NaN entry inside a threshold term's range (b = the threshold there, +inf outside) voids the term
template <typename T> __device__ __forceinline__ bool ind_void(int kind, T x, T b) {
  return kind == 1 && x != x && b < T(INFINITY);
}
// The rows of a sample split over `tps` consecutive lanes of a wave (tps a power of two <= 64; lane r of the
// group checks entries r, r + tps, ...; v[i * vs] = entry i): number of violated terms, on the group's lane
// r == 0 (zero elsewhere -- the callers' per-sample partial sums are added up over the group).  Every lane of
// the wave must call it (ballot).
template <typename T>
__device__ __forceinline__ T indicator_rows(const T* __restrict__ tab, int n_ind, const T* __restrict__ v, int vs,
                                            int no, int r, int tps) {
  T acc = T(0);
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < n_ind; ++k) {
    const T* tk = tab + (size_t)k * ind_stride(no);
    const int kind = (int)tk[0];
    bool viol = false, nan_in = false;
    for (int i = r; i < no; i += tps) {
      viol = viol || ind_entry<T>(kind, v[i * vs], tk[2 + i], tk[2 + no + i]);
      nan_in = nan_in || ind_void<T>(kind, v[i * vs], tk[2 + no + i]);
    }
    const unsigned long long bal = __ballot(viol), nal = __ballot(nan_in);
    const int sh = lane & ~(tps - 1);
    const unsigned long long msk = tps >= 64 ? ~0ull : (1ull << tps) - 1ull;
    const unsigned long long grp = tps >= 64 ? bal : (bal >> sh) & msk;
    const unsigned long long gna = tps >= 64 ? nal : (nal >> sh) & msk;
    if (r == 0 && grp != 0ull && gna == 0ull) acc += T(1);
  }
  return acc;
}
// ... with the whole observation in every lane: the count itself
template <typename T>
__device__ __forceinline__ T indicator_all(const T* __restrict__ tab, int n_ind, const T* __restrict__ v, int vs, int no) {
  T acc = T(0);
  for (int k = 0; k < n_ind; ++k) {
    const T* tk = tab + (size_t)k * ind_stride(no);
    const int kind = (int)tk[0];
    bool viol = false, nan_in = false;
    for (int i = 0; i < no; ++i) {
      viol = viol || ind_entry<T>(kind, v[i * vs], tk[2 + i], tk[2 + no + i]);
      nan_in = nan_in || ind_void<T>(kind, v[i * vs], tk[2 + no + i]);
    }
    if (viol && !nan_in) acc += T(1);
  }
  return acc;
}

}  // namespace ampc
