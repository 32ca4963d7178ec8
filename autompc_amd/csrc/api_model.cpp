// api_model.cpp -- what a handle holds: the model (ampc_set_mlp from host arrays, ampc_set_mlp_dev from device
// arrays, e.g. the parameters a PyTorch-ROCm fit has just produced; ampc_set_linear; ampc_set_sindy), cost blocks,
// indicator terms, control bounds -- and the batched model calls (ampc_mlp_pred_*, ampc_sindy_pred_*).
//
// The packed model buffer (layouts: mlp_tile.hpp, MlpDev) is a pure GATHER of the folded weights: which
// folded entry lands in which packed slot depends on the model's SHAPE only.  build_pack_map() writes
// that gather down once per (device, precision, shape) as an int32 index table; the host path applies it
// on the host, the device path applies it in one kernel after folding the normalisers on the device
// with the host's arithmetic (no contraction, same summation order), so both give the same bytes.
#include "host_common.hpp"

extern template int pred_impl<double>(ampc_handle*, const double*, const double*, double*, double*, double*, int);
extern template int pred_impl<float>(ampc_handle*, const double*, const double*, double*, double*, double*, int);

#include <map>
#include <memory>
#include <mutex>

// ---------------------------------------------------------------------------------------------
// packing orders, written over an element functor B(k, n) -> value (host: int32 source index)
// ---------------------------------------------------------------------------------------------
// N-split over W waves, NT tiles per wave.  own_first: wave w's stream starts at k-group w (8 k-steps
// per group) and wraps around -- the order TileNet::run consumes a hidden layer in when TileNet::OWN holds.
template <typename V, typename F>
static void pack_nsplit(std::vector<V>& dst, V zero, int kpad, int hpad, int NT, int W, F B, bool own_first = false) {
  const int KS = kpad / 4, G = 8, NG = KS / G;
  dst.assign((size_t)kpad * hpad, zero);
  for (int w = 0; w < W; ++w)
    for (int pos = 0; pos < KS; ++pos)
      for (int lane = 0; lane < 64; ++lane)
        for (int nt = 0; nt < NT; ++nt) {
          const int ks = own_first ? ((pos / G + w) % NG) * G + pos % G : pos;
          const int k = 4 * ks + (lane >> 4);
          const int n = 16 * (NT * w + nt) + (lane & 15);
          dst[(((size_t)w * KS + pos) * 64 + lane) * NT + nt] = B(k, n);
        }
}
// mirrors TileNet::OWN (mlp_tile.hpp): own columns = one k-group, power-of-two group count
static bool own_first_packing(int NT, int W) {
  const int ng = (16 * NT * W / 4) / 8;
  return 16 * NT == 32 && (ng & (ng - 1)) == 0;
}
// K-split over W waves, `tiles` 16-column tiles.
template <typename V, typename F>
static void pack_ksplit(std::vector<V>& dst, V zero, int hpad, int tiles, int W, F B) {
  const int KS = hpad / 4, KSW = KS / W;
  dst.assign((size_t)hpad * tiles * 16, zero);
  for (int w = 0; w < W; ++w)
    for (int ksl = 0; ksl < KSW; ++ksl)
      for (int lane = 0; lane < 64; ++lane)
        for (int t = 0; t < tiles; ++t) {
          const int k = 4 * (w * KSW + ksl) + (lane >> 4);
          const int n = 16 * t + (lane & 15);
          dst[(((size_t)w * KSW + ksl) * 64 + lane) * tiles + t] = B(k, n);
        }
}

// The folded parameters as ONE flat array: W'_0 | ... | W'_L | b'_0 | ... | b'_L  (torch.nn.Linear layout
// [out][in] per layer); FoldLayout gives the offsets.
struct FoldLayout {
  int L = 0, nx = 0, nu = 0;
  int in[kMaxHidden + 1], out[kMaxHidden + 1];
  size_t w_off[kMaxHidden + 1], b_off[kMaxHidden + 1], total = 0;
};
static FoldLayout fold_layout(const ampc_handle* h) {
  FoldLayout f;
  f.L = h->n_hidden; f.nx = h->nx; f.nu = h->nu;
  size_t o = 0;
  for (int l = 0; l <= f.L; ++l) {
    f.in[l] = l == 0 ? h->nx + h->nu : h->hidden[l - 1];
    f.out[l] = l == f.L ? h->nx : h->hidden[l];
    f.w_off[l] = o;
    o += (size_t)f.in[l] * f.out[l];
  }
  for (int l = 0; l <= f.L; ++l) { f.b_off[l] = o; o += f.out[l]; }
  f.total = o;
  return f;
}

// The gather: packed slot i takes folded[src[i]] (src < 0: zero padding); off[] = the arrays' starts in
// the order MlpDev's pointers are assigned (assign_pointers below).
struct PackMap {
  std::vector<int32_t> src;
  std::vector<size_t> off;
  size_t total = 0;
  bool tail4 = false;
  DevBuf dev;                 // src on the device (device path only, uploaded on first use)
  int device = -1;
  ~PackMap() { if (dev.p) { (void)hipSetDevice(device); dev.release(); } }
};

template <typename T> static bool wants_tail4(const ampc_handle* h) {
  return sizeof(T) == 8 && h->nx > 16 && h->nx <= 20 && env_int("AMPC_TAIL4", 1) != 0;
}

template <typename T> static void build_pack_map(const ampc_handle* h, PackMap* pm) {
  const FoldLayout f = fold_layout(h);
  const int L = f.L, nx = h->nx, kin = nx + h->nu;
  const int hpad = h->hpad, NT = h->nt, W = h->nw, k1p = h->k1p, nxp = h->nxp;
  typedef int32_t I;
  const I none = -1;
  std::vector<std::vector<I>> parts;  // in upload order
  auto push = [&](std::vector<I>&& v) { parts.emplace_back(std::move(v)); };
  auto Wsrc = [&](int l, int row, int col) { return (I)(f.w_off[l] + (size_t)row * f.in[l] + col); };
  // forward weights w[0..L]:  B[k][n] = W'_l[n][k]
  for (int l = 0; l <= L; ++l) {
    const int in = f.in[l], out = f.out[l];
    std::vector<I> pk;
    auto Bt = [&](int k, int n) { return (n < out && k < in) ? Wsrc(l, n, k) : none; };
    if (l < L) pack_nsplit(pk, none, l == 0 ? k1p : hpad, hpad, NT, W, Bt, l > 0 && own_first_packing(NT, W));
    else pack_ksplit(pk, none, hpad, nxp / 16, W, Bt);
    push(std::move(pk));
  }
  // tail fragments for the 4x4x4 output path (MlpDev::wt): f64, 16 < nx <= 20
  pm->tail4 = wants_tail4<T>(h);
  {
    const int KSW = hpad / 4 / W;
    std::vector<I> wt((size_t)W * KSW * 64, none);
    if (pm->tail4) {
      const int in = f.in[L];
      for (int w = 0; w < W; ++w)
        for (int ksl = 0; ksl < KSW; ++ksl)
          for (int lane = 0; lane < 64; ++lane) {
            const int k = 4 * (w * KSW + ksl) + lane / 16, col = 16 + lane % 4;
            wt[((size_t)w * KSW + ksl) * 64 + lane] = (col < nx && k < in) ? Wsrc(L, col, k) : none;
          }
    }
    push(std::move(wt));
  }
  // biases b[0..L]
  for (int l = 0; l <= L; ++l) {
    std::vector<I> bb(l < L ? hpad : nxp, none);
    for (int i = 0; i < f.out[l]; ++i) bb[i] = (I)(f.b_off[l] + i);
    push(std::move(bb));
  }
  // Jacobian-chain weights wj[0..L-1]: B[k][n] = W'_l[k][n]  (k = out index, n = in index)
  const int ni = (kin + 15) / 16;
  for (int l = 0; l < L; ++l) {
    const int in = f.in[l], out = f.out[l];
    std::vector<I> pk;
    auto Bn = [&](int k, int n) { return (k < out && n < in) ? Wsrc(l, k, n) : none; };
    if (l == 0) pack_ksplit(pk, none, hpad, ni, W, Bn);
    else pack_nsplit(pk, none, hpad, hpad, NT, W, Bn);
    push(std::move(pk));
  }
  // folded output weights in plain [nx][hpad]
  {
    std::vector<I> wp((size_t)nx * hpad, none);
    const int in = f.in[L];
    for (int i = 0; i < nx; ++i)
      for (int k = 0; k < in; ++k) wp[(size_t)i * hpad + k] = Wsrc(L, i, k);
    push(std::move(wp));
  }
  // four-wave packing of the forward weights (MlpDev::w4, ilqr_ls4.hpp): N-split layers as
  // [wave][k-step][chunk][lane][cw] with cw = 2 values per lane for even NT4, else 1 -- every fragment
  // load is then one fully coalesced 16- or 8-byte-per-lane access -- and a K-split output layer
  {
    const int NT4 = hpad / 64;
    const int cw = NT4 % 2 == 0 ? 2 : 1, chunks = NT4 / cw;
    for (int l = 0; l <= L; ++l) {
      const int in = f.in[l], out = f.out[l];
      std::vector<I> pk;
      auto Bt = [&](int k, int n) { return (n < out && k < in) ? Wsrc(l, n, k) : none; };
      if (l < L) {
        const int KS = (l == 0 ? k1p : hpad) / 4;
        pk.assign((size_t)KS * 4 * hpad, none);
        for (int w = 0; w < 4; ++w)
          for (int pos = 0; pos < KS; ++pos)
            for (int c = 0; c < chunks; ++c)
              for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < cw; ++e) {
                  const int nt = c * cw + e;
                  pk[((((size_t)w * KS + pos) * chunks + c) * 64 + lane) * cw + e] =
                      Bt(4 * pos + (lane >> 4), 16 * (NT4 * w + nt) + (lane & 15));
                }
      } else {
        pack_ksplit(pk, none, hpad, nxp / 16, 4, Bt);
      }
      push(std::move(pk));
    }
  }
  size_t total = 0;
  pm->off.clear();
  for (auto& v : parts) {
    pm->off.push_back(total);
    total += (v.size() + 3) / 4 * 4;  // keep every array 16/32-byte aligned
  }
  pm->total = total;
  pm->src.assign(total, none);
  for (size_t i = 0; i < parts.size(); ++i)
    std::memcpy(pm->src.data() + pm->off[i], parts[i].data(), parts[i].size() * sizeof(I));
}

// one map per (device, precision, shape), shared by every handle that stages such a model
static std::mutex g_map_mu;
static std::map<std::string, std::shared_ptr<PackMap>> g_maps;
template <typename T> static std::shared_ptr<PackMap> pack_map_of(const ampc_handle* h) {
  std::string key = std::to_string(h->device) + (sizeof(T) == 8 ? ":d:" : ":f:") + std::to_string(h->nx) + "," +
                    std::to_string(h->nu) + (wants_tail4<T>(h) ? ",t" : ",-");
  for (int l = 0; l < h->n_hidden; ++l) key += "," + std::to_string(h->hidden[l]);
  std::lock_guard<std::mutex> lock(g_map_mu);
  auto it = g_maps.find(key);
  if (it != g_maps.end()) return it->second;
  if (g_maps.size() >= 64) g_maps.clear();          // (a tuner walks through many shapes: bounded; users keep theirs alive)
  auto pm = std::make_shared<PackMap>();
  pm->device = h->device;
  build_pack_map<T>(h, pm.get());
  g_maps[key] = pm;
  return pm;
}

template <typename T> static void assign_pointers(ampc_handle* h, const PackMap& pm) {
  const int L = h->n_hidden;
  MlpDev<T>& m = model_of<T>(h);
  std::memset(&m, 0, sizeof(m));
  m.nx = h->nx; m.nu = h->nu; m.kin = h->nx + h->nu; m.k1p = h->k1p; m.n_hidden = L; m.hpad = h->hpad; m.nxp = h->nxp;
  m.act = h->act;
  const T* base = (const T*)h->model_buf.p;
  m.wbase = base;
  size_t idx = 0;
  for (int l = 0; l <= L; ++l) m.w[l] = base + pm.off[idx++];
  m.wt = base + pm.off[idx++];
  m.tail4 = pm.tail4 ? 1 : 0;
  for (int l = 0; l <= L; ++l) m.b[l] = base + pm.off[idx++];
  for (int l = 0; l < L; ++l) m.wj[l] = base + pm.off[idx++];
  h->wout_plain = (const void*)(base + pm.off[idx++]);
  for (int l = 0; l <= L; ++l) m.w4[l] = base + pm.off[idx++];
}

// ---------------------------------------------------------------------------------------------
// host path
// ---------------------------------------------------------------------------------------------
// Fold the affine normalisers into the first / last layer (double precision, see mlp_tile.hpp):
//   W'_0 = W_0 diag(1/xu_std), b'_0 = b_0 - W'_0 xu_mean; W'_L = diag(dy_std) W_L, b'_L = dy_std b_L + dy_mean
static void fold_host(const ampc_handle* h, const FoldLayout& f, std::vector<double>* folded) {
  const int L = f.L, nx = f.nx, kin = nx + f.nu;
  const double* xmean = h->norm.data();
  const double* xstd = xmean + kin;
  const double* dmean = xstd + kin;
  const double* dstd = dmean + nx;
  folded->assign(f.total, 0.0);
  double* F = folded->data();
  for (int l = 0; l <= L; ++l) {
    std::memcpy(F + f.w_off[l], h->W[l].data(), h->W[l].size() * 8);
    std::memcpy(F + f.b_off[l], h->b[l].data(), h->b[l].size() * 8);
  }
  for (int n = 0; n < f.out[0]; ++n) {
    double shift = 0.0;
    for (int k = 0; k < kin; ++k) {
      const double w = h->W[0][(size_t)n * kin + k] / xstd[k];
      F[f.w_off[0] + (size_t)n * kin + k] = w;
      shift += w * xmean[k];
    }
    F[f.b_off[0] + n] = h->b[0][n] - shift;
  }
  const int inL = f.in[L];
  for (int i = 0; i < nx; ++i) {
    for (int k = 0; k < inL; ++k) F[f.w_off[L] + (size_t)i * inL + k] *= dstd[i];
    F[f.b_off[L] + i] = F[f.b_off[L] + i] * dstd[i] + dmean[i];
  }
}

template <typename T> static int build_model(ampc_handle* h) {
  const FoldLayout f = fold_layout(h);
  std::vector<double> folded;
  fold_host(h, f, &folded);
  std::shared_ptr<PackMap> pm = pack_map_of<T>(h);
  std::vector<double> flat(pm->total);
  for (size_t i = 0; i < pm->total; ++i) flat[i] = pm->src[i] < 0 ? 0.0 : folded[(size_t)pm->src[i]];
  HIP_OK(h->model_buf.reserve(pm->total * sizeof(T)));
  HIP_OK(upload_converted<T>(h->model_buf.p, flat.data(), pm->total, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  assign_pointers<T>(h, *pm);
  return 0;
}

static int check_shape(ampc_handle* h, int nx, int nu, int n_hidden, const int* hidden_sizes, int activation,
                       const char* who) {
  const std::string w(who);
  REQUIRE(h, w + ": NULL handle");
  REQUIRE(nx >= 1 && nx <= 64, w + ": state dim must be in 1..64");
  REQUIRE(nu >= 1 && nu <= kMaxNu, w + ": ctrl dim must be in 1..16");
  REQUIRE(n_hidden >= 1 && n_hidden <= kMaxHidden, w + ": 1..4 hidden layers");
  REQUIRE(activation >= 0 && activation <= 4, w + ": unknown activation");
  REQUIRE(hidden_sizes, w + ": NULL argument");
  int hmax = 0;
  for (int l = 0; l < n_hidden; ++l) {
    REQUIRE(hidden_sizes[l] >= 1 && hidden_sizes[l] <= 256, w + ": hidden size 1..256");
    hmax = hidden_sizes[l] > hmax ? hidden_sizes[l] : hmax;
  }
  // (more than 32 states: the WIDE tile, three or four output column tiles, any hidden width)
  h->nx = nx; h->nu = nu; h->n_hidden = n_hidden; h->act = activation;
  for (int l = 0; l < kMaxHidden; ++l) h->hidden[l] = l < n_hidden ? hidden_sizes[l] : 0;
  // Workgroup shape: 8 waves (two per SIMD) whenever the padded width allows whole 16-column
  // tiles per wave, else 4 waves.  (W, NT) in {(4,1), (8,1), (4,3), (8,2)} for hpad 64..256.
  h->hpad = round_up(hmax, 64);
  h->nw = h->hpad % 128 == 0 ? 8 : 4;
  h->nt = h->hpad / (16 * h->nw);
  h->k1p = round_up(nx + nu, 8);
  h->nxp = round_up(nx, 16);
  return 0;
}

static void model_staged(ampc_handle* h) {
  h->has_mlp = true;
  h->has_sindy = false;
  h->has_lin = false;
  ampc_internal_jit_kick(h);   // (needs obs_dim: if the cost is set later, ampc_set_quad_costs starts it)
}

extern "C" int ampc_set_mlp(ampc_handle* h, int nx, int nu, int n_hidden, const int* hidden_sizes,
                            int activation, const double* const* weights,
                            const double* const* biases, const double* xu_mean,
                            const double* xu_std, const double* dy_mean, const double* dy_std) {
  if (int rc = check_shape(h, nx, nu, n_hidden, hidden_sizes, activation, "ampc_set_mlp")) return rc;
  REQUIRE(weights && biases && xu_mean && xu_std && dy_mean && dy_std, "ampc_set_mlp: NULL argument");
  HIP_OK(hipSetDevice(h->device));
  h->W.assign(n_hidden + 1, {});
  h->b.assign(n_hidden + 1, {});
  for (int l = 0; l <= n_hidden; ++l) {
    const int in = l == 0 ? nx + nu : hidden_sizes[l - 1];
    const int out = l == n_hidden ? nx : hidden_sizes[l];
    h->W[l].assign(weights[l], weights[l] + (size_t)in * out);
    h->b[l].assign(biases[l], biases[l] + out);
  }
  const int kin = nx + nu;
  h->norm.resize(2 * kin + 2 * nx);
  std::memcpy(h->norm.data(), xu_mean, kin * 8);
  std::memcpy(h->norm.data() + kin, xu_std, kin * 8);
  std::memcpy(h->norm.data() + 2 * kin, dy_mean, nx * 8);
  std::memcpy(h->norm.data() + 2 * kin + nx, dy_std, nx * 8);
  int rc = h->precision == AMPC_F64 ? build_model<double>(h) : build_model<float>(h);
  if (rc) return rc;
  model_staged(h);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// device path
// ---------------------------------------------------------------------------------------------
struct FoldArgs {
  FoldLayout f;
  const double* W[kMaxHidden + 1];
  const double* b[kMaxHidden + 1];
  const double* xmean;
  const double* xstd;
  const double* dmean;
  const double* dstd;
};

// Middle layers and raw biases: plain copies.  First layer: one thread per output row n, the same
// left-to-right sum as fold_host.  Last layer: one thread per entry.  No fused multiply-add anywhere: the host
// code (x86-64 baseline) has none, and HIP's __dmul_rn / __dadd_rn are plain operators that the compiler
// would contract -- the pragma is what keeps every product and sum separately rounded.
__global__ void fold_model_kernel(FoldArgs a, double* __restrict__ F) {
#pragma clang fp contract(off)
  const FoldLayout& f = a.f;
  const int L = f.L, kin = f.nx + f.nu;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
  for (int l = 1; l < L; ++l) {
    const size_t nw = (size_t)f.in[l] * f.out[l];
    for (size_t i = tid; i < nw; i += nthreads) F[f.w_off[l] + i] = a.W[l][i];
    for (size_t i = tid; i < (size_t)f.out[l]; i += nthreads) F[f.b_off[l] + i] = a.b[l][i];
  }
  for (size_t n = tid; n < (size_t)f.out[0]; n += nthreads) {
    double shift = 0.0;
    for (int k = 0; k < kin; ++k) {
      const double w = a.W[0][n * kin + k] / a.xstd[k];
      F[f.w_off[0] + n * kin + k] = w;
      const double prod = w * a.xmean[k];
      shift = shift + prod;
    }
    F[f.b_off[0] + n] = a.b[0][n] - shift;
  }
  const int inL = f.in[L];
  // (L >= 1: the first and the last layer are different layers, each written by one loop only)
  for (size_t i = tid; i < (size_t)f.nx * inL; i += nthreads) {
    const size_t row = i / inL;
    F[f.w_off[L] + i] = a.W[L][i] * a.dstd[row];
  }
  for (size_t i = tid; i < (size_t)f.nx; i += nthreads) {
    const double scaled = a.b[L][i] * a.dstd[i];
    F[f.b_off[L] + i] = scaled + a.dmean[i];
  }
}

template <typename T>
__global__ void gather_model_kernel(const int32_t* __restrict__ src, const double* __restrict__ F, T* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int32_t s = src[i];
    dst[i] = s < 0 ? (T)0 : (T)F[s];
  }
}

template <typename T> static int build_model_dev(ampc_handle* h, FoldArgs& a) {
  std::shared_ptr<PackMap> pm = pack_map_of<T>(h);
  {
    std::lock_guard<std::mutex> lock(g_map_mu);
    if (!pm->dev.p) {
      HIP_OK(pm->dev.reserve(pm->total * sizeof(int32_t)));
      HIP_OK(hipMemcpy(pm->dev.p, pm->src.data(), pm->total * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }
  ScopedBuf folded;
  HIP_OK(folded.reserve(a.f.total * sizeof(double)));
  HIP_OK(h->model_buf.reserve(pm->total * sizeof(T)));
  hipLaunchKernelGGL(fold_model_kernel, dim3(64), dim3(256), 0, h->stream, a, (double*)folded.p);
  hipLaunchKernelGGL(gather_model_kernel<T>, dim3((unsigned)((pm->total + 255) / 256)), dim3(256), 0, h->stream,
                     (const int32_t*)pm->dev.p, (const double*)folded.p, (T*)h->model_buf.p, pm->total);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  assign_pointers<T>(h, *pm);
  return 0;
}

extern "C" int ampc_set_mlp_dev(ampc_handle* h, int nx, int nu, int n_hidden, const int* hidden_sizes,
                                int activation, const double* const* weights_dev,
                                const double* const* biases_dev, const double* xu_mean_dev,
                                const double* xu_std_dev, const double* dy_mean_dev, const double* dy_std_dev) {
  if (int rc = check_shape(h, nx, nu, n_hidden, hidden_sizes, activation, "ampc_set_mlp_dev")) return rc;
  REQUIRE(weights_dev && biases_dev && xu_mean_dev && xu_std_dev && dy_mean_dev && dy_std_dev,
          "ampc_set_mlp_dev: NULL argument");
  HIP_OK(hipSetDevice(h->device));
  FoldArgs a;
  a.f = fold_layout(h);
  for (int l = 0; l <= kMaxHidden; ++l) { a.W[l] = nullptr; a.b[l] = nullptr; }
  for (int l = 0; l <= n_hidden; ++l) {
    REQUIRE(weights_dev[l] && biases_dev[l], "ampc_set_mlp_dev: NULL layer pointer");
    hipPointerAttribute_t attr;
    REQUIRE(hipPointerGetAttributes(&attr, weights_dev[l]) == hipSuccess && attr.type == hipMemoryTypeDevice &&
                attr.device == h->device,
            "ampc_set_mlp_dev: weights must be device memory of the handle's device");
    a.W[l] = weights_dev[l];
    a.b[l] = biases_dev[l];
  }
  a.xmean = xu_mean_dev; a.xstd = xu_std_dev; a.dmean = dy_mean_dev; a.dstd = dy_std_dev;
  h->W.clear(); h->b.clear(); h->norm.clear();       // (no host copy on this path)
  int rc = h->precision == AMPC_F64 ? build_model_dev<double>(h, a) : build_model_dev<float>(h, a);
  if (rc) return rc;
  model_staged(h);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// linear models, cost blocks, bounds, batched prediction
// ---------------------------------------------------------------------------------------------
// Wide linear model (65 .. 256 states; AMPC_LINEAR_WIDE = 1: any size): [A | B] packed in MFMA
// fragment order for linear_kernels.hpp, plus a plain copy.
constexpr int kLinMaxNx = 256;
static int set_linear_wide(ampc_handle* h, int nx, int nu, const double* A, const double* B) {
  HIP_OK(hipSetDevice(h->device));
  const int k = nx + nu, nxp = round_up(nx, 16), kp = round_up(k, 4), ntile = nxp / 16, ksn = kp / 4;
  const int ldj = round_up(k, 16), plain_sz = round_up(nx * k, 4);
  std::vector<double> buf((size_t)ntile * ksn * 64 + (size_t)plain_sz + (size_t)nxp * ldj, 0.0);
  auto Mat = [&](int row, int col) -> double {
    if (row >= nx || col >= k) return 0.0;
    return col < nx ? A[(size_t)row * nx + col] : B[(size_t)row * nu + (col - nx)];
  };
  for (int nt = 0; nt < ntile; ++nt)
    for (int ks = 0; ks < ksn; ++ks)
      for (int l = 0; l < 64; ++l)
        buf[((size_t)nt * ksn + ks) * 64 + l] = Mat(16 * nt + (l & 15), 4 * ks + (l >> 4));
  double* plain = buf.data() + (size_t)ntile * ksn * 64;
  for (int r = 0; r < nx; ++r)
    for (int c = 0; c < k; ++c) plain[(size_t)r * k + c] = Mat(r, c);
  double* jp = plain + plain_sz;                         // [nxp][ldj], zero padded: the wide iLQR sweep's J
  for (int r = 0; r < nx; ++r)
    for (int c = 0; c < k; ++c) jp[(size_t)r * ldj + c] = Mat(r, c);
  HIP_OK(h->lin_buf.reserve(buf.size() * h->esz()));
  if (h->precision == AMPC_F64) HIP_OK(upload_converted<double>(h->lin_buf.p, buf.data(), buf.size(), h->stream));
  else HIP_OK(upload_converted<float>(h->lin_buf.p, buf.data(), buf.size(), h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (nx != h->nx || nu != h->nu) { h->n_costs = 0; h->obs_dim = 0; h->has_bounds = false; h->n_ind = 0; }   // other dimensions
  h->nx = nx; h->nu = nu; h->l_nxp = nxp; h->l_kp = kp;
  h->n_hidden = 0; h->act = 4; h->hpad = 0;
  std::memset(&h->md, 0, sizeof(h->md));
  std::memset(&h->mf, 0, sizeof(h->mf));
  h->md.nx = h->mf.nx = nx; h->md.nu = h->mf.nu = nu; h->md.kin = h->mf.kin = nx + nu;
  h->has_lin = true; h->has_mlp = false; h->has_sindy = false;
  return 0;
}

// x' = A x + B u, staged as the one-hidden-layer identity-activation network
//   x' = x + I * ([A - I | B] [x; u])
// so that every kernel written for the MLP (rollout, Jacobians, iLQR, closed loop) serves the
// linear models too.  Multiplying by the identity output layer is exact; A - I rounds once.
extern "C" int ampc_set_linear(ampc_handle* h, int nx, int nu, const double* A, const double* B) {
  REQUIRE(h && A && B, "ampc_set_linear: NULL argument");
  REQUIRE(nx >= 1 && nx <= kLinMaxNx, "ampc_set_linear: state dim must be in 1..256");
  REQUIRE(nu >= 1 && nu <= kMaxNu, "ampc_set_linear: ctrl dim must be in 1..16");
  if (nx > 64 || env_int("AMPC_LINEAR_WIDE", 0) != 0) return set_linear_wide(h, nx, nu, A, B);
  const int kin = nx + nu;
  std::vector<double> w0((size_t)nx * kin), w1((size_t)nx * nx, 0.0), b0(nx, 0.0);
  for (int i = 0; i < nx; ++i) {
    for (int j = 0; j < nx; ++j) w0[(size_t)i * kin + j] = A[(size_t)i * nx + j] - (i == j ? 1.0 : 0.0);
    for (int j = 0; j < nu; ++j) w0[(size_t)i * kin + nx + j] = B[(size_t)i * nu + j];
    w1[(size_t)i * nx + i] = 1.0;
  }
  std::vector<double> zeros(kin, 0.0), ones(kin, 1.0);
  const double* ws[2] = {w0.data(), w1.data()};
  const double* bs[2] = {b0.data(), b0.data()};
  const int hidden = nx;
  return ampc_set_mlp(h, nx, nu, 1, &hidden, 4, ws, bs, zeros.data(), ones.data(), zeros.data(),
                      ones.data());
}

extern "C" int ampc_set_affine_quad_costs(ampc_handle* h, int n_costs, int obs_dim, const double* Q,
                                          const double* R, const double* F, const double* goal,
                                          const double* lin, const double* lin_term, const double* consts) {
  REQUIRE(h && Q && R && F && goal, "ampc_set_affine_quad_costs: NULL argument");
  REQUIRE(h->has_model(), "ampc_set_affine_quad_costs: set the model first");
  REQUIRE(n_costs >= 1, "ampc_set_affine_quad_costs: n_costs < 1");
  REQUIRE(obs_dim >= 1 && obs_dim <= h->nx, "ampc_set_affine_quad_costs: obs_dim must be <= state dim");
  HIP_OK(hipSetDevice(h->device));
  const int no = obs_dim, nu = h->nu;
  const int stride = cost_block_stride(no, nu);
  std::vector<double> flat((size_t)n_costs * stride, 0.0);
  bool affine = false;
  for (int c = 0; c < n_costs; ++c) {
    double* d = flat.data() + (size_t)c * stride;
    std::memcpy(d, Q + (size_t)c * no * no, no * no * 8);
    std::memcpy(d + no * no, R + (size_t)c * nu * nu, nu * nu * 8);
    std::memcpy(d + no * no + nu * nu, F + (size_t)c * no * no, no * no * 8);
    std::memcpy(d + cost_off_goal(no, nu), goal + (size_t)c * no, no * 8);
    if (lin) std::memcpy(d + cost_off_lin(no, nu), lin + (size_t)c * no, no * 8);
    if (lin_term) std::memcpy(d + cost_off_lint(no, nu), lin_term + (size_t)c * no, no * 8);
    if (consts) std::memcpy(d + cost_off_c(no, nu), consts + (size_t)c * 2, 2 * 8);
    for (int i = cost_off_lin(no, nu); i < cost_off_c(no, nu) + 2; ++i) affine = affine || d[i] != 0.0;
  }
  HIP_OK(h->cost_buf.reserve(flat.size() * h->esz()));
  if (h->precision == AMPC_F64) HIP_OK(upload_converted<double>(h->cost_buf.p, flat.data(), flat.size(), h->stream));
  else HIP_OK(upload_converted<float>(h->cost_buf.p, flat.data(), flat.size(), h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->n_costs = n_costs;
  if (h->obs_dim != obs_dim) h->n_ind = 0;      // (the indicator table is laid out for the old observation)
  h->obs_dim = obs_dim;
  h->cost_stride = stride;
  h->cost_affine = affine ? 1 : 0;
  bool diag = true;
  for (int c = 0; c < n_costs && diag; ++c) {
    for (int i = 0; i < no && diag; ++i)
      for (int j = 0; j < no; ++j)
        if (i != j && (Q[((size_t)c * no + i) * no + j] != 0.0 || F[((size_t)c * no + i) * no + j] != 0.0)) diag = false;
    for (int i = 0; i < nu && diag; ++i)
      for (int j = 0; j < nu; ++j)
        if (i != j && R[((size_t)c * nu + i) * nu + j] != 0.0) diag = false;
  }
  h->cost_diag = (diag && env_int("AMPC_DENSE_COST", 0) == 0) ? 1 : 0;
  ampc_internal_jit_kick(h);
  return 0;
}

extern "C" int ampc_set_quad_costs(ampc_handle* h, int n_costs, int obs_dim, const double* Q,
                                   const double* R, const double* F, const double* goal) {
  REQUIRE(h && Q && R && F && goal, "ampc_set_quad_costs: NULL argument");
  return ampc_set_affine_quad_costs(h, n_costs, obs_dim, Q, R, F, goal, nullptr, nullptr, nullptr);
}

// Indicator terms of the MPPI stage cost (mlp_tile.hpp: indicator_rows): threshold / box terms of the
// controller's cost, added to every cost block's stage cost of x_0 .. x_{H-1}
extern "C" int ampc_set_indicator_costs(ampc_handle* h, int n_terms, const int* kinds, const double* params) {
  REQUIRE(h, "ampc_set_indicator_costs: NULL handle");
  REQUIRE(n_terms >= 0 && n_terms <= kMaxInd, "ampc_set_indicator_costs: at most 8 indicator terms");
  if (n_terms == 0) { h->n_ind = 0; return 0; }
  REQUIRE(kinds && params, "ampc_set_indicator_costs: NULL argument");
  REQUIRE(h->n_costs > 0 && h->obs_dim > 0,
          "ampc_set_indicator_costs: set the quadratic part first (ampc_set_quad_costs; zeros for a cost without one)");
  HIP_OK(hipSetDevice(h->device));
  const int no = h->obs_dim, st = ind_stride(no);
  std::vector<double> tab((size_t)n_terms * st, 0.0);
  const double* par = params;
  for (int k = 0; k < n_terms; ++k) {
    double* t = tab.data() + (size_t)k * st;
    if (kinds[k] == SCORE_THRESHOLD) {            // goal[no] lo hi threshold  (as ampc_score_trajectories)
      const int lo = std::max(0, (int)par[no]), hi = std::min(no, (int)par[no + 1]);
      t[0] = 1.0;
      for (int i = 0; i < no; ++i) {
        t[2 + i] = par[i];
        t[2 + no + i] = (i >= lo && i < hi) ? par[no + 2] : INFINITY;
      }
      par += no + 3;
    } else if (kinds[k] == SCORE_BOX) {           // lower[no] upper[no]
      t[0] = 2.0;
      for (int i = 0; i < no; ++i) { t[2 + i] = par[i]; t[2 + no + i] = par[no + i]; }
      par += 2 * no;
    } else {
      return fail("ampc_set_indicator_costs: kinds must be 1 (threshold) or 2 (box)");
    }
  }
  HIP_OK(h->ind_buf.reserve(tab.size() * h->esz()));
  if (h->precision == AMPC_F64) HIP_OK(upload_converted<double>(h->ind_buf.p, tab.data(), tab.size(), h->stream));
  else HIP_OK(upload_converted<float>(h->ind_buf.p, tab.data(), tab.size(), h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->n_ind = n_terms;
  return 0;
}

extern "C" int ampc_set_ctrl_bounds(ampc_handle* h, const double* lo, const double* hi) {
  REQUIRE(h && lo && hi, "ampc_set_ctrl_bounds: NULL argument");
  REQUIRE(h->has_model(), "ampc_set_ctrl_bounds: set the model first");
  HIP_OK(hipSetDevice(h->device));
  const int nu = h->nu;
  h->lo.assign(lo, lo + nu);
  h->hi.assign(hi, hi + nu);
  // MPPI works in units of umax (ctrl_scale = umax, mppi.py:100-102): store lo/scale, hi/scale, scale.
  std::vector<double> flat(3 * nu);
  for (int j = 0; j < nu; ++j) {
    flat[j] = lo[j] / hi[j];
    flat[nu + j] = hi[j] / hi[j];
    flat[2 * nu + j] = hi[j];
  }
  HIP_OK(h->bounds_buf.reserve(flat.size() * h->esz()));
  if (h->precision == AMPC_F64) HIP_OK(upload_converted<double>(h->bounds_buf.p, flat.data(), flat.size(), h->stream));
  else HIP_OK(upload_converted<float>(h->bounds_buf.p, flat.data(), flat.size(), h->stream));
  std::vector<double> raw(2 * nu);
  for (int j = 0; j < nu; ++j) { raw[j] = lo[j]; raw[nu + j] = hi[j]; }
  HIP_OK(h->ubounds_buf.reserve(raw.size() * h->esz()));
  if (h->precision == AMPC_F64) HIP_OK(upload_converted<double>(h->ubounds_buf.p, raw.data(), raw.size(), h->stream));
  else HIP_OK(upload_converted<float>(h->ubounds_buf.p, raw.data(), raw.size(), h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->has_bounds = true;
  return 0;
}

// Model.pred_batch / pred_diff_batch of a wide linear model (linear_kernels.hpp)
template <typename T>
static int lin_pred_impl(ampc_handle* h, const double* states, const double* ctrls, double* out, double* jx,
                         double* ju, int n) {
  const int nx = h->nx, nu = h->nu;
  const LinDev<T> m = lin_of<T>(h);
  HIP_OK(h->s_states.reserve((size_t)n * nx * sizeof(T)));
  HIP_OK(h->s_ctrls.reserve((size_t)n * nu * sizeof(T)));
  HIP_OK(h->s_out.reserve((size_t)n * nx * sizeof(T)));
  HIP_OK(upload_converted<T>(h->s_states.p, states, (size_t)n * nx, h->stream));
  HIP_OK(upload_converted<T>(h->s_ctrls.p, ctrls, (size_t)n * nu, h->stream));
  const size_t lb = (size_t)16 * lin_xs(m.kp, (int)sizeof(T)) * sizeof(T);
  HIP_OK(allow_lds(linear_forward_kernel<T>, lb));
  hipLaunchKernelGGL(linear_forward_kernel<T>, dim3((n + 15) / 16), dim3(64 * kLinW), lb, h->stream, m,
                     (const T*)h->s_states.p, (const T*)h->s_ctrls.p, (T*)h->s_out.p, n);
  HIP_OK(hipGetLastError());
  if (jx) {
    HIP_OK(h->s_jx.reserve((size_t)n * nx * nx * sizeof(T)));
    HIP_OK(h->s_ju.reserve((size_t)n * nx * nu * sizeof(T)));
    hipLaunchKernelGGL(linear_jacobian_kernel<T>, dim3(1024), dim3(256), 0, h->stream, m, (T*)h->s_jx.p,
                       (T*)h->s_ju.p, n);
    HIP_OK(hipGetLastError());
    HIP_OK(download_converted<T>(jx, h->s_jx.p, (size_t)n * nx * nx, h->stream));
    HIP_OK(download_converted<T>(ju, h->s_ju.p, (size_t)n * nx * nu, h->stream));
  }
  HIP_OK(download_converted<T>(out, h->s_out.p, (size_t)n * nx, h->stream));
  return 0;
}

extern "C" int ampc_mlp_pred_batch(ampc_handle* h, const double* states, const double* ctrls,
                                   double* out, int n) {
  REQUIRE(h && states && ctrls && out, "ampc_mlp_pred_batch: NULL argument");
  if (h->has_sindy) return ampc_sindy_pred_batch(h, states, ctrls, out, n);
  if (h->has_lin) {
    if (n <= 0) return 0;
    HIP_OK(hipSetDevice(h->device));
    return h->precision == AMPC_F64 ? lin_pred_impl<double>(h, states, ctrls, out, nullptr, nullptr, n)
                                    : lin_pred_impl<float>(h, states, ctrls, out, nullptr, nullptr, n);
  }
  REQUIRE(h->has_mlp, "ampc_mlp_pred_batch: no model set");
  if (n <= 0) return 0;
  HIP_OK(hipSetDevice(h->device));
  return h->precision == AMPC_F64 ? pred_impl<double>(h, states, ctrls, out, nullptr, nullptr, n)
                                  : pred_impl<float>(h, states, ctrls, out, nullptr, nullptr, n);
}

extern "C" int ampc_mlp_pred_diff_batch(ampc_handle* h, const double* states, const double* ctrls,
                                        double* out, double* jx, double* ju, int n) {
  REQUIRE(h && states && ctrls && out && jx && ju, "ampc_mlp_pred_diff_batch: NULL argument");
  if (h->has_sindy) return ampc_sindy_pred_diff_batch(h, states, ctrls, out, jx, ju, n);
  if (h->has_lin) {
    if (n <= 0) return 0;
    HIP_OK(hipSetDevice(h->device));
    return h->precision == AMPC_F64 ? lin_pred_impl<double>(h, states, ctrls, out, jx, ju, n)
                                    : lin_pred_impl<float>(h, states, ctrls, out, jx, ju, n);
  }
  REQUIRE(h->has_mlp, "ampc_mlp_pred_diff_batch: no model set");
  if (n <= 0) return 0;
  HIP_OK(hipSetDevice(h->device));
  return h->precision == AMPC_F64 ? pred_impl<double>(h, states, ctrls, out, jx, ju, n)
                                  : pred_impl<float>(h, states, ctrls, out, jx, ju, n);
}

// ---------------------------------------------------------------------------------------------
// SINDy feature-library model
// ---------------------------------------------------------------------------------------------
template <typename T>
static int sindy_pred_impl(ampc_handle* h, const double* states, const double* ctrls, double* out,
                           double* jx, double* ju, int n) {
  const int nx = h->nx, nu = h->nu;
  const SindyDev<T> m = sindy_of<T>(h);
  HIP_OK(h->s_states.reserve((size_t)n * nx * sizeof(T)));
  HIP_OK(h->s_ctrls.reserve((size_t)n * nu * sizeof(T)));
  HIP_OK(h->s_out.reserve((size_t)n * nx * sizeof(T)));
  HIP_OK(upload_converted<T>(h->s_states.p, states, (size_t)n * nx, h->stream));
  HIP_OK(upload_converted<T>(h->s_ctrls.p, ctrls, (size_t)n * nu, h->stream));
  const size_t lb = (size_t)(2 * nx + nu + h->s_ntab) * 64 * sizeof(T) + sindy_stage_bytes<T>(h);
  HIP_OK(allow_lds(sindy_forward_kernel<T>, lb));
  hipLaunchKernelGGL(sindy_forward_kernel<T>, dim3((n + 63) / 64), dim3(64), lb, h->stream, m,
                     (const T*)h->s_states.p, (const T*)h->s_ctrls.p, (T*)h->s_out.p, n);
  if (jx) {
    HIP_OK(h->s_jx.reserve((size_t)n * nx * nx * sizeof(T)));
    HIP_OK(h->s_ju.reserve((size_t)n * nx * nu * sizeof(T)));
    hipLaunchKernelGGL(sindy_jacobian_kernel<T>, dim3((n + 63) / 64), dim3(64), 0, h->stream, m,
                       (const T*)h->s_states.p, (const T*)h->s_ctrls.p, (T*)h->s_jx.p, (T*)h->s_ju.p, n,
                       RowMap{n, 0, 0, nullptr});
  }
  HIP_OK(hipGetLastError());
  if (jx) {
    HIP_OK(download_converted<T>(jx, h->s_jx.p, (size_t)n * nx * nx, h->stream));
    HIP_OK(download_converted<T>(ju, h->s_ju.p, (size_t)n * nx * nu, h->stream));
  }
  HIP_OK(download_converted<T>(out, h->s_out.p, (size_t)n * nx, h->stream));
  return 0;
}

extern "C" int ampc_set_sindy(ampc_handle* h, int nx, int nu, int n_feat, const int* kind,
                              const int* arg0, const int* arg1, const double* param,
                              const double* xi, int continuous, double dt, int strict_reference,
                              int n_pairs, const int* pair_var, const int* pair_exp) {
  REQUIRE(h && kind && arg0 && arg1 && param && xi, "ampc_set_sindy: NULL argument");
  REQUIRE(nx >= 1 && nx <= 64 && nu >= 1 && nu <= kMaxNu, "ampc_set_sindy: nx in 1..64, nu in 1..16");
  REQUIRE(n_feat >= 1 && n_feat <= 4096, "ampc_set_sindy: n_feat in 1..4096");
  REQUIRE(n_pairs >= 0 && n_pairs <= 10 * 4096 && (n_pairs == 0 || (pair_var && pair_exp)),
          "ampc_set_sindy: bad monomial pair list");
  for (int j = 0; j < n_pairs; ++j)
    REQUIRE(pair_var[j] >= 0 && pair_var[j] < nx + nu && pair_exp[j] >= 1 && pair_exp[j] <= 64,
            "ampc_set_sindy: monomial pair needs a variable index and an exponent in 1..64");
  for (int k = 0; k < n_feat; ++k) {
    if (kind[k] == SF_MONO) {
      REQUIRE(arg1[k] >= 1 && arg1[k] <= 10 && arg0[k] >= 0 && arg0[k] + arg1[k] <= n_pairs,
              "ampc_set_sindy: monomial feature needs 1..10 pairs inside the pair list");
      continue;
    }
    REQUIRE(kind[k] >= 0 && kind[k] <= 5 && arg0[k] >= 0 && arg0[k] < nx + nu && arg1[k] >= 0 &&
                arg1[k] < nx + nu, "ampc_set_sindy: bad feature descriptor");
  }
  HIP_OK(hipSetDevice(h->device));
  // product form (SindyDev): distinct trig arguments and powers, two factor indices per feature
  std::vector<int> tvar, pvar, fx(n_feat, 0), fy(n_feat, 0), tslot(n_feat, -1);
  std::vector<double> tpar, ppar;
  for (int k = 0; k < n_feat; ++k) {
    if (kind[k] >= 1 && kind[k] <= 4) {
      const int var = kind[k] <= 2 ? arg0[k] : arg1[k];   // sin/cos(p v_a) vs v_a sin/cos(p v_b)
      int slot = -1;
      for (size_t j = 0; j < tvar.size(); ++j)
        if (tvar[j] == var && tpar[j] == param[k]) { slot = (int)j; break; }
      if (slot < 0) { slot = (int)tvar.size(); tvar.push_back(var); tpar.push_back(param[k]); }
      tslot[k] = slot;
    } else if (kind[k] == 5) {
      int slot = -1;
      for (size_t j = 0; j < pvar.size(); ++j)
        if (pvar[j] == arg0[k] && ppar[j] == param[k]) { slot = (int)j; break; }
      if (slot < 0) { slot = (int)pvar.size(); pvar.push_back(arg0[k]); ppar.push_back(param[k]); }
      tslot[k] = slot;
    }
  }
  std::vector<int> moff, mcnt;                                // one table entry per monomial feature
  for (int k = 0; k < n_feat; ++k)
    if (kind[k] == SF_MONO) { tslot[k] = (int)moff.size(); moff.push_back(arg0[k]); mcnt.push_back(arg1[k]); }
  int n_trig = (int)tvar.size(), n_pow = (int)pvar.size(), n_mon = (int)moff.size();
  int n_tab = 2 * n_trig + n_pow + n_mon + 1;
  if (n_tab > kSindyMaxTab) n_trig = n_pow = n_mon = n_tab = 0;      // direct evaluation instead
  if (n_tab > 0) {
    const int one = n_tab - 1;
    for (int k = 0; k < n_feat; ++k) {
      switch (kind[k]) {
        case 0: fx[k] = arg0[k]; fy[k] = one; break;
        case 1: fx[k] = -(2 * tslot[k]) - 1; fy[k] = one; break;
        case 2: fx[k] = -(2 * tslot[k] + 1) - 1; fy[k] = one; break;
        case 3: fx[k] = arg0[k]; fy[k] = 2 * tslot[k]; break;
        case 4: fx[k] = arg0[k]; fy[k] = 2 * tslot[k] + 1; break;
        case SF_MONO: fx[k] = -(2 * n_trig + n_pow + tslot[k]) - 1; fy[k] = one; break;
        default: fx[k] = -(2 * n_trig + tslot[k]) - 1; fy[k] = one; break;
      }
    }
  }
  std::vector<int> ints(9 * (size_t)n_feat + 2 * (size_t)n_pairs + 2, 0);
  std::memcpy(ints.data(), kind, n_feat * sizeof(int));
  std::memcpy(ints.data() + n_feat, arg0, n_feat * sizeof(int));
  std::memcpy(ints.data() + 2 * n_feat, arg1, n_feat * sizeof(int));
  std::memcpy(ints.data() + 3 * n_feat, fx.data(), n_feat * sizeof(int));
  std::memcpy(ints.data() + 4 * n_feat, fy.data(), n_feat * sizeof(int));
  if (n_trig > 0) std::memcpy(ints.data() + 5 * n_feat, tvar.data(), n_trig * sizeof(int));
  if (n_pow > 0) std::memcpy(ints.data() + 6 * n_feat, pvar.data(), n_pow * sizeof(int));
  if (n_mon > 0) {
    std::memcpy(ints.data() + 7 * n_feat, moff.data(), n_mon * sizeof(int));
    std::memcpy(ints.data() + 8 * n_feat, mcnt.data(), n_mon * sizeof(int));
  }
  for (int j = 0; j < n_pairs; ++j) {
    ints[9 * (size_t)n_feat + 2 * j] = pair_var[j];
    ints[9 * (size_t)n_feat + 2 * j + 1] = pair_exp[j];
  }
  HIP_OK(h->sindy_int.reserve(ints.size() * sizeof(int)));
  HIP_OK(hipMemcpy(h->sindy_int.p, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice));
  std::vector<double> flt((size_t)n_feat * (nx + 3), 0.0);
  std::memcpy(flt.data(), param, n_feat * 8);
  std::memcpy(flt.data() + n_feat, xi, (size_t)n_feat * nx * 8);
  if (n_trig > 0) std::memcpy(flt.data() + (size_t)n_feat * (nx + 1), tpar.data(), n_trig * 8);
  if (n_pow > 0) std::memcpy(flt.data() + (size_t)n_feat * (nx + 2), ppar.data(), n_pow * 8);
  h->s_ntrig = n_trig; h->s_npow = n_pow; h->s_ntab = n_tab;
  h->s_nmon = n_mon; h->s_npool = n_pairs;
  HIP_OK(h->sindy_flt.reserve(flt.size() * h->esz()));
  if (h->precision == AMPC_F64) HIP_OK(upload_converted<double>(h->sindy_flt.p, flt.data(), flt.size(), h->stream));
  else HIP_OK(upload_converted<float>(h->sindy_flt.p, flt.data(), flt.size(), h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->nx = nx; h->nu = nu; h->s_nfeat = n_feat; h->s_continuous = continuous ? 1 : 0;
  h->s_dt = dt; h->s_strict = strict_reference ? 1 : 0;
  std::memset(&h->md, 0, sizeof(h->md));
  std::memset(&h->mf, 0, sizeof(h->mf));
  h->md.nx = h->mf.nx = nx; h->md.nu = h->mf.nu = nu; h->md.kin = h->mf.kin = nx + nu;
  h->n_hidden = 0;
  h->has_sindy = true;
  h->has_mlp = false;
  h->has_lin = false;
  return 0;
}

extern "C" int ampc_sindy_pred_batch(ampc_handle* h, const double* states, const double* ctrls,
                                     double* out, int n) {
  REQUIRE(h && states && ctrls && out, "ampc_sindy_pred_batch: NULL argument");
  REQUIRE(h->has_sindy, "ampc_sindy_pred_batch: no SINDy model set");
  if (n <= 0) return 0;
  HIP_OK(hipSetDevice(h->device));
  return h->precision == AMPC_F64 ? sindy_pred_impl<double>(h, states, ctrls, out, nullptr, nullptr, n)
                                  : sindy_pred_impl<float>(h, states, ctrls, out, nullptr, nullptr, n);
}

extern "C" int ampc_sindy_pred_diff_batch(ampc_handle* h, const double* states, const double* ctrls,
                                          double* out, double* jx, double* ju, int n) {
  REQUIRE(h && states && ctrls && out && jx && ju, "ampc_sindy_pred_diff_batch: NULL argument");
  REQUIRE(h->has_sindy, "ampc_sindy_pred_diff_batch: no SINDy model set");
  if (n <= 0) return 0;
  HIP_OK(hipSetDevice(h->device));
  return h->precision == AMPC_F64 ? sindy_pred_impl<double>(h, states, ctrls, out, jx, ju, n)
                                  : sindy_pred_impl<float>(h, states, ctrls, out, jx, ju, n);
}
