// api_model.cpp -- staging an MLP into a handle: ampc_set_mlp (host arrays) and ampc_set_mlp_dev (device
// arrays, e.g. the parameters a PyTorch-ROCm fit has just produced).
//
// The packed model buffer (layouts: mlp_tile.hpp, MlpDev) is a pure GATHER of the folded weights: which
// folded entry lands in which packed slot depends on the model's SHAPE only.  build_pack_map() writes
// that gather down once per (device, precision, shape) as an int32 index table; the host path applies it
// on the host, the device path applies it in one kernel after folding the normalisers on the device
// with the host's arithmetic (no contraction, same summation order), so both give the same bytes.
#include "host_common.hpp"

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
