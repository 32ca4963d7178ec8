// Probe the lane layout and issue rate of v_mfma_f64_4x4x4_4b_f64 on this GPU (one-hot inputs).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma44_probe.cpp -o variants/mfma44_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

__global__ void rate(double* out, int iters) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

int main() {
  double *da, *db, *dd;
  hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dd, 64 * 8);
  // For every (la, lb): which output lanes see a[la]*b[lb]?
  std::vector<double> ha(64), hb(64), hd(64);
  printf("A lane -> (pairs with B lanes) -> D lanes\n");
  for (int la = 0; la < 64; ++la) {
    for (int lb = 0; lb < 64; ++lb) {
      for (int i = 0; i < 64; ++i) { ha[i] = 0; hb[i] = 0; }
      ha[la] = 1; hb[lb] = 1;
      hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice);
      hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
      probe<<<1, 64>>>(da, db, dd);
      hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; ++i)
        if (hd[i] != 0) printf("a%d b%d -> d%d\n", la, lb, i);
    }
  }
  // issue rate: 8 independent accumulators, 2 waves per SIMD on every CU
  double* out;
  const int blocks = 256 * 2, thr = 256, iters = 20000;
  hipMalloc(&out, (size_t)blocks * thr * 8);
  rate<<<blocks, thr>>>(out, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate<<<blocks, thr>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * (thr / 64) * iters * 8;
  printf("4x4x4 f64: %.3f ms, %.1f TFLOP/s (512 flop each), %.1f cycles per MFMA per SIMD at 2.4 GHz\n", ms,
         mfmas * 512 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (mfmas / (256 * 4)));
  return 0;
}
