// Issue rate of v_mfma_f64_4x4x4_4b_f64 with ONE wave per SIMD (the line-search kernel's occupancy)
// against two, and with the operand patterns that kernel uses (A shared by NT MFMAs, B by RB).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma44_rate1.cpp -o variants/mfma44_rate1
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RB, int NT>
__global__ void rate(double* out, const double* in, int iters) {
  double a[RB], b[NT], c[RB][NT];
  for (int r = 0; r < RB; ++r) a[r] = in[threadIdx.x + 64 * r];
  for (int n = 0; n < NT; ++n) b[n] = in[threadIdx.x + 64 * (RB + n)];
  for (int r = 0; r < RB; ++r) for (int n = 0; n < NT; ++n) c[r][n] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < RB; ++r) c[r][n] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[r], b[n], c[r][n], 0, 0, 0);
  }
  double s = 0;
  for (int r = 0; r < RB; ++r) for (int n = 0; n < NT; ++n) s += c[r][n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int RB, int NT> void run(int waves_per_simd, double* out, double* in) {
  const int blocks = 256 * waves_per_simd, thr = 256, iters = 20000;
  rate<RB, NT><<<blocks, thr>>>(out, in, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate<RB, NT><<<blocks, thr>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)waves_per_simd * iters * RB * NT;
  printf("RB %d NT %d, %d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s, %.2f cycles per MFMA per SIMD at 2.4 GHz\n", RB, NT,
         waves_per_simd, ms, per_simd * 1024 * 512 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / per_simd);
}

int main() {
  double *out, *in;
  hipMalloc(&out, (size_t)512 * 256 * 8); hipMalloc(&in, 64 * 16 * 8);
  hipMemset(in, 0, 64 * 16 * 8);
  run<1, 4>(1, out, in); run<1, 4>(2, out, in);
  run<1, 8>(1, out, in); run<1, 8>(2, out, in);
  run<3, 4>(1, out, in); run<3, 4>(2, out, in);
  run<2, 4>(1, out, in);
  return 0;
}
