// mfma_peak.cpp -- measure the dense MFMA issue ceiling of v_mfma_f64_16x16x4_f64 and
// v_mfma_f32_16x16x4_f32 on the GPU at hand (the roofline "peak" bench.py prices against).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.cpp -o variants/mfma_peak && variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ void k64(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k32(float* out, int iters) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K, typename T> double run(K kern, T* buf, int blocks, int iters, int nacc, double flop_per) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * nacc * flop_per;
  return flops / (ms * 1e-3) / 1e12;
}
int main() {
  void* buf; hipMalloc(&buf, 8 * 256 * 4096);
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    printf("blocks=%d (waves/SIMD=%d)  f64 16x16x4: 4acc %.1f TF  8acc %.1f TF | f32 16x16x4: 4acc %.1f TF 8acc %.1f TF\n",
           blocks, blocks / 256,
           run(k64<4>, (double*)buf, blocks, iters, 4, 2048.0), run(k64<8>, (double*)buf, blocks, iters, 8, 2048.0),
           run(k32<4>, (float*)buf, blocks, iters, 4, 2048.0), run(k32<8>, (float*)buf, blocks, iters, 8, 2048.0));
  }
  return 0;
}
