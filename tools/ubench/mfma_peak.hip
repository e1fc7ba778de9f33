// Practical f32-MFMA peak on this box: 4 (or 8) independent 32x32x2 accumulators per wave, no memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed * (threadIdx.x % 7 + 1) * 0.37f, b = seed * (threadIdx.x % 5 + 1) * 0.11f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a = -a;
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, float seed) {
  const int iters = 20000, grid = 256 * wgs_per_cu;
  float *out;
  hipMalloc(&out, grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, seed);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, seed);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * NACC * 4096.0;
  printf("NACC=%d wgs/CU=%d seed=%g: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, seed, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(1, 1.0f); run<4>(2, 1.0f); run<4>(3, 1.0f); run<8>(1, 1.0f); run<4>(1, 0.0f); run<4>(2, 0.0f);
  return 0;
}
