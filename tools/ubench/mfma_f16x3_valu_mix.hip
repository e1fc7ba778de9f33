// What would the Winograd chunk loops gain from the split-f16 MFMA (conv1x1_dma.hip's fp32-from-three-f16-MFMAs)?
// One "unit" is the work of EIGHT v_mfma_f32_32x32x2_f32 (a 32x32 tile over 16 channels, 512 SIMD cycles at the f32
// rate): either those eight MFMAs, or three v_mfma_f32_32x32x16_f16 (96 cycles), each unit followed by NV plain
// VALU ops (independent chains) -- the F(4x4) kernel carries ~16.7 non-MFMA instructions per f32 MFMA = ~134 per
// unit, and splitting a transformed operand into hi / lo halves adds ~3.5 per value.  Prints shader cycles per unit
// per SIMD, i.e. whether the f16 MFMAs hide under another wave's VALU stream and what the VALU-only floor is.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16x3_valu_mix.hip -o build/mfma_f16x3_valu_mix && build/mfma_f16x3_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool F16>
__global__ __launch_bounds__(512, 1) void k(float *out, long long *cyc, int iters, float seed) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed * (threadIdx.x % 7 + 1) * 0.37f, b = seed * (threadIdx.x % 5 + 1) * 0.11f;
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) {
    ah[i] = (_Float16)(a + i);
    bh[i] = (_Float16)(b - i);
  }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + i;
  const float c = 1.0001f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // one unit per accumulator
      if (F16) {
#pragma unroll
        for (int m = 0; m < 3; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(ah), "v"(bh));
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[v % 8]) : "v"(c));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) {
    s += x[i];
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, bool F16>
void run(int threads) {
  const int iters = 500, grid = 256;
  float *out;
  long long *cyc, h[256];
  hipMalloc(&out, grid * 512 * sizeof(float));
  hipMalloc(&cyc, grid * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, F16>), dim3(grid), dim3(threads), 0, 0, out, cyc, 20, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, F16>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < grid; ++i) mean += h[i];
  mean /= grid;
  const int wps = threads / 256;  // waves per SIMD
  const double units_per_simd = (double)iters * 8 * wps;
  // a unit is 2 * 32 * 32 * 16 fp32-equivalent FLOP
  printf("%s waves/SIMD=%d NV=%3d : %7.1f cycles/unit/SIMD  %6.3f ms -> %6.1f fp32-equivalent TFLOP/s\n",
         F16 ? "3 x f16 32x32x16" : "8 x f32 32x32x2 ", wps, NV, mean / units_per_simd, ms,
         (double)grid * 4 * units_per_simd * 32768.0 / ms / 1e9);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, false>(512);
  run<64, false>(512);
  run<128, false>(512);
  run<160, false>(512);
  run<0, true>(512);
  run<32, true>(512);
  run<64, true>(512);
  run<96, true>(512);
  run<128, true>(512);
  run<160, true>(512);
  run<192, true>(512);
  printf("-- one wave per SIMD --\n");
  run<128, false>(256);
  run<0, true>(256);
  run<64, true>(256);
  run<128, true>(256);
  run<160, true>(256);
  return 0;
}
