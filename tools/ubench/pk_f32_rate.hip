// Is packed fp32 math (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) full rate on gfx950, i.e. do the F(4x4) producers, whose
// lanes transform TWO channels in lockstep, halve their VALU instruction count for free?  One workgroup per CU, 2 waves per SIMD,
// 16 independent chains per wave; prints shader cycles per instruction and wave (4 = one wave64 instruction per 4 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_f32_rate.hip -o tools/ubench/pk_f32_rate.out && tools/ubench/pk_f32_rate.out
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int MF>  // MODE 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_add_f32, 3 v_pk_mul_f32, 4 v_add_f32; MF: one f16 MFMA per 16 ops
__global__ __launch_bounds__(512, 1) void k(float *out, long long *cyc, int iters, float seed) {
  float x[16];
  f2 y[16];
  for (int i = 0; i < 16; ++i) {
    x[i] = seed + i + threadIdx.x;
    y[i] = f2{seed + i, seed - i + threadIdx.x};
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) ah[i] = (_Float16)(seed + i), bh[i] = (_Float16)(seed - i);
  const float c = 1.0001f, d = 0.5f;
  const f2 c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (MF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(c2), "v"(d2));
        if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(c2));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(c2));
        if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i] + y[i][0] + y[i][1];
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int MF>
void run(const char *name) {
  float *out;
  long long *cyc;
  const int blocks = 256, iters = 2000;
  hipMalloc(&out, blocks * 512 * sizeof(float));
  hipMalloc(&cyc, blocks * sizeof(long long));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, MF>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += h[i];
  avg /= blocks;
  // readcyclecounter ticks at 100 MHz on this part? report both raw ticks per op and, via wall clock, ns per op
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, MF>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops_per_wave = (double)iters * 64;
  printf("%-28s %8.2f counter ticks / op / wave   %7.3f ns per op per SIMD (2 waves)   kernel %.3f ms\n", name, avg / ops_per_wave,
         ms * 1e6 / (ops_per_wave * 2), ms);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 0>("v_fma_f32");
  run<4, 0>("v_add_f32");
  run<1, 0>("v_pk_fma_f32");
  run<2, 0>("v_pk_add_f32");
  run<3, 0>("v_pk_mul_f32");
  run<0, 1>("v_fma_f32 + f16 MFMA / 16");
  run<1, 1>("v_pk_fma_f32 + f16 MFMA / 16");
  return 0;
}
