// How much VALU / transcendental / LDS work hides under f32 MFMAs on gfx950?
// One workgroup per CU, W waves per SIMD; every wave loops over 8 accumulators, each MFMA followed by NV plain
// VALU ops (independent chains), NT transcendental ops and NL ds_read_b32.  Prints shader cycles per MFMA per SIMD
// (64 = the pipe's floor).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_mix.hip -o build/mfma_valu_mix && build/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NT, int NL, int PK>
__global__ __launch_bounds__(512, 2) void k(float *out, long long *cyc, int iters, float seed) {
  __shared__ float lds[4096];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
  __syncthreads();
  float a = seed * (threadIdx.x % 7 + 1) * 0.37f, b = seed * (threadIdx.x % 5 + 1) * 0.11f;
  float x[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 y[4];
  for (int i = 0; i < 8; ++i) x[i] = seed + i;
  for (int i = 0; i < 4; ++i) y[i] = f2{seed + i, seed - i};
  const float c = 1.0001f;
  const f2 c2 = {1.0001f, 0.9999f};
  float l = 0.f;
  const float *lp = lds + (threadIdx.x & 63);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (PK) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[v % 4]) : "v"(c2));
        else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[v % 8]) : "v"(c));
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(t + 4) % 8]));
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        float tmp;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tmp) : "v"((int)(size_t)lp * 0 + (int)((threadIdx.x & 63) * 4)), "n"(256 * 0));
        l += 0.f * 0;
        (void)tmp;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s = l;
  for (int i = 0; i < 8; ++i) {
    s += x[i];
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  }
  for (int i = 0; i < 4; ++i) s += y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NT, int NL, int PK>
void run(int threads) {
  const int iters = 4000, grid = 256;
  float *out;
  long long *cyc, h[256];
  hipMalloc(&out, grid * 512 * sizeof(float));
  hipMalloc(&cyc, grid * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NT, NL, PK>), dim3(grid), dim3(threads), 0, 0, out, cyc, 50, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NT, NL, PK>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < grid; ++i) mean += h[i];
  mean /= grid;
  const int wps = threads / 256;  // waves per SIMD
  const double mf_per_simd = (double)iters * 8 * wps;
  printf("waves/SIMD=%d  NV=%2d%s NT=%d NL=%d : %7.1f cyc/MFMA/SIMD (s_memtime)  %6.3f ms  -> %.1f TFLOP/s MFMA\n", wps, NV,
         PK ? "(pk)" : "    ", NT, NL, mean / mf_per_simd, ms, (double)grid * 4 * mf_per_simd * 4096.0 / ms / 1e9);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 0, 0, 0>(512);
  run<4, 0, 0, 0>(512);
  run<6, 0, 0, 0>(512);
  run<8, 0, 0, 0>(512);
  run<10, 0, 0, 0>(512);
  run<12, 0, 0, 0>(512);
  run<16, 0, 0, 0>(512);
  run<6, 1, 0, 0>(512);
  run<6, 2, 0, 0>(512);
  run<0, 2, 0, 0>(512);
  run<0, 4, 0, 0>(512);
  run<4, 0, 0, 1>(512);
  run<6, 0, 0, 1>(512);
  run<8, 0, 0, 1>(512);
  run<6, 1, 2, 0>(512);
  run<6, 1, 4, 0>(512);
  run<0, 0, 4, 0>(512);
  printf("-- one wave per SIMD --\n");
  run<0, 0, 0, 0>(256);
  run<4, 0, 0, 0>(256);
  run<8, 0, 0, 0>(256);
  run<12, 0, 0, 0>(256);
  run<6, 2, 0, 0>(256);
  return 0;
}
