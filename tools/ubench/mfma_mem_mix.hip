// Marginal cost of memory-side instructions issued next to f32 MFMAs on gfx950 (2 waves per SIMD, 1 WG per CU).
// Each wave: 8 accumulators; after every MFMA one "filler group" chosen by the template kind.  Companion of
// mfma_valu_mix.hip.   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mem_mix.hip -o build/mfma_mem_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { NONE, DSW128, DSW32, DSW2ST, DSR32, DSR2, GLD128, BLD32, GLD128_DSW128, VALU4 };

template <int KIND, int EVERY>
__global__ __launch_bounds__(512, 2) void k(float *out, const float *src, int iters, float seed) {
  extern __shared__ float lds[];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed * (threadIdx.x % 7 + 1) * 0.37f, b = seed * (threadIdx.x % 5 + 1) * 0.11f;
  v4f w = {seed, seed, seed, seed};
  v4f g = {0, 0, 0, 0};
  float x[4] = {seed, seed + 1, seed + 2, seed + 3};
  float r0 = 0.f, r1 = 0.f;
  const int laddr = threadIdx.x * 16;
  const float *gp = src + (blockIdx.x % 64) * 4096 + threadIdx.x * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 20, 0x00020000);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      if (i % EVERY == 0) {
        if (KIND == DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"(laddr), "v"(w) : "memory");
        if (KIND == DSW32) asm volatile("ds_write_b32 %0, %1" ::"v"(laddr), "v"(a) : "memory");
        if (KIND == DSW2ST) asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:8" ::"v"(laddr / 4), "v"(a), "v"(b) : "memory");
        if (KIND == DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(laddr / 4) : "memory");
        if (KIND == DSR2) asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:8" : "=v"(*(float2 *)&x[0]) : "v"(laddr / 4) : "memory");
        if (KIND == GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g) : "v"(gp) : "memory");
        if (KIND == BLD32) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(r1) : "v"(laddr), "s"(rs) : "memory");
        if (KIND == GLD128_DSW128) {
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g) : "v"(gp) : "memory");
          asm volatile("ds_write_b128 %0, %1" ::"v"(laddr), "v"(w) : "memory");
        }
        if (KIND == VALU4) {
#pragma unroll
          for (int v = 0; v < 4; ++v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[v]) : "v"(1.0001f));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s = r0 + r1 + g.x + g.y + g.z + g.w + x[0] + x[1] + x[2] + x[3];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int EVERY>
double run(const char *name, double base) {
  const int iters = 4000, grid = 256;
  float *out, *src;
  hipMalloc(&out, grid * 512 * sizeof(float));
  hipMalloc(&src, 4 << 20);
  hipMemset(src, 0, 4 << 20);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<KIND, EVERY>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, EVERY>), dim3(grid), dim3(512), 32768, 0, out, src, 50, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, EVERY>), dim3(grid), dim3(512), 32768, 0, out, src, iters, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 8 * 2;  // MFMAs per SIMD
  const double tf = (double)grid * 4 * mf * 4096.0 / ms / 1e9;
  // marginal cost per filler instruction per wave, in units of "64-cycle MFMA slots x 64": assume base = 64 cyc
  const double cyc_per_mfma = base > 0 ? 64.0 * ms / base : 64.0;
  const int per_group = KIND == GLD128_DSW128 ? 2 : KIND == VALU4 ? 4 : 1;
  printf("%-16s every %d MFMA: %6.3f ms  %6.1f TF  -> %5.1f cyc/MFMA", name, EVERY, ms, tf, cyc_per_mfma);
  if (base > 0 && KIND != NONE) printf("  (+%.1f cyc per filler instruction per wave)", (cyc_per_mfma - 64.0) * EVERY / per_group);
  printf("\n");
  hipFree(out);
  hipFree(src);
  return ms;
}

int main() {
  const double base = run<NONE, 1>("none", 0);
  run<NONE, 1>("none", base);
  run<VALU4, 1>("4 x v_mul", base);
  run<DSR32, 1>("ds_read_b32", base);
  run<DSR2, 1>("ds_read2st64", base);
  run<DSW32, 1>("ds_write_b32", base);
  run<DSW32, 2>("ds_write_b32", base);
  run<DSW2ST, 1>("ds_write2st64", base);
  run<DSW2ST, 2>("ds_write2st64", base);
  run<DSW128, 1>("ds_write_b128", base);
  run<DSW128, 2>("ds_write_b128", base);
  run<DSW128, 4>("ds_write_b128", base);
  run<GLD128, 1>("global_load_x4", base);
  run<GLD128, 2>("global_load_x4", base);
  run<GLD128, 4>("global_load_x4", base);
  run<BLD32, 1>("buffer_load_dw", base);
  run<BLD32, 2>("buffer_load_dw", base);
  run<GLD128_DSW128, 4>("gld+dsw128", base);
  run<GLD128_DSW128, 8>("gld+dsw128", base);
  return 0;
}
