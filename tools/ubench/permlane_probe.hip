// permlane_probe.hip -- which halves does v_permlane32_swap_b32 exchange?  (development probe for attention.hip)
//   hipcc --offload-arch=gfx950 tools/ubench/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float *out) {
  const int l = threadIdx.x;
  float a = 100.f + l, b = 200.f + l;
  // (the builtin's second result is mis-lowered by this hipcc -- both stores read the first register -- hence asm)
  asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "+v"(b));
  out[l] = a;
  out[64 + l] = b;
}
int main() {
  float *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0=%g lane31=%g lane32=%g lane63=%g\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0=%g lane31=%g lane32=%g lane63=%g\n", h[64], h[95], h[96], h[127]);
  return 0;
}
