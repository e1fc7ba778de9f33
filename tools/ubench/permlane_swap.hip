// What v_permlane16_swap / v_permlane32_swap return through the clang builtins (gfx950): prints lane -> (r[0], r[1]) for
// vdst = lane id, src0 = 100 + lane id.   hipcc --offload-arch=gfx950 -O3 permlane_swap.hip -o permlane_swap.out
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
  const unsigned l = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
  o[4 * l + 0] = a[0];
  o[4 * l + 1] = a[1];
  o[4 * l + 2] = b[0];
  o[4 * l + 3] = b[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 8) printf("lane %2d: p16 (%3u, %3u)  p32 (%3u, %3u)\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
