// Probe: how does gfx950 range-check raw buffer loads whose 32-bit voffset is "negative"?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/bufload_probe.hip -o build/bufload_probe && build/bufload_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float *p, float *o, int bytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
  int neg4 = -4 + (int)threadIdx.x * 0;
  // (a) voffset = -4, inst_offset = 4  -> element 0 or 0.0?
  float a;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:4\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(neg4), "s"(r));
  // (b) dwordx4 at voffset = -4: lanes 1..3 = elements 0..2 ?
  v4f b;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(neg4), "s"(r));
  // (c) voffset = bytes - 4 (last element), inst_offset = 4 -> out of range -> 0
  float c;
  int last = bytes - 4;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:4\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(last), "s"(r));
  // (d) soffset: voffset = 0, soffset = bytes (past the end by the scalar offset) -> checked or not?
  float d;
  int zero = 0;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(zero), "s"(r), "s"(bytes));
  // (e) dwordx4 straddling the end: voffset = bytes - 8
  v4f e;
  int l2 = bytes - 8;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(l2), "s"(r));
  if (threadIdx.x == 0) {
    o[0] = a; o[1] = b.x; o[2] = b.y; o[3] = b.z; o[4] = b.w; o[5] = c; o[6] = d; o[7] = e.x; o[8] = e.y; o[9] = e.z; o[10] = e.w;
  }
}
int main() {
  const int n = 1024;  // probe a 64-float window in the middle of a larger allocation so nothing can fault
  float *h = (float *)malloc(n * 4), *d, *o, ho[16];
  for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
  hipMalloc(&d, n * 4); hipMalloc(&o, 64);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d + 512, o, 64 * 4);
  hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
  printf("window = elements 1512..1575\n");
  printf("(a) voff=-4 imm=4      : %g   (1512 = wraps in 32 bits, 0 = out of range)\n", ho[0]);
  printf("(b) x4 at voff=-4      : %g %g %g %g\n", ho[1], ho[2], ho[3], ho[4]);
  printf("(c) voff=last imm=4    : %g   (0 expected)\n", ho[5]);
  printf("(d) voff=0 soff=bytes  : %g   (0 = soffset range-checked, 1576 = not checked)\n", ho[6]);
  printf("(e) x4 at voff=bytes-8 : %g %g %g %g\n", ho[7], ho[8], ho[9], ho[10]);
  return 0;
}
