// Streaming-read patterns over one 537 MB activation ([B = 1024][C = 128][32 x 32] fp32), as the convolution kernels issue them:
// does the chip stream faster when a workgroup's concurrent requests are ONE contiguous block instead of one piece per channel
// plane (4 KB apart)?  Every mode reads every byte exactly once, one 256-byte row per wave-load, `PF` loads in flight per wave.
//   mode 0  channel-planar, conv_out / F(4x4) style: workgroup = (image, quarter of the rows), wave w walks channels w, w + 4, ...;
//           per channel it reads its 8 rows x 128 B = 1 KB (4 loads of 256 B): four waves -> four 1 KB pieces, 4 KB apart
//   mode 1  the same bytes per workgroup, but laid out channel-blocked per quarter: [image][quarter][channel][1 KB]: the four waves'
//           pieces are one contiguous 4 KB, consecutive channels follow each other
//   mode 2  whole planes: workgroup = image, wave w reads channel w, w + 4, ...: 4 KB contiguous per wave (16 loads)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/plane_stream.hip -o build/plane_stream && build/plane_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE, int PF>
__global__ __launch_bounds__(256) void stream(const float *__restrict__ in, float *__restrict__ out, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, q = blockIdx.x;  // image, quarter (modes 0 / 1)
  float acc = 0.f;
  if (MODE == 2) {
    const float *img = in + (size_t)n * C * 1024;
    for (int c = wave; c < C; c += 4 * PF) {
      float v[PF][16];
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[p][j] = (c + 4 * p < C) ? img[(size_t)(c + 4 * p) * 1024 + 64 * j + lane] : 0.f;
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[p][j];
    }
  } else {
    // piece (channel c) of this workgroup: 256 floats
    auto piece = [&](int c) {
      return MODE == 0 ? in + ((size_t)n * C + c) * 1024 + q * 256 : in + (((size_t)n * 4 + q) * C + c) * 256;
    };
    for (int c = wave; c < C; c += 4 * PF) {
      float v[PF][4];
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[p][j] = (c + 4 * p < C) ? piece(c + 4 * p)[64 * j + lane] : 0.f;
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[p][j];
    }
  }
  if (acc == 12345.678f) out[0] = acc;  // (keeps the loads)
}

template <int MODE, int PF>
static void run(const float *d, float *o, int B, int C, const char *name) {
  const dim3 grid(MODE == 2 ? 1 : 4, B);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<MODE, PF>), grid, dim3(256), 0, 0, d, o, C);
  hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((stream<MODE, PF>), grid, dim3(256), 0, 0, d, o, C);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)B * C * 4096;
  printf("%-58s PF=%d  %8.1f us  %6.2f TB/s\n", name, PF, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
}

int main() {
  const int B = 1024, C = 128;
  const size_t n = (size_t)B * C * 1024;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 64);
  hipMemset(d, 0, n * 4);
  run<0, 1>(d, o, B, C, "channel-planar, 1 KB pieces 4 KB apart (as the conv kernels)");
  run<0, 2>(d, o, B, C, "channel-planar, 1 KB pieces 4 KB apart (as the conv kernels)");
  run<0, 4>(d, o, B, C, "channel-planar, 1 KB pieces 4 KB apart (as the conv kernels)");
  run<1, 1>(d, o, B, C, "channel-blocked per quarter: the workgroup's pieces contiguous");
  run<1, 2>(d, o, B, C, "channel-blocked per quarter: the workgroup's pieces contiguous");
  run<1, 4>(d, o, B, C, "channel-blocked per quarter: the workgroup's pieces contiguous");
  run<2, 1>(d, o, B, C, "whole 4 KB planes per wave (workgroup = image)");
  run<2, 2>(d, o, B, C, "whole 4 KB planes per wave (workgroup = image)");
  return 0;
}
