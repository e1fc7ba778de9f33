// linear_skinny.hip -- Linear layers over a few hundred rows (timestep MLP and the fused time_emb_proj GEMM of
// DiffusionModelUNet.forward; reference call site /root/reference/src/trainers/reconstruct.py:151-153).
//
// out[m][n] = sum_k act(x[m][k]) w[n][k] + bias[n],  m < B <= 1024, w in torch layout [N][K].
//
// These GEMMs are a few hundred MFLOP: on the tiled conv kernel they were pure latency (one workgroup per 64 x 128
// output tile walking K serially through LDS, 32 chunks and barriers: 60 us for [256 x 512] x [512 x 512]).  Here a
// workgroup owns a 32 x 32 output tile and its four waves split K; operands go global -> registers as float4 along
// k (both matrices are k-contiguous), straight into v_mfma_f32_32x32x2_f32.  Lane (i, h) holds k = 8 j + 4 h + c
// for the c-th MFMA of step j: every k is visited once, in an order that is fixed by the shape alone.  The four
// partial tiles are added through LDS in wave order -> bit-reproducible.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace ddpm {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float v4fs __attribute__((ext_vector_type(4)));

constexpr int kSkinnyMaxRows = 1024;

bool linear_skinny_supported(const ddpm_conv_desc &d) {
  static const bool enabled = !(getenv("DDPM_LINEAR_SKINNY") && atoi(getenv("DDPM_LINEAR_SKINNY")) == 0);
  if (!enabled || d.ksize != 1 || d.mode != DDPM_CONV_NORMAL || !d.w_raw || d.force_direct) return false;
  if (d.Hi != 1 || d.Wi != 1 || d.Ho != 1 || d.Wo != 1 || d.Di > 1 || d.Do > 1) return false;
  if (d.C2 || d.gscale || d.residual || d.chan_add || d.out_act != DDPM_ACT_NONE) return false;
  if (d.act != DDPM_ACT_NONE && d.act != DDPM_ACT_SILU) return false;
  if ((reinterpret_cast<uintptr_t>(d.in1) | reinterpret_cast<uintptr_t>(d.w_raw)) & 15) return false;  // float4 loads
  return d.B <= kSkinnyMaxRows && d.C1 % 32 == 0 && d.Cout % 32 == 0;
}

__global__ __launch_bounds__(256) void linear_skinny_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            int B, int K, int N, int silu) {
  __shared__ float part[4][16][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int Kw = K >> 2, k0 = wave * Kw;  // this wave's quarter of K (a multiple of 8)
  const bool mok = m0 + i < B;
  const float *xr = x + (size_t)(mok ? m0 + i : 0) * K + k0 + 4 * h;
  const float *wr = w + (size_t)(n0 + i) * K + k0 + 4 * h;
  f32x16s acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kb = 0; kb < Kw; kb += 64) {
    v4fs a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (kb + 8 * j < Kw) {  // uniform
        a[j] = *reinterpret_cast<const v4fs *>(xr + kb + 8 * j);
        b[j] = *reinterpret_cast<const v4fs *>(wr + kb + 8 * j);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (kb + 8 * j < Kw) {
        v4fs av = a[j];
        if (silu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) av[c] = silu_fast(av[c]);
        }
        if (!mok) av = v4fs{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], b[j][c], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
  // wave v finishes accumulator registers 4 v .. 4 v + 3: D register r of lane (i, h) is row 8 (r / 4) + 4 h + r % 4
  const float bn = bias ? bias[n0 + i] : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * wave + q;
    const float v = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane] + bn;
    const int m = m0 + 8 * (r >> 2) + 4 * h + (r & 3);
    if (m < B) out[(size_t)m * N + n0 + i] = v;
  }
}

int launch_linear_skinny(const ddpm_conv_desc &d, hipStream_t s) {
  if (!linear_skinny_supported(d)) {
    set_error("linear_skinny: unsupported shape");
    return DDPM_EINVAL;
  }
  ProfScope prof(s, "linear_skinny", 2.0 * d.B * d.C1 * d.Cout, 4.0 * ((double)d.B * (d.C1 + d.Cout) + (double)d.C1 * d.Cout));
  dim3 grid((d.B + 31) / 32, d.Cout / 32);
  hipLaunchKernelGGL(linear_skinny_kernel, grid, dim3(256), 0, s, d.in1, d.w_raw, d.bias, d.out, d.B, d.C1, d.Cout,
                     d.act == DDPM_ACT_SILU ? 1 : 0);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
