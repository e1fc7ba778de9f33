// conv_direct.hip -- generic fp32 direct convolution (any H, W, Cin, Cout) on the vector ALU.
//
// Same fused semantics as conv_mfma.hip (affine+SiLU prologue, virtual concat, upsample /
// stride-2 indexing, bias + chan_add + residual epilogue).  It serves the layers the MFMA
// tiling does not cover -- conv_in (Cin = 1 / 3), conv_out (Cout = 1 / 3), odd extents such as
// 28x28 -- and is the on-device cross-check for the MFMA kernel in tests.
// Reference call site: /root/reference/src/trainers/reconstruct.py:151-153.
#include "common.h"

namespace ddpm {

template <int COB>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ddpm_conv_desc a) {
  const int HWo = a.Ho * a.Wo, HWi = a.Hi * a.Wi;
  const int Cin = a.C1 + a.C2;
  const int T = a.ksize * a.ksize;
  const int pad = a.ksize == 3 ? 1 : 0;
  const int s = a.mode == DDPM_CONV_STRIDE2 ? 2 : 1;
  const bool up = a.mode == DDPM_CONV_UPSAMPLE2;
  const int Hv = up ? a.Ho : a.Hi, Wv = up ? a.Wo : a.Wi;

  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.z;
  const int co0 = blockIdx.y * COB;
  if (p >= HWo) return;
  const int ho = p / a.Wo, wo = p - ho * a.Wo;

  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = 0.f;

  for (int ci = 0; ci < Cin; ++ci) {
    const float *plane = (ci < a.C1) ? a.in1 + ((size_t)n * a.C1 + ci) * HWi
                                     : a.in2 + ((size_t)n * a.C2 + (ci - a.C1)) * HWi;
    float sc = 1.f, sh = 0.f;
    if (a.gscale) {
      sc = a.gscale[(size_t)n * Cin + ci];
      sh = a.gshift[(size_t)n * Cin + ci];
    }
    for (int kh = 0; kh < a.ksize; ++kh) {
      for (int kw = 0; kw < a.ksize; ++kw) {
        const int hv = ho * s + kh - pad, wv = wo * s + kw - pad;
        float v = 0.f;
        if (hv >= 0 && hv < Hv && wv >= 0 && wv < Wv) {
          v = up ? plane[(hv >> 1) * a.Wi + (wv >> 1)] : plane[hv * a.Wi + wv];
          if (a.gscale) v = v * sc + sh;
          if (a.act == DDPM_ACT_SILU) v = silu_f(v);
        }
        const int t = kh * a.ksize + kw;
#pragma unroll
        for (int j = 0; j < COB; ++j) {
          const int co = (co0 + j < a.Cout) ? co0 + j : a.Cout - 1;
          acc[j] = fmaf(v, a.w_raw[((size_t)co * Cin + ci) * T + t], acc[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co = co0 + j;
    if (co < a.Cout) {
      const size_t idx = ((size_t)n * a.Cout + co) * HWo + p;
      float v = acc[j];
      if (a.bias) v += a.bias[co];
      if (a.chan_add) v += a.chan_add[(size_t)n * a.chan_add_stride + co];
      if (a.residual) v += a.residual[idx];
      a.out[idx] = v;
    }
  }
}

int launch_conv_direct(const ddpm_conv_desc &d, hipStream_t s) {
  DDPM_CHECK_ARG(d.w_raw != nullptr, "conv_direct: w_raw is NULL");
  DDPM_CHECK_ARG(d.ksize == 1 || d.ksize == 3, "conv_direct: ksize must be 1 or 3");
  DDPM_CHECK_ARG(d.B <= 65535, "conv_direct: batch > 65535");
  const int HWo = d.Ho * d.Wo;
  const double cin = d.C1 + d.C2, taps = d.ksize * d.ksize;
  ProfScope prof(s, "conv_direct", 2.0 * d.B * HWo * d.Cout * cin * taps,
                 4.0 * ((double)d.B * cin * d.Hi * d.Wi + (double)d.B * HWo * d.Cout * (d.residual ? 2 : 1) +
                        d.Cout * cin * taps));
  if (d.Cout <= 4) {
    dim3 grid((HWo + 255) / 256, d.Cout, d.B);
    hipLaunchKernelGGL(conv_direct_kernel<1>, grid, dim3(256), 0, s, d);
  } else {
    dim3 grid((HWo + 255) / 256, (d.Cout + 7) / 8, d.B);
    hipLaunchKernelGGL(conv_direct_kernel<8>, grid, dim3(256), 0, s, d);
  }
  DDPM_CHECK_LAUNCH();
  return 0;
}

int conv_dispatch(const ddpm_conv_desc &d, hipStream_t s) {
  DDPM_CHECK_ARG(d.in1 && d.out && d.B > 0 && d.Cout > 0 && d.C1 > 0, "conv: null tensor or empty shape");
  DDPM_CHECK_ARG(d.C2 == 0 || d.in2, "conv: C2 > 0 but in2 is NULL");
  DDPM_CHECK_ARG((d.gscale == nullptr) == (d.gshift == nullptr), "conv: gscale/gshift must come together");
  if (d.mode == DDPM_CONV_NORMAL)
    DDPM_CHECK_ARG(d.Hi == d.Ho && d.Wi == d.Wo, "conv: normal mode needs Hi == Ho, Wi == Wo");
  if (d.mode == DDPM_CONV_UPSAMPLE2)
    DDPM_CHECK_ARG(d.Ho == 2 * d.Hi && d.Wo == 2 * d.Wi && d.ksize == 3, "conv: upsample needs Ho == 2 Hi, k == 3");
  if (d.mode == DDPM_CONV_STRIDE2)
    DDPM_CHECK_ARG(d.Ho == (d.Hi + 1) / 2 && d.Wo == (d.Wi + 1) / 2 && d.ksize == 3,
                   "conv: stride-2 needs Ho == ceil(Hi / 2), k == 3");
  if (conv_mfma_supported(d)) return launch_conv_mfma(d, s);
  return launch_conv_direct(d, s);
}

}  // namespace ddpm
