// lpips.hip -- the LPIPS-AlexNet similarity score (second CSV column) on the device.
//
// Replaces lpips.LPIPS(net='alex', version='0.1', lpips=True, spatial=False)(in0, in1, normalize=True) as called by
// PerceptualLoss.forward (/root/reference/src/losses/perceptual_loss.py:105-186; call site
// /root/reference/src/trainers/reconstruct.py:172-187).  Three kernels:
//   lpips_conv_kernel    generic NCHW direct convolution + bias + ReLU for the two AlexNet layers that have no MFMA
//                        tiling here (11x11 stride 4 over 1 or 3 input channels, 5x5 over 64), with the "2x - 1" and
//                        ScalingLayer affine folded into the first layer's input and the 1 -> 3 channel broadcast
//                        done by indexing.  (The three 3x3 layers run on conv_mfma.hip with a ReLU epilogue.)
//   maxpool3s2_kernel    MaxPool2d(3, 2)
//   lpips_layer_kernel   unit-normalise both feature maps over channels, squared difference, 1x1 "lin" weights,
//                        spatial mean, accumulated over the five layers into one score per image pair
// This is 24 MFLOP per 32x32 image pair -- 6e-5 of a reconstruction -- so the kernels are written for clarity and
// coalesced access, not tuned: weights of the block's 4 output channels are wave-uniform (scalar loads), a thread owns
// one output position, adjacent threads adjacent positions.
#include "common.h"

namespace ddpm {

constexpr int kLpCob = 4;  // output channels per thread

__global__ __launch_bounds__(256) void lpips_conv_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                         const float *__restrict__ bias,
                                                         const float *__restrict__ in_scale,
                                                         const float *__restrict__ in_shift, float *__restrict__ out,
                                                         int N, int Cx, int Cin, int H, int W, int Cout, int Ho, int Wo,
                                                         int k, int stride, int pad, int relu) {
  const int64_t pos = blockIdx.x * (int64_t)256 + threadIdx.x;  // (n, ho, wo)
  const int co0 = blockIdx.y * kLpCob;
  const int HWo = Ho * Wo;
  if (pos >= (int64_t)N * HWo) return;
  const int n = (int)(pos / HWo), p = (int)(pos - (int64_t)n * HWo);
  const int ho = p / Wo, wo = p - ho * Wo;
  const int h0 = ho * stride - pad, w0 = wo * stride - pad;
  float acc[kLpCob];
#pragma unroll
  for (int j = 0; j < kLpCob; ++j) acc[j] = (co0 + j < Cout && bias) ? bias[co0 + j] : 0.f;
  for (int ci = 0; ci < Cin; ++ci) {
    const float *plane = in + ((size_t)n * Cx + (Cx == Cin ? ci : 0)) * H * W;  // 1-channel input feeds all three
    const float a = in_scale ? in_scale[ci] : 1.f, b = in_shift ? in_shift[ci] : 0.f;
    for (int kh = 0; kh < k; ++kh) {
      const int h = h0 + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int x = w0 + kw;
        if (x < 0 || x >= W) continue;
        const float v = plane[h * W + x] * a + b;  // zero padding applies to the scaled input
#pragma unroll
        for (int j = 0; j < kLpCob; ++j)
          if (co0 + j < Cout) acc[j] += v * w[(((size_t)(co0 + j) * Cin + ci) * k + kh) * k + kw];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kLpCob; ++j)
    if (co0 + j < Cout) out[((size_t)n * Cout + co0 + j) * HWo + p] = relu ? fmaxf(acc[j], 0.f) : acc[j];
}

__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         int64_t planes, int H, int W, int Ho, int Wo) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= planes * Ho * Wo) return;
  const int64_t pl = i / (Ho * Wo);
  const int p = (int)(i - pl * Ho * Wo), ho = p / Wo, wo = p - ho * Wo;
  const float *src = in + pl * H * W + (2 * ho) * W + 2 * wo;
  float m = src[0];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) m = fmaxf(m, src[a * W + b]);
  out[i] = m;
}

// one workgroup per image pair; threads walk the positions, each looping over the channels (coalesced across threads)
__global__ __launch_bounds__(256) void lpips_layer_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                          const float *__restrict__ lin, float *__restrict__ out, int C,
                                                          int HW, int accumulate) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float *a = f0 + (size_t)n * C * HW, *b = f1 + (size_t)n * C * HW;
  float acc = 0.f;
  for (int p = tid; p < HW; p += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float x = a[(size_t)c * HW + p], y = b[(size_t)c * HW + p];
      s0 += x * x;
      s1 += y * y;
    }
    const float r0 = 1.f / (sqrtf(s0) + 1e-10f), r1 = 1.f / (sqrtf(s1) + 1e-10f);
    float d2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = a[(size_t)c * HW + p] * r0 - b[(size_t)c * HW + p] * r1;
      d2 += lin[c] * (d * d);
    }
    acc += d2;
  }
  const float tot = block_sum_256(acc, red) / (float)HW;
  if (tid == 0) out[n] = accumulate ? out[n] + tot : tot;
}

int launch_lpips_conv(const float *in, const float *w, const float *bias, const float *in_scale, const float *in_shift,
                      float *out, int N, int Cx, int Cin, int H, int W, int Cout, int k, int stride, int pad, int relu,
                      hipStream_t s) {
  DDPM_CHECK_ARG(in && w && out, "lpips_conv: null pointer");
  DDPM_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, "lpips_conv: bad shape");
  DDPM_CHECK_ARG(Cx == Cin || Cx == 1, "lpips_conv: the input has %d channels, the layer wants %d (or 1, broadcast)", Cx, Cin);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  DDPM_CHECK_ARG(Ho > 0 && Wo > 0, "lpips_conv: image smaller than the kernel");
  const int64_t npos = (int64_t)N * Ho * Wo;
  ProfScope prof(s, "lpips_conv", 2.0 * npos * Cout * Cin * k * k, 4.0 * ((double)N * Cx * H * W + (double)npos * Cout));
  hipLaunchKernelGGL(lpips_conv_kernel, dim3((unsigned)((npos + 255) / 256), (Cout + kLpCob - 1) / kLpCob), dim3(256), 0, s,
                     in, w, bias, in_scale, in_shift, out, N, Cx, Cin, H, W, Cout, Ho, Wo, k, stride, pad, relu);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_maxpool3s2(const float *in, float *out, int64_t planes, int H, int W, hipStream_t s) {
  DDPM_CHECK_ARG(in && out && planes > 0 && H >= 3 && W >= 3, "maxpool3s2: bad arguments");
  const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
  const int64_t total = planes * Ho * Wo;
  ProfScope prof(s, "maxpool3s2", 9.0 * total, 4.0 * (planes * (double)H * W + total));
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, planes, H, W, Ho, Wo);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_lpips_layer(const float *f0, const float *f1, const float *lin, float *out, int N, int C, int HW,
                       int accumulate, hipStream_t s) {
  DDPM_CHECK_ARG(f0 && f1 && lin && out && N > 0 && C > 0 && HW > 0, "lpips_layer: bad arguments");
  ProfScope prof(s, "lpips_layer", 8.0 * N * C * (double)HW, 16.0 * N * C * (double)HW);
  hipLaunchKernelGGL(lpips_layer_kernel, dim3(N), dim3(256), 0, s, f0, f1, lin, out, C, HW, accumulate);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm

using namespace ddpm;

extern "C" int ddpm_lpips_conv_f32(const float *in, const float *w, const float *bias, const float *in_scale,
                                   const float *in_shift, float *out, int N, int Cx, int Cin, int H, int W, int Cout,
                                   int k, int stride, int pad, int relu, ddpm_stream_t stream) {
  return launch_lpips_conv(in, w, bias, in_scale, in_shift, out, N, Cx, Cin, H, W, Cout, k, stride, pad, relu,
                           as_stream(stream));
}

extern "C" int ddpm_maxpool3s2_f32(const float *in, float *out, int64_t planes, int H, int W, ddpm_stream_t stream) {
  return launch_maxpool3s2(in, out, planes, H, W, as_stream(stream));
}

extern "C" int ddpm_lpips_layer_f32(const float *f0, const float *f1, const float *lin, float *out, int N, int C, int HW,
                                    int accumulate, ddpm_stream_t stream) {
  return launch_lpips_layer(f0, f1, lin, out, N, C, HW, accumulate, as_stream(stream));
}
