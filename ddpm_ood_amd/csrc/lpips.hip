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
                                                         int k, int stride, int pad, int relu,
                                                         const float *__restrict__ bias_map) {
  const int64_t pos = blockIdx.x * (int64_t)256 + threadIdx.x;  // (n, ho, wo)
  const int co0 = blockIdx.y * kLpCob;
  const int HWo = Ho * Wo;
  if (pos >= (int64_t)N * HWo) return;
  const int n = (int)(pos / HWo), p = (int)(pos - (int64_t)n * HWo);
  const int ho = p / Wo, wo = p - ho * Wo;
  const int h0 = ho * stride - pad, w0 = wo * stride - pad;
  float acc[kLpCob];
#pragma unroll
  for (int j = 0; j < kLpCob; ++j)  // bias_map: a per-position bias [Cout][Ho * Wo] (the folded grey first layer)
    acc[j] = co0 + j >= Cout ? 0.f : bias_map ? bias_map[(size_t)(co0 + j) * HWo + p] : bias ? bias[co0 + j] : 0.f;
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

// ---- the 5x5 layer on the fp32 MFMA pipe -------------------------------------------------------------------------------
// For 2.5-D LPIPS over 128^3 volumes (cfg5: 128 slices x 3 views x 2 inputs per volume and t-start) the 5x5 layer
// (64 -> 192 over 15 x 15) is 35 GFLOP per call and the scalar kernel above ran it at 9 TFLOP/s -- a quarter of a
// cfg5 step.  Same-padded k x k stride-1 convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32:
//   one workgroup = one image, its whole input (all Cin planes with a zero halo) staged once in LDS (Cin * PS floats,
//   PS = 32 mod 64 so that the two channel planes a wave reads at once sit on disjoint banks);
//   A = weights, packed [cout block of 32][channel pair][tap][lhi][cout]: one coalesced 256-byte load per (pair, tap),
//       held in registers for a whole channel pair and prefetched one pair ahead;
//   B = pixels straight from the LDS planes (lane = pixel of a 32-pixel tile, lhi = channel of the pair);
//   a wave owns one cout block x 4 pixel tiles (4 accumulator tiles); 12 waves = 192 couts x 8 tiles for AlexNet at 128^2.
// Channels are accumulated in ascending order, taps in row-major order inside a channel pair (fp32 MFMA = fmaf chain).
constexpr int kL5Tiles = 4;   // pixel tiles (of 32) per wave
constexpr int kL5Threads = 768;

typedef float l5_f32x16 __attribute__((ext_vector_type(16)));

template <int KS>
__global__ __launch_bounds__(kL5Threads) void lpips_conv_mfma_kernel(const float *__restrict__ in,
                                                                     const float *__restrict__ wp,
                                                                     const float *__restrict__ bias,
                                                                     float *__restrict__ out, int Cin, int H, int W,
                                                                     int Cout, int PS, int relu) {
  extern __shared__ __attribute__((aligned(16))) float l5_smem[];
  constexpr int KK = KS * KS, PAD = KS / 2;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int HW = H * W, WP = W + 2 * PAD;
  for (int i = tid; i < Cin * PS; i += kL5Threads) l5_smem[i] = 0.f;
  __syncthreads();
  const float *src = in + (size_t)n * Cin * HW;
  for (int i = tid; i < Cin * HW; i += kL5Threads) {
    const int c = i / HW, p = i - c * HW;
    const int y = p / W, x = p - y * W;
    l5_smem[c * PS + (y + PAD) * WP + x + PAD] = src[i];
  }
  __syncthreads();
  const int ncb = Cout / 32, nt = (HW + 31) / 32, ngrp = (nt + kL5Tiles - 1) / kL5Tiles, nkk = Cin / 2;
  for (int u = wave; u < ncb * ngrp; u += kL5Threads / 64) {
    const int cb = u % ncb, grp = u / ncb;
    int pb[kL5Tiles];  // LDS float index of this lane's pixel (tap 0, 0) in channel plane lhi
#pragma unroll
    for (int t = 0; t < kL5Tiles; ++t) {
      const int p = min((grp * kL5Tiles + t) * 32 + l31, HW - 1);
      const int y = p / W, x = p - y * W;
      pb[t] = lhi * PS + y * WP + x;
    }
    l5_f32x16 acc[kL5Tiles];
#pragma unroll
    for (int t = 0; t < kL5Tiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float *wq = wp + (size_t)cb * nkk * KK * 64 + lane;
    float a_cur[KK], a_nxt[KK];
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) a_cur[tap] = wq[tap * 64];
    for (int kk = 0; kk < nkk; ++kk) {
      const float *wn = wq + (size_t)min(kk + 1, nkk - 1) * KK * 64;
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) a_nxt[tap] = wn[tap * 64];
      const float *plane = l5_smem + 2 * kk * PS;
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        const int toff = (tap / KS) * WP + tap % KS;
#pragma unroll
        for (int t = 0; t < kL5Tiles; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[tap], plane[pb[t] + toff], acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) a_cur[tap] = a_nxt[tap];
    }
    // D: lane = pixel l31, register r = cout 8 (r / 4) + 4 lhi + r % 4 of the block
#pragma unroll
    for (int t = 0; t < kL5Tiles; ++t) {
      const int p = (grp * kL5Tiles + t) * 32 + l31;
      if (p < HW) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cb * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
          const float v = acc[t][r] + (bias ? bias[co] : 0.f);
          out[((size_t)n * Cout + co) * HW + p] = relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
  }
}

// torch [Cout][Cin][k][k] -> [cout block][channel pair][tap][lhi][cout of 32]
__global__ void lpips_pack_conv_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int KK) {
  const int64_t total = (int64_t)Cout * Cin * KK;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % KK), ci = (int)((i / KK) % Cin), co = (int)(i / ((int64_t)KK * Cin));
    wp[((((size_t)(co / 32) * (Cin / 2) + ci / 2) * KK + tap) * 2 + (ci & 1)) * 32 + co % 32] = w[i];
  }
}

static int l5_plane(int H, int W, int k) {  // floats per padded channel plane, = 32 mod 64
  const int ps = (H + k - 1) * (W + k - 1);
  return ps + ((32 - ps) % 64 + 64) % 64;
}

bool lpips_conv_mfma_supported(int Cin, int H, int W, int Cout, int k) {
  if (k != 5 && k != 3) return false;
  if (Cin < 2 || Cin % 2 || Cout % 32 || H < 1 || W < 1 || H * W < 64) return false;  // tiny images: the scalar kernel
  return (size_t)Cin * l5_plane(H, W, k) * sizeof(float) <= 160 * 1024;
}

int launch_lpips_pack_conv(const float *w, float *wp, int Cout, int Cin, int k, hipStream_t s) {
  DDPM_CHECK_ARG(w && wp && Cout > 0 && Cout % 32 == 0 && Cin > 0 && Cin % 2 == 0 && k > 0, "lpips pack: bad arguments");
  const int64_t total = (int64_t)Cout * Cin * k * k;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(lpips_pack_conv_kernel, dim3(blocks), dim3(256), 0, s, w, wp, Cout, Cin, k * k);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_lpips_conv_mfma(const float *in, const float *wp, const float *bias, float *out, int N, int Cin, int H, int W,
                           int Cout, int k, int relu, hipStream_t s) {
  DDPM_CHECK_ARG(in && wp && out && N > 0, "lpips_conv_mfma: null pointer");
  DDPM_CHECK_ARG(lpips_conv_mfma_supported(Cin, H, W, Cout, k), "lpips_conv_mfma: unsupported shape");
  const int PS = l5_plane(H, W, k);
  const size_t lds = (size_t)Cin * PS * sizeof(float);
  typedef void (*kern_t)(const float *, const float *, const float *, float *, int, int, int, int, int, int);
  kern_t kern = k == 5 ? lpips_conv_mfma_kernel<5> : lpips_conv_mfma_kernel<3>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lpips_conv_mfma_kernel<5>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lpips_conv_mfma_kernel<3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const double npos = (double)N * H * W;
  ProfScope prof(s, "lpips_conv_mfma", 2.0 * npos * Cout * Cin * k * k, 4.0 * (npos * Cin + npos * Cout + (double)Cout * Cin * k * k));
  hipLaunchKernelGGL(kern, dim3(N), dim3(kL5Threads), lds, s, in, wp, bias, out, Cin, H, W, Cout, PS, relu);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_lpips_conv(const float *in, const float *w, const float *bias, const float *in_scale, const float *in_shift,
                      float *out, int N, int Cx, int Cin, int H, int W, int Cout, int k, int stride, int pad, int relu,
                      hipStream_t s, const float *bias_map = nullptr) {
  DDPM_CHECK_ARG(in && w && out, "lpips_conv: null pointer");
  DDPM_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, "lpips_conv: bad shape");
  DDPM_CHECK_ARG(Cx == Cin || Cx == 1, "lpips_conv: the input has %d channels, the layer wants %d (or 1, broadcast)", Cx, Cin);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  DDPM_CHECK_ARG(Ho > 0 && Wo > 0, "lpips_conv: image smaller than the kernel");
  const int64_t npos = (int64_t)N * Ho * Wo;
  ProfScope prof(s, "lpips_conv", 2.0 * npos * Cout * Cin * k * k, 4.0 * ((double)N * Cx * H * W + (double)npos * Cout));
  hipLaunchKernelGGL(lpips_conv_kernel, dim3((unsigned)((npos + 255) / 256), (Cout + kLpCob - 1) / kLpCob), dim3(256), 0, s,
                     in, w, bias, in_scale, in_shift, out, N, Cx, Cin, H, W, Cout, Ho, Wo, k, stride, pad, relu, bias_map);
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

extern "C" int ddpm_lpips_conv_biasmap_f32(const float *in, const float *w, const float *bias_map, float *out, int N, int Cin,
                                           int H, int W, int Cout, int k, int stride, int pad, int relu, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(bias_map != nullptr, "lpips_conv_biasmap: bias_map is NULL");
  return launch_lpips_conv(in, w, nullptr, nullptr, nullptr, out, N, Cin, Cin, H, W, Cout, k, stride, pad, relu,
                           as_stream(stream), bias_map);
}

extern "C" int ddpm_lpips_conv_mfma_supported(int Cin, int H, int W, int Cout, int k) {
  return lpips_conv_mfma_supported(Cin, H, W, Cout, k) ? 1 : 0;
}

extern "C" int ddpm_lpips_pack_conv_weight_f32(const float *w, float *w_packed, int Cout, int Cin, int k, ddpm_stream_t stream) {
  return launch_lpips_pack_conv(w, w_packed, Cout, Cin, k, as_stream(stream));
}

extern "C" int ddpm_lpips_conv_mfma_f32(const float *in, const float *w_packed, const float *bias, float *out, int N, int Cin,
                                        int H, int W, int Cout, int k, int relu, ddpm_stream_t stream) {
  return launch_lpips_conv_mfma(in, w_packed, bias, out, N, Cin, H, W, Cout, k, relu, as_stream(stream));
}

extern "C" int ddpm_maxpool3s2_f32(const float *in, float *out, int64_t planes, int H, int W, ddpm_stream_t stream) {
  return launch_maxpool3s2(in, out, planes, H, W, as_stream(stream));
}

extern "C" int ddpm_lpips_layer_f32(const float *f0, const float *f1, const float *lin, float *out, int N, int C, int HW,
                                    int accumulate, ddpm_stream_t stream) {
  return launch_lpips_layer(f0, f1, lin, out, N, C, HW, accumulate, as_stream(stream));
}
