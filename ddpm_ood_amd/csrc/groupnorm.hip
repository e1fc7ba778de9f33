// groupnorm.hip -- GroupNorm statistics -> per-(image, channel) scale / shift.
//
// Replaces the statistics half of F.group_norm inside ResnetBlock / AttentionBlock / out
// (SURVEY.md 2.3 row "group_norm(32 groups, eps=1e-6) + silu"; reference call site
// /root/reference/src/trainers/reconstruct.py:151-153).  The normalise + SiLU half is fused
// into the consuming convolution's staging (conv_mfma.hip), so this kernel is the only extra
// pass over the activation: one workgroup per (image, group), two-pass mean / variance in
// fp32, wave64 shuffle reductions.  Groups of up to 12 288 values (every level of the 32x32 and
// 3-D configurations) are held in registers between the passes -- all loads are issued before
// the first use -- larger ones re-read the group from L2.
// HBM-bound: algorithmic bytes = 4 * C * HW per image.  Handles a virtual torch.cat of two
// sources, including groups that straddle the seam (384 = 256 + 128 channels, 12 per group).
#include <stdlib.h>

#include "common.h"

namespace ddpm {

__global__ __launch_bounds__(256) void gn_scale_shift_kernel(const float *__restrict__ in1,
                                                             const float *__restrict__ in2, int C1, int C2,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float *__restrict__ scale,
                                                             float *__restrict__ shift, int HW, int G, float eps) {
  __shared__ float red[4];
  const int C = C1 + C2;
  const int cpg = C / G;
  const int n = blockIdx.y, g = blockIdx.x;
  const int c0 = g * cpg;
  const int tid = threadIdx.x;
  const int count = cpg * HW;

  auto plane = [&](int c) -> const float * {
    return (c < C1) ? in1 + ((size_t)n * C1 + c) * HW : in2 + ((size_t)n * C2 + (c - C1)) * HW;
  };

  constexpr int kHold = 12;  // float4 per thread kept in registers
  if ((HW & 3) == 0 && cpg * (HW >> 2) <= 256 * kHold) {
    const int hw4 = HW >> 2, n4 = cpg * hw4;
    float4 v[kHold];
#pragma unroll
    for (int i = 0; i < kHold; ++i) {
      const int e = tid + 256 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < n4) {
        const int c = e / hw4, p4 = e - c * hw4;
        v[i] = reinterpret_cast<const float4 *>(plane(c0 + c))[p4];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kHold; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = block_sum_256(s, red) / (float)count;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kHold; ++i) {
      if (tid + 256 * i < n4) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float var = block_sum_256(q, red) / (float)count;  // biased, as torch
    const float rstd = 1.0f / sqrtf(var + eps);
    if (tid < cpg) {
      const int c = c0 + tid;
      const float sc = rstd * gamma[c];
      scale[(size_t)n * C + c] = sc;
      shift[(size_t)n * C + c] = -sc * mean + beta[c];
    }
    return;
  }

  float s = 0.f;
  if ((HW & 3) == 0) {
    const int hw4 = HW >> 2;
    // eight 16-byte loads in flight per thread (one load per iteration was a latency chain: 1.8 TB/s on the 64x64 level
    // of the `big` UNet, where a group is 32 768 - 98 304 values)
    const int n4 = cpg * hw4;
    for (int e0 = tid; e0 < n4; e0 += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < n4) {
          const int c = e / hw4, p4 = e - c * hw4;
          v[i] = reinterpret_cast<const float4 *>(plane(c0 + c))[p4];
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  } else {
    for (int e = tid; e < count; e += 256) {
      const int c = e / HW, p = e - c * HW;
      s += plane(c0 + c)[p];
    }
  }
  const float mean = block_sum_256(s, red) / (float)count;

  float q = 0.f;
  if ((HW & 3) == 0) {
    const int hw4 = HW >> 2;
    const int n4 = cpg * hw4;
    for (int e0 = tid; e0 < n4; e0 += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + 256 * i;
        v[i] = make_float4(mean, mean, mean, mean);  // contributes zero
        if (e < n4) {
          const int c = e / hw4, p4 = e - c * hw4;
          v[i] = reinterpret_cast<const float4 *>(plane(c0 + c))[p4];
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
  } else {
    for (int e = tid; e < count; e += 256) {
      const int c = e / HW, p = e - c * HW;
      const float a = plane(c0 + c)[p] - mean;
      q += a * a;
    }
  }
  const float var = block_sum_256(q, red) / (float)count;  // biased, as torch
  const float rstd = 1.0f / sqrtf(var + eps);
  if (tid < cpg) {
    const int c = c0 + tid;
    const float sc = rstd * gamma[c];
    scale[(size_t)n * C + c] = sc;
    shift[(size_t)n * C + c] = -sc * mean + beta[c];
  }
}

// One WAVE per (image, group): no block barriers, no LDS.  The group (<= 64 * 4 * kWHold floats) is loaded with every
// 16-byte load in flight at once, reduced with wave64 shuffles (same two-pass mean / variance as above), and four
// groups share a 256-thread workgroup.  At 20 us per launch the block version was bound by its
// load -> barrier -> barrier latency chain, not by HBM (3.4 TB/s): 27 launches per `small` forward.
template <int kWHold>
__global__ __launch_bounds__(256) void gn_scale_shift_wave_kernel(const float *__restrict__ in1,
                                                                  const float *__restrict__ in2, int C1, int C2,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta,
                                                                  float *__restrict__ scale, float *__restrict__ shift,
                                                                  int HW, int G, float eps, int total) {
  const int C = C1 + C2;
  const int cpg = C / G;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);  // (image, group)
  if (item >= total) return;
  const int n = item / G, g = item - n * G;
  const int c0 = g * cpg;
  const int hw4 = HW >> 2, n4 = cpg * hw4;
  auto plane = [&](int c) -> const float * {
    return (c < C1) ? in1 + ((size_t)n * C1 + c) * HW : in2 + ((size_t)n * C2 + (c - C1)) * HW;
  };
  float4 v[kWHold];
#pragma unroll
  for (int i = 0; i < kWHold; ++i) {
    const int e = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
      const int c = e / hw4, p4 = e - c * hw4;
      v[i] = reinterpret_cast<const float4 *>(plane(c0 + c))[p4];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kWHold; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float count = (float)(cpg * HW);
  const float mean = wave_sum(s) / count;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kWHold; ++i) {
    if (lane + 64 * i < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = wave_sum(q) / count;  // biased, as torch
  const float rstd = 1.0f / sqrtf(var + eps);
  if (lane < cpg) {
    const int c = c0 + lane;
    const float sc = rstd * gamma[c];
    scale[(size_t)n * C + c] = sc;
    shift[(size_t)n * C + c] = -sc * mean + beta[c];
  }
}

int launch_gn_scale_shift(const float *in1, const float *in2, int C1, int C2, const float *gamma, const float *beta,
                          float *scale, float *shift, int B, int HW, int groups, float eps, hipStream_t s) {
  const int C = C1 + C2;
  DDPM_CHECK_ARG(in1 && gamma && beta && scale && shift, "gn: null pointer");
  DDPM_CHECK_ARG(C2 == 0 || in2, "gn: C2 > 0 but in2 is NULL");
  DDPM_CHECK_ARG(groups > 0 && C % groups == 0, "gn: C %% groups != 0");
  DDPM_CHECK_ARG(C / groups <= 256, "gn: more than 256 channels per group");
  DDPM_CHECK_ARG(B > 0 && B <= 65535 && HW > 0, "gn: bad B / HW");
  ProfScope prof(s, "gn_scale_shift", 5.0 * B * C * HW, 4.0 * B * C * (double)HW);
  const int cpg = C / groups;
  const long n4 = (HW & 3) == 0 ? (long)cpg * (HW >> 2) : -1;
  static const bool wave_ok = !(getenv("DDPM_GN_WAVE") && atoi(getenv("DDPM_GN_WAVE")) == 0);
  const int total = B * groups;
  if (wave_ok && n4 > 0 && cpg <= 64 && n4 <= 64 * 16) {  // a group of <= 4096 floats per wave
    const dim3 grid((total + 3) / 4);
    if (n4 <= 64 * 4)
      hipLaunchKernelGGL(gn_scale_shift_wave_kernel<4>, grid, dim3(256), 0, s, in1, in2, C1, C2, gamma, beta, scale,
                         shift, HW, groups, eps, total);
    else if (n4 <= 64 * 8)
      hipLaunchKernelGGL(gn_scale_shift_wave_kernel<8>, grid, dim3(256), 0, s, in1, in2, C1, C2, gamma, beta, scale,
                         shift, HW, groups, eps, total);
    else
      hipLaunchKernelGGL(gn_scale_shift_wave_kernel<16>, grid, dim3(256), 0, s, in1, in2, C1, C2, gamma, beta, scale,
                         shift, HW, groups, eps, total);
    DDPM_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(groups, B), dim3(256), 0, s, in1, in2, C1, C2, gamma, beta, scale,
                     shift, HW, groups, eps);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
