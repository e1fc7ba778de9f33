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

// ---- statistics produced elsewhere (ddpm_conv_desc.stats_out / channel_stats_kernel): scale / shift without a pass over
// the activation.  One workgroup per image: thread c folds channel c's slices to {sum of means, M2 about the channel's
// own mean} in LDS, thread g then merges its group's channels (the group mean first, then the squared deviations about
// it: the two-pass formula on pre-reduced slices) in a fixed order, and the channels' threads write scale / shift.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ st1, int parts1, int C1,
                                                          const float *__restrict__ st2, int parts2, int C2,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ scale,
                                                          float *__restrict__ shift, int HW, int G, float eps) {
  extern __shared__ float sm[];  // [C] channel means, [C] channel M2, [G] group mean, [G] group rstd
  const int C = C1 + C2, cpg = C / G;
  const int n = blockIdx.x, tid = threadIdx.x;
  float *cmean = sm, *cm2 = sm + C, *gmean = sm + 2 * C, *grstd = sm + 2 * C + G;
  // (gamma / beta of the thread's first two channels are requested now: one memory round trip less on the 7 us critical path)
  const float ga0 = tid < C ? gamma[tid] : 0.f, be0 = tid < C ? beta[tid] : 0.f;
  const float ga1 = tid + 256 < C ? gamma[tid + 256] : 0.f, be1 = tid + 256 < C ? beta[tid + 256] : 0.f;
  for (int c = tid; c < C; c += 256) {
    const bool first = c < C1;
    const int parts = first ? parts1 : parts2;
    const float2 *e = first ? reinterpret_cast<const float2 *>(st1) + ((size_t)n * C1 + c) * parts1
                            : reinterpret_cast<const float2 *>(st2) + ((size_t)n * C2 + (c - C1)) * parts2;
    float2 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = p < parts ? e[p] : make_float2(0.f, 0.f);
    float ms = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) ms += v[p].x;
    const float mean = ms / (float)parts;  // equal slices
    float q = 0.f, dv = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (p < parts) {
        const float dl = v[p].x - mean;
        q += v[p].y;
        dv += dl * dl;
      }
    }
    cmean[c] = mean;
    cm2[c] = q + (float)(HW / parts) * dv;
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    const int c0 = g * cpg;
    float ms = 0.f;
    for (int c = c0; c < c0 + cpg; ++c) ms += cmean[c];
    const float mean = ms / (float)cpg;  // every channel has HW values
    float q = 0.f, dv = 0.f;
    for (int c = c0; c < c0 + cpg; ++c) {
      const float dl = cmean[c] - mean;
      q += cm2[c];
      dv += dl * dl;
    }
    const float var = (q + (float)HW * dv) / ((float)cpg * (float)HW);  // biased, as torch
    gmean[g] = mean;
    grstd[g] = 1.0f / sqrtf(var + eps);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float ga = c == tid ? ga0 : c == tid + 256 ? ga1 : gamma[c], be = c == tid ? be0 : c == tid + 256 ? be1 : beta[c];
    const float sc = grstd[g] * ga;
    scale[(size_t)n * C + c] = sc;
    shift[(size_t)n * C + c] = -sc * gmean[g] + be;
  }
}

int launch_gn_finalize(const float *st1, int parts1, int C1, const float *st2, int parts2, int C2, const float *gamma,
                       const float *beta, float *scale, float *shift, int B, int HW, int groups, float eps, hipStream_t s) {
  const int C = C1 + C2;
  DDPM_CHECK_ARG(st1 && gamma && beta && scale && shift, "gn_finalize: null pointer");
  DDPM_CHECK_ARG(C2 == 0 || (st2 && parts2 > 0), "gn_finalize: C2 > 0 but no second slab");
  DDPM_CHECK_ARG(groups > 0 && C % groups == 0, "gn_finalize: C %% groups != 0");
  DDPM_CHECK_ARG(B > 0 && HW > 0 && parts1 > 0 && parts1 <= 8 && parts2 <= 8 && HW % parts1 == 0 &&
                     (C2 == 0 || HW % parts2 == 0),
                 "gn_finalize: bad B / HW / parts (1 .. 8 equal slices)");
  const size_t lds = (size_t)(2 * C + 2 * groups) * sizeof(float);
  DDPM_CHECK_ARG(lds <= 48 * 1024, "gn_finalize: too many channels");
  ProfScope prof(s, "gn_finalize", 6.0 * B * C * parts1, 8.0 * B * (double)(C1 * parts1 + C2 * parts2) + 8.0 * B * C);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), lds, s, st1, parts1, C1, st2, parts2, C2, gamma, beta, scale,
                     shift, HW, groups, eps);
  DDPM_CHECK_LAUNCH();
  return 0;
}

// Per-channel {mean, M2} of a tensor whose producer does not emit them: one wave per (image, channel), two passes (the
// plane stays in registers up to 64 * 4 * kCHold floats, larger planes are re-read from L2).
template <int kCHold>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float *__restrict__ in, float *__restrict__ stats,
                                                            int HW, int total) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);  // (image, channel)
  if (item >= total) return;
  const float *pl = in + (size_t)item * HW;
  float mean, q = 0.f;
  if (kCHold > 0) {
    const int n4 = HW >> 2;
    float4 v[kCHold > 0 ? kCHold : 1];
#pragma unroll
    for (int i = 0; i < kCHold; ++i) {
      const int e = lane + 64 * i;
      v[i] = e < n4 ? reinterpret_cast<const float4 *>(pl)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < kCHold; ++i) sm += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = wave_sum(sm) / (float)HW;
#pragma unroll
    for (int i = 0; i < kCHold; ++i) {
      if (lane + 64 * i < n4) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
  } else {
    float sm = 0.f;
    for (int e = lane; e < HW; e += 64) sm += pl[e];
    mean = wave_sum(sm) / (float)HW;
    for (int e = lane; e < HW; e += 64) {
      const float a = pl[e] - mean;
      q += a * a;
    }
  }
  q = wave_sum(q);
  if (lane == 0) reinterpret_cast<float2 *>(stats)[item] = make_float2(mean, q);
}

int launch_channel_stats(const float *in, float *stats, int B, int C, int HW, hipStream_t s) {
  DDPM_CHECK_ARG(in && stats && B > 0 && C > 0 && HW > 0, "channel_stats: bad argument");
  ProfScope prof(s, "gn_channel_stats", 5.0 * B * C * HW, 4.0 * B * C * (double)HW);
  const long total = (long)B * C;
  DDPM_CHECK_ARG(total < (1l << 31) - 4, "channel_stats: too many planes");
  const dim3 grid((unsigned)((total + 3) / 4));
  const bool v4 = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  const int n4 = HW >> 2;
  if (v4 && n4 <= 64)
    hipLaunchKernelGGL(channel_stats_kernel<1>, grid, dim3(256), 0, s, in, stats, HW, (int)total);
  else if (v4 && n4 <= 64 * 4)
    hipLaunchKernelGGL(channel_stats_kernel<4>, grid, dim3(256), 0, s, in, stats, HW, (int)total);
  else if (v4 && n4 <= 64 * 16)
    hipLaunchKernelGGL(channel_stats_kernel<16>, grid, dim3(256), 0, s, in, stats, HW, (int)total);
  else
    hipLaunchKernelGGL(channel_stats_kernel<0>, grid, dim3(256), 0, s, in, stats, HW, (int)total);
  DDPM_CHECK_LAUNCH();
  return 0;
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
