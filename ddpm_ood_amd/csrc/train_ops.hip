// train_ops.hip -- the element-wise, normalisation and reduction kernels of the native training step (SURVEY.md 8(f) row f-3;
// reference: /root/reference/src/trainers/ddpm_trainer.py:78-109 -- F.mse_loss(model(noisy, t), noise), loss.backward(),
// optimizer.step() -- with Adam(lr = 2.5e-5) of /root/reference/src/trainers/base.py:156).  Round 6.
//
// The forward of a training step runs on the inference path's convolution kernels (ddpm_conv_f32) over MATERIALISED
// GroupNorm + SiLU outputs (the backward needs them as the weight-gradient operand anyway); the backward is
//   3x3 / 1x1 input gradients    = ddpm_conv_f32 with the weights rotated by 180 degrees and transposed (conv_weight_rot180t),
//                                  a Downsample's through a zero-stuffed dY, an Upsample's followed by a 2x2 sum,
//   weight gradients             = train_gemm.hip,
//   everything else              = the kernels below.  All HBM-bound, all deterministic (fixed-order reductions, no atomics).
#include "common.h"

#include <initializer_list>

namespace ddpm {

namespace {

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- GroupNorm, training form: statistics kept for the backward --------------------------------------------------------------
// All four kernels: one workgroup per (image, group), one WAVE per channel plane of the group (channels w, w + 4, ... of the group
// for wave w), 16-byte loads when the plane length is a multiple of 4 (VEC).  Sums inside a plane: per-lane partial sums in element
// order, then the fixed butterfly of wave_sum -- deterministic.
typedef float f4 __attribute__((ext_vector_type(4)));

// mean and 1 / sqrt(var + eps) of the group's Cg * HW values (two passes: exact mean first; the second pass reads the L2's copy)
template <bool VEC>
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, float *__restrict__ mr, int C, int HW, int G,
                                                       float eps) {
  __shared__ float red[4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int Cg = C / G;
  const size_t n = (size_t)Cg * HW;
  const float *p = x + ((size_t)b * C + (size_t)g * Cg) * HW;
  float s = 0.f;
  if (VEC) {
    const f4 *p4 = reinterpret_cast<const f4 *>(p);
    for (size_t i = threadIdx.x; i < n / 4; i += 256) {
      const f4 v = p4[i];
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (size_t i = threadIdx.x; i < n; i += 256) s += p[i];
  }
  const float mean = block_sum_256(s, red) / (float)n;
  float q = 0.f;
  if (VEC) {
    const f4 *p4 = reinterpret_cast<const f4 *>(p);
    for (size_t i = threadIdx.x; i < n / 4; i += 256) {
      const f4 v = p4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[j] - mean;
        q = __builtin_fmaf(d, d, q);
      }
    }
  } else {
    for (size_t i = threadIdx.x; i < n; i += 256) {
      const float d = p[i] - mean;
      q = __builtin_fmaf(d, d, q);
    }
  }
  const float var = block_sum_256(q, red) / (float)n;
  if (threadIdx.x == 0) {
    mr[2 * blockIdx.x] = mean;
    mr[2 * blockIdx.x + 1] = 1.0f / sqrtf(var + eps);
  }
}

// y = act((x - mean) rstd gamma + beta)
template <bool VEC>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ mr,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ y, int C, int HW, int G, int act) {
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int Cg = C / G, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float mean = mr[2 * blockIdx.x], rstd = mr[2 * blockIdx.x + 1];
  for (int k = wave; k < Cg; k += 4) {
    const int c = g * Cg + k;
    const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
    const float *p = x + ((size_t)b * C + c) * HW;
    float *o = y + ((size_t)b * C + c) * HW;
    if (VEC) {
      for (int i = lane; i < HW / 4; i += 64) {
        f4 v = reinterpret_cast<const f4 *>(p)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = __builtin_fmaf(v[j], sc, sh);
          v[j] = act == DDPM_ACT_SILU ? z * sigmoid_f(z) : z;
        }
        reinterpret_cast<f4 *>(o)[i] = v;
      }
    } else {
      for (int i = lane; i < HW; i += 64) {
        const float z = __builtin_fmaf(p[i], sc, sh);
        o[i] = act == DDPM_ACT_SILU ? z * sigmoid_f(z) : z;
      }
    }
  }
}

// max over the workgroup of a float's magnitude bits (NaN orders above every finite value); `red4` = 4 words of LDS
__device__ __forceinline__ unsigned block_absmax_bits_256(unsigned b, unsigned *red4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = b;
  __syncthreads();
  return max(max(red4[0], red4[1]), max(red4[2], red4[3]));
}
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// out[blockIdx] = magnitude bits of the largest of the block's `chunk` contiguous floats (the maxima the kernels above emit, for the
// shapes their register forms do not take)
__global__ __launch_bounds__(256) void absmax_chunks_kernel(const float *__restrict__ x, size_t chunk, unsigned *__restrict__ out) {
  __shared__ unsigned red4[4];
  const float *p = x + (size_t)blockIdx.x * chunk;
  unsigned m = 0;
  for (size_t i = threadIdx.x; i < chunk; i += 256) m = max(m, abs_bits(p[i]));
  m = block_absmax_bits_256(m, red4);
  if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// statistics + apply in one kernel for planes of up to 1 024 values and groups of up to 16 channels: the group's values are read ONCE
// into registers (the wave's CPW channels, QPL quads per lane and channel), mean and centred variance are block sums over them, y is
// written from them -- x crosses the memory system once instead of three times (gn_stats twice, gn_apply once).
template <int CPW, int QPL>
__global__ __launch_bounds__(256) void gn_fwd_reg_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float *__restrict__ y, float *__restrict__ mr,
                                                         unsigned *__restrict__ amax, int C, int HW, int G, float eps, int act) {
  __shared__ float red[4];
  __shared__ unsigned red4[4];
  unsigned ymax = 0;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int Cg = C / G, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nq = HW / 4;
  f4 xv[CPW][QPL];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int k = wave + 4 * j;
    const bool kok = k < Cg;
    const f4 *p = reinterpret_cast<const f4 *>(x + ((size_t)b * C + g * Cg + (kok ? k : 0)) * HW);
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      const int i = lane + 64 * q;
      const bool ok = kok && i < nq;
      const f4 v = p[ok ? i : 0];
      xv[j][q] = ok ? v : f4{0.f, 0.f, 0.f, 0.f};
      s += (xv[j][q][0] + xv[j][q][1]) + (xv[j][q][2] + xv[j][q][3]);
    }
  }
  const float n = (float)Cg * (float)HW;
  const float mean = block_sum_256(s, red) / n;
  float qq = 0.f;
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const bool kok = wave + 4 * j < Cg;
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      const bool ok = kok && lane + 64 * q < nq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = ok ? xv[j][q][e] - mean : 0.f;
        qq = __builtin_fmaf(d, d, qq);
      }
    }
  }
  const float rstd = 1.0f / sqrtf(block_sum_256(qq, red) / n + eps);
  if (threadIdx.x == 0) {
    mr[2 * blockIdx.x] = mean;
    mr[2 * blockIdx.x + 1] = rstd;
  }
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int k = wave + 4 * j;
    if (k >= Cg) break;  // (uniform per wave)
    const int c = g * Cg + k;
    const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
    f4 *o = reinterpret_cast<f4 *>(y + ((size_t)b * C + c) * HW);
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      const int i = lane + 64 * q;
      if (i < nq) {
        f4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = __builtin_fmaf(xv[j][q][e], sc, sh);
          r[e] = act == DDPM_ACT_SILU ? z * sigmoid_f(z) : z;
          ymax = max(ymax, abs_bits(r[e]));
        }
        o[i] = r;
      }
    }
  }
  if (amax) {  // (uniform)
    ymax = block_absmax_bits_256(ymax, red4);
    if (threadIdx.x == 0) amax[blockIdx.x] = ymax;
  }
}

// backward: per channel plane  s1 = sum dz, s2 = sum dz xhat  with dz = dy act'(z), z = xhat gamma + beta  (ws[b][c] = {s1, s2} for
// the parameter gradients), then  dx = rstd (dz gamma - (A + xhat Bq) / (Cg HW)),  A = sum_{c in group} gamma_c s1_c,
// Bq = sum gamma_c s2_c.  The second pass reads x and dy again -- out of the L2, the group's planes were just streamed through it.
constexpr int kGnMaxCg = 64;  // channels per group the fused backward keeps sums for in LDS
template <bool VEC>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                     const float *__restrict__ mr, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float *__restrict__ ws, float *__restrict__ dx,
                                                     unsigned *__restrict__ amax, float *__restrict__ rowsum, int C, int HW, int G,
                                                     int act, int accumulate) {
  __shared__ float sums[2 * kGnMaxCg];
  __shared__ unsigned red4[4];
  unsigned dmax = 0;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int Cg = C / G, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float mean = mr[2 * blockIdx.x], rstd = mr[2 * blockIdx.x + 1];
  auto dz_of = [&](float d, float xh, float ga, float be) __attribute__((always_inline)) {
    if (act == DDPM_ACT_SILU) {
      const float z = __builtin_fmaf(xh, ga, be), sg = sigmoid_f(z);
      d *= sg * (1.0f + z * (1.0f - sg));
    }
    return d;
  };
  for (int k = wave; k < Cg; k += 4) {
    const int c = g * Cg + k;
    const float ga = gamma[c], be = beta[c];
    const float *p = x + ((size_t)b * C + c) * HW, *d = dy + ((size_t)b * C + c) * HW;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
      for (int i = lane; i < HW / 4; i += 64) {
        const f4 xv = reinterpret_cast<const f4 *>(p)[i], dv = reinterpret_cast<const f4 *>(d)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - mean) * rstd, dz = dz_of(dv[j], xh, ga, be);
          s1 += dz;
          s2 = __builtin_fmaf(dz, xh, s2);
        }
      }
    } else {
      for (int i = lane; i < HW; i += 64) {
        const float xh = (p[i] - mean) * rstd, dz = dz_of(d[i], xh, ga, be);
        s1 += dz;
        s2 = __builtin_fmaf(dz, xh, s2);
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      sums[2 * k] = s1;
      sums[2 * k + 1] = s2;
      ws[2 * ((size_t)b * C + c)] = s1;
      ws[2 * ((size_t)b * C + c) + 1] = s2;
    }
  }
  __syncthreads();
  float A = 0.f, Bq = 0.f;
  for (int k = 0; k < Cg; ++k) {
    const float gk = gamma[g * Cg + k];
    A = __builtin_fmaf(gk, sums[2 * k], A);
    Bq = __builtin_fmaf(gk, sums[2 * k + 1], Bq);
  }
  const float inv = 1.0f / ((float)Cg * (float)HW);
  for (int k = wave; k < Cg; k += 4) {
    const int c = g * Cg + k;
    const float ga = gamma[c], be = beta[c];
    const float *p = x + ((size_t)b * C + c) * HW, *d = dy + ((size_t)b * C + c) * HW;
    float *o = dx + ((size_t)b * C + c) * HW;
    float rs = 0.f;
    if (VEC) {
      for (int i = lane; i < HW / 4; i += 64) {
        const f4 xv = reinterpret_cast<const f4 *>(p)[i], dv = reinterpret_cast<const f4 *>(d)[i];
        f4 r = accumulate ? reinterpret_cast<const f4 *>(o)[i] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - mean) * rstd, dz = dz_of(dv[j], xh, ga, be);
          const float v = rstd * (dz * ga - (A + xh * Bq) * inv);
          r[j] = accumulate ? r[j] + v : v;
          rs += r[j];
          dmax = max(dmax, abs_bits(r[j]));
        }
        reinterpret_cast<f4 *>(o)[i] = r;
      }
    } else {
      for (int i = lane; i < HW; i += 64) {
        const float xh = (p[i] - mean) * rstd, dz = dz_of(d[i], xh, ga, be);
        const float v = rstd * (dz * ga - (A + xh * Bq) * inv);
        const float r = accumulate ? o[i] + v : v;
        o[i] = r;
        rs += r;
        dmax = max(dmax, abs_bits(r));
      }
    }
    if (rowsum) {  // (uniform)
      rs = wave_sum(rs);
      if (lane == 0) rowsum[(size_t)b * C + c] = rs;
    }
  }
  if (amax) {  // (uniform)
    dmax = block_absmax_bits_256(dmax, red4);
    if (threadIdx.x == 0) amax[blockIdx.x] = dmax;
  }
}

// The same backward for planes of up to 1 024 values and groups of up to 16 channels (every GroupNorm of a 32 x 32 UNet): x-hat and dz
// of the wave's CPW channels stay in registers between the two passes (QPL quads per lane and channel) -- no second read of x and
// dy, no second sigmoid.
template <int CPW, int QPL>
__global__ __launch_bounds__(256) void gn_bwd_reg_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         const float *__restrict__ mr, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float *__restrict__ ws, float *__restrict__ dx,
                                                         unsigned *__restrict__ amax, float *__restrict__ rowsum, int C, int HW, int G,
                                                         int act, int accumulate) {
  __shared__ float sums[2 * 16];
  __shared__ unsigned red4[4];
  unsigned dmax = 0;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int Cg = C / G, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nq = HW / 4;
  const float mean = mr[2 * blockIdx.x], rstd = mr[2 * blockIdx.x + 1];
  f4 xh[CPW][QPL], dz[CPW][QPL];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int k = wave + 4 * j;
    const bool kok = k < Cg;
    const int c = g * Cg + (kok ? k : 0);
    const float ga = gamma[c], be = beta[c];
    const f4 *p = reinterpret_cast<const f4 *>(x + ((size_t)b * C + c) * HW), *d = reinterpret_cast<const f4 *>(dy + ((size_t)b * C + c) * HW);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      const int i = lane + 64 * q;
      const bool ok = kok && i < nq;
      const f4 xv = p[ok ? i : 0], dv = d[ok ? i : 0];  // (an unused slot re-reads the plane's first quad and contributes zero)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float h = (xv[e] - mean) * rstd;
        float t = ok ? dv[e] : 0.f;
        if (act == DDPM_ACT_SILU) {
          const float z = __builtin_fmaf(h, ga, be), sg = sigmoid_f(z);
          t *= sg * (1.0f + z * (1.0f - sg));
        }
        xh[j][q][e] = h;
        dz[j][q][e] = t;
        s1 += t;
        s2 = __builtin_fmaf(t, h, s2);
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0 && kok) {
      sums[2 * k] = s1;
      sums[2 * k + 1] = s2;
      ws[2 * ((size_t)b * C + c)] = s1;
      ws[2 * ((size_t)b * C + c) + 1] = s2;
    }
  }
  __syncthreads();
  float A = 0.f, Bq = 0.f;
  for (int k = 0; k < Cg; ++k) {
    const float gk = gamma[g * Cg + k];
    A = __builtin_fmaf(gk, sums[2 * k], A);
    Bq = __builtin_fmaf(gk, sums[2 * k + 1], Bq);
  }
  const float inv = 1.0f / ((float)Cg * (float)HW);
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int k = wave + 4 * j;
    if (k >= Cg) break;  // (uniform per wave)
    const int c = g * Cg + k;
    const float ga = gamma[c];
    f4 *o = reinterpret_cast<f4 *>(dx + ((size_t)b * C + c) * HW);
    float rs = 0.f;
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      const int i = lane + 64 * q;
      if (i < nq) {
        f4 r = accumulate ? o[i] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = rstd * (dz[j][q][e] * ga - (A + xh[j][q][e] * Bq) * inv);
          r[e] = accumulate ? r[e] + v : v;
          rs += r[e];
          dmax = max(dmax, abs_bits(r[e]));
        }
        o[i] = r;
      }
    }
    if (rowsum) {  // (uniform)
      rs = wave_sum(rs);
      if (lane == 0) rowsum[(size_t)b * C + c] = rs;
    }
  }
  if (amax) {  // (uniform)
    dmax = block_absmax_bits_256(dmax, red4);
    if (threadIdx.x == 0) amax[blockIdx.x] = dmax;
  }
}

// ---- reductions ---------------------------------------------------------------------------------------------------------------
// out[r] = sum of the r-th row of `cols` contiguous floats (bias / temb gradients: rows = (image, channel) planes): one wave per row
template <bool VEC>
__global__ __launch_bounds__(256) void row_sum_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *p = in + (size_t)row * cols;
  float s = 0.f;
  if (VEC) {
    for (int i = lane; i < cols / 4; i += 64) {
      const f4 v = reinterpret_cast<const f4 *>(p)[i];
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (int i = lane; i < cols; i += 64) s += p[i];
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}
// out[c] (+)= alpha * sum_r in[r * stride + c].  A workgroup owns CW = 2^cwl columns (64, or every column of a narrow matrix) and
// splits the rows over its 256 / CW thread groups (group q: rows q, q + RG, ... in four interleaved chains, so that the loads of a
// long column -- the batch, or the loss's per-workgroup partial sums -- are in flight together instead of one latency after the
// other); the groups' sums are added in group order.  out1 != NULL: columns alternate between two outputs (out0[c / 2] for even c,
// out1[c / 2] for odd c -- the {s1, s2} pairs of the GroupNorm backward).
__global__ __launch_bounds__(256) void col_sum_kernel(const float *__restrict__ in, float *__restrict__ out0, float *__restrict__ out1,
                                                      int rows, int cols, long long stride, float alpha, int accumulate, int cwl) {
  __shared__ float part[256];
  const int CW = 1 << cwl, RG = 256 >> cwl;
  const int cl = threadIdx.x & (CW - 1), rg = threadIdx.x >> cwl;
  const int c = blockIdx.x * CW + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    const float *p = in + c;
    int r = rg;
    for (; r + 3 * RG < rows; r += 4 * RG) {
      s0 += p[(size_t)r * stride];
      s1 += p[(size_t)(r + RG) * stride];
      s2 += p[(size_t)(r + 2 * RG) * stride];
      s3 += p[(size_t)(r + 3 * RG) * stride];
    }
    for (; r < rows; r += RG) s0 += p[(size_t)r * stride];
  }
  part[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg != 0 || c >= cols) return;
  float s = 0.f;
  for (int k = 0; k < RG; ++k) s += part[k * CW + cl];
  s *= alpha;
  float *o = out1 ? ((c & 1) ? out1 : out0) + (c >> 1) : out0 + c;
  *o = accumulate ? *o + s : s;
}

// ---- element-wise -------------------------------------------------------------------------------------------------------------
__global__ void silu_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * sigmoid_f(x[i]);
}
__global__ void silu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], sg = sigmoid_f(v);
  dx[i] = dy[i] * sg * (1.0f + v * (1.0f - sg));
}
__global__ void axpby_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, float alpha, float beta,
                             int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
// x *= alpha in place; a non-finite value ORs `bit` into the device status word (the unscale + overflow check of a scaled backward)
__global__ void scale_check_kernel(float *__restrict__ x, float alpha, int64_t n, unsigned *__restrict__ status, unsigned bit) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (4 * i + 3 < n) {
    f4 v = reinterpret_cast<f4 *>(x)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bad |= (__float_as_uint(v[j]) & 0x7f800000u) == 0x7f800000u;
      v[j] *= alpha;
    }
    reinterpret_cast<f4 *>(x)[i] = v;
  } else {
    for (int64_t k = 4 * i; k < n; ++k) {
      bad |= (__float_as_uint(x[k]) & 0x7f800000u) == 0x7f800000u;
      x[k] *= alpha;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0 && status) atomicOr(status, bit);
}
// dst[b, cd0 + c, :] (+)= src[b, cs0 + c, :]  (torch.cat in the forward, its split in the backward)
// (per image the source and the destination block are contiguous runs of C HW floats: VEC moves them in 16-byte pieces)
template <bool VEC>
__global__ void chan_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int Cs, int cs0, int Cd, int cd0, int HW,
                                 int64_t n, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (VEC) {  // n and i count quads
    const int64_t run = (int64_t)C * HW / 4;
    const int64_t b = i / run, r = i - b * run;
    const f4 v = reinterpret_cast<const f4 *>(src + ((size_t)b * Cs + cs0) * HW)[r];
    f4 *o = reinterpret_cast<f4 *>(dst + ((size_t)b * Cd + cd0) * HW) + r;
    *o = accumulate ? *o + v : v;
    return;
  }
  const int px = (int)(i % HW);
  const int64_t r = i / HW;
  const int c = (int)(r % C), b = (int)(r / C);
  const float v = src[((size_t)b * Cs + cs0 + c) * HW + px];
  float *o = dst + ((size_t)b * Cd + cd0 + c) * HW + px;
  *o = accumulate ? *o + v : v;
}
// mode 0: nearest x2 (F.interpolate);  1: its adjoint (sum of each 2x2 block);  2: zero-stuffing x2 (value at even, even)
__global__ void resample2_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t planes, int H, int W, int mode) {
  // H, W: extent of the SMALL image
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (mode == 1) {
    if (i >= planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int64_t pl = i / ((int64_t)H * W);
    const float *p = in + pl * 4 * H * W + (size_t)(2 * y) * (2 * W) + 2 * x;
    out[i] = (p[0] + p[1]) + (p[2 * W] + p[2 * W + 1]);
  } else {
    if (i >= planes * 4 * H * W) return;
    const int x = (int)(i % (2 * W)), y = (int)((i / (2 * W)) % (2 * H));
    const int64_t pl = i / ((int64_t)4 * H * W);
    const float v = in[pl * H * W + (size_t)(y >> 1) * W + (x >> 1)];
    out[i] = mode == 0 ? v : (((x | y) & 1) ? 0.f : v);
  }
}
// the same three maps on volumes (D, H, W: the SMALL extent): nearest x2, its adjoint (sum of each 2x2x2 block), zero-stuffing x2
__global__ void resample3_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t planes, int D, int H, int W, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t small = (int64_t)D * H * W;
  if (mode == 1) {
    if (i >= planes * small) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)((i / ((int64_t)W * H)) % D);
    const int64_t pl = i / small;
    const float *p = in + pl * 8 * small + ((size_t)(2 * z) * (2 * H) + 2 * y) * (2 * W) + 2 * x;
    const size_t zs = (size_t)4 * H * W;
    out[i] = ((p[0] + p[1]) + (p[2 * W] + p[2 * W + 1])) + ((p[zs] + p[zs + 1]) + (p[zs + 2 * W] + p[zs + 2 * W + 1]));
  } else {
    if (i >= planes * 8 * small) return;
    const int x = (int)(i % (2 * W)), y = (int)((i / (2 * W)) % (2 * H)), z = (int)((i / ((int64_t)4 * W * H)) % (2 * D));
    const int64_t pl = i / (8 * small);
    const float v = in[pl * small + ((size_t)(z >> 1) * H + (y >> 1)) * W + (x >> 1)];
    out[i] = mode == 0 ? v : (((x | y | z) & 1) ? 0.f : v);
  }
}
// wt[ci][co][ky][kx] = w[co][ci][k - 1 - ky][k - 1 - kx]: the weights of the input-gradient convolution
__global__ void conv_weight_rot180t_kernel(const float *__restrict__ w, float *__restrict__ wt, int Cout, int Cin, int kk) {
  const int64_t n = (int64_t)Cout * Cin * kk;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % kk);
  const int64_t r = i / kk;
  const int co = (int)(r % Cout), ci = (int)(r / Cout);
  wt[i] = w[((size_t)co * Cin + ci) * kk + (kk - 1 - t)];
}
// softmax over rows of `cols` floats, in place (one wave per row, the attention block of the training forward)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ s, int64_t rows, int cols) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float *p = s + r * cols;
  float m = -3.402823466e+38f;
  for (int i = lane; i < cols; i += 64) m = fmaxf(m, p[i]);
  m = wave_max(m);
  float sum = 0.f;
  for (int i = lane; i < cols; i += 64) {
    const float e = expf(p[i] - m);
    p[i] = e;
    sum += e;
  }
  const float inv = 1.0f / wave_sum(sum);
  for (int i = lane; i < cols; i += 64) p[i] *= inv;
}
// ds = p (dp - sum_j dp p) per row, in place over dp
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float *__restrict__ pr, float *__restrict__ dp, int64_t rows, int cols) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float *p = pr + r * cols;
  float *d = dp + r * cols;
  float dot = 0.f;
  for (int i = lane; i < cols; i += 64) dot = __builtin_fmaf(d[i], p[i], dot);
  dot = wave_sum(dot);
  for (int i = lane; i < cols; i += 64) d[i] = p[i] * (d[i] - dot);
}
// F.mse_loss(pred, target): dpred = 2 (pred - target) / n, partial[block] = sum (pred - target)^2 of the block's elements
__global__ __launch_bounds__(256) void mse_grad_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                       float *__restrict__ dpred, float *__restrict__ partial, int64_t n, float scale) {
  __shared__ float red[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float q = 0.f;
  if (i < n) {
    const float d = pred[i] - target[i];
    dpred[i] = scale * d;
    q = d * d;
  }
  q = block_sum_256(q, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = q;
}
// torch.optim.Adam (no weight decay, no amsgrad), one flat parameter buffer: bias corrections c1 = 1 - b1^t, c2 = 1 - b2^t from the host
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n,
                            float lr, float b1, float b2, float eps, float c1, float sqrt_c2, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);  // torch: exp_avg.lerp_(grad, 1 - beta1)
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrt_c2 + eps;
  p[i] -= (lr / c1) * (mi / denom);
}

__global__ void fill_kernel(float *__restrict__ out, float value, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = value;
}
// Standard normals from Philox-4x32-10 (counter = (element / 4, stream), key = seed) + Box-Muller: four values per counter, a
// pure function of (seed, stream, element index) -- the training noise does not have to reproduce torch's generator, it has to
// be reproducible and independent across ranks / steps (stream = step counter, seed mixes in the rank).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u;
  k[1] += 0xBB67AE85u;
}
__global__ void randn_kernel(float *__restrict__ out, int64_t n, uint64_t seed, uint64_t stream) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // four outputs per thread
  if (4 * q >= n) return;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)c[2 * h] + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)c[2 * h + 1] * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * logf(u1 < 1e-30f ? 1e-30f : u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * q + j < n) out[4 * q + j] = z[j];
}

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

}  // namespace ddpm

using namespace ddpm;

namespace {
inline bool vec4_ok(int HW, std::initializer_list<const void *> ptrs) {
  if (HW & 3) return false;
  for (const void *q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) & 15) return false;
  return true;
}
inline int col_sum_cwl(int cols) {
  int l = 0;
  while (l < 6 && (1 << l) < cols) ++l;
  return l;
}
}  // namespace

extern "C" int ddpm_gn_stats_f32(const float *x, float *mean_rstd, int B, int C, int HW, int groups, float eps, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && mean_rstd && B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "gn_stats: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_gn_stats", 0.0, 4.0 * B * C * (double)HW);
  if (vec4_ok(HW, {x})) hipLaunchKernelGGL(gn_stats_kernel<true>, dim3(B * groups), dim3(256), 0, s, x, mean_rstd, C, HW, groups, eps);
  else hipLaunchKernelGGL(gn_stats_kernel<false>, dim3(B * groups), dim3(256), 0, s, x, mean_rstd, C, HW, groups, eps);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_gn_apply_f32(const float *x, const float *mean_rstd, const float *gamma, const float *beta, float *y, int B, int C,
                                 int HW, int groups, int act, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && mean_rstd && gamma && beta && y && groups > 0 && C % groups == 0 && (act == DDPM_ACT_NONE || act == DDPM_ACT_SILU),
                 "gn_apply: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_gn_apply", 0.0, 8.0 * B * C * (double)HW);
  if (vec4_ok(HW, {x, y}))
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(B * groups), dim3(256), 0, s, x, mean_rstd, gamma, beta, y, C, HW, groups, act);
  else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(B * groups), dim3(256), 0, s, x, mean_rstd, gamma, beta, y, C, HW, groups, act);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_gn_forward_f32(const float *x, const float *gamma, const float *beta, float *y, float *mean_rstd,
                                   unsigned *y_absmax, int B, int C, int HW, int groups, float eps, int act, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && gamma && beta && y && mean_rstd && B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0 &&
                     (act == DDPM_ACT_NONE || act == DDPM_ACT_SILU),
                 "gn_forward: bad arguments");
  const int Cg = C / groups;
  static const bool split = getenv("DDPM_GN_FWD_REG") && atoi(getenv("DDPM_GN_FWD_REG")) == 0;  // (A/B switch)
  if (!(vec4_ok(HW, {x, y}) && HW <= 1024 && Cg <= 16) || split) {
    int rc = ddpm_gn_stats_f32(x, mean_rstd, B, C, HW, groups, eps, stream);
    if (!rc) rc = ddpm_gn_apply_f32(x, mean_rstd, gamma, beta, y, B, C, HW, groups, act, stream);
    if (!rc && y_absmax) {
      hipLaunchKernelGGL(absmax_chunks_kernel, dim3(B * groups), dim3(256), 0, as_stream(stream), y, (size_t)Cg * HW, y_absmax);
      DDPM_CHECK_LAUNCH();
    }
    return rc;
  }
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_gn_forward", 0.0, 8.0 * B * C * (double)HW);
  const int cpw = (Cg + 3) / 4;
  const dim3 grid(B * groups);
#define DDPM_GN_FWD_REG(CPW, QPL) \
  hipLaunchKernelGGL((gn_fwd_reg_kernel<CPW, QPL>), grid, dim3(256), 0, s, x, gamma, beta, y, mean_rstd, y_absmax, C, HW, groups, eps, act)
  if (HW <= 256) {
    if (cpw == 1) DDPM_GN_FWD_REG(1, 1); else if (cpw == 2) DDPM_GN_FWD_REG(2, 1); else if (cpw == 3) DDPM_GN_FWD_REG(3, 1); else DDPM_GN_FWD_REG(4, 1);
  } else {
    if (cpw == 1) DDPM_GN_FWD_REG(1, 4); else if (cpw == 2) DDPM_GN_FWD_REG(2, 4); else if (cpw == 3) DDPM_GN_FWD_REG(3, 4); else DDPM_GN_FWD_REG(4, 4);
  }
#undef DDPM_GN_FWD_REG
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_gn_backward_f32(const float *x, const float *dy, const float *mean_rstd, const float *gamma, const float *beta,
                                    float *dx, int accumulate_dx, float *dgamma, float *dbeta, float *ws, unsigned *dx_absmax,
                                    float *dx_rowsum, int B, int C, int HW, int groups, int act, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && dy && mean_rstd && gamma && beta && dx && dgamma && dbeta && ws && groups > 0 && C % groups == 0,
                 "gn_backward: bad arguments");
  DDPM_CHECK_ARG(C / groups <= kGnMaxCg, "gn_backward: %d channels per group (at most %d)", C / groups, kGnMaxCg);
  hipStream_t s = as_stream(stream);
  // algorithmic traffic: x and dy read, dx written (and read when it accumulates)
  ProfScope prof(s, "train_gn_backward", 0.0, 4.0 * (3 + (accumulate_dx ? 1 : 0)) * B * C * (double)HW);
  const int Cg = C / groups;
  static const bool two_pass = getenv("DDPM_GN_BWD_REG") && atoi(getenv("DDPM_GN_BWD_REG")) == 0;  // (A/B switch)
  if (vec4_ok(HW, {x, dy, dx}) && HW <= 1024 && Cg <= 16 && !two_pass) {
    const int cpw = (Cg + 3) / 4;
    const dim3 grid(B * groups);
#define DDPM_GN_BWD_REG(CPW, QPL) \
  hipLaunchKernelGGL((gn_bwd_reg_kernel<CPW, QPL>), grid, dim3(256), 0, s, x, dy, mean_rstd, gamma, beta, ws, dx, dx_absmax, dx_rowsum, C, HW, groups, act, accumulate_dx)
    if (HW <= 256) {
      if (cpw == 1) DDPM_GN_BWD_REG(1, 1); else if (cpw == 2) DDPM_GN_BWD_REG(2, 1); else if (cpw == 3) DDPM_GN_BWD_REG(3, 1); else DDPM_GN_BWD_REG(4, 1);
    } else {
      if (cpw == 1) DDPM_GN_BWD_REG(1, 4); else if (cpw == 2) DDPM_GN_BWD_REG(2, 4); else if (cpw == 3) DDPM_GN_BWD_REG(3, 4); else DDPM_GN_BWD_REG(4, 4);
    }
#undef DDPM_GN_BWD_REG
  } else if (vec4_ok(HW, {x, dy, dx}))
    hipLaunchKernelGGL(gn_bwd_kernel<true>, dim3(B * groups), dim3(256), 0, s, x, dy, mean_rstd, gamma, beta, ws, dx, dx_absmax, dx_rowsum,
                       C, HW, groups, act, accumulate_dx);
  else
    hipLaunchKernelGGL(gn_bwd_kernel<false>, dim3(B * groups), dim3(256), 0, s, x, dy, mean_rstd, gamma, beta, ws, dx, dx_absmax, dx_rowsum,
                       C, HW, groups, act, accumulate_dx);
  // dbeta[c] = sum_b s1[b, c], dgamma[c] = sum_b s2[b, c]  (ws is [B][C][{s1, s2}]: 2 C columns, alternating outputs)
  const int cwl = col_sum_cwl(2 * C);
  hipLaunchKernelGGL(col_sum_kernel, dim3((2 * C + (1 << cwl) - 1) >> cwl), dim3(256), 0, s, ws, dbeta, dgamma, B, 2 * C, (long long)2 * C,
                     1.0f, 0, cwl);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_row_sum_f32(const float *in, float *out, int64_t rows, int cols, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(in && out && rows > 0 && rows <= 0x7fffffff && cols > 0, "row_sum: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_row_sum", 0.0, 4.0 * rows * (double)cols);
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (vec4_ok(cols, {in})) hipLaunchKernelGGL(row_sum_kernel<true>, grid, dim3(256), 0, s, in, out, rows, cols);
  else hipLaunchKernelGGL(row_sum_kernel<false>, grid, dim3(256), 0, s, in, out, rows, cols);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_col_sum_f32(const float *in, float *out, int rows, int cols, int64_t row_stride, float alpha, int accumulate,
                                ddpm_stream_t stream) {
  DDPM_CHECK_ARG(in && out && rows > 0 && cols > 0, "col_sum: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_col_sum", 0.0, 4.0 * rows * (double)cols);
  const int cwl = col_sum_cwl(cols);
  hipLaunchKernelGGL(col_sum_kernel, dim3((cols + (1 << cwl) - 1) >> cwl), dim3(256), 0, s, in, out, static_cast<float *>(nullptr), rows, cols,
                     (long long)row_stride, alpha, accumulate, cwl);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_silu_f32(const float *x, float *y, int64_t n, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && y && n > 0, "silu: bad arguments");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(silu_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, y, n);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_silu_backward_f32(const float *x, const float *dy, float *dx, int64_t n, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && dy && dx && n > 0, "silu_backward: bad arguments");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, dy, dx, n);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_axpby_f32(const float *a, const float *b, float *out, float alpha, float beta, int64_t n, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(a && out && n > 0, "axpby: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_axpby", 0.0, 12.0 * n);
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(n)), dim3(256), 0, s, a, b, out, alpha, beta, n);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_scale_check_f32(float *x, float alpha, int64_t n, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(x && n > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "scale_check: bad arguments (x 16-byte aligned)");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_scale_check", 0.0, 8.0 * n);
  hipLaunchKernelGGL(scale_check_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, x, alpha, n, status_word(), DDPM_STATUS_NONFINITE_GRAD);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_chan_copy_f32(const float *src, float *dst, int B, int C, int Csrc, int csrc0, int Cdst, int cdst0, int HW,
                                  int accumulate, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(src && dst && B > 0 && C > 0 && HW > 0 && csrc0 >= 0 && cdst0 >= 0 && csrc0 + C <= Csrc && cdst0 + C <= Cdst,
                 "chan_copy: bad arguments");
  hipStream_t s = as_stream(stream);
  const int64_t n = (int64_t)B * C * HW;
  ProfScope prof(s, "train_chan_copy", 0.0, 8.0 * n);
  if (vec4_ok(HW, {src, dst}))
    hipLaunchKernelGGL(chan_copy_kernel<true>, dim3(blocks_for(n / 4)), dim3(256), 0, s, src, dst, C, Csrc, csrc0, Cdst, cdst0, HW, n / 4, accumulate);
  else
    hipLaunchKernelGGL(chan_copy_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, s, src, dst, C, Csrc, csrc0, Cdst, cdst0, HW, n, accumulate);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_resample2_f32(const float *in, float *out, int64_t planes, int H, int W, int mode, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(in && out && planes > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2, "resample2: bad arguments");
  hipStream_t s = as_stream(stream);
  const int64_t n = mode == 1 ? planes * H * W : planes * 4 * H * W;
  ProfScope prof(s, "train_resample2", 0.0, 5.0 * 4 * planes * H * W);
  hipLaunchKernelGGL(resample2_kernel, dim3(blocks_for(n)), dim3(256), 0, s, in, out, planes, H, W, mode);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_resample3_f32(const float *in, float *out, int64_t planes, int D, int H, int W, int mode, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(in && out && planes > 0 && D > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2, "resample3: bad arguments");
  hipStream_t s = as_stream(stream);
  const int64_t n = (mode == 1 ? 1 : 8) * planes * D * H * W;
  ProfScope prof(s, "train_resample3", 0.0, 9.0 * 4 * planes * D * H * W);
  hipLaunchKernelGGL(resample3_kernel, dim3(blocks_for(n)), dim3(256), 0, s, in, out, planes, D, H, W, mode);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_conv_weight_rot180t_f32(const float *w, float *wt, int Cout, int Cin, int taps, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w && wt && Cout > 0 && Cin > 0 && taps > 0, "conv_weight_rot180t: bad arguments");
  hipStream_t s = as_stream(stream);
  const int64_t n = (int64_t)Cout * Cin * taps;
  hipLaunchKernelGGL(conv_weight_rot180t_kernel, dim3(blocks_for(n)), dim3(256), 0, s, w, wt, Cout, Cin, taps);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_softmax_rows_f32(float *s_inout, int64_t rows, int cols, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(s_inout && rows > 0 && cols > 0, "softmax_rows: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_softmax", 0.0, 8.0 * rows * (double)cols);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(blocks_for(rows, 4)), dim3(256), 0, s, s_inout, rows, cols);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_softmax_backward_rows_f32(const float *p, float *dp_inout, int64_t rows, int cols, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(p && dp_inout && rows > 0 && cols > 0, "softmax_backward_rows: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(s, "train_softmax_bwd", 0.0, 12.0 * rows * (double)cols);
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(blocks_for(rows, 4)), dim3(256), 0, s, p, dp_inout, rows, cols);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_mse_loss_grad_f32(const float *pred, const float *target, float *dpred, float *partial, int64_t n, float grad_scale,
                                      ddpm_stream_t stream) {
  DDPM_CHECK_ARG(pred && target && dpred && partial && n > 0, "mse_loss_grad: bad arguments");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(mse_grad_kernel, dim3(blocks_for(n)), dim3(256), 0, s, pred, target, dpred, partial, n, grad_scale);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
                                  int step, float grad_scale, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  hipStream_t s = as_stream(stream);
  const double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
  ProfScope prof(s, "train_adam", 0.0, 28.0 * n);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, (float)c1,
                     (float)sqrt(c2), grad_scale);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_fill_f32(float *out, float value, int64_t n, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(out && n > 0, "fill: bad arguments");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(n)), dim3(256), 0, s, out, value, n);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_randn_f32(float *out, int64_t n, uint64_t seed, uint64_t stream_id, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(out && n > 0, "randn: bad arguments");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(randn_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, out, n, seed, stream_id);
  DDPM_CHECK_LAUNCH();
  return 0;
}
