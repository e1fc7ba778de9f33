// vq.hip -- nearest-code search of the VQ-VAE quantiser (part of SURVEY 8 row f-1).
//
// Replaces EMAQuantizer.quantize + embedding lookup of MONAI-Generative's VQ-VAE (reference call sites
// /root/reference/src/trainers/reconstruct.py:124,166 through vqvae.decode_stage_2_outputs): for every latent vector
// z = x[b, :, p] the code k that minimises |z|^2 + |e_k|^2 - 2 z.e_k (first one on ties, as torch.max(-dist)),
// and the straight-through eval output x + (e_k - x).
// One thread per latent position keeps z in registers; codes are walked by all lanes of a wave together, so the
// codebook row is wave-uniform (scalar loads) and the inner product is D v_fmac with a scalar operand.  The four
// waves of a workgroup take a quarter of the codebook each for the same 64 positions and combine through LDS.
// 2 K D FLOP per position (0.5 MFLOP at K = 2 048, D = 128): VALU-bound, a few hundred microseconds per batch of
// 8^3 latents -- nowhere near the decode convolutions that follow.
#include "common.h"

namespace ddpm {

constexpr float kVqTieRel = 1e-5f;  // relative distance gap below which two codes count as a near-tie (ddpm_vq_near_ties_read)

template <int D>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float *__restrict__ x, const float *__restrict__ e,
                                                         const float *__restrict__ e2, int *__restrict__ idx,
                                                         float *__restrict__ out, int S, int K, long npos,
                                                         unsigned *__restrict__ status) {
  __shared__ float bd[4][64], bs[4][64];
  __shared__ int bi[4][64];
  const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
  const long pos = (long)blockIdx.x * 64 + lane;
  const bool live = pos < npos;
  const long b = live ? pos / S : 0, p = live ? pos - b * S : 0;
  const float *xp = x + (size_t)b * D * S + p;
  float z[D];
  float z2 = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    z[d] = live ? xp[(size_t)d * S] : 0.f;
    z2 += z[d] * z[d];
  }
  // a non-finite latent picks an arbitrary code: say so
  if (quarter == 0 && non_finite(z2) && status) atomicOr(status, (unsigned)DDPM_STATUS_NONFINITE_LATENT);
  const int kq = (K + 3) / 4, k0 = quarter * kq, k1 = min(K, k0 + kq);
  float best = INFINITY, second = INFINITY;  // (second: the runner-up distance -- only for the near-tie counter below)
  int besti = k0;
  for (int k = k0; k < k1; ++k) {
    const float *ek = e + (size_t)k * D;  // uniform over the wave
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dot = fmaf(z[d], ek[d], dot);
    const float dist = (z2 + e2[k]) - 2.f * dot;
    if (dist < best) {
      second = best;
      best = dist;
      besti = k;
    } else {
      second = fminf(second, dist);
    }
  }
  bd[quarter][lane] = best;
  bi[quarter][lane] = besti;
  bs[quarter][lane] = second;
  __syncthreads();
  if (quarter == 0 && live) {
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      if (bd[q][lane] < best) {  // strict: the earlier quarter (smaller index) wins a tie
        second = fminf(best, fminf(second, bs[q][lane]));
        best = bd[q][lane];
        besti = bi[q][lane];
      } else {
        second = fminf(second, bd[q][lane]);
      }
    }
    // a NEAR-TIE: the two nearest codes within kVqTieRel of each other -- a latent that differs in its 6th digit (another
    // machine's convolution rounding, the reference's own included) may pick the other one.  Counted, not flagged.
    if (status && second - best <= kVqTieRel * fabsf(best)) atomicAdd(status + 1, 1u);
    idx[pos] = besti;
    const float *ek = e + (size_t)besti * D;
    float *op = out + (size_t)b * D * S + p;
#pragma unroll
    for (int d = 0; d < D; ++d) op[(size_t)d * S] = z[d] + (ek[d] - z[d]);  // straight-through form, x + (q - x)
  }
}

// any embedding_dim (the built sizes above keep z in registers; this one re-reads it, cached, per code): same arithmetic order
__global__ __launch_bounds__(256) void vq_nearest_generic_kernel(const float *__restrict__ x, const float *__restrict__ e,
                                                                 const float *__restrict__ e2, int *__restrict__ idx,
                                                                 float *__restrict__ out, int D, int S, int K, long npos,
                                                                 unsigned *__restrict__ status) {
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= npos) return;
  const long b = pos / S, p = pos - b * S;
  const float *xp = x + (size_t)b * D * S + p;
  float z2 = 0.f;
  for (int d = 0; d < D; ++d) z2 += xp[(size_t)d * S] * xp[(size_t)d * S];
  if (non_finite(z2) && status) atomicOr(status, (unsigned)DDPM_STATUS_NONFINITE_LATENT);
  float best = INFINITY, second = INFINITY;
  int besti = 0;
  for (int k = 0; k < K; ++k) {
    const float *ek = e + (size_t)k * D;
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot = fmaf(xp[(size_t)d * S], ek[d], dot);
    const float dist = (z2 + e2[k]) - 2.f * dot;
    if (dist < best) {
      second = best;
      best = dist;
      besti = k;
    } else {
      second = fminf(second, dist);
    }
  }
  if (status && second - best <= kVqTieRel * fabsf(best)) atomicAdd(status + 1, 1u);
  idx[pos] = besti;
  const float *ek = e + (size_t)besti * D;
  float *op = out + (size_t)b * D * S + p;
  for (int d = 0; d < D; ++d) {
    const float zd = xp[(size_t)d * S];
    op[(size_t)d * S] = zd + (ek[d] - zd);
  }
}

__global__ void vq_code_norms_kernel(const float *__restrict__ e, float *__restrict__ e2, int K, int D) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s += e[(size_t)k * D + d] * e[(size_t)k * D + d];
  e2[k] = s;
}

int launch_vq_nearest(const float *x, const float *codebook, float *code_norms, int *idx, float *out, int B, int D, long S,
                      int K, hipStream_t s) {
  DDPM_CHECK_ARG(x && codebook && code_norms && idx && out, "vq_nearest: null pointer");
  DDPM_CHECK_ARG(B > 0 && S > 0 && K > 0, "vq_nearest: empty shape");
  DDPM_CHECK_ARG(D > 0, "vq_nearest: embedding_dim %d", D);
  DDPM_CHECK_ARG(S < (1L << 31), "vq_nearest: too many positions per image");
  const long npos = (long)B * S;
  hipLaunchKernelGGL(vq_code_norms_kernel, dim3((K + 255) / 256), dim3(256), 0, s, codebook, code_norms, K, D);
  ProfScope prof(s, "vq_nearest", 2.0 * npos * K * D, 4.0 * (2.0 * npos * D + (double)K * D));
  const dim3 grid((unsigned)((npos + 63) / 64));
#define DDPM_VQ_CASE(DD)                                                                                             \
  case DD:                                                                                                           \
    hipLaunchKernelGGL(vq_nearest_kernel<DD>, grid, dim3(256), 0, s, x, codebook, code_norms, idx, out, (int)S, K, npos, \
                       status_word());                                                                                   \
    break;
  switch (D) {
    DDPM_VQ_CASE(8)
    DDPM_VQ_CASE(16)
    DDPM_VQ_CASE(32)
    DDPM_VQ_CASE(64)
    DDPM_VQ_CASE(128)
    default:  // other embedding sizes: the generic kernel (x and out must not alias)
      hipLaunchKernelGGL(vq_nearest_generic_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, s, x, codebook, code_norms,
                         idx, out, D, (int)S, K, npos, status_word());
      break;
  }
#undef DDPM_VQ_CASE
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm

using namespace ddpm;

extern "C" int ddpm_vq_near_ties_read(unsigned *count, int clear, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(count != nullptr, "vq_near_ties_read: NULL pointer");
  unsigned *st = status_word();
  DDPM_CHECK_ARG(st != nullptr, "vq_near_ties_read: no status word on this device");
  hipError_t e = hipMemcpyAsync(count, st + 1, sizeof(unsigned), hipMemcpyDeviceToHost, as_stream(stream));
  if (e == hipSuccess && clear) e = hipMemsetAsync(st + 1, 0, sizeof(unsigned), as_stream(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
  if (e != hipSuccess) {
    set_error("vq_near_ties_read: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ddpm_vq_nearest_f32(const float *x, const float *codebook, float *code_norms, int *idx, float *out, int B,
                                   int D, int64_t S, int K, ddpm_stream_t stream) {
  return launch_vq_nearest(x, codebook, code_norms, idx, out, B, D, (long)S, K, as_stream(stream));
}
