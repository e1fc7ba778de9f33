// attention.hip -- fused self-attention core (QK^T -> softmax -> PV -> + residual), fp32 MFMA.
//
// Replaces torch.baddbmm / softmax / torch.bmm of MONAI-Generative's AttentionBlock._attention
// (reference call site /root/reference/src/trainers/reconstruct.py:151-153).
// q, k, v come straight from the fused QKV 1x1 convolution in channel-major NCHW form
// [B, 3C, N]: q[d][i], k[d][j], v[d][j] with the token index contiguous.  That is already the
// MFMA operand order (32 consecutive lanes = 32 consecutive tokens), so
//   S[i][j]  = sum_d q[d][i] k[d][j]   takes A = q^T and B = k directly from global / L2,
//   O[d][i]  = sum_j v[d][j] P[i][j]   takes A = v (staged through LDS with a +1 pad so the
//                                       32 lanes that differ in d hit 32 different banks)
//                                       and B = P^T from the score tile in LDS,
// and the output tile O[d][i] stores 128-byte rows back into NCHW, fused with the residual.
// One workgroup = (image, head, 64 queries); keys are walked in blocks of 64 with an online
// (running max / running sum) softmax, so n = 64 (small UNet @ 8x8) is a single block and
// n = 4096 (big UNet @ 64x64) never materialises the n x n score matrix.
// Head dim is fixed at 256 (num_head_channels = 256 in both reference configs,
// /root/reference/src/trainers/base.py:73,84).
#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDH = 256;   // head dim
constexpr int kQB = 64;    // queries per workgroup
constexpr int kKB = 64;    // keys per block
constexpr int kLd = 65;    // padded LDS leading dimension

__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ qkv,
                                                        const float *__restrict__ residual,
                                                        float *__restrict__ out, int C, int N, int heads,
                                                        float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Vl = smem;                    // [256][65]
  float *Sl = Vl + kDH * kLd;          // [64][65]
  float *mrow = Sl + kQB * kLd;        // [64]
  float *lrow = mrow + kQB;            // [64]
  float *arow = lrow + kQB;            // [64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int qblk = blockIdx.x, hh = blockIdx.y, n = blockIdx.z;
  const int i0 = qblk * kQB;

  const float *qp = qkv + ((size_t)n * 3 * C + hh * kDH) * N;
  const float *kp = qp + (size_t)C * N;
  const float *vp = kp + (size_t)C * N;

  if (tid < kQB) {
    mrow[tid] = -INFINITY;
    lrow[tid] = 0.f;
  }

  f32x16 o[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;

  const int qi = wave >> 1, kj = wave & 1;
  const int iq = min(i0 + qi * 32 + l31, N - 1);  // clamped query index for operand reads

  for (int j0 = 0; j0 < N; j0 += kKB) {
    __syncthreads();  // previous block's PV finished with Vl / Sl (also orders the m/l init)

    // ---- stage V block [256 d][64 keys] ----------------------------------------------------
    for (int e = tid; e < kDH * kKB; e += 256) {
      const int d = e >> 6, j = e & 63;
      Vl[d * kLd + j] = (j0 + j < N) ? vp[(size_t)d * N + j0 + j] : 0.f;
    }

    // ---- S quadrant: rows = queries qi*32.., cols = keys kj*32.. ----------------------------
    {
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const int jk = min(j0 + kj * 32 + l31, N - 1);
      const float *qa = qp + (size_t)lhi * N + iq;
      const float *kb = kp + (size_t)lhi * N + jk;
#pragma unroll 8
      for (int d = 0; d < kDH; d += 2) {
        const float av = qa[(size_t)d * N];
        const float bv = kb[(size_t)d * N];
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, sacc, 0, 0, 0);
      }
      const bool colok = (j0 + kj * 32 + l31) < N;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Sl[row * kLd + kj * 32 + l31] = colok ? sacc[r] * scale : -INFINITY;
      }
    }
    __syncthreads();

    // ---- online softmax: 4 threads per query row, 16 keys each ------------------------------
    {
      const int row = tid >> 2, part = tid & 3;
      float *sr = Sl + row * kLd + part * 16;
      float bm = -INFINITY;
#pragma unroll
      for (int c = 0; c < 16; ++c) bm = fmaxf(bm, sr[c]);
      bm = fmaxf(bm, __shfl_xor(bm, 1, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 2, 64));
      const float mo = mrow[row];
      const float mn = fmaxf(mo, bm);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float p = expf(sr[c] - mn);
        sr[c] = p;
        sum += p;
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float alpha = expf(mo - mn);  // exp(-inf) = 0 on the first block
      __syncthreads();                    // all 4 readers of mrow[row] are done
      if (part == 0) {
        mrow[row] = mn;
        lrow[row] = lrow[row] * alpha + sum;
        arow[row] = alpha;
      }
    }
    __syncthreads();

    // ---- O[d][i] = alpha_i * O[d][i] + sum_j V[d][j] P[i][j] --------------------------------
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float al = arow[b * 32 + l31];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[a][b][r] *= al;
    }
    const float *va = Vl + ((wave * 2) * 32 + l31) * kLd + lhi;
    const float *pb = Sl + l31 * kLd + lhi;
#pragma unroll 4
    for (int jj = 0; jj < kKB; jj += 2) {
      const float a0 = va[jj];
      const float a1 = va[32 * kLd + jj];
      const float b0 = pb[jj];
      const float b1 = pb[32 * kLd + jj];
      o[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, o[0][0], 0, 0, 0);
      o[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, o[0][1], 0, 0, 0);
      o[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, o[1][0], 0, 0, 0);
      o[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, o[1][1], 0, 0, 0);
    }
  }

  // ---- normalise, add residual, store [B, C, N] ------------------------------------------------
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int i = i0 + b * 32 + l31;
    if (i < N) {
      const float inv = 1.0f / lrow[b * 32 + l31];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = (wave * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const size_t idx = ((size_t)n * C + hh * kDH + d) * N + i;
          float v = o[a][b][r] * inv;
          if (residual) v += residual[idx];
          out[idx] = v;
        }
      }
    }
  }
}

int launch_attention(const float *qkv, const float *residual, float *out, int B, int C, int N, int heads, float scale,
                     hipStream_t s) {
  DDPM_CHECK_ARG(qkv && out && B > 0 && N > 0 && heads > 0, "attention: null pointer or empty shape");
  DDPM_CHECK_ARG(C == heads * kDH, "attention: only head dim 256 is built (C = %d, heads = %d)", C, heads);
  DDPM_CHECK_ARG(B <= 65535 && heads <= 65535, "attention: grid too large");
  const size_t lds = (size_t)(kDH * kLd + kQB * kLd + 3 * kQB) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_done = true;
  }
  dim3 grid((N + kQB - 1) / kQB, heads, B);
  ProfScope prof(s, "attention", 4.0 * B * (double)N * N * C, 4.0 * B * C * (double)N * (residual ? 5 : 4));
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), lds, s, qkv, residual, out, C, N, heads, scale);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
