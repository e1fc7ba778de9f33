// train_gemm.hip -- the contractions of the native training step (reference:
// /root/reference/src/trainers/ddpm_trainer.py:78-109 -- loss.backward() through generative's DiffusionModelUNet -- and
// /root/reference/src/trainers/base.py:156).  Round 6.
//
//   gemm_f32_kernel,       C[z] = alpha op(A[z]) op(B[z]) + beta C[z], every operand addressed by element strides, the K index and
//   gemm_f32_v4_kernel     the batch index each split in two levels (k = k0 K1 + k1, z = z0 Z1 + z1).  One entry point therefore
//                          covers everything in the backward pass that is not a 3x3 convolution: Linear weight / input gradients
//                          (time_embed, time_emb_proj, to_q / to_k / to_v), 1x1-convolution weight gradients (K = (image, pixel)),
//                          1x1 input gradients, and the five batched products of the attention block's forward and backward
//                          (Q^T K, V P^T, dO^T V, dS K, dS^T Q) on channel-major [B, C, N] tensors without a transpose pass.
//                          fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   conv3x3_wgrad_*        dW[co, ci, ky, kx] = sum over (image, output pixel) of dY[co, p] A[ci, s p + (ky, kx) - 1]: 64 couts x
//                          64 cins x 9 taps per workgroup (a wave holds nine 32x32 accumulator tiles, one per tap), the pixel
//                          stream split over workgroups, partial sums reduced in a fixed order by a second pass (no atomics:
//                          bit-reproducible).  _f16x3_kernel: stride 1, W in 8 .. 64, on v_mfma_f32_32x32x16_f16 at split
//                          precision (three f16 products per fp32 product, operands rescaled from their measured maxima: the
//                          same <= 3e-6 bound against float64 as the fp32 form is tested to); _staged_kernel / _kernel: fp32
//                          MFMA, stride 1 (other widths) and stride 2 (Downsample).
//
// Training needs gradients to ~1e-6 of their scale (the Adam step of the test is compared with CPU autograd): a single f16
// product (11 bits) does not give that, the split form does.
// Operand lanes of the fp32 MFMA: A: lane -> (m = lane % 32, k = lane / 32); B: lane -> (k = lane / 32, n = lane % 32);
// of the f16 MFMA: A: lane -> (m = lane % 32, k = 8 (lane / 32) .. + 7), B alike;
// accumulator register i of lane l (both): row m = 8 (i / 4) + 4 (l / 32) + i % 4, column n = l % 32.
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace ddpm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmP {
  const float *A, *B;
  float *C;
  int M, N, K, K1;  // K = K0 * K1
  long long sAm, sAk0, sAk1;
  long long sBn, sBk0, sBk1;
  long long sCm, sCn;
  int Z, Z1;  // Z = Z0 * Z1
  long long sAz0, sAz1, sBz0, sBz1, sCz0, sCz1;
  float alpha, beta;
  int S, kchunk;  // S > 1: the K range is cut into S slices of kchunk (a multiple of kGK); slice sp of batch item z writes its raw
  float *part;    // sums to part[(z S + sp) M N + m N + n], gemm_splitk_reduce_kernel adds them in order and applies alpha / beta
};

constexpr int kGT = 64;   // C tile (rows and columns) per workgroup
constexpr int kGK = 16;   // K per LDS stage
constexpr int kGP = 65;   // padded LDS row (k-major: [k][m])

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmP p) {
  __shared__ float As[kGK * kGP], Bs[kGK * kGP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z / p.S, sp = blockIdx.z - z * p.S;
  const int z0 = z / p.Z1, z1 = z - z0 * p.Z1;
  const int kbeg = sp * p.kchunk, kend = p.S > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const float *A = p.A + z0 * p.sAz0 + z1 * p.sAz1;
  const float *B = p.B + z0 * p.sBz0 + z1 * p.sBz1;
  float *C = p.C + z0 * p.sCz0 + z1 * p.sCz1;
  const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  // staging map: the index whose stride is 1 runs fastest over the threads
  const bool a_kfast = p.sAk1 == 1 && p.sAm != 1, b_kfast = p.sBk1 == 1 && p.sBn != 1;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int kt = kbeg; kt < kend; kt += kGK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;  // 0 .. 1023
      {
        const int k = a_kfast ? (e & 15) : (e >> 6), m = a_kfast ? (e >> 4) : (e & 63);
        const int kg = kt + k, mg = m0 + m;
        float v = 0.f;
        if (kg < kend && mg < p.M) {
          const int k0 = kg / p.K1, k1 = kg - k0 * p.K1;
          v = A[mg * p.sAm + k0 * p.sAk0 + k1 * p.sAk1];
        }
        As[k * kGP + m] = v;
      }
      {
        const int k = b_kfast ? (e & 15) : (e >> 6), n = b_kfast ? (e >> 4) : (e & 63);
        const int kg = kt + k, ng = n0 + n;
        float v = 0.f;
        if (kg < kend && ng < p.N) {
          const int k0 = kg / p.K1, k1 = kg - k0 * p.K1;
          v = B[ng * p.sBn + k0 * p.sBk0 + k1 * p.sBk1];
        }
        Bs[k * kGP + n] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGK; k += 2) {
      const float a = As[(k + (lane >> 5)) * kGP + wm + (lane & 31)];
      const float b = Bs[(k + (lane >> 5)) * kGP + wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn + (lane & 31);
  if (p.S > 1) {
    float *out = p.part + (size_t)blockIdx.z * p.M * p.N;
    if (n < p.N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (m < p.M) out[(size_t)m * p.N + n] = acc[i];
      }
    }
    return;
  }
  if (n < p.N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
      if (m < p.M) {
        float *c = C + m * p.sCm + n * p.sCn;
        const float v = p.alpha * acc[i];
        *c = p.beta != 0.f ? v + p.beta * *c : v;
      }
    }
  }
}

// The same product with 16-byte staging loads, 32 K per stage and the next stage's loads in flight while this one multiplies -- for
// operands whose unit-stride index is K (AK / BK) or the row / column index, 16-byte aligned with every other stride a multiple of
// four floats (gemm_fast_ok): the 1x1 convolutions' weight gradients (both operands K-major, K = images x pixels) ran at 19 TFLOP/s
// through the 4-byte, 16-K form above, whose loads touch 64 bytes per row.
constexpr int kGKF = 32;
template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_f32_v4_kernel(const GemmP p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  __shared__ float As[kGKF * kGP], Bs[kGKF * kGP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z / p.S, sp = blockIdx.z - z * p.S;
  const int z0 = z / p.Z1, z1 = z - z0 * p.Z1;
  const int kbeg = sp * p.kchunk, kend = p.S > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const float *A = p.A + z0 * p.sAz0 + z1 * p.sAz1;
  const float *B = p.B + z0 * p.sBz0 + z1 * p.sBz1;
  float *C = p.C + z0 * p.sCz0 + z1 * p.sCz1;
  const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  // staging slots: two quads per operand and thread.  K-major: quad = (row e / 8, k = 4 (e % 8) .. + 3); row-major: (k = e / 16,
  // rows 4 (e % 16) .. + 3)
  int a_k[2], a_m[2], b_k[2], b_n[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i;
    a_k[i] = AK ? 4 * (e & 7) : (e >> 4);
    a_m[i] = AK ? (e >> 3) : 4 * (e & 15);
    b_k[i] = BK ? 4 * (e & 7) : (e >> 4);
    b_n[i] = BK ? (e >> 3) : 4 * (e & 15);
  }
  v4 ra[2], rb[2];
  auto fetch = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      {
        const int kg = kt + a_k[i], mg = m0 + a_m[i];
        const int k0 = kg / p.K1, k1 = kg - k0 * p.K1;
        ra[i] = kg < kend && mg < p.M ? *reinterpret_cast<const v4 *>(A + mg * p.sAm + k0 * p.sAk0 + k1 * p.sAk1) : v4{0.f, 0.f, 0.f, 0.f};
      }
      {
        const int kg = kt + b_k[i], ng = n0 + b_n[i];
        const int k0 = kg / p.K1, k1 = kg - k0 * p.K1;
        rb[i] = kg < kend && ng < p.N ? *reinterpret_cast<const v4 *>(B + ng * p.sBn + k0 * p.sBk0 + k1 * p.sBk1) : v4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (kbeg < kend) fetch(kbeg);
  for (int kt = kbeg; kt < kend; kt += kGKF) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        As[AK ? (a_k[i] + j) * kGP + a_m[i] : a_k[i] * kGP + a_m[i] + j] = ra[i][j];
        Bs[BK ? (b_k[i] + j) * kGP + b_n[i] : b_k[i] * kGP + b_n[i] + j] = rb[i][j];
      }
    }
    __syncthreads();
    if (kt + kGKF < kend) fetch(kt + kGKF);
#pragma unroll
    for (int k = 0; k < kGKF; k += 2) {
      const float a = As[(k + (lane >> 5)) * kGP + wm + (lane & 31)];
      const float b = Bs[(k + (lane >> 5)) * kGP + wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  const int n = n0 + wn + (lane & 31);
  if (p.S > 1) {
    float *out = p.part + (size_t)blockIdx.z * p.M * p.N;
    if (n < p.N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (m < p.M) out[(size_t)m * p.N + n] = acc[i];
      }
    }
    return;
  }
  if (n < p.N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
      if (m < p.M) {
        float *c = C + m * p.sCm + n * p.sCn;
        const float v = p.alpha * acc[i];
        *c = p.beta != 0.f ? v + p.beta * *c : v;
      }
    }
  }
}

// Both operands K-major and inside the f16 exponent range (ddpm_gemm_desc.split_f16: the 1x1 weight gradients of a backward that
// runs under the training step's gradient scale -- dy pre-scaled, x an activation): the same 64 x 64 tile on
// v_mfma_f32_32x32x16_f16 at split precision, 2 x 3 MFMAs of 8 passes per 32-K stage instead of 16 of 16.  Staged quads are
// converted to (hi, lo 2^5) halves on their way into LDS; rows are 40 halves (80 bytes: five 16-byte slots, odd) apart.
constexpr int kGH = 40;
__global__ __launch_bounds__(256) void gemm_f16x3_kk_kernel(const GemmP p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  __shared__ _Float16 Ah[2][kGT * kGH], Bh[2][kGT * kGH];  // [hi, lo][row][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z / p.S, sp = blockIdx.z - z * p.S;
  const int z0 = z / p.Z1, z1 = z - z0 * p.Z1;
  const int kbeg = sp * p.kchunk, kend = p.S > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const float *A = p.A + z0 * p.sAz0 + z1 * p.sAz1;
  const float *B = p.B + z0 * p.sBz0 + z1 * p.sBz1;
  float *C = p.C + z0 * p.sCz0 + z1 * p.sCz1;
  const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  int s_k[2], s_r[2];  // staging slots: quad = (row e / 8, k = 4 (e % 8) .. + 3)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i;
    s_k[i] = 4 * (e & 7);
    s_r[i] = e >> 3;
  }
  v4 ra[2], rb[2];
  auto fetch = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kg = kt + s_k[i];
      const int k0 = kg / p.K1, k1 = kg - k0 * p.K1;
      const bool ka = kg < kend;
      const bool oa = ka && m0 + s_r[i] < p.M, ob = ka && n0 + s_r[i] < p.N;
      const v4 va = *reinterpret_cast<const v4 *>(oa ? A + (m0 + s_r[i]) * p.sAm + k0 * p.sAk0 + k1 : p.A);
      const v4 vb = *reinterpret_cast<const v4 *>(ob ? B + (n0 + s_r[i]) * p.sBn + k0 * p.sBk0 + k1 : p.B);
      ra[i] = oa ? va : v4{0.f, 0.f, 0.f, 0.f};
      rb[i] = ob ? vb : v4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto split4 = [](const v4 v, h4 &hi, h4 &lo) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const _Float16 h = (_Float16)v[t];
      hi[t] = h;
      lo[t] = (_Float16)((v[t] - (float)h) * kF16LoScale);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const f16x8 down = {(_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale),
                      (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale)};
  const int ro = (lane & 31) * kGH + 8 * (lane >> 5);  // this lane's operand row and k group
  if (kbeg < kend) fetch(kbeg);
  for (int kt = kbeg; kt < kend; kt += kGKF) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      h4 hi, lo;
      split4(ra[i], hi, lo);
      *reinterpret_cast<h4 *>(&Ah[0][s_r[i] * kGH + s_k[i]]) = hi;
      *reinterpret_cast<h4 *>(&Ah[1][s_r[i] * kGH + s_k[i]]) = lo;
      split4(rb[i], hi, lo);
      *reinterpret_cast<h4 *>(&Bh[0][s_r[i] * kGH + s_k[i]]) = hi;
      *reinterpret_cast<h4 *>(&Bh[1][s_r[i] * kGH + s_k[i]]) = lo;
    }
    __syncthreads();
    if (kt + kGKF < kend) fetch(kt + kGKF);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f16x8 ah = *reinterpret_cast<const f16x8 *>(&Ah[0][wm * kGH + ro + 16 * ks]);
      const f16x8 al = *reinterpret_cast<const f16x8 *>(&Ah[1][wm * kGH + ro + 16 * ks]);
      const f16x8 bh = *reinterpret_cast<const f16x8 *>(&Bh[0][wn * kGH + ro + 16 * ks]);
      const f16x8 bl = *reinterpret_cast<const f16x8 *>(&Bh[1][wn * kGH + ro + 16 * ks]);
      const f16x8 as = ah * down, bs = bh * down;
      DDPM_MFMA_F16X3(acc, ah, al, as, bh, bl, bs);
    }
  }
  const int n = n0 + wn + (lane & 31);
  if (p.S > 1) {
    float *out = p.part + (size_t)blockIdx.z * p.M * p.N;
    if (n < p.N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (m < p.M) out[(size_t)m * p.N + n] = acc[i];
      }
    }
    return;
  }
  if (n < p.N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
      if (m < p.M) {
        float *c = C + m * p.sCm + n * p.sCn;
        const float v = p.alpha * acc[i];
        *c = p.beta != 0.f ? v + p.beta * *c : v;
      }
    }
  }
}

__global__ void gemm_splitk_reduce_kernel(const GemmP p) {
  const size_t mn = (size_t)p.M * p.N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn * p.Z) return;
  const int z = (int)(i / mn);
  const size_t r = i - (size_t)z * mn;
  const int m = (int)(r / p.N), n = (int)(r - (size_t)m * p.N);
  const float *src = p.part + (size_t)z * p.S * mn + r;
  float sum = 0.f;
  for (int k = 0; k < p.S; ++k) sum += src[(size_t)k * mn];
  const int z0 = z / p.Z1, z1 = z - z0 * p.Z1;
  float *c = p.C + z0 * p.sCz0 + z1 * p.sCz1 + m * p.sCm + n * p.sCn;
  const float v = p.alpha * sum;
  *c = p.beta != 0.f ? v + p.beta * *c : v;
}

// ------------------------------------------------------------------------------------------------ 3x3 weight gradient
constexpr int kWT = 64;  // couts and cins per workgroup

struct WgradP {
  const float *a;   // [B, Cin, Hi, Wi]
  const float *dy;  // [B, Cout, Ho, Wo]
  float *part;      // [S][9][Cout][Cin]
  int B, Cin, Cout, Hi, Wi, Ho, Wo, stride;
  // 3-D (conv3d weight gradient, one launch per depth tap kd): an "image" is one (batch item, output slice zo); its input slice is
  // zi = stride zo + kd - 1 (outside the volume: the tile contributes nothing).  2-D: Do = Di = 1, kd = 1.
  int Di, Do, kd;
  long long a_cs, dy_cs;  // channel strides of a / dy in floats (Di Hi Wi / Do Ho Wo)
  int R;            // output rows per pixel tile
  int tiles_per_img, T, S;  // T = B * tiles_per_img pixel tiles, walked by S workgroups per (cout, cin) block
  int AR, AW, ACS;  // input tile: rows, row length (with halo), channel stride (odd)
  int DCS;          // dY tile channel stride (odd)
  // staged form (Wi % 4 == 0, Wo % 4 == 0, 16-byte aligned tensors): 16-byte global loads into registers one pixel tile ahead
  int fast;
  int LPR, RPI, NA;  // lanes per input row (Wi / 4), (channel, row) pairs per wave instruction, staging iterations of the input tile
  int LPD, CPI, ND;  // lanes per dY channel (PT / 4), channels per wave instruction, staging iterations of the dY tile
  // split-f16 form (stride 1, W a power of two in 8 .. 64, 64-pixel tiles): f16 LDS tiles, v_mfma_f32_32x32x16_f16
  int h16;                // the form applies
  int ACSh;               // input tile channel stride in halves ((R + 2) W + 8: a multiple of 8 with an odd 16-byte count)
  int DCSh;               // dY tile channel stride in halves (R rows of W + 8, an odd 16-byte count)
  int DHh, LDSh;          // halves of the dY tiles (8 in front + two planes) and of the whole LDS image
  int lw;                 // log2 W
  int nw;                 // waves per workgroup of the split-f16 form (4: 64 couts per workgroup, 8: 128)
  // float bits of partial maxima of |dY| (amax_nd of them) and of |a| (amax_na): wgrad_absmax_kernel's, or the caller's (the
  // kernels that wrote the two tensors -- ddpm_gn_backward_f32 / ddpm_gn_forward_f32 -- emit them for free)
  const unsigned *amax_d, *amax_a;
  int amax_nd, amax_na;
};

constexpr int kNA = 12, kND = 4;  // most staging iterations per thread (register arrays)

template <int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradP p) {
  extern __shared__ float smem[];
  float *Ds = smem;                   // [64 co][DCS]
  float *As = smem + kWT * p.DCS;     // [64 ci][ACS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cob = blockIdx.x * kWT, cib = blockIdx.y * kWT, sp = blockIdx.z;
  const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
  const int PT = p.R * p.Wo;  // pixels per tile (even)
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const size_t hw_i = (size_t)p.Hi * p.Wi, hw_o = (size_t)p.Ho * p.Wo;
  for (int tile = sp; tile < p.T; tile += p.S) {
    const int sl = tile / p.tiles_per_img, rb = tile - sl * p.tiles_per_img;
    const int b = sl / p.Do, zo = sl - b * p.Do, zi = STRIDE * zo + p.kd - 1;
    if (zi < 0 || zi >= p.Di) continue;  // (uniform over the workgroup)
    const int yo0 = rb * p.R;
    const int rows = min(p.R, p.Ho - yo0);  // (a ragged last tile: the missing rows are zero)
    // ---- dY tile: 64 couts x PT pixels (contiguous in memory)
    const float *dyb = p.dy + ((size_t)b * p.Cout + cob) * p.dy_cs + (size_t)zo * hw_o + (size_t)yo0 * p.Wo;
    for (int e = tid; e < kWT * PT; e += 256) {
      const int c = e / PT, px = e - c * PT;
      Ds[c * p.DCS + px] = px < rows * p.Wo ? dyb[(size_t)c * p.dy_cs + px] : 0.f;
    }
    // ---- input tile: 64 cins x AR rows x AW columns, zero halo; tile row r holds input row STRIDE yo0 - 1 + r
    const float *ab = p.a + ((size_t)b * p.Cin + cib) * p.a_cs + (size_t)zi * hw_i;
    const int yi0 = STRIDE * yo0 - 1;
    for (int e = tid; e < kWT * p.AR * p.AW; e += 256) {
      const int c = e / (p.AR * p.AW), rem = e - c * (p.AR * p.AW);
      const int r = rem / p.AW, col = rem - r * p.AW;
      const int yi = yi0 + r, xi = col - 1;
      float v = 0.f;
      if (yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi) v = ab[(size_t)c * p.a_cs + (size_t)yi * p.Wi + xi];
      As[c * p.ACS + r * p.AW + col] = v;
    }
    __syncthreads();
    const float *dsw = Ds + (wco + l31) * p.DCS + lhi;
    const float *asw = As + (wci + l31) * p.ACS;
    for (int px = 0; px < PT; px += 2) {
      const int pp = px + lhi;
      const int yo = pp / p.Wo, xo = pp - yo * p.Wo;
      const float a = dsw[px];
      const float *ap = asw + (STRIDE * yo) * p.AW + STRIDE * xo;  // tap (ky, kx): + ky AW + kx
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float bv = ap[(t / 3) * p.AW + (t % 3)];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // partial sums: part[sp][tap][co][ci] (ci fastest: coalesced)
  float *out = p.part + (size_t)sp * 9 * p.Cout * p.Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cob + wco + 8 * (i >> 2) + 4 * lhi + (i & 3);
      out[((size_t)t * p.Cout + co) * p.Cin + cib + wci + l31] = acc[t][i];
    }
}

// The same contraction with the staging rebuilt (the form above spent more time computing addresses for its 4-byte staging loads
// than multiplying: 33 TFLOP/s): every (channel, input row) of the tile is fetched by 16-byte loads -- Wi / 4 lanes per row --
// into registers ONE PIXEL TILE AHEAD, so the global-memory latency of tile i + 1 hides behind the 288 MFMAs per wave of tile i;
// LDS offsets and global offsets of a thread's loads are computed once per kernel, the K loop walks rows and pixel pairs
// without a division.  The halo columns of the input tile are zeroed once (no load ever writes them), rows outside the image
// are written as zeros per tile.
// (Measured and not kept: the same kernel WITHOUT the register set for the next tile, 256 registers per lane and two workgroups per
// CU, the second hiding the first's load latency -- 2 970-2 993 against 3 085-3 095 images/s at batch 64, 3 811 against 3 969 at 256:
// profiles/r06_wgrad_two_workgroups_per_cu_ab.log.)
template <int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_wgrad_staged_kernel(const WgradP p) {
  extern __shared__ float smem[];
  float *Ds = smem;                   // [64 co][DCS]
  float *As = smem + kWT * p.DCS;     // [64 ci][ACS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cob = blockIdx.x * kWT, cib = blockIdx.y * kWT, sp = blockIdx.z;
  const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
  const int hw_i = p.Hi * p.Wi, hw_o = p.Ho * p.Wo;
  // ---- this thread's staging slots
  int a_lds[kNA], a_g[kNA], a_r[kNA];
#pragma unroll
  for (int it = 0; it < kNA; ++it) {
    const int q = (it * 4 + wave) * p.RPI + lane / p.LPR, lc = lane % p.LPR;
    const bool ok = it < p.NA && q < kWT * p.AR && lane < p.RPI * p.LPR;
    const int c = q / p.AR, r = q - c * p.AR;
    a_lds[it] = c * p.ACS + r * p.AW + 1 + 4 * lc;
    a_g[it] = c * (int)p.a_cs + r * p.Wi + 4 * lc;
    a_r[it] = ok ? r : -(1 << 20);
  }
  int d_lds[kND], d_g[kND], d_px[kND];
#pragma unroll
  for (int it = 0; it < kND; ++it) {
    const int c = (it * 4 + wave) * p.CPI + lane / p.LPD, px0 = 4 * (lane % p.LPD);
    const bool ok = it < p.ND && c < kWT && lane < p.CPI * p.LPD;
    d_lds[it] = c * p.DCS + px0;
    d_g[it] = c * (int)p.dy_cs + px0;
    d_px[it] = ok ? px0 : (1 << 20);
  }
  for (int e = tid; e < kWT * p.ACS; e += 256) As[e] = 0.f;  // the halo columns stay zero for the whole kernel
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 ra[kNA], rd[kND];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int sl = tile / p.tiles_per_img, rb = tile - sl * p.tiles_per_img;
    const int b = sl / p.Do, zo = sl - b * p.Do, zi = STRIDE * zo + p.kd - 1;
    const bool zok = zi >= 0 && zi < p.Di;  // an input slice outside the volume: the whole tile is zero
    const int yo0 = rb * p.R, yi0 = STRIDE * yo0 - 1;
    const int rows_px = zok ? min(p.R, p.Ho - yo0) * p.Wo : 0;
    const float *ab = p.a + ((size_t)b * p.Cin + cib) * p.a_cs + (ptrdiff_t)(zok ? zi : 0) * hw_i + (ptrdiff_t)yi0 * p.Wi;
    const float *dyb = p.dy + ((size_t)b * p.Cout + cob) * p.dy_cs + (size_t)zo * hw_o + (size_t)yo0 * p.Wo;
#pragma unroll
    for (int it = 0; it < kNA; ++it) {
      const int yi = yi0 + a_r[it];
      ra[it] = zok && yi >= 0 && yi < p.Hi ? *reinterpret_cast<const v4 *>(ab + a_g[it]) : v4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < kND; ++it)
      rd[it] = d_px[it] < rows_px ? *reinterpret_cast<const v4 *>(dyb + d_g[it]) : v4{0.f, 0.f, 0.f, 0.f};
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const float *dsw = Ds + (wco + l31) * p.DCS + lhi;
  const float *asw = As + (wci + l31) * p.ACS + STRIDE * lhi;
  const int aw = p.AW, aw2 = 2 * p.AW;
  if (sp < p.T) fetch(sp);
  for (int tile = sp; tile < p.T; tile += p.S) {
    __syncthreads();  // the previous tile's MFMAs have read their operands (first pass: the zeroing above is done)
#pragma unroll
    for (int it = 0; it < kNA; ++it)
      if (a_r[it] >= 0) {
        float *o = As + a_lds[it];
        o[0] = ra[it][0]; o[1] = ra[it][1]; o[2] = ra[it][2]; o[3] = ra[it][3];
      }
#pragma unroll
    for (int it = 0; it < kND; ++it)
      if (d_px[it] < (1 << 20)) {
        float *o = Ds + d_lds[it];
        o[0] = rd[it][0]; o[1] = rd[it][1]; o[2] = rd[it][2]; o[3] = rd[it][3];
      }
    __syncthreads();
    if (tile + p.S < p.T) fetch(tile + p.S);  // in flight while this tile multiplies
    for (int yo = 0; yo < p.R; ++yo) {
      const float *drow = dsw + yo * p.Wo;
      const float *arow = asw + (STRIDE * yo) * aw;
#pragma unroll 2
      for (int xo = 0; xo < p.Wo; xo += 2) {
        const float a = drow[xo];
        const float *ap = arow + STRIDE * xo;
        const float b0 = ap[0], b1 = ap[1], b2 = ap[2], b3 = ap[aw], b4 = ap[aw + 1], b5 = ap[aw + 2], b6 = ap[aw2], b7 = ap[aw2 + 1],
                    b8 = ap[aw2 + 2];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b4, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b5, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b6, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b7, acc[7], 0, 0, 0);
        acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b8, acc[8], 0, 0, 0);
      }
    }
  }
  float *out = p.part + (size_t)sp * 9 * p.Cout * p.Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cob + wco + 8 * (i >> 2) + 4 * lhi + (i & 3);
      out[((size_t)t * p.Cout + co) * p.Cin + cib + wci + l31] = acc[t][i];
    }
}

// ---- the split-f16 form -------------------------------------------------------------------------------------------
// The same contraction on v_mfma_f32_32x32x16_f16 at split precision (common.h: a b ~= ah bh + (ah 2^-5)(bl 2^5) + (al 2^5)(bh 2^-5),
// fp32 accumulate), 16 pixels per instruction instead of 2: per 64-pixel tile a wave issues 4 x 27 MFMAs of 8 passes where the fp32
// form issues 32 x 9 of 16 -- 3 456 against 18 432 matrix-pipe cycles.
// K runs over INPUT pixels q = (y, x + kx - 1), so that the eight consecutive k of a lane are eight consecutive, 16-byte-aligned
// halves of one input row for every tap: dW[co][ci][ky][kx] = sum_{y, q} dY[co][y][q - kx + 1] a[ci][y + ky - 1][q].  The ky shift is
// a row offset into the input tile (rows y - 1 .. y + R, no column halo: q is a real pixel).  The kx shift sits on the dY side and is
// made IN REGISTERS: a lane reads its aligned eight halves dY[q .. q + 7] plus the words holding dY[q - 1] and dY[q + 8], and
// v_alignbit_b32 slides them by one half for kx = 2 / kx = 0.  dY rows are stored W + 8 halves apart, the 8 spare halves zero, so
// that the neighbour words of a row's first / last lane are the zeros the convolution's padding asks for.
// (First form, round 6: three pre-shifted copies of the dY tile in LDS -- 14 LDS stores per staged quad instead of 2, 90-106 KB
// per workgroup, one workgroup per CU: 193-265 TFLOP/s on the step's shapes, profiles/r06_wgrad_f16x3_ablations.log; staging was a
// third of the kernel, and nothing overlapped it.  This form fits two workgroups per CU.)
// Both operands are scaled by a power of two chosen from their largest magnitude (wgrad_absmax_kernel, read from device memory: no
// host round trip) so that the high halves sit at 2^13 .. 2^14 and the scaled low halves stay normal for every element within 2^-21
// of the maximum; the accumulators are scaled back as they are written.
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

constexpr int kAmaxBlocks = 512;  // most partial maxima per operand (two per thread of the consumer)

// partial maxima of |x| as float bit patterns, one per workgroup (no atomics: 8 192 waves on one address cost more than the read)
__global__ __launch_bounds__(256) void wgrad_absmax_kernel(const float *__restrict__ x, size_t n4, unsigned *__restrict__ out) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __shared__ unsigned red[4];
  // (NaN: fmaxf would drop it -- keep the bit pattern, which orders above every finite value)
  unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const u4 *x4 = reinterpret_cast<const u4 *>(x);
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight
    const u4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    m0 = max(m0, max(max(a[0] & 0x7fffffffu, a[1] & 0x7fffffffu), max(a[2] & 0x7fffffffu, a[3] & 0x7fffffffu)));
    m1 = max(m1, max(max(b[0] & 0x7fffffffu, b[1] & 0x7fffffffu), max(b[2] & 0x7fffffffu, b[3] & 0x7fffffffu)));
    m2 = max(m2, max(max(c[0] & 0x7fffffffu, c[1] & 0x7fffffffu), max(c[2] & 0x7fffffffu, c[3] & 0x7fffffffu)));
    m3 = max(m3, max(max(d[0] & 0x7fffffffu, d[1] & 0x7fffffffu), max(d[2] & 0x7fffffffu, d[3] & 0x7fffffffu)));
  }
  for (; i < n4; i += stride) {
    const u4 a = x4[i];
    m0 = max(m0, max(max(a[0] & 0x7fffffffu, a[1] & 0x7fffffffu), max(a[2] & 0x7fffffffu, a[3] & 0x7fffffffu)));
  }
  unsigned b = max(max(m0, m1), max(m2, m3));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

// 2^(13 - floor(log2 max)): max scale in [2^13, 2^14); 1 for an all-zero tensor
__device__ __forceinline__ float wgrad_scale_of(unsigned maxbits) {
  const int e = (int)((maxbits >> 23) & 255);
  if (maxbits == 0) return 1.f;
  const int se = min(254, max(1, 267 - e));
  return __uint_as_float((unsigned)se << 23);
}

__device__ __forceinline__ void wgrad_split4(const float (&v)[4], h4v &hi, h4v &lo) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)((v[t] - (float)h) * kF16LoScale);
  }
}

// Pipeline: the next tile's quads are fetched into registers while this tile multiplies, converted and stored between the tiles.
// (Measured, same shapes, batch 256, weighted by the step's counts -- profiles/r06_wgrad_f16x3_variants.log: three pre-shifted dY
// copies in LDS 6.78 ms; register-shifted dY 6.63 ms, with branch-free fetch / staging (this form) 5.95 ms; without the register
// prefetch at two workgroups per CU 7.05 ms; TWO LDS images
// with the conversion of tile i + 1 and the loads of tile i + 2 issued between the MFMAs of tile i, one barrier per tile 7.49 ms --
// with one wave per SIMD an MFMA hides about five other issues (MI355X_MICROARCH.md) and the 650 of a tile do not fit behind 108.
// A second register set (two tiles of loads in flight) 6.95 ms.  Ablations of this form on 4 waves (base 6.13 on that box): no MFMAs
// 6.08, no conversion + LDS stores 5.52, no global loads 4.63 -- the matrix pipe is hidden, the waves wait for their loads (49 % of
// wave cycles waiting, L2 hit rate 0.37, profiles/r06_pmc_wgrad_f16x3.log), and more loads in flight do not help; what helps is a
// second wave per SIMD with half the input traffic per MFMA: the 8-wave form below, 5.24 ms (its ablations: 3.24 / 3.90 / 4.08).)
// NA: staging slots of the input tile per thread (5, 6, 8, 12 for W = 8, 16, 32, 64: 64 (R + 2) (channel, row) pairs of W / 4
// quads over 256 threads, exactly -- with W a power of two and R W = 64 every slot of every lane is a real quad, so the staging code
// carries no branch and the scheduler is free to move it between the MFMAs)
// NW: waves per workgroup.  4: 64 couts x 64 cins, one wave per SIMD.  8: 128 couts x 64 cins (wave w: couts 32 (w / 2) .., cins
// 32 (w % 2) ..), two waves per SIMD and at most 256 registers each -- the input tile, which every cout block of a layer re-reads
// and re-converts, is staged once per 128 couts instead of once per 64, and a wave's load wait overlaps its SIMD partner's MFMAs.
template <int NA, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_wgrad_f16x3_kernel(const WgradP p) {
  constexpr int NT = 64 * NW, CO = 16 * NW;  // threads, couts per workgroup
  extern __shared__ float smem[];
  _Float16 *Dh = reinterpret_cast<_Float16 *>(smem) + 8;  // [hi, lo][CO co][DCSh]: R rows of W + 8 halves; 8 zero halves in front
  _Float16 *Ah = Dh - 8 + p.DHh;                           // [hi, lo][64 ci][ACSh]: rows y0 - 1 .. y0 + R of W pixels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // (An XCD-aware decode of the workgroup id -- the (cout, cin) blocks of one slice, which read the same tiles at the same time,
  // on one XCD's L2 -- measured +-0: 5.95 against 5.98 ms over the step's shapes.)
  const int cob = blockIdx.x * CO, cib = blockIdx.y * kWT, sp = blockIdx.z;
  const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
  const int hw_i = p.Hi * p.Wi, hw_o = p.Ho * p.Wo;
  const int W = p.Wo, ACS = p.ACSh, DCS = p.DCSh, lw = p.lw;
  const int aplane = kWT * ACS, dplane = CO * DCS;
  // ---- this thread's staging slots.  (channel, input row) pair q = it 4 RPI + (wave RPI + lane / LPR) -> row q / 64, channel
  // q % 64 (NW RPI pairs per step: 16 .. 256), so the `it` part of both is uniform (scalar registers) and only the lane part below lives in
  // vector registers -- the staged fp32 form's per-slot arrays were 48 of them, and this kernel has 256 to fit two workgroups per CU
  const int ql = wave * p.RPI + lane / p.LPR, lc = lane % p.LPR;
  const int a_c0 = ql & 63, a_r0 = ql >> 6;  // (a_r0 > 0 when a step covers more than 64 pairs)
  const int a_lds0 = a_c0 * ACS + a_r0 * W + 4 * lc, a_g0 = a_c0 * (int)p.a_cs + a_r0 * p.Wi + 4 * lc;
  const int qstep = NW * p.RPI;
  // dY: channel it 4 CPI + (wave CPI + lane / LPD), pixels 4 (lane % LPD) ..
  const int d_c0 = wave * p.CPI + lane / p.LPD, d_px0 = 4 * (lane % p.LPD);
  const int d_lds0 = d_c0 * DCS + (d_px0 >> lw) * (W + 8) + (d_px0 & (W - 1)), d_g0 = d_c0 * (int)p.dy_cs + d_px0;
  const int cstep = NW * p.CPI;
  {
    const int n16 = p.LDSh / 8;  // (a multiple of 8 halves)
    uint4 *z = reinterpret_cast<uint4 *>(smem);
    for (int e = tid; e < n16; e += NT) z[e] = uint4{0u, 0u, 0u, 0u};
  }
  float sD, sA;
  {  // the operand maxima from their per-workgroup parts
    unsigned md = 0, ma = 0;
    for (int e = tid; e < p.amax_nd; e += NT) md = max(md, p.amax_d[e]);
    for (int e = tid; e < p.amax_na; e += NT) ma = max(ma, p.amax_a[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      md = max(md, (unsigned)__shfl_xor((int)md, o, 64));
      ma = max(ma, (unsigned)__shfl_xor((int)ma, o, 64));
    }
    unsigned *red = reinterpret_cast<unsigned *>(smem);
    __syncthreads();  // (the zeroing above is complete before its first words are borrowed ...)
    if (lane == 0) { red[wave] = md; red[NW + wave] = ma; }
    __syncthreads();
    md = ma = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      md = max(md, red[w]);
      ma = max(ma, red[NW + w]);
    }
    __syncthreads();
    if (tid < 2 * NW) red[tid] = 0u;  // (... and they are zero again before the first tile is staged: the loop opens with a barrier)
    sD = wgrad_scale_of(md);
    sA = wgrad_scale_of(ma);
  }
  const float inv = 1.f / (sD * sA);
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 ra[NA], rd[kND];
  // the tile whose quads the fetch_* below load: base pointers and row limits (uniform)
  const float *f_ab = p.a, *f_dyb = p.dy;
  int f_yi0 = 0, f_rows_px = 0;
  bool f_zok = false;
  auto aim = [&](int tile) __attribute__((always_inline)) {
    const int sl = tile / p.tiles_per_img, rb = tile - sl * p.tiles_per_img;
    const int b = sl / p.Do, zo = sl - b * p.Do, zi = zo + p.kd - 1;
    f_zok = zi >= 0 && zi < p.Di;
    const int yo0 = rb * p.R;
    f_yi0 = yo0 - 1;
    f_rows_px = f_zok ? min(p.R, p.Ho - yo0) * p.Wo : 0;
    f_ab = p.a + ((size_t)b * p.Cin + cib) * p.a_cs + (ptrdiff_t)(f_zok ? zi : 0) * hw_i + (ptrdiff_t)f_yi0 * p.Wi;
    f_dyb = p.dy + ((size_t)b * p.Cout + cob) * p.dy_cs + (size_t)zo * hw_o + (size_t)yo0 * p.Wo;
  };
  // (rows outside the image / the volume: the load goes to the tensor's first quad and a select zeroes it -- no branch)
  auto fetch_a = [&](int it) __attribute__((always_inline)) {
    if (it >= NA) return;  // (compile time)
    const int qu = it * qstep, ru = qu >> 6, cu = qu & 63;  // (uniform)
    const int yi = f_yi0 + ru + a_r0;
    const bool ok = f_zok && yi >= 0 && yi < p.Hi;
    const v4 v = *reinterpret_cast<const v4 *>(ok ? f_ab + a_g0 + cu * (int)p.a_cs + ru * p.Wi : p.a);
    ra[it] = ok ? v : v4{0.f, 0.f, 0.f, 0.f};
  };
  auto fetch_d = [&](int it) __attribute__((always_inline)) {
    const bool ok = d_px0 < f_rows_px;
    const v4 v = *reinterpret_cast<const v4 *>(ok ? f_dyb + d_g0 + it * cstep * (int)p.dy_cs : p.dy);
    rd[it] = ok ? v : v4{0.f, 0.f, 0.f, 0.f};
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const _Float16 *dsw = Dh + (wco + l31) * DCS;
  const _Float16 *asw = Ah + (wci + l31) * ACS + 8 * lhi;
  const f16x8 down = {(_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale),
                      (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale), (_Float16)(1.f / kF16LoScale)};
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  // the aligned eight halves m = dY[q .. q + 7] with the word left of them (dY[q - 2], dY[q - 1]) / right of them (dY[q + 8], dY[q + 9])
  // -> dY[q + 1 .. q + 8] (kx = 0) and dY[q - 1 .. q + 6] (kx = 2)
  auto slide_up = [](const f16x8 m, unsigned right) __attribute__((always_inline)) {
    const u4v w = __builtin_bit_cast(u4v, m);
    const u4v r = {__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                   __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(right, w[3], 16)};
    return __builtin_bit_cast(f16x8, r);
  };
  auto slide_down = [](const f16x8 m, unsigned left) __attribute__((always_inline)) {
    const u4v w = __builtin_bit_cast(u4v, m);
    const u4v r = {__builtin_amdgcn_alignbit(w[0], left, 16), __builtin_amdgcn_alignbit(w[1], w[0], 16),
                   __builtin_amdgcn_alignbit(w[2], w[1], 16), __builtin_amdgcn_alignbit(w[3], w[2], 16)};
    return __builtin_bit_cast(f16x8, r);
  };
  // staging slot `it` of the tile in ra / rd -> LDS
  auto stage_a = [&](int it) __attribute__((always_inline)) {
    if (it >= NA) return;  // (compile time)
    const int qu = it * qstep, ru = qu >> 6, cu = qu & 63;
    const float v[4] = {ra[it][0] * sA, ra[it][1] * sA, ra[it][2] * sA, ra[it][3] * sA};
    h4v hi, lo;
    wgrad_split4(v, hi, lo);
    _Float16 *o = Ah + a_lds0 + cu * ACS + ru * W;
    *reinterpret_cast<h4v *>(o) = hi;
    *reinterpret_cast<h4v *>(o + aplane) = lo;
  };
  auto stage_d = [&](int it) __attribute__((always_inline)) {
    const float v[4] = {rd[it][0] * sD, rd[it][1] * sD, rd[it][2] * sD, rd[it][3] * sD};
    h4v hi, lo;
    wgrad_split4(v, hi, lo);
    _Float16 *o = Dh + d_lds0 + it * cstep * DCS;
    *reinterpret_cast<h4v *>(o) = hi;
    *reinterpret_cast<h4v *>(o + dplane) = lo;
  };
  if (sp < p.T) {
    aim(sp);
#pragma unroll
    for (int it = 0; it < NA; ++it) fetch_a(it);
#pragma unroll
    for (int it = 0; it < kND; ++it) fetch_d(it);
  }
  for (int tile = sp; tile < p.T; tile += p.S) {
    __syncthreads();  // the previous tile's MFMAs have read their operands (first pass: the zeroing above is done)
#pragma unroll
    for (int it = 0; it < NA; ++it) stage_a(it);
#pragma unroll
    for (int it = 0; it < kND; ++it) stage_d(it);
    __syncthreads();
    aim(tile + p.S < p.T ? tile + p.S : tile);  // in flight while this tile multiplies (no next tile: a tile nobody stages)
#pragma unroll
    for (int it = 0; it < NA; ++it) fetch_a(it);
#pragma unroll
    for (int it = 0; it < kND; ++it) fetch_d(it);
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // 16 pixels per step: this lane's k are pixels 16 s + 8 lhi .. + 7 of the tile
      const int px = 16 * s + 8 * lhi;
      const _Float16 *dp = dsw + (px >> lw) * (W + 8) + (px & (W - 1));
      f16x8 ah[3], al[3], as[3];
      {
        const f16x8 mh = *reinterpret_cast<const f16x8 *>(dp), ml = *reinterpret_cast<const f16x8 *>(dp + dplane);
        const unsigned hl = *reinterpret_cast<const unsigned *>(dp - 2), hr = *reinterpret_cast<const unsigned *>(dp + 8);
        const unsigned ll = *reinterpret_cast<const unsigned *>(dp + dplane - 2), lr = *reinterpret_cast<const unsigned *>(dp + dplane + 8);
        ah[0] = slide_up(mh, hr);   al[0] = slide_up(ml, lr);
        ah[1] = mh;                 al[1] = ml;
        ah[2] = slide_down(mh, hl); al[2] = slide_down(ml, ll);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) as[k] = ah[k] * down;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f16x8 bh = *reinterpret_cast<const f16x8 *>(asw + ky * W + 16 * s);  // input row y + ky - 1 = tile row y + ky
        const f16x8 bl = *reinterpret_cast<const f16x8 *>(asw + aplane + ky * W + 16 * s);
        const f16x8 bs = bh * down;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) DDPM_MFMA_F16X3(acc[3 * ky + kx], ah[kx], al[kx], as[kx], bh, bl, bs);
      }
    }
  }
  float *out = p.part + (size_t)sp * 9 * p.Cout * p.Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cob + wco + 8 * (i >> 2) + 4 * lhi + (i & 3);
      out[((size_t)t * p.Cout + co) * p.Cin + cib + wci + l31] = acc[t][i] * inv;
    }
}

// dW[co][ci][tap] = sum over the S partial slabs, slab order fixed
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, int S, int Cout, int Cin, int taps,
                                    int out_taps, int out_off) {
  const size_t cc_n = (size_t)Cout * Cin, n = cc_n * taps;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // consecutive threads walk (cout, cin) inside one tap: the S reads are contiguous, the one write is strided
  const int tap = (int)(i / cc_n);
  const size_t cc = i - (size_t)tap * cc_n;  // co * Cin + ci
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[((size_t)k * taps + tap) * cc_n + cc];
  dw[cc * out_taps + out_off + tap] = s;  // (out_taps = 27, out_off = 9 kd: one depth tap of a [Cout, Cin, 3, 3, 3] weight)
}

// Any (Cout, Cin), ksize 1 or 3, stride 1 or 2, padding ksize / 2: one workgroup per (cout, cin) pair and image slice, its
// ksize^2 taps summed over the slice's images and output pixels in a fixed order (the first / last convolution of the UNet -- 1 or
// 3 channels on one side -- and the reference of the MFMA kernel's tests).  gridDim.y > 1: image slices, partial sums to
// part[slice][tap][cout][cin] for wgrad_reduce_kernel (128 workgroups walking 65 536 pixels each were 0.4 ms per launch).
__global__ __launch_bounds__(256) void conv_wgrad_generic_kernel(const float *__restrict__ a, const float *__restrict__ dy,
                                                                 float *__restrict__ dw, float *__restrict__ part, int B, int Cin,
                                                                 int Cout, int Hi, int Wi, int Ho, int Wo, int k, int stride) {
  __shared__ float red[4];
  const int co = blockIdx.x / Cin, ci = blockIdx.x - co * Cin;
  const int pad = k / 2, taps = k * k;
  const int bper = (B + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  const int npx = Ho * Wo;
  for (int e = threadIdx.x; e < (b1 - b0) * npx; e += 256) {
    const int b = b0 + e / npx, px = e % npx;
    const int yo = px / Wo, xo = px - yo * Wo;
    const float g = dy[((size_t)b * Cout + co) * npx + px];
    const float *ap = a + ((size_t)b * Cin + ci) * Hi * Wi;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t < taps) {
        const int yi = yo * stride + t / k - pad, xi = xo * stride + t % k - pad;
        if (yi >= 0 && yi < Hi && xi >= 0 && xi < Wi) s[t] = __builtin_fmaf(g, ap[yi * Wi + xi], s[t]);
      }
    }
  }
  for (int t = 0; t < taps; ++t) {
    const float v = block_sum_256(s[t], red);
    if (threadIdx.x == 0) {
      if (gridDim.y > 1) part[(((size_t)blockIdx.y * taps + t) * Cout + co) * Cin + ci] = v;
      else dw[((size_t)co * Cin + ci) * taps + t] = v;
    }
  }
}

}  // namespace

}  // namespace ddpm

using namespace ddpm;

namespace {
// K slices of a product whose (M, N, batch) grid leaves most of the chip idle (a 1x1 convolution's weight gradient: 256 x 512
// outputs, K = images x pixels = 65 536): enough slices for two workgroups per CU, at least 64 K per slice
int gemm_ksplit(const ddpm_gemm_desc *g, int &kchunk) {
  const long wgs = (long)((g->N + kGT - 1) / kGT) * ((g->M + kGT - 1) / kGT) * g->batch;
  const long want = 2L * device_cus();
  kchunk = g->K;
  if (wgs >= want || g->K < 256) return 1;
  long S = (want + wgs - 1) / wgs;
  const long smax = g->K / 64;
  if (S > smax) S = smax;
  if (S * g->batch > 65535) S = 65535 / g->batch;
  if (S < 2) return 1;
  long kc = (g->K + S - 1) / S;
  kc = (kc + kGKF - 1) / kGKF * kGKF;
  S = (g->K + kc - 1) / kc;
  kchunk = (int)kc;
  return S < 2 ? 1 : (int)S;
}
}  // namespace

extern "C" size_t ddpm_gemm_scratch_floats(const ddpm_gemm_desc *g) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0 || g->batch <= 0) return 0;
  int kchunk;
  const int S = gemm_ksplit(g, kchunk);
  return S > 1 ? (size_t)S * g->batch * g->M * g->N : 0;
}

extern "C" int ddpm_gemm_f32(const ddpm_gemm_desc *g, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(g && g->A && g->B && g->C, "gemm: null operand");
  DDPM_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0 && g->batch > 0, "gemm: empty extent");
  GemmP p;
  p.A = g->A; p.B = g->B; p.C = g->C;
  p.M = g->M; p.N = g->N; p.K = g->K;
  p.K1 = g->k_inner > 0 ? g->k_inner : g->K;
  DDPM_CHECK_ARG(p.K % p.K1 == 0, "gemm: K %d is not a multiple of k_inner %d", p.K, p.K1);
  p.sAm = g->a_m; p.sAk0 = g->a_k_outer; p.sAk1 = g->a_k;
  p.sBn = g->b_n; p.sBk0 = g->b_k_outer; p.sBk1 = g->b_k;
  p.sCm = g->c_m; p.sCn = g->c_n;
  p.Z = g->batch;
  p.Z1 = g->batch_inner > 0 ? g->batch_inner : g->batch;  // one level: z1 = z, the *_batch strides
  DDPM_CHECK_ARG(p.Z % p.Z1 == 0, "gemm: batch %d is not a multiple of batch_inner %d", p.Z, p.Z1);
  DDPM_CHECK_ARG(p.Z <= 65535, "gemm: batch %d beyond the grid's z extent", p.Z);
  p.sAz0 = g->a_batch_outer; p.sAz1 = g->a_batch;
  p.sBz0 = g->b_batch_outer; p.sBz1 = g->b_batch;
  p.sCz0 = g->c_batch_outer; p.sCz1 = g->c_batch;
  p.alpha = g->alpha; p.beta = g->beta;
  p.S = 1; p.kchunk = g->K; p.part = nullptr;
  int kchunk;
  const int S = gemm_ksplit(g, kchunk);
  if (S > 1 && g->scratch && g->scratch_floats >= (size_t)S * g->batch * g->M * g->N) {
    p.S = S; p.kchunk = kchunk; p.part = g->scratch;
  }
  hipStream_t s = as_stream(stream);
  const char *kname = "train_gemm_f32";
  char kshape[96];
  if (g_prof_on && sw().prof_shapes) {  // development: one profile row per product shape
    snprintf(kshape, sizeof(kshape), "train_gemm_f32|%dx%dx%d b%d S%d", g->M, g->N, g->K, g->batch, p.S);
    kname = kshape;
  }
  ProfScope prof(s, kname, 2.0 * g->M * g->N * (double)g->K * g->batch,
                 4.0 * ((double)g->M * g->K + (double)g->K * g->N + (double)g->M * g->N) * g->batch);
  dim3 grid((g->N + kGT - 1) / kGT, (g->M + kGT - 1) / kGT, g->batch * p.S);
  // 16-byte staging: each operand either K-major (unit k stride; rows, K levels and batches at multiples of four floats, four
  // consecutive k inside one k_inner run) or row-major (unit row stride, the row count a multiple of four)
  const auto quad = [](long long v) { return (v & 3) == 0; };
  const bool kq = quad(p.K1) && quad(p.K) && quad(p.kchunk);
  const int fa = (reinterpret_cast<uintptr_t>(p.A) & 15) || !quad(p.sAz0) || !quad(p.sAz1) || !quad(p.sAk0) ? 0
                 : p.sAk1 == 1 && quad(p.sAm) && kq                                                         ? 1
                 : p.sAm == 1 && quad(p.sAk1) && quad(p.M)                                                  ? 2
                                                                                                            : 0;
  const int fb = (reinterpret_cast<uintptr_t>(p.B) & 15) || !quad(p.sBz0) || !quad(p.sBz1) || !quad(p.sBk0) ? 0
                 : p.sBk1 == 1 && quad(p.sBn) && kq                                                         ? 1
                 : p.sBn == 1 && quad(p.sBk1) && quad(p.N)                                                  ? 2
                                                                                                            : 0;
  static const bool plain = getenv("DDPM_GEMM_V4") && atoi(getenv("DDPM_GEMM_V4")) == 0;  // (A/B switch)
  static const bool no_h = getenv("DDPM_GEMM_F16X3") && atoi(getenv("DDPM_GEMM_F16X3")) == 0;  // (A/B switch)
  if (fa == 1 && fb == 1 && g->split_f16 && !plain && !no_h && split_f16_on(true)) {
    hipLaunchKernelGGL(gemm_f16x3_kk_kernel, grid, dim3(256), 0, s, p);
  } else if (fa && fb && !plain) {
    if (fa == 1 && fb == 1) hipLaunchKernelGGL((gemm_f32_v4_kernel<true, true>), grid, dim3(256), 0, s, p);
    else if (fa == 1) hipLaunchKernelGGL((gemm_f32_v4_kernel<true, false>), grid, dim3(256), 0, s, p);
    else if (fb == 1) hipLaunchKernelGGL((gemm_f32_v4_kernel<false, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_v4_kernel<false, false>), grid, dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, p);
  }
  if (p.S > 1) {
    const size_t n = (size_t)g->M * g->N * g->batch;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
  }
  DDPM_CHECK_LAUNCH();
  return 0;
}

namespace {
// DDPM_WGRAD_STAGED=0: the first form of the kernel (4-byte staging loads, no prefetch) -- the A/B partner of the staged form
bool wgrad_plain_form() {
  const char *v = getenv("DDPM_WGRAD_STAGED");
  return v && atoi(v) == 0;
}
bool wgrad_plan(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int ksize, int stride, WgradP &p, int Di = 1, int Do = 1) {
  if (ksize != 3 || (stride != 1 && stride != 2) || Cin % kWT || Cout % kWT) return false;
  if (stride == 1 ? (Ho != Hi || Wo != Wi) : (Ho != (Hi + 1) / 2 || Wo != (Wi + 1) / 2)) return false;
  if (Wo > 64 || (Wo & 1)) return false;  // pixel pairs stay inside a row
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.stride = stride;
  p.Di = Di; p.Do = Do; p.kd = 1;
  p.a_cs = (long long)Di * Hi * Wi; p.dy_cs = (long long)Do * Ho * Wo;
  if ((double)Cin * p.a_cs >= 2147483648.0 || (double)Cout * p.dy_cs >= 2147483648.0) return false;  // 32-bit offsets inside a tile
  // output rows per pixel tile: 64 pixels where the tiles fit 64 KB of LDS (two workgroups per CU), else fewer rows
  p.R = 64 / Wo < Ho ? 64 / Wo : Ho;
  if (p.R < 1) p.R = 1;
  for (;; p.R = (p.R + 1) / 2) {
    p.AR = stride * (p.R - 1) + 3;
    p.AW = Wi + 2;
    p.ACS = p.AR * p.AW;
    if (!(p.ACS & 1)) p.ACS += 1;
    p.DCS = p.R * Wo + 1;
    const size_t lds = (size_t)kWT * (p.ACS + p.DCS) * sizeof(float);
    if (lds <= 64 * 1024) break;
    if (p.R == 1) {  // a single row of a wide image: one workgroup per CU
      if (lds > 150 * 1024) return false;
      break;
    }
  }
  p.tiles_per_img = (Ho + p.R - 1) / p.R;
  p.T = B * Do * p.tiles_per_img;
  p.fast = 0;
  if (Wi % 4 == 0 && Wo % 4 == 0 && Wi <= 256 && (Hi * Wi) % 4 == 0) {
    p.LPR = Wi / 4;
    p.RPI = 64 / p.LPR;
    p.NA = (kWT * p.AR + 4 * p.RPI - 1) / (4 * p.RPI);
    p.LPD = p.R * Wo / 4;
    p.CPI = 64 / p.LPD;
    p.ND = p.CPI > 0 ? (kWT + 4 * p.CPI - 1) / (4 * p.CPI) : kND + 1;
    p.fast = p.RPI > 0 && p.NA <= kNA && p.ND <= kND;
  }
  p.h16 = 0; p.nw = 4; p.ACSh = p.DCSh = p.DHh = p.LDSh = p.lw = 0; p.amax_d = p.amax_a = nullptr; p.amax_nd = p.amax_na = 0;
  if (p.fast && stride == 1 && (Wo == 8 || Wo == 16 || Wo == 32 || Wo == 64) && p.R * Wo == 64) {
    p.h16 = p.NA == (Wo == 8 ? 5 : Wo == 16 ? 6 : Wo == 32 ? 8 : 12) && p.ND == 4 && p.RPI * p.LPR == 64 && p.CPI * p.LPD == 64;
    // 128 couts per workgroup where the staging slots still divide evenly (W >= 16) and the layer has them; DDPM_WGRAD_WAVES=4: A/B
    static const bool four = getenv("DDPM_WGRAD_WAVES") && atoi(getenv("DDPM_WGRAD_WAVES")) == 4;
    p.nw = p.h16 && Wo >= 16 && Cout % 128 == 0 && !four && split_f16_on(sw().wgrad_f16x3) && !wgrad_plain_form() ? 8 : 4;
    const int co = 16 * p.nw;
    p.ACSh = (p.R + 2) * Wo + 8;
    p.DCSh = p.R * (Wo + 8);
    if (!((p.DCSh / 8) & 1)) p.DCSh += 8;
    p.DHh = 8 + 2 * co * p.DCSh;
    p.LDSh = p.DHh + 2 * kWT * p.ACSh;
    p.lw = Wo == 8 ? 3 : Wo == 16 ? 4 : Wo == 32 ? 5 : 6;
  }
  const int blocks = (Cout / (16 * p.nw)) * (Cin / kWT);
  const int cus = device_cus();
  // one workgroup per CU is what the kernel's registers allow: the most slices that still run as ONE round (12 blocks x 22 slices
  // = 264 workgroups took two rounds on 256 CUs, the second for 8 of them: 137 against 220-250 TFLOP/s for the other shapes)
  int S = cus / blocks;
  if (S > p.T / 4) S = p.T / 4;  // at least four pixel tiles per slice: a small batch would spend more on adding slices than on making them
  if (S < 1) S = 1;
  p.S = S;
  return true;
}
}  // namespace

namespace {
constexpr size_t kWgradHead = 2 * kAmaxBlocks;  // words in front of the partial slabs: the operand maxima of the split-f16 form
size_t wgrad_scratch(const WgradP &p) { return kWgradHead + (size_t)p.S * 9 * p.Cout * p.Cin; }

void wgrad_attrs() {
  static bool done = false;
  if (done) return;
  for (const void *f : {reinterpret_cast<const void *>(&conv3x3_wgrad_kernel<1>), reinterpret_cast<const void *>(&conv3x3_wgrad_kernel<2>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_staged_kernel<1>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_staged_kernel<2>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<5, 4>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<6, 4>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<8, 4>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<12, 4>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<3, 8>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<4, 8>),
                        reinterpret_cast<const void *>(&conv3x3_wgrad_f16x3_kernel<6, 8>)})
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}

// the operand maxima of the split-f16 form into scratch[0 .. 1] (p.part already points behind the head)
bool wgrad_use_h16(const WgradP &p, bool aligned) { return p.h16 && aligned && split_f16_on(sw().wgrad_f16x3) && !wgrad_plain_form(); }
void wgrad_h16_maxima(WgradP &p, float *scratch, hipStream_t s, const unsigned *a_amax = nullptr, int a_n = 0,
                      const unsigned *dy_amax = nullptr, int dy_n = 0) {
  unsigned *mx = reinterpret_cast<unsigned *>(scratch);
  const size_t nd = (size_t)p.B * p.Cout * p.dy_cs / 4, na = (size_t)p.B * p.Cin * p.a_cs / 4;
  // 16 KB per workgroup and pass, up to two workgroups per CU
  if (dy_amax && dy_n > 0) {
    p.amax_d = dy_amax; p.amax_nd = dy_n;
  } else {
    p.amax_nd = (int)std::min<size_t>((nd + 4095) / 4096, kAmaxBlocks);
    hipLaunchKernelGGL(wgrad_absmax_kernel, dim3(p.amax_nd), dim3(256), 0, s, p.dy, nd, mx);
    p.amax_d = mx;
  }
  if (a_amax && a_n > 0) {
    p.amax_a = a_amax; p.amax_na = a_n;
  } else {
    p.amax_na = (int)std::min<size_t>((na + 4095) / 4096, kAmaxBlocks);
    hipLaunchKernelGGL(wgrad_absmax_kernel, dim3(p.amax_na), dim3(256), 0, s, p.a, na, mx + kAmaxBlocks);
    p.amax_a = mx + kAmaxBlocks;
  }
}
// one launch of the MFMA form the plan and the switches select
void wgrad_launch(const WgradP &p, bool aligned, hipStream_t s) {
  dim3 grid(p.Cout / kWT, p.Cin / kWT, p.S);
  const size_t lds = (size_t)kWT * (p.ACS + p.DCS) * sizeof(float);
  const bool staged = p.fast && aligned && !wgrad_plain_form();
  if (p.amax_d) {
    const size_t ldsh = (size_t)p.LDSh * sizeof(_Float16);
    if (p.nw == 8) {
      const dim3 g8(p.Cout / 128, p.Cin / kWT, p.S);
      switch (p.NA) {
        case 6: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<3, 8>), g8, dim3(512), ldsh, s, p); break;
        case 8: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<4, 8>), g8, dim3(512), ldsh, s, p); break;
        default: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<6, 8>), g8, dim3(512), ldsh, s, p); break;
      }
    } else {
      switch (p.NA) {
        case 5: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<5, 4>), grid, dim3(256), ldsh, s, p); break;
        case 6: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<6, 4>), grid, dim3(256), ldsh, s, p); break;
        case 8: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<8, 4>), grid, dim3(256), ldsh, s, p); break;
        default: hipLaunchKernelGGL((conv3x3_wgrad_f16x3_kernel<12, 4>), grid, dim3(256), ldsh, s, p); break;
      }
    }
  } else if (staged && p.stride == 1) hipLaunchKernelGGL(conv3x3_wgrad_staged_kernel<1>, grid, dim3(256), lds, s, p);
  else if (staged) hipLaunchKernelGGL(conv3x3_wgrad_staged_kernel<2>, grid, dim3(256), lds, s, p);
  else if (p.stride == 1) hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL(conv3x3_wgrad_kernel<2>, grid, dim3(256), lds, s, p);
}

// image slices of the generic form: enough workgroups for the chip when there are few (cout, cin) pairs and many pixels
int wgrad_generic_slices(int B, int Cin, int Cout, int Ho, int Wo) {
  // (32 workgroups per CU: each walks its pixels in a latency-bound loop of 4-byte loads -- 2 per CU took 0.5 ms for the first
  // convolution's 128 pairs x 262 144 pixels at batch 256)
  const long pairs = (long)Cout * Cin, want = 32L * device_cus();
  if (pairs >= want || (long)B * Ho * Wo < 16384) return 1;
  long S = (want + pairs - 1) / pairs;
  if (S > B) S = B;
  return S < 2 ? 1 : (int)S;
}
}  // namespace

extern "C" size_t ddpm_conv_wgrad_scratch_floats(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int ksize, int stride) {
  WgradP p;
  if (wgrad_plan(B, Cin, Cout, Hi, Wi, Ho, Wo, ksize, stride, p)) return wgrad_scratch(p);
  const int S = wgrad_generic_slices(B, Cin, Cout, Ho, Wo);
  return S > 1 ? (size_t)S * ksize * ksize * Cout * Cin : 0;
}

extern "C" int ddpm_conv_wgrad_f32(const float *a, const float *dy, float *dw, int B, int Cin, int Cout, int Hi, int Wi, int Ho,
                                   int Wo, int ksize, int stride, float *scratch, size_t scratch_floats, int force_generic,
                                   const unsigned *a_absmax, int a_absmax_n, const unsigned *dy_absmax, int dy_absmax_n,
                                   ddpm_stream_t stream) {
  DDPM_CHECK_ARG(a && dy && dw, "conv_wgrad: null operand");
  DDPM_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv_wgrad: ksize %d / stride %d", ksize, stride);
  DDPM_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && Ho == (Hi + 2 * (ksize / 2) - ksize) / stride + 1 &&
                     Wo == (Wi + 2 * (ksize / 2) - ksize) / stride + 1,
                 "conv_wgrad: extents %dx%d -> %dx%d do not match ksize %d stride %d", Hi, Wi, Ho, Wo, ksize, stride);
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * B * Ho * Wo * (double)Cout * Cin * ksize * ksize;
  const double bytes = 4.0 * ((double)B * Cin * Hi * Wi + (double)B * Cout * Ho * Wo + (double)Cout * Cin * ksize * ksize);
  WgradP p;
  if (!force_generic && wgrad_plan(B, Cin, Cout, Hi, Wi, Ho, Wo, ksize, stride, p) && scratch && scratch_floats >= wgrad_scratch(p)) {
    p.a = a; p.dy = dy; p.part = scratch + kWgradHead;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    wgrad_attrs();
    {
      const char *kname = "train_conv3x3_wgrad";
      char kshape[96];
      if (g_prof_on && sw().prof_shapes) {
        snprintf(kshape, sizeof(kshape), "train_conv3x3_wgrad|%d->%d@%dx%d s%d B%d S%d", Cin, Cout, Ho, Wo, stride, B, p.S);
        kname = kshape;
      }
      ProfScope prof(s, kname, flops, bytes);
      if (wgrad_use_h16(p, aligned)) wgrad_h16_maxima(p, scratch, s, a_absmax, a_absmax_n, dy_absmax, dy_absmax_n);
      wgrad_launch(p, aligned, s);
      DDPM_CHECK_LAUNCH();
    }
    ProfScope prof(s, "train_wgrad_reduce", 0.0, 4.0 * (p.S + 1.0) * 9 * Cout * Cin);
    const size_t n = (size_t)Cout * Cin * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.part, dw, p.S, Cout, Cin, 9, 9, 0);
    DDPM_CHECK_LAUNCH();
    return 0;
  }
  DDPM_CHECK_ARG((long long)Cout * Cin <= 0x7fffffffLL, "conv_wgrad: too many (cout, cin) pairs for the generic kernel");
  int S = force_generic ? 1 : wgrad_generic_slices(B, Cin, Cout, Ho, Wo);
  if (S > 1 && (!scratch || scratch_floats < (size_t)S * ksize * ksize * Cout * Cin)) S = 1;
  ProfScope prof(s, "train_conv_wgrad_generic", flops, bytes);
  hipLaunchKernelGGL(conv_wgrad_generic_kernel, dim3(Cout * Cin, S), dim3(256), 0, s, a, dy, dw, scratch, B, Cin, Cout, Hi, Wi, Ho, Wo,
                     ksize, stride);
  if (S > 1) {
    const size_t n = (size_t)Cout * Cin * ksize * ksize;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scratch, dw, S, Cout, Cin, ksize * ksize, ksize * ksize,
                       0);
  }
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t ddpm_conv3d_wgrad_scratch_floats(int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int stride) {
  WgradP p;
  if (!wgrad_plan(B, Cin, Cout, Hi, Wi, Ho, Wo, 3, stride, p, Di, Do)) return 0;
  return wgrad_scratch(p);
}

extern "C" int ddpm_conv3d_wgrad_f32(const float *a, const float *dy, float *dw, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                                     int Do, int Ho, int Wo, int stride, float *scratch, size_t scratch_floats, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(a && dy && dw && scratch, "conv3d_wgrad: null operand (the scratch of ddpm_conv3d_wgrad_scratch_floats is required)");
  DDPM_CHECK_ARG(stride == 1 || stride == 2, "conv3d_wgrad: stride %d", stride);
  DDPM_CHECK_ARG(Do == (Di - 1) / stride + 1 && Ho == (Hi - 1) / stride + 1 && Wo == (Wi - 1) / stride + 1,
                 "conv3d_wgrad: extents do not match kernel 3, padding 1, stride %d", stride);
  WgradP p;
  DDPM_CHECK_ARG(wgrad_plan(B, Cin, Cout, Hi, Wi, Ho, Wo, 3, stride, p, Di, Do),
                 "conv3d_wgrad: needs Cin %% 64 == 0, Cout %% 64 == 0 and an even W <= 64 (the 3-D latent UNet's shapes)");
  DDPM_CHECK_ARG(scratch_floats >= wgrad_scratch(p), "conv3d_wgrad: scratch too small");
  hipStream_t s = as_stream(stream);
  p.a = a; p.dy = dy; p.part = scratch + kWgradHead;
  wgrad_attrs();
  const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
  const double M = (double)B * Do * Ho * Wo;
  ProfScope prof(s, "train_conv3d_wgrad", 2.0 * M * Cout * Cin * 27,
                 4.0 * ((double)B * Cin * Di * Hi * Wi + M * Cout + 27.0 * Cout * Cin));
  if (wgrad_use_h16(p, aligned)) wgrad_h16_maxima(p, scratch, s);
  const size_t n = (size_t)Cout * Cin * 9;
  for (int kd = 0; kd < 3; ++kd) {  // one depth tap per launch: 9 accumulator tiles per wave is what the register file holds
    p.kd = kd;
    wgrad_launch(p, aligned, s);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.part, dw, p.S, Cout, Cin, 9, 27, 9 * kd);
  }
  DDPM_CHECK_LAUNCH();
  return 0;
}
