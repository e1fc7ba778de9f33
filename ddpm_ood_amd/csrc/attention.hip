// attention.hip -- fused self-attention core (QK^T -> softmax -> PV -> + residual): split-f16 products on the f16
// MFMA (default, see the F16X3 note at the kernel) or f32 MFMA.
//
// Replaces torch.baddbmm / softmax / torch.bmm of MONAI-Generative's AttentionBlock._attention
// (reference call site /root/reference/src/trainers/reconstruct.py:151-153).
// q, k, v come straight from the fused QKV 1x1 convolution in channel-major NCHW form
// [B, 3C, N]: q[d][i], k[d][j], v[d][j] with the token index contiguous.  That is already the
// MFMA operand order (32 consecutive lanes = 32 consecutive tokens):
//   S[i][j]  = sum_d q[d][i] k[d][j]   A = q^T (LDS [256][64], loaded once per workgroup),
//                                       B = k   (LDS [256][64], one key block at a time);
//   O[d][i]  = sum_j v[d][j] P[i][j]   A = v   (same LDS buffer re-used after QK^T, row stride 65 so
//                                       that the 32 lanes differing in d hit 32 banks),
//                                       B = P^T from the score tile in LDS;
// the output tile O[d][i] stores 128-byte rows back into NCHW, fused with the residual.
// One workgroup = (image, head, 64 queries); keys are walked in blocks of 64 with an online
// (running max / running sum) softmax, so n = 64 (small UNet @ 8x8) is a single block and
// n = 4096 (big UNet @ 64x64) never materialises the n x n score matrix.  The next block's K (V)
// is prefetched into registers while the PV (QK^T) MFMAs of the current block run, operand
// ds_reads are software-pipelined one k-step ahead of the MFMAs.
// Head dim is fixed at 256 (num_head_channels = 256 in both reference configs,
// /root/reference/src/trainers/base.py:73,84).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kDH = 256;   // head dim
constexpr int kQB = 64;    // queries per workgroup
constexpr int kKB = 64;    // keys per block
constexpr int kLd = 65;    // padded leading dimension (V tile, score tile)
constexpr int kPF = kDH * kKB / 256;  // prefetch floats per thread for one K or V block (64)

// The two MFMA loops are issued by hand: operand reads as inline-asm ds_read2 with immediate offsets, a ring of
// operand registers several MFMAs ahead, and counted lgkmcnt waits fused with the consuming MFMA.  Left to hipcc the
// ring collapses to "read, wait, use" (an exposed LDS latency every 8 MFMAs) and the unrolled QK^T loop gets
// branches; with one wave per SIMD nothing else covers those stalls.
typedef float f2a __attribute__((ext_vector_type(2)));
template <int O0, int O1>
__device__ __forceinline__ f2a lds_read2st64(int addr) {  // floats at addr + 256 O0, addr + 256 O1 bytes
  f2a v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
  return v;
}
template <int O0, int O1>
__device__ __forceinline__ f2a lds_read2(int addr) {  // floats at addr + 4 O0, addr + 4 O1 bytes
  f2a v;
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
  return v;
}
template <int WAIT>
__device__ __forceinline__ void mfma_w(f32x16 &c, float a, float b) {
  asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b), "n"(WAIT));
}
template <int WAIT>
__device__ __forceinline__ void mfma_first_w(f32x16 &c, float a, float b) {  // C = 0
  asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b), "n"(WAIT));
}
__device__ __forceinline__ void mfma_n(f32x16 &c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first_n(f32x16 &c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}
// the hazard recogniser does not see inside inline asm: a VALU read of an MFMA result needs the 16 passes to retire
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

// VEC: N % 4 == 0 -> 16-byte global loads of K / V / Q rows
// Workgroup -> (query block, head, image).  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own
// L2, and every workgroup of an (image, head) streams the same K / V rows (8 MB at n = 4096: 64 query blocks read them
// 64 times).  Dealt in launch order, the query blocks of one image land on all eight L2s and each L2 sees every image;
// instead XCD x takes the x-th eighth of the (image, head, query block) list, so the workgroups resident on an XCD
// walk the same K / V blocks at about the same time and all but the first read hits that L2.
__device__ __forceinline__ void attention_tile(int &qblk, int &hh, int &n) {
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  qblk = id % gx;
  hh = (id / gx) % gy;
  n = id / (gx * gy);
}

// F16X3: both contractions on the f16 MFMA, every fp32 product rebuilt from three v_mfma_f32_32x32x16_f16 (common.h
// split_f16x8; 22 mantissa bits per product, fp32 accumulate -- conv1x1_dma.hip has the error budget).  The operands
// are split in registers on their way from the fp32 LDS tiles to the MFMA; staging, softmax and epilogue are shared
// with the f32 MFMA form.  DDPM_ATTN_F16X3=0 selects the f32 loops.
template <bool VEC, bool F16X3>
__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ qkv,
                                                        const float *__restrict__ residual,
                                                        float *__restrict__ out, int C, int N, int heads,
                                                        float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ql = smem;                    // [256][64]   q tile, d-major
  float *KVl = Ql + kDH * kQB;         // [256][65]   K block (row stride 64) then V block (row stride 65)
  float *Sl = KVl + kDH * kLd;         // [64][65]    scores / probabilities
  float *mrow = Sl + kQB * kLd;        // [64] running max
  float *lrow = mrow + kQB;            // [64] running sum
  float *arow = lrow + kQB;            // [64] rescale factor of this block

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  int qblk, hh, n;
  attention_tile(qblk, hh, n);
  const int i0 = qblk * kQB;

  const float *qp = qkv + ((size_t)n * 3 * C + hh * kDH) * N;
  const float *kp = qp + (size_t)C * N;
  const float *vp = kp + (size_t)C * N;

  // ---- block loaders: a [256 d][64 token] block, zero beyond N ------------------------------------
  float pf[kPF];
  auto load_block = [&](const float *src, int t0) {
    if (VEC) {
#pragma unroll
      for (int r = 0; r < kPF / 4; ++r) {
        const int e4 = tid + 256 * r;
        const int d = e4 >> 4, j = (e4 & 15) * 4;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (t0 + j < N) v = *reinterpret_cast<const v4f *>(src + (size_t)d * N + t0 + j);
        pf[4 * r + 0] = v[0]; pf[4 * r + 1] = v[1]; pf[4 * r + 2] = v[2]; pf[4 * r + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int r = 0; r < kPF; ++r) {
        const int e = tid + 256 * r;
        const int d = e >> 6, j = e & 63;
        pf[r] = (t0 + j < N) ? src[(size_t)d * N + t0 + j] : 0.f;
      }
    }
  };
  auto store_block = [&](float *dst, int ld) {
    if (VEC) {
#pragma unroll
      for (int r = 0; r < kPF / 4; ++r) {
        const int e4 = tid + 256 * r;
        const int d = e4 >> 4, j = (e4 & 15) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[d * ld + j + c] = pf[4 * r + c];
      }
    } else {
#pragma unroll
      for (int r = 0; r < kPF; ++r) {
        const int e = tid + 256 * r;
        dst[(e >> 6) * ld + (e & 63)] = pf[r];
      }
    }
  };

  load_block(qp, i0);
  if constexpr (F16X3) {
    // The q tile is the same for every key block: split it once.  It passes through the (still unused) K / V buffer
    // as fp32 [d][token]; Ql then holds two f16 planes [d / 8][token][8 d] -- hi, and lo scaled by 2^5 -- so that an
    // MFMA A operand (this lane's eight head-dim channels of one query) is a single ds_read_b128 per plane.
    store_block(KVl, kQB);
    __syncthreads();
    f16x8 *Qh = reinterpret_cast<f16x8 *>(Ql), *Qlo = Qh + (kDH / 8) * kQB;
#pragma unroll
    for (int u = 0; u < (kDH / 8) * kQB / 256; ++u) {
      const int g = (tid >> 6) + 4 * u, tok = tid & 63;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = KVl[(8 * g + t) * kQB + tok];
      f16x8 hi, lo, hs;
      split_f16x8(v, hi, lo, hs);
      Qh[g * kQB + tok] = hi;
      Qlo[g * kQB + tok] = lo;
    }
  } else {
    store_block(Ql, kQB);
  }
  load_block(kp, 0);
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
  const int lds0 = static_cast<int>(reinterpret_cast<uintptr_t>(smem));  // LDS byte address of the dynamic block

  for (int j0 = 0; j0 < N; j0 += kKB) {
    __syncthreads();            // previous PV finished with KVl / Sl; orders the Q tile and m / l init
    store_block(KVl, kKB);      // K block, row stride 64
    __syncthreads();
    load_block(vp, j0);         // V of this block flies while QK^T runs

    // ---- S quadrant: rows = queries qi*32.., cols = keys kj*32.. ----------------------------
    if constexpr (F16X3) {
      // k-step ks = head-dim channels 16 ks .. 16 ks + 15; this lane's eight are 16 ks + 8 lhi + t
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const f16x8 *qh = reinterpret_cast<const f16x8 *>(Ql) + lhi * kQB + qi * 32 + l31;
      const f16x8 *ql = qh + (kDH / 8) * kQB;
      const float *kb = KVl + 8 * lhi * kKB + kj * 32 + l31;
      float bv[2][8];
      f16x8 ahv[2], alv[2];
      ahv[0] = qh[0];
      alv[0] = ql[0];
#pragma unroll
      for (int t = 0; t < 8; ++t) bv[0][t] = kb[t * kKB];
#pragma unroll
      for (int ks = 0; ks < kDH / 16; ++ks) {
        if (ks + 1 < kDH / 16) {
          ahv[(ks + 1) & 1] = qh[2 * (ks + 1) * kQB];
          alv[(ks + 1) & 1] = ql[2 * (ks + 1) * kQB];
#pragma unroll
          for (int t = 0; t < 8; ++t) bv[(ks + 1) & 1][t] = kb[(16 * (ks + 1) + t) * kKB];
        }
        f16x8 bh, bl, bs;
        split_f16x8(bv[ks & 1], bh, bl, bs);
        const f16x8 as = ahv[ks & 1] * (_Float16)(1.f / kF16LoScale);
        DDPM_MFMA_F16X3(sacc, ahv[ks & 1], alv[ks & 1], as, bh, bl, bs);
      }
      const bool colok = (j0 + kj * 32 + l31) < N;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Sl[row * kLd + kj * 32 + l31] = colok ? sacc[r] * (scale * 1.44269504088896341f) : -INFINITY;
      }
    } else {
      // two independent accumulator chains (even / odd k-steps); pair p = k-steps 2 p, 2 p + 1 = one ds_read2st64
      // per operand; ring of four pairs (8 MFMAs ahead of their use)
      f32x16 sacc, sacc2;
      const int qa = lds0 + (lhi * kQB + qi * 32 + l31) * 4;
      const int kb = lds0 + (kDH * kQB + lhi * kKB + kj * 32 + l31) * 4;
      f2a av[4], bv[4];
      auto ld = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        av[p & 3] = lds_read2st64<4 * p, 4 * p + 2>(qa);
        bv[p & 3] = lds_read2st64<4 * p, 4 * p + 2>(kb);
      };
      auto step = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int w = p <= 60 ? 6 : 2 * (63 - p);  // operand reads issued after this pair's
        if constexpr (p == 0) {
          mfma_first_w<w>(sacc, av[p & 3][0], bv[p & 3][0]);
          mfma_first_n(sacc2, av[p & 3][1], bv[p & 3][1]);
        } else {
          mfma_w<w>(sacc, av[p & 3][0], bv[p & 3][0]);
          mfma_n(sacc2, av[p & 3][1], bv[p & 3][1]);
        }
        if constexpr (p + 4 < 64) ld(std::integral_constant<int, p + 4>{});
      };
      ld(std::integral_constant<int, 0>{});
      ld(std::integral_constant<int, 1>{});
      ld(std::integral_constant<int, 2>{});
      ld(std::integral_constant<int, 3>{});
      static_for<64>(step);
      mfma_drain();
      const bool colok = (j0 + kj * 32 + l31) < N;
      // scores are kept in the log2 domain: s * scale * log2(e), so that the softmax below is one v_exp_f32 per key
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Sl[row * kLd + kj * 32 + l31] = colok ? (sacc[r] + sacc2[r]) * (scale * 1.44269504088896341f) : -INFINITY;
      }
    }
    __syncthreads();            // scores complete; every wave is done reading the K block

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
      float pr[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float p = __builtin_amdgcn_exp2f(sr[c] - mn);  // scores carry the log2(e) factor already
        if constexpr (F16X3) pr[c] = p; else sr[c] = p;
        sum += p;
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float alpha = __builtin_amdgcn_exp2f(mo - mn);  // exp2(-inf) = 0 on the first block
      store_block(KVl, kLd);              // V block over the K block, row stride 65
      __syncthreads();                    // all 4 readers of mrow[row] are done; V and P are visible
      if constexpr (F16X3) {
        // every score has been read: the probabilities replace them as two f16 planes [key / 8][query][8 keys] (hi,
        // lo 2^5), split once here instead of by each of the four waves -- the PV B operand is one ds_read_b128
        f16x8 *Ph = reinterpret_cast<f16x8 *>(Sl), *Pl = Ph + (kKB / 8) * kQB;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float v[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = pr[8 * u + t];
          f16x8 hi, lo, hs;
          split_f16x8(v, hi, lo, hs);
          Ph[(2 * part + u) * kQB + row] = hi;
          Pl[(2 * part + u) * kQB + row] = lo;
        }
      }
      if (part == 0) {
        mrow[row] = mn;
        lrow[row] = lrow[row] * alpha + sum;
        arow[row] = alpha;
      }
    }
    __syncthreads();
    if (j0 + kKB < N) load_block(kp, j0 + kKB);  // next K block flies while PV runs

    // ---- O[d][i] = alpha_i * O[d][i] + sum_j V[d][j] P[i][j] --------------------------------
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float al = arow[b * 32 + l31];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[a][b][r] *= al;
    }
    if constexpr (F16X3) {
      // k-step ks = keys 16 ks .. 16 ks + 15 of the block; A = V rows d (row stride 65), B = P rows i (row stride 65)
      const float *va = KVl + ((wave * 2) * 32 + l31) * kLd + 8 * lhi;
      const f16x8 *ph = reinterpret_cast<const f16x8 *>(Sl) + lhi * kQB + l31;
      const f16x8 *pl = ph + (kKB / 8) * kQB;
#pragma unroll
      for (int ks = 0; ks < kKB / 16; ++ks) {
        f16x8 ah[2], al[2], as[2], bh[2], bl[2], bs[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          float v[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = va[a * 32 * kLd + 16 * ks + t];
          split_f16x8(v, ah[a], al[a], as[a]);
          bh[a] = ph[2 * ks * kQB + 32 * a];
          bl[a] = pl[2 * ks * kQB + 32 * a];
          bs[a] = bh[a] * (_Float16)(1.f / kF16LoScale);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) DDPM_MFMA_F16X3(o[a][b], ah[a], al[a], as[a], bh[b], bl[b], bs[b]);
      }
    } else {
    // group g = k-steps 2 g, 2 g + 1 (keys 4 g + lhi, 4 g + 2 + lhi): one ds_read2 per operand row block, 8 MFMAs;
    // ring of three groups
    const int va0 = lds0 + (kDH * kQB + ((wave * 2) * 32 + l31) * kLd + lhi) * 4, va1 = va0 + 32 * kLd * 4;
    const int pb0 = lds0 + (kDH * kQB + kDH * kLd + l31 * kLd + lhi) * 4, pb1 = pb0 + 32 * kLd * 4;
    f2a a0[3], a1[3], b0[3], b1[3];
    auto ldg = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      a0[g % 3] = lds_read2<4 * g, 4 * g + 2>(va0);
      b0[g % 3] = lds_read2<4 * g, 4 * g + 2>(pb0);
      a1[g % 3] = lds_read2<4 * g, 4 * g + 2>(va1);
      b1[g % 3] = lds_read2<4 * g, 4 * g + 2>(pb1);
    };
    auto pv = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int w = g <= 13 ? 8 : 4 * (15 - g);
      constexpr int c = g % 3;
      mfma_w<w + 2>(o[0][0], a0[c][0], b0[c][0]);
      mfma_w<w>(o[0][1], a0[c][0], b1[c][0]);
      mfma_n(o[1][0], a1[c][0], b0[c][0]);
      mfma_n(o[1][1], a1[c][0], b1[c][0]);
      mfma_n(o[0][0], a0[c][1], b0[c][1]);
      mfma_n(o[0][1], a0[c][1], b1[c][1]);
      mfma_n(o[1][0], a1[c][1], b0[c][1]);
      mfma_n(o[1][1], a1[c][1], b1[c][1]);
      if constexpr (g + 3 < 16) ldg(std::integral_constant<int, g + 3>{});
    };
    ldg(std::integral_constant<int, 0>{});
    ldg(std::integral_constant<int, 1>{});
    ldg(std::integral_constant<int, 2>{});
    static_for<16>(pv);
    }
  }
  mfma_drain();

  // ---- normalise, add residual, store [B, C, N] ------------------------------------------------
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int i = i0 + b * 32 + l31;
    if (i < N) {
      const float inv = 1.0f / lrow[b * 32 + l31];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = (wave * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          rv[r] = residual ? residual[((size_t)n * C + hh * kDH + d) * N + i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = (wave * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          out[((size_t)n * C + hh * kDH + d) * N + i] = o[a][b][r] * inv + rv[r];
        }
      }
    }
  }
}

// ---- eight-wave split-f16 form -----------------------------------------------------------------------------------
// Same tile (64 queries x 64-key blocks), LDS image and arithmetic as attention_kernel<., true>, with 512 threads: two
// waves per SIMD, so one wave's staging stores, softmax and barrier waits overlap the other's MFMAs (with four waves
// ~60 % of a block's cycles were such stalls).  Work split:
//   QK^T   wave = (head-dim half kh, S quadrant): eight of the sixteen k-steps each; the kh = 1 waves park their partial
//          scores in Sl, the kh = 0 waves add them to their own (low half + high half: a fixed order), scale and mask
//   softmax  eight threads per query row, eight keys (= one f16 plane unit) each
//   PV     wave w owns head-dim rows 32 w .. 32 w + 31 for all 64 queries (two accumulator tiles)
constexpr int kPF8 = kDH * kKB / 512;  // staging floats per thread (32)

template <bool VEC>
__global__ __launch_bounds__(512) void attention8_kernel(const float *__restrict__ qkv,
                                                         const float *__restrict__ residual,
                                                         float *__restrict__ out, int C, int N, int heads,
                                                         float scale, float *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ql = smem;                    // two f16 planes [32][64][8]: q hi, q lo 2^5
  float *KVl = Ql + kDH * kQB;         // [256][65]   K block (row stride 64) then V block (row stride 65)
  float *Sl = KVl + kDH * kLd;         // [64][65]    scores, then two f16 planes [8][64][8] of probabilities
  float *mrow = Sl + kQB * kLd;
  float *lrow = mrow + kQB;
  float *arow = lrow + kQB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  int qblk, hh, n;
  attention_tile(qblk, hh, n);
  const int i0 = qblk * kQB;

  const float *qp = qkv + ((size_t)n * 3 * C + hh * kDH) * N;
  const float *kp = qp + (size_t)C * N;
  const float *vp = kp + (size_t)C * N;

  float pf[kPF8];
  auto load_block = [&](const float *src, int t0) {
    if (VEC) {
#pragma unroll
      for (int r = 0; r < kPF8 / 4; ++r) {
        const int e4 = tid + 512 * r;
        const int d = e4 >> 4, j = (e4 & 15) * 4;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (t0 + j < N) v = *reinterpret_cast<const v4f *>(src + (size_t)d * N + t0 + j);
        pf[4 * r + 0] = v[0]; pf[4 * r + 1] = v[1]; pf[4 * r + 2] = v[2]; pf[4 * r + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int r = 0; r < kPF8; ++r) {
        const int e = tid + 512 * r;
        const int d = e >> 6, j = e & 63;
        pf[r] = (t0 + j < N) ? src[(size_t)d * N + t0 + j] : 0.f;
      }
    }
  };
  auto store_block = [&](float *dst, int ld) {
    if (VEC) {
#pragma unroll
      for (int r = 0; r < kPF8 / 4; ++r) {
        const int e4 = tid + 512 * r;
        const int d = e4 >> 4, j = (e4 & 15) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[d * ld + j + c] = pf[4 * r + c];
      }
    } else {
#pragma unroll
      for (int r = 0; r < kPF8; ++r) {
        const int e = tid + 512 * r;
        dst[(e >> 6) * ld + (e & 63)] = pf[r];
      }
    }
  };

  // K block (round 3): loaded as (group of 8 head-dim channels, key) units and split ONCE, at staging, into the two MFMA-ready
  // f16 planes [d / 8][key][8 d] (hi, lo 2^5) -- as the q tile.  Before, every wave split the K values it multiplied (each
  // value by two waves) right before the MFMA: 8 split_f16x8 and 64 ds_read_b32 per lane and key block; now 4 splits per
  // thread at staging and 16 ds_read_b128 in the QK^T loop.
  auto load_k = [&](int t0) {
    const int key = tid & 63;
    const bool ok = t0 + key < N;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gq = (tid >> 6) + 8 * u;
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[8 * u + t] = ok ? kp[(size_t)(8 * gq + t) * N + t0 + key] : 0.f;
    }
  };
  auto store_k_planes = [&]() {
    f16x8 *Kh = reinterpret_cast<f16x8 *>(KVl), *Kl = Kh + (kDH / 8) * kKB;
    const int key = tid & 63;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gq = (tid >> 6) + 8 * u;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = pf[8 * u + t];
      f16x8 hi, lo, hs;
      split_f16x8(v, hi, lo, hs);
      Kh[gq * kKB + key] = hi;
      Kl[gq * kKB + key] = lo;
    }
  };

  // V block (round 3): as the K block -- (head-dim row d, 8 consecutive keys) units, split once at staging into the f16 planes
  // [key / 8][d][8 keys] (hi, lo 2^5) the PV A operand reads with ONE ds_read_b128 per plane and k-step (was: eight
  // ds_read_b32 of the fp32 block + a split_f16x8 per k-step in the PV loop, and 32 ds_write_b32 per thread at staging).
  // Plane stride 257 units: the eight key groups a thread octet stores land on disjoint banks.
  constexpr int kVG = kDH + 1;
  auto load_v = [&](int t0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e8 = tid + 512 * r;
      const int d = e8 >> 3, j = (e8 & 7) * 8;
      const float *src = vp + (size_t)d * N + t0 + j;
      if (VEC) {
        v4f v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (t0 + j < N) v0 = *reinterpret_cast<const v4f *>(src);
        if (t0 + j + 4 < N) v1 = *reinterpret_cast<const v4f *>(src + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          pf[8 * r + c] = v0[c];
          pf[8 * r + 4 + c] = v1[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) pf[8 * r + c] = (t0 + j + c < N) ? src[c] : 0.f;
      }
    }
  };
  auto store_v_planes = [&]() {
    f16x8 *Vh = reinterpret_cast<f16x8 *>(KVl), *Vl = Vh + (kKB / 8) * kVG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e8 = tid + 512 * r;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = pf[8 * r + t];
      f16x8 hi, lo, hs;
      split_f16x8(v, hi, lo, hs);
      Vh[(e8 & 7) * kVG + (e8 >> 3)] = hi;
      Vl[(e8 & 7) * kVG + (e8 >> 3)] = lo;
    }
  };
  static_assert(2 * (kKB / 8) * (kDH + 1) * 16 <= kDH * kLd * 4, "the V planes fit the K / V buffer");

  // q tile: fp32 through the K / V buffer, then split once into the two planes
  load_block(qp, i0);
  store_block(KVl, kQB);
  __syncthreads();
  {
    f16x8 *Qh = reinterpret_cast<f16x8 *>(Ql), *Qlo = Qh + (kDH / 8) * kQB;
#pragma unroll
    for (int u = 0; u < (kDH / 8) * kQB / 512; ++u) {
      const int g = (tid >> 6) + 8 * u, tok = tid & 63;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = KVl[(8 * g + t) * kQB + tok];
      f16x8 hi, lo, hs;
      split_f16x8(v, hi, lo, hs);
      Qh[g * kQB + tok] = hi;
      Qlo[g * kQB + tok] = lo;
    }
  }
  load_k(0);
  if (tid < kQB) {
    mrow[tid] = -INFINITY;
    lrow[tid] = 0.f;
  }

  f32x16 o[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[b][r] = 0.f;

  const int quad = wave & 3, kh = wave >> 2;
  const int qi = quad >> 1, kj = quad & 1;

  for (int j0 = 0; j0 < N; j0 += kKB) {
    __syncthreads();            // previous PV finished with KVl / Sl; orders the q planes and m / l init
    store_k_planes();           // K block as two f16 planes
    __syncthreads();
    load_v(j0);                 // V of this block flies while QK^T runs

    // ---- partial S quadrant over head-dim channels 128 kh .. 128 kh + 127 ------------------------------
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    {
      const f16x8 *qh = reinterpret_cast<const f16x8 *>(Ql) + (16 * kh + lhi) * kQB + qi * 32 + l31;
      const f16x8 *ql = qh + (kDH / 8) * kQB;
      const f16x8 *kbh = reinterpret_cast<const f16x8 *>(KVl) + (16 * kh + lhi) * kKB + kj * 32 + l31;
      const f16x8 *kbl = kbh + (kDH / 8) * kKB;
      f16x8 ahv[2], alv[2], bhv[2], blv[2];
      ahv[0] = qh[0];
      alv[0] = ql[0];
      bhv[0] = kbh[0];
      blv[0] = kbl[0];
#pragma unroll
      for (int ks = 0; ks < kDH / 32; ++ks) {
        if (ks + 1 < kDH / 32) {
          ahv[(ks + 1) & 1] = qh[2 * (ks + 1) * kQB];
          alv[(ks + 1) & 1] = ql[2 * (ks + 1) * kQB];
          bhv[(ks + 1) & 1] = kbh[2 * (ks + 1) * kKB];
          blv[(ks + 1) & 1] = kbl[2 * (ks + 1) * kKB];
        }
        const f16x8 as = ahv[ks & 1] * (_Float16)(1.f / kF16LoScale);
        const f16x8 bs = bhv[ks & 1] * (_Float16)(1.f / kF16LoScale);
        DDPM_MFMA_F16X3(sacc, ahv[ks & 1], alv[ks & 1], as, bhv[ks & 1], blv[ks & 1], bs);
      }
    }
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Sl[row * kLd + kj * 32 + l31] = sacc[r];
      }
    }
    __syncthreads();            // high-half partials parked; every wave is done reading the K block
    if (kh == 0) {
      const bool colok = (j0 + kj * 32 + l31) < N;
      // scores are kept in the log2 domain: s * scale * log2(e), so that the softmax below is one v_exp_f32 per key
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float *sp = Sl + row * kLd + kj * 32 + l31;
        *sp = colok ? (sacc[r] + *sp) * (scale * 1.44269504088896341f) : -INFINITY;
      }
    }
    __syncthreads();            // scores complete

    // ---- online softmax: 8 threads per query row, 8 keys each ------------------------------------------
    {
      const int row = tid >> 3, part = tid & 7;
      const float *sr = Sl + row * kLd + part * 8;
      float sv[8];
      float bm = -INFINITY;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        sv[c] = sr[c];
        bm = fmaxf(bm, sv[c]);
      }
      bm = fmaxf(bm, __shfl_xor(bm, 1, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 2, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 4, 64));
      const float mo = mrow[row];
      const float mn = fmaxf(mo, bm);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        sv[c] = __builtin_amdgcn_exp2f(sv[c] - mn);
        sum += sv[c];
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      const float alpha = __builtin_amdgcn_exp2f(mo - mn);  // exp2(-inf) = 0 on the first block
      store_v_planes();                   // V block over the K block, as two f16 planes
      __syncthreads();                    // every score and mrow[row] has been read; V is visible
      f16x8 *Ph = reinterpret_cast<f16x8 *>(Sl), *Pl = Ph + (kKB / 8) * kQB;
      f16x8 hi, lo, hs;
      split_f16x8(sv, hi, lo, hs);
      Ph[part * kQB + row] = hi;
      Pl[part * kQB + row] = lo;
      if (part == 0) {
        mrow[row] = mn;
        lrow[row] = lrow[row] * alpha + sum;
        arow[row] = alpha;
      }
    }
    __syncthreads();
    if (j0 + kKB < N) load_k(j0 + kKB);  // next K block flies while PV runs

    // ---- O[d][i] = alpha_i * O[d][i] + sum_j V[d][j] P[i][j], d = 32 wave + .. ---------------------------
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float al = arow[b * 32 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][r] *= al;
    }
    {
      const f16x8 *vh = reinterpret_cast<const f16x8 *>(KVl) + lhi * kVG + wave * 32 + l31;
      const f16x8 *vl = vh + (kKB / 8) * kVG;
      const f16x8 *ph = reinterpret_cast<const f16x8 *>(Sl) + lhi * kQB + l31;
      const f16x8 *pl = ph + (kKB / 8) * kQB;
#pragma unroll
      for (int ks = 0; ks < kKB / 16; ++ks) {
        const f16x8 ah = vh[2 * ks * kVG], al = vl[2 * ks * kVG];
        const f16x8 as = ah * (_Float16)(1.f / kF16LoScale);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f16x8 bh = ph[2 * ks * kQB + 32 * b], bl = pl[2 * ks * kQB + 32 * b];
          const f16x8 bs = bh * (_Float16)(1.f / kF16LoScale);
          DDPM_MFMA_F16X3(o[b], ah, al, as, bh, bl, bs);
        }
      }
    }
  }

  // ---- normalise, add residual, store [B, C, N] ------------------------------------------------
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int i = i0 + b * 32 + l31;
    if (i < N) {
      const float inv = 1.0f / lrow[b * 32 + l31];
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        rv[r] = residual ? residual[((size_t)n * C + hh * kDH + d) * N + i] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        o[b][r] = o[b][r] * inv + rv[r];
        out[((size_t)n * C + hh * kDH + d) * N + i] = o[b][r];
      }
    }
  }
  // ---- the next GroupNorm's per-channel statistics {mean, M2 about it} (launch_attention passes `stats` only for N == 64: this
  // workgroup then holds every token of its 256 channels; a channel = two values in each lane of a half-wave)
  if (stats) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // pairwise (Chan) merge on the DPP path: the lane's two values, then the 32 lanes of the half-wave (valid in its last lane)
      float mean = 0.5f * (o[0][r] + o[1][r]);
      const float dd = o[0][r] - o[1][r];
      float m2 = 0.5f * dd * dd;
      group_moments_last_lane(mean, m2, 2.f, 32);
      if (l31 == 31) {
        const int d = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        *reinterpret_cast<float2 *>(stats + ((size_t)n * C + hh * kDH + d) * 2) = make_float2(mean, m2);
      }
    }
  }
}

// true when launch_attention(...) with these arguments writes the per-channel {mean, M2} pairs of its output to `stats_out`
// ([B, C, 1 slice, 2]): the eight-wave split-f16 kernel with all 64 tokens of an image in one workgroup
bool attention_emits_stats(int B, int C, int N, int heads, const float *scratch, size_t scratch_floats) {
  const bool waves8 = sw().attn_waves8;
  const bool f16x3 = split_f16_on(sw().attn_f16x3);
  if (!f16x3 || !waves8 || N != kQB || C != heads * kDH) return false;
  return !(sw().attn_fa && attention_fa_supported(B, C, N, heads, scratch, scratch_floats));
}

int launch_attention(const float *qkv, const float *residual, float *out, int B, int C, int N, int heads, float scale,
                     hipStream_t s, float *scratch, size_t scratch_floats, float *stats_out) {
  DDPM_CHECK_ARG(qkv && out && B > 0 && N > 0 && heads > 0, "attention: null pointer or empty shape");
  DDPM_CHECK_ARG(C == heads * kDH, "attention: only head dim 256 is built (C = %d, heads = %d)", C, heads);
  DDPM_CHECK_ARG(B <= 65535 && heads <= 65535, "attention: grid too large");
  const size_t lds = (size_t)(kDH * kQB + kDH * kLd + kQB * kLd + 3 * kQB) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    for (const void *f : {reinterpret_cast<const void *>(&attention_kernel<true, false>),
                          reinterpret_cast<const void *>(&attention_kernel<false, false>),
                          reinterpret_cast<const void *>(&attention_kernel<true, true>),
                          reinterpret_cast<const void *>(&attention_kernel<false, true>),
                          reinterpret_cast<const void *>(&attention8_kernel<true>),
                          reinterpret_cast<const void *>(&attention8_kernel<false>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dim3 grid((N + kQB - 1) / kQB, heads, B);
  const bool f16x3 = split_f16_on(sw().attn_f16x3);
  // register-resident form (attention_fa.hip) whenever the caller gave scratch for the f16 planes and N is a multiple of 64
  if (f16x3 && sw().attn_fa && attention_fa_supported(B, C, N, heads, scratch, scratch_floats)) {
    ProfScope prof(s, "attention_fa", 4.0 * B * (double)N * N * C, 4.0 * B * C * (double)N * (residual ? 5 : 4));
    return launch_attention_fa(qkv, residual, out, B, C, N, heads, scale, scratch, s);
  }
  ProfScope prof(s, "attention", 4.0 * B * (double)N * N * C, 4.0 * B * C * (double)N * (residual ? 5 : 4));
  // eight-wave form by default (n = 4096: 892 vs 918 us, n = 256: 32.3 vs 34.8 us); DDPM_ATTN_WAVES=4 selects the other
  const bool waves8 = sw().attn_waves8;
  const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(qkv) & 15) == 0);
  if (f16x3 && waves8) {
    auto kern8 = vec ? attention8_kernel<true> : attention8_kernel<false>;
    float *const st = stats_out && attention_emits_stats(B, C, N, heads, scratch, scratch_floats) ? stats_out : nullptr;
    hipLaunchKernelGGL(kern8, grid, dim3(512), lds, s, qkv, residual, out, C, N, heads, scale, st);
    DDPM_CHECK_LAUNCH();
    return 0;
  }
  auto kern = f16x3 ? (vec ? attention_kernel<true, true> : attention_kernel<false, true>)
                    : (vec ? attention_kernel<true, false> : attention_kernel<false, false>);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, qkv, residual, out, C, N, heads, scale);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
