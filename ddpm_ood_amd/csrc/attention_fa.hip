// attention_fa.hip -- register-resident flash attention on the f16 MFMA pipe (round 4).
//
// Same op as attention.hip (generative's AttentionBlock inside DiffusionModelUNet.forward, call site
// /root/reference/src/trainers/reconstruct.py:151-153, constructor arguments /root/reference/src/trainers/base.py:66-86:
// softmax(q k^T / sqrt(d)) v + residual over qkv [B, 3 C, N], head dim 256) and the same arithmetic class (every fp32
// product rebuilt from three f16 products, fp32 accumulate), but a different machine:
//
//   attention8_kernel (round 2 / 3)                        this file
//   64 queries per workgroup, S and P through LDS,         a WAVE owns 16 queries for the whole key loop: its q tile (f16
//   five barriers per 64-key block, K / V split by         planes) and its 16 x 256 output accumulators live in registers, the
//   every workgroup that stages them (64 x per image),     scores never leave registers either -- S^T = K Q^T leaves the MFMA
//   MFMA pipe 21 % busy                                    in exactly the lane layout the next MFMA wants P^T in as its B
//                                                          operand (a key PERMUTATION inside each 32-key block, baked into
//                                                          the V planes, makes that true) -- so the only shared data are the
//                                                          K / V blocks: ONE barrier per 32-key block, LDS-DMA staging
//                                                          (no registers, no ds_write), operands by conflict-free ds_read_b128
//
// Pre-pass (attn_prep_kernel): q, k and v are split ONCE per call into MFMA-ready f16 planes in caller scratch (same bytes
// as the fp32 qkv): x' = 2^4 x, hi = f16(x'), lo = f16(x' - hi).  The power-of-two pre-scale keeps `lo` a normal f16 for
// |x| >= 2^-7 (below: absolute error <= 2^-29, graceful) and replaces the 2^5 / 2^-5 juggling of common.h::split_f16x8 --
// no per-operand v_pk_mul in the loops.  hi overflows above |x| = 4 094: q / k / v are outputs of a GroupNorm-ed 1x1
// convolution, and the numeric guard (include/ddpm_ood_hip.h) catches the inf if a checkpoint ever gets there.
// Products: a b ~= ah bh + ah bl + al bh (al bl <= 2^-22 |a b| is dropped), as attention.hip.
//
// MFMA: v_mfma_f32_16x16x32_f16.  A[m][k]: lane (g = lane / 16, c = lane % 16) holds m = c, k = 8 g .. 8 g + 7;
// B[k][n]: n = c, k = 8 g ..; D[m][n]: n = c, m = 4 g + r (r = 0 .. 3).
//   S^T tile t (16 keys x 16 queries) = sum over d of K[key][d] Q[query][d]:  A = K planes, B = the wave's q registers
//       -> lane (g, c) holds scores of query c for keys 16 t + 4 g + r
//   O^T (256 d x 16 queries) += V[d][key] P^T[key][query]:  A = V planes, B = P^T: lane (g, c) must hold 8 keys of query c --
//       it HAS eight: keys 4 g + r (tile 0) and 16 + 4 g + r (tile 1).  So k-slot 8 g + 4 t + r of the PV MFMA is defined to
//       be key 16 t + 4 g + r, and the pre-pass stores V's A operand in that order.
// The online softmax is per lane: a lane's O values all belong to query c, the row maximum needs two cross-lane steps
// (lanes c, c + 16, c + 32, c + 48), the row sum is kept as a per-lane partial and reduced once at the end.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace ddpm {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kD = 256;            // head dim (num_head_channels of both reference configurations)
constexpr int kKBk = 32;           // keys per block
constexpr int kQW = 16;            // queries per wave
constexpr float kPre = 16.f;       // 2^4 on q, k, v
constexpr float kPScale = 256.f;   // 2^8 on the probabilities
constexpr int kStageUnits = 2 * (kD / 8) * kKBk + 2 * (kKBk / 8) * kD;  // f16x8 units per stage: K planes + V planes (4096)
constexpr int kKUnits = 2 * (kD / 8) * kKBk;                            // 2048 units = 32 KB

// ---- pre-pass: fp32 qkv -> f16 planes ------------------------------------------------------------------------------
//   Qp [n][head][query tile N / 16][k-step 8][plane 2][lane 64]      unit = 8 d of one query  (B operand of S^T)
//   Kp [n][head][key block N / 32][plane 2][d group 32][key 32]      unit = 8 d of one key    (A operand of S^T)
//   Vp [n][head][key block N / 32][plane 2][g 4][d 256]              unit = keys {4 g + r, 16 + 4 g + r} of one d (A of PV)
__device__ __forceinline__ void split_store(const float (&v)[8], f16x8 *hi_dst, f16x8 *lo_dst) {
  f16x8 hi, lo;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = v[t] * kPre;
    const _Float16 h = (_Float16)x;
    hi[t] = h;
    lo[t] = (_Float16)(x - (float)h);
  }
  *hi_dst = hi;
  *lo_dst = lo;
}

__global__ __launch_bounds__(256) void attn_prep_kernel(const float *__restrict__ qkv, f16x8 *__restrict__ planes, int C, int N,
                                                        int heads) {
  // grid.x: units of one (image, head): q N * 32, k N * 32, v N * 32; grid.y = head, grid.z = image
  const int hh = blockIdx.y, n = blockIdx.z;
  const long per = (long)N * (kD / 8);            // units per tensor per plane
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= 3 * per) return;
  const int which = (int)(u / per);               // 0 q, 1 k, 2 v
  const long e = u - which * per;
  const float *src = qkv + (((size_t)n * 3 + which) * C + (size_t)hh * kD) * N;
  f16x8 *base = planes + (((size_t)n * heads + hh) * 3 + which) * 2 * per;
  float v[8];
  if (which < 2) {
    const int tok = (int)(e % N), dg = (int)(e / N);  // consecutive threads: consecutive tokens of one d group
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = src[(size_t)(8 * dg + t) * N + tok];
    if (which == 0) {
      const int qt = tok >> 4, c = tok & 15, ks = dg >> 2, g = dg & 3;
      f16x8 *p = base + ((size_t)(qt * 8 + ks) * 2) * 64 + 16 * g + c;
      split_store(v, p, p + 64);
    } else {
      const int jb = tok >> 5, key = tok & 31;
      f16x8 *p = base + (size_t)jb * kKUnits + dg * kKBk + key;
      split_store(v, p, p + (kD / 8) * kKBk);
    }
  } else {
    // unit (key block jb, g, d): consecutive threads -> g fastest (16-byte pieces of one row), then d
    const int g = (int)(e & 3), d = (int)((e >> 2) % kD), jb = (int)(e / (4 * kD));
    const float *row = src + (size_t)d * N + (size_t)jb * kKBk;
    const v4f a = *reinterpret_cast<const v4f *>(row + 4 * g), b = *reinterpret_cast<const v4f *>(row + 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = a[r];
      v[4 + r] = b[r];
    }
    f16x8 *p = base + (size_t)jb * kKUnits + g * kD + d;
    split_store(v, p, p + (kKBk / 8) * kD);
  }
}

// ---- main kernel -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fa_tile(int &qblk, int &hh, int &n) {  // XCD-aware workgroup order (as attention.hip)
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  qblk = id % gx;
  hh = (id / gx) % gy;
  n = id / (gx * gy);
}

#define FA_MFMA(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (acc), 0, 0, 0)

// max / sum over the four lanes that hold one query (c, c + 16, c + 32, c + 48) on the VALU: v_permlane16_swap /
// v_permlane32_swap exchange 16- / 32-lane halves of two registers -- fed the same value twice they return "my half" and
// "the other half" in every lane (no LDS round trip as with ds_bpermute: the softmax is a serial chain)
// (as inline asm: through __builtin_amdgcn_permlane16_swap hipcc 7.2 loses the SECOND result when both feed one arithmetic
// instruction -- it emitted "v_permlane16_swap v131, v0; v_add_f32 v15, v131, v131".  The s_nop covers the VALU-write ->
// permlane-swap hazard the compiler would otherwise pad itself.)
__device__ __forceinline__ void swap16(float v, float &mine, float &other) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  mine = a;
  other = b;
}
__device__ __forceinline__ void swap32(float v, float &mine, float &other) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  mine = a;
  other = b;
}
__device__ __forceinline__ float quad_max(float v) {
  float a, b;
  swap16(v, a, b);
  v = fmaxf(a, b);
  swap32(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float quad_sum(float v) {
  float a, b;
  swap16(v, a, b);
  v = a + b;
  swap32(v, a, b);
  return a + b;
}

// PIPE: the scores of block jb + 1 are computed in iteration jb, BEFORE the softmax of block jb -- its serial VALU chain (scale,
// max, exchange, exp2, split) then has 96 independent MFMAs of the same wave to hide behind instead of stalling both waves of
// the SIMD in lockstep (every wave passes the block barrier at the same time).  The K ring therefore runs one block ahead of
// the V ring: iteration jb fetches K(jb + 2) and V(jb + 1).
template <int NW, bool PIPE>
__global__ __launch_bounds__(64 * NW) void attention_fa_kernel(const f16x8 *__restrict__ planes,
                                                              const float *__restrict__ residual, float *__restrict__ out,
                                                              int C, int N, int heads, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];  // K ring: 2 x 2048 units, then V ring: 2 x 2048 units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  int qblk, hh, n;
  fa_tile(qblk, hh, n);
  const int i0 = qblk * (kQW * NW) + wave * kQW;  // this wave's first query
  const long per = (long)N * (kD / 8);
  const f16x8 *Qp = planes + ((size_t)n * heads + hh) * 6 * per;
  const f16x8 *Kp = Qp + 2 * per, *Vp = Qp + 4 * per;
  const int nblk = N / kKBk;
  f16x8 *const Kring = lds, *const Vring = lds + 2 * kKUnits;

  // a block's K planes / V planes: 2048 units = 32 pieces of 1 KiB, contiguous in global memory and in LDS
  auto dma_planes = [&](const f16x8 *src, f16x8 *dst) {
#pragma unroll
    for (int p = 0; p < 32 / NW; ++p) {
      const int piece = wave * (32 / NW) + p;
      __builtin_amdgcn_global_load_lds(src + piece * 64 + lane, dst + piece * 64, 16, 0, 0);
    }
  };
  dma_planes(Kp, Kring);
  if (PIPE && nblk > 1) dma_planes(Kp + kKUnits, Kring + kKUnits);
  dma_planes(Vp, Vring);

  // q tile of this wave: 8 k-steps x {hi, lo}
  f16x8 qh[8], ql[8];
  {
    const f16x8 *qsrc = Qp + (size_t)(i0 >> 4) * 8 * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qh[ks] = qsrc[(2 * ks) * 64];
      ql[ks] = qsrc[(2 * ks + 1) * 64];
    }
  }

  f32x4 o[16];
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;
  const float sscale = scale_log2e * (1.f / (kPre * kPre));  // the scores carry 2^8 from the q / k pre-scales

  // S^T = K Q^T of one block, two 16-key tiles.  Operands come through a three-deep register ring (the reads of k-step
  // ks + 2 are issued before the MFMAs of k-step ks: left alone, hipcc emits "2 reads, s_waitcnt lgkmcnt(0), 3 MFMAs" per step
  // and the matrix pipe idles for an LDS round trip 16 times per block), and the six MFMAs of a k-step alternate between four
  // accumulators -- cross terms (kh ql + kl qh) and main terms (kh qh) of the two tiles -- so that no MFMA waits for its
  // predecessor's result; the small cross terms are summed separately and added once.
  auto scores = [&](const f16x8 *Ks, f32x4 (&sacc)[2]) {
    f32x4 cr[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 mn_[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f16x8 kh[3][2], kl[3][2];
    auto rd = [&](int ks, int slot) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        kh[slot][t] = Ks[(4 * ks + g) * kKBk + 16 * t + c];
        kl[slot][t] = Ks[(kD / 8) * kKBk + (4 * ks + g) * kKBk + 16 * t + c];
      }
    };
    rd(0, 0);
    rd(1, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int sl = ks % 3;
      if (ks + 2 < 8) rd(ks + 2, (ks + 2) % 3);
      FA_MFMA(cr[0], kh[sl][0], ql[ks]);
      FA_MFMA(cr[1], kh[sl][1], ql[ks]);
      FA_MFMA(mn_[0], kh[sl][0], qh[ks]);
      FA_MFMA(mn_[1], kh[sl][1], qh[ks]);
      FA_MFMA(cr[0], kl[sl][0], qh[ks]);
      FA_MFMA(cr[1], kl[sl][1], qh[ks]);
      if (ks + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
    sacc[0] = mn_[0] + cr[0];
    sacc[1] = mn_[1] + cr[1];
  };

  f32x4 scur[2];
  if (PIPE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K(0) (and K(1), V(0)) have landed
    scores(Kring, scur);
  }

  for (int jb = 0; jb < nblk; ++jb) {
    const int stage = jb & 1;
    // every piece issued so far has landed -- a wave waits for ITS OWN pieces explicitly: hipcc's __syncthreads() does not put an
    // s_waitcnt vmcnt in front of the barrier for LDS-DMA writes issued in the previous iteration of a loop (it did in the
    // prologue); without it a block could be multiplied before it had arrived -- errors of 1e-3 that came and went with the
    // timing (found by test_big_forward_at_benchmarked_batch_vs_oracle) -- and last iteration's reads are done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 snext[2];
    if (PIPE) {
      // K(jb) was read in the previous iteration (or the prologue): its slot takes K(jb + 2); V(jb - 1)'s slot takes V(jb + 1)
#ifndef FA_NO_DMA  // (static ablations for tools/fa_abl.sh: timing only, wrong results)
      if (jb + 2 < nblk) dma_planes(Kp + (size_t)(jb + 2) * kKUnits, Kring + stage * kKUnits);
      if (jb + 1 < nblk) dma_planes(Vp + (size_t)(jb + 1) * kKUnits, Vring + (stage ^ 1) * kKUnits);
#endif
#ifndef FA_NO_S
      if (jb + 1 < nblk) scores(Kring + (stage ^ 1) * kKUnits, snext);
#else
      snext[0] = scur[0] * 1.0001f;
      snext[1] = scur[1] * 0.9999f;
#endif
    } else {
      if (jb + 1 < nblk) {
        dma_planes(Kp + (size_t)(jb + 1) * kKUnits, Kring + (stage ^ 1) * kKUnits);
        dma_planes(Vp + (size_t)(jb + 1) * kKUnits, Vring + (stage ^ 1) * kKUnits);
      }
      scores(Kring + stage * kKUnits, scur);
    }
    const f16x8 *Vs = Vring + stage * kKUnits;

    // ---- online softmax of query c over the block's 32 keys (log2 domain) ----------------------------------------------
    float x[8];
    float bm = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[4 * t + r] = scur[t][r] * sscale;
        bm = fmaxf(bm, x[4 * t + r]);
      }
#ifndef FA_NO_SOFTMAX
    bm = quad_max(bm);
#endif
    const float mn = fmaxf(m_run, bm);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mn);  // exp2(-inf) = 0 on the first block
    m_run = mn;
    float sum = 0.f;
    f16x8 ph, pl;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#ifndef FA_NO_SOFTMAX
      const float p = __builtin_amdgcn_exp2f(x[k] - mn);
#else
      const float p = x[k] - mn;
#endif
      sum += p;
      const float ps = p * kPScale;
      const _Float16 h = (_Float16)ps;
      ph[k] = h;
      pl[k] = (_Float16)(ps - (float)h);
    }
    l_part = l_part * alpha + sum;
    if (__any(alpha != 1.0f)) {  // (a running maximum that did not move leaves O alone: most blocks after the first few)
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) o[dt] *= alpha;
    }

    // ---- O^T += V P^T over the block's 32 keys (one k-step per 16-row tile of d): tiles in pairs, operands through a
    // three-deep ring, the six MFMAs of a pair alternating between its two accumulators ------------------------------------
    {
      f16x8 vh[3][2], vl[3][2];
      auto rd = [&](int pr, int slot) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          vh[slot][u] = Vs[g * kD + 16 * (2 * pr + u) + c];
          vl[slot][u] = Vs[(kKBk / 8) * kD + g * kD + 16 * (2 * pr + u) + c];
        }
      };
      rd(0, 0);
      rd(1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        const int sl = pr % 3;
        if (pr + 2 < 8) rd(pr + 2, (pr + 2) % 3);
#ifndef FA_NO_PV
        FA_MFMA(o[2 * pr], vh[sl][0], pl);
        FA_MFMA(o[2 * pr + 1], vh[sl][1], pl);
        FA_MFMA(o[2 * pr], vl[sl][0], ph);
        FA_MFMA(o[2 * pr + 1], vl[sl][1], ph);
        FA_MFMA(o[2 * pr], vh[sl][0], ph);
        FA_MFMA(o[2 * pr + 1], vh[sl][1], ph);
#else
        o[2 * pr] += f32x4{(float)vh[sl][0][0], (float)vl[sl][0][1], (float)pl[0], (float)ph[1]};
        o[2 * pr + 1] += f32x4{(float)vh[sl][1][0], (float)vl[sl][1][1], (float)pl[2], (float)ph[3]};
#endif
        if (pr + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
    }
    if (PIPE) {
      scur[0] = snext[0];
      scur[1] = snext[1];
    }
  }

  // ---- normalise, add residual, store [B, C, N] ---------------------------------------------------------------------
  const float l = quad_sum(l_part);
  const float inv = 1.0f / (l * (kPScale * kPre));
  const int i = i0 + c;
  const size_t row0 = ((size_t)n * C + (size_t)hh * kD) * N + i;
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) {
    float rv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[r] = residual ? residual[row0 + (size_t)(16 * dt + 4 * g + r) * N] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[row0 + (size_t)(16 * dt + 4 * g + r) * N] = o[dt][r] * inv + rv[r];
  }
}

}  // namespace

// scratch (in floats) with which launch_attention takes this kernel: the f16 planes of q, k and v.  From 1 024 tokens: below,
// the pre-pass and the 64-query workgroups cost more than the LDS-exchange kernel of attention.hip (measured, B = 16: n = 256
// 43 vs 34 us, n = 64 at B = 1 024 157 vs 74 us; n = 1 024 134 vs 199 us, n = 4 096 816 vs 1 470 us).  DDPM_ATTN_FA=2: every
// multiple of 64 (tests).
size_t attention_fa_scratch_floats(int B, int C, int N, int heads) {
  if (C != heads * kD || N < 64 || (N % 64) != 0) return 0;
  if (N < 1024 && sw().attn_fa != 2) return 0;
  return (size_t)B * 3 * C * N;
}

bool attention_fa_supported(int B, int C, int N, int heads, const float *scratch, size_t scratch_floats) {
  const size_t need = attention_fa_scratch_floats(B, C, N, heads);
  return need > 0 && scratch != nullptr && scratch_floats >= need && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0;
}

int launch_attention_fa(const float *qkv, const float *residual, float *out, int B, int C, int N, int heads, float scale,
                        float *scratch, hipStream_t s) {
  f16x8 *planes = reinterpret_cast<f16x8 *>(scratch);
  {
    const long units = 3L * N * (kD / 8);
    hipLaunchKernelGGL(attn_prep_kernel, dim3((unsigned)((units + 255) / 256), heads, B), dim3(256), 0, s, qkv, planes, C, N, heads);
    DDPM_CHECK_LAUNCH();
  }
  typedef void (*kern_t)(const f16x8 *, const float *, float *, int, int, int, float);
  static const kern_t kerns[2][2] = {{attention_fa_kernel<4, false>, attention_fa_kernel<4, true>},
                                     {attention_fa_kernel<8, false>, attention_fa_kernel<8, true>}};
  static bool attr_done = false;
  static int cus = 256;
  static bool pipe = true;
  if (!attr_done) {
    for (int i = 0; i < 4; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 2][i % 2]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    pipe = !(getenv("DDPM_ATTN_FA_PIPE") && atoi(getenv("DDPM_ATTN_FA_PIPE")) == 0);  // A/B: scores one block ahead
    attr_done = true;
  }
  const size_t lds = (size_t)4 * kKUnits * sizeof(f16x8);  // 128 KB: K ring + V ring, two 32 KB slots each
  const float sl2 = scale * 1.44269504088896341f;
  // 128-query workgroups (eight waves) when that still gives every CU a workgroup, 64-query ones otherwise
  const bool eight = (N % 128) == 0 && (long)B * heads * (N / 128) >= cus;
  if (eight) {
    hipLaunchKernelGGL(kerns[1][pipe], dim3(N / 128, heads, B), dim3(512), lds, s, planes, residual, out, C, N, heads, sl2);
  } else {
    hipLaunchKernelGGL(kerns[0][pipe], dim3(N / 64, heads, B), dim3(256), lds, s, planes, residual, out, C, N, heads, sl2);
  }
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
