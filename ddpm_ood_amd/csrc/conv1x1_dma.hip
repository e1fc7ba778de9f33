// conv1x1_dma.hip -- plain 1x1 convolution (ResnetBlock skip connections, attention proj_attn) as a GEMM whose
// two operands both arrive by LDS-DMA.
//
// Same op as conv_mfma.hip's single-tap variant without a GroupNorm / activation prologue (reference call site
// /root/reference/src/trainers/reconstruct.py:151-153): out[n, co, p] = sum_c W[co, c] x[n, c, p] (+ bias, temb,
// residual), x optionally a virtual concat of two tensors.  With no per-element work on the input, nothing has to
// pass through registers on its way to LDS: NCHW already is the [k][pixel] image the MFMA B operand wants, the
// packed weights (ddpm_pack_conv_weight_f32, [cout tile][k][128]) are the A image, so every chunk of 16 input
// channels is 40 x global_load_lds_dwordx4 per workgroup (5 per wave) and the loop is MFMAs + operand ds_reads only.
// conv_mfma's register-staged single-tap kernel ran these layers at ~50 % MFMA utilisation (32 MFMAs per barrier,
// staging VALU beside them); see DESIGN.md 3.4.
//
// Workgroup = 128 output channels x 256 pixels (consecutive pixels of one image, or whole images), 4 waves as
// 2 (cout halves) x 2 (pixel halves): a wave owns 64 couts x 128 pixels = 2 x 4 MFMA tiles (128 accumulator
// registers), 64 MFMAs per chunk from 6 ds_read_b32 per k-step.  Three LDS buffers of 24 KB: while chunk q is
// multiplied, chunk q + 1 has landed or is landing and chunk q + 2 is being requested; one barrier per chunk.
// Two workgroups fit a CU, so one's epilogue (these layers move as many bytes in the epilogue -- residual in, result
// out -- as in the main loop) overlaps the other's main loop; a 512-pixel / 8-wave version, one workgroup per CU,
// measured the same 100 TFLOP/s with and without deeper operand prefetch: the layers are HBM-side bound.
//
// Split-f16 form (conv1x1_dma_kernel<true>, the default; DDPM_CONV1X1_F16X3=0 restores the f32 MFMA loop).  The
// f32 loop above is MFMA-side limited for these layers (64 v_mfma_f32_32x32x2_f32 = 4 096 SIMD cycles per chunk
// against ~0.25 of that in HBM time).  v_mfma_f32_32x32x16_f16 multiplies a whole 16-channel chunk in one
// instruction at 16x the f32 rate, so an fp32 product is rebuilt from three of them:
//   x = xh + xl, w = wh + wl  (xh = f16(x) round-to-nearest, xl = f16(x - xh); x - xh is exact in fp32)
//   x w ~= xh wh + xh wl + xl wh     (fp32 accumulate in the MFMA; the dropped xl wl is <= 2^-22 |x w|)
// which keeps 22 mantissa bits per product -- the accumulation itself stays fp32 and channels are still added in
// ascending order.  The operands are split in registers right after the ds_read (3 VALU per value:
// v_cvt_pk_f16_f32, v_cvt_f32_f16, v_sub_f32, v_cvt_pk_f16_f32 per pair); the DMA pipeline, LDS image and epilogue
// are the f32 kernel's.  Per chunk and wave: 24 MFMAs (768 cycles) + ~160 VALU instead of 64 MFMAs (4 096 cycles).
// f16 range (5 exponent bits): the low halves would fall into f16 subnormals and lose their bits, so they are
// scaled up before the conversion and the factor is taken back on the HIGH half of the other operand (a power of
// two: exact while that half stays normal):   x w 2^6 = wh xh + (wh 2^-5)(xl 2^5) + (wl 2^5)(xh 2^-5)
// with w' = 2^6 w = wh + wl split in place of w and the accumulators multiplied by 2^-6 after the loop.  Every
// product keeps 22 bits for |x| in [4e-3, 6.5e4] and |w| in [6e-5, 1e3]; below those ranges an operand carries an
// ABSOLUTE error <= 2^-30 (x) / 2^-36 (w) -- it degrades gracefully, like fp32 flush-to-zero does much further
// down -- and above them the high half overflows to inf (loud; nothing a GroupNorm-ed UNet produces).
//
// Tried in round 4 and dropped (same-box A/B, tools/conv1x1_ab.py at B = 1 024): ONE eight-wave workgroup multiplying both
// 128-cout tiles of a 256-cout pair against one staged input tile, so that the input is read once instead of once per cout
// tile (the PMC pass had shown 878 MB per launch against 0.5 GB algorithmic on those layers).  256+256->256 @16x16 375 -> 379 us,
// 256+128->256 @16x16 282 -> 290 us, the q / k / v projections 158 -> 158 / 144 -> 171 / 129 -> 128 us: no gain -- the second
// read hits L2 and the 256-cout layers are bound by the operand SPLIT (VALU), not by bytes: per chunk a wave splits 16 weight
// and 32 input values per lane for 24 MFMAs, and both cout halves split the same input.  What would help is weights pre-split
// at pack time plus waves that own disjoint pixels (halves the split work); not built.
//
// GroupNorm prologue (split-f16 form only): the operands pass through registers for the split anyway, so the
// per-(image, channel) affine of a GroupNorm-ed input (the attention blocks' q / k / v projections) is applied
// there; its scale / shift pairs for the chunk ride the DMA ring (one global_load_lds_dword per wave and chunk:
// [4 groups of 64 pixels][16 scale | 16 shift]).  v = x * scale + shift as in conv_mfma.hip.
#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kF16WScale = 64.f;  // 2^6 on the weights before the hi / lo split, 2^-6 on the accumulators after

constexpr int kDC = 16;          // input channels per chunk
constexpr int kDM = 128;         // output channels per workgroup
constexpr int kDP = 256;         // pixels per workgroup
constexpr int kDBuf = kDC * (kDM + kDP);  // floats per LDS buffer (6 144)
constexpr int kDAff = 4 * 64;             // + the chunk's GroupNorm scale / shift pairs (AFFINE)

static bool conv1x1_f16x3_enabled() {
  return split_f16_on(sw().conv1x1_f16x3);
}

bool conv1x1_dma_supported(const ddpm_conv_desc &d) {
  static const bool enabled = !(getenv("DDPM_CONV1X1_DMA") && atoi(getenv("DDPM_CONV1X1_DMA")) == 0);
  const int Cin = d.C1 + d.C2;
  const long HW = (long)d.Ho * d.Wo;
  if (!enabled || d.force_direct || !d.w_packed) return false;
  if (d.ksize != 1 || d.mode != DDPM_CONV_NORMAL || d.act != DDPM_ACT_NONE) return false;
  // GroupNorm prologue: split-f16 form only; a 64-pixel group of a tile belongs to one image
  if (d.gscale && (!conv1x1_f16x3_enabled() || !d.gshift || HW % 64)) return false;
  if (d.Di > 1 || d.Do > 1) return false;
  if (Cin % kDC || (d.C2 > 0 && d.C1 % kDC) || d.Cout % kDM) return false;
  // a tile is 256 consecutive pixels of one image or a whole number of images; rows of 4 pixels never straddle
  if (HW % 4 || !((HW % kDP == 0) || (kDP % HW == 0))) return false;
  const long tiles = ((long)d.B * HW + kDP - 1) / kDP;
  // smaller launches stay with conv_mfma's 64 / 128-pixel tiles and its split-K (DDPM_CONV1X1_DMA_MIN_WG: A/B of the
  // threshold).  Split-f16: half a chip of workgroups already wins (8x8 skip at B = 256, 128 workgroups: 83 -> 55 us;
  // the `big` UNet's q / k / v at B = 16, 192 workgroups: 115 -> 70 us); the f32 loop needed 1.5 waves of the chip
  static const long min_wg_env = getenv("DDPM_CONV1X1_DMA_MIN_WG") ? atol(getenv("DDPM_CONV1X1_DMA_MIN_WG")) : -1;
  const long min_wg = min_wg_env >= 0 ? min_wg_env
                      : conv1x1_f16x3_enabled() ? 64  // (8x8 skip at B = 128, 64 workgroups: 62 -> 50 us)
                                                : 384;
  return tiles * (d.Cout / kDM) >= min_wg;
}

// PRE (round 4): the weights arrive PRE-SPLIT -- ddpm_pack_conv1x1_h_weight stores, per (cout tile, chunk), the two f16 planes
// hi = f16(2^6 w), lo = f16((2^6 w - hi) 2^5) as [plane][k group of 8][cout 128] units of eight k (the same 8 KB per chunk as
// the fp32 image, so the DMA ring is unchanged), i.e. exactly the registers split_f16x8 produced from the fp32 weights: the A
// operand of a cout block is one ds_read_b128 per plane and no VALU.  With the weight split gone the input split is what is
// left, so the waves own DISJOINT pixels (4 waves x (128 couts x 64 pixels) instead of 2 x 2 x (64 x 128)): each input value
// is split once per workgroup instead of twice.  Per chunk and wave: 16 instead of 48 values through split_f16x8 for the same
// 24 MFMAs.  Same products in the same order -> bit-identical to the non-PRE form (tests/test_gpu_ops.py).
template <bool F16X3, bool AFFINE, bool PRE = false>
__global__ __launch_bounds__(256, 2) void conv1x1_dma_kernel(const ddpm_conv_desc a) {
  static_assert(!PRE || F16X3, "pre-split weights are the split-f16 form's");
  constexpr int NI = PRE ? 4 : 2, NJ = PRE ? 2 : 4;  // cout blocks x pixel tiles (32 x 32) of a wave
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][ A [16][128] | B [16][256] | affine [4][64] ]
  constexpr int kBuf = kDBuf + (AFFINE ? kDAff : 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wco = PRE ? 0 : (wave & 1) * 64, wpx = PRE ? wave * 64 : (wave >> 1) * 128;  // 4 waves: 2 x 2 (PRE: 1 x 4)
  const int HW = a.Ho * a.Wo, Cin = a.C1 + a.C2, nchunks = Cin / kDC;
  // Workgroup -> (pixel tile, cout tile).  The cout tiles of one pixel tile read the same input; workgroups are dealt
  // round-robin to the 8 XCDs (id % 8), each with its own L2, so the cout tiles of a pixel tile take consecutive slots
  // of ONE XCD: they run side by side and all but the first read of the input hits that L2 (DDPM_CONV1X1_XCD=0: the
  // plain (pixel tile, cout tile) grid order, where they are a whole grid row apart).
  int nt;
  long t0;
  {
    const unsigned CT = a.Cout / kDM, PT = (unsigned)(((long)a.B * a.Ho * a.Wo + kDP - 1) / kDP);
    unsigned ptile;
    if (gridDim.y == 1) {
      const unsigned xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
      ptile = (m / CT) * 8 + xcd;
      nt = m % CT;
    } else {
      ptile = blockIdx.x;
      nt = blockIdx.y;
    }
    if (ptile >= PT) return;  // padding of the last group of eight pixel tiles
    t0 = (long)ptile * kDP;   // first pixel of the tile in the flattened (n, p) order
  }
  const long npix = (long)a.B * HW;

  // ---- DMA roles.  A: the chunk's 16 x 128 weights are 8 KB contiguous -> pieces 2 wave, 2 wave + 1 of 8.
  // B: one piece per channel (256 pixels = 1 KB); this wave copies channels 4 wave .. 4 wave + 3.  A lane copies 4
  // consecutive pixels (16 B) of its channel row; pixels past the end re-read the last group.
  const float *wsrc = (PRE ? reinterpret_cast<const float *>(a.w_wino44h) : a.w_packed) + (size_t)nt * nchunks * kDC * kDM +
                      wave * 512 + lane * 4;
  size_t boff1, boff2;  // element offset of (image, channel 0, pixel) in in1 / in2
  {
    long t = t0 + lane * 4;
    if (t >= npix) t = npix - 4;
    const long n = t / HW, p = t - n * HW;
    boff1 = ((size_t)n * a.C1) * HW + p;
    boff2 = ((size_t)n * a.C2) * HW + p;
  }
  // AFFINE: wave w fetches the pairs of the tile's pixel group w (64 pixels, one image): lanes 0-15 the chunk's 16
  // scales, 16-31 its shifts (lanes 32-63 repeat them into the unused half of the slot)
  const float *gsrc = nullptr;
  if constexpr (AFFINE) {
    long n = (t0 + 64 * wave) / HW;
    if (n >= a.B) n = a.B - 1;
    gsrc = ((lane & 16) ? a.gshift : a.gscale) + (size_t)n * Cin + (lane & 15);
  }
  auto dma_chunk = [&](int q, int buf) {
    float *dstA = smem + buf * kBuf;
    float *dstB = dstA + kDC * kDM;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(wsrc + (size_t)q * kDC * kDM + j * 256, dstA + wave * 512 + j * 256, 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * wave + j, cg = q * kDC + c;
      const bool first = cg < a.C1;  // uniform (C1 % 16 == 0)
      const float *src = first ? a.in1 + boff1 + (size_t)cg * HW : a.in2 + boff2 + (size_t)(cg - a.C1) * HW;
      __builtin_amdgcn_global_load_lds(src, dstB + c * kDP, 16, 0, 0);
    }
    if constexpr (AFFINE) __builtin_amdgcn_global_load_lds(gsrc + q * kDC, dstA + kDBuf + wave * 64, 4, 0, 0);
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  dma_chunk(0, 0);
  if (nchunks > 1) dma_chunk(1, 1);
  for (int q = 0; q < nchunks; ++q) {
    // chunk q's six pieces per wave are older than chunk q + 1's: a counted wait leaves the younger ones in flight
    if (q + 1 < nchunks) {
      if constexpr (AFFINE)
        asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    } else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // everyone has left chunk q - 1: its buffer takes chunk q + 2
    if (q + 2 < nchunks) dma_chunk(q + 2, (q + 2) % 3);
    if constexpr (F16X3) {
      // operand images: A[k][cout] / B[k][pixel]; this lane's eight k are 8 lhi .. 8 lhi + 7 (the 32x32x16 layout)
      const float *A = smem + (q % 3) * kBuf + 8 * lhi * kDM + wco + l31;
      const float *Bm = smem + (q % 3) * kBuf + kDC * kDM + 8 * lhi * kDP + wpx + l31;
      const float *G = smem + (q % 3) * kBuf + kDBuf + (wpx >> 6) * 64 + 8 * lhi;
      f16x8 ah[NI], al[NI], as[NI];
      if constexpr (PRE) {
        const f16x8 *Ah = reinterpret_cast<const f16x8 *>(smem + (q % 3) * kBuf) + lhi * kDM + l31, *Al = Ah + 2 * kDM;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          ah[i] = Ah[32 * i];
          al[i] = Al[32 * i];
          as[i] = ah[i] * (_Float16)(1.f / kF16LoScale);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          float v[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = A[t * kDM + 32 * i] * kF16WScale;
          split_f16x8(v, ah[i], al[i], as[i]);
        }
      }
      float bq[2][8];
#pragma unroll
      for (int t = 0; t < 8; ++t) bq[0][t] = Bm[t * kDP];
      float sc[8], sh[8];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) {
#pragma unroll
          for (int t = 0; t < 8; ++t) bq[(j + 1) & 1][t] = Bm[t * kDP + 32 * (j + 1)];
        }
        if constexpr (AFFINE) {
          if ((j & 1) == 0) {  // pixel tiles 2 g, 2 g + 1 are the wave's 64-pixel group g
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              sc[t] = G[(j >> 1) * 64 + t];
              sh[t] = G[(j >> 1) * 64 + 16 + t];
            }
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) bq[j & 1][t] = bq[j & 1][t] * sc[t] + sh[t];
        }
        f16x8 bh, bl, bs;
        split_f16x8(bq[j & 1], bh, bl, bs);
        // the two cross terms first, the two cout tiles interleaved (no MFMA waits on the one before it)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bs, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
    } else {
      const float *A = smem + (q % 3) * kBuf + lhi * kDM + wco + l31;
      const float *Bm = A - (lhi * kDM + wco + l31) + kDC * kDM + lhi * kDP + wpx + l31;
      // operands of k-step ks + 1 are requested before the 8 MFMAs of k-step ks (pinned: left alone, hipcc re-uses one
      // register set and issues the reads one MFMA ahead of their use)
      float av[2][2], bv[2][4];
      auto fetch = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) av[slot][i] = A[2 * ks * kDM + 32 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[slot][j] = Bm[2 * ks * kDP + 32 * j];
      };
      fetch(0, 0);
#pragma unroll
      for (int ks = 0; ks < kDC / 2; ++ks) {
        if (ks + 1 < kDC / 2) fetch(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: D[row = cout][col = pixel] -> NCHW, 128 B contiguous per (register, half-wave) -------------
  const int co_base = nt * kDM + wco + 4 * lhi;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long t = t0 + wpx + 32 * j + l31;
    if (t < npix) {
      const long n = t / HW, p = t - n * HW;
      const size_t obase = ((size_t)n * a.Cout + co_base) * HW + p;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float add[16], rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = 32 * i + (r & 3) + 8 * (r >> 2);
          add[r] = (a.bias ? a.bias[co_base + dco] : 0.f) +
                   (a.chan_add ? a.chan_add[(size_t)n * a.chan_add_stride + co_base + dco] : 0.f);
          rv[r] = a.residual ? a.residual[obase + (size_t)dco * HW] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = 32 * i + (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r];
          if constexpr (F16X3) v *= 1.f / kF16WScale;
          if (a.bias || a.chan_add) v += add[r];
          if (a.residual) v += rv[r];
          if (a.out_act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
          a.out[obase + (size_t)dco * HW] = v;
        }
      }
    }
  }
}

// ---- pre-split weights: torch [Cout][Cin] rows [cout_offset, cout_offset + Cout) of a (possibly fused) 1x1 weight ->
//   [cout tile 128][chunk 16][plane hi | lo][k group 2][cout 128][8 k] f16, values as split_f16x8 makes them from 2^6 w
__global__ void conv1x1_h_pack_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, int Cout, int Cin, int cout_offset,
                                      int Cout_total) {
  const int64_t total = (int64_t)Cout * Cin;
  const int nchunks = Cin / kDC;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)(i / Cin) + cout_offset;
    const float v = src[i] * kF16WScale;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)((v - (float)h) * kF16LoScale);
    const int tile = co / kDM, c128 = co % kDM, chunk = ci / kDC, kk = ci % kDC;
    const size_t unit = (((size_t)tile * nchunks + chunk) * 2 * 2 + (kk >> 3)) * kDM + c128;  // plane 0
    dst[unit * 8 + (kk & 7)] = h;
    dst[(unit + 2 * kDM) * 8 + (kk & 7)] = l;
  }
  (void)Cout_total;
}

size_t conv1x1_h_weight_halves(int Cout, int Cin) {
  return (Cout % kDM == 0 && Cin % kDC == 0) ? (size_t)Cout * Cin * 2 : 0;
}

int launch_pack_conv1x1_h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, int cout_offset, int Cout_total,
                                 hipStream_t s) {
  DDPM_CHECK_ARG(w_raw && dst && conv1x1_h_weight_halves(Cout_total, Cin) != 0 && cout_offset >= 0 && cout_offset + Cout <= Cout_total,
                 "conv1x1 pre-split pack: Cout_total %% 128 or Cin %% 16 != 0, or rows out of range");
  const int64_t total = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(conv1x1_h_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_raw,
                     reinterpret_cast<_Float16 *>(dst), Cout, Cin, cout_offset, Cout_total);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_conv1x1_dma(const ddpm_conv_desc &d, hipStream_t s) {
  if (!conv1x1_dma_supported(d)) {
    set_error("conv1x1_dma: unsupported shape");
    return DDPM_EINVAL;
  }
  const bool f16x3 = conv1x1_f16x3_enabled(), aff = d.gscale != nullptr;
  const size_t lds = (size_t)3 * (kDBuf + (aff ? kDAff : 0)) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    for (const void *f : {reinterpret_cast<const void *>(&conv1x1_dma_kernel<false, false>),
                          reinterpret_cast<const void *>(&conv1x1_dma_kernel<true, false>),
                          reinterpret_cast<const void *>(&conv1x1_dma_kernel<true, true>),
                          reinterpret_cast<const void *>(&conv1x1_dma_kernel<true, false, true>),
                          reinterpret_cast<const void *>(&conv1x1_dma_kernel<true, true, true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long HW = (long)d.Ho * d.Wo, npix = (long)d.B * HW;
  const int Cin = d.C1 + d.C2;
  ProfScope prof(s, aff ? "conv1x1_dma_gn" : "conv1x1_dma", 2.0 * npix * d.Cout * Cin,
                 4.0 * ((double)npix * Cin + (double)npix * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * Cin));
  static const bool xcd_order = !(getenv("DDPM_CONV1X1_XCD") && atoi(getenv("DDPM_CONV1X1_XCD")) == 0);
  const unsigned PT = (unsigned)((npix + kDP - 1) / kDP), CT = d.Cout / kDM;
  const dim3 grid = xcd_order ? dim3(8 * ((PT + 7) / 8) * CT) : dim3(PT, CT);
  static const bool pre_on = !(getenv("DDPM_CONV1X1_PRESPLIT") && atoi(getenv("DDPM_CONV1X1_PRESPLIT")) == 0);  // A/B
  const bool pre = f16x3 && pre_on && d.w_wino44h != nullptr;  // weights pre-split by ddpm_pack_conv1x1_h_weight
  if (pre && aff)
    hipLaunchKernelGGL((conv1x1_dma_kernel<true, true, true>), grid, dim3(256), lds, s, d);
  else if (pre)
    hipLaunchKernelGGL((conv1x1_dma_kernel<true, false, true>), grid, dim3(256), lds, s, d);
  else if (aff)
    hipLaunchKernelGGL((conv1x1_dma_kernel<true, true>), grid, dim3(256), lds, s, d);
  else if (f16x3)
    hipLaunchKernelGGL((conv1x1_dma_kernel<true, false>), grid, dim3(256), lds, s, d);
  else
    hipLaunchKernelGGL((conv1x1_dma_kernel<false, false>), grid, dim3(256), lds, s, d);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
