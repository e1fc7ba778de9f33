// conv_d3s.hip -- the ResnetBlock 3x3 convolution (stride 1, padding 1) of SMALL launches: the 8x8 / 16x16 levels of a UNet forward
// at a batch of a few images (BASELINE configs[0]: first_n = 16; generative's ResnetBlock.conv1 / conv2 inside
// DiffusionModelUNet.forward, reference call site /root/reference/src/trainers/reconstruct.py:151-153).  Round 4.
//
// Why another kernel.  At B = 16 a 256 -> 256 convolution over 8x8 images is 1.2 GFLOP -- 0.5 us of the chip -- yet every kernel
// built for throughput takes 25-40 us on it (rocprofv3, profiles/r04_b16_kernel_trace.csv): their chunk streams are software
// pipelines three to six stages deep (pixel loads, activation, transform, operand rings), so a stream of four chunks is all fill
// and drain, a chain of eight to ten dependent global-memory round trips.  This kernel has NO pipeline: a workgroup owns
// 64 couts x 128 pixels x 32 input channels, requests everything it needs at once (80 KB of weights by LDS-DMA, 16 KB of pixels
// into registers), stages the pixels, multiplies, stores: one memory round trip, one barrier before and one behind the staging.
// The channel slices (Cin / 32 of them) go to scratch slabs and conv_wino.hip's fixed-order reduce pass adds them up together with
// bias / temb / residual and emits the GroupNorm statistics: bit-reproducible, no atomics.
//
// Arithmetic: direct convolution on v_mfma_f32_32x32x16_f16, operands split into f16 hi / lo planes,
//     x' w' ~= wh xh + wh xl + wl xh,   x' = 2^3 act(x) (2^0 without prologue),  w' = 2^su w  (ddpm_pack_conv_d3h_weight)
// (three of the four partial products: 22 mantissa bits per product, fp32 accumulate); one K-step = 8 channels x two taps.
//
// Layout.  Weights in LDS [chunk 4][tap 10][plane 2][cout 64] units of 8 channels (the packed planes' 64-cout halves are 1 KB
// contiguous: 80 LDS-DMA pieces).  Pixels as [chunk][plane][haloed window] units: W = 8: two whole images per workgroup (4 waves,
// 128 pixels), rows of 10 units; W = 16: one whole image (8 waves, 256 pixels: half the weight traffic of 128-pixel tiles, which
// made 1 024 workgroups of 80 KB each), rows of 18 units; the halo is zeros written before the staging.
// Wave w owns pixel block w (32 pixels: 4 rows of an 8x8 image / 2 rows of a 16x16 one) x both 32-cout blocks.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace ddpm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

constexpr int kSM = 64;                   // couts per workgroup
constexpr int kSCh = 8;                   // input channels per chunk
constexpr int kSNC = 4;                   // chunks per workgroup (one channel slice = 32 channels)
constexpr int kSTaps = 10;                // nine taps + a zero tap (the packed planes of ddpm_pack_conv_d3h_weight)
constexpr int kSWU = kSNC * kSTaps * 2 * kSM;  // weight units in LDS (5 120 = 80 KB)
constexpr int kPackM = 128;               // couts per tile of the packed planes
constexpr float kXScale = 8.f;

struct D3SGeom {
  int W, HW;
  int TI;        // images per tile: 2 (8x8) or 1 (16x16)
  int RS, IS;    // units per haloed row, per image window (W + 2 rows)
  int XU;        // units per (chunk, plane): TI * IS
  int PT, CT, S; // pixel tiles, cout tiles (64), channel slices
  int nch;
  long long pstride;
};

bool d3s_geom(const ddpm_conv_desc &d, D3SGeom &g) {
  const int Cin = d.C1 + d.C2;
  const bool up = d.mode == DDPM_CONV_UPSAMPLE2;  // nearest x2 + 3x3 (generative's Upsample): the staging reads pixel (y >> 1, x >> 1)
  if (d.ksize != 3 || (d.mode != DDPM_CONV_NORMAL && !up) || d.dims == 3 || d.Di > 1 || d.Do > 1 || d.force_direct) return false;
  if (up ? (d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi || d.gscale || d.C2 || d.Wo == 8) : (d.Hi != d.Ho || d.Wi != d.Wo)) return false;
  if (d.Ho != d.Wo) return false;
  if (d.out_act != DDPM_ACT_NONE || d.act == DDPM_ACT_RELU) return false;
  if (d.gscale && (d.act != DDPM_ACT_SILU || !d.gshift)) return false;
  if (!d.gscale && d.act != DDPM_ACT_NONE) return false;
  if (Cin % (kSCh * kSNC) || (d.C2 > 0 && d.C1 % kSCh) || d.Cout % kPackM) return false;
  if (d.Wo != 8 && d.Wo != 16 && d.Wo != 32) return false;
  g.W = d.Wo;
  g.HW = d.Ho * d.Wo;
  g.TI = g.W == 8 ? 2 : 1;
  g.RS = g.W + 2;
  g.IS = (g.W == 32 ? 10 : g.W + 2) * g.RS;  // 32x32: eight rows of an image per workgroup
  g.XU = g.TI * g.IS;
  g.PT = g.W == 32 ? d.B * 4 : (d.B + g.TI - 1) / g.TI;
  g.CT = d.Cout / kSM;
  g.nch = Cin / kSCh;
  g.S = g.nch / kSNC;
  g.pstride = (long long)d.B * d.Cout * g.HW;
  if ((reinterpret_cast<uintptr_t>(d.in1) | reinterpret_cast<uintptr_t>(d.in2) | reinterpret_cast<uintptr_t>(d.gscale) |
       reinterpret_cast<uintptr_t>(d.gshift)) & 15)
    return false;  // 16-byte loads
  return true;
}

size_t d3s_lds_bytes(const D3SGeom &g) { return ((size_t)kSWU + (size_t)kSNC * 2 * g.XU + 1) * 16; }

// lanes 2 k / 2 k + 1 hold channels 0-3 / 4-7 of the same four pixels: they swap halves by DPP, the even lane
// assembles the whole units of pixels 0, 1, the odd lane those of pixels 2, 3
__device__ __forceinline__ v4i_t d3s_pair_unit(h4_t p, h4_t p2, bool odd) {
  const v2i_t a = __builtin_bit_cast(v2i_t, p), b = __builtin_bit_cast(v2i_t, p2);
  v2i_t own, got;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int send = odd ? a[d] : b[d];
    own[d] = odd ? b[d] : a[d];
    got[d] = __builtin_amdgcn_mov_dpp(send, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
  }
  return odd ? v4i_t{got[0], got[1], own[0], own[1]} : v4i_t{own[0], own[1], got[0], got[1]};
}

// NW waves = 32 NW pixels: 4 (W = 8: two images), 8 (W = 16: one image; W = 32: eight rows of an image)
template <bool AFFINE, int NW, int W, bool UP = false>
__global__ __launch_bounds__(64 * NW, 1) void conv_d3s_kernel(const ddpm_conv_desc a, const D3SGeom g, const uint16_t *__restrict__ wq) {
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  constexpr int HW = W * W, TI = NW == 4 ? 2 : 1, NT = 64 * NW;
  constexpr int TR = 32 * NW / (TI * W);  // image rows per tile: 8 / 16 / 8
  constexpr int WR = TR + 2;              // window rows
  static_assert(TI * TR * W == 32 * NW && (TR == W || W == 32), "tile = whole images, or eight rows of a 32x32 image");
  // staging items per chunk: TI images x WR window rows x (W / 4) pixel groups x 2 channel halves: 80 / 144 / 160 -- two rounds
  constexpr int kItemsPerChunk = TI * WR * (W / 4) * 2, kItems = kSNC * kItemsPerChunk;
  static_assert(kItems <= 2 * NT, "two staging rounds");
  const int Cin = a.C1 + a.C2;
  f16x8 *const Xb = lds + kSWU;

  // workgroup -> (channel slice, cout tile, pixel tile): the slices of one output tile are neighbours
  const int split = blockIdx.x % g.S, ct = (blockIdx.x / g.S) % g.CT, pt = blockIdx.x / (g.S * g.CT);
  const int n0 = W == 32 ? pt >> 2 : pt * TI;
  const int y0 = W == 32 ? (pt & 3) * TR : 0;
  const int ch0 = split * kSNC * kSCh;  // first input channel of the slice

  // ---- 1. the slice's weights: chunk q, tap t, plane p -> 64 units (1 KB) of the packed planes
  {
    const f16x8 *const wsrc = reinterpret_cast<const f16x8 *>(wq) +
                              ((size_t)(ct >> 1) * g.nch + (size_t)split * kSNC) * (kSTaps * 2 * kPackM) + (ct & 1) * kSM;
#pragma unroll
    for (int p = 0; p < kSNC * kSTaps * 2 / NW; ++p) {
      const int piece = p * NW + wave;  // (chunk, tap, plane) = piece / 20, (piece / 2) % 10, piece & 1
      __builtin_amdgcn_global_load_lds(wsrc + (size_t)piece * kPackM + lane, lds + piece * kSM, 16, 0, 0);
    }
  }

  // ---- 2. this thread's staging items (two rounds): item = chunk x (four consecutive pixels of a window row) x (channels 4 h .. 4 h + 3)
  v4f_t raw[2][4], gsa[2], gsb[2];
  int sbyte[2];  // byte offset of the first unit this lane stores inside the X region; -1: nothing to stage
  const bool odd = lane & 1;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int e = tid + NT * r;
    const int q = e / kItemsPerChunk, rem = e - q * kItemsPerChunk;
    constexpr int W4 = W >> 2;
    const int hsel = rem & 1, pg = rem >> 1;
    const int ti = pg / (WR * W4), rem2 = pg - ti * (WR * W4);
    const int srow = rem2 / W4, scol = (rem2 - srow * W4) * 4;  // window row 0 .. W + 1 <-> image row srow - 1
    const int yin = y0 + srow - 1, n = n0 + ti;
    const bool own = e < kItems && yin >= 0 && yin < W && n < a.B;
    sbyte[r] = own ? (((q * 2) * g.XU + ti * g.IS + srow * g.RS + scol + 1) + 2 * hsel) * 16 : -1;
    const int cg = ch0 + q * kSCh + 4 * hsel;  // first of the item's four channels
    const bool first = cg < a.C1;               // (C1 % 8 == 0: a chunk never straddles the concat seam)
    if (own && UP) {  // four pixels of the virtual image = two stored ones, each twice
      typedef float v2f_t __attribute__((ext_vector_type(2)));
      const float *src = a.in1 + ((size_t)n * a.C1 + cg) * (HW / 4) + (yin >> 1) * (W / 2) + (scol >> 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const v2f_t v = *reinterpret_cast<const v2f_t *>(src + (size_t)c * (HW / 4));
        raw[r][c] = v4f_t{v[0], v[0], v[1], v[1]};
      }
    } else if (own) {
      const float *src = first ? a.in1 + ((size_t)n * a.C1 + cg) * HW : a.in2 + ((size_t)n * a.C2 + (cg - a.C1)) * HW;
      src += yin * W + scol;
#pragma unroll
      for (int c = 0; c < 4; ++c) raw[r][c] = *reinterpret_cast<const v4f_t *>(src + (size_t)c * HW);
      if (AFFINE) {
        gsa[r] = *reinterpret_cast<const v4f_t *>(a.gscale + (size_t)n * Cin + cg);
        gsb[r] = *reinterpret_cast<const v4f_t *>(a.gshift + (size_t)n * Cin + cg);
      }
    }
  }

  // ---- 3. zeros under the window (halo columns, rows outside the image, images past the batch)
  {
    v4i_t z = {0, 0, 0, 0};
    for (int e = tid; e < kSNC * 2 * g.XU + 1; e += NT) *reinterpret_cast<v4i_t *>(Xb + e) = z;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the loads stay in flight)

  // ---- 4. activation, split, store
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (sbyte[r] < 0) continue;  // (lane pairs share their pixels: both lanes of a pair take the same way)
    h4_t hi[4], lo[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float y = raw[r][c][px];
        if (AFFINE) {
          const float v = __builtin_fmaf(y, gsa[r][c], gsb[r][c]);
          const float t = __builtin_fmaf(y, -1.44269504088896341f * gsa[r][c], -1.44269504088896341f * gsb[r][c]);
          y = (kXScale * v) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
        }
        const _Float16 h = (_Float16)y;
        hi[px][c] = h;
        lo[px][c] = (_Float16)(y - (float)h);
      }
    }
    char *Xw = reinterpret_cast<char *>(Xb) + sbyte[r];
    const int slo = g.XU * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<v4i_t *>(Xw + p * 16) = d3s_pair_unit(hi[p], hi[p + 2], odd);
      *reinterpret_cast<v4i_t *>(Xw + p * 16 + slo) = d3s_pair_unit(lo[p], lo[p + 2], odd);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // weights landed, window complete

  // ---- 5. 4 chunks x 5 K-steps x (2 cout blocks x 3 products)
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int xb;  // this lane's pixel in window units (tap (0, 0))
  {
    const int pp = wave * 32 + l31;
    if (TI == 2) {
      const int ti = pp >> 6, rem = pp & 63;
      xb = ti * g.IS + (rem >> 3) * g.RS + (rem & 7);
    } else {
      xb = (pp / W) * g.RS + (pp % W);
    }
  }
#pragma unroll
  for (int q = 0; q < kSNC; ++q) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tap = min(2 * j + lhi, 8);  // (the tenth tap's weights are zeros: it re-reads tap 8's pixels)
      const int dy = tap / 3, dx = tap - 3 * dy;
      const f16x8 *A = lds + ((q * kSTaps + 2 * j + lhi) * 2) * kSM + l31;
      const f16x8 *X = Xb + (q * 2) * g.XU + xb + dy * g.RS + dx;
      const f16x8 bh = X[0], bl = X[g.XU];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f16x8 ah = A[32 * i], al = A[kSM + 32 * i];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
      }
    }
  }

  // ---- 6. the slice's partial sums -> its scratch slab (NCHW; D[row = cout][col = pixel]: a 32-pixel block is 128 contiguous bytes)
  const float oscale = reinterpret_cast<const float *>(wq + (size_t)a.Cout * Cin * kSTaps * 2)[1] * (AFFINE ? 1.f : kXScale);
  const int pp = wave * 32 + l31;
  const int n = TI == 2 ? n0 + (pp >> 6) : n0;
  const int pix = TI == 2 ? (pp & 63) : y0 * W + pp;
  if (n < a.B) {
    float *const dst = a.scratch + (size_t)split * g.pstride + ((size_t)n * a.Cout + ct * kSM + 4 * lhi) * HW + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * HW] = acc[i][r] * oscale;
  }
}


// ============================================================================================================================
// The 1x1 convolutions of small launches (skip connections, the GroupNorm-ed fused q / k / v projection; generative's
// ResnetBlock.skip_connection / AttentionBlock.to_q, to_k, to_v, reference call site /root/reference/src/trainers/reconstruct.py:
// 151-153): the same one-shot scheme.  Workgroup = 64 couts x 128 consecutive pixels of the flattened (image, pixel) space x a
// slice of 128 input channels: 32 KB of weight planes by LDS-DMA, 64 KB of pixels through registers (GroupNorm affine, 2^3 /
// 2^0 pre-scale, hi / lo split) into [chunk][plane][pixel] units, 8 K-steps of 16 channels x 3 products x 2 cout blocks per wave.
// Weights: ddpm_pack_conv_d1s_weight -- [cout tile 64][chunk 8 ch][plane 2][cout 64] units of 2^su w, su PER 64-COUT TILE RANGE
// of one parameter tensor (the members of a fused q / k / v weight are packed one by one); behind the planes the tiles' epilogue
// scales 1 / (2^3 2^su) and the maxima they came from.
constexpr int kD1KC = 16;                       // chunks per channel slice (128 channels)
constexpr int kD1WU = kD1KC * 2 * kSM;          // weight units in LDS (2 048 = 32 KB)
constexpr int kD1XU = kD1KC * 2 * 128;          // pixel units in LDS (4 096 = 64 KB)

template <bool AFFINE>
__global__ __launch_bounds__(256, 1) void conv_d1s_kernel(const ddpm_conv_desc a, const int S, const long long pstride,
                                                          const uint16_t *__restrict__ wq) {
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HW = a.Ho * a.Wo, Cin = a.C1 + a.C2, nch = Cin / kSCh, CT = a.Cout / kSM;
  const long long npix = (long long)a.B * HW;
  f16x8 *const Xb = lds + kD1WU;
  const int split = blockIdx.x % S, ct = (blockIdx.x / S) % CT, pt = blockIdx.x / (S * CT);
  const int ch0 = split * kD1KC * kSCh;
  const long long p0 = (long long)pt * 128;

  // ---- 1. the slice's weights: 16 chunks x 2 planes x 64 units = 32 pieces of 1 KB, contiguous in the packed planes
  {
    const f16x8 *const wsrc = reinterpret_cast<const f16x8 *>(wq) + ((size_t)ct * nch + (size_t)split * kD1KC) * (2 * kSM);
#pragma unroll
    for (int p = 0; p < kD1KC * 2 / 4; ++p) {
      const int piece = p * 4 + wave;
      __builtin_amdgcn_global_load_lds(wsrc + (size_t)piece * kSM + lane, lds + piece * kSM, 16, 0, 0);
    }
  }
  // ---- 2. staging items: (chunk, four consecutive pixels, channels 4 h .. 4 h + 3): 16 x 32 x 2 = 1 024 = four per thread
  const bool odd = lane & 1;
#pragma unroll
  for (int half = 0; half < 2; ++half) {  // (two rounds of two items: 2 x 2 x 6 loads in flight)
    v4f_t raw[2][4], gsa[2], gsb[2];
    int sunit[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int e = tid + 256 * (2 * half + r);
      const int hsel = e & 1, g4 = (e >> 1) & 31, q = e >> 6;
      const long long p = p0 + 4 * g4;
      const bool own = p < npix;
      const int n = own ? (int)(p / HW) : 0, off = own ? (int)(p - (long long)n * HW) : 0;
      sunit[r] = own ? (q * 2) * 128 + 4 * g4 + 2 * hsel : -1;
      const int cg = ch0 + q * kSCh + 4 * hsel;
      const bool first = cg < a.C1;
      if (own) {
        const float *src = (first ? a.in1 + ((size_t)n * a.C1 + cg) * HW : a.in2 + ((size_t)n * a.C2 + (cg - a.C1)) * HW) + off;
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[r][c] = *reinterpret_cast<const v4f_t *>(src + (size_t)c * HW);
        if (AFFINE) {
          gsa[r] = *reinterpret_cast<const v4f_t *>(a.gscale + (size_t)n * Cin + cg);
          gsb[r] = *reinterpret_cast<const v4f_t *>(a.gshift + (size_t)n * Cin + cg);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      h4_t hi[4], lo[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float y = sunit[r] >= 0 ? raw[r][c][px] : 0.f;  // (pixels past the batch: zeros)
          if (AFFINE) y = sunit[r] >= 0 ? kXScale * __builtin_fmaf(y, gsa[r][c], gsb[r][c]) : 0.f;
          const _Float16 h = (_Float16)y;
          hi[px][c] = h;
          lo[px][c] = (_Float16)(y - (float)h);
        }
      }
      // (every unit of the tile is written, also for pixels past the batch: no zero fill)
      const int e = tid + 256 * (2 * half + r);
      const int u = ((e >> 6) * 2) * 128 + 4 * ((e >> 1) & 31) + 2 * (e & 1);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<v4i_t *>(Xb + u + p) = d3s_pair_unit(hi[p], hi[p + 2], odd);
        *reinterpret_cast<v4i_t *>(Xb + u + p + 128) = d3s_pair_unit(lo[p], lo[p + 2], odd);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- 3. 8 K-steps (lanes 0-31: chunk 2 j, lanes 32-63: chunk 2 j + 1) x 2 cout blocks x 3 products
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int j = 0; j < kD1KC / 2; ++j) {
    const f16x8 *A = lds + ((2 * j + lhi) * 2) * kSM + l31;
    const f16x8 *X = Xb + ((2 * j + lhi) * 2) * 128 + wave * 32 + l31;
    const f16x8 bh = X[0], bl = X[128];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f16x8 ah = A[32 * i], al = A[kSM + 32 * i];
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
    }
  }

  // ---- 4. partial sums -> the slice's slab
  const float oscale = reinterpret_cast<const float *>(wq + (size_t)a.Cout * Cin * 2)[ct] * (AFFINE ? 1.f : kXScale);
  const long long p = p0 + wave * 32 + l31;
  if (p < npix) {
    const int n = (int)(p / HW), off = (int)(p - (long long)n * HW);
    float *const dst = a.scratch + (size_t)split * pstride + ((size_t)n * a.Cout + ct * kSM + 4 * lhi) * HW + off;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * HW] = acc[i][r] * oscale;
  }
}

// ============================================================================================================================
// The Downsample convolutions (3x3, stride 2, padding 1, bias only; generative's Downsample.op, reference call site
// /root/reference/src/trainers/reconstruct.py:151-153) of small launches: the same one-shot scheme over 64 couts x 128 output
// pixels (two 8x8 output images, or eight rows of a 16x16 one) x 32 input channels.  The window is the 17 x (2 WO + 1) input
// patch of the tile's rows; a tap of output pixel (r, c) reads window unit (2 r + dy, 2 c + dx): lanes two units apart (a two-way
// bank conflict on the B operand, which an LDS pipe that is mostly idle here does not notice).  Eight waves: pixel block w & 3 x
// cout block w >> 2 (one accumulator tile each), so that the 1 088 staging items are two rounds + a short third.
template <int WO>
__global__ __launch_bounds__(512, 1) void conv_d3s2_kernel(const ddpm_conv_desc a, const int S, const long long pstride,
                                                           const uint16_t *__restrict__ wq) {
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  constexpr int TI = WO == 8 ? 2 : 1, WI = 2 * WO, HWi = WI * WI, HWo = WO * WO;
  constexpr int WR = 17, RS = WI + 2, IS = WR * RS, XU = TI * IS, W4 = WI / 4;
  constexpr int kItemsPerChunk = TI * WR * W4 * 2, kItems = kSNC * kItemsPerChunk;  // 272, 1 088
  const int Cin = a.C1, nch = Cin / kSCh, CT = a.Cout / kSM;
  f16x8 *const Xb = lds + kSWU;
  const int split = blockIdx.x % S, ct = (blockIdx.x / S) % CT, pt = blockIdx.x / (S * CT);
  const int n0 = WO == 8 ? pt * 2 : pt >> 1;
  const int y0 = WO == 8 ? 0 : (pt & 1) * 8;  // first output row of the tile
  const int ch0 = split * kSNC * kSCh;
  {
    const f16x8 *const wsrc = reinterpret_cast<const f16x8 *>(wq) +
                              ((size_t)(ct >> 1) * nch + (size_t)split * kSNC) * (kSTaps * 2 * kPackM) + (ct & 1) * kSM;
#pragma unroll
    for (int p = 0; p < kSNC * kSTaps * 2 / 8; ++p) {
      const int piece = p * 8 + wave;
      __builtin_amdgcn_global_load_lds(wsrc + (size_t)piece * kPackM + lane, lds + piece * kSM, 16, 0, 0);
    }
  }
  const bool odd = lane & 1;
  auto item = [&](int e, int &sunit, const float *&src) {  // staging item e -> first unit it stores (-1: none), its first pixel
    const int q = e / kItemsPerChunk, rem = e - q * kItemsPerChunk;
    const int hsel = rem & 1, pg = rem >> 1;
    const int ti = pg / (WR * W4), rem2 = pg - ti * (WR * W4);
    const int srow = rem2 / W4, scol = (rem2 - srow * W4) * 4;  // window row 0..16 <-> input row 2 y0 - 1 + srow
    const int yin = 2 * y0 - 1 + srow, n = n0 + ti;
    const bool own = e < kItems && yin >= 0 && yin < WI && n < a.B;
    sunit = own ? (q * 2) * XU + ti * IS + srow * RS + scol + 1 + 2 * hsel : -1;
    src = a.in1 + ((size_t)(own ? n : 0) * Cin + ch0 + q * kSCh + 4 * hsel) * HWi + (own ? yin * WI + scol : 0);
  };
  v4f_t raw[2][4];
  int sunit[2];
  auto request = [&](int r0) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float *src;
      item(tid + 512 * (r0 + r), sunit[r], src);
      if (sunit[r] >= 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[r][c] = *reinterpret_cast<const v4f_t *>(src + (size_t)c * HWi);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (sunit[r] < 0) continue;
      h4_t hi[4], lo[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float y = raw[r][c][px];
          const _Float16 h = (_Float16)y;
          hi[px][c] = h;
          lo[px][c] = (_Float16)(y - (float)h);
        }
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<v4i_t *>(Xb + sunit[r] + p) = d3s_pair_unit(hi[p], hi[p + 2], odd);
        *reinterpret_cast<v4i_t *>(Xb + sunit[r] + p + XU) = d3s_pair_unit(lo[p], lo[p + 2], odd);
      }
    }
  };
  request(0);
  {
    v4i_t z = {0, 0, 0, 0};
    for (int e = tid; e < kSNC * 2 * XU + 1; e += 512) *reinterpret_cast<v4i_t *>(Xb + e) = z;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  stage();
  request(2);  // (items 1 024 .. 1 087: the first 64 threads)
  stage();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int pp = (wave & 3) * 32 + l31, cb = wave >> 2;
  const int ti = WO == 8 ? pp >> 6 : 0, rem = WO == 8 ? pp & 63 : pp;
  const int pr = rem / WO, pc = rem % WO;
  const int xb = ti * IS + 2 * pr * RS + 2 * pc;
#pragma unroll
  for (int q = 0; q < kSNC; ++q) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tap = min(2 * j + lhi, 8);
      const int dy = tap / 3, dx = tap - 3 * dy;
      const f16x8 *A = lds + ((q * kSTaps + 2 * j + lhi) * 2) * kSM + 32 * cb + l31;
      const f16x8 *X = Xb + (q * 2) * XU + xb + dy * RS + dx;
      const f16x8 bh = X[0], bl = X[XU], ah = A[0], al = A[kSM];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  const float oscale = reinterpret_cast<const float *>(wq + (size_t)a.Cout * Cin * kSTaps * 2)[1] * kXScale;
  const int n = n0 + ti;
  if (n < a.B) {
    float *const dst = a.scratch + (size_t)split * pstride + ((size_t)n * a.Cout + ct * kSM + 32 * cb + 4 * lhi) * HWo + (y0 + pr) * WO + pc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * HWo] = acc[r] * oscale;
  }
}

__global__ void d1s_max_kernel(const float *__restrict__ src, unsigned *__restrict__ maxes, int Cin, int64_t total) {
  // one maximum per 64-cout tile of the member (blockIdx.y): rows are Cin floats; one atomic per wave (an atomic per element took
  // 4 ms per call: 0.4 s of start-up for the `big` UNet)
  const int64_t per_tile = (int64_t)kSM * Cin, base = blockIdx.y * per_tile;
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per_tile && base + i < total; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(src[base + i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(maxes + blockIdx.y, __builtin_bit_cast(unsigned, m));  // (non-negative floats order like their bits)
}

// member rows [cout_offset, cout_offset + Cout) of a [Cout_total][Cin] weight
__global__ void d1s_pack_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, int Cout, int Cin, int cout_offset, int Cout_total) {
  float *const scales = reinterpret_cast<float *>(dst + (size_t)Cout_total * Cin * 2);
  const float *const maxes = scales + Cout_total / kSM;
  const int nch = Cin / kSCh;
  const int64_t total = (int64_t)Cout * Cin;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = cout_offset + (int)(i / Cin);
    const int tile = co / kSM, c64 = co % kSM;
    // every tile of the member carries the member's scale: su from the largest of its tiles' maxima
    float umax = 0.f;
    for (int t = cout_offset / kSM; t < (cout_offset + Cout) / kSM; ++t) umax = fmaxf(umax, maxes[t]);
    int e = 0;
    (void)frexpf(umax, &e);
    const int su = umax > 0.f ? 15 - e : 0;
    if (ci == 0 && c64 == 0) scales[tile] = ldexpf(1.f / kXScale, -su);
    const float w = ldexpf(src[i], su);
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    const int chunk = ci / kSCh, cc = ci % kSCh;
    const size_t unit = ((size_t)tile * nch + chunk) * 2 * kSM + c64;
    dst[unit * 8 + cc] = h;
    dst[(unit + kSM) * 8 + cc] = l;
  }
}


// ---- weights of the 3x3 one-shot kernels (ddpm_pack_conv_d3h_weight; the direct split-f16 kernel these planes were first
// built for, conv_d3h.hip, measured 17-22 % slower than the F(4x4) form and was removed in round 5): torch [Cout][Cin][3][3] -> [cout tile 128][chunk 8][tap 10][plane hi | lo][cout 128][8 ch] f16 of 2^su w, tap 9 = 0;
// tail floats {max |w|, 1 / (2^3 2^su)}
__global__ void d3h_max_kernel(const float *__restrict__ src, unsigned *__restrict__ tail, int64_t total) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(src[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __builtin_bit_cast(unsigned, m));  // (non-negative floats order like their bits)
}

__global__ void d3h_pack_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, int Cout, int Cin) {
  float *tail = reinterpret_cast<float *>(dst + (size_t)Cout * Cin * kSTaps * 2);
  const float umax = tail[0];
  int e = 0;
  (void)frexpf(umax, &e);  // umax = m 2^e, m in [0.5, 1)
  const int su = umax > 0.f ? 15 - e : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[1] = ldexpf(1.f / kXScale, -su);
  const int nch = Cin / kSCh;
  const int64_t total = (int64_t)Cout * Cin * kSTaps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % kSTaps);
    const int ci = (int)((i / kSTaps) % Cin), co = (int)(i / ((int64_t)kSTaps * Cin));
    const float w = tap < 9 ? ldexpf(src[((size_t)co * Cin + ci) * 9 + tap], su) : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    const int tile = co / kPackM, c128 = co % kPackM, chunk = ci / kSCh, cc = ci % kSCh;
    const size_t unit = (((size_t)tile * nch + chunk) * kSTaps + tap) * 2 * kPackM + c128;  // plane 0
    dst[unit * 8 + cc] = h;
    dst[(unit + kPackM) * 8 + cc] = l;
  }
}

constexpr int kD3TailHalves = 64;  // behind the planes: float [0] = max |w|, float [1] = 1 / (2^3 2^su)

}  // namespace

size_t conv_d3h_weight_halves(int Cout, int Cin) {
  return (Cout % kPackM == 0 && Cin % kSCh == 0) ? (size_t)Cout * Cin * kSTaps * 2 + kD3TailHalves : 0;
}

int launch_pack_conv_d3h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, hipStream_t s) {
  DDPM_CHECK_ARG(w_raw && dst && conv_d3h_weight_halves(Cout, Cin) != 0, "conv_d3h pack: Cout %% 128 or Cin %% 8 != 0");
  unsigned *tail = reinterpret_cast<unsigned *>(dst + (size_t)Cout * Cin * kSTaps * 2);
  hipError_t e = hipMemsetAsync(tail, 0, kD3TailHalves * 2, s);
  if (e != hipSuccess) {
    set_error("conv_d3h pack: %s", hipGetErrorString(e));
    return (int)e;
  }
  const int64_t n9 = (int64_t)Cout * Cin * 9;
  hipLaunchKernelGGL(d3h_max_kernel, dim3((unsigned)((n9 + 255) / 256 > 1024 ? 1024 : (n9 + 255) / 256)), dim3(256), 0, s, w_raw, tail, n9);
  DDPM_CHECK_LAUNCH();
  const int64_t total = (int64_t)Cout * Cin * kSTaps;
  hipLaunchKernelGGL(d3h_pack_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, s, w_raw,
                     reinterpret_cast<_Float16 *>(dst), Cout, Cin);
  DDPM_CHECK_LAUNCH();
  return 0;
}


static bool d3s_take(const ddpm_conv_desc &d, D3SGeom &g, bool sizing) {
  if (!sw().conv_d3s || !split_f16_on(true) || !d.w_d3h || !d3s_geom(d, g)) return false;
  if (g.S < 2) return false;
  // small launches only: where the throughput kernels' pipelines are all fill and drain.  Upper bound: four workgroups per CU
  // (beyond that the F(4x4) kernel's 2.25-4x fewer multiplies win); `2`: any size (tests)
  const long wgs = (long)g.PT * g.CT * g.S;
  // (32x32: two workgroups per CU at most -- 1 024 workgroups of a 256 -> 128 layer at 16 images measured 56 us against the F(4x4)
  // kernel's 52)
  if (sw().conv_d3s != 2 && (wgs > (g.W == 32 ? 2L : 4L) * device_cus() || (long)g.PT * g.CT > device_cus() / 2 || (long)d.B * g.HW > (g.W == 32 ? 16384 : 4096))) return false;
  if (!sizing && (!d.scratch || d.scratch_floats < (size_t)g.S * (size_t)g.pstride)) return false;
  return true;
}

bool conv_d3s_supported(const ddpm_conv_desc &d) {
  D3SGeom g;
  return d3s_take(d, g, false);
}

size_t conv_d3s_scratch_floats(const ddpm_conv_desc &d) {
  D3SGeom g;
  return d3s_take(d, g, true) ? (size_t)g.S * (size_t)g.pstride : 0;
}

int conv_d3s_stats_parts(const ddpm_conv_desc &d) { return wino_split_reduce_stats_parts(d.Ho * d.Wo); }

int launch_conv_d3s(const ddpm_conv_desc &d, hipStream_t s) {
  D3SGeom g;
  if (!d3s_take(d, g, false)) {
    set_error("conv_d3s: unsupported shape");
    return DDPM_EINVAL;
  }
  static bool attr_done = false;
  if (!attr_done) {
    for (const void *f : {reinterpret_cast<const void *>(&conv_d3s_kernel<false, 4, 8>), reinterpret_cast<const void *>(&conv_d3s_kernel<true, 4, 8>),
                          reinterpret_cast<const void *>(&conv_d3s_kernel<false, 8, 16>), reinterpret_cast<const void *>(&conv_d3s_kernel<true, 8, 16>),
                          reinterpret_cast<const void *>(&conv_d3s_kernel<false, 8, 32>), reinterpret_cast<const void *>(&conv_d3s_kernel<true, 8, 32>),
                          reinterpret_cast<const void *>(&conv_d3s_kernel<false, 8, 16, true>), reinterpret_cast<const void *>(&conv_d3s_kernel<false, 8, 32, true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int Cin = d.C1 + d.C2;
  const double M = (double)d.B * g.HW;
  char kshape[160];
  const char *kname = d.mode == DDPM_CONV_UPSAMPLE2 ? "conv3x3_d3s_up" : d.gscale ? "conv3x3_d3s_gn_silu" : "conv3x3_d3s";
  if (g_prof_on && sw().prof_shapes) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, 2.0 * M * d.Cout * Cin * 9,
                 4.0 * (M * Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * Cin * 9));
  const dim3 grid((unsigned)(g.PT * g.CT * g.S));
  const size_t lds = d3s_lds_bytes(g);
  if (d.mode == DDPM_CONV_UPSAMPLE2) {
    if (g.W == 16) hipLaunchKernelGGL((conv_d3s_kernel<false, 8, 16, true>), grid, dim3(512), lds, s, d, g, d.w_d3h);
    else hipLaunchKernelGGL((conv_d3s_kernel<false, 8, 32, true>), grid, dim3(512), lds, s, d, g, d.w_d3h);
  } else if (g.W == 8) {
    if (d.gscale) hipLaunchKernelGGL((conv_d3s_kernel<true, 4, 8>), grid, dim3(256), lds, s, d, g, d.w_d3h);
    else hipLaunchKernelGGL((conv_d3s_kernel<false, 4, 8>), grid, dim3(256), lds, s, d, g, d.w_d3h);
  } else if (g.W == 16) {
    if (d.gscale) hipLaunchKernelGGL((conv_d3s_kernel<true, 8, 16>), grid, dim3(512), lds, s, d, g, d.w_d3h);
    else hipLaunchKernelGGL((conv_d3s_kernel<false, 8, 16>), grid, dim3(512), lds, s, d, g, d.w_d3h);
  } else {
    if (d.gscale) hipLaunchKernelGGL((conv_d3s_kernel<true, 8, 32>), grid, dim3(512), lds, s, d, g, d.w_d3h);
    else hipLaunchKernelGGL((conv_d3s_kernel<false, 8, 32>), grid, dim3(512), lds, s, d, g, d.w_d3h);
  }
  DDPM_CHECK_LAUNCH();
  ddpm_conv_desc dr = d;
  if (conv_d3s_stats_parts(d) == 0) dr.stats_out = nullptr;
  return launch_wino_split_reduce(dr, g.S, g.pstride, g.HW, s);
}

// ---- 1x1
size_t conv_d1s_weight_halves(int Cout, int Cin) {
  return (Cout % kSM == 0 && Cin % (kD1KC * kSCh) == 0) ? (size_t)Cout * Cin * 2 + 4 * (size_t)(Cout / kSM) + 64 : 0;
}

int launch_pack_conv_d1s_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, int cout_offset, int Cout_total, hipStream_t s) {
  DDPM_CHECK_ARG(w_raw && dst && conv_d1s_weight_halves(Cout_total, Cin) != 0 && Cout % kSM == 0 && cout_offset % kSM == 0 &&
                     cout_offset + Cout <= Cout_total,
                 "conv_d1s pack: Cout %% 64, Cin %% 128 or the member's rows are not whole 64-cout tiles");
  float *scales = reinterpret_cast<float *>(dst + (size_t)Cout_total * Cin * 2);
  unsigned *maxes = reinterpret_cast<unsigned *>(scales + Cout_total / kSM) + cout_offset / kSM;
  hipError_t e = hipMemsetAsync(maxes, 0, (size_t)(Cout / kSM) * 4, s);
  if (e != hipSuccess) {
    set_error("conv_d1s pack: %s", hipGetErrorString(e));
    return (int)e;
  }
  const int64_t n = (int64_t)Cout * Cin;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(d1s_max_kernel, dim3(16, (unsigned)(Cout / kSM)), dim3(256), 0, s, w_raw, maxes, Cin, n);
  DDPM_CHECK_LAUNCH();
  hipLaunchKernelGGL(d1s_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, reinterpret_cast<_Float16 *>(dst), Cout, Cin, cout_offset,
                     Cout_total);
  DDPM_CHECK_LAUNCH();
  return 0;
}

static bool d1s_take(const ddpm_conv_desc &d, int &S, long long &pstride, bool sizing) {
  const int Cin = d.C1 + d.C2;
  if (!sw().conv_d3s || !split_f16_on(true) || !d.w_d3h) return false;
  if (d.ksize != 1 || d.mode != DDPM_CONV_NORMAL || d.dims == 3 || d.Di > 1 || d.Do > 1 || d.force_direct) return false;
  if (d.act != DDPM_ACT_NONE || d.out_act != DDPM_ACT_NONE || (d.gscale && !d.gshift)) return false;
  if (conv_d1s_weight_halves(d.Cout, Cin) == 0 || (d.C2 > 0 && d.C1 % kSCh)) return false;
  const int HW = d.Ho * d.Wo;
  if (HW % 32 || d.Hi != d.Ho || d.Wi != d.Wo) return false;
  if ((reinterpret_cast<uintptr_t>(d.in1) | reinterpret_cast<uintptr_t>(d.in2) | reinterpret_cast<uintptr_t>(d.gscale) |
       reinterpret_cast<uintptr_t>(d.gshift)) & 15)
    return false;
  S = Cin / (kD1KC * kSCh);
  pstride = (long long)d.B * d.Cout * HW;
  const long long npix = (long long)d.B * HW;
  const long wgs = (long)((npix + 127) / 128) * (d.Cout / kSM) * S;
  if (sw().conv_d3s != 2 && (wgs > 4L * device_cus() || npix > sw().d1s_maxpx)) return false;
  if (!sizing && (!d.scratch || d.scratch_floats < (size_t)S * (size_t)pstride)) return false;
  return true;
}

bool conv_d1s_supported(const ddpm_conv_desc &d) {
  int S;
  long long ps;
  return d1s_take(d, S, ps, false);
}

size_t conv_d1s_scratch_floats(const ddpm_conv_desc &d) {
  int S;
  long long ps;
  return d1s_take(d, S, ps, true) ? (size_t)S * (size_t)ps : 0;
}

int launch_conv_d1s(const ddpm_conv_desc &d, hipStream_t s) {
  int S;
  long long pstride;
  if (!d1s_take(d, S, pstride, false)) {
    set_error("conv_d1s: unsupported shape");
    return DDPM_EINVAL;
  }
  static bool attr_done = false;
  if (!attr_done) {
    for (const void *f : {reinterpret_cast<const void *>(&conv_d1s_kernel<false>), reinterpret_cast<const void *>(&conv_d1s_kernel<true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int Cin = d.C1 + d.C2, HW = d.Ho * d.Wo;
  const double M = (double)d.B * HW;
  char kshape[160];
  const char *kname = d.gscale ? "conv1x1_d1s_gn" : "conv1x1_d1s";
  if (g_prof_on && sw().prof_shapes) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, 2.0 * M * d.Cout * Cin, 4.0 * (M * Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * Cin));
  const dim3 grid((unsigned)(((long long)M + 127) / 128 * (d.Cout / kSM) * S));
  const size_t lds = ((size_t)kD1WU + kD1XU) * 16;
  if (d.gscale) hipLaunchKernelGGL(conv_d1s_kernel<true>, grid, dim3(256), lds, s, d, S, pstride, d.w_d3h);
  else hipLaunchKernelGGL(conv_d1s_kernel<false>, grid, dim3(256), lds, s, d, S, pstride, d.w_d3h);
  DDPM_CHECK_LAUNCH();
  ddpm_conv_desc dr = d;
  dr.stats_out = nullptr;
  return launch_wino_split_reduce(dr, S, pstride, HW, s);
}

// ---- stride 2
static bool d3s2_take(const ddpm_conv_desc &d, int &S, long long &pstride, bool sizing) {
  if (!sw().conv_d3s || !split_f16_on(true) || !d.w_d3h) return false;
  if (d.ksize != 3 || d.mode != DDPM_CONV_STRIDE2 || d.dims == 3 || d.Di > 1 || d.Do > 1 || d.force_direct) return false;
  if (d.gscale || d.act != DDPM_ACT_NONE || d.out_act != DDPM_ACT_NONE || d.C2 != 0) return false;
  if ((d.Wo != 8 && d.Wo != 16) || d.Ho != d.Wo || d.Hi != 2 * d.Ho || d.Wi != 2 * d.Wo) return false;
  if (d.C1 % (kSCh * kSNC) || d.Cout % kPackM) return false;
  if (reinterpret_cast<uintptr_t>(d.in1) & 15) return false;
  S = d.C1 / (kSCh * kSNC);
  if (S < 2) return false;
  const int HWo = d.Ho * d.Wo;
  pstride = (long long)d.B * d.Cout * HWo;
  const long pt = d.Wo == 8 ? (d.B + 1) / 2 : 2L * d.B;
  const long wgs = pt * (d.Cout / kSM) * S;
  if (sw().conv_d3s != 2 && (wgs > 4L * device_cus() || (long)d.B * HWo > 4096)) return false;
  if (!sizing && (!d.scratch || d.scratch_floats < (size_t)S * (size_t)pstride)) return false;
  return true;
}

bool conv_d3s2_supported(const ddpm_conv_desc &d) {
  int S;
  long long ps;
  return d3s2_take(d, S, ps, false);
}

size_t conv_d3s2_scratch_floats(const ddpm_conv_desc &d) {
  int S;
  long long ps;
  return d3s2_take(d, S, ps, true) ? (size_t)S * (size_t)ps : 0;
}

int launch_conv_d3s2(const ddpm_conv_desc &d, hipStream_t s) {
  int S;
  long long pstride;
  if (!d3s2_take(d, S, pstride, false)) {
    set_error("conv_d3s2: unsupported shape");
    return DDPM_EINVAL;
  }
  static bool attr_done = false;
  if (!attr_done) {
    for (const void *f : {reinterpret_cast<const void *>(&conv_d3s2_kernel<8>), reinterpret_cast<const void *>(&conv_d3s2_kernel<16>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int HWo = d.Ho * d.Wo;
  const double M = (double)d.B * HWo;
  ProfScope prof(s, "conv3x3_d3s_stride2", 2.0 * M * d.Cout * d.C1 * 9, 4.0 * (4.0 * M * d.C1 + M * d.Cout + (double)d.Cout * d.C1 * 9));
  const long pt = d.Wo == 8 ? (d.B + 1) / 2 : 2L * d.B;
  const dim3 grid((unsigned)(pt * (d.Cout / kSM) * S));
  const int TI = d.Wo == 8 ? 2 : 1;
  const size_t lds = ((size_t)kSWU + (size_t)kSNC * 2 * TI * 17 * (2 * d.Wo + 2) + 1) * 16;
  if (d.Wo == 8) hipLaunchKernelGGL(conv_d3s2_kernel<8>, grid, dim3(512), lds, s, d, S, pstride, d.w_d3h);
  else hipLaunchKernelGGL(conv_d3s2_kernel<16>, grid, dim3(512), lds, s, d, S, pstride, d.w_d3h);
  DDPM_CHECK_LAUNCH();
  ddpm_conv_desc dr = d;
  if (wino_split_reduce_stats_parts(HWo) == 0) dr.stats_out = nullptr;
  return launch_wino_split_reduce(dr, S, pstride, HWo, s);
}

}  // namespace ddpm
