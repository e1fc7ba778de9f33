// conv_s2h.hip -- the Downsample convolution (3x3, stride 2, padding 1; generative's Downsample between the levels of the down
// path, reference call site /root/reference/src/trainers/reconstruct.py:151-153) as a direct convolution on the f16 MFMA with
// split-f16 operands.
//
// Until round 3 these two launches per `small` forward ran on conv_mfma.hip's fp32 MFMA loop (0.65 of that pipe: 1.5 ms of a
// 23.6 ms forward at B = 1 024).  Here the nine taps are nine K = 16 MFMA steps per 8 input channels, with the arithmetic of
// conv_wino44h.hip: every operand is carried as x = x_h + x_l (x_h = f16(x), x_l = f16(x - x_h)), the instruction's
// K = 16 holds 8 channels x {hi, lo} of the WEIGHT (A = [W_h | W_l], one ds_read_b128: lanes 0-31 the hi plane, 32-63 the lo
// plane) against the input's hi half, then against its lo half (B: both wave halves read the same 16 bytes) -- two MFMAs add all
// four partial products, each exact in the fp32 accumulator.  Weights are pre-scaled by 2^su per layer at pack time (largest
// |w| in [2^14, 2^15)), the input by 2^3 when it is split; 1 / (2^3 2^su) comes off in the epilogue's fma.
//
// Workgroup = 64 couts x 128 output pixels (TH rows of one image, or TI whole images), 4 waves = 4 pixel blocks of 32, each
// wave both 32-cout blocks (two accumulator tiles).  Per chunk of 8 input channels:
//   A   [tap 9][plane 2][cout 64][8 ch] f16 = 18 KB, contiguous in the packed weights (72 bytes per thread)
//   X   the input window of the tile as two f16 planes (hi, lo) of 16-byte units [image][row 2 TH + 1][column parity 2][PWh]:
//       splitting the columns by parity makes the stride-2 pixel walk of a tap a unit-stride walk over one parity plane
//       (tap kx reads parity kx & 1 at x + (kx >> 1)), so that every B operand is one conflict-free ds_read_b128; rows / columns
//       outside the image are stored as zeros.  A thread owns (two adjacent columns, 8 channels): eight coalesced 8-byte global
//       loads, two splits, four 16-byte stores.
// Two LDS buffers and two register sets: chunk q + 2 is requested (global -> registers) before the 36 MFMAs of chunk q, chunk
// q + 1 -- requested a chunk earlier -- is split / stored after them; one barrier per chunk.  Two workgroups per CU (80 KB each).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kS2K = 64;                       // couts per workgroup
constexpr int kS2P = 128;                      // output pixels per workgroup
constexpr int kS2C = 8;                        // input channels per chunk
constexpr int kS2A = 9 * 2 * kS2K * kS2C * 2;  // bytes of a chunk's weights (18 432)
constexpr int kS2NP = 2;                       // staging rounds per thread (<= 512 column pairs per chunk)
constexpr float kS2XScale = 8.f;
constexpr int kS2Tail = 64;                    // f16 slots behind the planes: two floats {max |w|, 1 / (2^3 2^su)}

struct S2Geom {
  int TH, TI;      // output rows per tile (TI == 1) / whole images per tile
  int R;           // input rows per image of a tile: 2 TH + 1
  int PWh;         // units per (row, column parity)
  int RU, IU;      // units per row (2 PWh), per image (R RU)
  int units;       // units per plane and chunk (TI IU)
  int pairs;       // column pairs per chunk (TI R Wo)
  int per;         // output pixels per image of a tile
  int TPI;         // tiles per image (TI == 1)
  int PT, CT;      // pixel tiles, cout tiles
  int nch;         // chunks
};

static bool s2h_geom(const ddpm_conv_desc &d, S2Geom &g) {
  const bool on = split_f16_on(sw().down_s2h != 0);  // DDPM_DOWN_S2H (2 / 3: force a kernel form)
  if (!on || d.force_direct || !d.w_wino44h) return false;
  if (d.mode != DDPM_CONV_STRIDE2 || d.ksize != 3 || d.dims == 3 || d.Di > 1 || d.Do > 1) return false;
  if (d.gscale || d.act != DDPM_ACT_NONE || d.C2 || d.chan_add || d.residual || d.out_act != DDPM_ACT_NONE) return false;
  if (d.C1 % kS2C || d.Cout % kS2K) return false;
  if (d.Hi != 2 * d.Ho || d.Wi != 2 * d.Wo || d.Wo < 2) return false;
  if ((double)d.B * d.C1 * d.Hi * d.Wi * 4 >= 2147483648.0) return false;  // 32-bit buffer offsets
  const int HWo = d.Ho * d.Wo;
  if (HWo >= kS2P) {
    if (kS2P % d.Wo) return false;
    g.TI = 1;
    g.TH = kS2P / d.Wo;
    if (d.Ho % g.TH) return false;
    g.TPI = d.Ho / g.TH;
    g.PT = d.B * g.TPI;
  } else {
    if (kS2P % HWo) return false;
    g.TI = kS2P / HWo;
    g.TH = d.Ho;
    g.TPI = 1;
    g.PT = (d.B + g.TI - 1) / g.TI;
  }
  g.per = g.TH * d.Wo;
  g.R = 2 * g.TH + 1;
  // units per parity row: Wo + 1 used; padded so that the 16 lanes the LDS serves per cycle of a ds_read_b128 (four rows of a
  // 32-pixel block at Wo = 8, two at 16) fall on 16 different 16-byte bank groups
  g.PWh = d.Wo == 16 ? 20 : d.Wo == 8 ? 10 : d.Wo + 1;
  g.RU = 2 * g.PWh;
  g.IU = g.R * g.RU;
  g.units = g.TI * g.IU;
  g.pairs = g.TI * g.R * d.Wo;
  if (g.pairs > 256 * kS2NP || (reinterpret_cast<uintptr_t>(d.in1) & 7)) return false;
  if (2 * (kS2A + 2 * g.units * 16) > 160 * 1024) return false;
  g.CT = d.Cout / kS2K;
  g.nch = d.C1 / kS2C;
  return true;
}

bool conv_s2h_supported(const ddpm_conv_desc &d) {
  S2Geom g;
  return s2h_geom(d, g);
}

// slices per (image, cout) of the GroupNorm statistics the epilogues write to desc.stats_out (0: none).  A wave holds 32 output
// pixels of each of its couts: an image's share of a tile has to be whole waves (32, 64 or 128 pixels); slices = tiles per image.
int conv_s2h_stats_parts(const ddpm_conv_desc &d) {
  S2Geom g;
  if (!s2h_geom(d, g) || g.per % 32) return 0;
  const int parts = g.TI == 1 ? g.TPI : 1;
  return parts <= 8 ? parts : 0;
}

namespace {
// sums over the two 32-lane halves of a wave on the DPP path (row_shr:1, 2, 4, 8, row_bcast:15): valid in lanes 31 and 63
__device__ __forceinline__ float half_sum_dpp(float v) {
  v += dpp_mov0<0x111>(v);
  v += dpp_mov0<0x112>(v);
  v += dpp_mov0<0x114>(v);
  v += dpp_mov0<0x118>(v);
  v += dpp_mov0<0x142>(v);
  return v;
}
// {mean, M2 about it} of the 32 values a half-wave holds of one cout (valid in lanes 31 / 63)
__device__ __forceinline__ float2 half_moments(float v, int lhi) {
  const float sm = half_sum_dpp(v);
  const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sm), 31));
  const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sm), 63));
  const float mu = (lhi ? s1 : s0) * (1.f / 32.f);
  const float dv = v - mu;
  return make_float2(mu, half_sum_dpp(dv * dv));
}
// W = 1, 2 or 4 such pairs (entries `stride` apart, 32 values each) merged pairwise in a fixed order (Chan)
__device__ __forceinline__ float2 merge_half_moments(const float2 *e, int stride, int W) {
  if (W == 1) return e[0];
  const float2 a = e[0], b = e[stride];
  const float d01 = b.x - a.x;
  const float2 m01 = make_float2(0.5f * (a.x + b.x), (a.y + b.y) + d01 * d01 * 16.f);
  if (W == 2) return m01;
  const float2 c = e[2 * stride], f = e[3 * stride];
  const float d23 = f.x - c.x;
  const float2 m23 = make_float2(0.5f * (c.x + f.x), (c.y + f.y) + d23 * d23 * 16.f);
  const float dd = m23.x - m01.x;
  return make_float2(0.5f * (m01.x + m23.x), (m01.y + m23.y) + dd * dd * 32.f);
}
}  // namespace

__global__ __launch_bounds__(256, 2) void conv_s2h_kernel(const ddpm_conv_desc a, const S2Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smb[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Cin = a.C1, HWin = a.Hi * a.Wi;

  // workgroup -> (pixel tile, cout tile): the cout tiles of a pixel tile are neighbours on ONE XCD (they read the same input)
  const unsigned xcd = blockIdx.x & 7, mm = blockIdx.x >> 3;
  const unsigned ptile = (mm / g.CT) * 8 + xcd;
  const int ct = mm % g.CT;
  if (ptile >= (unsigned)g.PT) return;
  int n0, y0;
  if (g.TI == 1) {
    n0 = ptile / g.TPI;
    y0 = (ptile - n0 * g.TPI) * g.TH;
  } else {
    n0 = ptile * g.TI;
    y0 = 0;
  }
  const int xbytes = g.units * 16;        // one plane of a chunk
  const int bufB = kS2A + 2 * xbytes;     // one LDS buffer: A | X hi | X lo

  // ---- staging roles (chunk-invariant): column PAIR e = tid + 256 r -> (image, row, j): input columns 2 j (parity plane 1,
  // unit j) and 2 j + 1 (parity plane 0, unit j + 1) of one row come with ONE 8-byte load per channel.  (Loading the two
  // parities separately fetched every cache line twice with half-used 4-byte lanes: the X loads were 36 % of the kernel.)
  int soff[kS2NP], u0[kS2NP];
  bool own[kS2NP];
#pragma unroll
  for (int r = 0; r < kS2NP; ++r) {
    const int e = tid + 256 * r;
    own[r] = e < g.pairs;
    soff[r] = -1;  // rows above / below the image, images past the batch: zeros
    u0[r] = 0;
    if (own[r]) {
      const int ti = e / (g.R * a.Wo), rem = e - ti * (g.R * a.Wo);
      const int row = rem / a.Wo, j = rem - row * a.Wo;
      const int n = n0 + ti, rin = 2 * y0 - 1 + row;
      if (n < a.B && rin >= 0 && rin < a.Hi) soff[r] = (n * Cin) * HWin + rin * a.Wi + 2 * j;  // (elements; < 2^29)
      u0[r] = ti * g.IU + row * g.RU + j;  // odd column -> unit u0 + 1 (parity 0), even column -> unit u0 + PWh (parity 1)
    }
  }
  const float *xsrc[kS2NP];
  float xmul[kS2NP];  // 2^3, or 0 for rows outside the image
#pragma unroll
  for (int r = 0; r < kS2NP; ++r) {
    xsrc[r] = a.in1 + (soff[r] >= 0 ? soff[r] : 0);
    xmul[r] = soff[r] >= 0 ? kS2XScale : 0.f;
  }
  // column -1 (parity 0, unit 0 of every row) is never loaded: zero in both buffers and planes, once
  for (int e = tid; e < 4 * g.TI * g.R; e += 256) {
    const int w = e / (g.TI * g.R), rr = e - w * (g.TI * g.R);  // w = buffer * 2 + plane
    const int ti = rr / g.R, row = rr - ti * g.R;
    f16x8 z;
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = (_Float16)0.f;
    reinterpret_cast<f16x8 *>(smb + (w >> 1) * bufB + kS2A + (w & 1) * xbytes)[ti * g.IU + row * g.RU] = z;
  }
  // Staging registers: TWO chunks in flight (a chunk's 36 MFMAs are ~0.3 us, an L2 round trip is 3 - 5 times that: with one
  // chunk of prefetch the kernel ran at 168 TFLOP/s, every chunk waiting for its operands).  Set s holds the pixels (8
  // channels x up to 4 units) and the 72 bytes of weights per thread of one chunk; everything goes global -> registers ->
  // LDS with plain loads, so the compiler's own vmcnt bookkeeping lets the younger set stay in flight.
  typedef float v4f_t __attribute__((ext_vector_type(4)));
  typedef float v2f_t __attribute__((ext_vector_type(2)));
  v2f_t xr[2][kS2NP][kS2C];  // [set][round][channel] = the pair's two columns (kept packed: unpacking at the load made hipcc wait for
                             // the data right there, i.e. no prefetch at all)
  v4f_t wr[2][5];
  const v4f_t *wsrc = reinterpret_cast<const v4f_t *>(a.w_wino44h + (size_t)ct * g.nch * (kS2A / 2)) + tid;
  auto load = [&](auto setc, int q) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int r = 0; r < kS2NP; ++r) {
      if (256 * r < g.pairs) {  // (uniform)
#pragma unroll
        for (int c = 0; c < kS2C; ++c) {
#ifdef S2H_NO_XLOAD  // (timing experiments: wrong results)
          xr[S][r][c] = v2f_t{(float)q, (float)q};
#else
          // (a plain 8-byte load: this toolchain lowers __builtin_amdgcn_raw_buffer_load_b64 to ONE buffer_load_dword)
          // unconditional (rows outside the image read the tensor's first pair and are zeroed when they are split): a
          // branch per load kept hipcc from issuing the sixteen loads back to back
          xr[S][r][c] = *reinterpret_cast<const v2f_t *>(xsrc[r] + (size_t)(q * kS2C + c) * HWin);
#endif
        }
      }
    }
    const v4f_t *src = wsrc + (size_t)q * (kS2A / 16);
#ifdef S2H_NO_WLOAD
#pragma unroll
    for (int j = 0; j < 5; ++j) wr[S][j] = v4f_t{(float)q, 0.f, 0.f, 0.f};
    (void)src;
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[S][j] = src[256 * j];
    if (tid < kS2A / 16 - 1024) wr[S][4] = src[1024];
#endif
  };
  auto store = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
#ifdef S2H_NO_STORE
    if (xr[S][0][0][0] == 12345.f && wr[S][0][0] == 54321.f)  // (keeps the loads alive)
#endif
    {
    v4f_t *Aw = reinterpret_cast<v4f_t *>(smb + buf * bufB) + tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) Aw[256 * j] = wr[S][j];
    if (tid < kS2A / 16 - 1024) Aw[1024] = wr[S][4];
    f16x8 *Xh = reinterpret_cast<f16x8 *>(smb + buf * bufB + kS2A), *Xl = Xh + g.units;
#pragma unroll
    for (int r = 0; r < kS2NP; ++r) {
      if (256 * r < g.pairs && own[r]) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
          f16x8 hi, lo;
#pragma unroll
          for (int c = 0; c < kS2C; ++c) {
            const float v = xr[S][r][c][col] * xmul[r];
            const _Float16 h = (_Float16)v;
            hi[c] = h;
            lo[c] = (_Float16)(v - (float)h);
          }
          const int u = u0[r] + (col ? 1 : g.PWh);
          Xh[u] = hi;
          Xl[u] = lo;
        }
      }
    }
    }
  };

  // ---- MFMA operand addresses
  const int aoff = (lhi * kS2K + l31) * 16;  // + tap 2 048 + cout block 512
  int boff;                                  // this lane's output pixel inside an X plane (bytes), tap (0, 0)
  int pn, py, px;
  {
    const int p = wave * 32 + l31;
    const int ti = p / g.per, rem = p - ti * g.per;
    const int ty = rem / a.Wo, x = rem - ty * a.Wo;
    boff = (ti * g.IU + 2 * ty * g.RU + x) * 16;
    pn = n0 + ti;
    py = y0 + ty;
    px = x;
  }
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  auto mfma_chunk = [&](int buf) {
#ifdef S2H_NO_MFMA
    (void)buf;
    return;
#endif
    const char *A = smb + buf * bufB + aoff;
    const char *Xh = smb + buf * bufB + kS2A + boff, *Xl = Xh + xbytes;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int toff = (ky * g.RU + (kx & 1) * g.PWh + (kx >> 1)) * 16;
      const f16x8 a0 = *reinterpret_cast<const f16x8 *>(A + tap * 2048), a1 = *reinterpret_cast<const f16x8 *>(A + tap * 2048 + 512);
      const f16x8 bh = *reinterpret_cast<const f16x8 *>(Xh + toff), bl = *reinterpret_cast<const f16x8 *>(Xl + toff);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bh, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bl, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bl, acc[1], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  load(S0{}, 0);
  if (g.nch > 1) load(S1{}, 1);
  store(S0{}, 0);
  __syncthreads();
  for (int q = 0; q < g.nch; q += 2) {
    // chunk q sits in buffer 0, set 1 holds chunk q + 1
    if (q + 2 < g.nch) load(S0{}, q + 2);
    mfma_chunk(0);
    if (q + 1 < g.nch) store(S1{}, 1);
    __syncthreads();
    if (q + 1 >= g.nch) break;
    if (q + 3 < g.nch) load(S1{}, q + 3);
    mfma_chunk(1);
    if (q + 2 < g.nch) store(S0{}, 0);
    __syncthreads();
  }

  // ---- epilogue: D[row = cout][col = pixel] -> NCHW; the power-of-two pre-scales come off in the fma that adds the bias
  const float oscale = reinterpret_cast<const float *>(a.w_wino44h + (size_t)a.Cout * Cin * 18)[1];
  const bool emit = a.stats_out != nullptr;  // (launch_conv_s2h passes it only where conv_s2h_stats_parts(d) > 0)
  float2 *const red = reinterpret_cast<float2 *>(smb);  // [wave][cout 64] (the operand buffers are free: the loop ends on a barrier)
  {
    const bool valid = pn < a.B;
    const size_t obase = ((size_t)min(pn, a.B - 1) * a.Cout + ct * kS2K + 4 * lhi) * (a.Ho * a.Wo) + (size_t)py * a.Wo + px;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dco = 32 * i + (r & 3) + 8 * (r >> 2);
        const float b = a.bias ? a.bias[ct * kS2K + 4 * lhi + dco] : 0.f;
        const float v = __builtin_fmaf(acc[i][r], oscale, b);
        if (valid) a.out[obase + (size_t)dco * (a.Ho * a.Wo)] = v;
        if (emit) {  // the next GroupNorm's statistics: this wave's 32 pixels of the cout
          const float2 mq = half_moments(v, lhi);
          if (l31 == 31) red[wave * kS2K + 4 * lhi + dco] = mq;
        }
      }
  }
  if (emit) {
    __syncthreads();
    const int W = g.per / 32, nimg = kS2P / g.per;  // waves per image share, image shares per tile
    const int parts = g.TI == 1 ? g.TPI : 1, part = g.TI == 1 ? (int)(ptile % g.TPI) : 0;
    for (int e = tid; e < nimg * kS2K; e += 256) {
      const int ti = e / kS2K, c = e - ti * kS2K;
      const int n = n0 + ti;
      if (n < a.B)
        *reinterpret_cast<float2 *>(a.stats_out + (((size_t)n * a.Cout + ct * kS2K + c) * parts + part) * 2) =
            merge_half_moments(red + (ti * W) * kS2K + c, kS2K, W);
    }
  }
}

// ---- the chip-filling form: 128 couts x FOUR 128-pixel tiles per workgroup -------------------------------------------------
// conv_s2h_kernel moves 2.35 GB per launch through L2 (5 TB/s, close to what the chip streams into LDS): every workgroup
// re-fetches its chunk's weights and both cout tiles stage the same input window.  Here a workgroup (512 threads, one per CU)
// keeps a chunk's weights of TWO cout tiles (36 KB) resident while FOUR consecutive pixel tiles pass under them -- eight
// accumulator tiles per wave (64 couts x 32 pixels x 4 tiles = 128 registers): weight traffic / 4, input traffic / 2.
// A step = (chunk q, tile sub): the next step's input window is requested before the step's 36 MFMAs and split / stored after
// them; the next chunk's weights are requested at sub 0 and stored after sub 3; one barrier per step.
constexpr int kS2A2 = 2 * kS2A;  // a chunk's weights of two cout tiles

__global__ __launch_bounds__(512, 1) void conv_s2h4_kernel(const ddpm_conv_desc a, const S2Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smb[];
  typedef float v4f_t __attribute__((ext_vector_type(4)));
  typedef float v2f_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb2 = wave >> 2, pb = wave & 3;  // cout tile of the pair, 32-pixel block of the tile
  const int Cin = a.C1, HWin = a.Hi * a.Wi, HWo = a.Ho * a.Wo;
  const int CT2 = g.CT / 2, PG = (g.PT + 3) / 4;

  const unsigned xcd = blockIdx.x & 7, mm = blockIdx.x >> 3;
  const unsigned pgrp = (mm / CT2) * 8 + xcd;
  const int ct2 = mm % CT2;
  if (pgrp >= (unsigned)PG) return;
  const int xbytes = g.units * 16;
  char *const XB = smb + 2 * kS2A2;  // X buffers: [2][hi | lo]

  // ---- staging roles: column pair e = tid (one round: pairs <= 512) of each of the four tiles
  const bool own = tid < g.pairs;
  int u0 = 0;
  const float *xsrc[4];
  float xmul[4];
  {
    int ti = 0, row = 0, j = 0;
    if (own) {
      ti = tid / (g.R * a.Wo);
      const int rem = tid - ti * (g.R * a.Wo);
      row = rem / a.Wo;
      j = rem - row * a.Wo;
      u0 = ti * g.IU + row * g.RU + j;
    }
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int st = 4 * pgrp + sub;
      int n0, y0;
      if (g.TI == 1) {
        n0 = st / g.TPI;
        y0 = (st - n0 * g.TPI) * g.TH;
      } else {
        n0 = st * g.TI;
        y0 = 0;
      }
      const int n = n0 + ti, rin = 2 * y0 - 1 + row;
      const bool ok = own && st < g.PT && n < a.B && rin >= 0 && rin < a.Hi;
      xsrc[sub] = a.in1 + (ok ? (size_t)(n * Cin) * HWin + rin * a.Wi + 2 * j : 0);
      xmul[sub] = ok ? kS2XScale : 0.f;
    }
  }
  for (int e = tid; e < 4 * g.TI * g.R; e += 512) {  // column -1: a zero unit per row, both buffers and planes, once
    const int w = e / (g.TI * g.R), rr = e - w * (g.TI * g.R);
    const int ti = rr / g.R, row = rr - ti * g.R;
    f16x8 z;
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = (_Float16)0.f;
    reinterpret_cast<f16x8 *>(XB + (w >> 1) * 2 * xbytes + (w & 1) * xbytes)[ti * g.IU + row * g.RU] = z;
  }
  // weights: 2 304 units of 16 bytes per chunk; unit u -> cout tile u / 1 152 of the pair
  const v4f_t *wsrc[5];
  int wu[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int u = j < 4 ? tid + 512 * j : 2048 + (tid & 255);
    wu[j] = u;
    const int t = u / (kS2A / 16), r = u - t * (kS2A / 16);
    wsrc[j] = reinterpret_cast<const v4f_t *>(a.w_wino44h + (size_t)(2 * ct2 + t) * g.nch * (kS2A / 2)) + r;
  }
  v2f_t xr[kS2C];
  v4f_t wr[5];
  auto load_x = [&](auto subc, int q) {
    constexpr int SUB = decltype(subc)::value;
#pragma unroll
    for (int c = 0; c < kS2C; ++c) xr[c] = *reinterpret_cast<const v2f_t *>(xsrc[SUB] + (size_t)(q * kS2C + c) * HWin);
  };
  auto store_x = [&](auto subc, int xbuf) {
    constexpr int SUB = decltype(subc)::value;
    if (!own) return;
    f16x8 *Xh = reinterpret_cast<f16x8 *>(XB + xbuf * 2 * xbytes), *Xl = Xh + g.units;
#pragma unroll
    for (int col = 0; col < 2; ++col) {
      f16x8 hi, lo;
#pragma unroll
      for (int c = 0; c < kS2C; ++c) {
        const float v = xr[c][col] * xmul[SUB];
        const _Float16 h = (_Float16)v;
        hi[c] = h;
        lo[c] = (_Float16)(v - (float)h);
      }
      const int u = u0 + (col ? 1 : g.PWh);
      Xh[u] = hi;
      Xl[u] = lo;
    }
  };
  auto load_w = [&](int q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[j] = wsrc[j][(size_t)q * (kS2A / 16)];
    if (tid < 256) wr[4] = wsrc[4][(size_t)q * (kS2A / 16)];
  };
  auto store_w = [&](int abuf) {
    v4f_t *Aw = reinterpret_cast<v4f_t *>(smb + abuf * kS2A2);
#pragma unroll
    for (int j = 0; j < 4; ++j) Aw[wu[j]] = wr[j];
    if (tid < 256) Aw[wu[4]] = wr[4];
  };

  // ---- MFMA operand addresses
  const int aoff = cb2 * kS2A + (lhi * kS2K + l31) * 16;
  int boff, pti, pty, pxx;
  {
    const int p = pb * 32 + l31;
    pti = p / g.per;
    const int rem = p - pti * g.per;
    pty = rem / a.Wo;
    pxx = rem - pty * a.Wo;
    boff = (pti * g.IU + 2 * pty * g.RU + pxx) * 16;
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
  auto mfma_step = [&](auto subc, int abuf, int xbuf) {
    constexpr int SUB = decltype(subc)::value;
    const char *A = smb + abuf * kS2A2 + aoff;
    const char *Xh = XB + xbuf * 2 * xbytes + boff, *Xl = Xh + xbytes;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int toff = (ky * g.RU + (kx & 1) * g.PWh + (kx >> 1)) * 16;
      const f16x8 a0 = *reinterpret_cast<const f16x8 *>(A + tap * 2048), a1 = *reinterpret_cast<const f16x8 *>(A + tap * 2048 + 512);
      const f16x8 bh = *reinterpret_cast<const f16x8 *>(Xh + toff), bl = *reinterpret_cast<const f16x8 *>(Xl + toff);
      acc[SUB][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bh, acc[SUB][0], 0, 0, 0);
      acc[SUB][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh, acc[SUB][1], 0, 0, 0);
      acc[SUB][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bl, acc[SUB][0], 0, 0, 0);
      acc[SUB][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bl, acc[SUB][1], 0, 0, 0);
    }
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>;

  load_w(0);
  load_x(T0{}, 0);
  store_w(0);
  store_x(T0{}, 0);
  __syncthreads();
  for (int q = 0; q < g.nch; ++q) {
    const int ab = q & 1;  // steps 4 q + sub use X buffer sub & 1
    const bool more = q + 1 < g.nch;
    if (more) load_w(q + 1);
    load_x(T1{}, q);
    mfma_step(T0{}, ab, 0);
    store_x(T1{}, 1);
    __syncthreads();
    load_x(T2{}, q);
    mfma_step(T1{}, ab, 1);
    store_x(T2{}, 0);
    __syncthreads();
    load_x(T3{}, q);
    mfma_step(T2{}, ab, 0);
    store_x(T3{}, 1);
    __syncthreads();
    if (more) load_x(T0{}, q + 1);
    mfma_step(T3{}, ab, 1);
    if (more) {
      store_w(ab ^ 1);
      store_x(T0{}, 0);
    }
    __syncthreads();
  }

  const float oscale = reinterpret_cast<const float *>(a.w_wino44h + (size_t)a.Cout * Cin * 18)[1];
  const int co0 = (2 * ct2 + cb2) * kS2K + 4 * lhi;
  const bool emit = a.stats_out != nullptr;
  float2 *const red = reinterpret_cast<float2 *>(smb);  // (the operand buffers are free: the step loop ends on a barrier)
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int st = 4 * pgrp + sub;
    int n0, y0;
    if (g.TI == 1) {
      n0 = st / g.TPI;
      y0 = (st - n0 * g.TPI) * g.TH;
    } else {
      n0 = st * g.TI;
      y0 = 0;
    }
    const int n = n0 + pti;
    const bool valid = st < g.PT && n < a.B;
    {
      const size_t obase = ((size_t)min(n, a.B - 1) * a.Cout + co0) * HWo + (size_t)(y0 + pty) * a.Wo + pxx;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = 32 * i + (r & 3) + 8 * (r >> 2);
          const float b = a.bias ? a.bias[co0 + dco] : 0.f;
          const float v = __builtin_fmaf(acc[sub][i][r], oscale, b);
          if (valid) a.out[obase + (size_t)dco * HWo] = v;
          if (emit) {  // [tile of the four][pixel block][cout of the pair's 128]
            const float2 mq = half_moments(v, lhi);
            if (l31 == 31) red[(sub * 4 + pb) * (2 * kS2K) + cb2 * kS2K + 4 * lhi + dco] = mq;
          }
        }
    }
  }
  if (emit) {
    __syncthreads();
    const int W = g.per / 32, nimg = kS2P / g.per;
    const int parts = g.TI == 1 ? g.TPI : 1;
    for (int e = tid; e < 4 * nimg * 2 * kS2K; e += 512) {
      const int sub = e / (nimg * 2 * kS2K), rem = e - sub * (nimg * 2 * kS2K);
      const int ti = rem / (2 * kS2K), c = rem - ti * (2 * kS2K);
      const int st = 4 * pgrp + sub;
      int n, part = 0;
      if (g.TI == 1) {
        n = st / g.TPI;
        part = st - n * g.TPI;
      } else {
        n = st * g.TI + ti;
      }
      if (st < g.PT && n < a.B)
        *reinterpret_cast<float2 *>(a.stats_out + (((size_t)n * a.Cout + (size_t)ct2 * 2 * kS2K + c) * parts + part) * 2) =
            merge_half_moments(red + (sub * 4 + ti * W) * (2 * kS2K) + c, 2 * kS2K, W);
    }
  }
}

int launch_conv_s2h(const ddpm_conv_desc &d, hipStream_t s) {
  S2Geom g;
  if (!s2h_geom(d, g)) {
    set_error("conv_s2h: unsupported shape");
    return DDPM_EINVAL;
  }
  const size_t lds = 2 * ((size_t)kS2A + 2 * (size_t)g.units * 16);
  ddpm_conv_desc dk = d;
  if (conv_s2h_stats_parts(d) == 0) dk.stats_out = nullptr;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const double M = (double)d.B * d.Ho * d.Wo;
  ProfScope prof(s, "conv3x3_s2h", 2.0 * M * d.Cout * (double)d.C1 * 9,
                 4.0 * ((double)d.B * d.C1 * d.Hi * d.Wi + M * d.Cout) + 2.0 * (double)d.Cout * d.C1 * 18);
  // launches of at least one workgroup per CU in the four-tile form take it (weights resident over four pixel tiles, both
  // cout tiles of a pair on one staged input window); DDPM_DOWN_S2H=2 / 3 force the small / the four-tile form (tests)
  const int force = sw().down_s2h;
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  const int PG = (g.PT + 3) / 4;
  const size_t lds4 = 2 * (size_t)kS2A2 + 4 * (size_t)g.units * 16;
  const bool can4 = d.Cout % (2 * kS2K) == 0 && g.pairs <= 512 && lds4 <= 160 * 1024;
  if (can4 && force != 2 && (force == 3 || (long)PG * (g.CT / 2) >= cus)) {
    static bool attr4_done = false;
    if (!attr4_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_s2h4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr4_done = true;
    }
    const unsigned grid4 = 8u * (unsigned)((PG + 7) / 8) * (unsigned)(g.CT / 2);
    hipLaunchKernelGGL(conv_s2h4_kernel, dim3(grid4), dim3(512), lds4, s, dk, g);
    DDPM_CHECK_LAUNCH();
    return 0;
  }
  const unsigned grid = 8u * (unsigned)((g.PT + 7) / 8) * (unsigned)g.CT;
  hipLaunchKernelGGL(conv_s2h_kernel, dim3(grid), dim3(256), lds, s, dk, g);
  DDPM_CHECK_LAUNCH();
  return 0;
}

// ---- weights: torch [Cout][Cin][3][3] -> 2^su w as f16 hi / lo planes in the order the kernel's LDS-DMA lands them:
//   [cout tile 64][chunk of 8 channels][tap 9][plane 2][cout 64][channel 8]   + two floats {max |w|, 1 / (2^3 2^su)}
__global__ void s2h_max_kernel(const float *__restrict__ src, unsigned *__restrict__ tail, int64_t total) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(src[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

__global__ void s2h_pack_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, int Cout, int Cin) {
  const int64_t total = (int64_t)Cout * Cin * 9;
  const int nch = Cin / kS2C;
  float *tail = reinterpret_cast<float *>(dst + (size_t)Cout * Cin * 18);
  int e = 0;
  const float wmax = tail[0];
  if (wmax > 0.f) (void)frexpf(wmax, &e);
  const int su = wmax > 0.f ? 15 - e : 0;  // max |2^su w| in [2^14, 2^15)
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[1] = ldexpf(1.f / kS2XScale, -su);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9), ci = (int)((i / 9) % Cin), o = (int)(i / (9 * (int64_t)Cin));
    const float w = ldexpf(src[i], su);
    const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
    const size_t base = ((((size_t)(o / kS2K) * nch + ci / kS2C) * 9 + tap) * 2) * kS2K;
    dst[(base + o % kS2K) * kS2C + ci % kS2C] = __builtin_bit_cast(uint16_t, hi);
    dst[(base + kS2K + o % kS2K) * kS2C + ci % kS2C] = __builtin_bit_cast(uint16_t, lo);
  }
}

size_t conv_s2h_weight_halves(int Cout, int Cin) {
  if (Cout % kS2K || Cin % kS2C) return 0;
  return (size_t)Cout * Cin * 18 + kS2Tail;
}

int launch_pack_conv_s2h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, hipStream_t s) {
  DDPM_CHECK_ARG(w_raw && dst && conv_s2h_weight_halves(Cout, Cin) != 0, "conv_s2h pack: Cout %% 64 or Cin %% 8 != 0");
  const int64_t total = (int64_t)Cout * Cin * 9;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  unsigned *tail = reinterpret_cast<unsigned *>(dst + (size_t)Cout * Cin * 18);
  hipError_t e = hipMemsetAsync(tail, 0, kS2Tail * sizeof(uint16_t), s);
  if (e != hipSuccess) {
    set_error("conv_s2h pack: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(s2h_max_kernel, dim3(blocks), dim3(256), 0, s, w_raw, tail, total);
  DDPM_CHECK_LAUNCH();
  hipLaunchKernelGGL(s2h_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, dst, Cout, Cin);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
