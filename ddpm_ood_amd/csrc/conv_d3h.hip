// conv_d3h.hip -- the ResnetBlock 3x3 convolution (stride 1, padding 1; generative's ResnetBlock.conv1 / conv2 inside
// DiffusionModelUNet.forward, reference call site /root/reference/src/trainers/reconstruct.py:151-153, layer list
// /root/reference/src/trainers/base.py:66-86) as a DIRECT convolution on the f16 MFMA pipe with split-f16 operands.  Round 4.
//
// Why next to conv_wino44h.hip.  The Winograd F(4x4) kernel executes 1.0x the algorithmic FLOPs on the matrix pipe but is bound
// by its V-transform producers: the MFMA pipe is 15-19 % busy (profiles/r03_pmc_per_kernel_b1024.csv), 395-446 algorithmic
// TFLOP/s.  A direct convolution needs no transform at all -- the only per-element work is GroupNorm + SiLU + the hi / lo split,
// done ONCE per staged input element and reused by nine taps x 128 output channels -- at the price of executing 3 x 10/9 x the
// algorithmic FLOPs on a pipe that has room for them (2.5 PFLOP/s dense f16): MFMA-bound instead of issue-bound.
//
// Arithmetic: x' = 2^3 act(x) (2^0 without a GroupNorm prologue), w' = 2^su w (su per layer: max |w'| in [2^14, 2^15));
// x' = xh + xl, w' = wh + wl with h = f16(.), l = f16(. - h) (exact differences); every fp32 product is
//     x' w' ~= wh xh + wh xl + wl xh        (the dropped wl xl is <= 2^-22 of the product; fp32 accumulate in the MFMA)
// and 1 / (2^3 2^su) comes off in the epilogue's fma.  No Winograd gain: operands up to 8 188 (65 504 without prologue).
//
// Tiling.  Workgroup = 128 couts x 512 pixels (TR whole rows of one image, or two whole 16x16 images), 8 waves = 2 cout halves x
// 4 pixel quarters, a wave owns 64 x 128 = 2 x 4 MFMA tiles (128 accumulator registers, pinned in a[0:127]).  Input channels
// advance in chunks of 8; one v_mfma_f32_32x32x16_f16 K-step multiplies 8 channels of TWO taps (lanes 0-31: tap 2 j, lanes
// 32-63: tap 2 j + 1; the tenth "tap" of the fifth step is a zero weight): 5 K-steps x 8 tiles x 3 products = 120 MFMAs per wave
// and chunk.  LDS per buffer (two buffers):
//   A  [tap 10][plane 2][cout 128] units of 8 channels (f16x8) = 40 KB, contiguous per (cout tile, chunk) in the packed weights
//      (ddpm_pack_conv_d3h_weight) -> 40 LDS-DMA pieces of 1 KB, five per wave, one behind each K-step
//   X  [plane 2][haloed pixel] units: (TR + 2) rows x S units, S = row stride chosen per width so that every ds_read_b128 lane
//      group of 16 falls on 16 different 16-byte bank groups whatever the tap offset (W = 64: 66, 32: 34, 16: 32); the halo and
//      the rows outside the image are zeros (written once: staging only ever writes in-image pixels)
// Staging: a lane = four pixels x four channels: four 16-byte loads (+ the GroupNorm pairs) one chunk ahead of their use, affine +
// SiLU + split dealt out between the MFMAs, lane pairs swap halves by DPP and store whole 16-byte units.  One barrier per chunk,
// with a counted vmcnt that leaves the prefetch in flight.
//
// STATUS: correct (tests/test_gpu_ops.py::test_conv_direct_split_f16_vs_conv2d, <= 1e-6 relative), NOT on the default path
// (DDPM_CONV_D3H=0): on the benchmark's layers at B = 1 024 it is 17-22 % SLOWER than the F(4x4) kernel (384 -> 128 at 32x32:
// 2 513 vs 2 066 us; six-layer sum 6 632 vs 5 685 us).  Measured anatomy of the 2 513 us (static ablations, tools/d3h_abl.sh):
// MFMAs + weight DMA alone 1 500; everything but the MFMAs 830; without the pixel loads 1 972; without the LDS stores 2 115;
// without transcendentals 2 463.  The matrix half and the staging half ADD instead of overlapping although two waves share each
// SIMD, and 1 500 us for the MFMAs alone is already 73 % of the F(4x4) kernel's total: 3.33x the matrix work costs more than the
// transforms it saves.  DESIGN.md 3.11 has the table and what would have to change (a 2-product scheme, or f16 staging output by
// the producing layer).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kM = 128;                 // couts per workgroup
constexpr int kP = 512;                 // pixels per workgroup
constexpr int kCh = 8;                  // input channels per chunk
constexpr int kTaps = 10;               // nine taps + one zero tap (five K-steps of two)
constexpr int kAUnits = kTaps * 2 * kM; // f16x8 units of a chunk's weights (2 560 = 40 KB)
constexpr float kXScale = 8.f;          // 2^3 behind the GroupNorm + SiLU prologue
constexpr int kTailHalves = 64;         // behind the planes: float [0] = max |w|, float [1] = 1 / (2^3 2^su)

// ---- accumulators: eight 32x32 fp32 tiles per wave in a[0:127], addressed BY NAME inside the asm statements (as
// conv_wino44h.hip): left to the register allocator as C++ values, 128 accumulator registers + the staging half's temporaries
// did not fit hipcc's allocation -- it spilled three tiles per chunk (and every B-operand address) to scratch, each reload an
// s_waitcnt vmcnt(0) in the middle of the MFMA stream: 109 TFLOP/s.  An empty asm with the 128 clobbers tells the compiler the
// registers are in use; it allocates everything else in the 128 architectural VGPRs.
__device__ __forceinline__ void reserve_agprs() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
               "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31",
               "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47",
               "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63",
               "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79",
               "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95",
               "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109",
               "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123",
               "a124", "a125", "a126", "a127");
}
#define D3H_TILES(X) X(0, 0, 15) X(1, 16, 31) X(2, 32, 47) X(3, 48, 63) X(4, 64, 79) X(5, 80, 95) X(6, 96, 111) X(7, 112, 127)
#define D3H_REGS(X) \
  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) \
  X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) \
  X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) \
  X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64) X(65) X(66) X(67) X(68) X(69) X(70) X(71) X(72) X(73) \
  X(74) X(75) X(76) X(77) X(78) X(79) X(80) X(81) X(82) X(83) X(84) X(85) X(86) X(87) X(88) X(89) X(90) X(91) \
  X(92) X(93) X(94) X(95) X(96) X(97) X(98) X(99) X(100) X(101) X(102) X(103) X(104) X(105) X(106) X(107) X(108) \
  X(109) X(110) X(111) X(112) X(113) X(114) X(115) X(116) X(117) X(118) X(119) X(120) X(121) X(122) X(123) X(124) \
  X(125) X(126) X(127)
__device__ __forceinline__ void mfma_pin(int T, const f16x8 &a, const f16x8 &b) {  // tile T (0 .. 7) += a b
  switch (T) {
#define X(t, lo, hi)                                                                                          \
  case t:                                                                                                     \
    asm volatile("v_mfma_f32_32x32x16_f16 a[" #lo ":" #hi "], %0, %1, a[" #lo ":" #hi "]" ::"v"(a), "v"(b)); \
    break;
    D3H_TILES(X)
#undef X
  }
}
__device__ __forceinline__ void zero_pinned_tiles() {
#define X(r) asm volatile("v_accvgpr_write_b32 a" #r ", 0");
  D3H_REGS(X)
#undef X
}
__device__ __forceinline__ float read_pinned(int r) {  // register r = 16 T + element
  float v = 0.f;
  switch (r) {
#define X(n)                                                \
  case n:                                                   \
    asm volatile("v_accvgpr_read_b32 %0, a" #n : "=v"(v)); \
    break;
    D3H_REGS(X)
#undef X
  }
  return v;
}

struct D3Geom {
  int TR;        // image rows per window (TI == 1) / rows of an image (TI == 2)
  int TI;        // images per tile: 1, or 2 whole 16x16 images
  int S;         // units per haloed row
  int IU;        // units per image window: (TR + 2) * S
  int XU;        // units per X plane: TI * IU
  int TPI;       // tiles per image (TI == 1)
  int PT, CT;    // pixel tiles, cout tiles
  int nch;       // chunks
  int bufX;      // units per X buffer: 2 planes + one dump unit
  int items;     // staging items per chunk: TI * (TR + 2) * W
};

bool d3h_geom(const ddpm_conv_desc &d, D3Geom &g) {
  const int Cin = d.C1 + d.C2;
  if (d.ksize != 3 || d.mode != DDPM_CONV_NORMAL || d.dims == 3 || d.Di > 1 || d.Do > 1 || d.force_direct) return false;
  if (d.Hi != d.Ho || d.Wi != d.Wo) return false;
  if (d.out_act != DDPM_ACT_NONE || d.act == DDPM_ACT_RELU) return false;
  if (d.gscale && (d.act != DDPM_ACT_SILU || !d.gshift)) return false;
  if (!d.gscale && d.act != DDPM_ACT_NONE) return false;
  if (Cin % kCh || (d.C2 > 0 && d.C1 % kCh) || d.Cout % kM) return false;
  if (d.Wo != 64 && d.Wo != 32 && d.Wo != 16) return false;
  const int HW = d.Ho * d.Wo;
  if (HW >= kP) {
    if (HW % kP) return false;
    g.TI = 1;
    g.TR = kP / d.Wo;
    g.TPI = HW / kP;
    g.PT = d.B * g.TPI;
  } else {
    if (kP != 2 * HW) return false;  // two whole 16x16 images
    g.TI = 2;
    g.TR = d.Ho;
    g.TPI = 1;
    g.PT = (d.B + 1) / 2;
  }
  g.S = d.Wo == 16 ? 32 : d.Wo + 2;
  g.IU = (g.TR + 2) * g.S;
  g.XU = g.TI * g.IU;
  g.CT = d.Cout / kM;
  g.nch = Cin / kCh;
  g.bufX = 2 * g.XU + 1;
  g.items = g.TI * (g.TR + 2) * (d.Wo / 4) * 2;  // staging items: (4 consecutive pixels of a window row) x (half of the 8 channels)
  if (g.items > 512) return false;
  if ((reinterpret_cast<uintptr_t>(d.in1) | reinterpret_cast<uintptr_t>(d.in2) | reinterpret_cast<uintptr_t>(d.gscale) |
       reinterpret_cast<uintptr_t>(d.gshift)) & 15)
    return false;  // 16-byte loads
  if ((size_t)(2 * kAUnits + 2 * g.bufX) * 16 > 160 * 1024) return false;
  if ((double)d.B * (d.C1 > d.C2 ? d.C1 : d.C2) * d.Ho * d.Wo * 4 >= 2147483648.0) return false;  // 32-bit element offsets x 4
  return true;
}

// Lanes 2 k (channels 0-3 of the chunk) and 2 k + 1 (channels 4-7) hold the same four pixels.  Each stored its own 8-byte half of a
// pixel's unit: lanes 64 bytes apart, four lanes per LDS bank group -- the stores alone cost 390 us of the 384 -> 128 layer's 2 500
// (ablation D3H_NO_LDSW).  Instead the pair swaps halves (one DPP move per dword): the even lane assembles the whole units of pixels
// 0, 1, the odd lane those of pixels 2, 3, and a wave's 16-byte stores land 32 bytes apart -- no conflict in any group of eight lanes.
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 d3h_pair_unit(h4_t p, h4_t p2, bool odd) {  // p: pixel 0 (1) of the item, p2: pixel 2 (3)
  const v2i_t a = __builtin_bit_cast(v2i_t, p), b = __builtin_bit_cast(v2i_t, p2);
  v2i_t own, got;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int send = odd ? a[d] : b[d];
    own[d] = odd ? b[d] : a[d];
    got[d] = __builtin_amdgcn_mov_dpp(send, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
  }
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  const v4i_t u = odd ? v4i_t{got[0], got[1], own[0], own[1]} : v4i_t{own[0], own[1], got[0], got[1]};
  return __builtin_bit_cast(f16x8, u);
}

// LDS (f16x8 units): A buffers [2][kAUnits] | X buffers [2][plane hi | plane lo | dump]
template <bool AFFINE>
__global__ __launch_bounds__(512, 1) void conv_d3h_kernel(const ddpm_conv_desc a, const D3Geom g, const uint16_t *__restrict__ wq) {
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cw = wave & 1, pw = (wave >> 1) & 3;  // cout half (64), pixel quarter (128 pixels = four 32-pixel blocks)
  const int W = a.Wo, H = a.Ho, HW = H * W, Cin = a.C1 + a.C2;
  f16x8 *const Xb = lds + 2 * kAUnits;

  // workgroup -> (pixel tile, cout tile): the cout tiles of a pixel tile are neighbours on ONE XCD (they stage the same input)
  const unsigned xcd = blockIdx.x & 7, mm = blockIdx.x >> 3;
  const unsigned ptile = (mm / g.CT) * 8 + xcd;
  const int ct = mm % g.CT;
  if (ptile >= (unsigned)g.PT) return;
  const int n0 = g.TI == 1 ? ptile / g.TPI : ptile * 2;
  const int y0 = g.TI == 1 ? (ptile - n0 * g.TPI) * g.TR : 0;

  // ---- zero both X buffers once (halo columns, rows outside the image / images past the batch: never written again)
  {
    f16x8 z;
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = (_Float16)0.f;
    for (int e = tid; e < 2 * g.bufX; e += 512) Xb[e] = z;
  }

  // ---- staging role: item tid = (four consecutive in-image pixels of a window row) x (channels 4 h .. 4 h + 3 of the chunk):
  // four 16-byte loads per chunk.  (One pixel x eight channels per thread -- eight 4-byte loads -- made 384 vector-memory
  // instructions per chunk and CU with the GroupNorm pairs: the address unit, not the bytes, was the limit, 2.7 us per chunk.)
  typedef float v4f_t __attribute__((ext_vector_type(4)));
  int sbyte;               // byte offset of the first unit this lane stores (pixel 0 for even, pixel 2 for odd lanes; the dump unit for
                           // items without pixels)
  int sstep, slo;          // bytes to its second unit (16; 0 for the dump), from a hi to its lo unit (XU * 16; 0 for the dump)
  const bool odd = lane & 1;  // (== the item's channel half: the items of a wave start at an even index)
  unsigned o1, o2, og;     // element offsets (< 2^29, d3h_geom) of the item's first pixel in in1 / in2, of its image's pairs in gscale
  {
    // the items are dealt evenly to the eight waves (36 - 40 lanes each): every wave carries the same staging work between its MFMAs
    const int IPW = ((g.items + 15) / 16) * 2;
    const int e = lane < IPW ? wave * IPW + lane : g.items, hsel = e & 1, pg = e >> 1, W4 = W >> 2;
    const int ti = pg / ((g.TR + 2) * W4), rem = pg - ti * ((g.TR + 2) * W4);
    const int srow = rem / W4, scol = (rem - srow * W4) * 4;  // window row 0 .. TR + 1 <-> image row y0 - 1 + srow
    const int yin = y0 - 1 + srow, n = n0 + ti;
    const bool own = e < g.items && yin >= 0 && yin < H && n < a.B;
    const int u = ti * g.IU + srow * g.S + scol + 1;
    sbyte = own ? (u + 2 * hsel) * 16 : 2 * g.XU * 16;
    sstep = own ? 16 : 0;
    slo = own ? g.XU * 16 : 0;
    const int nn = own ? n : n0;
    const int spix = own ? yin * W + scol : 0;
    o1 = (unsigned)(nn * a.C1 * HW + spix) + (unsigned)(4 * hsel * HW);
    o2 = (unsigned)(nn * a.C2 * HW + spix) + (unsigned)(4 * hsel * HW);
    og = (unsigned)(nn * Cin + 4 * hsel);
  }
  const f16x8 *const wsrc = reinterpret_cast<const f16x8 *>(wq) + (size_t)ct * g.nch * kAUnits;

  // ---- MFMA operand addresses (units): A[tap][plane][cout], tap = 2 j + lhi; X[plane][pixel + tap offset]
  const int aoff = lhi * (2 * kM) + cw * 64 + l31;  // + j * 4 kM (two taps) + plane * kM + 32 i
  int xb[4], xt[5];  // this lane's pixel of block jb in window units (tap (0, 0)); tap offset of K-step j (tap 2 j + lhi; 9 -> 8)
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    const int pp = pw * 128 + jb * 32 + l31;
    const int ti = pp / (g.TR * W), rem = pp - ti * (g.TR * W);
    const int pr = rem / W, pc = rem - pr * W;
    xb[jb] = ti * g.IU + pr * g.S + pc;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = min(2 * j + lhi, 8);
    const int dy = tap / 3, dx = tap - 3 * dy;  // window coordinates: output (r, c) reads rows r + dy, columns c + dx
    xt[j] = dy * g.S + dx;
  }

  reserve_agprs();       // accumulator tile 4 i + jb = cout block i x pixel block jb, in a[0:127]
  zero_pinned_tiles();

  auto adma = [&](int q, int buf) {  // this wave's five 1 KB pieces of chunk q's weights
#ifdef D3H_NO_DMA
    if (q > 0) return;
#endif
#pragma unroll
    for (int p5 = 0; p5 < 5; ++p5) {
      const int piece = wave * 5 + p5;
      __builtin_amdgcn_global_load_lds(wsrc + (size_t)min(q, g.nch - 1) * kAUnits + piece * 64 + lane, lds + buf * kAUnits + piece * 64, 16,
                                       0, 0);
    }
  };
  // Pixels and their GroupNorm pairs are requested a whole chunk before they are used (registers: with the accumulators pinned
  // in AGPRs there is room): a staging half is then VALU + LDS stores, not an exposed memory round trip.
  v4f_t raw[4], gsa, gsb;  // [channel of the half] x four pixels; the half's four scale / shift values
  auto xload = [&](int q) {
    const int cg0 = min(q, g.nch - 1) * kCh;
    const bool first = cg0 < a.C1;  // (uniform: C1 % 8 == 0)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#ifdef D3H_NO_XLOAD  // (static ablations for tools/d3h_abl.sh: timing only, wrong results)
      raw[c] = v4f_t{(float)(cg0 + c), 1.f, 2.f, 3.f};
#else
      raw[c] = first ? *reinterpret_cast<const v4f_t *>(a.in1 + o1 + (unsigned)((cg0 + c) * HW))
                     : *reinterpret_cast<const v4f_t *>(a.in2 + o2 + (unsigned)((cg0 + c - a.C1) * HW));
#endif
    }
    if (AFFINE) {
      gsa = *reinterpret_cast<const v4f_t *>(a.gscale + og + (unsigned)cg0);
      gsb = *reinterpret_cast<const v4f_t *>(a.gshift + og + (unsigned)cg0);
    }
  };
  auto xstore = [&](int buf) {  // activation + split + store of the chunk held in the registers
#ifndef D3H_NO_XSTORE
    char *Xw = reinterpret_cast<char *>(Xb + buf * g.bufX) + sbyte;
    h4_t hi[4], lo[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float y = raw[c][px];
        if (AFFINE) {
          const float v = __builtin_fmaf(y, gsa[c], gsb[c]);
          const float t = __builtin_fmaf(y, -1.44269504088896341f * gsa[c], -1.44269504088896341f * gsb[c]);
          y = (kXScale * v) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
        }
        const _Float16 h = (_Float16)y;
        hi[px][c] = h;
        lo[px][c] = (_Float16)(y - (float)h);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f16x8 *>(Xw + p * sstep) = d3h_pair_unit(hi[p], hi[p + 2], odd);
      *reinterpret_cast<f16x8 *>(Xw + p * sstep + slo) = d3h_pair_unit(lo[p], lo[p + 2], odd);
    }
#endif
  };

  // One chunk = the 120 MFMAs of chunk q ("matrix half": A buffer PAR, X buffer PAR) + the weight DMA of chunk q + 1, the
  // activation / split / store of X(q + 1) and the loads of X(q + 2) ("staging half").  Waves w and w + 4 share a SIMD: waves 0-3
  // run the matrix half first, waves 4-7 the staging half first -- one wave of a SIMD feeds the matrix pipe while the other
  // issues VALU / memory instructions.
  // Why 512 pixels per workgroup: a CU takes in ~12 bytes per cycle (global -> LDS / registers).  With 256-pixel tiles a chunk
  // brought 40 KB of weights + 10 KB of pixels for 3 840 matrix cycles per SIMD -- 13 B per cycle: the kernel ran at the SUM of its
  // matrix time and its data time whatever the instruction order (2 760-2 900 us for the 384 -> 128 layer at B = 1 024; matrix
  // half alone 1 557, staging half alone 1 241).  Twice the pixels under the same weights: 60 KB for 7 680 cycles.
  auto chunk = [&](auto parc, int q) {
    constexpr int PAR = decltype(parc)::value;
    const f16x8 *A = lds + PAR * kAUnits + aoff;
    const f16x8 *Xh = Xb + PAR * g.bufX, *Xl = Xh + g.XU;
    char *Xw = reinterpret_cast<char *>(Xb + (PAR ^ 1) * g.bufX) + sbyte;
    // (hidden from loop-invariant code motion: hipcc precomputed all 2 x 40 B-operand addresses of both buffers before the
    // chunk loop, kept them in scratch and reloaded one -- with an s_waitcnt vmcnt(0) -- in front of every ds_read)
    int xbl[4], xtl[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xbl[k] = xb[k];
      asm volatile("" : "+v"(xbl[k]));
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      xtl[k] = xt[k];
      asm volatile("" : "+v"(xtl[k]));
    }
    const int cq1 = min(q + 1, g.nch - 1), cq2 = min(q + 2, g.nch - 1);
    const bool first2 = cq2 * kCh < a.C1;  // (uniform: C1 % 8 == 0)
    h4_t hi, lo, hi0, lo0;  // the pixel in the making; the first pixel of the pair (0 of 0, 2; 1 of 1, 3)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      f16x8 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = A[j * 4 * kM + 32 * i];
        al[i] = A[j * 4 * kM + kM + 32 * i];
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // pixel blocks 2 half, 2 half + 1 under the same weight operands
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          bh[jj] = Xh[xbl[2 * half + jj] + xtl[j]];
          bl[jj] = Xl[xbl[2 * half + jj] + xtl[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          const int prod = m >> 2, i = (m >> 1) & 1, jj = m & 1;  // cross terms first, the main term last
#ifndef D3H_NO_MFMA
          mfma_pin(4 * i + 2 * half + jj, prod == 1 ? al[i] : ah[i], prod == 0 ? bl[jj] : bh[jj]);
#endif
          // ---- the staging work dealt out behind MFMA number `slot` of the chunk (0 .. 119): every wave's VALU / memory
          // instructions issue while the SIMD's other wave has the matrix pipe
          const int slot = 24 * j + 12 * half + m;
#ifdef D3H_DMA_SPREAD
          if (slot % 24 == 1) {  // weights of chunk q + 1: one 1 KB piece per K-step
            const int pc = j;
#else
          if (slot >= 1 && slot <= 9 && (slot & 1)) {  // weights of chunk q + 1: the five 1 KB pieces right behind the barrier (a whole
            const int pc = slot >> 1;                    // chunk to land: the next barrier waits for them)
#endif
#ifdef D3H_NO_DMA
            if (q < 0)
#endif
            __builtin_amdgcn_global_load_lds(wsrc + (size_t)cq1 * kAUnits + (wave * 5 + pc) * 64 + lane,
                                             lds + (PAR ^ 1) * kAUnits + (wave * 5 + pc) * 64, 16, 0, 0);
          }
#ifndef D3H_NO_XSTORE
          if (slot >= 40 && slot < 120 && (slot - 40) % 5 == 0) {  // value (pixel px, channel c) of X(q + 1): slots 40, 45, .., 115
            const int vi = (slot - 40) / 5, pi = vi >> 2, c = vi & 3, px = (pi >> 1) + 2 * (pi & 1);  // pixels 0, 2, 1, 3
            float y = raw[c][px];
            if (AFFINE) {
              const float v = __builtin_fmaf(y, gsa[c], gsb[c]);
              const float t = __builtin_fmaf(y, -1.44269504088896341f * gsa[c], -1.44269504088896341f * gsb[c]);
#ifdef D3H_NO_TRANS  // (ablation: no transcendentals)
              y = (kXScale * v) * (1.0f + t);
#else
              y = (kXScale * v) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
#endif
            }
            const _Float16 h = (_Float16)y;
            hi[c] = h;
            lo[c] = (_Float16)(y - (float)h);
            if (c == 3 && !(pi & 1)) hi0 = hi, lo0 = lo;
#ifdef D3H_NO_LDSW
            if (c == 3 && (pi & 1) && y == 12345.f) {
#else
            if (c == 3 && (pi & 1)) {  // the pair's units
#endif
              *reinterpret_cast<f16x8 *>(Xw + (pi >> 1) * sstep) = d3h_pair_unit(hi0, hi, odd);
              *reinterpret_cast<f16x8 *>(Xw + (pi >> 1) * sstep + slo) = d3h_pair_unit(lo0, lo, odd);
            }
          }
#endif
          // X(q + 2) and its GroupNorm pairs into the SAME registers, each right behind its last use (slots 101, 106, 111, 116; 117,
          // 118): they have until slot 40 of the next chunk to land -- the barrier in between waits with a COUNTED vmcnt that
          // leaves exactly these six (four) youngest loads in flight
          if (slot == 101 || slot == 106 || slot == 111 || slot >= 116 && slot <= 118) {
            const int k = slot <= 116 ? (slot - 101) / 5 : slot - 113;
            const int cg0 = cq2 * kCh;
#ifndef D3H_NO_XLOAD
            if (k < 4)
              raw[k] = first2 ? *reinterpret_cast<const v4f_t *>(a.in1 + o1 + (unsigned)((cg0 + k) * HW))
                              : *reinterpret_cast<const v4f_t *>(a.in2 + o2 + (unsigned)((cg0 + k - a.C1) * HW));
#endif
            if (AFFINE && k == 4) gsa = *reinterpret_cast<const v4f_t *>(a.gscale + og + (unsigned)cg0);
            if (AFFINE && k == 5) gsb = *reinterpret_cast<const v4f_t *>(a.gshift + og + (unsigned)cg0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- prologue: A(0) and X(0) into buffers 0, X(1) into registers
  adma(0, 0);
  xload(0);
  __syncthreads();  // the zero fill is complete
  xstore(0);
  xload(1);
  for (int q = 0; q < g.nch; q += 2) {
    // every LDS-DMA piece and every load issued during the previous chunk has landed (hipcc does not wait for LDS-DMA writes of
    // an earlier loop iteration by itself); buffers PAR are complete and the other buffers' readers are done
    // (counted: the weight pieces are older than the pixel loads issued at the end of the chunk, which stay in flight)
    // (a raw s_barrier: __syncthreads() is a fence, in front of which hipcc waits for ALL loads)
    if (AFFINE) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    chunk(S0{}, q);
    if (q + 1 >= g.nch) break;
    // (a raw s_barrier: __syncthreads() is a fence, in front of which hipcc waits for ALL loads)
    if (AFFINE) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    chunk(S1{}, q + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped repeats past the last chunk: nothing may land in LDS after the exit
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' passes (asm MFMAs: no interlock by the compiler)

  // ---- epilogue: D[row = cout][col = pixel] -> NCHW (a 32-pixel block is 128 contiguous bytes)
  const float oscale = reinterpret_cast<const float *>(wq + (size_t)a.Cout * Cin * kTaps * 2)[1] * (AFFINE ? 1.f : kXScale);
  int elane = lane;
  asm volatile("" : "+v"(elane));  // (keeps the epilogue's index arithmetic -- integer divisions by W -- out of the chunk loop's
                                   // live ranges: hoisted, it spilled 250 registers)
  const int el31 = elane & 31, elhi = elane >> 5;
  const int co_base = ct * kM + cw * 64 + 4 * elhi;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    const int pp = pw * 128 + jb * 32 + el31;
    const int ti = pp / (g.TR * W), rem = pp - ti * (g.TR * W);
    const int n = n0 + ti;
    if (n < a.B) {
      const size_t obase = ((size_t)n * a.Cout + co_base) * HW + (size_t)y0 * W + rem;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
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
          a.out[obase + (size_t)dco * HW] = __builtin_fmaf(read_pinned(16 * (4 * i + jb) + r), oscale, add[r]) + rv[r];
        }
        __builtin_amdgcn_sched_barrier(0);  // (one tile's addends at a time: hoisted together they spilled the accumulators)
      }
    }
  }
}

// ---- weights: torch [Cout][Cin][3][3] -> [cout tile 128][chunk 8][tap 10][plane hi | lo][cout 128][8 ch] f16 of 2^su w, tap 9 = 0;
// tail floats {max |w|, 1 / (2^3 2^su)}
__global__ void d3h_max_kernel(const float *__restrict__ src, unsigned *__restrict__ tail, int64_t total) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(src[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __builtin_bit_cast(unsigned, m));  // (non-negative floats order like their bits)
}

__global__ void d3h_pack_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, int Cout, int Cin) {
  float *tail = reinterpret_cast<float *>(dst + (size_t)Cout * Cin * kTaps * 2);
  const float umax = tail[0];
  int e = 0;
  (void)frexpf(umax, &e);  // umax = m 2^e, m in [0.5, 1)
  const int su = umax > 0.f ? 15 - e : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[1] = ldexpf(1.f / kXScale, -su);
  const int nch = Cin / kCh;
  const int64_t total = (int64_t)Cout * Cin * kTaps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % kTaps);
    const int ci = (int)((i / kTaps) % Cin), co = (int)(i / ((int64_t)kTaps * Cin));
    const float w = tap < 9 ? ldexpf(src[((size_t)co * Cin + ci) * 9 + tap], su) : 0.f;
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    const int tile = co / kM, c128 = co % kM, chunk = ci / kCh, cc = ci % kCh;
    const size_t unit = (((size_t)tile * nch + chunk) * kTaps + tap) * 2 * kM + c128;  // plane 0
    dst[unit * 8 + cc] = h;
    dst[(unit + kM) * 8 + cc] = l;
  }
}

}  // namespace

size_t conv_d3h_weight_halves(int Cout, int Cin) {
  return (Cout % kM == 0 && Cin % kCh == 0) ? (size_t)Cout * Cin * kTaps * 2 + kTailHalves : 0;
}

int launch_pack_conv_d3h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, hipStream_t s) {
  DDPM_CHECK_ARG(w_raw && dst && conv_d3h_weight_halves(Cout, Cin) != 0, "conv_d3h pack: Cout %% 128 or Cin %% 8 != 0");
  unsigned *tail = reinterpret_cast<unsigned *>(dst + (size_t)Cout * Cin * kTaps * 2);
  hipError_t e = hipMemsetAsync(tail, 0, kTailHalves * 2, s);
  if (e != hipSuccess) {
    set_error("conv_d3h pack: %s", hipGetErrorString(e));
    return (int)e;
  }
  const int64_t n9 = (int64_t)Cout * Cin * 9;
  hipLaunchKernelGGL(d3h_max_kernel, dim3((unsigned)((n9 + 255) / 256 > 1024 ? 1024 : (n9 + 255) / 256)), dim3(256), 0, s, w_raw, tail, n9);
  DDPM_CHECK_LAUNCH();
  const int64_t total = (int64_t)Cout * Cin * kTaps;
  hipLaunchKernelGGL(d3h_pack_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, s, w_raw,
                     reinterpret_cast<_Float16 *>(dst), Cout, Cin);
  DDPM_CHECK_LAUNCH();
  return 0;
}

bool conv_d3h_supported(const ddpm_conv_desc &d) {
  D3Geom g;
  if (!split_f16_on(sw().conv_d3h != 0) || !d.w_d3h || !d3h_geom(d, g)) return false;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return sw().conv_d3h == 2 || (long)g.PT * g.CT >= cus;  // launches smaller than the chip stay on the Winograd kernels' splits
}

int launch_conv_d3h(const ddpm_conv_desc &d, hipStream_t s) {
  D3Geom g;
  if (!d.w_d3h || !d3h_geom(d, g)) {
    set_error("conv_d3h: unsupported shape");
    return DDPM_EINVAL;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_d3h_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_d3h_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int Cin = d.C1 + d.C2;
  const double M = (double)d.B * d.Ho * d.Wo;
  ProfScope prof(s, d.gscale ? "conv3x3_d3h_gn_silu" : "conv3x3_d3h", 2.0 * M * d.Cout * (double)Cin * 9,
                 4.0 * (M * Cin + M * d.Cout * (d.residual ? 2 : 1)) + 2.0 * (double)d.Cout * Cin * kTaps * 2);
  const size_t lds = (size_t)(2 * kAUnits + 2 * g.bufX) * sizeof(f16x8);
  const dim3 grid(8 * ((g.PT + 7) / 8) * g.CT);
  if (d.gscale)
    hipLaunchKernelGGL(conv_d3h_kernel<true>, grid, dim3(512), lds, s, d, g, d.w_d3h);
  else
    hipLaunchKernelGGL(conv_d3h_kernel<false>, grid, dim3(512), lds, s, d, g, d.w_d3h);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
