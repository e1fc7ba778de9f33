// conv_wino44.hip -- 3x3 stride-1 convolution as Winograd F(4x4, 3x3) on the fp32 MFMA pipe.
//
// Same fused op as conv_wino.hip (GroupNorm-affine + SiLU prologue, virtual concat, bias / temb / residual epilogue;
// reference call site /root/reference/src/trainers/reconstruct.py:151-153) with 36 multiplies per 16 outputs
// instead of F(2x2, 3x3)'s 16 per 4: 1.78x fewer MFMAs again (4x fewer than the direct form).  Every 4x4 output tile is
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A,   d_c = the 6x6 input patch of channel c,
// with the standard interpolation points (0, +-1, +-2, inf).  fp32 throughout; the larger transform constants cost
// accuracy: a 512-channel layer is 3e-6 rms / 5e-5 max against fp64 where F(2x2) is 5e-7 / 3e-6 and the direct
// form 2e-7 / 1.5e-6 -- over whole PLMS trajectories the per-image Z-scores move by 5e-6 (F(2x2): 3e-6; bar 1e-4;
// DESIGN.md section 3.4 has the measurement).
//
// Work item = 64 output channels x 32 tiles (512 output pixels: 4 tile rows of a 32x32 image, two 16x16 images,
// eight 8x8 images) x all input channels, by 8 waves = 2 (cout block of 32) x 4 (position group of 9): a wave owns
// 32 couts x 32 tiles at 9 of the 36 positions = 9 accumulator tiles = 144 AGPRs, two waves per SIMD.  Input channels
// advance in chunks of 4 (the 36-position operand images are 2.25x larger per channel than F(2x2)'s: U 36 KB + V 18 KB per
// chunk, double-buffered) -- 18 MFMAs per wave and chunk.
//
// Persistent workgroups and the staging pipeline follow conv_wino.hip: pixel loads three chunks ahead (two register sets,
// a load has more than a whole chunk to land), activation two ahead into a zero-bordered pixel tile, patch transform one
// ahead, the U tile by LDS-DMA (buffer form) one ahead.  The 6x6 transform of a (tile, channel) pair is split over three
// lanes by output row pair -- rows (0, 5), (1, 2), (3, 4) share their inputs; lane = (tile of 16, channel of 4) over a pixel
// tile padded so that the 64 patch reads of a wave hit 64 banks.
// At the end of an item the 36 positions of every (cout, tile) meet through LDS (four passes of four accumulator
// registers; the last operand buffer + one extra slab) and each lane finishes one cout x one 4x4 tile: float4 rows.
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

namespace {

// s_setprio 1 around every MFMA issue paid 3 % in conv_wino.hip; here (18 MFMAs per chunk, longer staging streams) the 36
// extra scalar instructions cost 1.7 %: off unless built with -DW44_SETPRIO
#ifndef W44_SETPRIO
#define W44_PRIO1 ""
#define W44_PRIO0 ""
#else
#define W44_PRIO1 "s_setprio 1\n\t"
#define W44_PRIO0 "\n\ts_setprio 0"
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kT = 32;               // tiles per item
constexpr int kK = 64;               // output channels per item
constexpr int kC = 4;                // input channels per chunk
constexpr int kX = 36;               // transform positions
constexpr int kUF = kX * kC * kK;    // U floats per chunk and cout tile (9216): [xi][cout 64][lhi][e], channel = 2 e + lhi
constexpr int kVF = kX * kC * kT;    // V floats per chunk (4608):              [xi][tile 32][lhi][e]
constexpr int kBUF = kUF + kVF;      // one operand buffer (13824 floats = 3 exchange slabs)
constexpr int kXS = kX * 2 * 64;     // exchange slab: [xi][cout block][lane] of one accumulator register (4608)
constexpr int kNDMA = kUF / 256;     // 1 KB LDS-DMA transfers per U tile (36)

__device__ __forceinline__ void mfma_a(f32x16 &c, float a, float b) {
  asm volatile(W44_PRIO1 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" W44_PRIO0 : "+a"(c) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void mfma_a_wait(f32x16 &c, float a, float b) {
  asm volatile(W44_PRIO1 "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" W44_PRIO0
               : "+a"(c) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma_a_first_wait(f32x16 &c, float a, float b) {  // C = 0
  asm volatile(W44_PRIO1 "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" W44_PRIO0
               : "=a"(c) : "v"(a), "v"(b), "n"(N));
}
// the ninth accumulator tile of a wave lives in arch VGPRs: hipcc gives a 512-thread kernel 128 + 128 registers
__device__ __forceinline__ void mfma_v(f32x16 &c, float a, float b) {
  asm volatile(W44_PRIO1 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" W44_PRIO0 : "+v"(c) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void mfma_v_wait(f32x16 &c, float a, float b) {
  asm volatile(W44_PRIO1 "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" W44_PRIO0
               : "+v"(c) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma_v_first_wait(f32x16 &c, float a, float b) {  // C = 0
  asm volatile(W44_PRIO1 "s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" W44_PRIO0
               : "=v"(c) : "v"(a), "v"(b), "n"(N));
}
// two floats 16 bytes apart through the scalar cache (lgkmcnt, not vmcnt); the pointer must be wave-uniform
__device__ __forceinline__ void sload2(const float *p, float &x0, float &x1) {
  asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(x0), "=&s"(x1) : "s"(p) : "memory");
}
__device__ __forceinline__ f2 lds_b64(int byte_addr, int imm) {
  f2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(imm));
  return v;
}

// 1-D input transform B^T w, B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(const float (&w)[6], float (&t)[6]) {
  const float p = __builtin_fmaf(-4.f, w[2], w[4]), q = __builtin_fmaf(-4.f, w[1], w[3]);
  const float r = w[4] - w[2], s = w[3] - w[1];
  t[0] = __builtin_fmaf(4.f, w[0], __builtin_fmaf(-5.f, w[2], w[4]));
  t[1] = p + q;
  t[2] = p - q;
  t[3] = __builtin_fmaf(2.f, s, r);
  t[4] = __builtin_fmaf(-2.f, s, r);
  t[5] = __builtin_fmaf(4.f, w[1], __builtin_fmaf(-5.f, w[3], w[5]));
}
// 1-D output transform A^T m, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
  const float s = m1 + m2, d = m1 - m2, u = m3 + m4, v = m3 - m4;
  y[0] = (m0 + s) + u;
  y[1] = __builtin_fmaf(2.f, v, d);
  y[2] = __builtin_fmaf(4.f, u, s);
  y[3] = __builtin_fmaf(8.f, v, d) + m5;
}

}  // namespace

struct W44Geom {
  int TWc, THr;     // tile columns / rows per image (Wo / 4, Ho / 4)
  int TI, TR;       // images per item, tile rows per item (per image)
  int parts;        // items per image along the rows
  int Cin, nchunks, HW;
  int prow;         // pixel-tile rows per image of an item: 4 TR + 2
  int PW, IS, PCH;  // pixel tile in LDS: row length (>= W + 2), image stride (>= prow PW), floats per channel plane; padded so
                    // that the 16 tiles x 4 channels a patch-stage wave reads land on 64 different banks
  int UI;           // staging units of 64 pixels per image of an item; a wave pair stages units hv, hv + 2, ...
  int NR;           // staging rounds per wave: ceil(TI * UI / 2), at most 5
  int KT, NIT, IPW, NS, grid;  // as conv_wino.hip: cout tiles, items per (cout tile, part), items per workgroup, slots
  int abl;          // ablation mask (DDPM_W44_ABL, experiments): 1 no patch transform, 2 no pixel stage, 4 no U DMA
  int xmap;         // 1: an XCD serves ONE cout tile (its L2 keeps that tile's U stream); 0: the cout tiles of a slot share an XCD
  // launches with fewer items than CUs (the 8x8 level at B = 256; small batches): S workgroups share an item, each walks
  // nchunks / S chunks of the channel stream and writes its partial output (after the output transform) into slab
  // `split` of a scratch buffer; conv_wino.hip's reduce pass adds the slabs in a fixed order with bias / temb / residual
  int S;            // channel-stream splits per item (1: none)
  long long pstride;  // floats between two partial-output slabs (0 when S == 1)
  // 3-D (dims = 3, VQ-VAE residual units; as conv_wino.hip): an "image" is one (n, d) slice, the chunk stream of an item
  // walks (depth tap, channel chunk) -- 2-D F(4x4) per depth tap, the taps accumulated in the transform domain
  int D;            // slices per batch item (1: plain 2-D)
  int NIMG;         // images the items walk: B * D
  int nch_c;        // channel chunks per depth tap; nchunks = nkd * nch_c
  int kd0, nkd;     // depth taps kd0 .. kd0 + nkd - 1 (a depth-1 volume only has its centre tap)
  int CS;           // channel stride of the input / output tensors in floats: D * HW
  int nkd_w;        // depth-tap slabs per cout tile in w_wino44: 3 for a 3x3x3 weight, else 1
};

static int w44_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// sizing: only decide the split (the scratch-size query); else a split launch needs its scratch slabs in the descriptor
static bool w44_geom(const ddpm_conv_desc &d, W44Geom &g, bool sizing = false) {
  const int Cin = d.C1 + d.C2;
  const bool is3d = d.dims == 3;
  if (d.ksize != 3 || (!is3d && (d.Di > 1 || d.Do > 1)) || d.mode != DDPM_CONV_NORMAL) return false;
  if ((d.out_act != DDPM_ACT_NONE && !(is3d && d.out_act == DDPM_ACT_RELU)) || d.act == DDPM_ACT_RELU) return false;
  if (d.gscale && d.act != DDPM_ACT_SILU) return false;  // the affine variant has SiLU built in
  // 3-D: no GroupNorm / activation prologue (zero padding along the depth must stay zero), no concat, no temb
  if (is3d && (d.gscale || d.act != DDPM_ACT_NONE || d.C2 || d.chan_add)) return false;
  const int Dd = is3d ? (d.Di > 1 ? d.Di : 1) : 1;
  if (is3d && (d.Do > 1 ? d.Do : 1) != Dd) return false;
  if (Cin % 8 || (d.C2 > 0 && d.C1 % kC) || d.Cout % kK) return false;  // an even number of 4-channel chunks
  if ((d.Ho & 3) || (d.Wo & 3) || d.Hi != d.Ho || d.Wi != d.Wo) return false;
  if ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.residual)) & 15) return false;  // float4 rows
  if ((double)d.B * (d.C1 > d.C2 ? d.C1 : d.C2) * Dd * d.Ho * d.Wo * 4 >= 2147483648.0) return false;  // 32-bit buffer offsets
  if ((double)d.B * d.Cout * Dd * d.Ho * d.Wo * 4 >= 2147483648.0 * 2) return false;
  g.TWc = d.Wo / 4;
  g.THr = d.Ho / 4;
  const int per_img = g.TWc * g.THr;
  if (per_img >= kT) {
    if (kT % g.TWc) return false;
    g.TI = 1;
    g.TR = kT / g.TWc;
    if (g.THr % g.TR) return false;
    g.parts = g.THr / g.TR;
  } else {
    if (kT % per_img) return false;
    g.TI = kT / per_img;
    g.TR = g.THr;
    g.parts = 1;
  }
  if (is3d && g.TI != 1) return false;  // slices smaller than 32 tiles stay on conv_wino.hip / the direct kernel
  g.Cin = Cin;
  g.D = Dd;
  g.NIMG = d.B * Dd;
  g.nch_c = Cin / kC;
  g.kd0 = is3d && Dd == 1 ? 1 : 0;
  g.nkd = is3d && Dd > 1 ? 3 : 1;
  g.nkd_w = is3d ? 3 : 1;
  g.nchunks = g.nkd * g.nch_c;
  g.HW = d.Ho * d.Wo;
  g.CS = Dd * g.HW;
  g.prow = 4 * g.TR + 2;
  // bank of a patch element = (ch PCH + ti IS + 4 tr PW + 4 tc + const) % 64 for the wave's 16 tiles x 4 channels:
  // 4 tc covers 4 TWc banks, so the next tile row must start 4 TWc banks further (PW = TWc mod 16), the next image
  // 4 TWc THr further, and the channel planes fill the residues mod 4 (PCH = 1 mod 4).  Without room: plain W + 2.
  auto layout = [&](bool pad) {
    g.PW = d.Wi + 2;
    if (pad && g.TWc < 16) g.PW += ((g.TWc - g.PW) % 16 + 16) % 16;
    g.IS = g.prow * g.PW;
    if (pad && g.TI > 1 && per_img < 16) g.IS += ((4 * per_img - g.IS) % 64 + 64) % 64;
    g.PCH = g.TI * g.IS;
    g.PCH += ((1 - g.PCH) % 4 + 4) % 4;
    return ((size_t)2 * kBUF + kXS + 2 * kC * g.PCH + 64) * sizeof(float) <= 160 * 1024;
  };
  if (!layout(true) && !layout(false)) return false;
  const int rows = g.prow < d.Hi ? g.prow : d.Hi;
  g.UI = (rows * d.Wi + 63) / 64;
  g.NR = (g.TI * g.UI + 1) / 2;
  if (g.NR > (g.TI == 1 ? 5 : 4)) return false;
  // closed-form round addressing of the kernel: rounds of one image are 128 pixels = whole rows apart; all rounds but the
  // last of a one-image item are fully inside the item's rows even for the parts at the image's top / bottom (one row less);
  // images of several units are exactly four, images of one unit are whole
  if (128 % d.Wi) return false;
  if (g.TI == 1 && (rows - 1) * d.Wi < 128 * (g.NR - 1)) return false;
  if (g.TI > 1 && !((g.UI == 4 && rows * d.Wi == 256) || (g.UI == 1 && rows * d.Wi <= 64 && g.TI <= 8))) return false;
  g.KT = d.Cout / kK;
  g.NIT = (g.NIMG + g.TI - 1) / g.TI;
  const long items = (long)g.KT * g.parts * g.NIT;
  const int cus = w44_cus();
  // small launches: conv_wino.hip (half-size items) and its channel-stream split.  DDPM_CONV_WINO44=2 lifts the rule (tests)
  const bool any_size = sw().conv_wino44 == 2;  // (tests flip it: ddpm_reload_env)
  g.S = 1;
  g.pstride = 0;
  if (items < cus && !any_size) {
    if (is3d) return false;
    // every workgroup of a split walks an even number (>= 4) of chunks; DDPM_WINO44_SPLIT caps S (0 / 1: no split)
    const int sp_max = sw().wino44_split;
    for (int sp = 4; sp >= 2; sp >>= 1)
      if (sp <= sp_max && items * sp <= cus && g.nchunks % (2 * sp) == 0 && g.nchunks / sp >= 4) { g.S = sp; break; }
    // below three quarters of the chip conv_wino.hip's half-size items (and its own split) are faster -- measured at B = 16:
    // `small` 16x16 layers (32 items x 4) 48 / 74 us here vs 43 / 63 us there, but the 768-channel 16x16 layers of `big`
    // (96 items x 2 = 192 workgroups) 3.07 here vs 3.85 ms there per forward
    if (g.S == 1 || items * g.S * 4 < (long)cus * 3) return false;
    const size_t out_floats = (size_t)d.B * d.Cout * g.HW;
    if (!sizing && (!d.scratch || d.scratch_floats < g.S * out_floats)) return false;
    g.pstride = (long long)out_floats;
  }
  g.IPW = (int)((items + cus - 1) / cus);
  g.NS = g.parts * ((g.NIT + g.IPW - 1) / g.IPW) * g.S;
  g.grid = g.KT * ((g.NS + 7) / 8) * 8;
  g.abl = sw().w44_abl;
  g.xmap = (sw().wino44_xmap >= 0 ? sw().wino44_xmap != 0 : 1) && (8 % g.KT == 0);
  if (g.xmap) g.grid = 8 * ((g.NS + 8 / g.KT - 1) / (8 / g.KT));
  return true;
}

bool conv_wino44_supported(const ddpm_conv_desc &d) {
  const bool enabled = sw().conv_wino44 != 0;
  W44Geom g;
  return enabled && d.w_wino44 != nullptr && !d.force_direct && w44_geom(d, g);
}

// floats of scratch with which this descriptor runs as a split F(4x4) launch (0: unsplit, or not an F(4x4) shape)
size_t conv_wino44_scratch_floats(const ddpm_conv_desc &d) {
  W44Geom g;
  if (sw().conv_wino44 == 0 || !d.w_wino44 || d.force_direct || !w44_geom(d, g, true) || g.S == 1) return 0;
  return (size_t)g.S * d.B * d.Cout * g.HW;
}

// GD = consecutive staging rounds of a wave that belong to one image (they share a GroupNorm scale / shift pair):
// NR for one-image items, UI / 2 when an image is an even number of 64-pixel units, else 1
// D3: the 3-D form (images = (n, d) slices, chunk stream = (depth tap, channel chunk)); a template parameter so that the
// 2-D instantiations carry none of its address arithmetic
template <bool AFFINE, int NR, int GD, bool RES, bool D3 = false>
__global__ __launch_bounds__(512, 2) void conv_wino44_kernel(const ddpm_conv_desc a, const W44Geom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool ONEIMG = GD >= NR;
  constexpr int NGS = (NR + GD - 1) / GD;           // GroupNorm scale / shift pairs per chunk
  constexpr int NVM = NR + (AFFINE ? 2 * NGS : 0);  // vector-memory loads of one pixel stage
  float *const P = smem + 2 * kBUF + kXS;           // pixel tiles [2][4 channels][PCH] (zero borders) + 64 dump floats
  const int PB = kC * g.PCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb = wave & 1, pg = wave >> 1;  // MFMA role: cout block, position group (xi = 9 pg + x)
  const bool silu = a.act == DDPM_ACT_SILU;

  // ---- this workgroup's stream (as conv_wino.hip: the cout tiles of one slot share an XCD)
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  int kt = wj % g.KT, slot = (wj / g.KT) * 8 + xcd;
  if (g.xmap) {  // XCD x serves cout tile x % KT: every U chunk is fetched into that L2 once and hit by its other CUs
    kt = xcd % g.KT;
    slot = wj * (8 / g.KT) + xcd / g.KT;
  }
  if (slot >= g.NS) return;
  const int split = slot % g.S;  // the splits of an item sit on neighbouring slots
  slot /= g.S;
  const int part = slot % g.parts, it0 = (slot / g.parts) * g.IPW;
  const int nitems = min(g.IPW, g.NIT - it0);
  const int r0 = part * g.TR;
  const int n_first = it0 * g.TI, n_end = n_first + nitems * g.TI;
  const int ch_lo = split * (g.nchunks / g.S), last = ch_lo + g.nchunks / g.S - 1;  // this workgroup's chunk range
  float *const outp = a.out + (size_t)split * g.pstride;  // S > 1: slab `split` of the scratch buffer

  // ---- staging roles
  // pixels: channel sc of the chunk, units hv, hv + 2, ... of 64 pixels (unit u = image u / UI, pixels 64 (u % UI) ..)
  const int sc = wave & 3, hv = wave >> 2;
  const int row_lo = max(0, 4 * r0 - 1), row_hi = min(a.Ho, 4 * (r0 + g.TR) + 1);
  const int npx = (row_hi - row_lo) * a.Wo;
  // Round k of this wave = unit u = hv + 2 k.  Its pixel offset and pixel-tile slot follow from round 0 in closed form
  // (five address pairs per lane were the first thing hipcc spilled):
  //   one image (GD >= NR):  64 u pixels further: + 512 k bytes, + k (128 / W) rows of the tile; only the last round can hold
  //                          pixels past the item's rows, it keeps its own (masked) pair
  //   GD == 2 (four units per image):  k & 1 as above, k >> 1 = next image;     GD == 1 (one unit per image): image u
  int pix0, pw0, pixL = 0, pwL = 0;
  {
    const int e = hv * 64 + lane;
    const bool valid = e < npx;
    pix0 = valid ? ((row_lo + e / a.Wo) * a.Wo + e % a.Wo) * 4 : (int)0x80000000;  // out of range: the load returns 0
    pw0 = valid ? sc * g.PCH + (hv >= g.UI ? hv * g.IS : 0) + (row_lo + e / a.Wo - (4 * r0 - 1)) * g.PW + e % a.Wo + 1
                : 2 * PB + lane;
    if (GD == 1 && hv >= g.UI) {  // one unit per image: wave half hv starts at image hv
      pix0 = lane < npx ? ((row_lo + lane / a.Wo) * a.Wo + lane % a.Wo) * 4 : (int)0x80000000;
      pw0 = lane < npx ? sc * g.PCH + hv * g.IS + (row_lo + lane / a.Wo - (4 * r0 - 1)) * g.PW + lane % a.Wo + 1
                       : 2 * PB + lane;
    }
    if (ONEIMG) {
      const int eL = e + 128 * (NR - 1);
      const bool vL = eL < npx;
      pixL = vL ? ((row_lo + eL / a.Wo) * a.Wo + eL % a.Wo) * 4 : (int)0x80000000;
      pwL = vL ? sc * g.PCH + (row_lo + eL / a.Wo - (4 * r0 - 1)) * g.PW + eL % a.Wo + 1 : 2 * PB + lane;
    }
  }
  const int prs = (128 / a.Wo) * g.PW;  // pixel-tile floats between a lane's pixels of consecutive rounds of one image
  auto pix_of = [&](int k) { return ONEIMG ? (k == NR - 1 ? pixL : pix0 + 512 * k) : GD == 2 ? pix0 + 512 * (k & 1) : pix0; };
  auto pw_of = [&](int k) {
    return ONEIMG ? (k == NR - 1 ? pwL : pw0 + k * prs) : GD == 2 ? pw0 + (k & 1) * prs + (k >> 1) * g.IS : pw0 + 2 * k * g.IS;
  };
  auto img_of = [&](int k) { return ONEIMG ? 0 : GD == 2 ? (k >> 1) : min(hv + 2 * k, g.TI - 1); };  // wave-uniform
  // patches: lane = (tile of 16, channel of 4), wave = (tile half, output-row pair): rows (0, 5), (1, 2), (3, 4).  Waves 6 and
  // 7 repeat the work of waves 0 and 1 (same values to the same addresses): they share SIMDs 2 and 3 with waves 2 and 3, so
  // the copies are off the critical path and the stage needs no wave-dependent branch.
  const int trio = (wave / 3) & 1, third = wave % 3;
  const int st = trio * 16 + (lane & 15), tch = lane >> 4;
  int tbase;
  {
    const int per = g.TR * g.TWc;
    const int ti = st / per, rem = st - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tbase = tch * g.PCH + ti * g.IS + 4 * tr * g.PW + 4 * tc;
  }
  int vofs = kUF + st * 4 + (tch & 1) * 2 + (tch >> 1);  // + xi * 128
  // Output rows (rowA, rowB) of B^T d, each a combination of three input rows -- the same instruction stream for all:
  //   rows (0, 5): A = d4 - 5 d2 + 4 d0,  B = d5 - 5 d3 + 4 d1
  //   rows (1, 2): p = d4 - 2 d2 - 2 d2,  q = d3 - 2 d1 - 2 d1,  A = p + q, B = p - q     (the halves are exact)
  //   rows (3, 4): p = d4 - d2/2 - d2/2,  q = d3 - d1/2 - d1/2,  A = p + 2 q, B = p - 2 q
  const bool t0 = third == 0;
  const int rowA = t0 ? 0 : third == 1 ? 1 : 3, rowB = t0 ? 5 : third == 1 ? 2 : 4;
  const int rA2 = t0 ? 0 : 2, rB0 = t0 ? 5 : 3, rB1 = t0 ? 3 : 1;
  const float c1 = t0 ? -5.f : third == 1 ? -2.f : -0.5f, c2 = t0 ? 4.f : c1, bm = t0 ? 0.f : third == 1 ? 1.f : 2.f;

  const int bytes1 = a.B * a.C1 * (D3 ? g.CS : g.HW) * 4, bytes2 = a.B * a.C2 * g.HW * 4;
  const __amdgpu_buffer_rsrc_t rs_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sh =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  int vzero;  // keeps the uniform scale / shift loads on the vector memory path (see conv_wino.hip)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

  // ---- MFMA operands: A = U [xi][cout][lhi][e], B = V [xi][tile][lhi][e]; one ds_read_b64 = both k-steps of a position
  // (a wave's 64 lanes read 512 consecutive bytes; the patch stage writes 64 consecutive floats)
  const int ub = (cb * 32 + l31) * 4 + lhi * 2 + 9 * pg * kC * kK;
  const int vb = kUF + l31 * 4 + lhi * 2 + 9 * pg * kC * kT;

  f32x16 acc[8], acc8;  // positions 0..7 of the group in AGPRs, position 8 in arch VGPRs
  float praw[2][NR], gs[2][NGS], gh[2][NGS], drow[6], wA[6], wB[6];

  // transfer j of this wave's share of the U tile of chunk ch (1 KB each; waves 0..3 have five, the others four).
  // Buffer form: resource + scalar offset + one loop-invariant lane offset.  (With global_load_lds hipcc kept five 64-bit
  // per-lane addresses, spilled them, and every reload -- a scratch load -- waited vmcnt(0): all pixel loads drained
  // five times per chunk.)
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.w_wino44), 0, (int)((size_t)kX * a.Cout * g.Cin * g.nkd_w * 4), 0x00020000);
  const int ulane = lane * 16;
  const int ukt = D3 ? (kt * g.nkd_w + g.kd0) * g.nch_c : kt * g.nchunks;
  int u_soff = 0;  // byte offset of this wave's first transfer of the chunk being fetched
  auto dma_u = [&](int j, int nb) {
    if (j < 4 || wave < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_u, (__attribute__((address_space(3))) void *)(smem + nb + (wave + 8 * j) * 256), 16, ulane,
          u_soff + j * 8192, 0, 0);
  };
  // pixel stage addressing of the chunk being loaded: resource / channel offsets once per chunk, image offset per round
  __amdgpu_buffer_rsrc_t l_rs;
  int l_cx, l_cgl, l_cg;
  int l_soff3 = 0, l_dok = 1;  // D3: byte offset of the (batch item, channel, slice) plane; depth tap inside the volume
  int nL = n_first, chL = ch_lo;  // stream position of the pixel-load stage
  auto load_setup = [&](int ch) {
    l_cg = ch * kC + sc;
    if (D3) {  // stream chunk -> (depth tap, channel chunk); image -> (batch item, slice).  One image per item: once per chunk
      // (kept as two scalar divisions: tracking the position incrementally cost four more live SGPRs and hipcc spilled
      // vector registers inside the chunk loop instead -- see activate_px)
      const int kdi = ch / g.nch_c;
      l_cg = (ch - kdi * g.nch_c) * kC + sc;
      const int ni = min(nL, g.NIMG - 1);
      const int nb = ni / g.D, dsl = ni - nb * g.D + g.kd0 + kdi - 1;
      l_dok = dsl >= 0 && dsl < g.D;
      l_soff3 = ((nb * a.C1 + l_cg) * g.D + (l_dok ? dsl : 0)) * g.HW * 4;
    }
    const bool first = l_cg < a.C1;
    l_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2,
                                             0x00020000);
    l_cx = first ? a.C1 : a.C2;
    l_cgl = first ? l_cg : l_cg - a.C1;
  };
  auto load_px = [&](auto setc, int k, int n) {
    constexpr int S = decltype(setc)::value;
    const int ni = min(n + img_of(k), g.NIMG - 1);
    // D3: a depth tap outside the volume reads zeros -- the range check of a raw buffer load is on the VGPR offset, and
    // 0x80000000 is past every resource (as for the halo pixels in pix0)
    const int soff = D3 ? l_soff3 : (ni * l_cx + l_cgl) * g.HW * 4;
    const int voff = D3 && !l_dok ? (int)0x80000000 : pix_of(k);
    praw[S][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(l_rs, voff, soff, 0));
    if (AFFINE && k % GD == 0) {
      const int goff = (ni * g.Cin + l_cg) * 4;
      gs[S][k / GD] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, vzero, goff, 0));
      gh[S][k / GD] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, vzero, goff, 0));
    }
  };
  auto activate_px = [&](auto setc, int k, int pb) {
    constexpr int S = decltype(setc)::value;
    const float x = praw[S][k];
    if (AFFINE) {
      const float sa = gs[S][k / GD], sb = gh[S][k / GD];
      const float v = __builtin_fmaf(x, sa, sb);
      const float t = __builtin_fmaf(x, -1.44269504088896341f * sa, -1.44269504088896341f * sb);
      P[pb + pw_of(k)] = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
    } else {
      // (the 3-D form never has an input activation, but skipping the SiLU there -- tried together with an incremental
      // stream position -- made hipcc spill inside the chunk loop: scratch reloads next to the MFMAs, each a vmcnt(0)
      // drain, 37.8 -> 47.9 ms of F(4x4) time per two-volume decode; left as is)
      const float sv = silu_fast(x);
      P[pb + pw_of(k)] = silu ? sv : x;
    }
  };
  // patch transform of chunk c + 1 in nine steps (one per even MFMA step)
  auto rd = [&](int r, int pb) {
    const float *p = P + pb + tbase + r * g.PW;
#pragma unroll
    for (int j = 0; j < 6; ++j) drow[j] = p[j];
  };
  auto commit = [&](const float (&w)[6], int row, int nb) {
    float t[6];
    bt6(w, t);
    float *vl = smem + nb + vofs + row * 6 * (kC * kT);
#pragma unroll
    for (int j = 0; j < 6; ++j) vl[j * (kC * kT)] = t[j];
  };
  auto tstep = [&](int s, int pb, int nb) {
    if (s == 0) {
      rd(4, pb);
    } else if (s == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wA[j] = drow[j];
      rd(2, pb);
    } else if (s == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wA[j] = __builtin_fmaf(c1, drow[j], wA[j]);
      rd(rA2, pb);
    } else if (s == 3) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wA[j] = __builtin_fmaf(c2, drow[j], wA[j]);
      rd(rB0, pb);
    } else if (s == 4) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wB[j] = drow[j];
      rd(rB1, pb);
    } else if (s == 5) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wB[j] = __builtin_fmaf(c1, drow[j], wB[j]);
      rd(1, pb);
    } else if (s == 6) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float q = __builtin_fmaf(c2, drow[j], wB[j]), p = wA[j];
        wA[j] = __builtin_fmaf(bm, q, p);
        wB[j] = t0 ? q : __builtin_fmaf(-bm, q, p);
      }
    } else if (s == 7) {
      commit(wA, rowA, nb);
    } else if (s == 8) {
      commit(wB, rowB, nb);
    }
  };
  auto advance = [&](int &n, int &ch) {
    if (ch < last) {
      ++ch;
    } else if (n + g.TI < n_end) {
      n += g.TI;
      ch = ch_lo;
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- prologue: zero borders; pixel tiles of stream chunks 0 and 1; U and V of chunk 0; registers for chunk 2
  for (int i = tid; i < 2 * PB + 64; i += 512) P[i] = 0.f;
  u_soff = ((ukt + ch_lo) * kUF + wave * 256) * 4;
#pragma unroll
  for (int j = 0; j < 5; ++j) dma_u(j, 0);
  __syncthreads();
  load_setup(chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(S0{}, k, nL);
#pragma unroll
  for (int k = 0; k < NR; ++k) activate_px(S0{}, k, 0);
  advance(nL, chL);
  load_setup(chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(S1{}, k, nL);
#pragma unroll
  for (int k = 0; k < NR; ++k) activate_px(S1{}, k, PB);
  advance(nL, chL);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 9; ++s) tstep(s, 0, 0);
  load_setup(chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(S0{}, k, nL);
  advance(nL, chL);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NVM) : "memory");
  __builtin_amdgcn_sched_barrier(0);

  // One chunk = 9 positions x 2 k-steps = 18 MFMA steps per wave.  Staging slices pinned to the steps:
  //   step 0..2   LDS-DMA of the U tile of chunk c + 1          step 3..7    loads of pixel round s - 3 of chunk c + 3
  //   step 13..17 activation of the rounds of chunk c + 2 (loaded during chunk c - 1: 1.5 chunks to land)
  //   step 0, 2, .., 16  patch-transform step s / 2 of chunk c + 1
  // The chunk closes with vmcnt(NVM): the DMAs (issued first) have landed, this chunk's pixel loads stay in flight.
  auto chunk = [&](auto parc, auto firstc, int ch_cur) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr int cbuf = PAR * kBUF, nb = (1 - PAR) * kBUF;
    const int pb_t = (1 - PAR) * PB, pb_a = PAR * PB;
    const int ch_u = ch_cur < last ? ch_cur + 1 : ch_lo;
    // the staging addresses are derived from a handful of per-lane bases inside the chunk; hidden from the optimiser here,
    // else it hoists every base + offset combination out of the loop (twenty-odd registers) and spills them -- and a
    // spill reload is a scratch load: vmcnt(0), all pixel loads drained
    asm volatile("" : "+v"(pix0), "+v"(pw0), "+v"(pixL), "+v"(pwL), "+v"(tbase), "+v"(vofs));
    u_soff = ((ukt + ch_u) * kUF + wave * 256) * 4;
    load_setup(chL);
    f2 av[3], bv[3];
    const int ua = (cbuf + ub) * 4, va = (cbuf + vb) * 4;  // bytes
    auto load_pair = [&](int slot, int x) {
      av[slot] = lds_b64(ua, x * (kC * kK * 4));
      bv[slot] = lds_b64(va, x * (kC * kT * 4));
    };
    auto slice = [&](int s) {
#ifdef W44_ABLATION  // timing experiments only
      if (s < 3 && !(g.abl & 4)) {
        dma_u(2 * s, nb);
        if (s < 2) dma_u(2 * s + 1, nb);
      }
      if (!(g.abl & 2)) {
        if (s >= 3 && s < 3 + NR) load_px(std::integral_constant<int, 1 - PAR>{}, s - 3, nL);
        if (s >= 18 - NR) activate_px(std::integral_constant<int, PAR>{}, s - (18 - NR), pb_a);
      }
      if ((s & 1) == 0 && !(g.abl & 1)) tstep(s >> 1, pb_t, nb);
#else
      if (s < 3) {
        dma_u(2 * s, nb);
        if (s < 2) dma_u(2 * s + 1, nb);
      }
      if (s >= 3 && s < 3 + NR) load_px(std::integral_constant<int, 1 - PAR>{}, s - 3, nL);
      if (s >= 18 - NR) activate_px(std::integral_constant<int, PAR>{}, s - (18 - NR), pb_a);
      if ((s & 1) == 0) tstep(s >> 1, pb_t, nb);
#endif
    };
    load_pair(0, 0);
    load_pair(1, 1);
    load_pair(2, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 9; ++x) {
      if (x == 8) {
        if (FIRST) mfma_v_first_wait<0>(acc8, av[x % 3][0], bv[x % 3][0]);
        else mfma_v_wait<0>(acc8, av[x % 3][0], bv[x % 3][0]);
      } else if (FIRST) {
        if (x < 7) mfma_a_first_wait<4>(acc[x & 7], av[x % 3][0], bv[x % 3][0]);
        else mfma_a_first_wait<2>(acc[x & 7], av[x % 3][0], bv[x % 3][0]);
      } else {
        if (x < 7) mfma_a_wait<4>(acc[x & 7], av[x % 3][0], bv[x % 3][0]);
        else mfma_a_wait<2>(acc[x & 7], av[x % 3][0], bv[x % 3][0]);
      }
      slice(2 * x);
      __builtin_amdgcn_sched_barrier(0);
      if (x == 8) mfma_v(acc8, av[x % 3][1], bv[x % 3][1]);
      else mfma_a(acc[x & 7], av[x % 3][1], bv[x % 3][1]);
      if (x + 3 < 9) load_pair(x % 3, x + 3);
      slice(2 * x + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance(nL, chL);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NVM) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

#ifdef W44_PROBE  // timing experiment: cycle stamps of workgroup 0, wave 0 -> desc.scratch[item][4]
  int probe_item = 0;
#define W44_STAMP(i)                                                                              \
  if (blockIdx.x == 0 && tid == 0 && a.scratch)                                                   \
    reinterpret_cast<unsigned long long *>(a.scratch)[probe_item * 4 + (i)] = __builtin_readcyclecounter();
#else
#define W44_STAMP(i)
#endif
  for (int n_cur = n_first; n_cur < n_end; n_cur += g.TI) {
    W44_STAMP(0)
    chunk(S0{}, std::true_type{}, ch_lo);
    chunk(S1{}, std::false_type{}, ch_lo + 1);
    W44_STAMP(1)
    for (int ch = ch_lo + 2; ch <= last; ch += 2) {
      chunk(S0{}, std::false_type{}, ch);
      chunk(S1{}, std::false_type{}, ch + 1);
    }

    W44_STAMP(2)
    // ---- end of an item: Y = A^T M A.  The item's last chunk (odd) consumed operand buffer 1; buffer 1 + the extra slab
    // are four exchange slabs [xi][cout block][lane], one per accumulator register of a pass.  Pass q moves registers
    // 4 q .. 4 q + 3 of all 36 positions through them; wave (cb, pg) then finishes register 4 q + pg of cout block cb:
    // cout = 32 cb + 8 q + 4 lhi + pg, tile = l31.
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' 16 passes
    int elane = lane;  // keeps the epilogue's addressing out of the chunk loop's live ranges
    asm volatile("" : "+v"(elane));
    const int el31 = elane & 31, elhi = elane >> 5;
    const int per = g.TR * g.TWc;
    const int ti = el31 / per, rem = el31 - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    const int n = n_cur + ti, ncl = min(n, g.NIMG - 1);
    const int nbat = D3 ? ncl / g.D : ncl, dsl_o = D3 ? ncl - nbat * g.D : 0;  // (batch item, slice)
    const int cstr = D3 ? g.CS : g.HW;                                        // channel stride
    float *const XS = smem + kBUF;
    // Stores and loads share vmcnt and it retires in order: a load issued after a store cannot be waited for without
    // waiting for the store's acknowledgement.  So everything the epilogue reads is requested BEFORE the stores it could
    // queue behind: the per-channel addends of all four passes up front (scalar loads when the item is one image: they
    // count in lgkmcnt), the residual rows of pass q + 1 in the middle of pass q.
    float addv[4];
    if (ONEIMG) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co0 = kt * kK + cb * 32 + 8 * q + pg;  // lanes 0..31; lanes 32..63 hold co0 + 4
        float b0 = 0.f, b1 = 0.f, t0v = 0.f, t1v = 0.f;
        if (a.bias) sload2(a.bias + co0, b0, b1);
        if (a.chan_add) sload2(a.chan_add + (size_t)min(n_cur, g.NIMG - 1) * a.chan_add_stride + co0, t0v, t1v);
        addv[q] = elhi ? b1 + t1v : b0 + t0v;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = kt * kK + cb * 32 + 8 * q + 4 * elhi + pg;
        addv[q] = (a.bias ? a.bias[co] : 0.f) + (a.chan_add ? a.chan_add[(size_t)ncl * a.chan_add_stride + co] : 0.f);
      }
    }
    const size_t co_e = (size_t)kt * kK + cb * 32 + 4 * elhi + pg;
    const size_t obase0 = (D3 ? (((size_t)nbat * a.Cout + co_e) * g.D + dsl_o) * g.HW : ((size_t)ncl * a.Cout + co_e) * g.HW) +
                          (size_t)(4 * (r0 + tr)) * a.Wo + 4 * tc;  // pass q: + 8 q cstr
    v4f res[4];
    auto load_res = [&](int q) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        res[k] = *reinterpret_cast<const v4f *>(a.residual + obase0 + (size_t)(8 * q) * cstr + (size_t)k * a.Wo);
    };
    if (RES) load_res(0);
    auto pass = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const size_t obase = obase0 + (size_t)(8 * q) * cstr;
      {
        float *xw = XS + ((9 * pg) * 2 + cb) * 64 + elane;
#pragma unroll
        for (int x = 0; x < 9; ++x) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) xw[rr * kXS + x * 128] = x == 8 ? acc8[4 * q + rr] : acc[x & 7][4 * q + rr];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float *xr = XS + pg * kXS + cb * 64 + elane;  // + xi * 128
      const float ad = addv[q];
      // output rows (0, 1), then (2, 3): twelve live column sums instead of twenty-four, M is read twice
      auto half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        float w[2][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {  // columns of M through two rows of A^T
          const float m1 = xr[(1 * 6 + j) * 128], m2 = xr[(2 * 6 + j) * 128], m3 = xr[(3 * 6 + j) * 128],
                      m4 = xr[(4 * 6 + j) * 128];
          if (h == 0) {
            const float m0 = xr[(0 * 6 + j) * 128];
            w[0][j] = (m0 + (m1 + m2)) + (m3 + m4);
            w[1][j] = __builtin_fmaf(2.f, m3 - m4, m1 - m2);
          } else {
            const float m5 = xr[(5 * 6 + j) * 128];
            w[0][j] = __builtin_fmaf(4.f, m3 + m4, m1 + m2);
            w[1][j] = __builtin_fmaf(8.f, m3 - m4, m1 - m2) + m5;
          }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          constexpr int dummy = 0;
          (void)dummy;
          const int k = 2 * h + kk;
          float y[4];
          at4(w[kk][0], w[kk][1], w[kk][2], w[kk][3], w[kk][4], w[kk][5], y);
          v4f o = v4f{y[0] + ad, y[1] + ad, y[2] + ad, y[3] + ad};
          if (RES) o += res[k];
          if (D3 && a.out_act == DDPM_ACT_RELU) o = v4f{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
          if (n < g.NIMG) *reinterpret_cast<v4f *>(outp + obase + (size_t)k * a.Wo) = o;
        }
      };
      half(std::integral_constant<int, 0>{});
      v4f r01[2];  // next pass's rows 0, 1 go to fresh registers (this pass's rows 2, 3 are still needed)
      if (RES && q < 3) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          r01[k] = *reinterpret_cast<const v4f *>(a.residual + obase + (size_t)8 * cstr + (size_t)k * a.Wo);
      }
      half(std::integral_constant<int, 1>{});
      if (RES && q < 3) {
        res[0] = r01[0];
        res[1] = r01[1];
#pragma unroll
        for (int k = 2; k < 4; ++k)
          res[k] = *reinterpret_cast<const v4f *>(a.residual + obase + (size_t)8 * cstr + (size_t)k * a.Wo);
      }
      // nobody may overwrite the slabs (next pass, or the next chunk's staging) while a neighbour still reads them
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
#ifndef W44_SKIP_EPILOGUE  // timing experiment only
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
#endif
    W44_STAMP(3)
#ifdef W44_PROBE
    ++probe_item;
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

int launch_conv_wino44(const ddpm_conv_desc &d, hipStream_t s) {
  W44Geom g;
  if (!d.w_wino44 || !w44_geom(d, g)) {
    set_error("conv_wino44: unsupported shape");
    return DDPM_EINVAL;
  }
  const size_t lds = ((size_t)2 * kBUF + kXS + 2 * kC * g.PCH + 64) * sizeof(float);
  typedef void (*kern_t)(const ddpm_conv_desc, const W44Geom);
  // shapes: one-image items (five rounds), two rounds per image (16x16 images), one round per image
#define W44_K(A, N, G) {conv_wino44_kernel<A, N, G, false>, conv_wino44_kernel<A, N, G, true>}
  static const kern_t kerns[2][3][2] = {{W44_K(false, 5, 5), W44_K(false, 4, 2), W44_K(false, 4, 1)},
                                        {W44_K(true, 5, 5), W44_K(true, 4, 2), W44_K(true, 4, 1)}};
#undef W44_K
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 12; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 6][i / 2 % 3][i % 2]),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  ddpm_conv_desc dk = d;  // the descriptor the kernel sees
  if (g.S > 1) {          // partial sums go to the scratch slabs, the addends to the reduce pass
    dk.out = d.scratch;
    dk.bias = nullptr;
    dk.chan_add = nullptr;
    dk.residual = nullptr;
  }
  const int shape = g.TI == 1 ? 0 : g.UI == 4 ? 1 : 2;
  kern_t kern = kerns[d.gscale ? 1 : 0][shape][dk.residual ? 1 : 0];
  if (d.dims == 3) {  // only reached without prologue and with whole slices per item (w44_geom)
    static const kern_t kerns3d[2] = {conv_wino44_kernel<false, 5, 5, false, true>, conv_wino44_kernel<false, 5, 5, true, true>};
    static bool attr3_done = false;
    if (!attr3_done) {
      for (int i = 0; i < 2; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns3d[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
      attr3_done = true;
    }
    kern = kerns3d[d.residual ? 1 : 0];
  }
  const double M = (double)g.NIMG * g.HW;
  // algorithmic work = the direct convolution's: 2 M Cout Cin 9 (x 3 depth taps); 36 / 144 of it is executed
  const double flops = 2.0 * M * d.Cout * (double)g.Cin * 9 * g.nkd;
  const double bytes = 4.0 * (M * g.Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * g.Cin * 9 * g.nkd);
  const char *kname = d.dims == 3 ? "conv3d_wino44" : d.gscale ? "conv3x3_wino44_gn_silu" : "conv3x3_wino44";
  char kshape[160];
  if (g_prof_on && sw().prof_shapes) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  hipLaunchKernelGGL(kern, dim3(g.grid), dim3(512), lds, s, dk, g);
  DDPM_CHECK_LAUNCH();
  if (g.S > 1) {
    ddpm_conv_desc dr = d;
    dr.stats_out = nullptr;  // (ddpm_conv_stats_parts is 0 for this kernel: the field is ignored)
    return launch_wino_split_reduce(dr, g.S, g.pstride, g.HW, s);
  }
  return 0;
}

// ---- weights: torch [Cout][Cin][3][3] -> U = G g G^T (6 x 6), packed as the LDS image the kernel's MFMAs read:
//   [cout tile 64][chunk 4][xi 36][cout 64][lhi 2][e 2],  channel of the chunk = 2 e + lhi
// G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], evaluated in double
// A 3x3x3 weight (nkd = 3) is transformed per depth tap kd: slab [cout tile][kd][chunk] holds G w[:, :, kd] G^T.
__global__ void wino44_pack_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin, int nkd) {
  const int64_t total = (int64_t)Cout * Cin * nkd;
  const int nchunks = Cin / kC;
  const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},   {0, 0, 1}};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kd = (int)(i % nkd);
    const int ci = (int)((i / nkd) % Cin), o = (int)(i / ((int64_t)nkd * Cin));
    const float *w = src + ((size_t)o * Cin + ci) * 9 * nkd + kd * 9;
    double t[6][3];
    for (int r = 0; r < 6; ++r)
      for (int j = 0; j < 3; ++j) t[r][j] = G[r][0] * w[0 * 3 + j] + G[r][1] * w[1 * 3 + j] + G[r][2] * w[2 * 3 + j];
    const int tile = o / kK, k64 = o % kK, ch = ci / kC, cl = ci % kC;
    const int lhi = cl & 1, e = cl >> 1;
    float *d = dst + (((size_t)tile * nkd + kd) * nchunks + ch) * kUF + k64 * 4 + lhi * 2 + e;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c)
        d[(r * 6 + c) * (kC * kK)] = (float)(t[r][0] * G[c][0] + t[r][1] * G[c][1] + t[r][2] * G[c][2]);
  }
}

size_t wino44_weight_floats(int Cout, int Cin) {
  if (Cout % kK || Cin % 8) return 0;
  return (size_t)kX * Cout * Cin;
}

int launch_pack_wino44_weight(const float *w_raw, float *w_wino44, int Cout, int Cin, hipStream_t s, int nkd) {
  DDPM_CHECK_ARG(wino44_weight_floats(Cout, Cin) != 0 && (nkd == 1 || nkd == 3), "wino44 pack: Cout %% 64 or Cin %% 8 != 0");
  const int64_t total = (int64_t)Cout * Cin * nkd;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wino44_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_wino44, Cout, Cin, nkd);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
