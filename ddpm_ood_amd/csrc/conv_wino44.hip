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
// ahead, the U tile by LDS-DMA one ahead.  The 6x6 transform of a (tile, channel) pair is split over three lanes by output
// row pair -- rows (0, 5), (1, 2), (3, 4) share their inputs -- so that six of the eight waves carry 48 VALU ops each.
// At the end of an item the 36 positions of every (cout, tile) meet through LDS (four passes of four accumulator
// registers; the last operand buffer + one extra slab) and each lane finishes one cout x one 4x4 tile: float4 rows.
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kT = 32;               // tiles per item
constexpr int kK = 64;               // output channels per item
constexpr int kC = 4;                // input channels per chunk
constexpr int kX = 36;               // transform positions
constexpr int kUF = kX * kC * kK;    // U floats per chunk and cout tile (9216): [xi][lhi][cout 64][e], channel = 2 e + lhi
constexpr int kVF = kX * kC * kT;    // V floats per chunk (4608):              [xi][lhi][tile 32][e]
constexpr int kBUF = kUF + kVF;      // one operand buffer (13824 floats = 3 exchange slabs)
constexpr int kXS = kX * 2 * 64;     // exchange slab: [xi][cout block][lane] of one accumulator register (4608)
constexpr int kNDMA = kUF / 256;     // 1 KB LDS-DMA transfers per U tile (36)

__device__ __forceinline__ void mfma_a(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_setprio 0" : "+a"(c) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void mfma_a_wait(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_setprio 0"
               : "+a"(c) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma_a_first_wait(f32x16 &c, float a, float b) {  // C = 0
  asm volatile("s_setprio 1\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0\n\ts_setprio 0"
               : "=a"(c) : "v"(a), "v"(b), "n"(N));
}
// the ninth accumulator tile of a wave lives in arch VGPRs: hipcc gives a 512-thread kernel 128 + 128 registers
__device__ __forceinline__ void mfma_v(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_setprio 0" : "+v"(c) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void mfma_v_wait(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_setprio 0"
               : "+v"(c) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma_v_first_wait(f32x16 &c, float a, float b) {  // C = 0
  asm volatile("s_setprio 1\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0\n\ts_setprio 0"
               : "=v"(c) : "v"(a), "v"(b), "n"(N));
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
  int PW, PCH;      // pixel tile in LDS: padded row length (W + 2), floats per channel plane
  int UI;           // staging units of 64 pixels per image of an item; a wave pair stages units hv, hv + 2, ...
  int NR;           // staging rounds per wave: ceil(TI * UI / 2), at most 5
  int KT, NIT, IPW, NS, grid;  // as conv_wino.hip: cout tiles, items per (cout tile, part), items per workgroup, slots
  int xmap;         // 1: an XCD serves ONE cout tile (its L2 keeps that tile's U stream); 0: the cout tiles of a slot share an XCD
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

static bool w44_geom(const ddpm_conv_desc &d, W44Geom &g) {
  const int Cin = d.C1 + d.C2;
  if (d.ksize != 3 || d.dims == 3 || d.Di > 1 || d.Do > 1 || d.mode != DDPM_CONV_NORMAL) return false;
  if (d.out_act != DDPM_ACT_NONE || d.act == DDPM_ACT_RELU) return false;
  if (d.gscale && d.act != DDPM_ACT_SILU) return false;  // the affine variant has SiLU built in
  if (Cin % 8 || (d.C2 > 0 && d.C1 % kC) || d.Cout % kK) return false;  // an even number of 4-channel chunks
  if ((d.Ho & 3) || (d.Wo & 3) || d.Hi != d.Ho || d.Wi != d.Wo) return false;
  if ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.residual)) & 15) return false;  // float4 rows
  if ((double)d.B * (d.C1 > d.C2 ? d.C1 : d.C2) * d.Ho * d.Wo * 4 >= 2147483648.0) return false;  // 32-bit buffer offsets
  if ((double)d.B * d.Cout * d.Ho * d.Wo * 4 >= 2147483648.0 * 2) return false;
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
  g.Cin = Cin;
  g.nchunks = Cin / kC;
  g.HW = d.Ho * d.Wo;
  g.prow = 4 * g.TR + 2;
  g.PW = d.Wi + 2;
  g.PCH = (g.TI * g.prow * g.PW) | 1;
  const int rows = g.prow < d.Hi ? g.prow : d.Hi;
  g.UI = (rows * d.Wi + 63) / 64;
  g.NR = (g.TI * g.UI + 1) / 2;
  if (g.NR > 5) return false;
  if (((size_t)2 * kBUF + kXS + 2 * kC * g.PCH + 64) * sizeof(float) > 160 * 1024) return false;
  g.KT = d.Cout / kK;
  g.NIT = (d.B + g.TI - 1) / g.TI;
  const long items = (long)g.KT * g.parts * g.NIT;
  const int cus = w44_cus();
  // small launches: conv_wino.hip (half-size items) and its channel-stream split.  DDPM_CONV_WINO44=2 lifts the rule (tests)
  const char *sw = getenv("DDPM_CONV_WINO44");  // read per call: tests flip it
  const bool any_size = sw && atoi(sw) == 2;
  if (items < cus && !any_size) return false;
  g.IPW = (int)((items + cus - 1) / cus);
  g.NS = g.parts * ((g.NIT + g.IPW - 1) / g.IPW);
  g.grid = g.KT * ((g.NS + 7) / 8) * 8;
  const char *xm = getenv("DDPM_WINO44_XMAP");
  g.xmap = (xm ? atoi(xm) != 0 : 1) && (8 % g.KT == 0);
  if (g.xmap) g.grid = 8 * ((g.NS + 8 / g.KT - 1) / (8 / g.KT));
  return true;
}

bool conv_wino44_supported(const ddpm_conv_desc &d) {
  const char *sw = getenv("DDPM_CONV_WINO44");
  const bool enabled = !(sw && atoi(sw) == 0);
  W44Geom g;
  return enabled && d.w_wino44 != nullptr && !d.force_direct && w44_geom(d, g);
}

template <bool AFFINE, int NR, bool ONEIMG>
__global__ __launch_bounds__(512, 2) void conv_wino44_kernel(const ddpm_conv_desc a, const W44Geom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NGS = ONEIMG ? 1 : NR;              // GroupNorm scale / shift pairs per chunk: one per round's image
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
  const int part = slot % g.parts, it0 = (slot / g.parts) * g.IPW;
  const int nitems = min(g.IPW, g.NIT - it0);
  const int r0 = part * g.TR;
  const int n_first = it0 * g.TI, n_end = n_first + nitems * g.TI;
  const int last = g.nchunks - 1;

  // ---- staging roles
  // pixels: channel sc of the chunk, units hv, hv + 2, ... of 64 pixels (unit u = image u / UI, pixels 64 (u % UI) ..)
  const int sc = wave & 3, hv = wave >> 2;
  const int row_lo = max(0, 4 * r0 - 1), row_hi = min(a.Ho, 4 * (r0 + g.TR) + 1);
  const int npx = (row_hi - row_lo) * a.Wo;
  int pix[NR], pw[NR], tik[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int u = hv + 2 * k, ti = u / g.UI, e = (u - ti * g.UI) * 64 + lane;
    const bool valid = ti < g.TI && e < npx;
    const int row = row_lo + e / a.Wo, col = e % a.Wo;
    tik[k] = min(ti, g.TI - 1);
    pix[k] = valid ? (row * a.Wo + col) * 4 : (int)0x80000000;  // out of range: the buffer load returns 0
    pw[k] = valid ? sc * g.PCH + (ti * g.prow + row - (4 * r0 - 1)) * g.PW + col + 1 : 2 * PB + lane;
  }
  // patches: waves 0..5; (tile, channel) pair = 64 (wave / 3) + lane, output-row pair = wave % 3: rows (0, 5), (1, 2), (3, 4)
  const bool tact = wave < 6;
  const int trio = wave / 3, third = wave - 3 * trio;
  const int st = lane & 31, tch = 2 * trio + (lane >> 5);
  int tbase;
  {
    const int per = g.TR * g.TWc;
    const int ti = st / per, rem = st - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tbase = tch * g.PCH + (ti * g.prow + 4 * tr) * g.PW + 4 * tc;
  }
  const int vofs = kUF + ((tch & 1) * kT + st) * 2 + (tch >> 1);  // + xi * 2 * kT * 2
  const int rowA = third == 0 ? 0 : third == 1 ? 1 : 3, rowB = third == 0 ? 5 : third == 1 ? 2 : 4;  // output rows
  const int rB0 = third == 0 ? 5 : 3, rB1 = third == 0 ? 3 : 1;                                        // input rows
  const float c1 = third == 0 ? -5.f : third == 1 ? -4.f : -1.f, beta = third == 1 ? 1.f : 2.f;

  const int bytes1 = a.B * a.C1 * g.HW * 4, bytes2 = a.B * a.C2 * g.HW * 4;
  const __amdgpu_buffer_rsrc_t rs_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sh =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  int vzero;  // keeps the uniform scale / shift loads on the vector memory path (see conv_wino.hip)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

  // ---- MFMA operands: A = U [xi][lhi][cout][e], B = V [xi][lhi][tile][e]; one ds_read_b64 = both k-steps of a position
  const int ub = (lhi * kK + cb * 32 + l31) * 2 + 9 * pg * kC * kK;
  const int vb = kUF + (lhi * kT + l31) * 2 + 9 * pg * kC * kT;

  f32x16 acc[8], acc8;  // positions 0..7 of the group in AGPRs, position 8 in arch VGPRs
  float praw[2][NR], gs[2][NGS], gh[2][NGS], drow[6], wA[6], wB[6];

  // transfer j of this wave's share of the U tile of chunk ch (1 KB each; waves 0..3 have five, the others four).
  // Buffer form: resource + scalar offset + one loop-invariant lane offset.  (With global_load_lds hipcc kept five 64-bit
  // per-lane addresses, spilled them, and every reload -- a scratch load -- waited vmcnt(0): all pixel loads drained
  // five times per chunk.)
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.w_wino44), 0, (int)((size_t)kX * a.Cout * g.Cin * 4), 0x00020000);
  const int ulane = lane * 16;
  const int ukt = kt * g.nchunks;
  auto dma_u = [&](int j, int ch, int nb) {
    const int i = wave + 8 * j;
    if (i < kNDMA) {
      const int soff = ((ukt + ch) * kUF + i * 256) * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (__attribute__((address_space(3))) void *)(smem + nb + i * 256), 16,
                                               ulane, soff, 0, 0);
    }
  };
  auto load_px = [&](auto setc, int k, int n, int ch) {
    constexpr int S = decltype(setc)::value;
    const int cg = ch * kC + sc, ni = min(n + tik[k], a.B - 1);
    const bool first = cg < a.C1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2, 0x00020000);
    const int soff = first ? (ni * a.C1 + cg) * g.HW * 4 : (ni * a.C2 + cg - a.C1) * g.HW * 4;
    praw[S][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pix[k], soff, 0));
    if (AFFINE && (!ONEIMG || k == 0)) {
      const int goff = (ni * g.Cin + cg) * 4;
      gs[S][ONEIMG ? 0 : k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, vzero, goff, 0));
      gh[S][ONEIMG ? 0 : k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, vzero, goff, 0));
    }
  };
  auto activate_px = [&](auto setc, int k, int pb) {
    constexpr int S = decltype(setc)::value;
    const float x = praw[S][k];
    if (AFFINE) {
      const float sa = gs[S][ONEIMG ? 0 : k], sb = gh[S][ONEIMG ? 0 : k];
      const float v = __builtin_fmaf(x, sa, sb);
      const float t = __builtin_fmaf(x, -1.44269504088896341f * sa, -1.44269504088896341f * sb);
      P[pb + pw[k]] = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
    } else {
      const float sv = silu_fast(x);
      P[pb + pw[k]] = silu ? sv : x;
    }
  };
  // patch transform of chunk c + 1, nine steps: V rows (rowA, rowB) = B^T d B restricted to this lane's output rows.
  //   rows (0, 5): A = d4 - 5 d2 + 4 d0, B = d5 - 5 d3 + 4 d1;   rows (1, 2): p = d4 - 4 d2, q = d3 - 4 d1, p +- q;
  //   rows (3, 4): p = d4 - d2, q = d3 - d1, p +- 2 q
  auto rd = [&](int r, int pb) {
    const float *p = P + pb + tbase + r * g.PW;
#pragma unroll
    for (int j = 0; j < 6; ++j) drow[j] = p[j];
  };
  auto commit = [&](const float (&w)[6], int row, int nb) {
    float t[6];
    bt6(w, t);
    float *vl = smem + nb + vofs + row * 6 * (2 * kT * 2);
#pragma unroll
    for (int j = 0; j < 6; ++j) vl[j * (2 * kT * 2)] = t[j];
  };
  auto tstep = [&](int s, int pb, int nb) {
    if (!tact) return;
    if (s == 0) {
      rd(4, pb);
    } else if (s == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wA[j] = drow[j];
      rd(2, pb);
    } else if (s == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wA[j] = __builtin_fmaf(c1, drow[j], wA[j]);
      if (third == 0) rd(0, pb);
    } else if (s == 3) {
      if (third == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) wA[j] = __builtin_fmaf(4.f, drow[j], wA[j]);
      }
      rd(rB0, pb);
    } else if (s == 4) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wB[j] = drow[j];
      rd(rB1, pb);
    } else if (s == 5) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wB[j] = __builtin_fmaf(c1, drow[j], wB[j]);
      if (third == 0) rd(1, pb);
    } else if (s == 6) {
      if (third == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) wB[j] = __builtin_fmaf(4.f, drow[j], wB[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float p = wA[j], q = wB[j];
          wA[j] = __builtin_fmaf(beta, q, p);
          wB[j] = __builtin_fmaf(-beta, q, p);
        }
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
      ch = 0;
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- prologue: zero borders; pixel tiles of stream chunks 0 and 1; U and V of chunk 0; registers for chunk 2
  for (int i = tid; i < 2 * PB + 64; i += 512) P[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j) dma_u(j, 0, 0);
  int nL = n_first, chL = 0;  // stream position of the pixel-load stage
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(S0{}, k, nL, chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) activate_px(S0{}, k, 0);
  advance(nL, chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(S1{}, k, nL, chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) activate_px(S1{}, k, PB);
  advance(nL, chL);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 9; ++s) tstep(s, 0, 0);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(S0{}, k, nL, chL);
  advance(nL, chL);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NVM) : "memory");
  __builtin_amdgcn_sched_barrier(0);

  // One chunk = 9 positions x 2 k-steps = 18 MFMA steps per wave.  Staging slices pinned to the steps:
  //   step 0..4   LDS-DMA of the U tile of chunk c + 1          step 5..9    loads of pixel round s - 5 of chunk c + 3
  //   step 10..14 activation of round s - 10 of chunk c + 2 (loaded during chunk c - 1)
  //   step 0, 2, .., 16  patch-transform step s / 2 of chunk c + 1
  // The chunk closes with vmcnt(NVM): the DMAs (issued first) have landed, this chunk's pixel loads stay in flight.
  auto chunk = [&](auto parc, auto firstc, int ch_cur) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr int cbuf = PAR * kBUF, nb = (1 - PAR) * kBUF;
    const int pb_t = (1 - PAR) * PB, pb_a = PAR * PB;
    const int ch_u = ch_cur < last ? ch_cur + 1 : 0;
    f2 av[3], bv[3];
    const int ua = (cbuf + ub) * 4, va = (cbuf + vb) * 4;  // bytes
    auto load_pair = [&](int slot, int x) {
      av[slot] = lds_b64(ua, x * (kC * kK * 4));
      bv[slot] = lds_b64(va, x * (kC * kT * 4));
    };
    auto slice = [&](int s) {
      if (s < 5) dma_u(s, ch_u, nb);
      if (s >= 5 && s < 5 + NR) load_px(std::integral_constant<int, 1 - PAR>{}, s - 5, nL, chL);
      if (s >= 10 && s < 10 + NR) activate_px(std::integral_constant<int, PAR>{}, s - 10, pb_a);
      if ((s & 1) == 0) tstep(s >> 1, pb_t, nb);
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

  for (int n_cur = n_first; n_cur < n_end; n_cur += g.TI) {
    chunk(S0{}, std::true_type{}, 0);
    chunk(S1{}, std::false_type{}, 1);
    for (int ch = 2; ch <= last; ch += 2) {
      chunk(S0{}, std::false_type{}, ch);
      chunk(S1{}, std::false_type{}, ch + 1);
    }

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
    const int n = n_cur + ti, ncl = min(n, a.B - 1);
    float *const XS = smem + kBUF;
    auto pass = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int co = kt * kK + cb * 32 + 8 * q + 4 * elhi + pg;
      const size_t obase = ((size_t)ncl * a.Cout + co) * g.HW + (size_t)(4 * (r0 + tr)) * a.Wo + 4 * tc;
      // addends first: their latency passes under the exchange
      const float addv = (a.bias ? a.bias[co] : 0.f) +
                         (a.chan_add ? a.chan_add[(size_t)ncl * a.chan_add_stride + co] : 0.f);
      {
        float *xw = XS + ((9 * pg) * 2 + cb) * 64 + elane;
#pragma unroll
        for (int x = 0; x < 9; ++x) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            xw[rr * kXS + x * 128] = x == 8 ? acc8[4 * q + rr] : acc[x & 7][4 * q + rr];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      {
        const float *xr = XS + pg * kXS + cb * 64 + elane;  // + xi * 128
        float w[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {  // columns of M through A^T
          float y[4];
          at4(xr[(0 * 6 + j) * 128], xr[(1 * 6 + j) * 128], xr[(2 * 6 + j) * 128], xr[(3 * 6 + j) * 128],
              xr[(4 * 6 + j) * 128], xr[(5 * 6 + j) * 128], y);
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k][j] = y[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float y[4];
          at4(w[k][0], w[k][1], w[k][2], w[k][3], w[k][4], w[k][5], y);
          const v4f res = a.residual ? *reinterpret_cast<const v4f *>(a.residual + obase + (size_t)k * a.Wo) : v4f{0.f, 0.f, 0.f, 0.f};
          if (n < a.B)
            *reinterpret_cast<v4f *>(a.out + obase + (size_t)k * a.Wo) =
                v4f{y[0] + addv + res[0], y[1] + addv + res[1], y[2] + addv + res[2], y[3] + addv + res[3]};
        }
      }
      // nobody may overwrite the slabs (next pass, or the next chunk's staging) while a neighbour still reads them
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
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
  static const kern_t kerns[2][2][2] = {
      {{conv_wino44_kernel<false, 4, false>, conv_wino44_kernel<false, 5, false>},
       {conv_wino44_kernel<false, 4, true>, conv_wino44_kernel<false, 5, true>}},
      {{conv_wino44_kernel<true, 4, false>, conv_wino44_kernel<true, 5, false>},
       {conv_wino44_kernel<true, 4, true>, conv_wino44_kernel<true, 5, true>}}};
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 8; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 4][i / 2 % 2][i % 2]),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  kern_t kern = kerns[d.gscale ? 1 : 0][g.TI == 1 ? 1 : 0][g.NR <= 4 ? 0 : 1];
  const double M = (double)d.B * g.HW;
  // algorithmic work = the direct convolution's: 2 M Cout Cin 9; 36 / 144 of it is executed
  const double flops = 2.0 * M * d.Cout * (double)g.Cin * 9;
  const double bytes = 4.0 * (M * g.Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * g.Cin * 9);
  const char *kname = d.gscale ? "conv3x3_wino44_gn_silu" : "conv3x3_wino44";
  char kshape[160];
  if (g_prof_on && getenv("DDPM_PROF_SHAPES")) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  hipLaunchKernelGGL(kern, dim3(g.grid), dim3(512), lds, s, d, g);
  DDPM_CHECK_LAUNCH();
  return 0;
}

// ---- weights: torch [Cout][Cin][3][3] -> U = G g G^T (6 x 6), packed as the LDS image the kernel's MFMAs read:
//   [cout tile 64][chunk 4][xi 36][lhi 2][cout 64][e 2],  channel of the chunk = 2 e + lhi
// G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], evaluated in double
__global__ void wino44_pack_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin) {
  const int64_t total = (int64_t)Cout * Cin;
  const int nchunks = Cin / kC;
  const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},   {0, 0, 1}};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), o = (int)(i / Cin);
    const float *w = src + ((size_t)o * Cin + ci) * 9;
    double t[6][3];
    for (int r = 0; r < 6; ++r)
      for (int j = 0; j < 3; ++j) t[r][j] = G[r][0] * w[0 * 3 + j] + G[r][1] * w[1 * 3 + j] + G[r][2] * w[2 * 3 + j];
    const int tile = o / kK, k64 = o % kK, ch = ci / kC, cl = ci % kC;
    const int lhi = cl & 1, e = cl >> 1;
    float *d = dst + ((size_t)tile * nchunks + ch) * kUF + (lhi * kK + k64) * 2 + e;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c)
        d[(r * 6 + c) * (kC * kK)] = (float)(t[r][0] * G[c][0] + t[r][1] * G[c][1] + t[r][2] * G[c][2]);
  }
}

size_t wino44_weight_floats(int Cout, int Cin) {
  if (Cout % kK || Cin % 8) return 0;
  return (size_t)kX * Cout * Cin;
}

int launch_pack_wino44_weight(const float *w_raw, float *w_wino44, int Cout, int Cin, hipStream_t s) {
  DDPM_CHECK_ARG(wino44_weight_floats(Cout, Cin) != 0, "wino44 pack: Cout %% 64 or Cin %% 8 != 0");
  const int64_t total = (int64_t)Cout * Cin;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wino44_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_wino44, Cout, Cin);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
