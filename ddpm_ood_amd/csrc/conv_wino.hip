// conv_wino.hip -- 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the fp32 MFMA pipe.
//
// Same fused op as conv_mfma.hip's 9-tap kernel (GroupNorm-affine + SiLU prologue, virtual concat, bias /
// temb / residual epilogue; reference call site /root/reference/src/trainers/reconstruct.py:151-153) with
// 2.25x fewer multiplies: every 2x2 output tile is  Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A  where d_c is
// the 4x4 input patch of channel c.  The sum over channels at each of the 16 transform positions xi is a GEMM,
//     M_xi[cout][tile] = sum_c U_xi[cout][c] * V_xi[c][tile],
// and runs on v_mfma_f32_32x32x2_f32 exactly like the direct kernel: A = U_xi (weights, pre-transformed on the
// device by wino_pack_kernel), B = V_xi (input patches transformed while they are staged into LDS).
// fp32 throughout.  Rounding differs from the direct form (the transforms add values before multiplying):
// measured max error of a 512-channel layer vs fp64 is 1.6e-6 against 0.9e-6 for the direct fp32 conv
// (DESIGN.md section 3.3), far inside the 1e-4 parity bar.
//
// Work item = 64 output channels x 64 tiles (256 output pixels: whole tile rows of one image, or several whole
// images) x all input channels, computed by 8 waves as 2 (cout) x 2 (tile) x 2 (transform rows): a wave owns
// 32 couts x 32 tiles at 8 of the 16 positions = 8 accumulator tiles = 128 AGPRs, two waves per SIMD.  Input
// channels advance in chunks of 8 through double-buffered LDS operand images.
//
// The kernel is persistent: a workgroup owns one (cout tile, image part) and walks a range of images, its items
// forming ONE stream of chunks.  The staging pipeline (pixel loads three chunks ahead, activation two, patch
// transform one, U tile by LDS-DMA one) simply runs across item boundaries, so the next item's first chunks are
// already in LDS when the current item's last MFMA issues; only the output transform + store sits between two
// items.  With one workgroup per launch slot (one per CU: 150 KB of LDS, 8 x 256 registers) the per-item prologue
// and epilogue were a third of the time of a 128-channel layer (DESIGN.md 3.3 has the sampled timeline).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kWT = 64;    // tiles per item
constexpr int kWK = 64;    // output channels per item
constexpr int kWC = 8;     // input channels per chunk
constexpr int kWUF = 16 * kWC * kWK;  // U floats per chunk and cout tile (8192)
constexpr int kWVF = 16 * kWC * kWT;  // V floats per chunk (8192)

// The accumulator tiles must live in the AGPR half of the unified register file: with the builtin hipcc keeps
// them in arch VGPRs, funnels every MFMA through a[0:15] and spills.  The "+a" constraint pins each accumulator
// to its own AGPR tuple.
__device__ __forceinline__ void mfma_agpr(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_setprio 0" : "+a"(c) : "v"(a), "v"(b));
}
// first MFMA of an item into an accumulator: C = 0.  (Zeroing the tuples element by element makes hipcc build
// them in arch VGPRs first -- 128 of them inside the item loop -- and spill.)
__device__ __forceinline__ void mfma_agpr_first(f32x16 &c, float a, float b) {
  asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
// The MFMA operands are fetched from LDS by inline-asm ds_read_b64 and waited for by hand.  hipcc's own waits around
// the LDS-DMA and the staging reads were full lgkmcnt(0) drains several times per chunk; here the wait is "at most N
// LDS operations still in flight", N = the operand reads issued after this pair's (4, 2, 0) -- staging LDS traffic
// issued in between only makes the wait stricter, never wrong (the counter retires in order).
template <int N>
__device__ __forceinline__ void mfma_agpr_wait(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_setprio 0"
               : "+a"(c) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma_agpr_first_wait(f32x16 &c, float a, float b) {
  asm volatile("s_setprio 1\n\ts_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0\n\ts_setprio 0"
               : "=a"(c) : "v"(a), "v"(b), "n"(N));
}
__device__ __forceinline__ f2 lds_read_b64(int byte_addr, int imm) {
  f2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(imm));
  return v;
}

struct WinoGeom {
  int TWc, THr;     // tile columns / rows per image (Wo / 2, Ho / 2)
  int TI, TR;       // images per item, tile rows per item (per image)
  int parts;        // items per image along the rows (1 when an item holds whole images)
  int Cin, nchunks, HW;
  int up;           // 1: DDPM_CONV_UPSAMPLE2 -- tiles are low-res pixels, the pixel tile holds low-res rows
  int HWin;         // pixels of one input channel plane (Hi * Wi)
  int PW, PCH;      // pixel tile in LDS: padded row length (W + 2), floats per channel (TI * (2 TR + 2) rows; TR + 2 if up)
  int NRI;          // staging rounds of 64 pixels per image of an item (TI * NRI <= 6)
  int KT;           // cout tiles
  int NIT;          // items per (cout tile, part) = ceil(B / TI)
  int IPW;          // items per workgroup
  int NS;           // (part, image range) slots = parts * ceil(NIT / IPW)
  int grid;         // KT * NS rounded up so that the cout tiles of one slot share an XCD
  // 3-D (dims = 3, VQ-VAE residual units): an "image" is one (n, d) slice, the chunk stream of an item walks
  // (depth tap, channel chunk) -- 2-D Winograd per depth tap, the taps accumulated in the transform domain
  int D;            // slices per batch item (1: plain 2-D)
  int NIMG;         // images the items walk: B * D
  int nch_c;        // channel chunks per depth tap; nchunks = nkd * nch_c
  int kd0, nkd;     // depth taps kd0 .. kd0 + nkd - 1 (a depth-1 volume only has its centre tap)
  int CS;           // channel stride of the input / output tensors in floats: D * HW
  int nkd_w;        // depth-tap slabs per cout tile in w_wino: 3 for a 3x3x3 weight, else 1
  // small batches: fewer items than CUs, each a long serial chunk stream -> S workgroups share an item, each walks
  // nchunks / S chunks and writes its partial output (after the output transform) into slab `split` of a scratch
  // buffer; split_reduce_kernel adds the slabs in a fixed order together with bias / temb / residual.
  int S;            // channel-stream splits per item (1: none)
  long long pstride;  // floats between two partial-output slabs (0 when S == 1)
};

int device_cus() {
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

static bool wino_geom(const ddpm_conv_desc &d, WinoGeom &g) {
  const int Cin = d.C1 + d.C2;
  const bool is3d = d.dims == 3;
  if (d.ksize != 3 || (!is3d && (d.Di > 1 || d.Do > 1))) return false;
  if (d.out_act != DDPM_ACT_NONE && !(is3d && d.out_act == DDPM_ACT_RELU)) return false;
  if (d.mode != DDPM_CONV_NORMAL && d.mode != DDPM_CONV_UPSAMPLE2) return false;
  // 3-D: stride 1.  A depth tap outside the volume reads zero pixels AND zero GroupNorm scale / shift (out-of-range buffer
  // offsets), so the padding stays zero through the SiLU prologue
  if (is3d && d.mode != DDPM_CONV_NORMAL) return false;
  if (d.mode == DDPM_CONV_UPSAMPLE2 && d.out_act) return false;
  g.up = d.mode == DDPM_CONV_UPSAMPLE2;
  if (g.up && (d.gscale || d.act != DDPM_ACT_NONE || d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi)) return false;
  if (d.act == DDPM_ACT_RELU) return false;
  if (d.gscale && d.act != DDPM_ACT_SILU) return false;  // the affine variant has SiLU built in
  if (Cin % kWC || (d.C2 > 0 && d.C1 % kWC) || d.Cout % kWK) return false;
  if ((d.Ho & 1) || (d.Wo & 1)) return false;
  // pixels are fetched with 32-bit buffer offsets
  const int Dd = is3d ? (d.Di > 1 ? d.Di : 1) : 1;
  if (is3d && (d.Do > 1 ? d.Do : 1) != Dd) return false;
  if ((double)d.B * (d.C1 > d.C2 ? d.C1 : d.C2) * Dd * d.Ho * d.Wo * 4 >= 2147483648.0) return false;
  if ((double)d.B * d.Cout * Dd * d.Ho * d.Wo * 4 >= 2147483648.0 * 2) return false;
  g.TWc = d.Wo / 2;
  g.THr = d.Ho / 2;
  const int per_img = g.TWc * g.THr;
  if (per_img >= kWT) {
    if (kWT % g.TWc) return false;
    g.TI = 1;
    g.TR = kWT / g.TWc;
    if (g.THr % g.TR) return false;
    g.parts = g.THr / g.TR;
  } else {
    if (kWT % per_img) return false;
    g.TI = kWT / per_img;
    g.TR = g.THr;
    g.parts = 1;
  }
  g.Cin = Cin;
  g.D = Dd;
  g.NIMG = d.B * Dd;
  g.nch_c = Cin / kWC;
  g.kd0 = is3d && Dd == 1 ? 1 : 0;
  g.nkd = is3d && Dd > 1 ? 3 : 1;
  g.nkd_w = is3d ? 3 : 1;
  g.nchunks = g.nkd * g.nch_c;
  g.HW = d.Ho * d.Wo;
  g.HWin = d.Hi * d.Wi;
  const int prow = g.up ? g.TR + 2 : 2 * g.TR + 2;  // pixel-tile rows per image of an item
  g.PW = d.Wi + 2;
  g.PCH = (g.TI * prow * g.PW) | 1;  // odd: the two channel planes a patch-stage wave reads interleave over the LDS banks
  const int rows = prow < d.Hi ? prow : d.Hi;  // in-image rows an item reads, at most
  g.NRI = (rows * d.Wi + 63) / 64;
  if (g.TI * g.NRI > 6) return false;
  if ((2 * (kWUF + kWVF) + 2 * kWC * g.PCH + 64) * sizeof(float) > 160 * 1024) return false;
  g.CS = Dd * g.HW;
  g.KT = d.Cout / kWK;
  g.NIT = (g.NIMG + g.TI - 1) / g.TI;
  const long items = (long)g.KT * g.parts * g.NIT;
  const int cus = device_cus();
  g.IPW = (int)((items + cus - 1) / cus);
  g.NS = g.parts * ((g.NIT + g.IPW - 1) / g.IPW);
  // one workgroup per CU (150 KB of LDS): a grid a little over the chip -- 6 cout tiles x 48 slices = 288 workgroups on 256 CUs, the
  // 384-channel input gradient of the training step -- runs a second round for 32 of them.  More items per workgroup until the
  // grid is one round (the item -> workgroup map moves, no sum changes)
  while (items > cus && (long)g.KT * 8 <= cus && (long)g.KT * ((g.NS + 7) / 8) * 8 > cus && g.IPW < g.NIT) {
    ++g.IPW;
    g.NS = g.parts * ((g.NIT + g.IPW - 1) / g.IPW);
  }
  g.S = 1;
  g.pstride = 0;
  if (!is3d && !g.up && items * 2 <= cus) {  // launch would leave more than half of the chip idle
    for (int sp = sw().wino_split; sp >= 2; sp >>= 1)
      if (items * sp <= cus && g.nchunks % sp == 0 && g.nchunks / sp >= 4) { g.S = sp; break; }
  }
  g.grid = g.KT * ((g.NS * g.S + 7) / 8) * 8;
  return true;
}

// slices per (image, cout) of the statistics conv_wino_up_kernel writes to desc.stats_out (0: not emitted)
int conv_wino_stats_parts(const ddpm_conv_desc &d) {
  WinoGeom g;
  if (d.mode != DDPM_CONV_UPSAMPLE2 || d.dims == 3 || !d.w_wino || d.force_direct || !wino_geom(d, g) || g.S != 1) return 0;
  const int per = g.TR * g.TWc;
  if (g.TI == 1) return 2 * g.parts <= 8 ? 2 * g.parts : 0;
  return per == 4 || per == 16 || per == 32 ? 1 : 0;
}

bool conv_wino_supported(const ddpm_conv_desc &d) {
  static const bool enabled = !(getenv("DDPM_CONV_WINOGRAD") && atoi(getenv("DDPM_CONV_WINOGRAD")) == 0);
  WinoGeom g;
  return enabled && d.w_wino != nullptr && !d.force_direct && wino_geom(d, g);
}

// D3: the 3-D form (images = (n, d) slices, chunk stream = (depth tap, channel chunk)); a template parameter so that
// the 2-D instantiations carry none of its address arithmetic (as a run-time branch it cost them 9 %).
template <bool AFFINE, int NR, bool ONEIMG, bool D3 = false>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const ddpm_conv_desc a, const WinoGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = kWUF + kWVF;  // floats per operand buffer: U then V, both [xi 16][k-pair 2][k parity 2][64][2]
  constexpr int NGS = ONEIMG ? 1 : NR;             // GroupNorm scale / shift pairs per chunk: one per image
  constexpr int NVM = NR + (AFFINE ? 2 * NGS : 0);  // vector-memory loads of one pixel stage
  float *const P = smem + 2 * BUF;  // pixel tiles [2][8 channels][PCH] (zero-padded borders) + 64 dump floats
  const int PB = kWC * g.PCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  // 8 waves = 2 (cout block) x 2 (tile block) x 2 (transform rows {0,1} / {2,3}): two waves per SIMD, each wave
  // keeps 8 positions = 128 AGPRs
  const int cb = wave & 1, tb = (wave >> 1) & 1, hf = wave >> 2;
  const bool silu = a.act == DDPM_ACT_SILU;

  // ---- this workgroup's stream: cout tile kt, image part, images [n_first, n_end) in items of TI images.
  // Workgroup ids that differ by a multiple of 8 run on the same XCD (one L2): the cout tiles of one slot, which
  // read the same pixels, are placed there.
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const int kt = wj % g.KT, slot_s = (wj / g.KT) * 8 + xcd;
  if (slot_s >= g.NS * g.S) return;
  const int split = slot_s % g.S, slot = slot_s / g.S;  // the splits of an item sit on neighbouring XCD slots
  const int part = slot % g.parts, it0 = (slot / g.parts) * g.IPW;
  const int nitems = min(g.IPW, g.NIT - it0);
  const int r0 = part * g.TR;  // first tile row of the part
  const int n_first = it0 * g.TI, n_end = n_first + nitems * g.TI;
  const int ch_lo = split * (g.nchunks / g.S), last = ch_lo + g.nchunks / g.S - 1;  // this workgroup's chunk range
  float *const outp = a.out + (size_t)split * g.pstride;  // S > 1: slab `split` of the scratch buffer

  // ---- staging roles ---------------------------------------------------------------------------------
  // Pixels (stage A): wave = channel `sc` of the chunk; its lanes walk the channel's in-image pixels of the item's
  // rows (halo rows included) in NR rounds of 64, round k belonging to image n + k / NRI.  Every pixel is loaded,
  // normalised and activated ONCE and stored into the zero-bordered tile P; f32 MFMAs share the SIMD's FMA
  // hardware with the VALU (tools/ubench/mfma_valu_mix.hip: every VALU op next to an MFMA costs its ~3 cycles,
  // an exp / rcp ~9), so activating the 16 elements of every overlapping 4x4 patch -- each pixel four times --
  // is not free.
  // Patches (stage T): lane = tile `st`, wave = channel: 16 LDS reads of the 4x4 patch out of P (no masks: the
  // border is materialised), B^T d B, 16 LDS writes into the V image.
  // LDS banks: a wave of stage T serves the two channels that share a V pair (e = 0 / 1 of [.. tile][e]) for half of the
  // tiles -- lane = (tile, e) -- so that its 16 V writes are 64 consecutive floats, and the pair's two channel planes
  // sit next to each other in P with an ODD plane size, so that its 16 patch reads (tile stride 2 floats) interleave
  // even / odd banks.  With lane = tile, wave = channel both were two-way bank conflicts (PMC: 8-17 % of the CU's
  // cycles with an LDS conflict stall).
  const int sc = wave;
  auto plane_of = [](int c) { return (((c >> 2) * 2 + (c & 1)) * 2) + ((c >> 1) & 1); };  // channel -> plane of P
  const int tq = wave & 3, te = lane & 1;                 // stage T: V pair (kp, lhi') = (tq >> 1, tq & 1), e
  const int st = (wave >> 2) * 32 + (lane >> 1);           //          tile
  const int tch = 4 * (tq >> 1) + 2 * te + (tq & 1);       //          channel of the chunk this lane transforms
  const int row_lo = max(0, 2 * r0 - 1), row_hi = min(a.Ho, 2 * (r0 + g.TR) + 1);  // real rows an item reads
  const int npx = (row_hi - row_lo) * a.Wo;
  int pix[NR], pw[NR], tik[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int ti = k / g.NRI, e = lane + 64 * (k - ti * g.NRI);
    const bool valid = ti < g.TI && e < npx;
    const int row = row_lo + e / a.Wo, col = e % a.Wo;
    tik[k] = ti;
    pix[k] = valid ? (row * a.Wo + col) * 4 : (int)0x80000000;  // out of range: the buffer load returns 0
    pw[k] = valid ? plane_of(sc) * g.PCH + (ti * (2 * g.TR + 2) + row - (2 * r0 - 1)) * g.PW + col + 1 : 2 * PB + lane;
  }
  int tbase;  // patch origin of tile st inside the plane of channel tch
  {
    const int per = g.TR * g.TWc;
    const int ti = st / per, rem = st - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tbase = plane_of(tch) * g.PCH + (ti * (2 * g.TR + 2) + 2 * tr) * g.PW + 2 * tc;
  }
  const int bytes1 = a.B * a.C1 * (D3 ? g.CS : g.HW) * 4, bytes2 = a.B * a.C2 * (D3 ? g.CS : g.HW) * 4;
  const __amdgpu_buffer_rsrc_t rs_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sh =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  int vzero;  // a zero the compiler cannot see through: keeps the uniform scale / shift loads on the vector
              // memory path (scalar loads share lgkmcnt with LDS and would force full LDS drains)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

  // ---- MFMA operands.  This wave's positions are xi = 8 hf + x, x = 0..7; its 32 MFMAs of a chunk are 16 pairs
  // p = (k-pair kp = p >> 3, position x = p & 7), the two MFMAs of a pair covering channels 4 kp + lhi and
  // 4 kp + 2 + lhi of the chunk.  Both operand images are [xi][kp][lhi][cout or tile 64][2], so a pair's A and B
  // values are one ds_read_b64 each (256 B/clk against ds_read_b32's 128).
  const int ub = (lhi * kWK + cb * 32 + l31) * 2;         // + ((x + 8 hf) * 2 + kp) * 2 * 64 * 2
  const int vb = kWUF + (lhi * kWT + tb * 32 + l31) * 2;  // likewise
  // The U tile of a chunk (32 KB, already in LDS order in global memory) is copied by LDS-DMA: 4 x 1 KB per wave,
  // no registers, no ds_write pass, and -- unlike loads into registers -- nothing in the loop has to wait for it
  // before the chunk's closing barrier, so the (in-order) vmcnt waits never drag the slow pixel loads along.
  // U is stored [cout tile][depth tap][channel chunk][32 KB]: the chunk stream of an item reads it front to back

  f32x16 acc[8];

  // ---- staging registers, and the slices of staging work the chunk loop places between its MFMAs -----------
  float praw[NR], gs[NGS], gh[NGS], dreg[16], tt[16];
  // quarter i of this wave's share of the U tile of chunk ch -> operand buffer at float offset nb.  Buffer form of the
  // LDS-DMA: resource + scalar offset + one loop-invariant lane offset (with global_load_lds hipcc keeps 64-bit per-lane
  // addresses, spills them, and every reload -- a scratch load -- waits vmcnt(0))
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.w_wino), 0, (int)((size_t)16 * a.Cout * g.Cin * g.nkd_w * 4), 0x00020000);
  const int ulane = lane * 16;
  const int ubase_f = (D3 ? (kt * g.nkd_w + g.kd0) * g.nch_c : kt * g.nchunks) * kWUF + wave * 4 * 256;  // floats
  auto dma_u = [&](int i, int ch, int nb) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (__attribute__((address_space(3))) void *)(smem + nb + (wave * 4 + i) * 256),
                                             16, ulane, (ubase_f + ch * kWUF + i * 256) * 4, 0, 0);
  };
  // stage L: round k of chunk ch of the item at image n -> registers.  What depends on the chunk only (source tensor,
  // channel, depth tap) is set up once per chunk, the image offset per round
  __amdgpu_buffer_rsrc_t l_rs;
  int l_cx, l_cgl, l_cg, l_kd;
  auto load_setup = [&](int ch) {
    l_cg = ch * kWC + sc;
    l_kd = 0;
    if (D3) {  // stream chunk -> (depth tap, channel chunk)
      const int kdi = ch / g.nch_c;
      l_cg = (ch - kdi * g.nch_c) * kWC + sc;
      l_kd = g.kd0 + kdi - 1;
    }
    const bool first = l_cg < a.C1;
    l_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2,
                                             0x00020000);
    l_cx = first ? a.C1 : a.C2;
    l_cgl = first ? l_cg : l_cg - a.C1;
  };
  auto load_px = [&](int k, int n) {
    int ni = min(n + tik[k], g.NIMG - 1);
    int soff, voff = pix[k];
    bool dok3 = true;
    if (D3) {  // image -> (batch item, slice); a depth tap outside the volume reads zeros: the range check of a raw buffer
               // load is on the VGPR offset, and 0x80000000 is past every resource (as for the halo pixels in pix[])
      const int nb = ni / g.D, dsl = ni - nb * g.D + l_kd;
      const bool dok = dsl >= 0 && dsl < g.D;
      soff = ((nb * l_cx + l_cgl) * g.D + (dok ? dsl : 0)) * g.HW * 4;
      if (!dok) voff = (int)0x80000000;
      dok3 = dok;
      ni = nb;
    } else {
      soff = (ni * l_cx + l_cgl) * g.HW * 4;
    }
    praw[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(l_rs, voff, soff, 0));
    if (AFFINE && (!ONEIMG || k == 0)) {
      const int goff = (ni * g.Cin + l_cg) * 4;
      const int gv = D3 && !dok3 ? (int)0x80000000 : vzero;  // tap outside the volume: scale = shift = 0 -> SiLU(0) = 0
      gs[ONEIMG ? 0 : k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, gv, goff, 0));
      gh[ONEIMG ? 0 : k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, gv, goff, 0));
    }
  };
  // stage A: round k -> pixel tile `pb` (float offset of the P buffer).  With the GroupNorm affine the SiLU argument
  // is formed straight from the pixel: v = x a + b and t = -log2(e) v = x a' + b' are two independent fmas, then
  // v * rcp(1 + exp2(t)): 4 VALU + 2 transcendental per pixel
  auto activate_px = [&](int k, int pb) {
    const float x = praw[k];
    if (AFFINE) {
      const float sa = gs[ONEIMG ? 0 : k], sb = gh[ONEIMG ? 0 : k];
      const float v = __builtin_fmaf(x, sa, sb);
      const float t = __builtin_fmaf(x, -1.44269504088896341f * sa, -1.44269504088896341f * sb);
      P[pb + pw[k]] = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
    } else {
      const float sv = silu_fast(x);
      P[pb + pw[k]] = silu ? sv : x;
    }
  };
  // stage T: patch row i out of pixel tile `pb`
  auto read_patch = [&](int i, int pb) {
    const float *p = P + pb + tbase + i * g.PW;
#pragma unroll
    for (int j = 0; j < 4; ++j) dreg[4 * i + j] = p[j];
  };
  // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows first ...
  auto row_transform = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tt[0 * 4 + j] = dreg[0 * 4 + j] - dreg[2 * 4 + j];
      tt[1 * 4 + j] = dreg[1 * 4 + j] + dreg[2 * 4 + j];
      tt[2 * 4 + j] = dreg[2 * 4 + j] - dreg[1 * 4 + j];
      tt[3 * 4 + j] = dreg[1 * 4 + j] - dreg[3 * 4 + j];
    }
  };
  // ... then the columns of row i, written to the four positions (i, 0..3) of the V image
  auto col_commit = [&](int i, int nb) {
    // channel tch = 4 kp + 2 e + lhi  ->  V [xi][kp][lhi][tile][e]: the wave's lanes write 64 consecutive floats
    float *vl = smem + nb + kWUF + (tq * kWT + st) * 2 + te;
    vl[(i * 4 + 0) * kWC * kWT] = tt[i * 4 + 0] - tt[i * 4 + 2];
    vl[(i * 4 + 1) * kWC * kWT] = tt[i * 4 + 1] + tt[i * 4 + 2];
    vl[(i * 4 + 2) * kWC * kWT] = tt[i * 4 + 2] - tt[i * 4 + 1];
    vl[(i * 4 + 3) * kWC * kWT] = tt[i * 4 + 1] - tt[i * 4 + 3];
  };
  // stream position -> next stream position; past the end it sticks to the last chunk (its staging lands in
  // buffers nobody reads, which keeps the loop body free of branches)
  auto advance = [&](int &n, int &ch) {
    if (ch < last) {
      ++ch;
    } else if (n + g.TI < n_end) {
      n += g.TI;
      ch = ch_lo;
    }
  };

  // ---- prologue: zero borders; pixel tiles of stream chunks 0 and 1; U and V of chunk 0; registers for chunk 2
  for (int i = tid; i < 2 * PB + 64; i += 512) P[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_u(i, ch_lo, 0);
  int nL = n_first, chL = ch_lo;  // stream position of the pixel-load stage
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    load_setup(chL);
#pragma unroll
    for (int k = 0; k < NR; ++k) load_px(k, nL);
#pragma unroll
    for (int k = 0; k < NR; ++k) activate_px(k, c * PB);
    advance(nL, chL);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) read_patch(i, 0);
  row_transform();
#pragma unroll
  for (int i = 0; i < 4; ++i) col_commit(i, 0);
  load_setup(chL);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(k, nL);
  advance(nL, chL);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NVM) : "memory");
  __builtin_amdgcn_sched_barrier(0);

  // One chunk = 32 MFMA steps per wave, with three stages of staging in flight (c = stream index):
  //   T(c+1)  patches of chunk c + 1 out of P[(c+1) & 1] -> V of the other operand buffer
  //   A(c+2)  pixels of chunk c + 2 (registers) -> P[c & 1]          L(c+3)  pixel loads of chunk c + 3
  // plus the LDS-DMA of the U tile of chunk c + 1.  The work is cut into slices, one per MFMA step, and
  // sched_barriers pin every slice to its MFMA: left to itself the compiler emits the staging as one block between
  // two MFMAs and issues each operand read right before its use.
  //   step 0..     activation of pixel round s -> P (loaded a whole chunk ago; the vmcnt wait sees nothing younger)
  //   step 6..     loads of pixel round s - 6;   step 12..15  U quarter s - 12 by DMA (17 steps to land)
  //   step 16..19  patch row s - 16 -> registers;   step 22  row transform;   step 23..26  column transform + write
  // The chunk closes with a counted vmcnt (the DMAs are older than the pixel loads, which stay in flight) and a raw
  // s_barrier: __syncthreads() would wait for vmcnt(0), i.e. for HBM, every chunk.
  // The two waves of a SIMD (w, w + 4) share its matrix pipe and VALU issue, arbitrated by priority, then age; the
  // second-dispatched half loses every tie and trails its partner into each barrier.  One static priority bump for
  // that half evens them out (MI355X_MICROARCH.md, "Two waves per SIMD").
  int c = 0;  // stream index
  auto chunk = [&](auto first_c, int ch_cur) {
    constexpr bool FIRST = decltype(first_c)::value;  // first chunk of an item: its MFMAs start the accumulators
    const int cbuf = (c & 1) * BUF;
    const int nb = BUF - cbuf;
    const int pb_t = ((c + 1) & 1) * PB, pb_a = (c & 1) * PB;
    const int ch_u = ch_cur < last ? ch_cur + 1 : ch_lo;  // U depends on the chunk only, not on the image
    load_setup(chL);
    f2 av[3], bv[3];  // operand ring: three pairs
    const int ua = (cbuf + ub + 8 * hf * 2 * 2 * 64 * 2) * 4, va = (cbuf + vb + 8 * hf * 2 * 2 * 64 * 2) * 4;  // bytes
    auto load_pair = [&](int slot, int p) {
      const int imm = (((p & 7) * 2 + (p >> 3)) * 2 * 64 * 2) * 4;
      av[slot] = lds_read_b64(ua, imm);
      bv[slot] = lds_read_b64(va, imm);
    };
    auto slice = [&](int s) {
      if (s < NR) {
        activate_px(s, pb_a);
      } else if (s >= 6 && s < 6 + NR) {
        load_px(s - 6, nL);
      } else if (s >= 12 && s < 16) {
        dma_u(s - 12, ch_u, nb);
      } else if (s >= 16 && s < 20) {
        read_patch(s - 16, pb_t);
      } else if (s == 22) {
        row_transform();
      } else if (s >= 23 && s < 27) {
        col_commit(s - 23, nb);
      }
    };
    load_pair(0, 0);
    load_pair(1, 1);
    load_pair(2, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      // operand reads issued after this pair's: two more pairs (4 reads), one at p = 14, none at p = 15
      if (FIRST && p < 8)
        mfma_agpr_first_wait<4>(acc[p & 7], av[p % 3][0], bv[p % 3][0]);
      else if (p < 14)
        mfma_agpr_wait<4>(acc[p & 7], av[p % 3][0], bv[p % 3][0]);
      else if (p == 14)
        mfma_agpr_wait<2>(acc[p & 7], av[p % 3][0], bv[p % 3][0]);
      else
        mfma_agpr_wait<0>(acc[p & 7], av[p % 3][0], bv[p % 3][0]);
      slice(2 * p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_agpr(acc[p & 7], av[p % 3][1], bv[p % 3][1]);
      if (p + 3 < 16) load_pair(p % 3, p + 3);
      slice(2 * p + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance(nL, chL);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // DMAs landed; pixel loads are older
    // (measured: issuing the DMAs before the pixel loads and waiting vmcnt(NVM) here -- pixel loads left in flight
    //  across the barrier -- is 2 % slower: the wait just moves to the next chunk's first activation slice)
    __builtin_amdgcn_sched_barrier(0);
    ++c;
  };

  for (int n_cur = n_first; n_cur < n_end; n_cur += g.TI) {
    chunk(std::true_type{}, ch_lo);
    for (int ch = ch_lo + 1; ch <= last; ++ch) chunk(std::false_type{}, ch);
    const int cbuf = ((c - 1) & 1) * BUF;  // the operand buffer the item's last chunk consumed

    // ---- end of an item: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  The transform is linear in M, so each wave
    // applies it to its own two rows of M (hf = 0: rows 0, 1; hf = 1: rows 2, 3) and the two waves of a pair swap
    // halves through LDS: wave hf finishes accumulator registers 8 hf .. 8 hf + 7 (8 of its 16 couts per lane) and
    // sends the partial sums of the other eight.  The exchange uses the operand buffer the item's last chunk just
    // consumed; the next item's first chunk is already staged in the other one.
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' 16 passes
    // partial outputs of accumulator register r of this wave: its two rows of M through A^T (.) A
    auto partial = [&](int r, float *y) {
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ma = acc[j][r], mb = acc[4 + j][r];  // rows 2 hf and 2 hf + 1
        s0[j] = hf == 0 ? ma + mb : ma;
        s1[j] = hf == 0 ? mb : -ma - mb;
      }
      y[0] = s0[0] + s0[1] + s0[2];
      y[1] = s0[1] - s0[2] - s0[3];
      y[2] = s1[0] + s1[1] + s1[2];
      y[3] = s1[1] - s1[2] - s1[3];
    };
    int elane = lane;  // re-derived here so that the epilogue's addressing is not hoisted out of (and kept live
    asm volatile("" : "+v"(elane));  // across) the chunk loop
    const int tq = tb * 32 + (elane & 31);  // this lane's tile
    const int per = g.TR * g.TWc;
    const int ti = tq / per, rem = tq - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    const int n = n_cur + ti;
    const int co_base = kt * kWK + cb * 32 + 4 * (elane >> 5) + 16 * hf;
    const int ncl = min(n, g.NIMG - 1), nbat = D3 ? ncl / g.D : ncl, dsl_o = D3 ? ncl - nbat * g.D : 0;  // (item, slice)
    const int cstr = D3 ? g.CS : g.HW;  // channel stride
    const size_t obase = D3 ? (((size_t)nbat * a.Cout + co_base) * g.D + dsl_o) * g.HW + (size_t)(2 * (r0 + tr)) * a.Wo + 2 * tc
                            : ((size_t)nbat * a.Cout + co_base) * g.HW + (size_t)(2 * (r0 + tr)) * a.Wo + 2 * tc;
    // the addends (residual, bias + temb) are requested first: their latency passes under the transform + exchange
    f2 ra[8], rb[8];
    float addv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int cof = (rr & 3) + 8 * (rr >> 2);
      const size_t o = obase + (size_t)cof * cstr;
      ra[rr] = rb[rr] = f2{0.f, 0.f};
      if (a.residual) {
        ra[rr] = *reinterpret_cast<const f2 *>(a.residual + o);
        rb[rr] = *reinterpret_cast<const f2 *>(a.residual + o + a.Wo);
      }
      addv[rr] = (a.bias ? a.bias[co_base + cof] : 0.f) +
                 (a.chan_add ? a.chan_add[(size_t)nbat * a.chan_add_stride + co_base + cof] : 0.f);
    }
    {
      float *xw = smem + cbuf + (((wave & 3) * 2 + hf) * 32) * 64 + elane;  // [pair][from hf][rr * 4 + x][lane]
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        float ya[4], yb[4];
        partial(rr, ya);
        partial(rr + 8, yb);
#pragma unroll
        for (int x = 0; x < 4; ++x) xw[(rr * 4 + x) * 64] = hf == 0 ? yb[x] : ya[x];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
      const float *xr = smem + cbuf + (((wave & 3) * 2 + (1 - hf)) * 32) * 64 + elane;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        float ya[4], yb[4], yy[4];
        partial(rr, ya);  // recomputed rather than kept across the barrier: registers are the scarce resource
        partial(rr + 8, yb);
#pragma unroll
        for (int x = 0; x < 4; ++x) yy[x] = (hf == 0 ? ya[x] : yb[x]) + xr[(rr * 4 + x) * 64] + addv[rr];
        yy[0] += ra[rr][0]; yy[1] += ra[rr][1]; yy[2] += rb[rr][0]; yy[3] += rb[rr][1];
        if (D3 && a.out_act == DDPM_ACT_RELU) {
#pragma unroll
          for (int x = 0; x < 4; ++x) yy[x] = fmaxf(yy[x], 0.f);
        }
        if (n < g.NIMG) {
          const size_t o = obase + (size_t)((rr & 3) + 8 * (rr >> 2)) * cstr;
          *reinterpret_cast<f2 *>(outp + o) = f2{yy[0], yy[1]};
          *reinterpret_cast<f2 *>(outp + o + a.Wo) = f2{yy[2], yy[3]};
        }
      }
    }
    // nobody may stage the next chunk into this buffer while a neighbour still reads its exchange data
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}


// ---- Upsample: nearest x2 followed by the 3x3 conv, in the Winograd domain ---------------------------------------
// On a nearest-x2 image the 4x4 patch of the output tile (2 ty .. 2 ty + 1, 2 tx .. 2 tx + 1) has rows
// [L(ty-1), L(ty), L(ty), L(ty+1)] of the low-res image (zero outside), and the same for columns, so
//   B^T d = [L- - L0, 2 L0, 0, L0 - L+]:  row 2 and column 2 of V = B^T d B vanish.
// Only the 9 positions (i, j) in {0, 1, 3}^2 carry work: 2.25 multiplies per output against 4 for the folded 2x2-tap
// form (conv_mfma.hip) and 9 for the plain conv.  Tiles are low-res pixels (one 2x2 output tile each), so the item /
// stream geometry is the one of the normal kernel with TWc = Wi; the pixel tile holds low-res rows (TR + 2 per item,
// raw: Upsample has no GroupNorm / activation in front), a patch is 3x3 with stride 1.  The positions are dealt to
// the two waves of a pair as 5 + 4 (they are independent GEMMs and the output transform is linear, so any split
// works): wave half 0 takes xi = 0, 1, 3, 4, 5, half 1 takes 7, 12, 13, 15; 20 / 16 MFMAs per chunk.
constexpr int kUpPos[2][5] = {{0, 1, 3, 4, 5}, {7, 12, 13, 15, 15}};
constexpr int kUpNX[2] = {5, 4};
constexpr int kUpIdx[3] = {0, 1, 3};                       // transform index of the three surviving rows / columns
constexpr int kAT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};  // A^T

template <int NR>
__global__ __launch_bounds__(512, 2) void conv_wino_up_kernel(const ddpm_conv_desc a, const WinoGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = kWUF + kWVF;
  float *const P = smem + 2 * BUF;  // low-res pixel tiles [2][8 channels][PCH] (zero borders) + 64 dump floats
  const int PB = kWC * g.PCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb = wave & 1, tb = (wave >> 1) & 1, hf = wave >> 2;

  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const int kt = wj % g.KT, slot = (wj / g.KT) * 8 + xcd;
  if (slot >= g.NS) return;
  const int part = slot % g.parts, it0 = (slot / g.parts) * g.IPW;
  const int nitems = min(g.IPW, g.NIT - it0);
  const int r0 = part * g.TR;  // first low-res row (= tile row) of the part
  const int n_first = it0 * g.TI, n_end = n_first + nitems * g.TI;
  const int last = g.nchunks - 1;

  // ---- staging roles: wave = channel of the chunk; pixels in NR rounds of 64 lanes, patches one per lane ----
  const int sc = wave, st = lane;
  const int PR = g.TR + 2;
  const int row_lo = max(0, r0 - 1), row_hi = min(a.Hi, r0 + g.TR + 1);  // low-res rows an item reads
  const int npx = (row_hi - row_lo) * a.Wi;
  int pix[NR], pw[NR], tik[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int ti = k / g.NRI, e = lane + 64 * (k - ti * g.NRI);
    const bool valid = ti < g.TI && e < npx;
    const int row = row_lo + e / a.Wi, col = e % a.Wi;
    tik[k] = ti;
    pix[k] = valid ? (row * a.Wi + col) * 4 : (int)0x80000000;
    pw[k] = valid ? sc * g.PCH + (ti * PR + row - (r0 - 1)) * g.PW + col + 1 : 2 * PB + lane;
  }
  int tbase;  // 3x3 patch origin of tile st inside a channel tile of P
  {
    const int per = g.TR * g.TWc;
    const int ti = st / per, rem = st - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tbase = sc * g.PCH + (ti * PR + tr) * g.PW + tc;
  }
  const int bytes1 = a.B * a.C1 * g.HWin * 4, bytes2 = a.B * a.C2 * g.HWin * 4;

  const int ub = (lhi * kWK + cb * 32 + l31) * 2;
  const int vb = kWUF + (lhi * kWT + tb * 32 + l31) * 2;
  const float *usrc = a.w_wino + (size_t)kt * g.nchunks * kWUF + wave * 4 * 256;

  f32x16 acc[5];
  float praw[NR], dreg[9], vv[9];
  auto dma_u = [&](int i, int ch, int nb) {
    const float *ubase = usrc + (size_t)ch * kWUF + i * 256;  // uniform
    __builtin_amdgcn_global_load_lds(ubase + lane * 4, smem + nb + (wave * 4 + i) * 256, 16, 0, 0);
  };
  auto load_px = [&](int k, int n, int ch) {
    const int cg = ch * kWC + sc, ni = min(n + tik[k], a.B - 1);
    const bool first = cg < a.C1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2, 0x00020000);
    const int soff = first ? (ni * a.C1 + cg) * g.HWin * 4 : (ni * a.C2 + cg - a.C1) * g.HWin * 4;
    praw[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pix[k], soff, 0));
  };
  auto store_px = [&](int k, int pb) { P[pb + pw[k]] = praw[k]; };
  auto read_patch = [&](int i, int pb) {
    const float *p = P + pb + tbase + i * g.PW;
#pragma unroll
    for (int j = 0; j < 3; ++j) dreg[3 * i + j] = p[j];
  };
  // V at the 9 surviving positions: rows (L- - L0, 2 L0, L0 - L+), then the same along the columns
  auto transform = [&]() {
    float t[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      t[0 * 3 + j] = dreg[0 * 3 + j] - dreg[1 * 3 + j];
      t[1 * 3 + j] = dreg[1 * 3 + j] + dreg[1 * 3 + j];
      t[2 * 3 + j] = dreg[1 * 3 + j] - dreg[2 * 3 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      vv[i * 3 + 0] = t[i * 3 + 0] - t[i * 3 + 1];
      vv[i * 3 + 1] = t[i * 3 + 1] + t[i * 3 + 1];
      vv[i * 3 + 2] = t[i * 3 + 1] - t[i * 3 + 2];
    }
  };
  auto commit = [&](int i, int nb) {
    float *vl = smem + nb + kWUF + (((sc >> 2) * 2 + (sc & 1)) * kWT + st) * 2 + ((sc >> 1) & 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) vl[(kUpIdx[i] * 4 + kUpIdx[j]) * kWC * kWT] = vv[i * 3 + j];
  };
  auto advance = [&](int &n, int &ch) {
    if (ch < last) {
      ++ch;
    } else if (n + g.TI < n_end) {
      n += g.TI;
      ch = 0;
    }
  };

  // ---- prologue ----------------------------------------------------------------------------------------------
  for (int i = tid; i < 2 * PB + 64; i += 512) P[i] = 0.f;
  // positions 2, 6, 8..11, 14 of V are never written and never read; nothing to clear there
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_u(i, 0, 0);
  int nL = n_first, chL = 0;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int k = 0; k < NR; ++k) load_px(k, nL, chL);
#pragma unroll
    for (int k = 0; k < NR; ++k) store_px(k, c * PB);
    advance(nL, chL);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i) read_patch(i, 0);
  transform();
#pragma unroll
  for (int i = 0; i < 3; ++i) commit(i, 0);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(k, nL, chL);
  advance(nL, chL);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NR) : "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- chunk: 2 NX pairs of MFMAs (NX positions x 2 k-pairs), staging slices pinned to the steps -----------------
  //   step 0..3  U quarter by DMA;  step 0..2  patch row -> registers;  step 4  transform;  step 5..7  V writes
  //   step 9..   pixel round s - 9 -> P;  step 10..  loads of pixel round s - 10 (three chunks ahead)
  int c = 0;
  auto chunk = [&](auto first_c, auto hf_c, int ch_cur) {
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int HF = decltype(hf_c)::value;
    constexpr int NX = kUpNX[HF], NP = 2 * NX;
    const int cbuf = (c & 1) * BUF;
    const int nb = BUF - cbuf;
    const int pb_t = ((c + 1) & 1) * PB, pb_a = (c & 1) * PB;
    const int ch_u = ch_cur < last ? ch_cur + 1 : 0;
    f2 av[3], bv[3];
    const int ua = (cbuf + ub) * 4, va = (cbuf + vb) * 4;  // bytes
    auto load_pair = [&](int slot, int p) {
      const int imm = ((kUpPos[HF][p % NX] * 2 + p / NX) * 2 * 64 * 2) * 4;
      av[slot] = lds_read_b64(ua, imm);
      bv[slot] = lds_read_b64(va, imm);
    };
    auto slice = [&](int s) {
      if (s < 4) dma_u(s, ch_u, nb);
      if (s < 3) read_patch(s, pb_t);
      if (s == 4) transform();
      if (s >= 5 && s < 8) commit(s - 5, nb);
      if (s >= 9 && s < 9 + NR) store_px(s - 9, pb_a);
      if (s >= 10 && s < 10 + NR) load_px(s - 10, nL, chL);
    };
    load_pair(0, 0);
    load_pair(1, 1);
    load_pair(2, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      f32x16 &ac = acc[p % NX];
      if (FIRST && p < NX)
        mfma_agpr_first_wait<4>(ac, av[p % 3][0], bv[p % 3][0]);
      else if (p < NP - 2)
        mfma_agpr_wait<4>(ac, av[p % 3][0], bv[p % 3][0]);
      else if (p == NP - 2)
        mfma_agpr_wait<2>(ac, av[p % 3][0], bv[p % 3][0]);
      else
        mfma_agpr_wait<0>(ac, av[p % 3][0], bv[p % 3][0]);
      slice(2 * p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_agpr(ac, av[p % 3][1], bv[p % 3][1]);
      if (p + 3 < NP) load_pair(p % 3, p + 3);
      slice(2 * p + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance(nL, chL);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ++c;
  };

  for (int n_cur = n_first; n_cur < n_end; n_cur += g.TI) {
    if (hf == 0) {
      chunk(std::true_type{}, std::integral_constant<int, 0>{}, 0);
      for (int ch = 1; ch <= last; ++ch) chunk(std::false_type{}, std::integral_constant<int, 0>{}, ch);
    } else {
      chunk(std::true_type{}, std::integral_constant<int, 1>{}, 0);
      for (int ch = 1; ch <= last; ++ch) chunk(std::false_type{}, std::integral_constant<int, 1>{}, ch);
    }
    const int cbuf = ((c - 1) & 1) * BUF;

    // ---- end of an item: Y = A^T M A restricted to this wave's positions; halves swapped as in the normal kernel --
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int tq = tb * 32 + (elane & 31);
    const int per = g.TR * g.TWc;
    const int ti = tq / per, rem = tq - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    const int n = n_cur + ti;
    const int co_base = kt * kWK + cb * 32 + 4 * (elane >> 5) + 16 * hf;
    const size_t obase = ((size_t)min(n, a.B - 1) * a.Cout + co_base) * g.HW + (size_t)(2 * (r0 + tr)) * a.Wo + 2 * tc;
    f2 ra[8], rb[8];
    float addv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int cof = (rr & 3) + 8 * (rr >> 2);
      const size_t o = obase + (size_t)cof * g.HW;
      ra[rr] = rb[rr] = f2{0.f, 0.f};
      if (a.residual) {
        ra[rr] = *reinterpret_cast<const f2 *>(a.residual + o);
        rb[rr] = *reinterpret_cast<const f2 *>(a.residual + o + a.Wo);
      }
      addv[rr] = (a.bias ? a.bias[co_base + cof] : 0.f) +
                 (a.chan_add ? a.chan_add[(size_t)min(n, a.B - 1) * a.chan_add_stride + co_base + cof] : 0.f);
    }
    // y[2 a + b] = sum over this wave's positions (i, j) of A^T[a][i] A^T[b][j] M_ij, coefficients in {-1, 0, 1}
    auto partial = [&](auto hf_c, int r, float *y) {
      constexpr int HF = decltype(hf_c)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = 0.f;
#pragma unroll
      for (int x = 0; x < kUpNX[HF]; ++x) {
        const float m = acc[x][r];
        const int i = kUpPos[HF][x] >> 2, j = kUpPos[HF][x] & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cf = kAT[q >> 1][i] * kAT[q & 1][j];
          if (cf == 1) y[q] += m;
          if (cf == -1) y[q] -= m;
        }
      }
    };
    auto send = [&](auto hf_c) {
      constexpr int HF = decltype(hf_c)::value;
      float *xw = smem + cbuf + (((wave & 3) * 2 + HF) * 32) * 64 + elane;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        float y[4];
        partial(hf_c, rr + 8 * (1 - HF), y);  // the register half the partner finishes
#pragma unroll
        for (int x = 0; x < 4; ++x) xw[(rr * 4 + x) * 64] = y[x];
      }
    };
    if (hf == 0) send(std::integral_constant<int, 0>{});
    else send(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // GroupNorm statistics of the produced tensor (desc.stats_out, as conv_wino44h.hip): a wave finishes 32 tiles = one
    // slice of the image (TI == 1: 2 * parts slices of 32 low-res pixels) or whole images (`per` consecutive lanes each)
    const bool emit_stats = a.stats_out != nullptr;
    const int st_lanes = g.TI == 1 ? 32 : per;
    const int st_parts = g.TI == 1 ? 2 * g.parts : 1, st_slice = g.TI == 1 ? 2 * part + tb : 0;
    auto finish = [&](auto hf_c) {
      constexpr int HF = decltype(hf_c)::value;
      const float *xr = smem + cbuf + (((wave & 3) * 2 + (1 - HF)) * 32) * 64 + elane;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        float y[4], yy[4];
        partial(hf_c, rr + 8 * HF, y);
#pragma unroll
        for (int x = 0; x < 4; ++x) yy[x] = y[x] + xr[(rr * 4 + x) * 64] + addv[rr];
        yy[0] += ra[rr][0]; yy[1] += ra[rr][1]; yy[2] += rb[rr][0]; yy[3] += rb[rr][1];
        if (n < a.B) {
          const size_t o = obase + (size_t)((rr & 3) + 8 * (rr >> 2)) * g.HW;
          *reinterpret_cast<f2 *>(a.out + o) = f2{yy[0], yy[1]};
          *reinterpret_cast<f2 *>(a.out + o + a.Wo) = f2{yy[2], yy[3]};
        }
        if (emit_stats) {
          float mean = 0.25f * ((yy[0] + yy[1]) + (yy[2] + yy[3]));
          const float d0 = yy[0] - mean, d1 = yy[1] - mean, d2 = yy[2] - mean, d3 = yy[3] - mean;
          float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          group_moments_last_lane(mean, m2, 4.f, st_lanes);
          if (((elane & 31) & (st_lanes - 1)) == st_lanes - 1 && n < a.B) {
            const size_t co = (size_t)co_base + (rr & 3) + 8 * (rr >> 2);
            *reinterpret_cast<float2 *>(a.stats_out + (((size_t)n * a.Cout + co) * st_parts + st_slice) * 2) =
                make_float2(mean, m2);
          }
        }
      }
    };
    if (hf == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- four-wave variant: ONE wave per SIMD, all 16 transform positions of its 32 couts x 32 tiles in 256 AGPRs -----
// tools/ubench/mfma_peak.hip: a register-only f32 MFMA loop reaches 154 TFLOP/s with one wave per SIMD but only 137
// with two (arbitration between the two waves' MFMA streams), and the eight-wave kernel above additionally pays an
// LDS exchange + two barriers per item because the transform rows of a tile are split over two waves.  Here a wave
// owns whole tiles: the output transform A^T M A is register-only, items need no barrier of their own, and the 256
// arch VGPRs left beside the accumulators hold every staging value without spills.  Staging work per lane doubles
// (two channels of the chunk per wave) but so does the number of MFMA steps it is sliced into (64 per chunk).
// MEASURED (tools/wino_ab.py, B = 256, five layer shapes of the `small` UNet): 2 986 us against 2 486 us for the
// eight-wave kernel, i.e. 0.45-0.55 of the MFMA peak instead of 0.51-0.68; an ablation without any staging reached
// 0.62-0.78.  With a single wave per SIMD nothing covers the wave's own LDS operand latency and its VALU slices, and
// that costs more than the arbitration it avoids.  Kept as an opt-in A/B (DDPM_WINO_WAVES=4, parity-tested), NOT the
// default; the single barrier placed after pair 28 with the next chunk's operands prefetched across it (no restart
// bubble) made no measurable difference either.
template <bool AFFINE, int NR, bool ONEIMG>
__global__ __launch_bounds__(256, 1) void conv_wino4_kernel(const ddpm_conv_desc a, const WinoGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = kWUF + kWVF;
  constexpr int NGS = ONEIMG ? 1 : NR;
  float *const P = smem + 2 * BUF;          // pixel tiles [2][8 channels][PCH] + 64 dump floats each
  const int PB = kWC * g.PCH, PBS = PB + 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb = wave & 1, tb = wave >> 1;   // 4 waves = 2 (cout block) x 2 (tile block)
  const bool silu = a.act == DDPM_ACT_SILU;

  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const int kt = wj % g.KT, slot = (wj / g.KT) * 8 + xcd;
  if (slot >= g.NS) return;
  const int part = slot % g.parts, it0 = (slot / g.parts) * g.IPW;
  const int nitems = min(g.IPW, g.NIT - it0);
  const int r0 = part * g.TR;
  const int n_first = it0 * g.TI, n_end = n_first + nitems * g.TI;
  const int last = g.nchunks - 1;

  // ---- staging roles: this wave stages channels `wave` and `wave + 4` of every chunk ------------------------
  const int st = lane;
  const int row_lo = max(0, 2 * r0 - 1), row_hi = min(a.Ho, 2 * (r0 + g.TR) + 1);
  const int npx = (row_hi - row_lo) * a.Wo;
  int pix[NR], pw[NR], tik[NR];  // pw: offset inside ONE channel tile of P (-> dump slot when out of range)
  bool pvalid[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int ti = k / g.NRI, e = lane + 64 * (k - ti * g.NRI);
    const bool valid = ti < g.TI && e < npx;
    const int row = row_lo + e / a.Wo, col = e % a.Wo;
    tik[k] = ti;
    pvalid[k] = valid;
    pix[k] = valid ? (row * a.Wo + col) * 4 : (int)0x80000000;
    pw[k] = (ti * (2 * g.TR + 2) + row - (2 * r0 - 1)) * g.PW + col + 1;
  }
  int tbase;  // patch origin of tile st inside one channel tile of P
  {
    const int per = g.TR * g.TWc;
    const int ti = st / per, rem = st - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tbase = (ti * (2 * g.TR + 2) + 2 * tr) * g.PW + 2 * tc;
  }
  const int bytes1 = a.B * a.C1 * g.HW * 4, bytes2 = a.B * a.C2 * g.HW * 4;
  const __amdgpu_buffer_rsrc_t rs_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sh =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

  // operand images [xi 16][kp 2][lhi 2][cout or tile 64][2]; pair p = (kp = p >> 4, xi = p & 15)
  const int ub = (lhi * kWK + cb * 32 + l31) * 2;
  const int vb = kWUF + (lhi * kWT + tb * 32 + l31) * 2;
  const float *usrc = a.w_wino + (size_t)kt * g.nchunks * kWUF + wave * 8 * 256;  // 8 x 1 KB per wave and chunk

  f32x16 acc[16];

  float praw[2][NR], gs[2][NGS], gh[2][NGS], dreg[16], tt[16];
  auto dma_u = [&](int i, int ch, int nb) {
    const float *ubase = usrc + (size_t)ch * kWUF + i * 256;  // uniform
    __builtin_amdgcn_global_load_lds(ubase + lane * 4, smem + nb + (wave * 8 + i) * 256, 16, 0, 0);
  };
  auto load_px = [&](int h, int k, int n, int ch) {  // h = 0 / 1: channel wave / wave + 4 of the chunk
    const int cg = ch * kWC + wave + 4 * h, ni = min(n + tik[k], a.B - 1);
    const bool first = cg < a.C1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2, 0x00020000);
    const int soff = first ? (ni * a.C1 + cg) * g.HW * 4 : (ni * a.C2 + cg - a.C1) * g.HW * 4;
    praw[h][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pix[k], soff, 0));
    if (AFFINE && (!ONEIMG || k == 0)) {
      const int goff = (ni * g.Cin + cg) * 4;
      gs[h][ONEIMG ? 0 : k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, vzero, goff, 0));
      gh[h][ONEIMG ? 0 : k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, vzero, goff, 0));
    }
  };
  auto activate_px = [&](int h, int k, int pb) {
    const float x = praw[h][k];
    float v;
    if (AFFINE) {
      const float sa = gs[h][ONEIMG ? 0 : k], sb = gh[h][ONEIMG ? 0 : k];
      const float u = __builtin_fmaf(x, sa, sb);
      const float t = __builtin_fmaf(x, -1.44269504088896341f * sa, -1.44269504088896341f * sb);
      v = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
    } else {
      const float sv = silu_fast(x);
      v = silu ? sv : x;
    }
    P[pb + (pvalid[k] ? (wave + 4 * h) * g.PCH + pw[k] : PB + lane)] = v;
  };
  auto read_patch = [&](int h, int i, int pb) {
    const float *p = P + pb + (wave + 4 * h) * g.PCH + tbase + i * g.PW;
#pragma unroll
    for (int j = 0; j < 4; ++j) dreg[4 * i + j] = p[j];
  };
  auto row_transform = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tt[0 * 4 + j] = dreg[0 * 4 + j] - dreg[2 * 4 + j];
      tt[1 * 4 + j] = dreg[1 * 4 + j] + dreg[2 * 4 + j];
      tt[2 * 4 + j] = dreg[2 * 4 + j] - dreg[1 * 4 + j];
      tt[3 * 4 + j] = dreg[1 * 4 + j] - dreg[3 * 4 + j];
    }
  };
  auto col_commit = [&](int h, int i, int nb) {
    const int sc = wave + 4 * h;  // channel sc = 4 kp + 2 e + lhi  ->  V [xi][kp][lhi][tile][e]
    float *vl = smem + nb + kWUF + (((sc >> 2) * 2 + (sc & 1)) * kWT + st) * 2 + ((sc >> 1) & 1);
    vl[(i * 4 + 0) * kWC * kWT] = tt[i * 4 + 0] - tt[i * 4 + 2];
    vl[(i * 4 + 1) * kWC * kWT] = tt[i * 4 + 1] + tt[i * 4 + 2];
    vl[(i * 4 + 2) * kWC * kWT] = tt[i * 4 + 2] - tt[i * 4 + 1];
    vl[(i * 4 + 3) * kWC * kWT] = tt[i * 4 + 1] - tt[i * 4 + 3];
  };
  auto advance = [&](int &n, int &ch) {
    if (ch < last) {
      ++ch;
    } else if (n + g.TI < n_end) {
      n += g.TI;
      ch = 0;
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  for (int i = tid; i < 2 * PBS; i += 256) P[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_u(i, 0, 0);
  int nL = n_first, chL = 0;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < NR; ++k) load_px(h, k, nL, chL);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < NR; ++k) activate_px(h, k, c * PBS);
    advance(nL, chL);
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) read_patch(h, i, 0);
    row_transform();
#pragma unroll
    for (int i = 0; i < 4; ++i) col_commit(h, i, 0);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < NR; ++k) load_px(h, k, nL, chL);
  advance(nL, chL);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NR + (AFFINE ? 4 * NGS : 0)) : "memory");
  __builtin_amdgcn_sched_barrier(0);

  // One chunk = 64 MFMA steps per wave (32 pairs); the staging of three later chunks is cut into slices, one per
  // step, pinned with sched_barriers:
  //   step 0..7    U eighth s by DMA           step 8..16 / 17..25  patch rows, row transform, V writes of channel
  //   wave / wave + 4      step 28..  activation of pixel round s - 28 -> P      step 44..  loads of round s - 44
  int c = 0;
  f2 av[4], bv[4];  // operand ring: four pairs, running ACROSS chunk (and item) boundaries
  auto load_pair_at = [&](int slot, int buf, int p) {
    const int imm = (((p & 15) * 2 + (p >> 4)) * 2 * 64 * 2) * 4;
    av[slot] = lds_read_b64((buf + ub) * 4, imm);
    bv[slot] = lds_read_b64((buf + vb) * 4, imm);
  };
#pragma unroll
  for (int p = 0; p < 4; ++p) load_pair_at(p, 0, p);
  __builtin_amdgcn_sched_barrier(0);
  auto chunk = [&](auto first_c, int ch_cur) {
    constexpr bool FIRST = decltype(first_c)::value;
    const int cbuf = (c & 1) * BUF;
    const int nb = BUF - cbuf;
    const int pb_t = ((c + 1) & 1) * PBS, pb_a = (c & 1) * PBS;
    const int ch_u = ch_cur < last ? ch_cur + 1 : 0;
    auto slice = [&](int s) {
      if (s < 8) {
        dma_u(s, ch_u, nb);                       // first: 48 steps to land, and OLDER than this chunk's pixel loads
      } else if (s >= 8 && s < 12) {
        read_patch(0, s - 8, pb_t);
      } else if (s == 12) {
        row_transform();
      } else if (s >= 13 && s < 17) {
        col_commit(0, s - 13, nb);
      } else if (s >= 17 && s < 21) {
        read_patch(1, s - 17, pb_t);
      } else if (s == 21) {
        row_transform();
      } else if (s >= 22 && s < 26) {
        col_commit(1, s - 22, nb);
      } else if (s >= 27 && s < 27 + 2 * NR) {
        activate_px((s - 27) / NR, (s - 27) % NR, pb_a);   // pixels requested at steps 40.. of the PREVIOUS chunk
      } else if (s >= 40 && s < 40 + 2 * NR) {
        load_px((s - 40) / NR, (s - 40) % NR, nL, chL);
      }
    };
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      // six operand reads (three pairs) are always in flight behind this pair's
      if (FIRST && p < 16)
        mfma_agpr_first_wait<6>(acc[p & 15], av[p & 3][0], bv[p & 3][0]);
      else
        mfma_agpr_wait<6>(acc[p & 15], av[p & 3][0], bv[p & 3][0]);
      slice(2 * p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_agpr(acc[p & 15], av[p & 3][1], bv[p & 3][1]);
      if (p == 28) {
        // Every read of this chunk's operand buffer has been issued (pair 31 went out after pair 27) and every
        // staging write of the next chunk's buffer is done: ONE barrier here covers both hazards, and the next
        // chunk's first operands are fetched under this chunk's last MFMAs -- no restart bubble at the boundary.
        // The U DMAs are older than this chunk's pixel loads: the counted wait leaves those loads in flight.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NR + (AFFINE ? 4 * NGS : 0)) : "memory");
      }
      if (p + 4 < 32) load_pair_at(p & 3, cbuf, p + 4);
      else load_pair_at(p & 3, nb, p + 4 - 32);
      slice(2 * p + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance(nL, chL);
    ++c;
  };

  for (int n_cur = n_first; n_cur < n_end; n_cur += g.TI) {
    chunk(std::true_type{}, 0);
    for (int ch = 1; ch <= last; ++ch) chunk(std::false_type{}, ch);

    // ---- end of an item: Y = A^T M A, register-only (this wave holds all 16 positions of its tiles) --------------
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' 16 passes
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int tq = tb * 32 + (elane & 31);
    const int per = g.TR * g.TWc;
    const int ti = tq / per, rem = tq - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    const int n = n_cur + ti;
    const int nn = min(n, a.B - 1);
    const int co_base = kt * kWK + cb * 32 + 4 * (elane >> 5);
    const size_t obase = ((size_t)nn * a.Cout + co_base) * g.HW + (size_t)(2 * (r0 + tr)) * a.Wo + 2 * tc;
    // Y = A^T M A streamed over the 16 position tiles, one AGPR tuple at a time (a factored form that touches
    // element r of all 16 tuples at once makes hipcc copy every accumulator into arch VGPRs first and spill the
    // loop's registers): y_ab += A^T[a][i] * A^T[b][j] * M[i][j], all coefficients 0 / +1 / -1.
    f32x16 y00, y01, y10, y11;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int i = x >> 2, j = x & 3;
      f32x16 m;  // one AGPR tuple -> arch VGPRs, right here (a plain copy is hoisted and batched by hipcc)
      asm volatile("" : "=v"(m) : "0"(acc[x]));
      const int c00 = kAT[0][i] * kAT[0][j], c01 = kAT[0][i] * kAT[1][j];
      const int c10 = kAT[1][i] * kAT[0][j], c11 = kAT[1][i] * kAT[1][j];
      if (c00) y00 = (x == 0) ? m : y00 + m;                               // +1 only
      if (c01) y01 = (x == 1) ? m : (c01 > 0 ? y01 + m : y01 - m);
      if (c10) y10 = (x == 4) ? m : (c10 > 0 ? y10 + m : y10 - m);
      if (c11) y11 = (x == 5) ? m : (c11 > 0 ? y11 + m : y11 - m);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // 8 couts at a time: addends requested first, then added and stored
      f2 ra[8], rb[8];
      float addv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = half * 8 + q;
        const int cof = (r & 3) + 8 * (r >> 2);
        const size_t o = obase + (size_t)cof * g.HW;
        ra[q] = rb[q] = f2{0.f, 0.f};
        if (a.residual) {
          ra[q] = *reinterpret_cast<const f2 *>(a.residual + o);
          rb[q] = *reinterpret_cast<const f2 *>(a.residual + o + a.Wo);
        }
        addv[q] = (a.bias ? a.bias[co_base + cof] : 0.f) +
                  (a.chan_add ? a.chan_add[(size_t)nn * a.chan_add_stride + co_base + cof] : 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = half * 8 + q;
        if (n < a.B) {
          const size_t o = obase + (size_t)((r & 3) + 8 * (r >> 2)) * g.HW;
          *reinterpret_cast<f2 *>(a.out + o) = f2{y00[r] + addv[q] + ra[q][0], y01[r] + addv[q] + ra[q][1]};
          *reinterpret_cast<f2 *>(a.out + o + a.Wo) = f2{y10[r] + addv[q] + rb[q][0], y11[r] + addv[q] + rb[q][1]};
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

// out = sum of the S partial slabs (fixed order 0, 1, ..) + bias + temb + residual: the epilogue of a split launch.
// STATS: also the GroupNorm statistics of the finished tensor (desc.stats_out): a plane of HW4 float4 is GL = min(HW4, 64)
// consecutive lanes per slice; lanes merge {mean, M2} pairwise on the DPP path, the slice's last lane stores them.
template <bool STATS>
__global__ __launch_bounds__(256) void wino_split_reduce_kernel(const float *__restrict__ part, long long pstride, int S,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ chan_add, int chan_stride,
                                                                const float *__restrict__ residual,
                                                                float *__restrict__ out, int Cout, int HW4,
                                                                long long total4, float *__restrict__ stats, int GL) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= total4) return;  // (whole slices leave: total4 is a multiple of HW4)
  const int c = (int)((i / HW4) % Cout), n = (int)(i / ((long long)HW4 * Cout));
  v4f v = reinterpret_cast<const v4f *>(part)[i];
  for (int sp = 1; sp < S; ++sp) v += reinterpret_cast<const v4f *>(part + sp * pstride)[i];
  float add = bias ? bias[c] : 0.f;
  if (chan_add) add += chan_add[(size_t)n * chan_stride + c];
  v += add;
  if (residual) v += reinterpret_cast<const v4f *>(residual)[i];
  reinterpret_cast<v4f *>(out)[i] = v;
  if (STATS) {
    float mean = 0.25f * ((v[0] + v[1]) + (v[2] + v[3]));
    const v4f dv = v - mean;
    float m2 = (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
    group_moments_last_lane(mean, m2, 4.f, GL);
    const int e = (int)(i % HW4);
    if ((e & (GL - 1)) == GL - 1)
      reinterpret_cast<float2 *>(stats)[((size_t)n * Cout + c) * (HW4 / GL) + e / GL] = make_float2(mean, m2);
  }
}

int wino_split_reduce_stats_parts(int HW) {
  if (HW & 3) return 0;
  const int HW4 = HW / 4;
  if (HW4 == 4 || HW4 == 16 || HW4 == 64) return 1;
  return HW4 % 64 == 0 && HW4 / 64 <= 8 ? HW4 / 64 : 0;
}

// d.out = sum of the S slabs of d.scratch + d.bias + d.chan_add + d.residual (shared with conv_wino44.hip); d.stats_out: the
// caller has checked wino_split_reduce_stats_parts(HW) > 0
int launch_wino_split_reduce(const ddpm_conv_desc &d, int S, long long pstride, int HW, hipStream_t s) {
  const long long total4 = (long long)d.B * d.Cout * HW / 4;  // H, W even: HW % 4 == 0
  const int HW4 = HW / 4, GL = HW4 < 64 ? HW4 : 64;
  const dim3 grid((unsigned)((total4 + 255) / 256));
  if (d.stats_out && wino_split_reduce_stats_parts(HW) > 0)
    hipLaunchKernelGGL(wino_split_reduce_kernel<true>, grid, dim3(256), 0, s, d.scratch, pstride, S, d.bias, d.chan_add,
                       d.chan_add_stride, d.residual, d.out, d.Cout, HW4, total4, d.stats_out, GL);
  else
    hipLaunchKernelGGL(wino_split_reduce_kernel<false>, grid, dim3(256), 0, s, d.scratch, pstride, S, d.bias, d.chan_add,
                       d.chan_add_stride, d.residual, d.out, d.Cout, HW4, total4, nullptr, GL);
  DDPM_CHECK_LAUNCH();
  return 0;
}

size_t conv_wino_scratch_floats(const ddpm_conv_desc &d) {
  WinoGeom g;
  if (!conv_wino_supported(d) || !wino_geom(d, g) || g.S == 1) return 0;
  return (size_t)g.S * d.B * d.Cout * g.HW;
}

int launch_conv_wino(const ddpm_conv_desc &d, hipStream_t s) {
  WinoGeom g;
  if (!d.w_wino || !wino_geom(d, g)) {
    set_error("conv_wino: unsupported shape");
    return DDPM_EINVAL;
  }
  const size_t lds = ((size_t)2 * (kWUF + kWVF) + 2 * kWC * g.PCH + 64) * sizeof(float);
  typedef void (*kern_t)(const ddpm_conv_desc, const WinoGeom);
  static const kern_t kerns[2][2][3] = {
      {{conv_wino_kernel<false, 4, false>, conv_wino_kernel<false, 5, false>, conv_wino_kernel<false, 6, false>},
       {conv_wino_kernel<false, 4, true>, conv_wino_kernel<false, 5, true>, conv_wino_kernel<false, 6, true>}},
      {{conv_wino_kernel<true, 4, false>, conv_wino_kernel<true, 5, false>, conv_wino_kernel<true, 6, false>},
       {conv_wino_kernel<true, 4, true>, conv_wino_kernel<true, 5, true>, conv_wino_kernel<true, 6, true>}}};
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 12; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 6][i / 3 % 2][i % 3]),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const size_t out_floats = (size_t)d.B * d.Cout * g.HW;
  if (g.S > 1 && (!d.scratch || d.scratch_floats < g.S * out_floats)) {  // no scratch buffer: run unsplit
    g.S = 1;
    g.grid = g.KT * ((g.NS + 7) / 8) * 8;
  }
  ddpm_conv_desc dk = d;  // the descriptor the kernel sees
  if (conv_wino_stats_parts(d) == 0) dk.stats_out = nullptr;  // the ABI's promise: parts = 0 -> stats_out is ignored
  if (g.S > 1) {  // partial sums go to the scratch slabs, the addends to the reduce pass
    g.pstride = (long long)out_floats;
    dk.out = d.scratch;
    dk.bias = nullptr;
    dk.chan_add = nullptr;
    dk.residual = nullptr;
  }
  const int rounds = g.TI * g.NRI;
  kern_t kern = kerns[d.gscale ? 1 : 0][g.TI == 1 ? 1 : 0][rounds <= 4 ? 0 : rounds - 4];
  if (d.dims == 3) {  // the same variants with the 3-D stream (VQ-VAE residual units: no prologue, whole slices per item;
                      // 3-D UNet: GroupNorm + SiLU prologue, concat, four 8x8 slices per item)
    static const kern_t kerns3d[2][2][3] = {
        {{conv_wino_kernel<false, 4, false, true>, conv_wino_kernel<false, 5, false, true>, conv_wino_kernel<false, 6, false, true>},
         {conv_wino_kernel<false, 4, true, true>, conv_wino_kernel<false, 5, true, true>, conv_wino_kernel<false, 6, true, true>}},
        {{conv_wino_kernel<true, 4, false, true>, conv_wino_kernel<true, 5, false, true>, conv_wino_kernel<true, 6, false, true>},
         {conv_wino_kernel<true, 4, true, true>, conv_wino_kernel<true, 5, true, true>, conv_wino_kernel<true, 6, true, true>}}};
    static bool attr3_done = false;
    if (!attr3_done) {
      for (int i = 0; i < 12; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns3d[i / 6][i / 3 % 2][i % 3]),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr3_done = true;
    }
    kern = kerns3d[d.gscale ? 1 : 0][g.TI == 1 ? 1 : 0][rounds <= 4 ? 0 : rounds - 4];
  }
  // DDPM_WINO_WAVES=4: the one-wave-per-SIMD variant (A/B switch; see conv_wino4_kernel)
  static const bool four = getenv("DDPM_WINO_WAVES") && atoi(getenv("DDPM_WINO_WAVES")) == 4;
  int threads = 512;
  size_t lds_bytes = lds;
  if (four && !g.up && d.dims != 3 && g.S == 1) {
    static const kern_t kerns4[2][2][3] = {
        {{conv_wino4_kernel<false, 4, false>, conv_wino4_kernel<false, 5, false>, conv_wino4_kernel<false, 6, false>},
         {conv_wino4_kernel<false, 4, true>, conv_wino4_kernel<false, 5, true>, conv_wino4_kernel<false, 6, true>}},
        {{conv_wino4_kernel<true, 4, false>, conv_wino4_kernel<true, 5, false>, conv_wino4_kernel<true, 6, false>},
         {conv_wino4_kernel<true, 4, true>, conv_wino4_kernel<true, 5, true>, conv_wino4_kernel<true, 6, true>}}};
    static bool attr4_done = false;
    if (!attr4_done) {
      for (int i = 0; i < 12; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns4[i / 6][i / 3 % 2][i % 3]),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr4_done = true;
    }
    kern = kerns4[d.gscale ? 1 : 0][g.TI == 1 ? 1 : 0][rounds <= 4 ? 0 : rounds - 4];
    threads = 256;
    lds_bytes = ((size_t)2 * (kWUF + kWVF) + 2 * (kWC * g.PCH + 64)) * sizeof(float);
  }
  if (g.up) {
    static const kern_t up_kerns[2] = {conv_wino_up_kernel<2>, conv_wino_up_kernel<6>};
    static bool up_attr_done = false;
    if (!up_attr_done) {
      for (int i = 0; i < 2; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(up_kerns[i]),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      up_attr_done = true;
    }
    kern = up_kerns[rounds <= 2 ? 0 : 1];
  }
  const double M = (double)g.NIMG * g.HW;
  // algorithmic work = the direct convolution's: 2*M*Cout*Cin*9 (*3 depth taps); 16/36 of it is executed
  const double flops = 2.0 * M * d.Cout * (double)g.Cin * 9 * g.nkd;
  const double bytes = 4.0 * (M * g.Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * g.Cin * 9 * g.nkd);
  const char *kname = d.dims == 3 ? (d.gscale ? "conv3d_wino_gn_silu" : "conv3d_wino") : g.up ? "conv3x3_wino_up" : d.gscale ? "conv3x3_wino_gn_silu" : "conv3x3_wino";
  char kshape[160];
  if (g_prof_on && sw().prof_shapes) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  hipLaunchKernelGGL(kern, dim3(g.grid), dim3(threads), lds_bytes, s, dk, g);
  DDPM_CHECK_LAUNCH();
  if (g.S > 1) {
    ddpm_conv_desc dr = d;
    dr.stats_out = nullptr;  // (ddpm_conv_stats_parts is 0 for this kernel: the field is ignored)
    return launch_wino_split_reduce(dr, g.S, g.pstride, g.HW, s);
  }
  return 0;
}

// ---- weights: torch [Cout][Cin][3][3] -> U = G g G^T, packed as the LDS image the kernel's MFMAs read:
//   [cout tile 64][chunk 8][xi 16][kp 2][lhi 2][cout 64][e 2],  channel of the chunk = 4 kp + 2 e + lhi
// G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// A 3x3x3 weight (nkd = 3) is transformed per depth tap kd: slab [cout tile][kd][chunk] holds G w[:, :, kd] G^T.
__global__ void wino_pack_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin, int nkd) {
  const int64_t total = (int64_t)Cout * Cin * nkd;
  const int nchunks = Cin / kWC;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kd = (int)(i % nkd);
    const int ci = (int)((i / nkd) % Cin), o = (int)(i / ((int64_t)nkd * Cin));
    const float *w = src + ((size_t)o * Cin + ci) * 9 * nkd + kd * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // G g
      t[0][j] = w[0 * 3 + j];
      t[1][j] = 0.5f * (w[0 * 3 + j] + w[1 * 3 + j] + w[2 * 3 + j]);
      t[2][j] = 0.5f * (w[0 * 3 + j] - w[1 * 3 + j] + w[2 * 3 + j]);
      t[3][j] = w[2 * 3 + j];
    }
    const int tile = o / kWK, k64 = o % kWK, ch = ci / kWC, cl = ci % kWC;
    const int lhi = cl & 1, kp = cl >> 2, e = (cl >> 1) & 1;
    float *d = dst + (((size_t)tile * nkd + kd) * nchunks + ch) * kWUF + ((kp * 2 + lhi) * kWK + k64) * 2 + e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // (G g) G^T
      d[(r * 4 + 0) * kWC * kWK] = t[r][0];
      d[(r * 4 + 1) * kWC * kWK] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      d[(r * 4 + 2) * kWC * kWK] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      d[(r * 4 + 3) * kWC * kWK] = t[r][2];
    }
  }
}

size_t wino_weight_floats(int Cout, int Cin) {
  if (Cout % kWK || Cin % kWC) return 0;
  return (size_t)16 * Cout * Cin;
}

int launch_pack_wino_weight(const float *w_raw, float *w_wino, int Cout, int Cin, hipStream_t s, int nkd) {
  DDPM_CHECK_ARG(wino_weight_floats(Cout, Cin) != 0 && (nkd == 1 || nkd == 3), "wino pack: Cout %% 64 or Cin %% 8 != 0");
  const int64_t total = (int64_t)Cout * Cin * nkd;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_wino, Cout, Cin, nkd);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
