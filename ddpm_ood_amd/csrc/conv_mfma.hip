// conv_mfma.hip -- fused 3x3 / 1x1 convolution (and Linear, HW == 1) for gfx950.
//
// Replaces F.conv2d / F.linear inside DiffusionModelUNet.forward (reference call site
// /root/reference/src/trainers/reconstruct.py:151-153; layer list SURVEY.md 2.3) together with
// the GroupNorm-affine + SiLU prologue, virtual torch.cat, nearest-x2 upsample, stride-2
// downsample, bias, "+ temb[:, :, None, None]" and residual epilogues.
//
// Design (MI355X-first, not a cuDNN translation):
//   * NCHW stays NCHW.  A wave64 f32 MFMA (v_mfma_f32_32x32x2_f32) takes ONE f32 per lane for
//     each operand, lane l supplying row/col (l & 31) of k-slice (l >> 5).  With
//     A = weights[cout][k] and B = input[k][pixel], 32 consecutive lanes read 32 consecutive
//     pixels of one channel plane -- exactly the contiguous direction of NCHW -- so operand
//     fetches from LDS are bank-conflict-free ds_read_b32 and the D tile (rows = cout,
//     cols = pixel) stores 128-byte contiguous rows back to NCHW.
//   * f32-input MFMA is bit-exact fp32 FMA (k-ordered fmaf chain) at the fp32 vector peak but
//     needs 1 LDS dword per operand per 2048 FLOP instead of 1 per 2 FLOP, so LDS and VGPR
//     bandwidth stop being the limiter of a direct fp32 convolution.
//   * Workgroup tile: MT (128 or 64) output pixels -- whole rows of one image, or several whole
//     images when H*W < MT -- x 128 output channels, 4 waves, each wave 64x64 (MT = 128) or
//     64 px x 32 cout (MT = 64, used when the 128-pixel grid would leave CUs idle).
//   * Input channels are consumed in chunks of 4.  LDS is double-buffered
//     ([9][4][128] weights + [4][rows+2][W+2] haloed input per buffer, ~22 KB each): while the
//     72 MFMAs of chunk q run from buffer q & 1, the same wave commits chunk q + 1 (already in
//     registers; GroupNorm affine + SiLU applied here, zero halo written after the activation)
//     into the other buffer and issues the global loads of chunk q + 2.  The staging work is
//     spread over the chunk's k-steps in program order so that it issues in the shadow of the
//     wave's own 64-cycle MFMAs (hipcc's scheduler is left free to move it: pinning the order with
//     sched_barrier measured 2 % slower).  One barrier per chunk, three workgroups per CU.
//   * Weights arrive pre-packed as [cout_tile][chunk][tap][4][128]: a chunk is one contiguous
//     18 KB burst of coalesced 16-byte loads.  The input tile is loaded once per chunk and
//     reused by all 9 taps.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector types: arrays of HIP's struct float4 stay in scratch (SROA does not split them)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct ConvGeom {
  int M;        // output pixels of the launch: B * (output slices) * Ho * Wo (per parity for TRANSPOSE2 / folded)
  int HWo, HWi;
  int Di, Do;   // stored input / output depth (1 for 2-D); an "image" below is one (n, d) slice
  int NI;       // B * DS slices walked by the tiles
  int DS;       // slices per batch item: Do, or the LOW-RES depth Di for a 3-D TRANSPOSE2 (output slice 2 d + pz)
  int MT;       // pixels per workgroup tile (128 or 64)
  int TI, TH;   // images per tile, output rows per tile (per image)
  int TPX;      // valid output pixels per tile = TI * TH * Wo (<= MT; < MT for ragged extents such as 28 x 28)
  int TPI;      // tiles per image (TI == 1) -- 0 when a tile holds several whole images
  int ntiles;
  int IR, RS;   // LDS rows per image slot, LDS row stride
  int IRS;      // IR * RS
  int PS;       // LDS plane size (floats per channel)
  int pad;      // 1 for 3x3 / 4x4, 0 for 1x1
  int s;        // input step per output pixel (2 for stride-2)
  int Cin, nchunks;
  // ---- 3-D: the depth taps are part of the chunk stream (chunk = (depth tap, channel group)) ----------------
  int is3d;     // the depth logic below is active
  int kd0, nkd; // depth taps walked by this launch: kd0 .. kd0 + nkd - 1 (a depth-1 volume only has its centre tap)
  int nchunks_c;  // channel chunks per depth tap; nchunks = nkd * nchunks_c
  long long slab;      // floats between two depth-tap slabs of w_packed
  // input depth of tap kd for output slice dz: dv = dmul * dz + kd + doff (valid in [0, Dv)), stored slice dv >> dshift
  int dmul, doff, Dv, dshift;
  // ---- output parities (blockIdx.z): folded nearest-x2 upsample (4), ConvTranspose k4 s2 p1 (4 in 2-D, 8 in 3-D):
  // the tile walks LOW-RES pixels, tap (r, c) reads LDS at (py + r, px + c), the output pixel is (2h + py, 2w + px)
  int npar;
  long long par_slab;  // floats between two parity slabs of w_packed
  // ---- split-K (blockIdx.z of the non-parity launches): a launch with fewer workgroups than half the CUs -- the 4^3 / 2^3
  // levels of the latent UNet, small batches -- gives every tile to `ksplit` workgroups, each walking nchunks / ksplit
  // chunks; partial outputs go to slabs of d.scratch, conv_wino.hip's reduce pass adds them in a fixed order
  int ksplit;
  long long pstride;   // floats between two partial-output slabs
};

static bool make_geom(const ddpm_conv_desc &d, int MT, ConvGeom &g, int ps_cap = 0) {
  const int Cin = d.C1 + d.C2;
  g.Cin = Cin;
  g.MT = MT;
  g.HWo = d.Ho * d.Wo;
  g.HWi = d.Hi * d.Wi;
  g.Di = d.Di > 1 ? d.Di : 1;
  g.Do = d.Do > 1 ? d.Do : 1;
  g.is3d = d.dims == 3 && d.ksize != 1;
  g.npar = 1;
  g.par_slab = 0;
  g.DS = g.Do;
  g.NI = d.B * g.DS;
  g.M = g.NI * g.HWo;
  g.pad = d.ksize == 1 ? 0 : 1;
  g.s = d.mode == DDPM_CONV_STRIDE2 ? 2 : 1;
  if (g.HWo > MT) {
    // whole rows of one image: the largest divisor of Ho whose rows fit the tile
    if (d.Wo > MT) return false;
    g.TI = 1;
    g.TH = 0;
    for (int th = MT / d.Wo; th >= 1; --th)
      if (d.Ho % th == 0) { g.TH = th; break; }
    g.TPI = d.Ho / g.TH;
    g.ntiles = g.NI * g.TPI;
  } else {
    // several whole images per tile (the last tile may be short of images)
    g.TI = MT / g.HWo;
    g.TH = d.Ho;
    g.TPI = 0;
  }
  // stride 2: k = 3 reads input columns 2x-1 .. 2x+1, k = 4 reads 2x-1 .. 2x+2
  const int IC = (g.s == 2) ? (2 * d.Wo + d.ksize - 2) : (d.Wo + 2 * g.pad);
  g.IR = (g.s == 2) ? (2 * g.TH + d.ksize - 2) : (g.TH + 2 * g.pad);
  g.RS = IC;
  g.IRS = g.IR * g.RS;
  if (g.TPI == 0) {
    // tiny images (1x1, 2x2: the halo outweighs the pixels): cap the images per tile so that the staged
    // plane still fits the instantiated variants (geom_ok); the tile then carries a few idle MFMA columns
    const int max_ps = ps_cap ? ps_cap : d.ksize == 1 ? 256 : (d.gscale ? 2 * 256 : 3 * 256);
    if (g.TI * g.IRS > max_ps && max_ps / g.IRS >= 1) g.TI = max_ps / g.IRS;
    g.ntiles = (g.NI + g.TI - 1) / g.TI;
  }
  g.TPX = g.TI * g.TH * d.Wo;
  g.PS = g.TI * g.IRS;
  g.nchunks_c = Cin / kConvCc;
  // depth taps
  g.kd0 = 0;
  g.nkd = 1;
  g.slab = 0;
  g.dmul = g.s; g.doff = -1; g.Dv = g.Di; g.dshift = 0;
  if (g.is3d) {
    g.nkd = d.ksize;
    g.slab = (long long)d.Cout * Cin * d.ksize * d.ksize;
    if (d.mode == DDPM_CONV_UPSAMPLE2) { g.dmul = 1; g.Dv = g.Do; g.dshift = 1; }
    if (d.ksize == 3 && g.Di == 1 && g.Do == 1) { g.kd0 = 1; g.nkd = 1; }  // the outer taps only see padding
  }
  g.nchunks = g.nkd * g.nchunks_c;
  return true;
}

static bool geom_ok(const ddpm_conv_desc &d, const ConvGeom &g) {
  if (g.PS > 3 * 256) return false;
  if (d.gscale && g.PS > 2 * 256) return false;  // instantiated AFFINE variants: NPOS <= 2
  if (d.ksize == 1 && g.PS > 256) return false;
  if (4 * g.TPX < 3 * g.MT) return false;  // < 75 % of the tile's MFMA columns would carry real pixels
  return true;
}

// Pick the pixel-tile size: 128 unless that grid cannot give every CU two workgroups.
static bool pick_geom(const ddpm_conv_desc &d, ConvGeom &g, int ps_cap = 0) {
  ConvGeom g128, g64;
  const bool ok128 = make_geom(d, 128, g128, ps_cap) && geom_ok(d, g128);
  const bool ok64 = make_geom(d, 64, g64, ps_cap) && geom_ok(d, g64);
  if (!ok128 && !ok64) return false;
  const long wg128 = ok128 ? (long)g128.ntiles * (d.Cout / kConvNT) : 0;
  static const long min_wg128 = getenv("DDPM_CONV_MIN_WG128") ? atol(getenv("DDPM_CONV_MIN_WG128")) : 512;
  // prefer 128-pixel tiles when they fill the chip and waste no more MFMA columns than 64-pixel tiles would
  const bool fill128 = ok128 && (!ok64 || (long)g128.TPX * g64.MT >= (long)g64.TPX * g128.MT);
  if (ok128 && ((wg128 >= min_wg128 && fill128) || !ok64)) {
    g = g128;
  } else {
    g = g64;
  }
  return true;
}

// The low-res view of a parity launch (folded nearest-x2 upsample, ConvTranspose k4 s2 p1): tiles walk the INPUT
// pixels; every workgroup of grid.z computes one output parity with 2 x 2 (x 2) taps.
static ddpm_conv_desc lowres_view(const ddpm_conv_desc &d) {
  ddpm_conv_desc lr = d;
  lr.mode = DDPM_CONV_NORMAL;
  lr.ksize = 3;  // halo of one pixel on each side, as for a 3x3
  lr.Ho = d.Hi;
  lr.Wo = d.Wi;
  lr.Do = d.Di;
  return lr;
}

static bool transpose_geom(const ddpm_conv_desc &d, ConvGeom &g) {
  const int Cin = d.C1 + d.C2;
  if (d.ksize != 4 || d.gscale || d.C2 || Cin % 8 || d.Cout % kConvNT) return false;
  if (d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi) return false;
  const bool is3d = d.dims == 3;
  if (is3d && d.Do != 2 * (d.Di > 1 ? d.Di : 1)) return false;
  ddpm_conv_desc lr = lowres_view(d);
  lr.dims = 0;  // geometry of the in-plane tiling only; the depth walk is set below
  if (!pick_geom(lr, g, 2 * 256) || g.PS > 2 * 256) return false;  // parity variants: NPOS <= 2
  g.npar = is3d ? 8 : 4;
  g.nchunks_c = Cin / 8;               // KG = 2: eight channels per chunk
  g.par_slab = (long long)d.Cout * Cin * (is3d ? 8 : 4);
  if (is3d) {
    g.is3d = 1;
    g.Di = d.Di > 1 ? d.Di : 1;
    g.Do = d.Do;
    g.DS = g.Di;
    g.NI = d.B * g.DS;
    // pick_geom above tiled B * 1 slices: redo the tile count for B * Di slices
    g.ntiles = g.TPI > 0 ? g.NI * g.TPI : (g.NI + g.TI - 1) / g.TI;
    g.kd0 = 0; g.nkd = 2;
    g.slab = (long long)d.Cout * Cin * 4;
    g.dmul = 1; g.doff = -1; g.Dv = g.Di; g.dshift = 0;  // + pz, added in the kernel
  }
  g.M = g.NI * g.HWo;
  g.nchunks = g.nkd * g.nchunks_c;
  return true;
}

bool conv_mfma_supported(const ddpm_conv_desc &d) {
  if (!d.w_packed || d.force_direct) return false;
  const int Cin = d.C1 + d.C2;
  ConvGeom g;
  if (d.mode == DDPM_CONV_TRANSPOSE2) return transpose_geom(d, g);
  if (d.ksize != 1 && d.ksize != 3 && d.ksize != 4) return false;
  if (d.ksize == 4 && d.mode != DDPM_CONV_STRIDE2) return false;
  if (d.mode != DDPM_CONV_NORMAL && d.ksize == 1) return false;
  if (Cin % kConvCc || d.Cout % kConvNT) return false;
  if (d.C2 > 0 && (d.C1 % kConvCc)) return false;
  if (d.dims == 3 && d.ksize != 1 && d.mode == DDPM_CONV_UPSAMPLE2 && d.ksize != 3) return false;
  return pick_geom(d, g);
}

template <int NTAPS, int NPOS, bool AFFINE, int MT, int KG>
__global__ __launch_bounds__(256, (NPOS == 1 ? 3 : 2)) void conv_mfma_kernel(const ddpm_conv_desc a,
                                                                             const ConvGeom g) {
  // KG = groups of 4 input channels per chunk: 1 for 3x3 / 4x4 (72 / 128 MFMAs per chunk and barrier); 4 for plain
  // 1x1 convs / Linears, whose single tap would otherwise give a barrier every 8 MFMAs; 2 for the 2x2-tap
  // parity launches (folded upsample, ConvTranspose).
  constexpr int CC = kConvCc;                  // 4: granule of the packed weight layout
  constexpr int CPC = CC * KG;                 // input channels per chunk
  constexpr int WF = NTAPS * CPC * kConvNT;    // weight floats per chunk (4608 / 8192 / 2048 / 512)
  constexpr int NW4 = WF / 1024;               // full 16-byte rounds per thread
  constexpr bool HASREM = (WF % 1024) != 0;    // + one 8-byte round (512 floats)
  constexpr int NXP = NPOS * KG;               // input pieces: (position, channel group)
  constexpr int NP = NW4 + (HASREM ? 1 : 0) + NXP;  // staging pieces per chunk
  constexpr int NSTEP = NTAPS * (CPC / 2);     // k-steps (4 or 2 MFMAs each) per chunk
  constexpr int H = (NSTEP + 1) / 2;           // commit pieces go to steps [0, H), prefetch to [H, NSTEP)
  constexpr int NAB = MT == 128 ? 2 : 1;       // 32-cout blocks per wave
  constexpr bool PARITY = NTAPS == 4;          // low-res tiles, one output parity per blockIdx.z

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bufsz = WF + CPC * g.PS;           // floats per LDS buffer: [NTAPS][CPC][128] weights, [CPC][PS] input

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wco = MT == 128 ? (wave & 1) * 64 : wave * 32;  // wave's first cout inside the tile
  const int wpx = MT == 128 ? (wave >> 1) * 64 : 0;         // wave's first pixel inside the tile
  const int nt = blockIdx.y;
  // output parity of this workgroup (PARITY launches): out pixel (2 d + pz, 2 h + py, 2 w + px)
  const int par = PARITY ? blockIdx.z : 0;
  const int ks = PARITY ? 0 : blockIdx.z;                       // split-K share of this workgroup
  const int nch_s = g.nchunks / g.ksplit, q_lo = ks * nch_s, q_hi = q_lo + nch_s;
  float *const outp = a.out + (size_t)ks * g.pstride;
  const int px = par & 1, py = (par >> 1) & 1, pz = par >> 2;

  // ---- tile origin -------------------------------------------------------------------------
  const int tile = blockIdx.x;
  int n0, h0;  // first (n, d) slice of the tile, first output row inside it
  if (g.TPI > 0) {
    n0 = tile / g.TPI;
    h0 = (tile - n0 * g.TPI) * g.TH;
  } else {
    n0 = tile * g.TI;
    h0 = 0;
  }

  // ---- per-thread staging positions (chunk-invariant) -------------------------------------
  int soff[NPOS];   // offset inside one channel plane of the source, -1 => zero
  int nimg[NPOS];   // batch index of the position's slice
  int dzb[NPOS];    // dmul * (slice depth index) + doff (+ pz): add the depth tap to get the virtual input depth
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const int r = tid + 256 * j;
    soff[j] = -1;
    nimg[j] = 0;
    dzb[j] = 0;
    if (r < g.PS) {
      const int ti = r / g.IRS;
      const int rr = r - ti * g.IRS;
      const int ir = rr / g.RS;
      const int ic = rr - ir * g.RS;
      const int img = n0 + ti;       // (n, d) slice handled by this position
      const int n = img / g.DS;
      const int dz = img - n * g.DS;
      const int hv = g.s * h0 + ir - g.pad;
      const int wv = ic - g.pad;
      // bounds of the (virtual) conv input: the upsampled extent for UPSAMPLE2, else Hi x Wi
      const int Hv = (a.mode == DDPM_CONV_UPSAMPLE2) ? a.Ho : a.Hi;
      const int Wv = (a.mode == DDPM_CONV_UPSAMPLE2) ? a.Wo : a.Wi;
      if (img < g.NI && hv >= 0 && hv < Hv && wv >= 0 && wv < Wv) {
        nimg[j] = n;
        dzb[j] = g.dmul * dz + g.doff + pz;
        soff[j] = (a.mode == DDPM_CONV_UPSAMPLE2) ? ((hv >> 1) * a.Wi + (wv >> 1)) : (hv * a.Wi + wv);
      }
    }
  }

  // ---- per-lane MFMA operand bases (relative to an LDS buffer) --------------------------------
  int xb[2];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    const int qq = wpx + bb * 32 + l31;
    const int q = qq < g.TPX ? qq : 0;  // lanes past a ragged tile's last pixel compute a dummy column
    const int per_img = g.TH * a.Wo;
    const int ti = q / per_img;
    const int rem = q - ti * per_img;
    const int th = rem / a.Wo;
    const int tw = rem - th * a.Wo;
    xb[bb] = WF + lhi * g.PS + ti * g.IRS + th * g.s * g.RS + tw * g.s;
  }
  const int wb = lhi * kConvNT + wco + l31;

  f32x16 acc[NAB][2];
#pragma unroll
  for (int i = 0; i < NAB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- staging registers: one chunk in flight -----------------------------------------------
  v4f w4[NW4 > 0 ? NW4 : 1];
  v2f w2;
  float xreg[NXP][CC];
  v4f screg[NXP], shreg[NXP];
  unsigned xok = 0;  // bit xp: the staged piece is a real pixel (inside the image AND inside the depth range)

  const float *wsrc = a.w_packed + (size_t)par * g.par_slab + (size_t)nt * g.nchunks_c * WF;

  // piece p of chunk `ch` = (depth tap kdi, channel chunk cc): global -> registers
  auto prefetch_piece = [&](int p, int ch) {
    int kdi = 0, cc = ch;
    if (g.nkd > 1) {  // wave-uniform: scalar ALU
      kdi = ch / g.nchunks_c;
      cc = ch - kdi * g.nchunks_c;
    }
    const float *wp = wsrc + (size_t)(g.kd0 + kdi) * g.slab + (size_t)cc * WF;
    if (p < NW4) {
      w4[p < NW4 ? p : 0] = reinterpret_cast<const v4f *>(wp)[tid + 256 * p];
    } else if (HASREM && p == NW4) {
      w2 = reinterpret_cast<const v2f *>(wp)[NW4 * 512 + tid];
    } else {
      const int xp = p - NW4 - (HASREM ? 1 : 0);
      const int j = xp / KG, gk = xp % KG;
      const int cg0 = cc * CPC + gk * CC;
      const float *base;
      int Cs, cl0;
      if (cg0 < a.C1) {
        base = a.in1; Cs = a.C1; cl0 = cg0;
      } else {
        base = a.in2; Cs = a.C2; cl0 = cg0 - a.C1;
      }
      bool ok = soff[j] >= 0;
      int din = 0;
      if (g.is3d) {
        const int dv = dzb[j] + g.kd0 + kdi;
        ok = ok && dv >= 0 && dv < g.Dv;
        din = dv >> g.dshift;
      }
      if (ok) {
        const float *px_ = base + (((size_t)nimg[j] * Cs + cl0) * g.Di + din) * g.HWi + soff[j];
#pragma unroll
        for (int c = 0; c < CC; ++c) xreg[xp][c] = px_[(size_t)c * g.Di * g.HWi];
        if (AFFINE) {
          screg[xp] = *reinterpret_cast<const v4f *>(a.gscale + (size_t)nimg[j] * g.Cin + cg0);
          shreg[xp] = *reinterpret_cast<const v4f *>(a.gshift + (size_t)nimg[j] * g.Cin + cg0);
        }
        xok |= 1u << xp;
      } else {
#pragma unroll
        for (int c = 0; c < CC; ++c) xreg[xp][c] = 0.f;
        xok &= ~(1u << xp);
      }
    }
  };

  // piece p of the chunk held in registers -> LDS buffer at float offset `nb`
  auto commit_piece = [&](int p, int nb) {
    if (p < NW4) {
      reinterpret_cast<v4f *>(smem + nb)[tid + 256 * p] = w4[p < NW4 ? p : 0];
    } else if (HASREM && p == NW4) {
      reinterpret_cast<v2f *>(smem + nb)[NW4 * 512 + tid] = w2;
    } else {
      const int xp = p - NW4 - (HASREM ? 1 : 0);
      const int j = xp / KG, gk = xp % KG;
      const int r = tid + 256 * j;
      if (r < g.PS) {
        const bool valid = (xok >> xp) & 1u;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          float v = xreg[xp][c];
          if (AFFINE) v = v * screg[xp][c] + shreg[xp][c];
          // v_exp_f32 / v_rcp_f32 SiLU (6 instructions instead of ~45): end-to-end parity unchanged
          // (Z-score error 2.4e-6 vs 3.4e-6 with expf + IEEE divide)
          if (a.act == DDPM_ACT_SILU) v = silu_fast(v);
          if (a.act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
          smem[nb + WF + (gk * CC + c) * g.PS + r] = valid ? v : 0.f;
        }
      }
    }
  };

  // ---- prologue: chunk 0 -> LDS buffer 0, chunk 1 -> registers ----------------------------------
#pragma unroll
  for (int p = 0; p < NP; ++p) prefetch_piece(p, q_lo);
#pragma unroll
  for (int p = 0; p < NP; ++p) commit_piece(p, 0);
  if (nch_s > 1) {
#pragma unroll
    for (int p = 0; p < NP; ++p) prefetch_piece(p, q_lo + 1);
  }
  __syncthreads();

  // One chunk: MFMAs on buffer `cb`, with the commit of chunk q + 1 and the loads of chunk q + 2
  // woven between the MFMA groups.
  auto chunk = [&](auto commit_c, auto pref_c, int q) {
    constexpr bool DO_COMMIT = decltype(commit_c)::value;
    constexpr bool DO_PREF = decltype(pref_c)::value;
    const int cb = ((q - q_lo) & 1) * bufsz;
    const int nb = bufsz - cb;
    float av[2][NAB], bv[2][2];
    auto fetch = [&](int st, int slot) {
      const int t = st / (CPC / 2), kk = st % (CPC / 2);
      const int tapoff = (NTAPS == 9) ? ((t / 3) * g.RS + (t % 3))
                       : (NTAPS == 16) ? ((t >> 2) * g.RS + (t & 3))
                       : (NTAPS == 4) ? ((py + (t >> 1)) * g.RS + px + (t & 1)) : 0;
#pragma unroll
      for (int ab = 0; ab < NAB; ++ab)  // LDS weights are [channel group][tap][4][128], as packed
        av[slot][ab] = smem[cb + wb + (((kk >> 1) * NTAPS + t) * CC + 2 * (kk & 1)) * kConvNT + ab * 32];
      bv[slot][0] = smem[cb + xb[0] + 2 * kk * g.PS + tapoff];
      bv[slot][1] = smem[cb + xb[1] + 2 * kk * g.PS + tapoff];
    };
    fetch(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int cur = st & 1;
      if (st + 1 < NSTEP) fetch(st + 1, cur ^ 1);
      if (DO_COMMIT) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
          if ((p * H) / NP == st) commit_piece(p, nb);
      }
      if (DO_PREF) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
          if (H + (p * (NSTEP - H)) / NP == st) prefetch_piece(p, q + 2);
      }
#pragma unroll
      for (int ab = 0; ab < NAB; ++ab) {
        acc[ab][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][ab], bv[cur][0], acc[ab][0], 0, 0, 0);
        acc[ab][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][ab], bv[cur][1], acc[ab][1], 0, 0, 0);
      }
    }
    __syncthreads();
  };

  int q = q_lo;
  for (; q + 2 < q_hi; ++q) chunk(std::true_type{}, std::true_type{}, q);
  if (q + 1 < q_hi) {
    chunk(std::true_type{}, std::false_type{}, q);
    ++q;
  }
  chunk(std::false_type{}, std::false_type{}, q);

  // ---- epilogue: D[row = cout][col = pixel] -> NCHW, 128 B contiguous per (reg, half-wave) --
  // Every addend (bias, temb, residual) of a 16-value group is loaded before the group's first
  // store; the restrict copies state the no-alias contract of the ABI.
  const float *__restrict__ bias_p = a.bias;
  const float *__restrict__ chan_p = a.chan_add;
  const float *__restrict__ res_p = a.residual;
  const int co_base = nt * kConvNT + wco + 4 * lhi;
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    const int q = wpx + bb * 32 + l31;
    const int per_img = g.TH * a.Wo;
    const int ti = q / per_img;
    const int img = n0 + ti;
    if (q < g.TPX && img < g.NI) {
      const int p = h0 * a.Wo + (q - ti * per_img);  // flattened pixel inside the slice
      const int n = img / g.DS;
      const int dz = img - n * g.DS;
      size_t cstride = (size_t)g.Do * g.HWo;  // channel stride of the NC(D)HW output
      size_t obase = (((size_t)n * a.Cout + co_base) * g.Do + dz) * g.HWo + p;
      if (PARITY) {  // (d, h, w) of the low-res tile -> output voxel (2d + pz, 2h + py, 2w + px) of the 2x larger extent
        const int hl = p / a.Wo, wl = p - hl * a.Wo;
        const int dout = g.is3d ? 2 * dz + pz : 0;
        cstride = (size_t)g.Do * 4 * g.HWo;
        obase = (((size_t)n * a.Cout + co_base) * g.Do + dout) * (4 * (size_t)g.HWo) +
                (size_t)(2 * hl + py) * (2 * a.Wo) + 2 * wl + px;
      }
#pragma unroll
      for (int ab = 0; ab < NAB; ++ab) {
        float bvv[16], cv[16], rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = ab * 32 + (r & 3) + 8 * (r >> 2);
          bvv[r] = bias_p ? bias_p[co_base + dco] : 0.f;
          cv[r] = chan_p ? chan_p[(size_t)n * a.chan_add_stride + co_base + dco] : 0.f;
          rv[r] = res_p ? res_p[obase + (size_t)dco * cstride] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = ab * 32 + (r & 3) + 8 * (r >> 2);
          float v = acc[ab][bb][r];
          if (bias_p) v += bvv[r];
          if (chan_p) v += cv[r];
          if (res_p) v += rv[r];
          if (a.out_act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
          outp[obase + (size_t)dco * cstride] = v;
        }
      }
    }
  }
}

static int mfma_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int NTAPS, int NPOS, bool AFFINE, int MT, int KG = 1>
static int launch_variant(const ddpm_conv_desc &d, const ConvGeom &g_in, hipStream_t s) {
  ConvGeom g = g_in;
  g.nchunks_c = g.Cin / (kConvCc * KG);
  g.nchunks = g.nkd * g.nchunks_c;
  const size_t lds = (size_t)2 * (NTAPS * kConvCc * KG * kConvNT + kConvCc * KG * g.PS) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<NTAPS, NPOS, AFFINE, MT, KG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
    if (getenv("DDPM_CONV_DEBUG")) {
      int nb = -1;
      hipFuncAttributes fa{};
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, reinterpret_cast<const void *>(&conv_mfma_kernel<NTAPS, NPOS, AFFINE, MT, KG>), 256, lds);
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&conv_mfma_kernel<NTAPS, NPOS, AFFINE, MT, KG>));
      fprintf(stderr, "[conv_mfma<%d,%d,%d,%d>] lds=%zu B, numRegs=%d, occupancy=%d blocks/CU\n", NTAPS, NPOS,
              (int)AFFINE, MT, lds, fa.numRegs, nb);
    }
  }
  // split-K: see ConvGeom::ksplit.  Only plain layouts (no output parities, no output activation), whole float4 rows
  g.ksplit = 1;
  g.pstride = 0;
  ddpm_conv_desc dk = d;
  const size_t out_floats = (size_t)d.B * d.Cout * g.Do * g.HWo;
  const bool sk_on = sw().conv_splitk;  // DDPM_CONV_SPLITK=0: off (A/B, tests)
  if (sk_on && NTAPS != 4 && g.npar == 1 && d.scratch && d.out_act == DDPM_ACT_NONE && (g.Do * g.HWo) % 4 == 0) {
    const long wgs = (long)g.ntiles * (d.Cout / kConvNT);
    const int cus = mfma_cus();
    for (int sp = 8; sp >= 2; sp >>= 1)
      if (wgs * sp <= cus && g.nchunks % sp == 0 && g.nchunks / sp >= 8 && d.scratch_floats >= sp * out_floats) {
        g.ksplit = sp;
        break;
      }
    if (g.ksplit > 1) {
      g.pstride = (long long)out_floats;
      dk.out = d.scratch;
      dk.bias = nullptr;
      dk.chan_add = nullptr;
      dk.residual = nullptr;
    }
  }
  dim3 grid(g.ntiles, d.Cout / kConvNT, g.npar > 1 ? g.npar : g.ksplit);
  // algorithmic work of this launch: 2 * (output pixels) * Cout * Cin * taps FLOP, counted as the
  // op the launch replaces (a folded nearest-x2 upsample counts the 9 taps of the unfolded 3x3; a ConvTranspose
  // k4 s2 has 2 x 2 (x 2) live taps per output); bytes = input + output (+ residual) + weights, once each
  const bool transposed = d.mode == DDPM_CONV_TRANSPOSE2;
  const double out_px = (double)g.M * g.npar;
  const double taps = (NTAPS == 4 && !transposed) ? 9.0 : (double)NTAPS * g.nkd;
  const double flops = 2.0 * out_px * d.Cout * (double)g.Cin * taps;
  const double bytes = 4.0 * ((double)d.B * g.Cin * g.Di * g.HWi + out_px * d.Cout * (d.residual ? 2 : 1) +
                              (double)d.Cout * g.Cin * NTAPS * g.nkd * g.npar);
  const char *kname;
  if (g.is3d)
    kname = transposed ? "conv3d_transpose_k4s2" : NTAPS == 16 ? "conv3d_k4s2" : AFFINE ? "conv3d_k3_gn_silu" : "conv3d_k3";
  else if (transposed)
    kname = "conv2d_transpose_k4s2";
  else if (NTAPS == 16)
    kname = "conv2d_k4s2";
  else
    kname = NTAPS == 4 ? (MT == 128 ? "conv3x3_mfma_up_folded" : "conv3x3_mfma_up_folded_t64")
            : MT == 128 ? (NTAPS == 9 ? (AFFINE ? "conv3x3_mfma_gn_silu" : "conv3x3_mfma")
                                      : (AFFINE ? "conv1x1_mfma_gn" : "conv1x1_mfma"))
                        : (NTAPS == 9 ? (AFFINE ? "conv3x3_mfma_gn_silu_t64" : "conv3x3_mfma_t64")
                                      : (AFFINE ? "conv1x1_mfma_gn_t64" : "conv1x1_mfma_t64"));
  char kshape[160];
  if (g_prof_on && sw().prof_shapes) {  // development: one profile row per layer shape
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%dx%d m%d t%d", kname, d.C1, d.C2, d.Cout, g.Do, d.Ho, d.Wo, d.mode,
             MT);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  hipLaunchKernelGGL((conv_mfma_kernel<NTAPS, NPOS, AFFINE, MT, KG>), grid, dim3(256), lds, s, dk, g);
  DDPM_CHECK_LAUNCH();
  if (g.ksplit > 1) {
    ddpm_conv_desc dr = d;
    dr.stats_out = nullptr;  // (ddpm_conv_stats_parts is 0 for this kernel: the field is ignored)
    return launch_wino_split_reduce(dr, g.ksplit, g.pstride, g.Do * g.HWo, s);
  }
  return 0;
}

template <int MT>
static int launch_mt(const ddpm_conv_desc &d, const ConvGeom &g, hipStream_t s) {
  const int npos = (g.PS + 255) / 256;
  const bool aff = d.gscale != nullptr;
  if (d.ksize == 4) {  // k4 s2 p1 (VQ-VAE downsampling): 16 in-plane taps per depth tap
    if (npos == 1) return launch_variant<16, 1, false, MT>(d, g, s);
    if (npos == 2) return launch_variant<16, 2, false, MT>(d, g, s);
    return launch_variant<16, 3, false, MT>(d, g, s);
  }
  if (d.ksize == 3) {
    if (aff) {
      if (npos == 1) return launch_variant<9, 1, true, MT>(d, g, s);
      return launch_variant<9, 2, true, MT>(d, g, s);
    }
    if (npos == 1) return launch_variant<9, 1, false, MT>(d, g, s);
    if (npos == 2) return launch_variant<9, 2, false, MT>(d, g, s);
    return launch_variant<9, 3, false, MT>(d, g, s);
  }
  if (aff) return launch_variant<1, 1, true, MT>(d, g, s);
  // plain 1x1 / Linear: 16 channels per chunk when the channel counts allow it
  if (g.Cin % 16 == 0 && (d.C2 == 0 || d.C1 % 16 == 0)) return launch_variant<1, 1, false, MT, 4>(d, g, s);
  return launch_variant<1, 1, false, MT>(d, g, s);
}

// One launch, grid.z = output parity, 2 x 2 in-plane taps (x 2 depth taps in the chunk stream), 8 channels per chunk.
static int launch_parity(const ddpm_conv_desc &lr, const ConvGeom &g, hipStream_t s) {
  const int npos = (g.PS + 255) / 256;
  if (g.MT == 128)
    return npos == 1 ? launch_variant<4, 1, false, 128, 2>(lr, g, s) : launch_variant<4, 2, false, 128, 2>(lr, g, s);
  return npos == 1 ? launch_variant<4, 1, false, 64, 2>(lr, g, s) : launch_variant<4, 2, false, 64, 2>(lr, g, s);
}

// 3x3 conv over a nearest-x2 upsampled image as four 2x2-tap convs over the low-res image (one per output
// parity), weights pre-summed by fold_upsample_kernel: 16 instead of 36 multiply-adds per 4 outputs.
static int launch_upsample_folded(const ddpm_conv_desc &d, hipStream_t s, bool &taken) {
  taken = false;
  const int Cin = d.C1 + d.C2;
  if (!d.w_folded || d.gscale || d.dims == 3 || d.Di > 1 || d.Do > 1 || Cin % 8 || (d.C2 > 0 && d.C1 % 8)) return 0;
  ddpm_conv_desc lr = lowres_view(d);
  lr.w_packed = d.w_folded;
  ConvGeom g;
  if (!pick_geom(lr, g, 2 * 256) || g.PS > 2 * 256) return 0;
  taken = true;
  g.npar = 4;
  g.par_slab = (long long)d.Cout * Cin * 4;
  return launch_parity(lr, g, s);
}

// ConvTranspose k4 s2 p1 (VQ-VAE upsampling), 2-D or 3-D: out[2i + p] = sum over the two input positions
// i - 1 + p + r (r = 0, 1) with kernel index 3 - p - 2r, per axis -> one 2x2(x2)-tap conv per output parity.
static int launch_transpose(const ddpm_conv_desc &d, hipStream_t s) {
  ConvGeom g;
  if (!transpose_geom(d, g)) {
    set_error("conv_mfma: ConvTranspose k4 s2 p1 needs Cin %% 8 == 0, Cout %% 128 == 0 and Ho = 2 Hi, Wo = 2 Wi");
    return DDPM_EINVAL;
  }
  ddpm_conv_desc lr = lowres_view(d);
  lr.mode = DDPM_CONV_TRANSPOSE2;
  return launch_parity(lr, g, s);
}

// floats of scratch with which this descriptor's launch can be split over K (0: none): an upper bound of what
// launch_variant uses (the largest split that still fits the chip; the chunk count may force a smaller one)
size_t conv_mfma_scratch_floats(const ddpm_conv_desc &d) {
  if (d.mode == DDPM_CONV_TRANSPOSE2 || d.out_act != DDPM_ACT_NONE || !conv_mfma_supported(d)) return 0;
  ConvGeom g;
  if (!pick_geom(d, g) || (g.Do * g.HWo) % 4) return 0;
  const long wgs = (long)g.ntiles * (d.Cout / kConvNT);
  int sp = 8;
  while (sp >= 2 && wgs * sp > mfma_cus()) sp >>= 1;
  if (sp < 2) return 0;
  return (size_t)sp * d.B * d.Cout * g.Do * g.HWo;
}

int launch_conv_mfma(const ddpm_conv_desc &d, hipStream_t s) {
  if (d.mode == DDPM_CONV_TRANSPOSE2) return launch_transpose(d, s);
  if (d.mode == DDPM_CONV_UPSAMPLE2 && d.w_folded) {
    bool taken = false;
    const int rc = launch_upsample_folded(d, s, taken);
    if (taken || rc) return rc;
  }
  ConvGeom g;
  if (!conv_mfma_supported(d) || !pick_geom(d, g)) {
    set_error("conv_mfma: unsupported shape");
    return DDPM_EINVAL;
  }
  return g.MT == 128 ? launch_mt<128>(d, g, s) : launch_mt<64>(d, g, s);
}

// ---- weight packing: torch [Cout][Cin][T] -> [cout_tile][chunk][tap][4][128] -----------------
// `src_taps` / `tap_off` select T taps out of a wider torch kernel: the 9 (kh, kw) taps of depth tap kd
// of a [Cout][Cin][3][3][3] conv3d weight are src_taps = 27, tap_off = 9 * kd.
__global__ void pack_conv_weight_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin,
                                        int T, int cout_offset, int src_taps, int tap_off) {
  const int64_t total = (int64_t)Cout * Cin * T;
  const int nchunks = Cin / kConvCc;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int ci = (int)((i / T) % Cin);
    const int o = (int)(i / ((int64_t)T * Cin));
    const int og = cout_offset + o;
    const int tile = og / kConvNT, col = og % kConvNT;
    const int ch = ci / kConvCc, cl = ci % kConvCc;
    const size_t di = ((((size_t)tile * nchunks + ch) * T + t) * kConvCc + cl) * kConvNT + col;
    dst[di] = src[((size_t)o * Cin + ci) * src_taps + tap_off + t];
  }
}

// ---- folded upsample weights: torch [Cout][Cin][3][3] -> 4 x packed [cout_tile][chunk][2x2 tap][4][128] ----
// parity dy = 0: source rows {y-1: w[0], y: w[1] + w[2]};  dy = 1: {y: w[0] + w[1], y+1: w[2]}; same for columns
__global__ void fold_upsample_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin) {
  const int64_t total = (int64_t)Cout * Cin * 16;
  const int nchunks = Cin / kConvCc;
  const size_t slab = (size_t)Cout * Cin * 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), par = (int)((i >> 2) & 3);
    const int ci = (int)((i >> 4) % Cin), o = (int)((i >> 4) / Cin);
    const int dy = par >> 1, dx = par & 1, r = t >> 1, c = t & 1;
    const int kh0 = dy == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), kh1 = dy == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
    const int kw0 = dx == 0 ? (c == 0 ? 0 : 1) : (c == 0 ? 0 : 2), kw1 = dx == 0 ? (c == 0 ? 0 : 2) : (c == 0 ? 1 : 2);
    const float *w = src + ((size_t)o * Cin + ci) * 9;
    float acc = 0.f;
    for (int kh = kh0; kh <= kh1; ++kh)
      for (int kw = kw0; kw <= kw1; ++kw) acc += w[kh * 3 + kw];
    const int tile = o / kConvNT, col = o % kConvNT, ch = ci / kConvCc, cl = ci % kConvCc;
    dst[par * slab + ((((size_t)tile * nchunks + ch) * 4 + t) * kConvCc + cl) * kConvNT + col] = acc;
  }
}

size_t folded_upsample_weight_floats(int Cout, int Cin) {
  if (Cout % kConvNT || Cin % 8) return 0;
  return (size_t)16 * Cout * Cin;
}

int launch_fold_upsample_weight(const float *w_raw, float *w_folded, int Cout, int Cin, hipStream_t s) {
  DDPM_CHECK_ARG(folded_upsample_weight_floats(Cout, Cin) != 0, "fold: Cout %% 128 or Cin %% 8 != 0");
  const int64_t total = (int64_t)Cout * Cin * 16;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(fold_upsample_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_folded, Cout, Cin);
  DDPM_CHECK_LAUNCH();
  return 0;
}

size_t packed_conv_weight_floats(int Cout, int Cin, int ksize) {
  if (Cout % kConvNT || Cin % kConvCc || (ksize != 1 && ksize != 3 && ksize != 4)) return 0;
  return (size_t)Cout * Cin * ksize * ksize;
}

// ---- ConvTranspose k4 s2 p1 weights: torch [Cin][Cout][4][4]([4]) -> [parity][depth tap a][cout_tile][chunk][2x2 tap]
// [4][128]; tap (a, r, c) of parity (pz, py, px) is kernel element (3 - pz - 2a, 3 - py - 2r, 3 - px - 2c) ----------
__global__ void pack_convT_weight_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cin, int Cout,
                                         int dims) {
  const int npar = dims == 3 ? 8 : 4, nd = dims == 3 ? 2 : 1, K = dims == 3 ? 64 : 16;
  const int64_t total = (int64_t)Cout * Cin * npar * nd * 4;
  const int nchunks = Cin / kConvCc;
  const size_t slab = (size_t)Cout * Cin * 4, par_slab = slab * nd;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r_ = i;
    const int t = (int)(r_ & 3); r_ >>= 2;
    const int a = (int)(r_ % nd); r_ /= nd;
    const int par = (int)(r_ % npar); r_ /= npar;
    const int ci = (int)(r_ % Cin), o = (int)(r_ / Cin);
    const int px = par & 1, py = (par >> 1) & 1, pz = par >> 2;
    const int ky = 3 - py - 2 * (t >> 1), kx = 3 - px - 2 * (t & 1), kz = 3 - pz - 2 * a;
    const int kidx = dims == 3 ? (kz * 16 + ky * 4 + kx) : (ky * 4 + kx);
    const int tile = o / kConvNT, col = o % kConvNT, ch = ci / kConvCc, cl = ci % kConvCc;
    dst[par * par_slab + a * slab + ((((size_t)tile * nchunks + ch) * 4 + t) * kConvCc + cl) * kConvNT + col] =
        src[((size_t)ci * Cout + o) * K + kidx];
  }
}

size_t packed_convT_weight_floats(int Cout, int Cin, int dims) {
  if (Cout % kConvNT || Cin % 8 || (dims != 2 && dims != 3)) return 0;
  return (size_t)Cout * Cin * (dims == 3 ? 64 : 16);
}

int launch_pack_convT_weight(const float *w_raw, float *w_packed, int Cin, int Cout, int dims, hipStream_t s) {
  DDPM_CHECK_ARG(packed_convT_weight_floats(Cout, Cin, dims) != 0, "pack convT: Cout %% 128, Cin %% 8 or dims");
  const int64_t total = (int64_t)packed_convT_weight_floats(Cout, Cin, dims);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_convT_weight_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_packed, Cin, Cout, dims);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_pack_conv_weight(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize, int cout_offset,
                            int Cout_total, hipStream_t s, int src_taps, int tap_off) {
  if (src_taps <= 0) src_taps = ksize * ksize;
  DDPM_CHECK_ARG(packed_conv_weight_floats(Cout_total, Cin, ksize) != 0, "pack: Cout_total %% 128 or Cin %% 4 != 0");
  DDPM_CHECK_ARG(cout_offset >= 0 && cout_offset + Cout <= Cout_total, "pack: bad cout range");
  const int64_t total = (int64_t)Cout * Cin * ksize * ksize;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_packed, Cout, Cin,
                     ksize * ksize, cout_offset, src_taps, tap_off);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
