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
//   * f32-input MFMA is bit-exact fp32 FMA (k-ordered fmaf chain) at the fp32 vector peak
//     (157 TF) but needs 1 LDS dword per operand per 2048 FLOP instead of 1 per 2 FLOP, so LDS
//     and VGPR bandwidth stop being the limiter of a direct fp32 convolution.
//   * Workgroup tile: 128 output pixels (whole rows of one image, or several whole images when
//     H*W < 128) x 128 output channels, 4 waves as 2 (cout) x 2 (pixel), each wave 2x2 MFMA
//     tiles = 64 accumulator registers.  Input channels are consumed in chunks of 8:
//     the haloed input tile [8][rows+2][W+2] is staged ONCE per chunk and reused by all
//     9 taps; weights arrive pre-packed as [cout_tile][chunk][tap][8][128] so the chunk is one
//     contiguous 36 KB burst of coalesced 16-byte loads.
//   * Next chunk's global loads are issued into registers before the current chunk's 144 MFMAs
//     and written to LDS after them (async-stage split), so HBM/L2 latency hides under MFMA.
//   * GroupNorm arrives as per-(image, channel) scale/shift and is applied, with SiLU, while
//     staging; zero padding is applied AFTER the activation, as F.conv2d(pad=1) of the
//     activated tensor does.
#include <stdlib.h>

#include "common.h"

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned g_cu_ticket[2048];  // experiment knob 64 only

// native vector type: arrays of HIP's struct float4 stay in scratch (SROA does not split them)
typedef float v4f __attribute__((ext_vector_type(4)));

struct ConvGeom {
  int M;        // B * Ho * Wo
  int HWo, HWi;
  int TI, TH;   // images per tile, output rows per tile (per image)
  int IR, RS;   // LDS rows per image slot, LDS row stride
  int IRS;      // IR * RS
  int PS;       // LDS plane size (floats per channel)
  int pad;      // 1 for 3x3, 0 for 1x1
  int s;        // input step per output pixel (2 for stride-2)
  int Cin, nchunks;
  int xflags;   // experiment knobs (env DDPM_CONV_X): 1 = fast SiLU, 2 = skew co-resident workgroups
};

static bool make_geom(const ddpm_conv_desc &d, ConvGeom &g) {
  const int Cin = d.C1 + d.C2;
  g.Cin = Cin;
  g.HWo = d.Ho * d.Wo;
  g.HWi = d.Hi * d.Wi;
  g.M = d.B * g.HWo;
  g.pad = d.ksize == 3 ? 1 : 0;
  g.s = d.mode == DDPM_CONV_STRIDE2 ? 2 : 1;
  if (g.HWo >= kConvMT) {
    if (kConvMT % d.Wo) return false;
    g.TI = 1;
    g.TH = kConvMT / d.Wo;
    if (d.Ho % g.TH) return false;
  } else {
    if (kConvMT % g.HWo) return false;
    g.TI = kConvMT / g.HWo;
    g.TH = d.Ho;
  }
  const int IC = (g.s == 2) ? (2 * d.Wo + 1) : (d.Wo + 2 * g.pad);
  g.IR = (g.s == 2) ? (2 * g.TH + 1) : (g.TH + 2 * g.pad);
  g.RS = IC;
  g.IRS = g.IR * g.RS;
  g.PS = g.TI * g.IRS;
  g.nchunks = Cin / kConvCc;
  static const int xf = getenv("DDPM_CONV_X") ? atoi(getenv("DDPM_CONV_X")) : 0;
  g.xflags = xf;
  return true;
}

bool conv_mfma_supported(const ddpm_conv_desc &d) {
  if (!d.w_packed || d.force_direct) return false;
  const int Cin = d.C1 + d.C2;
  if (d.ksize != 1 && d.ksize != 3) return false;
  if (d.mode != DDPM_CONV_NORMAL && d.ksize != 3) return false;
  if (Cin % kConvCc || d.Cout % kConvNT) return false;
  if (d.C2 > 0 && (d.C1 % kConvCc)) return false;
  ConvGeom g;
  if (!make_geom(d, g)) return false;
  if (g.PS > 3 * 256) return false;
  if (d.gscale && g.PS > 2 * 256) return false;  // instantiated AFFINE variants: NPOS <= 2
  if (d.ksize == 1 && g.PS > 256) return false;
  return true;
}

template <int NTAPS, int NPOS, bool AFFINE>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ddpm_conv_desc a, const ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Wl = smem;                               // [NTAPS][8][128]
  float *Xl = smem + NTAPS * kConvCc * kConvNT;   // [8][PS]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int nt = blockIdx.y;
  const int P0 = blockIdx.x * kConvMT;

  if (g.xflags & 64) {
    // experiment: skew by a per-CU arrival ticket (robust to whatever the dispatcher does)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    const unsigned key = ((xcc & 7) << 8) | ((hw >> 8) & 0xff);       // xcc | se,sh,cu
    unsigned tk = 0;
    if (threadIdx.x == 0) tk = atomicAdd(&g_cu_ticket[key & 2047], 1u);
    unsigned *tks = reinterpret_cast<unsigned *>(smem);
    if (threadIdx.x == 0) tks[0] = tk;
    __syncthreads();
    const unsigned t0 = tks[0];
    __syncthreads();
    if (t0 & 1) __builtin_amdgcn_s_sleep(127);
  }
  if ((g.xflags & 14) && (((blockIdx.x + blockIdx.y * gridDim.x) >> 8) & 1)) {
    // experiment: offset every second 256-block "wave" of workgroups by ~half a chunk so that the two
    // workgroups sharing a CU do not stage (no MFMA issue) at the same time (s_sleep n = 64 n cycles)
    if (g.xflags & 2) __builtin_amdgcn_s_sleep(32);
    if (g.xflags & 4) __builtin_amdgcn_s_sleep(64);
    if (g.xflags & 8) __builtin_amdgcn_s_sleep(127);
  }
  // ---- tile origin ---------------------------------------------------------------------
  int n0, h0;
  if (g.TI == 1) {
    n0 = P0 / g.HWo;
    h0 = (P0 - n0 * g.HWo) / a.Wo;
  } else {
    n0 = P0 / g.HWo;
    h0 = 0;
  }

  // ---- per-thread staging positions (chunk-invariant) -------------------------------------
  int soff[NPOS];   // offset inside one channel plane of the source, -1 => zero
  int nimg[NPOS];
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const int r = tid + 256 * j;
    soff[j] = -1;
    nimg[j] = 0;
    if (r < g.PS) {
      const int ti = r / g.IRS;
      const int rr = r - ti * g.IRS;
      const int ir = rr / g.RS;
      const int ic = rr - ir * g.RS;
      const int n = n0 + ti;
      const int hv = g.s * h0 + ir - g.pad;
      const int wv = ic - g.pad;
      // bounds of the (virtual) conv input: the upsampled extent for UPSAMPLE2, else Hi x Wi
      const int Hv = (a.mode == DDPM_CONV_UPSAMPLE2) ? a.Ho : a.Hi;
      const int Wv = (a.mode == DDPM_CONV_UPSAMPLE2) ? a.Wo : a.Wi;
      if (n < a.B && hv >= 0 && hv < Hv && wv >= 0 && wv < Wv) {
        nimg[j] = n;
        soff[j] = (a.mode == DDPM_CONV_UPSAMPLE2) ? ((hv >> 1) * a.Wi + (wv >> 1)) : (hv * a.Wi + wv);
      }
    }
  }

  // ---- per-lane MFMA operand bases -----------------------------------------------------------
  int xb[2];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    const int q = (wm * 2 + bb) * 32 + l31;
    const int per_img = g.TH * a.Wo;
    const int ti = q / per_img;
    const int rem = q - ti * per_img;
    const int th = rem / a.Wo;
    const int tw = rem - th * a.Wo;
    xb[bb] = lhi * g.PS + ti * g.IRS + th * g.s * g.RS + tw * g.s;
  }
  const int wb = lhi * kConvNT + wn * 64 + l31;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prefetch registers ---------------------------------------------------------------------
  v4f wreg[NTAPS];
  float xreg[NPOS][kConvCc];
  v4f screg[NPOS][2], shreg[NPOS][2];

  const v4f *wsrc = reinterpret_cast<const v4f *>(a.w_packed) +
                       (size_t)nt * g.nchunks * (NTAPS * kConvCc * kConvNT / 4);

  // Software pipeline: iteration ch commits chunk ch (already in registers) to LDS, issues the
  // global loads of chunk ch + 1, then runs the MFMAs of chunk ch while those loads fly.
  // (Written inline, not as lambdas: by-reference captures kept the prefetch arrays in scratch.)
  for (int ch = -1; ch < g.nchunks; ++ch) {
    if (ch >= 0 && !((g.xflags & 16) && ch > 0)) {
      __syncthreads();  // everyone finished reading the previous chunk
      v4f *wl4 = reinterpret_cast<v4f *>(Wl);
#pragma unroll
      for (int i = 0; i < NTAPS; ++i) wl4[tid + 256 * i] = wreg[i];
#pragma unroll
      for (int j = 0; j < NPOS; ++j) {
        const int r = tid + 256 * j;
        if (r < g.PS) {
          const bool valid = soff[j] >= 0;
#pragma unroll
          for (int c = 0; c < kConvCc; ++c) {
            float v = xreg[j][c];
            if (AFFINE) {
              const float sc = screg[j][c >> 2][c & 3];
              const float sh = shreg[j][c >> 2][c & 3];
              v = v * sc + sh;
            }
            if (a.act == DDPM_ACT_SILU) v = (g.xflags & 1) ? silu_fast(v) : silu_f(v);
            Xl[c * g.PS + r] = valid ? v : 0.f;
          }
        }
      }
      __syncthreads();
    }

    if (ch + 1 < g.nchunks && !((g.xflags & 16) && ch >= 0)) {
      const int cn = ch + 1;
      const v4f *wp = wsrc + (size_t)cn * (NTAPS * kConvCc * kConvNT / 4);
#pragma unroll
      for (int i = 0; i < NTAPS; ++i) wreg[i] = wp[tid + 256 * i];
      const int cg0 = cn * kConvCc;
      const float *base;
      int Cs, cl0;
      if (cg0 < a.C1) {
        base = a.in1; Cs = a.C1; cl0 = cg0;
      } else {
        base = a.in2; Cs = a.C2; cl0 = cg0 - a.C1;
      }
#pragma unroll
      for (int j = 0; j < NPOS; ++j) {
        if (soff[j] >= 0) {
          const float *p = base + ((size_t)nimg[j] * Cs + cl0) * g.HWi + soff[j];
#pragma unroll
          for (int c = 0; c < kConvCc; ++c) xreg[j][c] = p[(size_t)c * g.HWi];
          if (AFFINE) {
            const v4f *sp = reinterpret_cast<const v4f *>(a.gscale + (size_t)nimg[j] * g.Cin + cg0);
            const v4f *hp = reinterpret_cast<const v4f *>(a.gshift + (size_t)nimg[j] * g.Cin + cg0);
            screg[j][0] = sp[0]; screg[j][1] = sp[1];
            shreg[j][0] = hp[0]; shreg[j][1] = hp[1];
          }
        } else {
#pragma unroll
          for (int c = 0; c < kConvCc; ++c) xreg[j][c] = 0.f;
        }
      }
    }

    if (ch >= 0 && !(g.xflags & 32)) {
      // Operand fetch is software-pipelined one k-step ahead of the MFMAs: hipcc otherwise issues each
      // step's ds_reads right in front of its four MFMAs and the matrix pipe idles for one LDS latency
      // (~70 of every 256 cycles with one computing wave per SIMD).  sched_barrier pins the order.
      constexpr int NSTEP = NTAPS * (kConvCc / 2);
      float a0[2], a1[2], b0[2], b1[2];
      auto fetch = [&](int st, int slot) {
        const int t = st / (kConvCc / 2), kk = st % (kConvCc / 2);
        const int tapoff = (NTAPS == 9) ? ((t / 3) * g.RS + (t % 3)) : 0;
        a0[slot] = Wl[wb + (t * kConvCc + 2 * kk) * kConvNT];
        a1[slot] = Wl[wb + (t * kConvCc + 2 * kk) * kConvNT + 32];
        b0[slot] = Xl[xb[0] + 2 * kk * g.PS + tapoff];
        b1[slot] = Xl[xb[1] + 2 * kk * g.PS + tapoff];
      };
      fetch(0, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int cur = st & 1;
        if (st + 1 < NSTEP) fetch(st + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b0[cur], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b1[cur], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b0[cur], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b1[cur], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: D[row = cout][col = pixel] -> NCHW, 128 B contiguous per (reg, half-wave) --
  // Two-phase per pixel block: every addend (bias, temb, residual) is loaded BEFORE the first store.
  // Interleaving "load residual -> add -> store" made each of the 64 stores wait for the previous one
  // (out may alias the addends as far as the compiler knows): ~85k cycles per workgroup, 22 % of a
  // 16-chunk convolution.  The restrict copies state the no-alias contract of the ABI.
  const float *__restrict__ bias_p = a.bias;
  const float *__restrict__ chan_p = a.chan_add;
  const float *__restrict__ res_p = a.residual;
  float *__restrict__ out_p = a.out;
  const int co_base = nt * kConvNT + wn * 64 + 4 * lhi;
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    const int P = P0 + (wm * 2 + bb) * 32 + l31;
    if (P < g.M && !((g.xflags & 256) && blockIdx.x != 0xfffff)) {  // knob 256: drop the epilogue (timing only)
      const int n = P / g.HWo;
      const int p = P - n * g.HWo;
      const size_t obase = ((size_t)n * a.Cout + co_base) * g.HWo + p;
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {  // 16 values at a time keeps the temporaries at 48 registers
        float bv[16], cv[16], rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = ab * 32 + (r & 3) + 8 * (r >> 2);
          bv[r] = bias_p ? bias_p[co_base + dco] : 0.f;
          cv[r] = chan_p ? chan_p[(size_t)n * a.chan_add_stride + co_base + dco] : 0.f;
          rv[r] = res_p ? res_p[obase + (size_t)dco * g.HWo] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dco = ab * 32 + (r & 3) + 8 * (r >> 2);
          float v = acc[ab][bb][r];
          if (bias_p) v += bv[r];
          if (chan_p) v += cv[r];
          if (res_p) v += rv[r];
          out_p[obase + (size_t)dco * g.HWo] = v;
        }
      }
    }
  }
}

template <int NTAPS, int NPOS, bool AFFINE>
static int launch_variant(const ddpm_conv_desc &d, const ConvGeom &g, hipStream_t s) {
  const size_t lds = (size_t)(NTAPS * kConvCc * kConvNT + kConvCc * g.PS) * sizeof(float) +
                     ((g.xflags & 128) ? 48 * 1024 : 0);  // knob 128: pad LDS so only one workgroup fits a CU
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<NTAPS, NPOS, AFFINE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
    if (getenv("DDPM_CONV_DEBUG")) {
      int nb = -1;
      hipFuncAttributes fa{};
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, reinterpret_cast<const void *>(&conv_mfma_kernel<NTAPS, NPOS, AFFINE>), 256, lds);
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&conv_mfma_kernel<NTAPS, NPOS, AFFINE>));
      fprintf(stderr, "[conv_mfma<%d,%d,%d>] lds=%zu B, numRegs=%d, static LDS=%zu, occupancy=%d blocks/CU\n", NTAPS,
              NPOS, (int)AFFINE, lds, fa.numRegs, fa.sharedSizeBytes, nb);
    }
  }
  dim3 grid((g.M + kConvMT - 1) / kConvMT, d.Cout / kConvNT);
  // algorithmic work of this launch: 2*M*Cout*Cin*taps FLOP; input + output (+ residual)
  // + weights bytes, each counted once
  const double flops = 2.0 * g.M * d.Cout * (double)g.Cin * NTAPS;
  const double bytes = 4.0 * ((double)d.B * g.Cin * g.HWi + (double)g.M * d.Cout * (d.residual ? 2 : 1) +
                              (double)d.Cout * g.Cin * NTAPS);
  const char *kname = NTAPS == 9 ? (AFFINE ? "conv3x3_mfma_gn_silu" : "conv3x3_mfma")
                                  : (AFFINE ? "conv1x1_mfma_gn" : "conv1x1_mfma");
  char kshape[160];
  if (g_prof_on && getenv("DDPM_PROF_SHAPES")) {  // development: one profile row per layer shape
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d m%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo, d.mode);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  hipLaunchKernelGGL((conv_mfma_kernel<NTAPS, NPOS, AFFINE>), grid, dim3(256), lds, s, d, g);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_conv_mfma(const ddpm_conv_desc &d, hipStream_t s) {
  ConvGeom g;
  if (!conv_mfma_supported(d) || !make_geom(d, g)) {
    set_error("conv_mfma: unsupported shape");
    return DDPM_EINVAL;
  }
  const int npos = (g.PS + 255) / 256;
  const bool aff = d.gscale != nullptr;
  if (d.ksize == 3) {
    if (aff) {
      if (npos == 1) return launch_variant<9, 1, true>(d, g, s);
      return launch_variant<9, 2, true>(d, g, s);
    }
    if (npos == 1) return launch_variant<9, 1, false>(d, g, s);
    if (npos == 2) return launch_variant<9, 2, false>(d, g, s);
    return launch_variant<9, 3, false>(d, g, s);
  }
  if (aff) return launch_variant<1, 1, true>(d, g, s);
  return launch_variant<1, 1, false>(d, g, s);
}

// ---- weight packing: torch [Cout][Cin][T] -> [cout_tile][chunk][tap][8][128] -----------------
__global__ void pack_conv_weight_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin,
                                        int T, int cout_offset) {
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
    dst[di] = src[i];
  }
}

size_t packed_conv_weight_floats(int Cout, int Cin, int ksize) {
  if (Cout % kConvNT || Cin % kConvCc || (ksize != 1 && ksize != 3)) return 0;
  return (size_t)Cout * Cin * ksize * ksize;
}

int launch_pack_conv_weight(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize, int cout_offset,
                            int Cout_total, hipStream_t s) {
  DDPM_CHECK_ARG(packed_conv_weight_floats(Cout_total, Cin, ksize) != 0, "pack: Cout_total %% 128 or Cin %% 8 != 0");
  DDPM_CHECK_ARG(cout_offset >= 0 && cout_offset + Cout <= Cout_total, "pack: bad cout range");
  const int64_t total = (int64_t)Cout * Cin * ksize * ksize;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_packed, Cout, Cin,
                     ksize * ksize, cout_offset);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
