// conv_wino44h.hip -- host side of the split-f16 Winograd F(4x4, 3x3) convolution: item geometry, dispatch, weight packing.
// Round 3; the kernel that ran here until round 5 (LDS-fed U ring, three barriers per chunk) was retired in round 6: its
// register-fed successor conv_wino44r.hip is bit-identical to it (held to it over 72 cases for a round) and 14-19 % faster.
//
// The op (GroupNorm-affine + SiLU prologue, virtual concat, bias / temb / residual epilogue; reference call site
// /root/reference/src/trainers/reconstruct.py:151-153, layer list /root/reference/src/trainers/base.py:66-86) and the work
// item (64 output channels x 32 tiles x all input channels, 8 waves = 2 cout blocks x 4 position groups, 9 accumulator tiles per
// wave): every M_xi[cout][tile] = sum_c U_xi[cout][c] V_xi[c][tile] runs on v_mfma_f32_32x32x16_f16.
//
// Arithmetic.  Transforms, activation, accumulation and the output transform stay fp32.  Each operand of a product is
// carried as TWO f16 numbers, x = xh + xl with xh = f16(x) and xl = f16(x - xh) (the difference is exact in fp32), and the
// MFMA's K = 16 holds 8 input channels x {hi, lo}:   A[k = 8 p + c] = U_p[c],   B[k] = V_h[c]  (first MFMA),  V_l[c] (second)
// so that two MFMAs per 8 channels add (Uh + Ul) Vh + (Uh + Ul) Vl = all four partial products, each exact in the fp32
// accumulator (11 x 11 significant bits).  What is lost is only the representation error of x ~ xh + xl: <= 2^-22 |x|
// as long as xl is a normal f16.  To keep it normal the operands are pre-scaled by powers of two (exact): U by 2^su when it
// is packed -- su chosen per layer so that max |2^su U| lies in [2^14, 2^15), i.e. 17 binades of full precision below the
// layer's largest transformed weight --, V by 2^3 through the activation that feeds the transform; the product of the two
// scales is stored behind the packed planes and goes back into the epilogue's fused multiply-add.  Full precision holds for
// |V| in [2^-6, 8188]; below, the absolute error of an operand is <= 2^-28 (V) / 2^-39 of the layer's largest |U|; above, the
// high half of V overflows to inf.  Behind a GroupNorm the input is bounded and 2^3 centres its range; the forms that read an
// UN-NORMALISED tensor (Upsample: the raw residual stream; the VQ-VAE's 3-D residual units: ReLU activations) use 2^0 instead
// -- full precision for |V| in [2^-3, 65504], i.e. patches up to ~650 in the worst case (B^T d B has gain <= 100) and ~5 000
// for typical data, graceful (absolute error <= 2^-25) below.  An overflow is never silent: the inf / NaN reaches the
// status word of the PLMS / clamp kernels (ddpm_status_read) and the caller re-runs the affected images on the fp32-MFMA kernels.
#include "wino44h_common.h"

namespace ddpm {

static int w44h_cus() {
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

bool w44h_geom(const ddpm_conv_desc &d, W44HGeom &g, bool sizing) {
  const int Cin = d.C1 + d.C2;
  const bool is3d = d.dims == 3;
  const bool up = d.mode == DDPM_CONV_UPSAMPLE2;
  if (d.ksize != 3 || (!is3d && (d.Di > 1 || d.Do > 1)) || (d.mode != DDPM_CONV_NORMAL && !up)) return false;
  // Upsample convolutions (F.interpolate(nearest, x2) + conv3x3 of generative's Upsample, between the up levels): plain
  // convolution of the virtual upsampled image; 64-pixel staging units must be an even number of rows (Wo <= 32)
  if (up) {
    if (!sw().up_wino44h || is3d || d.gscale || d.act != DDPM_ACT_NONE || d.C2 || d.chan_add || d.residual || d.out_act != DDPM_ACT_NONE)
      return false;
    if (d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi || d.Wo > 32) return false;
  }
  if ((d.out_act != DDPM_ACT_NONE && !(is3d && d.out_act == DDPM_ACT_RELU)) || d.act == DDPM_ACT_RELU) return false;
  if (d.gscale && d.act != DDPM_ACT_SILU) return false;  // the affine variant has SiLU built in
  // 3-D: no GroupNorm / activation prologue (zero padding along the depth must stay zero), no concat, no temb
  if (is3d && (d.gscale || d.act != DDPM_ACT_NONE || d.C2 || d.chan_add)) return false;
  const int Dd = is3d ? (d.Di > 1 ? d.Di : 1) : 1;
  if (is3d && (d.Do > 1 ? d.Do : 1) != Dd) return false;
  if (Cin % 16 || (d.C2 > 0 && d.C1 % 4) || d.Cout % kK) return false;  // an even number of 8-channel chunks, 4-channel halves
  if ((d.Ho & 3) || (d.Wo & 3) || (!up && (d.Hi != d.Ho || d.Wi != d.Wo))) return false;
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
  g.NCHc = Cin / kC;
  g.kd0 = is3d && Dd == 1 ? 1 : 0;
  g.nkd = is3d && Dd > 1 ? 3 : 1;
  // ddpm_conv_desc.depth_taps (ABI 10): a 3x3x3 weight whose first or last depth tap is all zeros (the parity convolutions a
  // ConvTranspose k4 s2 decomposes into, vqvae.py) walks two taps instead of three
  if (is3d && Dd > 1 && (d.depth_taps == 3 || d.depth_taps == 6)) {
    g.kd0 = d.depth_taps == 6 ? 1 : 0;
    g.nkd = 2;
  }
  g.nkd_w = is3d ? 3 : 1;
  g.NCH = g.nkd * g.NCHc;
  g.HW = d.Ho * d.Wo;
  g.up = up;
  g.HWin = up ? d.Hi * d.Wi : g.HW;
  g.CS = Dd * g.HW;
  g.prow = 4 * g.TR + 2;
  // the pixel-tile layout the kernel reads (conv_wino44r.hip: w44r_relayout) -- sized HERE, so that conv_wino44h_supported()
  // and the launch agree on whether the item fits the 160 KB of LDS (the launch used to check it on its own)
  w44r_relayout(d, g);
  if (w44h_lds_bytes(g) > 160 * 1024) return false;
  if (64 % d.Wo) return false;  // a staging unit is 64 pixels = whole rows (of the image the convolution sees)
  const int rows = g.prow < d.Ho ? g.prow : d.Ho;
  g.UI = (rows * d.Wo + 63) / 64;
  g.NRT = g.TI * g.UI;
  // kernel variants: one image per item with 9 or 10 units (32x32 / 64x64 images); whole images of 4 units or of 1 unit
  if (g.TI == 1) {
    if (g.NRT != 9 && g.NRT != 10) return false;
    if ((rows - 1) * d.Wo < 64 * (g.NRT - 1)) return false;  // only the last round can reach past the item's rows
  } else if (!((g.UI == 4 && rows * d.Wo == 256 && g.TI == 2) || (g.UI == 1 && rows * d.Wo == 64 && g.TI == 8))) {
    return false;
  }
  g.KT = d.Cout / kK;
  g.NIT = (g.NIMG + g.TI - 1) / g.TI;
  const long items = (long)g.KT * g.parts * g.NIT;
  const int cus = w44h_cus();
  const bool any_size = sw().conv_wino44 == 2;  // any launch size (tests)
  g.S = 1;
  g.pstride = 0;
  if (items < cus && !any_size) {
    if (is3d) return false;
    const int sp_max = sw().wino44_split;
    for (int sp = 4; sp >= 2; sp >>= 1)  // every workgroup of a split walks an even number of chunks
      if (sp <= sp_max && items * sp <= cus && g.NCH % (2 * sp) == 0) { g.S = sp; break; }
    if (g.S == 1 || items * g.S * 4 < (long)cus * 3) return false;  // below three quarters of the chip: conv_wino.hip
    const size_t out_floats = (size_t)d.B * d.Cout * g.HW;
    if (!sizing && (!d.scratch || d.scratch_floats < g.S * out_floats)) return false;
    g.pstride = (long long)out_floats;
  }
  g.IPW = (int)((items + cus - 1) / cus);
  g.NS = g.parts * ((g.NIT + g.IPW - 1) / g.IPW) * g.S;
  // one workgroup per CU: a grid a little over the chip (6 cout tiles x 48 slots = 288 on 256 CUs: the 384-channel input gradients
  // of the training step under DDPM_TRAIN_DGRAD=wino44h; no layer of the reconstruction path has 6 cout tiles) would run a second
  // round for 32 workgroups -- more items per workgroup until it is one round (as conv_wino.hip)
  while (g.S == 1 && items > cus && (long)g.KT * 8 <= cus && (long)g.KT * ((g.NS + 7) / 8) * 8 > cus && g.IPW < g.NIT) {
    ++g.IPW;
    g.NS = g.parts * ((g.NIT + g.IPW - 1) / g.IPW);
  }
  g.grid = g.KT * ((g.NS + 7) / 8) * 8;
  g.xitem = sw().w44h_xitem;
  g.rev = 0;
  // default 0: the cout tiles of a slot are neighbours on ONE XCD, so that the KT re-reads of the slot's input hit that XCD's L2
  // (rocprofv3 FETCH_SIZE / WRITE_SIZE at B = 1 024: 1.14 GB per launch against 1.49 GB with one cout tile per XCD, same time)
  g.xmap = (sw().wino44_xmap >= 0 ? sw().wino44_xmap != 0 : 0) && (8 % g.KT == 0);
  if (g.xmap) g.grid = 8 * ((g.NS + 8 / g.KT - 1) / (8 / g.KT));
  return true;
}

static bool w44h_enabled() {
  return sw().conv_wino44 != 0 && split_f16_on(sw().wino44_f16x3);
}

// slices per (image, cout) of the GroupNorm statistics the epilogue writes to desc.stats_out (0: none -- 3-D, split
// launches, tile counts per image that are not a power of two)
int conv_wino44h_stats_parts(const ddpm_conv_desc &d) {
  W44HGeom g;
  if (!w44h_enabled() || !d.w_wino44h || d.force_direct || d.dims == 3 || !w44h_geom(d, g, true)) return 0;
  const int per = g.TR * g.TWc;
  if (g.S != 1) return wino_split_reduce_stats_parts(g.HW);  // a channel-split launch: the reduce pass writes them
  if ((per != 4 && per != 16 && per != 32) || g.parts > 8) return 0;
  return g.parts;
}

bool conv_wino44h_supported(const ddpm_conv_desc &d) {
  W44HGeom g;
  return w44h_enabled() && d.w_wino44h != nullptr && !d.force_direct && w44h_geom(d, g);
}

size_t conv_wino44h_scratch_floats(const ddpm_conv_desc &d) {
  W44HGeom g;
  if (!w44h_enabled() || !d.w_wino44h || d.force_direct || !w44h_geom(d, g, true) || g.S == 1) return 0;
  return (size_t)g.S * d.B * d.Cout * g.HW;
}

int launch_conv_wino44h(const ddpm_conv_desc &d, hipStream_t s) {
  W44HGeom g;
  if (!d.w_wino44h || !w44h_geom(d, g)) {
    set_error("conv_wino44h: unsupported shape");
    return DDPM_EINVAL;
  }
  ddpm_conv_desc dk = d;
  // the ABI's promise: stats_out is ignored whenever ddpm_conv_stats_parts() is 0 for this descriptor (a split launch's
  // statistics come from its reduce pass, which is handed `d`, not `dk`)
  if (g.S > 1 || conv_wino44h_stats_parts(d) == 0) dk.stats_out = nullptr;
  if (g.S > 1) {  // partial sums go to the scratch slabs, the addends to the reduce pass
    dk.out = d.scratch;
    dk.bias = nullptr;
    dk.chan_add = nullptr;
    dk.residual = nullptr;
  }
  const double M = (double)g.NIMG * g.HW;
  // algorithmic work = the direct convolution's: 2 M Cout Cin 9 (x 3 depth taps)
  const double flops = 2.0 * M * d.Cout * (double)g.Cin * 9 * g.nkd;
  const double bytes = 4.0 * ((g.up ? 0.25 : 1.0) * M * g.Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * g.Cin * 9 * g.nkd);
  const char *kname = d.dims == 3 ? "conv3d_wino44h" : g.up ? "conv3x3_wino44h_up" : d.gscale ? "conv3x3_wino44h_gn_silu" : "conv3x3_wino44h";
  char kshape[160];
  if (g_prof_on && sw().prof_shapes) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  // serpentine item order across consecutive launches (DDPM_W44R_SERP=1; measured +-0, default off): results do not depend on it
  static unsigned launch_parity = 0;
  g.rev = sw().w44r_serp ? (int)(launch_parity++ & 1) : 0;
  if (const int rc = launch_conv_wino44r(dk, g, w44h_lds_bytes(g), s)) return rc;
  if (g.S > 1) return launch_wino_split_reduce(d, g.S, g.pstride, g.HW, s);
  return 0;
}

// ---- weights: torch [Cout][Cin][3][3] -> U = 2^su G g G^T (6 x 6) as f16 hi / lo planes in the order the kernel's LDS-DMA
// lands them:   [cout tile 64][chunk of 8 channels][phase 3][position 12][plane 2][cout 64][channel 8]   (f16)
// phase t holds transform rows (0, 5), (1, 2), (3, 4); position s = 6 (second row of the pair) + column.  Behind the planes:
// two floats, max |G g G^T| of the layer and 1 / (2^3 2^su) for the kernel's epilogue (su = 15 - exponent of the maximum).
__device__ __forceinline__ void wino44h_u(const float *w, double (&u)[6][6]) {
  const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},   {0, 0, 1}};
  double t[6][3];
  for (int r = 0; r < 6; ++r)
    for (int q = 0; q < 3; ++q) t[r][q] = G[r][0] * w[0 * 3 + q] + G[r][1] * w[1 * 3 + q] + G[r][2] * w[2 * 3 + q];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) u[r][c] = t[r][0] * G[c][0] + t[r][1] * G[c][1] + t[r][2] * G[c][2];
}

// (a 3x3x3 weight, nkd = 3, is transformed per depth tap: total counts (cout, cin, kd) triples, 9 floats each -- torch's
// [Cout][Cin][kd][3][3] order)
__global__ void wino44h_max_kernel(const float *__restrict__ src, unsigned *__restrict__ tail, int64_t total) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    double u[6][6];
    wino44h_u(src + i * 9, u);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) m = fmaxf(m, fabsf((float)u[r][c]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

__global__ void wino44h_pack_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, int Cout, int Cin, int nkd) {
  const int64_t total = (int64_t)Cout * Cin * nkd;
  const int nch = Cin / kC;
  float *tail = reinterpret_cast<float *>(dst + (size_t)kX * Cout * Cin * 2 * nkd);
  int e = 0;
  const float umax = tail[0];
  if (umax > 0.f) (void)frexpf(umax, &e);  // umax = f 2^e, f in [0.5, 1)
  const int su = umax > 0.f ? 15 - e : 0;  // max |2^su U| in [2^14, 2^15)
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[1] = ldexpf(1.f / kVScale, -su);
  const int trio_of[6] = {0, 1, 1, 2, 2, 0}, second_of[6] = {0, 0, 1, 0, 1, 1};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kd = (int)(i % nkd);
    const int ci = (int)((i / nkd) % Cin), o = (int)(i / ((int64_t)nkd * Cin));
    double uu[6][6];
    wino44h_u(src + i * 9, uu);
    const int tile = o / kK, k64 = o % kK, ch = ci / kC, c8 = ci % kC;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        const float u = ldexpf((float)uu[r][c], su);
        const _Float16 hi = (_Float16)u, lo = (_Float16)(u - (float)hi);
        const int s = second_of[r] * 6 + c;
        const size_t base = (((((size_t)tile * nkd + kd) * nch + ch) * 3 + trio_of[r]) * kPP + s) * 2;
        dst[((base + 0) * kK + k64) * kC + c8] = __builtin_bit_cast(uint16_t, hi);
        dst[((base + 1) * kK + k64) * kC + c8] = __builtin_bit_cast(uint16_t, lo);
      }
  }
}

size_t wino44h_weight_halves(int Cout, int Cin) {
  if (Cout % kK || Cin % 16) return 0;
  return (size_t)kX * Cout * Cin * 2 + kTail;
}

int launch_pack_wino44h_weight(const float *w_raw, uint16_t *w_wino44h, int Cout, int Cin, hipStream_t s, int nkd) {
  DDPM_CHECK_ARG(wino44h_weight_halves(Cout, Cin) != 0 && (nkd == 1 || nkd == 3), "wino44h pack: Cout %% 64 or Cin %% 16 != 0");
  const int64_t total = (int64_t)Cout * Cin * nkd;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  unsigned *tail = reinterpret_cast<unsigned *>(w_wino44h + (size_t)kX * Cout * Cin * 2 * nkd);
  hipError_t e = hipMemsetAsync(tail, 0, kTail * sizeof(uint16_t), s);
  if (e != hipSuccess) {
    set_error("wino44h pack: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(wino44h_max_kernel, dim3(blocks), dim3(256), 0, s, w_raw, tail, total);
  DDPM_CHECK_LAUNCH();
  hipLaunchKernelGGL(wino44h_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_wino44h, Cout, Cin, nkd);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
