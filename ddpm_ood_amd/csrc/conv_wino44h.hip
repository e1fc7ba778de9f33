// conv_wino44h.hip -- 3x3 stride-1 convolution as Winograd F(4x4, 3x3) with the position GEMMs on the f16 MFMA pipe
// (split-f16 products, fp32 accumulate).  Round 3.
//
// Same fused op as conv_wino44.hip (GroupNorm-affine + SiLU prologue, virtual concat, bias / temb / residual epilogue;
// reference call site /root/reference/src/trainers/reconstruct.py:151-153, layer list /root/reference/src/trainers/
// base.py:66-86) and the same work item (64 output channels x 32 tiles x all input channels, 8 waves = 2 cout blocks x
// 4 position groups, 9 accumulator tiles per wave), but every M_xi[cout][tile] = sum_c U_xi[cout][c] V_xi[c][tile] runs
// on v_mfma_f32_32x32x16_f16, which multiplies at 16x the rate of the fp32 MFMA the old kernel is bound by.
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
// status word of the PLMS / clamp kernels (ddpm_status_read) and the caller re-runs the batch on the fp32-MFMA kernels.
//
// Structure (why it is not the old kernel with other MFMAs).  K = 16 per instruction means 8 channels x 36 positions x
// (64 + 32) operand rows = 110 KB per K-step: the operand images cannot be double-buffered per chunk any more.  So a chunk
// of 8 channels is walked in THREE PHASES of 12 positions -- the transform rows (0, 5), (1, 2), (3, 4), which share their
// inputs -- and V is produced just in time:
//   waves 0..3  PRODUCERS: lane = (tile, channel pair); a task = the 12 positions of one phase for two channels (pair-packed
//               f16 stores), done in two halves over two phases; the two wave pairs alternate, so that phase m + 1's V
//               slot is written during phase m and a TWO-slot V ring suffices.  They also issue the LDS-DMA of the U slot
//               (24 KB per phase, packed in LDS order in global memory) one phase ahead.
//   waves 4..7  PIXEL waves: wave = one channel of a 4-channel half-chunk; loads two phases ahead of use, GroupNorm affine +
//               SiLU (x 2^3) once per pixel into a zero-bordered pixel tile (ring of four 4-channel half-tiles).
//   all 8 waves: 3 jobs x 2 MFMAs per phase (job = one position x one cout block: A, B_h, B_l by ds_read_b128).
// Waves w and w + 4 share a SIMD: one producer and one pixel wave each.  The two roles run separate instruction streams
// (same barrier sequence), so that neither pays for the other's registers.
// Per item the pipeline is filled and drained; the epilogue (output transform through four LDS exchange slabs, as
// conv_wino44.hip) then owns the operand rings -- but not the pixel ring: a workgroup's first item fills in six MFMA-free phases,
// every later one in two, because the pixel waves stage the next item's chunks 0 and 1 during the current item's last chunks
// (round 4, DDPM_W44H_XITEM).
//
// LDS: U ring 2 x 24 KB + V ring 2 x 12 KB (= the four 18 KB exchange slabs of the epilogue) + pixel ring.
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
  g.nkd_w = is3d ? 3 : 1;
  g.NCH = g.nkd * g.NCHc;
  g.HW = d.Ho * d.Wo;
  g.up = up;
  g.HWin = up ? d.Hi * d.Wi : g.HW;
  g.CS = Dd * g.HW;
  g.prow = 4 * g.TR + 2;
  auto layout = [&](bool pad) {
    g.PW = d.Wo + 2;
    if (pad && g.TWc < 16) g.PW += ((g.TWc - g.PW) % 16 + 16) % 16;
    g.IS = g.prow * g.PW;
    if (pad && g.TI > 1 && per_img < 16) g.IS += ((4 * per_img - g.IS) % 64 + 64) % 64;
    g.PCH = g.TI * g.IS;
    g.PCH += ((1 - g.PCH) % 4 + 4) % 4;
    g.HS = 4 * g.PCH + 1 + 64;  // + 64 dump floats for out-of-image lanes
    return w44h_lds_bytes(g) <= 160 * 1024;
  };
  if (!layout(true) && !layout(false)) return false;
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

// NRT = staging rounds of a pixel wave per half-chunk; UIT = 0: one image per item, else units per image (4 or 1)
// D3: the 3-D form (images = (n, d) slices, chunk stream = (depth tap, channel chunk)); a template parameter so that the 2-D
// instantiations carry none of its address arithmetic
template <bool AFFINE, int NRT, int UIT, bool RES, bool D3 = false, bool UP = false>
__global__ __launch_bounds__(512, 2) void conv_wino44h_kernel(const ddpm_conv_desc a, const W44HGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool ONEIMG = UIT == 0;
  constexpr int NGS = ONEIMG ? 1 : NRT / UIT;        // images per item = GroupNorm scale / shift pairs per half-chunk
  constexpr int GD = ONEIMG ? NRT : UIT;             // consecutive rounds that belong to one image
  constexpr int NVM = NRT;                           // vector-memory loads of one pixel stage
  float *const P = smem + kRINGF;                    // pixel ring: 4 half-tiles of [4 channels][PCH] + 1 + 64 dump floats
  char *const smb = reinterpret_cast<char *>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb = wave & 1, pg = wave >> 1;  // MFMA role: cout block, position group (positions 3 pg .. 3 pg + 2 of a phase)
  const bool silu = a.act == DDPM_ACT_SILU;

  // ---- this workgroup's items (as conv_wino44.hip)
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  int kt = wj % g.KT, slot = (wj / g.KT) * 8 + xcd;
  if (g.xmap) {
    kt = xcd % g.KT;
    slot = wj * (8 / g.KT) + xcd / g.KT;
  }
  if (slot >= g.NS) return;
  const int split = slot % g.S;
  slot /= g.S;
  const int part = slot % g.parts, it0 = (slot / g.parts) * g.IPW;
  const int nitems = min(g.IPW, g.NIT - it0);
  const int r0 = part * g.TR;
  const int n_first = it0 * g.TI, n_end = n_first + nitems * g.TI;
  const int NCHs = g.NCH / g.S, ch_lo = split * NCHs;  // this workgroup's chunk range (even length)
  const int NPH = 3 * NCHs;                            // MFMA phases per item
  float *const outp = a.out + (size_t)split * g.pstride;

  // ---- MFMA operand addresses (bytes): A = U slot [pos][plane = lhi][cout][8 ch], B = V slot [pos][plane][tile][8 ch]
  const int ua = (3 * pg * 2 + lhi) * (kK * 16) + (cb * 32 + l31) * 16;
  const int va = kVB0 + 3 * pg * (2 * kT * 16) + l31 * 16;
  f32x16 acc8;  // tiles 0..7: a[0:127] by name (see mfma_pin)
  reserve_agprs();

  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(a.w_wino44h), 0, (int)((size_t)kX * a.Cout * g.Cin * g.nkd_w * 4), 0x00020000);
  // U slot index of this workgroup's phase 0 (slots: [cout tile][depth tap][channel chunk][phase])
  const int ukt = ((kt * g.nkd_w + g.kd0) * g.NCHc + ch_lo) * 3;

#ifdef W44H_PROBE  // timing experiment: cycle stamps of workgroup 0, waves 0 / 2 / 4, block of phases 12..17 -> desc.scratch
  int probe_m = -100;
#define W44H_STAMP(i)                                                                                                   \
  if (blockIdx.x == 0 && (wave == 0 || wave == 2 || wave == 4) && probe_m >= 12 && probe_m < 18 && lane == 0 && a.scratch) \
    reinterpret_cast<unsigned long long *>(a.scratch)[((wave >> 1) * 6 + (probe_m - 12)) * 8 + (i)] = __builtin_readcyclecounter();
#else
#define W44H_STAMP(i)
#endif
  // One phase of MFMA work: 3 jobs (positions 3 pg + i of the phase) x 2 MFMAs; `slice(k)`, k = 0..5, is the role's staging
  // work pinned between them.  us / vs: ring slots (bytes) of the phase.
  auto mfma_phase = [&](auto tc, int us, int vs, auto &&slice) {
    constexpr int t = decltype(tc)::value;
    const int ua_s = ua + us, va_s = va + vs;
#ifndef W44H_PREFETCH2
    // all nine operand reads of the phase up front: with one job of read-ahead the LDS latency under load (250 .. 300
    // cycles) WAS the job time (probe: 280 cycles per job of two 32-cycle MFMAs)
    h8 A[3], Bh[3], Bl[3];
    W44H_STAMP(0)
#ifdef W44H_NO_MFMA
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      slice(k);
      __builtin_amdgcn_sched_barrier(0);
    }
    (void)A; (void)Bh; (void)Bl; (void)ua_s; (void)va_s; (void)t;
    return;
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      A[i] = lds_b128(ua_s, 2 * i * kK * 16);
      Bh[i] = lds_b128(va_s, 2 * i * kT * 16);
      Bl[i] = lds_b128(va_s, (2 * i + 1) * kT * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int x = 3 * t + i;  // accumulator tile
      // job i's three reads are the oldest outstanding: the later jobs' may stay in flight (plus whatever the slices issue)
      if (x == 8) {  // (the last job of a phase)
        mfma_v_pair_wait0(acc8, A[i], Bh[i], Bl[i]);
        slice(2 * i);
        __builtin_amdgcn_sched_barrier(0);
        slice(2 * i + 1);
        __builtin_amdgcn_sched_barrier(0);
        W44H_STAMP(1 + i)
        continue;
      }
      if (i == 0) mfma_pin_wait<6>(x, A[i], Bh[i]);
      else if (i == 1) mfma_pin_wait<3>(x, A[i], Bh[i]);
      else mfma_pin_wait<0>(x, A[i], Bh[i]);
      slice(2 * i);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pin(x, A[i], Bl[i]);
      slice(2 * i + 1);
      __builtin_amdgcn_sched_barrier(0);
      W44H_STAMP(1 + i)
    }
#else  // one job of read-ahead (two operand register sets)
    h8 A[2], Bh[2], Bl[2];
    A[0] = lds_b128(ua_s, 0);
    Bh[0] = lds_b128(va_s, 0);
    Bl[0] = lds_b128(va_s, kT * 16);
    A[1] = lds_b128(ua_s, 2 * kK * 16);
    Bh[1] = lds_b128(va_s, 2 * kT * 16);
    Bl[1] = lds_b128(va_s, 3 * kT * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int x = 3 * t + i;
      if (x == 8) {
        mfma_v_pair_wait0(acc8, A[i & 1], Bh[i & 1], Bl[i & 1]);
        slice(2 * i);
        __builtin_amdgcn_sched_barrier(0);
        slice(2 * i + 1);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (i < 2) mfma_pin_wait<3>(x, A[i & 1], Bh[i & 1]);
      else mfma_pin_wait<0>(x, A[i & 1], Bh[i & 1]);
      slice(2 * i);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pin(x, A[i & 1], Bl[i & 1]);
      if (i == 0) {
        A[0] = lds_b128(ua_s, 4 * kK * 16);
        Bh[0] = lds_b128(va_s, 4 * kT * 16);
        Bl[0] = lds_b128(va_s, 5 * kT * 16);
      }
      slice(2 * i + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  };
  auto phase_end = [&]() {  // pixel waves: the U slot they fetched has landed (and with it their older pixel loads)
#ifdef W44H_PROBE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    W44H_STAMP(4)
    asm volatile("s_barrier" ::: "memory");
    W44H_STAMP(5)
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_end_keep = [&](auto nc) {  // pixel waves: all but their N newest vector-memory operations have completed
    constexpr int N = decltype(nc)::value;
#ifdef W44H_PROBE
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    W44H_STAMP(4)
    asm volatile("s_barrier" ::: "memory");
    W44H_STAMP(5)
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_end_lds = [&]() {  // producers: no vector memory traffic
#ifdef W44H_PROBE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W44H_STAMP(4)
    asm volatile("s_barrier" ::: "memory");
    W44H_STAMP(5)
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
  };

  auto zero_accumulators = [&]() {
    zero_pinned_tiles();
    // (a literal zero vector is materialised THROUGH a0..a15 by hipcc: an opaque zero keeps it in arch VGPRs)
    float z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    acc8 = f32x16{z, z, z, z, z, z, z, z, z, z, z, z, z, z, z, z};
  };

  // ---- zero borders once (pixel writes only ever touch in-image pixels)
  for (int i = tid; i < 4 * g.HS; i += 512) P[i] = 0.f;
  __syncthreads();

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ================================================================================================ PRODUCER waves
  // lane = (tile of 16, channel pair j of 4); wave & 1 = tile half; (wave >> 1) = group: group G runs the first half (channel
  // 2 j) of the task of phase m + 2 when m + G is even and the second half (channel 2 j + 1, pack, store) of phase m + 1 else.
  auto producer_item = [&](auto grpc, bool first_item) {
    constexpr int GRP = decltype(grpc)::value;  // waves 0, 1: group 0; waves 2, 3: group 1
    const int st = (wave & 1) * 16 + (lane & 15), j = lane >> 4;
    int tb0;  // pixel-ring offset of this lane's patch origin in channel 2 j (half-chunk j >> 1, plane 2 (j & 1))
    {
      const int per = g.TR * g.TWc;
      const int ti = st / per, rem = st - ti * per;
      const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
      tb0 = (j >> 1) * g.HS + 2 * (j & 1) * g.PCH + ti * g.IS + 4 * tr * g.PW + 4 * tc;
    }
    int vw0 = kVB0 + st * 16 + 4 * j;  // V store: + slot + (2 pos + plane) 512
    const int ulane = lane * 16;
    // The U slot of phase m + 1 (24 transfers of 1 KB; an LDS-DMA issue costs 100 .. 150 cycles) is fetched by the waves with
    // slack: 16 transfers by the four pixel waves, 8 by the two producer waves that run the lighter FIRST half this phase
    auto dma_u = [&](int e, int mm, int us) {
      const int piece = 16 + (wave & 1) * 4 + e;
      // (the guard-free fill / tail phases ask for slots -5 .. -1 and NPH: clamped into the item's own range -- the scalar
      // offset of a raw buffer load is not part of its range check, so it must never point outside the packed planes)
      const int mc = min(max(mm, 0), NPH - 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (__attribute__((address_space(3))) void *)(smb + us + piece * 1024), 16,
                                               ulane, (ukt + mc) * kUSB + piece * 1024, 0, 0);
    };
    // column-pass results of the task's two channels (carried from its first half to its second) and the row being read
    float cA[2][6], cB[2][6], drow[6];

    // the transform of one channel of the patch: rows (0, 5), (1, 2) or (3, 4) of B^T d B (conv_wino44.hip's row pairs)
    //   rows (0, 5): A = d4 - 5 d2 + 4 d0,  B = d5 - 5 d3 + 4 d1
    //   rows (1, 2): p = d4 - 2 d2 - 2 d2,  q = d3 - 2 d1 - 2 d1,  A = p + q, B = p - q
    //   rows (3, 4): p = d4 - d2/2 - d2/2,  q = d3 - d1/2 - d1/2,  A = p + 2 q, B = p - 2 q
    auto rd = [&](int r, int pb) {
      const float *p = P + pb + r * g.PW;
#pragma unroll
      for (int q = 0; q < 6; ++q) drow[q] = p[q];
    };
    // column pass of channel c (0 / 1) in seven steps; every step consumes the row the previous one requested
    auto cstep = [&](auto tc, int c, int s, int pb) {
      constexpr int t = decltype(tc)::value;
      constexpr bool t0 = t == 0;
      constexpr float c1 = t0 ? -5.f : t == 1 ? -2.f : -0.5f, c2 = t0 ? 4.f : c1, bm = t0 ? 0.f : t == 1 ? 1.f : 2.f;
      float(&wA)[6] = cA[c], (&wB)[6] = cB[c];
      if (s == 0) {
        rd(4, pb);
      } else if (s == 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) wA[q] = drow[q];
        rd(2, pb);
      } else if (s == 2) {
#pragma unroll
        for (int q = 0; q < 6; ++q) wA[q] = __builtin_fmaf(c1, drow[q], wA[q]);
        rd(t0 ? 0 : 2, pb);
      } else if (s == 3) {
#pragma unroll
        for (int q = 0; q < 6; ++q) wA[q] = __builtin_fmaf(c2, drow[q], wA[q]);
        rd(t0 ? 5 : 3, pb);
      } else if (s == 4) {
#pragma unroll
        for (int q = 0; q < 6; ++q) wB[q] = drow[q];
        rd(t0 ? 3 : 1, pb);
      } else if (s == 5) {
#pragma unroll
        for (int q = 0; q < 6; ++q) wB[q] = __builtin_fmaf(c1, drow[q], wB[q]);
        rd(1, pb);
      } else if (s == 6) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const float qq = __builtin_fmaf(c2, drow[q], wB[q]), pp = wA[q];
          wA[q] = __builtin_fmaf(bm, qq, pp);
          wB[q] = t0 ? qq : __builtin_fmaf(-bm, qq, pp);
        }
      }
    };
    // A task = the 12 positions of one phase for the lane's two channels, in two halves of equal weight:
    //   first half   the column passes of both channels (2 x 6 row reads, 2 x 48 VALU)          -> cA, cB
    //   second half  four row passes (4 x 12), pair-split (4 per pair) and the 24 stores of the V slot
    auto first_half = [&](auto tc, int k, int pb) {
      const int pb1 = pb + g.PCH;  // channel 2 j + 1: the next plane of the half-tile
      if (k == 0) cstep(tc, 0, 0, pb);
      if (k == 1) { cstep(tc, 0, 1, pb); cstep(tc, 0, 2, pb); cstep(tc, 0, 3, pb); }
      if (k == 2) { cstep(tc, 0, 4, pb); cstep(tc, 0, 5, pb); cstep(tc, 0, 6, pb); cstep(tc, 1, 0, pb1); }
      if (k == 3) { cstep(tc, 1, 1, pb1); cstep(tc, 1, 2, pb1); cstep(tc, 1, 3, pb1); }
      if (k == 4) { cstep(tc, 1, 4, pb1); cstep(tc, 1, 5, pb1); }
      if (k == 5) cstep(tc, 1, 6, pb1);
    };
    float t0r[6], t1r[6];  // a transformed row of both channels (second half)
    uint32_t hi6[6], lo6[6];
    auto second_half = [&](int k, int vs) {
      // ONE address register + immediates (left to itself hipcc materialises a VGPR address per store and spills them:
      // ring offset + position offset exceed the 16-bit offset field when folded into one constant)
      int vwa = vw0 + vs;
      asm volatile("" : "+v"(vwa));
      // a row: six independent 4-instruction splits (ILP), then its twelve stores; the last slice of a phase has no LDS
      // traffic, so that the closing lgkmcnt(0) finds the queue almost drained
      auto split_row = [&]() {
#pragma unroll
        for (int q = 0; q < 6; ++q) split_pair(t0r[q], t1r[q], hi6[q], lo6[q]);
      };
      auto store_row = [&](int o) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(vwa), "v"(hi6[q]), "n"((2 * (o + q)) * (kT * 16)) : "memory");
          asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(vwa), "v"(lo6[q]), "n"((2 * (o + q) + 1) * (kT * 16)) : "memory");
        }
      };
      // (closing the FIRST half with the first row pass + split instead -- hi6 / lo6 carried in place of cA -- measured 4-8 %
      // slower: the first-half phases also carry the wave's LDS-DMA issues and the twelve patch-row reads)
      if (k == 0) { bt6(cA[0], t0r); bt6(cA[1], t1r); split_row(); }
      if (k == 1) { store_row(0); bt6(cB[0], t0r); }
      if (k == 2) { bt6(cB[1], t1r); split_row(); }
      if (k == 3) store_row(6);
    };
    // Phases m = -6 .. NPH - 1 in blocks of six; Q = m mod 6 is a compile-time constant of each instance, and with it the
    // trio, the ring slots and whether this group runs a first or a second half: no run-time branch inside a phase.  There
    // are no range guards either: the tasks and DMAs of phases before 0 / after NPH - 1 run on whatever the rings hold (buffer
    // addressing keeps the DMA in range, their results are overwritten before anyone reads them, everything has landed when
    // the last phase's barrier opens), and the MFMAs of the six fill phases accumulate garbage that is zeroed afterwards.
    auto body = [&](auto qc, int m) {
      constexpr int Q = decltype(qc)::value;
#ifdef W44H_PROBE
      probe_m = m;
#endif
      asm volatile("" : "+v"(tb0), "+v"(vw0));  // keep the per-lane bases out of LICM's reach (see conv_wino44.hip)
      constexpr int us = (Q & 1) * kUSB, vs = (Q & 1) * kVSB;                // ring slots of phase m
      constexpr int us1 = ((Q + 1) & 1) * kUSB, vs1 = ((Q + 1) & 1) * kVSB;  // ring slots of phase m + 1
      constexpr int Q2 = (Q + 2) % 6;
      // pixel half-tiles of the chunk a task belongs to: even chunks in ring slots 0, 1, odd chunks in 2, 3
      constexpr int pb2 = Q2 >= 3 ? 2 : 0;
      constexpr bool FIRST = GRP == (Q & 1);  // first half of the task of phase m + 2 (reads the pixel ring), else second half
                                              // of the task of phase m + 1 (registers -> V slot)
      const int pb = tb0 + pb2 * g.HS;
      auto slice = [&](int k) {
#ifndef W44H_NO_DMA  // (timing experiments: -DW44H_NO_DMA / _NO_PROD / _NO_PIXEL / _NO_MFMA build wrong-result variants)
        if (FIRST && k == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dma_u(e, m + 1, us1);
        }
#endif
#ifndef W44H_NO_PROD
        if (FIRST) first_half(std::integral_constant<int, Q2 % 3>{}, k, pb);
        else second_half(k, vs1);
#endif
      };
      mfma_phase(std::integral_constant<int, Q % 3>{}, us, vs, slice);
      if (FIRST) phase_end();
      else phase_end_lds();
    };
    for (int m = -6; m < NPH; m += 6) {
      if (m == 0) zero_accumulators();
      if (m >= 0 || first_item) {  // (a later item's fill is its last two phases: see the item loop)
        body(std::integral_constant<int, 0>{}, m);
        body(std::integral_constant<int, 1>{}, m + 1);
        body(std::integral_constant<int, 2>{}, m + 2);
        body(std::integral_constant<int, 3>{}, m + 3);
      }
      body(std::integral_constant<int, 4>{}, m + 4);
      body(std::integral_constant<int, 5>{}, m + 5);
    }
  };

  // ================================================================================================ PIXEL waves
  // wave - 4 = channel of a 4-channel half-chunk; round k = 64 pixels (unit k of the item: image k / GD, pixels 64 (k % GD) ..).
  //   phase m = 3 c + 0: loads (set 0) of half 0 of chunk c + 2;  activation (set 1) of half 1 of chunk c + 1
  //   phase m = 3 c + 1: loads (set 1) of half 1 of chunk c + 2
  //   phase m = 3 c + 2: activation (set 0) of half 0 of chunk c + 2
  // i.e. every load has two phases to land, and a half-tile is rewritten one phase after its last reader (the second half of
  // the task of the last phase of chunk c - 2) has passed its barrier.
  auto pixel_item = [&](int n_cur, bool first_item) {
    const int sc = wave - 4;
    const bool has_next = g.xitem && n_cur + g.TI < n_end;  // the item's last two chunks of staging fetch the NEXT item's chunks 0, 1
    const int row_lo = max(0, 4 * r0 - 1), row_hi = min(a.Ho, 4 * (r0 + g.TR) + 1);
    const int npx = (row_hi - row_lo) * a.Wo;
    const int dump = 4 * g.PCH + 1 + lane;  // relative to the half-tile
    int pix0, pw0, pixL = 0, pwL = 0;
    {
      const bool valid = lane < npx;
      // byte offset of pixel (row, col) of the image the convolution sees inside a stored channel plane; UP: nearest x2 of the
      // low-res plane -- a round of 64 pixels is an even number of rows, so rounds stay a constant stride apart (64 bytes)
      auto src_of = [&](int row, int col) { return UP ? ((row >> 1) * (a.Wo >> 1) + (col >> 1)) * 4 : (row * a.Wo + col) * 4; };
      pix0 = valid ? src_of(row_lo + lane / a.Wo, lane % a.Wo) : (int)0x80000000;  // out of range: the load returns 0
      pw0 = valid ? sc * g.PCH + (row_lo + lane / a.Wo - (4 * r0 - 1)) * g.PW + lane % a.Wo + 1 : dump;
      if (ONEIMG) {
        const int eL = lane + 64 * (NRT - 1);
        const bool vL = eL < npx;
        pixL = vL ? src_of(row_lo + eL / a.Wo, eL % a.Wo) : (int)0x80000000;
        pwL = vL ? sc * g.PCH + (row_lo + eL / a.Wo - (4 * r0 - 1)) * g.PW + eL % a.Wo + 1 : dump;
      }
    }
    const int prs = (64 / a.Wo) * g.PW;  // pixel-tile floats between a lane's pixels of consecutive rounds of one image
    constexpr int kRoundBytes = UP ? 64 : 256;
    auto pix_of = [&](int k) { return ONEIMG ? (k == NRT - 1 ? pixL : pix0 + kRoundBytes * k) : pix0 + kRoundBytes * (k % GD); };
    auto pw_of = [&](int k) { return ONEIMG ? (k == NRT - 1 ? pwL : pw0 + k * prs) : pw0 + (k % GD) * prs + (k / GD) * g.IS; };
    const int bytes1 = a.B * a.C1 * (D3 ? g.CS : UP ? g.HWin : g.HW) * 4, bytes2 = a.B * a.C2 * g.HW * 4;
    const __amdgpu_buffer_rsrc_t rs_sc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
    int vzero;  // keeps the wave-uniform scale / shift loads on the vector memory path (in-order vmcnt with the pixel loads)
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // ONE set of GroupNorm scale / shift pairs (eight images per item would otherwise hold 32 registers and push a value into a
    // pinned AGPR): fetched in the phase before the activation that uses them, after the previous user has finished
    float praw[2][NRT], gs[NGS], gh[NGS];
    const int ulane = lane * 16;
    // transfer e (0..3) of this wave's share of the U slot of phase mm (see the producers' dma_u)
    auto dma_u = [&](int e, int mm, int us) {
      const int mc = min(max(mm, 0), NPH - 1);  // (see the producers' dma_u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_u, (__attribute__((address_space(3))) void *)(smb + us + (sc + 4 * e) * 1024), 16, ulane,
          (ukt + mc) * kUSB + (sc + 4 * e) * 1024, 0, 0);
    };

    auto load_stage = [&](auto setc, int c, int half) {
      constexpr int S = decltype(setc)::value;
      // past the item's last chunk: the next item's first chunks (the stream of half-tiles runs on across the output transform,
      // which does not touch the pixel ring); behind the last item a harmless repeat (uniform vmcnt bookkeeping)
      const bool nxt = c >= NCHs && has_next;
      const int cl = nxt ? c - NCHs : min(max(c, 0), NCHs - 1);
      const int n_it = nxt ? n_cur + g.TI : n_cur;
      int cg = (ch_lo + cl) * kC + half * 4 + sc;
      int soff3 = 0;
      bool dok = true;  // D3: the depth tap's slice lies inside the volume
      if (D3) {  // stream chunk -> (depth tap, channel chunk); image -> (batch item, slice); one image per item
        const int kdi = cl / g.NCHc;
        cg = (cl - kdi * g.NCHc) * kC + half * 4 + sc;
        const int ni = min(n_it, g.NIMG - 1);
        const int nb = ni / g.D, dsl = ni - nb * g.D + g.kd0 + kdi - 1;
        dok = dsl >= 0 && dsl < g.D;
        soff3 = ((nb * a.C1 + cg) * g.D + (dok ? dsl : 0)) * g.HW * 4;
      }
      const bool first = cg < a.C1;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2, 0x00020000);
      const int cx = first ? a.C1 : a.C2, cgl = first ? cg : cg - a.C1;
#pragma unroll
      for (int k = 0; k < NRT; ++k) {
        const int ni = min(n_it + (ONEIMG ? 0 : k / GD), g.NIMG - 1);
        // D3: a depth tap outside the volume reads zeros -- the range check of a raw buffer load is on the VGPR offset, and
        // 0x80000000 is past every resource (as for the out-of-image lanes in pix0)
        const int soff = D3 ? soff3 : (ni * cx + cgl) * (UP ? g.HWin : g.HW) * 4;
        const int voff = D3 && !dok ? (int)0x80000000 : pix_of(k);
        praw[S][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
      }
    };
    auto load_affine = [&](int c, int half) {
      if (!AFFINE) return;
      const bool nxt = c >= NCHs && has_next;
      const int cl = nxt ? c - NCHs : min(max(c, 0), NCHs - 1);
      const int cg = (ch_lo + cl) * kC + half * 4 + sc;
#pragma unroll
      for (int i = 0; i < NGS; ++i) {
        const int ni = min((nxt ? n_cur + g.TI : n_cur) + i, g.NIMG - 1);
        const int goff = (ni * g.Cin + cg) * 4;
        gs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, vzero, goff, 0));
        gh[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, vzero, goff, 0));
      }
    };
    auto activate = [&](auto setc, int k, int ring) {  // pixel value x 2^3: the transform's output is the pre-scaled V
      constexpr int S = decltype(setc)::value;
      const float x = praw[S][k];
      float y;
      if (AFFINE) {
        const float sa = gs[k / GD], sb = gh[k / GD];
        const float v = __builtin_fmaf(x, sa, sb);
        const float t = __builtin_fmaf(x, -1.44269504088896341f * sa, -1.44269504088896341f * sb);
        y = (kVScale * v) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
      } else {
        y = kVScaleRaw * (silu ? silu_fast(x) : x);
      }
      P[ring * g.HS + pw_of(k)] = y;
    };
    auto body = [&](auto qc, int c, bool mini = false) {  // c = chunk of the phase (floor(m / 3); -2, -1 in the fill phases)
      constexpr int Q = decltype(qc)::value;
      const int m = 3 * c + Q % 3;
#ifdef W44H_PROBE
      probe_m = m;
#endif
      constexpr int R = Q % 3;
      constexpr int us1 = ((Q + 1) & 1) * kUSB;  // U slot of phase m + 1
      asm volatile("" : "+v"(pix0), "+v"(pw0), "+v"(pixL), "+v"(pwL));
      constexpr int us = (Q & 1) * kUSB, vs = (Q & 1) * kVSB;
      // ring slot of the half-tile written in this phase: even chunks 0 / 1, odd chunks 2 / 3 (a block starts on an even chunk)
      constexpr int ringB = Q == 0 ? 3 : 1;  // R == 0: half 1 of chunk c + 1
      constexpr int ringA = Q == 2 ? 0 : 2;  // R == 2: half 0 of chunk c + 2
      // (no range guards: a half-tile activated for a chunk outside the item is overwritten before anyone reads it)
      auto slice = [&](int k) {
        if (k == 0) {  // the U slot of the next phase first (waited for at the end of this phase), then the pixel loads, which
                       // stay in flight across the barrier: they are consumed two phases later
#ifndef W44H_NO_DMA
#pragma unroll
          for (int e = 0; e < 4; ++e) dma_u(e, m + 1, us1);
#endif
#ifndef W44H_NO_PIXEL
          if (R == 0) load_stage(I0{}, c + 2, 0);
          if (R == 1) { load_stage(I1{}, c + 2, 1); load_affine(c + 2, 0); }  // scale / shift of the half activated next phase
#endif
        }
#ifdef W44H_NO_PIXEL
        return;
#endif
        if (k >= 1 && R != 1) {
          // (the set being activated landed with an earlier phase's closing wait; in flight now: this phase's DMA and loads)
#pragma unroll
          for (int kk = 3 * (k - 1); kk < 3 * k && kk < NRT; ++kk) {  // three rounds per slice: done by slice 4
            if (R == 0) activate(I1{}, kk, ringB);
            if (R == 2 && !mini) activate(I0{}, kk, ringA);  // (mini: that half-tile was written during the previous item)
          }
          if (R == 2 && k == 5) load_affine(c + 2, 1);  // for the next phase's activation (half 1 of the same chunk)
        }
      };
      mfma_phase(std::integral_constant<int, Q % 3>{}, us, vs, slice);
      // the DMAs (issued first) have landed; the pixel loads of this phase -- NVM, + 2 NGS scale / shift loads in R == 1 --
      // stay in flight, those of earlier phases have landed too (in-order retirement): every load gets two phases
      if (R == 0) phase_end_keep(std::integral_constant<int, NVM>{});
      else if (R == 1) phase_end_keep(std::integral_constant<int, NVM + (AFFINE ? 2 * NGS : 0)>{});
      else phase_end_keep(std::integral_constant<int, (AFFINE ? 2 * NGS : 0)>{});  // R == 2: the scale / shift loads of its last slice
    };
    int c = -2;
    for (int m = -6; m < NPH; m += 6, c += 2) {
      if (m == 0) zero_accumulators();
      const bool mini = m < 0 && !first_item;
      if (!mini) {
        body(std::integral_constant<int, 0>{}, c);
        body(std::integral_constant<int, 1>{}, c);
        body(std::integral_constant<int, 2>{}, c);
        body(std::integral_constant<int, 3>{}, c + 1);
      }
      // a later item's fill: chunk 0 and half 0 of chunk 1 are in the pixel ring already (staged during the previous item's
      // last two chunks); what is left is the half in flight at the boundary -- loads of half 1 of chunk 1 (phase -2), its
      // scale / shift pairs (phase -1) -- and the producers' task halves + U slots of phases 0 and 1
      body(std::integral_constant<int, 4>{}, c + 1);
      body(std::integral_constant<int, 5>{}, c + 1, mini);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the repeats past the last chunk: nothing may land after the item
  };

#ifdef W44H_PRIO  // experiment: the producers are the critical path of every phase; the pixel wave of their SIMD yields to them
  if (wave < 4) asm volatile("s_setprio 1");
#endif
  for (int n_cur = n_first; n_cur < n_end; n_cur += g.TI) {
    // Items after the first skip four of the six fill phases: the pixel waves' stream of half-tiles (loads, GroupNorm + SiLU,
    // pixel ring) does not stop at an item's last chunk but goes on into the next item's chunks 0 and 1, and the pixel ring
    // survives the output transform (which owns only the operand rings).  (Round 4: -4 phases of ~0.8 us per item.)
    const bool first_item = n_cur == n_first || !g.xitem;
    if (wave < 2) producer_item(I0{}, first_item);
    else if (wave < 4) producer_item(I1{}, first_item);
    else pixel_item(n_cur, first_item);

    // ---- end of an item: Y = A^T M A through four exchange slabs [xi][cout block][lane] (the operand rings: every stage of
    // the item has finished).  Pass q moves accumulator registers 4 q .. 4 q + 3 of all 36 positions; wave (cb, pg) then
    // finishes register 4 q + pg of cout block cb: cout = 32 cb + 8 q + 4 lhi + pg, tile = l31 (as conv_wino44.hip).
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' passes
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int el31 = elane & 31, elhi = elane >> 5;
    const int per = g.TR * g.TWc;
    const int ti = el31 / per, rem = el31 - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    const int n = n_cur + ti, ncl = min(n, g.NIMG - 1);
    const int nbat = D3 ? ncl / g.D : ncl, dsl_o = D3 ? ncl - nbat * g.D : 0;  // (batch item, slice)
    const int cstr = D3 ? g.CS : g.HW;                                        // channel stride
    float *const XS = smem;
    // 1 / (2^3 2^su): the operands' power-of-two pre-scales, written behind the planes by the pack kernel.  Loaded here, per
    // item, through the scalar cache: one more live VGPR across the phase loops and hipcc parks a value in a0 (a pinned tile)
    float kOutScale, unused_umax;
    sload2(reinterpret_cast<const float *>(a.w_wino44h + (size_t)kX * a.Cout * g.Cin * 2 * g.nkd_w) + 1, kOutScale, unused_umax);
    (void)unused_umax;
    if (!AFFINE) kOutScale *= kVScale / kVScaleRaw;  // the packed tail carries 1 / (2^3 2^su)
    // accumulator tile 3 t + i of wave pg holds position s = 3 pg + i of phase t: row (0,5 | 1,2 | 3,4)[s / 6], column s % 6
    const int rsel = pg >> 1, cofs = 3 * (pg & 1);
    const int xb0 = (rsel ? 5 : 0) * 6 + cofs, xb1 = (rsel ? 2 : 1) * 6 + cofs, xb2 = (rsel ? 4 : 3) * 6 + cofs;
    float addv[4];
    if (ONEIMG) {
      // pass q, lanes 0..31: cout co0 + 8 q, lanes 32..63: co0 + 8 q + 4 -- floats 8 q and 8 q + 4 behind co0
      const int co0 = kt * kK + cb * 32 + pg;
      float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ts[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float *const tp = a.chan_add ? a.chan_add + (size_t)min(n_cur, g.NIMG - 1) * a.chan_add_stride + co0 : nullptr;
      if (a.bias && tp) sload8x2(a.bias + co0, tp, bs, ts);
      else if (a.bias) sload8(a.bias + co0, bs);
      else if (tp) sload8(tp, ts);
#pragma unroll
      for (int q = 0; q < 4; ++q) addv[q] = elhi ? bs[2 * q + 1] + ts[2 * q + 1] : bs[2 * q] + ts[2 * q];
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
    // GroupNorm statistics of the tensor this launch produces (desc.stats_out): per (image, cout, part) the mean and the
    // sum of squared deviations M2 of the item's pixels of that channel, merged pairwise (Chan) in a fixed order
    const bool emit_stats = !D3 && a.stats_out != nullptr && g.S == 1;
    auto pass = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const size_t obase = obase0 + (size_t)(8 * q) * cstr;
      float st_p = 0.f, st_s1 = 0.f, st_s2 = 0.f;  // this lane's 4x4 tile about a pivot (its first value)
      {
        float *xw = XS + cb * 64 + elane;
#pragma unroll
        for (int x = 0; x < 9; ++x) {
          const int xi = (x < 3 ? xb0 : x < 6 ? xb1 : xb2) + x % 3;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            xw[rr * kXS + xi * 128] = x == 8 ? acc8[4 * q + rr] : read_pinned(16 * (x & 7) + 4 * q + rr);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float *xr = XS + pg * kXS + cb * 64 + elane;  // + xi * 128
      const float ad = addv[q];
      auto half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        float w[2][6];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {  // columns of M through two rows of A^T
          const float m1 = xr[(1 * 6 + jj) * 128], m2 = xr[(2 * 6 + jj) * 128], m3 = xr[(3 * 6 + jj) * 128],
                      m4 = xr[(4 * 6 + jj) * 128];
          if (h == 0) {
            const float m0 = xr[(0 * 6 + jj) * 128];
            w[0][jj] = (m0 + (m1 + m2)) + (m3 + m4);
            w[1][jj] = __builtin_fmaf(2.f, m3 - m4, m1 - m2);
          } else {
            const float m5 = xr[(5 * 6 + jj) * 128];
            w[0][jj] = __builtin_fmaf(4.f, m3 + m4, m1 + m2);
            w[1][jj] = __builtin_fmaf(8.f, m3 - m4, m1 - m2) + m5;
          }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int k = 2 * h + kk;
          float y[4];
          at4(w[kk][0], w[kk][1], w[kk][2], w[kk][3], w[kk][4], w[kk][5], y);
          // the operands' power-of-two pre-scales come off here (exact), in the same fma that adds bias + temb
          v4f o = v4f{__builtin_fmaf(y[0], kOutScale, ad), __builtin_fmaf(y[1], kOutScale, ad),
                      __builtin_fmaf(y[2], kOutScale, ad), __builtin_fmaf(y[3], kOutScale, ad)};
          if (RES) o += res[k];
          if (D3 && a.out_act == DDPM_ACT_RELU) o = v4f{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
          if (n < g.NIMG) *reinterpret_cast<v4f *>(outp + obase + (size_t)k * a.Wo) = o;
          if (emit_stats) {
            if (k == 0) st_p = o[0];
            const v4f dd = o - st_p;
            st_s1 += (dd[0] + dd[1]) + (dd[2] + dd[3]);
            st_s2 += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
          }
        }
      };
      half(I0{});
      v4f r01[2];
      if (RES && q < 3) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          r01[k] = *reinterpret_cast<const v4f *>(a.residual + obase + (size_t)8 * cstr + (size_t)k * a.Wo);
      }
      half(I1{});
      if (RES && q < 3) {
        res[0] = r01[0];
        res[1] = r01[1];
#pragma unroll
        for (int k = 2; k < 4; ++k)
          res[k] = *reinterpret_cast<const v4f *>(a.residual + obase + (size_t)8 * cstr + (size_t)k * a.Wo);
      }
      if (emit_stats) {
        // lane: {mean, M2} of its tile; the item's tiles of one image are `per` consecutive lanes (4, 16 or 32)
        float mean = st_p + st_s1 * (1.f / 16.f);
        float m2 = fmaxf(st_s2 - st_s1 * st_s1 * (1.f / 16.f), 0.f);
        group_moments_last_lane(mean, m2, 16.f, per);
        if (rem == per - 1 && n < g.NIMG) {
          const size_t co = (size_t)kt * kK + cb * 32 + 8 * q + 4 * elhi + pg;
          *reinterpret_cast<float2 *>(a.stats_out + (((size_t)n * a.Cout + co) * g.parts + part) * 2) = make_float2(mean, m2);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
#ifndef W44H_NO_EPI  // (timing experiment: no output transform / stores -- wrong results)
    pass(I0{});
    pass(I1{});
    pass(I2{});
    pass(std::integral_constant<int, 3>{});
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

int launch_conv_wino44h(const ddpm_conv_desc &d, hipStream_t s) {
  W44HGeom g;
  if (!d.w_wino44h || !w44h_geom(d, g)) {
    set_error("conv_wino44h: unsupported shape");
    return DDPM_EINVAL;
  }
  const size_t lds = w44h_lds_bytes(g);
  typedef void (*kern_t)(const ddpm_conv_desc, const W44HGeom);
  // shapes: one image per item with 9 (32x32) or 10 (64x64) staging units, two 16x16 images, eight 8x8 images
#define W44H_K(A, N, U) {conv_wino44h_kernel<A, N, U, false>, conv_wino44h_kernel<A, N, U, true>}
  static const kern_t kerns[2][4][2] = {
      {W44H_K(false, 9, 0), W44H_K(false, 10, 0), W44H_K(false, 8, 4), W44H_K(false, 8, 1)},
      {W44H_K(true, 9, 0), W44H_K(true, 10, 0), W44H_K(true, 8, 4), W44H_K(true, 8, 1)}};
#undef W44H_K
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 16; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 8][i / 2 % 4][i % 2]),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
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
  const int shape = g.TI == 1 ? (g.NRT == 9 ? 0 : 1) : g.UI == 4 ? 2 : 3;
  kern_t kern = kerns[d.gscale ? 1 : 0][shape][dk.residual ? 1 : 0];
  if (g.up) {  // no prologue, no residual (w44h_geom)
    static const kern_t kerns_up[4] = {conv_wino44h_kernel<false, 9, 0, false, false, true>, conv_wino44h_kernel<false, 10, 0, false, false, true>,
                                       conv_wino44h_kernel<false, 8, 4, false, false, true>, conv_wino44h_kernel<false, 8, 1, false, false, true>};
    static bool attr_up_done = false;
    if (!attr_up_done) {
      for (int i = 0; i < 4; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns_up[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_up_done = true;
    }
    kern = kerns_up[shape];
  }
  if (d.dims == 3) {  // only reached without prologue and with whole slices per item (w44h_geom)
    static const kern_t kerns3d[2][2] = {{conv_wino44h_kernel<false, 9, 0, false, true>, conv_wino44h_kernel<false, 9, 0, true, true>},
                                         {conv_wino44h_kernel<false, 10, 0, false, true>, conv_wino44h_kernel<false, 10, 0, true, true>}};
    static bool attr3_done = false;
    if (!attr3_done) {
      for (int i = 0; i < 4; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns3d[i / 2][i % 2]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
      attr3_done = true;
    }
    kern = kerns3d[g.NRT == 9 ? 0 : 1][d.residual ? 1 : 0];
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
  if (sw().w44h_reg) {  // the register-fed form (conv_wino44r.hip, round 5): same item, same packed weights, bit-identical results
    W44HGeom gr = g;
    w44r_relayout(d, gr);
    // serpentine item order across consecutive launches (DDPM_W44R_SERP=1; measured +-0, default off): results do not depend on it
    static unsigned launch_parity = 0;
    gr.rev = sw().w44r_serp ? (int)(launch_parity++ & 1) : 0;
    if (w44h_lds_bytes(gr) > 160 * 1024) {
      set_error("conv_wino44r: pixel-tile layout does not fit");
      return DDPM_EINVAL;
    }
    if (const int rc = launch_conv_wino44r(dk, gr, w44h_lds_bytes(gr), s)) return rc;
  } else {
    hipLaunchKernelGGL(kern, dim3(g.grid), dim3(512), lds, s, dk, g);
    DDPM_CHECK_LAUNCH();
  }
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
