// conv_direct.hip -- generic fp32 direct convolution (any H, W, Cin, Cout) on the vector ALU.
//
// Same fused semantics as conv_mfma.hip (affine+SiLU prologue, virtual concat, upsample /
// stride-2 indexing, bias + chan_add + residual epilogue).  It serves the layers the MFMA
// tiling does not cover -- conv_in (Cin = 1 / 3), conv_out (Cout = 1 / 3), odd extents such as
// 28x28 -- and is the on-device cross-check for the MFMA kernel in tests.
// Reference call site: /root/reference/src/trainers/reconstruct.py:151-153.
#include <stdlib.h>

#include "common.h"

namespace ddpm {

template <int COB>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ddpm_conv_desc a) {
  const int HWo = a.Ho * a.Wo, HWi = a.Hi * a.Wi;
  const int Cin = a.C1 + a.C2;
  const int T = a.ksize * a.ksize;
  const int pad = a.ksize == 3 ? 1 : 0;
  const int s = a.mode == DDPM_CONV_STRIDE2 ? 2 : 1;
  const bool up = a.mode == DDPM_CONV_UPSAMPLE2;
  const int Hv = up ? a.Ho : a.Hi, Wv = up ? a.Wo : a.Wi;

  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.z;
  const int co0 = blockIdx.y * COB;
  if (p >= HWo) return;
  const int ho = p / a.Wo, wo = p - ho * a.Wo;

  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = 0.f;

  for (int ci = 0; ci < Cin; ++ci) {
    const float *plane = (ci < a.C1) ? a.in1 + ((size_t)n * a.C1 + ci) * HWi
                                     : a.in2 + ((size_t)n * a.C2 + (ci - a.C1)) * HWi;
    float sc = 1.f, sh = 0.f;
    if (a.gscale) {
      sc = a.gscale[(size_t)n * Cin + ci];
      sh = a.gshift[(size_t)n * Cin + ci];
    }
    for (int kh = 0; kh < a.ksize; ++kh) {
      for (int kw = 0; kw < a.ksize; ++kw) {
        const int hv = ho * s + kh - pad, wv = wo * s + kw - pad;
        float v = 0.f;
        if (hv >= 0 && hv < Hv && wv >= 0 && wv < Wv) {
          v = up ? plane[(hv >> 1) * a.Wi + (wv >> 1)] : plane[hv * a.Wi + wv];
          if (a.gscale) v = v * sc + sh;
          if (a.act == DDPM_ACT_SILU) v = silu_fast(v);
          if (a.act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
        }
        const int t = kh * a.ksize + kw;
#pragma unroll
        for (int j = 0; j < COB; ++j) {
          const int co = (co0 + j < a.Cout) ? co0 + j : a.Cout - 1;
          acc[j] = fmaf(v, a.w_raw[((size_t)co * Cin + ci) * T + t], acc[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co = co0 + j;
    if (co < a.Cout) {
      const size_t idx = ((size_t)n * a.Cout + co) * HWo + p;
      float v = acc[j];
      if (a.bias) v += a.bias[co];
      if (a.chan_add) v += a.chan_add[(size_t)n * a.chan_add_stride + co];
      if (a.residual) v += a.residual[idx];
      if (a.out_act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
      a.out[idx] = v;
    }
  }
}

// ---- 3x3 convolution with very few output channels (conv_out: 128 -> 1 or 3) -------------------
// HBM-bound: the output is tiny, the cost is ONE read of the input with the GroupNorm affine + SiLU
// applied once per element.  conv_direct_kernel evaluated the activation 9x per element (once per
// tap) and ran at 0.46 TB/s.  Here a 256-pixel tile (whole rows, one thread per output pixel) stages
// 8 channels at a time -- activated, zero halo -- through LDS and every thread then reads its 9 x 8
// neighbourhood from LDS.  Algorithmic bytes: 4 * Cin * H * W per image.
constexpr int kSC_CH = 8;

template <int NPOS>
__global__ __launch_bounds__(256) void conv_smallco_kernel(const ddpm_conv_desc a, int TH, int RS, int PS) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [8][PS]
  const int HW = a.Ho * a.Wo;
  const int Cin = a.C1 + a.C2;
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int h0 = blockIdx.x * TH;
  const bool live = tid < TH * a.Wo;  // ragged tiles (e.g. 9 x 28 = 252 pixels): the last threads only help staging
  const int th = live ? tid / a.Wo : 0, tw = live ? tid - th * a.Wo : 0;

  int soff[NPOS];
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const int r = tid + 256 * j;
    soff[j] = -1;
    if (r < PS) {
      const int ir = r / RS, ic = r - ir * RS;
      const int hv = h0 + ir - 1, wv = ic - 1;
      if (hv >= 0 && hv < a.Hi && wv >= 0 && wv < a.Wi) soff[j] = hv * a.Wi + wv;
    }
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int pix = th * RS + tw;  // top-left tap of this thread's 3x3 window inside the haloed plane

  // chunk c0 + 8 is loaded into registers while chunk c0 is consumed from LDS (the kernel is latency-bound otherwise:
  // 16 dependent load -> barrier -> compute rounds per workgroup)
  float pre[NPOS][kSC_CH], psc[kSC_CH], psh[kSC_CH];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int c = 0; c < kSC_CH; ++c) {
      const int ci = c0 + c;
      const bool okc = ci < Cin;
      const float *plane = !okc ? a.in1
                           : (ci < a.C1) ? a.in1 + ((size_t)n * a.C1 + ci) * HW
                                         : a.in2 + ((size_t)n * a.C2 + (ci - a.C1)) * HW;
#pragma unroll
      for (int j = 0; j < NPOS; ++j) pre[j][c] = (okc && soff[j] >= 0) ? plane[soff[j]] : 0.f;
      psc[c] = (okc && a.gscale) ? a.gscale[(size_t)n * Cin + ci] : 1.f;
      psh[c] = (okc && a.gscale) ? a.gshift[(size_t)n * Cin + ci] : 0.f;
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < Cin; c0 += kSC_CH) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
      const int r = tid + 256 * j;
      if (r < PS) {
#pragma unroll
        for (int c = 0; c < kSC_CH; ++c) {
          float v = 0.f;
          if (soff[j] >= 0 && c0 + c < Cin) {
            v = pre[j][c];
            if (a.gscale) v = v * psc[c] + psh[c];
            if (a.act == DDPM_ACT_SILU) v = silu_fast(v);
            if (a.act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
          }
          tile[c * PS + r] = v;
        }
      }
    }
    __syncthreads();
    if (c0 + kSC_CH < Cin) fetch(c0 + kSC_CH);
#pragma unroll
    for (int c = 0; c < kSC_CH; ++c) {
      const int ci = c0 + c;
      if (ci < Cin) {
        float x[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = tile[c * PS + pix + kh * RS + kw];
        for (int co = 0; co < a.Cout; ++co) {
          const float *w = a.w_raw + ((size_t)co * Cin + ci) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[co] = fmaf(x[t], w[t], acc[co]);
        }
      }
    }
  }
  const int p = (h0 + th) * a.Wo + tw;
  for (int co = 0; co < a.Cout && live; ++co) {
    const size_t idx = ((size_t)n * a.Cout + co) * HW + p;
    float v = acc[co];
    if (a.bias) v += a.bias[co];
    if (a.chan_add) v += a.chan_add[(size_t)n * a.chan_add_stride + co];
    if (a.residual) v += a.residual[idx];
    if (a.out_act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
    a.out[idx] = v;
  }
}

// ---- conv_out, wave-parallel over channels ------------------------------------------------------------------------
// conv_smallco_kernel above walks the channels 8 at a time with two block barriers per chunk: 16 dependent
// load -> barrier -> LDS -> barrier rounds per workgroup, 1.2 TB/s.  Here every WAVE owns a quarter of the channels
// and a private double-buffered LDS plane: it loads the haloed plane of its next channel (6 loads per lane in
// flight) while the 3x3 windows of the current one are read back, with no block barrier until the final four-way
// reduction of the per-wave partial sums.  Same tile (256 pixels = whole rows), one thread-quad of pixels per lane.
constexpr int kSW_NJ = 7;   // plane elements per lane: PS <= 448 (64x64 images: four rows + halo = 396)
constexpr int kSW_PPL = 4;  // output pixels per lane: 256 / 64

// NWV waves share the input channels (wave, wave + NWV, ...): 4, or 16 for launches of a few images, whose 64 workgroups x 4 waves
// each walked 32 channels one after the other (49 us for 16 images; the partial sums are added in wave order either way)
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void conv_smallco_wave_kernel(const ddpm_conv_desc a, int TH, int RS, int PS) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [NWV waves][2][PS] planes, then [NWV][256][4] partials
  const int HW = a.Ho * a.Wo;
  const int Cin = a.C1 + a.C2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const int h0 = blockIdx.x * TH;
  float *plane_lds = lds + wave * 2 * PS;

  int soff[kSW_NJ];
#pragma unroll
  for (int j = 0; j < kSW_NJ; ++j) {
    const int r = lane + 64 * j;
    soff[j] = -1;
    if (r < PS) {
      const int ir = r / RS, ic = r - ir * RS;
      const int hv = h0 + ir - 1, wv = ic - 1;
      if (hv >= 0 && hv < a.Hi && wv >= 0 && wv < a.Wi) soff[j] = hv * a.Wi + wv;
    }
  }
  int pix[kSW_PPL];  // top-left tap of each of this lane's pixels inside the haloed plane
#pragma unroll
  for (int k = 0; k < kSW_PPL; ++k) {
    const int q = lane + 64 * k;
    const int th = q / a.Wo, tw = q - th * a.Wo;
    pix[k] = (q < TH * a.Wo) ? th * RS + tw : 0;
  }
  float acc[kSW_PPL][4];
#pragma unroll
  for (int k = 0; k < kSW_PPL; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;

  // this wave's channels: wave, wave + 4, ... (the four waves read neighbouring planes at the same time); two planes
  // are always in flight per wave (register sets 0 / 1), a third is being consumed out of LDS
  float pre[2][kSW_NJ], psc[2] = {1.f, 1.f}, psh[2] = {0.f, 0.f};
  auto fetch = [&](int slot, int ci) {
    const float *plane = (ci < a.C1) ? a.in1 + ((size_t)n * a.C1 + ci) * HW
                                     : a.in2 + ((size_t)n * a.C2 + (ci - a.C1)) * HW;
#pragma unroll
    for (int j = 0; j < kSW_NJ; ++j) pre[slot][j] = soff[j] >= 0 ? plane[soff[j]] : 0.f;
    if (a.gscale) {
      psc[slot] = a.gscale[(size_t)n * Cin + ci];
      psh[slot] = a.gshift[(size_t)n * Cin + ci];
    }
  };
  auto consume = [&](int slot, int ci, int buf) {
    float *pl = plane_lds + buf * PS;
#pragma unroll
    for (int j = 0; j < kSW_NJ; ++j) {
      const int r = lane + 64 * j;
      if (r < PS) {
        float v = 0.f;
        if (soff[j] >= 0) {
          v = pre[slot][j];
          if (a.gscale) v = v * psc[slot] + psh[slot];
          if (a.act == DDPM_ACT_SILU) v = silu_fast(v);
          if (a.act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
        }
        pl[r] = v;
      }
    }
    if (ci + 2 * NWV < Cin) fetch(slot, ci + 2 * NWV);  // refill this register set: the plane two channels ahead
    // (LDS operations of one wave complete in order: the reads below see the writes above without a barrier)
    float wreg[4][9];
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      const float *w = a.w_raw + ((size_t)min(co, a.Cout - 1) * Cin + ci) * 9;  // wave-uniform
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[co][t] = w[t];
    }
#pragma unroll
    for (int k = 0; k < kSW_PPL; ++k) {
      float x[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = pl[pix[k] + kh * RS + kw];
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < a.Cout) {
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[k][co] = fmaf(x[t], wreg[co][t], acc[k][co]);
        }
    }
  };
  if (wave < Cin) fetch(0, wave);
  if (wave + NWV < Cin) fetch(1, wave + NWV);
  for (int ci = wave; ci < Cin; ci += 2 * NWV) {
    consume(0, ci, 0);
    if (ci + NWV < Cin) consume(1, ci + NWV, 1);
  }
  // ---- four-way reduction of the per-wave partial sums (fixed order: wave 0 + 1 + 2 + 3), epilogue, store ------------
  float *red = lds + NWV * 2 * PS;
#pragma unroll
  for (int k = 0; k < kSW_PPL; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[(wave * 256 + lane + 64 * k) * 4 + c] = acc[k][c];
  __syncthreads();
  if (tid < TH * a.Wo) {
    const int th = tid / a.Wo, tw = tid - th * a.Wo;
    const int p = (h0 + th) * a.Wo + tw;
    for (int co = 0; co < a.Cout; ++co) {
      float v = red[(0 * 256 + tid) * 4 + co];
#pragma unroll
      for (int wv = 1; wv < NWV; ++wv) v += red[(wv * 256 + tid) * 4 + co];  // (wave order: ((0 + 1) + 2) + 3 ...)
      const size_t idx = ((size_t)n * a.Cout + co) * HW + p;
      if (a.bias) v += a.bias[co];
      if (a.chan_add) v += a.chan_add[(size_t)n * a.chan_add_stride + co];
      if (a.residual) v += a.residual[idx];
      if (a.out_act == DDPM_ACT_RELU) v = fmaxf(v, 0.f);
      a.out[idx] = v;
    }
  }
}

// ---- conv_in: 3x3, 1..4 input channels, many output channels ------------------------------------------------------
// HBM-bound on the OUTPUT (134 MB at B = 256, 128 channels, 32x32; the input is 1 MB).  conv_direct_kernel<8> spends a
// workgroup per (256 pixels, 8 couts): 16 384 tiny workgroups, 1.5 TB/s.  Here a thread keeps its 9 x Cin input
// values in registers and walks ALL output channels (weights wave-uniform -> scalar loads), issuing one coalesced
// store per channel back to back.
template <int CIN>
__global__ __launch_bounds__(256) void conv_smallci_kernel(const ddpm_conv_desc a) {
  const int HW = a.Ho * a.Wo;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= HW) return;
  const int ho = p / a.Wo, wo = p - ho * a.Wo;
  float x[CIN * 9];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) {
    const float *plane = a.in1 + ((size_t)n * CIN + ci) * HW;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hv = ho + kh - 1, wv = wo + kw - 1;
        x[ci * 9 + kh * 3 + kw] = (hv >= 0 && hv < a.Hi && wv >= 0 && wv < a.Wi) ? plane[hv * a.Wi + wv] : 0.f;
      }
  }
  float *dst = a.out + (size_t)n * a.Cout * HW + p;
  // (gridDim.z slices of the output channels: a launch of a few images has too few pixel blocks to fill the chip and every
  // thread's channel loop is a serial chain of scalar loads and stores -- 38 us for 16 images, 128 channels)
  const int cpz = (a.Cout + gridDim.z - 1) / gridDim.z, co_end = min(a.Cout, ((int)blockIdx.z + 1) * cpz);
  // desc.stats_out (launch_conv_direct sets it only when HW is a multiple of 256: every lane of the workgroup holds a pixel): the
  // next GroupNorm's {mean, M2} of this workgroup's 256 pixels = one slice of the image -- the wave's 64 values by a DPP sum,
  // their squared deviations about the wave mean by another, the four waves merged (Chan, fixed order) through LDS at the end
  extern __shared__ float2 wred[];  // [cout of this z slice][wave]
  const bool emit = a.stats_out != nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 4
  for (int co = blockIdx.z * cpz; co < co_end; ++co) {  // (unrolled: the scalar weight loads of four channels go out together)
    const float *w = a.w_raw + (size_t)co * CIN * 9;  // wave-uniform
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < CIN * 9; ++k) acc = fmaf(x[k], w[k], acc);  // (ci, kh, kw) order, as conv_direct_kernel
    if (a.bias) acc += a.bias[co];
    dst[(size_t)co * HW] = acc;
    if (emit) {
      const float mean = wave_sum_dpp_last_lane(acc);  // (valid in lane 63)
      const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mean), 63)) * (1.f / 64.f);
      const float dv = acc - mu;
      const float q = wave_sum_dpp_last_lane(dv * dv);
      if (lane == 63) wred[(co - blockIdx.z * cpz) * 4 + wave] = make_float2(mu, q);
    }
  }
  if (emit) {
    __syncthreads();
    const int parts = HW / 256;
    for (int c = threadIdx.x; c < co_end - (int)blockIdx.z * cpz; c += 256) {
      const float2 w0 = wred[c * 4], w1 = wred[c * 4 + 1], w2 = wred[c * 4 + 2], w3 = wred[c * 4 + 3];
      const float m01 = 0.5f * (w0.x + w1.x), d01 = w1.x - w0.x, q01 = (w0.y + w1.y) + d01 * d01 * 32.f;
      const float m23 = 0.5f * (w2.x + w3.x), d23 = w3.x - w2.x, q23 = (w2.y + w3.y) + d23 * d23 * 32.f;
      const float mm = 0.5f * (m01 + m23), dd = m23 - m01, qq = (q01 + q23) + dd * dd * 64.f;
      const int co = blockIdx.z * cpz + c;
      *reinterpret_cast<float2 *>(a.stats_out + (((size_t)n * a.Cout + co) * parts + blockIdx.x) * 2) = make_float2(mm, qq);
    }
  }
}

static bool smallci_supported(const ddpm_conv_desc &d) {
  const bool on = sw().convin_fast;
  return on && d.C2 == 0 && d.C1 >= 1 && d.C1 <= 4 && d.ksize == 3 && d.mode == DDPM_CONV_NORMAL && !d.gscale &&
         d.act == DDPM_ACT_NONE && !d.chan_add && !d.residual && d.out_act == DDPM_ACT_NONE && d.Cout >= 16 &&
         d.Di <= 1 && d.Do <= 1 && d.dims != 3;
}

// slices per (image, cout) of the statistics conv_smallci_kernel writes to desc.stats_out (0: it does not)
static int smallci_stats_parts(const ddpm_conv_desc &d) {
  const int HW = d.Ho * d.Wo;
  if (!smallci_supported(d) || HW % 256 || HW / 256 > 8) return 0;
  return HW / 256;
}

static bool smallco_supported(const ddpm_conv_desc &d, int &TH, int &RS, int &PS) {
  if (d.Cout > 4 || d.ksize != 3 || d.mode != DDPM_CONV_NORMAL || d.Wo > 256 || d.Di > 1 || d.Do > 1) return false;
  TH = 0;
  for (int th = 256 / d.Wo; th >= 1; --th)
    if (d.Ho % th == 0) { TH = th; break; }
  if (TH == 0 || 2 * TH * d.Wo < 256) return false;
  RS = d.Wo + 2;
  PS = (TH + 2) * RS;
  return PS <= 512;
}

int launch_conv_direct(const ddpm_conv_desc &d, hipStream_t s) {
  DDPM_CHECK_ARG(d.w_raw != nullptr, "conv_direct: w_raw is NULL");
  DDPM_CHECK_ARG(d.ksize == 1 || d.ksize == 3, "conv_direct: ksize must be 1 or 3");
  DDPM_CHECK_ARG(d.B <= 65535, "conv_direct: batch > 65535");
  const int HWo = d.Ho * d.Wo;
  const double cin = d.C1 + d.C2, taps = d.ksize * d.ksize;
  const int Cin_i = d.C1 + d.C2;
  int TH, RS, PS;
  if (smallco_supported(d, TH, RS, PS)) {
    ProfScope prof(s, "conv3x3_small_cout", 2.0 * d.B * HWo * d.Cout * cin * 9,
                   4.0 * ((double)d.B * cin * HWo + (double)d.B * HWo * d.Cout + d.Cout * cin * 9));
    dim3 grid(d.Ho / TH, d.B);
    const bool wave_ok = sw().convout_wave;
    if (wave_ok && PS <= 64 * kSW_NJ && Cin_i >= 16) {
      const long w16_max = sw().convout_w16_maxwg;
      // up to two workgroups per CU (B <= 128 at 32x32: 54 -> 44 us at B = 128; at four per CU, B = 256, the four-wave form wins 70 vs 84)
      const long w16_wgs = w16_max >= 0 ? w16_max : 2L * device_cus();
      if ((long)grid.x * grid.y <= w16_wgs && Cin_i >= 64) {  // sixteen waves per workgroup
        const size_t lds_w = ((size_t)16 * 2 * PS + 16 * 256 * 4) * sizeof(float);
        static bool attr_done = false;
        if (!attr_done) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_smallco_wave_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr_done = true;
        }
        hipLaunchKernelGGL(conv_smallco_wave_kernel<16>, grid, dim3(1024), lds_w, s, d, TH, RS, PS);
        DDPM_CHECK_LAUNCH();
        return 0;
      }
      const size_t lds_w = ((size_t)4 * 2 * PS + 4 * 256 * 4) * sizeof(float);
      hipLaunchKernelGGL(conv_smallco_wave_kernel<4>, grid, dim3(256), lds_w, s, d, TH, RS, PS);
      DDPM_CHECK_LAUNCH();
      return 0;
    }
    const size_t lds = (size_t)kSC_CH * PS * sizeof(float);
    if (PS <= 256)
      hipLaunchKernelGGL(conv_smallco_kernel<1>, grid, dim3(256), lds, s, d, TH, RS, PS);
    else
      hipLaunchKernelGGL(conv_smallco_kernel<2>, grid, dim3(256), lds, s, d, TH, RS, PS);
    DDPM_CHECK_LAUNCH();
    return 0;
  }
  if (smallci_supported(d)) {
    ProfScope prof(s, "conv3x3_small_cin", 2.0 * d.B * HWo * d.Cout * cin * 9,
                   4.0 * ((double)d.B * cin * HWo + (double)d.B * HWo * d.Cout + d.Cout * cin * 9));
    const long blocks = (long)((HWo + 255) / 256) * d.B;
    int zs = 1;
    const long zblk = sw().convin_blocks_per_cu;  // (2 -> 8: 61 -> 35 us at B = 128, 75 -> 59 at B = 256)
    while (zs < 8 && blocks * zs * 2 <= device_cus() * zblk && d.Cout / (zs * 2) >= 16) zs *= 2;  // up to `zblk` blocks per CU
    dim3 grid((HWo + 255) / 256, d.B, zs);
    ddpm_conv_desc dk = d;
    if (smallci_stats_parts(d) == 0) dk.stats_out = nullptr;
    const size_t lds = dk.stats_out ? (size_t)((d.Cout + zs - 1) / zs) * 4 * sizeof(float2) : 0;
    switch (d.C1) {
      case 1: hipLaunchKernelGGL(conv_smallci_kernel<1>, grid, dim3(256), lds, s, dk); break;
      case 2: hipLaunchKernelGGL(conv_smallci_kernel<2>, grid, dim3(256), lds, s, dk); break;
      case 3: hipLaunchKernelGGL(conv_smallci_kernel<3>, grid, dim3(256), lds, s, dk); break;
      default: hipLaunchKernelGGL(conv_smallci_kernel<4>, grid, dim3(256), lds, s, dk); break;
    }
    DDPM_CHECK_LAUNCH();
    return 0;
  }
  ProfScope prof(s, "conv_direct", 2.0 * d.B * HWo * d.Cout * cin * taps,
                 4.0 * ((double)d.B * cin * d.Hi * d.Wi + (double)d.B * HWo * d.Cout * (d.residual ? 2 : 1) +
                        d.Cout * cin * taps));
  if (d.Cout <= 4) {
    dim3 grid((HWo + 255) / 256, d.Cout, d.B);
    hipLaunchKernelGGL(conv_direct_kernel<1>, grid, dim3(256), 0, s, d);
  } else {
    dim3 grid((HWo + 255) / 256, (d.Cout + 7) / 8, d.B);
    hipLaunchKernelGGL(conv_direct_kernel<8>, grid, dim3(256), 0, s, d);
  }
  DDPM_CHECK_LAUNCH();
  return 0;
}

size_t conv_scratch_floats(const ddpm_conv_desc &d) {
  const bool vol = d.dims == 3 || d.Di > 1 || d.Do > 1;  // the Winograd splits are 2-D only
  const size_t a = vol ? 0 : conv_wino44_scratch_floats(d), b = vol ? 0 : conv_wino_scratch_floats(d);
  const size_t c = linear_skinny_supported(d) ? 0 : conv_mfma_scratch_floats(d);
  const size_t h = vol ? 0 : conv_wino44h_scratch_floats(d);
  const size_t ab = (a > b ? a : b) > h ? (a > b ? a : b) : h;
  const size_t e1 = vol ? 0 : conv_d3s_scratch_floats(d), e2 = vol ? 0 : conv_d1s_scratch_floats(d);
  const size_t e3 = vol ? 0 : conv_d3s2_scratch_floats(d);
  const size_t e = (e1 > e2 ? e1 : e2) > e3 ? (e1 > e2 ? e1 : e2) : e3;
  const size_t abc = ab > c ? ab : c;
  return abc > e ? abc : e;
}

// mirrors conv_dispatch: which kernel takes d, and whether its epilogue writes desc.stats_out
int conv_stats_parts(const ddpm_conv_desc &d) {
  const bool is3d = d.dims == 3 && d.ksize != 1;
  if (is3d || d.ksize != 3 || d.Di > 1 || d.Do > 1 || linear_skinny_supported(d)) return 0;
  if (conv_d3s_supported(d) || conv_d3s2_supported(d)) return conv_d3s_stats_parts(d);  // (from its reduce pass)
  if (conv_wino44h_supported(d)) return conv_wino44h_stats_parts(d);
  if (conv_wino44_supported(d)) return 0;
  if (conv_wino_supported(d)) return conv_wino_stats_parts(d);  // (the Upsample form only)
  if (d.mode == DDPM_CONV_STRIDE2 && conv_s2h_supported(d)) return conv_s2h_stats_parts(d);  // Downsample
  int TH, RS, PS;
  if (d.mode == DDPM_CONV_NORMAL && !conv_mfma_supported(d) && !smallco_supported(d, TH, RS, PS))
    return smallci_stats_parts(d);  // conv_in (launch_conv_direct)
  return 0;
}

int conv_dispatch(const ddpm_conv_desc &d, hipStream_t s) {
  DDPM_CHECK_ARG(d.in1 && d.out && d.B > 0 && d.Cout > 0 && d.C1 > 0, "conv: null tensor or empty shape");
  DDPM_CHECK_ARG(d.C2 == 0 || d.in2, "conv: C2 > 0 but in2 is NULL");
  DDPM_CHECK_ARG((d.gscale == nullptr) == (d.gshift == nullptr), "conv: gscale/gshift must come together");
  if (d.mode == DDPM_CONV_NORMAL)
    DDPM_CHECK_ARG(d.Hi == d.Ho && d.Wi == d.Wo, "conv: normal mode needs Hi == Ho, Wi == Wo");
  if (d.mode == DDPM_CONV_UPSAMPLE2)
    DDPM_CHECK_ARG(d.Ho == 2 * d.Hi && d.Wo == 2 * d.Wi && d.ksize == 3, "conv: upsample needs Ho == 2 Hi, k == 3");
  if (d.mode == DDPM_CONV_STRIDE2 && d.ksize == 3)
    DDPM_CHECK_ARG(d.Ho == (d.Hi + 1) / 2 && d.Wo == (d.Wi + 1) / 2, "conv: stride-2 k3 needs Ho == ceil(Hi / 2)");
  if (d.mode == DDPM_CONV_STRIDE2 && d.ksize == 4)
    DDPM_CHECK_ARG(d.Ho == d.Hi / 2 && d.Wo == d.Wi / 2 && d.Ho > 0 && d.Wo > 0, "conv: stride-2 k4 needs Ho == Hi / 2");
  DDPM_CHECK_ARG(d.mode != DDPM_CONV_STRIDE2 || d.ksize == 3 || d.ksize == 4, "conv: stride-2 needs k == 3 or 4");
  const bool is3d = d.dims == 3 && d.ksize != 1;
  if (is3d || d.mode == DDPM_CONV_TRANSPOSE2 || d.ksize == 4) {
    // 3-D convolutions, k4 s2 and ConvTranspose only exist on the MFMA kernel (no generic fallback)
    const int Di = d.Di > 1 ? d.Di : 1, Do = d.Do > 1 ? d.Do : 1;
    if (is3d && d.mode == DDPM_CONV_NORMAL) DDPM_CHECK_ARG(Di == Do, "conv3d: normal mode needs Di == Do");
    if (is3d && d.mode == DDPM_CONV_UPSAMPLE2) DDPM_CHECK_ARG(Do == 2 * Di, "conv3d: upsample needs Do == 2 Di");
    if (is3d && d.mode == DDPM_CONV_STRIDE2)
      DDPM_CHECK_ARG(Do == (d.ksize == 3 ? (Di + 1) / 2 : Di / 2) && Do > 0, "conv3d: stride-2 output depth");
    if (is3d && conv_wino44h_supported(d)) return launch_conv_wino44h(d, s);         // VQ-VAE residual units, split-f16
    if (is3d && conv_wino44_supported(d)) return launch_conv_wino44(d, s);
    if (is3d && d.w_wino && conv_wino_supported(d)) return launch_conv_wino(d, s);
    DDPM_CHECK_ARG(conv_mfma_supported(d),
                   "conv: 3-D / k4 / transposed convolutions need an MFMA tiling (Cin %% 4 (8), Cout %% 128, packed weights)");
    return launch_conv_mfma(d, s);
  }
  DDPM_CHECK_ARG(d.Di <= 1 && d.Do <= 1, "conv: Di / Do > 1 needs dims == 3");
  if (linear_skinny_supported(d)) return launch_linear_skinny(d, s);  // Linear over <= 1024 rows: latency, not FLOPs
  if (conv_d3s_supported(d)) return launch_conv_d3s(d, s);          // small launches: one-shot direct 3x3, split-f16 (round 4)
  if (conv_wino44h_supported(d)) return launch_conv_wino44h(d, s);  // F(4x4) with split-f16 position GEMMs
  if (conv_wino44_supported(d)) return launch_conv_wino44(d, s);
  if (conv_wino_supported(d)) return launch_conv_wino(d, s);
  if (conv_d3s2_supported(d)) return launch_conv_d3s2(d, s);  // Downsample of small launches: one-shot, split-f16 (round 4)
  if (conv_s2h_supported(d)) return launch_conv_s2h(d, s);  // Downsample: direct 3x3 stride 2 on the f16 MFMA, split-f16 operands
  if (conv_d1s_supported(d)) return launch_conv_d1s(d, s);  // small launches: one-shot 1x1, split-f16 (round 4)
  if (conv1x1_dma_supported(d) && conv_mfma_supported(d)) return launch_conv1x1_dma(d, s);
  if (conv_mfma_supported(d)) return launch_conv_mfma(d, s);
  return launch_conv_direct(d, s);
}

}  // namespace ddpm

// Would ddpm_conv_f32 run this stride-1 3x3 descriptor on the split-f16 F(4x4) kernel if it were given w_wino44h?  (A caller that
// re-packs weights every step -- the training step -- packs the F(2x2) fallback form only when the answer is no.)
extern "C" int ddpm_conv_takes_wino44h(const ddpm_conv_desc *dp) {
  if (!dp) return 0;
  ddpm_conv_desc d = *dp;
  if (d.dims == 3 || d.ksize != 3) return 0;
  if (!d.w_wino44h) d.w_wino44h = reinterpret_cast<const uint16_t *>(uintptr_t(64));  // (only tested for non-NULL)
  if (!d.scratch) {  // (a launch split over channel slices needs scratch: the caller will size it with ddpm_conv_scratch_floats)
    d.scratch = reinterpret_cast<float *>(uintptr_t(64));
    d.scratch_floats = ~size_t(0);
  }
  if (ddpm::linear_skinny_supported(d) || (d.w_d3h && ddpm::conv_d3s_supported(d))) return 0;
  return ddpm::conv_wino44h_supported(d) ? 1 : 0;
}

