/*
 * ddpm_ood_hip.h -- C ABI of libddpm_ood_hip.so (gfx950 / CDNA4).
 *
 * Drop-in boundary for the multi-t DDPM reconstruction hot path of marksgraham/ddpm-ood.
 * The reference has no FFI of its own: its boundary is the duck-typed Python call surface
 * between src/trainers/reconstruct.py and the third-party `generative` package
 * (SURVEY.md 8b).  Each entry point below names the reference call site it replaces.
 * The Python mirror classes in ddpm_ood_amd/ bind these symbols with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated;
 *   - tensors are contiguous fp32 NC(D)HW, timesteps are int64 (torch.long);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - nothing here allocates, frees or synchronises: workspaces and packed-parameter blobs
 *     are caller-owned (PyTorch-ROCm caching allocator in the Python host);
 *   - return value 0 = success, otherwise a negative DDPM_E* code or a positive
 *     hipError_t; ddpm_last_error() returns a thread-local message.
 */
#ifndef DDPM_OOD_HIP_H
#define DDPM_OOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDPM_ABI_VERSION 10

#define DDPM_EINVAL (-1)      /* bad argument / unsupported shape      */
#define DDPM_ENOPARAM (-2)    /* unknown or missing parameter name     */
#define DDPM_EWORKSPACE (-3)  /* caller workspace too small            */

typedef void *ddpm_stream_t;

int ddpm_abi_version(void);
const char *ddpm_last_error(void);

/* ------------------------------------------------------------------------------------
 * Stand-alone operators (each replaces one ATen dispatch chain on the path).
 * ---------------------------------------------------------------------------------- */

/* conv modes / activation flags for ddpm_conv_f32 */
#define DDPM_CONV_NORMAL 0
#define DDPM_CONV_STRIDE2 1   /* kernel 3, stride 2, pad 1  (Downsample.op)               */
#define DDPM_CONV_UPSAMPLE2 2 /* nearest x2 folded into the input indexing (Upsample)     */
#define DDPM_CONV_TRANSPOSE2 3 /* ConvTranspose kernel 4, stride 2, pad 1 (VQ-VAE decoder upsampling); w_packed from
                                  ddpm_pack_convtr_weight_f32                                */
#define DDPM_ACT_NONE 0
#define DDPM_ACT_SILU 1
#define DDPM_ACT_RELU 2

/* Fused convolution / linear descriptor.  out = conv(act(affine(cat(in1,in2)))) + bias
 *                                               + chan_add[n, co] + residual
 * Replaces, inside DiffusionModelUNet.forward (reference call
 * src/trainers/reconstruct.py:151-153): F.group_norm+F.silu (via gscale/gshift),
 * torch.cat (in1/in2), F.interpolate(nearest) (mode), F.conv2d / F.linear (HW == 1),
 * the "+ temb[:, :, None, None]" broadcast (chan_add) and the residual add.            */
typedef struct ddpm_conv_desc {
  const float *in1;      /* [B, C1, Hi, Wi]                                              */
  const float *in2;      /* [B, C2, Hi, Wi] or NULL -- virtual channel concat            */
  int C1, C2;
  const float *w_packed; /* MFMA layout (ddpm_pack_conv_weight_f32) or NULL              */
  const float *w_raw;    /* torch layout [Cout, Cin, k, k]; used when w_packed == NULL or
                            the shape has no MFMA tiling                                 */
  const float *bias;     /* [Cout] or NULL                                               */
  const float *gscale;   /* [B, Cin] per-(image, channel) GroupNorm scale or NULL        */
  const float *gshift;   /* [B, Cin]                                                     */
  const float *chan_add; /* [B, chan_add_stride], entry [n, co] added, or NULL           */
  int chan_add_stride;
  const float *residual; /* [B, Cout, Ho, Wo] or NULL                                    */
  float *out;            /* [B, Cout, Ho, Wo]                                            */
  int B, Cout;
  int Hi, Wi;            /* stored input extent                                          */
  int Ho, Wo;            /* output extent                                                */
  int ksize;             /* 1, 3, or 4 (4: DDPM_CONV_STRIDE2 = kernel 4 stride 2 pad 1, or
                            DDPM_CONV_TRANSPOSE2)                                        */
  int mode;              /* DDPM_CONV_*                                                  */
  int act;               /* DDPM_ACT_* applied after the affine                          */
  int force_direct;      /* 1: take the generic direct kernel even if MFMA tiling exists */
  /* 3-D convolutions (F.conv3d of the LDM UNet, conv3d / conv_transpose3d of the VQ-VAE;
   * src/trainers/reconstruct.py:124,151-153,166) on NCDHW tensors: dims = 3, Di / Do = stored input / output
   * depth, kernel k x k x k.  w_packed then holds k depth slabs of k x k taps (ddpm_pack_conv3d_weight_f32) and
   * ONE launch walks (depth tap, channel group) chunks: out[n, :, d] = sum_kd conv2d(in[n, :, din(d, kd)],
   * w[:, :, kd]).  Supported: k = 3 with NORMAL / STRIDE2 / UPSAMPLE2, k = 4 with STRIDE2 / TRANSPOSE2; a 1x1x1
   * conv is the 2-D op over an (D*H) x W image.  dims = 0 or 2: plain 2-D (Di = Do = 0).                     */
  int Di, Do;
  int dims;
  /* dims = 3, kernel 3 (ABI 10; was reserved): bit kd set = depth tap kd of the 3x3x3 weight may be non-zero; 0 = all three.
   * 3 (taps 0, 1) and 6 (taps 1, 2) let the split-f16 F(4x4) form skip the all-zero tap; every other kernel ignores the field
   * (the zero tap then costs time, never correctness).  Set by the parity decomposition of ConvTranspose3d k4 s2
   * (ddpm_convtr3d_parity_weights_f32).  */
  int depth_taps;
  /* Optional, 2-D DDPM_CONV_UPSAMPLE2 only: weights folded by ddpm_fold_upsample_weight_f32.  A 3x3
   * conv over a nearest-x2 upsampled image is, for each of the 4 output parities (dy, dx), a 2x2 conv
   * over the low-res image whose taps are sums of the 3x3 taps that read the same source pixel:
   * 16 instead of 36 multiply-adds per 4 outputs, same result up to fp32 rounding of the tap sums.  */
  const float *w_folded;
  /* Activation applied to the finished output element (after bias / chan_add / residual / accumulate):
   * DDPM_ACT_NONE or DDPM_ACT_RELU.  Used by the VQ-VAE residual units, relu(x + conv2(relu(conv1(x)))).  */
  int out_act;
  int reserved;
  /* Optional, 2-D 3x3 only (DDPM_CONV_NORMAL or DDPM_CONV_UPSAMPLE2): weights pre-transformed by
   * ddpm_pack_wino_weight_f32 (U = G g G^T).  When present and the shape has a Winograd tiling (even H, W;
   * Cin % 8 == 0; Cout % 64 == 0) the conv runs as Winograd F(2x2, 3x3) on the fp32 MFMA pipe: 2.25x fewer
   * multiplies (4x fewer for UPSAMPLE2, where 9 of the 16 transform positions are non-zero on a nearest-x2
   * image); fp32 rounding differs from the direct form by ~1e-6 relative.                  */
  const float *w_wino;
  /* Optional scratch (ddpm_conv_scratch_floats): a launch with fewer work items than CUs (small batches; the F(4x4) kernel
   * at the 8x8 level) splits the channel stream of each item over 2 or 4 workgroups, whose partial outputs go to slabs of this buffer and are added in
   * a fixed order by a second pass.  NULL / too small: the convolution runs unsplit (same result up to fp32 rounding). */
  float *scratch;
  size_t scratch_floats;
  /* Optional, 2-D 3x3 DDPM_CONV_NORMAL only: weights pre-transformed by ddpm_pack_wino44_weight_f32 (U = G g G^T, 6 x 6).
   * When present, the shape has a 4x4 tiling (H, W % 4 == 0; Cin % 8 == 0; Cout % 64 == 0) and the launch fills at least
   * half of the chip (with `scratch`: after a 2- / 4-way channel split), the conv runs as Winograd F(4x4, 3x3): 4x fewer multiplies than the direct form (F(2x2): 2.25x); fp32
   * rounding differs from the direct form by ~3e-6 rms relative.  Takes precedence over w_wino.  */
  const float *w_wino44;
  /* Optional, 3x3 DDPM_CONV_NORMAL only (2-D, or dims = 3 with ddpm_pack_wino44h_weight3d; ABI 6): the F(4x4, 3x3) weights as split-f16 planes, packed by
   * ddpm_pack_wino44h_weight (U = 2^su G g G^T as hi = f16(U), lo = f16(U - hi), in the order the kernel's LDS-DMA lands
   * them; su per layer, the epilogue scale 1 / (2^3 2^su) stored behind the planes).  When present (Cin % 16 == 0, Cout % 64 == 0, same launch-size rule as w_wino44) the position GEMMs of the
   * Winograd convolution run on the f16 MFMA pipe with split-f16 products (every fp32 product rebuilt from four exact f16
   * partial products, fp32 accumulate); transforms and sums stay fp32.  Takes precedence over w_wino44;
   * DDPM_WINO44_F16X3=0 switches it off.  */
  const uint16_t *w_wino44h;
  /* Optional (ABI 7): GroupNorm statistics of the PRODUCED tensor, written by the convolution's epilogue so that the
   * following F.group_norm (reference call site src/trainers/reconstruct.py:151-153 -> generative ResnetBlock.norm2 /
   * the next block's norm1) needs no pass of its own over the activation.  Layout [B, Cout, parts, 2] floats:
   * {mean, sum of squared deviations} of each channel over one of `parts` equal slices of the image's pixels
   * (parts = ddpm_conv_stats_parts(d); 0 = this dispatch does not emit them and stats_out is ignored).  Merged pairwise
   * in a fixed order (no atomics): bit-reproducible.  ddpm_gn_finalize_f32 turns them into scale / shift.  Emitting
   * dispatches: the split-f16 F(4x4) 3x3 kernels (ResnetBlock convolutions, Upsample), the reduce pass of the small-launch
   * forms, and since round 5 conv_in's small-cin kernel (pixels per image a multiple of 256, at most 2 048) and the Downsample
   * kernels (an image's share of a 128-pixel tile at least 32 pixels).  */
  float *stats_out;
  /* Optional, 2-D 3x3 (ABI 8): the weights as split-f16 planes of the DIRECT convolution (nine taps on
   * v_mfma_f32_32x32x16_f16, three exact f16 partial products per fp32 product, fp32 accumulate), packed by
   * ddpm_pack_conv_d3h_weight (ddpm_conv_d3h_weight_halves(Cout, Cin) halves; 0: Cout % 128 or Cin % 8 != 0).  They feed the
   * one-shot kernel of launches far smaller than the chip (csrc/conv_d3s.hip: 8x8 / 16x16 images with at most 4 096 pixels
   * per launch, 32x32 images with at most 16 384, Cin % 32 == 0; DDPM_CONV_D3S=0 switches it off; also DDPM_CONV_STRIDE2
   * with 8x8 / 16x16 outputs and DDPM_CONV_UPSAMPLE2 with 16x16 / 32x32 outputs): channel slices of 32 into desc.scratch +
   * the fixed-order reduce pass, which also emits stats_out.  (The chip-filling direct kernel these planes were first built
   * for, round 4's conv_d3h.hip, measured 17-22 % slower than the Winograd form and was removed in round 5.)
   * For a 1x1 DDPM_CONV_NORMAL convolution the field carries the planes of ddpm_pack_conv_d1s_weight instead (the 1x1 form of
   * the small-launch kernel: at most 16 384 pixels per launch, Cout % 64 == 0, Cin % 128 == 0, act = none).  */
  const uint16_t *w_d3h;
} ddpm_conv_desc;

int ddpm_conv_f32(const ddpm_conv_desc *d, ddpm_stream_t stream);
/* Floats of scratch this descriptor can use (0: none); device- and shape-dependent, constant for a given process.  */
/* 1 if ddpm_conv_f32 would run this 2-D stride-1 3x3 descriptor on the split-f16 F(4x4) kernel given w_wino44h (which may be NULL
 * here): lets a caller that re-packs its weights every step (the training step) skip packing the F(2x2) fallback form.  */
int ddpm_conv_takes_wino44h(const ddpm_conv_desc *desc);
size_t ddpm_conv_scratch_floats(const ddpm_conv_desc *d);
/* Optional, 1x1 convolutions (ResnetBlock.skip_connection, the fused q / k / v projection; ABI 8): the weights pre-split into
 * the two f16 planes the DMA-fed split-f16 kernel multiplies (hi = f16(2^6 w), lo = f16((2^6 w - hi) 2^5), laid out
 * [cout tile 128][chunk of 16 channels][plane][k group][cout][8 k]), handed over in ddpm_conv_desc.w_wino44h.  Halves needed:
 * ddpm_conv1x1_h_weight_halves(Cout, Cin) (0: Cout % 128 or Cin % 16 != 0).  With them the kernel reads its A operand
 * straight from LDS and splits every input value once per workgroup instead of twice -- same products, same order,
 * bit-identical results; without them it splits the fp32 packed weights in registers as before. */
size_t ddpm_conv_d3h_weight_halves(int Cout, int Cin);
int ddpm_pack_conv_d3h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, ddpm_stream_t stream);
/* 1x1 planes of the small-launch kernel (ddpm_conv_desc.w_d3h of a ksize = 1 descriptor): [cout tile 64][chunk of 8 channels]
 * [plane hi | lo][cout 64][8] f16 of 2^su w, su per packed member; rows [cout_offset, cout_offset + Cout) of a
 * [Cout_total][Cin] weight are packed per call (the members of a fused q / k / v weight one by one; whole 64-cout tiles).
 * Halves needed for the whole weight: ddpm_conv_d1s_weight_halves(Cout_total, Cin) (0: Cout % 64 or Cin % 128 != 0).  */
size_t ddpm_conv_d1s_weight_halves(int Cout, int Cin);
int ddpm_pack_conv_d1s_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, int cout_offset, int Cout_total,
                              ddpm_stream_t stream);
size_t ddpm_conv1x1_h_weight_halves(int Cout, int Cin);
int ddpm_pack_conv1x1_h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, ddpm_stream_t stream);
/* Slices per (image, channel) of the statistics ddpm_conv_f32 writes to d->stats_out for this descriptor (1 .. 8), or 0
 * when the kernel it dispatches to does not emit them (the caller then runs ddpm_gn_scale_shift_f32 on the tensor).  */
int ddpm_conv_stats_parts(const ddpm_conv_desc *d);

/* Number of floats of the packed form of a [Cout, Cin, k, k] weight (0 if unpackable). */
size_t ddpm_packed_conv_weight_floats(int Cout, int Cin, int ksize);
/* Pack torch-layout weights into rows [cout_offset, cout_offset + Cout) of a packed
 * weight with `Cout_total` rows (lets q/k/v or all time_emb_proj share one GEMM).       */
int ddpm_pack_conv_weight_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize,
                              int cout_offset, int Cout_total, ddpm_stream_t stream);

/* [Cout, Cin, k, k, k] (k = 3 or 4) -> k packed depth slabs of k x k taps: the w_packed of a dims = 3 descriptor.  */
int ddpm_pack_conv3d_weight_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize,
                                ddpm_stream_t stream);

/* ConvTranspose k4 s2 p1 weights, torch layout [Cin, Cout, 4, 4] (dims = 2) or [Cin, Cout, 4, 4, 4] (dims = 3), packed
 * per output parity: out[2i + p] only sees kernel elements 3 - p - 2r (r = 0, 1) per axis, so every parity is a
 * 2 x 2 (x 2)-tap convolution over the input.  16 (64) * Cout * Cin floats; 0 if Cout % 128 or Cin % 8.           */
size_t ddpm_packed_convtr_weight_floats(int Cout, int Cin, int dims);
int ddpm_pack_convtr_weight_f32(const float *w_raw, float *w_packed, int Cin, int Cout, int dims, ddpm_stream_t stream);

/* VQ-VAE edge layers without an MFMA tiling (1 input / 1 output channel), NCDHW, stride 2, kernel 4, pad 1:
 *   ddpm_conv3d_k4s2_cin1_f32:   out[B, Cout, D/2, H/2, W/2] = relu?(conv3d(in[B, 1, D, H, W], w[Cout, 1, 4, 4, 4]) + bias)
 *   ddpm_convtr3d_k4s2_cout1_f32: out[B, 1, 2D, 2H, 2W] = conv_transpose3d(in[B, Cin, D, H, W], w[Cin, 1, 4, 4, 4]) + bias
 * (first encoder / last decoder layer at src/trainers/reconstruct.py:124,166; torch weight layouts, no packing).   */
int ddpm_conv3d_k4s2_cin1_f32(const float *in, const float *w, const float *bias, float *out, int B, int Cout, int D,
                              int H, int W, int relu, ddpm_stream_t stream);
/* Generic (transposed) convolution, dims = 2 ([B, C, 1, H, W] with Di = 1) or 3, any channel counts, symmetric kernel / stride
 * (1 or 2) / padding, torch weight layouts ([Cout, Cin, k..] or, transposed, [Cin, Cout, k..]; output_padding 0), optional
 * residual and ReLU epilogue: the always-available form behind the MFMA kernels, used by the VQ-VAE layers whose channel
 * counts have no MFMA tiling (nn.Conv3d / nn.ConvTranspose3d of generative's VQVAE, /root/reference/src/trainers/
 * reconstruct.py:124,166).  out extents: (e + 2 pad - k) / stride + 1, transposed (e - 1) stride - 2 pad + k.  */
int ddpm_convnd_generic_f32(const float *in, const float *w, const float *bias, const float *residual, float *out, int B, int Cin,
                            int Cout, int Di, int Hi, int Wi, int dims, int ksize, int stride, int pad, int transposed, int relu,
                            ddpm_stream_t stream);
int ddpm_convtr3d_k4s2_cout1_f32(const float *in, const float *w, const float *bias, float *out, int B, int Cin, int D,
                                int H, int W, ddpm_stream_t stream);
/* ConvTranspose3d(k = 4, stride 2, padding 1) as EIGHT stride-1 3x3x3 convolutions over the input grid, one per output parity
 * (qz, qy, qx): out[2j] = w[1] x[j] + w[3] x[j - 1], out[2j + 1] = w[2] x[j] + w[0] x[j + 1] per axis, i.e. 3-tap kernels
 * (w[3], w[1], 0) and (0, w[2], w[0]) -- so that the VQ-VAE decoder's largest up-convolution (256 -> 256, 32^3 -> 64^3; reference
 * call site src/trainers/reconstruct.py:166) runs on the split-f16 F(4x4) kernel (ddpm_conv_desc.depth_taps = 3 for qz = 0, 6 for
 * qz = 1) instead of the fp32-MFMA transposed kernel.  g: [8][Cout, Cin, 3, 3, 3] from the torch weight w [Cin, Cout, 4, 4, 4];
 * parity q = 4 qz + 2 qy + qx.  ddpm_parity_interleave3_f32: dst[pl, 2z + qz, 2y + qy, 2x + qx] = src[q][pl, z, y, x].  */
int ddpm_convtr3d_parity_weights_f32(const float *w, float *g, int Cin, int Cout, ddpm_stream_t stream);
int ddpm_parity_interleave3_f32(const float *src, float *dst, int64_t planes, int D, int H, int W, ddpm_stream_t stream);

/* Pack `ksize*ksize` taps starting at `tap_off` out of the `src_taps` taps of a wider torch kernel, e.g. depth
 * tap kd of a conv3d weight [Cout, Cin, 3, 3, 3]: src_taps = 27, tap_off = 9 * kd.                       */
int ddpm_pack_conv_weight_taps_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize, int src_taps,
                                   int tap_off, ddpm_stream_t stream);

/* Winograd-domain form of a [Cout, Cin, 3, 3] weight (see ddpm_conv_desc.w_wino): 16 * Cout * Cin floats.  */
size_t ddpm_wino_weight_floats(int Cout, int Cin);
int ddpm_pack_wino_weight_f32(const float *w_raw, float *w_wino, int Cout, int Cin, ddpm_stream_t stream);

/* F(4x4, 3x3) Winograd-domain form of a [Cout, Cin, 3, 3] weight (see ddpm_conv_desc.w_wino44): 36 * Cout * Cin floats
 * (0 if the channel counts have no tiling).                                                              */
size_t ddpm_wino44_weight_floats(int Cout, int Cin);
int ddpm_pack_wino44_weight_f32(const float *w_raw, float *w_wino44, int Cout, int Cin, ddpm_stream_t stream);
/* Split-f16 form of the same (see ddpm_conv_desc.w_wino44h): 2 * 36 * Cout * Cin + 64 f16 values (0: no tiling -- Cout % 64
 * or Cin % 16).  Replaces the weight operand of F.conv2d inside DiffusionModelUNet's ResnetBlocks
 * (/root/reference/src/trainers/reconstruct.py:151-153).  */
size_t ddpm_wino44h_weight_halves(int Cout, int Cin);
int ddpm_pack_wino44h_weight(const float *w_raw, uint16_t *w_wino44h, int Cout, int Cin, ddpm_stream_t stream);
/* The same for a [Cout, Cin, 3, 3, 3] weight (dims = 3 descriptors: the VQ-VAE residual units, nn.Conv3d inside generative's
 * VQVAE, /root/reference/src/trainers/reconstruct.py:124,166): one slab per depth tap, 3 * (halves - 64) + 64 f16 values.  */
int ddpm_pack_wino44h_weight3d(const float *w_raw, uint16_t *w_wino44h, int Cout, int Cin, ddpm_stream_t stream);

/* Downsample convolutions (3x3, stride 2, padding 1, 2-D, plain: bias only -- generative's Downsample between the levels of the
 * down path, reference call site src/trainers/reconstruct.py:151-153): the [Cout, Cin, 3, 3] weight as split-f16 planes for the
 * direct f16-MFMA kernel (conv_s2h.hip; Cin % 8 == 0, Cout % 64 == 0, even input extent): Cout * Cin * 18 + 64 f16 values,
 * carried in ddpm_conv_desc.w_wino44h of a DDPM_CONV_STRIDE2 descriptor (the field has no other meaning in that mode).
 * Same arithmetic as ddpm_pack_wino44h_weight's kernel: four exact f16 partial products per fp32 product, fp32 accumulate.
 * DDPM_DOWN_S2H=0 keeps such descriptors on the fp32 MFMA kernel.  */
size_t ddpm_conv_s2h_weight_halves(int Cout, int Cin);
int ddpm_pack_conv_s2h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, ddpm_stream_t stream);

/* Winograd-domain form of a [Cout, Cin, 3, 3, 3] conv3d weight: U_kd = G w[:, :, kd] G^T for each depth tap (3 * 16 * Cout *
 * Cin floats).  A dims = 3, stride-1 descriptor without GroupNorm / activation prologue (the VQ-VAE residual units) that
 * carries it in w_wino runs as 2-D Winograd F(2x2, 3x3) per depth tap, the taps accumulated in the transform domain.     */
int ddpm_pack_wino3d_weight_f32(const float *w_raw, float *w_wino, int Cout, int Cin, ddpm_stream_t stream);
/* The same for F(4x4, 3x3) (3 * 36 * Cout * Cin floats), carried in w_wino44: slices of at least 32 4x4 tiles (32 x 16 pixels
 * and up: the 32^3 and 64^3 levels of the README VQ-VAE, reference call site src/trainers/reconstruct.py:166) run as 2-D
 * F(4x4, 3x3) per depth tap -- 4x fewer multiplies than the direct form.  Takes precedence over w_wino.                      */
int ddpm_pack_wino44_weight3d_f32(const float *w_raw, float *w_wino44, int Cout, int Cin, ddpm_stream_t stream);

/* Folded form of an Upsample conv weight (see ddpm_conv_desc.w_folded): 4 packed 2x2-tap weights.     */
size_t ddpm_folded_upsample_weight_floats(int Cout, int Cin);
int ddpm_fold_upsample_weight_f32(const float *w_raw, float *w_folded, int Cout, int Cin, ddpm_stream_t stream);

/* GroupNorm statistics -> per-(image, channel) scale/shift so that
 * y = x * scale + shift == F.group_norm(x, G, gamma, beta, eps).  cat(in1, in2) virtual. */
int ddpm_gn_scale_shift_f32(const float *in1, const float *in2, int C1, int C2, const float *gamma,
                            const float *beta, float *scale, float *shift, int B, int HW, int groups,
                            float eps, ddpm_stream_t stream);

/* The same scale / shift from per-channel statistics slabs (ddpm_conv_desc.stats_out of the producing convolutions, or
 * ddpm_channel_stats_f32): st1 is [B, C1, parts1, 2], st2 [B, C2, parts2, 2] or NULL (C2 = 0); every entry covers
 * HW / parts pixels.  One thread per (image, group) merges the entries in a fixed order (mean, then squared deviations
 * about it): no pass over the activation.  */
int ddpm_gn_finalize_f32(const float *st1, int parts1, int C1, const float *st2, int parts2, int C2, const float *gamma,
                         const float *beta, float *scale, float *shift, int B, int HW, int groups, float eps,
                         ddpm_stream_t stream);
/* Per-channel statistics slab [B, C, 1, 2] = {mean, sum of squared deviations} of a [B, C, HW] tensor whose producer does
 * not emit them (one wave per (image, channel), two passes over registers).  */
int ddpm_channel_stats_f32(const float *in, float *stats, int B, int C, int HW, ddpm_stream_t stream);

/* Self-attention core of AttentionBlock (A.3): qkv is [B, 3C, N] (q rows, then k, then v,
 * channel-major exactly as a 1x1 conv over NCHW produces them); out = softmax(scale q^T k) v
 * + residual, written as [B, C, N].  Replaces torch.baddbmm / softmax / torch.bmm.  fp32 in and out; the two
 * contractions multiply on the f16 MFMA with every fp32 product rebuilt from three f16 products (22 mantissa bits,
 * fp32 accumulate) unless DDPM_ATTN_F16X3=0 (f32 MFMA, bit-exact fp32 products).                              */
int ddpm_attention_f32(const float *qkv, const float *residual, float *out, int B, int C, int N,
                       int num_heads, float scale, ddpm_stream_t stream);

/* get_timestep_embedding: out[b, :half] = cos(t_b * freqs), out[b, half:] = sin(...).   */
int ddpm_timestep_embedding_f32(const int64_t *timesteps, const float *freqs, float *out, int B, int dim,
                                ddpm_stream_t stream);

/* scheduler.add_noise (src/trainers/reconstruct.py:143-147):
 * out = sqrt_ac[b] * (x0 * b_scale) + sqrt_1m_ac[b] * noise ; coefficients are HOST arrays. */
int ddpm_add_noise_f32(const float *x0, const float *noise, const float *h_sqrt_ac, const float *h_sqrt_1m_ac,
                       float b_scale, float *out, int B, int64_t chw, ddpm_stream_t stream);

/* PNDMScheduler.step_plms + _get_prev_sample (src/trainers/reconstruct.py:155-157).
 * eps' = combination `kind` of up to four eps tensors (newest first):
 *   0: e0            1: (e0 + e1) / 2          2: (3 e0 - e1) / 2
 *   3: (23 e0 - 16 e1 + 5 e2) / 12             4: (1/24)(55 e0 - 59 e1 + 37 e2 - 9 e3)
 * v-prediction (v_a != 0 or v_b != 0... flag): eps' = v_a * eps' + v_b * sample
 * prev = sample_coeff * sample - (coef_eps * eps') / denom                                  */
int ddpm_plms_step_f32(const float *sample, const float *e0, const float *e1, const float *e2, const float *e3,
                       int kind, int v_prediction, float v_a, float v_b, float sample_coeff, float coef_eps,
                       float denom, float *prev, int64_t numel, ddpm_stream_t stream);

/* recon = clamp(recon * inv_b_scale... (recon / b_scale), 0, 1) in place and
 * mse[b] = mean((orig - recon)^2) (src/trainers/reconstruct.py:167-168,188-191).          */
int ddpm_clamp_mse_f32(const float *orig, float *recon, float b_scale, float *mse, int B, int64_t chw,
                       ddpm_stream_t stream);

/* VQ-VAE quantiser (EMAQuantizer.quantize + embedding lookup, eval path; reached from
 * src/trainers/reconstruct.py:124,166 via vqvae.decode_stage_2_outputs): for every latent vector z = x[b, :, p],
 * idx[b, p] = argmin_k |z|^2 + |e_k|^2 - 2 z.e_k (first on ties) and out[b, :, p] = z + (e_idx - z).
 * x, out: [B, D, S] channel-first; codebook [K, D]; code_norms: K floats of scratch (|e_k|^2, rewritten every call).
 * D in {8, 16, 32, 64, 128}.                                                                                  */
int ddpm_vq_nearest_f32(const float *x, const float *codebook, float *code_norms, int *idx, float *out, int B, int D,
                        int64_t S, int K, ddpm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * LPIPS-AlexNet (PerceptualLoss.forward, src/losses/perceptual_loss.py:105-186; call site
 * src/trainers/reconstruct.py:172-187).  The three 3x3 layers of AlexNet go through
 * ddpm_conv_f32 with out_act = DDPM_ACT_RELU; these cover the rest.
 * ---------------------------------------------------------------------------------- */
/* out[N,Cout,Ho,Wo] = relu?(conv2d(in * in_scale[c] + in_shift[c], w[Cout,Cin,k,k], stride, pad) + bias).
 * Cx = channels actually stored in `in` (Cin, or 1: the single plane feeds every input channel, the reference's
 * 1 -> 3 broadcast in the ScalingLayer); in_scale / in_shift may be NULL; zero padding pads the scaled input. */
int ddpm_lpips_conv_f32(const float *in, const float *w, const float *bias, const float *in_scale,
                        const float *in_shift, float *out, int N, int Cx, int Cin, int H, int W, int Cout, int k,
                        int stride, int pad, int relu, ddpm_stream_t stream);

/* The same convolution with a per-position bias: out = relu?(conv2d(in, w, stride, pad) + bias_map[Cout, Ho, Wo]).  Used for
 * the first AlexNet layer over GREY images: the ScalingLayer's 1 -> 3 broadcast and per-channel affine fold into ONE input
 * channel, conv(a_c x + b_c) = (sum_c a_c w_c) * x + (sum_c b_c w_c) * [inside the image] -- a third of the multiplies; the
 * second term (+ bias) does not depend on the image and is the bias map (it differs from a constant only at the border).  */
int ddpm_lpips_conv_biasmap_f32(const float *in, const float *w, const float *bias_map, float *out, int N, int Cin, int H,
                                int W, int Cout, int k, int stride, int pad, int relu, ddpm_stream_t stream);

/* The same-padded stride-1 k x k layers (k = 5: AlexNet's second layer, 64 -> 192; k = 3) on the fp32 MFMA pipe, for images
 * of at least 64 pixels whose Cin zero-haloed planes fit the 160 KB LDS (2.5-D LPIPS over 128^3 volumes runs the 5 x 5
 * layer over 15 x 15 maps 2 x 128 x 3 times per volume and t-start).  w_packed: Cout * Cin * k * k floats written by
 * ddpm_lpips_pack_conv_weight_f32 from the torch [Cout, Cin, k, k] layout (Cout % 32 == 0, Cin % 2 == 0).
 * out[N,Cout,H,W] = relu?(conv2d(in[N,Cin,H,W], w, padding = k / 2) + bias).                                             */
int ddpm_lpips_conv_mfma_supported(int Cin, int H, int W, int Cout, int k);
int ddpm_lpips_pack_conv_weight_f32(const float *w, float *w_packed, int Cout, int Cin, int k, ddpm_stream_t stream);
int ddpm_lpips_conv_mfma_f32(const float *in, const float *w_packed, const float *bias, float *out, int N, int Cin, int H,
                             int W, int Cout, int k, int relu, ddpm_stream_t stream);

/* MaxPool2d(kernel 3, stride 2, no padding) over `planes` = N * C planes of H x W. */
int ddpm_maxpool3s2_f32(const float *in, float *out, int64_t planes, int H, int W, ddpm_stream_t stream);

/* One LPIPS layer: out[n] (+)= mean_hw( sum_c lin[c] * (f0[n,c]/(|f0[n,:]| + 1e-10) - f1[n,c]/(|f1[n,:]| + 1e-10))^2 ). */
int ddpm_lpips_layer_f32(const float *f0, const float *f1, const float *lin, float *out, int N, int C, int HW,
                         int accumulate, ddpm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * UNet engine: DiffusionModelUNet(x, timesteps) as one native call
 * (ctor kwargs: src/trainers/base.py:66-86; call: src/trainers/reconstruct.py:151-153).
 * ---------------------------------------------------------------------------------- */
#define DDPM_MAX_LEVELS 8

typedef struct ddpm_unet_config {
  int spatial_dims;                    /* 2, or 3 for the latent-diffusion UNet (NCDHW)   */
  int in_channels, out_channels;
  int num_levels;
  int num_channels[DDPM_MAX_LEVELS];
  int attention_levels[DDPM_MAX_LEVELS];
  int num_res_blocks[DDPM_MAX_LEVELS];
  int num_head_channels[DDPM_MAX_LEVELS];
  int norm_num_groups;
  float norm_eps;
  int use_proj_attn;                   /* SURVEY A.3 open point; default 0                */
} ddpm_unet_config;

typedef struct ddpm_unet ddpm_unet;

ddpm_unet *ddpm_unet_create(const ddpm_unet_config *cfg);
void ddpm_unet_destroy(ddpm_unet *h);

/* Parameter storage: the caller allocates `ddpm_unet_param_blob_floats` floats on the device,
 * binds them, then pushes every state_dict tensor by its MONAI-Generative key name
 * (SURVEY A.5); conv / linear weights are re-laid-out for the MFMA kernels on the device.  */
size_t ddpm_unet_param_blob_floats(const ddpm_unet *h);
int ddpm_unet_bind_param_blob(ddpm_unet *h, float *blob);
int ddpm_unet_num_params(const ddpm_unet *h);
const char *ddpm_unet_param_name(const ddpm_unet *h, int i);
int64_t ddpm_unet_param_numel(const ddpm_unet *h, int i);
int ddpm_unet_set_param(ddpm_unet *h, const char *name, const float *src, int64_t numel, ddpm_stream_t stream);
/* "freqs" (timestep-embedding frequency table, [num_channels[0] / 2]) is pushed through
 * ddpm_unet_set_param too; it is computed by the host exactly as the reference does.      */

size_t ddpm_unet_workspace_bytes(const ddpm_unet *h, int B, int H, int W);
/* spatial_dims == 3 (LDM latents, NCDHW): same calls with an explicit depth */
size_t ddpm_unet_workspace_bytes3d(const ddpm_unet *h, int B, int D, int H, int W);
int ddpm_unet_forward3d(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B, int D, int H,
                        int W, void *workspace, size_t workspace_bytes, ddpm_stream_t stream);
int ddpm_unet_forward(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B, int H, int W,
                      void *workspace, size_t workspace_bytes, ddpm_stream_t stream);

/* Same call, replayed from a captured hipGraph (the launch-bound small-batch regime,
 * BASELINE configs[0]).  The 1st call with a given (x, timesteps, out, workspace, B, D, H, W) runs eagerly, the 2nd
 * captures the launch sequence, later ones are ONE hipGraphLaunch on a private stream ordered against `stream` by
 * events.  The caller must keep the four addresses stable and must not free them while the engine lives; rebinding
 * or updating parameters drops every captured graph.  ddpm_unet_num_graphs: instantiated graphs (for tests).      */
int ddpm_unet_forward_graphed(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B, int D, int H,
                              int W, void *workspace, size_t workspace_bytes, ddpm_stream_t stream);
int ddpm_unet_num_graphs(const ddpm_unet *h);

/* ------------------------------------------------------------------------------------
 * In-situ kernel timing (bench.py's roofline leg).  While enabled, every kernel launch of
 * this library is bracketed by hipEvents on the launch stream; the report synchronises those
 * events and writes one JSON object {"kernel": {"launches", "ms", "flops", "bytes"}, ...}
 * (algorithmic FLOPs / bytes per DESIGN.md) into buf.  Returns bytes written or < 0.
 * ---------------------------------------------------------------------------------- */
int ddpm_prof_enable(int on);
int ddpm_prof_report(char *buf, size_t cap);

/* ------------------------------------------------------------------------------------
 * Attention with caller scratch (ABI 8).  Same op as ddpm_attention_f32 -- generative's AttentionBlock inside
 * DiffusionModelUNet.forward, call site src/trainers/reconstruct.py:151-153: softmax(q k^T * scale) v (+ residual) over
 * qkv [B, 3 C, N], head dim 256 --; with `scratch` of at least ddpm_attention_scratch_floats(B, C, N, heads) floats (16-byte
 * aligned; 0 = this shape has no such form: N must be a multiple of 64) q, k and v are first split into MFMA-ready f16
 * planes there and the register-resident kernel of csrc/attention_fa.hip runs (one barrier per 32-key block, K / V by
 * LDS-DMA, scores and output never leave registers).  scratch = NULL: exactly ddpm_attention_f32.
 * ---------------------------------------------------------------------------------- */
size_t ddpm_attention_scratch_floats(int B, int C, int N, int num_heads);
int ddpm_attention_ws_f32(const float *qkv, const float *residual, float *out, int B, int C, int N, int num_heads,
                          float scale, float *scratch, size_t scratch_floats, ddpm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Numeric guard of the split-f16 kernel families (ABI 8).
 *
 * The reference computes this path in fp32 / fp16-autocast ATen ops (src/trainers/reconstruct.py:129,151-157); a
 * non-finite value there lands as NaN in its CSV.  Here the MFMA products of the 3x3 / 1x1 convolutions and of
 * attention are split-f16: exact for operands inside the f16 exponent range, inf beyond it
 * (F(4x4) input patches above ~80 in the worst case, ~600 typically).  So that an overflow can never reach the CSV
 * silently, the kernels every tensor of the path ends in -- the PLMS update (reads every eps), clamp + MSE (reads every
 * reconstruction), the VQ-VAE quantiser (reads every latent) -- OR a bit into a per-device status word when they meet a
 * non-finite value.  Protocol of the caller (ddpm_ood_amd/trainer.py::get_scores): read + clear after each batch; if set
 * while the split-f16 kernels are on, run the batch again after ddpm_set_split_f16(0); a word that is still set then is
 * a genuine fp32 overflow and the NaN is written like the reference would.
 * ---------------------------------------------------------------------------------- */
#define DDPM_STATUS_NONFINITE_EPS 1u    /* ddpm_plms_step_f32 read a non-finite model output           */
#define DDPM_STATUS_NONFINITE_RECON 2u  /* ddpm_clamp_mse_f32 read a non-finite reconstruction          */
#define DDPM_STATUS_NONFINITE_LATENT 4u /* ddpm_vq_nearest_f32 read a non-finite latent                 */
#define DDPM_STATUS_NONFINITE_GRAD 8u   /* ddpm_scale_check_f32 read a non-finite gradient (training)   */
/* Copies the current device's status word to *word (host memory), clears it if `clear`, and synchronises `stream`. */
int ddpm_status_read(unsigned *word, int clear, ddpm_stream_t stream);
/* ABI 9.  The quantiser is the one discontinuous op of the path (the reference re-quantises the denoised latent in
 * vqvae.decode_stage_2_outputs, src/trainers/reconstruct.py:166): where the two nearest codes lie within 1e-5 (relative) of
 * each other, a latent that differs in its 6th digit -- another machine's convolution rounding, the reference's own included --
 * may pick the other code, an O(1) change of that position's decoded block.  ddpm_vq_nearest_f32 counts such positions in a
 * per-device counter; this copies it to *count (host memory), clears it if `clear`, and synchronises `stream`.  It never
 * triggers the fp32 re-run: it tells the caller how close a run came to a flip (trainer.py: last_stats["vq_near_ties"]). */
int ddpm_vq_near_ties_read(unsigned *count, int clear, ddpm_stream_t stream);
/* on = 0: every split-f16 kernel family runs its fp32-MFMA form (bit-exact fp32 products) whatever the DDPM_*_F16X3
 * environment switches say; on = 1: back to what the environment selects (default: split-f16).  Returns the previous
 * setting.  Process-wide; not thread-safe against concurrent launches. */
int ddpm_set_split_f16(int on);
/* The master switch as ddpm_set_split_f16 set it (prev = get(); set(0); ...; set(prev) restores it).  */
int ddpm_get_split_f16(void);
/* ABI 10: 1 only when the master switch is on AND at least one split-f16 family is enabled by the environment -- "would
 * ddpm_set_split_f16(0) change which kernels run?", the question the caller's re-run guard asks (trainer.py::get_scores).
 * (ABI 9 answered it through ddpm_get_split_f16, which broke the get / set / restore idiom when every family was off.)  */
int ddpm_split_f16_active(void);
/* Parses the DDPM_* environment switches again (they are read once per process otherwise). */
int ddpm_reload_env(void);

/* ------------------------------------------------------------------------------------
 * Training step (ABI 10).  Replaces, for the epsilon-MSE step of
 * /root/reference/src/trainers/ddpm_trainer.py:78-109 over generative's DiffusionModelUNet with
 * torch.optim.Adam(lr = 2.5e-5) (/root/reference/src/trainers/base.py:156), what loss.backward() and
 * optimizer.step() dispatch to (cuDNN / cuBLAS weight and data gradients, ATen GroupNorm / SiLU / softmax backward,
 * the foreach Adam kernels).  The forward of a step runs on ddpm_conv_f32 over materialised GroupNorm + SiLU outputs;
 * input gradients of convolutions are ddpm_conv_f32 with ddpm_conv_weight_rot180t_f32's weights.  Everything below is
 * deterministic (fixed-order reductions, no atomics).  ddpm_ood_amd/train_native.py drives them.
 * ---------------------------------------------------------------------------------- */

/* C[z] = alpha * A[z] B[z] + beta * C[z] on the fp32 MFMA, every operand addressed by ELEMENT strides (a transpose is a
 * stride swap): A(z, m, k) = A[a_batch_outer z0 + a_batch z1 + a_m m + a_k_outer k0 + a_k k1] with k = k0 k_inner + k1
 * (k_inner = 0: one level, k1 = k) and z = z0 batch_inner + z1 (batch_inner = 0: one level, z1 = z); B(z, k, n), C(z, m, n)
 * alike.  One call covers a Linear's weight / input gradient, a 1x1 convolution's weight gradient over NCHW tensors
 * (K = (image, pixel): k_inner = HW), and the batched Q^T K, V P^T, dO^T V, dS K, dS^T Q products of AttentionBlock on
 * channel-major [B, C, N] tensors.  batch <= 65 535.  */
typedef struct ddpm_gemm_desc {
  const float *A, *B;
  float *C;
  int M, N, K, k_inner;
  int64_t a_m, a_k, a_k_outer;
  int64_t b_n, b_k, b_k_outer;
  int64_t c_m, c_n;
  int batch, batch_inner;
  int64_t a_batch, a_batch_outer, b_batch, b_batch_outer, c_batch, c_batch_outer;
  float alpha, beta;
  /* Optional: ddpm_gemm_scratch_floats(g) floats.  A product whose (M, N, batch) grid leaves most of the chip idle while K is
   * long (a 1x1 convolution's weight gradient) is cut into K slices, one workgroup each, whose partial sums go here and are
   * added in slice order by a second pass (deterministic).  NULL / too small: one workgroup walks the whole K range.  */
  float *scratch;
  size_t scratch_floats;
  /* != 0: the caller vouches that both operands sit in the f16 exponent range (|v| < 65 504; full precision for |v| >= 4e-3,
   * graceful below) -- the product of two K-major, 16-byte aligned operands then multiplies on the f16 MFMA at split precision
   * (three f16 products per fp32 product, fp32 accumulate: ~2^-22 relative per product).  The training step sets it for the
   * 1x1 weight gradients of a backward that runs under its gradient scale.  Ignored for other layouts and under
   * ddpm_set_split_f16(0).  */
  int split_f16;
  int reserved0;
} ddpm_gemm_desc;
size_t ddpm_gemm_scratch_floats(const ddpm_gemm_desc *g);
int ddpm_gemm_f32(const ddpm_gemm_desc *g, ddpm_stream_t stream);

/* Weight gradient of F.conv2d(a, w, stride, padding = ksize / 2): dw[Cout, Cin, k, k] (torch layout, overwritten) =
 * sum over images and output pixels of dy[b, co, p] a[b, ci, stride p + tap - pad].  a: [B, Cin, Hi, Wi] (the convolution's
 * input as it was multiplied: after GroupNorm / SiLU / concat / upsampling), dy: [B, Cout, Ho, Wo].  ksize 3 with
 * Cin % 64 == 0, Cout % 64 == 0 and an even Wo <= 64 runs on the matrix pipe (64 couts x 64 cins x 9 taps per workgroup, the
 * pixel stream split over workgroups into `scratch` -- ddpm_conv_wgrad_scratch_floats floats, 0 = no such tiling -- and
 * reduced in a fixed order: bit-reproducible): stride 1 with Wo in {8, 16, 32, 64} and 16-byte aligned tensors on the f16 MFMA
 * at split precision (three f16 products per fp32 product, both operands rescaled by a power of two from their maxima, which
 * are measured on the device into the head of `scratch`; within 3e-6 of float64 relative to the gradient's largest element, a
 * non-finite operand gives a non-finite result; DDPM_WGRAD_F16X3=0 or ddpm_set_split_f16(0) select the fp32 MFMA), every other
 * tiled shape on the fp32 MFMA; anything else (ksize 1; the 1- / 3-channel first and last convolutions) one workgroup per
 * (cout, cin) pair and image slice.  force_generic != 0: always the latter, unsliced (tests).  a_absmax / dy_absmax (optional):
 * *_n float bit patterns whose largest is the largest |a| / |dy| -- partial maxima from the kernel that wrote the tensor
 * (ddpm_gn_forward_f32 / ddpm_gn_backward_f32); NULL: the split form measures the tensor itself (one more read of it).  */
size_t ddpm_conv_wgrad_scratch_floats(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int ksize, int stride);
int ddpm_conv_wgrad_f32(const float *a, const float *dy, float *dw, int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo,
                        int ksize, int stride, float *scratch, size_t scratch_floats, int force_generic, const unsigned *a_absmax,
                        int a_absmax_n, const unsigned *dy_absmax, int dy_absmax_n, ddpm_stream_t stream);
/* The same for F.conv3d(a, w[Cout, Cin, 3, 3, 3], stride, padding = 1) on NCDHW tensors (the 3-D latent UNet of the LDM
 * configuration): one launch of the 3x3 kernel per depth tap, an "image" = (batch item, output slice).  Needs Cin % 64 == 0,
 * Cout % 64 == 0, an even Wo <= 64 and the scratch of ddpm_conv3d_wgrad_scratch_floats.  */
size_t ddpm_conv3d_wgrad_scratch_floats(int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int stride);
int ddpm_conv3d_wgrad_f32(const float *a, const float *dy, float *dw, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho,
                          int Wo, int stride, float *scratch, size_t scratch_floats, ddpm_stream_t stream);
/* wt[Cin, Cout, taps] = w[Cout, Cin, taps] with the taps in reverse order (taps = k^2 or k^3: the kernel rotated by 180 degrees
 * about every axis) and the channel axes transposed: conv(dy, wt, padding = k / 2) is the input gradient of a stride-1
 * convolution (of a stride-2 one after ddpm_resample2_f32 / ddpm_resample3_f32(mode 2) of dy).  */
int ddpm_conv_weight_rot180t_f32(const float *w, float *wt, int Cout, int Cin, int taps, ddpm_stream_t stream);

/* F.group_norm in its training form.  mean_rstd: [B, groups, 2] = {mean, 1 / sqrt(var + eps)} (biased variance), kept for the
 * backward; ddpm_gn_apply_f32: y = act((x - mean) rstd gamma + beta), act = DDPM_ACT_NONE or DDPM_ACT_SILU;
 * ddpm_gn_backward_f32: dx (+= if accumulate_dx) the gradient of that through act and the normalisation, dgamma / dbeta [C]
 * overwritten; ws: B * C * 2 floats of scratch; at most 64 channels per group.  x, y, dy, dx: [B, C, HW].  */
int ddpm_gn_stats_f32(const float *x, float *mean_rstd, int B, int C, int HW, int groups, float eps, ddpm_stream_t stream);
int ddpm_gn_apply_f32(const float *x, const float *mean_rstd, const float *gamma, const float *beta, float *y, int B, int C, int HW,
                      int groups, int act, ddpm_stream_t stream);
/* ddpm_gn_stats_f32 + ddpm_gn_apply_f32 in one call (one kernel, x read once, for planes of up to 1 024 values in groups of up to
 * 16 channels; the two launches otherwise): y and mean_rstd both written.  y_absmax (optional, B * groups words): the float bit
 * pattern of the largest |y| of each (image, group) -- what ddpm_conv_wgrad_f32 takes as a_absmax when y is the convolution's
 * input.
 * ddpm_gn_backward_f32's optional outputs, of the FINAL dx (after the accumulation when accumulate_dx): dx_absmax (B * groups words,
 * as above: ddpm_conv_wgrad_f32's dy_absmax when dx is the gradient of the preceding convolution's output) and dx_rowsum
 * ([B, C]: the sum of each plane -- that convolution's bias gradient before the sum over images, and the gradient of the time
 * embedding it adds).  */
int ddpm_gn_forward_f32(const float *x, const float *gamma, const float *beta, float *y, float *mean_rstd, unsigned *y_absmax, int B,
                        int C, int HW, int groups, float eps, int act, ddpm_stream_t stream);
int ddpm_gn_backward_f32(const float *x, const float *dy, const float *mean_rstd, const float *gamma, const float *beta, float *dx,
                         int accumulate_dx, float *dgamma, float *dbeta, float *ws, unsigned *dx_absmax, float *dx_rowsum, int B,
                         int C, int HW, int groups, int act, ddpm_stream_t stream);

/* out[r] = sum of row r of a [rows, cols] matrix (bias gradients: rows = (image, channel) planes);
 * out[c] (+= if accumulate) alpha * sum over rows of in[r * row_stride + c], in a fixed order (interleaved row groups).  */
int ddpm_row_sum_f32(const float *in, float *out, int64_t rows, int cols, ddpm_stream_t stream);
int ddpm_col_sum_f32(const float *in, float *out, int rows, int cols, int64_t row_stride, float alpha, int accumulate,
                     ddpm_stream_t stream);
int ddpm_silu_f32(const float *x, float *y, int64_t n, ddpm_stream_t stream);
int ddpm_silu_backward_f32(const float *x, const float *dy, float *dx, int64_t n, ddpm_stream_t stream);
/* out = alpha a + beta b (b may be NULL; out may alias a or b)  */
int ddpm_axpby_f32(const float *a, const float *b, float *out, float alpha, float beta, int64_t n, ddpm_stream_t stream);
/* x *= alpha in place (x 16-byte aligned); a non-finite value of x sets DDPM_STATUS_NONFINITE_GRAD in the device status word.  The
 * training step runs its backward on a gradient scaled by a power of two (the input gradients multiply on the f16 MFMA, where an
 * unscaled gradient of 1e-6 is subnormal) and takes the factor out of the flat gradient buffer with this call; a set bit means
 * the scale overflowed f16 somewhere: the caller lowers it and runs the backward again (train_native.py).  */
int ddpm_scale_check_f32(float *x, float alpha, int64_t n, ddpm_stream_t stream);
/* dst[b, cdst0 + c, :] (+= if accumulate) src[b, csrc0 + c, :], c < C: torch.cat in the forward, its split in the backward  */
int ddpm_chan_copy_f32(const float *src, float *dst, int B, int C, int Csrc, int csrc0, int Cdst, int cdst0, int HW, int accumulate,
                       ddpm_stream_t stream);
/* planes of H x W (the SMALL extent) <-> 2H x 2W.  mode 0: nearest x2 (F.interpolate of generative's Upsample); 1: its adjoint,
 * out[y, x] = sum of in's 2x2 block; 2: zero-stuffing, out[2y, 2x] = in[y, x], 0 elsewhere (a stride-2 convolution's dy).  */
int ddpm_resample2_f32(const float *in, float *out, int64_t planes, int H, int W, int mode, ddpm_stream_t stream);
/* the same three maps on volumes of D x H x W (the SMALL extent) <-> 2D x 2H x 2W  */
int ddpm_resample3_f32(const float *in, float *out, int64_t planes, int D, int H, int W, int mode, ddpm_stream_t stream);
/* softmax over each row in place; its backward ds = p (dp - sum(dp p)) in place over dp  */
int ddpm_softmax_rows_f32(float *s_inout, int64_t rows, int cols, ddpm_stream_t stream);
int ddpm_softmax_backward_rows_f32(const float *p, float *dp_inout, int64_t rows, int cols, ddpm_stream_t stream);
/* F.mse_loss(pred, target): dpred = grad_scale (pred - target) (grad_scale = 2 / n), partial[i] = sum of (pred - target)^2
 * over elements [256 i, 256 i + 256): the caller sums ceil(n / 256) partials (ddpm_col_sum_f32) and divides by n.  */
int ddpm_mse_loss_grad_f32(const float *pred, const float *target, float *dpred, float *partial, int64_t n, float grad_scale,
                           ddpm_stream_t stream);
int ddpm_fill_f32(float *out, float value, int64_t n, ddpm_stream_t stream);
/* torch.randn for the training noise (ddpm_trainer.py:81): standard normals from Philox-4x32-10 + Box-Muller, a pure function
 * of (seed, stream_id, element index).  Not torch's generator stream -- the noise of a training step only has to be
 * reproducible.  */
int ddpm_randn_f32(float *out, int64_t n, uint64_t seed, uint64_t stream_id, ddpm_stream_t stream);
/* torch.optim.Adam.step (no weight decay, no amsgrad) over one flat buffer; step = 1, 2, ...; g is multiplied by grad_scale first
 * (1 / world size after a sum all-reduce).  */
int ddpm_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
                       int step, float grad_scale, ddpm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DDPM_OOD_HIP_H */
