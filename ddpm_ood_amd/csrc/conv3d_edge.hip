// conv3d_edge.hip -- the two VQ-VAE layers that have no MFMA tiling: the first encoder convolution
// (Conv3d 1 -> C, kernel 4, stride 2, pad 1) and the last decoder convolution (ConvTranspose3d C -> 1, kernel 4,
// stride 2, pad 1) of generative.networks.nets.VQVAE as configured at /root/reference/README.md:153-158 and
// reached from /root/reference/src/trainers/reconstruct.py:124 (encode_stage_2_inputs) and :166
// (decode_stage_2_outputs).  Both are HBM-bound (one side of the layer is a single-channel volume): 8.6 GFLOP
// against 268 MB of activation traffic at 128^3 / 256 channels.  Weights are wave-uniform (scalar loads), the
// 64 (27) input values of a thread stay in registers across all output (input) channels.
#include "common.h"

namespace ddpm {

// out[n, co, z, y, x] = relu?(bias[co] + sum_{kz,ky,kx} w[co, 0, kz, ky, kx] * in[n, 0, 2z+kz-1, 2y+ky-1, 2x+kx-1])
__global__ __launch_bounds__(256) void conv3d_k4s2_cin1_kernel(const float *__restrict__ in,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               int Cout, int D, int H, int W, int relu) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const int npos = Do * Ho * Wo;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pos >= npos) return;
  const int x = pos % Wo, y = (pos / Wo) % Ho, z = pos / (Wo * Ho);
  const float *src = in + (size_t)n * D * H * W;
  float v[64];
#pragma unroll
  for (int kz = 0; kz < 4; ++kz)
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int iz = 2 * z + kz - 1, iy = 2 * y + ky - 1, ix = 2 * x + kx - 1;
        const bool ok = iz >= 0 && iz < D && iy >= 0 && iy < H && ix >= 0 && ix < W;
        v[(kz * 4 + ky) * 4 + kx] = ok ? src[((size_t)iz * H + iy) * W + ix] : 0.f;
      }
  float *dst = out + (size_t)n * Cout * npos + pos;
  for (int co = 0; co < Cout; ++co) {
    const float *wc = w + (size_t)co * 64;  // wave-uniform: scalar loads
    // torch's own accumulation order is unspecified; a fixed k-ascending fmaf chain keeps the result deterministic
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) acc = fmaf(v[k], wc[k], acc);
    if (bias) acc += bias[co];
    dst[(size_t)co * npos] = relu ? fmaxf(acc, 0.f) : acc;
  }
}

// out[n, 0, 2z+pz, 2y+py, 2x+px] = bias + sum_c sum_{a,r,s in {0,1}} in[n, c, z-1+pz+a, y-1+py+r, x-1+px+s]
//                                                  * w[c, 0, 3-pz-2a, 3-py-2r, 3-px-2s]
// A thread owns one low-res position and its eight output voxels: 27 neighbour loads and 64 FMAs per channel.
__global__ __launch_bounds__(256) void convT3d_k4s2_cout1_kernel(const float *__restrict__ in,
                                                                 const float *__restrict__ w,
                                                                 const float *__restrict__ bias,
                                                                 float *__restrict__ out, int Cin, int D, int H, int W) {
  const int npos = D * H * W;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pos >= npos) return;
  const int x = pos % W, y = (pos / W) % H, z = pos / (W * H);
  int off[27];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int iz = z + dz - 1, iy = y + dy - 1, ix = x + dx - 1;
        const bool ok = iz >= 0 && iz < D && iy >= 0 && iy < H && ix >= 0 && ix < W;
        off[(dz * 3 + dy) * 3 + dx] = ok ? (iz * H + iy) * W + ix : -1;
      }
  float acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) acc[p] = 0.f;
  const float *src = in + (size_t)n * Cin * npos;
  for (int c = 0; c < Cin; ++c) {
    const float *sc = src + (size_t)c * npos;
    const float *wc = w + (size_t)c * 64;  // wave-uniform: scalar loads
    float v[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = off[i] >= 0 ? sc[off[i]] : 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int px = p & 1, py = (p >> 1) & 1, pz = p >> 2;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[p] = fmaf(v[((pz + a) * 3 + py + r) * 3 + px + s],
                          wc[((3 - pz - 2 * a) * 4 + 3 - py - 2 * r) * 4 + 3 - px - 2 * s], acc[p]);
    }
  }
  const float b = bias ? bias[0] : 0.f;
  float *dst = out + (size_t)n * 8 * npos;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int px = p & 1, py = (p >> 1) & 1, pz = p >> 2;
    dst[((size_t)(2 * z + pz) * (2 * H) + 2 * y + py) * (2 * W) + 2 * x + px] = acc[p] + b;
  }
}

int launch_conv3d_k4s2_cin1(const float *in, const float *w, const float *bias, float *out, int B, int Cout, int D,
                            int H, int W, int relu, hipStream_t s) {
  DDPM_CHECK_ARG(in && w && out && B > 0 && Cout > 0, "conv3d_k4s2_cin1: null tensor or empty shape");
  DDPM_CHECK_ARG(D >= 2 && H >= 2 && W >= 2 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && B <= 65535,
                 "conv3d_k4s2_cin1: extents must be even and >= 2");
  const long npos = (long)(D / 2) * (H / 2) * (W / 2);
  ProfScope prof(s, "conv3d_k4s2_cin1", 2.0 * B * npos * Cout * 64, 4.0 * B * ((double)D * H * W + (double)npos * Cout));
  hipLaunchKernelGGL(conv3d_k4s2_cin1_kernel, dim3((unsigned)((npos + 255) / 256), B), dim3(256), 0, s, in, w, bias, out,
                     Cout, D, H, W, relu);
  DDPM_CHECK_LAUNCH();
  return 0;
}

int launch_convT3d_k4s2_cout1(const float *in, const float *w, const float *bias, float *out, int B, int Cin, int D,
                              int H, int W, hipStream_t s) {
  DDPM_CHECK_ARG(in && w && out && B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && B <= 65535,
                 "convT3d_k4s2_cout1: null tensor or empty shape");
  const long npos = (long)D * H * W;
  DDPM_CHECK_ARG(npos * 8 < (1L << 31), "convT3d_k4s2_cout1: volume too large for 32-bit voxel offsets");
  ProfScope prof(s, "convT3d_k4s2_cout1", 2.0 * B * npos * 8 * Cin * 8, 4.0 * B * ((double)npos * Cin + 8.0 * npos));
  hipLaunchKernelGGL(convT3d_k4s2_cout1_kernel, dim3((unsigned)((npos + 255) / 256), B), dim3(256), 0, s, in, w, bias,
                     out, Cin, D, H, W);
  DDPM_CHECK_LAUNCH();
  return 0;
}

// ---- generic (transposed) convolution, 2-D or 3-D, any channel counts: the VQ-VAE layers that have no MFMA tiling (channel
// counts that are not multiples of 4 / 128).  One thread per output element, taps and input channels in a fixed order, fp32
// fmaf chain -- the slow, always-available form behind the MFMA kernels (the product never leaves the HIP library: there is no
// PyTorch-ROCm / MIOpen route).  Reference ops: Conv3d / ConvTranspose3d inside generative's VQVAE,
// /root/reference/src/trainers/reconstruct.py:124,166.
__global__ __launch_bounds__(256) void convnd_generic_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ residual, float *__restrict__ out,
                                                             int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho,
                                                             int Wo, int kd, int k, int stride, int pad, int padd,
                                                             int transposed, int relu) {
  const long long total = (long long)B * Cout * Do * Ho * Wo;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), z = (int)((i / ((long long)Wo * Ho)) % Do);
    const int co = (int)((i / ((long long)Wo * Ho * Do)) % Cout), n = (int)(i / ((long long)Wo * Ho * Do * Cout));
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float *src = in + ((size_t)n * Cin + ci) * Di * Hi * Wi;
      // torch layouts: Conv [Cout, Cin, kd, k, k]; ConvTranspose [Cin, Cout, kd, k, k]
      const float *wt = w + (transposed ? ((size_t)ci * Cout + co) : ((size_t)co * Cin + ci)) * kd * k * k;
      for (int a = 0; a < kd; ++a) {
        int iz;
        if (!transposed) {
          iz = z * (kd == 1 ? 1 : stride) - padd + a;
        } else {  // out[z] receives in[iz] w[a] with z = iz * stride - pad + a
          const int tz = z + padd - a;
          if (kd > 1 && (tz < 0 || tz % stride)) continue;
          iz = kd == 1 ? z : tz / stride;
        }
        if (iz < 0 || iz >= Di) continue;
        for (int b = 0; b < k; ++b) {
          int iy;
          if (!transposed) {
            iy = y * stride - pad + b;
          } else {
            const int ty = y + pad - b;
            if (ty < 0 || ty % stride) continue;
            iy = ty / stride;
          }
          if (iy < 0 || iy >= Hi) continue;
          for (int c = 0; c < k; ++c) {
            int ix;
            if (!transposed) {
              ix = x * stride - pad + c;
            } else {
              const int tx = x + pad - c;
              if (tx < 0 || tx % stride) continue;
              ix = tx / stride;
            }
            if (ix < 0 || ix >= Wi) continue;
            acc = __builtin_fmaf(src[((size_t)iz * Hi + iy) * Wi + ix], wt[(a * k + b) * k + c], acc);
          }
        }
      }
    }
    if (residual) acc += residual[i];
    out[i] = relu ? fmaxf(acc, 0.f) : acc;
  }
}

int launch_convnd_generic(const float *in, const float *w, const float *bias, const float *residual, float *out, int B, int Cin,
                          int Cout, int Di, int Hi, int Wi, int dims, int k, int stride, int pad, int transposed, int relu,
                          hipStream_t s) {
  DDPM_CHECK_ARG(in && w && out && B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0, "convnd_generic: null tensor or empty shape");
  DDPM_CHECK_ARG((dims == 2 || dims == 3) && k >= 1 && k <= 7 && (stride == 1 || stride == 2) && pad >= 0 && pad < k,
                 "convnd_generic: dims 2 / 3, kernel 1 .. 7, stride 1 / 2");
  DDPM_CHECK_ARG(dims == 3 || Di == 1, "convnd_generic: a 2-D convolution has depth 1");
  const int kd = dims == 3 ? k : 1, padd = dims == 3 ? pad : 0;
  auto ext = [&](int e, int kk, int pp) { return transposed ? (e - 1) * stride - 2 * pp + kk : (e + 2 * pp - kk) / stride + 1; };
  const int Do = dims == 3 ? ext(Di, k, pad) : 1, Ho = ext(Hi, k, pad), Wo = ext(Wi, k, pad);
  DDPM_CHECK_ARG(Do > 0 && Ho > 0 && Wo > 0, "convnd_generic: empty output");
  const long long total = (long long)B * Cout * Do * Ho * Wo;
  const int blocks = (int)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
  ProfScope prof(s, "convnd_generic", 2.0 * total * Cin * kd * k * k, 4.0 * ((double)B * Cin * Di * Hi * Wi + total));
  hipLaunchKernelGGL(convnd_generic_kernel, dim3(blocks), dim3(256), 0, s, in, w, bias, residual, out, B, Cin, Cout, Di, Hi, Wi,
                     Do, Ho, Wo, kd, k, stride, pad, padd, transposed, relu);
  DDPM_CHECK_LAUNCH();
  return 0;
}


// ---- ConvTranspose3d k4 s2 p1 as eight parity convolutions (ddpm_convtr3d_parity_weights_f32 in the header) -----------------
// tap t (0..2, input offset t - 1) of parity q reads kernel element: q = 0: (3, 1, none), q = 1: (none, 2, 0)
__device__ __forceinline__ int convtr_parity_tap(int q, int t) { return q == 0 ? (t == 0 ? 3 : t == 1 ? 1 : -1) : (t == 1 ? 2 : t == 2 ? 0 : -1); }

__global__ void convtr3d_parity_weights_kernel(const float *__restrict__ w, float *__restrict__ g, int Cin, int Cout) {
  const int64_t per = (int64_t)Cout * Cin * 27, total = 8 * per;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i / per);
  const int64_t r = i - q * per;
  const int tap = (int)(r % 27);
  const int ci = (int)((r / 27) % Cin), co = (int)(r / ((int64_t)27 * Cin));
  const int kz = convtr_parity_tap(q >> 2, tap / 9), ky = convtr_parity_tap((q >> 1) & 1, (tap / 3) % 3), kx = convtr_parity_tap(q & 1, tap % 3);
  g[i] = (kz < 0 || ky < 0 || kx < 0) ? 0.f : w[(((size_t)ci * Cout + co) * 4 + kz) * 16 + ky * 4 + kx];
}

// one thread per (plane, output z, output y, input x): the two x parities of an output row pair as one 8-byte store
__global__ void parity_interleave3_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t planes, int D, int H, int W) {
  const int64_t n = planes * 2 * D * 2 * H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  const int y2 = (int)((i / W) % (2 * H)), z2 = (int)((i / ((int64_t)W * 2 * H)) % (2 * D));
  const int64_t pl = i / ((int64_t)W * 2 * H * 2 * D);
  const int q = ((z2 & 1) << 2) | ((y2 & 1) << 1);
  const size_t vol = (size_t)D * H * W, qs = (size_t)planes * vol;
  const size_t so = pl * vol + ((size_t)(z2 >> 1) * H + (y2 >> 1)) * W + x;
  const float2 v = make_float2(src[q * qs + so], src[(q | 1) * qs + so]);
  *reinterpret_cast<float2 *>(dst + ((pl * 2 * D + z2) * 2 * H + y2) * (size_t)(2 * W) + 2 * x) = v;
}

}  // namespace ddpm

extern "C" int ddpm_convtr3d_parity_weights_f32(const float *w, float *g, int Cin, int Cout, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w && g && Cin > 0 && Cout > 0, "convtr3d_parity_weights: bad arguments");
  const int64_t total = (int64_t)8 * Cout * Cin * 27;
  hipLaunchKernelGGL(ddpm::convtr3d_parity_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ddpm::as_stream(stream), w, g,
                     Cin, Cout);
  DDPM_CHECK_LAUNCH();
  return 0;
}

extern "C" int ddpm_parity_interleave3_f32(const float *src, float *dst, int64_t planes, int D, int H, int W, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(src && dst && planes > 0 && D > 0 && H > 0 && W > 0, "parity_interleave3: bad arguments");
  hipStream_t s = ddpm::as_stream(stream);
  const int64_t n = planes * 4 * D * H * W;
  ddpm::ProfScope prof(s, "convT3d_parity_interleave", 0.0, 64.0 * planes * D * H * W);
  hipLaunchKernelGGL(ddpm::parity_interleave3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, planes, D, H, W);
  DDPM_CHECK_LAUNCH();
  return 0;
}
