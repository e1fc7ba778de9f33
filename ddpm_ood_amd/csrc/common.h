// Shared host/device helpers for libddpm_ood_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/ddpm_ood_hip.h"

namespace ddpm {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(ddpm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define DDPM_CHECK_ARG(cond, ...)                \
  do {                                           \
    if (!(cond)) {                               \
      ::ddpm::set_error(__VA_ARGS__);            \
      return DDPM_EINVAL;                        \
    }                                            \
  } while (0)

#define DDPM_CHECK_LAUNCH()                                                    \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      ::ddpm::set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return (int)e_;                                                          \
    }                                                                          \
  } while (0)

constexpr int kWave = 64;

// ---- run-time switches (api.hip) ----------------------------------------------------------------
// The DDPM_* environment switches of DESIGN 4.1 that are looked at per launch, parsed ONCE (ddpm_reload_env() parses them
// again: the tests flip them inside one process) instead of 3-4 getenv calls per convolution, plus the run-time master
// switch of the split-f16 kernel families (ddpm_set_split_f16).
struct Switches {
  int conv_wino44 = 1;          // DDPM_CONV_WINO44: 0 no F(4x4) kernels, 2 also for launches smaller than the chip
  bool wino44_f16x3 = true;     // DDPM_WINO44_F16X3
  int wino44_split = 4;         // DDPM_WINO44_SPLIT
  int wino_split = 8;           // DDPM_WINO_SPLIT: most channel-stream splits of a conv_wino.hip launch smaller than half the chip (power of two)
  int wino44_xmap = -1;         // DDPM_WINO44_XMAP (-1: unset, each kernel has its own default)
  int w44_abl = 0;              // DDPM_W44_ABL
  int w44r_serp = 0;            // DDPM_W44R_SERP: 1 consecutive launches of conv_wino44r.hip walk their items in alternating order (measured: +-0)
  int w44h_xitem = 1;           // DDPM_W44H_XITEM: 0 every item of conv_wino44h.hip refills its pixel ring from scratch (A/B)
  bool up_wino44h = true;       // DDPM_UP_WINO44H
  int down_s2h = 1;             // DDPM_DOWN_S2H (0 off, 2 / 3 force a form)
  bool conv1x1_f16x3 = true;    // DDPM_CONV1X1_F16X3
  bool attn_f16x3 = true;       // DDPM_ATTN_F16X3
  int d1s_maxpx = 16384;        // DDPM_D1S_MAXPX: pixels per launch up to which a 1x1 takes the one-shot kernel (conv_d3s.hip)
  int conv_d3s = 1;             // DDPM_CONV_D3S: 0 never the one-shot small-launch 3x3 kernel (conv_d3s.hip), 2 for any launch size (tests)
  int attn_fa = 1;              // DDPM_ATTN_FA (0: the LDS-exchange kernels of attention.hip also when scratch is given; 2: the
                                // register-resident kernel for every multiple of 64 tokens, not only from 1 024)
  bool conv_splitk = true;      // DDPM_CONV_SPLITK
  bool gn_fused = true;         // DDPM_GN_FUSED
  bool attn_waves8 = true;      // DDPM_ATTN_WAVES: 4 selects the four-wave split-f16 attention kernel
  bool convin_fast = true;      // DDPM_CONVIN_FAST
  bool convout_wave = true;     // DDPM_CONVOUT_WAVE
  long convout_w16_maxwg = -1;  // DDPM_CONVOUT_W16_MAXWG (-1: two workgroups per CU)
  long convin_blocks_per_cu = 8;  // DDPM_CONVIN_BLOCKS_PER_CU
  bool wgrad_f16x3 = true;      // DDPM_WGRAD_F16X3: 0 the 3x3 weight gradient of the training step stays on the fp32 MFMA
  bool prof_shapes = false;     // DDPM_PROF_SHAPES
  bool split_f16 = true;        // ddpm_set_split_f16(): false = every split-f16 family runs its fp32-MFMA form
};
const Switches &sw();
unsigned switch_epoch();  // changes whenever ddpm_reload_env / ddpm_set_split_f16 may have changed the dispatch
inline bool split_f16_on(bool family) { return family && sw().split_f16; }

// ---- device status word (api.hip) ---------------------------------------------------------------
// One 32-bit word per device, set (atomicOr, only when something is wrong) by the cheap kernels every tensor of the path
// passes through; read with ddpm_status_read.  A non-finite value here means either a genuine fp32 overflow (the
// reference would write NaN to its CSV) or an operand beyond the f16 range of a split-f16 kernel -- the caller re-runs
// the batch with ddpm_set_split_f16(0) to tell the two apart (trainer.py::get_scores).
unsigned *status_word();  // device pointer of the current device's word (allocated and zeroed on first use; NULL on failure)
__device__ __forceinline__ bool non_finite(float v) { return !(__builtin_fabsf(v) <= 3.402823466e+38f); }

// ---- in-situ profiler (api.hip) ----------------------------------------------------------------
extern bool g_prof_on;
void prof_begin(hipStream_t s, const char *kernel, double flops, double bytes);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(hipStream_t st, const char *kernel, double flops, double bytes) : s(st), on(g_prof_on) {
    if (on) prof_begin(s, kernel, flops, bytes);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
// v_exp_f32 / v_rcp_f32 form (~1e-7 relative): 6 instructions instead of ~45
__device__ __forceinline__ float silu_fast(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Pairwise (Chan) merge of per-lane {mean, M2} of `cnt` values each over aligned groups of 4 / 16 / 32 lanes on the VALU's
// DPP path (no LDS traffic): row_shr:1, 2, 4, 8 then row_bcast:15 -- lane i takes in the finished half to its left, so the
// group's statistics are valid in its LAST lane only (other lanes end with partial garbage).  Exact to rounding whatever the
// means are, fixed order -> bit-reproducible.
template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ void chan_merge_dpp(float &mean, float &m2, float &halfn) {
  const float mo = dpp_mov0<CTRL>(mean), qo = dpp_mov0<CTRL>(m2);
  const float dl = mo - mean;
  m2 = (m2 + qo) + dl * dl * halfn;
  mean = 0.5f * (mean + mo);
  halfn += halfn;
}
// plain sum of a wave's 64 values on the same DPP path (valid in lane 63 only)
__device__ __forceinline__ float wave_sum_dpp_last_lane(float v) {
  v += dpp_mov0<0x111>(v);
  v += dpp_mov0<0x112>(v);
  v += dpp_mov0<0x114>(v);
  v += dpp_mov0<0x118>(v);
  v += dpp_mov0<0x142>(v);
  v += dpp_mov0<0x143>(v);
  return v;
}
__device__ __forceinline__ void group_moments_last_lane(float &mean, float &m2, float cnt, int lanes) {  // lanes: 4, 16, 32, 64
  float halfn = 0.5f * cnt;
  chan_merge_dpp<0x111>(mean, m2, halfn);
  chan_merge_dpp<0x112>(mean, m2, halfn);
  if (lanes > 4) {
    chan_merge_dpp<0x114>(mean, m2, halfn);
    chan_merge_dpp<0x118>(mean, m2, halfn);
  }
  if (lanes > 16) chan_merge_dpp<0x142>(mean, m2, halfn);
  if (lanes > 32) chan_merge_dpp<0x143>(mean, m2, halfn);  // row_bcast:31
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- split-f16 MFMA operands (conv1x1_dma.hip has the derivation and the error / range budget) --------------
// An fp32 product is rebuilt from three v_mfma_f32_32x32x16_f16: a b ~= ah bh + (ah 2^-5)(bl 2^5) + (al 2^5)(bh 2^-5),
// fp32 accumulate; the low halves are scaled into the f16 normal range, the factor goes back on a high half.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kF16LoScale = 32.f;  // 2^5 on a low half, 2^-5 on the high half it is multiplied with

// One MFMA operand (eight fp32 values: the 32x32x16 f16 instruction's k = 8 (lane / 32) .. + 7) as three f16
// vectors: hi = f16(v), lo = f16((v - hi) 2^5) (v - hi is exact in fp32), hs = hi 2^-5
__device__ __forceinline__ void split_f16x8(const float (&v)[8], f16x8 &hi, f16x8 &lo, f16x8 &hs) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)((v[t] - (float)h) * kF16LoScale);
  }
  hs = hi * (_Float16)(1.f / kF16LoScale);
}

// acc += a b for one 32x32x16 operand pair, cross terms first
#define DDPM_MFMA_F16X3(acc, ah, al, as, bh, bl, bs)                              \
  do {                                                                            \
    (acc) = __builtin_amdgcn_mfma_f32_32x32x16_f16((as), (bl), (acc), 0, 0, 0);   \
    (acc) = __builtin_amdgcn_mfma_f32_32x32x16_f16((al), (bs), (acc), 0, 0, 0);   \
    (acc) = __builtin_amdgcn_mfma_f32_32x32x16_f16((ah), (bh), (acc), 0, 0, 0);   \
  } while (0)

// Block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.
// Deterministic: fixed butterfly inside the wave, fixed order across waves.
__device__ __forceinline__ float block_sum_256(float v, float *red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- internal launchers shared between api.hip and the UNet engine ---------------------
int conv_dispatch(const ddpm_conv_desc &d, hipStream_t s);
bool conv_mfma_supported(const ddpm_conv_desc &d);
int launch_conv_mfma(const ddpm_conv_desc &d, hipStream_t s);
int launch_conv_direct(const ddpm_conv_desc &d, hipStream_t s);
size_t packed_conv_weight_floats(int Cout, int Cin, int ksize);
size_t folded_upsample_weight_floats(int Cout, int Cin);
bool conv1x1_dma_supported(const ddpm_conv_desc &d);
int launch_conv1x1_dma(const ddpm_conv_desc &d, hipStream_t s);
size_t conv1x1_h_weight_halves(int Cout, int Cin);
int launch_pack_conv1x1_h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, int cout_offset, int Cout_total,
                                 hipStream_t s);
bool linear_skinny_supported(const ddpm_conv_desc &d);
int launch_linear_skinny(const ddpm_conv_desc &d, hipStream_t s);
bool conv_wino_supported(const ddpm_conv_desc &d);
int launch_conv_wino(const ddpm_conv_desc &d, hipStream_t s);
size_t wino_weight_floats(int Cout, int Cin);
size_t conv_wino_scratch_floats(const ddpm_conv_desc &d);
int launch_wino_split_reduce(const ddpm_conv_desc &d, int S, long long pstride, int HW, hipStream_t s);
int wino_split_reduce_stats_parts(int HW);  // slices of desc.stats_out the reduce pass writes for planes of HW floats (0: none)
size_t conv_wino44_scratch_floats(const ddpm_conv_desc &d);
size_t conv_mfma_scratch_floats(const ddpm_conv_desc &d);
size_t conv_scratch_floats(const ddpm_conv_desc &d);  // what conv_dispatch can use: the largest of the kernels' needs
bool conv_wino44_supported(const ddpm_conv_desc &d);
int launch_conv_wino44(const ddpm_conv_desc &d, hipStream_t s);
size_t wino44_weight_floats(int Cout, int Cin);
bool conv_wino44h_supported(const ddpm_conv_desc &d);
int launch_conv_wino44h(const ddpm_conv_desc &d, hipStream_t s);
size_t conv_wino44h_scratch_floats(const ddpm_conv_desc &d);
bool conv_d3s2_supported(const ddpm_conv_desc &d);
int launch_conv_d3s2(const ddpm_conv_desc &d, hipStream_t s);
size_t conv_d3s2_scratch_floats(const ddpm_conv_desc &d);
bool conv_d1s_supported(const ddpm_conv_desc &d);
int launch_conv_d1s(const ddpm_conv_desc &d, hipStream_t s);
size_t conv_d1s_scratch_floats(const ddpm_conv_desc &d);
size_t conv_d1s_weight_halves(int Cout, int Cin);
int launch_pack_conv_d1s_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, int cout_offset, int Cout_total, hipStream_t s);
bool conv_d3s_supported(const ddpm_conv_desc &d);
int launch_conv_d3s(const ddpm_conv_desc &d, hipStream_t s);
size_t conv_d3s_scratch_floats(const ddpm_conv_desc &d);
int conv_d3s_stats_parts(const ddpm_conv_desc &d);
int conv_s2h_stats_parts(const ddpm_conv_desc &d);
int device_cus();
size_t conv_d3h_weight_halves(int Cout, int Cin);
int launch_pack_conv_d3h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, hipStream_t s);
bool conv_s2h_supported(const ddpm_conv_desc &d);
int launch_conv_s2h(const ddpm_conv_desc &d, hipStream_t s);
size_t conv_s2h_weight_halves(int Cout, int Cin);
int launch_pack_conv_s2h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, hipStream_t s);
int conv_wino44h_stats_parts(const ddpm_conv_desc &d);
int conv_wino_stats_parts(const ddpm_conv_desc &d);
int conv_stats_parts(const ddpm_conv_desc &d);
size_t wino44h_weight_halves(int Cout, int Cin);
int launch_pack_wino44h_weight(const float *w_raw, uint16_t *w_wino44h, int Cout, int Cin, hipStream_t s, int nkd = 1);
int launch_pack_wino44_weight(const float *w_raw, float *w_wino44, int Cout, int Cin, hipStream_t s, int nkd = 1);
int launch_pack_wino_weight(const float *w_raw, float *w_wino, int Cout, int Cin, hipStream_t s, int nkd = 1);
int launch_fold_upsample_weight(const float *w_raw, float *w_folded, int Cout, int Cin, hipStream_t s);
int launch_pack_conv_weight(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize, int cout_offset,
                            int Cout_total, hipStream_t s, int src_taps = 0, int tap_off = 0);
size_t packed_convT_weight_floats(int Cout, int Cin, int dims);
int launch_pack_convT_weight(const float *w_raw, float *w_packed, int Cin, int Cout, int dims, hipStream_t s);
int launch_conv3d_k4s2_cin1(const float *in, const float *w, const float *bias, float *out, int B, int Cout, int D,
                            int H, int W, int relu, hipStream_t s);
int launch_convT3d_k4s2_cout1(const float *in, const float *w, const float *bias, float *out, int B, int Cin, int D,
                              int H, int W, hipStream_t s);
int launch_convnd_generic(const float *in, const float *w, const float *bias, const float *residual, float *out, int B, int Cin,
                          int Cout, int Di, int Hi, int Wi, int dims, int k, int stride, int pad, int transposed, int relu,
                          hipStream_t s);
int launch_gn_scale_shift(const float *in1, const float *in2, int C1, int C2, const float *gamma, const float *beta,
                          float *scale, float *shift, int B, int HW, int groups, float eps, hipStream_t s);
int launch_gn_finalize(const float *st1, int parts1, int C1, const float *st2, int parts2, int C2, const float *gamma,
                       const float *beta, float *scale, float *shift, int B, int HW, int groups, float eps, hipStream_t s);
int launch_channel_stats(const float *in, float *stats, int B, int C, int HW, hipStream_t s);
int launch_attention(const float *qkv, const float *residual, float *out, int B, int C, int N, int heads, float scale,
                     hipStream_t s, float *scratch = nullptr, size_t scratch_floats = 0, float *stats_out = nullptr);
bool attention_emits_stats(int B, int C, int N, int heads, const float *scratch, size_t scratch_floats);
size_t attention_fa_scratch_floats(int B, int C, int N, int heads);
bool attention_fa_supported(int B, int C, int N, int heads, const float *scratch, size_t scratch_floats);
int launch_attention_fa(const float *qkv, const float *residual, float *out, int B, int C, int N, int heads, float scale,
                        float *scratch, hipStream_t s);
int launch_timestep_embedding(const int64_t *t, const float *freqs, float *out, int B, int dim, hipStream_t s);
int launch_copy_f32(const float *src, float *dst, int64_t n, hipStream_t s);

// MFMA conv tiling constants (shared by the packer and the kernel)
constexpr int kConvCc = 4;     // input channels staged per chunk
constexpr int kConvNT = 128;   // output channels per workgroup
constexpr int kConvMT = 128;   // output pixels per workgroup

}  // namespace ddpm
