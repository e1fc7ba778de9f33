// api.hip -- error plumbing and the extern "C" wrappers of the stand-alone operators.
#include <stdarg.h>

#include "common.h"

namespace ddpm {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace ddpm

using namespace ddpm;

extern "C" int ddpm_abi_version(void) { return DDPM_ABI_VERSION; }
extern "C" const char *ddpm_last_error(void) { return g_err; }

extern "C" int ddpm_conv_f32(const ddpm_conv_desc *d, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(d != nullptr, "conv: descriptor is NULL");
  return conv_dispatch(*d, as_stream(stream));
}

extern "C" size_t ddpm_packed_conv_weight_floats(int Cout, int Cin, int ksize) {
  return packed_conv_weight_floats(Cout, Cin, ksize);
}

extern "C" int ddpm_pack_conv_weight_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize,
                                         int cout_offset, int Cout_total, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_packed, "pack: NULL pointer");
  return launch_pack_conv_weight(w_raw, w_packed, Cout, Cin, ksize, cout_offset, Cout_total, as_stream(stream));
}

extern "C" int ddpm_gn_scale_shift_f32(const float *in1, const float *in2, int C1, int C2, const float *gamma,
                                       const float *beta, float *scale, float *shift, int B, int HW, int groups,
                                       float eps, ddpm_stream_t stream) {
  return launch_gn_scale_shift(in1, in2, C1, C2, gamma, beta, scale, shift, B, HW, groups, eps, as_stream(stream));
}

extern "C" int ddpm_attention_f32(const float *qkv, const float *residual, float *out, int B, int C, int N,
                                  int num_heads, float scale, ddpm_stream_t stream) {
  return launch_attention(qkv, residual, out, B, C, N, num_heads, scale, as_stream(stream));
}
