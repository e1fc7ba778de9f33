// api.hip -- error plumbing and the extern "C" wrappers of the stand-alone operators.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace ddpm {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- run-time switches --------------------------------------------------------------------------
static Switches g_sw;
static bool g_sw_loaded = false;
static unsigned g_sw_epoch = 0;  // bumped whenever a switch may have changed: captured hipGraphs bake the dispatch in
unsigned switch_epoch() { return g_sw_epoch; }
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static void load_switches() {
  const bool keep = g_sw.split_f16;
  Switches n;
  n.conv_wino44 = env_int("DDPM_CONV_WINO44", 1);
  n.wino44_f16x3 = env_int("DDPM_WINO44_F16X3", 1) != 0;
  n.wino44_split = env_int("DDPM_WINO44_SPLIT", 4);
  n.wino_split = env_int("DDPM_WINO_SPLIT", 8);
  n.w44h_xitem = env_int("DDPM_W44H_XITEM", 1);
  n.w44r_serp = env_int("DDPM_W44R_SERP", 0);
  n.wino44_xmap = env_int("DDPM_WINO44_XMAP", -1);
  n.w44_abl = env_int("DDPM_W44_ABL", 0);
  n.up_wino44h = env_int("DDPM_UP_WINO44H", 1) != 0;
  n.down_s2h = env_int("DDPM_DOWN_S2H", 1);
  n.conv1x1_f16x3 = env_int("DDPM_CONV1X1_F16X3", 1) != 0;
  n.attn_f16x3 = env_int("DDPM_ATTN_F16X3", 1) != 0;
  n.wgrad_f16x3 = env_int("DDPM_WGRAD_F16X3", 1) != 0;
  n.attn_fa = env_int("DDPM_ATTN_FA", 1);
  n.conv_d3s = env_int("DDPM_CONV_D3S", 1);
  n.d1s_maxpx = env_int("DDPM_D1S_MAXPX", 16384);
  n.conv_splitk = env_int("DDPM_CONV_SPLITK", 1) != 0;
  n.gn_fused = env_int("DDPM_GN_FUSED", 1) != 0;
  n.attn_waves8 = env_int("DDPM_ATTN_WAVES", 8) != 4;
  n.convin_fast = env_int("DDPM_CONVIN_FAST", 1) != 0;
  n.convout_wave = env_int("DDPM_CONVOUT_WAVE", 1) != 0;
  n.convout_w16_maxwg = getenv("DDPM_CONVOUT_W16_MAXWG") ? atol(getenv("DDPM_CONVOUT_W16_MAXWG")) : -1;
  n.convin_blocks_per_cu = getenv("DDPM_CONVIN_BLOCKS_PER_CU") ? atol(getenv("DDPM_CONVIN_BLOCKS_PER_CU")) : 8;
  n.prof_shapes = getenv("DDPM_PROF_SHAPES") != nullptr;
  n.split_f16 = g_sw_loaded ? keep : true;
  g_sw = n;
  g_sw_loaded = true;
  g_sw_epoch += 1;
}
const Switches &sw() {
  if (!g_sw_loaded) load_switches();
  return g_sw;
}

// ---- device status word -------------------------------------------------------------------------
static unsigned *g_status[64] = {};
unsigned *status_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!g_status[dev]) {
    unsigned *p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    g_status[dev] = p;
  }
  return g_status[dev];
}

// ---- in-situ profiler ---------------------------------------------------------------------------
struct ProfRec {
  hipEvent_t a, b;
  std::string kernel;
  double flops, bytes;
};
bool g_prof_on = false;
static std::vector<ProfRec> g_recs;

void prof_begin(hipStream_t s, const char *kernel, double flops, double bytes) {
  ProfRec r{nullptr, nullptr, std::string(kernel), flops, bytes};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}

void prof_end(hipStream_t s) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, s);
}

}  // namespace ddpm

using namespace ddpm;

extern "C" int ddpm_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

extern "C" int ddpm_prof_report(char *buf, size_t cap) {
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto &r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg &a = agg[r.kernel];
      a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  std::string out = "{";
  bool first = true;
  for (auto &kv : agg) {
    char line[768];
    snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += line;
    first = false;
  }
  out += "}";
  if (!buf || out.size() + 1 > cap) {
    set_error("prof_report: buffer too small (%zu needed)", out.size() + 1);
    return DDPM_EINVAL;
  }
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int)out.size();
}

extern "C" int ddpm_reload_env(void) {
  load_switches();
  return 0;
}

extern "C" int ddpm_set_split_f16(int on) {
  (void)sw();
  const int was = g_sw.split_f16 ? 1 : 0;
  if ((on != 0) != g_sw.split_f16) g_sw_epoch += 1;
  g_sw.split_f16 = on != 0;
  return was;
}

// the master switch, exactly what ddpm_set_split_f16 sets and returns (prev = get(); set(0); ...; set(prev) restores it)
extern "C" int ddpm_get_split_f16(void) { return sw().split_f16 ? 1 : 0; }

// 1 only when the master switch is on AND at least one split-f16 family is enabled: "would ddpm_set_split_f16(0) change
// which kernels run?" (the trainer's re-run guard asks exactly that)
extern "C" int ddpm_split_f16_active(void) {
  const Switches &w = sw();
  const bool any = w.wino44_f16x3 || w.conv1x1_f16x3 || w.attn_f16x3 || w.down_s2h != 0 || w.conv_d3s != 0 || w.up_wino44h;
  return w.split_f16 && any ? 1 : 0;
}

extern "C" int ddpm_status_read(unsigned *word, int clear, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(word != nullptr, "status_read: NULL pointer");
  unsigned *st = status_word();
  DDPM_CHECK_ARG(st != nullptr, "status_read: no status word on this device");
  hipError_t e = hipMemcpyAsync(word, st, sizeof(unsigned), hipMemcpyDeviceToHost, as_stream(stream));
  if (e == hipSuccess && clear) e = hipMemsetAsync(st, 0, sizeof(unsigned), as_stream(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
  if (e != hipSuccess) {
    set_error("status_read: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ddpm_abi_version(void) { return DDPM_ABI_VERSION; }
extern "C" const char *ddpm_last_error(void) { return g_err; }

extern "C" int ddpm_conv_f32(const ddpm_conv_desc *d, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(d != nullptr, "conv: descriptor is NULL");
  return conv_dispatch(*d, as_stream(stream));
}

extern "C" size_t ddpm_conv_scratch_floats(const ddpm_conv_desc *d) {
  if (!d) return 0;
  return conv_scratch_floats(*d);
}

extern "C" size_t ddpm_conv_s2h_weight_halves(int Cout, int Cin) { return conv_s2h_weight_halves(Cout, Cin); }

extern "C" int ddpm_pack_conv_s2h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, ddpm_stream_t stream) {
  return launch_pack_conv_s2h_weight(w_raw, dst, Cout, Cin, as_stream(stream));
}

extern "C" size_t ddpm_conv_d3h_weight_halves(int Cout, int Cin) { return conv_d3h_weight_halves(Cout, Cin); }

extern "C" size_t ddpm_conv_d1s_weight_halves(int Cout, int Cin) { return conv_d1s_weight_halves(Cout, Cin); }

extern "C" int ddpm_pack_conv_d1s_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, int cout_offset, int Cout_total,
                                         ddpm_stream_t stream) {
  return launch_pack_conv_d1s_weight(w_raw, dst, Cout, Cin, cout_offset, Cout_total, as_stream(stream));
}

extern "C" int ddpm_pack_conv_d3h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, ddpm_stream_t stream) {
  return launch_pack_conv_d3h_weight(w_raw, dst, Cout, Cin, as_stream(stream));
}

extern "C" size_t ddpm_conv1x1_h_weight_halves(int Cout, int Cin) { return conv1x1_h_weight_halves(Cout, Cin); }

extern "C" int ddpm_pack_conv1x1_h_weight(const float *w_raw, uint16_t *dst, int Cout, int Cin, ddpm_stream_t stream) {
  return launch_pack_conv1x1_h_weight(w_raw, dst, Cout, Cin, 0, Cout, as_stream(stream));
}

extern "C" int ddpm_conv_stats_parts(const ddpm_conv_desc *d) {
  if (!d) return 0;
  return conv_stats_parts(*d);
}

extern "C" size_t ddpm_packed_conv_weight_floats(int Cout, int Cin, int ksize) {
  return packed_conv_weight_floats(Cout, Cin, ksize);
}

extern "C" int ddpm_pack_conv_weight_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize,
                                         int cout_offset, int Cout_total, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_packed, "pack: NULL pointer");
  return launch_pack_conv_weight(w_raw, w_packed, Cout, Cin, ksize, cout_offset, Cout_total, as_stream(stream));
}

extern "C" int ddpm_pack_conv_weight_taps_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize,
                                              int src_taps, int tap_off, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_packed && src_taps >= ksize * ksize && tap_off >= 0 && tap_off + ksize * ksize <= src_taps,
                 "pack_taps: bad argument");
  return launch_pack_conv_weight(w_raw, w_packed, Cout, Cin, ksize, 0, Cout, as_stream(stream), src_taps, tap_off);
}

extern "C" int ddpm_pack_conv3d_weight_f32(const float *w_raw, float *w_packed, int Cout, int Cin, int ksize,
                                           ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_packed && (ksize == 3 || ksize == 4), "pack_conv3d: NULL pointer or ksize not 3 / 4");
  const size_t slab = (size_t)Cout * Cin * ksize * ksize;
  for (int kd = 0; kd < ksize; ++kd) {
    const int rc = launch_pack_conv_weight(w_raw, w_packed + kd * slab, Cout, Cin, ksize, 0, Cout, as_stream(stream),
                                           ksize * ksize * ksize, ksize * ksize * kd);
    if (rc) return rc;
  }
  return 0;
}

extern "C" size_t ddpm_packed_convtr_weight_floats(int Cout, int Cin, int dims) {
  return packed_convT_weight_floats(Cout, Cin, dims);
}

extern "C" int ddpm_pack_convtr_weight_f32(const float *w_raw, float *w_packed, int Cin, int Cout, int dims,
                                          ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_packed, "pack_convT: NULL pointer");
  return launch_pack_convT_weight(w_raw, w_packed, Cin, Cout, dims, as_stream(stream));
}

extern "C" int ddpm_conv3d_k4s2_cin1_f32(const float *in, const float *w, const float *bias, float *out, int B,
                                         int Cout, int D, int H, int W, int relu, ddpm_stream_t stream) {
  return launch_conv3d_k4s2_cin1(in, w, bias, out, B, Cout, D, H, W, relu, as_stream(stream));
}

extern "C" int ddpm_convtr3d_k4s2_cout1_f32(const float *in, const float *w, const float *bias, float *out, int B,
                                           int Cin, int D, int H, int W, ddpm_stream_t stream) {
  return launch_convT3d_k4s2_cout1(in, w, bias, out, B, Cin, D, H, W, as_stream(stream));
}

extern "C" size_t ddpm_wino_weight_floats(int Cout, int Cin) { return wino_weight_floats(Cout, Cin); }

extern "C" int ddpm_pack_wino_weight_f32(const float *w_raw, float *w_wino, int Cout, int Cin, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_wino, "wino pack: NULL pointer");
  return launch_pack_wino_weight(w_raw, w_wino, Cout, Cin, as_stream(stream));
}

extern "C" size_t ddpm_wino44_weight_floats(int Cout, int Cin) { return wino44_weight_floats(Cout, Cin); }

extern "C" int ddpm_pack_wino44_weight_f32(const float *w_raw, float *w_wino44, int Cout, int Cin, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_wino44, "wino44 pack: NULL pointer");
  return launch_pack_wino44_weight(w_raw, w_wino44, Cout, Cin, as_stream(stream));
}

extern "C" int ddpm_convnd_generic_f32(const float *in, const float *w, const float *bias, const float *residual, float *out,
                                       int B, int Cin, int Cout, int Di, int Hi, int Wi, int dims, int ksize, int stride,
                                       int pad, int transposed, int relu, ddpm_stream_t stream) {
  return launch_convnd_generic(in, w, bias, residual, out, B, Cin, Cout, Di, Hi, Wi, dims, ksize, stride, pad, transposed, relu,
                               as_stream(stream));
}

extern "C" size_t ddpm_wino44h_weight_halves(int Cout, int Cin) { return wino44h_weight_halves(Cout, Cin); }

extern "C" int ddpm_pack_wino44h_weight(const float *w_raw, uint16_t *w_wino44h, int Cout, int Cin, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_wino44h, "wino44h pack: NULL pointer");
  return launch_pack_wino44h_weight(w_raw, w_wino44h, Cout, Cin, as_stream(stream));
}

extern "C" int ddpm_pack_wino44h_weight3d(const float *w_raw, uint16_t *w_wino44h, int Cout, int Cin, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_wino44h, "wino44h 3d pack: NULL pointer");
  return launch_pack_wino44h_weight(w_raw, w_wino44h, Cout, Cin, as_stream(stream), 3);
}

extern "C" int ddpm_pack_wino44_weight3d_f32(const float *w_raw, float *w_wino44, int Cout, int Cin, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_wino44, "wino44 3d pack: NULL pointer");
  return launch_pack_wino44_weight(w_raw, w_wino44, Cout, Cin, as_stream(stream), 3);
}

extern "C" int ddpm_pack_wino3d_weight_f32(const float *w_raw, float *w_wino, int Cout, int Cin, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_wino, "wino3d pack: NULL pointer");
  return launch_pack_wino_weight(w_raw, w_wino, Cout, Cin, as_stream(stream), 3);
}

extern "C" size_t ddpm_folded_upsample_weight_floats(int Cout, int Cin) {
  return folded_upsample_weight_floats(Cout, Cin);
}

extern "C" int ddpm_fold_upsample_weight_f32(const float *w_raw, float *w_folded, int Cout, int Cin,
                                             ddpm_stream_t stream) {
  DDPM_CHECK_ARG(w_raw && w_folded, "fold: NULL pointer");
  return launch_fold_upsample_weight(w_raw, w_folded, Cout, Cin, as_stream(stream));
}

extern "C" int ddpm_gn_finalize_f32(const float *st1, int parts1, int C1, const float *st2, int parts2, int C2,
                                    const float *gamma, const float *beta, float *scale, float *shift, int B, int HW,
                                    int groups, float eps, ddpm_stream_t stream) {
  return launch_gn_finalize(st1, parts1, C1, st2, parts2, C2, gamma, beta, scale, shift, B, HW, groups, eps,
                            as_stream(stream));
}

extern "C" int ddpm_channel_stats_f32(const float *in, float *stats, int B, int C, int HW, ddpm_stream_t stream) {
  return launch_channel_stats(in, stats, B, C, HW, as_stream(stream));
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

extern "C" size_t ddpm_attention_scratch_floats(int B, int C, int N, int num_heads) {
  return attention_fa_scratch_floats(B, C, N, num_heads);
}

extern "C" int ddpm_attention_ws_f32(const float *qkv, const float *residual, float *out, int B, int C, int N, int num_heads,
                                     float scale, float *scratch, size_t scratch_floats, ddpm_stream_t stream) {
  return launch_attention(qkv, residual, out, B, C, N, num_heads, scale, as_stream(stream), scratch, scratch_floats);
}
