// unet_engine.hip -- native DiffusionModelUNet forward: one C call per UNet evaluation.
//
// Mirrors MONAI-Generative 0.2.x DiffusionModelUNet as constructed at
// /root/reference/src/trainers/base.py:66-86 and called at
// /root/reference/src/trainers/reconstruct.py:151-153.  The host walks the block list and
// enqueues the fused HIP kernels on the caller's stream; it never allocates (parameters live
// in a caller-owned blob, activations in a caller-owned workspace carved by a bump allocator)
// and never synchronises.
//
// Fusion plan per ResnetBlock (2 GroupNorm-stat launches + 2-3 conv launches instead of the
// reference's ~12 ATen dispatches):
//   gn_stats(x)  -> conv3x3[affine+SiLU prologue, +bias, +temb]      -> h1
//   gn_stats(h1) -> conv3x3[affine+SiLU prologue, +bias, +identity]  -> out     (Cin == Cout)
//                or conv3x3[...] -> h2 ; conv1x1(x)[+bias, +h2]      -> out     (Cin != Cout)
// torch.cat is virtual (two source pointers), nearest-x2 upsample and stride-2 are input
// indexing, the 11 time_emb_proj Linears are ONE GEMM, to_q/to_k/to_v are ONE 1x1 conv.
#include <deque>
#include <map>
#include <tuple>
#include <vector>

#include "common.h"

namespace ddpm {

struct ParamSlot {
  std::string name;
  int64_t numel = 0;
  size_t raw_off = 0;           // float offset of the torch-layout copy inside the blob
  bool is_conv = false;         // also packed for the MFMA kernel when the shape allows it
  size_t packed_base = 0;       // float offset of the (possibly shared) packed weight
  bool has_folded = false;      // Upsample conv: also keep the folded (4 x 2x2-tap) form
  size_t folded_base = 0;
  bool has_wino = false;        // 3x3 stride-1 conv: also keep the Winograd-domain form
  size_t wino_base = 0;
  bool has_wino44 = false;      // ... and the F(4x4, 3x3) form
  size_t wino44_base = 0;
  bool has_wino44h = false;     // ... and its split-f16 form (conv_wino44h.hip); base in floats, 2 f16 per float
  size_t wino44h_base = 0;
  bool has_s2h = false;         // Downsample conv: split-f16 planes of the direct stride-2 kernel (conv_s2h.hip), in wino44h_base
  bool has_d3h = false;         // stride-1 3x3 conv: split-f16 planes of the one-shot direct kernels (conv_d3s.hip)
  bool has_d1s = false;         // 1x1 conv: split-f16 planes of the small-launch kernel (conv_d3s.hip), in d3h_base
  size_t d3h_base = 0;
  bool has_h1 = false;          // 1x1 conv: pre-split f16 planes of the DMA-fed kernel (conv1x1_dma.hip), in wino44h_base
  int Cout = 0, Cin = 0, ksize = 1, cout_offset = 0, Cout_total = 0;
  int dims = 2;                 // 3: [Cout, Cin, k, k, k] packed as k slabs of 2-D taps (one per depth tap)
  bool optional = false;
  bool set = false;
};

struct ConvRef {  // a conv-like op: weight (raw + packed) and bias locations in the blob
  size_t w_raw = 0, w_packed = 0, bias = 0, w_folded = 0, w_wino = 0, w_wino44 = 0, w_wino44h = 0, w_d3h = 0;
  bool has_packed = false, has_folded = false, has_wino = false, has_wino44 = false, has_wino44h = false, has_d3h = false;
  int Cin = 0, Cout = 0, ksize = 1;
  int dims = 2;
};

struct GNRef {
  size_t gamma = 0, beta = 0;
  int C = 0;
};

struct ResRef {
  int Cin = 0, Cout = 0;
  GNRef n1, n2;
  ConvRef c1, c2, skip;
  bool has_skip = false;
  int temb_off = 0;
};

struct AttnRef {
  int C = 0, heads = 1;
  GNRef norm;
  ConvRef qkv, proj;
};

struct DownRef {
  std::vector<ResRef> res;
  std::vector<AttnRef> att;
  bool with_attn = false, has_down = false;
  ConvRef down;
};

struct UpRef {
  std::vector<ResRef> res;
  std::vector<AttnRef> att;
  bool with_attn = false, has_up = false;
  ConvRef up;
};

}  // namespace ddpm

using namespace ddpm;

namespace ddpm {
// One captured forward: the ~80 kernel launches of a UNet evaluation as a hipGraph, keyed by everything that is
// baked into the kernel arguments (tensor addresses and extents).
typedef std::tuple<const void *, const void *, void *, void *, int, int, int, int> GraphKey;
struct GraphEntry {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int calls = 0;
};
}  // namespace ddpm

struct ddpm_unet {
  ddpm_unet_config cfg;
  // hipGraph replay (ddpm_unet_forward_graphed): private capture / replay stream + the events that order it
  // against the caller's stream
  std::map<ddpm::GraphKey, ddpm::GraphEntry> graphs;
  unsigned graph_epoch = 0;  // ddpm::switch_epoch() the captured graphs were taken under
  hipStream_t gstream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  void drop_graphs() {
    // a replay may still be in flight on the private stream (the API does not ask the caller to synchronise before flipping a
    // switch or rebinding parameters): an executing graph must not be destroyed under it
    if (gstream && !graphs.empty()) (void)hipStreamSynchronize(gstream);
    for (auto &kv : graphs) {
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    graphs.clear();
  }
  ~ddpm_unet() {
    drop_graphs();
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (gstream) (void)hipStreamDestroy(gstream);
  }
  std::vector<ParamSlot> params;
  std::map<std::string, int> index;
  size_t blob_floats = 0;
  float *blob = nullptr;

  int ch0 = 0, ted = 0, temb_total = 0;
  size_t freqs_off = 0;
  ConvRef conv_in, te0, te2, temb_all, conv_out;
  GNRef out_norm;
  std::vector<DownRef> down;
  ResRef mid1, mid2;
  AttnRef mid_attn;
  std::vector<UpRef> up;

  size_t alloc(size_t n) {
    const size_t off = blob_floats;
    blob_floats += (n + 63) & ~size_t(63);  // 256-byte granules
    return off;
  }
  int add_raw(const std::string &name, int64_t numel, size_t off, bool optional = false) {
    ParamSlot p;
    p.name = name; p.numel = numel; p.raw_off = off; p.optional = optional;
    index[name] = (int)params.size();
    params.push_back(p);
    return (int)params.size() - 1;
  }
  // conv / linear with its own weight and bias
  // `dims` = spatial rank of the kernel (1x1 convs and Linears are rank-free: pass 2)
  ConvRef add_conv(const std::string &prefix, int Cout, int Cin, int k, bool optional = false, int dims = 2) {
    ConvRef r;
    if (k == 1) dims = 2;
    r.Cin = Cin; r.Cout = Cout; r.ksize = k; r.dims = dims;
    const size_t n = (size_t)Cout * Cin * (dims == 3 ? k * k * k : k * k);
    r.w_raw = alloc(n);
    r.has_packed = packed_conv_weight_floats(Cout, Cin, k) != 0;
    if (r.has_packed) r.w_packed = alloc(n);
    const bool h1 = k == 1 && r.has_packed && conv1x1_h_weight_halves(Cout, Cin) != 0;  // skip connections, proj_attn
    if (h1) {
      r.has_wino44h = true;
      r.w_wino44h = alloc(n);  // 2 f16 per weight
    }
    const size_t n1 = (k == 1 && sw().conv_d3s) ? conv_d1s_weight_halves(Cout, Cin) : 0;  // small-launch planes (conv_d3s.hip)
    if (n1) {
      r.has_d3h = true;
      r.w_d3h = alloc((n1 + 1) / 2);
    }
    r.bias = alloc(Cout);
    const int wi = add_raw(prefix + ".weight", (int64_t)n, r.w_raw, optional);
    params[wi].has_d1s = n1 != 0;
    params[wi].d3h_base = r.w_d3h;
    params[wi].is_conv = r.has_packed;
    params[wi].packed_base = r.w_packed;
    params[wi].has_h1 = h1;
    params[wi].wino44h_base = r.w_wino44h;
    params[wi].Cout = Cout; params[wi].Cin = Cin; params[wi].ksize = k;
    params[wi].cout_offset = 0; params[wi].Cout_total = Cout; params[wi].dims = dims;
    add_raw(prefix + ".bias", Cout, r.bias, optional);
    return r;
  }
  // member of a fused conv: rows [cout_offset, cout_offset + Cout) of a shared weight / bias
  void add_fused_member(const std::string &prefix, const ConvRef &shared, int Cout, int cout_offset) {
    const size_t n = (size_t)Cout * shared.Cin * shared.ksize * shared.ksize;
    const int wi = add_raw(prefix + ".weight", (int64_t)n,
                           shared.w_raw + (size_t)cout_offset * shared.Cin * shared.ksize * shared.ksize);
    params[wi].is_conv = shared.has_packed;
    params[wi].packed_base = shared.w_packed;
    params[wi].has_h1 = shared.ksize == 1 && shared.has_wino44h;
    params[wi].wino44h_base = shared.w_wino44h;
    params[wi].has_d1s = shared.ksize == 1 && shared.has_d3h && Cout % 64 == 0 && cout_offset % 64 == 0;
    params[wi].d3h_base = shared.w_d3h;
    params[wi].Cout = Cout; params[wi].Cin = shared.Cin; params[wi].ksize = shared.ksize;
    params[wi].cout_offset = cout_offset; params[wi].Cout_total = shared.Cout;
    add_raw(prefix + ".bias", Cout, shared.bias + cout_offset);
  }
  ConvRef alloc_shared(int Cout_total, int Cin, int k) {
    ConvRef r;
    r.Cin = Cin; r.Cout = Cout_total; r.ksize = k;
    const size_t n = (size_t)Cout_total * Cin * k * k;
    r.w_raw = alloc(n);
    r.has_packed = packed_conv_weight_floats(Cout_total, Cin, k) != 0;
    if (r.has_packed) r.w_packed = alloc(n);
    if (k == 1 && r.has_packed && conv1x1_h_weight_halves(Cout_total, Cin) != 0) {  // fused q / k / v (and the temb projections)
      r.has_wino44h = true;
      r.w_wino44h = alloc(n);
    }
    if (const size_t n1 = (k == 1 && sw().conv_d3s) ? conv_d1s_weight_halves(Cout_total, Cin) : 0) {
      r.has_d3h = true;
      r.w_d3h = alloc((n1 + 1) / 2);
    }
    r.bias = alloc(Cout_total);
    return r;
  }
  GNRef add_gn(const std::string &prefix, int C) {
    GNRef g;
    g.C = C;
    g.gamma = alloc(C);
    g.beta = alloc(C);
    add_raw(prefix + ".weight", C, g.gamma);
    add_raw(prefix + ".bias", C, g.beta);
    return g;
  }
};

namespace ddpm {

static ResRef build_res(ddpm_unet *u, const std::string &prefix, int Cin, int Cout, int &temb_cursor,
                        std::vector<std::pair<std::string, int>> &temb_members) {
  ResRef r;
  r.Cin = Cin; r.Cout = Cout;
  r.n1 = u->add_gn(prefix + ".norm1", Cin);
  r.c1 = u->add_conv(prefix + ".conv1.conv", Cout, Cin, 3, false, u->cfg.spatial_dims);
  r.temb_off = temb_cursor;
  temb_members.emplace_back(prefix + ".time_emb_proj", Cout);
  temb_cursor += Cout;
  r.n2 = u->add_gn(prefix + ".norm2", Cout);
  r.c2 = u->add_conv(prefix + ".conv2.conv", Cout, Cout, 3, false, u->cfg.spatial_dims);
  {  // both 3x3 convs of a ResnetBlock are stride 1: Winograd-domain weights too (3-D: one F(2x2) slab per depth tap)
    const bool d3 = u->cfg.spatial_dims == 3;
    ConvRef *cr[2] = {&r.c1, &r.c2};
    const char *nm[2] = {".conv1.conv.weight", ".conv2.conv.weight"};
    for (int i = 0; i < 2; ++i) {
      const size_t nw = wino_weight_floats(cr[i]->Cout, cr[i]->Cin) * (d3 ? 3 : 1);
      if (!nw) continue;
      cr[i]->has_wino = true;
      cr[i]->w_wino = u->alloc(nw);
      ParamSlot &ps = u->params[u->index[prefix + nm[i]]];
      ps.has_wino = true;
      ps.wino_base = cr[i]->w_wino;
      if (d3) continue;  // latent volumes (8^3) have no F(4x4) tiling
      if (const size_t n44 = wino44_weight_floats(cr[i]->Cout, cr[i]->Cin)) {
        cr[i]->has_wino44 = true;
        cr[i]->w_wino44 = u->alloc(n44);
        ps.has_wino44 = true;
        ps.wino44_base = cr[i]->w_wino44;
      }
      if (const size_t nh = wino44h_weight_halves(cr[i]->Cout, cr[i]->Cin)) {
        cr[i]->has_wino44h = true;
        cr[i]->w_wino44h = u->alloc(nh / 2);
        ps.has_wino44h = true;
        ps.wino44h_base = cr[i]->w_wino44h;
      }
      // direct split-f16 planes: read by the small-launch kernel (conv_d3s.hip: 8x8 / 16x16 layers of a few images) and by the
      // direct split-f16 planes for the one-shot small-launch kernels (conv_d3s.hip)
      if (const size_t nd = sw().conv_d3s ? conv_d3h_weight_halves(cr[i]->Cout, cr[i]->Cin) : 0) {
        cr[i]->has_d3h = true;
        cr[i]->w_d3h = u->alloc((nd + 1) / 2);
        ps.has_d3h = true;
        ps.d3h_base = cr[i]->w_d3h;
      }
    }
  }
  r.has_skip = Cin != Cout;
  if (r.has_skip) r.skip = u->add_conv(prefix + ".skip_connection.conv", Cout, Cin, 1);
  return r;
}

static AttnRef build_attn(ddpm_unet *u, const std::string &prefix, int C, int head_channels) {
  AttnRef a;
  a.C = C;
  a.heads = head_channels > 0 ? C / head_channels : 1;
  a.norm = u->add_gn(prefix + ".norm", C);
  a.qkv = u->alloc_shared(3 * C, C, 1);
  u->add_fused_member(prefix + ".to_q", a.qkv, C, 0);
  u->add_fused_member(prefix + ".to_k", a.qkv, C, C);
  u->add_fused_member(prefix + ".to_v", a.qkv, C, 2 * C);
  a.proj = u->add_conv(prefix + ".proj_attn", C, C, 1, /*optional=*/!u->cfg.use_proj_attn);
  return a;
}

}  // namespace ddpm

extern "C" ddpm_unet *ddpm_unet_create(const ddpm_unet_config *cfg) {
  if (!cfg) { set_error("unet_create: cfg is NULL"); return nullptr; }
  if (cfg->spatial_dims != 2 && cfg->spatial_dims != 3) {
    set_error("unet_create: spatial_dims must be 2 or 3, got %d", cfg->spatial_dims);
    return nullptr;
  }
  if (cfg->spatial_dims == 3) {
    // 3-D convolutions run as depth-tap launches of the MFMA kernel; there is no generic 3-D fallback
    bool ok = cfg->in_channels % 4 == 0 && cfg->out_channels % 128 == 0;
    for (int i = 0; i < cfg->num_levels; ++i) ok = ok && cfg->num_channels[i] % 128 == 0;
    if (!ok) {
      set_error("unet_create: spatial_dims = 3 needs in_channels %% 4 == 0 and out / num_channels %% 128 == 0");
      return nullptr;
    }
  }
  const int L = cfg->num_levels;
  if (L < 1 || L > DDPM_MAX_LEVELS) { set_error("unet_create: num_levels out of range"); return nullptr; }
  for (int i = 0; i < L; ++i) {
    if (cfg->num_channels[i] % cfg->norm_num_groups) {
      set_error("DiffusionModelUNet expects all num_channels being multiple of norm_num_groups");
      return nullptr;
    }
    if (cfg->num_res_blocks[i] < 1) { set_error("unet_create: num_res_blocks < 1"); return nullptr; }
  }
  ddpm_unet *u = new ddpm_unet();
  u->cfg = *cfg;
  u->ch0 = cfg->num_channels[0];
  u->ted = 4 * u->ch0;

  u->freqs_off = u->alloc(u->ch0 / 2);
  u->add_raw("freqs", u->ch0 / 2, u->freqs_off);
  const int sd = cfg->spatial_dims;
  u->conv_in = u->add_conv("conv_in.conv", u->ch0, cfg->in_channels, 3, false, sd);
  u->te0 = u->add_conv("time_embed.0", u->ted, u->ch0, 1);
  u->te2 = u->add_conv("time_embed.2", u->ted, u->ted, 1);

  int temb_cursor = 0;
  std::vector<std::pair<std::string, int>> temb_members;

  int out_c = u->ch0;
  for (int i = 0; i < L; ++i) {
    const int in_c = out_c;
    out_c = cfg->num_channels[i];
    DownRef d;
    d.with_attn = cfg->attention_levels[i] != 0;
    d.has_down = i != L - 1;
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < cfg->num_res_blocks[i]; ++j) {
      d.res.push_back(build_res(u, bp + ".resnets." + std::to_string(j), j == 0 ? in_c : out_c, out_c, temb_cursor,
                                temb_members));
      if (d.with_attn)
        d.att.push_back(build_attn(u, bp + ".attentions." + std::to_string(j), out_c, cfg->num_head_channels[i]));
    }
    if (d.has_down) {
      d.down = u->add_conv(bp + ".downsampler.op.conv", out_c, out_c, 3, false, sd);
      if (const size_t nh = sd == 2 ? conv_s2h_weight_halves(out_c, out_c) : 0) {  // split-f16 planes for conv_s2h.hip
        d.down.has_wino44h = true;
        d.down.w_wino44h = u->alloc(nh / 2);
        ParamSlot &ps = u->params[u->index[bp + ".downsampler.op.conv.weight"]];
        ps.has_s2h = true;
        ps.wino44h_base = d.down.w_wino44h;
      }
      if (const size_t nd = (sd == 2 && sw().conv_d3s) ? conv_d3h_weight_halves(out_c, out_c) : 0) {  // small launches (conv_d3s.hip)
        d.down.has_d3h = true;
        d.down.w_d3h = u->alloc((nd + 1) / 2);
        ParamSlot &ps = u->params[u->index[bp + ".downsampler.op.conv.weight"]];
        ps.has_d3h = true;
        ps.d3h_base = d.down.w_d3h;
      }
    }
    u->down.push_back(d);
  }
  const int cm = cfg->num_channels[L - 1];
  u->mid1 = build_res(u, "middle_block.resnet_1", cm, cm, temb_cursor, temb_members);
  u->mid_attn = build_attn(u, "middle_block.attention", cm, cfg->num_head_channels[L - 1]);
  u->mid2 = build_res(u, "middle_block.resnet_2", cm, cm, temb_cursor, temb_members);

  out_c = cfg->num_channels[L - 1];
  for (int i = 0; i < L; ++i) {
    const int ri = L - 1 - i;
    const int prev_c = out_c;
    out_c = cfg->num_channels[ri];
    const int in_c = cfg->num_channels[L - 1 - (i + 1 < L ? i + 1 : L - 1)];
    UpRef b;
    b.with_attn = cfg->attention_levels[ri] != 0;
    b.has_up = i != L - 1;
    const int nres = cfg->num_res_blocks[ri] + 1;
    const std::string bp = "up_blocks." + std::to_string(i);
    for (int j = 0; j < nres; ++j) {
      const int res_skip = (j == nres - 1) ? in_c : out_c;
      const int res_in = (j == 0) ? prev_c : out_c;
      b.res.push_back(build_res(u, bp + ".resnets." + std::to_string(j), res_in + res_skip, out_c, temb_cursor,
                                temb_members));
      if (b.with_attn)
        b.att.push_back(build_attn(u, bp + ".attentions." + std::to_string(j), out_c, cfg->num_head_channels[ri]));
    }
    if (b.has_up) {
      b.up = u->add_conv(bp + ".upsampler.conv.conv", out_c, out_c, 3, false, sd);
      if (sd == 2 && folded_upsample_weight_floats(out_c, out_c)) {  // nearest-x2 + 3x3 folded into 4 2x2 convs
        b.up.has_folded = true;
        b.up.w_folded = u->alloc(folded_upsample_weight_floats(out_c, out_c));
        ParamSlot &ps = u->params[u->index[bp + ".upsampler.conv.conv.weight"]];
        ps.has_folded = true;
        ps.folded_base = b.up.w_folded;
        if (const size_t nw = wino_weight_floats(out_c, out_c)) {  // Winograd on the upsampled image (9 of 16 positions)
          b.up.has_wino = true;
          b.up.w_wino = u->alloc(nw);
          ps.has_wino = true;
          ps.wino_base = b.up.w_wino;
        }
        if (const size_t nh = wino44h_weight_halves(out_c, out_c)) {  // split-f16 F(4x4) on the upsampled image (conv_wino44h.hip)
          b.up.has_wino44h = true;
          b.up.w_wino44h = u->alloc(nh / 2);
          ps.has_wino44h = true;
          ps.wino44h_base = b.up.w_wino44h;
        }
        if (const size_t nd = sw().conv_d3s ? conv_d3h_weight_halves(out_c, out_c) : 0) {  // small launches (conv_d3s.hip)
          b.up.has_d3h = true;
          b.up.w_d3h = u->alloc((nd + 1) / 2);
          ps.has_d3h = true;
          ps.d3h_base = b.up.w_d3h;
        }
      }
    }
    u->up.push_back(b);
  }
  u->out_norm = u->add_gn("out.0", u->ch0);
  u->conv_out = u->add_conv("out.2.conv", cfg->out_channels, u->ch0, 3, false, sd);

  // all time_emb_proj Linears share one [sum Cout, 4 ch0] GEMM (emb is loop-invariant in a forward)
  u->temb_total = temb_cursor;
  u->temb_all = u->alloc_shared(temb_cursor, u->ted, 1);
  int off = 0;
  for (auto &m : temb_members) {
    u->add_fused_member(m.first, u->temb_all, m.second, off);
    off += m.second;
  }
  return u;
}

extern "C" void ddpm_unet_destroy(ddpm_unet *h) { delete h; }

extern "C" size_t ddpm_unet_param_blob_floats(const ddpm_unet *h) { return h ? h->blob_floats : 0; }

extern "C" int ddpm_unet_bind_param_blob(ddpm_unet *h, float *blob) {
  DDPM_CHECK_ARG(h && blob, "bind_param_blob: NULL");
  h->blob = blob;
  h->drop_graphs();  // captured launches carry the old blob's addresses
  for (auto &p : h->params) p.set = false;
  return 0;
}

extern "C" int ddpm_unet_num_params(const ddpm_unet *h) { return h ? (int)h->params.size() : 0; }
extern "C" const char *ddpm_unet_param_name(const ddpm_unet *h, int i) {
  return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr;
}
extern "C" int64_t ddpm_unet_param_numel(const ddpm_unet *h, int i) {
  return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].numel : -1;
}

extern "C" int ddpm_unet_set_param(ddpm_unet *h, const char *name, const float *src, int64_t numel,
                                   ddpm_stream_t stream) {
  DDPM_CHECK_ARG(h && name && src, "set_param: NULL");
  DDPM_CHECK_ARG(h->blob, "set_param: bind a parameter blob first");
  auto it = h->index.find(name);
  if (it == h->index.end()) {
    set_error("set_param: unexpected key '%s'", name);
    return DDPM_ENOPARAM;
  }
  ParamSlot &p = h->params[it->second];
  if (p.numel != numel) {
    set_error("set_param: size mismatch for %s: expected %lld elements, got %lld", name, (long long)p.numel,
              (long long)numel);
    return DDPM_EINVAL;
  }
  hipStream_t s = as_stream(stream);
  int rc = launch_copy_f32(src, h->blob + p.raw_off, numel, s);
  if (rc) return rc;
  if (p.is_conv && p.dims == 3 && p.ksize == 3) {
    const size_t slab = (size_t)p.Cout_total * p.Cin * 9;  // one packed 2-D weight per depth tap
    for (int kd = 0; kd < 3 && !rc; ++kd)
      rc = launch_pack_conv_weight(src, h->blob + p.packed_base + kd * slab, p.Cout, p.Cin, 3, p.cout_offset,
                                   p.Cout_total, s, 27, 9 * kd);
    if (rc) return rc;
  } else if (p.is_conv) {
    rc = launch_pack_conv_weight(src, h->blob + p.packed_base, p.Cout, p.Cin, p.ksize, p.cout_offset, p.Cout_total, s);
    if (rc) return rc;
  }
  if (p.has_wino) {
    rc = launch_pack_wino_weight(src, h->blob + p.wino_base, p.Cout, p.Cin, s, p.dims == 3 ? 3 : 1);
    if (rc) return rc;
  }
  if (p.has_wino44) {
    rc = launch_pack_wino44_weight(src, h->blob + p.wino44_base, p.Cout, p.Cin, s);
    if (rc) return rc;
  }
  if (p.has_wino44h) {
    rc = launch_pack_wino44h_weight(src, reinterpret_cast<uint16_t *>(h->blob + p.wino44h_base), p.Cout, p.Cin, s);
    if (rc) return rc;
  }
  if (p.has_d3h) {
    rc = launch_pack_conv_d3h_weight(src, reinterpret_cast<uint16_t *>(h->blob + p.d3h_base), p.Cout, p.Cin, s);
    if (rc) return rc;
  }
  if (p.has_d1s) {
    rc = launch_pack_conv_d1s_weight(src, reinterpret_cast<uint16_t *>(h->blob + p.d3h_base), p.Cout, p.Cin, p.cout_offset,
                                     p.Cout_total, s);
    if (rc) return rc;
  }
  if (p.has_s2h) {
    rc = launch_pack_conv_s2h_weight(src, reinterpret_cast<uint16_t *>(h->blob + p.wino44h_base), p.Cout, p.Cin, s);
    if (rc) return rc;
  }
  if (p.has_h1) {
    rc = launch_pack_conv1x1_h_weight(src, reinterpret_cast<uint16_t *>(h->blob + p.wino44h_base), p.Cout, p.Cin, p.cout_offset,
                                      p.Cout_total, s);
    if (rc) return rc;
  }
  if (p.has_folded) {
    rc = launch_fold_upsample_weight(src, h->blob + p.folded_base, p.Cout, p.Cin, s);
    if (rc) return rc;
  }
  p.set = true;
  h->drop_graphs();  // (the packed copies live at fixed addresses, but keep replay and parameters in lock-step)
  return 0;
}

namespace ddpm {

struct Bump {
  char *base;
  size_t cap, off = 0, peak = 0;
  bool dry;
  bool over = false;  // a real run asked for more than the dry run had sized: its decisions differed (a bug -- fail loudly)
  int *rc_out = nullptr;  // the owner's error code, set at the moment of the overflow (Runner::run points it at Runner::rc)
  float *get(size_t floats) {
    const size_t bytes = (floats * sizeof(float) + 255) & ~size_t(255);
    // The dry run hands out NON-NULL, 256-byte aligned fake addresses (never dereferenced: every launch is skipped): the kernels'
    // supported() / scratch_floats() predicates look at which optional pointers are present and how they are aligned, and the
    // dry run must take the decisions of the real one.  (With NULL here a GroupNorm-ed convolution looked like "SiLU without
    // scale / shift" to the sizing pass; a kernel that requires the pair was sized out and then taken by the real run.)
    float *p = dry ? reinterpret_cast<float *>((uintptr_t(1) << 40) + off) : reinterpret_cast<float *>(base + off);
    off += bytes;
    if (off > peak) peak = off;
    if (!dry && off > cap) {
      // never hand out an address past the buffer: the start of the workspace is valid memory, the results are garbage and
      // `rc_out` (the Runner's rc) stops every later launch -- all of them are gated on it
      over = true;
      if (rc_out && !*rc_out) {
        ddpm::set_error("unet_forward: workspace overflow (%zu of %zu bytes): the sizing pass took other decisions than the run", off, cap);
        *rc_out = DDPM_EINVAL;
      }
      return reinterpret_cast<float *>(base);
    }
    return p;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

constexpr int kMaxStatParts = 8;  // slices per (image, channel) of a GroupNorm statistics slab (ddpm_conv_desc.stats_out)

struct Act {  // an activation tensor [B, C, H, W] (D == 1) or [B, C, D, H, W]
  float *p = nullptr;
  int C = 0, H = 0, W = 0, D = 1;
  // per-channel GroupNorm statistics [B, C, *sparts, 2] written by the tensor's producer (or by gn_channel_stats on first
  // use); *sparts == 0: not computed.  The counter lives in the Runner so that copies of the Act (skip stack) share it.
  float *stats = nullptr;
  int *sparts = nullptr;
  size_t voxels() const { return (size_t)D * H * W; }
};

struct Runner {
  ddpm_unet *u;
  Bump ws;
  hipStream_t s;
  int B;
  int rc = 0;
  float *temb = nullptr;  // [B, temb_total]
  std::deque<int> spool;  // Act::sparts storage (stable addresses)
  bool fuse_stats = sw().gn_fused;  // 0: every GroupNorm reads its input
  bool eager_stats = false;  // large launches: per-channel slabs for every tensor (a skip connection is reduced once, not twice)

  const float *P(size_t off) const { return u->blob + off; }

  void conv(const ConvRef &c, const Act &in1, const Act *in2, const float *gsc, const float *gsh, int act, int mode,
            const float *chan_add, int chan_stride, const float *residual, float *out, int Ho, int Wo, int Do = 1,
            const Act *outact = nullptr) {
    if (rc) return;
    ddpm_conv_desc d{};
    d.in1 = in1.p; d.C1 = in1.C;
    d.in2 = in2 ? in2->p : nullptr; d.C2 = in2 ? in2->C : 0;
    d.w_packed = c.has_packed ? P(c.w_packed) : nullptr;
    d.w_raw = P(c.w_raw);
    d.bias = P(c.bias);
    d.gscale = gsc; d.gshift = gsh;
    d.chan_add = chan_add; d.chan_add_stride = chan_stride;
    d.residual = residual;
    d.out = out;
    d.B = B; d.Cout = c.Cout;
    d.Hi = in1.H; d.Wi = in1.W; d.Ho = Ho; d.Wo = Wo;
    d.ksize = c.ksize; d.mode = mode; d.act = act;
    if (mode == DDPM_CONV_UPSAMPLE2 && c.has_folded) d.w_folded = P(c.w_folded);
    if ((mode == DDPM_CONV_NORMAL || mode == DDPM_CONV_UPSAMPLE2) && c.has_wino && c.dims == 2)
      d.w_wino = P(c.w_wino);
    if (mode == DDPM_CONV_NORMAL && c.has_wino44 && c.dims == 2) d.w_wino44 = P(c.w_wino44);
    if ((mode == DDPM_CONV_NORMAL || mode == DDPM_CONV_STRIDE2 || mode == DDPM_CONV_UPSAMPLE2) && c.has_d3h && c.dims == 2) d.w_d3h = reinterpret_cast<const uint16_t *>(P(c.w_d3h));
    if ((mode == DDPM_CONV_NORMAL || mode == DDPM_CONV_UPSAMPLE2 || mode == DDPM_CONV_STRIDE2) && c.has_wino44h && c.dims == 2)
      d.w_wino44h = reinterpret_cast<const uint16_t *>(P(c.w_wino44h));
    if (c.dims == 3 && c.ksize == 3 && mode == DDPM_CONV_NORMAL && c.has_wino) d.w_wino = P(c.w_wino);  // F(2x2) per depth tap
    if (c.dims == 3 && c.ksize == 3) {
      // F.conv3d: ONE launch walks the (depth tap, channel group) chunks (w_packed = three depth slabs); a
      // volume of depth 1 (input depth <= 2^(levels-1)) only has its centre tap
      d.dims = 3;
      d.Di = in1.D; d.Do = Do;
      d.w_raw = nullptr;  // the torch-layout tensor has 27 taps: never hand it to a 9-tap kernel
    } else if (in1.D > 1 || Do > 1) {
      // pointwise over a volume: the NCDHW tensor is an NCHW tensor of extent (D*H) x W
      d.Hi = in1.D * in1.H; d.Ho = Do * Ho;
    }
    // launches smaller than the chip: scratch slabs for the Winograd kernels' channel-stream split / the MFMA kernel's split-K (released with the enclosing block's
    // temporaries; the dry run that sizes the workspace takes the same decisions)
    if (const size_t need = conv_scratch_floats(d)) {
      d.scratch = ws.get(need);
      d.scratch_floats = need;
    }
    if (ws.dry) return;
    if (ws.over && !rc) {
      set_error("unet_forward: workspace overflow (%zu of %zu bytes): the sizing pass took other decisions than the run", ws.off, ws.cap);
      rc = DDPM_EINVAL;
    }
    if (rc) return;
    if (outact && outact->stats && fuse_stats) {  // the epilogue leaves the next GroupNorm's statistics behind
      const int sp = conv_stats_parts(d);
      if (sp > 0 && sp <= kMaxStatParts) {
        d.stats_out = outact->stats;
        *outact->sparts = sp;
      }
    }
    rc = conv_dispatch(d, s);
  }

  // GroupNorm -> scale / shift.  Sources whose producer left per-channel statistics need no pass over the activation; a
  // source without them is reduced per channel once (and keeps the slab for its later uses, e.g. as a skip connection) when
  // the other source has them; with no statistics at all the round-2 kernel reads the input(s).
  void gn(const GNRef &g, const Act &in1, const Act *in2, float *sc, float *sh) {
    if (ws.dry || rc) return;
    const bool h1 = in1.sparts && *in1.sparts > 0, h2 = in2 && in2->sparts && *in2->sparts > 0;
    const bool can1 = in1.stats != nullptr, can2 = !in2 || in2->stats != nullptr;
    if (fuse_stats && can1 && can2 && (h1 || h2 || eager_stats)) {
      const int HW = (int)in1.voxels();
      if (!h1) {
        rc = launch_channel_stats(in1.p, in1.stats, B, in1.C, HW, s);
        *in1.sparts = 1;
      }
      if (!rc && in2 && !h2) {
        rc = launch_channel_stats(in2->p, in2->stats, B, in2->C, HW, s);
        *in2->sparts = 1;
      }
      if (!rc)
        rc = launch_gn_finalize(in1.stats, *in1.sparts, in1.C, in2 ? in2->stats : nullptr, in2 ? *in2->sparts : 0,
                                in2 ? in2->C : 0, P(g.gamma), P(g.beta), sc, sh, B, HW, u->cfg.norm_num_groups,
                                u->cfg.norm_eps, s);
      return;
    }
    rc = launch_gn_scale_shift(in1.p, in2 ? in2->p : nullptr, in1.C, in2 ? in2->C : 0, P(g.gamma), P(g.beta), sc, sh,
                               B, (int)in1.voxels(), u->cfg.norm_num_groups, u->cfg.norm_eps, s);
  }

  // out must be allocated by the caller (so that it survives the temporaries released here)
  void resnet(const ResRef &r, const Act &in1, const Act *in2, Act &out) {
    const size_t m = ws.mark();
    const int H = in1.H, W = in1.W, D = in1.D;
    const size_t hw = in1.voxels();
    float *sc1 = ws.get((size_t)B * r.Cin), *sh1 = ws.get((size_t)B * r.Cin);
    gn(r.n1, in1, in2, sc1, sh1);
    Act h1 = new_act(r.Cout, H, W, D);
    conv(r.c1, in1, in2, sc1, sh1, DDPM_ACT_SILU, DDPM_CONV_NORMAL, temb + r.temb_off, u->temb_total, nullptr, h1.p,
         H, W, D, &h1);
    float *sc2 = ws.get((size_t)B * r.Cout), *sh2 = ws.get((size_t)B * r.Cout);
    gn(r.n2, h1, nullptr, sc2, sh2);
    if (!r.has_skip) {
      conv(r.c2, h1, nullptr, sc2, sh2, DDPM_ACT_SILU, DDPM_CONV_NORMAL, nullptr, 0, in1.p, out.p, H, W, D, &out);
    } else {
      // skip_connection(x) first, conv2 adds it: the block's output then comes from the kernel that emits statistics
      Act h2{ws.get((size_t)B * r.Cout * hw), r.Cout, H, W, D};
      conv(r.skip, in1, in2, nullptr, nullptr, DDPM_ACT_NONE, DDPM_CONV_NORMAL, nullptr, 0, nullptr, h2.p, H, W, D);
      conv(r.c2, h1, nullptr, sc2, sh2, DDPM_ACT_SILU, DDPM_CONV_NORMAL, nullptr, 0, h2.p, out.p, H, W, D, &out);
    }
    ws.release(m);
  }

  void attention(const AttnRef &a, const Act &x, Act &out) {
    const size_t m = ws.mark();
    const int N = (int)x.voxels();
    float *sc = ws.get((size_t)B * a.C), *sh = ws.get((size_t)B * a.C);
    gn(a.norm, x, nullptr, sc, sh);
    float *qkv = ws.get((size_t)B * 3 * a.C * N);
    conv(a.qkv, x, nullptr, sc, sh, DDPM_ACT_NONE, DDPM_CONV_NORMAL, nullptr, 0, nullptr, qkv, x.H, x.W, x.D);
    const float scale = 1.0f / sqrtf((float)a.C / (float)a.heads);
    const size_t nplanes = attention_fa_scratch_floats(B, a.C, N, a.heads);  // f16 planes of q / k / v (attention_fa.hip)
    float *planes = nplanes ? ws.get(nplanes) : nullptr;
    if (!u->cfg.use_proj_attn) {
      // 64 tokens: the kernel's epilogue leaves the next GroupNorm's statistics behind (one slice per channel)
      const bool st = fuse_stats && out.stats && out.sparts && attention_emits_stats(B, a.C, N, a.heads, planes, nplanes);
      if (!ws.dry && !rc) {
        rc = launch_attention(qkv, x.p, out.p, B, a.C, N, a.heads, scale, s, planes, nplanes, st ? out.stats : nullptr);
        if (st && !rc) *out.sparts = 1;
      }
    } else {
      Act o{ws.get((size_t)B * a.C * N), a.C, x.H, x.W, x.D};
      if (!ws.dry && !rc) rc = launch_attention(qkv, nullptr, o.p, B, a.C, N, a.heads, scale, s, planes, nplanes);
      conv(a.proj, o, nullptr, nullptr, nullptr, DDPM_ACT_NONE, DDPM_CONV_NORMAL, nullptr, 0, x.p, out.p, x.H, x.W,
           x.D);
    }
    ws.release(m);
  }

  Act new_act(int C, int H, int W, int D = 1) {
    Act a{ws.get((size_t)B * C * D * H * W), C, H, W, D};
    if (D == 1) {  // (the 3-D levels keep the reading GroupNorm: their producers emit no statistics)
      a.stats = ws.get((size_t)B * C * kMaxStatParts * 2);
      if (ws.dry) a.stats = reinterpret_cast<float *>(16);  // non-NULL marker: the dry run takes the same decisions
      spool.push_back(0);
      a.sparts = &spool.back();
    }
    return a;
  }

  int run(const float *x, const int64_t *timesteps, float *out, int H, int W, int D = 1) {
    const ddpm_unet_config &cfg = u->cfg;
    ws.rc_out = &rc;  // an overflowing ws.get() stops every launch behind it (conv, GroupNorm, attention, temb alike)
    eager_stats = (size_t)B * H * W >= 64 * 1024;
    // ---- timestep embedding + MLP + all time projections ------------------------------------------
    Act temb0{ws.get((size_t)B * u->ch0), u->ch0, 1, 1};
    if (!ws.dry && !rc) rc = launch_timestep_embedding(timesteps, P(u->freqs_off), temb0.p, B, u->ch0, s);
    Act e1{ws.get((size_t)B * u->ted), u->ted, 1, 1};
    conv(u->te0, temb0, nullptr, nullptr, nullptr, DDPM_ACT_NONE, DDPM_CONV_NORMAL, nullptr, 0, nullptr, e1.p, 1, 1);
    Act emb{ws.get((size_t)B * u->ted), u->ted, 1, 1};
    conv(u->te2, e1, nullptr, nullptr, nullptr, DDPM_ACT_SILU, DDPM_CONV_NORMAL, nullptr, 0, nullptr, emb.p, 1, 1);
    temb = ws.get((size_t)B * u->temb_total);
    conv(u->temb_all, emb, nullptr, nullptr, nullptr, DDPM_ACT_SILU, DDPM_CONV_NORMAL, nullptr, 0, nullptr, temb, 1,
         1);

    // ---- conv_in + down path ------------------------------------------------------------------------
    Act xin{const_cast<float *>(x), cfg.in_channels, H, W, D};
    Act h = new_act(u->ch0, H, W, D);
    conv(u->conv_in, xin, nullptr, nullptr, nullptr, DDPM_ACT_NONE, DDPM_CONV_NORMAL, nullptr, 0, nullptr, h.p, H, W,
         D, &h);
    std::vector<Act> skips;
    skips.push_back(h);
    for (size_t i = 0; i < u->down.size(); ++i) {
      const DownRef &d = u->down[i];
      for (size_t j = 0; j < d.res.size(); ++j) {
        Act o = new_act(d.res[j].Cout, h.H, h.W, h.D);
        resnet(d.res[j], h, nullptr, o);
        h = o;
        if (d.with_attn) {
          Act o2 = new_act(h.C, h.H, h.W, h.D);
          attention(d.att[j], h, o2);
          h = o2;
        }
        skips.push_back(h);
      }
      if (d.has_down) {
        const int Ho = (h.H + 1) / 2, Wo = (h.W + 1) / 2, Do = h.D > 1 ? (h.D + 1) / 2 : 1;
        Act o = new_act(h.C, Ho, Wo, Do);
        conv(d.down, h, nullptr, nullptr, nullptr, DDPM_ACT_NONE, DDPM_CONV_STRIDE2, nullptr, 0, nullptr, o.p, Ho, Wo,
             Do, &o);
        h = o;
        skips.push_back(h);
      }
    }
    // ---- middle ----------------------------------------------------------------------------------------
    {
      Act o = new_act(h.C, h.H, h.W, h.D);
      resnet(u->mid1, h, nullptr, o);
      Act o2 = new_act(h.C, h.H, h.W, h.D);
      attention(u->mid_attn, o, o2);
      Act o3 = new_act(h.C, h.H, h.W, h.D);
      resnet(u->mid2, o2, nullptr, o3);
      h = o3;
    }
    // ---- up path ----------------------------------------------------------------------------------------
    for (size_t i = 0; i < u->up.size(); ++i) {
      const UpRef &b = u->up[i];
      for (size_t j = 0; j < b.res.size(); ++j) {
        if (skips.empty()) { set_error("unet_forward: skip stack underflow"); return DDPM_EINVAL; }
        Act sk = skips.back();
        skips.pop_back();
        if (sk.H != h.H || sk.W != h.W || sk.D != h.D) {
          set_error("unet_forward: skip extent %dx%dx%d does not match %dx%dx%d (every input extent must be divisible "
                    "by 2^%d)", sk.D, sk.H, sk.W, h.D, h.H, h.W, (int)u->down.size() - 1);
          return DDPM_EINVAL;
        }
        Act o = new_act(b.res[j].Cout, h.H, h.W, h.D);
        resnet(b.res[j], h, &sk, o);
        h = o;
        if (b.with_attn) {
          Act o2 = new_act(h.C, h.H, h.W, h.D);
          attention(b.att[j], h, o2);
          h = o2;
        }
      }
      if (b.has_up) {
        const int Do = cfg.spatial_dims == 3 ? 2 * h.D : 1;  // nearest x2 doubles a depth of 1 as well
        Act o = new_act(h.C, 2 * h.H, 2 * h.W, Do);
        conv(b.up, h, nullptr, nullptr, nullptr, DDPM_ACT_NONE, DDPM_CONV_UPSAMPLE2, nullptr, 0, nullptr, o.p, 2 * h.H,
             2 * h.W, Do, &o);
        h = o;
      }
    }
    // ---- out: GroupNorm + SiLU + conv ----------------------------------------------------------------
    float *sc = ws.get((size_t)B * u->ch0), *sh = ws.get((size_t)B * u->ch0);
    gn(u->out_norm, h, nullptr, sc, sh);
    Act o{out, cfg.out_channels, H, W, D};
    conv(u->conv_out, h, nullptr, sc, sh, DDPM_ACT_SILU, DDPM_CONV_NORMAL, nullptr, 0, nullptr, o.p, H, W, D);
    return rc;
  }
};

}  // namespace ddpm

extern "C" size_t ddpm_unet_workspace_bytes3d(const ddpm_unet *h, int B, int D, int H, int W) {
  if (!h || B <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
  if ((h->cfg.spatial_dims == 3) != (D > 1) && !(h->cfg.spatial_dims == 3 && D == 1)) {
    set_error("unet: a 2-D network takes D == 1");
    return 0;
  }
  Runner r{const_cast<ddpm_unet *>(h), Bump{nullptr, 0, 0, 0, true}, nullptr, B};
  if (r.run(nullptr, nullptr, nullptr, H, W, D) != 0) return 0;
  return r.ws.peak + 256;
}

extern "C" size_t ddpm_unet_workspace_bytes(const ddpm_unet *h, int B, int H, int W) {
  return ddpm_unet_workspace_bytes3d(h, B, 1, H, W);
}

extern "C" int ddpm_unet_forward3d(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B, int D,
                                   int H, int W, void *workspace, size_t workspace_bytes, ddpm_stream_t stream);

extern "C" int ddpm_unet_forward(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B, int H,
                                 int W, void *workspace, size_t workspace_bytes, ddpm_stream_t stream) {
  return ddpm_unet_forward3d(h, x, timesteps, out, B, 1, H, W, workspace, workspace_bytes, stream);
}

extern "C" int ddpm_unet_forward3d(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B, int D,
                                   int H, int W, void *workspace, size_t workspace_bytes, ddpm_stream_t stream) {
  DDPM_CHECK_ARG(h && x && timesteps && out && workspace, "unet_forward: NULL argument");
  DDPM_CHECK_ARG(h->blob, "unet_forward: no parameter blob bound");
  for (auto &p : h->params) {
    if (!p.set && !p.optional) {
      set_error("unet_forward: missing key '%s' in state_dict", p.name.c_str());
      return DDPM_ENOPARAM;
    }
  }
  const size_t need = ddpm_unet_workspace_bytes3d(h, B, D, H, W);
  if (need == 0) return DDPM_EINVAL;
  if (workspace_bytes < need) {
    set_error("unet_forward: workspace too small: %zu < %zu bytes", workspace_bytes, need);
    return DDPM_EWORKSPACE;
  }
  char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
  Runner r{h, Bump{base, workspace_bytes, 0, 0, false}, as_stream(stream), B};
  return r.run(x, timesteps, out, H, W, D);
}

// ---- hipGraph replay --------------------------------------------------------------------------------------------
// SURVEY.md section 7 step 6 / hard part 5: at small batch (BASELINE configs[0]: first_n = 16) a forward is ~80
// launches of 5-30 us kernels and the host's launch path, not the GPU, sets the pace.  The first call with a given
// (x, timesteps, out, workspace, B, D, H, W) runs eagerly (it also performs every one-time hipFuncSetAttribute); the
// second captures the same launch sequence on a private stream into a hipGraph; later calls replay it with one
// hipGraphLaunch.  The caller keeps the four addresses stable (the Python mirror owns static I/O tensors).  Ordering
// against the caller's stream is by events, so the call remains asynchronous like ddpm_unet_forward3d.
extern "C" int ddpm_unet_forward_graphed(ddpm_unet *h, const float *x, const int64_t *timesteps, float *out, int B,
                                         int D, int H, int W, void *workspace, size_t workspace_bytes,
                                         ddpm_stream_t stream) {
  DDPM_CHECK_ARG(h && x && timesteps && out && workspace, "unet_forward_graphed: NULL argument");
  if (g_prof_on)  // per-kernel hipEvents are not graph nodes: profile the eager path
    return ddpm_unet_forward3d(h, x, timesteps, out, B, D, H, W, workspace, workspace_bytes, stream);
  // a captured graph bakes in WHICH kernels ran: after ddpm_set_split_f16 / ddpm_reload_env the old captures are stale (the
  // trainer's fp32 re-run of a flagged batch would otherwise replay the split-f16 kernels it is trying to rule out)
  if (h->graph_epoch != switch_epoch()) {
    h->drop_graphs();
    h->graph_epoch = switch_epoch();
  }
  const GraphKey key(x, timesteps, out, workspace, B, D, H, W);
  GraphEntry &e = h->graphs[key];
  e.calls += 1;
  if (e.calls == 1)
    return ddpm_unet_forward3d(h, x, timesteps, out, B, D, H, W, workspace, workspace_bytes, stream);
  hipStream_t cs = as_stream(stream);
  if (!h->gstream) {
    if (hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess) {
      set_error("unet_forward_graphed: cannot create the replay stream / events");
      return DDPM_EINVAL;
    }
  }
  if (!e.exec) {
    hipError_t rc = hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal);
    if (rc != hipSuccess) {
      set_error("unet_forward_graphed: hipStreamBeginCapture: %s", hipGetErrorString(rc));
      return (int)rc;
    }
    const int frc = ddpm_unet_forward3d(h, x, timesteps, out, B, D, H, W, workspace, workspace_bytes, h->gstream);
    rc = hipStreamEndCapture(h->gstream, &e.graph);
    if (frc != 0 || rc != hipSuccess || !e.graph) {
      if (e.graph) (void)hipGraphDestroy(e.graph);
      e.graph = nullptr;
      h->graphs.erase(key);
      if (frc == 0) set_error("unet_forward_graphed: hipStreamEndCapture: %s", hipGetErrorString(rc));
      return frc ? frc : (int)rc;
    }
    rc = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
    if (rc != hipSuccess) {
      set_error("unet_forward_graphed: hipGraphInstantiate: %s", hipGetErrorString(rc));
      (void)hipGraphDestroy(e.graph);
      h->graphs.erase(key);
      return (int)rc;
    }
  }
  hipError_t rc = hipEventRecord(h->ev_in, cs);
  if (rc == hipSuccess) rc = hipStreamWaitEvent(h->gstream, h->ev_in, 0);
  if (rc == hipSuccess) rc = hipGraphLaunch(e.exec, h->gstream);
  if (rc == hipSuccess) rc = hipEventRecord(h->ev_out, h->gstream);
  if (rc == hipSuccess) rc = hipStreamWaitEvent(cs, h->ev_out, 0);
  if (rc != hipSuccess) {
    set_error("unet_forward_graphed: replay: %s", hipGetErrorString(rc));
    return (int)rc;
  }
  return 0;
}

extern "C" int ddpm_unet_num_graphs(const ddpm_unet *h) {
  int n = 0;
  if (h)
    for (auto &kv : h->graphs) n += kv.second.exec != nullptr;
  return n;
}
