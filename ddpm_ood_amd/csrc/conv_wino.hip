// conv_wino.hip -- 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the fp32 MFMA pipe.
//
// Same fused op as conv_mfma.hip's 9-tap kernel (GroupNorm-affine + SiLU prologue, virtual concat, bias /
// temb / residual epilogue; reference call site /root/reference/src/trainers/reconstruct.py:151-153) with
// 2.25x fewer multiplies: every 2x2 output tile is  Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A  where d_c is
// the 4x4 input patch of channel c.  The sum over channels at each of the 16 transform positions xi is a GEMM,
//     M_xi[cout][tile] = sum_c U_xi[cout][c] * V_xi[c][tile],
// and runs on v_mfma_f32_32x32x2_f32 exactly like the direct kernel: A = U_xi (weights, pre-transformed on the
// device by wino_pack_kernel), B = V_xi (input patches transformed while they are staged into LDS).
// fp32 throughout.  Rounding differs from the direct form (the transforms add values before multiplying):
// measured max error of a 512-channel layer vs fp64 is 1.6e-6 against 0.9e-6 for the direct fp32 conv
// (DESIGN.md section 3.3), far inside the 1e-4 parity bar.
//
// Workgroup = 64 output channels x 64 tiles (256 output pixels: whole tile rows of one image, or several
// whole images), 8 waves as 2 (cout) x 2 (tile) x 2 (transform rows): a wave owns 32 couts x 32 tiles at 8 of
// the 16 positions = 8 accumulator tiles = 128 AGPRs, two waves per SIMD.  (A first version with 4 waves x 16
// positions = 256 AGPRs made the output transform lane-local but ran one wave per SIMD: the ~1000 staging
// instructions per chunk could not hide under that wave's own 64 MFMAs and the kernel only matched the direct
// one.)  Input channels advance in chunks of 8; LDS is double-buffered ([16][8][64] U + [16][8][64] V = 64 KB
// per buffer): while the 32 MFMAs of chunk q run, the wave commits its share of chunk q + 1 (activation +
// B^T d B applied here) and issues the loads of chunk q + 2.  One barrier per chunk.  The output transform is
// linear in M, so each wave transforms its own two rows and the pair sums through LDS.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef WINO_TRACE
__device__ unsigned long long g_wino_trace[4];
extern "C" int ddpm_debug_wino_trace(unsigned long long *out, int reset) {
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_trace), 32);
}
#endif

namespace ddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kWT = 64;    // tiles per workgroup
constexpr int kWK = 64;    // output channels per workgroup
constexpr int kWC = 8;     // input channels per chunk
constexpr int kWUF = 16 * kWC * kWK;  // U floats per chunk and cout tile (8192)
constexpr int kWVF = 16 * kWC * kWT;  // V floats per chunk (8192)

// The 16 accumulator tiles (256 registers) must live in the AGPR half of the unified register file: with the
// builtin hipcc keeps them in arch VGPRs, funnels every MFMA through a[0:15] and spills 1.2 KB per lane.
// The "+a" constraint pins each accumulator to its own AGPR tuple; successive MFMAs never share one (16 apart),
// so no wait states are needed between them (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void mfma_agpr(f32x16 &c, float a, float b) {
  asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

struct WinoGeom {
  int TWc, THr;     // tile columns / rows per image (Wo / 2, Ho / 2)
  int TI, TR;       // images per workgroup, tile rows per workgroup (per image)
  int TPI;          // workgroups per image (0 when a workgroup holds several whole images)
  int ntiles;       // workgroups along the tile axis
  int Cin, nchunks, HW;
  int PW, PCH;      // pixel tile in LDS: padded row length (Wo + 2), floats per channel (TI * (2 TR + 2) rows)
  int NRI;          // staging rounds of 64 pixels per image of the workgroup (TI * NRI <= 6)
};

static bool wino_geom(const ddpm_conv_desc &d, WinoGeom &g) {
  const int Cin = d.C1 + d.C2;
  if (d.ksize != 3 || d.mode != DDPM_CONV_NORMAL || d.Di > 1 || d.Do > 1 || d.accumulate || d.out_act) return false;
  if (d.act == DDPM_ACT_RELU) return false;
  if (Cin % kWC || (d.C2 > 0 && d.C1 % kWC) || d.Cout % kWK) return false;
  if ((d.Ho & 1) || (d.Wo & 1)) return false;
  // patches are fetched with 32-bit buffer offsets (12-bit immediate for the column)
  if ((double)d.B * (d.C1 > d.C2 ? d.C1 : d.C2) * d.Ho * d.Wo * 4 >= 2147483648.0) return false;
  g.TWc = d.Wo / 2;
  g.THr = d.Ho / 2;
  const int per_img = g.TWc * g.THr;
  if (per_img >= kWT) {
    if (kWT % g.TWc) return false;
    g.TI = 1;
    g.TR = kWT / g.TWc;
    if (g.THr % g.TR) return false;
    g.TPI = g.THr / g.TR;
    g.ntiles = d.B * g.TPI;
  } else {
    if (kWT % per_img) return false;
    g.TI = kWT / per_img;
    g.TR = g.THr;
    g.TPI = 0;
    g.ntiles = (d.B + g.TI - 1) / g.TI;
  }
  g.Cin = Cin;
  g.nchunks = Cin / kWC;
  g.HW = d.Ho * d.Wo;
  g.PW = d.Wo + 2;
  g.PCH = g.TI * (2 * g.TR + 2) * g.PW;
  const int rows = 2 * g.TR + 2 < d.Ho ? 2 * g.TR + 2 : d.Ho;  // in-image rows a workgroup reads, at most
  g.NRI = (rows * d.Wo + 63) / 64;
  if (g.TI * g.NRI > 6) return false;
  if ((2 * (kWUF + kWVF) + 2 * kWC * g.PCH + 64) * sizeof(float) > 160 * 1024) return false;
  return true;
}

bool conv_wino_supported(const ddpm_conv_desc &d) {
  static const bool enabled = !(getenv("DDPM_CONV_WINOGRAD") && atoi(getenv("DDPM_CONV_WINOGRAD")) == 0);
  WinoGeom g;
  return enabled && d.w_wino != nullptr && !d.force_direct && wino_geom(d, g);
}

template <bool AFFINE, int NR>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const ddpm_conv_desc a, const WinoGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = kWUF + kWVF;  // floats per operand buffer: U then V, both [xi 16][k-pair 2][k parity 2][64][2]
  float *const P = smem + 2 * BUF;  // pixel tiles [2][8 channels][PCH] (zero-padded borders) + 64 dump floats
  const int PB = kWC * g.PCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  // 8 waves = 2 (cout block) x 2 (tile block) x 2 (transform rows {0,1} / {2,3}): two waves per SIMD, each wave
  // keeps 8 positions = 128 AGPRs
  const int cb = wave & 1, tb = (wave >> 1) & 1, hf = wave >> 2;
  const int kt = blockIdx.y;
  const bool silu = a.act == DDPM_ACT_SILU;

  int n0, r0;  // first image of the workgroup, first tile row inside it
  if (g.TPI > 0) {
    n0 = blockIdx.x / g.TPI;
    r0 = (blockIdx.x - n0 * g.TPI) * g.TR;
  } else {
    n0 = blockIdx.x * g.TI;
    r0 = 0;
  }

  // ---- staging roles ---------------------------------------------------------------------------------
  // Pixels (stage A): wave = channel `sc` of the chunk; its lanes walk the channel's in-image pixels of this
  // workgroup's rows (halo rows included) in NR rounds of 64, round k belonging to image n0 + k / NRI.  Every pixel
  // is loaded, normalised and activated ONCE and stored into the zero-bordered tile P; f32 MFMAs share the SIMD's
  // FMA hardware with the VALU (tools/ubench/mfma_valu_mix.hip: every VALU op next to an MFMA costs its ~2.4
  // cycles, an exp / rcp ~9), so activating the 16 elements of every overlapping 4x4 patch -- each pixel four
  // times -- cost a quarter of the kernel.
  // Patches (stage T): lane = tile `st`, wave = channel: 16 LDS reads of the 4x4 patch out of P (no masks: the
  // border is materialised), B^T d B, 16 LDS writes into the V image.
  const int sc = wave, st = lane;
  const int row_lo = max(0, 2 * r0 - 1), row_hi = min(a.Ho, 2 * (r0 + g.TR) + 1);  // real rows this workgroup reads
  const int npx = (row_hi - row_lo) * a.Wo;
  int pix[NR], pw[NR], nimg[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int ti = k / g.NRI, e = lane + 64 * (k - ti * g.NRI);
    const bool valid = ti < g.TI && e < npx && n0 + ti < a.B;
    const int row = row_lo + e / a.Wo, col = e % a.Wo;
    nimg[k] = min(n0 + ti, a.B - 1);
    pix[k] = valid ? (row * a.Wo + col) * 4 : (int)0x80000000;  // out of range: the buffer load returns 0
    pw[k] = valid ? sc * g.PCH + (ti * (2 * g.TR + 2) + row - (2 * r0 - 1)) * g.PW + col + 1 : 2 * PB + lane;
  }
  int tbase;  // patch origin of tile st inside a channel tile of P
  {
    const int per = g.TR * g.TWc;
    const int ti = st / per, rem = st - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tbase = sc * g.PCH + (ti * (2 * g.TR + 2) + 2 * tr) * g.PW + 2 * tc;
  }
  const int bytes1 = a.B * a.C1 * g.HW * 4, bytes2 = a.B * a.C2 * g.HW * 4;
  const __amdgpu_buffer_rsrc_t rs_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sh =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  int vzero;  // a zero the compiler cannot see through: keeps the uniform scale / shift loads on the vector
              // memory path (scalar loads share lgkmcnt with LDS and would force full LDS drains)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

  // ---- MFMA operands.  This wave's positions are xi = 8 hf + x, x = 0..7; its 32 MFMAs of a chunk are 16 pairs
  // p = (k-pair kp = p >> 3, position x = p & 7), the two MFMAs of a pair covering channels 4 kp + lhi and
  // 4 kp + 2 + lhi of the chunk.  Both operand images are [xi][kp][lhi][cout or tile 64][2], so a pair's A and B
  // values are one ds_read_b64 each (256 B/clk against ds_read_b32's 128).
  const int ub = (lhi * kWK + cb * 32 + l31) * 2;         // + ((x + 8 hf) * 2 + kp) * 2 * 64 * 2
  const int vb = kWUF + (lhi * kWT + tb * 32 + l31) * 2;  // likewise
  // The U tile of a chunk (32 KB, already in LDS order in global memory) is copied by LDS-DMA: 4 x 1 KB per wave,
  // no registers, no ds_write pass, and -- unlike loads into registers -- nothing in the loop has to wait for it
  // before the chunk's closing barrier, so the (in-order) vmcnt waits never drag the slow pixel loads along.
  const float *usrc = a.w_wino + (size_t)kt * g.nchunks * kWUF + wave * 4 * 256 + lane * 4;

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  // ---- staging registers, and the slices of staging work the chunk loop places between its MFMAs -----------
  float praw[NR], gs[NR], gh[NR], dreg[16], tt[16];
  const int last = g.nchunks - 1;  // chunk indices past the end are clamped: their staging lands in buffers
                                   // nobody reads, which keeps the loop body free of branches
  // quarter i of this wave's share of the U tile of chunk ch -> operand buffer at float offset nb
  auto dma_u = [&](int i, int ch, int nb) {
    __builtin_amdgcn_global_load_lds(usrc + (size_t)min(ch, last) * kWUF + i * 256,
                                     smem + nb + (wave * 4 + i) * 256, 16, 0, 0);
  };
  // stage L: round k of chunk ch -> registers
  auto load_px = [&](int k, int ch) {
    const int cg = min(ch, last) * kWC + sc;
    const bool first = cg < a.C1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2, 0x00020000);
    const int soff = first ? (nimg[k] * a.C1 + cg) * g.HW * 4 : (nimg[k] * a.C2 + cg - a.C1) * g.HW * 4;
    praw[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pix[k], soff, 0));
    if (AFFINE) {
      const int goff = (nimg[k] * g.Cin + cg) * 4;
      gs[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, vzero, goff, 0));
      gh[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, vzero, goff, 0));
    }
  };
  // stage A: round k -> pixel tile `pb` (float offset of the P buffer)
  auto activate_px = [&](int k, int pb) {
    float v = praw[k];
    if (AFFINE) v = v * gs[k] + gh[k];
    const float sv = silu_fast(v);
    P[pb + pw[k]] = silu ? sv : v;
  };
  // stage T: patch row i out of pixel tile `pb`
  auto read_patch = [&](int i, int pb) {
    const float *p = P + pb + tbase + i * g.PW;
#pragma unroll
    for (int j = 0; j < 4; ++j) dreg[4 * i + j] = p[j];
  };
  // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows first ...
  auto row_transform = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tt[0 * 4 + j] = dreg[0 * 4 + j] - dreg[2 * 4 + j];
      tt[1 * 4 + j] = dreg[1 * 4 + j] + dreg[2 * 4 + j];
      tt[2 * 4 + j] = dreg[2 * 4 + j] - dreg[1 * 4 + j];
      tt[3 * 4 + j] = dreg[1 * 4 + j] - dreg[3 * 4 + j];
    }
  };
  // ... then the columns of row i, written to the four positions (i, 0..3) of the V image
  auto col_commit = [&](int i, int nb) {
    // channel sc = 4 kp + 2 e + lhi  ->  V [xi][kp][lhi][tile][e]
    float *vl = smem + nb + kWUF + (((sc >> 2) * 2 + (sc & 1)) * kWT + st) * 2 + ((sc >> 1) & 1);
    vl[(i * 4 + 0) * kWC * kWT] = tt[i * 4 + 0] - tt[i * 4 + 2];
    vl[(i * 4 + 1) * kWC * kWT] = tt[i * 4 + 1] + tt[i * 4 + 2];
    vl[(i * 4 + 2) * kWC * kWT] = tt[i * 4 + 2] - tt[i * 4 + 1];
    vl[(i * 4 + 3) * kWC * kWT] = tt[i * 4 + 1] - tt[i * 4 + 3];
  };

  // ---- prologue: zero borders; pixel tiles of chunks 0 and 1; U and V of chunk 0; registers for chunk 2 -------
  for (int i = tid; i < 2 * PB + 64; i += 512) P[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_u(i, 0, 0);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int k = 0; k < NR; ++k) load_px(k, c);
#pragma unroll
    for (int k = 0; k < NR; ++k) activate_px(k, c * PB);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) read_patch(i, 0);
  row_transform();
#pragma unroll
  for (int i = 0; i < 4; ++i) col_commit(i, 0);
#pragma unroll
  for (int k = 0; k < NR; ++k) load_px(k, 2);
  __syncthreads();

  // One chunk = 32 MFMA steps per wave, with three stages of staging in flight:
  //   T(q+1)  patches of chunk q + 1 out of P[(q+1) & 1] -> V of the other operand buffer
  //   A(q+2)  pixels of chunk q + 2 (registers) -> P[q & 1]          L(q+3)  pixel loads of chunk q + 3
  // plus the LDS-DMA of the U tile of chunk q + 1.  The work is cut into slices, one per MFMA step, and
  // sched_barriers pin every slice to its MFMA: left to itself the compiler emits the staging as one block between
  // two MFMAs and issues each operand read right before its use.
  //   step 0..3    U quarter s by DMA, patch row s -> registers
  //   step 6       row transform;   step 7..10  column transform + LDS write of row s - 7
  //   step 12..    activation of pixel round s - 12 -> P;   step 20..  loads of pixel round s - 20
  // The chunk closes with a counted vmcnt (the DMAs are older than the pixel loads, which stay in flight) and a raw
  // s_barrier: __syncthreads() would wait for vmcnt(0), i.e. for HBM, every chunk.
#ifdef WINO_TRACE
  unsigned long long tr_t0 = 0, tr_period = 0, tr_seg = 0;
#endif
  for (int q = 0; q < g.nchunks; ++q) {
    const int cbuf = (q & 1) * BUF;
    const int nb = BUF - cbuf;
    const int pb_t = ((q + 1) & 1) * PB, pb_a = (q & 1) * PB;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 av[3], bv[3];  // operand ring: three pairs
    auto load_pair = [&](int slot, int p) {
      const int off = (((p & 7) + 8 * hf) * 2 + (p >> 3)) * 2 * 64 * 2;
      av[slot] = *reinterpret_cast<const f2 *>(smem + cbuf + ub + off);
      bv[slot] = *reinterpret_cast<const f2 *>(smem + cbuf + vb + off);
    };
    auto slice = [&](int s) {
#ifdef WINO_TRACE
      if (s == 0) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (q > 0) tr_period += t - tr_t0;
        tr_t0 = t;
      }
      if (s == WINO_TRACE) tr_seg += __builtin_readcyclecounter() - tr_t0;
#endif
      if (s < 4) {
        dma_u(s, q + 1, nb);
        read_patch(s, pb_t);
      } else if (s == 6) {
        row_transform();
      } else if (s >= 7 && s < 11) {
        col_commit(s - 7, nb);
      } else if (s >= 12 && s < 12 + NR) {
        activate_px(s - 12, pb_a);
      } else if (s >= 20 && s < 20 + NR) {
        load_px(s - 20, q + 3);
      }
    };
    load_pair(0, 0);
    load_pair(1, 1);
    load_pair(2, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      mfma_agpr(acc[p & 7], av[p % 3][0], bv[p % 3][0]);
      slice(2 * p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_agpr(acc[p & 7], av[p % 3][1], bv[p % 3][1]);
      if (p + 3 < 16) load_pair(p % 3, p + 3);
      slice(2 * p + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NR * (AFFINE ? 3 : 1)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();

#ifdef WINO_TRACE
  if (tid == 0) {
    atomicAdd(&g_wino_trace[0], tr_period);
    atomicAdd(&g_wino_trace[1], tr_seg);
    atomicAdd(&g_wino_trace[2], (unsigned long long)g.nchunks);
  }
#endif
  // the last MFMAs are inline asm: give them their 16 passes before the accumulators are read back
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  The transform is linear in M, so each wave applies it
  // to its own two rows of M (hf = 0: rows 0, 1; hf = 1: rows 2, 3); the hf = 1 wave hands its four partial
  // outputs per (cout, tile) to its hf = 0 partner through LDS (the operand buffers are free by now).
  float y[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ma = acc[j][r], mb = acc[4 + j][r];  // rows 2 hf and 2 hf + 1
      s0[j] = hf == 0 ? ma + mb : ma;
      s1[j] = hf == 0 ? mb : -ma - mb;
    }
    y[r][0] = s0[0] + s0[1] + s0[2];
    y[r][1] = s0[1] - s0[2] - s0[3];
    y[r][2] = s1[0] + s1[1] + s1[2];
    y[r][3] = s1[1] - s1[2] - s1[3];
  }
  float *xch = smem + (wave & 3) * 64 * 64 + lane;  // [pair][r * 4 + x][lane]
  if (hf == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) xch[(r * 4 + x) * 64] = y[r][x];
  }
  __syncthreads();
  if (hf == 1) return;

  const int tq = tb * 32 + l31;  // this lane's tile
  const int per = g.TR * g.TWc;
  const int ti = tq / per, rem = tq - ti * per;
  const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
  const int n = n0 + ti;
  if (n < a.B) {
    const int co_base = kt * kWK + cb * 32 + 4 * lhi;
    const size_t pix = (size_t)(2 * (r0 + tr)) * a.Wo + 2 * tc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + (r & 3) + 8 * (r >> 2);
      float yy[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) yy[x] = y[r][x] + xch[(r * 4 + x) * 64];
      const size_t o = ((size_t)n * a.Cout + co) * g.HW + pix;
      float2 r0v = make_float2(0.f, 0.f), r1v = make_float2(0.f, 0.f);
      if (a.residual) {
        r0v = *reinterpret_cast<const float2 *>(a.residual + o);
        r1v = *reinterpret_cast<const float2 *>(a.residual + o + a.Wo);
      }
      const float add = a.bias ? a.bias[co] : 0.f;
      const float ca = a.chan_add ? a.chan_add[(size_t)n * a.chan_add_stride + co] : 0.f;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (a.bias) yy[x] += add;
        if (a.chan_add) yy[x] += ca;
      }
      if (a.residual) {
        yy[0] += r0v.x; yy[1] += r0v.y; yy[2] += r1v.x; yy[3] += r1v.y;
      }
      *reinterpret_cast<float2 *>(a.out + o) = make_float2(yy[0], yy[1]);
      *reinterpret_cast<float2 *>(a.out + o + a.Wo) = make_float2(yy[2], yy[3]);
    }
  }
}

int launch_conv_wino(const ddpm_conv_desc &d, hipStream_t s) {
  WinoGeom g;
  if (!d.w_wino || !wino_geom(d, g)) {
    set_error("conv_wino: unsupported shape");
    return DDPM_EINVAL;
  }
  const size_t lds = ((size_t)2 * (kWUF + kWVF) + 2 * kWC * g.PCH + 64) * sizeof(float);
  typedef void (*kern_t)(const ddpm_conv_desc, const WinoGeom);
  static const kern_t kerns[2][3] = {
      {conv_wino_kernel<false, 4>, conv_wino_kernel<false, 5>, conv_wino_kernel<false, 6>},
      {conv_wino_kernel<true, 4>, conv_wino_kernel<true, 5>, conv_wino_kernel<true, 6>}};
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 6; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 3][i % 3]),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int rounds = g.TI * g.NRI;
  const kern_t kern = kerns[d.gscale ? 1 : 0][rounds <= 4 ? 0 : rounds - 4];
  dim3 grid(g.ntiles, d.Cout / kWK);
  const double M = (double)d.B * g.HW;
  // algorithmic work = the direct convolution's: 2*M*Cout*Cin*9; 16/36 of it is executed
  const double flops = 2.0 * M * d.Cout * (double)g.Cin * 9;
  const double bytes = 4.0 * (M * g.Cin + M * d.Cout * (d.residual ? 2 : 1) + (double)d.Cout * g.Cin * 9);
  const char *kname = d.gscale ? "conv3x3_wino_gn_silu" : "conv3x3_wino";
  char kshape[160];
  if (g_prof_on && getenv("DDPM_PROF_SHAPES")) {
    snprintf(kshape, sizeof(kshape), "%s|%d+%d->%d@%dx%d", kname, d.C1, d.C2, d.Cout, d.Ho, d.Wo);
    kname = kshape;
  }
  ProfScope prof(s, kname, flops, bytes);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, d, g);
  DDPM_CHECK_LAUNCH();
  return 0;
}

// ---- weights: torch [Cout][Cin][3][3] -> U = G g G^T, packed as the LDS image the kernel's MFMAs read:
//   [cout tile 64][chunk 8][xi 16][kp 2][lhi 2][cout 64][e 2],  channel of the chunk = 4 kp + 2 e + lhi
// G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
__global__ void wino_pack_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin) {
  const int64_t total = (int64_t)Cout * Cin;
  const int nchunks = Cin / kWC;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), o = (int)(i / Cin);
    const float *w = src + i * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // G g
      t[0][j] = w[0 * 3 + j];
      t[1][j] = 0.5f * (w[0 * 3 + j] + w[1 * 3 + j] + w[2 * 3 + j]);
      t[2][j] = 0.5f * (w[0 * 3 + j] - w[1 * 3 + j] + w[2 * 3 + j]);
      t[3][j] = w[2 * 3 + j];
    }
    const int tile = o / kWK, k64 = o % kWK, ch = ci / kWC, cl = ci % kWC;
    const int lhi = cl & 1, kp = cl >> 2, e = (cl >> 1) & 1;
    float *d = dst + ((size_t)tile * nchunks + ch) * kWUF + ((kp * 2 + lhi) * kWK + k64) * 2 + e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // (G g) G^T
      d[(r * 4 + 0) * kWC * kWK] = t[r][0];
      d[(r * 4 + 1) * kWC * kWK] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      d[(r * 4 + 2) * kWC * kWK] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      d[(r * 4 + 3) * kWC * kWK] = t[r][2];
    }
  }
}

size_t wino_weight_floats(int Cout, int Cin) {
  if (Cout % kWK || Cin % kWC) return 0;
  return (size_t)16 * Cout * Cin;
}

int launch_pack_wino_weight(const float *w_raw, float *w_wino, int Cout, int Cin, hipStream_t s) {
  DDPM_CHECK_ARG(wino_weight_floats(Cout, Cin) != 0, "wino pack: Cout %% 64 or Cin %% 8 != 0");
  const int64_t total = (int64_t)Cout * Cin;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, s, w_raw, w_wino, Cout, Cin);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
