// conv_wino44r.hip -- the split-f16 Winograd F(4x4, 3x3) convolution of conv_wino44h.hip with its chunk pipeline rebuilt.  Round 5.
//
// Same fused op (GroupNorm-affine + SiLU prologue, virtual concat, bias / temb / residual epilogue, GroupNorm statistics of the
// output; reference call site /root/reference/src/trainers/reconstruct.py:151-153, layer list /root/reference/src/trainers/
// base.py:66-86), same arithmetic (every fp32 product from four exact f16 partial products, fp32 accumulate, fp32 transforms),
// same work item (64 couts x 32 tiles, nine 32x32 accumulator tiles per wave, eight pinned in a[0:127]), same packed weights,
// same output transform -- BIT-IDENTICAL results (tests/test_gpu_wino44h.py holds the two kernels to each other).  What changed is
// how an 8-channel chunk moves through the workgroup.  Two rounds of tuning left conv_wino44h_kernel at ~2 000 cycles per
// 12-position phase with every pipe under 50 % (MFMA 18 %): three barriers per chunk, each preceded by a drain of the LDS queue and
// of the U slot's LDS-DMA, each followed by a ~300-cycle wait for nine operand reads before the first MFMA can issue.  Here:
//
//   * U never touches LDS.  Wave (cb, pg) is the ONLY reader of its 3 x 3 x 1 KB of transformed weights per chunk (32 couts x
//     8 channels x {hi, lo} of one position = exactly one MFMA A operand): it loads them straight into registers
//     (buffer_load_dwordx4, the packed order is already the operand order), six jobs ahead, through a register ring.  No LDS-DMA
//     issue (100-150 cycles apiece, 24 per phase), no U ring (48 KB), no A-operand ds_read, no vmcnt(0) in front of a barrier.
//   * ONE barrier per chunk.  The freed LDS holds a V ring of two WHOLE chunks (2 x 36 KB = the epilogue's four exchange slabs),
//     so everything a chunk interval writes is read in the next one and nothing else orders the waves.
//   * A wave's chunk interval is two SEGMENTS: [the 18 MFMAs of chunk c, with the pixel staging in their shadow] and [its V task
//     of chunk c + 1, unsliced].  Waves 0-3 run the V task last, waves 4-7 (their SIMD partners) first: while one wave of a SIMD
//     sits in its MFMA segment the other one issues VALU and LDS work -- the pairing MI355X_MICROARCH.md describes ("Two waves
//     per SIMD"); waves 4-7 run at s_setprio 1 (the younger half loses every arbitration otherwise).
//   * All eight waves stage pixels (one channel of the chunk each); six of them (0, 1, 2, 4, 5, 6) run one V task per chunk =
//     the 12 positions of a transform-row pair for 16 tiles x 8 channels (the task conv_wino44h.hip splits in halves over two
//     phases), waves 3 and 7 none.
//   * Staging runs one interval ahead in a second register set (every shape but eight images per item): interval c requests the
//     pixels of chunk c + 3 behind its odd MFMA jobs and activates chunk c + 2 (GroupNorm affine + SiLU x 2^3 into the pixel ring)
//     behind its even ones -- no wave ever waits for a pixel load.
//   * Pixel-tile layout, patch reads and V stores are conflict-free and wide (w44r_relayout below): a patch row is one
//     ds_read_b128 + one ds_read_b64, a position's V plane of 16 tiles is 256 lane-linear bytes written by ds_write_addtid_b32.
// DESIGN.md 3.13 has the measurements (-12.8 % per launch at B = 1 024, -17 % at B = 128 against conv_wino44h.hip, same box),
// the cycle budget of an interval and everything that was tried and dropped (tools/w44r_abl.sh builds the variants,
// tools/w44r_probe.py reads the cycle stamps of a -DW44R_PROBE build).
//
// LDS: V ring 2 x [row pair 3][position 12][plane 2][tile 32][8 ch f16] + pixel ring of four 4-channel half-tiles.
#include "wino44h_common.h"

namespace ddpm {

namespace {

constexpr int kVCB = 3 * kVSB;  // bytes of one V slot: a whole chunk (36 864)
static_assert(2 * kVCB == kRINGF * 4, "the V ring is the epilogue's four exchange slabs");

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

// the same six positions through ds_write_addtid_b32 (address = M0 + offset + 4 lane, no address VGPR, twice the rate of
// ds_write_b32): with lane = 4 tile + channel pair a position's plane of 16 tiles x 8 channels is 256 lane-linear bytes.  m0base:
// wave-uniform byte address of the task's block (V slot + row pair + tile half).  (M0 is not otherwise used by this kernel: no
// LDS-DMA, no movrel; the compiler sets it right before any use of its own.)
template <int O>
__device__ __forceinline__ void v_store_row_addtid(int m0base, const uint32_t (&hi6)[6], const uint32_t (&lo6)[6]) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" ::"s"(m0base));  // (SALU write of M0 -> add-TID LDS instruction: one wait state)
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(hi6[q]), "n"((2 * (O + q)) * (kT * 16)) : "memory");
    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(lo6[q]), "n"((2 * (O + q) + 1) * (kT * 16)) : "memory");
  }
}

// a wave-uniform pointer the compiler has moved to VGPRs (SGPR pressure) back into SGPRs for the scalar-load asm statements
template <class T>
__device__ __forceinline__ const T *uniform_ptr(const T *p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const T *>(((uint64_t)hi << 32) | lo);
}

// six positions (O .. O + 5 of a row pair's 12) of a lane's channel pair into the V slot: hi plane at 2 pos, lo plane at 2 pos + 1
template <int O>
__device__ __forceinline__ void v_store_row(int vwa, const uint32_t (&hi6)[6], const uint32_t (&lo6)[6]) {
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(vwa), "v"(hi6[q]), "n"((2 * (O + q)) * (kT * 16)) : "memory");
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(vwa), "v"(lo6[q]), "n"((2 * (O + q) + 1) * (kT * 16)) : "memory");
  }
}

}  // namespace

// NRT = staging rounds of a wave per chunk (one channel); UIT = 0: one image per item, else units per image (4 or 1)
template <bool AFFINE, int NRT, int UIT, bool RES, bool D3 = false, bool UP = false>
__global__ __launch_bounds__(512, 2) void conv_wino44r_kernel(const ddpm_conv_desc a, const W44HGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool ONEIMG = UIT == 0;
  constexpr int NGS = ONEIMG ? 1 : NRT / UIT;        // images per item = GroupNorm scale / shift pairs per channel
  constexpr int GD = ONEIMG ? NRT : UIT;             // consecutive rounds that belong to one image
  float *const P = smem + kRINGF;                    // pixel ring: 4 half-tiles of [4 channels][PCH] + 1 + 64 dump floats

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cb = wave & 1, pg = wave >> 1;  // MFMA role: cout block, position group (positions 3 pg .. 3 pg + 2 of a row pair)
#ifndef W44R_ROLE
#define W44R_ROLE 0
#endif
  const bool lateprod = W44R_ROLE == 0 ? wave < 4 : W44R_ROLE == 1 ? (wave & 1) == 0 : ((wave >> 1) & 1) == 0;  // waves 0-3: V task at the END of a chunk interval; waves 4-7 (SIMD partners): first
  const bool silu = a.act == DDPM_ACT_SILU;

  // ---- this workgroup's items (as conv_wino44h.hip)
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
  const int istep = g.rev ? -g.TI : g.TI;  // images from an item to the next one of this workgroup
  const int NCHs = g.NCH / g.S, ch_lo = split * NCHs;  // this workgroup's chunk range (even length)
  float *const outp = a.out + (size_t)split * g.pstride;

  f32x16 acc8;  // tiles 0..7: a[0:127] by name (mfma_pin)
  reserve_agprs();

  // ---- U: wave (cb, pg), job (row pair t, i): position 3 pg + i of the pair, planes {hi, lo} x 32 couts x 8 channels = 1 KB at
  //   slot (chunk, t) * kUSB + (2 (3 pg + i) + plane) * 1024 + (32 cb + cout) * 16       [packed order = MFMA A operand order]
  const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(a.w_wino44h), 0, (int)((size_t)kX * a.Cout * g.Cin * g.nkd_w * 4), 0x00020000);
  const int ukt = ((kt * g.nkd_w + g.kd0) * g.NCHc + ch_lo) * 3;  // U slot index of this workgroup's chunk 0, row pair 0
  int ua = (3 * pg * 2 + lhi) * (kK * 16) + (cb * 32 + l31) * 16;
  // ---- V (LDS bytes): slot (chunk & 1) * kVCB + t * kVSB + (2 position + plane) * 512 + tile * 16
  int va = 3 * pg * (2 * kT * 16) + l31 * 16;

  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto zero_accumulators = [&]() __attribute__((always_inline)) {
    zero_pinned_tiles();
    float z;  // (a literal zero vector is materialised THROUGH a0..a15 by hipcc: an opaque zero keeps it in arch VGPRs)
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    acc8 = f32x16{z, z, z, z, z, z, z, z, z, z, z, z, z, z, z, z};
  };

#ifdef W44R_PROBE
  if (blockIdx.x == 0 && lane == 0 && a.scratch) reinterpret_cast<unsigned long long *>(a.scratch)[512 + wave * 32 + 31] = __builtin_readcyclecounter();
#endif
  // ---- borders are zeroed once (pixel writes only ever touch in-image pixels): behind the first item's first requests, whose
  // memory round trip the zeroing then hides
  auto zero_borders = [&]() __attribute__((always_inline)) {
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < g.HS; i += 512) *reinterpret_cast<v4f *>(P + 4 * i) = z4;  // (4 HS floats; P is 16-byte aligned)
    __syncthreads();
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

#ifdef W44R_PROBE  // timing experiment: cycle stamps of workgroup 0, every wave, chunk intervals 4..7 of the first item -> desc.scratch
  int probe_cc = -100;
  int n_idx_probe = 0;
#define W44R_STAMP(i)                                                                                                  \
  if (blockIdx.x == 0 && probe_cc >= 4 && probe_cc < 8 && lane == 0 && a.scratch)                                      \
    reinterpret_cast<unsigned long long *>(a.scratch)[((wave * 4 + (probe_cc - 4)) * 16 + (i)) & 511] = __builtin_readcyclecounter();
// phase stamps of the first item (fill, first V task, chunk loop, output-transform passes): slot 512 + 32 wave + i
#define W44R_FSTAMP(i)                                                                                                 \
  if (blockIdx.x == 0 && n_idx_probe == 0 && lane == 0 && a.scratch)                                                    \
    reinterpret_cast<unsigned long long *>(a.scratch)[512 + wave * 32 + (i)] = __builtin_readcyclecounter();
#else
#define W44R_STAMP(i)
#define W44R_FSTAMP(i)
#endif
  W44R_FSTAMP(0)
  // ================================================================================================ V tasks
  // Waves 0, 1, 2, 4, 5, 6 = tasks q = 0..5: row pair q / 2, tile half q & 1.  lane = (tile of 16, channel pair j of 4); a task =
  // the 12 positions of the pair for the lane's two channels: column passes of both channels (12 patch-row reads, 96 VALU), four
  // row passes (48), twelve pair splits (48), 24 stores.
  // NEWLAY: lane = 4 tile + j and the pixel-tile layout of w44r_relayout() -- patch rows are a conflict-free ds_read_b128 and a
  // ds_read_b64, a position's V plane of 16 tiles is lane-linear (ds_write_addtid_b32).
#ifndef W44R_NEWLAY8
#define W44R_NEWLAY8 1  // 0: eight 8x8 images per item keep conv_wino44h.hip's layout and its 4-byte patch reads
#endif
  constexpr bool NEWLAY = UIT != 1 || W44R_NEWLAY8;
  const int ptask = (wave & 3) == 3 ? -1 : wave - (wave >> 2);
  const int pst = (ptask & 1) * 16 + (NEWLAY ? lane >> 2 : lane & 15), pj = NEWLAY ? lane & 3 : lane >> 4;
  int tb0;  // pixel-ring offset (floats) of this lane's patch origin in channel 2 j of an EVEN chunk (half-tile j >> 1, plane 2 (j & 1))
  {
    const int per = g.TR * g.TWc;
    const int ti = pst / per, rem = pst - ti * per;
    const int tr = rem / g.TWc, tc = rem - tr * g.TWc;
    tb0 = (pj >> 1) * g.HS + 2 * (pj & 1) * g.PCH + ti * g.IS + 4 * tr * g.PW + 4 * tc;
  }
  int vw0 = pst * 16 + 4 * pj;  // V store (bytes): + slot + t * kVSB + (2 position + plane) * 512
  auto produce = [&](auto tc_, int cc) __attribute__((always_inline)) {  // V of stream chunk cc (slot cc & 1) from the pixel half-tiles 2 (cc & 1), 2 (cc & 1) + 1
    constexpr int t = decltype(tc_)::value;
    constexpr bool t0 = t == 0;
    constexpr float c1 = t0 ? -5.f : t == 1 ? -2.f : -0.5f, c2 = t0 ? 4.f : c1, bm = t0 ? 0.f : t == 1 ? 1.f : 2.f;
    asm volatile("" : "+v"(tb0), "+v"(vw0));  // keep the per-lane bases out of LICM's reach
    const int pb0 = tb0 + (cc & 1) * 2 * g.HS;
    float cA[2][6], cB[2][6];
    // column pass of one channel: rows (0, 5): A = d4 - 5 d2 + 4 d0, B = d5 - 5 d3 + 4 d1;  rows (1, 2): p = d4 - 4 d2, q = d3 - 4 d1,
    // A = p + q, B = p - q;  rows (3, 4): p = d4 - d2, q = d3 - d1, A = p + 2 q, B = p - 2 q   (conv_wino44h.hip's cstep, unsliced)
#ifdef W44R_VREADS_UPFRONT
    float r4[2][6], r2[2][6], r1[2][6], r3[2][6];
    if (!t0 && NEWLAY) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float *p = P + pb0 + c * g.PCH;
        auto rowu = [&](int r, float (&dst)[6]) __attribute__((always_inline)) {
          const v4f lo = *reinterpret_cast<const v4f *>(p + r * g.PW);
          const v2f_t hi = *reinterpret_cast<const v2f_t *>(p + r * g.PW + 4);
          dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = lo[2]; dst[3] = lo[3]; dst[4] = hi[0]; dst[5] = hi[1];
        };
        rowu(4, r4[c]); rowu(2, r2[c]); rowu(1, r1[c]); rowu(3, r3[c]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float *p = P + pb0 + c * g.PCH;
      float d4[6], d2[6], dx[6], dy[6], dz[6], dw[6];
      // rows 4, 2, 1 and: pair (0, 5) rows 0, 5, 3; pairs (1, 2), (3, 4) row 3 -- those pairs use rows 2 and 1 twice (same
      // operation order as conv_wino44h.hip's cstep, which re-reads them: bit-identical)
      auto row = [&](int r, float (&dst)[6]) __attribute__((always_inline)) {
        if (NEWLAY) {  // 16-byte aligned (w44r_relayout): columns 0-3 conflict-free by ds_read_b128, columns 4-5 by ds_read_b64
          // (as a second b128 with two dead floats hipcc overlapped the destination registers of consecutive rows and put a
          // full lgkmcnt(0) between them: eight exposed LDS latencies per task)
          const v4f lo = *reinterpret_cast<const v4f *>(p + r * g.PW);
          const v2f_t hi = *reinterpret_cast<const v2f_t *>(p + r * g.PW + 4);
          dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = lo[2]; dst[3] = lo[3]; dst[4] = hi[0]; dst[5] = hi[1];
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) dst[q] = p[r * g.PW + q];
        }
      };
#ifdef W44R_VREADS_UPFRONT
      if (!t0 && NEWLAY) {
#pragma unroll
        for (int q = 0; q < 6; ++q) d4[q] = r4[c][q], d2[q] = r2[c][q], dw[q] = r1[c][q], dy[q] = r3[c][q];
      } else
#endif
      {
        row(4, d4);
        row(2, d2);
        row(1, dw);
        row(t0 ? 5 : 3, dy);
      }
      if (t0) {
        row(0, dx);
        row(3, dz);
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) dx[q] = d2[q], dz[q] = dw[q];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float wa = __builtin_fmaf(c2, dx[q], __builtin_fmaf(c1, d2[q], d4[q]));
        const float qq = __builtin_fmaf(c2, dw[q], __builtin_fmaf(c1, dz[q], dy[q]));
        cA[c][q] = __builtin_fmaf(bm, qq, wa);
        cB[c][q] = t0 ? qq : __builtin_fmaf(-bm, qq, wa);
      }
    }
    int vwa = vw0 + (cc & 1) * kVCB + t * kVSB;
    asm volatile("" : "+v"(vwa));  // ONE address register + immediates
    float t0r[6], t1r[6];
    uint32_t hi6[6], lo6[6];
    bt6(cA[0], t0r);
    bt6(cA[1], t1r);
#pragma unroll
    for (int q = 0; q < 6; ++q) split_pair(t0r[q], t1r[q], hi6[q], lo6[q]);
    const int m0base = __builtin_amdgcn_readfirstlane((cc & 1) * kVCB + t * kVSB + (ptask & 1) * 256);
    if (NEWLAY) v_store_row_addtid<0>(m0base, hi6, lo6);
    else v_store_row<0>(vwa, hi6, lo6);
    bt6(cB[0], t0r);
    bt6(cB[1], t1r);
#pragma unroll
    for (int q = 0; q < 6; ++q) split_pair(t0r[q], t1r[q], hi6[q], lo6[q]);
    if (NEWLAY) v_store_row_addtid<6>(m0base, hi6, lo6);
    else v_store_row<6>(vwa, hi6, lo6);
  };
  auto produce_task = [&](int cc) __attribute__((always_inline)) {
    if (ptask < 0) return;
    const int t = ptask >> 1;
    if (t == 0) produce(I0{}, cc);
    else if (t == 1) produce(I1{}, cc);
    else produce(I2{}, cc);
  };

  // ================================================================================================ pixel staging
  // every wave: one channel of the 8-channel chunk (half-chunk wave >> 2, channel wave & 3 of it); round k = 64 pixels
  const int sc = wave & 3, phalf = wave >> 2;
  const int row_lo = max(0, 4 * r0 - 1), row_hi = min(a.Ho, 4 * (r0 + g.TR) + 1);
  const int npx = (row_hi - row_lo) * a.Wo;
  // QUAD (every shape but eight images per item and the Upsample form): a lane stages FOUR consecutive pixels of a row per round
  // (one buffer_load_dwordx4: a round = 256 pixels), a third of the vector-memory instructions of the 64-pixel rounds -- their
  // issue (~45 cycles apiece in this kernel) was a quarter of the MFMA segment.  NR = rounds per channel and item.
#ifndef W44R_QUAD
#define W44R_QUAD 0  // measured: same speed to 1.5 % slower than the 64-pixel rounds (the loads' issue is not what the MFMA segment waits for)
#endif
  constexpr bool QUAD = W44R_QUAD && UIT != 1 && !UP;
  constexpr int NR = QUAD ? (ONEIMG ? 3 : 2) : NRT;
  constexpr int LP = QUAD ? 4 : 1;                              // pixels per lane and round
  const int dump = 4 * g.PCH + 1 + LP * lane;  // relative to the half-tile (w44r_relayout keeps 256 dump floats behind the planes)
  int pix0, pw0, pixL = 0, pwL = 0;
  {
    const int e0 = LP * lane;
    const bool valid = e0 < npx;
    auto src_of = [&](int row, int col) __attribute__((always_inline)) { return UP ? ((row >> 1) * (a.Wo >> 1) + (col >> 1)) * 4 : (row * a.Wo + col) * 4; };
    pix0 = valid ? src_of(row_lo + e0 / a.Wo, e0 % a.Wo) : (int)0x80000000;  // out of range: the load returns 0
    pw0 = valid ? sc * g.PCH + (row_lo + e0 / a.Wo - (4 * r0 - 1)) * g.PW + e0 % a.Wo + 1 : dump;
    if (ONEIMG) {
      const int eL = e0 + 64 * LP * (NR - 1);
      const bool vL = eL < npx;
      pixL = vL ? src_of(row_lo + eL / a.Wo, eL % a.Wo) : (int)0x80000000;
      pwL = vL ? sc * g.PCH + (row_lo + eL / a.Wo - (4 * r0 - 1)) * g.PW + eL % a.Wo + 1 : dump;
    }
  }
  const int prs = (64 * LP / a.Wo) * g.PW;  // pixel-tile floats between a lane's pixels of consecutive rounds of one image
  constexpr int kRoundBytes = UP ? 64 : 256 * LP;
  constexpr int GDR = QUAD ? (ONEIMG ? NR : 1) : GD;  // consecutive rounds that belong to one image
  auto pix_of = [&](int k) __attribute__((always_inline)) { return ONEIMG ? (k == NR - 1 ? pixL : pix0 + kRoundBytes * k) : pix0 + kRoundBytes * (k % GDR); };
  auto pw_of = [&](int k) __attribute__((always_inline)) { return ONEIMG ? (k == NR - 1 ? pwL : pw0 + k * prs) : pw0 + (k % GDR) * prs + (k / GDR) * g.IS; };
  const int bytes1 = a.B * a.C1 * (D3 ? g.CS : UP ? g.HWin : g.HW) * 4, bytes2 = a.B * a.C2 * g.HW * 4;
  const __amdgpu_buffer_rsrc_t rs_sc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gscale), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sh =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gshift), 0, AFFINE ? a.B * g.Cin * 4 : 0, 0x00020000);
  int vzero;  // keeps the wave-uniform scale / shift loads on the vector memory path
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  // PACKED (eight images per item): the item's eight GroupNorm pairs of a channel live in ONE register pair, lane i = image i
  // (read back with v_readlane at the activation), instead of sixteen wave-uniform registers -- that is what lets this shape hold
  // two sets as well.
  constexpr bool PACKED = UIT == 1;
  constexpr int NGL = PACKED ? 1 : NGS;  // GroupNorm scale / shift registers per set
#ifndef W44R_DEEP8
#define W44R_DEEP8 1  // 0: eight images per item stage from ONE set (loads in front of the MFMA segment, activation behind it)
#endif
  // DEEP: TWO register sets.
  // Interval c requests chunk c + 3 into set (c + 1) & 1 and activates chunk c + 2 from set c & 1 (requested an interval earlier),
  // one round behind each MFMA job: nothing in a wave ever waits for a pixel load, and the loads' issue and the GroupNorm +
  // SiLU arithmetic hide behind the MFMAs.  Otherwise one set: loads in front of the MFMA segment, activation behind it.
  constexpr bool DEEP = UIT != 1 || W44R_DEEP8;
#ifndef W44R_SETS
#define W44R_SETS 2
#endif
  // DEEP3 (W44R_SETS=3): THREE sets, requests TWO intervals ahead (interval c requests chunk c + 4): the set of a chunk is its
  // running index over the workgroup's whole chunk stream modulo 3, so the two-chunk loop body exists in three rotations
  constexpr bool DEEP3 = DEEP && W44R_SETS == 3;
#ifndef W44R_RELOAD
#define W44R_RELOAD 0  // measured: 6 % SLOWER (5 152 / 5 196 vs 4 864 / 4 875 us over the six layers, same box)
#endif
  // RELOAD (two sets): a round's registers are requested again -- for the chunk TWO intervals on -- right behind the activation that
  // consumed them, instead of filling the other set one interval ahead: twice the lead for the same registers
  constexpr bool RELOAD = DEEP && !DEEP3 && W44R_RELOAD;
  constexpr int NSET = DEEP3 ? 3 : DEEP ? 2 : 1;
  constexpr int kAhead = DEEP3 || RELOAD ? 4 : DEEP ? 3 : 2;  // interval c requests chunk c + kAhead
  using praw_t = std::conditional_t<QUAD, v4f, float>;
  praw_t praw[NSET][NR];
  float gs[NSET][NGL], gh[NSET][NGL];
  struct LoadCtx {  // wave-uniform addressing of one stream chunk's loads (SGPRs)
    __amdgpu_buffer_rsrc_t rs;
    int n_it, cx, cgl, cga, soff3;
    bool dok;
  };
  // stream chunk cc of the item that starts at image n_cur; past the item's last chunk: the next item's first chunks (the pixel
  // ring survives the output transform), behind the last item a harmless repeat
  auto load_prep = [&](int cc, int n_cur, bool has_next) __attribute__((always_inline)) {
    LoadCtx L;
    const bool nxt = cc >= NCHs && has_next;
    const int cl = nxt ? cc - NCHs : min(max(cc, 0), NCHs - 1);
    L.n_it = nxt ? n_cur + istep : n_cur;
    int cg = (ch_lo + cl) * kC + phalf * 4 + sc;
    L.cga = cg;
    L.soff3 = 0;
    L.dok = true;  // D3: the depth tap's slice lies inside the volume
    if (D3) {  // stream chunk -> (depth tap, channel chunk); image -> (batch item, slice); one image per item
      const int kdi = cl / g.NCHc;
      cg = (cl - kdi * g.NCHc) * kC + phalf * 4 + sc;
      const int ni = min(L.n_it, g.NIMG - 1);
      const int nb = ni / g.D, dsl = ni - nb * g.D + g.kd0 + kdi - 1;
      L.dok = dsl >= 0 && dsl < g.D;
      L.soff3 = ((nb * a.C1 + cg) * g.D + (L.dok ? dsl : 0)) * g.HW * 4;
    }
    const bool first = cg < a.C1;
    L.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(first ? a.in1 : a.in2), 0, first ? bytes1 : bytes2, 0x00020000);
    L.cx = first ? a.C1 : a.C2;
    L.cgl = first ? cg : cg - a.C1;
    return L;
  };
  auto load_round = [&](const LoadCtx &L, auto setc, int k) __attribute__((always_inline)) {
    constexpr int S = decltype(setc)::value;
    const int ni = min(L.n_it + (ONEIMG ? 0 : k / GDR), g.NIMG - 1);
    const int soff = D3 ? L.soff3 : (ni * L.cx + L.cgl) * (UP ? g.HWin : g.HW) * 4;
    const int voff = D3 && !L.dok ? (int)0x80000000 : pix_of(k);
    if constexpr (QUAD)
      praw[S][k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(L.rs, voff, soff, 0));
    else
      praw[S][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(L.rs, voff, soff, 0));
  };
  auto load_affine = [&](const LoadCtx &L, auto setc, int i) __attribute__((always_inline)) {
    constexpr int S = decltype(setc)::value;
    if (AFFINE) {
      const int nb = min(L.n_it + i, g.NIMG - 1);
      const int goff = (nb * g.Cin + L.cga) * 4;
      // PACKED: lane l = image min(n_it + (l & 7), NIMG - 1)
      const int voff = PACKED ? max(0, min(lane & 7, g.NIMG - 1 - L.n_it)) * (g.Cin * 4) : vzero;
      gs[S][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sc, voff, goff, 0));
      gh[S][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sh, voff, goff, 0));
    }
  };
  auto load_stage = [&](auto setc, int cc, int n_cur, bool has_next) __attribute__((always_inline)) {
    const LoadCtx L = load_prep(cc, n_cur, has_next);
#pragma unroll
    for (int k = 0; k < NR; ++k) load_round(L, setc, k);
#pragma unroll
    for (int i = 0; i < NGL; ++i) load_affine(L, setc, i);
  };
  // pixel value x 2^3 (2^0 without prologue): the transform's output is the pre-scaled V
  auto activate_round = [&](auto setc, int cc, int k) __attribute__((always_inline)) {
    constexpr int S = decltype(setc)::value;
    float *const Pr = P + (2 * (cc & 1) + phalf) * g.HS + pw_of(k);
    float sa = 0.f, sb = 0.f, ta = 0.f, tb = 0.f;
    if (AFFINE) {
      if (PACKED) {
        sa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gs[S][0]), k / GDR));
        sb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gh[S][0]), k / GDR));
      } else {
        sa = gs[S][k / GDR];
        sb = gh[S][k / GDR];
      }
      ta = -1.44269504088896341f * sa;
      tb = -1.44269504088896341f * sb;
    }
    auto act = [&](float x) __attribute__((always_inline)) {
      if (AFFINE) {
        const float v = __builtin_fmaf(x, sa, sb);
        const float t = __builtin_fmaf(x, ta, tb);
        return (kVScale * v) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
      }
      return kVScaleRaw * (silu ? silu_fast(x) : x);
    };
    if constexpr (QUAD) {
      const v4f x = praw[S][k];
#pragma unroll
      for (int i = 0; i < 4; ++i) Pr[i] = act(x[i]);
    } else {
      Pr[0] = act(praw[S][k]);
    }
  };
  auto activate_stage = [&](auto setc, int cc) __attribute__((always_inline)) {
    asm volatile("" : "+v"(pix0), "+v"(pw0), "+v"(pixL), "+v"(pwL));
#pragma unroll
    for (int k = 0; k < NR; ++k) activate_round(setc, cc, k);
  };

  // ================================================================================================ MFMA jobs
  // job jj = 3 t + i of a chunk: accumulator tile jj += U[pair t, position 3 pg + i] (Vh + Vl).  A comes from the register ring
  // (six jobs ahead; 9 jobs per chunk, ring of 6: the indices repeat every two chunks, hence the two-chunk loop body), Bh / Bl by
  // ds_read_b128 one job ahead.
#ifndef W44R_AR
#define W44R_AR 6  // A-operand ring depth: a divisor of 18 (static indices over the two-chunk body)
#endif
  constexpr int kAR = W44R_AR;
  h8 Ar[kAR];
  auto load_a = [&](int cl, int jj) __attribute__((always_inline)) {  // job jj of chunk cl (clamped into the item: behind the last chunk a harmless repeat)
    const int t = jj / 3, i = jj - 3 * t;
    const int c2 = min(cl, NCHs - 1);
    const v4i_t v = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(rs_u, ua + i * (2 * kK * 16), (ukt + 3 * c2 + t) * kUSB, 0));
    return __builtin_bit_cast(h8, v);
  };
  auto mfma_chunk = [&](auto parc, int cl, auto &&slice) __attribute__((always_inline)) {  // PAR = chunk parity inside the two-chunk body: ring indices are constants
    constexpr int PAR = decltype(parc)::value;
    asm volatile("" : "+v"(ua), "+v"(va));
    const int vb = va + (cl & 1) * kVCB;
    h8 Bh[2], Bl[2];
#ifndef W44R_BVIS
#define W44R_BVIS 0  // (1: measured equal, 4 808 vs 4 810 us over the six layers; the hand-counted form is the one the full suite validated)
#endif
#if W44R_BVIS
    // B operands as PLAIN LDS loads: the compiler counts them together with the slices' pixel-ring stores and puts the exact
    // lgkmcnt in front of each MFMA.  (Hand-counted waits were off by the stores: "all but the next job's two reads" also
    // waited for the next job's FIRST read whenever a slice had stored in between -- one exposed LDS latency per job.)
    const char *const smbc = reinterpret_cast<const char *>(smem);
#define W44R_BREAD(off) (*reinterpret_cast<const h8 *>(smbc + vb + (off)))
#else
#define W44R_BREAD(off) lds_b128(vb, off)
#endif
    Bh[0] = W44R_BREAD(0);
    Bl[0] = W44R_BREAD(kT * 16);
#pragma unroll
    for (int jj = 0; jj < 9; ++jj) {
      if (jj < 8) {  // next job's B operands: row pair (jj + 1) / 3, position 3 pg + (jj + 1) % 3 (offsets fold to immediates)
        Bh[(jj + 1) & 1] = W44R_BREAD(((jj + 1) / 3) * kVSB + 2 * ((jj + 1) % 3) * kT * 16);
        Bl[(jj + 1) & 1] = W44R_BREAD(((jj + 1) / 3) * kVSB + (2 * ((jj + 1) % 3) + 1) * kT * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int ri = (PAR * 9 + jj) % kAR;  // ring index of this job's A
#if W44R_BVIS
      if (jj == 8) {  // the ninth tile (arch VGPRs): both MFMAs and their completion in one statement
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %3, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7"
                     : "+v"(acc8) : "v"(Ar[ri]), "v"(Bh[jj & 1]), "v"(Bl[jj & 1]));
      } else {
        mfma_pin(jj, Ar[ri], Bh[jj & 1]);
        __builtin_amdgcn_sched_barrier(0);
        slice(jj, 0);  // in the shadow of the first MFMA (the second one, on the same tile, cannot issue before it has finished)
        __builtin_amdgcn_sched_barrier(0);
        mfma_pin(jj, Ar[ri], Bl[jj & 1]);
      }
#else
      if (jj == 8) {
        mfma_v_pair_wait0(acc8, Ar[ri], Bh[jj & 1], Bl[jj & 1]);
      } else {
        // outstanding LDS reads, oldest first: Bh(jj), Bl(jj), Bh(jj + 1), Bl(jj + 1) (+ whatever the slices issued: newer)
        mfma_pin_wait<3>(jj, Ar[ri], Bh[jj & 1]);
        __builtin_amdgcn_sched_barrier(0);
        slice(jj, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        mfma_pin(jj, Ar[ri], Bl[jj & 1]);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      // the A operand six jobs ahead (this ring register is free: the MFMAs above have read it)
      Ar[ri] = load_a(jj + kAR < 9 ? cl : cl + 1, (jj + kAR) % 9);
      if (jj == 8) slice(jj, 0);
      slice(jj, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef W44R_BREAD
  };

  // static priority for the second-dispatched half of the workgroup: at equal priority the older wave of a SIMD wins every VALU
  // arbitration and waves 4-7 ran every segment ~15 % slower than their partners (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
  if (wave >= 4) asm volatile("s_setprio 1");
  int grot = 0;  // DEEP3: running index (over this workgroup's items) of the next interval's chunk, modulo 3
  for (int n_idx = 0; n_idx < nitems; ++n_idx) {
    const int n_cur = g.rev ? n_end - (n_idx + 1) * g.TI : n_first + n_idx * g.TI;
    const bool first_item = n_idx == 0 || !g.xitem;
    const bool has_next = g.xitem && n_idx + 1 < nitems;
    // ---- fill.  A workgroup's first item stages chunks 0 and 1 from scratch; later items find them in the pixel ring (staged
    // during the previous item's last two chunk intervals: the ring survives the output transform) and only owe V of chunk 0.
#ifdef W44R_PROBE
    n_idx_probe = n_idx;
#endif
    W44R_FSTAMP(1)
    if (first_item) {
      grot = 0;
      constexpr bool FILL2 = DEEP && !DEEP3 && !RELOAD;  // chunks 0 and 1 requested together, into their own sets: one memory round trip
      load_stage(I0{}, 0, n_cur, has_next);
      if (FILL2) load_stage(I1{}, 1, n_cur, has_next);
      if (n_idx == 0) zero_borders();
      W44R_FSTAMP(2)
      activate_stage(I0{}, 0);
      W44R_FSTAMP(3)
      if (FILL2) {
        load_stage(I0{}, 2, n_cur, has_next);
        activate_stage(I1{}, 1);
      } else {
        load_stage(I0{}, 1, n_cur, has_next);
        activate_stage(I0{}, 1);
      }
      W44R_FSTAMP(4)
      // (a later item finds these in flight: requested in its predecessor's last interval(s))
      if (DEEP3) {  // first item: running index = chunk index
        load_stage(I2{}, 2, n_cur, has_next);
        load_stage(I0{}, 3, n_cur, has_next);
      } else if (RELOAD) {
        load_stage(I0{}, 2, n_cur, has_next);
        load_stage(I1{}, 3, n_cur, has_next);
      }
      barrier();
    } else if (RELOAD) {
      load_stage(I1{}, 3, n_cur, has_next);  // (chunk 2 has been in flight since the previous item's last-but-one interval)
    }
    W44R_FSTAMP(5)
#pragma unroll
    for (int jj = 0; jj < kAR; ++jj) Ar[jj] = load_a(0, jj);
    produce_task(0);
    zero_accumulators();
    W44R_FSTAMP(6)
    barrier();
    W44R_FSTAMP(7)
    // ---- chunk intervals.  Interval c of a wave: loads of chunk c + 2, the 18 MFMAs of chunk c, activation of chunk c + 2, and
    // its V task of chunk c + 1 -- waves 0-3 run that task LAST (before the interval's barrier), waves 4-7 FIRST (behind the
    // previous interval's barrier: their "interval" is shifted by one segment, so that the two waves of a SIMD are never in their
    // MFMA segments at the same time).  In program order both are: LMA(c); V task; with the barrier in front of the task (waves
    // 4-7, task of chunk c + 2) or behind it (waves 0-3, task of chunk c + 1).  Everything an interval writes (V slot, pixel
    // half-tiles) is read in the next one; the two-slot rings need nothing else.  Chunk NCHs is the next item's chunk 0: its V has
    // to wait for the output transform, which owns the V ring.
    const int pahead = lateprod ? 1 : 2;
    if (!lateprod) produce_task(1);
    auto interval = [&](auto parc, auto rotc, int cc) __attribute__((always_inline)) {
      constexpr int PAR = decltype(parc)::value;
      constexpr int ROT = decltype(rotc)::value;  // DEEP3: running chunk index of chunk cc, modulo 3
      using SA = std::integral_constant<int, DEEP3 ? (ROT + 2) % 3 : DEEP ? PAR : 0>;      // the set activated in this interval (chunk cc + 2)
      using SL = std::integral_constant<int, DEEP3 ? (ROT + 1) % 3 : RELOAD ? PAR : DEEP ? 1 - PAR : 0>;  // the set requested in this interval (chunk cc + kAhead)
#ifdef W44R_PROBE
      probe_cc = n_idx == 0 ? cc : -100;
#endif
      W44R_STAMP(0)
#ifndef W44R_LOADS_AT_V
#define W44R_LOADS_AT_V 0  // measured 3 % slower than requesting them behind the odd MFMA jobs (same-box A/B)
#endif
      constexpr bool kLoadsAtV = W44R_LOADS_AT_V != 0;
      const LoadCtx L = load_prep(cc + kAhead, n_cur, has_next);
      if (!DEEP) {
#pragma unroll
        for (int k = 0; k < NR; ++k) load_round(L, SL{}, k);
#pragma unroll
        for (int i = 0; i < NGL; ++i) load_affine(L, SL{}, i);
      }
      W44R_STAMP(1)
      if (DEEP) asm volatile("" : "+v"(pix0), "+v"(pw0), "+v"(pixL), "+v"(pwL));
      mfma_chunk(parc, cc, [&](int jj, int part) __attribute__((always_inline)) {
        if (!DEEP) return;
        // behind the EVEN jobs: two activation rounds of chunk cc + 2 at a time (two independent fma -> exp -> rcp chains
        // interleave; one round per job left each job waiting for a ~100-cycle dependent chain); behind the ODD jobs: the
        // requests of chunk cc + 3, three rounds at a time (+ the GroupNorm pairs)
        // part 0 runs between a job's two MFMAs, part 1 behind them.  Even jobs: two activation rounds of chunk cc + 2 (one per
        // part); odd jobs: the requests of chunk cc + 3 (three rounds: one + two) and the GroupNorm pairs.
        if (QUAD) {  // (a round = four pixels per lane: round q at job 2 q, its request at job 2 q + 1)
          if (part == 0) return;
          if ((jj & 1) == 0 && jj / 2 < NR) activate_round(SA{}, cc + 2, jj / 2);
          if (!kLoadsAtV) {
            if ((jj & 1) == 1 && jj / 2 < NR) load_round(L, SL{}, jj / 2);
            if (jj == 7) {
#pragma unroll
              for (int i = 0; i < NGL; ++i) load_affine(L, SL{}, i);
            }
          }
        } else if (RELOAD) {
          // even jobs: two activation rounds (one per part); odd jobs: the two rounds just consumed are requested again, for chunk cc + 4
          const int k0 = 2 * (jj / 2) + part;
          if ((jj & 1) == 0) {
#pragma unroll
            for (int kk = 0; kk < NRT; ++kk)
              if (kk == k0) activate_round(SA{}, cc + 2, kk);
          } else if (cc + 1 < NCHs) {
#pragma unroll
            for (int kk = 0; kk < NRT; ++kk)
              if (kk == k0) load_round(L, SL{}, kk);
          }
          // (an item's LAST interval requests nothing: its chunk cc + 4 is the next item's chunk 3, requested behind the output
          // transform instead -- only ONE set, the next item's chunk 2, is in flight across the transform, whose residual rows
          // and exchange temporaries need the registers)
          if (jj == 8 && part == 1 && cc + 1 < NCHs) {  // rounds 8 (and 9) were activated in this job: their requests and the GroupNorm pairs close the interval
#pragma unroll
            for (int kk = 8; kk < NRT; ++kk) load_round(L, SL{}, kk);
#pragma unroll
            for (int i = 0; i < NGL; ++i) load_affine(L, SL{}, i);
          }
        } else if ((jj & 1) == 0) {
          const int k = 2 * (jj / 2) + part;
          if (k < NRT) {
#pragma unroll
            for (int kk = 0; kk < NRT; ++kk)
              if (kk == k) activate_round(SA{}, cc + 2, kk);
          }
        } else if (!kLoadsAtV) {
#pragma unroll
          for (int k = 0; k < NRT; ++k)
            if (k / 3 == jj / 2 && (k % 3 == 0) == (part == 0)) load_round(L, SL{}, k);
          if (part == 1) {
#pragma unroll
            for (int i = 0; i < NGL; ++i)
              if (i == jj / 2) load_affine(L, SL{}, i);
          }
        }
      });
      W44R_STAMP(2)
      if (!DEEP) activate_stage(SA{}, cc + 2);
      W44R_STAMP(3)
      if (!lateprod) barrier();
      W44R_STAMP(4)
      if (DEEP && kLoadsAtV) {
        // the requests of chunk cc + 3, in front of the V task: loads return IN ORDER per wave, so an activation load (HBM, microseconds)
        // issued among the MFMA jobs held back every later U load (L2, a few hundred cycles) behind it -- the MFMA segment then
        // waited for HBM although nothing in it needs the pixels.  Here the U ring is full (its six loads are older) and no
        // further U load is issued until the next MFMA segment, a V task later.
#pragma unroll
        for (int k = 0; k < NR; ++k) load_round(L, SL{}, k);
#pragma unroll
        for (int i = 0; i < NGL; ++i) load_affine(L, SL{}, i);
      }
      if (cc + pahead < NCHs) produce_task(cc + pahead);
      W44R_STAMP(5)
      if (lateprod) barrier();
      W44R_STAMP(6)
    };
    if constexpr (DEEP3) {
      for (int c = 0; c < NCHs; c += 2) {
        if (grot == 0) { interval(I0{}, I0{}, c); interval(I1{}, I1{}, c + 1); }
        else if (grot == 1) { interval(I0{}, I1{}, c); interval(I1{}, I2{}, c + 1); }
        else { interval(I0{}, I2{}, c); interval(I1{}, I0{}, c + 1); }
        grot = grot == 0 ? 2 : grot - 1;  // (+ 2 modulo 3)
      }
    } else {
      for (int c = 0; c < NCHs; c += 2) {
        interval(I0{}, I0{}, c);
        interval(I1{}, I0{}, c + 1);
      }
    }
    // (no vmcnt(0) here: every vector-memory operation of this kernel is a compiler-visible builtin, so the waitcnt pass orders
    // the epilogue's register reuse against whatever is still in flight -- the next item's chunk-2 pixels keep landing into their
    // own register set during the output transform)

    // ---- end of an item: Y = A^T M A through four exchange slabs [xi][cout block][lane] (the V ring: every stage of the item has
    // finished).  Pass q moves accumulator registers 4 q .. 4 q + 3 of all 36 positions; wave (cb, pg) then finishes register
    // 4 q + pg of cout block cb: cout = 32 cb + 8 q + 4 lhi + pg, tile = l31 (conv_wino44h.hip's output transform, verbatim).
    W44R_FSTAMP(8)
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
    float kOutScale, unused_umax;
    sload2(uniform_ptr(reinterpret_cast<const float *>(a.w_wino44h + (size_t)kX * a.Cout * g.Cin * 2 * g.nkd_w) + 1), kOutScale, unused_umax);
    (void)unused_umax;
    if (!AFFINE) kOutScale *= kVScale / kVScaleRaw;  // the packed tail carries 1 / (2^3 2^su)
    // accumulator tile 3 t + i of wave pg holds position s = 3 pg + i of pair t: row (0,5 | 1,2 | 3,4)[s / 6], column s % 6
    const int rsel = pg >> 1, cofs = 3 * (pg & 1);
    const int xb0 = (rsel ? 5 : 0) * 6 + cofs, xb1 = (rsel ? 2 : 1) * 6 + cofs, xb2 = (rsel ? 4 : 3) * 6 + cofs;
    float addv[4];
    if (ONEIMG) {
      const int co0 = kt * kK + cb * 32 + pg;
      float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ts[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float *const tp = a.chan_add ? a.chan_add + (size_t)min(n_cur, g.NIMG - 1) * a.chan_add_stride + co0 : nullptr;
      if (a.bias && tp) sload8x2(uniform_ptr(a.bias + co0), uniform_ptr(tp), bs, ts);
      else if (a.bias) sload8(uniform_ptr(a.bias + co0), bs);
      else if (tp) sload8(uniform_ptr(tp), ts);
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
    auto load_res = [&](int q) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        res[k] = *reinterpret_cast<const v4f *>(a.residual + obase0 + (size_t)(8 * q) * cstr + (size_t)k * a.Wo);
    };
    if (RES) load_res(0);
    const bool emit_stats = !D3 && a.stats_out != nullptr && g.S == 1;
    auto pass = [&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      const size_t obase = obase0 + (size_t)(8 * q) * cstr;
      float st_p = 0.f, st_s1 = 0.f, st_s2 = 0.f;  // this lane's 4x4 tile about a pivot (its first value)
#ifndef W44R_XADDTID
#define W44R_XADDTID 1  // 0: the exchange stores as ds_write_b32 (half the rate)
#endif
      if (W44R_XADDTID) {
        // the slabs are [xi][cout block][lane]: lane-linear, so the 36 stores are ds_write_addtid_b32 (address = M0 + offset +
        // 4 lane; twice the rate of ds_write_b32).  M0 = cout block + the wave's position row (x / 3) + slab pair; offsets: slab of
        // the pair, column x % 3.
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
          for (int x3 = 0; x3 < 3; ++x3) {
            const int m0v = __builtin_amdgcn_readfirstlane(((x3 == 0 ? xb0 : x3 == 1 ? xb1 : xb2) * 128 + cb * 64 + hh * 2 * kXS) * 4);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" ::"s"(m0v));
#pragma unroll
            for (int xc = 0; xc < 3; ++xc) {
              const int x = 3 * x3 + xc;
#pragma unroll
              for (int r2 = 0; r2 < 2; ++r2) {
                const int rr = 2 * hh + r2;
                const float v = x == 8 ? acc8[4 * q + rr] : read_pinned(16 * (x & 7) + 4 * q + rr);
                if (r2 == 0) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(xc * 512) : "memory");
                else asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(kXS * 4 + xc * 512) : "memory");
              }
            }
          }
        }
      } else {
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
      auto half = [&](auto hc) __attribute__((always_inline)) {
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
    W44R_FSTAMP(9)
    pass(I0{});
    W44R_FSTAMP(10)
    pass(I1{});
    W44R_FSTAMP(11)
    pass(I2{});
    W44R_FSTAMP(12)
    pass(std::integral_constant<int, 3>{});
    W44R_FSTAMP(13)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The pixel-tile layout of the register-fed kernel (every shape but eight images per item).  A V task's lane = (tile of 16,
// channel pair j of 4) reads, per patch row, columns 0-3 and 4-7 as two ds_read_b128.  ds_read_b128 is served in four groups of 16
// lanes = 4 tiles x 4 pairs (tiles {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15} of the task's 16): conflict-free iff the 16
// addresses are distinct modulo 256 bytes, i.e. in 16-byte units  U(tile) + O(j)  distinct modulo 16.  With
//   U(tile) = tile (mod 16):  row length PW = 8 (mod 16) units for 8 tile columns (32x32), 4 (mod 16) for 4 (16x16), anything for 16,
//   O(j) = 0, 8, 4, 12:       channel-plane stride PCH = 16 (mod 32) floats, half-tile stride HS = 16 (mod 64) floats
// every group sees {0,3,5,6} + {0,8,4,12} = all 16 residues.  Rows are 16-byte aligned (PW % 4 == 0; pixel (r, c) at r PW + c + 1,
// so a tile's patch starts at column 4 tc).
void w44r_relayout(const ddpm_conv_desc &d, W44HGeom &g) {
  if (g.TI == 8 && !W44R_NEWLAY8) return;
  const int w = d.Wo + 2;
  int pw = (w + 3) & ~3;
  if (g.TWc == 8) while (pw % 16 != 8) pw += 4;
  if (g.TWc == 4) while (pw % 16 != 4) pw += 4;
  g.PW = pw;
  g.IS = g.prow * g.PW;
  // eight 8x8 images per item (tile of 16 = 4 image + 2 tile row + tile column): the tile rows of an image fall into different
  // groups, so U only has to tell apart (image parity, tile column): image stride IS = 8 (mod 16) floats (10 rows of 12: as it is)
  if (g.TI == 8) while (g.IS % 16 != 8) g.IS += 4;
  g.PCH = g.TI * g.IS;
  while (g.PCH % 32 != 16) g.PCH += 1;
  g.HS = 4 * g.PCH + 272;  // = 16 (mod 64); the 256 dump floats of out-of-image lanes (four per lane) sit at 4 PCH + 1 ..
}

int launch_conv_wino44r(const ddpm_conv_desc &dk, const W44HGeom &g, size_t lds, hipStream_t s) {
  typedef void (*kern_t)(const ddpm_conv_desc, const W44HGeom);
  // shapes: one image per item with 9 (32x32) or 10 (64x64) staging units, two 16x16 images, eight 8x8 images
#define W44R_K(A, N, U) {conv_wino44r_kernel<A, N, U, false>, conv_wino44r_kernel<A, N, U, true>}
  static const kern_t kerns[2][4][2] = {
      {W44R_K(false, 9, 0), W44R_K(false, 10, 0), W44R_K(false, 8, 4), W44R_K(false, 8, 1)},
      {W44R_K(true, 9, 0), W44R_K(true, 10, 0), W44R_K(true, 8, 4), W44R_K(true, 8, 1)}};
#undef W44R_K
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 16; ++i)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[i / 8][i / 2 % 4][i % 2]),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int shape = g.TI == 1 ? (g.NRT == 9 ? 0 : 1) : g.UI == 4 ? 2 : 3;
  kern_t kern = kerns[dk.gscale ? 1 : 0][shape][dk.residual ? 1 : 0];
  if (g.up) {  // no prologue, no residual (w44h_geom)
    static const kern_t kerns_up[4] = {conv_wino44r_kernel<false, 9, 0, false, false, true>, conv_wino44r_kernel<false, 10, 0, false, false, true>,
                                       conv_wino44r_kernel<false, 8, 4, false, false, true>, conv_wino44r_kernel<false, 8, 1, false, false, true>};
    static bool attr_up_done = false;
    if (!attr_up_done) {
      for (int i = 0; i < 4; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns_up[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_up_done = true;
    }
    kern = kerns_up[shape];
  }
  if (dk.dims == 3) {  // only reached without prologue and with whole slices per item (w44h_geom)
    static const kern_t kerns3d[2][2] = {{conv_wino44r_kernel<false, 9, 0, false, true>, conv_wino44r_kernel<false, 9, 0, true, true>},
                                         {conv_wino44r_kernel<false, 10, 0, false, true>, conv_wino44r_kernel<false, 10, 0, true, true>}};
    static bool attr3_done = false;
    if (!attr3_done) {
      for (int i = 0; i < 4; ++i)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kerns3d[i / 2][i % 2]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
      attr3_done = true;
    }
    kern = kerns3d[g.NRT == 9 ? 0 : 1][dk.residual ? 1 : 0];
  }
  hipLaunchKernelGGL(kern, dim3(g.grid), dim3(512), lds, s, dk, g);
  DDPM_CHECK_LAUNCH();
  return 0;
}

}  // namespace ddpm
