// wino44h_common.h -- shared by the split-f16 Winograd F(4x4, 3x3) kernel (conv_wino44r.hip) and its host side (conv_wino44h.hip:
// geometry, dispatch, weight packing): item constants, the pinned-accumulator MFMA helpers, the 1-D transforms, the pair split,
// the item geometry and the packed-weight layout.  See conv_wino44h.hip's header comment for the arithmetic.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace ddpm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int kT = 32;                    // tiles per item
constexpr int kK = 64;                    // output channels per item
constexpr int kC = 8;                     // input channels per chunk
constexpr int kX = 36;                    // transform positions
constexpr int kPP = 12;                   // positions per phase
constexpr int kUSB = kPP * 2 * kK * 16;   // bytes of one U slot: [pos 12][plane 2][cout 64][8 ch f16]  (24576)
constexpr int kVSB = kPP * 2 * kT * 16;   // bytes of one V slot: [pos 12][plane 2][tile 32][8 ch f16]  (12288)
constexpr int kVB0 = 2 * kUSB;            // byte offset of the V ring
constexpr int kRINGF = (2 * kUSB + 2 * kVSB) / 4;  // floats of both rings (18432)
constexpr int kXS = kX * 2 * 64;          // exchange slab of the epilogue: [xi][cout block][lane] (4608 floats)
static_assert(kRINGF == 4 * kXS, "the operand rings are the epilogue's four exchange slabs");
constexpr float kVScale = 8.f;            // 2^3 on V (through the activation) behind a GroupNorm + SiLU prologue
constexpr float kVScaleRaw = 1.f;         // 2^0 on V for un-normalised inputs (Upsample, VQ-VAE residual units, plain convs)
constexpr int kTail = 64;                 // f16 slots behind the packed planes: float [0] = max |U|, float [1] = 1 / (2^3 2^su)

// ---- accumulators.  A wave owns nine 32x32 fp32 tiles.  Eight of them live in a[0:127], addressed BY NAME inside the asm
// statements: they are not C++ objects, so the compiler can neither spill nor copy them.  (It must not: hipcc does not know
// that an asm MFMA writes its destination over the next passes -- as C++ variables with "+a" constraints, one tile of the
// 16x16 / 8x8 variants was spilled around its MFMAs and the store, placed right behind the asm statement, saved stale
// values: registers 0..3 of that tile wrong by 1e-3, differently on every run.  tools/check_acc_spills.py now fails the build
// on any compiler-generated AGPR or scratch access inside the MFMA loops.)  The ninth tile is a C++ variable in arch VGPRs
// (a 512-thread kernel gets 128 + 128 registers); its two MFMAs are ONE asm statement that ends after their last pass.
#define W44H_CLOB16(b) "a" #b
__device__ __forceinline__ void reserve_agprs() {  // the only place the compiler learns that a0..a127 are in use
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
               "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31",
               "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47",
               "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63",
               "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79",
               "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95",
               "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109",
               "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123",
               "a124", "a125", "a126", "a127");
}
#undef W44H_CLOB16
// (register numbers are pasted into the asm text: an "n" operand above 63 would be printed in hex)
#define W44H_TILES(X) X(0, 0, 15) X(1, 16, 31) X(2, 32, 47) X(3, 48, 63) X(4, 64, 79) X(5, 80, 95) X(6, 96, 111) X(7, 112, 127)
#define W44H_REGS(X) \
  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) \
  X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) \
  X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) \
  X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64) X(65) X(66) X(67) X(68) X(69) X(70) X(71) X(72) X(73) \
  X(74) X(75) X(76) X(77) X(78) X(79) X(80) X(81) X(82) X(83) X(84) X(85) X(86) X(87) X(88) X(89) X(90) X(91) \
  X(92) X(93) X(94) X(95) X(96) X(97) X(98) X(99) X(100) X(101) X(102) X(103) X(104) X(105) X(106) X(107) X(108) \
  X(109) X(110) X(111) X(112) X(113) X(114) X(115) X(116) X(117) X(118) X(119) X(120) X(121) X(122) X(123) X(124) \
  X(125) X(126) X(127)
// tile T (0..7) += A B, after at most N LDS operations remain outstanding
template <int N>
__device__ __forceinline__ void mfma_pin_wait(int T, const h8 &a, const h8 &b) {
  switch (T) {
#define X(t, lo, hi)                                                                                                       \
  case t:                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(%2)\n\tv_mfma_f32_32x32x16_f16 a[" #lo ":" #hi "], %0, %1, a[" #lo ":" #hi "]"          \
                 ::"v"(a), "v"(b), "n"(N));                                                                                \
    break;
    W44H_TILES(X)
#undef X
  }
}
__device__ __forceinline__ void mfma_pin(int T, const h8 &a, const h8 &b) {
  switch (T) {
#define X(t, lo, hi)                                                                                                       \
  case t:                                                                                                                  \
    asm volatile("v_mfma_f32_32x32x16_f16 a[" #lo ":" #hi "], %0, %1, a[" #lo ":" #hi "]" ::"v"(a), "v"(b));              \
    break;
    W44H_TILES(X)
#undef X
  }
}
__device__ __forceinline__ void zero_pinned_tiles() {
#define X(r) asm volatile("v_accvgpr_write_b32 a" #r ", 0");
  W44H_REGS(X)
#undef X
}
__device__ __forceinline__ float read_pinned(int r) {  // register r = 16 T + element
  float v = 0.f;
  switch (r) {
#define X(n)                                                \
  case n:                                                   \
    asm volatile("v_accvgpr_read_b32 %0, a" #n : "=v"(v)); \
    break;
    W44H_REGS(X)
#undef X
  }
  return v;
}
// the ninth tile: both MFMAs of its job and their completion in one statement (8 passes of 4 cycles each; the second
// issues when the first has finished).  Nothing the compiler places behind this statement can see the tile half-written.
__device__ __forceinline__ void mfma_v_pair_wait0(f32x16 &c, const h8 &a, const h8 &bh, const h8 &bl) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %3, %0\n\t"
               "s_nop 15\n\ts_nop 15\n\ts_nop 7"
               : "+v"(c) : "v"(a), "v"(bh), "v"(bl));
}
__device__ __forceinline__ h8 lds_b128(int byte_addr, int imm) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(imm));
  return v;
}
// two floats 16 bytes apart through the scalar cache (lgkmcnt, not vmcnt); the pointer must be wave-uniform
__device__ __forceinline__ void sload2(const float *p, float &x0, float &x1) {
  asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(x0), "=&s"(x1) : "s"(p) : "memory");
}

// eight floats at byte offsets 32 q + {0, 16} (q = 0..3) from each of two wave-uniform pointers, ONE wait: the epilogue's bias /
// temb addends.  (As four + four sload2 calls an item paid eight serialized scalar-cache round trips before its first pass.)
__device__ __forceinline__ void sload8(const float *p, float (&x)[8]) {
  asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x10\n\ts_load_dword %2, %8, 0x20\n\ts_load_dword %3, %8, 0x30\n\t"
               "s_load_dword %4, %8, 0x40\n\ts_load_dword %5, %8, 0x50\n\ts_load_dword %6, %8, 0x60\n\ts_load_dword %7, %8, 0x70\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(x[0]), "=&s"(x[1]), "=&s"(x[2]), "=&s"(x[3]), "=&s"(x[4]), "=&s"(x[5]), "=&s"(x[6]), "=&s"(x[7])
               : "s"(p) : "memory");
}
__device__ __forceinline__ void sload8x2(const float *p, const float *r, float (&x)[8], float (&y)[8]) {
  asm volatile("s_load_dword %0, %16, 0x0\n\ts_load_dword %1, %16, 0x10\n\ts_load_dword %2, %16, 0x20\n\ts_load_dword %3, %16, 0x30\n\t"
               "s_load_dword %4, %16, 0x40\n\ts_load_dword %5, %16, 0x50\n\ts_load_dword %6, %16, 0x60\n\ts_load_dword %7, %16, 0x70\n\t"
               "s_load_dword %8, %17, 0x0\n\ts_load_dword %9, %17, 0x10\n\ts_load_dword %10, %17, 0x20\n\ts_load_dword %11, %17, 0x30\n\t"
               "s_load_dword %12, %17, 0x40\n\ts_load_dword %13, %17, 0x50\n\ts_load_dword %14, %17, 0x60\n\ts_load_dword %15, %17, 0x70\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(x[0]), "=&s"(x[1]), "=&s"(x[2]), "=&s"(x[3]), "=&s"(x[4]), "=&s"(x[5]), "=&s"(x[6]), "=&s"(x[7]),
                 "=&s"(y[0]), "=&s"(y[1]), "=&s"(y[2]), "=&s"(y[3]), "=&s"(y[4]), "=&s"(y[5]), "=&s"(y[6]), "=&s"(y[7])
               : "s"(p), "s"(r) : "memory");
}

// 1-D input transform B^T w (as conv_wino44.hip)
__device__ __forceinline__ void bt6(const float (&w)[6], float (&t)[6]) {
  const float p = __builtin_fmaf(-4.f, w[2], w[4]), q = __builtin_fmaf(-4.f, w[1], w[3]);
  const float r = w[4] - w[2], s = w[3] - w[1];
  t[0] = __builtin_fmaf(4.f, w[0], __builtin_fmaf(-5.f, w[2], w[4]));
  t[1] = p + q;
  t[2] = p - q;
  t[3] = __builtin_fmaf(2.f, s, r);
  t[4] = __builtin_fmaf(-2.f, s, r);
  t[5] = __builtin_fmaf(4.f, w[1], __builtin_fmaf(-5.f, w[3], w[5]));
}
// 1-D output transform A^T m
__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
  const float s = m1 + m2, d = m1 - m2, u = m3 + m4, v = m3 - m4;
  y[0] = (m0 + s) + u;
  y[1] = __builtin_fmaf(2.f, v, d);
  y[2] = __builtin_fmaf(4.f, u, s);
  y[3] = __builtin_fmaf(8.f, v, d) + m5;
}

// x -> (f16(x), f16(x - f16(x))) for two channels, pair-packed: hi = {h(a), h(b)}, lo = {l(a), l(b)}.  Four instructions:
// the remainders come from v_fma_mix_f32 reading the packed f16 halves directly (x - h is exact in fp32).  (Left to hipcc
// the same source became nine: it re-derived each half with v_fma_mixlo / mixhi_f16 from the transform's last fma, converted
// back and subtracted.)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t &hi, uint32_t &lo) {
  float la, lb;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(la), "v"(lb));
}

}  // namespace

struct W44HGeom {
  int TWc, THr;     // tile columns / rows per image
  int TI, TR;       // images per item, tile rows per item (per image)
  int parts;        // items per image along the rows
  int Cin, NCH, HW; // NCH = chunks of 8 channels
  int prow;         // pixel-tile rows per image of an item: 4 TR + 2
  int PW, IS, PCH;  // pixel tile: row length, image stride, floats per channel plane (conv_wino44.hip's padding rules)
  int HS;           // floats per half-tile: 4 channel planes + 1 (the second half-chunk sits one bank further) + 64 dump floats
  int UI;           // 64-pixel staging units per image of an item
  int NRT;          // staging rounds per pixel wave and half-chunk: TI * UI
  int KT, NIT, IPW, NS, grid;
  int xmap;
  int xitem;        // 1: the pixel waves' staging stream runs on across item boundaries (DDPM_W44H_XITEM, A/B)
  int rev;          // conv_wino44r.hip: 1 = a workgroup walks its items LAST to first (serpentine across consecutive launches: the
                    // consumer starts on what its producer wrote last, i.e. on what the 256 MB Infinity Cache still holds)
  int up, HWin;     // 1: DDPM_CONV_UPSAMPLE2 -- the pixel waves read the nearest-x2 image from the stored low-res one (HWin pixels)
  int S;            // channel-stream splits per item (1: none)
  long long pstride;
  int NIMG;
  // 3-D (dims = 3, the VQ-VAE residual units; as conv_wino44.hip): an "image" is one (n, d) slice, the chunk stream of an item
  // walks (depth tap, channel chunk) -- 2-D F(4x4) per depth tap, the taps accumulated in the transform domain
  int D;            // slices per batch item (1: plain 2-D)
  int CS;           // channel stride of the tensors in floats: D * HW
  int NCHc;         // channel chunks per depth tap; NCH = nkd * NCHc
  int kd0, nkd;     // depth taps kd0 .. kd0 + nkd - 1 (a depth-1 volume only has its centre tap)
  int nkd_w;        // depth-tap slabs per cout tile in w_wino44h: 3 for a 3x3x3 weight, else 1
};


bool w44h_geom(const ddpm_conv_desc &d, W44HGeom &g, bool sizing = false);  // conv_wino44h.hip
inline size_t w44h_lds_bytes(const W44HGeom &g) { return ((size_t)kRINGF + 4 * (size_t)g.HS) * sizeof(float); }
// conv_wino44r.hip: the same fused convolution, item and packed weights on the register-fed form of the kernel
int launch_conv_wino44r(const ddpm_conv_desc &dk, const W44HGeom &g, size_t lds, hipStream_t s);
void w44r_relayout(const ddpm_conv_desc &d, W44HGeom &g);  // its pixel-tile layout (called by w44h_geom, which checks the LDS size)

}  // namespace ddpm
