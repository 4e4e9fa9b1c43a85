// orl_mlp.h - wave-level MLP tower primitives for gfx950, shared by the rollout and update kernels.
//
// Tile = 16 batch rows per wavefront.  All activations live in registers in the "T layout"
// (transposed: features down the MFMA M dimension, batch rows across the N dimension), i.e. the
// C/D fragment of v_mfma_f32_16x16x4_f32 computing  Z^T[64 x 16] = W[64 x K] * X^T[K x 16]:
//
//   lane l:  j = l & 15 (batch row of the tile),  q = l >> 4;
//   act[m][r]  (m = 0..3 M-tiles, r = 0..3)  holds feature f = 16*m + 4*q + r of batch row j.
//
// Because the reduction index of a GEMM can be enumerated in any order as long as A and B agree,
// the C fragment of one layer is fed DIRECTLY as the B operand of the next layer: k-step (m', r')
// consumes register act[m'][r'] (lane (j,q) contributes feature 16m'+4q+r') and the matching A
// operand is W[out][16m'+4q+r'], which for r' = 0..3 is one 16-byte LDS read.  No cross-lane
// movement between layers; LayerNorm statistics of a batch row are an in-lane sum of 16 values
// plus two xor-shuffles (lanes l, l^16, l^32, l^48 share j).
//
// fp32 MFMA is bit-for-bit an fmaf chain (MI355X guide section 3), i.e. plain fp32 GEMM numerics.
#pragma once
#include "orl_common.h"

namespace orl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 64;    // hidden width supported by the MFMA towers
constexpr int W2S = 68;    // padded LDS row stride (floats) of the 64x64 matrices
constexpr int TILE_B = 16; // batch rows per wave tile

#define ORL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// timing probe (ORL_BUILD_DEFS=-DORL_SPLIT_PROBE, never shipped: WRONG numerics): only the three largest products of every split
// GEMM - what a two-term split (e.g. fp16 pairs) would issue; measures what halving the MFMA count of the towers could buy
#ifdef ORL_SPLIT_PROBE
#define ORL_IF_FULL(x)
#else
#define ORL_IF_FULL(x) x
#endif

// ---- three-term bf16 splitting (tools/split_bf16_gemm.hip) ----------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (32 cycles per 2 048 flop) and blocks the SIMD's VALU while it
// does; v_mfma_f32_16x16x32_bf16 retires 16 384 flop in 16 cycles and overlaps VALU.  An fp32 number is EXACTLY
// hi + mid + lo with three bf16 terms when each term truncates the running remainder (8 + 8 + 8 significand bits), so
// a.b is the sum of 9 bf16 products; the 6 largest leave an error of about one fp32 rounding of the product (measured
// on N(0,1) operands over K = 64: rms 8.9e-7 against 1.17e-6 for the fp32 MFMA itself, profiles/r03_split_bf16_gemm.txt).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bf16 image row stride (elements).  mm64_T_split's A fragments are ds_read_b128 of row j at 16 q bytes: serviced in four
// 16-lane groups {0-3,12-15,20-27} .. over 64 banks (MI355X guide, LDS section), a row stride of 36 dwords (72 elements,
// rounds 3 - 4) puts two lanes of EVERY group on the same 4 banks - each fragment read takes 8 LDS cycles instead of 4, and the
// 48 reads of a tile were 81 % of the kernel's SQ_LDS_BANK_CONFLICT cycles (237 per tile, profiles/r04_pmc_tower.txt).  40
// dwords (80 elements) is conflict free for that pattern (tools/lds_bank_model.py enumerates the strides).
#ifndef ORL_WBS
#define ORL_WBS 80
#endif
constexpr int WBS = ORL_WBS;
constexpr int WB_IMG_FLOATS = 3 * HID * WBS / 2;   // three parts x 64 rows, in floats (7 680 at WBS = 80)

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
// one dword = {bf16 truncation of a (low half), of b (high half)}: v_perm_b32 picks the two upper halves
__device__ __forceinline__ unsigned pack_hi16(float a, float b) { return __builtin_amdgcn_perm(f2u(b), f2u(a), 0x07060302u); }

// Remainder of a truncation (round 5 experiments; build-time switch ORL_SPLIT_FORM):
//   0  v_and_b32 + v_sub_f32 per value (rounds 3 - 4): 11 VALU per pair of values;
//   1  ONE v_dot2c_f32_bf16 per value: pk = {bf16 of a (low half), bf16 of b (high half)} is the dword pack_hi16 has just
//      produced as the MFMA operand, and D += A.lo * B.lo + A.hi * B.hi with B = (-1, -0) resp. (-0, -1) returns a - bf16(a)
//      resp. b - bf16(b) exactly (the remainder of a truncation is representable): 7 VALU per pair.  The constants carry -0 in
//      the unused half ON PURPOSE: hipcc encodes the bf16 pair (lo = -1, hi = +0) as the INLINE constant -1.0, which the
//      hardware reads as the fp32 pattern 0xbf800000 = (lo = +0, hi = -1) - the other half (tools/split_dot2c_probe.hip's
//      first run: every pair wrong).  A pattern with a sign bit in the other half is no inline constant and travels as a
//      32-bit literal.  Measured: NOT faster than form 0 (profiles/r05_experiments.md) - v_dot2c is not a full-rate op;
//   2  the two subtractions of a pair as ONE v_pk_add_f32 with negated operand: 9 VALU per pair.
#ifndef ORL_SPLIT_FORM
#define ORL_SPLIT_FORM 0
#endif
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rem_lo(float a, unsigned pk) {
#if ORL_SPLIT_FORM == 1
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, 0x8000bf80u), a, false);
#else
  return a - u2f(f2u(a) & 0xffff0000u);
#endif
}
__device__ __forceinline__ float rem_hi(float b, unsigned pk) {
#if ORL_SPLIT_FORM == 1
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, 0xbf808000u), b, false);
#else
  return b - u2f(f2u(b) & 0xffff0000u);
#endif
}
// both remainders of a pair
__device__ __forceinline__ void rem_pair(float a, float b, unsigned pk, float& ra, float& rb) {
#if ORL_SPLIT_FORM == 2
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const f32x2 ab = f32x2{a, b};
  const f32x2 t = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2_, ab) & u32x2_{0xffff0000u, 0xffff0000u});
  const f32x2 r = ab - t;
  ra = r[0];
  rb = r[1];
#else
  ra = rem_lo(a, pk);
  rb = rem_hi(b, pk);
#endif
}

// 8 fp32 values -> three bf16x8 MFMA fragments (hi, mid, lo); x = hi + mid + lo exactly.  11 / 7 / 9 VALU per 2 values.
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    hi[p] = pack_hi16(a, b);
    float ra, rb, sa, sb;
    rem_pair(a, b, hi[p], ra, rb);
    mid[p] = pack_hi16(ra, rb);
    rem_pair(ra, rb, mid[p], sa, sb);
    lo[p] = pack_hi16(sa, sb);
  }
}

__device__ __forceinline__ f32x4 mfma_bf16_16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16_32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// bf16 weight image of a 64 x 64 matrix A (rows = MFMA M index, 64 reduction indices k): part p (0 hi, 1 mid, 2 lo),
// row o; the 64 k's of a row are stored as [K-step h][lane group q][slot s] with k = 16 (2h + s/4) + 4q + s%4 - the
// order in which a T-layout activation tile (lane (j, q) holds k = 16m + 4q + r) fills the 8 slots of its B fragment,
// so the A fragment of lane (j, q) for (row block mo, step h) is ONE 16-byte read.
__device__ __forceinline__ int wb_off(int p, int o, int h, int q) { return (p * HID + o) * WBS + h * 32 + q * 8; }

// store element A[o][k] = w into the three images at `img`
__device__ __forceinline__ void split_weight_store(unsigned short* __restrict__ img, int o, int k, float w) {
  const int m = k >> 4, q = (k >> 2) & 3, r = k & 3, h = m >> 1, sl = (m & 1) * 4 + r;
  const float h1 = u2f(f2u(w) & 0xffff0000u), r1 = w - h1;
  const float m1 = u2f(f2u(r1) & 0xffff0000u), l1 = r1 - m1;
  img[wb_off(0, o, h, q) + sl] = (unsigned short)(f2u(h1) >> 16);
  img[wb_off(1, o, h, q) + sl] = (unsigned short)(f2u(m1) >> 16);
  img[wb_off(2, o, h, q) + sl] = (unsigned short)(f2u(l1) >> 16);
}

// two neighbouring reduction indices k (even), k + 1 of row o at once: their slots are adjacent, so each part is ONE dword
// store (pack_hi16) instead of two sub-dword ones - half the LDS store instructions of an image's staging
__device__ __forceinline__ void split_weight_store2(unsigned short* __restrict__ img, int o, int k, float w0, float w1) {
  const int m = k >> 4, q = (k >> 2) & 3, r = k & 3, h = m >> 1, sl = (m & 1) * 4 + r;  // r is 0 or 2
  const float r0 = w0 - u2f(f2u(w0) & 0xffff0000u), r1 = w1 - u2f(f2u(w1) & 0xffff0000u);
  const float s0 = r0 - u2f(f2u(r0) & 0xffff0000u), s1 = r1 - u2f(f2u(r1) & 0xffff0000u);
  *(unsigned*)(img + wb_off(0, o, h, q) + sl) = pack_hi16(w0, w1);
  *(unsigned*)(img + wb_off(1, o, h, q) + sl) = pack_hi16(r0, r1);
  *(unsigned*)(img + wb_off(2, o, h, q) + sl) = pack_hi16(s0, s1);
}

// the B fragments of a T-layout activation tile: xs[h][part]
__device__ __forceinline__ void split_T(const f32x4 (&in)[4], u32x4 (&xs)[2][3]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = in[2 * h][r]; x[4 + r] = in[2 * h + 1][r]; }
    split8(x, xs[h][0], xs[h][1], xs[h][2]);
  }
}

// ---- LDS reads the compiler cannot reschedule.  hipcc, close to the VGPR limit, sinks every ds_read of a software-
// pipelined loop right in front of its consumer and waits lgkmcnt(0) there: the GEMM loops then pay a full LDS round
// trip per 16-byte fragment (measured: 4 500 cycles per chunk against 770 of MFMA work).  These reads are volatile asm
// (issued where they are written), and the wait is an asm that "modifies" the registers it guards, so that the MFMAs
// consuming them cannot move above it.  LDS operations of a wave complete in order, so lgkmcnt(n) with n younger reads in
// flight guarantees the guarded ones; compiler-issued LDS / scalar loads in between only make a wait conservative.
__device__ __forceinline__ unsigned gt_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF>
__device__ __forceinline__ u32x4 gt_ds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void gt_lds_wait(u32x4& a, u32x4& b, u32x4& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int N>
__device__ __forceinline__ void gt_lds_wait(u32x4& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

// compile-time loop: f(integral_constant<int, I>) for I = B .. E-1 (asm immediates need constant expressions)
template <int I> struct GtIdx { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void gt_static_for(F&& f) {
  if constexpr (B < E) {
    f(GtIdx<B>{});
    gt_static_for<B + 1, E>(f);
  }
}

// acc += A in (both T layout), A = the bf16 images at Wb: 48 bf16 MFMAs (6 products x 4 row blocks x 2 K-steps),
// 24 ds_read_b128.  Products are issued smallest first.
#ifndef ORL_MM64_PF   // build-time experiment: A-fragment read-ahead (steps) of the pinned variant; 0 = the plain loop below
#define ORL_MM64_PF 0
#endif
#if ORL_MM64_PF > 0
// Experiment (ORL_BUILD_DEFS=-DORL_MM64_PF=1|2): the fragment reads pinned (volatile asm, issued PF steps = 3 PF reads ahead
// of their MFMAs, explicit lgkmcnt waits).  Left to itself hipcc batches the reads of two steps right in front of their
// twelve MFMAs and waits lgkmcnt(0) there, so in ONE wave every pair of steps opens with an exposed LDS round trip - but
// with two waves per SIMD the other wave fills it: 0.2278 - 0.2290 ms per pair launch with PF = 2 (256 VGPRs, 4 spilled),
// 0.2297 - 0.2319 with PF = 1, against 0.2266 - 0.2287 for the plain loop (interleaved A/B on one box).  Not the default.
__device__ __forceinline__ void mm64_T_split(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][3],
                                             f32x4 (&acc)[4], int j, int q) {
  constexpr int PF = ORL_MM64_PF, NS = 8;
  const unsigned base = gt_lds_addr(Wb) + (unsigned)(j * WBS + q * 8) * 2u;
  u32x4 w[PF + 1][3];
#define ORL_FRAG_OFF(p, s) ((((p) * HID + 16 * ((s) & 3)) * WBS + ((s) >> 2) * 32) * 2)
  gt_static_for<0, PF>([&](auto si) {
    constexpr int s = decltype(si)::value;
    w[s][0] = gt_ds_read128<ORL_FRAG_OFF(0, s)>(base);
    w[s][1] = gt_ds_read128<ORL_FRAG_OFF(1, s)>(base);
    w[s][2] = gt_ds_read128<ORL_FRAG_OFF(2, s)>(base);
  });
  gt_static_for<0, NS>([&](auto si) {
    constexpr int s = decltype(si)::value;
    if constexpr (s + PF < NS) {
      w[(s + PF) % (PF + 1)][0] = gt_ds_read128<ORL_FRAG_OFF(0, s + PF)>(base);
      w[(s + PF) % (PF + 1)][1] = gt_ds_read128<ORL_FRAG_OFF(1, s + PF)>(base);
      w[(s + PF) % (PF + 1)][2] = gt_ds_read128<ORL_FRAG_OFF(2, s + PF)>(base);
    }
    constexpr int ahead = (NS - 1 - s) < PF ? (NS - 1 - s) : PF;
    u32x4& wh = w[s % (PF + 1)][0];
    u32x4& wm = w[s % (PF + 1)][1];
    u32x4& wl = w[s % (PF + 1)][2];
    gt_lds_wait<3 * ahead>(wh, wm, wl);
    constexpr int h = s >> 2, mo = s & 3;
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wl, xs[h][0], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wh, xs[h][2], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wm, xs[h][1], acc[mo]);)
    acc[mo] = mfma_bf16_16(wm, xs[h][0], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][0], acc[mo]);
  });
#undef ORL_FRAG_OFF
}
#elif defined(ORL_MM64_SINGLE)
// build-time experiment (round 5, three waves per SIMD): A fragments NOT double-buffered - 12 registers less, the LDS round trip
// of every step is left to the other waves of the SIMD
__device__ __forceinline__ void mm64_T_split(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][3],
                                             f32x4 (&acc)[4], int j, int q) {
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    const u32x4 wh = *(const u32x4*)(Wb + wb_off(0, 16 * mo + j, h, q));
    const u32x4 wm = *(const u32x4*)(Wb + wb_off(1, 16 * mo + j, h, q));
    const u32x4 wl = *(const u32x4*)(Wb + wb_off(2, 16 * mo + j, h, q));
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wl, xs[h][0], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wh, xs[h][2], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wm, xs[h][1], acc[mo]);)
    acc[mo] = mfma_bf16_16(wm, xs[h][0], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][0], acc[mo]);
  }
}
#else
__device__ __forceinline__ void mm64_T_split(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][3],
                                             f32x4 (&acc)[4], int j, int q) {
  u32x4 w[2][3];  // A fragments double-buffered over the 8 (h, mo) steps
#pragma unroll
  for (int p = 0; p < 3; ++p) w[0][p] = *(const u32x4*)(Wb + wb_off(p, j, 0, q));
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    if (st < 7) {
      const int h2 = (st + 1) >> 2, mo2 = (st + 1) & 3;
#pragma unroll
      for (int p = 0; p < 3; ++p) w[(st + 1) & 1][p] = *(const u32x4*)(Wb + wb_off(p, 16 * mo2 + j, h2, q));
    }
    const u32x4 wh = w[st & 1][0], wm = w[st & 1][1], wl = w[st & 1][2];
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wl, xs[h][0], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wh, xs[h][2], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wm, xs[h][1], acc[mo]);)
    acc[mo] = mfma_bf16_16(wm, xs[h][0], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][0], acc[mo]);
  }
}
#endif

// acc += A^T in (both T layout) with A = the SAME bf16 images mm64_T_split reads (A = W2 -> acc += W2^T in: the dgrad)
// through gfx950's transposing LDS read (tools/ds_read_tr_probe.hip pins its semantics: within a 16-lane group lane s
// passes the address of an 8-byte chunk E_s and lane n receives E_{(n >> 2) + 4 jj}[n & 3], jj = 0..3).  Lane s of group
// q points at image row o = 32 h + 16 half + 4 q + (s >> 2), the 4 consecutive elements i = 16 mo + 4 (s & 3) .. + 3 of
// that row (wb_off stores them contiguously); lane n then holds W2[o = 32 h + 16 half + 4 q + jj][i = 16 mo + n], jj =
// 0..3 - slots 4 half .. 4 half + 3 of its A fragment for (row block mo, K-step h), whose slot s carries reduction
// index o = 16 (2 h + s / 4) + 4 q + s % 4 exactly as split_T lays out the B operand.  Two 8-byte reads per fragment
// instead of one 16-byte read, and no second (transposed) image in LDS: - 27.6 KB.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 ds_read_tr16(const unsigned short* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(u32x2, v);
}
__device__ __forceinline__ void mm64_T_split_tr(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][3],
                                                f32x4 (&acc)[4], int j, int q) {
  const unsigned short* base = Wb + (4 * q + (j >> 2)) * WBS + (j & 3) * 8;
  auto frag = [&](int p, int h, int mo) -> u32x4 {
    const unsigned short* a = base + (p * HID + 32 * h) * WBS + (mo >> 1) * 32 + (mo & 1) * 4;
    const u32x2 lo = ds_read_tr16(a), hi = ds_read_tr16(a + 16 * WBS);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
#ifdef ORL_MM64_SINGLE
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    const u32x4 wh = frag(0, h, mo), wm = frag(1, h, mo), wl = frag(2, h, mo);
#else
  u32x4 w[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) w[0][p] = frag(p, 0, 0);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    if (st < 7) {
#pragma unroll
      for (int p = 0; p < 3; ++p) w[(st + 1) & 1][p] = frag(p, (st + 1) >> 2, (st + 1) & 3);
    }
    const u32x4 wh = w[st & 1][0], wm = w[st & 1][1], wl = w[st & 1][2];
#endif
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wl, xs[h][0], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wh, xs[h][2], acc[mo]);)
    ORL_IF_FULL(acc[mo] = mfma_bf16_16(wm, xs[h][1], acc[mo]);)
    acc[mo] = mfma_bf16_16(wm, xs[h][0], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_bf16_16(wh, xs[h][0], acc[mo]);
  }
}

// ---- two-term fp16 splitting (round 6; tools/split_f16_gemm.hip) --------------------------------------------------------------
// A timing probe that issued 3 of the 6 bf16 products ran the tower pair 21 % faster (profiles/r06_experiments.md section 8):
// the kernel is bound by its MFMA + splitting work.  x = hi + lo with hi = rn16(x), lo = rn16(x - hi) carries 11 + 11
// significand bits (|x - hi - lo| <= 2^-22 |x|, the remainder x - hi is exact), and a.b ~ hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_f16 (the dropped lo.lo <= 2^-22 |a b|): THREE products instead of six, 4 VALU per pair of values
// instead of 11.  Measured on one wave against float64 (N(0,1) operands, K = 64): rms 8.0e-8 of the output's rms, against
// 8.3e-8 for the six bf16 products and 1.07e-7 for v_mfma_f32_16x16x4_f32 itself.  fp16's RANGE is the catch (the bf16
// split has fp32's): an operand has to sit in [2^-14, 2^16) to keep its 22 bits, and gradients of 1e-6 lose everything
// unscaled (rms 1e-2).  So every operand is scaled by an exact power of two chosen where it is produced -
//   * a weight image by 2^kw, kw from the image's own maximum (stage_tower: max -> [2^13, 2^14)),
//   * LayerNorm outputs (|xhat| < 8) as they are,
//   * a tile of gradients by its own maximum (orl_ppo_tower.h: max -> [2^11, 2^12)),
// and the accumulators are unscaled by the inverse power (exact).  fp16 subnormals pass through the MFMA unflushed
// (probe), so a value far below its tile's maximum degrades gracefully: its absolute error is 2^-25 x 2^-scale, i.e.
// <= 2^-36 of the tile's maximum.  ORL_TOWER_F16 = 0 keeps the three-term bf16 kernels of rounds 3 - 5 (A/B switch).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int WBH_IMG_FLOATS = 2 * HID * WBS / 2;  // two parts x 64 rows (5 120 floats at WBS = 80)
constexpr int TOWER_SPLIT_IMG_FLOATS = ORL_TOWER_F16 ? WBH_IMG_FLOATS : WB_IMG_FLOATS;
constexpr int TOWER_SPLIT_PARTS = ORL_TOWER_F16 ? 2 : 3;
// one dword = {rn16(a) (low half), rn16(b)}: v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
// x - (float)half: ONE v_fma_mix_f32 (the fp16 half widened inside the FMA; the result is exact).  hipcc does not form it
// from `x - (float)h` (it emits v_cvt_f32_f16 + v_sub_f32), hence the asm.
__device__ __forceinline__ float rem16_lo(unsigned pk, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
__device__ __forceinline__ float rem16_hi(unsigned pk, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
// 8 fp32 values -> two f16x8 MFMA fragments (hi, lo): 4 VALU per 2 values
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    hi[p] = cvt_pk_f16(a, b);
    lo[p] = cvt_pk_f16(rem16_lo(hi[p], a), rem16_hi(hi[p], b));
  }
}
__device__ __forceinline__ f32x4 mfma_f16_16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16_32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the two fp16 images of a (pre-scaled) 64 x 64 matrix: wb_off's layout with parts 0 (hi), 1 (lo)
__device__ __forceinline__ void split_weight_store2h(unsigned short* __restrict__ img, int o, int k, float w0, float w1) {
  const int m = k >> 4, q = (k >> 2) & 3, r = k & 3, h = m >> 1, sl = (m & 1) * 4 + r;  // r is 0 or 2
  const unsigned hi = cvt_pk_f16(w0, w1);
  *(unsigned*)(img + wb_off(0, o, h, q) + sl) = hi;
  *(unsigned*)(img + wb_off(1, o, h, q) + sl) = cvt_pk_f16(rem16_lo(hi, w0), rem16_hi(hi, w1));
}
// the B fragments of a T-layout activation tile: xs[h][part]
__device__ __forceinline__ void split_Th(const f32x4 (&in)[4], u32x4 (&xs)[2][2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = in[2 * h][r]; x[4 + r] = in[2 * h + 1][r]; }
    split8h(x, xs[h][0], xs[h][1]);
  }
}
// acc += A in (both T layout), A = the fp16 images at Wb: 24 MFMAs (3 products x 4 row blocks x 2 K-steps), 16 ds_read_b128
__device__ __forceinline__ void mm64_T_h2(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][2], f32x4 (&acc)[4], int j,
                                          int q) {
  u32x4 w[2][2];  // A fragments double-buffered over the 8 (h, mo) steps
#pragma unroll
  for (int p = 0; p < 2; ++p) w[0][p] = *(const u32x4*)(Wb + wb_off(p, j, 0, q));
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    if (st < 7) {
      const int h2 = (st + 1) >> 2, mo2 = (st + 1) & 3;
#pragma unroll
      for (int p = 0; p < 2; ++p) w[(st + 1) & 1][p] = *(const u32x4*)(Wb + wb_off(p, 16 * mo2 + j, h2, q));
    }
    const u32x4 wh = w[st & 1][0], wl = w[st & 1][1];
    acc[mo] = mfma_f16_16(wl, xs[h][0], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][0], acc[mo]);
  }
}
// acc += A^T in through the transposing LDS read of the SAME images (mm64_T_split_tr's addressing)
__device__ __forceinline__ void mm64_T_h2_tr(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][2], f32x4 (&acc)[4],
                                             int j, int q) {
  const unsigned short* base = Wb + (4 * q + (j >> 2)) * WBS + (j & 3) * 8;
  auto frag = [&](int p, int h, int mo) -> u32x4 {
    const unsigned short* a = base + (p * HID + 32 * h) * WBS + (mo >> 1) * 32 + (mo & 1) * 4;
    const u32x2 lo = ds_read_tr16(a), hi = ds_read_tr16(a + 16 * WBS);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  u32x4 w[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) w[0][p] = frag(p, 0, 0);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    if (st < 7) {
#pragma unroll
      for (int p = 0; p < 2; ++p) w[(st + 1) & 1][p] = frag(p, (st + 1) >> 2, (st + 1) & 3);
    }
    const u32x4 wh = w[st & 1][0], wl = w[st & 1][1];
    acc[mo] = mfma_f16_16(wl, xs[h][0], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][0], acc[mo]);
  }
}
// maximum of |x| over the wave (every lane returns it): 4 DPP rotations inside the 16-lane rows + the two half-wave swaps
__device__ __forceinline__ float wave_absmax(float v) {
  v = fmaxf(v, u2f(__builtin_amdgcn_update_dpp(0u, f2u(v), 0x121, 0xf, 0xf, false)));  // row_ror:1
  v = fmaxf(v, u2f(__builtin_amdgcn_update_dpp(0u, f2u(v), 0x122, 0xf, 0xf, false)));  // row_ror:2
  v = fmaxf(v, u2f(__builtin_amdgcn_update_dpp(0u, f2u(v), 0x124, 0xf, 0xf, false)));  // row_ror:4
  v = fmaxf(v, u2f(__builtin_amdgcn_update_dpp(0u, f2u(v), 0x128, 0xf, 0xf, false)));  // row_ror:8
  return row_allmax(v);
}
// biased exponent E of a non-negative maximum, clamped so that 2^(138 - E) (which moves the maximum into [2^11, 2^12)) and its
// inverse stay finite: zero / tiny tiles get the largest scale (their values underflow harmlessly), inf / nan a finite one
__device__ __forceinline__ int scale_exponent(float mx) {
  int e = (int)(f2u(mx) >> 23) & 0xff;
  e = e < 20 ? 20 : e;
  return e > 254 ? 254 : e;
}

// LDS image of one tower (offsets in floats, all multiples of 4 => 16-byte aligned).
struct TowerLds {
  int DP, n_out, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, logstd, W2T, W3P, wsc, total;
  __host__ __device__ TowerLds() {}
  // with_w3p: a second image of W3, zero padded to 16 rows at the conflict-free stride W2S - the MFMA operand of the
  // head GEMMs of wide heads (orl_ppo_tower.h, NO > 4)
  // split: W2 (and W2T) are the three-term bf16 images of split_weight_store (mm64_T_split's A operands) instead of
  // fp32 rows - the update towers' SPLIT builds (orl_ppo_tower.h)
  __host__ __device__ TowerLds(int D, int n_out_, bool gaussian, bool with_w2t, bool with_w3p = false,
                               bool split = false) {
    const int w2_floats = split ? TOWER_SPLIT_IMG_FLOATS : HID * W2S;
    DP = (D + 3) & ~3;
    n_out = n_out_;
    const int no4 = (n_out + 3) & ~3;
    int o = 0;
    W1 = o; o += HID * DP;
    b1 = o; o += HID;
    g1 = o; o += HID;
    be1 = o; o += HID;
    W2 = o; o += w2_floats;
    b2 = o; o += HID;
    g2 = o; o += HID;
    be2 = o; o += HID;
    W3 = o; o += with_w3p ? 0 : no4 * HID;  // wide heads read the padded image W3P only
    b3 = o; o += no4;
    logstd = o; o += gaussian ? no4 : 0;
    W2T = o; o += with_w2t ? w2_floats : 0;
    W3P = o; o += with_w3p ? 16 * W2S : 0;
    // fp16 split images (ORL_TOWER_F16): {kw as a float, 2^kw, 2^-kw, 1e-5 x 4^kw} of the image's scale, then 16 words of
    // scratch for the staging's maximum
    wsc = o; o += (split && ORL_TOWER_F16) ? 20 : 0;
    total = o;
  }
};

// Cooperative global -> LDS staging of a tower (any thread count).  W1 is zero padded to DP
// columns, W2 re-strided to W2S, optionally also stored transposed (for the backward GEMM).
__device__ inline void stage_tower(float* __restrict__ lds, const float* __restrict__ theta, const TowerLayout& tl,
                                   const TowerLds& tw, bool with_w2t, int tid, int nthreads, bool with_w3p = false,
                                   bool split = false, bool fold = false) {
  // fold (the update towers): the LayerNorm affines are folded into the next Linear - exact algebra,
  //   W2 (xhat1 * g1 + be1) + b2 = (W2 diag(g1)) xhat1 + (b2 + W2 be1),   W3 (xhat2 * g2 + be2) + b3 likewise,
  // so a tile's forward feeds xhat straight into the GEMMs and its backward gets d xhat = (W diag(g))^T d out from the
  // same images: no per-tile reads of g / be, no affine or d*g passes.  The images of W2, W2^T, W3 hold W diag(g), the
  // b2 / b3 slots hold the folded biases; the raw gradient sums (G = dz2^T xhat1, ...) are unchanged.
  // Staging runs in ROUNDS: in round r every array contributes its element tid + r * nthreads, all of the round's global loads
  // are issued before the first LDS store (phase 1 / phase 2 below).  Array by array - one rolled loop each - every loop
  // iteration was a global round trip of its own: 12 in a row for a wide-observation tower on 512 threads, 12 000 cycles of
  // prologue per launch (tools/tower_phase_prof.py).  Same expressions per element as before: bit-identical images.
  const int D = tl.D;
  const int no4 = (tl.n_out + 3) & ~3;
  const int n_w3p = with_w3p ? 16 * W2S : 0;
  const int n_w1 = HID * tw.DP;
  const int n_w2 = split ? (HID / 2) * (HID / 2) : HID * HID;  // 2 x 2 blocks (bf16 images) or elements
  const int n_w3 = with_w3p ? 0 : no4 * HID;
  const int rows = HID + tl.n_out;  // folded biases: b2' rows, then b3' rows; 8 threads (aligned lane groups) per row
  const int n_fb = fold ? rows * 8 : 0;
  int n_max = n_w3p > n_w1 ? n_w3p : n_w1;
  n_max = n_w2 > n_max ? n_w2 : n_max;
  n_max = n_w3 > n_max ? n_w3 : n_max;
  n_max = n_fb > n_max ? n_fb : n_max;
  n_max = HID > n_max ? HID : n_max;
  unsigned short* b2i = (unsigned short*)(lds + tw.W2);
  unsigned short* b2t = (unsigned short*)(lds + tw.W2T);
  // fp16 images: the scale 2^kw that moves the image's largest |W2 diag(g1)| into [2^13, 2^14) - one pass over W2 (L2 hits
  // again in the rounds below), a wave maximum per wave, one barrier.  The folded bias b2' is stored scaled as well: fc2's
  // accumulators start from it and LayerNorm 2 takes them scaled (with eps scaled by 4^kw it returns the same xhat2).
  float wscale = 1.f;
  struct Round {  // one round's loaded values of one thread
    float p_w, p_g, w1v, vb1, vg1, vbe1, vb2, vg2, vbe2, w00, w01, w10, w11, g0, g1v, w3v, w3g, vb3, vls, fw[8], fbe[8], fb;
  };
  // ---- phase 1: loads -----------------------------------------------------------------------------------------------------
  auto load_round = [&](int e, Round& R) {
    R.p_w = 0.f; R.p_g = 1.f;                      // W3P
    const int pc = e / W2S, pi = e - pc * W2S;
    if (e < n_w3p && pc < tl.n_out && pi < HID) {
      R.p_w = theta[tl.oW3 + pc * HID + pi];
      if (fold) R.p_g = theta[tl.og2 + pi];
    }
    R.w1v = 0.f;                                   // W1
    const int w1f = e / tw.DP, w1k = e - w1f * tw.DP;
    if (e < n_w1 && w1k < D) R.w1v = theta[tl.oW1 + w1f * D + w1k];
    R.vb1 = R.vg1 = R.vbe1 = R.vb2 = R.vg2 = R.vbe2 = 0.f;  // the 64-wide vectors
    if (e < HID) {
      R.vb1 = theta[tl.ob1 + e]; R.vg1 = theta[tl.og1 + e]; R.vbe1 = theta[tl.obe1 + e];
      if (!fold) R.vb2 = theta[tl.ob2 + e];  // (fold: ONE writer per slot - the folded biases below)
      R.vg2 = theta[tl.og2 + e]; R.vbe2 = theta[tl.obe2 + e];
    }
    R.w00 = R.w01 = R.w10 = R.w11 = 0.f; R.g0 = R.g1v = 1.f;  // W2
    const int bo = split ? 2 * (e >> 5) : (e >> 6), bi = split ? 2 * (e & 31) : (e & 63);
    if (e < n_w2) {
      if (fold) R.g0 = theta[tl.og1 + bi];
      R.w00 = theta[tl.oW2 + bo * HID + bi];
      if (split) {
        if (fold) R.g1v = theta[tl.og1 + bi + 1];
        R.w01 = theta[tl.oW2 + bo * HID + bi + 1];
        R.w10 = theta[tl.oW2 + (bo + 1) * HID + bi];
        R.w11 = theta[tl.oW2 + (bo + 1) * HID + bi + 1];
      }
    }
    R.w3v = 0.f; R.w3g = 1.f;                      // W3 (narrow heads)
    if (e < n_w3 && e < tl.n_out * HID) {
      R.w3v = theta[tl.oW3 + e];
      if (fold) R.w3g = theta[tl.og2 + (e & 63)];
    }
    R.vb3 = R.vls = 0.f;                           // b3 / logstd
    if (e < no4 && e < tl.n_out) {
      if (!fold) R.vb3 = theta[tl.ob3 + e];
      if (tl.head == ORL_HEAD_GAUSSIAN) R.vls = theta[tl.ologstd + e];
    }
    const int fo = e >> 3, part = e & 7;           // folded biases: 8 terms of row fo, part `part`
    R.fb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) R.fw[k] = R.fbe[k] = 0.f;
    if (fold && fo < rows) {
      const float* wrow = fo < HID ? theta + tl.oW2 + fo * HID : theta + tl.oW3 + (fo - HID) * HID;
      const float* be = theta + (fo < HID ? tl.obe1 : tl.obe2);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        R.fw[k] = wrow[8 * part + k];
        R.fbe[k] = be[8 * part + k];
      }
      if (part == 0) R.fb = fo < HID ? theta[tl.ob2 + fo] : theta[tl.ob3 + (fo - HID)];
    }
  };
  // ---- phase 2: stores ----------------------------------------------------------------------------------------------------
  auto store_round = [&](int e, const Round& R) {
    const int pc = e / W2S, pi = e - pc * W2S;
    if (e < n_w3p) lds[tw.W3P + e] = (pc < tl.n_out && pi < HID) ? R.p_w * R.p_g : 0.f;
    if (e < n_w1) lds[tw.W1 + e] = R.w1v;
    if (e < HID) {
      lds[tw.b1 + e] = R.vb1; lds[tw.g1 + e] = R.vg1; lds[tw.be1 + e] = R.vbe1;
      if (!fold) lds[tw.b2 + e] = R.vb2 * wscale;
      lds[tw.g2 + e] = R.vg2; lds[tw.be2 + e] = R.vbe2;
    }
    if (e < n_w2) {
      const int bo = split ? 2 * (e >> 5) : (e >> 6), bi = split ? 2 * (e & 31) : (e & 63);
      if (split) {
        // bf16 images: fc2's A = W2 (rows o, reduction over i), dgrad's A = W2^T (rows i, reduction over o).  One 2 x 2
        // block of W2 per thread and round: its two row pairs are dword stores of W2's image, its two column pairs dword
        // stores of the transposed image (split_weight_store2) - no sub-dword LDS store in the staging
        const float a00 = R.w00 * R.g0, a01 = R.w01 * R.g1v, a10 = R.w10 * R.g0, a11 = R.w11 * R.g1v;
#if ORL_TOWER_F16
        split_weight_store2h(b2i, bo, bi, a00 * wscale, a01 * wscale);
        split_weight_store2h(b2i, bo + 1, bi, a10 * wscale, a11 * wscale);
        if (with_w2t) {
          split_weight_store2h(b2t, bi, bo, a00 * wscale, a10 * wscale);
          split_weight_store2h(b2t, bi + 1, bo, a01 * wscale, a11 * wscale);
        }
#else
        split_weight_store2(b2i, bo, bi, a00, a01);
        split_weight_store2(b2i, bo + 1, bi, a10, a11);
        if (with_w2t) {
          split_weight_store2(b2t, bi, bo, a00, a10);
          split_weight_store2(b2t, bi + 1, bo, a01, a11);
        }
#endif
      } else {
        const float w = R.w00 * R.g0;
        lds[tw.W2 + bo * W2S + bi] = w;
        if (with_w2t) lds[tw.W2T + bi * W2S + bo] = w;
      }
    }
    if (e < n_w3) lds[tw.W3 + e] = e < tl.n_out * HID ? R.w3v * R.w3g : 0.f;
    if (e < no4) {
      if (!fold || e >= tl.n_out) lds[tw.b3 + e] = R.vb3;
      if (tl.head == ORL_HEAD_GAUSSIAN) lds[tw.logstd + e] = R.vls;
    }
    if (fold) {
      const int fo = e >> 3, part = e & 7;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += R.fw[k] * R.fbe[k];
      // sum over the 8 aligned lanes of a row on the VALU (DPP quad permutes + half-row mirror: the same pairs as the
      // ds_bpermute butterfly acc += xor 1, 2, 4 - three LDS round trips per round of the prologue)
      acc += u2f(__builtin_amdgcn_update_dpp(0u, f2u(acc), 0xB1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
      acc += u2f(__builtin_amdgcn_update_dpp(0u, f2u(acc), 0x4E, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
      acc += u2f(__builtin_amdgcn_update_dpp(0u, f2u(acc), 0x141, 0xf, 0xf, false));  // row_half_mirror
      if (fo < rows && part == 0) {
        if (fo < HID) lds[tw.b2 + fo] = (R.fb + acc) * wscale;
        else lds[tw.b3 + (fo - HID)] = R.fb + acc;
      }
    }
  };
  // one round per trip (two rounds in flight per trip were measured no faster at configuration 2 and slower at the wide
  // towers: 9 500 -> 10 900 cycles of prologue); every lane of a wave runs the same trips - the shuffles of the folded
  // biases are wave-wide
  // (fp16 images: round 0's loads are in flight while the image's maximum is formed - one global round trip for both)
  {
    Round R;
    load_round(tid, R);
    if (split && ORL_TOWER_F16) {
      float mx = 0.f;
      for (int e = tid; e < HID * HID; e += nthreads) {
        const float w = theta[tl.oW2 + e] * (fold ? theta[tl.og1 + (e & 63)] : 1.f);
        mx = fmaxf(mx, fabsf(w));
      }
      mx = wave_absmax(mx);
      if ((tid & 63) == 0) lds[tw.wsc + 4 + (tid >> 6)] = mx;
      __syncthreads();
      mx = 0.f;
      for (int w = 0; w < (nthreads + 63) / 64; ++w) mx = fmaxf(mx, lds[tw.wsc + 4 + w]);
      int kw = 0;
      const int eb = (int)(f2u(mx) >> 23) & 0xff;  // mx in [2^(eb - 127), 2^(eb - 126))
      if (eb > 0 && eb < 255) kw = 13 - (eb - 127);
      kw = kw < -40 ? -40 : (kw > 40 ? 40 : kw);
      wscale = __builtin_ldexpf(1.f, kw);
      if (tid == 0) {
        lds[tw.wsc + 0] = (float)kw;
        lds[tw.wsc + 1] = wscale;
        lds[tw.wsc + 2] = __builtin_ldexpf(1.f, -kw);
        lds[tw.wsc + 3] = __builtin_ldexpf(1e-5f, 2 * kw);
      }
    }
    store_round(tid, R);
  }
  for (int e0 = nthreads; e0 < n_max; e0 += nthreads) {
    Round R;
    load_round(e0 + tid, R);
    store_round(e0 + tid, R);
  }
}

// acc[m] <- vec[16m+4q .. +3]   (bias init of an accumulator in T layout)
__device__ inline void load_vec_T(const float* __restrict__ v, int q, f32x4 (&acc)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = *(const f32x4*)(v + 16 * m + 4 * q);
}

// fc1: acc += W1[64 x DP] * X^T ; xb(s) returns this lane's B operand x[j][4s+q] (0 beyond D).
template <class XB>
__device__ inline void fc1_T(const float* __restrict__ W1s, int DP, XB xb, f32x4 (&acc)[4], int j, int q) {
  for (int s = 0; s < (DP >> 2); ++s) {
    const float b = xb(s);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float a = W1s[(16 * m + j) * DP + 4 * s + q];
      acc[m] = ORL_MFMA(a, b, acc[m]);
    }
  }
}

// acc += Ws[64 x 64 (stride W2S)] * in   (both T layout).  64 MFMAs, 16 ds_read_b128.
__device__ inline void mm64_T(const float* __restrict__ Ws, const f32x4 (&in)[4], f32x4 (&acc)[4], int j, int q) {
  // A operands double-buffered: the four 16-byte reads of k-block mi+1 are issued BEFORE the 16 MFMAs of k-block mi,
  // otherwise every block opens with an exposed LDS round trip (measured: a 64-MFMA phase took ~2 500 cycles of its
  // 2 048-cycle floor on a wave running alone)
  f32x4 a4[2][4];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) a4[0][mo] = *(const f32x4*)(Ws + (16 * mo + j) * W2S + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    if (mi < 3) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
        a4[(mi + 1) & 1][mo] = *(const f32x4*)(Ws + (16 * mo + j) * W2S + 16 * (mi + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) acc[mo] = ORL_MFMA(a4[mi & 1][mo][r], in[mi][r], acc[mo]);
    }
  }
}

// In-lane part of a 16-value reduction: the four f32x4 registers are added element-wise first (v_pk_add_f32 / v_pk_fma_f32: two
// values per instruction), the horizontal sum of the one remaining vector comes last - 9 instructions per sum (11 for a sum of
// products) instead of 15 / 24 scalar ones (round 5; ORL_PACKED_SUMS = 0 restores the scalar order for A/B).
#ifndef ORL_PACKED_SUMS
#define ORL_PACKED_SUMS 1
#endif
__device__ __forceinline__ float hsum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ float lane_sum16(const f32x4 (&x)[4]) {
#if ORL_PACKED_SUMS
  return hsum4((x[0] + x[1]) + (x[2] + x[3]));
#else
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) s += (x[m][0] + x[m][1]) + (x[m][2] + x[m][3]);
  return s;
#endif
}
__device__ __forceinline__ float lane_dot16(const f32x4 (&x)[4], const f32x4 (&y)[4]) {
#if ORL_PACKED_SUMS
  f32x4 v = x[0] * y[0];
  v = x[1] * y[1] + v;
  v = x[2] * y[2] + v;
  v = x[3] * y[3] + v;
  return hsum4(v);
#else
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) s += (x[m][0] * y[m][0] + x[m][1] * y[m][1]) + (x[m][2] * y[m][2] + x[m][3] * y[m][3]);
  return s;
#endif
}

// sum over the 64 features of batch row j (in-lane 16 + lanes l^16, l^32)
__device__ inline float feat_sum(const f32x4 (&x)[4]) {
  return row_allsum(lane_sum16(x));
}

// LayerNorm statistics + normalisation in place: x <- (x - mean) * rstd   (eps = 1e-5, biased var)
// ORL_LN_ONEPASS (build-time switch, round 5): sum and sum of squares in ONE cross-lane butterfly (row_allsum2), var = E[x^2] -
// mean^2 (clamped at 0), x <- x * rstd - mean * rstd: one dependent cross-lane round trip and 16 subtractions less per pass.  The
// subtraction loses ~eps32 * mean^2 / var of relative accuracy in var - 1e-7 for the activations of these towers (mean^2 ~ var),
// and is bounded by eps = 1e-5 under the root when a row is nearly constant.
// Round 6 (VERDICT r5 item 4 / ADVICE r5): the shortcut is GUARDED.  E[x^2] - mean^2 is trusted only while it keeps at least
// 1 / LN_GUARD of E[x^2], i.e. mean^2 <= (LN_GUARD - 1) var: the cancellation then costs at most ~LN_GUARD eps32 ~ 1e-6 of relative
// accuracy in var.  A tile with ANY row beyond that (a large common offset: big b1 / b2 after long training, un-normalised
// observations) takes the reference's two-pass form (x - mean first, then the centred sum of squares - nn.LayerNorm,
// /root/reference/openrl/modules/networks/utils/mlp.py:8-46, is stable for any row).  The test is one v_cmp + one wave-uniform
// branch on the one-pass path; the mean of both forms is the same sum.  tests/test_layernorm_adversarial_gpu.py drives
// mean / std up to 3 000 through act / evaluate / update / recurrent update.
#ifndef ORL_LN_ONEPASS
#define ORL_LN_ONEPASS 1   // round 5: headline 2.702 -> 2.668 ms, 512-env shard 0.848 -> 0.831 ms in three same-box alternations; every parity test unchanged
#endif
#ifndef ORL_LN_GUARD
#define ORL_LN_GUARD 17.0f
#endif
// eps: 1e-5, or 1e-5 x 4^k for rows that arrive scaled by 2^k (the fp16 split GEMMs' accumulators): the same xhat comes out and
// rstd is that of the scaled row (= the true one x 2^-k: v_rsq of an argument scaled by an even power of two has the same mantissa)
__device__ inline void ln_normalize_T(f32x4 (&x)[4], float& rstd, const float eps = 1e-5f) {
#if ORL_LN_ONEPASS
  float s1 = lane_sum16(x), s2 = lane_dot16(x, x);
  row_allsum2(s1, s2);
  const float mean1 = s1 * (1.0f / 64.0f);
  const float ex2 = s2 * (1.0f / 64.0f);
  const float var1 = ex2 - mean1 * mean1;
#ifndef ORL_LN_NOGUARD   // (A/B switch: the unguarded round-5 form)
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(var1 * ORL_LN_GUARD < ex2) != 0ull, 0)) {
    // ill-conditioned row in this tile: centre first, then both statistics of the CENTRED values in one butterfly.  The first
    // mean carries a rounding error of ~eps32 |mean| - at mean / std = 3 000 that alone is 2e-4 std, a common shift of the row
    // (measured: values 3.7 x further from the float64 result than torch's own fp32 LayerNorm); the centred values are small
    // and their sum is nearly exact, so one more subtraction removes it (error ~eps32 std: closer to float64 than torch's).
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = x[m] - mean1;
    float c1 = lane_sum16(x), c2 = lane_dot16(x, x);
    row_allsum2(c1, c2);
    const float dm = c1 * (1.0f / 64.0f);
    const float v = fmaxf(c2 * (1.0f / 64.0f) - dm * dm, 0.f);
    rstd = __builtin_amdgcn_rsqf(v + eps);
    const float sh2 = -dm * rstd;
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = x[m] * rstd + sh2;
    return;
  }
#endif
  rstd = __builtin_amdgcn_rsqf(fmaxf(var1, 0.f) + eps);
  const float shift = -mean1 * rstd;
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = x[m] * rstd + shift;
  return;
#endif
  const float mean = feat_sum(x) * (1.0f / 64.0f);
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = x[m] - mean;
  float v = lane_dot16(x, x);
  v = row_allsum(v);
  // v_rsq_f32 (1 ulp) instead of the ~20-instruction correctly-rounded sqrt + IEEE division sequence
  rstd = __builtin_amdgcn_rsqf(v * (1.0f / 64.0f) + eps);
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = x[m] * rstd;
}

// out = xhat * g + be
__device__ inline void ln_affine_T(const f32x4 (&xhat)[4], const float* __restrict__ g, const float* __restrict__ be,
                                   int q, f32x4 (&out)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x4 gg = *(const f32x4*)(g + 16 * m + 4 * q);
    const f32x4 bb = *(const f32x4*)(be + 16 * m + 4 * q);
    out[m] = xhat[m] * gg + bb;
  }
}

__device__ inline void relu_T(f32x4 (&x)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) x[m][r] = fmaxf(x[m][r], 0.f);
  }
}

// head: out[c] = b3[c] + sum_f W3[c][f] * n2[f]  for c < NO (compile-time bound, runtime n_out).
// Every lane of a batch row ends up with the full result.
template <int NO>
__device__ inline void head_T(const float* __restrict__ W3s, const float* __restrict__ b3s, int n_out,
                              const f32x4 (&n2)[4], int q, float (&out)[NO]) {
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    float p = 0.f;
    if (c < n_out) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 w = *(const f32x4*)(W3s + c * HID + 16 * m + 4 * q);
        p += (w[0] * n2[m][0] + w[1] * n2[m][1]) + (w[2] * n2[m][2] + w[3] * n2[m][3]);
      }
      p = row_allsum(p);
      p += b3s[c];
    }
    out[c] = p;
  }
}

// head_T with the head's weights resident in registers (the fused rollout: W3 / b3 do not change during a launch, but the
// compiler may not keep LDS contents in registers across a barrier - 8 ds_read_b128 + 2 ds_read_b32 and their waits on every
// step's serial chain at Discrete(2)).  Same products, same order of additions as head_T; classes >= n_out hold zeros, so
// all NO chains run unconditionally (no scalar branch between them: they interleave) and return 0 as head_T does.
template <int NO>
struct HeadRegs {
  f32x4 w[NO][4];
  float b[NO];
};

template <int NO>
__device__ inline void head_regs_load(const float* __restrict__ W3s, const float* __restrict__ b3s, int n_out, int q,
                                      HeadRegs<NO>& H) {
#pragma unroll
  for (int c = 0; c < NO; ++c) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      H.w[c][m] = c < n_out ? *(const f32x4*)(W3s + c * HID + 16 * m + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    H.b[c] = c < n_out ? b3s[c] : 0.f;
  }
}

template <int NO>
__device__ inline void head_T_regs(const HeadRegs<NO>& H, const f32x4 (&n2)[4], float (&out)[NO]) {
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    float p = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 w = H.w[c][m];
      p += (w[0] * n2[m][0] + w[1] * n2[m][1]) + (w[2] * n2[m][2] + w[3] * n2[m][3]);
    }
    p = row_allsum(p);
    out[c] = p + H.b[c];
  }
}

// LDS operations of one wave execute in issue order, so a later ds_read observes an earlier ds_write of
// ANY lane of the same wave; only the compiler has to be kept from reordering across this point (it still
// inserts the lgkmcnt wait before a read's first use).
__device__ inline void wave_lds_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Wide heads (5..16 outputs) on MFMA: logits^T[16 c x 16 rows] = W3p[16 x 64] n2^T in 16 MFMAs, where W3p is the
// zero-padded [16][W2S] image TowerLds::W3P.  Lane (j, q) receives logits 4q..4q+3 of row j; the tile takes one trip
// through a wave-private [16][16] LDS scratch so that every lane of a row ends up with all of them, as head_T does.
template <int NO>
__device__ inline void head_mfma_T(const float* __restrict__ W3p, const float* __restrict__ b3s, int n_out,
                                   const f32x4 (&n2)[4], float* __restrict__ tile, int j, int q, float (&out)[NO]) {
  const int no4 = (n_out + 3) & ~3;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (4 * q < no4) acc = *(const f32x4*)(b3s + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const f32x4 a4 = *(const f32x4*)(W3p + j * W2S + 16 * mi + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = ORL_MFMA(a4[r], n2[mi][r], acc);
  }
  *(f32x4*)(tile + j * 16 + 4 * q) = acc;
  wave_lds_fence();
#pragma unroll
  for (int b = 0; b < NO / 4; ++b) {
    const f32x4 v = *(const f32x4*)(tile + j * 16 + 4 * b);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[4 * b + r] = v[r];
  }
  wave_lds_fence();
}

// Whole trunk forward for one tile: returns n2 (post-LN2 activations) in T layout.
template <class XB>
__device__ inline void trunk_fwd_T(const float* __restrict__ lds, const TowerLds& tw, XB xb, int j, int q,
                                   f32x4 (&n2)[4]) {
  f32x4 z[4], n1[4];
  float rstd;
  load_vec_T(lds + tw.b1, q, z);
  fc1_T(lds + tw.W1, tw.DP, xb, z, j, q);
  relu_T(z);
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lds + tw.g1, lds + tw.be1, q, n1);
  load_vec_T(lds + tw.b2, q, z);
  mm64_T(lds + tw.W2, n1, z, j, q);
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lds + tw.g2, lds + tw.be2, q, n2);
}

// ---- distribution heads ----------------------------------------------------------------------------

// Categorical over lg[0..n_out): masked logits, log-sum-exp, returns lse.  (distributions.py:68-72)
template <int NO>
__device__ inline float cat_lse(float (&lg)[NO], int n_out, const float* __restrict__ amask_row) {
  float mx = -3.0e38f;
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    if (c < n_out) {
      if (amask_row != nullptr && amask_row[c] == 0.f) lg[c] = -6e4f;
      mx = fmaxf(mx, lg[c]);
    }
  }
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    if (c < n_out) se += __expf(lg[c] - mx);  // v_exp_f32 / v_log_f32: 1-ulp hardware transcendentals
  }
  return mx + __logf(se);
}

// inverse-CDF sample of softmax(lg) with uniform u in [0,1)
template <int NO>
__device__ inline int cat_sample(const float (&lg)[NO], int n_out, float lse, float u) {
  float p[NO];
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    p[c] = (c < n_out) ? __expf(lg[c] - lse) : 0.f;
    tot += p[c];
  }
  const float ut = u * tot;
  int a = -1, last = 0;
  float cum = 0.f;
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    if (c < n_out) {
      cum += p[c];
      if (p[c] > 0.f) last = c;
      if (a < 0 && cum > ut) a = c;
    }
  }
  return a < 0 ? last : a;
}

template <int NO>
__device__ inline int cat_mode(const float (&lg)[NO], int n_out) {
  int a = 0;
  float best = lg[0];
#pragma unroll
  for (int c = 1; c < NO; ++c) {
    if (c < n_out && lg[c] > best) { best = lg[c]; a = c; }
  }
  return a;
}

template <int NO>
__device__ inline float pick(const float (&v)[NO], int a) {
  // every candidate goes through an opaque copy: hipcc recognises the plain select chain as a table lookup and compiles it to an
  // INDEXED SCRATCH LOAD of v[] followed by s_waitcnt vmcnt(0) (found in cfg4's rollout, profiles/r05_experiments.md section 8)
  float r = v[0];
#pragma unroll
  for (int c = 1; c < NO; ++c) {
    float x = v[c];
    asm volatile("" : "+v"(x));
    r = (c == a) ? x : r;
  }
  return r;
}

// standard normal pair from two uniforms (Box-Muller)
__device__ inline void box_muller(uint32_t x0, uint32_t x1, float& n0, float& n1) {
  const float u1 = u01_open0(x0), u2 = u01(x1);
  const float rad = sqrtf(-2.0f * logf(u1));
  const float ang = 6.28318530717958647692f * u2;
  n0 = rad * cosf(ang);
  n1 = rad * sinf(ang);
}

}  // namespace orl
