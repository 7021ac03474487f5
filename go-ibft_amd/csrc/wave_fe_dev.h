// wave_fe_dev.h — ONE WAVEFRONT PER SIGNATURE: field elements spread over the lanes of a row.
//
// Product code (the host build exists only for tests, through wave_emul.h).  The lane- and
// group-kernels keep a whole field element in one lane (10 VGPRs) and pay ≈224 dependent VALU
// instructions per multiplication; at N = 1 024 rows the chip (1 024 SIMDs) has exactly one
// SIMD per signature, so what matters is the LATENCY of one signature, not lane throughput.
// Here a field element is one VGPR: lane 16·ρ + i of DPP row ρ holds limb i (radix 2^26,
// i = 0..9; lanes 10..15 of every row hold 0).  A multiplication is then
//   ten broadcasts of a_i (row_newbcast) × ten row-shifted copies of b (row_shr) feeding one
//   v_mad_u64_u32 each — lane k accumulates column k, lanes 10..15 the columns 10..15, three more
//   mads give columns 16..18 — and a carry-save reduction with 2^260 ≡ 0x3D10 + 0x400·2^26,
// 72 VALU instructions for FOUR independent products (one per row) instead of 224 for one.  The
// four rows carry four independent pieces of the scalar multiplication (GLV half × upper /
// lower 64 bits), so the per-row instruction stream is an ordinary sequential point formula
// and every branch is wave-uniform by construction (a wavefront holds one signature).
//
// Contents: cross-lane primitives · wfe_mul / linear ops · row ↔ lane layout · group law per row ·
// √ chain riding on the 3-row prefix doublings · modinv_wave (safegcd, limbs over lanes) ·
// recover_pubkey_wave (cold path) · verify_known_wave (warm path).  DESIGN.md §4 has the measurements.
//
// RULE: a cross-lane primitive must never sit under lane-dependent control flow (`c ? f(dpp) : x` with
// a per-row c executes the DPP in some rows only).  The host emulator tags every rendezvous with its
// primitive and aborts the test when lanes disagree.
//
// Magnitudes: U = 2^26 + 2^20; "magnitude m" = every limb ≤ m·U.  wfe_mul accepts magnitudes
// ≤ 15 on both inputs (10·(15U)² < 2^64) and returns magnitude 1; wfe_neg(a, m) = K_m − a with
// K_m ≡ 0 (mod p), K_m,i ∈ [m·U, m·U + 2^26), so the result has magnitude m + 1.
#pragma once
#include "modinv_dev.h"
#include "recover_dev.h"
#include "verify_dev.h"
#if !defined(__HIP_DEVICE_COMPILE__) && defined(IBFT_WAVE_EMUL)
#include "wave_emul.h"  // tests only: 64 lockstep coroutines stand in for the lanes
#endif

// big functions: forced inline on the device (the only outlined piece is wfe_mul_fn), ordinary
// inline + an outlined multiply in the host test build (forcing them there makes the compile explode)
#if defined(__HIP_DEVICE_COMPILE__)
#define WVF __host__ __device__ __forceinline__
#else
#define WVF __host__ __device__ inline
#endif

namespace wv {

using secp::aff;
using secp::fe;
using secp::jac;
using secp::M26;
using secp::u256;

// ---- cross-lane primitives -----------------------------------------------------------------
HD uint32_t lane_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  return __lane_id();
#elif defined(IBFT_WAVE_EMUL)
  return (uint32_t)wave_emul::lane();
#else
  return 0;  // host pass of the product build: never called
#endif
}
// lane k of a row ← lane k − N (zero shifted in)
template <int N>
HD uint32_t row_shr(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true);
#elif defined(IBFT_WAVE_EMUL)
  const int l = wave_emul::lane();
  return wave_emul::xchg(v, (l & 15) >= N ? l - N : -1, 0x100u + N);
#else
  return v;
#endif
}
// lane k of a row ← lane k + N (zero shifted in)
template <int N>
HD uint32_t row_shl(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
#elif defined(IBFT_WAVE_EMUL)
  const int l = wave_emul::lane();
  return wave_emul::xchg(v, (l & 15) + N <= 15 ? l + N : -1, 0x200u + N);
#else
  return v;
#endif
}
// every lane of a row ← lane N of that row (gfx90a+ row_newbcast)
template <int N>
HD uint32_t row_bcast(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xF, 0xF, true);
#elif defined(IBFT_WAVE_EMUL)
  const int l = wave_emul::lane();
  return wave_emul::xchg(v, (l & ~15) + N, 0x300u + N);
#else
  return v;
#endif
}
HD uint32_t lane_xor(uint32_t v, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__shfl_xor((int)v, off, 64);
#elif defined(IBFT_WAVE_EMUL)
  return wave_emul::xchg(v, wave_emul::lane() ^ off, 0x400u + (uint32_t)off);
#else
  return v + (uint32_t)off;
#endif
}
// value of v held by lane `src` (any lane of the wavefront; ds_bpermute)
HD uint32_t lane_perm(uint32_t v, uint32_t src) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
#elif defined(IBFT_WAVE_EMUL)
  return wave_emul::xchg(v, (int)src, 0x500u);
#else
  return v + src;
#endif
}
// gfx950's row swaps (v_permlane16_swap_b32 / v_permlane32_swap_b32): VALU instructions that exchange whole DPP rows
// between TWO registers — no LDS round trip, unlike ds_bpermute (which a lone wavefront per SIMD waits out in full).
// With d = [d0 d1 d2 d3], s = [s0 s1 s2 s3] (one letter per row of sixteen lanes):
//   rows_swap16(d, s): a = [d0 s0 d2 s2], b = [d1 s1 d3 s3]     (odd rows of d ↔ even rows of s)
//   rows_swap32(d, s): a = [d0 d1 s0 s1], b = [d2 d3 s2 s3]     (upper half of d ↔ lower half of s)
// so rows_swap16(v, v) followed by rows_swap32 of a result with itself broadcasts one row to all four.
struct u32x2 {
  uint32_t a, b;
};
#ifndef IBFT_ROW_SWAPS
#define IBFT_ROW_SWAPS 1  // 0: the same exchanges through ds_bpermute (A/B timing)
#endif
HD u32x2 rows_swap16(uint32_t d, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
#if IBFT_ROW_SWAPS
  const auto r = __builtin_amdgcn_permlane16_swap(d, s, false, false);
  return u32x2{r[0], r[1]};
#else
  const uint32_t l = __lane_id();
  const uint32_t sd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l - 16u) << 2), (int)s);
  const uint32_t du = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l + 16u) << 2), (int)d);
  return u32x2{(l & 16u) ? sd : d, (l & 16u) ? s : du};
#endif
#elif defined(IBFT_WAVE_EMUL)
  const int l = wave_emul::lane();
  const bool odd = (l & 16) != 0;
  const uint32_t sd = wave_emul::xchg(s, odd ? l - 16 : l, 0x610u);
  const uint32_t du = wave_emul::xchg(d, odd ? l : l + 16, 0x611u);
  return u32x2{odd ? sd : d, odd ? s : du};
#else
  return u32x2{d, s};
#endif
}
HD u32x2 rows_swap32(uint32_t d, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
#if IBFT_ROW_SWAPS
  const auto r = __builtin_amdgcn_permlane32_swap(d, s, false, false);
  return u32x2{r[0], r[1]};
#else
  const uint32_t l = __lane_id();
  const uint32_t sd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l - 32u) << 2), (int)s);
  const uint32_t du = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((l + 32u) << 2), (int)d);
  return u32x2{(l & 32u) ? sd : d, (l & 32u) ? s : du};
#endif
#elif defined(IBFT_WAVE_EMUL)
  const int l = wave_emul::lane();
  const bool up = (l & 32) != 0;
  const uint32_t sd = wave_emul::xchg(s, up ? l - 32 : l, 0x620u);
  const uint32_t du = wave_emul::xchg(d, up ? l : l + 32, 0x621u);
  return u32x2{up ? sd : d, up ? s : du};
#else
  return u32x2{d, s};
#endif
}
// the value the same lane of the row 16 / 32 lanes away holds (lane ^ 16, lane ^ 32)
HD uint32_t row_partner16(uint32_t v) {
  const u32x2 r = rows_swap16(v, v);  // a = [v0 v0 v2 v2], b = [v1 v1 v3 v3]
  return (lane_id() & 16u) ? r.a : r.b;
}
HD uint32_t row_partner32(uint32_t v) {
  const u32x2 r = rows_swap32(v, v);  // a = [v0 v1 v0 v1], b = [v2 v3 v2 v3]
  return (lane_id() & 32u) ? r.a : r.b;
}
HD bool any(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any(c ? 1 : 0) != 0;
#elif defined(IBFT_WAVE_EMUL)
  return wave_emul::ballot(c) != 0;
#else
  return c;
#endif
}

// ---- per-lane constants ----------------------------------------------------------------------
constexpr uint32_t WFE_U = (1u << 26) + (1u << 20);
HD uint32_t wneg_limb(int which, int i) {  // K_1, K_2, K_8 (tests/test_dev_wave_host.py re-derives them)
  const uint32_t K[3][10] = {
      {0x07FFBF1Fu, 0x07FFFBBEu, 0x07FFFFFEu, 0x07FFFFFEu, 0x07FFFFFEu, 0x07FFFFFEu, 0x07FFFFFEu, 0x07FFFFFEu,
       0x07FFFFFEu, 0x043FFFFEu},
      {0x0BFF820Fu, 0x0BFFF7BDu, 0x0BFFFFFDu, 0x0BFFFFFDu, 0x0BFFFFFDu, 0x0BFFFFFDu, 0x0BFFFFFDu, 0x0BFFFFFDu,
       0x0BFFFFFDu, 0x083FFFFDu},
      {0x23FE0C0Du, 0x23FFDF37u, 0x23FFFFF7u, 0x23FFFFF7u, 0x23FFFFF7u, 0x23FFFFF7u, 0x23FFFFF7u, 0x23FFFFF7u,
       0x23FFFFF7u, 0x20BFFFF7u}};
  return K[which][i];
}
struct wk {
  uint32_t li;    // lane within the row
  uint32_t row;   // 0..3
  uint32_t act;   // all ones on limb lanes (li < 10)
  uint32_t m3;    // li < 3 ? M26 : (li < 10 ? all ones : 0): the last AND of wfe_reduce (clears the idle lanes too)
  uint32_t lt3;   // all ones for li < 3
  uint32_t lt9;   // all ones for li < 9
  uint32_t kr;    // 2^260 mod p as limbs: 0x3D10 in lane 0, 0x400 in lane 1
  uint32_t k1, k2, k8;  // negation constants
};
HD wk wk_init() {
  wk k;
  const uint32_t l = lane_id();
  k.li = l & 15u;
  k.row = l >> 4;
  k.act = k.li < 10 ? 0xFFFFFFFFu : 0u;
  k.m3 = k.li < 3 ? M26 : (k.li < 10 ? 0xFFFFFFFFu : 0u);
  k.lt3 = k.li < 3 ? 0xFFFFFFFFu : 0u;
  k.lt9 = k.li < 9 ? 0xFFFFFFFFu : 0u;
  k.kr = k.li == 0 ? 0x3D10u : (k.li == 1 ? 0x400u : 0u);
  k.k1 = k.k2 = k.k8 = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    k.k1 = k.li == (uint32_t)i ? wneg_limb(0, i) : k.k1;
    k.k2 = k.li == (uint32_t)i ? wneg_limb(1, i) : k.k2;
    k.k8 = k.li == (uint32_t)i ? wneg_limb(2, i) : k.k8;
  }
  return k;
}

// ---- multiplication --------------------------------------------------------------------------
HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
HD uint32_t mul24(uint32_t a, uint32_t b) {  // both < 2^24
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}

template <int I>
HD void wfe_mul_step(uint32_t a, uint32_t b, uint64_t &lo, uint64_t &hi) {
  const uint32_t ai = row_bcast<I>(a);
  const uint32_t tl = row_shr<(I == 0 ? 1 : I)>(b);  // (I == 0 uses b itself below)
  lo = mad64(ai, I == 0 ? b : tl, lo);
  if (I >= 7) hi = mad64(ai, row_shl<16 - (I >= 7 ? I : 7)>(b), hi);  // columns 16..18 in lanes 0..2
}

// column sums of a·b added to (lo, hi): lane k of lo = column k (k = 0..15), lane k of hi = column 16 + k (k = 0..2)
HD void wfe_prod(uint32_t a, uint32_t b, uint64_t &lo, uint64_t &hi) {
  {
    const uint32_t a0 = row_bcast<0>(a);
    lo = mad64(a0, b, lo);
  }
  wfe_mul_step<1>(a, b, lo, hi);
  wfe_mul_step<2>(a, b, lo, hi);
  wfe_mul_step<3>(a, b, lo, hi);
  wfe_mul_step<4>(a, b, lo, hi);
  wfe_mul_step<5>(a, b, lo, hi);
  wfe_mul_step<6>(a, b, lo, hi);
  wfe_mul_step<7>(a, b, lo, hi);
  wfe_mul_step<8>(a, b, lo, hi);
  wfe_mul_step<9>(a, b, lo, hi);
}
// Code-generation hints (device build only; the host build computes the same integers without them).
//   fence_v: an empty asm that redefines a VGPR value.  Between two additions it keeps instruction selection from
//     fusing them into v_add3_u32 — whose operands cannot carry a DPP modifier, so each shifted summand then costs a
//     v_mov_b32_dpp of its own — and lets the DPP combiner fold every shifted summand into a v_add_u32_dpp.
//   opaque_s: a wave-uniform constant the optimiser cannot see through.  ·0x400 would otherwise be strength-reduced to
//     v_lshlrev_b64 + v_lshl_add_u64 (two issue slots) where one v_mad_u64_u32 does it.
// One wavefront per SIMD issues one instruction every ≈4 cycles whatever it is: instructions are the cost.
HD uint32_t fence_v(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(v));
#endif
  return v;
}
HD uint32_t opaque_s(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(v));
#endif
  return v;
}
// 64-bit column sums (each < 2^64) → magnitude 1 (< 2^26 + 2^16), idle lanes zero.
// m3z: M26 in lanes 0..2, all ones in lanes 3..9, ZERO in the idle lanes (wk::m3z) — the last AND also clears what the
// second fold leaves in lanes 10, 11.
HD uint32_t wfe_reduce(uint64_t lo, uint64_t hi, uint32_t act, uint32_t m3z, uint32_t lt3) {
  const uint32_t c400 = opaque_s(0x400u);
  // cut into 26-bit chunks
  const uint32_t c0 = (uint32_t)lo & M26, c1 = (uint32_t)(lo >> 26) & M26, c2 = (uint32_t)(lo >> 52);
  const uint32_t h0 = (uint32_t)hi & M26, h1 = (uint32_t)(hi >> 26) & M26, h2 = (uint32_t)(hi >> 52);
  // S: positions 0..15, T: positions 16..20 (lane j = position 16 + j)
  // (nested so that every addition has ONE shifted summand — shr2 = shr1∘shr1, shl15 = shl14∘shl1, zero-filling all —
  // and folds it as a DPP operand; only the last one needs the fence)
  const uint32_t S = c0 + row_shr<1>(c1 + row_shr<1>(c2));
  const uint32_t T0 = h0 + row_shr<1>(h1 + row_shr<1>(h2));
  const uint32_t T = fence_v(T0) + row_shl<14>(c2 + row_shl<1>(c1));
  // H: lane j = position 10 + j (j = 0..10); fold with 2^260 ≡ 0x3D10 + 0x400·2^26
  const uint32_t H = row_shl<10>(S) + row_shr<6>(T);
  uint64_t V = mad64(H, 0x3D10u, (uint64_t)(S & act));
  V = mad64(row_shr<1>(H), c400, V);  // < 2^42 in lanes 0..11
  const uint32_t U = ((uint32_t)V & M26) + row_shr<1>((uint32_t)(V >> 26));  // positions 0..11
  // positions 10, 11 once more (lanes 10, 11 of U itself stay in V2 and are cleared by m3z)
  const uint32_t H2 = row_shl<10>(U);  // lanes 0, 1
  uint64_t V2 = mad64(H2, 0x3D10u, (uint64_t)U);
  V2 = mad64(row_shr<1>(H2), c400, V2);  // lanes 0..2 < 2^41, lanes 3..9 = U
  return ((uint32_t)V2 & m3z) + row_shr<1>((uint32_t)(V2 >> 26) & lt3);
}
// both inputs of magnitude ≤ 15 with zero idle lanes; result magnitude 1 (< 2^26 + 2^16), idle lanes zero
HD uint32_t wfe_mul_body(uint32_t a, uint32_t b, uint32_t act, uint32_t m3, uint32_t lt3) {
  uint64_t lo = 0, hi = 0;
  wfe_prod(a, b, lo, hi);
  return wfe_reduce(lo, hi, act, m3, lt3);
}
// a·b + c·d with ONE reduction (107 VALU instructions instead of 2 × 72 + the carry pass of the sum): the point
// formulas end in "product − product" (Y3 = E·(D − X3) − 8·B², Y3 = r·(V − X3) − 2·Y1·J), and the subtrahend enters
// as a negated factor.  Magnitudes: ma·mb + mc·md ≤ 225 (10·225·U² < 2^64).
HD uint32_t wfe_mul2_body(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t act, uint32_t m3, uint32_t lt3) {
  uint64_t lo = 0, hi = 0;
  wfe_prod(a, b, lo, hi);
  wfe_prod(c, d, lo, hi);
  return wfe_reduce(lo, hi, act, m3, lt3);
}
// INL = true pastes the 72 VALU instructions in place (hot loops: no call/return, no argument moves,
// and the scheduler can fill the DPP wait states across neighbouring multiplications); INL = false
// calls one outlined copy (straight-line code that runs once — table, G additions, joins — where
// 500-byte bodies would only thrash the 64 KB instruction cache: measured +0.03 ms when inlined).
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __attribute__((noinline)) uint32_t wfe_mul_fn(uint32_t a, uint32_t b, uint32_t act, uint32_t m3,
                                                               uint32_t lt3) {
  return wfe_mul_body(a, b, act, m3, lt3);
}
static __device__ __attribute__((noinline)) uint32_t wfe_mul2_fn(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t act,
                                                                uint32_t m3, uint32_t lt3) {
  return wfe_mul2_body(a, b, c, d, act, m3, lt3);
}
template <bool INL = false>
HD uint32_t wfe_mul(uint32_t a, uint32_t b, const wk &k) {
  return INL ? wfe_mul_body(a, b, k.act, k.m3, k.lt3) : wfe_mul_fn(a, b, k.act, k.m3, k.lt3);
}
template <bool INL = false>
HD uint32_t wfe_mul2(uint32_t a, uint32_t b, uint32_t c, uint32_t d, const wk &k) {
  return INL ? wfe_mul2_body(a, b, c, d, k.act, k.m3, k.lt3) : wfe_mul2_fn(a, b, c, d, k.act, k.m3, k.lt3);
}
#else
static __host__ __device__ __attribute__((noinline)) uint32_t wfe_mul_fn(uint32_t a, uint32_t b, uint32_t act,
                                                                        uint32_t m3, uint32_t lt3) {
  return wfe_mul_body(a, b, act, m3, lt3);
}
static __host__ __device__ __attribute__((noinline)) uint32_t wfe_mul2_fn(uint32_t a, uint32_t b, uint32_t c, uint32_t d,
                                                                         uint32_t act, uint32_t m3, uint32_t lt3) {
  return wfe_mul2_body(a, b, c, d, act, m3, lt3);
}
template <bool INL = false>
__host__ __device__ inline uint32_t wfe_mul(uint32_t a, uint32_t b, const wk &k) {
  return wfe_mul_fn(a, b, k.act, k.m3, k.lt3);
}
template <bool INL = false>
__host__ __device__ inline uint32_t wfe_mul2(uint32_t a, uint32_t b, uint32_t c, uint32_t d, const wk &k) {
  return wfe_mul2_fn(a, b, c, d, k.act, k.m3, k.lt3);
}
#endif
template <bool INL = false>
HD uint32_t wfe_sqr(uint32_t a, const wk &k) { return wfe_mul<INL>(a, a, k); }
HD uint32_t wfe_sqr_n(uint32_t a, int n, const wk &k) {
  for (int i = 0; i < n; i++) a = wfe_sqr(a, k);
  return a;
}

// ---- linear operations -------------------------------------------------------------------------
HD uint32_t wfe_neg1(uint32_t a, const wk &k) { return k.k1 - a; }  // a magnitude ≤ 1 → 2
HD uint32_t wfe_neg2(uint32_t a, const wk &k) { return k.k2 - a; }  // ≤ 2 → 3
HD uint32_t wfe_neg8(uint32_t a, const wk &k) { return k.k8 - a; }  // ≤ 8 → 9
// any limbs < 2^32 → magnitude 1: one carry pass, the carry out of limb 9 folded with 2^260 mod p
HD uint32_t wfe_weak(uint32_t v, const wk &k) {
  const uint32_t c = v >> 26;
  const uint32_t c9 = row_bcast<9>(c);
  return (v & M26) + row_shr<1>(c & k.lt9) + mul24(c9, k.kr);
}

// ---- row layout ↔ lane layout -------------------------------------------------------------------
// every lane of the row gets the whole element
HD fe gather(uint32_t w) {
  fe r;
  r.n[0] = row_bcast<0>(w);
  r.n[1] = row_bcast<1>(w);
  r.n[2] = row_bcast<2>(w);
  r.n[3] = row_bcast<3>(w);
  r.n[4] = row_bcast<4>(w);
  r.n[5] = row_bcast<5>(w);
  r.n[6] = row_bcast<6>(w);
  r.n[7] = row_bcast<7>(w);
  r.n[8] = row_bcast<8>(w);
  r.n[9] = row_bcast<9>(w);
  return r;
}
// a identical in all lanes of the row (or at least: lane i holds the right n[i])
HD uint32_t scatter(const fe &a, const wk &k) {
  uint32_t w = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) w = k.li == (uint32_t)i ? a.n[i] : w;
  return w;
}
HD bool wfe_is_zero(uint32_t w) { return secp::fe_is_zero(gather(w)); }  // magnitude ≤ 31
// Cheap NECESSARY condition for z ≡ 0 (mod p), z an output of wfe_mul / wfe_mul2: lane 0 then holds exactly z mod 2^26
// (wfe_reduce masks it and shifts nothing in) and the value is below 2^260·(1 + 2^-10) < 17p, so a multiple of p is m·p with
// m ≤ 16 and −z mod 2^26 = m·977.  Never misses a zero; says "maybe" for 2.3·10^-4 of the non-zero values.  One
// broadcast, a subtraction, a mask and a compare instead of the ≈75 instructions of wfe_is_zero.
HD bool wfe_z_maybe_zero(uint32_t z) { return ((0u - row_bcast<0>(z)) & M26) <= 16u * 977u; }

// ---- group law (Jacobian, a = 0), one point per row ------------------------------------------------
struct wjac {
  uint32_t x, y, z;
  bool inf;  // uniform within a row
};
struct waff {
  uint32_t x, y;
};
HD wjac wjac_inf() { return wjac{0u, 0u, 0u, true}; }
HD wjac wjac_select(bool c, const wjac &a, const wjac &b) {
  wjac r;
  r.x = c ? a.x : b.x;
  r.y = c ? a.y : b.y;
  r.z = c ? a.z : b.z;
  r.inf = c ? a.inf : b.inf;
  return r;
}
HD wjac wjac_from_aff(const waff &a, const wk &k) { return wjac{a.x, a.y, k.li == 0 ? 1u : 0u, false}; }

// dbl-2009-l with S = M (a squaring costs what a multiplication costs here): D = 2·((X + B)² − A − C) is 4·X·B —
// one multiplication that waits for B only —, and Y3 = E·(D − X3) − 8·B² is ONE fused multiply-add, so C = B² is never
// formed: five multiplications and a fused pair (≈ 6.5) instead of seven, the linear work of t and D gone.
template <bool INL = false>
WVF wjac wjac_dbl(const wjac &p, const wk &k) {
  const uint32_t A = wfe_sqr<INL>(p.x, k);
  const uint32_t B = wfe_sqr<INL>(p.y, k);
  const uint32_t XB = wfe_mul<INL>(p.x, B, k);
  wjac r;
  r.z = wfe_mul<INL>(2u * p.y, p.z, k);
  const uint32_t E = 3u * A;                             // 3
  const uint32_t F = wfe_sqr<INL>(E, k);
  r.x = wfe_weak(F + wfe_neg8(8u * XB, k), k);           // X3 = F − 2D, 2D = 8·X·B: 1 + 9 → 1
  // Y3 = E·(D − X3) + (−8B)·B: magnitudes 3·(4 + 2) + 9·1 = 27
  r.y = wfe_mul2<INL>(E, 4u * XB + wfe_neg1(r.x, k), wfe_neg8(8u * B, k), B, k);
  r.inf = p.inf;
  return r;
}
// add-2007-bl with the exceptional cases (P = Q, P = −Q, ∞) resolved per row
template <bool INL = false>
WVF wjac wjac_add(const wjac &p, const wjac &q, const wk &k) {
  const uint32_t z1z1 = wfe_sqr<INL>(p.z, k), z2z2 = wfe_sqr<INL>(q.z, k);
  const uint32_t u1 = wfe_mul<INL>(p.x, z2z2, k), u2 = wfe_mul<INL>(q.x, z1z1, k);
  const uint32_t s1 = wfe_mul<INL>(wfe_mul<INL>(p.y, q.z, k), z2z2, k);
  const uint32_t s2 = wfe_mul<INL>(wfe_mul<INL>(q.y, p.z, k), z1z1, k);
  const uint32_t h = u2 + wfe_neg1(u1, k);   // 3
  const uint32_t rr = s2 + wfe_neg1(s1, k);  // 3
  const uint32_t i = wfe_sqr<INL>(2u * h, k);     // in 6
  const uint32_t j = wfe_mul<INL>(h, i, k);
  const uint32_t r2 = 2u * rr;               // 6
  const uint32_t v = wfe_mul<INL>(u1, i, k);
  wjac r;
  r.x = wfe_weak(wfe_sqr<INL>(r2, k) + wfe_neg1(j, k) + wfe_neg2(2u * v, k), k);                 // 6 → 1
  const uint32_t s1j2 = 2u * wfe_mul<INL>(s1, j, k);                                            // 2
  r.y = wfe_weak(wfe_mul<INL>(r2, v + wfe_neg1(r.x, k), k) + wfe_neg2(s1j2, k), k);              // 4 → 1
  const uint32_t zz = wfe_sqr<INL>(p.z + q.z, k) + wfe_neg1(z1z1, k) + wfe_neg1(z2z2, k);        // 5
  r.z = wfe_mul<INL>(zz, h, k);
  r.inf = false;
  // P = ±Q ⇔ H ≡ 0 ⇔ Z3 ≡ 0 (Z1·Z2 ≢ 0 for finite points): wfe_z_maybe_zero(Z3) never misses it, and the exact
  // tests (a gather and a normalisation each) run for one addition in a thousand instead of for every one
  const bool both = !p.inf && !q.inf;
  const bool zmz = wfe_z_maybe_zero(r.z);  // (cross-lane: evaluated by every lane, then combined)
  const bool maybe = both && zmz;
  if (any(maybe)) {  // (measured: laying this path out of line with __builtin_expect costs the rows kernel 1.4 %)
    const bool hz = wfe_is_zero(h) && maybe;
    bool rz = false;
    if (any(hz)) rz = wfe_is_zero(rr);
    const bool same = hz && rz, opposite = hz && !rz;
    if (any(same)) r = wjac_select(same, wjac_dbl<false>(p, k), r);
    r = wjac_select(opposite, wjac_inf(), r);
  }
  r = wjac_select(q.inf, p, r);
  r = wjac_select(p.inf, q, r);
  return r;
}
// madd-2007-bl (q affine, never infinity) with Z3 = (Z1 + H)² − Z1Z1 − HH written as 2·Z1·H (S = M here) and
// Y3 = r·(V − X3) − 2·Y1·J as ONE fused multiply-add: nine multiplications and a fused pair instead of eleven.
// RARE_INL: the doubling of the exceptional case P = Q with its multiplications pasted in too — for a caller that must stay a
// LEAF function (rows_two_adds_fn: a nested call makes the compiler save a register to the stack on every entry)
template <bool INL = false, bool RARE_INL = false>
WVF wjac wjac_add_aff(const wjac &p, const waff &q, const wk &k) {
  const uint32_t z1z1 = wfe_sqr<INL>(p.z, k);
  const uint32_t u2 = wfe_mul<INL>(q.x, z1z1, k);
  const uint32_t s2 = wfe_mul<INL>(wfe_mul<INL>(q.y, p.z, k), z1z1, k);
  const uint32_t h = u2 + wfe_neg1(p.x, k);   // 3
  const uint32_t rr = s2 + wfe_neg2(p.y, k);  // 4 (p.y may be a negated table entry: magnitude ≤ 2)
  const uint32_t hh = wfe_sqr<INL>(h, k);
  const uint32_t i = 4u * hh;                 // 4
  const uint32_t j = wfe_mul<INL>(h, i, k);
  const uint32_t r2 = 2u * rr;                // 8
  const uint32_t v = wfe_mul<INL>(p.x, i, k);
  wjac r;
  r.z = wfe_mul<INL>(2u * p.z, h, k);         // 2 · 3
  r.x = wfe_weak(wfe_sqr<INL>(r2, k) + wfe_neg1(j, k) + wfe_neg2(2u * v, k), k);
  // Y3 = r2·(V − X3) + (−2·Y1)·J: magnitudes 8·3 + 9·1 = 33
  r.y = wfe_mul2<INL>(r2, v + wfe_neg1(r.x, k), wfe_neg8(2u * p.y, k), j, k);
  r.inf = false;
  const wjac qj = wjac_from_aff(q, k);
  const bool zmz = wfe_z_maybe_zero(r.z);  // Z3 = 2·Z1·H: H ≡ 0 ⇒ Z3 ≡ 0 (cross-lane: evaluated by every lane)
  const bool maybe = !p.inf && zmz;
  if (any(maybe)) {  // (measured: laying this path out of line with __builtin_expect costs the rows kernel 1.4 %)
    const bool hz = wfe_is_zero(h) && maybe;
    bool rz = false;
    if (any(hz)) rz = wfe_is_zero(rr);
    const bool same = hz && rz, opposite = hz && !rz;
    if (any(same)) r = wjac_select(same, wjac_dbl<RARE_INL>(qj, k), r);
    r = wjac_select(opposite, wjac_inf(), r);
  }
  r = wjac_select(p.inf, qj, r);
  return r;
}
// Two mixed additions in a row — acc ← acc + q1 (if t1), then + q2 (if t2) — as ONE outlined leaf function with the
// multiplications pasted in (round 6, the A/B form IBFT_ROWS_SHARED_ADDS: see the macro for what it measured).  The idea: the
// row-per-signature recover runs this pair 32 times in its main loop and 8 times for the fixed-base windows, whose own pasted copy
// issues at 3.5 ns per instruction where the main loop's identical additions take 2.0 (profiles/r06e_rows_stage_issue.txt).
#if defined(__HIP_DEVICE_COMPILE__)
struct wjac4 {
  uint32_t x, y, z, inf;
};
static __device__ __attribute__((noinline)) wjac4 rows_two_adds_fn(uint32_t x, uint32_t y, uint32_t z, uint32_t inf, uint32_t q1x,
                                                                  uint32_t q1y, uint32_t q2x, uint32_t q2y, uint32_t t1, uint32_t t2,
                                                                  uint32_t li, uint32_t row, uint32_t act, uint32_t m3, uint32_t lt3,
                                                                  uint32_t lt9, uint32_t kr, uint32_t k1, uint32_t k2, uint32_t k8) {
  wk k;
  k.li = li; k.row = row; k.act = act; k.m3 = m3; k.lt3 = lt3; k.lt9 = lt9; k.kr = kr; k.k1 = k1; k.k2 = k2; k.k8 = k8;
  wjac acc = wjac{x, y, z, inf != 0};
  const wjac s1 = wjac_add_aff<true, true>(acc, waff{q1x, q1y}, k);
  acc = wjac_select(t1 != 0, s1, acc);
  const wjac s2 = wjac_add_aff<true, true>(acc, waff{q2x, q2y}, k);
  acc = wjac_select(t2 != 0, s2, acc);
  return wjac4{acc.x, acc.y, acc.z, acc.inf ? 1u : 0u};
}
#endif
WVF wjac rows_two_adds(const wjac &acc, const waff &q1, const waff &q2, bool t1, bool t2, const wk &k) {
#if defined(__HIP_DEVICE_COMPILE__)
  const wjac4 r = rows_two_adds_fn(acc.x, acc.y, acc.z, acc.inf ? 1u : 0u, q1.x, q1.y, q2.x, q2.y, t1 ? 1u : 0u, t2 ? 1u : 0u, k.li,
                                   k.row, k.act, k.m3, k.lt3, k.lt9, k.kr, k.k1, k.k2, k.k8);
  return wjac{r.x, r.y, r.z, r.inf != 0};
#else
  wjac a = acc;
  const wjac s1 = wjac_add_aff<true>(a, q1, k);
  a = wjac_select(t1, s1, a);
  const wjac s2 = wjac_add_aff<true>(a, q2, k);
  return wjac_select(t2, s2, a);
#endif
}
HD wjac wjac_lane_xor(const wjac &p, int off) {
  wjac r;
  if (off == 16) {
    r.x = row_partner16(p.x);
    r.y = row_partner16(p.y);
    r.z = row_partner16(p.z);
    r.inf = row_partner16(p.inf ? 1u : 0u) != 0;
  } else if (off == 32) {
    r.x = row_partner32(p.x);
    r.y = row_partner32(p.y);
    r.z = row_partner32(p.z);
    r.inf = row_partner32(p.inf ? 1u : 0u) != 0;
  } else {
    r.x = lane_xor(p.x, off);
    r.y = lane_xor(p.y, off);
    r.z = lane_xor(p.z, off);
    r.inf = lane_xor(p.inf ? 1u : 0u, off) != 0;
  }
  return r;
}
HD jac wjac_gather(const wjac &p) {
  jac r;
  r.x = gather(p.x);
  r.y = gather(p.y);
  r.z = gather(p.z);
  r.inf = p.inf;
  return r;
}

// a^((p+1)/4), same addition chain as secp::fe_sqrt_candidate (a of magnitude ≤ 8)
WVF uint32_t wfe_sqrt_candidate(uint32_t a, const wk &k) {
  const uint32_t x2 = wfe_mul(wfe_sqr(a, k), a, k);
  const uint32_t x3 = wfe_mul(wfe_sqr(x2, k), a, k);
  const uint32_t x6 = wfe_mul(wfe_sqr_n(x3, 3, k), x3, k);
  const uint32_t x9 = wfe_mul(wfe_sqr_n(x6, 3, k), x3, k);
  const uint32_t x11 = wfe_mul(wfe_sqr_n(x9, 2, k), x2, k);
  const uint32_t x22 = wfe_mul(wfe_sqr_n(x11, 11, k), x11, k);
  const uint32_t x44 = wfe_mul(wfe_sqr_n(x22, 22, k), x22, k);
  const uint32_t x88 = wfe_mul(wfe_sqr_n(x44, 44, k), x44, k);
  const uint32_t x176 = wfe_mul(wfe_sqr_n(x88, 88, k), x88, k);
  const uint32_t x220 = wfe_mul(wfe_sqr_n(x176, 44, k), x44, k);
  const uint32_t x223 = wfe_mul(wfe_sqr_n(x220, 3, k), x3, k);
  uint32_t t = wfe_mul(wfe_sqr_n(x223, 23, k), x22, k);
  t = wfe_mul(wfe_sqr_n(t, 6, k), x2, k);
  return wfe_sqr_n(t, 2, k);
}

// The same chain as ONE doubly nested loop around a single pasted multiplication: 14 segments of
// "n squarings, then × a saved power".  The row-per-signature recover has no spare row to hide the chain on, so it pays
// for every instruction of it: no call / return / argument moves per step (≈15 of ≈90 instructions with the outlined
// multiply), and one 600-byte body instead of thirteen call sites.
//   INVSQRT = false: a^((p+1)/4), the square-root candidate (exponent bits 223 ones, 0, 22 ones, 00001100);
//   INVSQRT = true:  a^((p−3)/4) = a^((p+1)/4) / a (bits 223 ones, 0, 22 ones, 00001011) — for a square a this is
//                    ±1/√a, the one exponentiation that gives recover_pubkey_row BOTH √(x³ + 7) and the inverse of the
//                    final Z (see there).
template <bool INVSQRT>
WVF uint32_t wfe_pow_chain_rolled(uint32_t a, const wk &k) {
  //                                 x2 x3 x6 x9 x11 x22 x44 x88 x176 x220 x223  t   t   t
  const uint8_t NSQ_SQRT[14] = {1, 1, 3, 3, 2, 11, 22, 44, 88, 44, 3, 23, 6, 1};   // last segment: two bare squarings
  const uint8_t NSQ_INV[14] = {1, 1, 3, 3, 2, 11, 22, 44, 88, 44, 3, 23, 5, 3};    // … ·2^5·a, then ·2^3·x2
  uint32_t cur = a, x2 = 0, x3 = 0, x11 = 0, x22 = 0, x44 = 0, x88 = 0;
#pragma unroll 1
  for (int seg = 0; seg < 14; seg++) {
    uint32_t opnd = a;                        // segments 0, 1 (and 12 of the inverse square root)
    opnd = (seg == 2 || seg == 3 || seg == 10) ? x3 : opnd;
    opnd = (seg == 4 || seg == (INVSQRT ? 13 : 12)) ? x2 : opnd;
    opnd = seg == 5 ? x11 : opnd;
    opnd = (seg == 6 || seg == 11) ? x22 : opnd;
    opnd = (seg == 7 || seg == 9) ? x44 : opnd;
    opnd = seg == 8 ? x88 : opnd;
    const int nsq = INVSQRT ? NSQ_INV[seg] : NSQ_SQRT[seg];
#pragma unroll 1
    for (int i = 0; i <= nsq; i++) {
      const uint32_t b = (i < nsq || (!INVSQRT && seg == 13)) ? cur : opnd;  // wave-uniform choice
      cur = wfe_mul<true>(cur, b, k);
    }
    x2 = seg == 0 ? cur : x2;
    x3 = seg == 1 ? cur : x3;
    x11 = seg == 4 ? cur : x11;
    x22 = seg == 5 ? cur : x22;
    x44 = seg == 6 ? cur : x44;
    x88 = seg == 7 ? cur : x88;
  }
  return cur;
}
WVF uint32_t wfe_sqrt_candidate_rolled(uint32_t a, const wk &k) { return wfe_pow_chain_rolled<false>(a, k); }

// ---- √ on a spare row ------------------------------------------------------------------------------
// The a^((p+1)/4) chain (secp::fe_sqrt_candidate) is 266 dependent multiplications — but it needs
// only ONE row.  While rows 0..2 run the 64 prefix doublings (three wfe_mul each, below), row 3
// feeds its own operands into the very same wfe_mul calls: chain step t = 3·i + j rides on call j
// of doubling i.  Every step squares the running value except
//   step   1    3    7   11   14   26   49   94  183 | 228  232  256  263      (266 steps in all)
//   (i,j) 0,1  1,0  2,1  3,2  4,2  8,2 16,1 31,1 61,0 | after the loop (sqrt_side_tail)
//   by     a    a   x3   x3   x2  x11  x22  x44  x88 | x44   x3  x22   x2
//   save  x2   x3    –    –  x11  x22  x44  x88    – |   –    –    –    –
// so only the doublings i ∈ {0,1,2,3,4,8,16,31,61} (SPECIAL) carry the operand selects.
struct sqrt_side {
  uint32_t cur, a, x2, x3, x11, x22, x44, x88;  // row-3 values (other rows: don't care)
};
HD sqrt_side sqrt_side_init(uint32_t a) { return sqrt_side{a, a, 0u, 0u, 0u, 0u, 0u, 0u}; }
HD bool sqrt_side_special(int i) { return ((0x200000008001011Full >> i) & 1ull) != 0; }  // bits 0,1,2,3,4,8,16,31,61

// one wfe_mul shared by the doubling (rows 0..2: a·b) and the chain (row 3); i is wave-uniform
template <int J, bool SPECIAL>
WVF uint32_t mul_with_side(uint32_t a, uint32_t b, sqrt_side &sd, int i, const wk &k) {
  uint32_t so = sd.cur;
  if (SPECIAL) {
    if (J == 0) {
      so = i == 1 ? sd.a : so;
      so = i == 61 ? sd.x88 : so;
    } else if (J == 1) {
      so = i == 0 ? sd.a : so;
      so = i == 2 ? sd.x3 : so;
      so = i == 16 ? sd.x22 : so;
      so = i == 31 ? sd.x44 : so;
    } else {
      so = i == 3 ? sd.x3 : so;
      so = i == 4 ? sd.x2 : so;
      so = i == 8 ? sd.x11 : so;
    }
  }
  const bool side = k.row == 3;
  const uint32_t m = wfe_mul<true>(side ? sd.cur : a, side ? so : b, k);
  sd.cur = m;
  if (SPECIAL) {
    if (J == 0) {
      sd.x3 = i == 1 ? m : sd.x3;
    } else if (J == 1) {
      sd.x2 = i == 0 ? m : sd.x2;
      sd.x44 = i == 16 ? m : sd.x44;
      sd.x88 = i == 31 ? m : sd.x88;
    } else {
      sd.x11 = i == 4 ? m : sd.x11;
      sd.x22 = i == 8 ? m : sd.x22;
    }
  }
  return m;
}
// steps 192..265 (after 64 doublings): 36 squarings, ·x44, 3, ·x3, 23, ·x22, 6, ·x2, 2
WVF uint32_t sqrt_side_tail(const sqrt_side &sd, const wk &k) {
  uint32_t t = wfe_mul(wfe_sqr_n(sd.cur, 36, k), sd.x44, k);
  t = wfe_mul(wfe_sqr_n(t, 3, k), sd.x3, k);
  t = wfe_mul(wfe_sqr_n(t, 23, k), sd.x22, k);
  t = wfe_mul(wfe_sqr_n(t, 6, k), sd.x2, k);
  return wfe_sqr_n(t, 2, k);
}

// ---- prefix doublings with the three rows working on ONE point ----------------------------------------
// dbl-2009-l has three dependent levels of multiplications: {X², Y², 2Y·Z} → {B², (X+B)², (3A)²} →
// {E·(D − X3)}.  Rows 0..2 take one product each, row swaps move the results between rows (five swaps a doubling;
// ds_bpermute, two LDS round trips a doubling, is what this replaced): three wfe_mul per doubling instead of seven.  In: X, Y valid in rows 0..2 (replicated), Z valid in row 2.
template <bool SPECIAL>
WVF void prefix_dbl(uint32_t &X, uint32_t &Y, uint32_t &Z, sqrt_side &sd, int i, const wk &k) {
  const bool r0 = k.row == 0, r1 = k.row == 1, r2 = k.row == 2;
  // level 1: row 0: A = X², row 1: B = Y², row 2: Z3 = 2Y·Z
  const uint32_t m1 = mul_with_side<0, SPECIAL>(r0 ? X : (r2 ? 2u * Y : Y), r0 ? X : (r2 ? Z : Y), sd, i, k);
  // rows 0, 1 ← B (row 1), every row ← A (row 0): m1 = [A B Z3 ·] → [A A Z3 Z3], [B B · ·] → [A A A A]
  const u32x2 e1 = rows_swap16(m1, m1);
  const uint32_t A = rows_swap32(e1.a, e1.a).a;
  const uint32_t v1 = r2 ? A : e1.b;
  // level 2: row 0: C = B², row 1: (X + B)², row 2: F = (3A)²
  const uint32_t o2 = r1 ? X + v1 : (r2 ? 3u * v1 : v1);  // magnitudes 1, 2, 3
  const uint32_t m2 = mul_with_side<1, SPECIAL>(o2, o2, sd, i, k);
  // m2 = [C T F ·] → [C C F F], [T T · ·] → [C C C C], [F F F F] and [T T T T]
  const u32x2 e2 = rows_swap16(m2, m2);
  const u32x2 cf = rows_swap32(e2.a, e2.a);
  const uint32_t C = cf.a, F = cf.b;
  const uint32_t T = rows_swap32(e2.b, e2.b).a;
  const uint32_t t = T + wfe_neg1(A, k) + wfe_neg1(C, k);    // 5
  const uint32_t D = wfe_weak(2u * t, k);                    // 10 → 1
  const uint32_t X3 = wfe_weak(F + wfe_neg2(2u * D, k), k);  // 4 → 1
  // level 3 (every row): Y3 = E·(D − X3) − 8C
  const uint32_t m3 = mul_with_side<2, SPECIAL>(3u * A, D + wfe_neg1(X3, k), sd, i, k);
  X = X3;
  Y = wfe_weak(m3 + wfe_neg8(8u * C, k), k);
  Z = m1;  // row 2
}
// (X, Y, 1) → 2^ndbl·(X, Y, 1) in every row, and w^((p+1)/4) in every row — the latter only for
// ndbl = 64, the schedule the chain is laid out on (tests use smaller ndbl to check the doublings)
WVF void prefix_and_sqrt(uint32_t &X, uint32_t &Y, uint32_t &Z, uint32_t &root, uint32_t w, int ndbl, const wk &k) {
  sqrt_side sd = sqrt_side_init(w);
  Z = k.li == 0 ? 1u : 0u;
#pragma unroll 1
  for (int i = 0; i < ndbl; i++) {
    if (sqrt_side_special(i))
      prefix_dbl<true>(X, Y, Z, sd, i, k);
    else
      prefix_dbl<false>(X, Y, Z, sd, i, k);
  }
  const uint32_t rt = sqrt_side_tail(sd, k);
  X = lane_perm(X, k.li);
  Y = lane_perm(Y, k.li);
  Z = lane_perm(Z, 32u | k.li);
  root = lane_perm(rt, 48u | k.li);
}

// ---- modular inversion with the limbs of d, e, f, g spread over lanes ---------------------------------
// safegcd as in modinv_dev.h, for ONE value per wavefront.  The 30 divsteps of a batch look only at
// the low words and stay scalar-like (every lane computes the same 2×2 matrix, variable-time form);
// but applying the matrix to the four 9-limb numbers — two thirds of the lane-layout cost — becomes
// four/six v_mad_i64_i32 per lane: lane i holds limb i (radix 2^30, signed) of d, e, f and g.
// Carries are not rippled: column c_i = lo30 + 2^30·hi gives limb_j = lo_{j+1} + hi_j, one more
// parallel pass leaves limbs in [−4, 2^30 + 4) (top limb unmasked) — bounded, not canonical, which is
// all the next batch needs (only f, g mod 2^30 and the sign of the top limbs of d, e are read).
// The 30 divsteps of a batch: secp::divsteps_30_lockstep (modinv_dev.h) — written for lanes / rows that hold
// DIFFERENT values and advance in lockstep, which is what the row-per-signature recover needs.
WVF int32_t divsteps_30_rows(int32_t zeta, uint32_t f0, uint32_t g0, secp::trans2x2 &t) {
  return secp::divsteps_30_lockstep(zeta, f0, g0, t);
}

template <int WITH_MOD>
WVF int32_t modinv_wave_apply(int32_t m1, int32_t a, int32_t m2, int32_t b, int32_t mod_limb, int32_t mm, uint32_t li) {
  int64_t c = (int64_t)m1 * a + (int64_t)m2 * b;
  if (WITH_MOD) c += (int64_t)mod_limb * mm;
  const uint32_t lo = (uint32_t)c & (uint32_t)secp::M30;
  const int64_t n = (int64_t)row_shl<1>(lo) + (c >> 30);  // limb j = lo_{j+1} + hi_j  (exact: lo_0 = 0)
  const bool top = li >= 8;
  const int32_t carry = top ? 0 : (int32_t)(n >> 30);
  const int32_t keep = top ? (int32_t)n : (int32_t)((uint32_t)n & (uint32_t)secp::M30);
  return keep + (int32_t)row_shr<1>((uint32_t)carry);
}
// One body for both moduli (is_p: the field prime, else the group order) so that the kernels carry ONE copy
// of this code: the one-wavefront kernels sit right at the 64 KB of the instruction cache.
WVF u256 modinv_wave_body(const u256 &x, bool is_p, uint32_t li) {
  const secp::s30 xs = secp::s30_from_u256(x);
  int32_t mod[9];
#pragma unroll
  for (int i = 0; i < 9; i++) mod[i] = is_p ? secp::ModP::limb(i) : secp::ModN::limb(i);
  const uint32_t inv30 = is_p ? secp::ModP::inv30() : secp::ModN::inv30();
  int32_t f = 0, g = 0, d = 0, e = li == 0 ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    f = li == (uint32_t)i ? mod[i] : f;
    g = li == (uint32_t)i ? xs.v[i] : g;
  }
  const int32_t mod_limb = f;
  int32_t zeta = -1;
#pragma unroll 1
  for (int b = 0; b < 20; b++) {
    secp::trans2x2 t;
    zeta = divsteps_30_rows(zeta, row_bcast<0>((uint32_t)f), row_bcast<0>((uint32_t)g), t);
    // (d, e) ← t·(d, e)/2^30 mod M: the multiple of M that makes it divisible (modinv_dev.h:update_de_30)
    const int32_t d0 = (int32_t)row_bcast<0>((uint32_t)d), e0 = (int32_t)row_bcast<0>((uint32_t)e);
    const int32_t sd = (int32_t)row_bcast<8>((uint32_t)d) >> 31, se = (int32_t)row_bcast<8>((uint32_t)e) >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    const uint32_t cd0 = (uint32_t)t.u * (uint32_t)d0 + (uint32_t)t.v * (uint32_t)e0;
    const uint32_t ce0 = (uint32_t)t.q * (uint32_t)d0 + (uint32_t)t.r * (uint32_t)e0;
    md -= (int32_t)((inv30 * cd0 + (uint32_t)md) & (uint32_t)secp::M30);
    me -= (int32_t)((inv30 * ce0 + (uint32_t)me) & (uint32_t)secp::M30);
    const int32_t nd = modinv_wave_apply<1>(t.u, d, t.v, e, mod_limb, md, li);
    const int32_t ne = modinv_wave_apply<1>(t.q, d, t.r, e, mod_limb, me, li);
    const int32_t nf = modinv_wave_apply<0>(t.u, f, t.v, g, 0, 0, li);
    const int32_t ng = modinv_wave_apply<0>(t.q, f, t.r, g, 0, 0, li);
    d = nd;
    e = ne;
    f = nf;
    g = ng;
    if (!any(g != 0)) break;
  }
  // collect d (and the sign of f = ±1: −1 ≡ 3 mod 4 whatever the limb form), ripple the carries once
  secp::s30 D;
  D.v[0] = (int32_t)row_bcast<0>((uint32_t)d);
  D.v[1] = (int32_t)row_bcast<1>((uint32_t)d);
  D.v[2] = (int32_t)row_bcast<2>((uint32_t)d);
  D.v[3] = (int32_t)row_bcast<3>((uint32_t)d);
  D.v[4] = (int32_t)row_bcast<4>((uint32_t)d);
  D.v[5] = (int32_t)row_bcast<5>((uint32_t)d);
  D.v[6] = (int32_t)row_bcast<6>((uint32_t)d);
  D.v[7] = (int32_t)row_bcast<7>((uint32_t)d);
  D.v[8] = (int32_t)row_bcast<8>((uint32_t)d);
  const bool fneg = (row_bcast<0>((uint32_t)f) & 3u) == 3u;
  // D → [0, M), negated first when f = −1 (secp::normalize_30 with the modulus at run time).
  // Range of D: (−2M − ε, M + ε), NOT (−2M, M).  The batches decide "d < 0" from the top limb of a carry-save form
  // (limbs in [−4, 2^30 + 4)), which calls a value in [0, 4·2^210) negative when its top limb is −1 under a limb 7 of
  // 2^30 + c; such a d gets one M too many.  A row that is done (g = 0) while other rows of the wavefront still work
  // goes through the remaining batches with the matrix (2^30, 0; 0, 1) and keeps exactly that surplus: 1/1 came out
  // as M + 1 (tests/test_dev_wave_host.py::test_modular_inverse_of_four_different_values_in_lockstep).  So: add M
  // while negative (twice after the negation), then take M off once if the value reached it.
  auto ripple = [&]() {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      D.v[i + 1] += D.v[i] >> 30;
      D.v[i] &= secp::M30;
    }
  };
  auto add_m_if_negative = [&]() {
    const int32_t neg = D.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) D.v[i] += mod[i] & neg;
    ripple();
  };
  ripple();
  add_m_if_negative();  // (−M − ε, M + ε)
  const int32_t nm = fneg ? -1 : 0;
#pragma unroll
  for (int i = 0; i < 9; i++) D.v[i] = (D.v[i] ^ nm) - nm;
  ripple();
  add_m_if_negative();  // (−ε, M + ε)
  add_m_if_negative();  // [0, M + ε)
  {
    secp::s30 E = D;
#pragma unroll
    for (int i = 0; i < 9; i++) E.v[i] -= mod[i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      E.v[i + 1] += E.v[i] >> 30;
      E.v[i] &= secp::M30;
    }
    const int32_t keep = E.v[8] >> 31;  // D − M < 0: D was below M already
#pragma unroll
    for (int i = 0; i < 9; i++) D.v[i] = (D.v[i] & keep) | (E.v[i] & ~keep);
  }
  return secp::s30_to_u256(D);
}
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __attribute__((noinline)) u256 modinv_wave_fn(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3,
                                                               uint32_t x4, uint32_t x5, uint32_t x6, uint32_t x7,
                                                               uint32_t is_p, uint32_t li) {
  u256 x;
  x.v[0] = x0; x.v[1] = x1; x.v[2] = x2; x.v[3] = x3;
  x.v[4] = x4; x.v[5] = x5; x.v[6] = x6; x.v[7] = x7;
  return modinv_wave_body(x, is_p != 0, li);
}
#endif
template <class MOD> struct wave_mod_is_p;
template <> struct wave_mod_is_p<secp::ModP> { static constexpr uint32_t value = 1; };
template <> struct wave_mod_is_p<secp::ModN> { static constexpr uint32_t value = 0; };
template <class MOD>
WVF u256 modinv_wave(const u256 &x, const wk &k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return modinv_wave_fn(x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7], wave_mod_is_p<MOD>::value, k.li);
#else
  return modinv_wave_body(x, wave_mod_is_p<MOD>::value != 0, k.li);
#endif
}
// Jacobian (lane layout, the same in every lane) → affine; r.x / r.y canonical; false for infinity
WVF bool jac_to_aff_wave(aff &r, const jac &p, const wk &k) {
  const fe zi = secp::fe_from_u256(modinv_wave<secp::ModP>(secp::fe_to_u256(p.z), k));
  const fe zi2 = secp::fe_sqr(zi);
  r.x = secp::fe_normalize(secp::fe_mul(p.x, zi2));
  r.y = secp::fe_normalize(secp::fe_mul(p.y, secp::fe_mul(zi2, zi)));
  return !(p.inf || secp::fe_is_zero(p.z));
}

// The same from the row layout, multiplications included: Z⁻¹ goes back into the row and X·Z⁻², Y·Z⁻³ are four
// wfe_mul (≈90 instructions each, and code that is hot anyway) instead of four lane-layout multiplications
// (≈224 each, fetched for this one use).
WVF bool wjac_to_aff(aff &r, const wjac &p, const wk &k) {
  const fe zf = gather(p.z);
  const uint32_t zi = scatter(secp::fe_from_u256(modinv_wave<secp::ModP>(secp::fe_to_u256(zf), k)), k);
  const uint32_t zi2 = wfe_sqr(zi, k);
  r.x = secp::fe_normalize(gather(wfe_mul(p.x, zi2, k)));
  r.y = secp::fe_normalize(gather(wfe_mul(p.y, wfe_mul(zi2, zi, k), k)));
  return !(p.inf || secp::fe_is_zero(zf));
}

// u1·G summed into acc: the GTAB_WINDOWS fixed-base windows dealt to the four rows (rows hold partial sums: the caller joins)
// Table point t of this row's share (window row·WPR + t) — the load only; gen_windows_wave consumes it.
WVF waff gen_window_point(const uint32_t *__restrict__ gtab, const u256 &u1, int t, const wk &k) {
  constexpr int WPR = ibftk::GTAB_WINDOWS / 4;
  const int win = (int)k.row * WPR + t;
  const int bit = win * ibftk::GTAB_BITS;
  const uint32_t dgt = (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
  const uint32_t *e = gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)win * ibftk::GTAB_ENTRIES + dgt);
  const uint32_t ld = k.li < 10 ? k.li : 0u;
  waff pt;
  pt.x = e[ld] & k.act;
  pt.y = e[10 + ld] & k.act;
  return pt;
}
// Round 6: software-pipelined by one — the point of window t + 1 is asked for before the addition of window t runs, and the
// FIRST point is handed in by the caller, who asked for it as soon as u1 was known (in front of the main loop): a dependent
// read of the 84 MB table takes ≈ 1.8 µs with one resident wavefront per SIMD, the same order as the addition it used to
// wait in front of (four times per signature on the latency-bound path of the small batches).
WVF wjac gen_windows_wave(const uint32_t *__restrict__ gtab, const u256 &u1, wjac acc, waff first, const wk &k) {
  constexpr int WPR = ibftk::GTAB_WINDOWS / 4;
  waff cur = first;
#pragma unroll 1
  for (int t = 0; t < WPR; t++) {
    const waff nxt = gen_window_point(gtab, u1, t + 1 < WPR ? t + 1 : t, k);  // (the last iteration re-reads its own point)
    const int bit = ((int)k.row * WPR + t) * ibftk::GTAB_BITS;
    const uint32_t dgt = (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    const wjac sum = wjac_add_aff(acc, cur, k);  // (called multiply: four iterations do not pay for 6 KB of code)
    acc = wjac_select(dgt != 0, sum, acc);
    cur = nxt;
  }
  return acc;
}
WVF wjac gen_windows_wave(const uint32_t *__restrict__ gtab, const u256 &u1, wjac acc, const wk &k) {
  return gen_windows_wave(gtab, u1, acc, gen_window_point(gtab, u1, 0, k), k);
}
#ifndef IBFT_WAVE_COMMON_Z
#define IBFT_WAVE_COMMON_Z 1  // 1: the one-wavefront recover brings each row's table to one common Z (mixed additions); 0: A/B
#endif
// ---- the recover, one signature per wavefront -------------------------------------------------------
// Same contract and rejection list as ibftk::recover_pubkey (recover_dev.h); every lane of the
// wavefront passes the same (z, r, s, v) and gets the same answer.
//
// Row ρ computes the piece (half = ρ & 1, upper = ρ >> 1) of u2·R = k1·(±R) + k2·λ(±R): 64 bits of
// the 128-bit |k_half| in signed radix-16 digits on the base 2^(64·upper)·(β^half·x, ±y), then
// its share of the u1·G window points; two row-xor additions join the four rows.
//
// y = √(x³ + 7) is NOT on the critical path: with w = x³ + 7 the Jacobian triple (w·x, w², y)
// is the point (x, y), and the a = 0 formulas never look at the curve constant, so everything up to
// the G additions runs on (w·x, w², 1) — the isomorphic curve y² = x³ + 7w³ — and the accumulator's Z
// is multiplied by y once the chain (computed by row 3 during the prefix doublings) has delivered it.
// STOP < 99 cuts the function short after a stage (devtest timing breakdown only; addr then holds junk).
//
// TWO WAVEFRONTS PER SIGNATURE (round 4, n ≤ 512 — at most half of the chip's SIMDs would otherwise work): PAIR = true is
// the MAIN wavefront of a pair.  Everything on its critical path stays (√ riding on the prefix doublings, table, the 64
// doublings of the main loop, joins, Z⁻¹, Keccak); what does not depend on the curve work — r⁻¹ mod n, u₁, u₂, the GLV split,
// and the sixteen fixed-base additions of u₁·G with their two joins — is done by the HELPER wavefront (recover_helper_wave)
// meanwhile and handed over through `sh` (LDS on the device) at two workgroup barriers (`sync`): the split scalars before the
// table is built, the point u₁·G before the last addition.  Same verdicts: point addition is associative, and the rare cases
// (∞, equal or opposite operands) are handled by wjac_add wherever they fall.
struct pair_shared {
  uint32_t k1[4], k2[4];            // |k1|, |k2| of the GLV split of u₂ (128 bits each)
  uint32_t neg;                     // bit 0: k1 negative, bit 1: k2 negative
  uint32_t ginf;                    // u₁·G is the point at infinity (u₁ = 0)
  uint32_t gx[16], gy[16], gz[16];  // u₁·G, Jacobian, limb i in slot i (slots 10…15 zero: the row layout's idle lanes)
};
struct no_sync {
  HD void operator()() const {}
};
template <int STOP = 99, bool PAIR = false, class SYNC = no_sync>
WVF bool recover_pubkey_wave(const uint32_t *__restrict__ gtab, const u256 &z_raw, const u256 &r, const u256 &s,
                            uint32_t v, uint32_t flags, uint32_t addr[5], aff &Qa, const pair_shared *sh = nullptr,
                            SYNC sync = SYNC()) {
#define WV_STAGE(n, keep)     \
  if (STOP == (n)) {          \
    addr[0] = (keep);         \
    return ok;                \
  }
  const wk k = wk_init();
  bool ok = ibftk::sig_in_range(r, s, v, flags);
  const fe rx = secp::fe_from_u256(r);
  const uint32_t x = scatter(rx, k);
  const uint32_t one = k.li == 0 ? 1u : 0u;
  const uint32_t w = wfe_weak(wfe_mul(wfe_sqr(x, k), x, k) + (k.li == 0 ? 7u : 0u), k);  // magnitude 1
  const uint32_t X0 = wfe_mul(w, x, k), Y0 = wfe_sqr(w, k);
  // prefix: 2^64·(X0, Y0, 1) on rows 0..2, √w on row 3
  uint32_t PX = X0, PY = Y0, PZ, yc;
  prefix_and_sqrt(PX, PY, PZ, yc, w, 64, k);
  WV_STAGE(1, PX ^ PY ^ PZ ^ yc)
  // u1 = −z/r, u2 = s/r (mod n); u2 = k1 + k2·λ — computed here, or (PAIR) by the helper wavefront while the prefix ran
  u256 u1 = secp::zero256();
  secp::glv_split sp;
  waff gfirst = waff{0u, 0u};  // this row's first table point of u1·G (not PAIR)
  if constexpr (PAIR) {
    sync();  // barrier 1: the helper has written the split scalars
    sp.k1 = sp.k2 = secp::zero256();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      sp.k1.v[i] = sh->k1[i];
      sp.k2.v[i] = sh->k2[i];
    }
    sp.neg1 = (sh->neg & 1u) != 0;
    sp.neg2 = (sh->neg & 2u) != 0;
  } else {
    const secp::sc rinv = secp::sc_from_u256(modinv_wave<secp::ModN>(r, k));
    u1 = secp::sc_neg_canon(secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), rinv)));
    const u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(s), rinv));
    gfirst = gen_window_point(gtab, u1, 0, k);  // on its way while the split, the table and the main loop run
    sp = secp::sc_split_lambda(u2);
  }
  WV_STAGE(2, PX ^ PY ^ PZ ^ yc ^ u1.v[0] ^ sp.k1.v[0] ^ sp.k2.v[1])
  const bool half = (k.row & 1u) != 0, upper = (k.row & 2u) != 0;
  // signed radix-16 digits of this row's |k|: k + 0x88…8 has nibbles d_j + 8, bit 128 is the top digit
  uint32_t dw[5];
  {
    const u256 &kk = half ? sp.k2 : sp.k1;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) dw[i] = secp::addc(kk.v[i], 0x88888888u, c);
    dw[4] = c;
  }
  const uint32_t d_lo = upper ? dw[2] : dw[0], d_hi = upper ? dw[3] : dw[1];
  const bool top = upper && dw[4] != 0;
  // this row's base: (β^half · X, ±Y, Z) of the point or of its 2^64-multiple
  wjac base;
  const uint32_t beta = scatter(secp::GLV_CONST(1), k);
  base.x = wfe_mul(upper ? PX : X0, half ? beta : one, k);
  base.y = upper ? PY : Y0;
  {
    const uint32_t yn = wfe_weak(wfe_neg1(base.y, k), k);  // cross-lane ops: computed by every row, then selected
    base.y = (half ? sp.neg2 : sp.neg1) ? yn : base.y;
  }
  base.z = upper ? PZ : one;
  base.inf = false;
  // table 1..8 of the base
  wjac T[9];
  T[1] = base;
  T[2] = wjac_dbl(T[1], k);
  T[3] = wjac_add(T[2], T[1], k);
  T[4] = wjac_dbl(T[2], k);
  T[5] = wjac_add(T[4], T[1], k);
  T[6] = wjac_dbl(T[3], k);
  T[7] = wjac_add(T[6], T[1], k);
  T[8] = wjac_dbl(T[4], k);
#if IBFT_WAVE_COMMON_Z
  // ONE common Z for the row's table, as in the row-per-signature recover: with Zc = z1·…·z8 and s_i = Zc / z_i entry i
  // is the affine point (x_i·s_i², y_i·s_i³) of an isomorphic curve, the sixteen additions of the main loop are MIXED
  // (10.5 multiplications instead of 16) and Zc goes into the accumulator's Z once behind the loop: 51 multiplications
  // spent, 88 saved per row.  pre[i] = z1·…·z_i forward, suf = z_{i+1}·…·z8 while walking down.
  uint32_t AX[9], AY[9], Zc;
  {
    uint32_t pre[8];
    pre[1] = T[1].z;
#pragma unroll
    for (int i = 2; i <= 7; i++) pre[i] = wfe_mul(pre[i - 1], T[i].z, k);
    uint32_t suf = one;
#pragma unroll
    for (int i = 8; i >= 1; i--) {
      const uint32_t sc = i == 8 ? pre[7] : (i == 1 ? suf : wfe_mul(pre[i - 1], suf, k));
      const uint32_t s2 = wfe_sqr(sc, k);
      AX[i] = wfe_mul(T[i].x, s2, k);
      AY[i] = wfe_mul(T[i].y, wfe_mul(s2, sc, k), k);
      suf = i == 8 ? T[8].z : wfe_mul(suf, T[i].z, k);
    }
    Zc = suf;
  }
  WV_STAGE(3, AX[3] ^ AY[5] ^ AX[6] ^ AX[7] ^ AY[8] ^ Zc ^ yc ^ u1.v[0])
  wjac acc = wjac_select(top, wjac_from_aff(waff{AX[1], AY[1]}, k), wjac_inf());
#pragma unroll 1
  for (int jd = 15; jd >= 0; jd--) {
#pragma unroll 1
    for (int d = 0; d < 4; d++) acc = wjac_dbl<true>(acc, k);
    const uint32_t word = jd >= 8 ? d_hi : d_lo;
    const int dg = (int)((word >> (4 * (jd & 7))) & 15u) - 8;
    const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg);
    waff q = waff{AX[1], AY[1]};
#pragma unroll
    for (int e = 2; e <= 8; e++) {
      q.x = mag == (uint32_t)e ? AX[e] : q.x;
      q.y = mag == (uint32_t)e ? AY[e] : q.y;
    }
    q.y = dg < 0 ? wfe_neg1(q.y, k) : q.y;  // magnitude ≤ 2
    const wjac sum = wjac_add_aff<true>(acc, q, k);
    acc = wjac_select(mag != 0, sum, acc);
  }
  acc.z = wfe_mul(acc.z, Zc, k);  // back from the table's curve (an accumulator at infinity keeps its flag)
#else
  WV_STAGE(3, T[3].x ^ T[5].y ^ T[6].z ^ T[7].x ^ T[8].y ^ yc ^ u1.v[0])
  wjac acc = wjac_select(top, T[1], wjac_inf());
#pragma unroll 1
  for (int jd = 15; jd >= 0; jd--) {
#pragma unroll 1
    for (int d = 0; d < 4; d++) acc = wjac_dbl<true>(acc, k);
    const uint32_t word = jd >= 8 ? d_hi : d_lo;
    const int dg = (int)((word >> (4 * (jd & 7))) & 15u) - 8;
    const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg);
    wjac q = T[1];
#pragma unroll
    for (int e = 2; e <= 8; e++) q = wjac_select(mag == (uint32_t)e, T[e], q);
    q.y = dg < 0 ? wfe_neg1(q.y, k) : q.y;  // magnitude ≤ 2
    const wjac sum = wjac_add<true>(acc, q, k);
    acc = wjac_select(mag != 0, sum, acc);
  }
#endif
  WV_STAGE(4, acc.x ^ acc.y ^ acc.z ^ yc ^ u1.v[0])
  // back to the real curve: y² = w ?  parity(y) = v; Z ← Z·y
  ok = ok && wfe_is_zero(wfe_sqr(yc, k) + wfe_neg1(w, k));
  fe y = secp::fe_normalize(gather(yc));
  y = secp::l26_select((y.n[0] & 1u) != v, secp::fe_normalize_weak(secp::fe_neg(y, 1)), y);
  acc.z = wfe_mul(acc.z, scatter(y, k), k);
  // u1·G: the fixed-base windows are dealt to the rows (PAIR: the helper wavefront has summed them meanwhile)
  if constexpr (!PAIR) acc = gen_windows_wave(gtab, u1, acc, gfirst, k);
  acc = wjac_add(acc, wjac_lane_xor(acc, 16), k);
  acc = wjac_add(acc, wjac_lane_xor(acc, 32), k);
  if constexpr (PAIR) {
    sync();  // barrier 2: u₁·G is in `sh`
    wjac g;
    g.x = sh->gx[k.li];
    g.y = sh->gy[k.li];
    g.z = sh->gz[k.li];
    g.inf = sh->ginf != 0;
    acc = wjac_add(acc, g, k);
  }
  WV_STAGE(5, acc.x ^ acc.y ^ acc.z)
  const jac Q = wjac_gather(acc);
  ok = jac_to_aff_wave(Qa, Q, k) && ok;  // every row holds the same point after the joins
  WV_STAGE(6, Qa.x.n[0] ^ Qa.y.n[1])
  u256 qx = secp::l26_to_u256(Qa.x), qy = secp::l26_to_u256(Qa.y);
  keccak::address_from_xy(qx.v, qy.v, addr);
  return ok;
#undef WV_STAGE
}

// The HELPER wavefront of a pair (see recover_pubkey_wave<…, PAIR = true>): r⁻¹ mod n, u₁ = −z/r, u₂ = s/r, the GLV split of
// u₂ → `sh`, barrier 1; then u₁·G (sixteen table additions over the four rows, two joins) → `sh`, barrier 2.
template <class SYNC>
WVF void recover_helper_wave(const uint32_t *__restrict__ gtab, const u256 &z_raw, const u256 &r, const u256 &s, pair_shared *sh,
                             SYNC sync) {
  const wk k = wk_init();
  const secp::sc rinv = secp::sc_from_u256(modinv_wave<secp::ModN>(r, k));
  const u256 u1 = secp::sc_neg_canon(secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), rinv)));
  const u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(s), rinv));
  const secp::glv_split sp = secp::sc_split_lambda(u2);
  if (lane_id() == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      sh->k1[i] = sp.k1.v[i];
      sh->k2[i] = sp.k2.v[i];
    }
    sh->neg = (sp.neg1 ? 1u : 0u) | (sp.neg2 ? 2u : 0u);
  }
  sync();  // barrier 1
  wjac acc = gen_windows_wave(gtab, u1, wjac_inf(), k);
  acc = wjac_add(acc, wjac_lane_xor(acc, 16), k);
  acc = wjac_add(acc, wjac_lane_xor(acc, 32), k);  // every row holds u₁·G now
  if (lane_id() < 16) {
    sh->gx[k.li] = acc.x;
    sh->gy[k.li] = acc.y;
    sh->gz[k.li] = acc.z;
  }
  if (lane_id() == 0) sh->ginf = acc.inf ? 1u : 0u;
  sync();  // barrier 2
}

// One dword per lane from global memory straight into LDS — no VGPR in between, nothing for the wavefront to wait for until it
// says so: on gfx950 `global_load_lds_dword` (LDS address = M0 + 4·lane, the global address is per lane); the load is asynchronous
// and counted by vmcnt.  Written as an asm statement on purpose: hipcc does not count it, so it puts NO wait in front of later
// LDS reads (it would wait vmcnt(0) at the very next one if it knew) — the reader says lds_prefetch_wait() before it reads
// what was prefetched.  dst64: a wave-uniform pointer to 64 dwords of LDS (one per lane).  The emulator copies.
WVF void lds_prefetch_dword(uint32_t *dst64, const uint32_t *src) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32 *)dst64);
  uint32_t keep;  // (M0 is the compiler's: saved and restored inside the statement that borrows it)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(m0v)
               : "memory");
#else
  dst64[lane_id()] = *src;
#endif
}
WVF void lds_prefetch_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

WVF waff load_waff(const uint32_t *__restrict__ e20, const wk &k) {
  const uint32_t ld = k.li < 10 ? k.li : 0u;
  waff pt;
  pt.x = e20[ld] & k.act;
  pt.y = e20[10 + ld] & k.act;
  return pt;
}
// ---- the end of the row-per-signature recover: ONE exponentiation for √(x³ + 7) AND the final inversion ------------
// The rows run u2·R without knowing R's y: with t = x³ + 7 the point R′ = (x·t, t²) lies on y² = x³ + 7t³, which is the
// curve itself seen through (x, y) → (x·u², y·u³), u = y_R = √t, and the a = 0 formulas never look at the constant.  A
// Jacobian result (X′, Y′, Z′) there is the point (X′, Y′, Z′·u) here.  Its sum with P1 = u1·G (kept in an accumulator of
// its own) can be written down with u as a symbol, because u² = t is known:
//     U1 = X1·Z′²t   U2 = X′·Z1²   S2 = Y′·Z1³   S1 = ŝ·u, ŝ = Y1·Z′³t   H = U2 − U1   V = U1·H²
//     X3 = A + B·u    A = S2² + ŝ²t − H³ − 2V            B = −2·S2·ŝ
//     Y3 = C + D·u    C = S2·(V − A) + ŝ·B·t             D = −S2·B − ŝ·(V − A + H³)
//     Z3 = m·u        m = Z1·Z′·H
// and the affine point is x = (A + B·u)/(m²t), y = (C·u + D·t)/(m³t²).  With q = t·m² and I = q^((p−3)/4) = ±1/(u·m):
//     I² = 1/(m²t)        ±u = I·m·t        1/(m³t²) = I⁴·m
// — the square root and the inverse come out of the SAME 254-squaring chain; the sign of u is fixed by the recovery id
// exactly as before (y_R has parity v), and I enters only through I², I⁴ and I·(the sign that is being fixed).  What this
// replaces: a √ chain at the start AND a safegcd inversion of Z at the end (≈9 k issue slots) for ≈35 multiplications.
// t not a square (r is no x coordinate): (I·m·t)² = −t ≠ t, the row is rejected like before.
// Rare rows — P1 or the u2·R′ result at infinity, or H = 0 (u1·G = ±u2·R) — go through the same chain with other inputs
// (in the function).
WVF bool rows_finish_deferred(aff &Qa, const wjac &p1, const wjac &p2, uint32_t t, uint32_t v, const wk &k) {
  const uint32_t z2s = wfe_mul(wfe_sqr(p2.z, k), t, k);  // Z2² = Z′²·t
  const uint32_t z1s = wfe_sqr(p1.z, k);
  const uint32_t U1 = wfe_mul(p1.x, z2s, k), U2 = wfe_mul(p2.x, z1s, k);
  const uint32_t S2 = wfe_mul(p2.y, wfe_mul(z1s, p1.z, k), k);
  const uint32_t sh = wfe_mul(p1.y, wfe_mul(z2s, p2.z, k), k);  // ŝ
  const uint32_t H = U2 + wfe_neg1(U1, k);                      // 3
  const uint32_t HH = wfe_sqr(H, k);
  const uint32_t HHH = wfe_mul(HH, H, k);
  const uint32_t V = wfe_mul(U1, HH, k);
  // A = S2² + ŝ²·t − H³ − 2V: one reduction for the two products (1·1 + 1·2), then 1 + 2 + 3 → 1
  uint32_t A = wfe_weak(wfe_mul2(S2, S2, wfe_sqr(sh, k), t, k) + wfe_neg1(HHH, k) + wfe_neg2(2u * V, k), k);
  uint32_t Bn = wfe_mul(2u * S2, sh, k);                        // −B
  const uint32_t VA = V + wfe_neg1(A, k);                       // 3
  uint32_t C = wfe_mul2(S2, VA, wfe_neg1(wfe_mul(sh, t, k), k), Bn, k);   // 1·3 + 2·1
  uint32_t D = wfe_mul2(S2, Bn, wfe_neg1(sh, k), VA + HHH, k);            // 1·1 + 2·4
  uint32_t m = wfe_mul(wfe_mul(p1.z, p2.z, k), H, k);
  // H ≡ 0 ⇔ m ≡ 0 (finite points have Z ≢ 0).  The cheap filter says "maybe" for one row in 4 000 — once per launch of
  // 4 096 rows — so the exact test decides.
  const bool mz = wfe_z_maybe_zero(m);  // (cross-lane: every lane evaluates it)
  bool hz = false;
  if (any(mz)) hz = wfe_is_zero(m) && mz;
  // The rare rows ride the SAME exponentiation with other (A, B, C, D, m) — a launch lasts as long as its slowest wavefront,
  // and a validator can make any of these cases on purpose (z = ±s·k, z = 0), so none of them may cost a second chain
  // or an inversion (the first version's textbook route did: +20 % for the whole launch):
  //   P1 = ∞ (u1 = 0):           the sum is P2:      x = X′/(Z′²t), y = Y′·u/(Z′³t²)     →  (X′, 0, Y′, 0, Z′)
  //   H = 0 (P2 = ±P1):          tentatively 2·P1 =: S, a point of the curve itself: x = X_S/Z_S², y = Y_S/Z_S³
  //                                                                                     →  (X_S·t, 0, 0, Y_S·t, Z_S)
  //                              and P2 = +P1 ⇔ S2 = ŝ·u is checked once u is known; P2 = −P1: the key would be ∞, rejected
  //   P2′ = ∞ (cannot be the outcome of a valid row): S := P1, the same shape;   both ∞: rejected.
  bool chk = false, rej = false;
  if (any(p1.inf || p2.inf || hz)) {
    const wjac d1 = wjac_dbl(p1, k);
    const bool only1 = !p1.inf && (p2.inf || hz);  // the answer is a point S of the curve itself
    const uint32_t xs = p2.inf ? p1.x : d1.x, ys = p2.inf ? p1.y : d1.y, zs = p2.inf ? p1.z : d1.z;
    const uint32_t as = wfe_mul(xs, t, k), ds = wfe_mul(ys, t, k);
    A = only1 ? as : (p1.inf ? p2.x : A);
    Bn = (only1 || p1.inf) ? 0u : Bn;
    C = only1 ? 0u : (p1.inf ? p2.y : C);
    D = only1 ? ds : (p1.inf ? 0u : D);
    m = only1 ? zs : (p1.inf ? p2.z : m);
    chk = hz && !p1.inf && !p2.inf;
    rej = p1.inf && p2.inf;
  }
  const uint32_t I = wfe_pow_chain_rolled<true>(wfe_mul(wfe_sqr(m, k), t, k), k);
  const uint32_t I2 = wfe_sqr(I, k);
  const uint32_t ut = wfe_mul(I, wfe_mul(m, t, k), k);          // ±√t
  bool ok = wfe_is_zero(wfe_sqr(ut, k) + wfe_neg2(t, k));       // t is a square: (x, ·) is on the curve
  const fe un = secp::fe_normalize(gather(ut));
  const uint32_t u = ((un.n[0] & 1u) != v) ? wfe_neg1(ut, k) : ut;               // y_R: parity v (magnitude ≤ 2)
  if (any(chk)) {
    const bool same = wfe_is_zero(S2 + wfe_neg1(wfe_mul(sh, u, k), k));           // S1 = ŝ·u = S2: P2 = P1, the doubling stands
    ok = ok && (!chk || same);
  }
  const uint32_t xq = wfe_mul(A + wfe_neg1(wfe_mul(Bn, u, k), k), I2, k);         // (A + B·u)·I²
  const uint32_t yq = wfe_mul(wfe_mul2(C, u, D, t, k), wfe_mul(wfe_sqr(I2, k), m, k), k);  // (C·u + D·t)·I⁴·m
  Qa.x = secp::fe_normalize(gather(xq));
  Qa.y = secp::fe_normalize(gather(yq));
  return ok && !rej;
}

// ---- sixteen lanes per signature: every ROW of the wavefront recovers its own signature -------------------
// For batches between the one-wavefront form (n ≤ 2 048) and the point where eight lanes per signature
// fill the chip (n = 8 192): four signatures per wavefront, so n = 4 096 is again one wavefront per SIMD.
// Nothing is shared between rows, so there are no pieces, no prefix and no joins: a row runs the
// textbook interleaved GLV multiplication — 128 doublings shared by k1 and k2, two signed radix-16
// tables (the λ table is the first one with X·β), 33 digits each — then its sixteen G-table additions.
// Lane-layout values (r, s, z, the scalars, the digits) simply differ from row to row; the only wave-wide
// operations are the `any` votes, which merely make every row wait for the slowest one.
// STOP < 99 cuts the function short after a stage (devtest timing breakdown only; addr then holds junk).
// wtab: ROW_TAB_SLOTS × 64 dwords of wave-private scratch (LDS in the kernels): element (slot, lane) at
// wtab[slot·64 + lane]; a lane only ever touches its own column, so no barrier is involved.
#ifndef IBFT_ROWS_VARIANT
#define IBFT_ROWS_VARIANT 2  // 1: table in registers, straight-line build (kept for A/B timing); 2: table in LDS, rolled loops
#endif
#ifndef IBFT_ROWS_PEEL
#define IBFT_ROWS_PEEL 1  // 1: the first digit of the main loop and the first G window are choices, not additions (A/B: 0)
#endif
#ifndef IBFT_ROWS_DEFER_SQRT
#define IBFT_ROWS_DEFER_SQRT 1  // 1: √ and the final inversion from one exponentiation at the end (rows_finish_deferred); 0: √ first, safegcd last (A/B)
#endif
// IBFT_ROWS_G_PREFETCH = 1: the sixteen table points of u1·G go into wave-private LDS as soon as u1 is known (global → LDS
// directly, no register, no wait: lds_prefetch_dword) and are long there when the G additions come.  BUILT AND MEASURED in round
// 6, NOT ADOPTED: the dependent table reads in front of the fifteen additions are not what those additions wait for — with the
// prefetch the kernel is 0.9 % SLOWER on a resident batch (0.3329 → 0.3360 ms: 32 DMA instructions, 32 KB more LDS per
// workgroup) and 0.4 % slower on fresh batches whose entries come from beyond the L2 (0.3349 → 0.3362 ms), and the stage
// between the main loop and the closing exponentiation takes 0.038 ms as before (profiles/r06b_kernel_ab.txt,
// r06b_fresh_batch_ab.txt, r06b_rows_stage_ms.txt).  Kept behind the macro for the A/B.
#ifndef IBFT_ROWS_G_PREFETCH
#define IBFT_ROWS_G_PREFETCH 0
#endif
// IBFT_ROWS_G_MERGED = 1 (round 6): the fixed-base additions run as further iterations of the MAIN LOOP — two table points per
// iteration through the main loop's own two pasted mixed additions, no doublings — instead of a loop of their own around a third
// pasted copy.  What the per-stage counters showed (profiles/r06e_rows_stage_issue.txt): the G stage issues 10.2 k instructions in
// 36 µs, 3.5 ns each where the main loop's identical additions take 2.0 — and the excess does not move when the table points are
// prefetched (profiles/r06f_gpre_stage_ab.txt): it is the first walk through 5.5 KB of code the instruction cache has never
// seen, by every wavefront of the chip at once (so the reading went).
// — BUILT AND MEASURED, NOT ADOPTED: the fixed-base stage fell from 0.035 to 0.023 ms, the main loop rose from 0.205 to 0.216 ms
// (≈ 110 more instructions per iteration for the phase logic, and a worse schedule): N = 4 096 0.3338 → 0.3402 ms
// (profiles/r06g_kernel_ab.txt, r06g_rows_stage_ms.txt).  What replaced it: IBFT_ROWS_SHARED_ADDS.
// IBFT_ROWS_G_MERGED = 2 (with IBFT_ROWS_G_PREFETCH = 1): the uniform form — both phases read their operands from LDS by slot
// number; 0.3336 → 0.3415 ms (profiles/r06p_kernel_ab.txt), not adopted either.
#ifndef IBFT_ROWS_G_MERGED
#define IBFT_ROWS_G_MERGED 0
#endif
// IBFT_ROWS_SHARED_ADDS = 1: both loops keep their shape and CALL one outlined pair of pasted mixed additions (rows_two_adds,
// a leaf function: no stack).  BUILT AND MEASURED, NOT ADOPTED either: the main loop loses the same 0.010 ms (its two additions
// no longer share a scheduling region with the doublings and the table reads around them) and the fixed-base stage gains
// nothing: N = 4 096 0.3337 → 0.3406 ms (profiles/r06h_kernel_ab.txt, r06h_rows_stage_ms.txt).  The three pasted copies of the
// mixed addition are a local optimum of THIS compiler's schedule; DESIGN.md §9 has the table.
#ifndef IBFT_ROWS_SHARED_ADDS
#define IBFT_ROWS_SHARED_ADDS 0
#endif
constexpr int ROW_TAB_G0 = 32;  // 8 entries × (x, y, z → X·β) + 8 prefix products, then (x, y) of the GTAB_WINDOWS points of u1·G
#ifndef IBFT_ROWS_TAB_PAD_SLOTS
#define IBFT_ROWS_TAB_PAD_SLOTS 0  // experiment: unused slots (what does a workgroup's LDS SIZE alone cost?)
#endif
constexpr int ROW_TAB_SLOTS = ROW_TAB_G0 + (IBFT_ROWS_G_PREFETCH ? 2 * ibftk::GTAB_WINDOWS : 0) + IBFT_ROWS_TAB_PAD_SLOTS;
template <int STOP = 99>
WVF bool recover_pubkey_row(const uint32_t *__restrict__ gtab, const u256 &z_raw, const u256 &r, const u256 &s,
                           uint32_t v, uint32_t flags, uint32_t addr[5], aff &Qa, uint32_t *wtab) {
#define WV_STAGE(n, keep)     \
  if (STOP == (n)) {          \
    lds_prefetch_wait();      \
    addr[0] = (keep);         \
    return ok;                \
  }
  // (the wait: a wavefront must not end with a prefetch into its LDS still on its way)
  const wk k = wk_init();
#if defined(__HIP_DEVICE_COMPILE__) && defined(IBFT_ROWS_PAD_DWORDS)
  // code-placement experiment (profiles/r04n_*): every instruction behind this point moves by 4·IBFT_ROWS_PAD_DWORDS bytes
#pragma unroll
  for (int pad_ = 0; pad_ < IBFT_ROWS_PAD_DWORDS; pad_++) asm volatile("s_nop 0");
#endif
  bool ok = ibftk::sig_in_range(r, s, v, flags);
  const fe rx = secp::fe_from_u256(r);
  const uint32_t x = scatter(rx, k);
  const uint32_t one = k.li == 0 ? 1u : 0u;
  const uint32_t rhs = wfe_mul(wfe_sqr(x, k), x, k) + (k.li == 0 ? 7u : 0u);  // t = x³ + 7, magnitude 2
#if IBFT_ROWS_DEFER_SQRT
  // y = √t is not computed here: R′ = (x·t, t²) on the isomorphic curve stands in for R until the very end, where ONE
  // exponentiation yields both √t and the inverse of the final Z (rows_finish_deferred)
  const waff R1 = waff{wfe_mul(x, rhs, k), wfe_sqr(rhs, k)};
  const uint32_t skeep = R1.x ^ R1.y;
#else
  const uint32_t yc = wfe_sqrt_candidate_rolled(rhs, k);
  const bool on_curve = wfe_is_zero(wfe_sqr(yc, k) + wfe_neg2(rhs, k));  // cross-lane: every row evaluates it
  ok = ok && on_curve;
  fe y = secp::fe_normalize(gather(yc));
  y = secp::l26_select((y.n[0] & 1u) != v, secp::fe_normalize_weak(secp::fe_neg(y, 1)), y);
  const waff R1 = waff{x, scatter(y, k)};
  const uint32_t skeep = y.n[0] ^ y.n[3];
#endif
  WV_STAGE(1, skeep)
  // u1 = −z/r, u2 = s/r (mod n); u2 = k1 + k2·λ
  const secp::sc rinv = secp::sc_from_u256(modinv_wave<secp::ModN>(r, k));
  WV_STAGE(21, skeep ^ rinv.n[0] ^ rinv.n[9])
  const u256 u1 = secp::sc_neg_canon(secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), rinv)));
  const u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(s), rinv));
  WV_STAGE(22, skeep ^ u1.v[0] ^ u2.v[3])
#if IBFT_ROWS_G_PREFETCH
  // (A/B form, see the macro: the sixteen table points of u1·G are asked for NOW — global memory → wave-private LDS — and are
  // long there when the G additions come, ≈ 0.25 ms later)
  {
    const uint32_t ld = k.li < 10 ? k.li : 0u;  // (idle lanes fetch limb 0 and are masked at use)
#pragma unroll
    for (int win = 0; win < ibftk::GTAB_WINDOWS; win++) {
      const int bit = win * ibftk::GTAB_BITS;
      const uint32_t dgt = (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
      const uint32_t *e = gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)win * ibftk::GTAB_ENTRIES + dgt) + ld;
      lds_prefetch_dword(wtab + (ROW_TAB_G0 + 2 * win) * 64, e);
      lds_prefetch_dword(wtab + (ROW_TAB_G0 + 2 * win + 1) * 64, e + 10);
    }
  }
#endif
  const secp::glv_split sp = secp::sc_split_lambda(u2);
  WV_STAGE(2, skeep ^ u1.v[0] ^ sp.k1.v[0] ^ sp.k2.v[1])
  // signed radix-16 digits of |k1|, |k2|: k + 0x88…8 has nibbles d_j + 8, bit 128 is digit 32
  uint32_t w1[5], w2[5];
  {
    uint32_t c1 = 0, c2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      w1[i] = secp::addc(sp.k1.v[i], 0x88888888u, c1);
      w2[i] = secp::addc(sp.k2.v[i], 0x88888888u, c2);
    }
    w1[4] = c1;
    w2[4] = c2;
  }
#if IBFT_ROWS_VARIANT == 1
  // tables 1..8 of R (affine start: mixed additions) and, through X·β, of λR
  wjac T[9];
  T[1] = wjac_from_aff(R1, k);
  T[2] = wjac_dbl(T[1], k);
  T[3] = wjac_add_aff(T[2], R1, k);
  T[4] = wjac_dbl(T[2], k);
  T[5] = wjac_add_aff(T[4], R1, k);
  T[6] = wjac_dbl(T[3], k);
  T[7] = wjac_add_aff(T[6], R1, k);
  T[8] = wjac_dbl(T[4], k);
  // One common Z for the whole table ("effective affine"): with Zc = z2·z3·…·z8 and s_i = Zc / z_i, entry i is
  // (x_i·s_i², y_i·s_i³, Zc) — the SAME Z everywhere, i.e. the affine point (x_i·s_i², y_i·s_i³) of the isomorphic
  // curve y² = x³ + 7·Zc⁶.  The a = 0 formulas never look at the curve constant, so the main loop adds table
  // entries with MIXED additions (11 multiplications instead of 16, 66 times) and Zc is multiplied into the
  // accumulator's Z once, before the G additions bring it back among points of the real curve.  Prefix products
  // forward, suffix products backward: 6 + 12 multiplications, then 4 per entry.
  uint32_t AX[9], AY[9];
  uint32_t Zc;
  {
    uint32_t pre[9];
    pre[2] = T[2].z;
#pragma unroll
    for (int i = 3; i <= 7; i++) pre[i] = wfe_mul(pre[i - 1], T[i].z, k);
    uint32_t suf = T[8].z;  // z_{i+1}·…·z8 while walking down
#pragma unroll
    for (int i = 8; i >= 1; i--) {
      // s_i = (z2…z_{i-1})·(z_{i+1}…z8)
      const uint32_t sc = i == 8 ? pre[7] : (i <= 2 ? suf : wfe_mul(pre[i - 1], suf, k));
      const uint32_t s2 = wfe_sqr(sc, k);
      AX[i] = wfe_mul(T[i].x, s2, k);
      AY[i] = wfe_mul(T[i].y, wfe_mul(s2, sc, k), k);
      if (i <= 7 && i >= 3) suf = wfe_mul(suf, T[i].z, k);  // after the step for i: suf = z_i·…·z8 (needed by i − 1)
      if (i == 2) Zc = wfe_mul(suf, T[2].z, k);              // s_1 = z2·…·z8 = Zc: computed before the step for i = 1
      if (i == 2) suf = Zc;
    }
  }
  const uint32_t beta = scatter(secp::GLV_CONST(1), k);
  uint32_t TX[9];
#pragma unroll
  for (int e = 1; e <= 8; e++) TX[e] = wfe_mul(AX[e], beta, k);
  WV_STAGE(3, AX[3] ^ AY[5] ^ AX[6] ^ AX[7] ^ AY[8] ^ TX[2] ^ TX[8] ^ u1.v[0] ^ w1[0] ^ w2[1] ^ Zc)
  wjac acc = wjac_inf();
#pragma unroll 1
  for (int jd = 32; jd >= 0; jd--) {
    if (jd < 32) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) acc = wjac_dbl<true>(acc, k);
    }
    // digit jd of both scalars (digit 32 is the carry bit, never negative)
    const int n1 = (int)((w1[jd >> 3] >> (4 * (jd & 7))) & 15u), n2 = (int)((w2[jd >> 3] >> (4 * (jd & 7))) & 15u);
    const int d1 = jd == 32 ? (int)(w1[4] & 1u) : n1 - 8, d2 = jd == 32 ? (int)(w2[4] & 1u) : n2 - 8;
    const uint32_t m1 = (uint32_t)(d1 < 0 ? -d1 : d1), m2 = (uint32_t)(d2 < 0 ? -d2 : d2);
    waff q1 = waff{AX[1], AY[1]}, q2 = waff{TX[1], AY[1]};
#pragma unroll
    for (int e = 2; e <= 8; e++) {
      q1.x = m1 == (uint32_t)e ? AX[e] : q1.x;
      q1.y = m1 == (uint32_t)e ? AY[e] : q1.y;
      q2.x = m2 == (uint32_t)e ? TX[e] : q2.x;
      q2.y = m2 == (uint32_t)e ? AY[e] : q2.y;
    }
    q1.y = ((d1 < 0) != sp.neg1) ? wfe_neg1(q1.y, k) : q1.y;  // magnitude ≤ 2
    q2.y = ((d2 < 0) != sp.neg2) ? wfe_neg1(q2.y, k) : q2.y;
    const wjac s1 = wjac_add_aff<true>(acc, q1, k);
    acc = wjac_select(m1 != 0, s1, acc);
    const wjac s2 = wjac_add_aff<true>(acc, q2, k);
    acc = wjac_select(m2 != 0, s2, acc);
  }
#else
  // Tables 1..8 of R (affine start: mixed additions) and, through X·β, of λR — in wave-private LDS, built and
  // brought to ONE common Z by rolled loops: code that runs once per signature is fetched, not executed
  // (profiles/r02c_rows_stage_ms.txt: the same arithmetic written straight-line cost 0.05 ms of instruction-cache
  // misses per launch), so every loop body below appears once.
  //   "Effective affine": with Zc = z2·z3·…·z8 and s_i = Zc / z_i, entry i is (x_i·s_i², y_i·s_i³, Zc) — the affine
  //   point (x_i·s_i², y_i·s_i³) of the isomorphic curve y² = x³ + 7·Zc⁶.  The a = 0 formulas never look at the
  //   curve constant, so the main loop adds table entries with MIXED additions (11 multiplications instead of 16,
  //   66 times) and Zc is multiplied into the accumulator's Z once, before the G additions.
  const uint32_t lane_ = lane_id();
#define WT(slot) wtab[(slot) * 64 + lane_]
  {
    wjac cur = wjac_from_aff(R1, k);
    WT(0) = cur.x;
    WT(1) = cur.y;
    WT(2) = cur.z;
#pragma unroll 1
    for (int i = 2; i <= 8; i++) {  // T[i] = i odd ? T[i−1] + R : 2·T[i/2]   (cur = T[i−1] on entry)
      if (i & 1) {
        cur = wjac_add_aff(cur, R1, k);
      } else {
        const int h = 3 * ((i >> 1) - 1);
        const wjac half = wjac{WT(h), WT(h + 1), WT(h + 2), false};
        cur = wjac_dbl(half, k);
      }
      WT(3 * (i - 1)) = cur.x;
      WT(3 * (i - 1) + 1) = cur.y;
      WT(3 * (i - 1) + 2) = cur.z;
    }
  }
  uint32_t Zc;
  {
    // prefix products pre[i] = z2·…·z_i (pre[0] = pre[1] = 1), then walking down with the suffix product
    // suf = z_{i+1}·…·z8:  s_i = pre[i−1]·suf.  Uniform body (z1 = 1 costs a few multiplications by one).
    WT(24) = one;
    WT(25) = one;
    uint32_t run = one;
#pragma unroll 1
    for (int i = 2; i <= 7; i++) {
      run = wfe_mul(run, WT(3 * (i - 1) + 2), k);
      WT(24 + i) = run;
    }
    const uint32_t beta = scatter(secp::GLV_CONST(1), k);
    uint32_t suf = one;
#pragma unroll 1
    for (int i = 8; i >= 1; i--) {
      const int b = 3 * (i - 1);
      const uint32_t xi = WT(b), yi = WT(b + 1), zi = WT(b + 2), pr = WT(24 + i - 1);
      const uint32_t sc = wfe_mul(pr, suf, k);
      const uint32_t s2 = wfe_sqr(sc, k);
      const uint32_t ax = wfe_mul(xi, s2, k);
      WT(b) = ax;
      WT(b + 1) = wfe_mul(yi, wfe_mul(s2, sc, k), k);
      WT(b + 2) = wfe_mul(ax, beta, k);  // the λ table: X·β
      suf = wfe_mul(suf, zi, k);
    }
    Zc = suf;
  }
  WV_STAGE(3, WT(0) ^ WT(7) ^ WT(23) ^ u1.v[0] ^ w1[0] ^ w2[1] ^ Zc)
  wjac acc = wjac_inf();
#if IBFT_ROWS_PEEL
  // Digit 32 (the carry bit of the recoding, 0 or 1, never negative) outside the loop: the accumulator is still at
  // infinity, so k1's digit is a choice between T[1] and infinity, not an addition, and k2's is ONE addition that runs
  // once per signature (outlined multiply: its code is fetched, not executed).
  {
    const uint32_t y1 = WT(1);
    const uint32_t yn = wfe_neg1(y1, k);  // magnitude 2
    const wjac first = wjac_from_aff(waff{WT(0), sp.neg1 ? yn : y1}, k);
    acc = wjac_select((w1[4] & 1u) != 0, first, acc);
    const wjac s2 = wjac_add_aff(acc, waff{WT(2), sp.neg2 ? yn : y1}, k);
    acc = wjac_select((w2[4] & 1u) != 0, s2, acc);
  }
#endif
#if IBFT_ROWS_G_MERGED == 2
  // The UNIFORM form (needs IBFT_ROWS_G_PREFETCH): both phases read their two operands from wave-private LDS by slot number —
  // table entries in the window iterations, the prefetched G-table points in the fixed-base ones — so an iteration is the same
  // instruction stream in both phases: a slot computation, 4 or 0 doublings, two additions.  No operand path of its own for the
  // fixed-base phase, no preload logic: what the first merged form (IBFT_ROWS_G_MERGED = 1) paid the main loop.
  static_assert(ibftk::GTAB_WINDOWS % 2 == 0 && IBFT_ROWS_G_PREFETCH, "two prefetched fixed-base windows per iteration");
  constexpr int GIT = ibftk::GTAB_WINDOWS / 2;
  wjac hold = wjac_inf();
#pragma unroll 1
  for (int it = 0; it < 32 + GIT; it++) {
    const bool is_g = it >= 32;  // (wave-uniform)
    const int jd = is_g ? 0 : 31 - it, g = is_g ? it - 32 : 0;
    const int n1 = (int)((w1[jd >> 3] >> (4 * (jd & 7))) & 15u), n2 = (int)((w2[jd >> 3] >> (4 * (jd & 7))) & 15u);
    const int d1 = n1 - 8, d2 = n2 - 8;
    const uint32_t m1 = (uint32_t)(d1 < 0 ? -d1 : d1), m2 = (uint32_t)(d2 < 0 ? -d2 : d2);
    const int b1 = 3 * (int)((m1 ? m1 : 1u) - 1u), b2 = 3 * (int)((m2 ? m2 : 1u) - 1u);
    const int gs = ROW_TAB_G0 + 4 * g;
    const int s1x = is_g ? gs : b1, s1y = is_g ? gs + 1 : b1 + 1, s2x = is_g ? gs + 2 : b2 + 2, s2y = is_g ? gs + 3 : b2 + 1;
    if (it == 32) {
      acc.z = wfe_mul(acc.z, Zc, k);  // back from the isomorphic curve (an accumulator at infinity keeps its flag)
      WV_STAGE(4, acc.x ^ acc.y ^ acc.z ^ u1.v[0])
      hold = acc;
      acc = wjac_inf();
      lds_prefetch_wait();  // (asked for a quarter of a millisecond ago)
    }
    waff q1 = waff{WT(s1x) & k.act, WT(s1y) & k.act}, q2 = waff{WT(s2x) & k.act, WT(s2y) & k.act};
    const int ndbl = is_g ? 0 : 4;
#pragma unroll 1
    for (int d = 0; d < ndbl; d++) acc = wjac_dbl<true>(acc, k);
    const bool f1 = !is_g && ((d1 < 0) != sp.neg1), f2 = !is_g && ((d2 < 0) != sp.neg2);
    q1.y = f1 ? wfe_neg1(q1.y, k) : q1.y;  // magnitude ≤ 2
    q2.y = f2 ? wfe_neg1(q2.y, k) : q2.y;
    const int bit1 = (2 * g) * ibftk::GTAB_BITS, bit2 = (2 * g + 1) * ibftk::GTAB_BITS;
    const uint32_t gd1 = (u1.v[bit1 >> 5] >> (bit1 & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    const uint32_t gd2 = (u1.v[bit2 >> 5] >> (bit2 & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    const bool t1 = is_g ? gd1 != 0 : m1 != 0, t2 = is_g ? gd2 != 0 : m2 != 0;
    const wjac s1 = wjac_add_aff<true>(acc, q1, k);
    acc = wjac_select(t1, s1, acc);
    const wjac s2 = wjac_add_aff<true>(acc, q2, k);
    acc = wjac_select(t2, s2, acc);
  }
  const wjac accg_merged = acc;
  acc = hold;
#elif IBFT_ROWS_G_MERGED
  // iterations jd = 31 … 0: digit jd of both scalars (four doublings, two table additions); iterations jd = −1 … −GIT: the
  // fixed-base windows 2g, 2g + 1 (g = −1 − jd) of u1 into an accumulator of their own — the u2·R′ sum steps aside at jd = −1.
  // The two table points of an iteration are asked for one iteration earlier.
  static_assert(ibftk::GTAB_WINDOWS % 2 == 0, "two fixed-base windows per iteration");
  constexpr int GIT = ibftk::GTAB_WINDOWS / 2;
  wjac hold = wjac_inf();
  waff pre1 = waff{0u, 0u}, pre2 = waff{0u, 0u};
  auto g_digit = [&](int win) -> uint32_t {
    const int bit = win * ibftk::GTAB_BITS;
    return (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
  };
#pragma unroll 1
  for (int jd = 31; jd >= -GIT; jd--) {
    waff q1, q2;
    bool t1, t2;
    if (jd >= 0) {  // (wave-uniform)
      const int n1 = (int)((w1[jd >> 3] >> (4 * (jd & 7))) & 15u), n2 = (int)((w2[jd >> 3] >> (4 * (jd & 7))) & 15u);
      const int d1 = n1 - 8, d2 = n2 - 8;
      const uint32_t m1 = (uint32_t)(d1 < 0 ? -d1 : d1), m2 = (uint32_t)(d2 < 0 ? -d2 : d2);
      const int e1 = 3 * (int)((m1 ? m1 : 1u) - 1u), e2 = 3 * (int)((m2 ? m2 : 1u) - 1u);
      q1 = waff{WT(e1), WT(e1 + 1)};  // (read before the doublings that hide the latency)
      q2 = waff{WT(e2 + 2), WT(e2 + 1)};
#pragma unroll 1
      for (int d = 0; d < 4; d++) acc = wjac_dbl<true>(acc, k);
      q1.y = ((d1 < 0) != sp.neg1) ? wfe_neg1(q1.y, k) : q1.y;  // magnitude ≤ 2
      q2.y = ((d2 < 0) != sp.neg2) ? wfe_neg1(q2.y, k) : q2.y;
      t1 = m1 != 0;
      t2 = m2 != 0;
    } else {
      if (jd == -1) {
        acc.z = wfe_mul(acc.z, Zc, k);  // back from the isomorphic curve (an accumulator at infinity keeps its flag)
        WV_STAGE(4, acc.x ^ acc.y ^ acc.z ^ u1.v[0])
        hold = acc;
        acc = wjac_inf();
      }
      const int g = -1 - jd;
      q1 = pre1;
      q2 = pre2;
      t1 = g_digit(2 * g) != 0;
      t2 = g_digit(2 * g + 1) != 0;
    }
    if (jd <= 0 && jd > -GIT) {  // the two points of the NEXT iteration: on their way while this one's additions run
      const int g = -jd;
      pre1 = load_waff(gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)(2 * g) * ibftk::GTAB_ENTRIES + g_digit(2 * g)), k);
      pre2 = load_waff(gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)(2 * g + 1) * ibftk::GTAB_ENTRIES + g_digit(2 * g + 1)), k);
    }
    const wjac s1 = wjac_add_aff<true>(acc, q1, k);
    acc = wjac_select(t1, s1, acc);
    const wjac s2 = wjac_add_aff<true>(acc, q2, k);
    acc = wjac_select(t2, s2, acc);
  }
  const wjac accg_merged = acc;
  acc = hold;
#else
#pragma unroll 1
  for (int jd = IBFT_ROWS_PEEL ? 31 : 32; jd >= 0; jd--) {
    // digit jd of both scalars (digit 32 is the carry bit, never negative); the table reads are issued before the
    // doublings that hide their latency
    const int n1 = (int)((w1[jd >> 3] >> (4 * (jd & 7))) & 15u), n2 = (int)((w2[jd >> 3] >> (4 * (jd & 7))) & 15u);
    const int d1 = jd == 32 ? (int)(w1[4] & 1u) : n1 - 8, d2 = jd == 32 ? (int)(w2[4] & 1u) : n2 - 8;
    const uint32_t m1 = (uint32_t)(d1 < 0 ? -d1 : d1), m2 = (uint32_t)(d2 < 0 ? -d2 : d2);
    const int e1 = 3 * (int)((m1 ? m1 : 1u) - 1u), e2 = 3 * (int)((m2 ? m2 : 1u) - 1u);
    waff q1 = waff{WT(e1), WT(e1 + 1)}, q2 = waff{WT(e2 + 2), WT(e2 + 1)};
    if (jd < 32) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) acc = wjac_dbl<true>(acc, k);
    }
    q1.y = ((d1 < 0) != sp.neg1) ? wfe_neg1(q1.y, k) : q1.y;  // magnitude ≤ 2
    q2.y = ((d2 < 0) != sp.neg2) ? wfe_neg1(q2.y, k) : q2.y;
#if IBFT_ROWS_SHARED_ADDS
    acc = rows_two_adds(acc, q1, q2, m1 != 0, m2 != 0, k);
#else
    const wjac s1 = wjac_add_aff<true>(acc, q1, k);
    acc = wjac_select(m1 != 0, s1, acc);
    const wjac s2 = wjac_add_aff<true>(acc, q2, k);
    acc = wjac_select(m2 != 0, s2, acc);
#endif
  }
#endif  // IBFT_ROWS_G_MERGED
#undef WT
#endif
#if IBFT_ROWS_G_MERGED
  const wjac accg = accg_merged;
#else
  acc.z = wfe_mul(acc.z, Zc, k);  // back from the isomorphic curve (an accumulator at infinity keeps its flag)
  WV_STAGE(4, acc.x ^ acc.y ^ acc.z ^ u1.v[0])
  // u1·G: all the fixed-base windows in this row — into the accumulator itself, or (deferred √) into one of its own: the
  // table points are points of the curve, the accumulator still lives on the isomorphic one
#if IBFT_ROWS_DEFER_SQRT
  wjac accg = wjac_inf();
#else
  wjac &accg = acc;
#endif
#if IBFT_ROWS_G_PREFETCH
  lds_prefetch_wait();  // (issued a quarter of a millisecond ago)
  const uint32_t gl_ = lane_id();
#define WV_GPOINT(win) waff{wtab[(ROW_TAB_G0 + 2 * (win)) * 64 + gl_] & k.act, wtab[(ROW_TAB_G0 + 2 * (win) + 1) * 64 + gl_] & k.act}
#else
#define WV_GPOINT(win) load_waff(gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)(win) * ibftk::GTAB_ENTRIES + dgt), k)
#endif
#if IBFT_ROWS_SHARED_ADDS
  // two windows per step through the main loop's own pair of additions (rows_two_adds); the two table points of the next
  // step are asked for before this step's additions run
  {
    static_assert(ibftk::GTAB_WINDOWS % 2 == 0, "two fixed-base windows per step");
    auto g_digit = [&](int win) -> uint32_t {
      const int bit = win * ibftk::GTAB_BITS;
      return (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    };
    auto g_point = [&](int win) -> waff {
      return load_waff(gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)win * ibftk::GTAB_ENTRIES + g_digit(win)), k);
    };
    waff c1 = g_point(0), c2 = g_point(1);
#pragma unroll 1
    for (int g = 0; g < ibftk::GTAB_WINDOWS / 2; g++) {
      const int gn = g + 1 < ibftk::GTAB_WINDOWS / 2 ? g + 1 : g;  // (the last step re-reads its own points)
      const waff n1 = g_point(2 * gn), n2 = g_point(2 * gn + 1);
      accg = rows_two_adds(accg, c1, c2, g_digit(2 * g) != 0, g_digit(2 * g + 1) != 0, k);
      c1 = n1;
      c2 = n2;
    }
  }
#else
#if IBFT_ROWS_PEEL && IBFT_ROWS_DEFER_SQRT
  // window 0 into an accumulator at infinity is the table point itself (or still infinity for a zero digit)
  {
    const uint32_t dgt = u1.v[0] & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    const waff pt = WV_GPOINT(0);
    accg = wjac_select(dgt != 0, wjac_from_aff(pt, k), accg);
  }
#endif
#pragma unroll 1
  for (int win = (IBFT_ROWS_PEEL && IBFT_ROWS_DEFER_SQRT) ? 1 : 0; win < ibftk::GTAB_WINDOWS; win++) {
    const int bit = win * ibftk::GTAB_BITS;
    const uint32_t dgt = (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    const waff pt = WV_GPOINT(win);
    const wjac sum = wjac_add_aff<true>(accg, pt, k);
    accg = wjac_select(dgt != 0, sum, accg);
  }
#endif  // IBFT_ROWS_SHARED_ADDS
#undef WV_GPOINT
#endif  // IBFT_ROWS_G_MERGED
  WV_STAGE(5, acc.x ^ acc.y ^ acc.z ^ accg.x ^ accg.z)
#if IBFT_ROWS_DEFER_SQRT
  ok = rows_finish_deferred(Qa, accg, acc, rhs, v, k) && ok;
#else
  ok = wjac_to_aff(Qa, acc, k) && ok;
#endif
  WV_STAGE(6, Qa.x.n[0] ^ Qa.y.n[1])
  u256 qx = secp::l26_to_u256(Qa.x), qy = secp::l26_to_u256(Qa.y);
  keccak::address_from_xy(qx.v, qy.v, addr);
  return ok;
#undef WV_STAGE
}

// ---- the warm path, one signature per wavefront --------------------------------------------------------
// Same contract as ibftk::verify_known (verify_dev.h): accept ⇔ R′ = (z/s)·G + (r/s)·Q is finite,
// R′.x = r and parity(R′.y) = v.  The 32 + GTAB_WINDOWS table points are dealt to the four rows;
// a row adds its share with mixed additions (11 multiplications of ≈72 instructions for the four
// rows together, against 11 × 224 per lane in the lane layout), two row-xor additions join them.
WVF bool verify_known_wave(const uint32_t *__restrict__ gtab, const uint32_t *__restrict__ qtab_v, const u256 &z_raw,
                           const u256 &r, const u256 &s, uint32_t v, uint32_t flags) {
  const wk k = wk_init();
  const bool ok = ibftk::sig_in_range(r, s, v, flags);
  // u1 = z/s, u2 = r/s (mod n)
  const secp::sc sinv = secp::sc_from_u256(modinv_wave<secp::ModN>(s, k));
  const u256 u1 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), sinv));
  const u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(r), sinv));
  constexpr int POINTS = ibftk::QTAB_WINDOWS + ibftk::GTAB_WINDOWS;
  static_assert(POINTS % 4 == 0, "points are dealt to four rows");
  wjac acc = wjac_inf();
  // this row's table point of step `it` (point 4·it + row: rows interleave, so every row mixes Q- and G-table points)
  auto point_of = [&](int it, uint32_t &dgt) -> const uint32_t * {
    const int p = 4 * it + (int)k.row;
    if (p < ibftk::QTAB_WINDOWS) {
      dgt = (u2.v[p >> 2] >> (8 * (p & 3))) & 255u;
      return qtab_v + (size_t)ibftk::GTAB_ENTRY_DWORDS * (p * ibftk::QTAB_ENTRIES + dgt);
    }
    const int w = p - ibftk::QTAB_WINDOWS;
    const int bit = w * ibftk::GTAB_BITS;
    dgt = (u1.v[bit >> 5] >> (bit & 31)) & (uint32_t)(ibftk::GTAB_ENTRIES - 1);
    return gtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ((size_t)w * ibftk::GTAB_ENTRIES + dgt);
  };
  // Round 6: software-pipelined by one — the point of step it + 1 is asked for before the addition of step it runs (a
  // dependent read of the validator's table used to open each of the twelve steps: ≈ 1–2 µs in front of a 1.3 µs addition)
  uint32_t dgt;
  waff cur = load_waff(point_of(0, dgt), k);
#pragma unroll 1
  for (int it = 0; it < POINTS / 4; it++) {
    uint32_t dn;
    const waff nxt = load_waff(point_of(it + 1 < POINTS / 4 ? it + 1 : it, dn), k);  // (the last step re-reads its own point)
    const wjac sum = wjac_add_aff<true>(acc, cur, k);
    acc = wjac_select(dgt != 0, sum, acc);
    cur = nxt;
    dgt = dn;
  }
  acc = wjac_add(acc, wjac_lane_xor(acc, 16), k);
  acc = wjac_add(acc, wjac_lane_xor(acc, 32), k);
  aff A;
  const bool fin = jac_to_aff_wave(A, wjac_gather(acc), k);
  const fe rx = secp::fe_from_u256(r);  // r < n < p: canonical limbs
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) diff |= A.x.n[i] ^ rx.n[i];
  return fin && diff == 0 && (A.y.n[0] & 1u) == v && ok;
}

}  // namespace wv
