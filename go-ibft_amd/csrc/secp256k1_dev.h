// secp256k1_dev.h — 256-bit field / scalar / group arithmetic for the gfx950 kernels.
//
// Product code (go-ibft_amd).  Implements what an application's Backend does behind
// go-ibft's Verifier.IsValidCommittedSeal / IsValidValidator
// (/root/reference/core/backend.go:41-45, 53-55): secp256k1 ECDSA public-key recovery.
// The reference ships no arithmetic; conventions are fixed in include/ibftgpu.h.
//
// Representation — chosen from measurements on MI355X (profiles/r01_ubench_int.txt and
// the ISA of the first version): v_mad_u64_u32 issues at the same rate as a carry-add,
// every VCC carry dependency costs two wait states (`s_nop 1`), and 64-bit C accumulators
// over saturated 32-bit limbs compile into ~230 v_mov per multiply.  So field elements
// and scalars use a REDUCED RADIX: 10 limbs of 26 bits in 32-bit VGPRs, lazily carried.
//   * products accumulate in 64-bit columns with pure v_mad_u64_u32 chains
//     (10 × 2^30·2^30 < 2^64: no carry flags anywhere in the hot path),
//   * add / negate are 10 plain 32-bit VALU ops,
//   * reduction folds 2^260 ≡ 0x3D10 + 0x400·2^26 (mod p) with two more mads per limb.
// "magnitude m" below means: every limb ≤ 2m·(2^26−1) (limb 9: 2m·(2^22−1)).
// fe_mul / fe_sqr take magnitude ≤ 8 and return magnitude 1.
//
// One value per lane, straight-line unrolled code, branch-free on data except the rare
// exceptional-point paths, so a wavefront stays converged.  Integer only: no MFMA.
//
// Everything is __host__ __device__ so the same source is unit-tested on the CPU
// (tests/test_dev_arith_host.py builds csrc/host_arith_harness.hip with hipcc's host
// pass); the shipped library only ever runs it on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HD __host__ __device__ __forceinline__

namespace secp {

// ------------------------------------------------------------------ 8×32 "canonical" integers
// Used at the edges only (unpack, range checks, digit extraction, hashing).
struct u256 {
  uint32_t v[8];
};

HD uint32_t N_LIMB(int i) {
  const uint32_t n[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                         0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  return n[i];
}
HD uint32_t NHALF_LIMB(int i) {  // (n-1)/2
  const uint32_t h[8] = {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u,
                         0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
  return h[i];
}
HD uint32_t P_LIMB(int i) { return i == 0 ? 0xFFFFFC2Fu : (i == 1 ? 0xFFFFFFFEu : 0xFFFFFFFFu); }
struct NL {
  HD uint32_t operator()(int i) const { return N_LIMB(i); }
};
struct NHL {
  HD uint32_t operator()(int i) const { return NHALF_LIMB(i); }
};
struct PL {
  HD uint32_t operator()(int i) const { return P_LIMB(i); }
};

HD bool is_zero(const u256 &a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
HD bool eq(const u256 &a, const u256 &b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}
HD u256 zero256() {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}
HD u256 one256() {
  u256 r = zero256();
  r.v[0] = 1;
  return r;
}
HD u256 select(bool c, const u256 &a, const u256 &b) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
HD u256 from_be32(const uint8_t *b) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint8_t *q = b + 4 * (7 - i);
    r.v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
  return r;
}
HD void to_be32(uint8_t *b, const u256 &a) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint8_t *q = b + 4 * (7 - i);
    q[0] = (uint8_t)(a.v[i] >> 24);
    q[1] = (uint8_t)(a.v[i] >> 16);
    q[2] = (uint8_t)(a.v[i] >> 8);
    q[3] = (uint8_t)a.v[i];
  }
}
// explicit carry chains (edge code only: each step costs a VCC wait state on gfx950)
HD uint32_t addc(uint32_t a, uint32_t b, uint32_t &c) {
  unsigned co;
  uint32_t r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
}
HD uint32_t subb(uint32_t a, uint32_t b, uint32_t &c) {
  unsigned co;
  uint32_t r = __builtin_subc(a, b, c, &co);
  c = co;
  return r;
}
template <typename LIMB>
HD bool geq_const(const u256 &a, LIMB limb) {  // a >= constant
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) (void)subb(a.v[i], limb(i), br);
  return br == 0;
}
template <typename LIMB>
HD void sub_const_if(u256 &a, bool c, LIMB limb) {  // a -= c ? constant : 0
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) a.v[i] = subb(a.v[i], c ? limb(i) : 0u, br);
}
HD uint32_t sub256(u256 &r, const u256 &a, const u256 &b) {
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = subb(a.v[i], b.v[i], br);
  return br;
}
HD uint32_t nibble(const u256 &k, int idx) {  // idx 0 = least significant 4 bits
  return (k.v[idx >> 3] >> (4 * (idx & 7))) & 15u;
}
// word `w` of k picked by a chain of selects: a dynamically indexed k.v[w] sends the whole array to the private segment (the
// last 64–112 B of scratch per lane the lane / group kernels still had once their window tables had moved to LDS)
template <int WORDS = 8>
HD uint32_t word_sel(const u256 &k, uint32_t w) {
  uint32_t r = k.v[0];
#pragma unroll
  for (int i = 1; i < WORDS; i++) r = w == (uint32_t)i ? k.v[i] : r;
  return r;
}
HD uint32_t nibble5(const u256 &k, int idx) {  // nibble idx of a value below 2^160 (a biased 128-bit scalar: 33 digits)
  return (word_sel<5>(k, (uint32_t)idx >> 3) >> (4 * (idx & 7))) & 15u;
}
// Digits WITHOUT an index (round 6): a loop that walks the digits of a scalar keeps the scalar in a shift register — the
// current digit sits at a fixed place, a handful of funnel shifts per step move the next one there.  word_sel's select chain
// on a wave-uniform index was turned back into an indexed private-segment array by the compiler (the lane kernel's last
// 112 B of scratch: 18 words written, one read per window); a shift register has nothing to index.
// top digit of a value below 2^(32·(WORDS−1)+4) whose top nibble sits in the low four bits of its top word
template <int WORDS>
HD uint32_t top_nibble(const u256 &k) { return k.v[WORDS - 1] & 15u; }
template <int WORDS>
HD void shl4(u256 &k) {  // k ← k·16 (within WORDS words)
#pragma unroll
  for (int i = WORDS - 1; i >= 1; i--) k.v[i] = (k.v[i] << 4) | (k.v[i - 1] >> 28);
  k.v[0] <<= 4;
}
template <int BITS>
HD void shr_bits(u256 &k) {  // k ← k >> BITS, 0 < BITS < 32
#pragma unroll
  for (int i = 0; i < 7; i++) k.v[i] = (k.v[i] >> BITS) | (k.v[i + 1] << (32 - BITS));
  k.v[7] >>= BITS;
}
template <int WORDS_DOWN>
HD void shr_words(u256 &k) {  // k ← k >> 32·WORDS_DOWN
#pragma unroll
  for (int i = 0; i < 8; i++) k.v[i] = i + WORDS_DOWN < 8 ? k.v[i + WORDS_DOWN] : 0u;
}
template <int BITS>
HD void shr_const(u256 &k) {  // k ← k >> BITS, any constant 0 < BITS ≤ 256
  if constexpr (BITS >= 256) {
    k = u256{{0, 0, 0, 0, 0, 0, 0, 0}};
  } else {
    if constexpr (BITS / 32 > 0) shr_words<BITS / 32>(k);
    if constexpr (BITS % 32 > 0) shr_bits<BITS % 32>(k);
  }
}
// k ← k >> (UNIT·m) for a per-lane m < 2^LEVELS: binary decomposition, each level computed and selected (no divergence)
template <int UNIT, int LEVELS>
HD void shr_units(u256 &k, uint32_t m) {
  if constexpr (LEVELS > 0) {
    u256 t = k;
    shr_const<UNIT>(t);
#pragma unroll
    for (int i = 0; i < 8; i++) k.v[i] = (m & 1u) ? t.v[i] : k.v[i];
    shr_units<UNIT * 2, LEVELS - 1>(k, m >> 1);
  }
}

// ------------------------------------------------------------------ 10×26 limbs
constexpr uint32_t M26 = 0x3FFFFFFu;
constexpr uint32_t M22 = 0x03FFFFFu;

struct l26 {  // 260-bit container: Σ n[i]·2^(26i)
  uint32_t n[10];
};
using fe = l26;  // field element mod p (magnitude-tracked, lazily reduced)
using sc = l26;  // scalar mod n ("weak": limbs < 2^26, value < 2^260)

HD l26 l26_from_u256(const u256 &a) {
  l26 r;
  r.n[0] = a.v[0] & M26;
  r.n[1] = ((a.v[0] >> 26) | (a.v[1] << 6)) & M26;
  r.n[2] = ((a.v[1] >> 20) | (a.v[2] << 12)) & M26;
  r.n[3] = ((a.v[2] >> 14) | (a.v[3] << 18)) & M26;
  r.n[4] = ((a.v[3] >> 8) | (a.v[4] << 24)) & M26;
  r.n[5] = (a.v[4] >> 2) & M26;
  r.n[6] = ((a.v[4] >> 28) | (a.v[5] << 4)) & M26;
  r.n[7] = ((a.v[5] >> 22) | (a.v[6] << 10)) & M26;
  r.n[8] = ((a.v[6] >> 16) | (a.v[7] << 16)) & M26;
  r.n[9] = a.v[7] >> 10;
  return r;
}
// requires limbs < 2^26 and value < 2^256
HD u256 l26_to_u256(const l26 &a) {
  u256 r;
  r.v[0] = a.n[0] | (a.n[1] << 26);
  r.v[1] = (a.n[1] >> 6) | (a.n[2] << 20);
  r.v[2] = (a.n[2] >> 12) | (a.n[3] << 14);
  r.v[3] = (a.n[3] >> 18) | (a.n[4] << 8);
  r.v[4] = (a.n[4] >> 24) | (a.n[5] << 2) | (a.n[6] << 28);
  r.v[5] = (a.n[6] >> 4) | (a.n[7] << 22);
  r.v[6] = (a.n[7] >> 10) | (a.n[8] << 16);
  r.v[7] = (a.n[8] >> 16) | (a.n[9] << 10);
  return r;
}
HD l26 l26_select(bool c, const l26 &a, const l26 &b) {
  l26 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = c ? a.n[i] : b.n[i];
  return r;
}

// 19 column sums C[k] = Σ_{i+j=k} a_i·b_j; limbs ≤ 2^30 keep every column < 2^64
HD void mul_columns(uint64_t C[19], const l26 &a, const l26 &b) {
#pragma unroll
  for (int k = 0; k < 19; k++) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = k - i;
      if (j >= 0 && j < 10) acc += (uint64_t)a.n[i] * b.n[j];
    }
    C[k] = acc;
  }
}
// squaring: off-diagonal products once against the pre-doubled operand (55 mads, not 100)
HD void sqr_columns(uint64_t C[19], const l26 &a) {
  uint32_t d[10];
#pragma unroll
  for (int i = 0; i < 10; i++) d[i] = a.n[i] << 1;  // ≤ 2^31
#pragma unroll
  for (int k = 0; k < 19; k++) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = k - i;
      if (j >= 0 && j < 10 && i < j) acc += (uint64_t)d[i] * a.n[j];
    }
    if ((k & 1) == 0) acc += (uint64_t)a.n[k / 2] * a.n[k / 2];
    C[k] = acc;
  }
}

// ------------------------------------------------------------------ field mod p = 2^256 − 2^32 − 977
// 2^260 ≡ 16·(2^32 + 977) = R0 + R1·2^26
constexpr uint32_t FE_R0 = 0x3D10u, FE_R1 = 0x400u;

HD uint32_t P26(int i) { return i == 0 ? 0x3FFFC2Fu : (i == 1 ? 0x3FFFFBFu : (i == 9 ? M22 : M26)); }

// columns (19 × <2^64) -> magnitude-1 element
HD fe fe_reduce_columns(uint64_t C[19]) {
  // (1) normalise the high columns 10..18 to 26-bit limbs U[10..20]
  uint32_t U[11];
#pragma unroll
  for (int k = 10; k < 18; k++) {
    U[k - 10] = (uint32_t)C[k] & M26;
    C[k + 1] += C[k] >> 26;
  }
  U[8] = (uint32_t)C[18] & M26;
  uint64_t top = C[18] >> 26;  // < 2^38
  U[9] = (uint32_t)top & M26;
  U[10] = (uint32_t)(top >> 26);  // < 2^12
  // (2) fold: limb k gets R0·U[k+10] + R1·U[k+9]
#pragma unroll
  for (int k = 0; k < 10; k++) {
    C[k] += (uint64_t)U[k] * FE_R0;
    if (k >= 1) C[k] += (uint64_t)U[k - 1] * FE_R1;
  }
  // what lands on limbs 10 and 11 (≥ 2^260 again) is folded once more; both are < 2^38
  uint64_t L10 = (uint64_t)U[9] * FE_R1 + (uint64_t)U[10] * FE_R0;
  uint64_t L11 = (uint64_t)U[10] * FE_R1;
  C[0] += L10 * FE_R0;
  C[1] += L10 * FE_R1 + L11 * FE_R0;
  C[2] += L11 * FE_R1;
  // (3) carry the low ten columns
  fe r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    r.n[k] = (uint32_t)C[k] & M26;
    C[k + 1] += C[k] >> 26;
  }
  // (4) limb 9 holds bits ≥ 234; everything above bit 256 folds with 2^256 ≡ 977 + 2^6·2^26
  uint64_t x = C[9] >> 22;  // < 2^42
  r.n[9] = (uint32_t)C[9] & M22;
  uint64_t t0 = (uint64_t)r.n[0] + x * 977u;
  r.n[0] = (uint32_t)t0 & M26;
  uint64_t t1 = (uint64_t)r.n[1] + (x << 6) + (t0 >> 26);
  r.n[1] = (uint32_t)t1 & M26;
  uint64_t t2 = (uint64_t)r.n[2] + (t1 >> 26);  // ≤ 2^26 + 2^22: magnitude 1 allows it
  r.n[2] = (uint32_t)t2;
  return r;
}
// Opaque constant: keeps the compiler from strength-reducing ·0x400 into a 64-bit shift + add
// (3 instructions) where a single v_mad_u64_u32 with the constant in an SGPR does it.
HD uint32_t opaque_const(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(x));
#endif
  return x;
}

// Fused multiply: one running 64-bit accumulator walks the high columns (10..18), emitting the
// 26-bit limbs U[] that fold back with 2^260 ≡ R0 + R1·2^26, then walks the low columns (0..9)
// adding products, folds and the carry in the same v_mad_u64_u32 chain — no separate carry adds.
// PROD(k) must add Σ_{i+j=k} a_i·b_j to acc.  Bounds: products < 2^63.3, everything else < 2^51.
#define SECP_FE_FUSED_BODY(PROD)                                                                  \
  const uint32_t R0 = FE_R0, R1 = opaque_const(FE_R1);                                            \
  const uint32_t R0R1 = FE_R0 * FE_R1, R0R0 = FE_R0 * FE_R0, R1R1 = opaque_const(FE_R1 * FE_R1);   \
  uint32_t U[11];                                                                                 \
  uint64_t acc = 0;                                                                               \
  _Pragma("unroll") for (int k = 10; k <= 18; k++) {                                              \
    PROD(k);                                                                                      \
    U[k - 10] = (uint32_t)acc & M26;                                                              \
    acc >>= 26;                                                                                   \
  }                                                                                               \
  U[9] = (uint32_t)acc & M26; /* acc < 2^38 */                                                    \
  U[10] = (uint32_t)(acc >> 26);                                                                  \
  fe r;                                                                                           \
  acc = 0;                                                                                        \
  _Pragma("unroll") for (int k = 0; k <= 9; k++) {                                                \
    PROD(k);                                                                                      \
    acc += (uint64_t)U[k] * R0;                                                                   \
    if (k >= 1) acc += (uint64_t)U[k - 1] * R1;                                                   \
    /* limbs 10 and 11 of the first fold (U9·R1 + U10·R0, U10·R1) folded once more */             \
    if (k == 0) acc += (uint64_t)U[9] * R0R1 + (uint64_t)U[10] * R0R0;                            \
    if (k == 1) acc += (uint64_t)U[9] * R1R1 + (uint64_t)U[10] * (2u * R0R1);                      \
    if (k == 2) acc += (uint64_t)U[10] * R1R1;                                                    \
    if (k < 9) {                                                                                  \
      r.n[k] = (uint32_t)acc & M26;                                                               \
      acc >>= 26;                                                                                 \
    }                                                                                             \
  }                                                                                               \
  /* column 9 holds bits ≥ 234: everything above bit 256 folds with 2^256 ≡ 977 + 2^6·2^26 */     \
  uint64_t x = acc >> 22; /* < 2^42 */                                                            \
  r.n[9] = (uint32_t)acc & M22;                                                                   \
  uint64_t t0 = (uint64_t)r.n[0] + x * 977u;                                                      \
  r.n[0] = (uint32_t)t0 & M26;                                                                    \
  uint64_t t1 = (uint64_t)r.n[1] + (x << 6) + (t0 >> 26);                                         \
  r.n[1] = (uint32_t)t1 & M26;                                                                    \
  r.n[2] += (uint32_t)(t1 >> 26); /* ≤ 2^26 + 2^23: magnitude 1 allows it */                      \
  return r;

HD fe fe_mul_inl(const fe &a, const fe &b) {
#define SECP_PROD_MUL(k)                                              \
  _Pragma("unroll") for (int i = 0; i < 10; i++) {                    \
    const int j = (k)-i;                                              \
    if (j >= 0 && j < 10) acc += (uint64_t)a.n[i] * b.n[j];           \
  }
  SECP_FE_FUSED_BODY(SECP_PROD_MUL)
#undef SECP_PROD_MUL
}
HD fe fe_sqr_inl(const fe &a) {
  uint32_t d[10];
#pragma unroll
  for (int i = 0; i < 10; i++) d[i] = a.n[i] << 1;  // ≤ 2^31
#define SECP_PROD_SQR(k)                                              \
  _Pragma("unroll") for (int i = 0; i < 10; i++) {                    \
    const int j = (k)-i;                                              \
    if (j >= 0 && j < 10 && i < j) acc += (uint64_t)d[i] * a.n[j];    \
  }                                                                   \
  if (((k)&1) == 0) acc += (uint64_t)a.n[(k) / 2] * a.n[(k) / 2];
  SECP_FE_FUSED_BODY(SECP_PROD_SQR)
#undef SECP_PROD_SQR
}
// On the device the multiply / square bodies are REAL functions (s_swappc), not inlined: the
// fully inlined kernel was 53 k instructions (420 KB) with a 100 KB main-loop body, far beyond
// the 64 KB instruction cache, and instruction fetch — not the VALU — set the time.  Arguments
// are passed as 20 scalars so they travel in VGPRs (a by-value struct pair goes through scratch).
#if defined(__HIP_DEVICE_COMPILE__)
#define SECP_ARGS10(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, \
                       uint32_t p##5, uint32_t p##6, uint32_t p##7, uint32_t p##8, uint32_t p##9
#define SECP_PACK10(p) {{p##0, p##1, p##2, p##3, p##4, p##5, p##6, p##7, p##8, p##9}}
#define SECP_PASS10(x) x.n[0], x.n[1], x.n[2], x.n[3], x.n[4], x.n[5], x.n[6], x.n[7], x.n[8], x.n[9]
static __device__ __attribute__((noinline)) fe fe_mul_fn(SECP_ARGS10(a), SECP_ARGS10(b)) {
  fe a = SECP_PACK10(a), b = SECP_PACK10(b);
  return fe_mul_inl(a, b);
}
static __device__ __attribute__((noinline)) fe fe_sqr_fn(SECP_ARGS10(a)) {
  fe a = SECP_PACK10(a);
  return fe_sqr_inl(a);
}
HD fe fe_mul(const fe &a, const fe &b) { return fe_mul_fn(SECP_PASS10(a), SECP_PASS10(b)); }
HD fe fe_sqr(const fe &a) { return fe_sqr_fn(SECP_PASS10(a)); }
#else
HD fe fe_mul(const fe &a, const fe &b) { return fe_mul_inl(a, b); }
HD fe fe_sqr(const fe &a) { return fe_sqr_inl(a); }
#endif
HD fe fe_add(const fe &a, const fe &b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + b.n[i];
  return r;
}
// −a for a of magnitude ≤ m; result magnitude m+1
HD fe fe_neg(const fe &a, uint32_t m) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = 2u * (m + 1u) * P26(i) - a.n[i];
  return r;
}
// a − b for b of magnitude ≤ mb; result magnitude ma + mb + 1
HD fe fe_sub(const fe &a, const fe &b, uint32_t mb) { return fe_add(a, fe_neg(b, mb)); }
HD fe fe_mul_int(const fe &a, uint32_t k) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] * k;
  return r;
}
// magnitude ≤ 32 -> magnitude 1 (one carry pass + one fold of the bits above 2^256)
HD fe fe_normalize_weak(const fe &a) {
  fe r;
  uint32_t x = a.n[9] >> 22;
  uint32_t t = a.n[0] + x * 977u;
  r.n[0] = t & M26;
  t = a.n[1] + (x << 6) + (t >> 26);
  r.n[1] = t & M26;
#pragma unroll
  for (int i = 2; i < 9; i++) {
    t = a.n[i] + (t >> 26);
    r.n[i] = t & M26;
  }
  r.n[9] = (a.n[9] & M22) + (t >> 26);
  return r;
}
// any magnitude ≤ 32 -> the canonical representative in [0, p), limbs < 2^26
HD fe fe_normalize(const fe &a) {
  fe r = fe_normalize_weak(a);
  // value < 2^256 + 2^240 < 2p after the weak pass: subtract p once if value ≥ p.
  // value ≥ p  ⇔  value + 2^32 + 977 ≥ 2^256
  uint32_t u[10];
  uint32_t t = r.n[0] + 977u;
  u[0] = t & M26;
  t = r.n[1] + 64u + (t >> 26);
  u[1] = t & M26;
#pragma unroll
  for (int i = 2; i < 9; i++) {
    t = r.n[i] + (t >> 26);
    u[i] = t & M26;
  }
  t = r.n[9] + (t >> 26);
  u[9] = t & M22;
  bool ge = (t >> 22) != 0;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = ge ? u[i] : r.n[i];
  return r;
}
HD bool fe_is_zero(const fe &a) {  // a of magnitude ≤ 32: value ≡ 0 (mod p)?
  // after one weak pass the value is in [0, 2^256 + 2^240) with limbs 0..8 < 2^26, so it is a
  // multiple of p only as the all-zero pattern or exactly p's limb pattern
  fe r = fe_normalize_weak(a);
  uint32_t z0 = 0, z1 = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    z0 |= r.n[i];
    z1 |= r.n[i] ^ P26(i);
  }
  return z0 == 0 || z1 == 0;
}
// Cheap NECESSARY condition for z ≡ 0 (mod p), z an output of fe_mul / fe_sqr / fe_normalize_weak: limb 0 is then exactly
// z mod 2^26 and the value is below 2^256 + 2^240 < 2p, so a multiple of p is 0 or p itself and limb 0 is 0 or P26(0).
// Never misses a zero; says "maybe" for one non-zero z in 2^25.  The point additions use it on Z3 (= 2·Z1·H or
// 2·Z1·Z2·H) in place of the full fe_is_zero(H) — ≈60 instructions per addition that decided nothing.
HD bool fe_z_maybe_zero(const fe &z) { return z.n[0] == 0u || z.n[0] == P26(0); }
HD bool fe_equal(const fe &a, const fe &b, uint32_t mb) { return fe_is_zero(fe_sub(a, b, mb)); }
HD bool fe_is_odd(const fe &a_normalized) { return (a_normalized.n[0] & 1u) != 0; }
HD fe fe_from_u256(const u256 &a) { return l26_from_u256(a); }     // a < p assumed by callers
HD u256 fe_to_u256(const fe &a) { return l26_to_u256(fe_normalize(a)); }
HD fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = 0;
  return r;
}
HD fe fe_one() {
  fe r = fe_zero();
  r.n[0] = 1;
  return r;
}
// n squarings in a row (the √ and inversion chains: 253 of a recovery's ≈266 √-chain operations).  On the device ONE
// outlined function holds a rolled loop around the INLINED squaring: a run of n squarings costs one call instead of n
// (round 4: the argument moves and register traffic around a call cost more than the ≈30 instructions they look like).
// n must be wave-uniform (it always is a constant of the chain).
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __attribute__((noinline)) fe fe_sqr_n_fn(SECP_ARGS10(a), int n) {
  fe a = SECP_PACK10(a);
#pragma unroll 1
  for (int i = 0; i < n; i++) a = fe_sqr_inl(a);
  return a;
}
HD fe fe_sqr_n(fe a, int n) { return fe_sqr_n_fn(SECP_PASS10(a), n); }
#else
HD fe fe_sqr_n(fe a, int n) {
  for (int i = 0; i < n; i++) a = fe_sqr(a);
  return a;
}
#endif
// shared prefix of the p−2 and (p+1)/4 addition chains: x223 = a^(2^223 − 1) etc.
struct fe_chain {
  fe x2, x3, x22, x223;
};
HD fe_chain fe_chain_223(const fe &a) {
  fe_chain ch;
  ch.x2 = fe_mul(fe_sqr(a), a);
  ch.x3 = fe_mul(fe_sqr(ch.x2), a);
  fe x6 = fe_mul(fe_sqr_n(ch.x3, 3), ch.x3);
  fe x9 = fe_mul(fe_sqr_n(x6, 3), ch.x3);
  fe x11 = fe_mul(fe_sqr_n(x9, 2), ch.x2);
  ch.x22 = fe_mul(fe_sqr_n(x11, 11), x11);
  fe x44 = fe_mul(fe_sqr_n(ch.x22, 22), ch.x22);
  fe x88 = fe_mul(fe_sqr_n(x44, 44), x44);
  fe x176 = fe_mul(fe_sqr_n(x88, 88), x88);
  fe x220 = fe_mul(fe_sqr_n(x176, 44), x44);
  ch.x223 = fe_mul(fe_sqr_n(x220, 3), ch.x3);
  return ch;
}
// a^(p−2): exponent bits = 223 ones, 0, 22 ones, 0000101101   (a of magnitude ≤ 8)
HD fe fe_inv(const fe &a) {
  fe_chain ch = fe_chain_223(a);
  fe t = fe_mul(fe_sqr_n(ch.x223, 23), ch.x22);
  t = fe_mul(fe_sqr_n(t, 5), a);
  t = fe_mul(fe_sqr_n(t, 3), ch.x2);
  t = fe_mul(fe_sqr_n(t, 2), a);
  return t;
}
// a^((p+1)/4): exponent bits = 223 ones, 0, 22 ones, 00001100; caller checks r² == a
HD fe fe_sqrt_candidate(const fe &a) {
  fe_chain ch = fe_chain_223(a);
  fe t = fe_mul(fe_sqr_n(ch.x223, 23), ch.x22);
  t = fe_mul(fe_sqr_n(t, 6), ch.x2);
  return fe_sqr_n(t, 2);
}

// ------------------------------------------------------------------ scalars mod n
// 2^260 ≡ 16·(2^256 − n) = K (133 bits, six 26-bit limbs)
HD uint32_t SC_K(int i) {
  // 16·(2^256 − n) = 0x14551231950B75FC4402DA1732FC9BEBF0 in 26-bit limbs
  const uint32_t k[6] = {0x09BEBF0u, 0x285CCBFu, 0x3C4402Du, 0x2542DD7u, 0x0551231u, 0x5u};
  return k[i];
}

// generic: fold limbs [10, 10+nh) of L (26-bit each) down with K; L has room for the result
template <int NH>
HD void sc_fold(uint64_t *E, const uint32_t *hi) {
  // E[k] += Σ_{i+j=k} hi[i]·K[j]
#pragma unroll
  for (int i = 0; i < NH; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) E[i + j] += (uint64_t)hi[i] * SC_K(j);
}
// carry-normalise NC 64-bit columns into NC+1 26-bit limbs
template <int NC>
HD void sc_carry(uint32_t *L, uint64_t *E) {
#pragma unroll
  for (int k = 0; k < NC - 1; k++) {
    L[k] = (uint32_t)E[k] & M26;
    E[k + 1] += E[k] >> 26;
  }
  L[NC - 1] = (uint32_t)E[NC - 1] & M26;
  L[NC] = (uint32_t)(E[NC - 1] >> 26);
}
// 19 product columns -> weak scalar (limbs < 2^26, value < 2^260, ≡ product mod n)
HD sc sc_reduce_columns(uint64_t C[19]) {
  uint32_t L[20];
  sc_carry<19>(L, C);  // 20 limbs; L[19] < 2^26
  // round 1: hi = L[10..19] (10 limbs) × K -> columns 0..14
  uint64_t E[15];
#pragma unroll
  for (int k = 0; k < 15; k++) E[k] = k < 10 ? L[k] : 0;
  sc_fold<10>(E, L + 10);
  uint32_t L2[16];
  sc_carry<15>(L2, E);  // 16 limbs
  // round 2: hi = L2[10..15] (6 limbs) × K -> columns 0..10
  uint64_t F[11];
#pragma unroll
  for (int k = 0; k < 11; k++) F[k] = k < 10 ? L2[k] : 0;
  sc_fold<6>(F, L2 + 10);
  uint32_t L3[12];
  sc_carry<11>(L3, F);  // 12 limbs: value < 2^260 + 2^(156+133)
  // round 3: hi = L3[10..11] (≤ 2^29·…) × K -> columns 0..6
  uint64_t G[10];
#pragma unroll
  for (int k = 0; k < 10; k++) G[k] = L3[k];
  sc_fold<2>(G, L3 + 10);
  uint32_t L4[11];
  sc_carry<10>(L4, G);  // L4[10] ∈ {0,1}
  // round 4: one last unit of 2^260
  uint64_t H[10];
#pragma unroll
  for (int k = 0; k < 10; k++) H[k] = L4[k];
#pragma unroll
  for (int j = 0; j < 6; j++) H[j] += (uint64_t)L4[10] * SC_K(j);
  uint32_t L5[11];
  sc_carry<10>(L5, H);
  sc r;
#pragma unroll
  for (int k = 0; k < 10; k++) r.n[k] = L5[k];
  return r;
}
HD sc sc_mul_inl(const sc &a, const sc &b) {
  uint64_t C[19];
  mul_columns(C, a, b);
  return sc_reduce_columns(C);
}
HD sc sc_sqr_inl(const sc &a) {
  uint64_t C[19];
  sqr_columns(C, a);
  return sc_reduce_columns(C);
}
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __attribute__((noinline)) sc sc_mul_fn(SECP_ARGS10(a), SECP_ARGS10(b)) {
  sc a = SECP_PACK10(a), b = SECP_PACK10(b);
  return sc_mul_inl(a, b);
}
static __device__ __attribute__((noinline)) sc sc_sqr_fn(SECP_ARGS10(a)) {
  sc a = SECP_PACK10(a);
  return sc_sqr_inl(a);
}
HD sc sc_mul(const sc &a, const sc &b) { return sc_mul_fn(SECP_PASS10(a), SECP_PASS10(b)); }
HD sc sc_sqr(const sc &a) { return sc_sqr_fn(SECP_PASS10(a)); }
#else
HD sc sc_mul(const sc &a, const sc &b) { return sc_mul_inl(a, b); }
HD sc sc_sqr(const sc &a) { return sc_sqr_inl(a); }
#endif
HD sc sc_from_u256(const u256 &a) { return l26_from_u256(a); }  // any 256-bit value
// weak scalar -> canonical [0, n) in 8×32 words
HD u256 sc_canon(const sc &a) {
  // value = low256 + x·2^256, x = a.n[9] >> 22 (≤ 4 bits); 2^256 ≡ c = 2^256 − n (129 bits)
  uint32_t x = a.n[9] >> 22;
  l26 lo = a;
  lo.n[9] &= M22;
  u256 r = l26_to_u256(lo);
  const uint32_t c[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u};
  uint32_t carry = 0, mc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t m = (i < 5 ? (uint64_t)c[i] * x : 0ull) + mc;  // x·c word i (+ carry of the product)
    mc = (uint32_t)(m >> 32);
    r.v[i] = addc(r.v[i], (uint32_t)m, carry);
  }
  // a carry out of 2^256 (rare) is one more c
  uint32_t wrap = carry;
  carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = addc(r.v[i], (wrap && i < 5) ? c[i] : 0u, carry);
  sub_const_if(r, geq_const(r, NL()), NL());
  sub_const_if(r, geq_const(r, NL()), NL());
  return r;
}
HD u256 sc_neg_canon(const u256 &a) {  // a in [0,n)
  u256 nn, r;
#pragma unroll
  for (int i = 0; i < 8; i++) nn.v[i] = N_LIMB(i);
  sub256(r, nn, a);
  return select(is_zero(a), a, r);
}
HD sc sc_sqr_n(sc a, int n) {
  for (int i = 0; i < n; i++) a = sc_sqr(a);
  return a;
}
// a^(n−2) mod n.  n−2 = [127 ones][0] ‖ 0xBAAEDCE6AF48A03BBFD25E8CD036413F: the run of ones
// by an addition chain, the low 128 bits with 4-bit fixed windows over {a^1..a^15}.
HD sc sc_inv(const sc &a) {
  sc x2 = sc_mul(sc_sqr(a), a);
  sc x3 = sc_mul(sc_sqr(x2), a);
  sc x6 = sc_mul(sc_sqr_n(x3, 3), x3);
  sc x12 = sc_mul(sc_sqr_n(x6, 6), x6);
  sc x24 = sc_mul(sc_sqr_n(x12, 12), x12);
  sc x48 = sc_mul(sc_sqr_n(x24, 24), x24);
  sc x96 = sc_mul(sc_sqr_n(x48, 48), x48);
  sc x120 = sc_mul(sc_sqr_n(x96, 24), x24);
  sc x126 = sc_mul(sc_sqr_n(x120, 6), x6);
  sc t = sc_mul(sc_sqr(x126), a);  // x127
  t = sc_sqr(t);                   // the single 0 bit (bit 128)
  const uint32_t low[4] = {0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u};
  for (int w = 3; w >= 0; w--) {
    uint32_t bits = low[w];
    for (int b = 31; b >= 0; b--) {
      t = sc_sqr(t);
      sc tm = sc_mul(t, a);
      t = l26_select(((bits >> b) & 1u) != 0, tm, t);
    }
  }
  return t;
}

// ------------------------------------------------------------------ GLV endomorphism
// λ·(x, y) = (β·x, y) with λ³ ≡ 1 (mod n), β³ ≡ 1 (mod p).  Any scalar k splits as
// k ≡ k1 + k2·λ (mod n) with |k1|, |k2| < 2^128 (lattice basis (a1,b1),(a2,b2) of the
// endomorphism; g1, g2 = round(2^384·b2/n), round(2^384·(−b1)/n)), which halves the number of
// doublings of the variable-base multiplication.  Constants are the curve's standard ones
// (checked in tests/test_dev_arith_host.py: λ·G = (β·Gx, Gy), k1 + k2·λ ≡ k, 128-bit bounds).
HD l26 GLV_CONST(int which) {
  const uint32_t c[6][10] = {
      // λ
      {0x323BD72u, 0x0A59F06u, 0x2678DF0u, 0x3A88205u, 0x2122E22u, 0x2049916u, 0x261C028u, 0x0C38294u, 0x14CC05Cu, 0x014D8EBu},
      // β
      {0x19501EEu, 0x25B0A1Cu, 0x0995C13u, 0x1D44BD6u, 0x19CF049u, 0x30D0D3Au, 0x24479EAu, 0x01C41B9u, 0x22B657Cu, 0x01EBA5Au},
      // −b1
      {0x2BFE4C3u, 0x11FEA42u, 0x08286F5u, 0x358043Au, 0x0E4437Eu, 0, 0, 0, 0, 0},
      // −b2 (= n − b2)
      {0x1B1562Cu, 0x1736A0Fu, 0x346DD76u, 0x3141DD0u, 0x28A280Au, 0x3FFFFFFu, 0x3FFFFFFu, 0x3FFFFFFu, 0x3FFFFFFu, 0x03FFFFFu},
      // g1
      {0x1DBB031u, 0x0C82691u, 0x0A7FE89u, 0x051C7A3u, 0x13DAA8Au, 0x0A13AC5u, 0x2C90E49u, 0x1AF37A1u, 0x221A7D4u, 0x00C21B4u},
      // g2
      {0x2C47F71u, 0x06D2BA2u, 0x06C6157u, 0x2B277D4u, 0x0221208u, 0x2AFF931u, 0x147FA90u, 0x220A1BDu, 0x2D6010Eu, 0x03910DFu}};
  l26 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = c[which][i];
  return r;
}
// round(k·g / 2^384) for 256-bit k, g (exact 512-bit product, no modular reduction)
HD sc mul_shift_384(const sc &k, const sc &g) {
  uint64_t C[19];
  mul_columns(C, k, g);
  uint32_t L[20];
  sc_carry<19>(L, C);
  // bit 384 = limb 14, bit 20; rounding bit 383 = limb 14, bit 19
  sc r;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    uint32_t lo = (14 + i < 20) ? (L[14 + i] >> 20) : 0u;
    uint32_t hi = (15 + i < 20) ? (L[15 + i] << 6) : 0u;
    r.n[i] = (lo | hi) & M26;
  }
  r.n[0] += (L[14] >> 19) & 1u;  // ≤ 2^26: still a legal sc_mul operand
  return r;
}
HD u256 n256() {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = N_LIMB(i);
  return r;
}
HD u256 add_mod_n(const u256 &a, const u256 &b) {  // a, b in [0, n)
  u256 r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = addc(a.v[i], b.v[i], c);
  sub_const_if(r, c != 0 || geq_const(r, NL()), NL());
  return r;
}
struct glv_split {
  u256 k1, k2;  // magnitudes, < 2^128
  bool neg1, neg2;
};
// low 256 bits of c·w for c < 2^128 (a mul_shift_384 result: limbs 5..9 are zero) and a constant w < 2^130 (five 26-bit limbs)
HD u256 mul_c_by_const(const sc &c, const uint32_t (&w)[5]) {
  uint64_t E[9];
#pragma unroll
  for (int k = 0; k < 9; k++) E[k] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 5; j++) E[i + j] += (uint64_t)c.n[i] * w[j];  // < 5·2^53
  l26 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r.n[k] = (uint32_t)E[k] & M26;
    E[k + 1] += E[k] >> 26;
  }
  r.n[8] = (uint32_t)E[8] & M26;
  r.n[9] = (uint32_t)(E[8] >> 26) & M22;  // (bits ≥ 256 are dropped: the caller works modulo 2^256)
  return l26_to_u256(r);
}
// Round 4: the split in EXACT INTEGERS.  With c1 = round(k·g1/2^384), c2 = round(k·g2/2^384) the halves are
//   k1 = k − c1·a1 − c2·a2      k2 = −c1·b1 − c2·b2 = c1·|b1| − c2·a1        (b2 = a1)
// and both lie in (−2^128, 2^128): computed modulo 2^256 in two's complement they are exact, the sign is bit 255.  Four
// 128 × 130-bit products instead of three multiplications modulo n with their four-round reductions, three
// canonicalisations and two modular additions (the same k1, k2: k1 + k2·λ ≡ k and the bounds are what the tests check).
HD glv_split sc_split_lambda(const u256 &k) {  // k in [0, n)
  const uint32_t A1[5] = {0x284EB15u, 0x3243924u, 0x2BCDE86u, 0x0869F51u, 0x03086D2u};   // a1 = b2
  const uint32_t B1M[5] = {0x2BFE4C3u, 0x11FEA42u, 0x08286F5u, 0x358043Au, 0x0E4437Eu};  // −b1
  const uint32_t A2[5] = {0x144CFD8u, 0x0442367u, 0x33F657Cu, 0x3DEA38Bu, 0x114CA50u};   // a2 (129 bits)
  const sc ks = sc_from_u256(k);
  const sc c1 = mul_shift_384(ks, GLV_CONST(4)), c2 = mul_shift_384(ks, GLV_CONST(5));
  u256 k2w, t, k1w;
  sub256(k2w, mul_c_by_const(c1, B1M), mul_c_by_const(c2, A1));
  {
    const u256 p = mul_c_by_const(c1, A1), q = mul_c_by_const(c2, A2);
    uint32_t cy = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(p.v[i], q.v[i], cy);
  }
  sub256(k1w, k, t);
  glv_split s;
  s.neg1 = (k1w.v[7] >> 31) != 0;
  s.neg2 = (k2w.v[7] >> 31) != 0;
  u256 n1, n2;
  sub256(n1, zero256(), k1w);
  sub256(n2, zero256(), k2w);
  s.k1 = select(s.neg1, n1, k1w);
  s.k2 = select(s.neg2, n2, k2w);
  return s;
}

// true if the predicate holds on ANY lane of the wavefront (host build: the single "lane").
// Used to turn the rare exceptional-point paths into WAVE-UNIFORM branches: measured on
// gfx950 / ROCm 7.2, a call to an outlined device function that executes under a partial EXEC
// mask can return wrong values (tests/test_gpu_arith.py::test_gtab_and_full_recover_on_gpu
// caught it), so no function call in this code base may sit under divergent control flow.
HD bool wave_any(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any(c ? 1 : 0) != 0;
#else
  return c;
#endif
}

// ------------------------------------------------------------------ group (Jacobian, a = 0)
// Coordinates are kept at magnitude 1 between operations.
struct jac {
  fe x, y, z;
  bool inf;
};
struct aff {
  fe x, y;
};

HD jac jac_inf() {
  jac r;
  r.x = fe_zero();
  r.y = fe_zero();
  r.z = fe_zero();
  r.inf = true;
  return r;
}
HD jac jac_select(bool c, const jac &a, const jac &b) {
  jac r;
  r.x = l26_select(c, a.x, b.x);
  r.y = l26_select(c, a.y, b.y);
  r.z = l26_select(c, a.z, b.z);
  r.inf = c ? a.inf : b.inf;
  return r;
}
HD jac jac_from_aff(const aff &a) {
  jac r;
  r.x = a.x;
  r.y = a.y;
  r.z = fe_one();
  r.inf = false;
  return r;
}
// The point formulas exist in two code shapes (template parameter INL):
//   false  every field multiplication is a CALL of fe_mul_fn / fe_sqr_fn (≈30 instructions of argument moves and
//          s_swappc per call, small code) — everything that runs a handful of times per signature;
//   true   the multiplications are INLINED into the formula (no call overhead, ≈1.5–2.5 k instructions per formula) —
//          for the ONE copy of the doubling and of the mixed addition inside a rolled main loop, where ≈3/4 of a
//          recovery's multiplications are (round 4; the fully inlined kernel of round 1 was 53 k instructions and
//          instruction fetch set its time — one inlined copy per loop stays far inside the 64 KB instruction cache).
template <bool INL>
HD fe fe_mul_t(const fe &a, const fe &b) {
  if constexpr (INL)
    return fe_mul_inl(a, b);
  else
    return fe_mul(a, b);
}
template <bool INL>
HD fe fe_sqr_t(const fe &a) {
  if constexpr (INL)
    return fe_sqr_inl(a);
  else
    return fe_sqr(a);
}
// dbl-2009-l: 2M + 5S.  Magnitudes in comments.
template <bool INL>
HD jac jac_dbl_t(const jac &p) {
  fe A = fe_sqr_t<INL>(p.x);                            // 1
  fe B = fe_sqr_t<INL>(p.y);                            // 1
  fe C = fe_sqr_t<INL>(B);                              // 1
  fe t = fe_sqr_t<INL>(fe_add(p.x, B));                 // in 2 -> 1
  t = fe_add(fe_add(t, fe_neg(A, 1)), fe_neg(C, 1));    // 1+2+2 = 5
  fe D = fe_normalize_weak(fe_mul_int(t, 2));           // 10 -> 1
  fe E = fe_mul_int(A, 3);                              // 3
  fe F = fe_sqr_t<INL>(E);                              // 1
  jac r;
  r.x = fe_normalize_weak(fe_add(F, fe_neg(fe_mul_int(D, 2), 2)));       // 1 + 3 = 4 -> 1
  fe C8 = fe_mul_int(C, 8);                                              // 8
  fe y3 = fe_add(fe_mul_t<INL>(E, fe_add(D, fe_neg(r.x, 1))), fe_neg(C8, 8));   // E:3, D−X3: 1+2=3; 1 + 9 = 10
  r.y = fe_normalize_weak(y3);
  r.z = fe_mul_t<INL>(fe_mul_int(p.y, 2), p.z);                          // in 2,1 -> 1
  r.inf = p.inf;  // no point of order 2 on this curve: y = 0 cannot occur for on-curve input
  return r;
}
HD jac jac_dbl(const jac &p) { return jac_dbl_t<false>(p); }
// add-2007-bl: 11M + 5S, exceptional cases handled (rare, divergent)
template <bool INL>
HD jac jac_add_t(const jac &p, const jac &q) {
  fe z1z1 = fe_sqr_t<INL>(p.z);
  fe z2z2 = fe_sqr_t<INL>(q.z);
  fe u1 = fe_mul_t<INL>(p.x, z2z2);
  fe u2 = fe_mul_t<INL>(q.x, z1z1);
  fe s1 = fe_mul_t<INL>(fe_mul_t<INL>(p.y, q.z), z2z2);
  fe s2 = fe_mul_t<INL>(fe_mul_t<INL>(q.y, p.z), z1z1);
  fe h = fe_add(u2, fe_neg(u1, 1));    // 3
  fe rr = fe_add(s2, fe_neg(s1, 1));   // 3
  fe i = fe_sqr_t<INL>(fe_mul_int(h, 2));  // in 6 -> 1
  fe j = fe_mul_t<INL>(h, i);          // 1
  fe r2 = fe_mul_int(rr, 2);           // 6
  fe v = fe_mul_t<INL>(u1, i);         // 1
  jac r;
  // X3 = r2² − J − 2V
  r.x = fe_normalize_weak(fe_add(fe_add(fe_sqr_t<INL>(r2), fe_neg(j, 1)), fe_neg(fe_mul_int(v, 2), 2)));  // 1+2+3 = 6 -> 1
  // Y3 = r2·(V − X3) − 2·S1·J
  fe s1j2 = fe_mul_int(fe_mul_t<INL>(s1, j), 2);                                                          // 2
  r.y = fe_normalize_weak(fe_add(fe_mul_t<INL>(r2, fe_add(v, fe_neg(r.x, 1))), fe_neg(s1j2, 2)));         // 1 + 3 -> 1
  // Z3 = ((Z1+Z2)² − Z1Z1 − Z2Z2)·H
  fe zz = fe_add(fe_add(fe_sqr_t<INL>(fe_add(p.z, q.z)), fe_neg(z1z1, 1)), fe_neg(z2z2, 1));              // 5
  r.z = fe_mul_t<INL>(zz, h);                                                                             // in 5,3 -> 1
  r.inf = false;
  // exceptional cases (P = ±Q ⇔ H ≡ 0): Z1·Z2 ≢ 0 for finite points, so H ≡ 0 ⇔ Z3 ≡ 0, and fe_z_maybe_zero(Z3) never
  // misses it — the exact tests run once in 2^25 additions, for the whole wave if any lane asks
  const bool both = !p.inf && !q.inf;
  const bool maybe = both && fe_z_maybe_zero(r.z);
  if (__builtin_expect(wave_any(maybe), false)) {  // (the rare path is laid out of line: the common one falls through)
    const bool hz = maybe && fe_is_zero(h), rz = fe_is_zero(rr);
    const bool same = hz && rz;       // P == Q
    const bool opposite = hz && !rz;  // P == −Q
    if (wave_any(same)) r = jac_select(same, jac_dbl_t<false>(p), r);
    r = jac_select(opposite, jac_inf(), r);
  }
  r = jac_select(q.inf, p, r);
  r = jac_select(p.inf, q, r);
  return r;
}
HD jac jac_add(const jac &p, const jac &q) { return jac_add_t<false>(p, q); }
// madd-2007-bl: 7M + 4S (q affine, never infinity)
template <bool INL>
HD jac jac_add_aff_t(const jac &p, const aff &q) {
  fe z1z1 = fe_sqr_t<INL>(p.z);
  fe u2 = fe_mul_t<INL>(q.x, z1z1);
  fe s2 = fe_mul_t<INL>(fe_mul_t<INL>(q.y, p.z), z1z1);
  fe h = fe_add(u2, fe_neg(p.x, 1));   // 3
  fe rr = fe_add(s2, fe_neg(p.y, 1));  // 3
  fe hh = fe_sqr_t<INL>(h);            // 1
  fe i = fe_mul_int(hh, 4);            // 4
  fe j = fe_mul_t<INL>(h, i);          // 1
  fe r2 = fe_mul_int(rr, 2);           // 6
  fe v = fe_mul_t<INL>(p.x, i);        // 1
  jac r;
  r.x = fe_normalize_weak(fe_add(fe_add(fe_sqr_t<INL>(r2), fe_neg(j, 1)), fe_neg(fe_mul_int(v, 2), 2)));  // 6 -> 1
  fe y1j2 = fe_mul_int(fe_mul_t<INL>(p.y, j), 2);                                                         // 2
  r.y = fe_normalize_weak(fe_add(fe_mul_t<INL>(r2, fe_add(v, fe_neg(r.x, 1))), fe_neg(y1j2, 2)));         // 4 -> 1
  // Z3 = (Z1+H)² − Z1Z1 − HH
  r.z = fe_normalize_weak(fe_add(fe_add(fe_sqr_t<INL>(fe_add(p.z, h)), fe_neg(z1z1, 1)), fe_neg(hh, 1)));  // in 4; 5 -> 1
  r.inf = false;
  const jac qj = jac_from_aff(q);
  const bool maybe = !p.inf && fe_z_maybe_zero(r.z);  // Z3 = 2·Z1·H: H ≡ 0 ⇒ Z3 ∈ {0, p} (see fe_z_maybe_zero)
  if (__builtin_expect(wave_any(maybe), false)) {  // (the rare path is laid out of line: the common one falls through)
    const bool hz = maybe && fe_is_zero(h), rz = fe_is_zero(rr);
    const bool same = hz && rz;
    const bool opposite = hz && !rz;
    if (wave_any(same)) r = jac_select(same, jac_dbl_t<false>(qj), r);  // (rare: the call shape keeps the loop body small)
    r = jac_select(opposite, jac_inf(), r);
  }
  r = jac_select(p.inf, qj, r);
  return r;
}
HD jac jac_add_aff(const jac &p, const aff &q) { return jac_add_aff_t<false>(p, q); }
// returns false if p is infinity; r.x / r.y are canonical (fully normalised)
HD bool jac_to_aff(aff &r, const jac &p) {
  fe zi = fe_inv(p.z);
  fe zi2 = fe_sqr(zi);
  r.x = fe_normalize(fe_mul(p.x, zi2));
  r.y = fe_normalize(fe_mul(p.y, fe_mul(zi2, zi)));
  return !(p.inf || fe_is_zero(p.z));
}

HD aff generator() {
  u256 gx, gy;
  const uint32_t x[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                         0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
  const uint32_t y[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                         0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    gx.v[i] = x[i];
    gy.v[i] = y[i];
  }
  aff g;
  g.x = fe_from_u256(gx);
  g.y = fe_from_u256(gy);
  return g;
}

}  // namespace secp
