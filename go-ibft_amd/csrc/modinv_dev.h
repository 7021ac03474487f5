// modinv_dev.h — constant-time modular inversion by "safegcd" divsteps (Bernstein & Yang,
// "Fast constant-time gcd computation and modular inversion", 2019) for the gfx950 kernels.
//
// Product code.  Replaces the Fermat inversions (a^(p−2), a^(n−2): ≈56 k and ≈171 k VALU
// instructions per signature) by 20 batches of 30 divsteps (≈12 k): every lane runs the same
// straight-line instruction stream whatever its data, which is exactly what a wavefront wants.
//
// Half-delta variant: ζ = −(δ + ½) starts at −1; 590 divsteps are enough for any 256-bit
// modulus (computer-verified bound used by libsecp256k1's modinv), we run 600.
//   divstep: if ζ < 0 and g odd: (ζ, f, g) ← (−ζ − 2, g, (g − f)/2)
//            else:               (ζ, f, g) ← (ζ − 1, f, (g + (g mod 2)·f)/2)
// A batch of 30 steps only looks at the low 32 bits of f, g and yields a 2×2 integer matrix
// t with t·(f, g) = 2^30·(f', g'); the same matrix is applied to the Bézout pair (d, e)
// modulo M, made divisible by 2^30 by adding a multiple of M (needs M⁻¹ mod 2^30).
// Limbs: 9 signed limbs of 30 bits.  Invariant: d·x ≡ f and e·x ≡ g (mod M) throughout, so
// when g reaches 0 and f = ±1 the inverse is ±d.
#pragma once
#include "secp256k1_dev.h"

namespace secp {

constexpr int32_t M30 = 0x3FFFFFFF;

struct s30 {
  int32_t v[9];
};
struct trans2x2 {
  int32_t u, v, q, r;
};

// ---- moduli in radix 2^30 and their inverses mod 2^30 (tests/test_dev_arith_host.py re-derives them)
struct ModP {
  static HD int32_t limb(int i) {
    const int32_t m[9] = {0x3FFFFC2F, 0x3FFFFFFB, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF,
                          0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF};
    return m[i];
  }
  static HD uint32_t inv30() { return 0x2DDACACFu; }
};
struct ModN {
  static HD int32_t limb(int i) {
    const int32_t m[9] = {0x10364141, 0x3F497A33, 0x348A03BB, 0x2BB739AB, 0x3FFFFEBA,
                          0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF};
    return m[i];
  }
  static HD uint32_t inv30() { return 0x2A774EC1u; }
};

HD s30 s30_from_u256(const u256 &a) {
  s30 r;
  r.v[0] = (int32_t)(a.v[0] & M30);
  r.v[1] = (int32_t)(((a.v[0] >> 30) | (a.v[1] << 2)) & M30);
  r.v[2] = (int32_t)(((a.v[1] >> 28) | (a.v[2] << 4)) & M30);
  r.v[3] = (int32_t)(((a.v[2] >> 26) | (a.v[3] << 6)) & M30);
  r.v[4] = (int32_t)(((a.v[3] >> 24) | (a.v[4] << 8)) & M30);
  r.v[5] = (int32_t)(((a.v[4] >> 22) | (a.v[5] << 10)) & M30);
  r.v[6] = (int32_t)(((a.v[5] >> 20) | (a.v[6] << 12)) & M30);
  r.v[7] = (int32_t)(((a.v[6] >> 18) | (a.v[7] << 14)) & M30);
  r.v[8] = (int32_t)(a.v[7] >> 16);
  return r;
}
// limbs 0..7 in [0, 2^30), limb 8 in [0, 2^16)
HD u256 s30_to_u256(const s30 &a) {
  u256 r;
  const uint32_t *v = reinterpret_cast<const uint32_t *>(a.v);
  r.v[0] = v[0] | (v[1] << 30);
  r.v[1] = (v[1] >> 2) | (v[2] << 28);
  r.v[2] = (v[2] >> 4) | (v[3] << 26);
  r.v[3] = (v[3] >> 6) | (v[4] << 24);
  r.v[4] = (v[4] >> 8) | (v[5] << 22);
  r.v[5] = (v[5] >> 10) | (v[6] << 20);
  r.v[6] = (v[6] >> 12) | (v[7] << 18);
  r.v[7] = (v[7] >> 14) | (v[8] << 16);
  return r;
}

// 30 divsteps on the low words; returns the new ζ and the transition matrix
HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, trans2x2 &t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;  // two's complement, wrap-around arithmetic
  uint32_t f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);  // all ones if ζ < 0
    uint32_t c2 = 0u - (g & 1u);           // all ones if g odd
    // x, y, z = ±(f, u, v): negated when ζ < 0
    uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
    // g odd: (g, q, r) += (x, y, z)
    g += x & c2;
    q += y & c2;
    r += z & c2;
    // swap case (ζ < 0 and g was odd): ζ ← −ζ − 2, (f, u, v) += new (g, q, r); otherwise ζ ← ζ − 1
    c1 &= c2;
    zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;  // g is even here; only the low bits matter inside a batch
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return zeta;
}

// (helpers of divsteps_30_lockstep)  opaque_mask: the optimiser must not see a boolean in a select mask;
// mul24_lo: a product of which only the low bits are used — v_mul_u32_u24 on the device (it reads the low 24 bits of its
// operands), the plain product elsewhere.
HD uint32_t opaque_mask(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(v));
#endif
  return v;
}
// the loop's vote: every lane of the wavefront (device), every coroutine of the emulated wavefront (tests: the rounds are
// then really taken in lockstep, finished lanes idling), or just this value (plain host code)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(IBFT_WAVE_EMUL)
}  // namespace secp
namespace wave_emul {
inline uint64_t ballot(bool c);  // wave_emul.h
inline int lane();               // −1 outside an emulated wavefront (the harness also calls plain host code)
}
namespace secp {
#endif
HD bool lockstep_any(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any(c ? 1 : 0) != 0;
#elif defined(IBFT_WAVE_EMUL)
  return wave_emul::lane() < 0 ? c : wave_emul::ballot(c) != 0;
#else
  return c;
#endif
}
HD uint32_t mul24_lo(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}
// ---- variable-time divsteps for lanes in lockstep -------------------------------------------------------
// 30 divsteps on the low words, variable time, written so that the lanes of a wavefront (or the DPP rows of the
// row-per-signature recover, wave_fe_dev.h) may hold DIFFERENT values: no branch but the loop's own vote; lanes that are
// done idle through the remaining rounds (nothing selected, a zero multiplier).  One round =
//   strip the zero run of g with one ctz (a sentinel bit bounds it by the steps that are left): those steps only halve;
//   ζ < 0 (g is odd now): the swap of the divstep, (ζ, f, g, u, v, q, r) ← (−ζ − 1, g, −f, q, r, −u, −v) — its
//     subtraction is left to the next line (−f + 1·g);
//   ζ ≥ 0 everywhere now, so the next min(ζ + 1, steps left) steps cannot swap: each adds f to g when g is odd, and
//     halves.  Up to SIX of them are taken at once (round 4): w = −g/f mod 2^L makes g + w·f ≡ 0 (mod 2^L), and
//     (g, q, r) += w·(f, u, v) is exactly what those L steps add — Σ b_j·2^j·(f, u, v) with b_j the parities they
//     would have met — the halvings follow as the next zero run.  −1/f mod 64 = f·(f² − 2)  (f·that = (f² − 1)² − 1,
//     and f² − 1 ≡ 0 mod 8).  Four rows in lockstep need 231 rounds per inversion this way, 311 one step at a time.
// Same transition matrix and ζ as divsteps_30, step for step (tests/test_dev_arith_host.py, tests/test_dev_wave_host.py).

HD int32_t divsteps_30_lockstep(int32_t zeta, uint32_t f0, uint32_t g0, trans2x2 &t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
  int i = 30;
#define WV_STRIP_ZEROS()                         \
  {                                              \
    const uint32_t m = g | (1u << i);            \
    const int z = __builtin_ctz(m);              \
    g >>= z;                                     \
    u <<= z;                                     \
    v <<= z;                                     \
    zeta -= z;                                   \
    i -= z;                                      \
  }
  // "strip; do { round; strip; } while (vote)": the loop closes with ONE vote and ONE branch (a round with no row live
  // changes nothing, so entering it unasked is harmless)
  WV_STRIP_ZEROS()
  do {
    // (the optimiser must not see a boolean in the mask: it would turn the bit-selects below into compares +
    // conditional moves; as a plain mask each is one v_bfi_b32)
    const uint32_t c = opaque_mask((uint32_t)(zeta >> 31) & (i != 0 ? 0xFFFFFFFFu : 0u));
    const uint32_t nf = 0u - f, nu = 0u - u, nv = 0u - v;
    const uint32_t f2 = (g & c) | (f & ~c), u2 = (q & c) | (u & ~c), v2 = (r & c) | (v & ~c);
    g = (nf & c) | (g & ~c);
    q = (nu & c) | (q & ~c);
    r = (nv & c) | (r & ~c);
    f = f2;
    u = u2;
    v = v2;
    zeta = (int32_t)((uint32_t)zeta ^ c);  // −ζ − 1 = ~ζ
    // L = min(ζ + 1, steps left, 6); a finished row has no steps left: L = 0, w = 0
    const int cap = i < 6 ? i : 6;
    int L = zeta + 1;
    L = L < 0 ? 0 : L;
    L = L > cap ? cap : L;
    const uint32_t w = mul24_lo(mul24_lo(mul24_lo(f, f) - 2u, f), g) & ((1u << L) - 1u);  // only bits 0..5 matter
    g += w * f;
    q += w * u;
    r += w * v;
    WV_STRIP_ZEROS()
  } while (lockstep_any(i != 0));
#undef WV_STRIP_ZEROS
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return zeta;
}


// (f, g) ← t·(f, g) / 2^30  (exact)
HD void update_fg_30(s30 &f, s30 &g, const trans2x2 &t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f.v[0] + v * g.v[0];
  int64_t cg = q * f.v[0] + r * g.v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cf += u * f.v[i] + v * g.v[i];
    cg += q * f.v[i] + r * g.v[i];
    f.v[i - 1] = (int32_t)cf & M30;
    g.v[i - 1] = (int32_t)cg & M30;
    cf >>= 30;
    cg >>= 30;
  }
  f.v[8] = (int32_t)cf;
  g.v[8] = (int32_t)cg;
}

// (d, e) ← t·(d, e) / 2^30 (mod M); d, e stay in (−2M, M)
template <class MOD>
HD void update_de_30(s30 &d, s30 &e, const trans2x2 &t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;  // sign masks
  // start with the multiple of M that offsets a negative d / e, then fix divisibility by 2^30
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = u * d.v[0] + v * e.v[0];
  int64_t ce = q * d.v[0] + r * e.v[0];
  md -= (int32_t)((MOD::inv30() * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((MOD::inv30() * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)MOD::limb(0) * md;
  ce += (int64_t)MOD::limb(0) * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cd += u * d.v[i] + v * e.v[i] + (int64_t)MOD::limb(i) * md;
    ce += q * d.v[i] + r * e.v[i] + (int64_t)MOD::limb(i) * me;
    d.v[i - 1] = (int32_t)cd & M30;
    e.v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}

// r in (−2M, M) → [0, M), negated first when neg is set
template <class MOD>
HD void normalize_30(s30 &r, bool neg) {
  int32_t add = r.v[8] >> 31;  // negative: add M
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += MOD::limb(i) & add;
  int32_t nm = neg ? -1 : 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (r.v[i] ^ nm) - nm;
  // carry
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
  add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += MOD::limb(i) & add;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
}

// x⁻¹ mod M for canonical x in [0, M); 0 maps to 0 (like the Fermat chain).
// Constant-time batches.  (Round 4 tried divsteps_30_lockstep here too: with 64 different values in lockstep a batch
// needs ≈14 rounds of 33 instructions instead of 30 × 20, but the loop inside the 20 batches pushes the lane / group
// kernels past 256 registers or into spills — N = 16 384 unchanged, N = 65 536 cold +0.5 %, warm +3.7 % slower,
// profiles/r04f_lockstep_lane_ab.txt.  The lockstep form stays where a value owns a DPP row: wave_fe_dev.h.)
template <class MOD>
HD u256 modinv(const u256 &x) {
  s30 d, e, f, g = s30_from_u256(x);
#pragma unroll
  for (int i = 0; i < 9; i++) {
    d.v[i] = 0;
    e.v[i] = 0;
    f.v[i] = MOD::limb(i);
  }
  e.v[0] = 1;
  int32_t zeta = -1;
  for (int b = 0; b < 20; b++) {
    trans2x2 t;
    zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    update_de_30<MOD>(d, e, t);
    update_fg_30(f, g, t);
  }
  // g = 0 and f = ±gcd; the inverse is d·sign(f)
  normalize_30<MOD>(d, f.v[8] < 0);
  return s30_to_u256(d);
}

// ---- ONE copy for both moduli (round 5) --------------------------------------------------------------------------------
// A lane-layout kernel inverts twice per signature — s or r mod n at the start, Z mod p at the end — and the two template
// instances above are two pasted copies of ≈ 1 050 instructions (6.6 KB each).  On one lease in four an instruction fetch
// past the 64 KB instruction cache costs 65 % more, so kernel CODE is a resource: here the modulus is a
// (wave-uniform) run-time value and the device build calls ONE outlined copy.  Same algorithm, same batches, same results
// (tests/test_dev_arith_host.py checks it against the template form and big ints).
struct mod_rt {
  int32_t m[9];
  uint32_t inv30;
};
HD mod_rt mod_rt_select(bool is_p) {
  mod_rt r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.m[i] = is_p ? ModP::limb(i) : ModN::limb(i);
  r.inv30 = is_p ? ModP::inv30() : ModN::inv30();
  return r;
}
HD void update_de_30_rt(s30 &d, s30 &e, const trans2x2 &t, const mod_rt &M) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = u * d.v[0] + v * e.v[0];
  int64_t ce = q * d.v[0] + r * e.v[0];
  md -= (int32_t)((M.inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((M.inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)M.m[0] * md;
  ce += (int64_t)M.m[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cd += u * d.v[i] + v * e.v[i] + (int64_t)M.m[i] * md;
    ce += q * d.v[i] + r * e.v[i] + (int64_t)M.m[i] * me;
    d.v[i - 1] = (int32_t)cd & M30;
    e.v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}
HD void normalize_30_rt(s30 &r, bool neg, const mod_rt &M) {
  int32_t add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += M.m[i] & add;
  int32_t nm = neg ? -1 : 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (r.v[i] ^ nm) - nm;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
  add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += M.m[i] & add;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
}
HD u256 modinv_rt(const u256 &x, bool is_p) {
  const mod_rt M = mod_rt_select(is_p);
  s30 d, e, f, g = s30_from_u256(x);
#pragma unroll
  for (int i = 0; i < 9; i++) {
    d.v[i] = 0;
    e.v[i] = 0;
    f.v[i] = M.m[i];
  }
  e.v[0] = 1;
  int32_t zeta = -1;
  for (int b = 0; b < 20; b++) {
    trans2x2 t;
    zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    update_de_30_rt(d, e, t, M);
    update_fg_30(f, g, t);
  }
  normalize_30_rt(d, f.v[8] < 0, M);
  return s30_to_u256(d);
}
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __attribute__((noinline)) u256 modinv_rt_fn(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4,
                                                             uint32_t x5, uint32_t x6, uint32_t x7, uint32_t is_p) {
  u256 x;
  x.v[0] = x0; x.v[1] = x1; x.v[2] = x2; x.v[3] = x3; x.v[4] = x4; x.v[5] = x5; x.v[6] = x6; x.v[7] = x7;
  return modinv_rt(x, is_p != 0);
}
#endif
template <class MOD> struct mod_is_p;
template <> struct mod_is_p<ModP> { static constexpr uint32_t value = 1; };
template <> struct mod_is_p<ModN> { static constexpr uint32_t value = 0; };
// x⁻¹ mod M through the shared copy (device) / the run-time form pasted in (host tests)
template <class MOD>
HD u256 modinv_shared(const u256 &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return modinv_rt_fn(x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7], mod_is_p<MOD>::value);
#else
  return modinv_rt(x, mod_is_p<MOD>::value != 0);
#endif
}

// ---- variable-time divsteps ----------------------------------------------------------------------------
// Same 30 divsteps, but runs of even g are stripped with one count-trailing-zeros.  Control flow
// depends on the data, so this is for code where a whole wavefront works on ONE value
// (wave_fe_dev.h:modinv_wave): every branch is then wave-uniform.  Nothing here is secret —
// signatures and public keys — so timing is not a concern; the constant-time form above is used where
// lanes hold different values because it never diverges.
HD int32_t divsteps_30_var(int32_t zeta, uint32_t f0, uint32_t g0, trans2x2 &t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
  int i = 30;
  for (;;) {
    // g even ×z: z divsteps (ζ, f, g) ← (ζ − 1, f, g/2); the sentinel bit stops at the batch end
    const uint32_t m = g | (1u << i);
    const int z = __builtin_ctz(m);
    g >>= z;
    u <<= z;
    v <<= z;
    zeta -= z;
    i -= z;
    if (i == 0) break;
    // g odd: the add/subtract half of the divstep (its halving is the next round's first zero)
    if (zeta < 0) {
      zeta = -zeta - 1;
      const uint32_t nf = g, nu = q, nv = r;
      g -= f;
      q -= u;
      r -= v;
      f = nf;
      u = nu;
      v = nv;
    } else {
      g += f;
      q += u;
      r += v;
    }
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return zeta;
}
HD fe fe_inv_safegcd(const fe &a) {  // a of magnitude ≤ 32
  return fe_from_u256(modinv_shared<ModP>(fe_to_u256(a)));
}
HD sc sc_inv_safegcd(const sc &a) { return sc_from_u256(modinv_shared<ModN>(sc_canon(a))); }

// Jacobian → affine with the safegcd inverse; r.x / r.y canonical; false for infinity
HD bool jac_to_aff_fast(aff &r, const jac &p) {
  fe zi = fe_inv_safegcd(p.z);
  fe zi2 = fe_sqr(zi);
  r.x = fe_normalize(fe_mul(p.x, zi2));
  r.y = fe_normalize(fe_mul(p.y, fe_mul(zi2, zi)));
  return !(p.inf || fe_is_zero(p.z));
}

}  // namespace secp
