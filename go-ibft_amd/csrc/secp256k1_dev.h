// secp256k1_dev.h — 256-bit field / scalar / group arithmetic for the gfx950 kernels.
//
// Product code (go-ibft_amd).  Implements what an application's Backend does
// behind go-ibft's Verifier.IsValidCommittedSeal / IsValidValidator
// (/root/reference/core/backend.go:41-45, 53-55): secp256k1 ECDSA public-key
// recovery.  The reference ships no arithmetic; conventions are fixed in
// include/ibftgpu.h and DESIGN.md.
//
// Representation: 8 × 32-bit little-endian limbs held in VGPRs, one value per lane
// (lane-per-signature kernels) — every function here is straight-line, fully
// unrolled and branch-free on data except the rare exceptional-point paths, so
// all 64 lanes of a wavefront stay converged.  Integer work only: v_mad_u64_u32 +
// carry chains; there is no dense contraction here, so no MFMA.
//
// Everything is __host__ __device__ so the same source is unit-tested on the CPU
// (tests/test_dev_arith_host.py builds csrc/host_arith_harness.hip with hipcc's
// host pass); the shipped library only ever runs it on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HD __host__ __device__ __forceinline__

namespace secp {

struct u256 {
  uint32_t v[8];
};

// p = 2^256 - 2^32 - 977
HD uint32_t P_LIMB(int i) { return i == 0 ? 0xFFFFFC2Fu : (i == 1 ? 0xFFFFFFFEu : 0xFFFFFFFFu); }
// n (group order)
HD uint32_t N_LIMB(int i) {
  const uint32_t n[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                         0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  return n[i];
}
// 2^256 - n  (129 bits: 5 limbs, top limb = 1)
HD uint32_t NC_LIMB(int i) {
  const uint32_t c[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u};
  return c[i];
}
// (n-1)/2
HD uint32_t NHALF_LIMB(int i) {
  const uint32_t h[8] = {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u,
                         0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
  return h[i];
}

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
HD u256 select(bool c, const u256 &a, const u256 &b) {  // c ? a : b
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// big-endian 32 bytes -> limbs
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

// r = a + b, returns carry
HD uint32_t add256(u256 &r, const u256 &a, const u256 &b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}
// r = a - b, returns borrow
HD uint32_t sub256(u256 &r, const u256 &a, const u256 &b) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
    r.v[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
  return (uint32_t)br;
}
template <typename LIMB>
HD bool geq_const(const u256 &a, LIMB limb) {  // a >= constant
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.v[i] - limb(i) - br;
    br = (t >> 32) & 1;
  }
  return br == 0;
}
template <typename LIMB>
HD void sub_const_if(u256 &a, bool c, LIMB limb) {  // a -= c ? constant : 0
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.v[i] - (c ? limb(i) : 0u) - br;
    a.v[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
}
template <typename LIMB>
HD void add_const_if(u256 &a, bool c, LIMB limb) {  // a += c ? constant : 0 (mod 2^256)
  uint64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    cy += (uint64_t)a.v[i] + (c ? limb(i) : 0u);
    a.v[i] = (uint32_t)cy;
    cy >>= 32;
  }
}
struct PL {
  HD uint32_t operator()(int i) const { return P_LIMB(i); }
};
struct NL {
  HD uint32_t operator()(int i) const { return N_LIMB(i); }
};
struct NHL {
  HD uint32_t operator()(int i) const { return NHALF_LIMB(i); }
};

// 8x8 limbs -> 16 limbs, operand scanning; each step a*b + r + carry < 2^64
HD void mul_wide(uint32_t r[16], const u256 &a, const u256 &b) {
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t t = (uint64_t)a.v[i] * b.v[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    r[i + 8] = carry;
  }
}
// 36 products instead of 64: off-diagonal once, doubled, plus the diagonal
HD void sqr_wide(uint32_t r[16], const u256 &a) {
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = i + 1; j < 8; j++) {
      uint64_t t = (uint64_t)a.v[i] * a.v[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    r[i + 8] = carry;
  }
  // double
  uint32_t top = 0;
#pragma unroll
  for (int i = 1; i < 16; i++) {
    uint32_t nt = r[i] >> 31;
    r[i] = (r[i] << 1) | top;
    top = nt;
  }
  // add the squares
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t sq = (uint64_t)a.v[i] * a.v[i];
    c += (uint64_t)r[2 * i] + (uint32_t)sq;
    r[2 * i] = (uint32_t)c;
    c >>= 32;
    c += (uint64_t)r[2 * i + 1] + (uint32_t)(sq >> 32);
    r[2 * i + 1] = (uint32_t)c;
    c >>= 32;
  }
}

// ---------------------------------------------------------------- field mod p
// 512 -> 256 bits using 2^256 ≡ 2^32 + 977 (mod p); result fully reduced
HD u256 fe_reduce(const uint32_t w[16]) {
  u256 t;
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c += (uint64_t)w[k] + (uint64_t)w[8 + k] * 977u + (k > 0 ? w[8 + k - 1] : 0u);
    t.v[k] = (uint32_t)c;
    c >>= 32;
  }
  uint64_t top = c + w[15];  // < 2^34
  uint64_t m = top * 977u;   // < 2^44
  c = (uint64_t)t.v[0] + (uint32_t)m;
  t.v[0] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)t.v[1] + (m >> 32) + (uint32_t)top;
  t.v[1] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)t.v[2] + (top >> 32);
  t.v[2] = (uint32_t)c;
  c >>= 32;
#pragma unroll
  for (int k = 3; k < 8; k++) {
    c += t.v[k];
    t.v[k] = (uint32_t)c;
    c >>= 32;
  }
  // a carry out means the value wrapped past 2^256 once more: add 2^32+977
  uint32_t wrap = (uint32_t)c;
  c = (uint64_t)t.v[0] + (wrap ? 977u : 0u);
  t.v[0] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)t.v[1] + wrap;
  t.v[1] = (uint32_t)c;
  c >>= 32;
#pragma unroll
  for (int k = 2; k < 8; k++) {
    c += t.v[k];
    t.v[k] = (uint32_t)c;
    c >>= 32;
  }
  sub_const_if(t, geq_const(t, PL()), PL());
  return t;
}
HD u256 fe_mul(const u256 &a, const u256 &b) {
  uint32_t w[16];
  mul_wide(w, a, b);
  return fe_reduce(w);
}
HD u256 fe_sqr(const u256 &a) {
  uint32_t w[16];
  sqr_wide(w, a);
  return fe_reduce(w);
}
HD u256 fe_add(const u256 &a, const u256 &b) {
  u256 r;
  uint32_t c = add256(r, a, b);
  sub_const_if(r, c || geq_const(r, PL()), PL());
  return r;
}
HD u256 fe_sub(const u256 &a, const u256 &b) {
  u256 r;
  uint32_t br = sub256(r, a, b);
  add_const_if(r, br != 0, PL());
  return r;
}
HD u256 fe_neg(const u256 &a) { return fe_sub(zero256(), a); }
HD u256 fe_dbl(const u256 &a) { return fe_add(a, a); }
HD u256 fe_sqr_n(u256 a, int n) {
  for (int i = 0; i < n; i++) a = fe_sqr(a);
  return a;
}
// shared prefix of the p-2 and (p+1)/4 addition chains: x223 = a^(2^223-1) etc.
struct fe_chain {
  u256 x2, x3, x22, x223;
};
HD fe_chain fe_chain_223(const u256 &a) {
  fe_chain ch;
  ch.x2 = fe_mul(fe_sqr(a), a);
  ch.x3 = fe_mul(fe_sqr(ch.x2), a);
  u256 x6 = fe_mul(fe_sqr_n(ch.x3, 3), ch.x3);
  u256 x9 = fe_mul(fe_sqr_n(x6, 3), ch.x3);
  u256 x11 = fe_mul(fe_sqr_n(x9, 2), ch.x2);
  ch.x22 = fe_mul(fe_sqr_n(x11, 11), x11);
  u256 x44 = fe_mul(fe_sqr_n(ch.x22, 22), ch.x22);
  u256 x88 = fe_mul(fe_sqr_n(x44, 44), x44);
  u256 x176 = fe_mul(fe_sqr_n(x88, 88), x88);
  u256 x220 = fe_mul(fe_sqr_n(x176, 44), x44);
  ch.x223 = fe_mul(fe_sqr_n(x220, 3), ch.x3);
  return ch;
}
// a^(p-2): exponent bits = 223 ones, 0, 22 ones, 0000101101
HD u256 fe_inv(const u256 &a) {
  fe_chain ch = fe_chain_223(a);
  u256 t = fe_mul(fe_sqr_n(ch.x223, 23), ch.x22);
  t = fe_mul(fe_sqr_n(t, 5), a);
  t = fe_mul(fe_sqr_n(t, 3), ch.x2);
  t = fe_mul(fe_sqr_n(t, 2), a);
  return t;
}
// a^((p+1)/4): exponent bits = 223 ones, 0, 22 ones, 00001100; caller checks r^2 == a
HD u256 fe_sqrt_candidate(const u256 &a) {
  fe_chain ch = fe_chain_223(a);
  u256 t = fe_mul(fe_sqr_n(ch.x223, 23), ch.x22);
  t = fe_mul(fe_sqr_n(t, 6), ch.x2);
  return fe_sqr_n(t, 2);
}

// ---------------------------------------------------------------- scalars mod n
// 512 -> 256 bits using 2^256 ≡ c (mod n), c = 2^256 - n (129 bits)
HD u256 sc_reduce(const uint32_t w[16]) {
  // pass 1: x = lo + hi*c, hi = w[8..15] (8 limbs) * c (5 limbs) -> 13 limbs
  uint32_t x[14];
#pragma unroll
  for (int i = 0; i < 14; i++) x[i] = i < 8 ? w[i] : 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      uint64_t t = (uint64_t)w[8 + i] * NC_LIMB(j) + x[i + j] + carry;
      x[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    // propagate the row carry (x may already hold data above i+5 from lo / earlier rows)
#pragma unroll
    for (int k = i + 5; k < 14; k++) {
      uint64_t t = (uint64_t)x[k] + carry;
      x[k] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
  }
  // pass 2: hi2 = x[8..13] (≤ 2^(130+1)) * c -> fold again
  uint32_t y[14];
#pragma unroll
  for (int i = 0; i < 14; i++) y[i] = i < 8 ? x[i] : 0u;
#pragma unroll
  for (int i = 0; i < 5; i++) {  // x[8..12]; x[13] is always 0 (value < 2^(256+130))
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      uint64_t t = (uint64_t)x[8 + i] * NC_LIMB(j) + y[i + j] + carry;
      y[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
#pragma unroll
    for (int k = i + 5; k < 14; k++) {
      uint64_t t = (uint64_t)y[k] + carry;
      y[k] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
  }
  // pass 3: y < 2^256 + 2^(131+129): y[8] small (fits one limb), y[9..] = 0
  u256 r;
  {
    uint32_t h = y[8];
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t t = (uint64_t)h * (j < 5 ? NC_LIMB(j) : 0u) + y[j] + carry;
      r.v[j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    // pass 4: a final wrap adds c once more (value then < 2^256 for sure)
    uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      cy += (uint64_t)r.v[j] + ((carry && j < 5) ? NC_LIMB(j) : 0u);
      r.v[j] = (uint32_t)cy;
      cy >>= 32;
    }
  }
  sub_const_if(r, geq_const(r, NL()), NL());
  sub_const_if(r, geq_const(r, NL()), NL());
  return r;
}
HD u256 sc_mul(const u256 &a, const u256 &b) {
  uint32_t w[16];
  mul_wide(w, a, b);
  return sc_reduce(w);
}
HD u256 sc_sqr(const u256 &a) {
  uint32_t w[16];
  sqr_wide(w, a);
  return sc_reduce(w);
}
HD u256 sc_neg(const u256 &a) {  // a in [0,n)
  u256 r;
  u256 nn;
#pragma unroll
  for (int i = 0; i < 8; i++) nn.v[i] = N_LIMB(i);
  sub256(r, nn, a);
  return select(is_zero(a), a, r);
}
HD u256 sc_normalize(const u256 &a) {  // any 256-bit value -> [0,n)
  u256 r = a;
  sub_const_if(r, geq_const(r, NL()), NL());
  return r;
}
HD u256 sc_sqr_n(u256 a, int n) {
  for (int i = 0; i < n; i++) a = sc_sqr(a);
  return a;
}
// a^(n-2) mod n.  n-2 = [127 ones][0] ‖ 0xBAAEDCE6AF48A03BBFD25E8CD036413F; the top
// run uses an addition chain, the low 128 bits a plain left-to-right scan.
HD u256 sc_inv(const u256 &a) {
  u256 x2 = sc_mul(sc_sqr(a), a);
  u256 x3 = sc_mul(sc_sqr(x2), a);
  u256 x6 = sc_mul(sc_sqr_n(x3, 3), x3);
  u256 x12 = sc_mul(sc_sqr_n(x6, 6), x6);
  u256 x24 = sc_mul(sc_sqr_n(x12, 12), x12);
  u256 x48 = sc_mul(sc_sqr_n(x24, 24), x24);
  u256 x96 = sc_mul(sc_sqr_n(x48, 48), x48);
  u256 x120 = sc_mul(sc_sqr_n(x96, 24), x24);
  u256 x126 = sc_mul(sc_sqr_n(x120, 6), x6);
  u256 t = sc_mul(sc_sqr(x126), a);  // x127
  t = sc_sqr(t);                     // the single 0 bit (bit 128)
  const uint32_t low[4] = {0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u};
  for (int w = 3; w >= 0; w--) {
    uint32_t bits = low[w];
    for (int b = 31; b >= 0; b--) {
      t = sc_sqr(t);
      u256 tm = sc_mul(t, a);
      t = select(((bits >> b) & 1u) != 0, tm, t);
    }
  }
  return t;
}

// ---------------------------------------------------------------- group (Jacobian, a = 0)
struct jac {
  u256 x, y, z;
  bool inf;
};
struct aff {
  u256 x, y;
};

HD jac jac_inf() {
  jac r;
  r.x = zero256();
  r.y = zero256();
  r.z = zero256();
  r.inf = true;
  return r;
}
HD jac jac_from_aff(const aff &a) {
  jac r;
  r.x = a.x;
  r.y = a.y;
  r.z = one256();
  r.inf = false;
  return r;
}
// dbl-2009-l: 2M + 5S
HD jac jac_dbl(const jac &p) {
  u256 A = fe_sqr(p.x);
  u256 B = fe_sqr(p.y);
  u256 C = fe_sqr(B);
  u256 t = fe_sqr(fe_add(p.x, B));
  t = fe_sub(fe_sub(t, A), C);
  u256 D = fe_dbl(t);
  u256 E = fe_add(fe_dbl(A), A);
  u256 F = fe_sqr(E);
  jac r;
  r.x = fe_sub(fe_sub(F, D), D);
  u256 C8 = fe_dbl(fe_dbl(fe_dbl(C)));
  r.y = fe_sub(fe_mul(E, fe_sub(D, r.x)), C8);
  r.z = fe_dbl(fe_mul(p.y, p.z));
  r.inf = p.inf || is_zero(p.y);
  return r;
}
// add-2007-bl: 11M + 5S, with the exceptional cases handled (rare, divergent)
HD jac jac_add(const jac &p, const jac &q) {
  u256 z1z1 = fe_sqr(p.z);
  u256 z2z2 = fe_sqr(q.z);
  u256 u1 = fe_mul(p.x, z2z2);
  u256 u2 = fe_mul(q.x, z1z1);
  u256 s1 = fe_mul(fe_mul(p.y, q.z), z2z2);
  u256 s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  u256 h = fe_sub(u2, u1);
  u256 rr = fe_sub(s2, s1);
  u256 i = fe_sqr(fe_dbl(h));
  u256 j = fe_mul(h, i);
  u256 r2 = fe_dbl(rr);
  u256 v = fe_mul(u1, i);
  jac r;
  r.x = fe_sub(fe_sub(fe_sub(fe_sqr(r2), j), v), v);
  r.y = fe_sub(fe_mul(r2, fe_sub(v, r.x)), fe_dbl(fe_mul(s1, j)));
  r.z = fe_mul(fe_sub(fe_sub(fe_sqr(fe_add(p.z, q.z)), z1z1), z2z2), h);
  r.inf = false;
  if (p.inf) return q;
  if (q.inf) return p;
  if (is_zero(h)) {
    if (is_zero(rr)) return jac_dbl(p);
    return jac_inf();
  }
  return r;
}
// madd-2007-bl: 7M + 4S (q affine, never infinity)
HD jac jac_add_aff(const jac &p, const aff &q) {
  u256 z1z1 = fe_sqr(p.z);
  u256 u2 = fe_mul(q.x, z1z1);
  u256 s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  u256 h = fe_sub(u2, p.x);
  u256 rr = fe_sub(s2, p.y);
  u256 hh = fe_sqr(h);
  u256 i = fe_dbl(fe_dbl(hh));
  u256 j = fe_mul(h, i);
  u256 r2 = fe_dbl(rr);
  u256 v = fe_mul(p.x, i);
  jac r;
  r.x = fe_sub(fe_sub(fe_sub(fe_sqr(r2), j), v), v);
  r.y = fe_sub(fe_mul(r2, fe_sub(v, r.x)), fe_dbl(fe_mul(p.y, j)));
  r.z = fe_sub(fe_sub(fe_sqr(fe_add(p.z, h)), z1z1), hh);
  r.inf = false;
  if (p.inf) return jac_from_aff(q);
  if (is_zero(h)) {
    if (is_zero(rr)) return jac_dbl(jac_from_aff(q));
    return jac_inf();
  }
  return r;
}
// returns false if p is infinity
HD bool jac_to_aff(aff &r, const jac &p) {
  u256 zi = fe_inv(p.z);
  u256 zi2 = fe_sqr(zi);
  r.x = fe_mul(p.x, zi2);
  r.y = fe_mul(p.y, fe_mul(zi2, zi));
  return !(p.inf || is_zero(p.z));
}

HD aff generator() {
  aff g;
  const uint32_t gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                          0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
  const uint32_t gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                          0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    g.x.v[i] = gx[i];
    g.y.v[i] = gy[i];
  }
  return g;
}

HD uint32_t nibble(const u256 &k, int idx) {  // idx 0 = least significant 4 bits
  return (k.v[idx >> 3] >> (4 * (idx & 7))) & 15u;
}

}  // namespace secp
