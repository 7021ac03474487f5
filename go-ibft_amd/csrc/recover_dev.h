// recover_dev.h — ECDSA recover -> address, the per-row body of the a2/a3 kernels.
//
// Product code (__host__ __device__ so tests can run the identical source on the CPU;
// the shipped library only runs it on gfx950).  Follows SEC 1 v2 §4.1.6 with the
// conventions of include/ibftgpu.h; the reference call sites it serves are
// /root/reference/core/ibft.go:943 (IsValidCommittedSeal) and :1128 (IsValidValidator).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keccak_dev.h"
#include "secp256k1_dev.h"
#include "modinv_dev.h"

namespace ibftk {

using secp::aff;
using secp::jac;
using secp::u256;

// Fixed-base window width for u1·G.  16-bit windows: 16 lookups + 16 mixed additions per
// signature from an 84 MB table (16 × 65 536 affine points × 80 B) that lives in HBM and is
// served from the 256 MB Infinity Cache — memory is the cheap resource on this part, VALU
// issue slots are not.  (The CPU test harness builds the same code with 8-bit windows so that
// its table takes 0.4 s instead of a minute to fill.)
#ifndef IBFT_GTAB_BITS
#define IBFT_GTAB_BITS 16
#endif
constexpr int GTAB_BITS = IBFT_GTAB_BITS;
constexpr int GTAB_WINDOWS = 256 / GTAB_BITS;
constexpr int GTAB_ENTRIES = 1 << GTAB_BITS;  // entry 0 unused
static_assert(32 % GTAB_BITS == 0, "a window must not straddle a 32-bit word");

// ---- validator table (open addressing, linear probing) ---------------------------
// slot = 6 dwords: addr[5], index+1 (0 = empty)
__host__ __device__ __forceinline__ uint32_t addr_hash(const uint32_t a[5]) {
  uint32_t h = a[0] * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + a[1] * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + a[2] * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + a[3] * 0x27D4EB2Fu;
  h = (h ^ (h >> 15)) + a[4] * 0x165667B1u;
  return h ^ (h >> 16);
}

// ---- fixed-base table: gtab[w][e] = e * 2^(B·w) * G, affine ------------------------------
// Each entry is 20 dwords: x then y, ten 26-bit limbs each (the kernels' native form, so a
// lookup is five 16-byte loads and no repacking).
constexpr int GTAB_ENTRY_DWORDS = 20;

__host__ __device__ inline void gtab_entry(int w, int e, uint32_t *out) {
  if (e == 0) {
    for (int i = 0; i < GTAB_ENTRY_DWORDS; i++) out[i] = 0;
    return;
  }
  // base = 2^(B·w) G, then e*base by double-and-add (B bits)
  jac base = secp::jac_from_aff(secp::generator());
  for (int k = 0; k < GTAB_BITS * w; k++) base = secp::jac_dbl(base);
  jac acc = secp::jac_inf();
  for (int b = GTAB_BITS - 1; b >= 0; b--) {
    acc = secp::jac_dbl(acc);
    // branch-free: the add is always computed and then selected, so no function call ever
    // executes under a partial EXEC mask (lanes of one wavefront hold different e)
    jac sum = secp::jac_add(acc, base);
    acc = secp::jac_select(((e >> b) & 1) != 0, sum, acc);
  }
  aff a;
  secp::jac_to_aff_fast(a, acc);
  for (int i = 0; i < 10; i++) {
    out[i] = a.x.n[i];
    out[10 + i] = a.y.n[i];
  }
}

// u1*G from the fixed-base table (GTAB_WINDOWS mixed adds, no doublings)
__host__ __device__ __forceinline__ jac ecmult_gen(const uint32_t *__restrict__ gtab, const u256 &k, jac acc) {
  for (int w = 0; w < GTAB_WINDOWS; w++) {
    uint32_t dgt = (k.v[(w * GTAB_BITS) >> 5] >> ((w * GTAB_BITS) & 31)) & (uint32_t)(GTAB_ENTRIES - 1);
    const uint4 *e = reinterpret_cast<const uint4 *>(gtab + (size_t)GTAB_ENTRY_DWORDS * ((size_t)w * GTAB_ENTRIES + dgt));
    uint4 t0 = e[0], t1 = e[1], t2 = e[2], t3 = e[3], t4 = e[4];
    aff q;
    q.x.n[0] = t0.x; q.x.n[1] = t0.y; q.x.n[2] = t0.z; q.x.n[3] = t0.w;
    q.x.n[4] = t1.x; q.x.n[5] = t1.y; q.x.n[6] = t1.z; q.x.n[7] = t1.w;
    q.x.n[8] = t2.x; q.x.n[9] = t2.y; q.y.n[0] = t2.z; q.y.n[1] = t2.w;
    q.y.n[2] = t3.x; q.y.n[3] = t3.y; q.y.n[4] = t3.z; q.y.n[5] = t3.w;
    q.y.n[6] = t4.x; q.y.n[7] = t4.y; q.y.n[8] = t4.z; q.y.n[9] = t4.w;
    jac sum = secp::jac_add_aff(acc, q);
    if (dgt != 0) acc = sum;
  }
  return acc;
}

// u2*R through the GLV split: u2 = k1 + k2·λ with 128-bit |k1|, |k2|, so 128 doublings instead
// of 256.  One per-lane table of j·(±R), j = 1..15; the λ-multiple of an entry is (β·X, Y, Z),
// its sign is folded into Y.  4-bit fixed windows, MSB first, both scalars share the doublings.
__host__ __device__ __forceinline__ jac ecmult_var(const aff &R, const u256 &k) {
  secp::glv_split sp = secp::sc_split_lambda(k);
  aff R1 = R;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(R.y, 1)), R.y);
  const bool flip2 = sp.neg1 != sp.neg2;
  const secp::fe beta = secp::GLV_CONST(1);
  jac tab[16];
  tab[0] = secp::jac_inf();
  tab[1] = secp::jac_from_aff(R1);
  tab[2] = secp::jac_dbl(tab[1]);
  for (int i = 3; i < 16; i++) tab[i] = secp::jac_add_aff(tab[i - 1], R1);
  jac acc = secp::jac_inf();
  for (int nib = 31; nib >= 0; nib--) {
    for (int d = 0; d < 4; d++) acc = secp::jac_dbl(acc);
    uint32_t d1 = secp::nibble(sp.k1, nib), d2 = secp::nibble(sp.k2, nib);
    jac s1 = secp::jac_add(acc, tab[d1]);
    if (d1 != 0) acc = s1;
    jac q = tab[d2];
    q.x = secp::fe_mul(q.x, beta);
    q.y = secp::l26_select(flip2, secp::fe_neg(q.y, 1), q.y);  // magnitude ≤ 2
    jac s2 = secp::jac_add(acc, q);
    if (d2 != 0) acc = s2;
  }
  return acc;
}

// Recover the signer address of (digest z, r, s, v); returns false if the signature
// is rejected (same rejection list as oracle/secp256k1.c:orc_ecrecover).
__host__ __device__ __forceinline__ bool recover_pubkey(const uint32_t *__restrict__ gtab, const u256 &z_raw,
                                                        const u256 &r, const u256 &s, uint32_t v,
                                                        uint32_t flags, uint32_t addr[5], aff &Qa) {
  bool ok = v <= 1;
  ok = ok && !secp::is_zero(r) && !secp::geq_const(r, secp::NL());
  ok = ok && !secp::is_zero(s) && !secp::geq_const(s, secp::NL());
  if (flags & 1u) {  // strict low-s: reject s > (n-1)/2, i.e. s - 1 >= (n-1)/2
    u256 sm1;
    secp::sub256(sm1, s, secp::one256());
    ok = ok && !secp::geq_const(sm1, secp::NHL());
  }
  // R = (r, y), y^2 = r^3 + 7, parity(y) = v        (r < n < p, so x = r)
  secp::fe rx = secp::fe_from_u256(r);
  secp::fe seven = secp::fe_zero();
  seven.n[0] = 7;
  secp::fe rhs = secp::fe_add(secp::fe_mul(secp::fe_sqr(rx), rx), seven);  // magnitude 2
  secp::fe y = secp::fe_sqrt_candidate(rhs);
  ok = ok && secp::fe_equal(secp::fe_sqr(y), rhs, 2);
  y = secp::fe_normalize(y);
  secp::fe yneg = secp::fe_normalize_weak(secp::fe_neg(y, 1));
  y = secp::l26_select((y.n[0] & 1u) != v, yneg, y);
  aff R;
  R.x = rx;
  R.y = y;
  // u1 = -z/r, u2 = s/r (mod n)
  secp::sc rinv = secp::sc_from_u256(secp::modinv<secp::ModN>(r));  // r is canonical, in [1, n)
  u256 u1 = secp::sc_neg_canon(secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), rinv)));
  u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(s), rinv));
  jac Q = ecmult_var(R, u2);
  Q = ecmult_gen(gtab, u1, Q);
  ok = secp::jac_to_aff_fast(Qa, Q) && ok;
  u256 qx = secp::l26_to_u256(Qa.x), qy = secp::l26_to_u256(Qa.y);
  keccak::address_from_xy(qx.v, qy.v, addr);
  return ok;
}
__host__ __device__ __forceinline__ bool recover_address(const uint32_t *__restrict__ gtab, const u256 &z_raw,
                                                         const u256 &r, const u256 &s, uint32_t v,
                                                         uint32_t flags, uint32_t addr[5]) {
  aff Qa;
  return recover_pubkey(gtab, z_raw, r, s, v, flags, addr, Qa);
}

}  // namespace ibftk
