// sign_dev.h — the signing side: one committed seal per row.
//
// Product code (__host__ __device__ like recover_dev.h, so tests can run the identical source on
// the CPU; the shipped library only runs it on gfx950).  The reference produces a committed seal
// inside Backend.BuildCommitMessage (/root/reference/core/backend.go:12-34, called from
// core/ibft.go:898-909 sendCommitMessage); a real validator signs ONE seal per round with ITS key,
// which is not a batch problem.  The batch exists for simulators, load generators and test rigs
// that play thousands of validators in one process (the shape of the reference's own
// core/consensus_test.go clusters) — that is the only use this entry point is meant for: the keys
// cross PCIe in the clear and sit in HBM for the duration of the call.
//
// The signature is plain ECDSA over secp256k1 with the low-s rule and v = parity(R.y) (flipped when
// s is negated), i.e. what ibft_verify_seals accepts under every flag.  The nonce is deterministic
// and is the one the CPU oracle uses (oracle/secp256k1.c:orc_sign):
//     k = keccak256(sk32 ‖ digest32 ‖ LE32(ctr)) mod n,   ctr = 0, 1, … until (k, r, s) are all usable
// so that a device signature can be compared byte for byte with the oracle's.  (It is NOT RFC 6979;
// nothing on the verify side depends on how k was chosen.)
#pragma once
#include "recover_dev.h"

namespace ibftk {

constexpr uint32_t SIGN_MAX_TRIES = 1024;  // same bound as the oracle; a retry has probability ≈2^-128

// keccak256 of the 68-byte nonce preimage, as a 256-bit big-endian integer
__host__ __device__ __forceinline__ u256 sign_nonce_hash(const uint8_t *sk32, const uint8_t *digest32, uint32_t ctr) {
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint64_t a = 0, b = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      a |= (uint64_t)sk32[8 * j + t] << (8 * t);
      b |= (uint64_t)digest32[8 * j + t] << (8 * t);
    }
    s[j] = a;
    s[4 + j] = b;
  }
  // bytes 64..67 = ctr (the oracle writes two counter bytes and two zeros; ctr < 1024), byte 68 = 0x01 pad
  s[8] = (uint64_t)(ctr & 0xFFFFu) | (0x01ULL << 32);
  s[16] ^= 0x8000000000000000ULL;
  keccak::f1600(s);
  u256 k;
  keccak::digest_to_limbs(s, k.v);
  return k;
}

// One row.  Returns false (and writes zeros) for a key outside [1, n) or when no nonce was usable.
// `tries` lets the caller keep a wavefront convergent: the loop body is executed by every lane until
// all lanes of the wavefront are done (on the device), lanes that finished discard the extra attempts.
__host__ __device__ __forceinline__ bool sign_row(const uint32_t *__restrict__ gtab, const uint8_t *sk32,
                                                  const uint8_t *digest32, u256 &r_out, u256 &s_out, uint32_t &v_out,
                                                  uint32_t addr[5]) {
  const u256 d = secp::from_be32(sk32);
  const bool key_ok = !secp::is_zero(d) && !secp::geq_const(d, secp::NL());
  u256 z = secp::from_be32(digest32);
  secp::sub_const_if(z, secp::geq_const(z, secp::NL()), secp::NL());  // z mod n (z < 2^256 < 2n)
  const secp::sc d_sc = secp::sc_from_u256(d);

  bool done = !key_ok, ok = false;
  r_out = secp::zero256();
  s_out = secp::zero256();
  v_out = 0;
  for (uint32_t ctr = 0; ctr < SIGN_MAX_TRIES; ctr++) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(!done) == 0) break;
#else
    if (done) break;
#endif
    u256 k = sign_nonce_hash(sk32, digest32, ctr);
    secp::sub_const_if(k, secp::geq_const(k, secp::NL()), secp::NL());
    bool good = !secp::is_zero(k);
    const u256 k_safe = secp::select(good, k, secp::one256());  // keep the inversion's precondition
    // R = k·G
    jac R = ecmult_gen(gtab, k_safe, secp::jac_inf());
    aff Ra;
    good = secp::jac_to_aff_fast(Ra, R) && good;
    const u256 rx = secp::l26_to_u256(Ra.x), ry = secp::l26_to_u256(Ra.y);
    good = good && !secp::geq_const(rx, secp::NL()) && !secp::is_zero(rx);  // r = x would need v ≥ 2: next nonce
    // s = k⁻¹ (z + r·d) mod n
    const secp::sc kinv = secp::sc_from_u256(secp::modinv<secp::ModN>(k_safe));
    const u256 rd = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(rx), d_sc));
    const u256 t = secp::add_mod_n(rd, z);
    u256 s = secp::sc_canon(secp::sc_mul(kinv, secp::sc_from_u256(t)));
    good = good && !secp::is_zero(s);
    uint32_t v = ry.v[0] & 1u;
    // low-s: s > (n−1)/2  ⇔  s − 1 ≥ (n−1)/2
    u256 sm1;
    secp::sub256(sm1, s, secp::one256());
    const bool high = !secp::is_zero(s) && secp::geq_const(sm1, secp::NHL());
    s = secp::select(high, secp::sc_neg_canon(s), s);
    v ^= high ? 1u : 0u;
    if (!done && good) {
      r_out = rx;
      s_out = s;
      v_out = v;
      ok = true;
      done = true;
    }
  }
  // the signer's address, keccak256(X‖Y)[12..32) of d·G — what the validator set is keyed by
  jac Q = ecmult_gen(gtab, secp::select(key_ok, d, secp::one256()), secp::jac_inf());
  aff Qa;
  (void)secp::jac_to_aff_fast(Qa, Q);
  const u256 qx = secp::l26_to_u256(Qa.x), qy = secp::l26_to_u256(Qa.y);
  keccak::address_from_xy(qx.v, qy.v, addr);
  if (!key_ok) {
#pragma unroll
    for (int i = 0; i < 5; i++) addr[i] = 0;
  }
  return ok;
}

}  // namespace ibftk
