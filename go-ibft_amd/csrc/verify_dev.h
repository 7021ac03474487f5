// verify_dev.h — the "warm" path: ECDSA verification against per-validator fixed-base tables.
//
// Product code.  IsValidCommittedSeal / IsValidValidator ask "was this signed by the key whose
// address is msg.From?" (/root/reference/core/backend.go:41-45, 53-55).  The cold path answers
// by recovering the key (recover_dev.h): one square root, a 128-doubling variable-base
// multiplication and two inversions — ≈676 k VALU instructions, inherently sequential.  But a
// validator set is small and stable: once a validator's public key Q has been recovered ONCE
// (and hashed to its address), every later signature (z, r, s, v) from it can be checked as
//
//     R' = (z/s)·G + (r/s)·Q ,   accept  ⇔  R' ≠ ∞  ∧  R'.x = r  ∧  parity(R'.y) = v
//
// which gives the same verdict as recover-and-compare: accept ⇔ the recovered key equals Q
// (a different key with the same address would be a Keccak collision).  Both bases are now
// FIXED, so with per-validator tables Q·e·2^(8w) (32 windows × 256 affine points = 655 KB per
// validator; 1 024 validators = 0.67 GB, 65 536 = 43 GB of the 288 GB HBM) there are no
// doublings at all: 32 + 16 table points are summed.  That sum is a reduction, so several lanes
// can share one signature: G lanes fetch their share of the points and a log2(G)-level butterfly of
// point additions finishes it (verify_known_group_kernel, kernels.hip.h); with one wavefront per
// signature the sum runs in the row layout of wave_fe_dev.h (verify_known_wave); the lane-per-
// signature form below is used when there are enough rows to fill the chip anyway.
#pragma once
#include "recover_dev.h"

namespace ibftk {

constexpr int QTAB_WINDOWS = 32;  // 8-bit windows over u2
constexpr int QTAB_ENTRIES = 256;
constexpr size_t QTAB_DWORDS_PER_VALIDATOR = (size_t)QTAB_WINDOWS * QTAB_ENTRIES * GTAB_ENTRY_DWORDS;

__host__ __device__ __forceinline__ aff load_affine(const uint32_t *__restrict__ e20) {
  const uint4 *e = reinterpret_cast<const uint4 *>(e20);
  uint4 t0 = e[0], t1 = e[1], t2 = e[2], t3 = e[3], t4 = e[4];
  aff q;
  q.x.n[0] = t0.x; q.x.n[1] = t0.y; q.x.n[2] = t0.z; q.x.n[3] = t0.w;
  q.x.n[4] = t1.x; q.x.n[5] = t1.y; q.x.n[6] = t1.z; q.x.n[7] = t1.w;
  q.x.n[8] = t2.x; q.x.n[9] = t2.y; q.y.n[0] = t2.z; q.y.n[1] = t2.w;
  q.y.n[2] = t3.x; q.y.n[3] = t3.y; q.y.n[4] = t3.z; q.y.n[5] = t3.w;
  q.y.n[6] = t4.x; q.y.n[7] = t4.y; q.y.n[8] = t4.z; q.y.n[9] = t4.w;
  return q;
}
__host__ __device__ __forceinline__ void store_affine(uint32_t *__restrict__ e20, const aff &a) {
  for (int i = 0; i < 10; i++) {
    e20[i] = a.x.n[i];
    e20[10 + i] = a.y.n[i];
  }
}

// signature range checks shared by both paths (same list as oracle/secp256k1.c:orc_ecrecover)
__host__ __device__ __forceinline__ bool sig_in_range(const u256 &r, const u256 &s, uint32_t v, uint32_t flags) {
  bool ok = v <= 1;
  ok = ok && !secp::is_zero(r) && !secp::geq_const(r, secp::NL());
  ok = ok && !secp::is_zero(s) && !secp::geq_const(s, secp::NL());
  if (flags & 1u) {
    u256 sm1;
    secp::sub256(sm1, s, secp::one256());
    ok = ok && !secp::geq_const(sm1, secp::NHL());
  }
  return ok;
}

// u1 = z/s, u2 = r/s (mod n), canonical
__host__ __device__ __forceinline__ void verify_scalars(const u256 &z_raw, const u256 &r, const u256 &s, u256 &u1,
                                                        u256 &u2) {
  secp::sc sinv = secp::sc_from_u256(secp::modinv_shared<secp::ModN>(s));
  u1 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), sinv));
  u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(r), sinv));
}

// final check on R' = u1·G + u2·Q
__host__ __device__ __forceinline__ bool verify_finish(const jac &Rp, const u256 &r, uint32_t v) {
  aff A;
  bool fin = secp::jac_to_aff_fast(A, Rp);
  secp::fe rx = secp::fe_from_u256(r);  // r < n < p: canonical limbs
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) diff |= A.x.n[i] ^ rx.n[i];
  return fin && diff == 0 && (A.y.n[0] & 1u) == v;
}

// lane-per-signature form: 32 + GTAB_WINDOWS mixed additions, no doublings
__host__ __device__ __forceinline__ bool verify_known(const uint32_t *__restrict__ gtab,
                                                      const uint32_t *__restrict__ qtab_v, const u256 &z_raw,
                                                      const u256 &r, const u256 &s, uint32_t v, uint32_t flags) {
  bool ok = sig_in_range(r, s, v, flags);
  u256 u1, u2;
  verify_scalars(z_raw, r, s, u1, u2);
  jac acc = secp::jac_inf();
  // ONE rolled loop over the 32 + GTAB_WINDOWS table points (one inlined copy of the mixed addition: secp256k1_dev.h).
  // Round 6: software-pipelined by one — the entry of step w + 1 is asked for before the addition of step w runs.  Every step
  // used to open with a dependent read of a table far beyond any cache (655 KB per validator: 43 GB at 65 536 validators;
  // ≈ 2 µs with one resident wavefront per SIMD) in front of a 4.4 µs addition: a third of this kernel's time was that wait.
  // The scalars are shift registers (the current window in the low bits of word 0): nothing is indexed by w, nothing lives
  // in the private segment (the 80 B of scratch this kernel had were u1 and u2).
  u256 k2 = u2, k1 = u1;
  uint32_t dgt = k2.v[0] & 255u;
  const uint32_t *entry = qtab_v + (size_t)GTAB_ENTRY_DWORDS * dgt;
  aff cur = load_affine(entry);
  constexpr int POINTS = QTAB_WINDOWS + GTAB_WINDOWS;
#pragma unroll 1
  for (int w = 0; w < POINTS; w++) {
    uint32_t dn = dgt;                  // (the last step re-reads its own entry)
    const int wn = w + 1;
    if (wn < QTAB_WINDOWS) {            // (wave-uniform)
      secp::shr_bits<8>(k2);
      dn = k2.v[0] & 255u;
      entry = qtab_v + (size_t)GTAB_ENTRY_DWORDS * (wn * QTAB_ENTRIES + dn);
    } else if (wn < POINTS) {
      const int g = wn - QTAB_WINDOWS;
      if (g > 0) secp::shr_bits<GTAB_BITS>(k1);
      dn = k1.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
      entry = gtab + (size_t)GTAB_ENTRY_DWORDS * ((size_t)g * GTAB_ENTRIES + dn);
    }
    const aff nxt = load_affine(entry);
    jac sum = secp::jac_add_aff_t<true>(acc, cur);
    acc = secp::jac_select(dgt != 0, sum, acc);
    cur = nxt;
    dgt = dn;
  }
  return verify_finish(acc, r, v) && ok;
}

// ---- table build: one (validator, window) per lane, 16 entries per Montgomery batch ------------
// Writes qtab[v][w][e] = e·2^(8w)·Q_v for e = 1..255 (entry 0 zeroed).  Every lane of a wavefront
// executes the same instruction stream (same w, uniform trip counts); lanes with nothing to
// build run on the generator and skip only the stores — no call sits under a partial EXEC mask.
constexpr int QTAB_CHUNK = 16;

__host__ __device__ inline void qtab_build_window(const aff &Q, int w, uint32_t *__restrict__ out, bool do_store) {
  // base = 2^(8w)·Q, affine
  jac b = secp::jac_from_aff(Q);
  for (int k = 0; k < 8 * w; k++) b = secp::jac_dbl(b);
  aff B;
  secp::jac_to_aff_fast(B, b);
  if (do_store)
    for (int i = 0; i < GTAB_ENTRY_DWORDS; i++) out[i] = 0;
  jac run = secp::jac_inf();
  for (int c = 0; c < QTAB_ENTRIES / QTAB_CHUNK; c++) {
    jac buf[QTAB_CHUNK];
    secp::fe pre[QTAB_CHUNK];  // pre[i] = Z_0·…·Z_i over the finite entries of the chunk
    secp::fe acc = secp::fe_one();
    for (int i = 0; i < QTAB_CHUNK; i++) {
      const int e = c * QTAB_CHUNK + i;
      if (e > 0) run = secp::jac_add_aff(run, B);  // e = 0 stays ∞ (uniform: e is lane-independent)
      buf[i] = run;
      secp::fe z = secp::l26_select(run.inf, secp::fe_one(), run.z);
      acc = secp::fe_mul(acc, z);
      pre[i] = acc;
    }
    secp::fe inv = secp::fe_inv_safegcd(acc);
    for (int i = QTAB_CHUNK - 1; i >= 0; i--) {
      const int e = c * QTAB_CHUNK + i;
      secp::fe z = secp::l26_select(buf[i].inf, secp::fe_one(), buf[i].z);
      secp::fe zi = i > 0 ? secp::fe_mul(inv, pre[i - 1]) : inv;
      inv = secp::fe_mul(inv, z);
      secp::fe zi2 = secp::fe_sqr(zi);
      aff a;
      a.x = secp::fe_normalize(secp::fe_mul(buf[i].x, zi2));
      a.y = secp::fe_normalize(secp::fe_mul(buf[i].y, secp::fe_mul(zi2, zi)));
      if (do_store && e > 0) store_affine(out + (size_t)GTAB_ENTRY_DWORDS * e, a);
    }
  }
}

}  // namespace ibftk
