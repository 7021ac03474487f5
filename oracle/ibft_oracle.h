/*
 * oracle/ibft_oracle.h — CPU oracle for the go-ibft verifier hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under go-ibft_amd/ (the product) may
 * include, link or call this.  Allowed callers: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.
 *
 * PARITY STATUS
 *   numerics (Keccak-256, secp256k1, ECDSA recover, address derivation):
 *     "parity unpinned" by the reference — go-ibft implements none of it; the
 *     three verifiers are interface methods (/root/reference/core/backend.go:37-56)
 *     whose only in-repo implementation is the byte-comparing test mock
 *     (/root/reference/core/mock_test.go:105-151).  Pinned instead by public
 *     known answers (Keccak-256 vectors; G, 2G; sk = 1, 2 -> address; go-ethereum's
 *     signature test triple msg/sig/pubkey and the ecrecover-precompile example:
 *     tests/golden/kats.json, tests/test_oracle_kat.py), a pure-Python big-int
 *     re-derivation (oracle/pyref.py) and OpenSSL libcrypto's secp256k1
 *     (oracle/openssl_xcheck.c).
 *   semantics (which verifier is called with what, nil handling, quorum rule,
 *     dedup by sender): pinned by the reference's own unit tables, replayed in
 *     tests/test_semantics_*.py.
 *
 * Conventions fixed once here and obeyed bit-for-bit by the HIP path
 * (SURVEY.md §8c "What must a CPU restatement follow"):
 *   - proposal hash  = keccak256(RawProposal ‖ BE64(Round))
 *   - seal digest    = the 32-byte proposalHash itself
 *                      (/root/reference/core/backend.go:53-55)
 *   - sender digest  = keccak256(PayloadNoSig)
 *                      (/root/reference/messages/proto/helper.go:12-27)
 *   - signature      = 65 bytes r‖s‖v, r,s big-endian in [1,n-1], v ∈ {0,1}
 *   - address        = keccak256(X‖Y)[12:32]
 *   - low-s is NOT required unless ORC_FLAG_STRICT_LOW_S is set
 */
#ifndef IBFT_ORACLE_H
#define IBFT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- keccak.c ---------------------------------------------------------- */
void orc_keccak_f1600(uint64_t state[25]);
void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
/* the same sponge with the first padding byte given: 0x01 = Keccak-256, 0x06 = NIST SHA3-256 (the form a third-party
 * implementation can check on arbitrary inputs) */
void orc_sponge256(const uint8_t *in, size_t len, uint8_t pad, uint8_t out[32]);

/* ---- secp256k1.c ------------------------------------------------------- */
#define ORC_FLAG_STRICT_LOW_S 1u

/* pub64 = X‖Y big-endian.  Returns 1 on success, 0 if sk is 0 or >= n. */
int orc_pubkey(const uint8_t sk32[32], uint8_t pub64[64]);
void orc_address(const uint8_t pub64[64], uint8_t addr20[20]);
/* Deterministic ECDSA over a 32-byte digest; nonce k = keccak256(sk‖digest‖ctr)
 * mod n (a vector GENERATOR — any valid k yields a valid signature).  Emits
 * low-s, v = parity(R.y) adjusted for the s-negation. Returns 1 on success. */
int orc_sign(const uint8_t sk32[32], const uint8_t digest32[32], uint8_t sig65[65]);
/* the same with the nonce of RFC 6979 (HMAC-SHA-256): pins the signing path against the published secp256k1 vectors */
int orc_sign_rfc6979(const uint8_t sk32[32], const uint8_t digest32[32], uint8_t sig65[65]);
/* the TUNED form of the recovery (recover_tuned.inc: endomorphism split, width-5 NAF, addition chains, binary Euclid) — for
 * bench.py's cpu_baseline leg only; the checker is orc_ecrecover, against which tests/test_oracle_tuned.py holds this one */
int orc_ecrecover_tuned(const uint8_t digest32[32], const uint8_t sig65[65], uint32_t flags, uint8_t pub64[64]);
int orc_recover_address_tuned(const uint8_t digest32[32], const uint8_t sig65[65], uint32_t flags, uint8_t addr20[20]);
void orc_sha256(const uint8_t *in, size_t len, uint8_t out[32]);
void orc_hmac_sha256(const uint8_t *key, size_t klen, const uint8_t *msg1, size_t n1, const uint8_t *msg2, size_t n2, uint8_t out[32]);
/* ECDSA public-key recovery.  Returns 1 and fills pub64 on success, else 0.
 * Rejects: r or s == 0, r or s >= n, v > 1, r^3+7 non-residue, Q at infinity,
 * and (with ORC_FLAG_STRICT_LOW_S) s > n/2. */
int orc_ecrecover(const uint8_t digest32[32], const uint8_t sig65[65], uint32_t flags,
                  uint8_t pub64[64]);
int orc_recover_address(const uint8_t digest32[32], const uint8_t sig65[65], uint32_t flags,
                        uint8_t addr20[20]);
/* small arithmetic exports used by tests to cross-check against pyref/OpenSSL */
void orc_fe_mul(const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32]);
void orc_fe_inv(const uint8_t a32[32], uint8_t out32[32]);
int orc_fe_sqrt(const uint8_t a32[32], uint8_t out32[32]); /* 1 if a is a QR */
void orc_sc_mul(const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32]);
void orc_sc_inv(const uint8_t a32[32], uint8_t out32[32]);
/* out = k1*G + k2*P (P = pub64 affine).  Returns 0 if result is infinity. */
int orc_ecmult2(const uint8_t k1[32], const uint8_t k2[32], const uint8_t p64[64],
                uint8_t out64[64]);

/* ---- ibft_oracle.c: the verifier hot path, row by row ------------------ */
/* pre_flags bits, one byte per row, set by the host when flattening messages */
#define ORC_ROW_NIL 0x01u      /* nil payload / nil seal: verdict false           */
#define ORC_ROW_BADLEN 0x02u   /* signature length != 65 or hash length != 32    */
#define ORC_ROW_HASH_BAD 0x04u /* a1 already failed: a2 is short-circuited       */

typedef struct {
  uint64_t quorum_lo, quorum_hi; /* floor(2*total/3)+1 as a 128-bit value        */
  uint64_t power_lo, power_hi;   /* Σ power over distinct valid senders ∈ set    */
  uint32_t valid_rows;           /* popcount of the verdict mask                 */
  uint32_t distinct_senders;     /* distinct member senders among valid rows     */
  uint32_t has_quorum;           /* power >= quorum                              */
  uint32_t reserved;
} orc_tally_t;

typedef struct orc_valset orc_valset_t;
/* validators: n 20-byte addresses + u64 voting power each
 * (/root/reference/core/validator_manager.go:17-20, 61-75).  Returns NULL if the
 * total voting power is zero (errVotingPowerNotCorrect). Duplicate addresses keep
 * the last power, like a Go map literal built in a loop. */
orc_valset_t *orc_valset_new(const uint8_t *addrs20, const uint64_t *power, size_t n);
void orc_valset_free(orc_valset_t *vs);
int orc_valset_index(const orc_valset_t *vs, const uint8_t addr20[20]); /* -1 if absent */
void orc_valset_quorum(const orc_valset_t *vs, uint64_t *lo, uint64_t *hi);

/* a1: IsValidProposalHash over a batch (/root/reference/core/ibft.go:858-861, 938).
 * hash_len[i] is the byte length of row i's hash field (0 = nil). verdict[i]∈{0,1}. */
void orc_verify_hashes(const uint8_t *raw, size_t raw_len, uint64_t round, const uint8_t *hash32,
                       const uint8_t *hash_len, size_t n, uint8_t *verdict);
void orc_proposal_hash(const uint8_t *raw, size_t raw_len, uint64_t round, uint8_t out[32]);

/* a2: IsValidCommittedSeal over a batch (/root/reference/core/ibft.go:931-944,
 * /root/reference/core/backend.go:53-55).  hash32 is N×32 (per-row proposalHash). */
void orc_verify_seals(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                      const uint8_t *signer20, const uint8_t *pre_flags, size_t n, uint32_t flags,
                      uint8_t *verdict);
/* same, rows sharded over nthreads pthreads (CPU baseline B2) */
void orc_verify_seals_mt(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                         const uint8_t *signer20, const uint8_t *pre_flags, size_t n,
                         uint32_t flags, uint8_t *verdict, int nthreads);

/* same rows through orc_recover_address_tuned: the cpu_baseline leg of bench.py, never the checker */
void orc_verify_seals_tuned_mt(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                               const uint8_t *signer20, const uint8_t *pre_flags, size_t n,
                               uint32_t flags, uint8_t *verdict, int nthreads);

/* a3: IsValidValidator over a batch (/root/reference/core/backend.go:41-45):
 * payload = concatenated PayloadNoSig bytes, row i = payload[off[i]..off[i+1]). */
void orc_verify_senders(const orc_valset_t *vs, const uint8_t *payload, const uint32_t *off,
                        const uint8_t *sig65, const uint8_t *from20, const uint8_t *pre_flags,
                        size_t n, uint32_t flags, uint8_t *verdict);

/* a8: HasQuorum (/root/reference/core/validator_manager.go:77-96) over the rows
 * whose verdict is 1: distinct senders, unknown senders contribute 0. */
void orc_tally(const orc_valset_t *vs, const uint8_t *sender20, const uint8_t *verdict, size_t n,
               orc_tally_t *out);

#ifdef __cplusplus
}
#endif
#endif
