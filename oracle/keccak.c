/*
 * oracle/keccak.c — CPU restatement of Keccak-256 (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker for the HIP path, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it.
 *
 * PARITY UNPINNED BY THE REFERENCE: go-ibft ships no hashing at all. The only
 * statement it makes is "Keccak hash of the proposal"
 * (/root/reference/messages/proto/messages.proto:51,61,67 and the comment at
 * /root/reference/core/ibft.go:648). The algorithm restated here is the
 * published Keccak-f[1600] permutation (FIPS-202 §3) with the ORIGINAL Keccak
 * multi-rate padding 0x01…0x80 (Ethereum's Keccak-256), rate 136 bytes,
 * capacity 512 bits — NOT NIST SHA3-256 (0x06 pad).  It is pinned by public
 * known-answer vectors in tests/test_oracle_kat.py (keccak256("") =
 * c5d24601…, keccak256("abc") = 4e03657a…), by an independent pure-Python
 * derivation in oracle/pyref.py, and — permutation, absorption and squeezing, on
 * arbitrary inputs — by Python's hashlib.sha3_256 through orc_sponge256(…, 0x06).
 */
#include "ibft_oracle.h"
#include <string.h>

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL,
    0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL,
    0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL,
    0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

/* rho rotation offsets, indexed [x + 5*y] */
static const unsigned KECCAK_RHO[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55,
                                        20, 3,  10, 43, 25, 39, 41, 45, 15,
                                        21, 8,  18, 2,  61, 56, 14};

static inline uint64_t rol64(uint64_t v, unsigned n) {
  return n ? (v << n) | (v >> (64 - n)) : v;
}

void orc_keccak_f1600(uint64_t A[25]) {
  for (int round = 0; round < 24; round++) {
    uint64_t C[5], D[5], B[25];
    /* theta */
    for (int x = 0; x < 5; x++)
      C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
    for (int x = 0; x < 5; x++)
      D[x] = C[(x + 4) % 5] ^ rol64(C[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) A[i] ^= D[i % 5];
    /* rho + pi: B[y, 2x+3y] = rot(A[x,y], r[x,y]) */
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++)
        B[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(A[x + 5 * y], KECCAK_RHO[x + 5 * y]);
    /* chi */
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++)
        A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
    /* iota */
    A[0] ^= KECCAK_RC[round];
  }
}

void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]) { orc_sponge256(in, len, 0x01, out); }

/* the sponge at rate 136 / capacity 512 with the first padding byte as a parameter: 0x01 = Keccak-256 (Ethereum), 0x06 =
 * NIST SHA3-256 — the same permutation, absorption and squeezing, so that tests can hold this code against a third-party
 * SHA3-256 (Python's hashlib) on arbitrary inputs; only the domain byte then rests on the Keccak known answers */
void orc_sponge256(const uint8_t *in, size_t len, uint8_t pad, uint8_t out[32]) {
  enum { RATE = 136 };
  uint64_t A[25];
  uint8_t block[RATE];
  memset(A, 0, sizeof A);
  while (len >= RATE) {
    for (int i = 0; i < RATE / 8; i++) {
      uint64_t w = 0;
      for (int b = 0; b < 8; b++) w |= (uint64_t)in[8 * i + b] << (8 * b);
      A[i] ^= w;
    }
    orc_keccak_f1600(A);
    in += RATE;
    len -= RATE;
  }
  memset(block, 0, RATE);
  if (len) memcpy(block, in, len);
  block[len] ^= pad;       /* 0x01: Keccak (pre-NIST) domain/pad start */
  block[RATE - 1] ^= 0x80; /* pad end */
  for (int i = 0; i < RATE / 8; i++) {
    uint64_t w = 0;
    for (int b = 0; b < 8; b++) w |= (uint64_t)block[8 * i + b] << (8 * b);
    A[i] ^= w;
  }
  orc_keccak_f1600(A);
  for (int i = 0; i < 4; i++)
    for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(A[i] >> (8 * b));
}
