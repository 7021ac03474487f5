/* oracle/sha256.c — SHA-256 (FIPS 180-4) and HMAC-SHA-256 (RFC 2104): TEST INFRASTRUCTURE, needed only by the RFC 6979 nonce
 * (oracle/secp256k1.c: orc_sign_rfc6979) through which the oracle's signing path is pinned against the published RFC 6979
 * secp256k1 vectors of tests/golden/kats.json.  The round constants below were derived (first 32 bits of the fractional parts of the
 * cube / square roots of the first 64 / 8 primes), and tests/test_oracle_kat.py checks the functions against Python's hashlib / hmac. */
#include <stdint.h>
#include <string.h>

#include "ibft_oracle.h"

static const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_block(uint32_t st[8], const uint8_t b[64]) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    const uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    const uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = st[0], bb = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; i++) {
    const uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = h + S1 + ch + K256[i] + w[i];
    const uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22);
    const uint32_t maj = (a & bb) ^ (a & c) ^ (bb & c);
    const uint32_t t2 = S0 + maj;
    h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
  }
  st[0] += a; st[1] += bb; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

/* SHA-256 of the concatenation of up to three byte strings (what HMAC needs without a streaming interface) */
static void sha256_3(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, const uint8_t *c, size_t nc, uint8_t out[32]) {
  uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  uint8_t blk[64];
  size_t fill = 0;
  const uint8_t *parts[3] = {a, b, c};
  const size_t lens[3] = {na, nb, nc};
  for (int p = 0; p < 3; p++)
    for (size_t i = 0; i < lens[p]; i++) {
      blk[fill++] = parts[p][i];
      if (fill == 64) {
        sha256_block(st, blk);
        fill = 0;
      }
    }
  const uint64_t bits = (uint64_t)(na + nb + nc) * 8;
  blk[fill++] = 0x80;
  if (fill > 56) {
    memset(blk + fill, 0, 64 - fill);
    sha256_block(st, blk);
    fill = 0;
  }
  memset(blk + fill, 0, 56 - fill);
  for (int i = 0; i < 8; i++) blk[56 + i] = (uint8_t)(bits >> (56 - 8 * i));
  sha256_block(st, blk);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24);
    out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8);
    out[4 * i + 3] = (uint8_t)st[i];
  }
}

void orc_sha256(const uint8_t *in, size_t len, uint8_t out[32]) { sha256_3(in, len, NULL, 0, NULL, 0, out); }

/* HMAC-SHA-256 over the concatenation msg1 ‖ msg2 */
void orc_hmac_sha256(const uint8_t *key, size_t klen, const uint8_t *msg1, size_t n1, const uint8_t *msg2, size_t n2, uint8_t out[32]) {
  uint8_t k0[64], ipad[64], opad[64], inner[32];
  memset(k0, 0, 64);
  if (klen > 64) orc_sha256(key, klen, k0); else memcpy(k0, key, klen);
  for (int i = 0; i < 64; i++) {
    ipad[i] = k0[i] ^ 0x36;
    opad[i] = k0[i] ^ 0x5c;
  }
  sha256_3(ipad, 64, msg1, n1, msg2, n2, inner);
  sha256_3(opad, 64, inner, 32, NULL, 0, out);
}
