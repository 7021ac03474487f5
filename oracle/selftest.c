/*
 * oracle/selftest.c — stand-alone self-test of the C oracle, meant to be built with
 * -fsanitize=address,undefined (`make -C oracle asan_selftest && oracle/asan_selftest`).
 * TEST INFRASTRUCTURE ONLY.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ibft_oracle.h"

static int fails;
#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);     \
      fails++;                                                          \
    }                                                                   \
  } while (0)

static void hex(const uint8_t *b, size_t n, char *out) {
  for (size_t i = 0; i < n; i++) sprintf(out + 2 * i, "%02x", b[i]);
}

int main(void) {
  uint8_t h[32], pub[64], addr[20], sig[65], sk[32] = {0};
  char s[200];
  orc_keccak256((const uint8_t *)"", 0, h);
  hex(h, 32, s);
  CHECK(strcmp(s, "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470") == 0);
  orc_keccak256((const uint8_t *)"abc", 3, h);
  hex(h, 32, s);
  CHECK(strcmp(s, "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45") == 0);
  sk[31] = 1;
  CHECK(orc_pubkey(sk, pub) == 1);
  orc_address(pub, addr);
  hex(addr, 20, s);
  CHECK(strcmp(s, "7e5f4552091a69125d5dfcb7b8c2659029395bdf") == 0);

  /* a small Byzantine round through every batch entry point */
  enum { N = 64 };
  static uint8_t addrs[N * 20], hash32[N * 32], seal[N * 65], signer[N * 20], pre[N], verdict[N], hlen[N];
  static uint64_t power[N];
  uint8_t raw[100], H[32];
  for (int i = 0; i < 100; i++) raw[i] = (uint8_t)(i * 7);
  orc_proposal_hash(raw, sizeof raw, 3, H);
  for (int i = 0; i < N; i++) {
    uint8_t k[32] = {0};
    k[30] = (uint8_t)(i + 1);
    k[31] = 0x5a;
    CHECK(orc_pubkey(k, pub));
    orc_address(pub, addrs + 20 * i);
    memcpy(signer + 20 * i, addrs + 20 * i, 20);
    memcpy(hash32 + 32 * i, H, 32);
    hlen[i] = 32;
    CHECK(orc_sign(k, H, seal + 65 * i));
    power[i] = 1 + (uint64_t)(i % 5);
    pre[i] = 0;
    if (i % 7 == 3) seal[65 * i + 40] ^= 0x10; /* corrupt s */
    if (i % 11 == 5) pre[i] = ORC_ROW_NIL;
  }
  orc_valset_t *vs = orc_valset_new(addrs, power, N);
  CHECK(vs != NULL);
  orc_verify_seals(vs, hash32, seal, signer, pre, N, 0, verdict);
  int good = 0;
  for (int i = 0; i < N; i++) {
    int exp = !(i % 7 == 3) && !(i % 11 == 5);
    CHECK(verdict[i] == exp);
    good += verdict[i];
  }
  static uint8_t v2[N];
  orc_verify_seals_mt(vs, hash32, seal, signer, pre, N, 0, v2, 4);
  CHECK(memcmp(verdict, v2, N) == 0);
  memset(v2, 7, N);
  orc_verify_seals_tuned_mt(vs, hash32, seal, signer, pre, N, 0, v2, 3); /* the tuned recovery: same verdicts, and clean under
                                                                            the sanitizers (signed limbs, 128-bit carries) */
  CHECK(memcmp(verdict, v2, N) == 0);
  for (int i = 0; i < N; i++) {
    uint8_t p1[64], p2[64];
    const int a = orc_ecrecover(hash32 + 32 * i, seal + 65 * i, 0, p1), b = orc_ecrecover_tuned(hash32 + 32 * i, seal + 65 * i, 0, p2);
    CHECK(a == b && (!a || memcmp(p1, p2, 64) == 0));
  }
  orc_tally_t t;
  orc_tally(vs, signer, verdict, N, &t);
  CHECK((int)t.valid_rows == good && t.distinct_senders == (uint32_t)good);
  orc_verify_hashes(raw, sizeof raw, 3, hash32, hlen, N, v2);
  for (int i = 0; i < N; i++) CHECK(v2[i] == 1);
  orc_verify_hashes(raw, sizeof raw, 4, hash32, hlen, N, v2);
  for (int i = 0; i < N; i++) CHECK(v2[i] == 0);
  /* senders over a tiny payload */
  static uint8_t payload[N * 10], msig[N * 65];
  static uint32_t off[N + 1];
  for (int i = 0; i < N; i++) {
    uint8_t k[32] = {0}, d[32];
    k[30] = (uint8_t)(i + 1);
    k[31] = 0x5a;
    off[i] = 10 * i;
    memset(payload + 10 * i, i, 10);
    orc_keccak256(payload + 10 * i, 10, d);
    CHECK(orc_sign(k, d, msig + 65 * i));
  }
  off[N] = 10 * N;
  orc_verify_senders(vs, payload, off, msig, signer, NULL, N, 0, v2);
  for (int i = 0; i < N; i++) CHECK(v2[i] == 1);
  orc_valset_free(vs);
  uint64_t zero[2] = {0, 0};
  CHECK(orc_valset_new(addrs, zero, 2) == NULL);
  /* RFC 6979: SHA-256("abc") (FIPS 180-4 appendix B.1) and the published secp256k1 vector sk = 1, "Satoshi Nakamoto" */
  orc_sha256((const uint8_t *)"abc", 3, h);
  hex(h, 32, s);
  CHECK(strcmp(s, "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad") == 0);
  memset(sk, 0, 32);
  sk[31] = 1;
  orc_sha256((const uint8_t *)"Satoshi Nakamoto", 16, h);
  CHECK(orc_sign_rfc6979(sk, h, sig));
  hex(sig, 65, s);
  CHECK(strcmp(s, "934b1ea10a4b3c1757e2b0c017d0b6143ce3c9a7e6a4a49860d7a6ab210ee3d8"
                  "2442ce9d2b916064108014783e923ec36b49743e2ffa1c4496f01a512aafd9e501") == 0);
  printf(fails ? "oracle selftest: %d failure(s)\n" : "oracle selftest: ok\n", fails);
  return fails ? 1 : 0;
}
