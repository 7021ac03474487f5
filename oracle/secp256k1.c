/*
 * oracle/secp256k1.c — CPU restatement of secp256k1 ECDSA public-key recovery
 * (TEST INFRASTRUCTURE ONLY; see ibft_oracle.h for who may call it).
 *
 * PARITY UNPINNED BY THE REFERENCE: go-ibft contains no elliptic-curve code;
 * "recover the signature and the signer matches from address"
 * (/root/reference/core/backend.go:41-45) and "signature for proposal hash in
 * committed seal is signed by a validator" (/root/reference/core/backend.go:53-55)
 * are the whole specification.  The algorithm restated here is SEC 1 v2 §4.1.6
 * (public key recovery) on the SEC 2 curve secp256k1, Ethereum conventions
 * (65-byte r‖s‖v, v∈{0,1}, address = keccak256(X‖Y)[12:]).  Pinned by public
 * KATs (G, 2G, sk=1 → 0x7E5F4552091A69125d5DfCb7b8C2659029395Bdf), by the
 * independent pure-Python big-int derivation in oracle/pyref.py, and by
 * OpenSSL libcrypto EC_POINT arithmetic (oracle/openssl_xcheck.c).
 *
 * Representation: 4×64-bit little-endian limbs, always fully reduced.
 */
#include "ibft_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct {
  uint64_t d[4];
} u256;

static const u256 FE_P = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                           0xFFFFFFFFFFFFFFFFULL}};
static const u256 SC_N = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL,
                           0xFFFFFFFFFFFFFFFFULL}};
/* 2^256 - n */
static const uint64_t SC_C[3] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1ULL};
/* (n-1)/2, for the optional low-s rule */
static const u256 SC_HALF = {{0xDFE92F46681B20A0ULL, 0x5D576E7357A4501DULL, 0xFFFFFFFFFFFFFFFFULL,
                              0x7FFFFFFFFFFFFFFFULL}};
static const u256 G_X = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL,
                          0x79BE667EF9DCBBACULL}};
static const u256 G_Y = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL,
                          0x483ADA7726A3C465ULL}};

/* ---- 256-bit helpers ---------------------------------------------------- */
static int u256_is_zero(const u256 *a) { return (a->d[0] | a->d[1] | a->d[2] | a->d[3]) == 0; }
static int u256_cmp(const u256 *a, const u256 *b) {
  for (int i = 3; i >= 0; i--) {
    if (a->d[i] < b->d[i]) return -1;
    if (a->d[i] > b->d[i]) return 1;
  }
  return 0;
}
static uint64_t u256_add(u256 *r, const u256 *a, const u256 *b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a->d[i] + b->d[i];
    r->d[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static uint64_t u256_sub(u256 *r, const u256 *a, const u256 *b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a->d[i] - b->d[i] - borrow;
    r->d[i] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
  return borrow;
}
static void u256_from_be(u256 *r, const uint8_t b[32]) {
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int k = 0; k < 8; k++) w = (w << 8) | b[(3 - i) * 8 + k];
    r->d[i] = w;
  }
}
static void u256_to_be(uint8_t b[32], const u256 *a) {
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) b[(3 - i) * 8 + k] = (uint8_t)(a->d[i] >> (8 * (7 - k)));
}
static void mul_4x4(uint64_t r[8], const u256 *a, const u256 *b) {
  memset(r, 0, 8 * sizeof(uint64_t));
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) {
      u128 t = (u128)a->d[i] * b->d[j] + r[i + j] + carry;
      r[i + j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    r[i + 4] = carry;
  }
}

/* ---- field mod p = 2^256 - 2^32 - 977 ----------------------------------- */
static void fe_reduce512(u256 *out, const uint64_t r[8]) {
  const uint64_t K = 0x1000003D1ULL; /* 2^256 mod p */
  uint64_t t[4];
  u128 acc = 0;
  for (int i = 0; i < 4; i++) {
    acc += (u128)r[4 + i] * K + r[i];
    t[i] = (uint64_t)acc;
    acc >>= 64;
  }
  uint64_t top = (uint64_t)acc; /* < 2^34 */
  acc = (u128)top * K + t[0];
  t[0] = (uint64_t)acc;
  acc >>= 64;
  for (int i = 1; i < 4; i++) {
    acc += t[i];
    t[i] = (uint64_t)acc;
    acc >>= 64;
  }
  if (acc) { /* wrapped past 2^256 once more: add K (cannot wrap again) */
    acc = (u128)t[0] + K;
    t[0] = (uint64_t)acc;
    acc >>= 64;
    for (int i = 1; i < 4; i++) {
      acc += t[i];
      t[i] = (uint64_t)acc;
      acc >>= 64;
    }
  }
  memcpy(out->d, t, sizeof t);
  if (u256_cmp(out, &FE_P) >= 0) u256_sub(out, out, &FE_P);
}
static void fe_mul(u256 *r, const u256 *a, const u256 *b) {
  uint64_t w[8];
  mul_4x4(w, a, b);
  fe_reduce512(r, w);
}
static void fe_sqr(u256 *r, const u256 *a) { fe_mul(r, a, a); }
static void fe_add(u256 *r, const u256 *a, const u256 *b) {
  uint64_t c = u256_add(r, a, b);
  if (c || u256_cmp(r, &FE_P) >= 0) u256_sub(r, r, &FE_P);
}
static void fe_sub(u256 *r, const u256 *a, const u256 *b) {
  if (u256_sub(r, a, b)) u256_add(r, r, &FE_P);
}
static void fe_neg(u256 *r, const u256 *a) {
  if (u256_is_zero(a))
    *r = *a;
  else
    u256_sub(r, &FE_P, a);
}
static void fe_mul_small(u256 *r, const u256 *a, unsigned k) { /* k in 2..8, by additions */
  u256 acc = *a;
  for (unsigned i = 1; i < k; i++) fe_add(&acc, &acc, a);
  *r = acc;
}

typedef void (*mulfn_t)(u256 *, const u256 *, const u256 *);
/* r = base^e with fixed 4-bit windows (same code path for field and scalar) */
static void pow_w4(u256 *r, const u256 *base, const u256 *e, mulfn_t mul, const u256 *one) {
  u256 tab[16];
  tab[0] = *one;
  tab[1] = *base;
  for (int i = 2; i < 16; i++) mul(&tab[i], &tab[i - 1], base);
  u256 acc = *one;
  for (int nib = 63; nib >= 0; nib--) {
    for (int k = 0; k < 4; k++) mul(&acc, &acc, &acc);
    unsigned dgt = (unsigned)(e->d[nib / 16] >> (4 * (nib % 16))) & 15u;
    if (dgt) mul(&acc, &acc, &tab[dgt]);
  }
  *r = acc;
}
static const u256 U256_ONE = {{1, 0, 0, 0}};
static void fe_inv(u256 *r, const u256 *a) {
  u256 e = FE_P;
  e.d[0] -= 2; /* p-2; no borrow: low limb ends in ...FC2F */
  pow_w4(r, a, &e, fe_mul, &U256_ONE);
}
/* returns 1 and r = sqrt(a) if a is a quadratic residue, else 0 */
static int fe_sqrt(u256 *r, const u256 *a) {
  /* (p+1)/4 = 0x3FFFFFFF...FFFFFFFFBFFFFF0C */
  static const u256 E = {{0xFFFFFFFFBFFFFF0CULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                          0x3FFFFFFFFFFFFFFFULL}};
  u256 y, y2;
  pow_w4(&y, a, &E, fe_mul, &U256_ONE);
  fe_sqr(&y2, &y);
  if (u256_cmp(&y2, a) != 0) return 0;
  *r = y;
  return 1;
}

/* ---- scalars mod n -------------------------------------------------------- */
static void sc_reduce512(u256 *out, const uint64_t r[8]) {
  uint64_t x[8];
  memcpy(x, r, sizeof x);
  for (int iter = 0; iter < 6; iter++) {
    if ((x[4] | x[5] | x[6] | x[7]) == 0) break;
    uint64_t pr[8] = {0};
    for (int i = 0; i < 4; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < 3; j++) {
        u128 t = (u128)x[4 + i] * SC_C[j] + pr[i + j] + carry;
        pr[i + j] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
      }
      pr[i + 3] = carry;
    }
    u128 acc = 0;
    for (int i = 0; i < 8; i++) {
      acc += (u128)pr[i] + (i < 4 ? x[i] : 0);
      x[i] = (uint64_t)acc;
      acc >>= 64;
    }
  }
  memcpy(out->d, x, 4 * sizeof(uint64_t));
  while (u256_cmp(out, &SC_N) >= 0) u256_sub(out, out, &SC_N);
}
static void sc_mul(u256 *r, const u256 *a, const u256 *b) {
  uint64_t w[8];
  mul_4x4(w, a, b);
  sc_reduce512(r, w);
}
static void sc_add(u256 *r, const u256 *a, const u256 *b) {
  uint64_t c = u256_add(r, a, b);
  if (c || u256_cmp(r, &SC_N) >= 0) u256_sub(r, r, &SC_N);
}
static void sc_neg(u256 *r, const u256 *a) {
  if (u256_is_zero(a))
    *r = *a;
  else
    u256_sub(r, &SC_N, a);
}
static void sc_inv(u256 *r, const u256 *a) {
  u256 e = SC_N;
  e.d[0] -= 2;
  pow_w4(r, a, &e, sc_mul, &U256_ONE);
}
static void sc_from_be_mod(u256 *r, const uint8_t b[32]) { /* 256-bit value mod n */
  u256_from_be(r, b);
  if (u256_cmp(r, &SC_N) >= 0) u256_sub(r, r, &SC_N);
}

/* ---- group: Jacobian coordinates, a = 0 ------------------------------------ */
typedef struct {
  u256 x, y, z;
  int inf;
} jac_t;
typedef struct {
  u256 x, y;
} aff_t;

static void jac_set_inf(jac_t *r) {
  memset(r, 0, sizeof *r);
  r->inf = 1;
}
static void jac_from_aff(jac_t *r, const aff_t *a) {
  r->x = a->x;
  r->y = a->y;
  r->z = U256_ONE;
  r->inf = 0;
}
static void jac_dbl(jac_t *r, const jac_t *p) {
  if (p->inf || u256_is_zero(&p->y)) {
    jac_set_inf(r);
    return;
  }
  u256 A, B, C, D, E, F, t;
  fe_sqr(&A, &p->x);
  fe_sqr(&B, &p->y);
  fe_sqr(&C, &B);
  fe_add(&t, &p->x, &B);
  fe_sqr(&t, &t);
  fe_sub(&t, &t, &A);
  fe_sub(&t, &t, &C);
  fe_add(&D, &t, &t);
  fe_mul_small(&E, &A, 3);
  fe_sqr(&F, &E);
  u256 x3, y3, z3;
  fe_sub(&x3, &F, &D);
  fe_sub(&x3, &x3, &D);
  fe_sub(&t, &D, &x3);
  fe_mul(&y3, &E, &t);
  fe_mul_small(&t, &C, 8);
  fe_sub(&y3, &y3, &t);
  fe_mul(&z3, &p->y, &p->z);
  fe_add(&z3, &z3, &z3);
  r->x = x3;
  r->y = y3;
  r->z = z3;
  r->inf = 0;
}
static void jac_add(jac_t *r, const jac_t *p, const jac_t *q) {
  if (p->inf) {
    *r = *q;
    return;
  }
  if (q->inf) {
    *r = *p;
    return;
  }
  u256 z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  fe_sqr(&z1z1, &p->z);
  fe_sqr(&z2z2, &q->z);
  fe_mul(&u1, &p->x, &z2z2);
  fe_mul(&u2, &q->x, &z1z1);
  fe_mul(&s1, &p->y, &q->z);
  fe_mul(&s1, &s1, &z2z2);
  fe_mul(&s2, &q->y, &p->z);
  fe_mul(&s2, &s2, &z1z1);
  fe_sub(&h, &u2, &u1);
  fe_sub(&rr, &s2, &s1);
  if (u256_is_zero(&h)) {
    if (u256_is_zero(&rr))
      jac_dbl(r, p);
    else
      jac_set_inf(r);
    return;
  }
  fe_add(&i, &h, &h);
  fe_sqr(&i, &i);
  fe_mul(&j, &h, &i);
  fe_add(&rr, &rr, &rr);
  fe_mul(&v, &u1, &i);
  u256 x3, y3, z3;
  fe_sqr(&x3, &rr);
  fe_sub(&x3, &x3, &j);
  fe_sub(&x3, &x3, &v);
  fe_sub(&x3, &x3, &v);
  fe_sub(&t, &v, &x3);
  fe_mul(&y3, &rr, &t);
  fe_mul(&t, &s1, &j);
  fe_add(&t, &t, &t);
  fe_sub(&y3, &y3, &t);
  fe_add(&z3, &p->z, &q->z);
  fe_sqr(&z3, &z3);
  fe_sub(&z3, &z3, &z1z1);
  fe_sub(&z3, &z3, &z2z2);
  fe_mul(&z3, &z3, &h);
  r->x = x3;
  r->y = y3;
  r->z = z3;
  r->inf = 0;
}
static void jac_add_aff(jac_t *r, const jac_t *p, const aff_t *q) {
  jac_t qq;
  if (p->inf) {
    jac_from_aff(r, q);
    return;
  }
  u256 z1z1, u2, s2, h, hh, i, j, rr, v, t;
  fe_sqr(&z1z1, &p->z);
  fe_mul(&u2, &q->x, &z1z1);
  fe_mul(&s2, &q->y, &p->z);
  fe_mul(&s2, &s2, &z1z1);
  fe_sub(&h, &u2, &p->x);
  fe_sub(&rr, &s2, &p->y);
  if (u256_is_zero(&h)) {
    if (u256_is_zero(&rr)) {
      jac_from_aff(&qq, q);
      jac_dbl(r, &qq);
    } else
      jac_set_inf(r);
    return;
  }
  fe_sqr(&hh, &h);
  fe_mul_small(&i, &hh, 4);
  fe_mul(&j, &h, &i);
  fe_add(&rr, &rr, &rr);
  fe_mul(&v, &p->x, &i);
  u256 x3, y3, z3;
  fe_sqr(&x3, &rr);
  fe_sub(&x3, &x3, &j);
  fe_sub(&x3, &x3, &v);
  fe_sub(&x3, &x3, &v);
  fe_sub(&t, &v, &x3);
  fe_mul(&y3, &rr, &t);
  fe_mul(&t, &p->y, &j);
  fe_add(&t, &t, &t);
  fe_sub(&y3, &y3, &t);
  fe_add(&z3, &p->z, &h);
  fe_sqr(&z3, &z3);
  fe_sub(&z3, &z3, &z1z1);
  fe_sub(&z3, &z3, &hh);
  r->x = x3;
  r->y = y3;
  r->z = z3;
  r->inf = 0;
}
/* returns 0 if p is infinity */
static int jac_to_aff(aff_t *r, const jac_t *p) {
  if (p->inf || u256_is_zero(&p->z)) return 0;
  u256 zi, zi2, zi3;
  fe_inv(&zi, &p->z);
  fe_sqr(&zi2, &zi);
  fe_mul(&zi3, &zi2, &zi);
  fe_mul(&r->x, &p->x, &zi2);
  fe_mul(&r->y, &p->y, &zi3);
  return 1;
}

/* ---- fixed-base table for G: 32 windows of 8 bits, affine ------------------- */
static aff_t *g_tab; /* [32][256], entry 0 unused */
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;
static void g_tab_build(void) {
  const int W = 32, E = 256;
  jac_t *jt = (jac_t *)malloc(sizeof(jac_t) * W * E);
  g_tab = (aff_t *)calloc((size_t)W * E, sizeof(aff_t));
  aff_t g = {G_X, G_Y};
  jac_t base;
  jac_from_aff(&base, &g);
  for (int w = 0; w < W; w++) {
    jac_set_inf(&jt[w * E]);
    jt[w * E + 1] = base;
    for (int e = 2; e < E; e++) jac_add(&jt[w * E + e], &jt[w * E + e - 1], &base);
    for (int k = 0; k < 8; k++) jac_dbl(&base, &base);
  }
  /* batch-normalise with Montgomery's trick */
  u256 *pre = (u256 *)malloc(sizeof(u256) * W * E);
  u256 acc = U256_ONE;
  for (int i = 0; i < W * E; i++) {
    pre[i] = acc;
    if (!jt[i].inf) fe_mul(&acc, &acc, &jt[i].z);
  }
  u256 inv;
  fe_inv(&inv, &acc);
  for (int i = W * E - 1; i >= 0; i--) {
    if (jt[i].inf) continue;
    u256 zi, zi2, zi3;
    fe_mul(&zi, &inv, &pre[i]);
    fe_mul(&inv, &inv, &jt[i].z);
    fe_sqr(&zi2, &zi);
    fe_mul(&zi3, &zi2, &zi);
    fe_mul(&g_tab[i].x, &jt[i].x, &zi2);
    fe_mul(&g_tab[i].y, &jt[i].y, &zi3);
  }
  free(pre);
  free(jt);
}

/* r = k1*G + k2*P */
static void ecmult2(jac_t *r, const u256 *k1, const u256 *k2, const aff_t *p) {
  pthread_once(&g_tab_once, g_tab_build);
  jac_t acc;
  jac_set_inf(&acc);
  if (p && !u256_is_zero(k2)) {
    jac_t tab[16];
    jac_set_inf(&tab[0]);
    jac_from_aff(&tab[1], p);
    jac_dbl(&tab[2], &tab[1]);
    for (int i = 3; i < 16; i++) jac_add_aff(&tab[i], &tab[i - 1], p);
    for (int nib = 63; nib >= 0; nib--) {
      if (!acc.inf)
        for (int k = 0; k < 4; k++) jac_dbl(&acc, &acc);
      unsigned dgt = (unsigned)(k2->d[nib / 16] >> (4 * (nib % 16))) & 15u;
      if (dgt) jac_add(&acc, &acc, &tab[dgt]);
    }
  }
  for (int w = 0; w < 32; w++) {
    unsigned dgt = (unsigned)(k1->d[w / 8] >> (8 * (w % 8))) & 255u;
    if (dgt) jac_add_aff(&acc, &acc, &g_tab[w * 256 + dgt]);
  }
  *r = acc;
}

/* ---- public API --------------------------------------------------------------- */
int orc_pubkey(const uint8_t sk32[32], uint8_t pub64[64]) {
  u256 k;
  u256_from_be(&k, sk32);
  if (u256_is_zero(&k) || u256_cmp(&k, &SC_N) >= 0) return 0;
  jac_t q;
  aff_t a;
  u256 zero = {{0, 0, 0, 0}};
  ecmult2(&q, &k, &zero, NULL);
  if (!jac_to_aff(&a, &q)) return 0;
  u256_to_be(pub64, &a.x);
  u256_to_be(pub64 + 32, &a.y);
  return 1;
}

void orc_address(const uint8_t pub64[64], uint8_t addr20[20]) {
  uint8_t h[32];
  orc_keccak256(pub64, 64, h);
  memcpy(addr20, h + 12, 20);
}

/* the signature of digest z under key d with nonce k (0 < k < n): r = (k·G).x mod n, s = (z + r·d)/k, low-s, v = parity of R.y
 * after the low-s flip; 0 when this nonce cannot be used (r = 0, s = 0, or r would need recovery id ≥ 2) */
static int sign_with_nonce(const u256 *d, const u256 *z, const u256 *k, uint8_t sig65[65]) {
  jac_t R;
  aff_t Ra;
  u256 zero = {{0, 0, 0, 0}};
  ecmult2(&R, k, &zero, NULL);
  if (!jac_to_aff(&Ra, &R)) return 0;
  if (u256_cmp(&Ra.x, &SC_N) >= 0) return 0; /* r overflow would need v>=2: skip */
  u256 r = Ra.x;
  if (u256_is_zero(&r)) return 0;
  u256 kinv, s, t;
  sc_inv(&kinv, k);
  sc_mul(&t, &r, d);
  sc_add(&t, &t, z);
  sc_mul(&s, &kinv, &t);
  if (u256_is_zero(&s)) return 0;
  unsigned v = (unsigned)(Ra.y.d[0] & 1);
  if (u256_cmp(&s, &SC_HALF) > 0) {
    sc_neg(&s, &s);
    v ^= 1;
  }
  u256_to_be(sig65, &r);
  u256_to_be(sig65 + 32, &s);
  sig65[64] = (uint8_t)v;
  return 1;
}

int orc_sign(const uint8_t sk32[32], const uint8_t digest32[32], uint8_t sig65[65]) {
  u256 d, z;
  u256_from_be(&d, sk32);
  if (u256_is_zero(&d) || u256_cmp(&d, &SC_N) >= 0) return 0;
  sc_from_be_mod(&z, digest32);
  for (uint32_t ctr = 0; ctr < 1024; ctr++) {
    uint8_t buf[68], kh[32];
    memcpy(buf, sk32, 32);
    memcpy(buf + 32, digest32, 32);
    buf[64] = (uint8_t)ctr;
    buf[65] = (uint8_t)(ctr >> 8);
    buf[66] = 0;
    buf[67] = 0;
    orc_keccak256(buf, 68, kh);
    u256 k;
    u256_from_be(&k, kh);
    if (u256_cmp(&k, &SC_N) >= 0) u256_sub(&k, &k, &SC_N);
    if (u256_is_zero(&k)) continue;
    if (sign_with_nonce(&d, &z, &k, sig65)) return 1;
  }
  return 0;
}

/* The same signature with the nonce of RFC 6979 §3.2 (HMAC-SHA-256 DRBG, hlen = qlen = 256: bits2octets(h1) = h1 mod n) — the
 * recipe SURVEY.md §8d names for synthetic seals.  It exists to PIN the oracle's signing path (k·G, the inversion of k, the
 * arithmetic mod n) against the published RFC 6979 secp256k1 vectors (tests/golden/kats.json: private key, message digest → the
 * exact r, s of bitcoinjs-lib's / btcd's suites); the product's batch signer and the committed fixtures keep orc_sign's nonce. */
int orc_sign_rfc6979(const uint8_t sk32[32], const uint8_t digest32[32], uint8_t sig65[65]) {
  u256 d, z;
  u256_from_be(&d, sk32);
  if (u256_is_zero(&d) || u256_cmp(&d, &SC_N) >= 0) return 0;
  sc_from_be_mod(&z, digest32);
  uint8_t h1[32], V[32], K[32], buf[97];
  u256_to_be(h1, &z); /* bits2octets: the digest reduced mod n */
  memset(V, 0x01, 32);
  memset(K, 0x00, 32);
  for (int round = 0; round < 2; round++) { /* K = HMAC_K(V ‖ 0x00 / 0x01 ‖ int2octets(x) ‖ bits2octets(h1)); V = HMAC_K(V) */
    memcpy(buf, V, 32);
    buf[32] = (uint8_t)round;
    memcpy(buf + 33, sk32, 32);
    memcpy(buf + 65, h1, 32);
    orc_hmac_sha256(K, 32, buf, 97, NULL, 0, K);
    orc_hmac_sha256(K, 32, V, 32, NULL, 0, V);
  }
  for (int tries = 0; tries < 1024; tries++) {
    orc_hmac_sha256(K, 32, V, 32, NULL, 0, V); /* T = V (tlen = qlen after one block) */
    u256 k;
    u256_from_be(&k, V);
    if (!u256_is_zero(&k) && u256_cmp(&k, &SC_N) < 0 && sign_with_nonce(&d, &z, &k, sig65)) return 1;
    memcpy(buf, V, 32);
    buf[32] = 0x00;
    orc_hmac_sha256(K, 32, buf, 33, NULL, 0, K);
    orc_hmac_sha256(K, 32, V, 32, NULL, 0, V);
  }
  return 0;
}

int orc_ecrecover(const uint8_t digest32[32], const uint8_t sig65[65], uint32_t flags,
                  uint8_t pub64[64]) {
  u256 r, s, z;
  u256_from_be(&r, sig65);
  u256_from_be(&s, sig65 + 32);
  unsigned v = sig65[64];
  if (v > 1) return 0;
  if (u256_is_zero(&r) || u256_cmp(&r, &SC_N) >= 0) return 0;
  if (u256_is_zero(&s) || u256_cmp(&s, &SC_N) >= 0) return 0;
  if ((flags & ORC_FLAG_STRICT_LOW_S) && u256_cmp(&s, &SC_HALF) > 0) return 0;
  /* R = (r, y) with y^2 = r^3 + 7 and parity(y) = v   (r < n < p, so x = r) */
  aff_t R;
  u256 rhs, seven = {{7, 0, 0, 0}};
  R.x = r;
  fe_sqr(&rhs, &r);
  fe_mul(&rhs, &rhs, &r);
  fe_add(&rhs, &rhs, &seven);
  if (!fe_sqrt(&R.y, &rhs)) return 0;
  if ((R.y.d[0] & 1) != v) fe_neg(&R.y, &R.y);
  /* Q = r^-1 (s R - z G) = u1 G + u2 R,  u1 = -z r^-1,  u2 = s r^-1 */
  u256 rinv, u1, u2;
  sc_from_be_mod(&z, digest32);
  sc_inv(&rinv, &r);
  sc_mul(&u1, &z, &rinv);
  sc_neg(&u1, &u1);
  sc_mul(&u2, &s, &rinv);
  jac_t Q;
  aff_t Qa;
  ecmult2(&Q, &u1, &u2, &R);
  if (!jac_to_aff(&Qa, &Q)) return 0;
  u256_to_be(pub64, &Qa.x);
  u256_to_be(pub64 + 32, &Qa.y);
  return 1;
}

int orc_recover_address(const uint8_t digest32[32], const uint8_t sig65[65], uint32_t flags,
                        uint8_t addr20[20]) {
  uint8_t pub[64];
  if (!orc_ecrecover(digest32, sig65, flags, pub)) return 0;
  orc_address(pub, addr20);
  return 1;
}

void orc_fe_mul(const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32]) {
  u256 a, b, r;
  u256_from_be(&a, a32);
  u256_from_be(&b, b32);
  if (u256_cmp(&a, &FE_P) >= 0) u256_sub(&a, &a, &FE_P);
  if (u256_cmp(&b, &FE_P) >= 0) u256_sub(&b, &b, &FE_P);
  fe_mul(&r, &a, &b);
  u256_to_be(out32, &r);
}
void orc_fe_inv(const uint8_t a32[32], uint8_t out32[32]) {
  u256 a, r;
  u256_from_be(&a, a32);
  if (u256_cmp(&a, &FE_P) >= 0) u256_sub(&a, &a, &FE_P);
  fe_inv(&r, &a);
  u256_to_be(out32, &r);
}
int orc_fe_sqrt(const uint8_t a32[32], uint8_t out32[32]) {
  u256 a, r;
  u256_from_be(&a, a32);
  if (u256_cmp(&a, &FE_P) >= 0) u256_sub(&a, &a, &FE_P);
  if (!fe_sqrt(&r, &a)) return 0;
  u256_to_be(out32, &r);
  return 1;
}
void orc_sc_mul(const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32]) {
  u256 a, b, r;
  sc_from_be_mod(&a, a32);
  sc_from_be_mod(&b, b32);
  sc_mul(&r, &a, &b);
  u256_to_be(out32, &r);
}
void orc_sc_inv(const uint8_t a32[32], uint8_t out32[32]) {
  u256 a, r;
  sc_from_be_mod(&a, a32);
  sc_inv(&r, &a);
  u256_to_be(out32, &r);
}
int orc_ecmult2(const uint8_t k1[32], const uint8_t k2[32], const uint8_t p64[64],
                uint8_t out64[64]) {
  u256 a, b;
  aff_t p, o;
  jac_t q;
  sc_from_be_mod(&a, k1);
  sc_from_be_mod(&b, k2);
  u256_from_be(&p.x, p64);
  u256_from_be(&p.y, p64 + 32);
  ecmult2(&q, &a, &b, &p);
  if (!jac_to_aff(&o, &q)) return 0;
  u256_to_be(out64, &o.x);
  u256_to_be(out64 + 32, &o.y);
  return 1;
}

#include "recover_tuned.inc"
