/*
 * oracle/openssl_xcheck.c — third, independent derivation of ECDSA recover on
 * secp256k1 using OpenSSL libcrypto's EC_POINT/BN arithmetic (TEST INFRASTRUCTURE
 * ONLY).  Shares no code with secp256k1.c or pyref.py; used by
 * tests/test_oracle_xcheck.py to pin the C oracle because the reference itself
 * pins no numerics (oracle/ibft_oracle.h "PARITY STATUS").
 * OpenSSL 3.0.2 has no KECCAK-256 digest, so only the curve part is checked here.
 */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/ecdsa.h>
#include <openssl/obj_mac.h>
#include <stdint.h>
#include <string.h>

/* returns 1 and pub64 = X‖Y on success */
int ossl_ecrecover(const uint8_t digest32[32], const uint8_t sig65[65], uint8_t pub64[64]) {
  int ok = 0;
  EC_GROUP *grp = EC_GROUP_new_by_curve_name(NID_secp256k1);
  BN_CTX *ctx = BN_CTX_new();
  BIGNUM *r = BN_bin2bn(sig65, 32, NULL), *s = BN_bin2bn(sig65 + 32, 32, NULL);
  BIGNUM *z = BN_bin2bn(digest32, 32, NULL), *n = BN_new(), *rinv = BN_new();
  BIGNUM *u1 = BN_new(), *u2 = BN_new(), *x = BN_new(), *y = BN_new();
  EC_POINT *R = EC_POINT_new(grp), *Q = EC_POINT_new(grp);
  EC_GROUP_get_order(grp, n, ctx);
  if (sig65[64] > 1) goto done;
  if (BN_is_zero(r) || BN_is_zero(s) || BN_cmp(r, n) >= 0 || BN_cmp(s, n) >= 0) goto done;
  if (!EC_POINT_set_compressed_coordinates(grp, R, r, sig65[64], ctx)) goto done;
  BN_mod_inverse(rinv, r, n, ctx);
  BN_mod(z, z, n, ctx);
  BN_mod_mul(u1, z, rinv, n, ctx);
  BN_mod_sub(u1, n, u1, n, ctx);
  BN_mod_mul(u2, s, rinv, n, ctx);
  if (!EC_POINT_mul(grp, Q, u1, R, u2, ctx)) goto done;
  if (EC_POINT_is_at_infinity(grp, Q)) goto done;
  if (!EC_POINT_get_affine_coordinates(grp, Q, x, y, ctx)) goto done;
  memset(pub64, 0, 64);
  BN_bn2binpad(x, pub64, 32);
  BN_bn2binpad(y, pub64 + 32, 32);
  ok = 1;
done:
  BN_free(r); BN_free(s); BN_free(z); BN_free(n); BN_free(rinv);
  BN_free(u1); BN_free(u2); BN_free(x); BN_free(y);
  EC_POINT_free(R); EC_POINT_free(Q); BN_CTX_free(ctx); EC_GROUP_free(grp);
  return ok;
}

/* standard ECDSA verify of (r,s) over digest against pub64; 1 = valid */
int ossl_verify(const uint8_t digest32[32], const uint8_t sig65[65], const uint8_t pub64[64]) {
  int ok = 0;
  EC_KEY *key = EC_KEY_new_by_curve_name(NID_secp256k1);
  BIGNUM *x = BN_bin2bn(pub64, 32, NULL), *y = BN_bin2bn(pub64 + 32, 32, NULL);
  ECDSA_SIG *sig = ECDSA_SIG_new();
  if (EC_KEY_set_public_key_affine_coordinates(key, x, y) == 1) {
    ECDSA_SIG_set0(sig, BN_bin2bn(sig65, 32, NULL), BN_bin2bn(sig65 + 32, 32, NULL));
    ok = ECDSA_do_verify(digest32, 32, sig, key) == 1;
  }
  ECDSA_SIG_free(sig); BN_free(x); BN_free(y); EC_KEY_free(key);
  return ok;
}

/* pub64 = sk*G */
int ossl_pubkey(const uint8_t sk32[32], uint8_t pub64[64]) {
  int ok = 0;
  EC_GROUP *grp = EC_GROUP_new_by_curve_name(NID_secp256k1);
  BN_CTX *ctx = BN_CTX_new();
  BIGNUM *k = BN_bin2bn(sk32, 32, NULL), *x = BN_new(), *y = BN_new();
  EC_POINT *Q = EC_POINT_new(grp);
  if (EC_POINT_mul(grp, Q, k, NULL, NULL, ctx) && !EC_POINT_is_at_infinity(grp, Q) &&
      EC_POINT_get_affine_coordinates(grp, Q, x, y, ctx)) {
    BN_bn2binpad(x, pub64, 32);
    BN_bn2binpad(y, pub64 + 32, 32);
    ok = 1;
  }
  BN_free(k); BN_free(x); BN_free(y); EC_POINT_free(Q); BN_CTX_free(ctx); EC_GROUP_free(grp);
  return ok;
}
