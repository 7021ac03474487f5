// host_arith_harness.hip — TEST-ONLY: runs the device arithmetic headers
// (secp256k1_dev.h, keccak_dev.h, the __host__ __device__ parts of kernels.hip.h)
// on the CPU so tests/test_dev_arith_host.py can compare the exact kernel source
// against the oracle in this GPU-less container.  Built with hipcc's host pass;
// never linked into libibftgpu.so and never used as a fallback.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "recover_dev.h"

using secp::u256;

static std::vector<uint32_t> g_gtab;

extern "C" {

void dev_fe_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) {
  secp::to_be32(out, secp::fe_mul(secp::from_be32(a), secp::from_be32(b)));
}
void dev_fe_sqr(const uint8_t *a, uint8_t *out) { secp::to_be32(out, secp::fe_sqr(secp::from_be32(a))); }
void dev_fe_add(const uint8_t *a, const uint8_t *b, uint8_t *out) {
  secp::to_be32(out, secp::fe_add(secp::from_be32(a), secp::from_be32(b)));
}
void dev_fe_sub(const uint8_t *a, const uint8_t *b, uint8_t *out) {
  secp::to_be32(out, secp::fe_sub(secp::from_be32(a), secp::from_be32(b)));
}
void dev_fe_inv(const uint8_t *a, uint8_t *out) { secp::to_be32(out, secp::fe_inv(secp::from_be32(a))); }
int dev_fe_sqrt(const uint8_t *a, uint8_t *out) {
  u256 x = secp::from_be32(a);
  u256 y = secp::fe_sqrt_candidate(x);
  secp::to_be32(out, y);
  return secp::eq(secp::fe_sqr(y), x) ? 1 : 0;
}
void dev_sc_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) {
  secp::to_be32(out, secp::sc_mul(secp::from_be32(a), secp::from_be32(b)));
}
void dev_sc_sqr(const uint8_t *a, uint8_t *out) { secp::to_be32(out, secp::sc_sqr(secp::from_be32(a))); }
void dev_sc_inv(const uint8_t *a, uint8_t *out) { secp::to_be32(out, secp::sc_inv(secp::from_be32(a))); }

void dev_gtab_init(void) {
  if (!g_gtab.empty()) return;
  g_gtab.resize((size_t)ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES * 16);
  for (int t = 0; t < ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES; t++)
    ibftk::gtab_entry(t / ibftk::GTAB_ENTRIES, t % ibftk::GTAB_ENTRIES, g_gtab.data() + 16 * t);
}
const uint32_t *dev_gtab_ptr(void) {
  dev_gtab_init();
  return g_gtab.data();
}

// returns 1 and the 20-byte address on success
int dev_recover_address(const uint8_t *digest32, const uint8_t *sig65, uint32_t flags, uint8_t *addr20) {
  dev_gtab_init();
  uint32_t a[5];
  bool ok = ibftk::recover_address(g_gtab.data(), secp::from_be32(digest32), secp::from_be32(sig65),
                                   secp::from_be32(sig65 + 32), sig65[64], flags, a);
  memcpy(addr20, a, 20);
  return ok ? 1 : 0;
}

void dev_keccak256(const uint8_t *in, uint32_t len, uint8_t *out32) {
  uint64_t d[4];
  keccak::hash_bytes(in, len, d);
  memcpy(out32, d, 32);
}
void dev_digest_limbs_roundtrip(const uint8_t *in, uint32_t len, uint8_t *out32) {
  uint64_t d[4];
  keccak::hash_bytes(in, len, d);
  u256 z;
  keccak::digest_to_limbs(d, z.v);
  secp::to_be32(out32, z);
}
uint32_t dev_addr_hash(const uint8_t *addr20) {
  uint32_t a[5];
  memcpy(a, addr20, 20);
  return ibftk::addr_hash(a);
}
}
