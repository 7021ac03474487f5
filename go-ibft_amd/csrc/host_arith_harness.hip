// host_arith_harness.hip — TEST-ONLY: runs the device arithmetic headers
// (secp256k1_dev.h, keccak_dev.h, the __host__ __device__ parts of kernels.hip.h)
// on the CPU so tests/test_dev_arith_host.py can compare the exact kernel source
// against the oracle in this GPU-less container.  Built with hipcc's host pass;
// never linked into libibftgpu.so and never used as a fallback.
#define IBFT_GTAB_BITS 8  // small table for the CPU harness (see recover_dev.h)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "recover_dev.h"
#include "verify_dev.h"
#include "sign_dev.h"
#include "modinv_dev.h"
#include "wire_dev.h"

using secp::u256;

static std::vector<uint32_t> g_gtab;

extern "C" {

static secp::fe fin(const uint8_t *a) { return secp::fe_from_u256(secp::from_be32(a)); }
static void fout(uint8_t *out, const secp::fe &v) { secp::to_be32(out, secp::fe_to_u256(v)); }
static secp::sc sin_(const uint8_t *a) { return secp::sc_from_u256(secp::from_be32(a)); }
static void sout(uint8_t *out, const secp::sc &v) { secp::to_be32(out, secp::sc_canon(v)); }

void dev_fe_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) { fout(out, secp::fe_mul(fin(a), fin(b))); }
void dev_fe_sqr(const uint8_t *a, uint8_t *out) { fout(out, secp::fe_sqr(fin(a))); }
void dev_fe_add(const uint8_t *a, const uint8_t *b, uint8_t *out) { fout(out, secp::fe_add(fin(a), fin(b))); }
void dev_fe_sub(const uint8_t *a, const uint8_t *b, uint8_t *out) { fout(out, secp::fe_sub(fin(a), fin(b), 1)); }
void dev_fe_inv(const uint8_t *a, uint8_t *out) { fout(out, secp::fe_inv(fin(a))); }
int dev_fe_sqrt(const uint8_t *a, uint8_t *out) {
  secp::fe x = fin(a);
  secp::fe y = secp::fe_sqrt_candidate(x);
  fout(out, y);
  return secp::fe_equal(secp::fe_sqr(y), x, 1) ? 1 : 0;
}
// stress the lazy representation: ((a+b)*k - c)^2 * (a - b) with un-normalised intermediates
void dev_fe_lazy(const uint8_t *a, const uint8_t *b, const uint8_t *c, uint32_t k, uint8_t *out) {
  secp::fe A = fin(a), B = fin(b), Cc = fin(c);
  secp::fe t = secp::fe_add(secp::fe_mul_int(secp::fe_add(A, B), k), secp::fe_neg(Cc, 1));  // 2k + 2
  secp::fe u = secp::fe_sub(A, B, 1);                                                       // 3
  fout(out, secp::fe_mul(secp::fe_sqr(t), u));
}
// the additions' filter: Z3 = 2·z·(a − b) as madd-2007-bl forms it; bit 0 = fe_z_maybe_zero(Z3), bit 1 = fe_is_zero(Z3),
// bit 2 = the same through a squaring-shaped product and fe_normalize_weak (the other producers of a Z3)
int dev_fe_zero_filter(const uint8_t *a, const uint8_t *b, const uint8_t *z) {
  const secp::fe h = secp::fe_sub(fin(a), fin(b), 1);  // magnitude 3, ≡ 0 iff a ≡ b
  const secp::fe z3 = secp::fe_mul(secp::fe_mul_int(fin(z), 2), h);
  const secp::fe zs = secp::fe_normalize_weak(secp::fe_add(secp::fe_sqr(h), secp::fe_mul_int(z3, 3)));
  return (secp::fe_z_maybe_zero(z3) ? 1 : 0) | (secp::fe_is_zero(z3) ? 2 : 0) | (secp::fe_z_maybe_zero(zs) ? 4 : 0);
}
int dev_fe_equal(const uint8_t *a, const uint8_t *b) { return secp::fe_equal(fin(a), fin(b), 1) ? 1 : 0; }
void dev_sc_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) { sout(out, secp::sc_mul(sin_(a), sin_(b))); }
void dev_sc_sqr(const uint8_t *a, uint8_t *out) { sout(out, secp::sc_sqr(sin_(a))); }
void dev_sc_inv(const uint8_t *a, uint8_t *out) { sout(out, secp::sc_inv(sin_(a))); }

void dev_fe_inv_safegcd(const uint8_t *a, uint8_t *out) { fout(out, secp::fe_inv_safegcd(fin(a))); }
// one batch of variable-time divsteps against the constant-time one: returns 1 if (ζ, u, v, q, r) agree
int dev_divsteps_agree(int32_t zeta, uint32_t f0, uint32_t g0) {
  secp::trans2x2 a, b;
  int32_t za = secp::divsteps_30(zeta, f0, g0, a), zb = secp::divsteps_30_var(zeta, f0, g0, b);
  secp::trans2x2 c;  // the lockstep form (up to six steps cancelled per round): what every verification kernel runs
  const int32_t zc = secp::divsteps_30_lockstep(zeta, f0, g0, c);
  return za == zb && a.u == b.u && a.v == b.v && a.q == b.q && a.r == b.r &&
         za == zc && a.u == c.u && a.v == c.v && a.q == c.q && a.r == c.r;
}
void dev_sc_inv_safegcd(const uint8_t *a, uint8_t *out) { sout(out, secp::sc_inv_safegcd(sin_(a))); }
// GLV split: out = k1(32 BE) ‖ k2(32 BE), returns neg1 | neg2<<1
int dev_glv_split(const uint8_t *k, uint8_t *out64) {
  secp::glv_split s = secp::sc_split_lambda(secp::from_be32(k));
  secp::to_be32(out64, s.k1);
  secp::to_be32(out64 + 32, s.k2);
  return (s.neg1 ? 1 : 0) | (s.neg2 ? 2 : 0);
}
// out = k * P (affine, 64 BE bytes) through ecmult_var; returns 0 for infinity
int dev_ecmult_var(const uint8_t *k, const uint8_t *p64, uint8_t *out64) {
  ibftk::aff P;
  P.x = fin(p64);
  P.y = fin(p64 + 32);
  ibftk::jac q = ibftk::ecmult_var(P, secp::from_be32(k));
  ibftk::aff a;
  bool ok = secp::jac_to_aff(a, q);
  secp::to_be32(out64, secp::l26_to_u256(a.x));
  secp::to_be32(out64 + 32, secp::l26_to_u256(a.y));
  return ok ? 1 : 0;
}

// the same with the window table in "LDS" (round 5: recover_dev.h ltab) — here a heap block that stands for the columns of TPB = 3
// lanes of a workgroup, this call being lane `lane` of them (the other columns must stay untouched: returned in guard_ok)
int dev_ecmult_var_lds(const uint8_t *k, const uint8_t *p64, uint8_t *out64, int lane, int *guard_ok) {
  ibftk::aff P;
  P.x = fin(p64);
  P.y = fin(p64 + 32);
  std::vector<uint32_t> mem((size_t)ibftk::LTAB_WORDS * 3, 0xA5A5A5A5u);
  ibftk::jac q = ibftk::ecmult_var_lds<3>(P, secp::from_be32(k), mem.data() + lane);
  int ok_guard = 1;
  for (size_t i = 0; i < mem.size(); i++)
    if ((int)(i % 3) != lane && mem[i] != 0xA5A5A5A5u) ok_guard = 0;
  if (guard_ok) *guard_ok = ok_guard;
  ibftk::aff a;
  bool ok = secp::jac_to_aff(a, q);
  secp::to_be32(out64, secp::l26_to_u256(a.x));
  secp::to_be32(out64 + 32, secp::l26_to_u256(a.y));
  return ok ? 1 : 0;
}

// the same through round 1's form (unsigned 4-bit windows, 15 Jacobian multiples, full additions): the cross-check of the
// signed-window / common-Z form above
int dev_ecmult_var_v1(const uint8_t *k, const uint8_t *p64, uint8_t *out64) {
  ibftk::aff P;
  P.x = fin(p64);
  P.y = fin(p64 + 32);
  ibftk::jac q = ibftk::ecmult_var_v1(P, secp::from_be32(k));
  ibftk::aff a;
  bool ok = secp::jac_to_aff(a, q);
  secp::to_be32(out64, secp::l26_to_u256(a.x));
  secp::to_be32(out64 + 32, secp::l26_to_u256(a.y));
  return ok ? 1 : 0;
}
// signed digits of a 128-bit scalar as ecmult_var recodes them: out[i] = digit i + 8 (33 bytes)
void dev_window_digits(const uint8_t *k, uint8_t *out33) {
  const secp::u256 b = ibftk::window_bias(secp::from_be32(k));
  for (int i = 0; i < ibftk::WINDOW_DIGITS; i++) out33[i] = (uint8_t)secp::nibble(b, i);
}

// affine x of k1·G + k2·G through jac_add (via_aff = 0) or jac_add_aff (1); returns 0 for infinity
int dev_point_add_case(const uint8_t *k1, const uint8_t *k2, int via_aff, uint8_t *out32) {
  ibftk::aff g = secp::generator();
  ibftk::jac p1 = ibftk::ecmult_var(g, secp::from_be32(k1)), p2 = ibftk::ecmult_var(g, secp::from_be32(k2));
  ibftk::jac sum;
  if (!via_aff) {
    sum = secp::jac_add(p1, p2);
  } else {
    ibftk::aff a2;
    bool fin2 = secp::jac_to_aff_fast(a2, p2);
    sum = fin2 ? secp::jac_add_aff(p1, a2) : p1;
  }
  ibftk::aff af;
  bool fin = secp::jac_to_aff_fast(af, sum);
  secp::to_be32(out32, fin ? secp::l26_to_u256(af.x) : secp::zero256());
  return fin ? 1 : 0;
}

// warm path on the host: build the per-key table (cached for the last key) and verify
static std::vector<uint32_t> g_qtab;
static uint8_t g_qtab_key[64];
static bool g_qtab_valid = false;
void dev_gtab_init(void);
int dev_verify_known(const uint8_t *digest32, const uint8_t *sig65, const uint8_t *pub64, uint32_t flags) {
  dev_gtab_init();
  if (!g_qtab_valid || memcmp(g_qtab_key, pub64, 64) != 0) {
    g_qtab.assign(ibftk::QTAB_DWORDS_PER_VALIDATOR, 0);
    ibftk::aff Q;
    Q.x = fin(pub64);
    Q.y = fin(pub64 + 32);
    for (int w = 0; w < ibftk::QTAB_WINDOWS; w++)
      ibftk::qtab_build_window(Q, w, g_qtab.data() + (size_t)ibftk::GTAB_ENTRY_DWORDS * ibftk::QTAB_ENTRIES * w, true);
    memcpy(g_qtab_key, pub64, 64);
    g_qtab_valid = true;
  }
  return ibftk::verify_known(g_gtab.data(), g_qtab.data(), secp::from_be32(digest32), secp::from_be32(sig65),
                             secp::from_be32(sig65 + 32), sig65[64], flags) ? 1 : 0;
}
// table entry (w, e) of the last key: x ‖ y big-endian
void dev_qtab_entry(int w, int e, uint8_t *out64) {
  const uint32_t *p = g_qtab.data() + (size_t)ibftk::GTAB_ENTRY_DWORDS * (w * ibftk::QTAB_ENTRIES + e);
  ibftk::aff a = ibftk::load_affine(p);
  secp::to_be32(out64, secp::l26_to_u256(a.x));
  secp::to_be32(out64 + 32, secp::l26_to_u256(a.y));
}

void dev_gtab_init(void) {
  if (!g_gtab.empty()) return;
  g_gtab.resize((size_t)ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES * ibftk::GTAB_ENTRY_DWORDS);
  for (int t = 0; t < ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES; t++)
    ibftk::gtab_entry(t / ibftk::GTAB_ENTRIES, t % ibftk::GTAB_ENTRIES, g_gtab.data() + ibftk::GTAB_ENTRY_DWORDS * t);
}
const uint32_t *dev_gtab_ptr(void) {
  dev_gtab_init();
  return g_gtab.data();
}

// returns 1 and the 20-byte address on success
// the same recover with the window table in "LDS" (the lane kernel's form since round 5)
int dev_recover_address_lds(const uint8_t *digest32, const uint8_t *sig65, uint32_t flags, uint8_t *addr20) {
  std::vector<uint32_t> mem((size_t)ibftk::LTAB_WORDS * 2, 0u);
  uint32_t a[5];
  ibftk::aff Qa;
  bool ok = ibftk::recover_pubkey_with(g_gtab.data(), secp::from_be32(digest32), secp::from_be32(sig65), secp::from_be32(sig65 + 32),
                                       sig65[64], flags, a, Qa, ibftk::var_mult_lds<2>{mem.data() + 1});
  memcpy(addr20, a, 20);
  return ok ? 1 : 0;
}

// the same recover with the table in the private segment, co-Z build, one loop (round 6: the lane kernel's form beyond 65 536 rows)
int dev_recover_address_private(const uint8_t *digest32, const uint8_t *sig65, uint32_t flags, uint8_t *addr20) {
  dev_gtab_init();
  uint32_t a[5];
  ibftk::aff Qa;
  bool ok = ibftk::recover_pubkey_with(g_gtab.data(), secp::from_be32(digest32), secp::from_be32(sig65), secp::from_be32(sig65 + 32),
                                       sig65[64], flags, a, Qa, ibftk::var_mult_private<false>{});
  memcpy(addr20, a, 20);
  return ok ? 1 : 0;
}

int dev_recover_address(const uint8_t *digest32, const uint8_t *sig65, uint32_t flags, uint8_t *addr20) {
  dev_gtab_init();
  uint32_t a[5];
  bool ok = ibftk::recover_address(g_gtab.data(), secp::from_be32(digest32), secp::from_be32(sig65),
                                   secp::from_be32(sig65 + 32), sig65[64], flags, a);
  memcpy(addr20, a, 20);
  return ok ? 1 : 0;
}

// the signing row (sign_dev.h): sig65 = r ‖ s ‖ v and the signer's address; returns 0 for an unusable key
int dev_sign(const uint8_t *sk32, const uint8_t *digest32, uint8_t *sig65, uint8_t *addr20) {
  dev_gtab_init();
  u256 r, s;
  uint32_t v, a[5];
  bool ok = ibftk::sign_row(g_gtab.data(), sk32, digest32, r, s, v, a);
  secp::to_be32(sig65, r);
  secp::to_be32(sig65 + 32, s);
  sig65[64] = (uint8_t)v;
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

// §8f rank 3: the device wire walker on the CPU.  out: row_info (80 B) ‖ digest (32) ‖ sig (65) ‖ from (20)
// ‖ seal (65) ‖ pre_flag (1) = 263 bytes
void dev_wire_row(const uint8_t *m, uint32_t n, uint8_t *out263) {
  wire::row_info ri;
  wire::process_row(m, n, &ri, out263 + 80, out263 + 112, out263 + 177, out263 + 197, out263 + 262);
  memcpy(out263, &ri, 80);
}

// §8f rank 2 from bytes: the certificate tree on the CPU — the same wire:: building blocks in the order the cert_*
// kernels apply them (parse a level, count + list the nested messages, next level, …; propagate bottom-up; digests;
// hash bits).  Columns are rows_cap long; returns the number of rows, or -1 when the tree does not fit rows_cap.
// The wire buffer must carry 8 bytes of slack behind off[n] (dword reads).
int64_t dev_cert_tree(const uint8_t *wire_bytes, const uint32_t *off, uint32_t n, uint32_t rows_cap, uint8_t *nodes_out,
                      uint8_t *rows_out, uint8_t *digest32, uint8_t *sig65, uint8_t *from20, uint8_t *pre_flags,
                      uint8_t *prop_digest32, uint8_t *cls, uint8_t *hash_bits, uint8_t *self_bits) {
  if (n > rows_cap) return -1;
  std::vector<wire::node_info> nodes(rows_cap);
  std::vector<wire::row_info> rows(rows_cap);
  std::vector<uint32_t> span(2 * (size_t)rows_cap);
  for (uint32_t i = 0; i < n; i++) {
    wire::node_info nd{};
    nd.off = off[i];
    nd.len = off[i + 1] - off[i];
    nd.parent = wire::NO_PARENT;
    nd.ordinal = i;
    nodes[i] = nd;
  }
  std::vector<std::pair<uint32_t, uint32_t>> levels;
  uint32_t lo = 0, hi = n;
  for (;;) {
    for (uint32_t row = lo; row < hi; row++)
      wire::process_tree_row(wire_bytes + nodes[row].off, nodes[row].len, &rows[row], &nodes[row], &span[2 * (size_t)row],
                             digest32 + 32ull * row, sig65 + 65ull * row, from20 + 20ull * row, pre_flags + row);
    levels.push_back({lo, hi});
    // count (and check) the nested messages of every row of the level, then list them behind the level
    uint32_t next = hi;
    for (int fill = 0; fill < 2; fill++) {
      uint32_t base = hi;
      for (uint32_t row = lo; row < hi; row++) {
        wire::node_info &nd = nodes[row];
        if (!fill) nd.first_child = base;  // every row gets the running sum, like cert_scan_kernel
        if (rows[row].status != wire::STATUS_OK || !(nd.flags & wire::TREE_HAS_CERT)) continue;
        const bool pc = rows[row].payload_kind == wire::KIND_ROUND_CHANGE;
        uint32_t pos = nd.off + span[2 * (size_t)row], last = 0, count = 0;
        const uint32_t end = pos + span[2 * (size_t)row + 1];
        bool ok = true;
        while (pos < end) {
          uint32_t len;
          uint8_t role;
          if (!wire::cert_child_header(wire_bytes, end, pc, last, pos, len, role)) {
            ok = false;
            break;
          }
          if (fill) {
            if ((uint64_t)nd.first_child + count >= rows_cap) return -1;
            wire::node_info c{};
            c.off = pos;
            c.len = len;
            c.parent = row;
            c.ordinal = count;
            c.level = (uint8_t)(nd.level + 1);
            c.role = role;
            nodes[nd.first_child + count] = c;
          }
          count++;
          pos += len;
        }
        if (!fill) {
          if (!ok) {
            rows[row].status = wire::STATUS_NEEDS_HOST;
            count = 0;
          }
          nd.n_children = count;
          nd.first_child = base;
          base += count;
        }
      }
      if (!fill) {
        next = base;
        if (next > rows_cap) return -1;
        if (next == hi) break;
      }
    }
    if (next == hi) break;
    lo = hi;
    hi = next;
  }
  const uint32_t total = hi;
  for (size_t l = levels.size(); l-- > 1;)
    for (uint32_t row = levels[l].first; row < levels[l].second; row++)
      if (rows[row].status != wire::STATUS_OK && nodes[row].parent != wire::NO_PARENT) rows[nodes[row].parent].status = wire::STATUS_NEEDS_HOST;
  for (uint32_t row = 0; row < total; row++)
    wire::tree_digest_row(wire_bytes, &rows[row], &nodes[row], digest32 + 32ull * row, prop_digest32 + 32ull * row, pre_flags + row, true);
  for (uint32_t row = 0; row < total; row++) {
    bool hb, sb;
    wire::tree_compare_row(nodes.data(), rows.data(), prop_digest32, row, hb, sb, cls[row]);
    hash_bits[row] = hb;
    self_bits[row] = sb;
  }
  memcpy(nodes_out, nodes.data(), (size_t)total * sizeof(wire::node_info));
  memcpy(rows_out, rows.data(), (size_t)total * sizeof(wire::row_info));
  return total;
}
}
