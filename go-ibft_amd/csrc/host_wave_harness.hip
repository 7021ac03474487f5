// host_wave_harness.hip — TEST-ONLY: runs wave_fe_dev.h (one wavefront per signature, limbs spread
// over lanes with DPP) on the CPU through the 64-coroutine lockstep emulator in wave_emul.h, so
// tests/test_dev_wave_host.py can check the exact kernel source in this GPU-less container.
// Built with hipcc's host pass; never linked into libibftgpu.so, never a fallback.
#define IBFT_GTAB_BITS 8
#define IBFT_WAVE_EMUL 1
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "wave_fe_dev.h"

using secp::u256;

static std::vector<uint32_t> g_gtab;

namespace {

// limbs in / out: [row][10] little-endian 26-bit-radix limbs (any 32-bit values)
struct mul_job {
  const uint32_t *a, *b;  // [4][10]
  uint32_t *out;          // [4][16]  all 16 lanes of each row, so tests can check idle lanes are zero
  int op;
};
void lane_mul(void *p) {
  mul_job *j = (mul_job *)p;
  const wv::wk k = wv::wk_init();
  const uint32_t a = k.li < 10 ? j->a[k.row * 10 + k.li] : 0u;
  const uint32_t b = k.li < 10 ? j->b[k.row * 10 + k.li] : 0u;
  uint32_t r = 0;
  switch (j->op) {
    case 0: r = wv::wfe_mul(a, b, k); break;
    case 1: r = wv::wfe_weak(a, k); break;
    case 2: r = wv::wfe_neg1(a, k) + b; break;
    case 3: r = wv::wfe_neg2(a, k) + b; break;
    case 4: r = wv::wfe_neg8(a, k) + b; break;
    case 5: r = wv::wfe_sqrt_candidate(a, k); break;
    case 6: r = wv::scatter(wv::gather(a), k); break;
    case 7: r = wv::wfe_is_zero(a) ? 1u : 0u; break;
    case 8: r = wv::wfe_z_maybe_zero(wv::wfe_mul(a, b, k)) ? 1u : 0u; break;  // the additions' filter on Z3
  }
  j->out[k.row * 16 + k.li] = r;
}

struct pt_job {
  const uint32_t *p, *q;  // [4][31]: x[10] y[10] z[10] inf
  uint32_t *out;          // [4][31]
  int op;
};
wv::wjac load_pt(const uint32_t *src, const wv::wk &k) {
  const uint32_t *r = src + 31 * k.row;
  wv::wjac p;
  p.x = k.li < 10 ? r[k.li] : 0u;
  p.y = k.li < 10 ? r[10 + k.li] : 0u;
  p.z = k.li < 10 ? r[20 + k.li] : 0u;
  p.inf = r[30] != 0;
  return p;
}
void lane_pt(void *vp) {
  pt_job *j = (pt_job *)vp;
  const wv::wk k = wv::wk_init();
  wv::wjac p = load_pt(j->p, k), q = load_pt(j->q, k), r;
  switch (j->op) {
    case 0: r = wv::wjac_dbl(p, k); break;
    case 1: r = wv::wjac_add(p, q, k); break;
    case 2: r = wv::wjac_add_aff(p, wv::waff{q.x, q.y}, k); break;
    default: r = wv::wjac_add(p, wv::wjac_lane_xor(p, 16), k); break;
  }
  uint32_t *o = j->out + 31 * k.row;
  if (k.li < 10) {
    o[k.li] = r.x;
    o[10 + k.li] = r.y;
    o[20 + k.li] = r.z;
  }
  if (k.li == 0) o[30] = r.inf ? 1u : 0u;
}

struct rec_job {
  const uint8_t *hash32, *sig65;
  uint32_t flags;
  uint8_t *addr20;  // [64][20]  every lane's answer
  int *ok;          // [64]
};
void lane_rec(void *vp) {
  rec_job *j = (rec_job *)vp;
  u256 z = secp::from_be32(j->hash32), r = secp::from_be32(j->sig65), s = secp::from_be32(j->sig65 + 32);
  uint32_t a[5];
  secp::aff Q;
  bool ok = wv::recover_pubkey_wave(g_gtab.data(), z, r, s, j->sig65[64], j->flags, a, Q);
  const int l = wave_emul::lane();
  memcpy(j->addr20 + 20 * l, a, 20);
  j->ok[l] = ok ? 1 : 0;
}

// the two-wavefront form: the helper's half, then the main half, on one emulated wavefront each, `sh` in plain memory
struct rec2_job {
  const uint8_t *hash32, *sig65;
  uint32_t flags;
  uint8_t *addr20;
  int *ok;
  wv::pair_shared *sh;
};
void lane_rec2_helper(void *vp) {
  rec2_job *j = (rec2_job *)vp;
  u256 z = secp::from_be32(j->hash32), r = secp::from_be32(j->sig65), s = secp::from_be32(j->sig65 + 32);
  wv::recover_helper_wave(g_gtab.data(), z, r, s, j->sh, wv::no_sync());
}
void lane_rec2_main(void *vp) {
  rec2_job *j = (rec2_job *)vp;
  u256 z = secp::from_be32(j->hash32), r = secp::from_be32(j->sig65), s = secp::from_be32(j->sig65 + 32);
  uint32_t a[5];
  secp::aff Q;
  bool ok = wv::recover_pubkey_wave<99, true>(g_gtab.data(), z, r, s, j->sig65[64], j->flags, a, Q, j->sh, wv::no_sync());
  const int l = wave_emul::lane();
  memcpy(j->addr20 + 20 * l, a, 20);
  j->ok[l] = ok ? 1 : 0;
}

struct pre_job {
  const uint32_t *x;  // [10]
  uint32_t *out;      // [4][16] PX, then PY, PZ, yc : 4 × 64
  int ndbl;
};
void lane_prefix(void *vp) {
  pre_job *j = (pre_job *)vp;
  const wv::wk k = wv::wk_init();
  const uint32_t x = k.li < 10 ? j->x[k.li] : 0u;
  const uint32_t w = wv::wfe_weak(wv::wfe_mul(wv::wfe_sqr(x, k), x, k) + (k.li == 0 ? 7u : 0u), k);
  uint32_t PX = wv::wfe_mul(w, x, k), PY = wv::wfe_sqr(w, k), PZ, root;
  wv::prefix_and_sqrt(PX, PY, PZ, root, w, j->ndbl, k);
  const int l = wave_emul::lane();
  j->out[l] = PX;
  j->out[64 + l] = PY;
  j->out[128 + l] = PZ;
  j->out[192 + l] = root;
}

struct rec4_job {
  const uint8_t *hash32x4, *sig65x4;  // four signatures, one per row
  uint8_t *addr20;                    // [64][20]
  int *ok;                            // [64]
};
void lane_rec4(void *vp) {
  rec4_job *j = (rec4_job *)vp;
  const int l = wave_emul::lane(), row = l >> 4;
  const uint8_t *h = j->hash32x4 + 32 * row, *sg = j->sig65x4 + 65 * row;
  u256 z = secp::from_be32(h), r = secp::from_be32(sg), s = secp::from_be32(sg + 32);
  uint32_t a[5];
  secp::aff Q;
  static uint32_t wtab[wv::ROW_TAB_SLOTS * 64];  // one emulated wavefront at a time; a lane touches only its own column
  bool ok = wv::recover_pubkey_row(g_gtab.data(), z, r, s, sg[64], 0, a, Q, wtab);
  memcpy(j->addr20 + 20 * l, a, 20);
  j->ok[l] = ok ? 1 : 0;
}

// rows_finish_deferred on its own: per row P1 (on the curve), P2′ (on y² = x³ + 7t³), t and the recovery id
struct fin_job {
  const uint32_t *p1, *p2;  // [4][31] each
  const uint32_t *t;        // [4][10]
  const uint32_t *v;        // [4]
  uint32_t *out;            // [4][21]: canonical x[10], y[10], ok
};
void lane_fin(void *vp) {
  fin_job *j = (fin_job *)vp;
  const wv::wk k = wv::wk_init();
  const wv::wjac p1 = load_pt(j->p1, k), p2 = load_pt(j->p2, k);
  const uint32_t t = k.li < 10 ? j->t[k.row * 10 + k.li] : 0u;
  secp::aff Q;
  const bool ok = wv::rows_finish_deferred(Q, p1, p2, t, j->v[k.row], k);
  if (k.li == 0) {
    uint32_t *o = j->out + 21 * k.row;
    for (int i = 0; i < 10; i++) {
      o[i] = Q.x.n[i];
      o[10 + i] = Q.y.n[i];
    }
    o[20] = ok ? 1u : 0u;
  }
}

struct inv_job {
  const uint8_t *x32;
  uint8_t *out;  // [64][32]
  int which;
};
void lane_inv(void *vp) {
  inv_job *j = (inv_job *)vp;
  const wv::wk k = wv::wk_init();
  const u256 x = secp::from_be32(j->x32);
  const u256 r = j->which == 0 ? wv::modinv_wave<secp::ModP>(x, k) : wv::modinv_wave<secp::ModN>(x, k);
  secp::to_be32(j->out + 32 * wave_emul::lane(), r);
}

void lane_inv_rows(void *vp) {
  inv_job *j = (inv_job *)vp;
  const wv::wk k = wv::wk_init();
  const u256 x = secp::from_be32(j->x32 + 32 * (wave_emul::lane() >> 4));
  const u256 r = j->which == 0 ? wv::modinv_wave<secp::ModP>(x, k) : wv::modinv_wave<secp::ModN>(x, k);
  secp::to_be32(j->out + 32 * wave_emul::lane(), r);
}

struct vk_job {
  const uint32_t *qtab;
  const uint8_t *hash32, *sig65;
  uint32_t flags;
  int *ok;  // [64]
};
void lane_vk(void *vp) {
  vk_job *j = (vk_job *)vp;
  u256 z = secp::from_be32(j->hash32), r = secp::from_be32(j->sig65), s = secp::from_be32(j->sig65 + 32);
  bool ok = wv::verify_known_wave(g_gtab.data(), j->qtab, z, r, s, j->sig65[64], j->flags);
  j->ok[wave_emul::lane()] = ok ? 1 : 0;
}

}  // namespace

extern "C" {
// per-validator fixed-base table (32 windows × 256 entries × 20 dwords) for the public key X‖Y (64 BE bytes)
void wvh_build_qtab(const uint8_t *pub64, uint32_t *qtab) {
  secp::aff Q;
  Q.x = secp::fe_from_u256(secp::from_be32(pub64));
  Q.y = secp::fe_from_u256(secp::from_be32(pub64 + 32));
  for (int w = 0; w < ibftk::QTAB_WINDOWS; w++)
    ibftk::qtab_build_window(Q, w, qtab + (size_t)ibftk::GTAB_ENTRY_DWORDS * ibftk::QTAB_ENTRIES * w, true);
}
void wvh_recover4(const uint8_t *hash32x4, const uint8_t *sig65x4, uint8_t *addr64x20, int *ok64) {
  rec4_job j{hash32x4, sig65x4, addr64x20, ok64};
  wave_emul::run(lane_rec4, &j);
}
void wvh_modinv(int which, const uint8_t *x32, uint8_t *out64x32) {
  inv_job j{x32, out64x32, which};
  wave_emul::run(lane_inv, &j);
}
// four DIFFERENT values, one per DPP row (what the row-per-signature recover does): x4x32 = 4 × 32 bytes
void wvh_modinv_rows(int which, const uint8_t *x4x32, uint8_t *out64x32) {
  inv_job j{x4x32, out64x32, which};
  wave_emul::run(lane_inv_rows, &j);
}
void wvh_verify_known(const uint32_t *qtab, const uint8_t *hash32, const uint8_t *sig65, uint32_t flags, int *ok64) {
  vk_job j{qtab, hash32, sig65, flags, ok64};
  wave_emul::run(lane_vk, &j);
}
void wvh_prefix(const uint32_t *x10, int ndbl, uint32_t *out256) {
  pre_job j{x10, out256, ndbl};
  wave_emul::run(lane_prefix, &j);
}

void wvh_init_gtab(void) {
  if (!g_gtab.empty()) return;
  g_gtab.resize((size_t)ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES * ibftk::GTAB_ENTRY_DWORDS);
  for (int t = 0; t < ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES; t++)
    ibftk::gtab_entry(t / ibftk::GTAB_ENTRIES, t % ibftk::GTAB_ENTRIES,
                      g_gtab.data() + (size_t)ibftk::GTAB_ENTRY_DWORDS * t);
}
void wvh_fe_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out64) {
  mul_job j{a, b, out64, op};
  wave_emul::run(lane_mul, &j);
}
void wvh_pt_op(int op, const uint32_t *p, const uint32_t *q, uint32_t *out) {
  pt_job j{p, q, out, op};
  wave_emul::run(lane_pt, &j);
}
void wvh_rows_finish(const uint32_t *p1, const uint32_t *p2, const uint32_t *t, const uint32_t *v, uint32_t *out) {
  fin_job j{p1, p2, t, v, out};
  wave_emul::run(lane_fin, &j);
}
void wvh_recover(const uint8_t *hash32, const uint8_t *sig65, uint32_t flags, uint8_t *addr64x20, int *ok64) {
  rec_job j{hash32, sig65, flags, addr64x20, ok64};
  wave_emul::run(lane_rec, &j);
}
// two wavefronts per signature: helper (scalars, u1·G) then main, sharing `sh` — sequential here, concurrent on the device
void wvh_recover2(const uint8_t *hash32, const uint8_t *sig65, uint32_t flags, uint8_t *addr64x20, int *ok64) {
  wv::pair_shared sh;
  memset(&sh, 0xA5, sizeof sh);  // (nothing may be read before the helper wrote it)
  rec2_job j{hash32, sig65, flags, addr64x20, ok64, &sh};
  wave_emul::run(lane_rec2_helper, &j);
  wave_emul::run(lane_rec2_main, &j);
}
uint32_t wvh_neg_limb(int which, int i) { return wv::wneg_limb(which, i); }

}  // extern "C"
