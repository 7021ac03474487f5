// recover_dev.h — ECDSA recover -> address, the per-row body of the a2/a3 kernels.
//
// Product code (__host__ __device__ so tests can run the identical source on the CPU;
// the shipped library only runs it on gfx950).  Follows SEC 1 v2 §4.1.6 with the
// conventions of include/ibftgpu.h; the reference call sites it serves are
// /root/reference/core/ibft.go:943 (IsValidCommittedSeal) and :1128 (IsValidValidator).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keccak_dev.h"
#include "secp256k1_dev.h"
#include "modinv_dev.h"

namespace ibftk {

using secp::aff;
using secp::jac;
using secp::u256;

// Fixed-base window width for u1·G.  16-bit windows: 16 lookups + 16 mixed additions per
// signature from an 84 MB table (16 × 65 536 affine points × 80 B) that lives in HBM and is
// served from the 256 MB Infinity Cache — memory is the cheap resource on this part, VALU
// issue slots are not.  (The CPU test harness builds the same code with 8-bit windows so that
// its table takes 0.4 s instead of a minute to fill.)
#ifndef IBFT_GTAB_BITS
#define IBFT_GTAB_BITS 16
#endif
constexpr int GTAB_BITS = IBFT_GTAB_BITS;
constexpr int GTAB_WINDOWS = 256 / GTAB_BITS;
constexpr int GTAB_ENTRIES = 1 << GTAB_BITS;  // entry 0 unused
static_assert(32 % GTAB_BITS == 0, "a window must not straddle a 32-bit word");

// ---- validator table (open addressing, linear probing) ---------------------------
// slot = 6 dwords: addr[5], index+1 (0 = empty)
__host__ __device__ __forceinline__ uint32_t addr_hash(const uint32_t a[5]) {
  uint32_t h = a[0] * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + a[1] * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + a[2] * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + a[3] * 0x27D4EB2Fu;
  h = (h ^ (h >> 15)) + a[4] * 0x165667B1u;
  return h ^ (h >> 16);
}

// ---- fixed-base table: gtab[w][e] = e * 2^(B·w) * G, affine ------------------------------
// Each entry is 20 dwords: x then y, ten 26-bit limbs each (the kernels' native form, so a
// lookup is five 16-byte loads and no repacking).
constexpr int GTAB_ENTRY_DWORDS = 20;

__host__ __device__ inline void gtab_entry(int w, int e, uint32_t *out) {
  if (e == 0) {
    for (int i = 0; i < GTAB_ENTRY_DWORDS; i++) out[i] = 0;
    return;
  }
  // base = 2^(B·w) G, then e*base by double-and-add (B bits)
  jac base = secp::jac_from_aff(secp::generator());
  for (int k = 0; k < GTAB_BITS * w; k++) base = secp::jac_dbl(base);
  jac acc = secp::jac_inf();
  for (int b = GTAB_BITS - 1; b >= 0; b--) {
    acc = secp::jac_dbl(acc);
    // branch-free: the add is always computed and then selected, so no function call ever
    // executes under a partial EXEC mask (lanes of one wavefront hold different e)
    jac sum = secp::jac_add(acc, base);
    acc = secp::jac_select(((e >> b) & 1) != 0, sum, acc);
  }
  aff a;
  secp::jac_to_aff_fast(a, acc);
  for (int i = 0; i < 10; i++) {
    out[i] = a.x.n[i];
    out[10 + i] = a.y.n[i];
  }
}

// u1*G from the fixed-base table (GTAB_WINDOWS mixed adds, no doublings); one inlined copy of the mixed addition
// in a rolled loop (secp256k1_dev.h: the INL code shape).
// Round 6: (1) software-pipelined by one — entry w + 1 is asked for before the addition of entry w runs: a dependent read of
// the 84 MB table takes ≈ 1.8 µs with one resident wavefront per SIMD and used to be waited out sixteen times per signature
// (the addition itself takes 4.4 µs in this layout and now covers it); (2) the scalar is a shift register, its current
// window in the low bits of word 0 — nothing is indexed, nothing lives in the private segment.
struct gtab_raw {
  uint4 t0, t1, t2, t3, t4;
};
__host__ __device__ __forceinline__ gtab_raw gtab_load(const uint32_t *__restrict__ gtab, int w, uint32_t dgt) {
  const uint4 *e = reinterpret_cast<const uint4 *>(gtab + (size_t)GTAB_ENTRY_DWORDS * ((size_t)w * GTAB_ENTRIES + dgt));
  return gtab_raw{e[0], e[1], e[2], e[3], e[4]};
}
__host__ __device__ __forceinline__ aff gtab_point(const gtab_raw &g) {
  aff q;
  q.x.n[0] = g.t0.x; q.x.n[1] = g.t0.y; q.x.n[2] = g.t0.z; q.x.n[3] = g.t0.w;
  q.x.n[4] = g.t1.x; q.x.n[5] = g.t1.y; q.x.n[6] = g.t1.z; q.x.n[7] = g.t1.w;
  q.x.n[8] = g.t2.x; q.x.n[9] = g.t2.y; q.y.n[0] = g.t2.z; q.y.n[1] = g.t2.w;
  q.y.n[2] = g.t3.x; q.y.n[3] = g.t3.y; q.y.n[4] = g.t3.z; q.y.n[5] = g.t3.w;
  q.y.n[6] = g.t4.x; q.y.n[7] = g.t4.y; q.y.n[8] = g.t4.z; q.y.n[9] = g.t4.w;
  return q;
}
__host__ __device__ __forceinline__ jac ecmult_gen(const uint32_t *__restrict__ gtab, const u256 &k, jac acc) {
  u256 kk = k;
  uint32_t dgt = kk.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
  gtab_raw cur = gtab_load(gtab, 0, dgt);
#pragma unroll 1
  for (int w = 0; w < GTAB_WINDOWS; w++) {
    secp::shr_bits<GTAB_BITS>(kk);
    const bool last = w + 1 == GTAB_WINDOWS;
    const uint32_t dn = last ? dgt : kk.v[0] & (uint32_t)(GTAB_ENTRIES - 1);  // (the last step re-reads its own entry)
    const gtab_raw nxt = gtab_load(gtab, last ? w : w + 1, dn);
    const jac sum = secp::jac_add_aff_t<true>(acc, gtab_point(cur));
    acc = secp::jac_select(dgt != 0, sum, acc);
    cur = nxt;
    dgt = dn;
  }
  return acc;
}

// ---- u2·R: GLV split, signed radix-16 windows, a window table with ONE common Z (round 4) -----------------------
// u2 = k1 + k2·λ with 128-bit |k1|, |k2| (128 doublings instead of 256).  Each half is recoded into 33 signed digits
// e_i ∈ [−8, 7] without a carry chain: with C = Σ_{i<33} 8·16^i, digit i of k is nibble_i(k + C) − 8 — so the table
// holds the multiples 1…8 of ±R only (the sign of a digit is a negation of Y) instead of 1…15.
// The eight multiples are brought to ONE common Z ("effective affine"): with Zc = z₂·…·z₈ and s_i = Zc / z_i
// (prefix / suffix products, no inversion) entry i becomes (x_i·s_i², y_i·s_i³) — the AFFINE point of i·R on the
// isomorphic curve y² = x³ + 7·Zc⁶.  The a = 0 formulas never read the curve constant, so the main loop runs on that
// curve with MIXED additions (7M + 4S instead of 11M + 5S, 66 times), the λ-multiples are the same entries with X·β
// (computed once per entry instead of once per addition), and Zc is multiplied into the accumulator's Z once at the end.
// Round 1's form (15 Jacobian multiples, full additions, β per addition) is kept below for the CPU cross-check.
struct wtab {
  secp::fe x[9], y[9], bx[9];  // entries 1…8 (0 unused); bx = β·x
  secp::fe zc;
};
__host__ __device__ __forceinline__ void ecmult_table(const aff &R1, wtab &t) {
  jac m[9];
  m[1] = secp::jac_from_aff(R1);
  m[2] = secp::jac_dbl(m[1]);
  m[3] = secp::jac_add_aff(m[2], R1);
  m[4] = secp::jac_dbl(m[2]);
  m[5] = secp::jac_add_aff(m[4], R1);
  m[6] = secp::jac_dbl(m[3]);
  m[7] = secp::jac_add_aff(m[6], R1);
  m[8] = secp::jac_dbl(m[4]);
  secp::fe pre[9], suf[10];  // pre[i] = z₂·…·z_i, suf[i] = z_i·…·z₈
  pre[1] = secp::fe_one();
  pre[2] = m[2].z;
#pragma unroll 1
  for (int i = 3; i <= 8; i++) pre[i] = secp::fe_mul(pre[i - 1], m[i].z);
  suf[9] = secp::fe_one();
  suf[8] = m[8].z;
#pragma unroll 1
  for (int i = 7; i >= 2; i--) suf[i] = secp::fe_mul(suf[i + 1], m[i].z);
  t.zc = pre[8];
  const secp::fe beta = secp::GLV_CONST(1);
#pragma unroll 1
  for (int i = 1; i <= 8; i++) {
    const secp::fe s = i == 1 ? pre[8] : secp::fe_mul(pre[i - 1], suf[i + 1]);  // Zc / z_i  (z₁ = 1)
    const secp::fe s2 = secp::fe_sqr(s);
    t.x[i] = secp::fe_mul(m[i].x, s2);
    t.y[i] = secp::fe_mul(m[i].y, secp::fe_mul(s2, s));
    t.bx[i] = secp::fe_mul(t.x[i], beta);
  }
}
// k + Σ_{i<33} 8·16^i (k < 2^128): nibble i of the sum, minus 8, is the signed digit i of k
__host__ __device__ __forceinline__ u256 window_bias(const u256 &k) {
  u256 r = secp::zero256();
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) r.v[i] = secp::addc(i < 4 ? k.v[i] : 0u, i < 4 ? 0x88888888u : 0x8u, c);
  return r;
}
constexpr int WINDOW_DIGITS = 33;
// one signed-window addition: acc ± entry |e| of the table (X or β·X), nothing for e = 0; the INL code shape.
// In two halves, because the table lives in the lane's private segment: window_operand READS the entry — the callers issue
// it in front of the four doublings of the window, whose ≈520 instructions then cover the latency of the scratch loads (a
// lone wavefront per SIMD otherwise waits it out, 66 times per signature; hosts differ in how fast scratch is served:
// DESIGN.md §5.1) — and window_add_q adds it.
#ifndef IBFT_WINDOW_PREFETCH
#define IBFT_WINDOW_PREFETCH 1  // 0: the entry is read where it is used, behind the doublings (A/B)
#endif
__host__ __device__ __forceinline__ aff window_operand(const wtab &t, int e, bool lambda, bool flip) {
  const uint32_t mag = (uint32_t)(e < 0 ? -e : e);
  const uint32_t idx = mag ? mag : 1u;  // (a dummy operand for e = 0: the sum is computed and dropped)
  aff q;
  q.x = secp::l26_select(lambda, t.bx[idx], t.x[idx]);
  const secp::fe y = t.y[idx];
  q.y = secp::l26_select((e < 0) != flip, secp::fe_neg(y, 1), y);  // magnitude ≤ 2
  return q;
}
__host__ __device__ __forceinline__ jac window_add_q(const jac &acc, const aff &q, int e) {
  const jac sum = secp::jac_add_aff_t<true>(acc, q);
  return secp::jac_select(e != 0, sum, acc);
}
__host__ __device__ __forceinline__ jac window_add(const jac &acc, const wtab &t, int e, bool lambda, bool flip) {
  return window_add_q(acc, window_operand(t, e, lambda, flip), e);
}
template <bool PREFETCH>
__host__ __device__ __forceinline__ jac ecmult_var_t(const aff &R, const u256 &k) {
  secp::glv_split sp = secp::sc_split_lambda(k);
  aff R1 = R;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(R.y, 1)), R.y);
  const bool flip2 = sp.neg1 != sp.neg2;
  wtab t;
  ecmult_table(R1, t);
  const u256 k1 = window_bias(sp.k1), k2 = window_bias(sp.k2);
  jac acc = secp::jac_inf();
#pragma unroll 1
  for (int i = WINDOW_DIGITS - 1; i >= 0; i--) {
    const int e1 = (int)secp::nibble(k1, i) - 8, e2 = (int)secp::nibble(k2, i) - 8;
    aff q1, q2;
    if (PREFETCH) {
      q1 = window_operand(t, e1, false, false);
      q2 = window_operand(t, e2, true, flip2);
    }
    if (i != WINDOW_DIGITS - 1) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) acc = secp::jac_dbl_t<true>(acc);
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      if (PREFETCH) {
        aff q;  // (one inlined copy of the mixed addition: the operand is chosen, not the code)
        q.x = secp::l26_select(h != 0, q2.x, q1.x);
        q.y = secp::l26_select(h != 0, q2.y, q1.y);
        acc = window_add_q(acc, q, h ? e2 : e1);
      } else {
        acc = window_add(acc, t, h ? e2 : e1, h != 0, h != 0 && flip2);
      }
    }
  }
  acc.z = secp::fe_mul(acc.z, t.zc);  // back from the isomorphic curve
  return acc;
}
__host__ __device__ __forceinline__ jac ecmult_var(const aff &R, const u256 &k) { return ecmult_var_t<IBFT_WINDOW_PREFETCH != 0>(R, k); }

// ---- the same multiplication with the window table in LDS (round 5) -------------------------------------------------
// The private-segment table above costs 3 056 B of scratch per lane — 1.3 GB of HBM traffic per N = 65 536 launch, 172 × the
// algorithmic bytes (profiles/r04x_n65536_traffic.json), served at a speed that differs by 12–19 % between hosts.  Here the
// eight (x, y) pairs of the common-Z table live in the workgroup's LDS: 16 field elements = 640 B per lane, exactly what 160 KB
// give four wavefronts of a compute unit (one per SIMD: the occupancy these kernels have anyway).  Element slot s of lane
// `tid` of a TPB-lane workgroup: words col[(10·s + w)·TPB], col = lds + tid — a lane touches its own column only (no barrier),
// consecutive lanes hit consecutive banks whatever entry each of them asks for (no conflict).  What does not fit is dropped:
// β·X is multiplied at use (one product per λ-window, +1.5 % instructions), the Z's and prefix products of the build live in
// registers (the build is straight-line code over outlined multiplications; z₁ = 1).
constexpr int LTAB_SLOTS = 16;
constexpr int LTAB_WORDS = LTAB_SLOTS * 10;  // 32-bit words per lane
template <int TPB>
struct ltab {
  uint32_t *col;
  secp::fe zc;
};
template <int TPB>
__host__ __device__ __forceinline__ void ltab_put(const ltab<TPB> &t, int s, const secp::fe &v) {
#pragma unroll
  for (int w = 0; w < 10; w++) t.col[(size_t)(10 * s + w) * TPB] = v.n[w];
}
template <int TPB>
__host__ __device__ __forceinline__ secp::fe ltab_get(const ltab<TPB> &t, int s) {
  secp::fe v;
#pragma unroll
  for (int w = 0; w < 10; w++) v.n[w] = t.col[(size_t)(10 * s + w) * TPB];
  return v;
}
// The same sixteen slots in the lane's PRIVATE segment (plus eight for β·x of the entries: memory is not scarce there) — the
// lane kernel's form beyond 65 536 rows, where two wavefronts per SIMD are resident and LDS cannot hold two tables per lane.
// A private array indexed by a loop counter is exactly what the private segment is for.
struct ptab {
  secp::fe e[24];  // slots 0…15: (x, y) of entries 1…8; 16…23: β·x of entries 1…8
  secp::fe zc;
};
template <int TPB>
__host__ __device__ __forceinline__ void tab_put(ltab<TPB> &t, int s, const secp::fe &v) { ltab_put(t, s, v); }
template <int TPB>
__host__ __device__ __forceinline__ secp::fe tab_get(const ltab<TPB> &t, int s) { return ltab_get(t, s); }
__host__ __device__ __forceinline__ void tab_put(ptab &t, int s, const secp::fe &v) { t.e[s] = v; }
__host__ __device__ __forceinline__ secp::fe tab_get(const ptab &t, int s) { return t.e[s]; }

// entry e (1…8): x in slot 2(e−1), y in slot 2(e−1)+1
//
// Round 6 — the table is built with CO-Z additions (Meloni 2007; the "ZADDU" form of Goundar, Joye, Miyaji), in place, by
// rolled loops.  Until round 5 the seven multiples were computed as Jacobian points (4 doublings, 3 mixed additions), their
// seven Z's kept in 70 registers and brought to one common Z by prefix / suffix products: ≈ 111 multiplications written out as
// 21 KB of straight-line code (a register array cannot be indexed by a loop counter without going to the private segment), every
// line of it fetched once per launch by every instruction cache — on a lease whose instruction fetch is slow that is what the
// kernel waits for.  Two points that SHARE their Z add in 5M + 2S, and the addition hands back its first
// operand rescaled to the sum's Z for nothing:
//     C = (X1−X2)²   W1 = X1·C   W2 = X2·C   D = (Y1−Y2)²   A1 = Y1·(W1−W2)
//     X3 = D − W1 − W2     Y3 = (Y1−Y2)·(W1−X3) − A1     Z3 = Z·(X1−X2)        P′ = (W1, A1) ≡ P at Z3
// so the table can be kept at ONE Z all the time: k·R = (k−1)·R + R with both operands at the current Z, then the older
// entries are brought along by λ = X1 − X2 (x·λ², y·λ³: two multiplications each; λ² = C is there already).  The Z of the
// last addition IS the common Z.  ≈ 100 multiplications, no Z kept but the current one, every entry addressed in LDS by a
// loop counter: 4 KB of code instead of 21.  No exceptional case exists: (k−1)·R = ±R would make R a point of order ≤ 9, and the
// curve's order is prime (a row whose r is no x coordinate computes garbage here and is rejected by its failed square root).
template <class TAB>
__host__ __device__ __forceinline__ void ecmult_table_coz(const aff &R1, TAB &t) {
  // 2R from the affine R (Z = 1: mdbl-2007-bl), then R brought to 2R's Z = 2y — for which nothing has to be multiplied but
  // x: Z² = 4·y², Z³·y = 8·y⁴, and y², y⁴ are the doubling's own YY, YYYY.  Seven multiplications for both.
  secp::fe z;
  {
    const secp::fe xx = secp::fe_sqr(R1.x), yy = secp::fe_sqr(R1.y), yyyy = secp::fe_sqr(yy);
    secp::fe tt = secp::fe_sqr(secp::fe_add(R1.x, yy));                                        // in 2 → 1
    tt = secp::fe_add(secp::fe_add(tt, secp::fe_neg(xx, 1)), secp::fe_neg(yyyy, 1));           // 5
    const secp::fe sS = secp::fe_normalize_weak(secp::fe_mul_int(tt, 2));                      // 10 → 1
    const secp::fe mM = secp::fe_mul_int(xx, 3);                                               // 3
    const secp::fe x3 = secp::fe_normalize_weak(secp::fe_add(secp::fe_sqr(mM), secp::fe_neg(secp::fe_mul_int(sS, 2), 2)));  // 4 → 1
    const secp::fe y8 = secp::fe_mul_int(yyyy, 8);                                             // 8
    const secp::fe y3 = secp::fe_normalize_weak(
        secp::fe_add(secp::fe_mul(mM, secp::fe_add(sS, secp::fe_neg(x3, 1))), secp::fe_neg(y8, 8)));  // 1 + 9 → 1
    tab_put(t, 2, x3);
    tab_put(t, 3, y3);
    z = secp::fe_normalize_weak(secp::fe_mul_int(R1.y, 2));
    tab_put(t, 0, secp::fe_mul(R1.x, secp::fe_mul_int(yy, 4)));   // x·Z²
    tab_put(t, 1, secp::fe_normalize_weak(y8));                  // y·Z³ = 8·y⁴
  }
#pragma unroll 1
  for (int k = 3; k <= 8; k++) {  // entry k = entry (k − 1) + entry 1, both at Z = z
    const int sp = 2 * (k - 2);   // slots of P = entry k − 1; its sum goes to sp + 2
    const secp::fe x1 = tab_get(t, sp), y1 = tab_get(t, sp + 1), x2 = tab_get(t, 0), y2 = tab_get(t, 1);
    const secp::fe dx = secp::fe_add(x1, secp::fe_neg(x2, 1));  // 3
    const secp::fe dy = secp::fe_add(y1, secp::fe_neg(y2, 1));  // 3
    const secp::fe c = secp::fe_sqr(dx);
    const secp::fe w1 = secp::fe_mul(x1, c), w2 = secp::fe_mul(x2, c);
    const secp::fe d = secp::fe_sqr(dy);
    const secp::fe a1 = secp::fe_mul(y1, secp::fe_add(w1, secp::fe_neg(w2, 1)));
    const secp::fe x3 = secp::fe_normalize_weak(secp::fe_add(secp::fe_add(d, secp::fe_neg(w1, 1)), secp::fe_neg(w2, 1)));  // 5 → 1
    const secp::fe y3 = secp::fe_normalize_weak(
        secp::fe_add(secp::fe_mul(dy, secp::fe_add(w1, secp::fe_neg(x3, 1))), secp::fe_neg(a1, 1)));  // 3 → 1
    z = secp::fe_mul(z, dx);
    tab_put(t, sp, w1);  // entry k − 1 at the new Z: free
    tab_put(t, sp + 1, a1);
    tab_put(t, sp + 2, x3);
    tab_put(t, sp + 3, y3);
    const secp::fe c3 = secp::fe_mul(c, dx);  // λ³
    tab_put(t, 0, w2);   // entry 1 at the new Z: x2·λ² is W2
    tab_put(t, 1, secp::fe_mul(y2, c3));
#pragma unroll 1
    for (int j = 2; j <= k - 2; j++) {  // entries 2 … k − 2 follow
      const int sj = 2 * (j - 1);
      tab_put(t, sj, secp::fe_mul(tab_get(t, sj), c));
      tab_put(t, sj + 1, secp::fe_mul(tab_get(t, sj + 1), c3));
    }
  }
  t.zc = z;
}
template <int TPB>
__host__ __device__ __forceinline__ void ecmult_table_lds(const aff &R1, ltab<TPB> &t) { ecmult_table_coz(R1, t); }
template <int TPB>
__host__ __device__ __forceinline__ aff window_operand_lds(const ltab<TPB> &t, int e, bool flip) {
  const uint32_t mag = (uint32_t)(e < 0 ? -e : e);
  const int sl = 2 * (int)((mag ? mag : 1u) - 1u);  // (a dummy operand for e = 0: the sum is computed and dropped)
  aff q;
  q.x = ltab_get(t, sl);
  const secp::fe y = ltab_get(t, sl + 1);
  q.y = secp::l26_select((e < 0) != flip, secp::fe_neg(y, 1), y);  // magnitude ≤ 2
  return q;
}
template <int TPB>
__host__ __device__ __forceinline__ jac ecmult_var_lds(const aff &R, const u256 &k, uint32_t *col) {
  secp::glv_split sp = secp::sc_split_lambda(k);
  aff R1 = R;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(R.y, 1)), R.y);
  const bool flip2 = sp.neg1 != sp.neg2;
  ltab<TPB> t;
  t.col = col;
  ecmult_table_lds<TPB>(R1, t);
  const u256 k1 = window_bias(sp.k1), k2 = window_bias(sp.k2);
  const secp::fe beta = secp::GLV_CONST(1);
  jac acc = secp::jac_inf();
  u256 r1 = k1, r2 = k2;  // shift registers: the current digit is the low nibble of word 4 (secp256k1_dev.h: top_nibble)
#pragma unroll 1
  for (int i = WINDOW_DIGITS - 1; i >= 0; i--) {
    if (i != WINDOW_DIGITS - 1) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) acc = secp::jac_dbl_t<true>(acc);
    }
    const int e1 = (int)secp::top_nibble<5>(r1) - 8, e2 = (int)secp::top_nibble<5>(r2) - 8;
    secp::shl4<5>(r1);
    secp::shl4<5>(r2);
#pragma unroll 1
    for (int h = 0; h < 2; h++) {  // (one inlined copy of the mixed addition; h is uniform: the β product is no divergent call)
      const int e = h ? e2 : e1;
      aff q = window_operand_lds<TPB>(t, e, h != 0 && flip2);
      if (h) q.x = secp::fe_mul(q.x, beta);
      acc = window_add_q(acc, q, e);
    }
  }
  acc.z = secp::fe_mul(acc.z, t.zc);  // back from the isomorphic curve
  return acc;
}
// ---- u2·R + u1·G in ONE loop of mixed additions (round 6) -----------------------------------------------------------------
// The lane kernel carried TWO pasted copies of the mixed addition — one in the window loop above, one in ecmult_gen's loop —
// 18.5 KB each, in a kernel whose code has to be fetched through a 64 KB instruction cache.  Here the 66
// window additions and the 16 fixed-base additions are steps of one loop around one copy: what differs from step to step is
// where the operand comes from (an LDS table entry, with β·X on the odd steps — or the G-table entry asked for one step
// earlier) and whether four doublings run first; all of that hangs on the step number, which is wave-uniform.  The first
// G-table entry is asked for before the window table is even built.
#ifndef IBFT_LANE_MERGED
#define IBFT_LANE_MERGED 1  // 0: ecmult_gen(ecmult_var_lds(…)) — two copies of the mixed addition (round 5; A/B)
#endif
template <int TPB>
__host__ __device__ __forceinline__ jac ecmult_var_gen_lds(const aff &R, const u256 &k, const uint32_t *__restrict__ gtab,
                                                           const u256 &u1, uint32_t *col) {
  u256 kk = u1;  // shift register: the current 16-bit window of u1 in the low bits of word 0
  uint32_t dgt = kk.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
  gtab_raw cur = gtab_load(gtab, 0, dgt);
  secp::glv_split sp = secp::sc_split_lambda(k);
  aff R1 = R;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(R.y, 1)), R.y);
  const bool flip2 = sp.neg1 != sp.neg2;
  ltab<TPB> t;
  t.col = col;
  ecmult_table_lds<TPB>(R1, t);
  u256 r1 = window_bias(sp.k1), r2 = window_bias(sp.k2);  // shift registers: the current digit is the low nibble of word 4
  const secp::fe beta = secp::GLV_CONST(1);
  jac acc = secp::jac_inf();
  int e1 = 0, e2 = 0;
  constexpr int WSTEPS = 2 * WINDOW_DIGITS;
#pragma unroll 1
  for (int st = 0; st < WSTEPS + GTAB_WINDOWS; st++) {
    aff q;
    bool take;
    gtab_raw nxt = cur;
    uint32_t dn = dgt;
    if (st < WSTEPS) {  // (wave-uniform)
      const bool h = (st & 1) != 0;
      if (!h) {
        if (st != 0) {
#pragma unroll 1
          for (int d = 0; d < 4; d++) acc = secp::jac_dbl_t<true>(acc);
        }
        e1 = (int)secp::top_nibble<5>(r1) - 8;
        e2 = (int)secp::top_nibble<5>(r2) - 8;
        secp::shl4<5>(r1);
        secp::shl4<5>(r2);
      }
      const int e = h ? e2 : e1;
      q = window_operand_lds<TPB>(t, e, h && flip2);
      if (h) q.x = secp::fe_mul(q.x, beta);
      take = e != 0;
    } else {
      if (st == WSTEPS) acc.z = secp::fe_mul(acc.z, t.zc);  // back from the isomorphic curve: the G-table points live on the curve itself
      const int w = st - WSTEPS;
      const bool last = w + 1 == GTAB_WINDOWS;
      secp::shr_bits<GTAB_BITS>(kk);
      dn = last ? dgt : kk.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
      nxt = gtab_load(gtab, last ? w : w + 1, dn);  // asked for before this step's addition runs (the last step re-reads its own)
      q = gtab_point(cur);
      take = dgt != 0;
    }
    const jac sum = secp::jac_add_aff_t<true>(acc, q);
    acc = secp::jac_select(take, sum, acc);
    if (st >= WSTEPS) {  // (only the fixed-base steps move the pipeline: the 66 window steps copy nothing)
      cur = nxt;
      dgt = dn;
    }
  }
  return acc;
}

// The same loop over a table in the private segment (the lane kernel beyond 65 536 rows: two resident wavefronts per SIMD).
// The co-Z build works in place there too; β·x of the eight entries is computed once behind it (memory is not scarce in the
// private segment), so the λ steps multiply nothing.  101 → 58 KB of code for the last kernel the AUTO rule dispatches that
// was larger than the instruction cache.
__host__ __device__ __forceinline__ jac ecmult_var_gen_private(const aff &R, const u256 &k, const uint32_t *__restrict__ gtab,
                                                               const u256 &u1) {
  u256 kk = u1;
  uint32_t dgt = kk.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
  gtab_raw cur = gtab_load(gtab, 0, dgt);
  secp::glv_split sp = secp::sc_split_lambda(k);
  aff R1 = R;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(R.y, 1)), R.y);
  const bool flip2 = sp.neg1 != sp.neg2;
  ptab t;
  ecmult_table_coz(R1, t);
  {
    const secp::fe beta = secp::GLV_CONST(1);
#pragma unroll 1
    for (int j = 0; j < 8; j++) t.e[16 + j] = secp::fe_mul(t.e[2 * j], beta);
  }
  u256 r1 = window_bias(sp.k1), r2 = window_bias(sp.k2);
  jac acc = secp::jac_inf();
  int e1 = 0, e2 = 0;
  constexpr int WSTEPS = 2 * WINDOW_DIGITS;
#pragma unroll 1
  for (int st = 0; st < WSTEPS + GTAB_WINDOWS; st++) {
    aff q;
    bool take;
    gtab_raw nxt = cur;
    uint32_t dn = dgt;
    if (st < WSTEPS) {  // (wave-uniform)
      const bool h = (st & 1) != 0;
      if (!h) {
        if (st != 0) {
#pragma unroll 1
          for (int d = 0; d < 4; d++) acc = secp::jac_dbl_t<true>(acc);
        }
        e1 = (int)secp::top_nibble<5>(r1) - 8;
        e2 = (int)secp::top_nibble<5>(r2) - 8;
        secp::shl4<5>(r1);
        secp::shl4<5>(r2);
      }
      const int e = h ? e2 : e1;
      const int mag = e < 0 ? -e : e;
      const int ent = (mag ? mag : 1) - 1;  // (a dummy operand for e = 0: the sum is computed and dropped)
      q.x = t.e[h ? 16 + ent : 2 * ent];
      const secp::fe y = t.e[2 * ent + 1];
      q.y = secp::l26_select((e < 0) != (h && flip2), secp::fe_neg(y, 1), y);  // magnitude ≤ 2
      take = e != 0;
    } else {
      if (st == WSTEPS) acc.z = secp::fe_mul(acc.z, t.zc);  // back from the isomorphic curve
      const int w = st - WSTEPS;
      const bool last = w + 1 == GTAB_WINDOWS;
      secp::shr_bits<GTAB_BITS>(kk);
      dn = last ? dgt : kk.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
      nxt = gtab_load(gtab, last ? w : w + 1, dn);
      q = gtab_point(cur);
      take = dgt != 0;
    }
    const jac sum = secp::jac_add_aff_t<true>(acc, q);
    acc = secp::jac_select(take, sum, acc);
    if (st >= WSTEPS) {
      cur = nxt;
      dgt = dn;
    }
  }
  return acc;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Round 1's u2·R (4-bit unsigned windows over 15 Jacobian multiples, full additions): host builds only — the CPU
// harness checks the new form against it on random and adversarial scalars (tests/test_dev_arith_host.py)
__host__ __device__ __forceinline__ jac ecmult_var_v1(const aff &R, const u256 &k) {
  secp::glv_split sp = secp::sc_split_lambda(k);
  aff R1 = R;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(R.y, 1)), R.y);
  const bool flip2 = sp.neg1 != sp.neg2;
  const secp::fe beta = secp::GLV_CONST(1);
  jac tab[16];
  tab[0] = secp::jac_inf();
  tab[1] = secp::jac_from_aff(R1);
  tab[2] = secp::jac_dbl(tab[1]);
  for (int i = 3; i < 16; i++) tab[i] = secp::jac_add_aff(tab[i - 1], R1);
  jac acc = secp::jac_inf();
  for (int nib = 31; nib >= 0; nib--) {
    for (int d = 0; d < 4; d++) acc = secp::jac_dbl(acc);
    uint32_t d1 = secp::nibble(sp.k1, nib), d2 = secp::nibble(sp.k2, nib);
    jac s1 = secp::jac_add(acc, tab[d1]);
    if (d1 != 0) acc = s1;
    jac q = tab[d2];
    q.x = secp::fe_mul(q.x, beta);
    q.y = secp::l26_select(flip2, secp::fe_neg(q.y, 1), q.y);  // magnitude ≤ 2
    jac s2 = secp::jac_add(acc, q);
    if (d2 != 0) acc = s2;
  }
  return acc;
}
#endif

// Recover the signer address of (digest z, r, s, v); returns false if the signature
// is rejected (same rejection list as oracle/secp256k1.c:orc_ecrecover).
// how u2·R is computed: the table in the private segment (PREFETCH: its entries read in front of the doublings — one resident
// wavefront per SIMD; without: two) or in the workgroup's LDS
// (each returns u2·R + u1·G)
template <bool PREFETCH>
struct var_mult_private {
  __host__ __device__ __forceinline__ jac operator()(const aff &R, const u256 &k, const uint32_t *__restrict__ gtab,
                                                     const u256 &u1) const {
#if IBFT_LANE_MERGED
    if constexpr (!PREFETCH) return ecmult_var_gen_private(R, k, gtab, u1);  // (the form the AUTO rule dispatches beyond 65 536 rows)
#endif
    return ecmult_gen(gtab, u1, ecmult_var_t<PREFETCH>(R, k));  // (round 4's form with the entries read in front of the doublings: A/B)
  }
};
template <int TPB>
struct var_mult_lds {
  uint32_t *col;
  __host__ __device__ __forceinline__ jac operator()(const aff &R, const u256 &k, const uint32_t *__restrict__ gtab,
                                                     const u256 &u1) const {
#if IBFT_LANE_MERGED
    return ecmult_var_gen_lds<TPB>(R, k, gtab, u1, col);
#else
    // (round 5 measured a first form of the merged loop at −2.6 % on a lease of the slow kind, +1.5 % on the fast kind and kept
    // the two copies: profiles/r05la_lane_merge_ab.txt)
    return ecmult_gen(gtab, u1, ecmult_var_lds<TPB>(R, k, col));
#endif
  }
};
template <class VARMULT>
__host__ __device__ __forceinline__ bool recover_pubkey_with(const uint32_t *__restrict__ gtab, const u256 &z_raw,
                                                             const u256 &r, const u256 &s, uint32_t v,
                                                             uint32_t flags, uint32_t addr[5], aff &Qa, const VARMULT &var_mult) {
  bool ok = v <= 1;
  ok = ok && !secp::is_zero(r) && !secp::geq_const(r, secp::NL());
  ok = ok && !secp::is_zero(s) && !secp::geq_const(s, secp::NL());
  if (flags & 1u) {  // strict low-s: reject s > (n-1)/2, i.e. s - 1 >= (n-1)/2
    u256 sm1;
    secp::sub256(sm1, s, secp::one256());
    ok = ok && !secp::geq_const(sm1, secp::NHL());
  }
  // R = (r, y), y^2 = r^3 + 7, parity(y) = v        (r < n < p, so x = r)
  secp::fe rx = secp::fe_from_u256(r);
  secp::fe seven = secp::fe_zero();
  seven.n[0] = 7;
  secp::fe rhs = secp::fe_add(secp::fe_mul(secp::fe_sqr(rx), rx), seven);  // magnitude 2
  secp::fe y = secp::fe_sqrt_candidate(rhs);
  ok = ok && secp::fe_equal(secp::fe_sqr(y), rhs, 2);
  y = secp::fe_normalize(y);
  secp::fe yneg = secp::fe_normalize_weak(secp::fe_neg(y, 1));
  y = secp::l26_select((y.n[0] & 1u) != v, yneg, y);
  aff R;
  R.x = rx;
  R.y = y;
  // u1 = -z/r, u2 = s/r (mod n)
  secp::sc rinv = secp::sc_from_u256(secp::modinv_shared<secp::ModN>(r));  // r is canonical, in [1, n)
  u256 u1 = secp::sc_neg_canon(secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z_raw), rinv)));
  u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(s), rinv));
  jac Q = var_mult(R, u2, gtab, u1);
  ok = secp::jac_to_aff_fast(Qa, Q) && ok;
  u256 qx = secp::l26_to_u256(Qa.x), qy = secp::l26_to_u256(Qa.y);
  keccak::address_from_xy(qx.v, qy.v, addr);
  return ok;
}
__host__ __device__ __forceinline__ bool recover_pubkey(const uint32_t *__restrict__ gtab, const u256 &z_raw,
                                                        const u256 &r, const u256 &s, uint32_t v,
                                                        uint32_t flags, uint32_t addr[5], aff &Qa) {
  return recover_pubkey_with(gtab, z_raw, r, s, v, flags, addr, Qa, var_mult_private<IBFT_WINDOW_PREFETCH != 0>{});
}
__host__ __device__ __forceinline__ bool recover_address(const uint32_t *__restrict__ gtab, const u256 &z_raw,
                                                         const u256 &r, const u256 &s, uint32_t v,
                                                         uint32_t flags, uint32_t addr[5]) {
  aff Qa;
  return recover_pubkey(gtab, z_raw, r, s, v, flags, addr, Qa);
}

}  // namespace ibftk
