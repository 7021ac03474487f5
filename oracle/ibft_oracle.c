/*
 * oracle/ibft_oracle.c — CPU restatement of the verifier hot path, row by row
 * (TEST INFRASTRUCTURE ONLY; see ibft_oracle.h for who may call it).
 *
 * Each function follows the reference call site it names; the per-row loop is
 * the loop of /root/reference/messages/messages.go:183-191 with the closure
 * bodies of /root/reference/core/ibft.go:858-861 (PREPARE) and :932-944 (COMMIT).
 */
#include "ibft_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ---- validator set: /root/reference/core/validator_manager.go:22-75 -------- */
struct orc_valset {
  size_t n;         /* distinct addresses */
  uint8_t *addr;    /* n × 20, sorted */
  uint64_t *power;  /* n */
  u128 total;
  u128 quorum;
};

static int cmp20(const void *a, const void *b) { return memcmp(a, b, 20); }

typedef struct {
  uint8_t a[20];
  uint32_t idx;
} ent_t;

/* order by (addr, original index) so that "last writer wins" can be applied */
static int cmp_ent(const void *x, const void *y) {
  const ent_t *p = (const ent_t *)x, *q = (const ent_t *)y;
  int c = memcmp(p->a, q->a, 20);
  if (c) return c;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

orc_valset_t *orc_valset_new(const uint8_t *addrs20, const uint64_t *power, size_t n) {
  ent_t *e = (ent_t *)malloc(sizeof(ent_t) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) {
    memcpy(e[i].a, addrs20 + 20 * i, 20);
    e[i].idx = (uint32_t)i;
  }
  qsort(e, n, sizeof(ent_t), cmp_ent);
  orc_valset_t *vs = (orc_valset_t *)calloc(1, sizeof *vs);
  vs->addr = (uint8_t *)malloc(20 * (n ? n : 1));
  vs->power = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    if (i + 1 < n && memcmp(e[i].a, e[i + 1].a, 20) == 0) continue; /* later dup wins */
    memcpy(vs->addr + 20 * m, e[i].a, 20);
    vs->power[m] = power[e[i].idx];
    m++;
  }
  free(e);
  vs->n = m;
  vs->total = 0;
  for (size_t i = 0; i < m; i++) vs->total += vs->power[i];
  if (vs->total == 0) { /* errVotingPowerNotCorrect, validator_manager.go:68-70 */
    orc_valset_free(vs);
    return NULL;
  }
  /* calculateQuorum, validator_manager.go:130-135: floor(2*total/3) + 1 */
  vs->quorum = (vs->total * 2) / 3 + 1;
  return vs;
}

void orc_valset_free(orc_valset_t *vs) {
  if (!vs) return;
  free(vs->addr);
  free(vs->power);
  free(vs);
}

int orc_valset_index(const orc_valset_t *vs, const uint8_t addr20[20]) {
  if (!vs || !vs->n) return -1;
  const uint8_t *hit = (const uint8_t *)bsearch(addr20, vs->addr, vs->n, 20, cmp20);
  return hit ? (int)((hit - vs->addr) / 20) : -1;
}

void orc_valset_quorum(const orc_valset_t *vs, uint64_t *lo, uint64_t *hi) {
  *lo = (uint64_t)vs->quorum;
  *hi = (uint64_t)(vs->quorum >> 64);
}

/* ---- a1: IsValidProposalHash ------------------------------------------------- */
void orc_proposal_hash(const uint8_t *raw, size_t raw_len, uint64_t round, uint8_t out[32]) {
  uint8_t *buf = (uint8_t *)malloc(raw_len + 8);
  if (raw_len) memcpy(buf, raw, raw_len);
  for (int i = 0; i < 8; i++) buf[raw_len + i] = (uint8_t)(round >> (8 * (7 - i)));
  orc_keccak256(buf, raw_len + 8, out);
  free(buf);
}

void orc_verify_hashes(const uint8_t *raw, size_t raw_len, uint64_t round, const uint8_t *hash32,
                       const uint8_t *hash_len, size_t n, uint8_t *verdict) {
  uint8_t H[32];
  orc_proposal_hash(raw, raw_len, round, H);
  for (size_t i = 0; i < n; i++)
    verdict[i] = (hash_len[i] == 32 && memcmp(hash32 + 32 * i, H, 32) == 0) ? 1 : 0;
}

/* ---- a2: IsValidCommittedSeal ------------------------------------------------ */
typedef int (*recover_fn)(const uint8_t *, const uint8_t *, uint32_t, uint8_t *);
static inline uint8_t seal_row_with(recover_fn recover, const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                                    const uint8_t *signer20, uint8_t pre, uint32_t flags) {
  if (pre) return 0; /* nil seal / bad length / a1 short-circuit: ibft.go:938-943 */
  uint8_t addr[20];
  if (!recover(hash32, sig65, flags, addr)) return 0;
  if (memcmp(addr, signer20, 20) != 0) return 0;
  return orc_valset_index(vs, signer20) >= 0 ? 1 : 0;
}
static uint8_t seal_row(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                        const uint8_t *signer20, uint8_t pre, uint32_t flags) {
  return seal_row_with(orc_recover_address, vs, hash32, sig65, signer20, pre, flags);
}

void orc_verify_seals(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                      const uint8_t *signer20, const uint8_t *pre_flags, size_t n, uint32_t flags,
                      uint8_t *verdict) {
  for (size_t i = 0; i < n; i++)
    verdict[i] = seal_row(vs, hash32 + 32 * i, sig65 + 65 * i, signer20 + 20 * i,
                          pre_flags ? pre_flags[i] : 0, flags);
}

typedef struct {
  const orc_valset_t *vs;
  const uint8_t *hash32, *sig65, *signer20, *pre;
  size_t lo, hi;
  uint32_t flags;
  uint8_t *verdict;
  recover_fn recover;
} seal_job_t;

static void *seal_worker(void *arg) {
  seal_job_t *j = (seal_job_t *)arg;
  for (size_t i = j->lo; i < j->hi; i++)
    j->verdict[i] = seal_row_with(j->recover, j->vs, j->hash32 + 32 * i, j->sig65 + 65 * i, j->signer20 + 20 * i,
                                  j->pre ? j->pre[i] : 0, j->flags);
  return NULL;
}

static void verify_seals_threads(recover_fn recover, const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                                 const uint8_t *signer20, const uint8_t *pre_flags, size_t n, uint32_t flags,
                                 uint8_t *verdict, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  uint8_t dummy_pub[64], one[32] = {0};
  one[31] = 1;
  orc_pubkey(one, dummy_pub); /* force the G-table build before spawning */
  pthread_t th[256];
  seal_job_t jobs[256];
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (seal_job_t){vs, hash32, sig65, signer20, pre_flags, n * (size_t)t / nthreads,
                           n * (size_t)(t + 1) / nthreads, flags, verdict, recover};
    pthread_create(&th[t], NULL, seal_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

void orc_verify_seals_mt(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                         const uint8_t *signer20, const uint8_t *pre_flags, size_t n,
                         uint32_t flags, uint8_t *verdict, int nthreads) {
  verify_seals_threads(orc_recover_address, vs, hash32, sig65, signer20, pre_flags, n, flags, verdict, nthreads);
}

/* the same rows through the tuned recovery (recover_tuned.inc): bench.py's cpu_baseline only, never the checker */
void orc_verify_seals_tuned_mt(const orc_valset_t *vs, const uint8_t *hash32, const uint8_t *sig65,
                               const uint8_t *signer20, const uint8_t *pre_flags, size_t n,
                               uint32_t flags, uint8_t *verdict, int nthreads) {
  verify_seals_threads(orc_recover_address_tuned, vs, hash32, sig65, signer20, pre_flags, n, flags, verdict, nthreads);
}

/* ---- a3: IsValidValidator ------------------------------------------------------ */
void orc_verify_senders(const orc_valset_t *vs, const uint8_t *payload, const uint32_t *off,
                        const uint8_t *sig65, const uint8_t *from20, const uint8_t *pre_flags,
                        size_t n, uint32_t flags, uint8_t *verdict) {
  for (size_t i = 0; i < n; i++) {
    verdict[i] = 0;
    if (pre_flags && pre_flags[i]) continue;
    uint8_t digest[32], addr[20];
    orc_keccak256(payload + off[i], off[i + 1] - off[i], digest);
    if (!orc_recover_address(digest, sig65 + 65 * i, flags, addr)) continue;
    if (memcmp(addr, from20 + 20 * i, 20) != 0) continue;
    if (orc_valset_index(vs, from20 + 20 * i) < 0) continue;
    verdict[i] = 1;
  }
}

/* ---- a8: HasQuorum -------------------------------------------------------------- */
void orc_tally(const orc_valset_t *vs, const uint8_t *sender20, const uint8_t *verdict, size_t n,
               orc_tally_t *out) {
  memset(out, 0, sizeof *out);
  if (!vs) return; /* "if not initialized correctly return false", validator_manager.go:82-84 */
  uint8_t *seen = (uint8_t *)calloc(vs->n ? vs->n : 1, 1);
  u128 sum = 0;
  for (size_t i = 0; i < n; i++) {
    if (!verdict[i]) continue;
    out->valid_rows++;
    int idx = orc_valset_index(vs, sender20 + 20 * i);
    if (idx < 0 || seen[idx]) continue; /* unknown senders contribute 0; set semantics */
    seen[idx] = 1;
    out->distinct_senders++;
    sum += vs->power[idx];
  }
  free(seen);
  out->power_lo = (uint64_t)sum;
  out->power_hi = (uint64_t)(sum >> 64);
  out->quorum_lo = (uint64_t)vs->quorum;
  out->quorum_hi = (uint64_t)(vs->quorum >> 64);
  out->has_quorum = sum >= vs->quorum ? 1 : 0;
}
