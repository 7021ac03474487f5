// ibftgpu.hip — host side of libibftgpu.so: the C ABI declared in include/ibftgpu.h.
//
// Product code.  No CPU verification path exists here on purpose: if the device or a
// HIP call fails the entry point returns a negative code and the caller (the Go shim,
// INTEGRATION.md) runs its own per-message Verifier for that batch.
#include "../../include/ibftgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is dlopen()ed on first use (ibft_comm_* / ibft_group_*)
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

#include "kernels.hip.h"

namespace {

constexpr uint32_t DEFAULT_MAX_ROWS = 65536;

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

}  // namespace

// What belongs to a DEVICE rather than to a context: the fixed-base table of G (84 MB, 9 ms to build) and — with
// IBFT_FLAG_PUBKEY_CACHE — the validators' key tables.  Every context of a device shares one copy (the four contexts a
// Backend holds for its four goroutines, the several ranks of a group that lists one device more than once); the tables
// are keyed by ADDRESS, so a validator keeps its slot and its table when the validator set changes around it.
struct KeyAddr {
  uint8_t b[20];
  bool operator==(const KeyAddr &o) const { return memcmp(b, o.b, 20) == 0; }
};
struct KeyAddrHash {
  size_t operator()(const KeyAddr &a) const {
    uint64_t x, y;
    uint32_t z;
    memcpy(&x, a.b, 8);
    memcpy(&y, a.b + 8, 8);
    memcpy(&z, a.b + 16, 4);
    return (size_t)((x * 0x9E3779B97F4A7C15ull) ^ (y * 0xC2B2AE3D27D4EB4Full) ^ z);
  }
};
struct DeviceShared {
  std::mutex mu;  // taken around every launch that reads the pointers below and around every change of them
  int device = 0;
  DevBuf d_gtab;
  // key cache: slot s holds pub (80 B), state (0 unknown, 1 key known, 2 table built) and a 655 KB table
  DevBuf d_pub, d_state, d_qtab, d_learned;  // d_learned: {keys learned, a learned slot} (two u32), bumped by the kernels
  uint32_t cap = 0;                          // slots allocated
  std::unordered_map<KeyAddr, uint32_t, KeyAddrHash> slot_of;
  std::vector<uint32_t> refs;                // contexts whose current validator set holds the slot's address
  std::vector<KeyAddr> addr_of;
  std::vector<uint8_t> built;                // host shadow of state == 2 (refreshed after every build pass)
  std::vector<uint32_t> free_slots;
  uint32_t used = 0;                         // slots handed out so far (high-water mark)
  uint32_t learned_seen = 0;                 // value of the device counter up to which tables are built (or being built)
  uint32_t dummy_slot = 0;
  uint64_t build_epoch = 1;                  // bumped by every build pass: contexts recount their built validators
  uint64_t generation = 1;                   // bumped when the buffers move: contexts re-read the pointers
};
std::mutex g_devices_mu;
std::map<int, std::weak_ptr<DeviceShared>> g_devices;

struct ibft_ctx {
  std::mutex mu;
  std::shared_ptr<DeviceShared> dev;   // the G table and the key cache of this context's device
  DevBuf d_vslot;                      // validator index → key-cache slot (0xFFFFFFFF = none)
  std::vector<uint32_t> vslot;         // the same on the host
  uint32_t my_built = 0;               // validators of the current set whose table is built
  uint64_t seen_build_epoch = 0;
  int device = 0;
  uint32_t flags = 0;
  uint32_t max_rows = DEFAULT_MAX_ROWS;
  uint32_t row_cap = 0;  // rows the columns can hold: max_rows, or 2·⌈max_rows/64⌉·64 once a message set was verified
  uint32_t kernel = IBFT_KERNEL_AUTO;
  hipStream_t stream = nullptr;
  std::string last_error;

  // columns in HBM
  DevBuf d_hash, d_sig, d_signer, d_pre, d_hash_len, d_payload, d_off, d_raw;
  // the SECOND staging slot of the seal columns (ibft_seals_stage_next / ibft_seals_swap): batch k+1 is copied here on a
  // copy stream of its own while the verdict kernels read batch k from the columns above; a swap exchanges the two sets
  DevBuf d_hash_nx, d_sig_nx, d_signer_nx, d_pre_nx;
  hipStream_t cstream = nullptr;
  hipEvent_t ev_staged = nullptr, ev_cols_read = nullptr;  // the copy into the spare slot is done / the spare slot's last reader is
  uint32_t next_n = 0;
  bool next_pre = false, next_valid = false;
  // pipelined passes (ibft_seals_submit / ibft_seals_collect): two host-visible result slots, an event behind each pass's tally
  uint64_t *p_mask[2] = {nullptr, nullptr}, *p_tally[2] = {nullptr, nullptr}, *dp_mask[2] = {nullptr, nullptr}, *dp_tally[2] = {nullptr, nullptr};
  hipEvent_t ev_pass[2] = {nullptr, nullptr};
  uint32_t pass_issued = 0, pass_collected = 0, pass_n[2] = {0, 0};
  int learn_rc = IBFT_OK;  // a table build behind a DELIVERED pass failed: reported by the next ibft_seals_submit
  bool split_large = true;    // batches of 65 537 … 98 304 rows as two launches (enqueue_recover); IBFT_SPLIT_LARGE=0 turns it off (A/B)
  uint32_t split_launches = 0;
  int tally_slot = -1;      // ≥ 0: the next tally delivers into pipeline slot `tally_slot` instead of h_mask / h_tally
  // The tally of a pipelined pass on a stream of its own (round 6): pass k's tally (one small workgroup, 9–15 µs with its
  // dependency gap) runs NEXT TO pass k + 1's verdict kernel instead of in front of it.  What the two would share is doubled:
  // the work mask and the validator-index column exist twice (d_mask / d_vidx are the pair the next verdict launch writes,
  // d_mask_b / d_vidx_b the pair the tally in flight reads and zeroes; ibft_seals_submit swaps them), and pair P(k) = P(k − 2) is
  // free again because pass k − 2 was collected before pass k could be submitted (at most two in flight).  Every OTHER entry
  // point puts the main stream behind the side stream's tallies first (ctx_lock → join_side).
  // MEASURED (profiles/r06v…r06y_side_tally_ab.txt, r06v_side_tally_trace.txt).  Two things had to be learned: (1) the tally
  // must FIT next to a resident verdict kernel — the one-workgroup form (sixteen wavefronts of 126 registers) fits next to
  // nothing and simply waited for the NEXT verdict kernel to end (N = 4 096: 0.351 → 0.505 ms per step); the ticket form (42
  // registers) does, so a side-stream tally always takes that one (enqueue_tally); (2) the lane / group cold kernels fill the
  // LDS of every compute unit with their window tables: a tally workgroup placed first takes the room of a verdict workgroup,
  // which then runs as a second round (N = 16 384: 0.71 → 1.37 ms, 65 536: 0.92 → 1.87).  Behind everything else it pays:
  // cold N = 64 … 8 192 −0.8 … −1.7 % per step, warm N = 1 024 … 65 536 −2.3 … −6 %.  Hence side_tally: 0 never, 1 always
  // (A/B), 2 = AUTO (ibft_seals_submit).  IBFT_SIDE_TALLY=0|1 pins it.
  int side_tally = 2;
  bool side_pending = false;
  hipStream_t tstream = nullptr;
  hipEvent_t ev_rec[2] = {nullptr, nullptr};   // pass k's verdict kernels are done (recorded on the main stream)
  DevBuf d_mask_b, d_vidx_b;
  uint32_t mask_dirty_words_b = ~0u;
  uint32_t side_tallies = 0;                   // passes whose tally ran on the side stream (ibft_last_dispatch-style counter)
  uint32_t launched_n = 0;  // rows of the last ibft_seals_launch: what ibft_seals_fetch delivers (a swap may have changed staged_n since)
  DevBuf d_mask, d_vidx, d_tally, d_H;
  DevBuf d_mask_out;        // verdict words after the tally consumed d_mask (what fetch / export read)
  // d_mask words [0, mask_dirty_words) may hold bits; 0 = the whole work mask is zero (the tally left it so)
  // and no memset is needed in front of the atomicOr kernels
  uint32_t mask_dirty_words = ~0u;
  DevBuf d_wire_rows, d_seal;  // §8f rank 3: per-row parse results and the COMMIT seals found in the wire bytes
  DevBuf d_noseal;             // message sets from wire bytes: rows (PREPAREs) whose closure has no seal
  uint32_t wire_n = 0;         // rows of the last ibft_verify_senders_wire
  bool wire_valid = false;     // its columns are still the resident ones
  // fixed-base table for G
  // validator table
  DevBuf d_vtab, d_vpower;
  uint32_t vslot_mask = 0;
  uint32_t n_validators = 0;
  bool have_valset = false;
  uint32_t power_words = 1;                          // 64-bit words per voting power: 1 (u64) or 4 (256-bit)
  uint64_t quorum_w[ibftk::TALLY_SUM_WORDS] = {0};   // ⌊2·total/3⌋+1, little-endian words
  DevBuf d_seen, d_acc, d_quorum;                    // tally: distinct-sender bitmap, launch-wide sums + ticket, quorum words
  uint64_t last_wide[ibftk::TALLY_SUM_WORDS] = {0};  // full-width power of the last fetched tally
  uint64_t height = 0;
  // the seal-digest convention of the embedding Backend (ibft_set_seal_digest): 0 = the proposalHash itself
  uint32_t seal_digest_mode = 0;
  uint64_t seal_suffix_words[9] = {0};
  DevBuf d_hash_copy;            // message sets under a non-identity convention: the carried hashes a1 compares
  std::vector<uint32_t> h_vtab;  // host copy of the validator table (6 dwords per slot): the proposer's seat is looked up here
  // HasPrepareQuorum: set by an entry point that was given a proposer, consumed by the next tally it enqueues
  bool next_prop_on = false;
  int32_t next_prop_vidx = -1;
  // … and that tally is ONE SHARD's share of a sharded PREPARE set whose exchange follows in the same call (set by
  // ibft_group_verify_messages only): the proposer's seat is then left to the merge.  Every other tally — ibft_tally_prepare,
  // a one-shot ibft_verify_messages[_wire] — adds the seat itself, also on a context that is a member of a group or of a
  // communicator (round-4 advice: the seat was dropped by membership, and HasPrepareQuorum under-counted there).
  bool next_shard = false;
  uint32_t last_proposer_rows = 0;

  // multi-GPU exchange (ibft_comm_*): rows of a batch sharded over `xworld` contexts, one all-reduce merges them
  ncclComm_t comm = nullptr;
  bool xlocal = false;  // a rank of a group whose collective is the library's own sum kernel (no communicator)
  uint32_t xrank = 0, xworld = 1;
  hipStream_t xstream = nullptr;
  DevBuf d_xbuf[2], d_xres[2];
  DevBuf d_seen_out;    // the last tally's distinct-sender bitmap (⌈n_validators/64⌉ u64 words): what the ranks exchange
  hipEvent_t ev_xpack = nullptr;  // local collective: this rank's buffer is packed / the summed buffers are back
  uint32_t x_K[2] = {1, 1};       // verdict arrays of the exchange in each slot (1: seal / sender batch, 2: message set)
  uint32_t set_n = 0;             // rows of the last message set (its words are in d_set)
  uint64_t *h_xres[2] = {nullptr, nullptr}, *dh_xres[2] = {nullptr, nullptr};
  size_t h_xres_words = 0;
  hipEvent_t ev_xdone[2] = {nullptr, nullptr};
  uint64_t x_total[2] = {0, 0};   // n_total of the exchange in each slot
  uint32_t x_issued = 0, x_fetched = 0;  // exchanges enqueued / consumed (at most 2 in flight)

  // warm path (IBFT_FLAG_PUBKEY_CACHE): recovered keys + per-validator fixed-base tables
  bool cache_on = false;       // flag set AND the device's key cache could be set up for this validator set
  DevBuf d_warm_done;
  std::vector<uint8_t> valset_addrs;  // last address list, to keep the cache across identical sets
  uint32_t warm_passes = 0, cold_passes = 0, last_group = 0, last_cold_group = 1;
  bool cold_group_auto = true;
  uint32_t cold_group_force = 0;  // IBFT_COLD_LANES=1|2|4|8|64 (experiments: pin the cold kernel variant)
  // IBFT_COLD_TABLE=lds|private|private2 pins where the lane / group cold kernels keep their window tables: 1 LDS, 2 private
  // segment with the prefetch (round 4's form), 3 private segment without it (two resident wavefronts; lane kernel only)
  uint32_t cold_table_force = 0, last_cold_table = 0;
  uint32_t warm_group_force = 0;  // IBFT_WARM_LANES=1|2|…|64 (experiments: pin the warm kernel variant)
  uint32_t rows_kernel_max = 8192;  // AUTO: a DPP row per signature above wave_rows_max up to this many rows
                                    // (4 096 rows: 0.55 ms vs 0.84 ms for the 8-lane kernel; 8 192: 0.84 vs 0.86)
  uint32_t pair_rows_max = 512;   // AUTO: TWO wavefronts per signature up to this many rows (≤ one wavefront per SIMD in all;
                                  // IBFT_PAIR_ROWS_MAX=0 turns the form off)
  uint32_t wave_rows_max = 2048;  // AUTO: one wavefront per signature up to this many rows (two per SIMD: 0.45 ms); the
                                  // row-per-signature kernel (0.55 ms up to 4 096 rows) wins from there

  // a1: the proposal whose Keccak the device holds in d_H (raw ‖ BE64(round)); the same proposal is checked
  // against every PREPARE and COMMIT set of a round and on every wake-up, so it is hashed once
  std::vector<uint8_t> hashed_proposal;
  std::vector<uint8_t> padded_proposal;  // the same with Keccak's pad10*1, as uploaded
  bool have_H = false;
  // the proposal is hashed on a stream of its own (one lane, ≈9 µs per 136-byte block): a message set's verdict launch does
  // not depend on it — only the combine step inside the tally does — so the two overlap
  hipStream_t hstream = nullptr;
  hipEvent_t ev_H = nullptr;
  bool H_pending = false;  // the main stream has not yet been told to wait for ev_H
  // Where the proposal is hashed: on the host by default (IBFT_PROPOSAL_HASH=device selects the wavefront sponge kernel)
  bool hash_on_host = true;
  bool H_copy_issued = false;   // ev_H marks a copy out of the pinned staging slot
  uint8_t H_host[32] = {0};     // the digest d_H holds, when the host computed it
  bool have_H_host = false;

  // staged batch
  uint32_t staged_n = 0;
  bool staged_pre = false;

  // pinned host mirrors for results
  uint64_t *h_mask = nullptr;
  uint64_t *h_tally = nullptr;
  uint64_t *h_digest = nullptr;  // 32-byte staging for a1 digests (never shared with results a kernel may still deliver)
  uint64_t *dh_mask = nullptr, *dh_tally = nullptr;  // the same pinned buffers as the device sees them
  // message sets (ibft_verify_messages): sender words then valid words, ⌈max_rows/64⌉ each
  uint64_t *h_set = nullptr, *dh_set = nullptr;
  DevBuf d_set;
  uint8_t *h_class = nullptr, *dh_class = nullptr;  // one routing byte per wire row (ibft_verify_messages_wire)
  DevBuf d_class;
  // certificates from wire bytes (ibft_verify_certificates_wire): tree nodes, where each row's nested messages lie,
  // child counts of the level being expanded, proposal digests, hash / self words; the row count of the next level
  // comes back through one mapped word
  DevBuf d_cert_nodes, d_cert_span, d_cert_count, d_cert_prop, d_cert_masks, d_cert_total, d_cert_slot, d_cert_tiles;
  uint32_t *h_cert_total = nullptr, *dh_cert_total = nullptr;
  hipEvent_t ev_cert_fork = nullptr, ev_cert_join = nullptr;  // the side stream (hstream) hashes the deferred rows next to the verdict launch
  bool cert_overlap = true;                                   // IBFT_CERT_OVERLAP=0: everything on one stream, one verdict launch
  bool gather_pinned = true;  // columns in ibft_pinned_alloc buffers are read by one gather launch (IBFT_NO_GATHER=1: never)
  // … whose extra blocks can hash PayloadNoSig straight from the host column (IBFT_DIGEST_FUSION=1).  Off by default: it
  // saves nothing measurable (gather + digest ≈ 46 µs either way) and the 8 192-row known-key kernel of a COMMIT set ran
  // 0.25 ms instead of 0.17 ms behind the variant of the launch that carries the digest blocks' 32 KiB of LDS
  // (profiles/r02q_n4096_seq_kernel_stats.csv against r02r)
  bool digest_in_gather = false;
  uint32_t gathers = 0;       // batches whose columns came in through the gather launch
  bool host_direct = false;                          // the last tally kernel delivered its results there
  hipEvent_t ev_ready = nullptr, ev_read = nullptr;  // ibft_seals_export_on: results ready / results read
  bool read_pending = false;                         // the next tally must wait for ev_read

  // dominant-kernel timing
  uint32_t time_every = 1;     // HIP-event pair around the verdict kernels of every n-th staged pass (0 = never)
  uint32_t pass_counter = 0;
  std::vector<hipEvent_t> ev;  // pairs
  uint32_t ev_used = 0;
};

namespace {

#define HIPCHK(ctx, expr)                                                        \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);     \
      return IBFT_E_HIP;                                                         \
    }                                                                            \
  } while (0)

int ensure(ibft_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return IBFT_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes < 256 ? 256 : bytes;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    c->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
    return IBFT_E_NOMEM;
  }
  b.cap = want;
  return IBFT_OK;
}

void release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

int mask_words(size_t n) { return (int)((n + 63) / 64); }

int alloc_rows(ibft_ctx *c, uint32_t rows) {
  size_t m = rows;
  int rc;
  const void *old_mask = c->d_mask.p;
  if ((rc = ensure(c, c->d_hash, m * 32))) return rc;
  if ((rc = ensure(c, c->d_sig, m * 65 + 64))) return rc;
  if ((rc = ensure(c, c->d_signer, m * 20))) return rc;
  if ((rc = ensure(c, c->d_pre, m))) return rc;
  if ((rc = ensure(c, c->d_hash_len, m))) return rc;
  if ((rc = ensure(c, c->d_off, (m + 1) * 4))) return rc;
  if ((rc = ensure(c, c->d_mask, (size_t)mask_words(m) * 8))) return rc;
  if ((rc = ensure(c, c->d_mask_out, (size_t)mask_words(m) * 8))) return rc;
  // fresh memory — or words beyond the rows the last clean covered (a small context's mask buffer is never smaller
  // than 256 bytes, so growing row_cap need not reallocate it): the next verdict launch zeroes the whole mask first
  if (c->d_mask.p != old_mask || rows > c->row_cap) c->mask_dirty_words = ~0u;
  if ((rc = ensure(c, c->d_vidx, m * 4))) return rc;
  {  // the pair a side-stream tally works on (ibft_seals_submit): same sizes, same rule for fresh words
    const void *old_b = c->d_mask_b.p;
    if ((rc = ensure(c, c->d_mask_b, (size_t)mask_words(m) * 8))) return rc;
    if (c->d_mask_b.p != old_b || rows > c->row_cap) c->mask_dirty_words_b = ~0u;
    if ((rc = ensure(c, c->d_vidx_b, m * 4))) return rc;
  }
  if ((rc = ensure(c, c->d_tally, (size_t)ibftk::TALLY_OUT_WORDS * 8))) return rc;
  if ((rc = ensure(c, c->d_acc, (size_t)ibftk::TALLY_ACC_WORDS * 8))) return rc;
  if ((rc = ensure(c, c->d_quorum, (size_t)ibftk::TALLY_SUM_WORDS * 8))) return rc;
  if ((rc = ensure(c, c->d_H, 4 * 8))) return rc;
  if ((rc = ensure(c, c->d_warm_done, m))) return rc;
  c->row_cap = std::max(c->row_cap, rows);
  return IBFT_OK;
}

// row_base (a multiple of 64): the batch is rows [row_base, row_base + n) of the columns — its own verdict words
ibftk::recover_args make_args(ibft_ctx *c, uint32_t n, bool with_pre, uint32_t row_base = 0) {
  ibftk::recover_args a{};
  a.hash32 = (const uint8_t *)c->d_hash.p + 32ull * row_base;
  a.sig65 = (const uint8_t *)c->d_sig.p + 65ull * row_base;
  a.signer20 = (const uint8_t *)c->d_signer.p + 20ull * row_base;
  a.pre_flags = with_pre ? (const uint8_t *)c->d_pre.p + row_base : nullptr;
  a.payload = (const uint8_t *)c->d_payload.p;
  a.off = (const uint32_t *)c->d_off.p + row_base;  // (the offsets themselves are absolute positions in `payload`)
  a.gtab = (const uint32_t *)c->dev->d_gtab.p;
  a.vtab = (const uint32_t *)c->d_vtab.p;
  a.vslot_mask = c->vslot_mask;
  a.n = n;
  a.flags = c->flags;
  a.mask = (uint64_t *)c->d_mask.p + row_base / 64;
  a.vidx = (int32_t *)c->d_vidx.p + row_base;
  if (c->cache_on) {  // (the caller holds c->dev->mu: the pointers cannot move before the launch is enqueued)
    a.pub = (uint32_t *)c->dev->d_pub.p;
    a.pub_state = (uint32_t *)c->dev->d_state.p;
    a.learned = (uint32_t *)c->dev->d_learned.p;
    a.qtab = (const uint32_t *)c->dev->d_qtab.p;
    a.dummy_validator = c->dev->dummy_slot;
    a.vslot = (const uint32_t *)c->d_vslot.p;
  }
  return a;
}

int next_events(ibft_ctx *c, hipEvent_t *start, hipEvent_t *stop) {
  if ((size_t)c->ev_used * 2 + 2 > c->ev.size()) {
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    c->ev.push_back(a);
    c->ev.push_back(b);
  }
  *start = c->ev[(size_t)c->ev_used * 2];
  *stop = c->ev[(size_t)c->ev_used * 2 + 1];
  c->ev_used++;
  return IBFT_OK;
}

// zero the work mask unless the last tally already left it zero
int clean_mask(ibft_ctx *c) {
  if (c->mask_dirty_words == 0) return IBFT_OK;
  HIPCHK(c, hipMemsetAsync(c->d_mask.p, 0, (size_t)mask_words(c->row_cap) * 8, c->stream));
  c->mask_dirty_words = 0;
  return IBFT_OK;
}

// The main stream behind every tally the pipeline put on the side stream (a device-side wait; nothing when none is pending).
int join_side(ibft_ctx *c) {
  if (!c->side_pending) return IBFT_OK;
  HIPCHK(c, hipSetDevice(c->device));
  for (int i = 0; i < 2; i++)
    if (c->ev_pass[i] && hipStreamWaitEvent(c->stream, c->ev_pass[i], 0) != hipSuccess) HIPCHK(c, hipStreamSynchronize(c->tstream));
  c->side_pending = false;
  return IBFT_OK;
}
// The lock every entry point takes — and, for all but the pipeline's own calls (ibft_seals_submit / _collect / _stage_next /
// _swap / _rows, the timing getters), the join above: whatever the call enqueues on the main stream finds the buffers a side
// tally works on (work mask, validator indices, tally sums, verdict words) finished with.
struct ctx_lock {
  std::lock_guard<std::mutex> g;
  explicit ctx_lock(ibft_ctx *c) : g(c->mu) {
    if (c->side_pending && join_side(c) != IBFT_OK) {  // (cannot report from here: drain the side stream the hard way)
      (void)hipStreamSynchronize(c->tstream);
      c->side_pending = false;
    }
  }
};

// enqueue the verdict kernels over the resident columns: warm kernel first when tables exist
// (its rows are then skipped by the recover kernel), recover kernel for everything else
// keep_mask: a second batch of the same call (rows [row_base, row_base + n)): the work mask already holds the first batch's
// bits and must not be zeroed again (the caller saw to it that this batch's words are clean)
// validators of this context's set whose table is built, recounted when a build pass has run since the last count
void recount_built(ibft_ctx *c) {  // c->dev->mu held
  if (!c->cache_on || c->seen_build_epoch == c->dev->build_epoch) return;
  uint32_t k = 0;
  for (uint32_t sl : c->vslot) k += (sl != 0xFFFFFFFFu && c->dev->built[sl]) ? 1u : 0u;
  c->my_built = k;
  c->seen_build_epoch = c->dev->build_epoch;
}

// Batches just beyond what one wavefront per SIMD can take (65 536 rows through the lane kernel with its tables in LDS): the
// private-segment form that serves everything larger keeps TWO wavefronts on as many SIMDs as there are rows beyond 65 536 — and
// the launch lasts as long as those: 1.72 ms for 70 000 rows where 65 536 take 0.92 (profiles/r06j_kernel_ab.txt).  Up to
// SPLIT_ROWS_MAX rows the batch is two launches instead — 65 536 rows through the LDS-table lane kernel, the rest through whatever
// the AUTO rule picks for that many rows (rows 65 536 … n are a batch of their own: every kernel takes a row base): 0.92 + 0.33 ms at
// 70 000 rows, 0.92 + 0.75 at 98 304; beyond that the private-segment form wins again (131 072: 1.76 ms).  Cold contexts without
// pinned kernels only (round 6).
constexpr uint32_t SPLIT_ROWS_MIN = 65536u, SPLIT_ROWS_MAX = 98304u;

int enqueue_recover(ibft_ctx *c, uint32_t n, bool with_pre, int mode, bool time_it, uint32_t row_base = 0, bool keep_mask = false) {
  if (n == 0) return IBFT_OK;
  if (n > SPLIT_ROWS_MIN && n <= SPLIT_ROWS_MAX && !c->cache_on && c->cold_group_auto && !c->cold_group_force && !c->cold_table_force &&
      row_base % 64 == 0 && c->split_large) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (time_it) {
      int rc = next_events(c, &e0, &e1);
      if (rc) return rc;
      HIPCHK(c, hipEventRecord(e0, c->stream));
    }
    // The work mask is cleared HERE, once, for both launches: the lane kernel of the first part stores whole verdict words and
    // never asks for a clean mask, the kernel of the second part ORs its bits into words it expects to be zero (a dirty or
    // freshly allocated mask gave wrong verdicts in the rows beyond 65 536: caught by the Byzantine parity test, not by the
    // all-valid A/B rounds).
    int rc = keep_mask ? (int)IBFT_OK : clean_mask(c);
    if (rc) return rc;
    if ((rc = enqueue_recover(c, SPLIT_ROWS_MIN, with_pre, mode, false, row_base, true))) return rc;
    if ((rc = enqueue_recover(c, n - SPLIT_ROWS_MIN, with_pre, mode, false, row_base + SPLIT_ROWS_MIN, true))) return rc;
    if (time_it) HIPCHK(c, hipEventRecord(e1, c->stream));
    c->last_cold_group = 1;   // (what ibft_last_dispatch reports for a split batch: the lane kernel took the bulk)
    c->last_cold_table = 1;
    c->split_launches++;
    return IBFT_OK;
  }
  std::unique_lock<std::mutex> dev_lk(c->dev->mu, std::defer_lock);
  if (c->cache_on) {
    dev_lk.lock();
    recount_built(c);
  }
  ibftk::recover_args a = make_args(c, n, with_pre, row_base);
  struct dirty_on_exit {  // whatever is launched below writes verdict bits into the first ⌈n/64⌉ words of d_mask
    ibft_ctx *c;
    uint32_t words;
    ~dirty_on_exit() { c->mask_dirty_words = std::max(c->mask_dirty_words, words); }
  } mark_dirty{c, (uint32_t)mask_words((size_t)row_base + n)};
  auto clean_mask = [keep_mask](ibft_ctx *cc) { return keep_mask ? (int)IBFT_OK : ::clean_mask(cc); };
  int rc_clean = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (time_it) {
    int rc = next_events(c, &e0, &e1);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(e0, c->stream));
  }
  const bool warm = c->cache_on && c->my_built > 0;
  if (warm) {
    a.warm_done = (uint8_t *)c->d_warm_done.p + row_base;
    // lanes per signature: ≈ one wavefront per SIMD (1024 SIMDs × 64 lanes / n rows), a power of two
    uint32_t G = 1;
    if (c->warm_group_force) {
      G = c->warm_group_force;
    } else if (c->kernel == IBFT_KERNEL_WAVE) {
      G = 64;
    } else if (c->kernel != IBFT_KERNEL_LANE) {
      while (G < 64 && (uint64_t)n * (G * 2) <= 65536ull) G *= 2;
    }
    if (G > 1) {
      if ((rc_clean = clean_mask(c))) return rc_clean;
      const uint32_t rows_per_wave = 64 / G;
      dim3 grid((n + rows_per_wave - 1) / rows_per_wave), block(64);
#define IBFT_LAUNCH_GROUP(GG)                                                                                  \
  if (mode == 0)                                                                                               \
    hipLaunchKernelGGL((ibftk::verify_known_group_kernel<0, GG>), grid, block, 0, c->stream, a);               \
  else                                                                                                         \
    hipLaunchKernelGGL((ibftk::verify_known_group_kernel<1, GG>), grid, block, 0, c->stream, a);
      const dim3 wgrid((n + ibftk::WAVE_KERNEL_WAVES - 1) / ibftk::WAVE_KERNEL_WAVES), wblock(64 * ibftk::WAVE_KERNEL_WAVES);
      switch (G) {
        case 64:
          if (mode == 0)
            hipLaunchKernelGGL(ibftk::verify_known_wave_kernel<0>, wgrid, wblock, 0, c->stream, a);
          else
            hipLaunchKernelGGL(ibftk::verify_known_wave_kernel<1>, wgrid, wblock, 0, c->stream, a);
          break;
        case 32: IBFT_LAUNCH_GROUP(32) break;
        case 16: IBFT_LAUNCH_GROUP(16) break;
        case 8: IBFT_LAUNCH_GROUP(8) break;
        case 4: IBFT_LAUNCH_GROUP(4) break;
        default: IBFT_LAUNCH_GROUP(2) break;
      }
#undef IBFT_LAUNCH_GROUP
    } else {
      dim3 grid((n + ibftk::ROWS_PER_BLOCK - 1) / ibftk::ROWS_PER_BLOCK), block(ibftk::ROWS_PER_BLOCK);
      if (mode == 0)
        hipLaunchKernelGGL(ibftk::verify_known_lane_kernel<0>, grid, block, 0, c->stream, a);
      else
        hipLaunchKernelGGL(ibftk::verify_known_lane_kernel<1>, grid, block, 0, c->stream, a);
    }
    c->last_group = G;
    HIPCHK(c, hipGetLastError());
    c->warm_passes++;
  } else {
    c->cold_passes++;
  }
  if (warm && c->my_built >= c->n_validators) {
    // every validator has a table: the warm kernel decides every row (non-members included)
    if (time_it) HIPCHK(c, hipEventRecord(e1, c->stream));
    c->last_cold_group = 0;
    return IBFT_OK;
  }
  // cold kernel: lane groups when the batch is too small to fill the chip (and nothing is warm:
  // with a warm kernel in front only the stragglers are left and the group kernel's atomicOr
  // merge needs the mask it already holds)
  uint32_t CG = 1;
  if (c->cold_group_force) {
    CG = c->cold_group_force;
  } else if (c->cold_group_auto) {
    if ((uint64_t)n <= c->pair_rows_max) CG = 128;
    else if ((uint64_t)n <= c->wave_rows_max) CG = 64;
    else if ((uint64_t)n <= c->rows_kernel_max) CG = 16;
    else if ((uint64_t)n * 8 <= 65536ull) CG = 8;
    else if ((uint64_t)n * 4 <= 65536ull) CG = 4;
    else if ((uint64_t)n * 2 <= 65536ull) CG = 2;
  }
  if (CG == 16) {
    if (!warm && (rc_clean = clean_mask(c))) return rc_clean;
    const uint32_t waves = (n + 3) / 4;
    const dim3 rgrid((waves + ibftk::WAVE_KERNEL_WAVES - 1) / ibftk::WAVE_KERNEL_WAVES), rblock(64 * ibftk::WAVE_KERNEL_WAVES);
    if (mode == 0)
      hipLaunchKernelGGL(ibftk::ecrecover_rows_kernel<0>, rgrid, rblock, 0, c->stream, a);
    else
      hipLaunchKernelGGL(ibftk::ecrecover_rows_kernel<1>, rgrid, rblock, 0, c->stream, a);
  } else if (CG == 128) {
    if (!warm && (rc_clean = clean_mask(c))) return rc_clean;
    const dim3 pgrid((n + ibftk::PAIRS_PER_BLOCK - 1) / ibftk::PAIRS_PER_BLOCK), pblock(128 * ibftk::PAIRS_PER_BLOCK);
    if (mode == 0)
      hipLaunchKernelGGL(ibftk::ecrecover_wave2_kernel<0>, pgrid, pblock, 0, c->stream, a);
    else
      hipLaunchKernelGGL(ibftk::ecrecover_wave2_kernel<1>, pgrid, pblock, 0, c->stream, a);
  } else if (CG == 64) {
    if (!warm && (rc_clean = clean_mask(c))) return rc_clean;
    if (mode == 0)
      hipLaunchKernelGGL(ibftk::ecrecover_wave_kernel<0>, dim3((n + ibftk::WAVE_KERNEL_WAVES - 1) / ibftk::WAVE_KERNEL_WAVES),
                         dim3(64 * ibftk::WAVE_KERNEL_WAVES), 0, c->stream, a);
    else
      hipLaunchKernelGGL(ibftk::ecrecover_wave_kernel<1>, dim3((n + ibftk::WAVE_KERNEL_WAVES - 1) / ibftk::WAVE_KERNEL_WAVES),
                         dim3(64 * ibftk::WAVE_KERNEL_WAVES), 0, c->stream, a);
  } else if (CG > 1) {
    if (!warm && (rc_clean = clean_mask(c))) return rc_clean;
    const uint32_t rows_per_wave = 64 / CG;
    dim3 cgrid((n + rows_per_wave - 1) / rows_per_wave), cblock(64);
    // the lanes' window tables: LDS (no private segment) unless pinned otherwise (IBFT_COLD_TABLE=private: round 4's form, A/B)
    const bool lds_tab = c->cold_table_force != 2;
#define IBFT_LAUNCH_COLD(GG)                                                                                          \
  if (mode == 0 && lds_tab)                                                                                           \
    hipLaunchKernelGGL((ibftk::ecrecover_group_kernel<0, GG, ibftk::TAB_LDS>), cgrid, cblock, 0, c->stream, a);       \
  else if (mode == 0)                                                                                                 \
    hipLaunchKernelGGL((ibftk::ecrecover_group_kernel<0, GG, ibftk::TAB_PRIVATE_PREFETCH>), cgrid, cblock, 0, c->stream, a); \
  else if (lds_tab)                                                                                                   \
    hipLaunchKernelGGL((ibftk::ecrecover_group_kernel<1, GG, ibftk::TAB_LDS>), cgrid, cblock, 0, c->stream, a);       \
  else                                                                                                                \
    hipLaunchKernelGGL((ibftk::ecrecover_group_kernel<1, GG, ibftk::TAB_PRIVATE_PREFETCH>), cgrid, cblock, 0, c->stream, a);
    switch (CG) {
      case 8: IBFT_LAUNCH_COLD(8) break;
      case 4: IBFT_LAUNCH_COLD(4) break;
      default: IBFT_LAUNCH_COLD(2) break;
    }
#undef IBFT_LAUNCH_COLD
    c->last_cold_table = lds_tab ? 1 : 2;
  } else {
    dim3 grid((n + ibftk::ROWS_PER_BLOCK - 1) / ibftk::ROWS_PER_BLOCK), block(ibftk::ROWS_PER_BLOCK);
    // one lane per signature: up to one wavefront per SIMD offered (n ≤ 65 536) the table lives in LDS; beyond, in the private
    // segment WITHOUT the prefetch — 256 registers, two resident wavefronts per SIMD (15 against 18 ns per verify)
    const uint32_t tab = c->cold_table_force ? c->cold_table_force : ((uint64_t)n <= 65536ull ? 1u : 3u);
#define IBFT_LAUNCH_LANE(TT)                                                                       \
  if (mode == 0)                                                                                   \
    hipLaunchKernelGGL((ibftk::ecrecover_lane_kernel<0, TT>), grid, block, 0, c->stream, a);       \
  else                                                                                             \
    hipLaunchKernelGGL((ibftk::ecrecover_lane_kernel<1, TT>), grid, block, 0, c->stream, a);
    switch (tab) {
      case 1: IBFT_LAUNCH_LANE(ibftk::TAB_LDS) break;
      case 2: IBFT_LAUNCH_LANE(ibftk::TAB_PRIVATE_PREFETCH) break;
      default: IBFT_LAUNCH_LANE(ibftk::TAB_PRIVATE) break;
    }
#undef IBFT_LAUNCH_LANE
    c->last_cold_table = tab;
  }
  c->last_cold_group = CG;
  HIPCHK(c, hipGetLastError());
  if (time_it) HIPCHK(c, hipEventRecord(e1, c->stream));
  return IBFT_OK;
}

// after a fetch: build tables for keys learned since the last build (stream-ordered, asynchronous)
// Keys were learned since the last build pass (the device counter moved): tables for them, on this context's stream.
// One context builds; the others see the new tables at their next launch (state 2) and recount.
int build_new_tables(ibft_ctx *c, uint32_t learned_total, uint32_t any_slot) {
  if (!c->cache_on) return IBFT_OK;
  DeviceShared &d = *c->dev;
  std::lock_guard<std::mutex> lk(d.mu);
  if ((int32_t)(learned_total - d.learned_seen) <= 0) {  // (another context's pass may already have covered a newer count)
    recount_built(c);
    return IBFT_OK;
  }
  const uint32_t ns = d.used;
  hipLaunchKernelGGL(ibftk::qtab_claim_kernel, dim3((ns + 255) / 256), dim3(256), 0, c->stream, (uint32_t *)d.d_state.p, ns);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(ibftk::qtab_build_kernel, dim3(((ns + 63) / 64) * ibftk::QTAB_WINDOWS), dim3(64), 0, c->stream,
                     (const uint32_t *)d.d_pub.p, (const uint32_t *)d.d_state.p, (uint32_t *)d.d_qtab.p, ns);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(ibftk::qtab_commit_kernel, dim3((ns + 255) / 256), dim3(256), 0, c->stream, (uint32_t *)d.d_state.p, ns);
  HIPCHK(c, hipGetLastError());
  // which slots hold a table now: the host's shadow of the states (a learn event is rare once the set is known)
  std::vector<uint32_t> st(ns);
  HIPCHK(c, hipMemcpyAsync(st.data(), d.d_state.p, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (uint32_t i = 0; i < ns; i++) d.built[i] = st[i] == ibftk::KEY_BUILT;
  d.learned_seen = learned_total;
  if (any_slot < ns && d.built[any_slot]) d.dummy_slot = any_slot;
  d.build_epoch++;
  recount_built(c);
  return IBFT_OK;
}

// `on`: the stream the tally is enqueued on (default: the context's main stream; ibft_seals_submit passes the side stream)
int enqueue_tally(ibft_ctx *c, uint32_t n, const ibftk::set_args *set = nullptr, hipStream_t on = nullptr) {
  const hipStream_t ts = on ? on : c->stream;
  if (c->read_pending) {  // a consumer stream is still copying the previous results (ibft_seals_export_on / exchange)
    HIPCHK(c, hipStreamWaitEvent(ts, c->ev_read, 0));
    c->read_pending = false;
  }
  ibftk::tally_args t{};
  t.work_mask = (uint64_t *)c->d_mask.p;
  t.mask = (uint64_t *)c->d_mask_out.p;
  t.vidx = (const int32_t *)c->d_vidx.p;
  t.vpower32 = (const uint32_t *)c->d_vpower.p;
  t.n = n;
  t.n_validators = c->n_validators;
  t.seen = (uint32_t *)c->d_seen.p;
  t.acc = (uint64_t *)c->d_acc.p;
  t.quorum = (const uint64_t *)c->d_quorum.p;
  t.out = (uint64_t *)c->d_tally.p;
  t.host_mask = c->tally_slot >= 0 ? c->dp_mask[c->tally_slot] : c->dh_mask;
  t.host_tally = c->tally_slot >= 0 ? c->dp_tally[c->tally_slot] : c->dh_tally;
  t.set_on = set ? 1u : 0u;
  if (set) t.set = *set;
  // HasPrepareQuorum: on one device the proposer's seat joins the bitmap here; a rank of a sharded batch only counts the
  // proposer's rows and leaves the seat to the merge (exchange_unpack_kernel)
  t.prop_on = c->next_prop_on ? 1u : 0u;
  t.prop_vidx = c->next_prop_vidx;
  t.prop_seat = c->next_shard ? 0u : 1u;
  c->next_prop_on = false;
  c->next_prop_vidx = -1;
  c->next_shard = false;
  if (c->cache_on) t.learned_src = (const uint64_t *)c->dev->d_learned.p;
  if (c->comm || c->xlocal) {  // a rank of a sharded batch: the bitmap of this launch is what the exchange merges
    const size_t bytes = (size_t)((c->n_validators + 63) / 64) * 8;
    if (bytes > c->d_seen_out.cap) {
      int rc = ensure(c, c->d_seen_out, bytes);
      if (rc) return rc;
      HIPCHK(c, hipMemsetAsync(c->d_seen_out.p, 0, c->d_seen_out.cap, ts));  // (an odd trailing 32-bit word stays 0)
    }
    t.seen_out = (uint32_t *)c->d_seen_out.p;
  }
  const dim3 grid(std::max(1u, (n + ibftk::TALLY_ROWS_PER_BLOCK - 1) / ibftk::TALLY_ROWS_PER_BLOCK)), block(ibftk::TALLY_THREADS);
  // one workgroup (n ≤ 4 096): everything stays in the workgroup — no global atomics on the latency-critical sizes;
  // beyond: ticket form, each workgroup merging its LDS bitmap into the HBM one word by word.  A validator set whose
  // bitmap does not fit 48 KiB of dynamic LDS (> 393 216 validators) sends every row's bit to HBM directly.
  const size_t lds = (size_t)((c->n_validators + 31) / 32) * 4;
  t.lds_bitmap = lds <= 49152 ? 1u : 0u;
  const size_t dyn = t.lds_bitmap ? lds : 0;
  // (a side-stream tally has to FIT next to a resident verdict kernel: the one-workgroup form holds 126 registers in each of its
  // sixteen wavefronts — nothing else fits a compute unit beside it —, the ticket form 42)
  const bool single = grid.x == 1 && t.lds_bitmap && !on;
  if (single) {
    if (c->power_words == 1)
      hipLaunchKernelGGL((ibftk::tally_kernel<1, false>), grid, block, dyn, ts, t);
    else
      hipLaunchKernelGGL((ibftk::tally_kernel<4, false>), grid, block, dyn, ts, t);
  } else {
    if (c->power_words == 1)
      hipLaunchKernelGGL((ibftk::tally_kernel<1, true>), grid, block, dyn, ts, t);
    else
      hipLaunchKernelGGL((ibftk::tally_kernel<4, true>), grid, block, dyn, ts, t);
  }
  HIPCHK(c, hipGetLastError());
  c->host_direct = c->dh_mask != nullptr && c->tally_slot < 0;  // results of THIS tally are on their way to h_mask / h_tally
  if ((uint32_t)mask_words(n) >= c->mask_dirty_words) c->mask_dirty_words = 0;  // ... and it zeroed every word that held bits
  return IBFT_OK;
}

int fetch_results(ibft_ctx *c, uint32_t n, uint64_t *out_mask, ibft_tally_t *tally, bool have_tally) {
  size_t mw = (size_t)mask_words(n);
  const bool direct = c->host_direct && have_tally;  // the tally kernel already wrote h_mask / h_tally
  c->host_direct = false;
  if (!direct) {
    if (out_mask && mw)
      HIPCHK(c, hipMemcpyAsync(c->h_mask, have_tally ? c->d_mask_out.p : c->d_mask.p, mw * 8, hipMemcpyDeviceToHost,
                               c->stream));
    // d_tally holds {power_lo, power_hi, counts, has_quorum, learned|any_validator, power in full}: one copy
    if (tally && have_tally)
      HIPCHK(c, hipMemcpyAsync(c->h_tally, c->d_tally.p, (size_t)(ibftk::TALLY_OUT_PROPOSER_ROWS + 1) * 8,
                               hipMemcpyDeviceToHost, c->stream));
    // {keys learned, a learned slot}: the device-wide counter itself (a tally kernel passes it on in word 4 of its results;
    // calls without a tally — the certificate tree, plain hash batches — read it here)
    if (c->cache_on)
      HIPCHK(c, hipMemcpyAsync(c->h_tally + 4, c->dev->d_learned.p, 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->cache_on) {
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(c->h_tally + 4);
    int rcb = build_new_tables(c, lw[0], lw[1]);
    if (rcb) return rcb;
  }
  if (out_mask && mw) {
    memcpy(out_mask, c->h_mask, mw * 8);
    // clear the padding bits of the last word
    if (n & 63) out_mask[mw - 1] &= (~0ull) >> (64 - (n & 63));
  }
  if (have_tally)
    for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) c->last_wide[i] = c->h_tally[ibftk::TALLY_OUT_WIDE + i];
  if (tally) {
    memset(tally, 0, sizeof *tally);
    tally->quorum_lo = c->quorum_w[0];
    tally->quorum_hi = c->quorum_w[1];
    if (have_tally) {
      tally->power_lo = c->h_tally[0];
      tally->power_hi = c->h_tally[1];
      tally->valid_rows = (uint32_t)(c->h_tally[2] & 0xFFFFFFFFull);
      tally->distinct_senders = (uint32_t)(c->h_tally[2] >> 32);
      tally->has_quorum = (uint32_t)c->h_tally[3];
      tally->proposer_rows = (uint32_t)c->h_tally[ibftk::TALLY_OUT_PROPOSER_ROWS];
    }
  }
  return IBFT_OK;
}

// validator index of an address in the current set (−1: no validator) — the device's open-addressing table, on the host
int32_t host_lookup(const ibft_ctx *c, const uint8_t addr20[20]) {
  if (c->h_vtab.empty()) return -1;
  uint32_t a[5];
  memcpy(a, addr20, 20);
  uint32_t s = ibftk::addr_hash(a) & c->vslot_mask;
  for (uint32_t probe = 0; probe <= c->vslot_mask; probe++) {
    const uint32_t *e = &c->h_vtab[(size_t)s * 6];
    if (e[5] == 0) return -1;
    if (memcmp(e, a, 20) == 0) return (int32_t)e[5] - 1;
    s = (s + 1) & c->vslot_mask;
  }
  return -1;
}
// the next tally this context enqueues is HasPrepareQuorum with this proposer (null: plain HasQuorum)
void note_proposer(ibft_ctx *c, const uint8_t *proposer20, bool shard = false) {
  c->next_prop_on = proposer20 != nullptr;
  c->next_prop_vidx = proposer20 ? host_lookup(c, proposer20) : -1;
  c->next_shard = shard && proposer20 != nullptr;
}

int upload(ibft_ctx *c, DevBuf &b, const void *src, size_t bytes);

// rows [row0, row0 + n) of the hash column: carried proposalHash → the digest the seal signs (no-op under the identity
// convention); keep_copy: the carried hashes stay available in d_hash_copy (rows [0, n)) for the a1 compare
int apply_seal_digest(ibft_ctx *c, uint32_t row0, uint32_t n, bool keep_copy) {
  if (c->seal_digest_mode == 0 || n == 0) return IBFT_OK;
  ibftk::seal_digest_args a{};
  a.hash32 = (uint8_t *)c->d_hash.p + 32ull * row0;
  if (keep_copy) {
    int rc = ensure(c, c->d_hash_copy, (size_t)n * 32);
    if (rc) return rc;
    a.copy32 = (uint8_t *)c->d_hash_copy.p;
  }
  a.n = n;
  memcpy(a.suffix_words, c->seal_suffix_words, sizeof a.suffix_words);
  hipLaunchKernelGGL(ibftk::seal_digest_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  return IBFT_OK;
}

// d_H ← keccak256(raw ‖ BE64(round)) unless it already holds exactly that (enqueued on the context's stream)
// d_H ← keccak256(raw ‖ BE64(round)) unless it already holds exactly that.  Two steps so that a message set can put the
// hash commands BEHIND its verdict launch in host time as well (≈20 µs of runtime calls that would otherwise delay the
// uploads): note_proposal decides and pads, launch_proposal_hash enqueues upload + kernel on the side stream.
// Every reader of d_H is a synchronous call that waits for ev_H first (wait_proposal_hash), so nothing else orders the two streams.
bool note_proposal(ibft_ctx *c, const uint8_t *raw, size_t raw_len, uint64_t round) {
  uint8_t be[8];
  for (int i = 0; i < 8; i++) be[i] = (uint8_t)(round >> (8 * (7 - i)));
  if (c->have_H && c->hashed_proposal.size() == raw_len + 8 && (raw_len == 0 || memcmp(c->hashed_proposal.data(), raw, raw_len) == 0) &&
      memcmp(c->hashed_proposal.data() + raw_len, be, 8) == 0)
    return false;
  c->have_H = false;
  c->hashed_proposal.resize(raw_len + 8);
  if (raw_len) memcpy(c->hashed_proposal.data(), raw, raw_len);
  memcpy(c->hashed_proposal.data() + raw_len, be, 8);
  return true;
}
// Keccak-256 of a ‖ b on the HOST, with the device code's own permutation (keccak_dev.h compiles for both sides).
// Keccak is a sequential sponge: one host core absorbs ≈340–360 MB/s here, ≈600 MB/s on the GPU box's cores (3.6 µs for 1 KiB, ≈3 ms for 1 MiB) where the one
// wavefront that can work on a message absorbs 25 MB/s (profiles/r02_a1_sizes_v2.json: 41 ms for 1 MiB).
// the permutation for a host that has BMI1/BMI2 (andn, rorx: every x86-64 core since 2013): the same code, compiled for them
__attribute__((target("bmi,bmi2"), noinline)) static void f1600_bmi(uint64_t s[25]) { keccak::f1600(s); }
static void f1600_plain(uint64_t s[25]) { keccak::f1600(s); }
static void (*const host_f1600)(uint64_t *) = (__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2")) ? f1600_bmi : f1600_plain;

void host_keccak256(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, uint8_t out32[32]) {
  uint64_t s[25] = {0};
  uint8_t block[136];
  size_t fill = 0;
  auto absorb = [&](const uint8_t *p, size_t n) {
    while (n) {
      if (fill == 0 && n >= 136) {  // whole blocks straight from the source
        for (int i = 0; i < 17; i++) {
          uint64_t w;
          memcpy(&w, p + 8 * i, 8);
          s[i] ^= w;
        }
        host_f1600(s);
        p += 136;
        n -= 136;
        continue;
      }
      const size_t k = std::min(n, (size_t)136 - fill);
      memcpy(block + fill, p, k);
      fill += k;
      p += k;
      n -= k;
      if (fill == 136) {
        for (int i = 0; i < 17; i++) {
          uint64_t w;
          memcpy(&w, block + 8 * i, 8);
          s[i] ^= w;
        }
        host_f1600(s);
        fill = 0;
      }
    }
  };
  if (na) absorb(a, na);
  if (nb) absorb(b, nb);
  memset(block + fill, 0, 136 - fill);
  block[fill] ^= 0x01;
  block[135] ^= 0x80;
  for (int i = 0; i < 17; i++) {
    uint64_t w;
    memcpy(&w, block + 8 * i, 8);
    s[i] ^= w;
  }
  host_f1600(s);
  memcpy(out32, s, 32);
}

int launch_proposal_hash(ibft_ctx *c) {
  if (c->hash_on_host) {
    // the digest is computed here, on the calling thread — behind the verdict launch of a message set, which is already
    // running on the device — and reaches d_H through the side stream like the device-computed one did
    if (c->H_copy_issued) HIPCHK(c, hipEventSynchronize(c->ev_H));  // the staging slot is free again
    uint8_t *stage = reinterpret_cast<uint8_t *>(c->h_digest) + 32;
    host_keccak256(c->hashed_proposal.data(), c->hashed_proposal.size(), nullptr, 0, stage);
    memcpy(c->H_host, stage, 32);
    HIPCHK(c, hipMemcpyAsync(c->d_H.p, stage, 32, hipMemcpyHostToDevice, c->hstream));
    HIPCHK(c, hipEventRecord(c->ev_H, c->hstream));
    c->H_copy_issued = true;
    c->H_pending = true;
    c->have_H = true;
    c->have_H_host = true;
    return IBFT_OK;
  }
  c->have_H_host = false;
  // Keccak padding on the host: pad10*1 up to a multiple of the 136-byte rate (the kernel XORs whole 64-bit words)
  const size_t mlen = c->hashed_proposal.size();
  const size_t blocks = mlen / 136 + 1;
  c->padded_proposal.assign(blocks * 136, 0);
  memcpy(c->padded_proposal.data(), c->hashed_proposal.data(), mlen);
  c->padded_proposal[mlen] ^= 0x01;
  c->padded_proposal[blocks * 136 - 1] ^= 0x80;
  int rc = ensure(c, c->d_raw, blocks * 136);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_raw.p, c->padded_proposal.data(), blocks * 136, hipMemcpyHostToDevice, c->hstream));
  hipLaunchKernelGGL(ibftk::proposal_hash_kernel, dim3(1), dim3(64), 0, c->hstream, (const uint64_t *)c->d_raw.p,
                     (uint32_t)blocks, (uint64_t *)c->d_H.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev_H, c->hstream));
  c->H_pending = true;
  c->have_H = true;  // valid once the main stream has waited for ev_H (wait_proposal_hash): every reader does that first
  return IBFT_OK;
}
int ensure_proposal_hash(ibft_ctx *c, const uint8_t *raw, size_t raw_len, uint64_t round) {
  return note_proposal(c, raw, raw_len, round) ? launch_proposal_hash(c) : IBFT_OK;
}
// the main stream goes on only when d_H holds the hash launched by ensure_proposal_hash
int wait_proposal_hash(ibft_ctx *c) {
  if (!c->H_pending) return IBFT_OK;
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_H, 0));
  c->H_pending = false;
  return IBFT_OK;
}

int upload(ibft_ctx *c, DevBuf &b, const void *src, size_t bytes) {
  if (!bytes) return IBFT_OK;
  int rc = ensure(c, b, bytes);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  return IBFT_OK;
}

// Blocks handed out by ibft_pinned_alloc: a column that lies inside one can be read by the device directly.
struct PinnedRegistry {
  std::mutex mu;
  std::map<uintptr_t, size_t> blocks;
  bool covers(const void *p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(mu);
    auto it = blocks.upper_bound(a);
    if (it == blocks.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
  }
};
PinnedRegistry &pinned_registry() {
  static PinnedRegistry r;
  return r;
}

// The host→HBM column copies of one batch.  flush(): ONE gather launch when every source is pinned (the device
// reads the columns itself), otherwise one hipMemcpyAsync per column as before.
struct ColumnCopies {
  struct Seg {
    void *dst;
    const void *src;
    size_t bytes;
  };
  Seg seg[ibftk::GATHER_MAX];
  int n = 0;
  // optional: the PayloadNoSig column of a message set.  Pinned: it is never copied — digest blocks of the gather launch
  // hash it straight from the host column into `digest_dst` (digest_done).  Pageable: payload and offsets are copied to
  // pay_dst / off_dst like any column and the caller launches payload_digest_kernel.
  const void *pay_src = nullptr, *off_src = nullptr;
  void *pay_dst = nullptr, *off_dst = nullptr, *digest_dst = nullptr;
  size_t pay_bytes = 0, rows = 0;
  bool digest_done = false;
  void add(void *dst, const void *src, size_t bytes) {
    if (bytes && n < ibftk::GATHER_MAX) seg[n++] = Seg{dst, src, bytes};
  }
  void payload(void *pdst, const void *psrc, size_t pbytes, void *odst, const void *osrc, size_t n_rows, void *ddst) {
    pay_dst = pdst; pay_src = psrc; pay_bytes = pbytes; off_dst = odst; off_src = osrc; rows = n_rows; digest_dst = ddst;
  }
  int flush(ibft_ctx *c) {
    const bool job = rows != 0;
    if (n == 0 && !job) return IBFT_OK;
    bool pinned = c->gather_pinned;
    for (int i = 0; i < n && pinned; i++) pinned = seg[i].bytes < (1ull << 31) && pinned_registry().covers(seg[i].src, seg[i].bytes);
    if (pinned && job)
      pinned = pay_bytes < (1ull << 31) && pinned_registry().covers(off_src, (rows + 1) * 4) &&
               (pay_bytes == 0 || pinned_registry().covers(pay_src, pay_bytes));
    if (!pinned) {
      if (job) {
        if (pay_bytes) HIPCHK(c, hipMemcpyAsync(pay_dst, pay_src, pay_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(off_dst, off_src, (rows + 1) * 4, hipMemcpyHostToDevice, c->stream));
      }
      for (int i = 0; i < n; i++) HIPCHK(c, hipMemcpyAsync(seg[i].dst, seg[i].src, seg[i].bytes, hipMemcpyHostToDevice, c->stream));
      n = 0;
      return IBFT_OK;
    }
    if (job && !c->digest_in_gather) {  // the payload and its offsets as two more copied columns; the caller hashes from HBM
      if (n + 2 > ibftk::GATHER_MAX) return IBFT_E_INVAL;
      if (pay_bytes) seg[n++] = Seg{pay_dst, pay_src, pay_bytes};
      seg[n++] = Seg{off_dst, off_src, (rows + 1) * 4};
    }
    const bool fused = job && c->digest_in_gather;
    ibftk::gather_args a{};
    uint32_t blocks = 0;
    for (int i = 0; i < n; i++) {
      a.src[i] = (const uint8_t *)seg[i].src;
      a.dst[i] = (uint8_t *)seg[i].dst;
      a.bytes[i] = (uint32_t)seg[i].bytes;
      a.first_block[i] = blocks;
      blocks += (uint32_t)((seg[i].bytes + ibftk::GATHER_BLOCK_BYTES - 1) / ibftk::GATHER_BLOCK_BYTES);
    }
    a.first_block[n] = blocks;
    a.n = (uint32_t)n;
    if (fused) {
      a.pay_src = (const uint8_t *)pay_src;
      a.off_src = (const uint32_t *)off_src;
      a.digest_dst = (uint8_t *)digest_dst;
      a.n_rows = (uint32_t)rows;
      a.pay_bytes = (uint32_t)pay_bytes;
      blocks += (uint32_t)((rows + 63) / 64);
      digest_done = true;
    }
    if (fused)
      hipLaunchKernelGGL(ibftk::gather_columns_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, a);
    else
      hipLaunchKernelGGL(ibftk::gather_columns_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    c->gathers++;
    n = 0;
    return IBFT_OK;
  }
};

// ---- RCCL, loaded on first use: a single-GPU deployment never needs the library ----------------------
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(ncclComm_t, int *) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi *rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {getenv("IBFT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      if (!nm || !*nm) continue;
      if ((api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!api.handle) return;
#define IBFT_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name))
    IBFT_SYM(GetUniqueId, "ncclGetUniqueId");
    IBFT_SYM(CommInitRank, "ncclCommInitRank");
    IBFT_SYM(CommDestroy, "ncclCommDestroy");
    IBFT_SYM(AllReduce, "ncclAllReduce");
    IBFT_SYM(CommCount, "ncclCommCount");
    IBFT_SYM(CommUserRank, "ncclCommUserRank");
    IBFT_SYM(CommCuDevice, "ncclCommCuDevice");
    IBFT_SYM(GroupStart, "ncclGroupStart");
    IBFT_SYM(GroupEnd, "ncclGroupEnd");
    IBFT_SYM(GetErrorString, "ncclGetErrorString");
#undef IBFT_SYM
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GroupStart && api.GroupEnd;
  });
  return api.ok ? &api : nullptr;
}
#define NCCLCHK(ctx, api, expr)                                                                              \
  do {                                                                                                       \
    ncclResult_t r_ = (expr);                                                                                \
    if (r_ != ncclSuccess) {                                                                                 \
      (ctx)->last_error = std::string(#expr) + ": " + ((api)->GetErrorString ? (api)->GetErrorString(r_) : "rccl error"); \
      return IBFT_E_RCCL;                                                                                    \
    }                                                                                                        \
  } while (0)

void comm_release(ibft_ctx *c) {
  if (c->comm) {
    if (RcclApi *a = rccl()) (void)a->CommDestroy(c->comm);
    c->comm = nullptr;
  }
  for (int i = 0; i < 2; i++) {
    if (c->ev_xdone[i]) (void)hipEventDestroy(c->ev_xdone[i]);
    c->ev_xdone[i] = nullptr;
    if (c->h_xres[i]) (void)hipHostFree(c->h_xres[i]);
    c->h_xres[i] = c->dh_xres[i] = nullptr;
  }
  c->h_xres_words = 0;
  if (c->ev_xpack) (void)hipEventDestroy(c->ev_xpack);
  c->ev_xpack = nullptr;
  c->xlocal = false;
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  c->xstream = nullptr;
  c->xrank = 0;
  c->xworld = 1;
  c->x_issued = c->x_fetched = 0;
}

// shard layout (pure): contiguous 64-aligned row ranges, the last rank takes the remainder
uint64_t shard_rows_per_rank(uint64_t n_total, uint32_t world) {
  return (((n_total + world - 1) / world) + 63) / 64 * 64;
}

int ensure_handoff_events(ibft_ctx *c) {
  if (!c->ev_ready) {
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_read, hipEventDisableTiming));
  }
  return IBFT_OK;
}

struct xplan {
  uint32_t slot, w, total_words, seen_words, n_pieces, slots, my_words, K;
  // HasPrepareQuorum of THIS exchange (fixed when the plan is made: a tally some other call enqueues on the context between
  // pack and unpack must not change what the merge applies)
  bool prop_on;
  int32_t prop_vidx;
};
// u64 slots of the exchange buffer: K verdict arrays, one bitmap segment per rank, the count of valid rows
uint32_t exchange_slots(uint32_t K, uint32_t words_per_rank, uint32_t world, uint32_t n_validators) {
  return K * words_per_rank * world + world * ((n_validators + 63) / 64) + 1;
}
// before the collective: order behind the tally, pack this rank's words + its sender bitmap into the exchange buffer.
// n_local rows of this rank (seal batch: the staged one; message set: the last set), K verdict arrays.
int exchange_pre(ibft_ctx *c, uint64_t n_total, xplan &x, uint32_t K = 1, const uint8_t *proposer20 = nullptr) {
  if (!(c->comm || c->xlocal) || !c->have_valset) return (c->comm || c->xlocal) ? IBFT_E_NOVALSET : IBFT_E_INVAL;
  if (c->x_issued - c->x_fetched >= 2) {
    c->last_error = "two exchanges already in flight: call ibft_seals_fetch_merged first";
    return IBFT_E_INVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t per = shard_rows_per_rank(n_total, c->xworld);
  const uint32_t n_local = K == 2 ? c->set_n : c->staged_n;
  x.K = K;
  x.prop_on = proposer20 != nullptr;
  x.prop_vidx = proposer20 ? host_lookup(c, proposer20) : -1;
  x.slot = c->x_issued & 1u;
  x.w = (uint32_t)(per / 64);
  x.total_words = x.w * c->xworld;
  x.seen_words = (c->n_validators + 63) / 64;
  x.n_pieces = 2 * c->power_words;
  x.slots = exchange_slots(K, x.w, c->xworld, c->n_validators);
  x.my_words = (uint32_t)mask_words(n_local);
  const uint64_t lo = std::min<uint64_t>((uint64_t)c->xrank * per, n_total), hi = std::min<uint64_t>(lo + per, n_total);
  if ((uint64_t)n_local != hi - lo) {
    c->last_error = "the resident batch is not this rank's shard of n_total rows (ibft_shard_range)";
    return IBFT_E_INVAL;
  }
  if (!c->d_seen_out.p) {
    c->last_error = "no tally has run since the context joined the exchange";
    return IBFT_E_INVAL;
  }
  int rc;
  if ((rc = ensure(c, c->d_xbuf[x.slot], (size_t)x.slots * 8))) return rc;
  const size_t res_words = (size_t)K * x.total_words + 16;
  if ((rc = ensure(c, c->d_xres[x.slot], res_words * 8))) return rc;
  if (res_words > c->h_xres_words) {  // (re)allocate the mapped result buffers; nothing is in flight on a fresh size
    if (c->x_issued != c->x_fetched) HIPCHK(c, hipStreamSynchronize(c->xstream));
    for (int i = 0; i < 2; i++) {
      if (c->h_xres[i]) (void)hipHostFree(c->h_xres[i]);
      c->h_xres[i] = c->dh_xres[i] = nullptr;
      if (hipHostMalloc((void **)&c->h_xres[i], res_words * 8) != hipSuccess) return IBFT_E_NOMEM;
      void *d = nullptr;
      if (hipHostGetDevicePointer(&d, c->h_xres[i], 0) == hipSuccess) c->dh_xres[i] = (uint64_t *)d;
    }
    c->h_xres_words = res_words;
  }
  if ((rc = ensure_handoff_events(c))) return rc;
  HIPCHK(c, hipEventRecord(c->ev_ready, c->stream));  // everything launched so far, the tally included
  HIPCHK(c, hipStreamWaitEvent(c->xstream, c->ev_ready, 0));
  ibftk::exchange_pack_args pa{};
  if (K == 2) {
    pa.mask[0] = (const uint64_t *)c->d_set.p;
    pa.mask[1] = (const uint64_t *)c->d_set.p + mask_words(c->max_rows);
  } else {
    pa.mask[0] = pa.mask[1] = (const uint64_t *)c->d_mask_out.p;
  }
  pa.seen = (const uint32_t *)c->d_seen_out.p;
  pa.tally_out = (const uint64_t *)c->d_tally.p;
  pa.xbuf = (uint64_t *)c->d_xbuf[x.slot].p;
  pa.K = K;
  pa.world = c->xworld;
  pa.rank = c->xrank;
  pa.words_per_rank = x.w;
  pa.my_words = x.my_words;
  pa.seen_words = x.seen_words;
  hipLaunchKernelGGL(ibftk::exchange_pack_kernel, dim3((x.slots + 255) / 256), dim3(256), 0, c->xstream, pa);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev_read, c->xstream));
  c->read_pending = true;  // the next tally overwrites d_mask_out / d_set / d_tally / d_seen_out only after the pack has read them
  return IBFT_OK;
}
int exchange_collective(ibft_ctx *c, RcclApi *api, const xplan &x) {
  NCCLCHK(c, api, api->AllReduce(c->d_xbuf[x.slot].p, c->d_xbuf[x.slot].p, x.slots, ncclUint64, ncclSum, c->comm, c->xstream));
  return IBFT_OK;
}
// The same step without a communicator: every rank's buffer is addressable from rank 0's device (one device listed
// several times, or peer access), so rank 0's exchange stream sums the buffers in place of ncclAllReduce — behind every
// rank's pack, in front of every rank's unpack.
int exchange_collective_local(const std::vector<ibft_ctx *> &ctx, const std::vector<xplan> &plan) {
  ibft_ctx *c0 = ctx[0];
  ibftk::exchange_sum_args sa{};
  sa.world = (uint32_t)ctx.size();
  sa.slots = plan[0].slots;
  for (size_t i = 0; i < ctx.size(); i++) {
    ibft_ctx *c = ctx[i];
    if (plan[i].slots != sa.slots) return IBFT_E_INVAL;
    sa.buf[i] = (uint64_t *)c->d_xbuf[plan[i].slot].p;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev_xpack, c->xstream));
  }
  HIPCHK(c0, hipSetDevice(c0->device));
  for (size_t i = 1; i < ctx.size(); i++) HIPCHK(c0, hipStreamWaitEvent(c0->xstream, ctx[i]->ev_xpack, 0));
  hipLaunchKernelGGL(ibftk::exchange_sum_kernel, dim3((sa.slots + 255) / 256), dim3(256), 0, c0->xstream, sa);
  HIPCHK(c0, hipGetLastError());
  HIPCHK(c0, hipEventRecord(c0->ev_xpack, c0->xstream));
  for (size_t i = 1; i < ctx.size(); i++) {
    HIPCHK(ctx[i], hipSetDevice(ctx[i]->device));
    HIPCHK(ctx[i], hipStreamWaitEvent(ctx[i]->xstream, c0->ev_xpack, 0));
  }
  return IBFT_OK;
}
int exchange_post(ibft_ctx *c, uint64_t n_total, const xplan &x) {
  HIPCHK(c, hipSetDevice(c->device));
  ibftk::exchange_unpack_args ua{};
  ua.xbuf = (const uint64_t *)c->d_xbuf[x.slot].p;
  ua.vpower32 = (const uint32_t *)c->d_vpower.p;
  ua.quorum = (const uint64_t *)c->d_quorum.p;
  ua.dst = (uint64_t *)c->d_xres[x.slot].p;
  ua.host_dst = c->dh_xres[x.slot];
  ua.K = x.K;
  ua.world = c->xworld;
  ua.words_per_rank = x.w;
  ua.seen_words = x.seen_words;
  ua.n_pieces = x.n_pieces;
  ua.n_validators = c->n_validators;
  ua.prop_on = x.prop_on ? 1u : 0u;
  ua.prop_vidx = x.prop_vidx;
  const uint32_t copy_blocks = (x.K * x.total_words + ibftk::XUNPACK_THREADS - 1) / ibftk::XUNPACK_THREADS;
  hipLaunchKernelGGL(ibftk::exchange_unpack_kernel, dim3(copy_blocks + 1), dim3(ibftk::XUNPACK_THREADS), 0, c->xstream, ua);
  HIPCHK(c, hipGetLastError());
  if (!c->dh_xres[x.slot])
    HIPCHK(c, hipMemcpyAsync(c->h_xres[x.slot], c->d_xres[x.slot].p, ((size_t)x.K * x.total_words + 16) * 8,
                             hipMemcpyDeviceToHost, c->xstream));
  HIPCHK(c, hipEventRecord(c->ev_xdone[x.slot], c->xstream));
  c->x_total[x.slot] = n_total;
  c->x_K[x.slot] = x.K;
  c->x_issued++;
  return IBFT_OK;
}
// out_mask: the first verdict array; out_mask2: the second one of a message-set exchange (K = 2)
int fetch_merged_locked(ibft_ctx *c, uint64_t *out_mask, ibft_tally_t *tally, uint64_t *out_mask2 = nullptr) {
  if (c->x_fetched == c->x_issued) return IBFT_E_INVAL;
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t slot = c->x_fetched & 1u;
  HIPCHK(c, hipEventSynchronize(c->ev_xdone[slot]));
  c->x_fetched++;
  if (c->cache_on && c->dh_tally) {  // keys this shard's rows taught the device (delivered by the tally kernel) → tables
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(c->h_tally + 4);
    int rcb = build_new_tables(c, lw[0], lw[1]);
    if (rcb) return rcb;
  }
  const uint64_t n_total = c->x_total[slot];
  const uint32_t K = c->x_K[slot];
  const uint64_t per = shard_rows_per_rank(n_total, c->xworld);
  const size_t total_words = (size_t)(per / 64) * c->xworld, mw = (size_t)((n_total + 63) / 64);
  const uint64_t *res = c->h_xres[slot];
  if (out_mask && mw) memcpy(out_mask, res, mw * 8);
  if (out_mask2 && mw && K == 2) memcpy(out_mask2, res + total_words, mw * 8);
  const uint64_t *t = res + (size_t)K * total_words;
  for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) c->last_wide[i] = t[4 + i];
  if (tally) {
    memset(tally, 0, sizeof *tally);
    tally->quorum_lo = c->quorum_w[0];
    tally->quorum_hi = c->quorum_w[1];
    tally->power_lo = t[0];
    tally->power_hi = t[1];
    tally->valid_rows = (uint32_t)(t[2] & 0xFFFFFFFFull);
    tally->distinct_senders = (uint32_t)(t[2] >> 32);
    tally->has_quorum = (uint32_t)t[3];
    tally->shard_overlap = (uint32_t)t[4 + ibftk::TALLY_SUM_WORDS];
    tally->proposer_rows = (uint32_t)t[5 + ibftk::TALLY_SUM_WORDS];
  }
  return IBFT_OK;
}
int comm_attach(ibft_ctx *c, ncclComm_t comm, uint32_t rank, uint32_t world) {
  HIPCHK(c, hipSetDevice(c->device));
  c->comm = comm;
  c->xlocal = comm == nullptr;
  c->xrank = rank;
  c->xworld = world;
  HIPCHK(c, hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_xdone[i], hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_xpack, hipEventDisableTiming));
  return IBFT_OK;
}

}  // namespace

// One host thread per device of a group: staging and launching a shard costs tens of microseconds of host time per
// device, which one thread would pay W times in a row.
class GroupWorker {
 public:
  GroupWorker() : th_([this] { loop(); }) {}
  ~GroupWorker() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  void post(std::function<int()> fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      task_ = std::move(fn);
      busy_ = true;
    }
    cv_.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return !busy_; });
    return rc_;
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [this] { return stop_ || (busy_ && task_); });
      if (stop_) return;
      std::function<int()> fn = std::move(task_);
      task_ = nullptr;
      lk.unlock();
      const int rc = fn();
      lk.lock();
      rc_ = rc;
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::function<int()> task_;
  bool busy_ = false, stop_ = false;
  int rc_ = 0;
  std::thread th_;
};

struct ibft_group {
  std::mutex mu;
  std::vector<ibft_ctx *> ctx;
  bool local = false;  // the collective is exchange_sum_kernel on rank 0's exchange stream, not ncclAllReduce
  std::vector<std::unique_ptr<GroupWorker>> worker;
  // run fn(i) for every rank on its own thread; the first non-zero return code (by rank) is the result
  int each(const std::function<int(uint32_t)> &fn) {
    if (ctx.size() == 1) return fn(0);
    for (uint32_t i = 0; i < ctx.size(); i++) worker[i]->post([&fn, i] { return fn(i); });
    int rc = IBFT_OK;
    for (uint32_t i = 0; i < ctx.size(); i++) {
      const int r = worker[i]->wait();
      if (r && !rc) rc = r;
    }
    return rc;
  }
};

extern "C" {

static void key_cache_unmap(ibft_ctx *c);

int ibft_version(void) { return 4; }  // 2: ibft_tally_t.proposer_rows, proposer20 arguments, ibft_tally_prepare, ibft_comm_info;
                                      // 3: ibft_seals_stage_next / _swap / _submit / _collect, ibft_last_cold_table, ibft_comm_preload, ibft_seals_rows, ibft_issue_probe;
                                      // 4: ibft_pipeline_stats (the side-stream tally of submitted passes changes no signature)

const char *ibft_strerror(int code) {
  switch (code) {
    case IBFT_OK: return "ok";
    case IBFT_E_INVAL: return "invalid argument";
    case IBFT_E_NODEVICE: return "no usable gfx950 device";
    case IBFT_E_NOMEM: return "out of memory";
    case IBFT_E_HIP: return "HIP runtime error";
    case IBFT_E_NOVALSET: return "validator set not configured";
    case IBFT_E_POWER: return "total voting power is zero or less";
    case IBFT_E_TOOBIG: return "batch exceeds max_rows";
    case IBFT_E_RCCL: return "RCCL unavailable or a collective failed";
    default: return "unknown error";
  }
}

const char *ibft_last_error(const ibft_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int ibft_ctx_create(const ibft_cfg *cfg, ibft_ctx **out) {
  if (!out) return IBFT_E_INVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return IBFT_E_NODEVICE;
  int dev = cfg ? cfg->device : 0;
  if (dev < 0 || dev >= ndev) return IBFT_E_NODEVICE;
  if (hipSetDevice(dev) != hipSuccess) return IBFT_E_NODEVICE;
  // experiment: how much of a step is the host waking up?  IBFT_SPIN=1 asks the runtime to spin in synchronize calls
  if (const char *e = getenv("IBFT_SPIN"))
    if (atoi(e) == 1) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return IBFT_E_NODEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return IBFT_E_NODEVICE;  // gfx950-only code object

  ibft_ctx *c = new (std::nothrow) ibft_ctx();
  if (!c) return IBFT_E_NOMEM;
  c->device = dev;
  c->flags = cfg ? cfg->flags : 0;
  c->max_rows = (cfg && cfg->max_rows) ? cfg->max_rows : DEFAULT_MAX_ROWS;
  c->kernel = cfg ? cfg->kernel : IBFT_KERNEL_AUTO;
  c->cold_group_auto = c->kernel != IBFT_KERNEL_LANE;
  if (const char *e = getenv("IBFT_SPLIT_LARGE")) c->split_large = strcmp(e, "0") != 0;
  if (const char *e = getenv("IBFT_SIDE_TALLY")) c->side_tally = strcmp(e, "0") == 0 ? 0 : strcmp(e, "1") == 0 ? 1 : 2;
  if (const char *e = getenv("IBFT_COLD_TABLE")) {
    if (!strcmp(e, "lds")) c->cold_table_force = 1;
    else if (!strcmp(e, "private")) c->cold_table_force = 2;
    else if (!strcmp(e, "private2")) c->cold_table_force = 3;
  }
  if (const char *e = getenv("IBFT_COLD_LANES")) {
    const int g = atoi(e);
    if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 64 || g == 128) c->cold_group_force = (uint32_t)g;
  }
  if (const char *e = getenv("IBFT_WARM_LANES")) {
    const int g = atoi(e);
    if (g >= 1 && g <= 64 && (g & (g - 1)) == 0) c->warm_group_force = (uint32_t)g;
  }
  if (const char *e = getenv("IBFT_NO_EVENTS"))
    if (atoi(e) == 1) c->time_every = 0;
  if (getenv("IBFT_NO_GATHER")) c->gather_pinned = false;
  if (const char *e = getenv("IBFT_PROPOSAL_HASH")) c->hash_on_host = strcmp(e, "device") != 0;
  if (const char *e = getenv("IBFT_CERT_OVERLAP")) c->cert_overlap = atoi(e) != 0;
  if (getenv("IBFT_DIGEST_FUSION")) c->digest_in_gather = true;
  if (const char *e = getenv("IBFT_WAVE_ROWS_MAX")) c->wave_rows_max = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("IBFT_PAIR_ROWS_MAX")) c->pair_rows_max = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("IBFT_ROWS_KERNEL_MAX")) c->rows_kernel_max = (uint32_t)strtoul(e, nullptr, 10);
  int rc = IBFT_OK;
  do {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = IBFT_E_HIP; break; }
    if (hipStreamCreateWithFlags(&c->hstream, hipStreamNonBlocking) != hipSuccess) { rc = IBFT_E_HIP; break; }
    if (hipEventCreateWithFlags(&c->ev_H, hipEventDisableTiming) != hipSuccess) { rc = IBFT_E_HIP; break; }
    if ((rc = alloc_rows(c, c->max_rows))) break;
    if (hipMemsetAsync(c->d_acc.p, 0, c->d_acc.cap, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_tally.p, 0, c->d_tally.cap, c->stream) != hipSuccess) { rc = IBFT_E_HIP; break; }
    if (hipHostMalloc((void **)&c->h_mask, (size_t)mask_words(c->max_rows) * 8 + 64) != hipSuccess) { rc = IBFT_E_NOMEM; break; }
    if (hipHostMalloc((void **)&c->h_tally, 128) != hipSuccess) { rc = IBFT_E_NOMEM; break; }
    if (hipHostMalloc((void **)&c->h_digest, 64) != hipSuccess) { rc = IBFT_E_NOMEM; break; }
    if (hipHostMalloc((void **)&c->h_set, (size_t)mask_words(c->max_rows) * 16 + 64) != hipSuccess) { rc = IBFT_E_NOMEM; break; }
    if ((rc = ensure(c, c->d_set, (size_t)mask_words(c->max_rows) * 16))) break;
    if (hipHostMalloc((void **)&c->h_class, (size_t)c->max_rows + 64) != hipSuccess) { rc = IBFT_E_NOMEM; break; }
    if ((rc = ensure(c, c->d_class, (size_t)c->max_rows + 64))) break;
    // zero-copy result delivery (tally_kernel writes the verdict words and its own result into the
    // pinned buffers); IBFT_NO_HOST_DIRECT=1 keeps the two device-to-host copies instead
    if (!getenv("IBFT_NO_HOST_DIRECT")) {
      void *dm = nullptr, *dt = nullptr;
      if (hipHostGetDevicePointer(&dm, c->h_mask, 0) == hipSuccess && hipHostGetDevicePointer(&dt, c->h_tally, 0) == hipSuccess) {
        c->dh_mask = (uint64_t *)dm;
        c->dh_tally = (uint64_t *)dt;
        void *ds = nullptr;
        if (hipHostGetDevicePointer(&ds, c->h_set, 0) == hipSuccess) c->dh_set = (uint64_t *)ds;
        void *dc = nullptr;
        if (hipHostGetDevicePointer(&dc, c->h_class, 0) == hipSuccess) c->dh_class = (uint8_t *)dc;
      }
    }
    {  // the device's shared object: created (and the G table built) by the first context of the device
      std::lock_guard<std::mutex> glk(g_devices_mu);
      c->dev = g_devices[c->device].lock();
      if (!c->dev) {
        c->dev = std::make_shared<DeviceShared>();
        c->dev->device = c->device;
        g_devices[c->device] = c->dev;
      }
    }
    {
      std::lock_guard<std::mutex> dlk(c->dev->mu);
      if (!c->dev->d_gtab.p) {
        if ((rc = ensure(c, c->dev->d_gtab, (size_t)ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES * ibftk::GTAB_ENTRY_DWORDS * 4))) break;
        int threads = 64, total = ibftk::GTAB_WINDOWS * ibftk::GTAB_ENTRIES;
        hipLaunchKernelGGL(ibftk::gtab_build_kernel, dim3((total + threads - 1) / threads), dim3(threads), 0,
                           c->stream, (uint32_t *)c->dev->d_gtab.p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
          release(c->dev->d_gtab);
          rc = IBFT_E_HIP;
          break;
        }
      }
    }
  } while (0);
  if (rc) {
    ibft_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return IBFT_OK;
}

void ibft_ctx_destroy(ibft_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->hstream) (void)hipStreamSynchronize(c->hstream);
  if (c->tstream) (void)hipStreamSynchronize(c->tstream);
  for (DevBuf *b : {&c->d_hash, &c->d_sig, &c->d_signer, &c->d_pre, &c->d_hash_len, &c->d_payload,
                    &c->d_off, &c->d_raw, &c->d_mask, &c->d_mask_out, &c->d_vidx, &c->d_tally, &c->d_H,
                    &c->d_vtab, &c->d_vpower, &c->d_vslot,
                    &c->d_warm_done, &c->d_seen, &c->d_acc, &c->d_quorum, &c->d_wire_rows, &c->d_seal,
                    &c->d_xbuf[0], &c->d_xbuf[1], &c->d_xres[0], &c->d_xres[1], &c->d_set, &c->d_noseal, &c->d_class,
                    &c->d_cert_nodes, &c->d_cert_span, &c->d_cert_count, &c->d_cert_prop, &c->d_cert_masks, &c->d_cert_total,
                    &c->d_cert_slot, &c->d_cert_tiles, &c->d_hash_copy, &c->d_seen_out, &c->d_hash_nx, &c->d_sig_nx,
                    &c->d_signer_nx, &c->d_pre_nx, &c->d_mask_b, &c->d_vidx_b})
    release(*b);
  if (c->tstream) {
    (void)hipStreamSynchronize(c->tstream);
    (void)hipStreamDestroy(c->tstream);
  }
  for (int i = 0; i < 2; i++)
    if (c->ev_rec[i]) (void)hipEventDestroy(c->ev_rec[i]);
  if (c->cstream) {
    (void)hipStreamSynchronize(c->cstream);
    (void)hipStreamDestroy(c->cstream);
  }
  if (c->ev_staged) (void)hipEventDestroy(c->ev_staged);
  if (c->ev_cols_read) (void)hipEventDestroy(c->ev_cols_read);
  if (c->dev) {
    {
      std::lock_guard<std::mutex> dlk(c->dev->mu);
      key_cache_unmap(c);
    }
    std::lock_guard<std::mutex> glk(g_devices_mu);
    if (c->dev.use_count() == 1) {  // the last context of the device: its tables go with it
      for (DevBuf *b : {&c->dev->d_gtab, &c->dev->d_pub, &c->dev->d_state, &c->dev->d_qtab, &c->dev->d_learned}) release(*b);
      g_devices.erase(c->device);
    }
    c->dev.reset();
  }
  if (c->h_cert_total) (void)hipHostFree(c->h_cert_total);
  if (c->ev_cert_fork) (void)hipEventDestroy(c->ev_cert_fork);
  if (c->ev_cert_join) (void)hipEventDestroy(c->ev_cert_join);
  comm_release(c);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_read) (void)hipEventDestroy(c->ev_read);
  for (int i = 0; i < 2; i++) {
    if (c->p_mask[i]) (void)hipHostFree(c->p_mask[i]);
    if (c->p_tally[i]) (void)hipHostFree(c->p_tally[i]);
    if (c->ev_pass[i]) (void)hipEventDestroy(c->ev_pass[i]);
  }
  if (c->h_mask) (void)hipHostFree(c->h_mask);
  if (c->h_tally) (void)hipHostFree(c->h_tally);
  if (c->h_digest) (void)hipHostFree(c->h_digest);
  if (c->h_set) (void)hipHostFree(c->h_set);
  if (c->h_class) (void)hipHostFree(c->h_class);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  if (c->ev_H) (void)hipEventDestroy(c->ev_H);
  if (c->hstream) (void)hipStreamDestroy(c->hstream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// powers: n × pw little-endian 64-bit words (pw = 1: ibft_set_validators, pw = 4: ibft_set_validators_u256)
// Drop this context's references to its slots; a slot nobody refers to any more is free for another address.
static void key_cache_unmap(ibft_ctx *c) {  // c->dev->mu held
  DeviceShared &d = *c->dev;
  for (uint32_t sl : c->vslot) {
    if (sl == 0xFFFFFFFFu || sl >= d.refs.size() || d.refs[sl] == 0) continue;
    if (--d.refs[sl] == 0) {
      d.slot_of.erase(d.addr_of[sl]);
      d.free_slots.push_back(sl);
    }
  }
  c->vslot.clear();
  c->my_built = 0;
  c->cache_on = false;
}

// Give every validator of the new set its slot (c->dev->mu held).  Grows the pool when the set needs more slots than it
// has (within IBFT_QTAB_BUDGET_GB, default 64: 655 KB per slot); validators beyond the budget simply have no slot.
static int key_cache_map(ibft_ctx *c, const std::vector<KeyAddr> &vaddr) {
  DeviceShared &d = *c->dev;
  const size_t nv = vaddr.size();
  if (nv == 0) {
    key_cache_unmap(c);
    return IBFT_E_INVAL;
  }
  // new references first, then the old ones go: an address in both sets never drops to zero in between
  std::vector<uint32_t> ns(nv, 0xFFFFFFFFu);
  size_t need_new = 0;
  for (size_t v = 0; v < nv; v++) {
    auto it = d.slot_of.find(vaddr[v]);
    if (it != d.slot_of.end()) {
      ns[v] = it->second;
      d.refs[it->second]++;
    } else {
      need_new++;
    }
  }
  {
    std::vector<uint32_t> old;
    old.swap(c->vslot);
    for (uint32_t sl : old) {
      if (sl == 0xFFFFFFFFu || d.refs[sl] == 0) continue;
      if (--d.refs[sl] == 0) {
        d.slot_of.erase(d.addr_of[sl]);
        d.free_slots.push_back(sl);
      }
    }
  }
  size_t budget = 64ull << 30;
  if (const char *e = getenv("IBFT_QTAB_BUDGET_GB")) budget = (size_t)strtoull(e, nullptr, 10) << 30;
  const size_t slot_bytes = (size_t)ibftk::QTAB_DWORDS_PER_VALIDATOR * 4;
  const size_t max_slots = budget / slot_bytes;
  const size_t avail = d.free_slots.size() + (d.cap - d.used);
  if (need_new > avail) {  // grow: the existing tables move to the new buffers (device copies), the pointers change
    size_t want = std::min(max_slots, std::max((size_t)d.used + (need_new - d.free_slots.size()), (size_t)d.cap * 2));
    if (want > d.cap) {
      DevBuf nq, np, nst;
      if (ensure(c, nq, want * slot_bytes) == IBFT_OK && ensure(c, np, want * ibftk::GTAB_ENTRY_DWORDS * 4) == IBFT_OK &&
          ensure(c, nst, want * 4) == IBFT_OK && (d.d_learned.p || ensure(c, d.d_learned, 8) == IBFT_OK)) {
        bool ok = hipMemset(nst.p, 0, want * 4) == hipSuccess;
        if (d.used) {
          ok = ok && hipMemcpy(nq.p, d.d_qtab.p, (size_t)d.used * slot_bytes, hipMemcpyDeviceToDevice) == hipSuccess;
          ok = ok && hipMemcpy(np.p, d.d_pub.p, (size_t)d.used * ibftk::GTAB_ENTRY_DWORDS * 4, hipMemcpyDeviceToDevice) == hipSuccess;
          ok = ok && hipMemcpy(nst.p, d.d_state.p, (size_t)d.used * 4, hipMemcpyDeviceToDevice) == hipSuccess;
        } else {
          ok = ok && hipMemset(d.d_learned.p, 0, 8) == hipSuccess;
        }
        if (ok) {
          release(d.d_qtab);   // (hipFree waits for every kernel of the device that may still read the old buffers;
          release(d.d_pub);    //  launches that would read the pointers take d.mu first, which we hold)
          release(d.d_state);
          d.d_qtab = nq;
          d.d_pub = np;
          d.d_state = nst;
          d.cap = (uint32_t)want;
          d.refs.resize(want, 0);
          d.addr_of.resize(want);
          d.built.resize(want, 0);
          d.generation++;
        } else {
          release(nq);
          release(np);
          release(nst);
        }
      } else {
        release(nq);
        release(np);
        release(nst);
      }
    }
  }
  std::vector<uint32_t> fresh;  // slots handed to new addresses: their state starts at 0
  for (size_t v = 0; v < nv; v++) {
    if (ns[v] != 0xFFFFFFFFu) continue;
    uint32_t sl;
    if (!d.free_slots.empty()) {
      sl = d.free_slots.back();
      d.free_slots.pop_back();
    } else if (d.used < d.cap) {
      sl = d.used++;
    } else {
      continue;  // beyond the budget: this validator stays on the recover path
    }
    d.slot_of[vaddr[v]] = sl;
    d.addr_of[sl] = vaddr[v];
    d.refs[sl] = 1;
    if (d.built[sl]) {
      d.built[sl] = 0;
      d.build_epoch++;
    }
    ns[v] = sl;
    fresh.push_back(sl);
  }
  if (d.cap == 0) {
    c->vslot.clear();
    return IBFT_E_NOMEM;
  }
  // a recycled slot must read "unknown" again before any kernel of this context can claim it
  std::sort(fresh.begin(), fresh.end());
  for (size_t i = 0; i < fresh.size();) {
    size_t j = i;
    while (j + 1 < fresh.size() && fresh[j + 1] == fresh[j] + 1) j++;
    HIPCHK(c, hipMemsetAsync((uint32_t *)d.d_state.p + fresh[i], 0, (j - i + 1) * 4, c->stream));
    i = j + 1;
  }
  int rc = ensure(c, c->d_vslot, nv * 4);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_vslot.p, ns.data(), nv * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->vslot.swap(ns);
  c->cache_on = true;
  c->seen_build_epoch = 0;
  recount_built(c);  // how many of THIS set's validators have a table already (all who stayed, any another context learned)
  return IBFT_OK;
}

static int set_validators_impl(ibft_ctx *c, uint64_t height, const uint8_t *addrs20, const uint64_t *power, uint32_t pw,
                               size_t n) {
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  // Host-side build of the open-addressing table; a repeated address keeps the LAST
  // power (a Go map assignment in a loop), mirroring oracle/ibft_oracle.c.
  uint32_t slots = 64;
  while (slots < 2 * n + 2) slots <<= 1;
  std::vector<uint32_t> tab((size_t)slots * 6, 0);
  std::vector<uint64_t> pwv;  // distinct validators × pw words
  pwv.reserve((n ? n : 1) * pw);
  std::vector<KeyAddr> vaddr;  // distinct validators' addresses, by validator index
  vaddr.reserve(n);
  for (size_t i = 0; i < n; i++) {
    uint32_t a[5];
    memcpy(a, addrs20 + 20 * i, 20);
    uint32_t s = ibftk::addr_hash(a) & (slots - 1);
    for (;;) {
      uint32_t *e = &tab[(size_t)s * 6];
      if (e[5] == 0) {
        memcpy(e, a, 20);
        pwv.insert(pwv.end(), power + i * pw, power + (i + 1) * pw);
        e[5] = (uint32_t)(pwv.size() / pw);
        KeyAddr ka;
        memcpy(ka.b, a, 20);
        vaddr.push_back(ka);  // address of validator index e[5] − 1
        break;
      }
      if (memcmp(e, a, 20) == 0) {
        memcpy(&pwv[(size_t)(e[5] - 1) * pw], power + i * pw, (size_t)pw * 8);
        break;
      }
      s = (s + 1) & (slots - 1);
    }
  }
  const size_t nv = pwv.size() / pw;
  // total voting power and quorum = ⌊2·total/3⌋ + 1 (calculateQuorum, validator_manager.go:130-135) in
  // TALLY_SUM_WORDS × 64 bits, as 32-bit halves: total → ×2 → long division by 3 → +1
  constexpr int H = 2 * ibftk::TALLY_SUM_WORDS;
  uint64_t piece[H] = {0};
  for (size_t v = 0; v < nv; v++)
    for (uint32_t k = 0; k < 2 * pw; k++) piece[k] += (pwv[v * pw + k / 2] >> (32 * (k & 1))) & 0xFFFFFFFFull;
  uint32_t half[H];
  {
    uint64_t carry = 0;
    for (int k = 0; k < H; k++) {
      const uint64_t t = carry + piece[k];
      half[k] = (uint32_t)t;
      carry = t >> 32;
    }
  }
  bool zero = true;
  for (int k = 0; k < H; k++) zero = zero && half[k] == 0;
  if (zero) return IBFT_E_POWER;  // validator_manager.go:68-70
  {
    uint32_t carry = 0;  // × 2
    for (int k = 0; k < H; k++) {
      const uint32_t top = half[k] >> 31;
      half[k] = (half[k] << 1) | carry;
      carry = top;
    }
    uint64_t rem = 0;    // ÷ 3
    for (int k = H - 1; k >= 0; k--) {
      const uint64_t cur = (rem << 32) | half[k];
      half[k] = (uint32_t)(cur / 3);
      rem = cur % 3;
    }
    for (int k = 0; k < H; k++)  // + 1
      if (++half[k] != 0) break;
  }
  uint64_t quorum_w[ibftk::TALLY_SUM_WORDS];
  for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) quorum_w[i] = (uint64_t)half[2 * i] | ((uint64_t)half[2 * i + 1] << 32);
  int rc;
  if ((rc = upload(c, c->d_vtab, tab.data(), tab.size() * 4))) return rc;
  if ((rc = upload(c, c->d_vpower, pwv.data(), pwv.size() * 8))) return rc;
  if ((rc = upload(c, c->d_quorum, quorum_w, sizeof quorum_w))) return rc;
  const size_t seen_bytes = ((nv + 31) / 32) * 4;
  if (seen_bytes > c->d_seen.cap) {  // the tally keeps the bitmap zeroed between launches: zero it when it is (re)allocated
    if ((rc = ensure(c, c->d_seen, seen_bytes))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_seen.p, 0, c->d_seen.cap, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // warm path: the validators of this set get their slots in the device's key cache — an address that had one keeps it
  // (and its table), whatever changed around it
  if (c->flags & IBFT_FLAG_PUBKEY_CACHE) {
    std::lock_guard<std::mutex> dlk(c->dev->mu);
    c->cache_on = key_cache_map(c, vaddr) == IBFT_OK;
  }
  c->valset_addrs.assign(addrs20, addrs20 + n * 20);
  c->h_vtab = std::move(tab);
  c->vslot_mask = slots - 1;
  c->n_validators = (uint32_t)nv;
  c->power_words = pw;
  memcpy(c->quorum_w, quorum_w, sizeof quorum_w);
  c->height = height;
  c->have_valset = true;
  return IBFT_OK;
}

int ibft_set_validators(ibft_ctx *c, uint64_t height, const uint8_t *addrs20, const uint64_t *power, size_t n) {
  if (!c || (n && (!addrs20 || !power)) || n > (1u << 20)) return IBFT_E_INVAL;
  return set_validators_impl(c, height, addrs20, power, 1, n);
}

int ibft_set_validators_u256(ibft_ctx *c, uint64_t height, const uint8_t *addrs20, const uint8_t *power_be32, size_t n) {
  if (!c || (n && (!addrs20 || !power_be32)) || n > (1u << 20)) return IBFT_E_INVAL;
  std::vector<uint64_t> words(n * 4);  // big.Int.FillBytes(32) → four little-endian words
  for (size_t i = 0; i < n; i++)
    for (int w = 0; w < 4; w++) {
      uint64_t v = 0;
      for (int b = 0; b < 8; b++) v = (v << 8) | power_be32[32 * i + 8 * (3 - w) + b];
      words[4 * i + w] = v;
    }
  return set_validators_impl(c, height, addrs20, words.data(), 4, n);
}

int ibft_last_tally_wide(ibft_ctx *c, ibft_tally_wide_t *out) {
  if (!c || !out) return IBFT_E_INVAL;
  ctx_lock lk(c);
  memset(out, 0, sizeof *out);
  for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) {
    out->quorum[i] = c->quorum_w[i];
    out->power[i] = c->last_wide[i];
  }
  bool ge = true;
  for (int i = ibftk::TALLY_SUM_WORDS - 1; i >= 0; i--)
    if (out->power[i] != out->quorum[i]) {
      ge = out->power[i] > out->quorum[i];
      break;
    }
  out->has_quorum = ge ? 1 : 0;
  return IBFT_OK;
}

int ibft_keccak256(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, uint8_t out32[32]) {
  if (!out32 || (na && !a) || (nb && !b)) return IBFT_E_INVAL;
  host_keccak256(a, na, b, nb, out32);
  return IBFT_OK;
}

int ibft_proposal_hash(ibft_ctx *c, const uint8_t *raw, size_t raw_len, uint64_t round, uint8_t out32[32]) {
  if (!c || (raw_len && !raw) || !out32 || raw_len > (1ull << 31)) return IBFT_E_INVAL;
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_proposal_hash(c, raw, raw_len, round))) return rc;
  if (c->have_H && c->have_H_host) {  // hashed on the host: the digest is at hand (d_H receives it on the side stream)
    memcpy(out32, c->H_host, 32);
    return IBFT_OK;
  }
  if ((rc = wait_proposal_hash(c))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_digest, c->d_H.p, 32, hipMemcpyDeviceToHost, c->stream));
  if (hipStreamSynchronize(c->stream) != hipSuccess) {
    c->have_H = false;
    c->last_error = "hipStreamSynchronize failed";
    return IBFT_E_HIP;
  }
  memcpy(out32, c->h_digest, 32);
  return IBFT_OK;
}

static int hash_eq_locked(ibft_ctx *c, const uint8_t *hash32, const uint8_t *hash_len, size_t n, uint64_t *out_mask) {
  int rc;
  c->wire_valid = false;
  if ((rc = wait_proposal_hash(c))) return rc;
  ColumnCopies cc;
  cc.add(c->d_hash.p, hash32, n * 32);
  cc.add(c->d_hash_len.p, hash_len, n);
  if ((rc = cc.flush(c))) return rc;
  if (n) {
    hipLaunchKernelGGL(ibftk::hash_eq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       (const uint8_t *)c->d_hash.p, (const uint8_t *)c->d_hash_len.p,
                       (const uint64_t *)c->d_H.p, (uint32_t)n, (uint64_t *)c->d_mask.p);
    HIPCHK(c, hipGetLastError());
    c->mask_dirty_words = std::max(c->mask_dirty_words, (uint32_t)mask_words(n));  // ballot words, no tally follows
  }
  rc = fetch_results(c, (uint32_t)n, out_mask, nullptr, false);
  if (rc) c->have_H = false;  // a failed stream leaves nothing to trust
  return rc;
}

int ibft_forget_proposal(ibft_ctx *c) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  c->have_H = false;
  return IBFT_OK;
}

int ibft_verify_hashes(ibft_ctx *c, const uint8_t *raw, size_t raw_len, uint64_t round,
                       const uint8_t *hash32, const uint8_t *hash_len, size_t n, uint64_t *out_mask) {
  if (!c || (raw_len && !raw) || (n && (!hash32 || !hash_len || !out_mask)) || raw_len > (1ull << 31))
    return IBFT_E_INVAL;
  ctx_lock lk(c);  // context state is only read inside the critical section
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_proposal_hash(c, raw, raw_len, round))) return rc;
  return hash_eq_locked(c, hash32, hash_len, n, out_mask);
}

int ibft_verify_hashes_digest(ibft_ctx *c, const uint8_t digest32[32], const uint8_t *hash32, const uint8_t *hash_len,
                              size_t n, uint64_t *out_mask) {
  if (!c || !digest32 || (n && (!hash32 || !hash_len || !out_mask))) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  HIPCHK(c, hipSetDevice(c->device));
  {
    int rcw = wait_proposal_hash(c);  // a hash still on its way must not land on top of the caller's digest
    if (rcw) return rcw;
  }
  c->have_H = false;  // d_H now holds the caller's digest, not the hash of the remembered proposal
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is free again
  memcpy(c->h_digest, digest32, 32);
  HIPCHK(c, hipMemcpyAsync(c->d_H.p, c->h_digest, 32, hipMemcpyHostToDevice, c->stream));
  return hash_eq_locked(c, hash32, hash_len, n, out_mask);
}

static int seals_stage_locked(ibft_ctx *c, const uint8_t *hash32, const uint8_t *sig65, const uint8_t *signer20,
                              const uint8_t *pre_flags, size_t n, bool wait) {
  if (n && (!hash32 || !sig65 || !signer20)) return IBFT_E_INVAL;
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  c->wire_valid = false;
  ColumnCopies cc;
  cc.add(c->d_hash.p, hash32, n * 32);
  cc.add(c->d_sig.p, sig65, n * 65);
  cc.add(c->d_signer.p, signer20, n * 20);
  if (pre_flags) cc.add(c->d_pre.p, pre_flags, n);
  if ((rc = cc.flush(c))) return rc;
  if ((rc = apply_seal_digest(c, 0, (uint32_t)n, false))) return rc;
  // ibft_seals_stage promises the caller its buffers back; the one-shot call waits once, at the end
  if (wait) HIPCHK(c, hipStreamSynchronize(c->stream));
  c->staged_n = (uint32_t)n;
  c->staged_pre = pre_flags != nullptr;
  return IBFT_OK;
}

int ibft_seals_stage(ibft_ctx *c, const uint8_t *hash32, const uint8_t *sig65, const uint8_t *signer20,
                     const uint8_t *pre_flags, size_t n) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  return seals_stage_locked(c, hash32, sig65, signer20, pre_flags, n, true);
}

// Double-buffered staging: the copy of batch k+1 overlaps the verdict kernels of batch k.
int ibft_seals_stage_next(ibft_ctx *c, const uint8_t *hash32, const uint8_t *sig65, const uint8_t *signer20,
                          const uint8_t *pre_flags, size_t n) {
  if (!c || (n && (!hash32 || !sig65 || !signer20))) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->cstream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_cols_read, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_cols_read, c->stream));
  }
  int rc;
  const size_t m = std::max<size_t>(c->max_rows, c->row_cap);
  if ((rc = ensure(c, c->d_hash_nx, m * 32))) return rc;
  if ((rc = ensure(c, c->d_sig_nx, m * 65 + 64))) return rc;
  if ((rc = ensure(c, c->d_signer_nx, m * 20))) return rc;
  if ((rc = ensure(c, c->d_pre_nx, m))) return rc;
  // the spare slot was the resident one until the last swap: kernels enqueued before that swap may still read it
  HIPCHK(c, hipStreamWaitEvent(c->cstream, c->ev_cols_read, 0));
  if (n) {
    HIPCHK(c, hipMemcpyAsync(c->d_hash_nx.p, hash32, n * 32, hipMemcpyHostToDevice, c->cstream));
    HIPCHK(c, hipMemcpyAsync(c->d_sig_nx.p, sig65, n * 65, hipMemcpyHostToDevice, c->cstream));
    HIPCHK(c, hipMemcpyAsync(c->d_signer_nx.p, signer20, n * 20, hipMemcpyHostToDevice, c->cstream));
    if (pre_flags) HIPCHK(c, hipMemcpyAsync(c->d_pre_nx.p, pre_flags, n, hipMemcpyHostToDevice, c->cstream));
  }
  HIPCHK(c, hipEventRecord(c->ev_staged, c->cstream));
  c->next_n = (uint32_t)n;
  c->next_pre = pre_flags != nullptr;
  c->next_valid = true;
  return IBFT_OK;
}

int ibft_seals_swap(ibft_ctx *c, int wait_for_copy) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (!c->next_valid) {
    c->last_error = "ibft_seals_swap without a batch staged by ibft_seals_stage_next";
    return IBFT_E_INVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  // everything enqueued so far read the columns that become the spare slot now; what is enqueued from here on reads the
  // other set, once its copy has landed (a device-side wait: the host does not block unless asked to)
  HIPCHK(c, hipEventRecord(c->ev_cols_read, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_staged, 0));
  std::swap(c->d_hash, c->d_hash_nx);
  std::swap(c->d_sig, c->d_sig_nx);
  std::swap(c->d_signer, c->d_signer_nx);
  std::swap(c->d_pre, c->d_pre_nx);
  c->staged_n = c->next_n;
  c->staged_pre = c->next_pre;
  c->next_valid = false;
  c->wire_valid = false;
  int rc;
  if ((rc = apply_seal_digest(c, 0, c->staged_n, false))) return rc;
  if (wait_for_copy) HIPCHK(c, hipEventSynchronize(c->ev_staged));  // the caller's source buffers are free again
  return IBFT_OK;
}

static int seals_launch_locked(ibft_ctx *c, uint32_t repeat) {
  if (!c->have_valset) return IBFT_E_NOVALSET;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->ev_used >= 4096) c->ev_used = 0;  // event pairs accumulate until ibft_last_kernel_ms reads (and resets) them
  if (repeat == 0) repeat = 1;
  c->launched_n = c->staged_n;
  for (uint32_t k = 0; k < repeat; k++) {
    int rc;
    const bool time_it = c->time_every && (c->pass_counter++ % c->time_every) == 0;
    if ((rc = enqueue_recover(c, c->staged_n, c->staged_pre, 0, time_it))) return rc;
    if ((rc = enqueue_tally(c, c->staged_n))) return rc;
  }
  return IBFT_OK;
}

// Pipelined passes over the resident batch: submit enqueues one pass whose results go to one of two host-visible slots,
// collect waits for the OLDEST submitted pass only (the event behind its tally, not the stream).
int ibft_seals_submit(ibft_ctx *c) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (!c->have_valset) return IBFT_E_NOVALSET;
  if (c->pass_issued - c->pass_collected >= 2) {
    c->last_error = "two passes already in flight: call ibft_seals_collect first";
    return IBFT_E_INVAL;
  }
  if (c->learn_rc != IBFT_OK) {  // the table build behind an already delivered pass failed (ibft_seals_collect): say so once
    const int rc = c->learn_rc;
    c->learn_rc = IBFT_OK;
    return rc;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t s = c->pass_issued & 1u;
  // mapped result slots (the tally kernel writes them itself) and the pass events, on first use — piece by piece, so that a
  // failure half way leaves nothing to leak and nothing half set up for the next call
  for (int i = 0; i < 2; i++) {
    if (!c->p_mask[i] &&
        hipHostMalloc((void **)&c->p_mask[i], (size_t)mask_words(std::max<size_t>(c->max_rows, c->row_cap)) * 8 + 64) != hipSuccess)
      return IBFT_E_NOMEM;
    if (!c->p_tally[i] && hipHostMalloc((void **)&c->p_tally[i], 128) != hipSuccess) return IBFT_E_NOMEM;
    if (!c->dp_mask[i] || !c->dp_tally[i]) {
      void *dm = nullptr, *dt = nullptr;
      if (hipHostGetDevicePointer(&dm, c->p_mask[i], 0) != hipSuccess || hipHostGetDevicePointer(&dt, c->p_tally[i], 0) != hipSuccess)
        return IBFT_E_HIP;
      c->dp_mask[i] = (uint64_t *)dm;
      c->dp_tally[i] = (uint64_t *)dt;
    }
    if (!c->ev_pass[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pass[i], hipEventDisableTiming));
  }
  if (c->ev_used >= 4096) c->ev_used = 0;
  const bool time_it = c->time_every && (c->pass_counter++ % c->time_every) == 0;
  int rc;
  // The verdict launch writes the CURRENT pair of work mask / validator indices; its last reader was the tally of pass k − 2
  // (collected, or the two-in-flight check above would have refused) or something the main stream is already behind.
  if ((rc = enqueue_recover(c, c->staged_n, c->staged_pre, 0, time_it))) return rc;
  // Side-stream tally: not for a rank of a sharded batch (its exchange follows the tally on the main stream) and not while
  // keys are still being learned (the tally passes the device's learned-key counter on; with every table built nothing moves it)
  const bool all_warm = c->cache_on && c->my_built >= c->n_validators;
  // AUTO: behind the warm kernels, and behind the cold kernels that leave the tally room (rows / wave / two-wave forms: a quarter
  // of the LDS, a third of the registers) — NOT behind the lane / group kernels whose tables fill the LDS (see side_tally above)
  const bool side = !c->comm && !c->xlocal && (!c->cache_on || all_warm) &&
                    (c->side_tally == 1 || (c->side_tally == 2 && (all_warm || (!c->cache_on && c->last_cold_group >= 16))));
  if (side) {
    if (!c->tstream) HIPCHK(c, hipStreamCreateWithFlags(&c->tstream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++)
      if (!c->ev_rec[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_rec[i], hipEventDisableTiming));
    // (a timed pass already has an event right behind its verdict kernels — the stop event of the pair: one marker less)
    hipEvent_t after = time_it && c->ev_used > 0 ? c->ev[(size_t)(c->ev_used - 1) * 2 + 1] : nullptr;
    if (!after) {
      after = c->ev_rec[s];
      HIPCHK(c, hipEventRecord(after, c->stream));
    }
    HIPCHK(c, hipStreamWaitEvent(c->tstream, after, 0));
    c->tally_slot = (int)s;
    rc = enqueue_tally(c, c->staged_n, nullptr, c->tstream);
    c->tally_slot = -1;
    if (rc) return rc;
    if (hipEventRecord(c->ev_pass[s], c->tstream) != hipSuccess) {
      // the tally is on its way without an event anybody could wait for: drain its stream before reporting, or the next call on
      // the main stream would share its buffers with it
      (void)hipStreamSynchronize(c->tstream);
      c->last_error = "hipEventRecord behind a side-stream tally failed";
      return IBFT_E_HIP;
    }
    // the next verdict launch gets the other pair; this one belongs to the tally just enqueued until the pass is collected
    std::swap(c->d_mask, c->d_mask_b);
    std::swap(c->d_vidx, c->d_vidx_b);
    std::swap(c->mask_dirty_words, c->mask_dirty_words_b);
    c->side_pending = true;
    c->side_tallies++;
  } else {
    if ((rc = join_side(c))) return rc;   // a main-stream tally shares the sums and verdict words with the side stream's
    c->tally_slot = (int)s;
    rc = enqueue_tally(c, c->staged_n);
    c->tally_slot = -1;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_pass[s], c->stream));
  }
  c->pass_n[s] = c->staged_n;
  c->launched_n = c->staged_n;  // (an ibft_seals_fetch behind a submit delivers this pass through the copy route)
  c->pass_issued++;
  return IBFT_OK;
}

int ibft_seals_collect(ibft_ctx *c, uint64_t *out_mask, ibft_tally_t *tally) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (c->pass_collected == c->pass_issued) {
    c->last_error = "ibft_seals_collect without a submitted pass";
    return IBFT_E_INVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t s = c->pass_collected & 1u;
  HIPCHK(c, hipEventSynchronize(c->ev_pass[s]));
  const uint64_t *hm = c->p_mask[s], *ht = c->p_tally[s];
  const uint32_t n = c->pass_n[s];
  const size_t mw = (size_t)mask_words(n);
  if (out_mask && mw) {
    memcpy(out_mask, hm, mw * 8);
    if (n & 63) out_mask[mw - 1] &= (~0ull) >> (64 - (n & 63));
  }
  for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) c->last_wide[i] = ht[ibftk::TALLY_OUT_WIDE + i];
  if (tally) {
    memset(tally, 0, sizeof *tally);
    tally->quorum_lo = c->quorum_w[0];
    tally->quorum_hi = c->quorum_w[1];
    tally->power_lo = ht[0];
    tally->power_hi = ht[1];
    tally->valid_rows = (uint32_t)(ht[2] & 0xFFFFFFFFull);
    tally->distinct_senders = (uint32_t)(ht[2] >> 32);
    tally->has_quorum = (uint32_t)ht[3];
    tally->proposer_rows = (uint32_t)ht[ibftk::TALLY_OUT_PROPOSER_ROWS];
  }
  // The pass is DELIVERED (its verdict words and tally are in the caller's buffers) before anything else can fail
  // (the count used to move first and a failing table build dropped a finished pass for good).
  c->pass_collected++;
  if (c->cache_on) {  // keys this pass taught the device (passed on by its tally kernel) → tables
    // (build_new_tables waits for the context's whole stream when keys were learned — also for a newer pass in flight:
    // the cold-to-warm transition costs one full drain, include/ibftgpu.h.)  A failure here takes nothing from the verdicts
    // just delivered: the next ibft_seals_submit reports it.
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(ht + 4);
    const int rcb = build_new_tables(c, lw[0], lw[1]);
    if (rcb) c->learn_rc = rcb;
  }
  return IBFT_OK;
}

// Rows of the resident batch and of the oldest submitted pass (0 when none is in flight): a binding sizes its verdict
// buffers from THESE, never from a count its caller passes along (shim/go/ibftgpu).
int ibft_seals_rows(ibft_ctx *c, uint32_t *resident_rows, uint32_t *oldest_pass_rows) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (resident_rows) *resident_rows = c->staged_n;
  if (oldest_pass_rows) *oldest_pass_rows = c->pass_collected == c->pass_issued ? 0u : c->pass_n[c->pass_collected & 1u];
  return IBFT_OK;
}

int ibft_seals_launch(ibft_ctx *c, uint32_t repeat) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  return seals_launch_locked(c, repeat);
}

int ibft_seals_fetch(ibft_ctx *c, uint64_t *out_mask, ibft_tally_t *tally) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  return fetch_results(c, c->launched_n, out_mask, tally, true);
}

int ibft_seals_run(ibft_ctx *c, uint64_t *out_mask, ibft_tally_t *tally) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  int rc = seals_launch_locked(c, 1);
  if (rc) return rc;
  return fetch_results(c, c->staged_n, out_mask, tally, true);
}

int ibft_seals_device_ptrs(ibft_ctx *c, void **d_mask, size_t *mask_words_out, void **d_tally) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (d_mask) *d_mask = c->d_mask_out.p;
  if (mask_words_out) *mask_words_out = (size_t)mask_words(c->staged_n);
  if (d_tally) *d_tally = c->d_tally.p;
  return IBFT_OK;
}

int ibft_seals_export(ibft_ctx *c, void *d_mask_dst, void *d_tally_dst) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  size_t mw = (size_t)mask_words(c->staged_n);
  if (d_mask_dst && mw)
    HIPCHK(c, hipMemcpyAsync(d_mask_dst, c->d_mask_out.p, mw * 8, hipMemcpyDeviceToDevice, c->stream));
  if (d_tally_dst)
    HIPCHK(c, hipMemcpyAsync(d_tally_dst, c->d_tally.p, 4 * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return IBFT_OK;
}

int ibft_seals_export_on(ibft_ctx *c, void *d_mask_dst, void *d_tally_dst, void *consumer_stream) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t cs = (hipStream_t)consumer_stream;
  {
    int rce = ensure_handoff_events(c);
    if (rce) return rce;
  }
  HIPCHK(c, hipEventRecord(c->ev_ready, c->stream));  // everything launched so far, the tally included
  HIPCHK(c, hipStreamWaitEvent(cs, c->ev_ready, 0));
  size_t mw = (size_t)mask_words(c->staged_n);
  if (d_mask_dst && mw) HIPCHK(c, hipMemcpyAsync(d_mask_dst, c->d_mask_out.p, mw * 8, hipMemcpyDeviceToDevice, cs));
  if (d_tally_dst) HIPCHK(c, hipMemcpyAsync(d_tally_dst, c->d_tally.p, 4 * 8, hipMemcpyDeviceToDevice, cs));
  HIPCHK(c, hipEventRecord(c->ev_read, cs));
  c->read_pending = true;  // enqueue_tally: the next tally overwrites d_mask_out / d_tally only after ev_read
  return IBFT_OK;
}

int ibft_last_kernel_ms(ibft_ctx *c, float *ms, uint32_t *launches) {
  if (!c || !ms) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float total = 0.f;
  for (uint32_t i = 0; i < c->ev_used; i++) {
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev[(size_t)i * 2], c->ev[(size_t)i * 2 + 1]));
    total += t;
  }
  *ms = total;
  if (launches) *launches = c->ev_used;
  c->ev_used = 0;
  return IBFT_OK;
}

int ibft_set_kernel_timing(ibft_ctx *c, uint32_t every_n) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  c->time_every = every_n;
  c->pass_counter = 0;
  return IBFT_OK;
}

int ibft_cache_stats(ibft_ctx *c, uint32_t *tables, uint32_t *warm_passes, uint32_t *cold_passes,
                     uint32_t *lanes_per_signature) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (tables) *tables = c->cache_on ? c->my_built : 0;
  if (warm_passes) *warm_passes = c->warm_passes;
  if (cold_passes) *cold_passes = c->cold_passes;
  if (lanes_per_signature) *lanes_per_signature = c->last_group;
  return IBFT_OK;
}

int ibft_cache_memory(ibft_ctx *c, uint64_t *device_bytes, uint32_t *slots_in_use, uint32_t *slots_allocated,
                      uint32_t *contexts_sharing) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  std::lock_guard<std::mutex> dlk(c->dev->mu);
  const DeviceShared &d = *c->dev;
  if (device_bytes) *device_bytes = (uint64_t)d.d_gtab.cap + d.d_qtab.cap + d.d_pub.cap + d.d_state.cap + d.d_learned.cap;
  if (slots_in_use) *slots_in_use = d.used - (uint32_t)d.free_slots.size();
  if (slots_allocated) *slots_allocated = d.cap;
  if (contexts_sharing) *contexts_sharing = (uint32_t)c->dev.use_count();
  return IBFT_OK;
}

int ibft_last_dispatch(ibft_ctx *c, uint32_t *cold_lanes, uint32_t *warm_lanes) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (cold_lanes) *cold_lanes = c->last_cold_group;
  if (warm_lanes) *warm_lanes = c->last_group;
  return IBFT_OK;
}

int ibft_last_cold_table(ibft_ctx *c, uint32_t *table) {
  if (!c || !table) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  *table = (c->last_cold_group == 1 || c->last_cold_group == 2 || c->last_cold_group == 4) ? c->last_cold_table : 0u;
  return IBFT_OK;
}

int ibft_pipeline_stats(ibft_ctx *c, uint32_t *side_tallies, uint32_t *split_batches) {
  if (!c) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(c->mu);  // (a call of the pipeline itself: no join_side)
  if (side_tallies) *side_tallies = c->side_tallies;
  if (split_batches) *split_batches = c->split_launches;
  return IBFT_OK;
}

int ibft_column_stats(ibft_ctx *c, uint32_t *gather_batches) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (gather_batches) *gather_batches = c->gathers;
  return IBFT_OK;
}

void *ibft_pinned_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return nullptr;
  void *dp = nullptr;
  // the gather launch dereferences the host address itself: only register blocks the device sees at that address
  if (hipHostGetDevicePointer(&dp, p, 0) == hipSuccess && dp == p) {
    std::lock_guard<std::mutex> lk(pinned_registry().mu);
    pinned_registry().blocks[(uintptr_t)p] = bytes ? bytes : 1;
  }
  return p;
}
void ibft_pinned_free(void *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(pinned_registry().mu);
    pinned_registry().blocks.erase((uintptr_t)p);
  }
  (void)hipHostFree(p);
}

// A whole PREPARE or COMMIT set in one call (kernels.hip.h: "a message set in one pass").
// a message set up to and including its tally, everything asynchronous past the uploads (c->mu held)
static int messages_launch_locked(ibft_ctx *c, const uint8_t *payload, const uint32_t *off, const uint8_t *msg_sig65,
                                  const uint8_t *from20, const uint8_t *hash32, const uint8_t *hash_len, const uint8_t *seal65,
                                  const uint8_t *sender_pre, const uint8_t *valid_pre, size_t n, const uint8_t *raw, size_t raw_len,
                                  uint64_t round, const uint8_t *digest32) {
  if (n && (!off || !msg_sig65 || !from20 || !hash32 || !hash_len)) return IBFT_E_INVAL;
  if ((raw_len && !raw) || raw_len > (1ull << 31)) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  if (n && off[n] && !payload) return IBFT_E_INVAL;
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  if (!c->have_valset) return IBFT_E_NOVALSET;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  c->wire_valid = false;
  c->staged_n = 0;
  c->set_n = (uint32_t)n;
  const uint32_t half = ((uint32_t)n + 63u) & ~63u;
  const uint32_t rows = seal65 ? half + (uint32_t)n : (uint32_t)n;  // verdict rows of the one launch
  if ((rc = alloc_rows(c, 2 * (((uint32_t)c->max_rows + 63u) & ~63u)))) return rc;
  // the proposal the set is checked against: its digest, or raw ‖ BE64(round) hashed once and remembered
  bool hash_needed = false;
  if (digest32) {
    if ((rc = wait_proposal_hash(c))) return rc;  // a hash still on its way must not land on top of the caller's digest
    c->have_H = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is free again
    memcpy(c->h_digest, digest32, 32);
    HIPCHK(c, hipMemcpyAsync(c->d_H.p, c->h_digest, 32, hipMemcpyHostToDevice, c->stream));
  } else {
    hash_needed = note_proposal(c, raw, raw_len, round);
  }
  if (n == 0) {
    if (hash_needed && (rc = launch_proposal_hash(c))) return rc;
    // an empty shard of a sharded set still contributes (zero) words, an empty bitmap and a zero count to the exchange;
    // HasPrepareQuorum of no messages is still a question (the proposer alone may be a quorum: validator_manager.go:109-126)
    if (c->comm || c->xlocal || c->next_prop_on) return enqueue_tally(c, 0);
    return IBFT_OK;
  }
  uint8_t *d_hash = (uint8_t *)c->d_hash.p, *d_sig = (uint8_t *)c->d_sig.p, *d_signer = (uint8_t *)c->d_signer.p,
          *d_pre = (uint8_t *)c->d_pre.p;
  const size_t pbytes = off[n];
  if ((rc = ensure(c, c->d_payload, pbytes + 256))) return rc;
  ColumnCopies cc;
  cc.payload(c->d_payload.p, payload, pbytes, c->d_off.p, off, n, d_hash);
  cc.add(d_sig, msg_sig65, n * 65);
  cc.add(d_signer, from20, n * 20);
  cc.add(d_hash + 32ull * half, hash32, n * 32);  // the hashes the messages carry
  cc.add(c->d_hash_len.p, hash_len, n);
  // rows the host rejected before any crypto: applied to the verdict words afterwards (the verdict launch runs them
  // like any other row — skipping work inside a lock-step wavefront saves nothing)
  if (sender_pre) cc.add(d_pre, sender_pre, n);
  if (valid_pre) cc.add(d_pre + half, valid_pre, n);
  if (seal65) {
    cc.add(d_sig + 65ull * half, seal65, n * 65);
    cc.add(d_signer + 20ull * half, from20, n * 20);  // the seal's signer is the message's From
    if (half != n)  // rows n..half−1 sit between the two groups: a zero signature is rejected by every kernel
      HIPCHK(c, hipMemsetAsync(d_sig + 65ull * n, 0, 65ull * (half - n), c->stream));
  }
  if ((rc = cc.flush(c))) return rc;
  if (!cc.digest_done) {  // pageable columns: the payload was copied, hash it from HBM
    hipLaunchKernelGGL(ibftk::payload_digest_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream,
                       (const uint8_t *)c->d_payload.p, (const uint32_t *)c->d_off.p, (uint32_t)n, d_hash);
    HIPCHK(c, hipGetLastError());
  }
  // the seals sign digest(carried hash) under a non-identity convention; a1 still compares the carried hashes (the copy)
  const bool converted = seal65 && c->seal_digest_mode != 0;
  if (converted && (rc = apply_seal_digest(c, half, (uint32_t)n, true))) return rc;
  c->ev_used = 0;
  const bool time_it = c->time_every && (c->pass_counter++ % c->time_every) == 0;
  if ((rc = enqueue_recover(c, rows, false, 0, time_it))) return rc;
  ibftk::set_args sa{};
  sa.sender_pre = sender_pre ? d_pre : nullptr;
  sa.valid_pre = valid_pre ? d_pre + half : nullptr;
  sa.hash32 = converted ? (const uint8_t *)c->d_hash_copy.p : d_hash + 32ull * half;
  sa.hash_len = (const uint8_t *)c->d_hash_len.p;
  sa.H4 = (const uint64_t *)c->d_H.p;
  sa.n = (uint32_t)n;
  sa.half_words = seal65 ? half / 64 : 0;
  sa.sender_out = (uint64_t *)c->d_set.p;
  sa.valid_out = (uint64_t *)c->d_set.p + mask_words(c->max_rows);
  sa.host_sender = c->dh_set;
  sa.host_valid = c->dh_set ? c->dh_set + mask_words(c->max_rows) : nullptr;
  // the verdict launch above did not need the proposal's hash, the combine step does: hashed on the side stream meanwhile
  if (hash_needed && (rc = launch_proposal_hash(c))) return rc;
  if ((rc = wait_proposal_hash(c))) return rc;
  if ((rc = enqueue_tally(c, (uint32_t)n, &sa))) return rc;
  c->mask_dirty_words = 0;  // the combine kernel zeroed the seal words, the tally the sender words
  if (!c->dh_set)
    HIPCHK(c, hipMemcpyAsync(c->h_set, c->d_set.p, (size_t)mask_words(c->max_rows) * 16, hipMemcpyDeviceToHost, c->stream));
  return IBFT_OK;
}

int ibft_verify_messages(ibft_ctx *c, const uint8_t *payload, const uint32_t *off, const uint8_t *msg_sig65,
                         const uint8_t *from20, const uint8_t *hash32, const uint8_t *hash_len, const uint8_t *seal65,
                         const uint8_t *sender_pre, const uint8_t *valid_pre, size_t n, const uint8_t *raw, size_t raw_len,
                         uint64_t round,
                         const uint8_t *digest32, const uint8_t *proposer20, uint64_t *out_sender_mask,
                         uint64_t *out_valid_mask, ibft_tally_t *tally) {
  if (!c || (n && (!out_sender_mask || !out_valid_mask))) return IBFT_E_INVAL;
  ctx_lock lk(c);
  note_proposer(c, proposer20);
  int rc = messages_launch_locked(c, payload, off, msg_sig65, from20, hash32, hash_len, seal65, sender_pre, valid_pre, n, raw,
                                  raw_len, round, digest32);
  c->next_prop_on = c->next_shard = false;  // (an error in front of the tally must not leave the proposer to a later call)
  if (rc) return rc;
  if (n == 0) {
    if (proposer20) return fetch_results(c, 0, nullptr, tally, true);
    if (tally) {
      memset(tally, 0, sizeof *tally);
      tally->quorum_lo = c->quorum_w[0];
      tally->quorum_hi = c->quorum_w[1];
    }
    for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) c->last_wide[i] = 0;
    return IBFT_OK;
  }
  const size_t mw = (size_t)mask_words(n);
  if ((rc = fetch_results(c, (uint32_t)n, nullptr, tally, true))) {
    c->have_H = false;
    return rc;
  }
  memcpy(out_sender_mask, c->h_set, mw * 8);
  memcpy(out_valid_mask, c->h_set + mask_words(c->max_rows), mw * 8);
  return IBFT_OK;
}

// f4: the committed seals of a simulated validator set, one per row.
int ibft_sign_seals(ibft_ctx *c, const uint8_t *sk32, const uint8_t *hash32, size_t n, uint8_t *out_sig65,
                    uint8_t *out_signer20, uint8_t *out_ok) {
  if (!c || (n && (!sk32 || !hash32 || !out_sig65))) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  HIPCHK(c, hipSetDevice(c->device));
  c->wire_valid = false;
  c->staged_n = 0;
  if (n == 0) return IBFT_OK;
  int rc;
  if ((rc = upload(c, c->d_payload, sk32, n * 32))) return rc;  // the sender-payload column is free during a seal batch
  if ((rc = upload(c, c->d_hash, hash32, n * 32))) return rc;
  if ((rc = apply_seal_digest(c, 0, (uint32_t)n, false))) return rc;  // a seal signs what the convention says it signs
  ibftk::sign_args a;
  a.gtab = (const uint32_t *)c->dev->d_gtab.p;
  a.sk32 = (const uint8_t *)c->d_payload.p;
  a.hash32 = (const uint8_t *)c->d_hash.p;
  a.sig65 = (uint8_t *)c->d_sig.p;
  a.signer20 = (uint8_t *)c->d_signer.p;
  a.ok = (uint8_t *)c->d_pre.p;
  a.n = (uint32_t)n;
  const uint32_t blocks = (uint32_t)((n + ibftk::ROWS_PER_BLOCK - 1) / ibftk::ROWS_PER_BLOCK);
  hipLaunchKernelGGL(ibftk::sign_lane_kernel, dim3(blocks), dim3(ibftk::ROWS_PER_BLOCK), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemsetAsync(c->d_payload.p, 0, n * 32, c->stream));  // the keys do not outlive the call in HBM
  HIPCHK(c, hipMemcpyAsync(out_sig65, c->d_sig.p, n * 65, hipMemcpyDeviceToHost, c->stream));
  if (out_signer20) HIPCHK(c, hipMemcpyAsync(out_signer20, c->d_signer.p, n * 20, hipMemcpyDeviceToHost, c->stream));
  if (out_ok) HIPCHK(c, hipMemcpyAsync(out_ok, c->d_pre.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // hash32 / sig65 / signer20 now hold exactly what ibft_seals_stage would have uploaded for these seals:
  // the batch is staged (rows whose key was refused carry a zero signature, which every verifier rejects)
  c->staged_n = (uint32_t)n;
  c->staged_pre = false;
  return IBFT_OK;
}

// Device canary: see issue_probe_kernel (kernels.hip.h).  Three untimed launches, then the median of five timed ones.
int ibft_issue_probe(ibft_ctx *c, float *ns_per_inst, float *kernel_ms) {
  if (!c || !ns_per_inst) return IBFT_E_INVAL;
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t *d_out = nullptr;
  HIPCHK(c, hipMalloc(&d_out, 64));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = IBFT_OK;
  float t[5] = {0, 0, 0, 0, 0};
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = IBFT_E_HIP;
  for (int k = 0; k < 8 && rc == IBFT_OK; k++) {
    if (k >= 3 && hipEventRecord(e0, c->stream) != hipSuccess) rc = IBFT_E_HIP;
    hipLaunchKernelGGL(ibftk::issue_probe_kernel, dim3(256), dim3(256), 0, c->stream, d_out, (uint32_t)(k + 1));
    if (k >= 3) {
      if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
          hipEventElapsedTime(&t[k - 3], e0, e1) != hipSuccess)
        rc = IBFT_E_HIP;
    }
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) rc = IBFT_E_HIP;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_out);
  if (rc != IBFT_OK) {
    c->last_error = "ibft_issue_probe: a HIP call failed";
    return rc;
  }
  std::sort(t, t + 5);
  if (kernel_ms) *kernel_ms = t[2];
  *ns_per_inst = t[2] * 1e6f / (float)(ibftk::ISSUE_PROBE_ITERS * 64);
  return IBFT_OK;
}

int ibft_sync(ibft_ctx *c) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->xstream) HIPCHK(c, hipStreamSynchronize(c->xstream));  // exchanges in flight (ibft_seals_exchange)
  if (c->cstream) HIPCHK(c, hipStreamSynchronize(c->cstream));  // a batch on its way into the spare slot (ibft_seals_stage_next)
  return IBFT_OK;
}

int ibft_verify_seals(ibft_ctx *c, const uint8_t *hash32, const uint8_t *sig65, const uint8_t *signer20,
                      const uint8_t *pre_flags, size_t n, uint64_t *out_mask, ibft_tally_t *tally) {
  if (!c || (n && !out_mask)) return IBFT_E_INVAL;
  ctx_lock lk(c);  // one critical section: stage + launch + fetch
  if (!c->have_valset) return IBFT_E_NOVALSET;
  int rc;
  if ((rc = seals_stage_locked(c, hash32, sig65, signer20, pre_flags, n, false))) return rc;
  if ((rc = seals_launch_locked(c, 1))) return rc;
  return fetch_results(c, c->staged_n, out_mask, tally, true);
}

// a3 up to and including the tally (c->mu held)
static int senders_launch_locked(ibft_ctx *c, const uint8_t *payload, const uint32_t *off, const uint8_t *sig65,
                                 const uint8_t *from20, const uint8_t *pre_flags, size_t n) {
  if (n && (!off || !sig65 || !from20)) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  if (n && off[n] && !payload) return IBFT_E_INVAL;
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  if (!c->have_valset) return IBFT_E_NOVALSET;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  c->wire_valid = false;
  size_t pbytes = n ? off[n] : 0;
  if ((rc = ensure(c, c->d_payload, pbytes + 256))) return rc;
  ColumnCopies cc;
  cc.add(c->d_payload.p, payload, pbytes);
  if (n) cc.add(c->d_off.p, off, (n + 1) * 4);
  cc.add(c->d_sig.p, sig65, n * 65);
  cc.add(c->d_signer.p, from20, n * 20);
  if (pre_flags) cc.add(c->d_pre.p, pre_flags, n);
  if ((rc = cc.flush(c))) return rc;
  c->staged_n = (uint32_t)n;
  c->staged_pre = pre_flags != nullptr;
  c->ev_used = 0;
  if ((rc = enqueue_recover(c, (uint32_t)n, pre_flags != nullptr, 1, true))) return rc;
  return enqueue_tally(c, (uint32_t)n);
}

int ibft_verify_senders(ibft_ctx *c, const uint8_t *payload, const uint32_t *off, const uint8_t *sig65,
                        const uint8_t *from20, const uint8_t *pre_flags, size_t n, uint64_t *out_mask,
                        ibft_tally_t *tally) {
  if (!c || (n && (!off || !out_mask))) return IBFT_E_INVAL;
  ctx_lock lk(c);
  int rc = senders_launch_locked(c, payload, off, sig65, from20, pre_flags, n);
  if (rc) return rc;
  return fetch_results(c, (uint32_t)n, out_mask, tally, true);
}

int ibft_verify_senders_wire(ibft_ctx *c, const uint8_t *wire_bytes, const uint32_t *off, size_t n, uint64_t *out_mask,
                             ibft_wire_row_t *out_rows, ibft_tally_t *tally) {
  static_assert(sizeof(ibft_wire_row_t) == sizeof(wire::row_info), "ABI");
  if (!c || (n && (!off || !out_mask))) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  if (n && off[n] && !wire_bytes) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  if (!c->have_valset) return IBFT_E_NOVALSET;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const size_t wbytes = n ? off[n] : 0;
  if ((rc = ensure(c, c->d_payload, wbytes + 256))) return rc;
  if ((rc = ensure(c, c->d_wire_rows, (size_t)c->max_rows * sizeof(wire::row_info)))) return rc;
  if ((rc = ensure(c, c->d_seal, (size_t)c->max_rows * 65 + 64))) return rc;
  if (wbytes) HIPCHK(c, hipMemcpyAsync(c->d_payload.p, wire_bytes, wbytes, hipMemcpyHostToDevice, c->stream));
  if ((rc = upload(c, c->d_off, off, (n + 1) * 4))) return rc;
  c->staged_n = (uint32_t)n;
  c->staged_pre = true;
  c->wire_n = (uint32_t)n;
  c->wire_valid = true;
  c->ev_used = 0;
  if (n) {
    hipLaunchKernelGGL(ibftk::wire_parse_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream,
                       (const uint8_t *)c->d_payload.p, (const uint32_t *)c->d_off.p, (uint32_t)n,
                       (wire::row_info *)c->d_wire_rows.p, (uint8_t *)c->d_hash.p, (uint8_t *)c->d_sig.p,
                       (uint8_t *)c->d_signer.p, (uint8_t *)c->d_seal.p, (uint8_t *)c->d_pre.p);
    HIPCHK(c, hipGetLastError());
  }
  // the digest column now holds keccak256(PayloadNoSig): the sender check is the seal-style pass (mode 0)
  if ((rc = enqueue_recover(c, (uint32_t)n, true, 0, true))) return rc;
  if ((rc = enqueue_tally(c, (uint32_t)n))) return rc;
  if (out_rows && n)
    HIPCHK(c, hipMemcpyAsync(out_rows, c->d_wire_rows.p, n * sizeof(ibft_wire_row_t), hipMemcpyDeviceToHost, c->stream));
  return fetch_results(c, (uint32_t)n, out_mask, tally, true);
}

// A batch of raw IbftMessages judged completely: the wire walk, then BOTH signatures of every message in one verdict
// launch (sender rows [0, n), seal rows [half, half + n)), the closure's hash compare folded into the combine step.
int ibft_verify_messages_wire(ibft_ctx *c, const uint8_t *wire_bytes, const uint32_t *off, size_t n, uint64_t height,
                              uint64_t round, const uint8_t *raw, size_t raw_len, uint64_t proposal_round,
                              const uint8_t *digest32, uint64_t *out_sender_mask, uint64_t *out_valid_mask,
                              uint8_t *out_class, ibft_wire_row_t *out_rows, const uint8_t *proposer20, ibft_tally_t *tally) {
  if (!c || (n && (!off || !out_sender_mask || !out_valid_mask))) return IBFT_E_INVAL;
  if ((raw_len && !raw) || raw_len > (1ull << 31)) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  if (n && off[n] && !wire_bytes) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  if (!c->have_valset) return IBFT_E_NOVALSET;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  c->wire_valid = false;
  c->staged_n = 0;
  const uint32_t half = ((uint32_t)n + 63u) & ~63u;
  if ((rc = alloc_rows(c, 2 * (((uint32_t)c->max_rows + 63u) & ~63u)))) return rc;
  bool hash_needed = false;
  if (digest32) {
    if ((rc = wait_proposal_hash(c))) return rc;  // a hash still on its way must not land on top of the caller's digest
    c->have_H = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is free again
    memcpy(c->h_digest, digest32, 32);
    HIPCHK(c, hipMemcpyAsync(c->d_H.p, c->h_digest, 32, hipMemcpyHostToDevice, c->stream));
  } else {
    hash_needed = note_proposal(c, raw, raw_len, proposal_round);
  }
  if (n == 0) {
    if (hash_needed && (rc = launch_proposal_hash(c))) return rc;
    if (proposer20) {  // HasPrepareQuorum of no messages: the proposer alone
      note_proposer(c, proposer20);
      if ((rc = enqueue_tally(c, 0))) return rc;
      return fetch_results(c, 0, nullptr, tally, true);
    }
    if (tally) {
      memset(tally, 0, sizeof *tally);
      tally->quorum_lo = c->quorum_w[0];
      tally->quorum_hi = c->quorum_w[1];
    }
    for (int i = 0; i < ibftk::TALLY_SUM_WORDS; i++) c->last_wide[i] = 0;
    return IBFT_OK;
  }
  const size_t wbytes = off[n];
  if ((rc = ensure(c, c->d_payload, wbytes + 256))) return rc;
  if ((rc = ensure(c, c->d_wire_rows, (size_t)c->max_rows * sizeof(wire::row_info)))) return rc;
  if ((rc = ensure(c, c->d_seal, (size_t)c->max_rows * 65 + 64))) return rc;
  if ((rc = ensure(c, c->d_noseal, c->max_rows))) return rc;
  ColumnCopies cc;
  cc.add(c->d_payload.p, wire_bytes, wbytes);
  cc.add(c->d_off.p, off, (n + 1) * 4);
  if ((rc = cc.flush(c))) return rc;
  uint8_t *d_hash = (uint8_t *)c->d_hash.p, *d_sig = (uint8_t *)c->d_sig.p, *d_signer = (uint8_t *)c->d_signer.p,
          *d_pre = (uint8_t *)c->d_pre.p;
  hipLaunchKernelGGL(ibftk::wire_parse_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream,
                     (const uint8_t *)c->d_payload.p, (const uint32_t *)c->d_off.p, (uint32_t)n,
                     (wire::row_info *)c->d_wire_rows.p, d_hash, d_sig, d_signer, (uint8_t *)c->d_seal.p, d_pre);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(ibftk::wire_set_stage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     (const wire::row_info *)c->d_wire_rows.p, (const uint8_t *)c->d_seal.p, (uint32_t)n, half, height, round,
                     d_hash, (uint8_t *)c->d_hash_len.p, d_sig, d_signer, d_pre, (uint8_t *)c->d_noseal.p,
                     (uint8_t *)c->d_class.p, c->dh_class);
  HIPCHK(c, hipGetLastError());
  if (half != n) {
    HIPCHK(c, hipMemsetAsync(d_sig + 65ull * n, 0, 65ull * (half - n), c->stream));
    HIPCHK(c, hipMemsetAsync(d_pre + n, 1, half - n, c->stream));
  }
  const bool converted = c->seal_digest_mode != 0;
  if (converted && (rc = apply_seal_digest(c, half, (uint32_t)n, true))) return rc;
  c->ev_used = 0;
  const bool time_it = c->time_every && (c->pass_counter++ % c->time_every) == 0;
  // with the pre column: wavefronts whose rows are all dead (the seal rows of PREPAREs, other views, odd encodings) exit at once
  if ((rc = enqueue_recover(c, half + (uint32_t)n, true, 0, time_it))) return rc;
  ibftk::set_args sa{};
  sa.hash32 = converted ? (const uint8_t *)c->d_hash_copy.p : d_hash + 32ull * half;
  sa.hash_len = (const uint8_t *)c->d_hash_len.p;
  sa.H4 = (const uint64_t *)c->d_H.p;
  sa.sender_pre = d_pre;  // not canonical here, or From / Signature of a length no signature check can pass
  sa.valid_pre = d_pre + half;
  sa.no_seal = (const uint8_t *)c->d_noseal.p;
  sa.n = (uint32_t)n;
  sa.half_words = half / 64;
  const size_t mw = (size_t)mask_words(n);
  sa.sender_out = (uint64_t *)c->d_set.p;
  sa.valid_out = (uint64_t *)c->d_set.p + mask_words(c->max_rows);
  sa.host_sender = c->dh_set;
  sa.host_valid = c->dh_set ? c->dh_set + mask_words(c->max_rows) : nullptr;
  // the verdict launch above did not need the proposal's hash, the combine step does: hashed on the side stream meanwhile
  if (hash_needed && (rc = launch_proposal_hash(c))) return rc;
  if ((rc = wait_proposal_hash(c))) return rc;
  note_proposer(c, proposer20);
  if ((rc = enqueue_tally(c, (uint32_t)n, &sa))) return rc;
  c->mask_dirty_words = 0;
  if (!c->dh_set)
    HIPCHK(c, hipMemcpyAsync(c->h_set, c->d_set.p, (size_t)mask_words(c->max_rows) * 16, hipMemcpyDeviceToHost, c->stream));
  if (out_class && !c->dh_class)
    HIPCHK(c, hipMemcpyAsync(c->h_class, c->d_class.p, n, hipMemcpyDeviceToHost, c->stream));
  if (out_rows)  // 80 B per row through a copy command: ask for it only when the fields are needed (out_class routes)
    HIPCHK(c, hipMemcpyAsync(out_rows, c->d_wire_rows.p, n * sizeof(ibft_wire_row_t), hipMemcpyDeviceToHost, c->stream));
  if ((rc = fetch_results(c, (uint32_t)n, nullptr, tally, true))) {
    c->have_H = false;
    return rc;
  }
  memcpy(out_sender_mask, c->h_set, mw * 8);
  memcpy(out_valid_mask, c->h_set + mask_words(c->max_rows), mw * 8);
  if (out_class) memcpy(out_class, c->h_class, n);
  return IBFT_OK;
}

// §8f rank 2 from the transport's bytes: the whole certificate tree of a batch of PREPREPARE / ROUND_CHANGE messages —
// every nested message a row — expanded level by level on the device (kernels.hip.h: cert_*_kernel), then ONE verdict
// launch over all rows.  The host takes part once per level: it reads the next level's row count to size the launches.
int ibft_verify_certificates_wire(ibft_ctx *c, const uint8_t *wire_bytes, const uint32_t *off, size_t n, size_t rows_cap,
                                  size_t *out_n_rows, ibft_cert_node_t *out_nodes, ibft_wire_row_t *out_rows,
                                  uint8_t *out_class, uint64_t *out_sender_mask, uint64_t *out_hash_mask,
                                  uint64_t *out_self_mask) {
  static_assert(sizeof(ibft_cert_node_t) == sizeof(wire::node_info), "ABI");
  static_assert(IBFT_CERT_DIGEST_MAX_BYTES == wire::TREE_DIGEST_MAX_BYTES, "ABI");
  if (!c || !out_n_rows || (n && (!off || !out_sender_mask))) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  if (n && off[n] && !wire_bytes) return IBFT_E_INVAL;
  *out_n_rows = 0;
  ctx_lock lk(c);
  const size_t cap = std::min<size_t>(rows_cap, c->max_rows);
  if (n > cap) return IBFT_E_TOOBIG;
  if (!c->have_valset) return IBFT_E_NOVALSET;
  if (n == 0) return IBFT_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  c->wire_valid = false;
  c->staged_n = 0;
  const size_t wbytes = off[n], m = c->max_rows;
  if ((rc = ensure(c, c->d_payload, wbytes + 256))) return rc;
  if ((rc = ensure(c, c->d_wire_rows, m * sizeof(wire::row_info)))) return rc;
  if ((rc = ensure(c, c->d_cert_nodes, m * sizeof(wire::node_info)))) return rc;
  if ((rc = ensure(c, c->d_cert_span, m * 8))) return rc;
  if ((rc = ensure(c, c->d_cert_count, m * 4))) return rc;
  if ((rc = ensure(c, c->d_cert_slot, m * 4))) return rc;
  if ((rc = ensure(c, c->d_cert_prop, m * 32))) return rc;
  if ((rc = ensure(c, c->d_cert_masks, (size_t)mask_words(m) * 16))) return rc;
  if ((rc = ensure(c, c->d_cert_total, 64))) return rc;
  if ((rc = ensure(c, c->d_cert_tiles, (m / 1024 + 2) * 8))) return rc;
  if (!c->h_cert_total) {
    if (hipHostMalloc((void **)&c->h_cert_total, 64) != hipSuccess) return IBFT_E_NOMEM;
    void *d = nullptr;
    if (!getenv("IBFT_NO_HOST_DIRECT") && hipHostGetDevicePointer(&d, c->h_cert_total, 0) == hipSuccess) c->dh_cert_total = (uint32_t *)d;
  }
  if (!c->ev_cert_fork) {
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_cert_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_cert_join, hipEventDisableTiming));
  }
  // the columns hold the tree's rows and, with two verdict launches, the batch of the deferred rows behind them (64-aligned)
  if (c->cert_overlap && (rc = alloc_rows(c, 2 * (((uint32_t)c->max_rows + 63u) & ~63u)))) return rc;
  const uint8_t *d_wire = (const uint8_t *)c->d_payload.p;
  wire::node_info *d_nodes = (wire::node_info *)c->d_cert_nodes.p;
  wire::row_info *d_rows = (wire::row_info *)c->d_wire_rows.p;
  uint2 *d_span = (uint2 *)c->d_cert_span.p;
  uint32_t *d_count = (uint32_t *)c->d_cert_count.p, *d_total = (uint32_t *)c->d_cert_total.p, *d_slot = (uint32_t *)c->d_cert_slot.p;
  uint8_t *d_digest = (uint8_t *)c->d_hash.p, *d_sig = (uint8_t *)c->d_sig.p, *d_from = (uint8_t *)c->d_signer.p,
          *d_pre = (uint8_t *)c->d_pre.p, *d_prop = (uint8_t *)c->d_cert_prop.p;
  if (wbytes) HIPCHK(c, hipMemcpyAsync(c->d_payload.p, wire_bytes, wbytes, hipMemcpyHostToDevice, c->stream));
  std::vector<wire::node_info> level0(n);
  for (size_t i = 0; i < n; i++) {
    wire::node_info nd{};
    nd.off = off[i];
    nd.len = off[i + 1] - off[i];
    nd.parent = wire::NO_PARENT;
    nd.ordinal = (uint32_t)i;
    nd.role = wire::ROLE_ROOT;
    level0[i] = nd;
  }
  HIPCHK(c, hipMemcpyAsync(d_nodes, level0.data(), n * sizeof(wire::node_info), hipMemcpyHostToDevice, c->stream));
  std::vector<std::pair<uint32_t, uint32_t>> levels;
  uint32_t lo = 0, hi = (uint32_t)n, carriers = 0;
  for (uint32_t level = 0;; level++) {
    const uint32_t cnt = hi - lo;
    hipLaunchKernelGGL(ibftk::cert_parse_kernel, dim3((cnt + 63) / 64), dim3(64), 0, c->stream, d_wire, d_nodes, lo, hi, d_rows, d_span,
                       d_digest, d_sig, d_from, d_pre);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(ibftk::cert_walk_kernel<false>, dim3(cnt), dim3(64), 0, c->stream, d_wire, d_nodes, d_rows, (const uint2 *)d_span, lo, hi,
                       d_count);
    HIPCHK(c, hipGetLastError());
    if (cnt <= 8192u) {
      hipLaunchKernelGGL(ibftk::cert_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)d_count, d_nodes, lo, hi, hi, carriers,
                         d_slot, d_total, c->dh_cert_total);
    } else {  // a long level: per-tile sums, their scan, per-tile scans 
      const uint32_t tiles = (cnt + 1023u) / 1024u;
      uint2 *d_tiles = (uint2 *)c->d_cert_tiles.p;
      hipLaunchKernelGGL(ibftk::cert_scan_tiles_kernel, dim3(tiles), dim3(1024), 0, c->stream, (const uint32_t *)d_count, cnt, d_tiles);
      hipLaunchKernelGGL(ibftk::cert_scan_offsets_kernel, dim3(1), dim3(1024), 0, c->stream, d_tiles, tiles, hi, carriers, d_total,
                         c->dh_cert_total);
      hipLaunchKernelGGL(ibftk::cert_scan_apply_kernel, dim3(tiles), dim3(1024), 0, c->stream, (const uint32_t *)d_count, d_nodes, lo, cnt,
                         (const uint2 *)d_tiles, d_slot);
    }
    HIPCHK(c, hipGetLastError());
    if (!c->dh_cert_total) HIPCHK(c, hipMemcpyAsync(c->h_cert_total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    levels.push_back({lo, hi});
    const uint32_t total = ((volatile uint32_t *)c->h_cert_total)[0];
    carriers += ((volatile uint32_t *)c->h_cert_total)[1];
    if (total == 0) break;
    if ((uint64_t)hi + total > cap || level == 254) return IBFT_E_TOOBIG;
    hipLaunchKernelGGL(ibftk::cert_walk_kernel<true>, dim3(cnt), dim3(64), 0, c->stream, d_wire, d_nodes, d_rows, (const uint2 *)d_span, lo, hi,
                       d_count);
    HIPCHK(c, hipGetLastError());
    lo = hi;
    hi += total;
  }
  const uint32_t rows = hi;
  for (size_t l = levels.size(); l-- > 1;) {
    const uint32_t cnt = levels[l].second - levels[l].first;
    hipLaunchKernelGGL(ibftk::cert_propagate_kernel, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, (const wire::node_info *)d_nodes, d_rows,
                       levels[l].first, levels[l].second);
    HIPCHK(c, hipGetLastError());
  }
  // The deferred digests (messages that carry certificates, long messages) — a wavefront per message, a sequential sponge each — run
  // on the side stream NEXT TO the verdict launch over all the other rows; the deferred rows then get a small verdict launch of their
  // own as rows [region, region + carriers) of the columns.  That pays when the verdict launch is in its throughput regime (many
  // wavefronts per SIMD: ≥ 4 096 rows, or the known-key kernels): N = 1 024 cold 20.7 → 17.8 ms, warm 12.3 → 10.1; N = 256 warm
  // 2.04 → 1.89.  A small cold launch is one wavefront per SIMD whose duration is its slowest wavefront, and a wavefront that shares
  // its SIMD with a digest wavefront runs at ≈55 % speed (N = 64 cold 1.08 → 1.12 ms): one stream, one verdict launch there.
  // IBFT_CERT_OVERLAP=0: never.
  const bool two = c->cert_overlap && carriers != 0 && (rows >= 4096u || (c->cache_on && c->my_built > 0));
  const uint32_t region = two ? ((rows + 63u) & ~63u) : 0u;
  hipStream_t ds = two ? c->hstream : c->stream;
  if (two) {
    if ((rc = alloc_rows(c, 2 * (((uint32_t)c->max_rows + 63u) & ~63u)))) return rc;
    d_digest = (uint8_t *)c->d_hash.p, d_sig = (uint8_t *)c->d_sig.p, d_from = (uint8_t *)c->d_signer.p, d_pre = (uint8_t *)c->d_pre.p;
    HIPCHK(c, hipEventRecord(c->ev_cert_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(ds, c->ev_cert_fork, 0));
  }
  HIPCHK(c, hipMemsetAsync(d_prop, 0, (size_t)rows * 32, ds));  // rows without a Proposal: a zero digest (cert_finish_kernel skips them)
  if (carriers) {
    hipLaunchKernelGGL(ibftk::cert_digest_wave_kernel, dim3(2 * carriers), dim3(64), 0, ds, d_wire, (const wire::node_info *)d_nodes,
                       (const wire::row_info *)d_rows, (const uint32_t *)d_slot, region, d_digest, d_prop);
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(ibftk::cert_finish_kernel, dim3((rows + 63) / 64), dim3(64), 0, ds, d_wire, d_nodes, (const wire::row_info *)d_rows, rows,
                     d_digest, d_prop, d_pre, two ? 1u : 0u);
  HIPCHK(c, hipGetLastError());
  if (two) {
    hipLaunchKernelGGL(ibftk::cert_carrier_stage_kernel, dim3((carriers + 255) / 256), dim3(256), 0, ds, (const wire::node_info *)d_nodes,
                       (const wire::row_info *)d_rows, (const uint32_t *)d_slot, carriers, region, d_sig, d_from, d_pre);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_cert_join, ds));
  }
  c->ev_used = 0;
  // the verdict launch over all rows of all levels: the digest column holds keccak256(PayloadNoSig) — the seal-style pass; rows that
  // are not judged here (and, with two launches, the deferred rows) are pre-flagged
  if ((rc = enqueue_recover(c, rows, true, 0, false))) return rc;
  if (two) {
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_cert_join, 0));
    // the deferred rows' verdict words: zero whatever an earlier call left there (the first launch only vouches for its own words)
    HIPCHK(c, hipMemsetAsync((uint64_t *)c->d_mask.p + region / 64, 0, (size_t)mask_words(carriers) * 8, c->stream));
    if ((rc = enqueue_recover(c, carriers, true, 0, false, region, true))) return rc;
    hipLaunchKernelGGL(ibftk::cert_scatter_kernel, dim3((carriers + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)d_slot, carriers, region,
                       (uint64_t *)c->d_mask.p);
    HIPCHK(c, hipGetLastError());
  }
  uint64_t *d_hash_mask = (uint64_t *)c->d_cert_masks.p, *d_self_mask = d_hash_mask + mask_words(m);
  hipLaunchKernelGGL(ibftk::cert_compare_kernel, dim3((rows + 255) / 256), dim3(256), 0, c->stream, (const wire::node_info *)d_nodes,
                     (const wire::row_info *)d_rows, (const uint8_t *)d_prop, rows, d_hash_mask, d_self_mask, (uint8_t *)c->d_class.p);
  HIPCHK(c, hipGetLastError());
  const size_t mw = (size_t)mask_words(rows);
  if (out_nodes) HIPCHK(c, hipMemcpyAsync(out_nodes, d_nodes, (size_t)rows * sizeof(wire::node_info), hipMemcpyDeviceToHost, c->stream));
  if (out_rows) HIPCHK(c, hipMemcpyAsync(out_rows, d_rows, (size_t)rows * sizeof(wire::row_info), hipMemcpyDeviceToHost, c->stream));
  if (out_class) HIPCHK(c, hipMemcpyAsync(out_class, c->d_class.p, rows, hipMemcpyDeviceToHost, c->stream));
  if (out_hash_mask) HIPCHK(c, hipMemcpyAsync(out_hash_mask, d_hash_mask, mw * 8, hipMemcpyDeviceToHost, c->stream));
  if (out_self_mask) HIPCHK(c, hipMemcpyAsync(out_self_mask, d_self_mask, mw * 8, hipMemcpyDeviceToHost, c->stream));
  // no tally: the rows of a tree belong to many certificates; their quorum rules are the caller's (From / type / view columns)
  if ((rc = fetch_results(c, rows, out_sender_mask, nullptr, false))) return rc;
  *out_n_rows = rows;
  return IBFT_OK;
}

int ibft_wire_stage_seals(ibft_ctx *c) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (!c->wire_valid) return IBFT_E_INVAL;  // no parsed batch resident (another call restaged the columns)
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t n = c->wire_n;
  if (n) {
    hipLaunchKernelGGL(ibftk::wire_stage_seals_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                       (const wire::row_info *)c->d_wire_rows.p, (const uint8_t *)c->d_seal.p, n,
                       (uint8_t *)c->d_hash.p, (uint8_t *)c->d_sig.p, (uint8_t *)c->d_pre.p);
    HIPCHK(c, hipGetLastError());
    int rc = apply_seal_digest(c, 0, n, false);
    if (rc) return rc;
  }
  c->staged_pre = true;
  return IBFT_OK;
}

static int tally_impl(ibft_ctx *c, const uint8_t *sender20, const uint64_t *mask, size_t n, const uint8_t *proposer20,
                      ibft_tally_t *tally);
int ibft_tally(ibft_ctx *c, const uint8_t *sender20, const uint64_t *mask, size_t n, ibft_tally_t *tally) {
  return tally_impl(c, sender20, mask, n, nullptr, tally);
}
int ibft_tally_prepare(ibft_ctx *c, const uint8_t *sender20, const uint64_t *mask, size_t n, const uint8_t proposer20[20],
                       ibft_tally_t *tally) {
  if (!proposer20) return IBFT_E_INVAL;
  return tally_impl(c, sender20, mask, n, proposer20, tally);
}
static int tally_impl(ibft_ctx *c, const uint8_t *sender20, const uint64_t *mask, size_t n, const uint8_t *proposer20,
                      ibft_tally_t *tally) {
  if (!c || !tally || (n && (!sender20 || !mask))) return IBFT_E_INVAL;
  ctx_lock lk(c);
  if (n > c->max_rows) return IBFT_E_TOOBIG;
  if (!c->have_valset) return IBFT_E_NOVALSET;
  HIPCHK(c, hipSetDevice(c->device));
  // resolve sender -> validator index with a small lookup pass over the table in HBM
  int rc;
  c->wire_valid = false;
  if ((rc = upload(c, c->d_signer, sender20, n * 20))) return rc;
  c->mask_dirty_words = std::max(c->mask_dirty_words, (uint32_t)mask_words(n));
  if ((rc = upload(c, c->d_mask, mask, (size_t)mask_words(n) * 8))) return rc;
  ibftk::lookup_proposer lp{};
  if (proposer20) {
    lp.on = 1;
    memcpy(lp.a, proposer20, 20);
  }
  if (n) {
    hipLaunchKernelGGL(ibftk::lookup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       (const uint8_t *)c->d_signer.p, (const uint32_t *)c->d_vtab.p, c->vslot_mask,
                       (uint32_t)n, (int32_t *)c->d_vidx.p, lp);
    HIPCHK(c, hipGetLastError());
  }
  note_proposer(c, proposer20);
  if ((rc = enqueue_tally(c, (uint32_t)n))) return rc;
  return fetch_results(c, (uint32_t)n, nullptr, tally, true);
}


/* ---- multi-GPU: validator shards + one RCCL all-reduce of verdict words and tally partials ---- */
int ibft_shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi) {
  if (!world || rank >= world || !lo || !hi) return IBFT_E_INVAL;
  const uint64_t per = shard_rows_per_rank(n_total, world);
  *lo = std::min<uint64_t>((uint64_t)rank * per, n_total);
  *hi = std::min<uint64_t>(*lo + per, n_total);
  return IBFT_OK;
}

int ibft_exchange_layout(uint64_t n_total, uint32_t world, uint32_t n_validators, uint32_t n_masks, uint32_t *words_per_rank,
                         uint32_t *seen_words, uint32_t *slots) {
  if (!world || (n_masks != 1 && n_masks != 2)) return IBFT_E_INVAL;
  const uint64_t w = shard_rows_per_rank(n_total, world) / 64;
  if (words_per_rank) *words_per_rank = (uint32_t)w;
  if (seen_words) *seen_words = (n_validators + 63) / 64;
  if (slots) *slots = exchange_slots(n_masks, (uint32_t)w, world, n_validators);
  return IBFT_OK;
}

int ibft_comm_preload(void) { return rccl() ? IBFT_OK : IBFT_E_RCCL; }

int ibft_comm_unique_id(uint8_t id[IBFT_COMM_ID_BYTES]) {
  static_assert(IBFT_COMM_ID_BYTES == sizeof(ncclUniqueId), "ABI");
  if (!id) return IBFT_E_INVAL;
  RcclApi *api = rccl();
  if (!api) return IBFT_E_RCCL;
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return IBFT_E_RCCL;
  memcpy(id, &u, sizeof u);
  return IBFT_OK;
}

int ibft_comm_init(ibft_ctx *c, const uint8_t id[IBFT_COMM_ID_BYTES], uint32_t rank, uint32_t world) {
  if (!c || !id || !world || rank >= world) return IBFT_E_INVAL;
  RcclApi *api = rccl();
  if (!api) return IBFT_E_RCCL;
  ctx_lock lk(c);
  if (c->comm) return IBFT_E_INVAL;
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  NCCLCHK(c, api, api->CommInitRank(&comm, (int)world, u, (int)rank));  // collective: blocks until every rank has joined
  return comm_attach(c, comm, rank, world);
}

int ibft_comm_destroy(ibft_ctx *c) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  (void)hipSetDevice(c->device);
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  comm_release(c);
  return IBFT_OK;
}

int ibft_seals_exchange(ibft_ctx *c, uint64_t n_total) {
  if (!c) return IBFT_E_INVAL;
  if (c->xlocal) return IBFT_E_INVAL;  // a rank of a local group: only the group can run the collective
  RcclApi *api = rccl();
  if (!api) return IBFT_E_RCCL;
  ctx_lock lk(c);
  xplan x{};
  int rc;
  if ((rc = exchange_pre(c, n_total, x))) return rc;
  if ((rc = exchange_collective(c, api, x))) return rc;
  return exchange_post(c, n_total, x);
}

int ibft_comm_info(ibft_ctx *c, uint32_t *rccl_nranks, uint32_t *rccl_rank, int32_t *rccl_device) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  RcclApi *api = rccl();
  if (!c->comm || !api || !api->CommCount || !api->CommUserRank || !api->CommCuDevice) return IBFT_E_INVAL;
  int n = 0, r = 0, d = 0;
  NCCLCHK(c, api, api->CommCount(c->comm, &n));
  NCCLCHK(c, api, api->CommUserRank(c->comm, &r));
  NCCLCHK(c, api, api->CommCuDevice(c->comm, &d));
  if (rccl_nranks) *rccl_nranks = (uint32_t)n;
  if (rccl_rank) *rccl_rank = (uint32_t)r;
  if (rccl_device) *rccl_device = d;
  return IBFT_OK;
}

int ibft_seals_fetch_merged(ibft_ctx *c, uint64_t *out_mask, ibft_tally_t *tally) {
  if (!c) return IBFT_E_INVAL;
  ctx_lock lk(c);
  return fetch_merged_locked(c, out_mask, tally);
}

int ibft_group_create(const int32_t *devices, uint32_t n_dev, uint32_t flags, uint32_t max_rows_total, ibft_group **out) {
  if (!out || !devices || !n_dev || n_dev > (uint32_t)ibftk::XSUM_MAX_RANKS) return IBFT_E_INVAL;
  *out = nullptr;
  // Which collective: RCCL needs one rank per DEVICE.  A group that lists a device more than once (several contexts
  // sharing one MI355X: how the exchange is exercised with W > 1 on a one-GPU box) — or IBFT_GROUP_COLLECTIVE=local
  // with peer-accessible devices — sums the buffers with the library's own kernel instead.
  bool local = false;
  for (uint32_t i = 0; i < n_dev; i++)
    for (uint32_t j = 0; j < i; j++) local = local || devices[i] == devices[j];
  const char *mode = getenv("IBFT_GROUP_COLLECTIVE");
  if (mode && !strcmp(mode, "local")) local = true;
  RcclApi *api = local ? nullptr : rccl();
  if (!local && !api) return IBFT_E_RCCL;
  ibft_group *g = new (std::nothrow) ibft_group();
  if (!g) return IBFT_E_NOMEM;
  g->local = local;
  const uint64_t total = max_rows_total ? max_rows_total : (uint64_t)DEFAULT_MAX_ROWS * n_dev;
  int rc = IBFT_OK;
  for (uint32_t i = 0; i < n_dev && rc == IBFT_OK; i++) {
    ibft_cfg cfg{devices[i], flags, (uint32_t)shard_rows_per_rank(total, n_dev), IBFT_KERNEL_AUTO};
    ibft_ctx *c = nullptr;
    rc = ibft_ctx_create(&cfg, &c);
    if (rc == IBFT_OK) g->ctx.push_back(c);
  }
  if (rc == IBFT_OK && local) {
    // rank 0's device must be able to address every rank's exchange buffer
    for (uint32_t i = 1; i < n_dev && rc == IBFT_OK; i++) {
      if (devices[i] == devices[0]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) != hipSuccess || !can) {
        rc = IBFT_E_INVAL;
        break;
      }
      (void)hipSetDevice(devices[0]);
      const hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = IBFT_E_HIP;
      (void)hipGetLastError();
    }
    for (uint32_t i = 0; i < n_dev && rc == IBFT_OK; i++) rc = comm_attach(g->ctx[i], nullptr, i, n_dev);
  } else if (rc == IBFT_OK) {
    ncclUniqueId u;
    std::vector<ncclComm_t> comms(n_dev, nullptr);
    // one thread initialises every rank: the calls only complete inside the group (ncclCommInitRank's contract)
    bool ok = api->GetUniqueId(&u) == ncclSuccess && api->GroupStart() == ncclSuccess;
    for (uint32_t i = 0; ok && i < n_dev; i++)
      ok = hipSetDevice(devices[i]) == hipSuccess && api->CommInitRank(&comms[i], (int)n_dev, u, (int)i) == ncclSuccess;
    ok = api->GroupEnd() == ncclSuccess && ok;
    for (uint32_t i = 0; ok && i < n_dev; i++) ok = comm_attach(g->ctx[i], comms[i], i, n_dev) == IBFT_OK;
    if (!ok) rc = IBFT_E_RCCL;
  }
  if (rc != IBFT_OK) {
    ibft_group_destroy(g);
    return rc;
  }
  if (n_dev > 1)
    for (uint32_t i = 0; i < n_dev; i++) g->worker.emplace_back(new GroupWorker());
  *out = g;
  return IBFT_OK;
}

void ibft_group_destroy(ibft_group *g) {
  if (!g) return;
  g->worker.clear();  // joins the threads
  for (ibft_ctx *c : g->ctx) ibft_ctx_destroy(c);
  delete g;
}
int ibft_group_is_local(const ibft_group *g) { return g ? (g->local ? 1 : 0) : IBFT_E_INVAL; }

uint32_t ibft_group_size(const ibft_group *g) { return g ? (uint32_t)g->ctx.size() : 0; }
ibft_ctx *ibft_group_ctx(ibft_group *g, uint32_t i) { return (g && i < g->ctx.size()) ? g->ctx[i] : nullptr; }

int ibft_set_seal_digest(ibft_ctx *c, uint32_t mode, const uint8_t *suffix, size_t suffix_len) {
  if (!c || mode > IBFT_SEAL_DIGEST_KECCAK_SUFFIX || (suffix_len && !suffix) || suffix_len > IBFT_SEAL_SUFFIX_MAX) return IBFT_E_INVAL;
  ctx_lock lk(c);
  uint8_t block[72] = {0};
  if (mode == IBFT_SEAL_DIGEST_KECCAK_SUFFIX) {
    if (suffix_len) memcpy(block, suffix, suffix_len);
    block[suffix_len] = 0x01;  // pad10*1 starts right behind the message
  }
  c->seal_digest_mode = mode;
  memcpy(c->seal_suffix_words, block, sizeof block);
  c->staged_n = 0;  // a resident batch was staged under the previous convention
  c->wire_valid = false;
  return IBFT_OK;
}
int ibft_group_set_seal_digest(ibft_group *g, uint32_t mode, const uint8_t *suffix, size_t suffix_len) {
  if (!g) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  for (ibft_ctx *c : g->ctx) {
    int rc = ibft_set_seal_digest(c, mode, suffix, suffix_len);
    if (rc) return rc;
  }
  return IBFT_OK;
}

int ibft_group_set_validators(ibft_group *g, uint64_t height, const uint8_t *addrs20, const uint64_t *power, size_t n) {
  if (!g) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  // the validator table is replicated on every device (28 B per validator); every device on its own thread
  return g->each([&](uint32_t i) { return ibft_set_validators(g->ctx[i], height, addrs20, power, n); });
}

int ibft_group_set_validators_u256(ibft_group *g, uint64_t height, const uint8_t *addrs20, const uint8_t *power_be32, size_t n) {
  if (!g) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  return g->each([&](uint32_t i) { return ibft_set_validators_u256(g->ctx[i], height, addrs20, power_be32, n); });
}

// the exchange of a group call: every rank packed (exchange_pre) → collective → unpack → results of rank 0 to the caller
static int group_exchange(ibft_group *g, uint64_t n, std::vector<xplan> &plan, uint64_t *out_mask, uint64_t *out_mask2,
                          ibft_tally_t *tally) {
  const uint32_t world = (uint32_t)g->ctx.size();
  int rc;
  if (g->local) {
    if ((rc = exchange_collective_local(g->ctx, plan))) return rc;
  } else {
    RcclApi *api = rccl();
    if (!api) return IBFT_E_RCCL;
    // single thread: the calls go inside a group
    bool ok = api->GroupStart() == ncclSuccess;
    for (uint32_t i = 0; ok && i < world; i++) ok = exchange_collective(g->ctx[i], api, plan[i]) == IBFT_OK;
    ok = api->GroupEnd() == ncclSuccess && ok;
    if (!ok) return IBFT_E_RCCL;
  }
  for (uint32_t i = 0; i < world; i++)
    if ((rc = exchange_post(g->ctx[i], n, plan[i]))) return rc;
  for (uint32_t i = world; i-- > 0;)  // every rank holds the merged result; device 0's copy is handed to the caller
    if ((rc = fetch_merged_locked(g->ctx[i], i == 0 ? out_mask : nullptr, i == 0 ? tally : nullptr, i == 0 ? out_mask2 : nullptr)))
      return rc;
  return IBFT_OK;
}

int ibft_group_verify_seals(ibft_group *g, const uint8_t *hash32, const uint8_t *sig65, const uint8_t *signer20,
                            const uint8_t *pre_flags, size_t n, uint64_t *out_mask, ibft_tally_t *tally) {
  if (!g || (n && (!hash32 || !sig65 || !signer20 || !out_mask))) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  const uint32_t world = (uint32_t)g->ctx.size();
  std::vector<std::unique_lock<std::mutex>> locks;
  for (ibft_ctx *c : g->ctx) locks.emplace_back(c->mu);
  for (ibft_ctx *c : g->ctx)
    if (!c->have_valset) return IBFT_E_NOVALSET;
  // every device verifies its own 64-aligned row range: stage, verdict kernel, tally, pack — all asynchronous past the
  // upload, every device driven by its own host thread
  std::vector<xplan> plan(world);
  int rc = g->each([&](uint32_t i) {
    ibft_ctx *c = g->ctx[i];
    uint64_t lo, hi;
    (void)ibft_shard_range(n, i, world, &lo, &hi);
    int r;
    if ((r = seals_stage_locked(c, hash32 + 32 * lo, sig65 + 65 * lo, signer20 + 20 * lo, pre_flags ? pre_flags + lo : nullptr,
                                (size_t)(hi - lo), false)))
      return r;
    if ((r = seals_launch_locked(c, 1))) return r;
    return exchange_pre(c, n, plan[i], 1);
  });
  if (rc) return rc;
  return group_exchange(g, n, plan, out_mask, nullptr, tally);
}

// rows [lo, hi) of a payload column as a column of their own: offsets rebased to the shard's first byte
static void shard_offsets(const uint32_t *off, uint64_t lo, uint64_t hi, std::vector<uint32_t> &out) {
  out.resize((size_t)(hi - lo) + 1);
  for (uint64_t i = lo; i <= hi; i++) out[(size_t)(i - lo)] = off[i] - off[lo];
}

int ibft_group_verify_senders(ibft_group *g, const uint8_t *payload, const uint32_t *off, const uint8_t *sig65,
                              const uint8_t *from20, const uint8_t *pre_flags, size_t n, uint64_t *out_mask,
                              ibft_tally_t *tally) {
  if (!g || (n && (!off || !sig65 || !from20 || !out_mask))) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  const uint32_t world = (uint32_t)g->ctx.size();
  std::vector<std::unique_lock<std::mutex>> locks;
  for (ibft_ctx *c : g->ctx) locks.emplace_back(c->mu);
  std::vector<xplan> plan(world);
  std::vector<std::vector<uint32_t>> loff(world);
  int rc = g->each([&](uint32_t i) {
    ibft_ctx *c = g->ctx[i];
    uint64_t lo, hi;
    (void)ibft_shard_range(n, i, world, &lo, &hi);
    if (n) shard_offsets(off, lo, hi, loff[i]);
    int r = senders_launch_locked(c, n ? payload + off[lo] : nullptr, n ? loff[i].data() : nullptr, sig65 + 65 * lo,
                                  from20 + 20 * lo, pre_flags ? pre_flags + lo : nullptr, (size_t)(hi - lo));
    if (r) return r;
    return exchange_pre(c, n, plan[i], 1);
  });
  if (rc) return rc;
  return group_exchange(g, n, plan, out_mask, nullptr, tally);
}

int ibft_group_verify_messages(ibft_group *g, const uint8_t *payload, const uint32_t *off, const uint8_t *msg_sig65,
                               const uint8_t *from20, const uint8_t *hash32, const uint8_t *hash_len, const uint8_t *seal65,
                               const uint8_t *sender_pre, const uint8_t *valid_pre, size_t n, const uint8_t *raw,
                               size_t raw_len, uint64_t round, const uint8_t *digest32, const uint8_t *proposer20,
                               uint64_t *out_sender_mask, uint64_t *out_valid_mask, ibft_tally_t *tally) {
  if (!g || (n && (!off || !msg_sig65 || !from20 || !hash32 || !hash_len || !out_sender_mask || !out_valid_mask)))
    return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  const uint32_t world = (uint32_t)g->ctx.size();
  std::vector<std::unique_lock<std::mutex>> locks;
  for (ibft_ctx *c : g->ctx) locks.emplace_back(c->mu);
  std::vector<xplan> plan(world);
  std::vector<std::vector<uint32_t>> loff(world);
  // every device judges its own 64-aligned range of MESSAGES completely (both signatures of a COMMIT are rows of its
  // one verdict launch); the proposal is hashed (and remembered) by every device for itself
  int rc = g->each([&](uint32_t i) {
    ibft_ctx *c = g->ctx[i];
    uint64_t lo, hi;
    (void)ibft_shard_range(n, i, world, &lo, &hi);
    if (n) shard_offsets(off, lo, hi, loff[i]);
    note_proposer(c, proposer20, /*shard=*/true);
    int r = messages_launch_locked(c, n ? payload + off[lo] : nullptr, n ? loff[i].data() : nullptr, msg_sig65 + 65 * lo,
                                   from20 + 20 * lo, hash32 + 32 * lo, hash_len + lo, seal65 ? seal65 + 65 * lo : nullptr,
                                   sender_pre ? sender_pre + lo : nullptr, valid_pre ? valid_pre + lo : nullptr,
                                   (size_t)(hi - lo), raw, raw_len, round, digest32);
    c->next_prop_on = c->next_shard = false;
    if (r) return r;
    return exchange_pre(c, n, plan[i], 2, proposer20);
  });
  if (rc) {
    for (ibft_ctx *c : g->ctx) c->have_H = false;
    return rc;
  }
  return group_exchange(g, n, plan, out_sender_mask, out_valid_mask, tally);
}

// Certificates sharded by CARRIER: every device expands and judges the trees of its own contiguous range of the call's
// messages (ibft_verify_certificates_wire on its context, all devices at once); nothing has to be exchanged — the verdicts
// are per row and no tally is taken — so the only work here is the renumbering: the merged rows are in the breadth-first
// order ONE call over all n messages would produce (level by level, within a level rank by rank, which is message order).
int ibft_group_verify_certificates_wire(ibft_group *g, const uint8_t *wire_bytes, const uint32_t *off, size_t n, size_t rows_cap,
                                        size_t *out_n_rows, ibft_cert_node_t *out_nodes, ibft_wire_row_t *out_rows,
                                        uint8_t *out_class, uint64_t *out_sender_mask, uint64_t *out_hash_mask,
                                        uint64_t *out_self_mask) {
  if (!g || !out_n_rows || (n && (!off || !out_sender_mask))) return IBFT_E_INVAL;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return IBFT_E_INVAL;
  *out_n_rows = 0;
  if (n == 0) return IBFT_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  const uint32_t world = (uint32_t)g->ctx.size();
  const size_t per = (n + world - 1) / world;
  struct Part {
    size_t lo = 0, hi = 0, rows = 0;
    uint32_t shift = 0;  // bytes in front of the shard's first message in the buffer its device sees
    std::vector<uint32_t> off;
    std::vector<ibft_cert_node_t> nodes;
    std::vector<ibft_wire_row_t> wrows;
    std::vector<uint8_t> cls;
    std::vector<uint64_t> ms, mh, mself;
  };
  std::vector<Part> part(world);
  int rc = g->each([&](uint32_t i) {
    Part &p = part[i];
    p.lo = std::min(n, (size_t)i * per);
    p.hi = std::min(n, p.lo + per);
    if (p.hi == p.lo) return (int)IBFT_OK;
    ibft_ctx *c = g->ctx[i];
    const size_t cap = std::min<size_t>(rows_cap, c->max_rows), words = (cap + 63) / 64;
    // (a shard that does not start the buffer keeps a few bytes in front of it: no message of it lies at offset 0, so a
    // node's raw_off of 0 can only mean "not parsed" and the offsets can be moved back into the whole buffer below)
    p.shift = std::min<uint32_t>(off[p.lo], 64u);
    p.off.resize(p.hi - p.lo + 1);
    for (size_t k = p.lo; k <= p.hi; k++) p.off[k - p.lo] = off[k] - off[p.lo] + p.shift;
    p.nodes.resize(cap);
    if (out_rows) p.wrows.resize(cap);
    p.cls.assign(cap, 0);
    p.ms.assign(words, 0);
    p.mh.assign(words, 0);
    p.mself.assign(words, 0);
    return ibft_verify_certificates_wire(c, wire_bytes + off[p.lo] - p.shift, p.off.data(), p.hi - p.lo, cap, &p.rows, p.nodes.data(),
                                         out_rows ? p.wrows.data() : nullptr, p.cls.data(), p.ms.data(), p.mh.data(), p.mself.data());
  });
  if (rc) return rc;
  // rows per (rank, level); a rank's rows are breadth first, so its levels are contiguous blocks
  size_t total = 0, levels = 1;
  for (Part &p : part) {
    total += p.rows;
    for (size_t j = 0; j < p.rows; j++) levels = std::max<size_t>(levels, (size_t)p.nodes[j].level + 1);
  }
  if (total > rows_cap) return IBFT_E_TOOBIG;
  std::vector<std::vector<size_t>> count(world, std::vector<size_t>(levels + 2, 0)), start(world, std::vector<size_t>(levels + 2, 0)),
      dest(world, std::vector<size_t>(levels + 2, 0));
  for (uint32_t r = 0; r < world; r++) {
    for (size_t j = 0; j < part[r].rows; j++) count[r][part[r].nodes[j].level]++;
    for (size_t l = 1; l < levels + 2; l++) start[r][l] = start[r][l - 1] + count[r][l - 1];
  }
  size_t base = 0;
  for (size_t l = 0; l < levels + 2; l++)
    for (uint32_t r = 0; r < world; r++) {
      dest[r][l] = base;
      base += count[r][l];
    }
  const size_t words = (total + 63) / 64;
  memset(out_sender_mask, 0, words * 8);
  if (out_hash_mask) memset(out_hash_mask, 0, words * 8);
  if (out_self_mask) memset(out_self_mask, 0, words * 8);
  for (uint32_t r = 0; r < world; r++) {
    const Part &p = part[r];
    const uint32_t byte_base = p.rows ? off[p.lo] - p.shift : 0;
    auto global = [&](size_t local, size_t level) { return dest[r][level] + (local - start[r][level]); };
    for (size_t j = 0; j < p.rows; j++) {
      const size_t level = p.nodes[j].level, at = global(j, level);
      if (out_nodes) {
        ibft_cert_node_t nd = p.nodes[j];
        nd.off += byte_base;
        if (nd.raw_off) nd.raw_off += byte_base;
        if (nd.parent == IBFT_CERT_NO_PARENT)
          nd.ordinal += (uint32_t)p.lo;  // its number among the call's messages
        else
          nd.parent = (uint32_t)global(nd.parent, level - 1);
        // a leaf's first_child is whatever the walk left there: only a node WITH children has a range to renumber (round-3
        // advice: the subtraction underflowed for leaves and consumers that test the bounds first took the slow route)
        nd.first_child = nd.n_children ? (uint32_t)global(nd.first_child, level + 1) : (uint32_t)at;
        out_nodes[at] = nd;
      }
      if (out_rows) out_rows[at] = p.wrows[j];
      if (out_class) out_class[at] = p.cls[j];
      const uint64_t bit = 1ull << (at & 63);
      if ((p.ms[j >> 6] >> (j & 63)) & 1) out_sender_mask[at >> 6] |= bit;
      if (out_hash_mask && ((p.mh[j >> 6] >> (j & 63)) & 1)) out_hash_mask[at >> 6] |= bit;
      if (out_self_mask && ((p.mself[j >> 6] >> (j & 63)) & 1)) out_self_mask[at >> 6] |= bit;
    }
  }
  *out_n_rows = total;
  return IBFT_OK;
}

}  // extern "C"
