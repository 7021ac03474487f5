/*
 * include/ibft_host.h — C ABI of libibft_host.so: the host-side mirror of go-ibft's
 * message store, quorum manager and hot-path callers, sitting ABOVE include/ibftgpu.h.
 *
 * In production this layer is Go (messages/messages.go, core/validator_manager.go,
 * core/ibft.go stay as they are; INTEGRATION.md shows the few lines that change).  This
 * image has no Go toolchain, so the same semantics are provided in C++ and exported here
 * so that the parity tests can drive them exactly like the reference's own unit tests
 * drive the Go code (mock backend via callbacks, or the GPU backend).
 *
 * Messages cross this ABI as protobuf wire bytes (messages/proto/messages.proto).
 * Message lists are packed as repeated { u32 little-endian length, bytes }.
 * Seal lists are packed as repeated { u8 present, u32 signer_len, signer, u32 sig_len, sig }.
 * Returned buffers are owned by the caller: release with ibft_host_buf_free.
 */
#ifndef IBFT_HOST_H
#define IBFT_HOST_H
#include <stddef.h>
#include <stdint.h>

#include "ibftgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ibft_host ibft_host;

typedef struct {
  uint8_t *data;
  size_t len;
  size_t count; /* number of packed items */
} ibft_host_buf;

/* Mock/stock Verifier (core/backend.go:37-56) as callbacks; has_* = 0 means a nil argument. */
typedef struct {
  int (*is_valid_proposal_hash)(void *user, int has_proposal, const uint8_t *raw, size_t raw_len,
                                uint64_t round, int has_hash, const uint8_t *hash, size_t hash_len);
  int (*is_valid_committed_seal)(void *user, int has_hash, const uint8_t *hash, size_t hash_len,
                                 int has_seal, const uint8_t *signer, size_t signer_len,
                                 const uint8_t *sig, size_t sig_len);
  int (*is_valid_validator)(void *user, const uint8_t *wire, size_t len);
  void *user;
  /* consulted by the certificate checks only; NULL = mock defaults (IsProposer false,
   * IsValidProposal true), core/mock_test.go:105-151 */
  int (*is_proposer)(void *user, const uint8_t *id, size_t id_len, uint64_t height, uint64_t round);
  int (*is_valid_proposal)(void *user, const uint8_t *raw, size_t raw_len);
} ibft_host_verifier;

typedef int (*ibft_host_msg_pred)(void *user, const uint8_t *wire, size_t len);
typedef int (*ibft_host_rcc_pred)(void *user, uint64_t round, size_t n_messages);

ibft_host *ibft_host_new(void);
void ibft_host_free(ibft_host *h);
void ibft_host_buf_free(ibft_host_buf *b);

/* SURVEY.md §8f rank 3 — IsValidValidator (core/backend.go:41-45) for n messages given as wire bytes
 * (row i = wire[off[i]..off[i+1])).  mode 0: device walk (ibft_verify_senders_wire) with the stock
 * route for the rows it flags; mode 1: stock route for every row (decode, PayloadNoSig, flatten,
 * ibft_verify_senders) — the "before" the walk replaces.  verdict: n bytes.  host_ms (optional):
 * milliseconds spent in proto decode + re-marshal + flattening on the host; host_rows (optional): rows
 * that took the stock route.  Returns 0, or the negative libibftgpu code.                       */
int ibft_host_verify_senders_wire(ibft_ctx *ctx, const uint8_t *wire, const uint32_t *off, size_t n, int mode,
                                  uint8_t *verdict, double *host_ms, size_t *host_rows);

/* wire helpers: PayloadNoSig (messages/proto/helper.go:12-27) and canonical re-encode */
int ibft_host_payload_no_sig(const uint8_t *wire, size_t len, ibft_host_buf *out);
int ibft_host_reencode(const uint8_t *wire, size_t len, ibft_host_buf *out);

/* messages.Messages (messages/messages.go) */
int ibft_host_store_add(ibft_host *h, const uint8_t *wire, size_t len);
size_t ibft_host_store_num(ibft_host *h, uint64_t height, uint64_t round, uint32_t type);
void ibft_host_store_prune(ibft_host *h, uint64_t height);
int ibft_host_store_get_valid(ibft_host *h, uint64_t height, uint64_t round, uint32_t type,
                              ibft_host_msg_pred pred, void *user, ibft_host_buf *out);
int ibft_host_store_get_extended_rcc(ibft_host *h, uint64_t height, ibft_host_msg_pred pred,
                                     ibft_host_rcc_pred rcc_pred, void *user, ibft_host_buf *out);
/* The same with the candidate set handed to the RCC predicate as packed messages — what isValidRCC(round, msgs) of
 * messages/messages.go:202-245 receives (core/ibft.go:487-495 takes HasQuorum over their senders).                    */
typedef int (*ibft_host_rcc_msgs_pred)(void *user, uint64_t round, const uint8_t *packed, size_t len, size_t n_messages);
int ibft_host_store_get_extended_rcc_msgs(ibft_host *h, uint64_t height, ibft_host_msg_pred pred,
                                          ibft_host_rcc_msgs_pred rcc_pred, void *user, ibft_host_buf *out);
int ibft_host_store_get_most_rc(ibft_host *h, uint64_t min_round, uint64_t height, ibft_host_buf *out);

/* messages/helpers.go */
int ibft_host_has_unique_senders(const uint8_t *packed, size_t len);
int ibft_host_are_valid_pc_messages(const uint8_t *packed, size_t len, uint64_t height, uint64_t round_limit);
/* returns 0 and fills out on success, -1 on ErrWrongCommitMessageType */
int ibft_host_extract_committed_seals(const uint8_t *packed, size_t len, ibft_host_buf *out);

/* core.ValidatorManager: addresses packed as repeated { u32 len, bytes } */
int ibft_host_vm_init(ibft_host *h, const uint8_t *packed_addrs, size_t len, const uint64_t *power, size_t n);
int ibft_host_vm_has_quorum(ibft_host *h, const uint8_t *packed_senders, size_t len);
int ibft_host_vm_has_prepare_quorum(ibft_host *h, const uint8_t *proposal_wire, size_t proposal_len,
                                    const uint8_t *packed_msgs, size_t len);
void ibft_host_vm_quorum(ibft_host *h, uint64_t *lo, uint64_t *hi);

/* the slice of core/state.go the hot path reads */
int ibft_host_set_state(ibft_host *h, uint64_t height, uint64_t round, const uint8_t *proposal_wire,
                        size_t proposal_len);
void ibft_host_set_verifier(ibft_host *h, const ibft_host_verifier *v);
/* Attach the GPU batch backend (BatchVerifier); ctx stays owned by the caller. */
void ibft_host_attach_gpu(ibft_host *h, ibft_ctx *ctx);
void ibft_host_use_batch(ibft_host *h, int on);

/* IBFT.AddMessage (core/ibft.go:1101-1123): 0 rejected, 1 stored, 2 stored + SignalEvent */
int ibft_host_add_message(ibft_host *h, const uint8_t *wire, size_t len);
/* Same decisions as ibft_host_add_message with the O(1) incremental quorum probe (Σ power per
 * view maintained from the store's insert/prune hooks) instead of re-walking the view on every
 * message (core/ibft.go:1113-1120 is O(#stored) per call, O(N²) per phase).  Call
 * ibft_host_enable_quorum_index once after ibft_host_new; ibft_host_vm_init invalidates it.   */
void ibft_host_enable_quorum_index(ibft_host *h);
int ibft_host_add_message_fast(ibft_host *h, const uint8_t *wire, size_t len);
/* Batched ingest: IsValidValidator for many messages in one device call,
 * then the same store/probe logic per accepted message; results[i] as above. */
int ibft_host_add_messages_batch(ibft_host *h, const uint8_t *packed, size_t len, uint8_t *results, size_t n);
/* The receive side: n messages as the transport delivered them (packed wire bytes).  ONE
 * device call answers IsValidValidator for the messages not seen before — byte-identical re-deliveries come
 * from a verdict cache keyed by the full wire bytes (cleared when the validator set changes, pruned with the
 * store) — then IBFT.AddMessage runs per message with the verdict attached (the O(1) quorum probe when
 * ibft_host_enable_quorum_index was called).  results[i]: -1 undecodable, else 0 / 1 / 2 as above.  With no
 * batch backend, or when the device call fails, the per-message verifier answers (ibft_host_fallbacks).   */
int ibft_host_ingest_wire(ibft_host *h, const uint8_t *packed, size_t len, int8_t *results, size_t n,
                          size_t *device_rows, size_t *cache_hits, size_t *device_calls);
/* The same for a micro-batch in the DEVICE's layout: row i is wire[off[i] .. off[i+1]), rows back to back, n + 1 offsets —
 * what a transport that receives into one buffer (or the cgo shim's SoA batcher) holds.  The mirror copies the bytes once
 * (every decoded message of the batch points into that copy) and hands them to the device as they are.               */
int ibft_host_ingest_flat(ibft_host *h, const uint8_t *wire, const uint32_t *off, size_t n, int8_t *results,
                          size_t *device_rows, size_t *cache_hits, size_t *device_calls);
/* Test hook: the receive side looks at a message before decoding it (view, type, payload member — to drop stale views and
 * to route certificate carriers while a helper thread decodes).  out[0..5] = that look {ok, has_view, height, round, type,
 * payload kind}, out[6..11] = the same six from the full decoder; the look must succeed whenever the decoder does and agree
 * with it.                                                                                                            */
int ibft_host_peek_vs_decode(const uint8_t *wire, size_t len, uint64_t out[12]);
/* Test hook: the look has a shortcut for the shape every honest PREPARE / COMMIT has; 0 = shortcut and general walk
 * disagree about this message (a bug), 1 = they agree, 2 = they agree and the message is a row candidate (`simple`).   */
int ibft_host_peek_shortcut_agrees(const uint8_t *wire, size_t len);
/* The receive-side queue: transport threads PUSH what arrives (rows back to back + offsets, copied,
 * never blocks on an ingest in progress); one worker per mirror takes everything pending and ingests it as ONE batch
 * (ibft_host_ingest_flat on at most max_rows rows at a time).  The batch size adapts to the load by itself: arrivals pile up
 * while a device call is in flight and form the next batch.  linger_us > 0: the worker waits that long after the last push
 * for a burst to finish arriving (0 = never wait).  on_signal (optional) is the SignalEvent(type, view) of
 * core/ibft.go:1118-1119, once per ingested batch and message type that reached its quorum probe, called from the worker
 * WITHOUT the mirror's lock (the callee may call ibft_host_handle_* at once).  drain waits until everything pushed before
 * the call has been ingested and reports the counters since queue_start.  The mirror's other entry points stay usable from
 * any thread meanwhile (they are serialised with the worker).                                                       */
typedef struct {
  uint64_t pushed, ingested, stored, rejected, undecodable;
  uint64_t batches, device_calls, cache_hits, max_batch_rows;
  uint64_t signals[4]; /* results of 2 by IbftMessage.type */
  uint64_t ingest_us, device_us; /* wall time of the worker inside ingests, and the part of it inside device calls */
} ibft_host_queue_stats;
typedef void (*ibft_host_signal_fn)(void *user, uint32_t type, uint64_t height, uint64_t round);
int ibft_host_queue_start(ibft_host *h, size_t max_rows, uint32_t linger_us);
void ibft_host_queue_on_signal(ibft_host *h, ibft_host_signal_fn fn, void *user);
int ibft_host_queue_push(ibft_host *h, const uint8_t *wire, const uint32_t *off, size_t n);
/* The queue is BOUNDED (defaults: 256 MiB / 1 Mi messages pending, i.e. pushed and not yet taken by the worker; the byte cap
 * never exceeds 4 GiB − 64 KiB: pending offsets are 32 bits wide): a push that would take it past a cap waits until the
 * worker has taken what is pending — the back-pressure the reference's synchronous AddMessage applies to a transport thread
 * (core/ibft.go:1101-1123) — and a single push larger than a cap returns −2 with nothing queued.  0 = default.          */
void ibft_host_queue_set_caps(ibft_host *h, size_t max_pending_bytes, size_t max_pending_rows);
uint64_t ibft_host_queue_backpressure_waits(ibft_host *h);
int ibft_host_queue_drain(ibft_host *h, ibft_host_queue_stats *out);
void ibft_host_queue_stop(ibft_host *h);
/* Receive-side memory (bounded): messages remembered for byte-identical re-deliveries — only messages that AddMessage
 * STORED are remembered (the entry is the stored message itself, verdicts included; it goes when the store prunes its
 * height), rejected ones leave a 16-byte keyed fingerprint in a FIFO of rejected_cap entries.  When stored_cap entries are
 * reached the table is dropped and re-deliveries are judged again.  Defaults: 262 144 / 16 384.                        */
size_t ibft_host_seen_entries(ibft_host *h);
void ibft_host_set_seen_caps(ibft_host *h, size_t stored_cap, size_t rejected_cap);
/* Message sets (include/ibftgpu.h: ibft_verify_messages).  When the proposal of the current view is already
 * accepted, ibft_host_ingest_wire sends the PREPARE / COMMIT messages of that view through ONE set call per
 * type: IsValidValidator and the handlePrepare / handleCommit closure (core/ibft.go:856-862, :932-944) are
 * answered together, the closure verdicts wait in a table keyed by the stored message, and
 * ibft_host_handle_prepare / _commit only send the device what the table cannot answer
 * (ibft_host_closure_hits = messages of the last handle call answered from it).  The table is dropped when
 * the proposal, the round or the validator set changes and pruned with the store.  use_sets(0) = off.      */
void ibft_host_use_sets(ibft_host *h, int on);
size_t ibft_host_last_set_rows(ibft_host *h);   /* rows of the last ingest that went through set calls */
size_t ibft_host_closure_hits(ibft_host *h);
size_t ibft_host_loop_batch_set_calls(ibft_host *h);
/* Certificates on arrival (include/ibftgpu.h: ibft_verify_certificates_wire).  ibft_host_ingest_wire sends the PREPREPARE
 * and ROUND_CHANGE messages of a micro-batch to the device AS THEY ARRIVED: ONE call settles their own IsValidValidator and
 * every IsValidValidator / IsValidProposalHash that validateProposal, validPC and handleRoundChangeMessage
 * (core/ibft.go:470-551, 683-788, 1162-1231) will ask about the messages NESTED in them — no PayloadNoSig re-marshal of
 * nested messages on the host.  The verdicts wait in tables keyed by the decoded (stored) message objects; the certificate
 * walks (ibft_host_handle_preprepare, ibft_host_handle_round_change) then send the device only what the tables cannot
 * answer.  Messages the device will not vouch for (non-canonical bytes, envelopes longer than it hashes) take the stock
 * route, per message.  use_certs(0) = off.  cert_stats: certificate calls made by ingest, rows (messages, nested ones
 * included) they judged, sender verdicts the last certificate walk took from the tables.                               */
void ibft_host_use_certs(ibft_host *h, int on);
/* Rows instead of objects (default on; needs the quorum index, use_sets, the accepted proposal and a batch backend): a
 * PREPARE / COMMIT of the current view that the backend judged completely FROM ITS BYTES (ibft_verify_messages_wire: the
 * canonical encoding, IsValidValidator, the handlePrepare / handleCommit closure) is stored as a row — where its bytes
 * lie, where From / proposalHash / committedSeal are inside them, its closure verdict — and never decoded;
 * handlePrepare / handleCommit filter the rows, the quorum comes from the index, the committed seals are read off the
 * bytes.  Anything that asks the store for OBJECTS (store_* accessors, the per-message walks, a validator-set or proposal
 * change) turns the view's rows into objects first, verdicts noted: the answers are the same either way
 * (tests/test_host_rows.py).  use_rows(0) = every message becomes an object on arrival.  rows_kept: messages stored as
 * rows so far.                                                                                                        */
/* Round-change certificates judged from the backend's rows (default on; needs use_certs and a batch backend):
 * ibft_verify_certificates_wire returns, for every message nested in a ROUND_CHANGE message's PreparedCertificate, its view,
 * type, From and carried hash next to its IsValidValidator and IsValidProposalHash bits — everything validPC
 * (core/ibft.go:1162-1231) and proposalMatchesCertificate (:516-551) look at.  For a message the backend vouched for the
 * mirror evaluates both on those rows when the message ARRIVES, notes the verdict in the stored message and does not
 * decode the certificate at all (it is decoded if somebody asks for the objects, and copied byte for byte when the message
 * is encoded again); handleRoundChangeMessage takes the noted verdict.  Irregular shapes (a nested message the backend did
 * not judge, type and payload that disagree, hashes that are not 32 bytes, …) are left to the walk over the decoded objects,
 * and so is everything once the validator set changes.  use_rc_rows(0) = decode and walk, as before.  rc_from_rows:
 * ROUND_CHANGE messages decided that way so far.                                                                      */
/* Who pays for a certificate.  Judging a carrier's whole tree on arrival means that ONE message from anybody — its From is
 * just a field — buys up to N² signature checks, where the reference spends one IsValidValidator on it.  mode 1: the
 * envelopes of the carriers of a micro-batch are judged first (one more backend call) and only authenticated carriers have
 * their trees expanded; mode 0: never (one call, lowest latency); mode 2 (default): mode 1 while forged carriers keep
 * arriving — a count of carriers whose own envelope failed, halved every batch, at 4 or more — mode 0 otherwise, so honest
 * traffic keeps its single call and a flood costs the attacker's messages one check each after the first few.  The verdicts
 * are the same in every mode.  roots_first_calls: how often the extra call was made.                                   */
void ibft_host_cert_roots_first(ibft_host *h, int mode);
size_t ibft_host_roots_first_calls(ibft_host *h);
/* wall time the last ibft_host_ingest_* call spent inside the batch backend (the device calls), in milliseconds */
double ibft_host_last_ingest_device_ms(ibft_host *h);
void ibft_host_use_rc_rows(ibft_host *h, int on);
size_t ibft_host_rc_from_rows(ibft_host *h);
/* The same switch covers the RoundChangeCertificate of a PREPREPARE message (validateProposal, core/ibft.go:683-788):
 * unique senders, their quorum, every ROUND_CHANGE message of the proposal's view and validly signed, validPC of every
 * nested PreparedCertificate and the highest prepared round with its hash are decided from the rows on arrival and the
 * certificate stays undecoded; handlePrePrepare then asks only what depends on this node and on the application
 * (the common checks, IsProposer(ID), IsValidProposal, the final IsValidProposalHash).  pp_from_rows: PREPREPARE messages
 * decided that way so far.                                                                                             */
size_t ibft_host_pp_from_rows(ibft_host *h);
/* PROCESS-WIDE, optional: keep up to `bytes` of freed C heap in the process (glibc mallopt: M_TRIM_THRESHOLD, M_TOP_PAD,
 * M_MMAP_THRESHOLD) instead of handing it back to the kernel.  The mirror's memory has the rhythm of the chain — the
 * messages of a height are stored, then pruned — and with the default thresholds every height's buffers are fresh pages
 * again: first-touch page faults were ≈25 % of the mirror's own time for a height of config #3.  Call it
 * once at start-up if the C heap of the process is yours to tune (in a Go node it is: the Go heap is not malloc's).
 * 0 = set, −1 = the C library refused a value.                                                                        */
int ibft_host_retain_heap(size_t bytes);
/* Rows point into the buffer of the batch they arrived in and keep it alive.  When less than a quarter of the bytes of a
 * batch of at least repack_min_bytes (default 256 KiB) was stored — a flood of rejected messages around a few honest ones
 * — the stored rows are moved into a buffer of their own and the batch's buffer is let go, so that rejected bytes are not
 * held until the height is pruned.  repacked_bytes: bytes moved that way so far.                                      */
void ibft_host_set_repack_min_bytes(ibft_host *h, size_t bytes);
size_t ibft_host_repacked_bytes(ibft_host *h);
/* a8 on the device: handlePrepare / handleCommit take the quorum decision from ibft_tally_prepare / ibft_tally (tally_kernel:
 * HasPrepareQuorum's proposer rule included, /root/reference/core/validator_manager.go:77-127) over the senders that survived
 * the walk, instead of the mirror's quorum index; the index's answer is kept as a cross-check — `mismatches` counts the
 * decisions on which the two differed (must stay 0), `calls` the decisions the device took.  Off by default.               */
void ibft_host_use_device_quorum(ibft_host *h, int on);
void ibft_host_device_quorum_stats(ibft_host *h, size_t *calls, size_t *mismatches);
void ibft_host_use_rows(ibft_host *h, int on);
size_t ibft_host_rows_kept(ibft_host *h);
/* What the rows of one view hold: live rows, row slots (replaced / pruned ones included until the next compaction:
 * ≤ 2·live + 32) and batch buffers still referenced — a buffer is let go when its last live row is replaced or pruned,
 * so a sender that keeps replacing its message pins one batch, not one per replacement.                                */
void ibft_host_lean_stats(ibft_host *h, uint64_t height, uint64_t round, uint32_t type, size_t *live, size_t *slots,
                          size_t *buffers);
void ibft_host_cert_stats(ibft_host *h, size_t *calls, size_t *rows, size_t *hits);
size_t ibft_host_loop_batch_cert_calls(ibft_host *h);
/* handlePrePrepare (core/ibft.go:792-813): 1 = a stored PREPREPARE of (height, round) passes validateProposal0 (round 0) /
 * validateProposal; the first such message in msg.  Rejected ones are pruned from the store, as in the reference.   */
int ibft_host_handle_preprepare(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *msg);
/* Measurement aid (tools/cert_from_wire.py): the sender checks of every message of a batch of certificate trees, route 0 = the
 * bytes straight to ibft_verify_certificates_wire, route 1 = decode + collect nested messages + PayloadNoSig re-marshal +
 * flatten on the host (host_ms) + ibft_verify_senders.  rows_cap: rows the tree may expand to (0 = 65 536; the context's
 * max_rows bounds it too).  rows / valid = messages judged / accepted.                                                   */
int ibft_host_cert_routes(ibft_ctx *ctx, const uint8_t *wire, const uint32_t *off, size_t n, int route, size_t rows_cap,
                          size_t *rows, size_t *valid, double *host_ms, double *total_ms);
/* A batch backend that loops over the callback Verifier (no device): the batch control flow — one call per walk,
 * verdict tables, fallback — for CPU-side tests.  fail_mask bits: 1 hash batches, 2 seal batches, 4 sender
 * batches, 8 message-set calls, 16 certificate-tree calls report "device unavailable"; 32: so does the quorum call
 * (ibft_host_use_device_quorum), 64: the quorum call answers the OPPOSITE (the mirror then counts a mismatch).                       */
void ibft_host_use_loop_batch(ibft_host *h, int fail_mask);
size_t ibft_host_loop_batch_calls(ibft_host *h);
/* batches that fell back to the per-message verifier because the batch backend reported failure            */
size_t ibft_host_fallbacks(ibft_host *h);
/* SURVEY.md §5 "min batch for GPU" (round 6): a batch of fewer than `rows` rows is DECLINED by the batch backend — it
 * answers "not offered", as it does when the device is unavailable — and the stock per-message closures of
 * core/ibft.go:856-862 / 932-944 / 1128 run: a launch has a floor of ≈ 0.2 ms whatever the row count, one host core
 * recovers a signature in 31–50 µs, and every validator count the reference itself tests (4, 6, ≤ 30:
 * core/consensus_test.go:139, core/byzantine_test.go:21, core/rapid_test.go:156) lies around that crossover
 * (INTEGRATION.md §2 has the measured table).  0 = never decline (the default; the environment variable
 * IBFT_MIN_DEVICE_ROWS sets it for every mirror of the process that does not call this).  Verdicts are the same either
 * way — the declined batch takes the fallback path every handler has.                                                   */
void ibft_host_set_min_device_rows(ibft_host *h, size_t rows);
size_t ibft_host_declined_batches(ibft_host *h);
/* Certificate checks (core/ibft.go: validPC :1162-1231, proposalMatchesCertificate :516-551,
 * validateProposal0 :658-680, validateProposal :683-788).  NULL wire pointers are Go nils.  With
 * ibft_host_use_batch(1) and a GPU attached, all sender signatures / hashes of the certificate go
 * to the device in one batch; ibft_host_last_cert_batch reports how many.  Return 1/0, -1 = undecodable. */
void ibft_host_set_id(ibft_host *h, const uint8_t *id, size_t len);
/* A native Backend.IsProposer (core/backend.go:46-48) for measurements and simulators: the round-robin rule of the
 * reference's cluster tests (core/helpers_test.go:214-225), proposer = addrs[(height·use_height + round) mod n] — a
 * certificate walk asks IsProposer once per nested message (core/ibft.go:1206-1225), 29 000 times for a round change at
 * N = 256, which a callback into an interpreter would dominate.  An empty list restores the is_proposer callback.  */
int ibft_host_set_round_robin_proposer(ibft_host *h, const uint8_t *packed_addrs, size_t len, int use_height);
int ibft_host_valid_pc(ibft_host *h, const uint8_t *pc_wire, size_t len, uint64_t round_limit, uint64_t height);
int ibft_host_proposal_matches_certificate(ibft_host *h, const uint8_t *proposal_wire, size_t plen,
                                           const uint8_t *pc_wire, size_t clen);
int ibft_host_validate_proposal0(ibft_host *h, const uint8_t *msg_wire, size_t len, uint64_t height, uint64_t round);
int ibft_host_validate_proposal(ibft_host *h, const uint8_t *msg_wire, size_t len, uint64_t height, uint64_t round);
void ibft_host_last_cert_batch(ibft_host *h, size_t *senders, size_t *hashes);
/* handlePrepare / handleCommit (core/ibft.go:855-889, 931-967): 1 = quorum reached */
int ibft_host_handle_prepare(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *prepared);
int ibft_host_handle_commit(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *seals);
/* handleRoundChangeMessage (core/ibft.go:470-512): 1 = an extended RCC exists for (height, round), its messages in
 * rcc.  In batch mode every nested signature of every stored ROUND-CHANGE message's prepared certificate goes to
 * the device in one sender batch and the certificate hashes in one hash batch per distinct proposal
 * (ibft_host_last_cert_batch reports both counts).                                                          */
int ibft_host_handle_round_change(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *rcc);

#ifdef __cplusplus
}
#endif
#endif
