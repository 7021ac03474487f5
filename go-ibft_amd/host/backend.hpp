// backend.hpp — the Verifier boundary as the host sees it, and the hot-path callers.
//
// Product host code.
//   Verifier        = core.Verifier, per message (/root/reference/core/backend.go:37-56)
//   BatchVerifier   = the OPTIONAL batch interface INTEGRATION.md adds next to it; one call
//                     per GetValidMessages walk instead of one per message
//   GpuBackend      = BatchVerifier on top of libibftgpu.so (include/ibftgpu.h): flattens
//                     messages into byte columns (SoA), one C-ABI call per batch
//   HotPath         = the callers around the boundary, restated from core/ibft.go:
//                     AddMessage :1101-1123, isAcceptableMessage :1126-1149,
//                     handlePrepare :855-889, handleCommit :931-967,
//                     hasQuorumByMsgType :1273-1284.  RunSequence itself is untouched Go.
#pragma once
#include <memory>
#include <unordered_map>

#include "../../include/ibftgpu.h"
#include "messages.hpp"

namespace ibft {

struct Verifier {
  virtual ~Verifier() = default;
  // nil-able arguments are pointers, exactly like the Go interface
  virtual bool IsValidProposalHash(const Proposal *proposal, const bytes *hash) = 0;
  virtual bool IsValidCommittedSeal(const bytes *proposalHash, const CommittedSeal *seal) = 0;
  virtual bool IsValidValidator(const IbftMessage &msg) = 0;
  // the rest of core.Verifier / Backend that the certificate checks consult (cold path, per node)
  virtual bool IsProposer(const bytes & /*id*/, uint64_t /*height*/, uint64_t /*round*/) { return false; }  // mock default
  virtual bool IsValidProposal(const bytes & /*rawProposal*/) { return true; }                               // mock default
  virtual bytes ID() { return bytes(); }
};

// rows of a certificate tree (ibft_verify_certificates_wire): nodes[row] says where a row's nested messages are
// (first_child, n_children, role); cls / sender / hash / self are one byte per row
struct CertVerdicts {
  size_t n_rows = 0;
  std::vector<ibft_cert_node_t> nodes;
  std::vector<ibft_wire_row_t> rows;  // the parsed fields of every row (view, type, payload kind, From, carried proposal hash)
  std::vector<uint8_t> cls, sender, hash, self;
};

struct BatchVerifier {
  virtual ~BatchVerifier() = default;
  // SURVEY §5 "min batch for GPU" (round 6).  A launch has a floor of ≈ 0.2 ms whatever the row count, one core recovers a
  // signature in ≈ 31–50 µs: below a handful of rows the per-message Verifier is FASTER than the device — and every validator
  // count the reference itself tests lives there (4: core/consensus_test.go:139, 6: core/byzantine_test.go:21, ≤ 30:
  // core/rapid_test.go:156).  A batch with fewer than min_device_rows rows is DECLINED: the batch method answers false /
  // −1, exactly what it answers when the device is unavailable, and the caller runs the stock closures (same verdicts by
  // construction — the fallback path of every handler).  0 = never decline.  The measured crossover: INTEGRATION.md §2.
  size_t min_device_rows = 0;
  size_t declined = 0;  // batches answered "not offered" for being too small
  bool declines(size_t rows) {
    if (rows == 0 || rows >= min_device_rows) return false;
    declined++;
    return true;
  }
  // verdict[i] == what handlePrepare's closure would return for msgs[i] (ibft.go:856-862)
  virtual bool VerifyPrepareBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs,
                                  std::vector<uint8_t> &verdict) = 0;
  // verdict[i] == handleCommit's closure (ibft.go:932-944): a1 ∧ a2 with a2 short-circuited
  virtual bool VerifyCommitBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs,
                                 std::vector<uint8_t> &verdict) = 0;
  // verdict[i] == IsValidValidator(msgs[i]) (ibft.go:1128)
  virtual bool VerifySenderBatch(const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &verdict) = 0;
  // A whole PREPARE or COMMIT set at once (include/ibftgpu.h: ibft_verify_messages): sender[i] == IsValidValidator(msgs[i])
  // and closure[i] == what handlePrepare's / handleCommit's closure returns for msgs[i] against `proposal` — all three
  // predicates are pure, so a message can be judged completely when it arrives.  false = not offered / device unavailable.
  virtual bool VerifyMessageSet(const Proposal * /*proposal*/, MessageType /*type*/, const std::vector<MsgPtr> & /*msgs*/,
                                std::vector<uint8_t> & /*sender*/, std::vector<uint8_t> & /*closure*/) {
    return false;
  }
  // §8f rank 2 from the transport's bytes (include/ibftgpu.h: ibft_verify_certificates_wire): n raw messages of any type;
  // every IbftMessage nested in them — RoundChangeCertificate, PreparedCertificate, to any depth — becomes a row, breadth
  // first, with its IsValidValidator verdict, the proposal-hash bits of proposalMatchesCertificate / validateProposalCommon
  // and a class byte (0 = judged; otherwise the stock route decides that message).  false = not offered / device
  // unavailable / tree too large.
  virtual bool VerifyCertificatesWire(const uint8_t * /*wire*/, const uint32_t * /*off*/, size_t /*n*/, CertVerdicts & /*out*/) {
    return false;
  }
  // A batch of raw messages judged COMPLETELY from their bytes (include/ibftgpu.h: ibft_verify_messages_wire): sender[i] =
  // IsValidValidator(message i); judged[i] != 0 for the messages the backend vouches are the CANONICAL encoding of a PREPARE
  // / COMMIT of the view (height, round) — closure[i] is then the handlePrepare / handleCommit closure against `proposal`.
  // Such a message can be kept as a row (messages.hpp: LeanRow).  false = not offered / device unavailable.
  virtual bool VerifyMessagesWire(const uint8_t * /*wire*/, const uint32_t * /*off*/, size_t /*n*/, uint64_t /*height*/,
                                  uint64_t /*round*/, const Proposal & /*proposal*/, std::vector<uint8_t> & /*sender*/,
                                  std::vector<uint8_t> & /*closure*/, std::vector<uint8_t> & /*judged*/) {
    return false;
  }
  // a8 on the device (include/ibftgpu.h: ibft_tally / ibft_tally_prepare): ValidatorManager.HasQuorum over the sender set
  // (proposer == nullptr) or HasPrepareQuorum with proposalMessage.From = *proposer (validator_manager.go:77-127).
  // 1 / 0 = the decision, −1 = not offered / device unavailable / an address that is not 20 bytes (the host decides).
  virtual int QuorumOfSenders(const std::vector<bytes> & /*senders*/, const bytes * /*proposer*/) { return -1; }
};

// SoA columns handed to the C ABI (plain bytes, no pointers inside: cgo-safe layout)
struct SealColumns {
  std::vector<uint8_t> hash32, hash_len, sig65, signer20, pre_flags;
  size_t n = 0;
};
struct SenderColumns {
  std::vector<uint8_t> payload, sig65, from20, pre_flags;
  std::vector<uint32_t> off;
  size_t n = 0;
};
// Flatten COMMIT messages (ExtractCommitHash / ExtractCommittedSeal) into columns; rows the
// reference would reject without calling the crypto (nil payload, wrong lengths) get pre_flags.
void flatten_commits(const std::vector<MsgPtr> &msgs, SealColumns &out);
void flatten_prepares(const std::vector<MsgPtr> &msgs, SealColumns &out);  // hash32/hash_len only
void flatten_senders(const std::vector<MsgPtr> &msgs, SenderColumns &out);

class GpuBackend : public BatchVerifier {
 public:
  // IBFT_MIN_DEVICE_ROWS (environment): the node operator's knob, read once per backend; ibft_host_set_min_device_rows overrides
  explicit GpuBackend(ibft_ctx *ctx);
  bool VerifyPrepareBatch(const Proposal *, const std::vector<MsgPtr> &, std::vector<uint8_t> &) override;
  bool VerifyCommitBatch(const Proposal *, const std::vector<MsgPtr> &, std::vector<uint8_t> &) override;
  bool VerifySenderBatch(const std::vector<MsgPtr> &, std::vector<uint8_t> &) override;
  bool VerifyMessageSet(const Proposal *, MessageType, const std::vector<MsgPtr> &, std::vector<uint8_t> &,
                        std::vector<uint8_t> &) override;
  // §8f rank 3: IsValidValidator for n messages straight from their wire bytes.  The device walks,
  // hashes and verifies the PREPARE/COMMIT rows it can vouch for (ibft_verify_senders_wire); rows
  // it flags (other payload kinds, unknown fields, non-canonical encodings) take the stock route —
  // proto decode, PayloadNoSig re-marshal, ibft_verify_senders — and rows that do not decode are
  // verdict 0 (the reference drops them before AddMessage).  Same verdicts as VerifySenderBatch on
  // the decoded messages.  stats (optional): rows sent to the host route, host milliseconds.
  struct WireStats {
    size_t host_rows = 0;
    double host_ms = 0.0;
  };
  bool VerifySendersWire(const uint8_t *wire, const uint32_t *off, size_t n, std::vector<uint8_t> &verdict,
                         WireStats *stats = nullptr);
  // The same batch judged completely (ibft_verify_messages_wire): sender[i] as above; for the canonical PREPARE / COMMIT
  // messages of the view (height, round) — judged[i] != 0 — closure[i] is the handlePrepare / handleCommit closure against
  // `proposal`, both signatures of a COMMIT verified in the same launch.  Rows the device flags take the stock sender route.
  bool VerifyMessagesWire(const uint8_t *wire, const uint32_t *off, size_t n, uint64_t height, uint64_t round,
                          const Proposal &proposal, std::vector<uint8_t> &sender, std::vector<uint8_t> &closure,
                          std::vector<uint8_t> &judged) override;
  bool VerifyCertificatesWire(const uint8_t *wire, const uint32_t *off, size_t n, CertVerdicts &out) override;
  int QuorumOfSenders(const std::vector<bytes> &senders, const bytes *proposer) override;
  size_t cert_rows_cap = 65536;  // rows one certificate call may expand to (the context's max_rows bounds it too)
  int last_rc = 0;

 private:
  bool sender_batch(const std::vector<MsgPtr> &, std::vector<uint8_t> &);  // VerifySenderBatch without the min-rows rule
  ibft_ctx *ctx_;
};

// BatchVerifier that answers every batch by looping over a per-message Verifier: the batch control flow of the
// hot path (one callback per walk, verdict tables, device-failure fallback) without a device — what the
// CPU-side tests drive.  fail_* make the corresponding batch call report "device unavailable".
class LoopBatch : public BatchVerifier {
 public:
  explicit LoopBatch(Verifier *v) : v_(v) {}
  bool VerifyPrepareBatch(const Proposal *, const std::vector<MsgPtr> &, std::vector<uint8_t> &) override;
  bool VerifyCommitBatch(const Proposal *, const std::vector<MsgPtr> &, std::vector<uint8_t> &) override;
  bool VerifySenderBatch(const std::vector<MsgPtr> &, std::vector<uint8_t> &) override;
  bool VerifyMessageSet(const Proposal *, MessageType, const std::vector<MsgPtr> &, std::vector<uint8_t> &,
                        std::vector<uint8_t> &) override;
  // decodes the messages and asks the per-message Verifier about every nested message, in the row order of the device
  bool VerifyCertificatesWire(const uint8_t *wire, const uint32_t *off, size_t n, CertVerdicts &out) override;
  // decodes, re-encodes (judged = the bytes ARE the canonical encoding of a PREPARE / COMMIT of the view) and asks the
  // per-message Verifier: the row-keeping ingest path without a device (counts as a message-set call; fail_sets fails it)
  bool VerifyMessagesWire(const uint8_t *wire, const uint32_t *off, size_t n, uint64_t height, uint64_t round,
                          const Proposal &proposal, std::vector<uint8_t> &sender, std::vector<uint8_t> &closure,
                          std::vector<uint8_t> &judged) override;
  // a8 without a device: HasQuorum / HasPrepareQuorum restated from scratch over `quorum_vm`'s powers (a set of addresses,
  // the proposer's seat, a sender equal to the proposer voids) — the CPU-side stand-in for ibft_tally[_prepare]; fail_quorum
  // makes it report "device unavailable", wrong_quorum inverts the answer (the mirror must count the mismatch)
  int QuorumOfSenders(const std::vector<bytes> &senders, const bytes *proposer) override;
  const ValidatorManager *quorum_vm = nullptr;
  bool fail_quorum = false, wrong_quorum = false;
  size_t quorum_calls = 0;
  bool fail_hashes = false, fail_seals = false, fail_senders = false, fail_sets = false, fail_certs = false;
  size_t calls = 0;      // batch calls answered
  size_t set_calls = 0;  // of which message-set calls
  size_t cert_calls = 0; // of which certificate-tree calls

 private:
  Verifier *v_;
};

enum class StateName { newRound, prepare, commit, fin };

class HotPath {
 public:
  Messages messages;
  ValidatorManager validatorManager;
  Verifier *verifier = nullptr;     // stock path
  BatchVerifier *batch = nullptr;   // optional; used when set and use_batch is true
  bool use_batch = false;
  // the slice of core/state.go the hot path reads
  uint64_t height = 0, round = 0;
  StateName stateName = StateName::newRound;
  MsgPtr proposalMessage;  // state.getProposalMessage()
  std::vector<std::optional<CommittedSeal>> committedSeals;
  std::vector<MsgPtr> preparedMessages;  // PC.PrepareMessages after finalizePrepare

  const Proposal *getProposal() const {  // state.go:135-144
    return proposalMessage ? extract_proposal(*proposalMessage) : nullptr;
  }
  // IBFT.AddMessage: 0 = rejected, 1 = stored, 2 = stored and SignalEvent fired
  int AddMessage(MsgPtr m, bool accepted = false);  // accepted: isAcceptableMessage already answered true
  // Same decisions with the O(1) quorum probe (QuorumIndex) instead of the O(#stored) walk;
  // call EnableQuorumIndex() once, and NotifyValidatorSetChanged() after validatorManager.Init.
  int AddMessageFast(MsgPtr m, bool accepted = false);
  void EnableQuorumIndex();
  void NotifyValidatorSetChanged() {
    quorumIndex.Invalidate();
    valset_epoch_++;   // every sender verdict noted in a message object was computed against the old set: ignored from now on
    seen_.clear();
    seen_has_votes_ = false;
    seen_rejected_.clear();
  }
  QuorumIndex quorumIndex;
  // sender_ok: IsValidValidator already answered (by a device batch or the arrival-time verdict in the message); null = ask
  // the per-message verifier (the stock path)
  bool isAcceptableMessage(const IbftMessage &m, const bool *sender_ok = nullptr);
  bool hasQuorumByMsgType(const std::vector<MsgPtr> &msgs, uint32_t type);
  bool hasQuorumOfStoredView(const View &view, uint32_t type, const std::vector<MsgPtr> &msgs);
  bool handlePrepare(const View &view);
  bool handleCommit(const View &view);
  // handlePrePrepare (core/ibft.go:792-813): the first stored PREPREPARE of the view that validateProposal0 (round 0) /
  // validateProposal accepts; nil = none
  MsgPtr handlePrePrepare(const View &view);
  // handleRoundChangeMessage (core/ibft.go:470-512): the extended RCC for `view`, empty = nil.  With use_batch the
  // signatures of every prepared certificate carried by every stored ROUND-CHANGE message of the height and all
  // their proposal-hash checks (grouped by the proposal they refer to) are answered from ONE sender batch and one
  // hash batch per distinct proposal, gathered by a pre-pass under the store's lock; the walk is the reference's.
  std::vector<MsgPtr> handleRoundChangeMessage(const View &view);
  // device batches of the last handle* / certificate call that fell back to the per-message verifier
  size_t fallbacks = 0;

  // ---- §8f rank 1: the receive side.  Messages arrive as wire bytes; IngestWire answers IsValidValidator for a
  // whole micro-batch with one device call (verdicts of byte-identical messages — gossip re-deliveries — come
  // from a cache keyed by the full wire bytes: the verdict is a pure function of them and of the validator set,
  // so NotifyValidatorSetChanged clears it, and a height prune drops what can no longer be accepted), then runs
  // AddMessage (AddMessageFast when the quorum index is enabled) per message with the verdict attached.
  // results[i]: −1 undecodable, else AddMessage's 0 / 1 / 2.
  // With use_sets (default) the PREPARE / COMMIT messages of the CURRENT view are judged completely on arrival when the
  // proposal is already accepted: one VerifyMessageSet call per type answers IsValidValidator and the handlePrepare /
  // handleCommit closure together (both signatures of a COMMIT in one verdict launch); the closure verdicts wait in a
  // table keyed by the stored message and handlePrepare / handleCommit only send the device what the table cannot answer.
  struct IngestStats {
    size_t device_rows = 0, cache_hits = 0, device_calls = 0, set_rows = 0;
    double device_ms = 0.0;  // wall time inside the batch backend's calls (the rest of an ingest is the mirror's own work)
  };
  bool use_sets = true;
  // device_quorum: the quorum decision of handlePrepare / handleCommit is the DEVICE's (BatchVerifier::QuorumOfSenders →
  // ibft_tally_prepare / ibft_tally over the senders that survived the walk) — hasQuorumByMsgType computed by tally_kernel,
  // HasPrepareQuorum's proposer rule included; the quorum index's answer is kept as a cross-check (mismatches counted, the
  // device's answer taken).  Off by default: the index answers in O(1) without a device call.
  bool device_quorum = false;
  size_t device_quorum_calls = 0, device_quorum_mismatches = 0;
  bool quorumDecision(uint32_t type, const std::vector<bytes> &senders, bool host_answer);
  size_t closure_hits = 0;  // messages of the last handlePrepare / handleCommit whose closure verdict was already known
  // With use_certs (default) the PREPREPARE / ROUND_CHANGE messages of a micro-batch go to the device as they arrived
  // (VerifyCertificatesWire): their own IsValidValidator AND every IsValidValidator / IsValidProposalHash that
  // validateProposal, validPC and handleRoundChangeMessage will later ask about the messages nested in them are settled
  // in that one call — no PayloadNoSig re-marshal of nested messages on the host — and wait in tables keyed by the
  // decoded (stored) message objects, which a decoded message lists in the device's row order.  The certificate walks
  // then send the device only what the tables cannot answer (normally nothing).
  bool use_certs = true;
  // With use_lean (default; needs the quorum index, use_sets, the proposal at hand and a backend that judges bytes
  // completely): a PREPARE / COMMIT the backend vouched for is kept as a ROW — no object is built for it — and
  // handlePrepare / handleCommit run over the rows; anything that wants objects materialises them (messages.hpp).
  bool use_lean = true;
  // With use_rc_rows (default): the certificate of a ROUND_CHANGE message the backend vouched for is judged from the
  // backend's rows when the message arrives and is NOT decoded (proto.hpp: RoundChangeMessage::certificate_deferred);
  // handleRoundChangeMessage takes the noted verdict.  rc_from_rows: messages decided that way so far.
  bool use_rc_rows = true;
  // 0 = the trees of all carriers of a micro-batch are expanded and judged in the one certificate call; 1 = the carriers'
  // own envelopes are judged first (one more backend call) and only authenticated carriers have their trees expanded;
  // 2 (default) = 1 while forged carriers keep arriving (a decayed count of carriers whose envelope failed), 0 otherwise.
  int cert_roots_first = 2;
  size_t roots_first_calls = 0;
  size_t rc_from_rows = 0;
  size_t repack_min_bytes = 256 << 10;  // batches of at least this size are checked for a low stored share (< 1/4 of the bytes)
  size_t repacked_bytes = 0;            // bytes of stored rows moved out of such batches' buffers
  size_t pp_from_rows = 0;  // PREPREPARE messages whose RoundChangeCertificate was judged from rows (and left undecoded)
  double last_ingest_device_ms = 0.0;  // wall time the last IngestFlat spent inside the batch backend's calls
  size_t lean_rows = 0;        // messages ingested as rows so far
  bool prepared_as_rows = false;  // the last successful handlePrepare ran over rows: PC.PrepareMessages = PreparedWire()
  View prepared_view{};
  bool committed_as_rows = false;  // the last successful handleCommit ran over rows: their seals are read on demand
  View committed_view{};
  const std::vector<std::optional<CommittedSeal>> &CommittedSeals();  // state.getCommittedSeals() after handleCommit
  size_t PackCommittedSeals(bytes &out);                              // the same, packed for the C API; returns their number
  std::vector<bytes> PreparedWire();  // the prepared messages' wire bytes (rows: as stored; objects: encoded)
  size_t cert_calls = 0, cert_rows = 0;  // certificate-tree calls made by IngestWire, rows they judged
  size_t cert_hits = 0;  // sender verdicts the last certificate walk took from the arrival-time tables
  bool IngestWire(const std::vector<bytes> &raw, std::vector<int> &results, IngestStats *stats = nullptr);
  // The same for a micro-batch in the device's own layout — row i is wire[off[i] .. off[i+1]), rows back to back: what a
  // transport that receives into one buffer (or the cgo shim's SoA batcher) holds.  The bytes are copied ONCE (the buffer
  // every decoded message of the batch points into) and, when no row is answered from the cache, handed to the device
  // as they are.  results[i]: −1 undecodable, else AddMessage's 0 / 1 / 2.
  // types (optional, n bytes): IbftMessage.type of every row that decoded (0xFF otherwise) — what a caller needs to turn a
  // result of 2 into the SignalEvent(type, view) of core/ibft.go:1119.
  // owned: a buffer the rows lie in that the caller SHARES with the mirror (it must stay unchanged for as long as anybody
  // holds it): the stored messages point into it and keep it alive, and the batch is not copied.
  bool IngestFlat(const uint8_t *wire, const uint32_t *off, size_t n, int8_t *results, IngestStats *stats = nullptr,
                  uint8_t *types = nullptr, const std::shared_ptr<const void> &owned = nullptr);
  // IBFT.AddMessage with IsValidValidator already answered (AddMessageFast when the quorum index is enabled)
  int addWithVerdict(MsgPtr m, bool sender_ok);
  void PruneVerdictCache(uint64_t below_height);
  // Bounds of the receive-side memory: messages remembered for re-delivery (each entry pins one stored message)
  // and fingerprints of rejected ones.  When `seen_cap` is reached the table is dropped (re-deliveries are judged again).
  size_t seen_cap = 1u << 18, rejected_cap = 1u << 14;
  size_t seen_entries() const { return seen_.size(); }
  HotPath();
  // Certificate checks (§8f rank 2), restated from core/ibft.go: validPC :1162-1231,
  // proposalMatchesCertificate :516-551, validateProposalCommon :629-655, validateProposal0
  // :658-680, validateProposal :683-788.  With use_batch, every IsValidValidator /
  // IsValidProposalHash the walk can reach is answered from ONE device batch gathered up front
  // (the predicates are pure, so evaluating verdicts the short-circuit would have skipped cannot
  // change the result).
  bool validPC(const PreparedCertificate *certificate, uint64_t roundLimit, uint64_t height);  // batches on its own when called directly
  bool proposalMatchesCertificate(const Proposal *proposal, const PreparedCertificate *certificate);
  bool validateProposalCommon(const IbftMessage &msg, const View &view);
  bool validateProposal0(const IbftMessage &msg, const View &view);
  bool validateProposal(const IbftMessage &msg, const View &view);
  // number of sender signatures / hashes the last batched certificate check sent to the device
  size_t last_cert_senders = 0, last_cert_hashes = 0;

 private:
  // verdict tables filled by the batch pre-pass of a certificate walk (transient: cleared when the walk returns)
  std::unordered_map<const IbftMessage *, bool> sender_verdict_;
  struct PairHash {
    size_t operator()(const std::pair<const Proposal *, const bytes *> &k) const {
      return std::hash<const void *>()(k.first) * 1000003u ^ std::hash<const void *>()(k.second);
    }
  };
  std::unordered_map<std::pair<const Proposal *, const bytes *>, bool, PairHash> hash_verdict_;  // (proposal, hash) by identity
  bool isValidValidatorCached(const IbftMessage &m);
  // m: the message that carries `hash` (its arrival-time verdict, if any, is in m.verdicts)
  bool lookupHashVerdict(const IbftMessage *m, const Proposal *proposal, const bytes *hash, bool &ok) const;
  bool isValidProposalHashCached(const IbftMessage &m, const Proposal *proposal, const bytes *hash);
  void prefetchCertificateHashes(const std::vector<MsgPtr> &rcs);
  bool index_enabled_ = false;
  // Arrival-time verdicts live IN the message objects (proto.hpp: Verdicts) and are valid for these epochs
  uint32_t valset_epoch_ = 1;   // bumped by NotifyValidatorSetChanged
  uint32_t closure_epoch_ = 1;  // bumped when the proposal the closure verdicts refer to changes
  bytes closure_key_;           // raw proposal ‖ BE64(round) the closure verdicts refer to
  void syncClosureKey(const Proposal *proposal);
  bool senderKnown(const IbftMessage &m) const { return m.verdicts.sender_epoch == valset_epoch_; }
  bool closureKnown(const IbftMessage &m) const { return m.verdicts.closure_epoch == closure_epoch_; }
  void noteSender(const IbftMessage &m, bool ok) const {
    m.verdicts.sender = ok;
    m.verdicts.sender_epoch = valset_epoch_;
  }
  void noteClosure(const IbftMessage &m, bool ok) const {
    m.verdicts.closure = ok;
    m.verdicts.closure_epoch = closure_epoch_;
  }
  // the handle* walks: arrival-time verdicts first, one batch call for the rest, per-message closure when that fails
  std::vector<uint8_t> closureVerdicts(const Proposal *proposal, MessageType type, const std::vector<MsgPtr> &all);
  // Messages seen before, found by a seeded 128-bit fingerprint of their wire bytes.  The fingerprint is a FILTER, not an
  // identity — multiply-fold mixing has seed-independent collisions (a 16-byte block that zeroes a multiplicand, round-3
  // advice), so nothing is decided on it alone: a byte-identical re-delivery (gossip) of a STORED message is answered
  // with the stored object after a byte-for-byte compare; a re-delivery of a REJECTED one is recognised by the Keccak-256 of
  // its bytes (computed when it was rejected, and again only for a message whose fingerprint hits), from a bounded FIFO.  A
  // message that merely collides with a rejected forgery is judged like any other.
  struct Seen {
    uint64_t fp2;
    MsgPtr msg;           // the stored message (keeps its bytes: wire / len point into its backing)
    const uint8_t *wire;
    uint32_t len;
    uint64_t height;
  };
  // (a message stored as a ROW needs no entry: the store finds its sender's row and compares the bytes)
  // IBFT.AddMessage for a message kept as a row (sender and view already accepted): store + quorum probe
  int addLeanRow(uint32_t type, uint64_t h, uint64_t r, const LeanRow &row, const std::shared_ptr<const void> &backing);
  void addLeanRun(uint32_t type, const std::vector<const LeanRow *> &rows, const std::vector<size_t> &at,
                  const std::shared_ptr<const void> &backing, int8_t *results);
  int quorumProbe(uint32_t type, const View &view);  // the 1 / 2 of AddMessageFast, from the quorum index
  bool handleLean(const View &view, MessageType type, bool &quorum);  // true = the view was held as rows and is handled
  std::unordered_map<uint64_t, Seen> seen_;
  bool seen_has_votes_ = false;  // seen_ holds (or held, since it was last empty) PREPARE / COMMIT objects
  struct Rejected {
    uint64_t fp2;
    uint8_t digest[32];  // keccak256 of the rejected bytes
  };
  std::unordered_map<uint64_t, Rejected> seen_rejected_;  // fp1 → the rejected message's identity
  std::vector<uint64_t> rejected_fifo_;
  size_t rejected_head_ = 0;
  uint64_t fp_seed_;
  void noteCertificateTree(const CertVerdicts &cv, size_t row, const MsgPtr &root, bool note);
  // isValidMsgFn of handleRoundChangeMessage for the ROUND_CHANGE message at `row`, from the rows alone (no nested message
  // is decoded): 1 / 0 = the verdict, −1 = not decided here (an irregular shape: the object walk decides)
  int roundChangeVerdictFromRows(const CertVerdicts &cv, size_t row);
  int pcVerdictFromRows(const CertVerdicts &cv, size_t row, uint64_t limit, uint64_t height, bool match_proposal);
  bool proposalVerdictFromRows(const CertVerdicts &cv, size_t row, ProposalVerdict &out);
  double forged_carriers_ = 0.0;  // carriers whose own envelope failed, halved every batch
  std::vector<uint64_t> rc_set_;  // scratch: the sender set of one certificate
  bool validPCImpl(const PreparedCertificate *certificate, uint64_t roundLimit, uint64_t height);
  void prefetchSenders(const std::vector<const IbftMessage *> &msgs);

 public:
};

}  // namespace ibft
