// proto.hpp — C++ mirror of go-ibft's wire types (messages/proto/messages.proto).
//
// Product host code.  Field numbers and zero/empty omission follow
// /root/reference/messages/proto/messages.proto:7-110 (proto3); PayloadNoSig follows
// /root/reference/messages/proto/helper.go:12-27 (marshal of the message with Signature
// cleared).  The encoder emits what protobuf-go emits for these messages: known fields in
// field-number order, minimal varints, zero scalars / empty bytes omitted, present
// sub-messages emitted even when empty, unknown fields preserved and appended last.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "bytes.hpp"

namespace ibft {

enum MessageType : uint32_t { PREPREPARE = 0, PREPARE = 1, COMMIT = 2, ROUND_CHANGE = 3 };

struct View {
  uint64_t height = 0, round = 0;
  bytes unknown;
};

struct Proposal {
  bytes raw_proposal;
  uint64_t round = 0;
  bytes unknown;
};

struct IbftMessage;
using MsgPtr = std::shared_ptr<IbftMessage>;

struct PreparedCertificate {
  MsgPtr proposal_message;               // nil-able
  std::vector<MsgPtr> prepare_messages;  // repeated
  bytes unknown;
};

struct RoundChangeCertificate {
  std::vector<MsgPtr> round_change_messages;
  bytes unknown;
};

struct PrePrepareMessage {
  std::optional<Proposal> proposal;
  bytes proposal_hash;
  std::optional<RoundChangeCertificate> certificate;
  bytes unknown;
  // a certificate that has not been decoded yet — see RoundChangeMessage below
  mutable bool certificate_deferred = false;
  mutable bytes certificate_wire;
  mutable std::shared_ptr<const void> certificate_backing;
  bool realise_certificate() const;
};
struct PrepareMessage {
  bytes proposal_hash;
  bytes unknown;
};
struct CommitMessage {
  bytes proposal_hash;
  bytes committed_seal;
  bytes unknown;
};
struct RoundChangeMessage {
  std::optional<Proposal> last_prepared_proposal;
  std::optional<PreparedCertificate> latest_prepared_certificate;
  bytes unknown;
  // A certificate that has not been decoded yet (decode_in(..., defer_certificate)): its bytes, to be decoded by
  // realise_certificate() when somebody asks for the objects (extract_latest_pc, nested_messages) and copied as they are
  // by encode().  Only a certificate whose bytes are known to be well-formed AND canonical may stay in this state — the
  // receive path keeps it for messages a batch backend vouched for (class 0 of ibft_verify_certificates_wire) and decodes
  // every other one at once, dropping the message if that fails as proto.Unmarshal would have.
  mutable bool certificate_deferred = false;
  mutable bytes certificate_wire;                      // a view into `certificate_backing`
  mutable std::shared_ptr<const void> certificate_backing;
  bool realise_certificate() const;                    // false: the deferred bytes do not decode (the certificate stays absent)
};

// Which member of the `payload` oneof is set (messages.proto:36-43); NONE = nil Payload.
enum class PayloadKind { NONE, PREPREPARE, PREPARE, COMMIT, ROUND_CHANGE };

// What the hot path has learned about a message — NOT part of its wire state (never encoded, never compared).  The three
// Verifier predicates are pure functions of the message bytes, the proposal and the validator set, so a verdict computed
// when the message arrived can sit in the message object itself: it lives exactly as long as the message, needs no side
// table, and is ignored (epoch mismatch) once the validator set or the proposal it was computed against has changed.
// validateProposal (core/ibft.go:683-788) as far as it depends only on a PREPREPARE message's RoundChangeCertificate and on the
// validator set, decided from a batch backend's rows when the message arrived (HotPath::proposalVerdictFromRows)
struct ProposalVerdict {
  bool ok = false;            // unique senders, quorum, every ROUND_CHANGE message of the proposal's view and validly signed
  bool has_prepared = false;  // at least one valid PreparedCertificate: (max_round, hash) = the highest prepared round and its hash
  uint64_t max_round = 0;
  uint8_t hash[32] = {0};
  uint32_t rows = 0;          // the nested messages the verdict covers
};

struct Verdicts {
  uint32_t sender_epoch = 0;   // validator-set epoch of `sender` (0 = unknown)
  uint32_t closure_epoch = 0;  // proposal epoch of `closure` (0 = unknown)
  const void *hash_of = nullptr;   // the Proposal object `hash` was judged against (a nested message of a PreparedCertificate:
                                   // its carrier's lastPreparedProposal — proposalMatchesCertificate, core/ibft.go:516-551)
  const void *self_of = nullptr;   // a PREPREPARE's own Proposal object `self` was judged against (validateProposalCommon)
  uint8_t sender = 0, closure = 0, hash = 0, self = 0;
  // a ROUND_CHANGE message: validPC(latestPC, view.round, view.height) ∧ proposalMatchesCertificate(lastPreparedProposal,
  // latestPC) — handleRoundChangeMessage's isValidMsgFn (core/ibft.go:478-489) — decided when the message arrived, from the
  // rows a batch backend returned for its certificate; rc_rows = the nested messages that verdict covers
  uint32_t rc_epoch = 0;  // validator-set epoch of `rc_ok` (0 = unknown)
  uint32_t rc_rows = 0;
  uint8_t rc_ok = 0;
  uint32_t pp_epoch = 0;  // validator-set epoch of the message's ProposalVerdict (IbftMessage::proposal_verdict; 0 = none)
};

// A member most messages never use (the PREPREPARE / ROUND_CHANGE payloads, ≈ 500 bytes of the message object between them):
// allocated on the first write, read as an empty value until then.  The object, once allocated, never moves (pointers into
// it — extract_proposal, extract_latest_pc — stay valid for the life of the message).
template <class T>
class Lazy {
 public:
  Lazy() = default;
  Lazy(const Lazy &o) : p_(o.p_ ? new T(*o.p_) : nullptr) {}
  Lazy(Lazy &&) noexcept = default;
  Lazy &operator=(const Lazy &o) {
    if (this != &o) p_.reset(o.p_ ? new T(*o.p_) : nullptr);
    return *this;
  }
  Lazy &operator=(Lazy &&) noexcept = default;
  const T &get() const {
    static const T empty{};
    return p_ ? *p_ : empty;
  }
  T &mut() {
    if (!p_) p_.reset(new T());
    return *p_;
  }
  void reset() { p_.reset(); }

 private:
  std::unique_ptr<T> p_;
};

struct IbftMessage {
  std::optional<View> view;  // nil-able pointer in Go
  bytes from;
  bytes signature;
  uint32_t type = PREPREPARE;
  PayloadKind kind = PayloadKind::NONE;
  PrepareMessage prepare;
  CommitMessage commit;
  const PrePrepareMessage &preprepare() const { return preprepare_.get(); }
  PrePrepareMessage &preprepare_mut() { return preprepare_.mut(); }
  const RoundChangeMessage &round_change() const { return round_change_.get(); }
  RoundChangeMessage &round_change_mut() { return round_change_.mut(); }
  Lazy<PrePrepareMessage> preprepare_;
  Lazy<RoundChangeMessage> round_change_;
  mutable Lazy<ProposalVerdict> proposal_verdict;  // (with verdicts.pp_epoch)
  bytes unknown;
  // the buffer the byte fields of a DECODED message (and of everything nested in it) point into; null for a message
  // that was built field by field
  std::shared_ptr<const void> backing;
  mutable Verdicts verdicts;
};

// ---- wire encoding ---------------------------------------------------------------------
bytes encode(const IbftMessage &m, bool with_signature = true);
inline bytes payload_no_sig(const IbftMessage &m) { return encode(m, false); }
bytes encode(const View &v);
bytes encode(const Proposal &p);
bytes encode(const PreparedCertificate &pc);
bytes encode(const RoundChangeCertificate &rcc);

// ---- wire decoding (what proto.Unmarshal does for these messages) -------------------------
// Returns false on malformed input (truncated varint/length, bad wire type for a known field).
bool decode(const uint8_t *p, size_t n, IbftMessage &out);
// The same without copying the wire: [p, p + n) lies inside `backing`, which the message (and every message nested in it)
// keeps alive; all byte fields are views into it.
// defer_certificate: the PreparedCertificate of a top-level ROUND_CHANGE payload / the RoundChangeCertificate of a top-level
// PREPREPARE payload is not decoded (RoundChangeMessage, PrePrepareMessage above).
bool decode_in(const std::shared_ptr<const void> &backing, const uint8_t *p, size_t n, IbftMessage &out,
               bool defer_certificate = false);
// What the receive side needs to know about a message BEFORE it is decoded (and without allocating anything): whether the
// top-level walk succeeds at all, the view, the type and which payload member is set — merged exactly as decode() merges
// repeated fields (the last View fields and the last payload member on the wire win).
struct Peek {
  bool ok = false, has_view = false;
  uint64_t height = 0, round = 0;
  uint32_t type = PREPREPARE;
  PayloadKind kind = PayloadKind::NONE;
  // where the byte fields of a PREPARE / COMMIT lie (offsets from the message's first byte; length 0 = absent), for a
  // message whose fields occur ONCE (what a canonical encoding guarantees): From, Signature, the payload's proposalHash
  // and committedSeal.  simple = false when a field repeats or the payload holds anything else (the row is then not kept
  // as a row: it is decoded).
  uint32_t from_off = 0, from_len = 0, sig_off = 0, sig_len = 0, hash_off = 0, hash_len = 0, seal_off = 0, seal_len = 0;
  bool simple = false;
};
Peek peek(const uint8_t *p, size_t n);
Peek peek_general(const uint8_t *p, size_t n);  // the field walk without the shortcut for the regular PREPARE / COMMIT shape
// a heap copy of [p, p + n) to decode into (one allocation)
std::shared_ptr<const void> make_backing(const uint8_t *p, size_t n, const uint8_t **copy);
bool decode(const uint8_t *p, size_t n, PreparedCertificate &out);
bool decode(const uint8_t *p, size_t n, Proposal &out);

// ---- messages/helpers.go Extract* ------------------------------------------------------------
// A null pointer return mirrors a nil []byte / nil pointer in the reference.
struct CommittedSeal {  // messages/helpers.go:15-19
  bytes signer, signature;
  MsgPtr keep;  // set by extract_committed_seals: signer / signature are views into this message …
  std::shared_ptr<const void> keep_buf;  // … or into this buffer (a seal read off a message that is kept as a row)
};
const bytes *extract_commit_hash(const IbftMessage &m);        // helpers.go:51-62
std::optional<CommittedSeal> extract_committed_seal(const IbftMessage &m);  // helpers.go:38-48
const bytes *extract_prepare_hash(const IbftMessage &m);       // helpers.go:107-118
const Proposal *extract_proposal(const IbftMessage &m);        // helpers.go:65-76
const bytes *extract_proposal_hash(const IbftMessage &m);      // helpers.go:79-90
const RoundChangeCertificate *extract_round_change_certificate(const IbftMessage &m);  // :93-104
const PreparedCertificate *extract_latest_pc(const IbftMessage &m);                    // :121-132
const Proposal *extract_last_prepared_proposal(const IbftMessage &m);                  // :135-146
// helpers.go:22-35: false = ErrWrongCommitMessageType
bool extract_committed_seals(const std::vector<MsgPtr> &msgs, std::vector<std::optional<CommittedSeal>> &out);
bool has_unique_senders(const std::vector<MsgPtr> &msgs);                                // :149-166
bool are_valid_pc_messages(const std::vector<MsgPtr> &msgs, uint64_t height, uint64_t round_limit);  // :169-213

}  // namespace ibft
