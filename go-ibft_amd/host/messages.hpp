// messages.hpp — C++ mirror of messages.Messages (the store) and ValidatorManager.
//
// Product host code.  Semantics follow /root/reference/messages/messages.go:54-65
// (AddMessage: last writer wins per sender), :96-119 (numMessages), :123-148
// (PruneByHeight), :169-199 (GetValidMessages: invalid messages are DELETED), :202-245
// (GetExtendedRCC: non-pruning; round 0 can never be returned because of the
// `round <= highestRound` test), :249-286 (GetMostRoundChangeMessages) and
// /root/reference/core/validator_manager.go:50-155.
//
// Iteration order: Go map iteration is random, so callers may only rely on the SET of
// returned messages; this mirror iterates in sender-byte order (deterministic).
#pragma once
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <unordered_map>

#include "proto.hpp"

namespace ibft {

using Predicate = std::function<bool(const IbftMessage &)>;

class Messages {
 public:
  // Observer of the sender SET of a view: called with +1 when a sender appears in
  // [type][height][round] (not when it overwrites its own message) and −1 when one is removed by
  // the prune-on-invalid of GetValidMessages.  PruneByHeight reports whole heights instead.
  // Lets a caller keep Σ power per view incrementally (QuorumIndex below) instead of re-walking
  // the view on every AddMessage (/root/reference/core/ibft.go:1113-1120).
  using SenderHook = std::function<void(uint32_t type, uint64_t height, uint64_t round, const bytes &from, int delta)>;
  using HeightHook = std::function<void(uint64_t below_height)>;
  void SetHooks(SenderHook s, HeightHook h) {
    sender_hook_ = std::move(s);
    height_hook_ = std::move(h);
  }
  bool Has(const View &view, MessageType type, const bytes &from);
  void AddMessage(MsgPtr m);
  size_t numMessages(const View &view, MessageType type);
  void PruneByHeight(uint64_t height);
  std::vector<MsgPtr> GetValidMessages(const View &view, MessageType type, const Predicate &isValid);
  // Batched form used by the GPU backend: `verdicts(msgs)` is called ONCE with every stored
  // message of the view (under the same per-type lock the reference holds across its
  // callback loop) and returns one verdict per message; rejected ones are deleted.
  using BatchPredicate = std::function<std::vector<uint8_t>(const std::vector<MsgPtr> &)>;
  std::vector<MsgPtr> GetValidMessagesBatch(const View &view, MessageType type, const BatchPredicate &verdicts);
  // prepass (optional): called ONCE, under the same lock, with every ROUND-CHANGE message stored for `height`
  // before the walk — lets a batch backend answer all the nested signature / hash questions of the walk with one
  // device call; the walk itself and what it returns are untouched.
  std::vector<MsgPtr> GetExtendedRCC(uint64_t height, const Predicate &isValidMessage,
                                     const std::function<bool(uint64_t, const std::vector<MsgPtr> &)> &isValidRCC,
                                     const std::function<void(const std::vector<MsgPtr> &)> &prepass = nullptr);
  std::vector<MsgPtr> GetMostRoundChangeMessages(uint64_t minRound, uint64_t height);

 private:
  using protoMessages = std::map<bytes, MsgPtr>;             // sender -> message
  using roundMessageMap = std::map<uint64_t, protoMessages>;  // round -> ...
  using heightMessageMap = std::map<uint64_t, roundMessageMap>;
  heightMessageMap maps_[4];
  struct LastView {
    protoMessages *msgs = nullptr;
    uint64_t height = 0, round = 0;
  };
  LastView last_[4];  // where the senders of the view of the last AddMessage are (per type)
  std::shared_mutex mux_[4];
  SenderHook sender_hook_;
  HeightHook height_hook_;
  static int slot(uint32_t type) { return type <= 3 ? (int)type : -1; }
};

// core/validator_manager.go.  Powers are u64 here (128-bit sums); the reference uses
// *big.Int — sets that need more must stay on the Go path (include/ibftgpu.h).
class ValidatorManager {
 public:
  // setCurrentVotingPower (:61-75); false = errVotingPowerNotCorrect
  bool Init(const std::vector<std::pair<bytes, uint64_t>> &powers);
  bool HasQuorum(const std::set<bytes> &senders) const;                                  // :77-96
  // HasQuorum(convertMessageToAddressSet(msgs) ∪ {extra}) without materialising the set
  bool HasQuorumOf(const std::vector<MsgPtr> &msgs, const bytes *extra = nullptr) const;
  // :99-127.  proposal == nullptr -> false
  bool HasPrepareQuorum(const IbftMessage *proposal, const std::vector<MsgPtr> &msgs) const;
  unsigned __int128 quorum() const { return quorum_; }
  bool initialized() const { return initialized_; }
  const std::map<bytes, uint64_t> &powers() const { return power_; }
  // O(1): voting power of `from` (0 for a non-member) and membership
  uint64_t powerOf(const bytes &from) const {
    auto it = fast_.find(from);
    return it == fast_.end() ? 0 : it->second;
  }
  bool isValidator(const bytes &from) const { return fast_.find(from) != fast_.end(); }

 private:
  std::map<bytes, uint64_t> power_;
  std::unordered_map<bytes, uint64_t> fast_;
  unsigned __int128 quorum_ = 0;
  bool initialized_ = false;
};

std::set<bytes> convertMessageToAddressSet(const std::vector<MsgPtr> &msgs);  // :147-155

// Σ voting power of the distinct senders stored per (type, height, round), maintained from the
// store's hooks: the O(1) replacement of the per-message GetValidMessages + HasQuorum probe of
// IBFT.AddMessage.  Rebuilt lazily when the validator set changes.
class QuorumIndex {
 public:
  void OnSender(uint32_t type, uint64_t height, uint64_t round, const bytes &from, int delta,
                const ValidatorManager &vm);
  void OnPrune(uint64_t below_height);
  void Invalidate() { epoch_++; }  // validator set changed: sums are recomputed on next use
  // (Σ power, number of stored senders); `rebuild` lists the view's senders when the entry is stale
  std::pair<unsigned __int128, size_t> Get(uint32_t type, uint64_t height, uint64_t round,
                                           const std::function<std::vector<bytes>()> &rebuild,
                                           const ValidatorManager &vm);

 private:
  struct Entry {
    unsigned __int128 power = 0;
    size_t count = 0;
    uint64_t epoch = 0;
    bool valid = false;
  };
  std::map<std::tuple<uint32_t, uint64_t, uint64_t>, Entry> e_;
  std::tuple<uint32_t, uint64_t, uint64_t> last_key_{};
  Entry *last_ = nullptr;  // entry of the last view asked about (std::map nodes do not move)
  Entry *find(uint32_t type, uint64_t height, uint64_t round) {
    const std::tuple<uint32_t, uint64_t, uint64_t> k{type, height, round};
    if (last_ && last_key_ == k) return last_;
    auto it = e_.find(k);
    if (it == e_.end()) return nullptr;
    last_key_ = k;
    last_ = &it->second;
    return last_;
  }
  uint64_t epoch_ = 1;
  std::mutex mu_;
};

}  // namespace ibft
