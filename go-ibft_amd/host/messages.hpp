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
// returned messages; this mirror iterates in the senders' arrival order (deterministic).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <string_view>
#include <unordered_map>

#include "proto.hpp"

namespace ibft {

using Predicate = std::function<bool(const IbftMessage &)>;

// sender → message of one (type, height, round): a flat hash index over a vector of messages.  The key of an entry is the
// From of the message it holds (no copy, no tree node per sender); iteration is in ARRIVAL order of the senders
// (deterministic; Go's map order is random, callers may only rely on the set).
class SenderMap {
 public:
  size_t size() const { return live_; }
  bool contains(const bytes &from) const { return locate(from) != npos; }
  // insert, or replace the message of the same sender (last writer wins); true = a new sender
  bool put(MsgPtr m) {
    const size_t at = locate(m->from);
    if (at != npos) {
      entries_[at] = std::move(m);
      return false;
    }
    if ((entries_.size() + 1) * 2 > index_.size()) rebuild(std::max<size_t>(64, (entries_.size() + 1) * 4));
    entries_.push_back(std::move(m));
    link(entries_.size() - 1);
    live_++;
    return true;
  }
  bool erase(const bytes &from) {  // (the slot stays in the index: locate() steps over it, the next compaction drops it)
    const size_t at = locate(from);
    if (at == npos) return false;
    entries_[at].reset();
    live_--;
    return true;
  }
  // visit the live entries in arrival order; f(const MsgPtr &) → false erases the entry
  template <class F>
  void filter(F &&f) {
    bool erased = false;
    for (size_t i = 0; i < entries_.size(); i++) {
      if (!entries_[i]) continue;
      if (!f(entries_[i])) {
        entries_[i].reset();
        live_--;
        erased = true;
      }
    }
    if (erased) compact();
  }
  template <class F>
  void for_each(F &&f) const {
    for (const MsgPtr &m : entries_)
      if (m) f(m);
  }

 private:
  static constexpr size_t npos = (size_t)-1;
  static size_t hash_of(const bytes &b) { return bytes_hash()(b); }
  size_t locate(const bytes &from) const {
    if (index_.empty()) return npos;
    const size_t mask = index_.size() - 1;
    for (size_t s = hash_of(from) & mask;; s = (s + 1) & mask) {
      const uint32_t e = index_[s];
      if (e == 0) return npos;
      if (entries_[e - 1] && entries_[e - 1]->from == from) return e - 1;
    }
  }
  void link(size_t i) {
    const size_t mask = index_.size() - 1;
    size_t s = hash_of(entries_[i]->from) & mask;
    while (index_[s] != 0) s = (s + 1) & mask;
    index_[s] = (uint32_t)(i + 1);
  }
  void rebuild(size_t slots) {
    size_t n = 64;
    while (n < slots) n <<= 1;
    index_.assign(n, 0);
    for (size_t i = 0; i < entries_.size(); i++)
      if (entries_[i]) link(i);
  }
  void compact() {  // drop the erased entries (arrival order of the rest is kept) and re-index
    size_t k = 0;
    for (size_t i = 0; i < entries_.size(); i++)
      if (entries_[i]) {
        if (k != i) entries_[k] = std::move(entries_[i]);
        k++;
      }
    entries_.resize(k);
    rebuild(std::max<size_t>(64, k * 4));
  }
  std::vector<MsgPtr> entries_;
  std::vector<uint32_t> index_;  // open addressing: entry number + 1, 0 = empty
  size_t live_ = 0;
};

// A PREPARE / COMMIT message kept as a ROW instead of an object: where its bytes lie and where the few fields the hot path
// reads are inside them.  What a batch backend that takes the transport's bytes has judged completely (canonical encoding,
// envelope signature, closure) needs no pointer graph until somebody asks for one — a certificate being built, a test, a
// walk by the per-message path — and is decoded ("materialised") then, with its verdicts noted in the object.
struct LeanRow {
  const uint8_t *wire = nullptr;  // the message's bytes, inside buffers[buf] of its view
  uint32_t len = 0, buf = 0;
  uint32_t from_off = 0, from_len = 0, hash_off = 0, hash_len = 0, seal_off = 0, seal_len = 0;
  uint8_t closure = 0;            // handlePrepare's / handleCommit's closure verdict (against the view's closure epoch)
  uint8_t dead = 0;               // (the store's: pruned)
  uint64_t sender_hash = 0;       // hash_key(from()); 0 = not computed yet
  std::string_view from() const { return std::string_view((const char *)wire + from_off, from_len); }
};
class LeanView {
 public:
  uint32_t closure_epoch = 0, valset_epoch = 0;  // what the rows' verdicts were computed against
  // buffers[b] keeps the bytes of the rows with buf == b alive; it is let go (null) as soon as no LIVE row points into it —
  // a batch buffer also holds other senders' rejected bytes, and a sender that keeps replacing its message (canonical /
  // re-encoded variants, round-3 advice) must not pin one whole batch per replacement until the height is pruned
  std::vector<std::shared_ptr<const void>> buffers;
  size_t size() const { return live_; }
  size_t slots() const { return rows_.size(); }           // rows held, dead ones included (bounded: ≤ 2·live + 32)
  size_t buffers_held() const {
    size_t k = 0;
    for (const auto &b : buffers) k += b != nullptr;
    return k;
  }
  bool contains(std::string_view from) const { return locate(from) != npos; }
  bool erase(std::string_view from) {
    const size_t at = locate(from);
    if (at == npos) return false;
    row_died(at);
    return true;
  }
  const LeanRow *find(std::string_view from) const { return find(from, hash_key(from.data(), from.size())); }
  const LeanRow *find(std::string_view from, uint64_t hash) const {
    const size_t at = locate(from, hash);
    return at == npos ? nullptr : &rows_[at];
  }
  // insert, or replace the row of the same sender; true = a new sender.  (Pointers into the view — find() — do not survive
  // a put: rows may move.)
  bool put(LeanRow row, const std::shared_ptr<const void> &backing) {
    const std::string_view from = row.from();
    if (row.sender_hash == 0) row.sender_hash = hash_key(from.data(), from.size());
    const uint64_t h = row.sender_hash;
    row.dead = 0;
    if ((dead_ > live_ && dead_ >= 32) || null_bufs_ >= 64) compact();  // (before any index is taken: compaction renumbers)
    const size_t at = locate(from, h);
    row.buf = buffer_index(backing);
    buf_live_[row.buf]++;
    if (at != npos) {
      const uint32_t old = rows_[at].buf;
      rows_[at] = row;
      unref(old);
      return false;
    }
    if ((rows_.size() + 1) * 2 > index_.size()) rebuild(std::max<size_t>(64, (rows_.size() + 1) * 4));
    rows_.push_back(row);
    link(rows_.size() - 1, h);
    live_++;
    return true;
  }
  // The rows that point into `backing` get a buffer of their own (their bytes, back to back) and `backing` is let go: for a
  // batch of which little was stored — the rest of its buffer is other people's rejected bytes.  Returns the bytes copied.
  size_t repack(const std::shared_ptr<const void> &backing) {
    size_t total = 0;
    for (size_t idx = 0; idx < buffers.size(); idx++) {
      if (!buffers[idx] || buffers[idx] != backing) continue;
      size_t bytes = 0;
      for (const LeanRow &r : rows_)
        if (!r.dead && r.buf == idx) bytes += r.len;
      uint8_t *mem = static_cast<uint8_t *>(malloc(bytes ? bytes : 1));
      size_t at = 0;
      for (LeanRow &r : rows_)
        if (!r.dead && r.buf == idx) {
          memcpy(mem + at, r.wire, r.len);
          r.wire = mem + at;
          at += r.len;
        }
      buffers[idx] = std::shared_ptr<const void>(mem, free);
      total += bytes;
    }
    return total;
  }
  template <class F>
  void filter(F &&f) {  // f(const LeanRow &) → false erases the row
    for (size_t i = 0; i < rows_.size(); i++)
      if (!rows_[i].dead && !f(rows_[i])) row_died(i);
  }
  template <class F>
  void for_each(F &&f) const {
    for (size_t i = 0; i < rows_.size(); i++)
      if (!rows_[i].dead) f(rows_[i]);
  }

 private:
  static constexpr size_t npos = (size_t)-1;
  size_t locate(std::string_view from) const { return locate(from, hash_key(from.data(), from.size())); }
  // index entry: the key's hash (upper half) next to the row number + 1 (lower half) — a probe that lands on another
  // sender's slot is told apart without touching that sender's bytes.  A dead row's bytes are never read (its buffer may
  // be gone).
  size_t locate(std::string_view from, uint64_t h) const {
    if (index_.empty()) return npos;
    const size_t mask = index_.size() - 1;
    const uint64_t tag = h & 0xFFFFFFFF00000000ull;
    for (size_t s = h & mask;; s = (s + 1) & mask) {
      const uint64_t e = index_[s];
      if (e == 0) return npos;
      if ((e & 0xFFFFFFFF00000000ull) != tag) continue;
      const size_t at = (size_t)(e & 0xFFFFFFFFull) - 1;
      if (!rows_[at].dead && rows_[at].from() == from) return at;
    }
  }
  void link(size_t i, uint64_t h) {
    const size_t mask = index_.size() - 1;
    size_t s = h & mask;
    while (index_[s] != 0) s = (s + 1) & mask;
    index_[s] = (h & 0xFFFFFFFF00000000ull) | (uint64_t)(i + 1);
  }
  void rebuild(size_t slots) {
    size_t n = 64;
    while (n < slots) n <<= 1;
    index_.assign(n, 0);
    for (size_t i = 0; i < rows_.size(); i++)
      if (!rows_[i].dead) link(i, rows_[i].sender_hash);
  }
  // a batch arrives as runs of rows over ONE backing: the last buffer first, then the few others (a re-delivery from an
  // older batch must find its buffer again instead of pinning it a second time)
  uint32_t buffer_index(const std::shared_ptr<const void> &backing) {
    for (size_t b = buffers.size(); b-- > 0;)
      if (buffers[b] == backing) return (uint32_t)b;
    buffers.push_back(backing);
    buf_live_.push_back(0);
    return (uint32_t)buffers.size() - 1;
  }
  void unref(uint32_t b) {
    if (--buf_live_[b] == 0) {
      buffers[b].reset();
      null_bufs_++;
    }
  }
  void row_died(size_t at) {
    rows_[at].dead = 1;
    rows_[at].wire = nullptr;
    live_--;
    dead_++;
    unref(rows_[at].buf);
  }
  // dead rows outnumber the live ones: drop them and the buffers nobody points into, renumber, re-index
  void compact() {
    std::vector<uint32_t> remap(buffers.size(), 0);
    size_t nb = 0;
    for (size_t b = 0; b < buffers.size(); b++)
      if (buffers[b]) {
        remap[b] = (uint32_t)nb;
        if (nb != b) {
          buffers[nb] = std::move(buffers[b]);
          buf_live_[nb] = buf_live_[b];
        }
        nb++;
      }
    buffers.resize(nb);
    buf_live_.resize(nb);
    size_t k = 0;
    for (size_t i = 0; i < rows_.size(); i++)
      if (!rows_[i].dead) {
        rows_[i].buf = remap[rows_[i].buf];
        if (k != i) rows_[k] = rows_[i];
        k++;
      }
    rows_.resize(k);
    dead_ = 0;
    null_bufs_ = 0;
    rebuild(std::max<size_t>(64, (k + 1) * 4));
  }
  std::vector<LeanRow> rows_;
  std::vector<uint64_t> index_;
  std::vector<uint32_t> buf_live_;  // live rows per buffer
  size_t live_ = 0, dead_ = 0, null_bufs_ = 0;
};

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
  // objects_only: the view's ROWS are left alone (not decoded, not shown to `verdicts`) — the walk over a view that is
  // held partly as rows judges the rows by their noted verdicts (FilterLean) and only its objects here
  std::vector<MsgPtr> GetValidMessagesBatch(const View &view, MessageType type, const BatchPredicate &verdicts,
                                            bool objects_only = false);
  // prepass (optional): called ONCE, under the same lock, with every ROUND-CHANGE message stored for `height`
  // before the walk — lets a batch backend answer all the nested signature / hash questions of the walk with one
  // device call; the walk itself and what it returns are untouched.
  std::vector<MsgPtr> GetExtendedRCC(uint64_t height, const Predicate &isValidMessage,
                                     const std::function<bool(uint64_t, const std::vector<MsgPtr> &)> &isValidRCC,
                                     const std::function<void(const std::vector<MsgPtr> &)> &prepass = nullptr);
  std::vector<MsgPtr> GetMostRoundChangeMessages(uint64_t minRound, uint64_t height);
  // ---- rows (LeanRow above).  A (type, height, round) may hold rows AND objects, a sender in at most one of them (the
  // last writer wins, as in the reference's map): storing an object drops its sender's row and vice versa, and the sender
  // hook fires only when the sender is new to the VIEW.  AddLean refuses (false) when the view's rows were judged against
  // other epochs; the accessors that hand out objects turn the view's rows into objects first.
  bool AddLean(uint32_t type, uint64_t height, uint64_t round, const LeanRow &row, const std::shared_ptr<const void> &backing,
               uint32_t closure_epoch, uint32_t valset_epoch);
  // A run of rows of ONE view under one lock.  after(k, new_sender, view_rows) is called for every row once it is stored
  // — new_sender: the sender had nothing in the view, neither row nor object (the sender hook is NOT called: the caller
  // keeps the counters of a run itself).  Returns the rows taken: 0 when the view's rows were judged against other epochs
  // (they become objects; the caller stores these messages one by one).
  size_t AddLeanRun(uint32_t type, uint64_t height, uint64_t round, const LeanRow *const *rows, size_t n,
                    const std::shared_ptr<const void> &backing, uint32_t closure_epoch, uint32_t valset_epoch,
                    const std::function<void(size_t, bool, const LeanView &, const SenderMap *)> &after);
  // the view's rows when it is held as rows AND they were judged against these epochs; otherwise the rows (if any) are
  // materialised and nullptr is returned.  The pointer is valid until the next call that touches the view.
  LeanView *LeanFor(const View &view, MessageType type, uint32_t closure_epoch, uint32_t valset_epoch);
  // prune the rows f rejects (hooks fire, as for GetValidMessages); returns the survivors' count
  size_t FilterLean(const View &view, MessageType type, const std::function<bool(const LeanRow &)> &f);
  // a batch buffer of which little was stored: the view's rows that point into it move into a buffer of their own
  size_t RepackLean(const View &view, MessageType type, const std::shared_ptr<const void> &backing);
  void MaterializeAll();  // every view held as rows becomes objects (validator set changed: the rows' verdicts are void)
  std::vector<bytes> SendersOf(const View &view, MessageType type);  // distinct senders of a view, rows or objects
  // what a view's rows hold: live rows, row slots (dead ones included), batch buffers still referenced
  void LeanStats(const View &view, MessageType type, size_t *live, size_t *slots, size_t *buffers);

 private:
  using protoMessages = SenderMap;                           // sender -> message
  using roundMessageMap = std::map<uint64_t, protoMessages>;  // round -> ...
  using heightMessageMap = std::map<uint64_t, roundMessageMap>;
  heightMessageMap maps_[4];
  struct LastView {
    protoMessages *msgs = nullptr;
    uint64_t height = 0, round = 0;
  };
  LastView last_[4];  // where the senders of the view of the last AddMessage are (per type)
  std::map<std::pair<uint64_t, uint64_t>, LeanView> lean_[4];  // (height, round) → rows; only PREPARE / COMMIT are ever used
  void materialize_locked(int s, uint64_t height, uint64_t round);  // mux_[s] held
  SenderMap *objects_of(int s, uint64_t height, uint64_t round);     // mux_[s] held; nullptr = none
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
  uint64_t powerOf(const bytes &from) const { return powerOf(std::string_view(from.data(), from.size())); }
  uint64_t powerOf(std::string_view from) const { return powerOf(from, hash_key(from.data(), from.size())); }
  uint64_t powerOf(std::string_view from, uint64_t hash) const {  // hash = hash_key(from)
    const int64_t at = seat(from, hash);
    return at < 0 ? 0 : seat_power_[(size_t)at];
  }
  bool isValidator(const bytes &from) const { return seat(std::string_view(from.data(), from.size())) >= 0; }

 private:
  std::map<bytes, uint64_t> power_;
  // open addressing over the addresses: slot = hash (upper half) | seat + 1 (lower half), 0 = empty — one probe and,
  // on a matching hash, one 20-byte compare per lookup
  std::vector<uint64_t> seat_slot_;
  std::vector<bytes> seat_addr_;
  std::vector<uint64_t> seat_power_;
  int64_t seat(std::string_view from) const { return seat(from, hash_key(from.data(), from.size())); }
  int64_t seat(std::string_view from, uint64_t h) const {
    if (seat_slot_.empty()) return -1;
    const size_t mask = seat_slot_.size() - 1;
    const uint64_t tag = h & 0xFFFFFFFF00000000ull;
    for (size_t s = h & mask;; s = (s + 1) & mask) {
      const uint64_t e = seat_slot_[s];
      if (e == 0) return -1;
      if ((e & 0xFFFFFFFF00000000ull) != tag) continue;
      const bytes &a = seat_addr_[(size_t)(e & 0xFFFFFFFFull) - 1];
      if (a.size() == from.size() && memcmp(a.data(), from.data(), from.size()) == 0) return (int64_t)(e & 0xFFFFFFFFull) - 1;
    }
  }
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
  // the counters of a view moved by (Δpower, Δsenders) — a run of rows stored without the per-sender hook
  void Add(uint32_t type, uint64_t height, uint64_t round, unsigned __int128 dpower, size_t dcount);
  void OnPrune(uint64_t below_height);
  void Invalidate() {  // validator set changed: sums are recomputed on next use
    std::lock_guard<std::mutex> lk(mu_);
    epoch_++;
  }
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
  uint64_t changes_ = 0;  // every hook call: a Get that listed the view with the lock released sees whether it must list again
  std::mutex mu_;
};

}  // namespace ibft
