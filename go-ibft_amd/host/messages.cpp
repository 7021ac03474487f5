// messages.cpp — see messages.hpp for the reference lines each method follows.
#include "messages.hpp"

#include <string_view>
#include <unordered_set>

namespace ibft {

void Messages::AddMessage(MsgPtr m) {
  int s = slot(m->type);
  if (s < 0 || !m->view) return;  // the reference would panic on an unknown type / nil view
  std::unique_lock lk(mux_[s]);
  const uint64_t h = m->view->height, r = m->view->round;
  bool had_row = false;
  if (!lean_[s].empty()) {  // the sender's row, if any, is replaced by this object (last writer wins)
    auto lv = lean_[s].find({h, r});
    if (lv != lean_[s].end()) had_row = lv->second.erase(std::string_view(m->from.data(), m->from.size()));
  }
  // consecutive messages mostly belong to one view: remember where its senders are (std::map nodes do not move)
  LastView &lv = last_[s];
  if (!lv.msgs || lv.height != h || lv.round != r) {
    lv.msgs = &maps_[s][h][r];
    lv.height = h;
    lv.round = r;
  }
  protoMessages &view_msgs = *lv.msgs;
  const IbftMessage *raw = m.get();
  if (view_msgs.put(std::move(m)) && !had_row) {  // a new sender (otherwise: last writer wins, sender set unchanged)
    if (sender_hook_) sender_hook_((uint32_t)s, h, r, raw->from, +1);
  }
}

bool Messages::Has(const View &view, MessageType type, const bytes &from) {
  int s = slot(type);
  if (s < 0) return false;
  std::shared_lock lk(mux_[s]);
  auto lv = lean_[s].find({view.height, view.round});
  if (lv != lean_[s].end() && lv->second.contains(std::string_view(from.data(), from.size()))) return true;
  auto h = maps_[s].find(view.height);
  if (h == maps_[s].end()) return false;
  auto r = h->second.find(view.round);
  return r != h->second.end() && r->second.contains(from);
}

size_t Messages::numMessages(const View &view, MessageType type) {
  int s = slot(type);
  if (s < 0) return 0;
  std::shared_lock lk(mux_[s]);
  auto lv = lean_[s].find({view.height, view.round});
  const size_t rows = lv != lean_[s].end() ? lv->second.size() : 0;
  auto h = maps_[s].find(view.height);
  if (h == maps_[s].end()) return rows;
  auto r = h->second.find(view.round);
  return rows + (r == h->second.end() ? 0 : r->second.size());
}

void Messages::PruneByHeight(uint64_t height) {
  for (int s = 0; s < 4; s++) {
    std::unique_lock lk(mux_[s]);
    auto &m = maps_[s];
    m.erase(m.begin(), m.lower_bound(height));  // delete every msgHeight < height
    last_[s] = LastView{};
    lean_[s].erase(lean_[s].begin(), lean_[s].lower_bound({height, 0}));
  }
  if (height_hook_) height_hook_(height);
}

std::vector<MsgPtr> Messages::GetValidMessages(const View &view, MessageType type, const Predicate &isValid) {
  std::vector<MsgPtr> valid;
  int s = slot(type);
  if (s < 0) return valid;
  std::unique_lock lk(mux_[s]);  // write lock held across the callbacks, messages.go:174-176
  if (!lean_[s].empty()) materialize_locked(s, view.height, view.round);
  auto h = maps_[s].find(view.height);
  if (h == maps_[s].end()) return valid;
  auto r = h->second.find(view.round);
  if (r == h->second.end()) return valid;
  valid.reserve(r->second.size());
  r->second.filter([&](const MsgPtr &m) {
    if (!isValid(*m)) {
      if (sender_hook_) sender_hook_((uint32_t)s, view.height, view.round, m->from, -1);
      return false;  // prune out invalid messages, messages.go:193-196
    }
    valid.push_back(m);
    return true;
  });
  return valid;
}

std::vector<MsgPtr> Messages::GetValidMessagesBatch(const View &view, MessageType type,
                                                    const BatchPredicate &verdicts, bool objects_only) {
  std::vector<MsgPtr> valid;
  int s = slot(type);
  if (s < 0) return valid;
  std::unique_lock lk(mux_[s]);
  if (!objects_only && !lean_[s].empty()) materialize_locked(s, view.height, view.round);
  auto h = maps_[s].find(view.height);
  if (h == maps_[s].end()) return valid;
  auto r = h->second.find(view.round);
  if (r == h->second.end()) return valid;
  std::vector<MsgPtr> all;
  all.reserve(r->second.size());
  r->second.for_each([&](const MsgPtr &m) { all.push_back(m); });
  std::vector<uint8_t> v = verdicts(all);
  if (v.size() != all.size()) return valid;  // backend failure: nothing pruned, nothing returned
  bool any_bad = false;
  for (uint8_t x : v) any_bad = any_bad || !x;
  if (!any_bad) return all;  // (the usual case: every stored message survives)
  size_t i = 0;
  r->second.filter([&](const MsgPtr &m) {
    const bool ok = v[i++] != 0;
    if (!ok) {
      if (sender_hook_) sender_hook_((uint32_t)s, view.height, view.round, m->from, -1);
    } else {
      valid.push_back(m);
    }
    return ok;
  });
  return valid;
}

std::vector<MsgPtr> Messages::GetExtendedRCC(
    uint64_t height, const Predicate &isValidMessage,
    const std::function<bool(uint64_t, const std::vector<MsgPtr> &)> &isValidRCC,
    const std::function<void(const std::vector<MsgPtr> &)> &prepass) {
  std::unique_lock lk(mux_[ROUND_CHANGE]);
  std::vector<MsgPtr> extended;
  auto h = maps_[ROUND_CHANGE].find(height);
  if (h == maps_[ROUND_CHANGE].end()) return extended;
  if (prepass) {
    std::vector<MsgPtr> all;
    for (auto &rm : h->second)
      rm.second.for_each([&](const MsgPtr &m) { all.push_back(m); });
    prepass(all);
  }
  uint64_t highest = 0;
  for (auto &rm : h->second) {
    const uint64_t round = rm.first;
    if (round <= highest) continue;  // messages.go:222-224 (so round 0 is never considered)
    std::vector<MsgPtr> valid;
    rm.second.for_each([&](const MsgPtr &m) {
      if (isValidMessage(*m)) valid.push_back(m);
    });
    if (!isValidRCC(round, valid)) continue;
    highest = round;
    extended = std::move(valid);
  }
  return extended;
}

std::vector<MsgPtr> Messages::GetMostRoundChangeMessages(uint64_t minRound, uint64_t height) {
  std::shared_lock lk(mux_[ROUND_CHANGE]);
  std::vector<MsgPtr> out;
  auto h = maps_[ROUND_CHANGE].find(height);
  if (h == maps_[ROUND_CHANGE].end()) return out;
  uint64_t best = 0;
  size_t best_count = 0;
  for (auto &rm : h->second) {
    if (rm.first < minRound) continue;
    if (rm.second.size() > best_count) {  // strict: ties keep the first seen (map order is random in Go)
      best = rm.first;
      best_count = rm.second.size();
    }
  }
  if (best == 0) return out;  // "no messages found" — also when the best round IS 0, messages.go:273-276
  h->second[best].for_each([&](const MsgPtr &m) { out.push_back(m); });
  return out;
}

// ---- rows ---------------------------------------------------------------------------------------------------------
void Messages::materialize_locked(int s, uint64_t height, uint64_t round) {
  auto it = lean_[s].find({height, round});
  if (it == lean_[s].end()) return;
  LeanView lv = std::move(it->second);
  lean_[s].erase(it);
  protoMessages &view_msgs = maps_[s][height][round];
  last_[s] = LastView{};
  lv.for_each([&](const LeanRow &row) {
    auto m = std::make_shared<IbftMessage>();
    if (!decode_in(lv.buffers[row.buf], row.wire, row.len, *m)) return;  // (a row was vouched canonical: it decodes)
    // what was known about the row goes with the object
    m->verdicts.sender = 1;
    m->verdicts.sender_epoch = lv.valset_epoch;
    m->verdicts.closure = row.closure;
    m->verdicts.closure_epoch = lv.closure_epoch;
    view_msgs.put(std::move(m));  // (the sender hooks fired when the row was added: nothing changes for the counters)
  });
}

bool Messages::AddLean(uint32_t type, uint64_t height, uint64_t round, const LeanRow &row,
                       const std::shared_ptr<const void> &backing, uint32_t closure_epoch, uint32_t valset_epoch) {
  int s = slot(type);
  if (s != PREPARE && s != COMMIT) return false;
  std::unique_lock lk(mux_[s]);
  auto it = lean_[s].find({height, round});
  if (it == lean_[s].end()) {
    it = lean_[s].emplace(std::make_pair(height, round), LeanView{}).first;
    it->second.closure_epoch = closure_epoch;
    it->second.valset_epoch = valset_epoch;
  } else if (it->second.closure_epoch != closure_epoch || it->second.valset_epoch != valset_epoch) {
    materialize_locked(s, height, round);  // judged against another proposal / validator set: objects from here on
    return false;
  }
  SenderMap *objs = objects_of(s, height, round);
  const std::string_view f = row.from();
  const bool had_object = objs && objs->size() && objs->erase(bytes::view(f.data(), f.size()));
  if (it->second.put(row, backing) && !had_object && sender_hook_)
    sender_hook_((uint32_t)s, height, round, bytes::view(f.data(), f.size()), +1);
  return true;
}

SenderMap *Messages::objects_of(int s, uint64_t height, uint64_t round) {
  auto h = maps_[s].find(height);
  if (h == maps_[s].end()) return nullptr;
  auto r = h->second.find(round);
  return r == h->second.end() ? nullptr : &r->second;
}

size_t Messages::AddLeanRun(uint32_t type, uint64_t height, uint64_t round, const LeanRow *const *rows, size_t n,
                            const std::shared_ptr<const void> &backing, uint32_t closure_epoch, uint32_t valset_epoch,
                            const std::function<void(size_t, bool, const LeanView &, const SenderMap *)> &after) {
  int s = slot(type);
  if (s != PREPARE && s != COMMIT) return 0;
  std::unique_lock lk(mux_[s]);
  auto it = lean_[s].find({height, round});
  if (it == lean_[s].end()) {
    it = lean_[s].emplace(std::make_pair(height, round), LeanView{}).first;
    it->second.closure_epoch = closure_epoch;
    it->second.valset_epoch = valset_epoch;
  } else if (it->second.closure_epoch != closure_epoch || it->second.valset_epoch != valset_epoch) {
    materialize_locked(s, height, round);
    return 0;
  }
  LeanView &lv = it->second;
  SenderMap *objs = objects_of(s, height, round);
  if (objs && objs->size() == 0) objs = nullptr;
  for (size_t k = 0; k < n; k++) {
    bool had_object = false;
    if (objs) {
      const std::string_view f = rows[k]->from();
      had_object = objs->erase(bytes::view(f.data(), f.size()));
    }
    const bool fresh = lv.put(*rows[k], backing) && !had_object;
    after(k, fresh, lv, objs);  // (objs: the view's OBJECTS — a sender may be held as either)
  }
  return n;
}

LeanView *Messages::LeanFor(const View &view, MessageType type, uint32_t closure_epoch, uint32_t valset_epoch) {
  int s = slot(type);
  if (s < 0) return nullptr;
  std::unique_lock lk(mux_[s]);
  auto it = lean_[s].find({view.height, view.round});
  if (it == lean_[s].end()) return nullptr;
  if (it->second.closure_epoch != closure_epoch || it->second.valset_epoch != valset_epoch) {
    materialize_locked(s, view.height, view.round);
    return nullptr;
  }
  return &it->second;
}

size_t Messages::FilterLean(const View &view, MessageType type, const std::function<bool(const LeanRow &)> &f) {
  int s = slot(type);
  if (s < 0) return 0;
  std::unique_lock lk(mux_[s]);  // the lock GetValidMessages holds across its walk
  auto it = lean_[s].find({view.height, view.round});
  if (it == lean_[s].end()) return 0;
  it->second.filter([&](const LeanRow &row) {
    if (f(row)) return true;
    if (sender_hook_) {
      const std::string_view fr = row.from();
      sender_hook_((uint32_t)s, view.height, view.round, bytes::view(fr.data(), fr.size()), -1);
    }
    return false;  // pruned, as GetValidMessages prunes what its predicate rejects (messages.go:193-196)
  });
  return it->second.size();
}

size_t Messages::RepackLean(const View &view, MessageType type, const std::shared_ptr<const void> &backing) {
  int s = slot(type);
  if (s < 0) return 0;
  std::unique_lock lk(mux_[s]);
  auto it = lean_[s].find({view.height, view.round});
  return it == lean_[s].end() ? 0 : it->second.repack(backing);
}

void Messages::MaterializeAll() {
  for (int s : {(int)PREPARE, (int)COMMIT}) {
    std::unique_lock lk(mux_[s]);
    while (!lean_[s].empty()) {
      const auto key = lean_[s].begin()->first;
      materialize_locked(s, key.first, key.second);
    }
  }
}

void Messages::LeanStats(const View &view, MessageType type, size_t *live, size_t *slots, size_t *buffers) {
  size_t a = 0, b = 0, c = 0;
  int s = slot(type);
  if (s >= 0) {
    std::shared_lock lk(mux_[s]);
    auto lv = lean_[s].find({view.height, view.round});
    if (lv != lean_[s].end()) {
      a = lv->second.size();
      b = lv->second.slots();
      c = lv->second.buffers_held();
    }
  }
  if (live) *live = a;
  if (slots) *slots = b;
  if (buffers) *buffers = c;
}

std::vector<bytes> Messages::SendersOf(const View &view, MessageType type) {
  std::vector<bytes> out;
  int s = slot(type);
  if (s < 0) return out;
  std::shared_lock lk(mux_[s]);
  auto lv = lean_[s].find({view.height, view.round});
  if (lv != lean_[s].end())
    lv->second.for_each([&](const LeanRow &row) {
      const std::string_view f = row.from();
      out.emplace_back(f.data(), f.size());
    });
  auto h = maps_[s].find(view.height);
  if (h == maps_[s].end()) return out;
  auto r = h->second.find(view.round);
  if (r == h->second.end()) return out;
  r->second.for_each([&](const MsgPtr &m) { out.push_back(m->from); });
  return out;
}

bool ValidatorManager::Init(const std::vector<std::pair<bytes, uint64_t>> &powers) {
  std::map<bytes, uint64_t> p;
  for (auto &kv : powers) p[kv.first] = kv.second;
  unsigned __int128 total = 0;
  for (auto &kv : p) total += kv.second;
  if (total == 0) return false;  // errVotingPowerNotCorrect: state is left unchanged
  power_ = std::move(p);
  seat_addr_.clear();
  seat_power_.clear();
  size_t slots = 64;
  while (slots < power_.size() * 4) slots <<= 1;
  seat_slot_.assign(slots, 0);
  for (auto &kv : power_) {
    seat_addr_.push_back(kv.first);
    seat_power_.push_back(kv.second);
    const uint64_t hk = hash_key(kv.first.data(), kv.first.size());
    size_t sl = hk & (slots - 1);
    while (seat_slot_[sl] != 0) sl = (sl + 1) & (slots - 1);
    seat_slot_[sl] = (hk & 0xFFFFFFFF00000000ull) | (uint64_t)seat_addr_.size();
  }
  quorum_ = (total * 2) / 3 + 1;  // calculateQuorum :130-135
  initialized_ = true;
  return true;
}

bool ValidatorManager::HasQuorum(const std::set<bytes> &senders) const {
  if (!initialized_) return false;  // :82-84
  unsigned __int128 sum = 0;
  for (auto &s : senders) sum += powerOf(s);  // unknown senders contribute 0 (:88-92)
  return sum >= quorum_;
}

// The same over the From of a list of messages (convertMessageToAddressSet + HasQuorum, :77-96, :147-155): the SET of
// senders, each counted once — a flat hash set of views instead of one tree node per sender.
bool ValidatorManager::HasQuorumOf(const std::vector<MsgPtr> &msgs, const bytes *extra) const {
  if (!initialized_) return false;
  std::unordered_set<std::string_view, sv_hash> seen;  // (keyed: the senders are the network's choice)
  seen.reserve(msgs.size() * 2 + 2);
  unsigned __int128 sum = 0;
  if (extra && seen.insert(std::string_view(extra->data(), extra->size())).second) sum += powerOf(*extra);
  for (auto &m : msgs)
    if (seen.insert(std::string_view(m->from.data(), m->from.size())).second) sum += powerOf(m->from);
  return sum >= quorum_;
}

bool ValidatorManager::HasPrepareQuorum(const IbftMessage *proposal, const std::vector<MsgPtr> &msgs) const {
  if (!proposal) return false;
  for (auto &m : msgs)
    if (m->from == proposal->from) return false;  // proposer among PREPARE signers :117-121
  return HasQuorumOf(msgs, &proposal->from);
}

void QuorumIndex::OnSender(uint32_t type, uint64_t height, uint64_t round, const bytes &from, int delta,
                           const ValidatorManager &vm) {
  std::lock_guard<std::mutex> lk(mu_);
  changes_++;
  Entry *e = find(type, height, round);
  if (!e || !e->valid || e->epoch != epoch_) return;  // stale: rebuilt on demand
  const unsigned __int128 w = vm.powerOf(from);       // unknown senders contribute 0
  if (delta > 0) {
    e->power += w;
    e->count++;
  } else {
    e->power -= w;
    e->count--;
  }
}

void QuorumIndex::Add(uint32_t type, uint64_t height, uint64_t round, unsigned __int128 dpower, size_t dcount) {
  std::lock_guard<std::mutex> lk(mu_);
  changes_++;
  Entry *e = find(type, height, round);
  if (!e || !e->valid || e->epoch != epoch_) return;  // stale: rebuilt on demand
  e->power += dpower;
  e->count += dcount;
}
void QuorumIndex::OnPrune(uint64_t below_height) {
  std::lock_guard<std::mutex> lk(mu_);
  changes_++;
  for (auto it = e_.begin(); it != e_.end();)
    it = std::get<1>(it->first) < below_height ? e_.erase(it) : std::next(it);
  last_ = nullptr;
}

std::pair<unsigned __int128, size_t> QuorumIndex::Get(uint32_t type, uint64_t height, uint64_t round,
                                                      const std::function<std::vector<bytes>()> &rebuild,
                                                      const ValidatorManager &vm) {
  // Lock order: the store's hooks call OnSender / OnPrune UNDER the store's per-type lock, so this index's lock is always
  // the inner one — `rebuild` (which takes the store's lock to list the view's senders) therefore runs with this lock
  // RELEASED (ThreadSanitizer: lock-order inversion, round 4).  Whatever changed meanwhile bumped `changes_`: list again.
  std::unique_lock<std::mutex> lk(mu_);
  for (;;) {
    Entry *pe = find(type, height, round);
    if (pe && pe->valid && pe->epoch == epoch_) return {pe->power, pe->count};
    const uint64_t seen = changes_, epoch = epoch_;
    lk.unlock();
    unsigned __int128 power = 0;
    size_t count = 0;
    for (const bytes &from : rebuild()) {
      power += vm.powerOf(from);
      count++;
    }
    lk.lock();
    if (changes_ != seen || epoch_ != epoch) continue;
    pe = find(type, height, round);
    if (!pe) {
      pe = &e_[{type, height, round}];
      last_key_ = {type, height, round};
      last_ = pe;
    }
    pe->power = power;
    pe->count = count;
    pe->valid = true;
    pe->epoch = epoch;
    return {power, count};
  }
}

std::set<bytes> convertMessageToAddressSet(const std::vector<MsgPtr> &msgs) {
  std::set<bytes> s;
  for (auto &m : msgs) s.insert(m->from);
  return s;
}

}  // namespace ibft
