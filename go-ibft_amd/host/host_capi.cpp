// host_capi.cpp — C ABI of the host mirror (include/ibft_host.h).
#include "../../include/ibft_host.h"

#include <cstdlib>
#include <malloc.h>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "backend.hpp"

#include <chrono>

using namespace ibft;

namespace {

struct CallbackVerifier : Verifier {
  ibft_host_verifier cb{};
  bool IsValidProposalHash(const Proposal *p, const bytes *hash) override {
    if (!cb.is_valid_proposal_hash) return true;  // mockBackend default (mock_test.go:105-151)
    return cb.is_valid_proposal_hash(cb.user, p != nullptr, p ? (const uint8_t *)p->raw_proposal.data() : nullptr,
                                     p ? p->raw_proposal.size() : 0, p ? p->round : 0, hash != nullptr,
                                     hash ? (const uint8_t *)hash->data() : nullptr, hash ? hash->size() : 0) != 0;
  }
  bool IsValidCommittedSeal(const bytes *hash, const CommittedSeal *seal) override {
    if (!cb.is_valid_committed_seal) return true;
    return cb.is_valid_committed_seal(cb.user, hash != nullptr, hash ? (const uint8_t *)hash->data() : nullptr,
                                      hash ? hash->size() : 0, seal != nullptr,
                                      seal ? (const uint8_t *)seal->signer.data() : nullptr,
                                      seal ? seal->signer.size() : 0,
                                      seal ? (const uint8_t *)seal->signature.data() : nullptr,
                                      seal ? seal->signature.size() : 0) != 0;
  }
  bool IsValidValidator(const IbftMessage &m) override {
    if (!cb.is_valid_validator) return true;
    bytes w = encode(m);
    return cb.is_valid_validator(cb.user, (const uint8_t *)w.data(), w.size()) != 0;
  }
  bool IsProposer(const bytes &id, uint64_t height, uint64_t round) override {
    if (!rr_addrs.empty()) return id == rr_addrs[(size_t)(((unsigned __int128)height * rr_height_weight + round) % rr_addrs.size())];
    if (!cb.is_proposer) return false;
    return cb.is_proposer(cb.user, (const uint8_t *)id.data(), id.size(), height, round) != 0;
  }
  bool IsValidProposal(const bytes &raw) override {
    if (!cb.is_valid_proposal) return true;
    return cb.is_valid_proposal(cb.user, (const uint8_t *)raw.data(), raw.size()) != 0;
  }
  bytes ID() override { return id; }
  bytes id;
  std::vector<bytes> rr_addrs;  // a native round-robin proposer rule (ibft_host_set_round_robin_proposer)
  uint64_t rr_height_weight = 1;
};

void pack_bytes(bytes &o, const bytes &item) {
  uint32_t l = (uint32_t)item.size();
  o.append((const char *)&l, 4);
  o += item;
}
bool unpack_list(const uint8_t *p, size_t len, std::vector<bytes> &out) {
  size_t pos = 0;
  while (pos < len) {
    if (len - pos < 4) return false;
    uint32_t l;
    memcpy(&l, p + pos, 4);
    pos += 4;
    if (len - pos < l) return false;
    out.emplace_back((const char *)p + pos, l);
    pos += l;
  }
  return true;
}
bool unpack_msgs(const uint8_t *p, size_t len, std::vector<MsgPtr> &out) {
  std::vector<bytes> items;
  if (!unpack_list(p, len, items)) return false;
  for (auto &w : items) {
    auto m = std::make_shared<IbftMessage>();
    if (!decode((const uint8_t *)w.data(), w.size(), *m)) return false;
    out.push_back(std::move(m));
  }
  return true;
}
void to_buf(const bytes &o, size_t count, ibft_host_buf *out) {
  out->data = (uint8_t *)malloc(o.size() ? o.size() : 1);
  memcpy(out->data, o.data(), o.size());
  out->len = o.size();
  out->count = count;
}
void msgs_to_buf(const std::vector<MsgPtr> &msgs, ibft_host_buf *out) {
  bytes o;
  for (auto &m : msgs) pack_bytes(o, encode(*m));
  to_buf(o, msgs.size(), out);
}
void seals_to_buf(const std::vector<std::optional<CommittedSeal>> &seals, ibft_host_buf *out) {
  static const bytes none;
  bytes o;
  size_t total = 0;
  for (auto &s : seals) total += 9 + (s ? s->signer.size() + s->signature.size() : 0);
  o.reserve(total);
  for (auto &s : seals) {
    o.push_back(s ? 1 : 0);
    pack_bytes(o, s ? s->signer : none);
    pack_bytes(o, s ? s->signature : none);
  }
  to_buf(o, seals.size(), out);
}

}  // namespace

struct ibft_host;
namespace {
// The receive-side queue ("queue → micro-batches → device"): transport threads push what arrives and
// return at once; ONE worker per mirror takes EVERYTHING that is pending and ingests it as one batch.  The batch size
// adapts by itself — while a device call is in flight (0.3 ms for any batch up to a few thousand rows: the device is
// latency-bound there) arrivals pile up and form the next, larger batch — so under load a height's 8 191 messages reach
// the device in a handful of calls instead of 32 micro-batches, and an idle mirror still answers a lone message at once.
class IngestQueue {
 public:
  IngestQueue(ibft_host *h, size_t max_rows, uint32_t linger_us)
      : h_(h), max_rows_(max_rows ? max_rows : 65536), linger_us_(linger_us), th_([this] { loop(); }) {}
  ~IngestQueue() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    space_cv_.notify_all();
    th_.join();
  }
  // Bounded: the pending bytes are offsets of 32 bits and memory of the process.  A pusher that would take the queue past
  // its caps WAITS for the worker to take what is pending (the reference's AddMessage is synchronous: a transport thread that
  // delivers faster than messages are verified is slowed down, not buffered without end — round-3 advice); a single push
  // that could never fit is refused (−2), nothing of it queued.
  int push(const uint8_t *wire, const uint32_t *off, size_t n) {
    if (!n) return 0;
    const size_t bytes = (size_t)off[n] - off[0];
    {
      std::unique_lock<std::mutex> lk(mu_);
      // "could never fit" is judged under the lock, and again whenever the waiter wakes: set_caps may lower the caps while
      // a pusher waits, and a push larger than the NEW caps would otherwise wait for ever on an empty queue (round-4 advice)
      auto never = [&] { return bytes > cap_bytes_ || n > cap_rows_; };
      if (never()) return -2;
      if (wire_.size() + bytes > cap_bytes_ || off_.size() - 1 + n > cap_rows_) {
        backpressure_waits_++;
        cv_.notify_all();
        space_cv_.wait(lk, [&] {
          return stop_ || never() || (wire_.size() + bytes <= cap_bytes_ && off_.size() - 1 + n <= cap_rows_);
        });
        if (stop_) return -1;
        if (never()) return -2;
      }
      const uint32_t base = (uint32_t)wire_.size();
      wire_.insert(wire_.end(), wire + off[0], wire + off[n]);
      for (size_t i = 1; i <= n; i++) off_.push_back(base + (off[i] - off[0]));
      pushed_ += n;
      last_push_ = std::chrono::steady_clock::now();
    }
    cv_.notify_all();
    return 0;
  }
  void set_caps(size_t cap_bytes, size_t cap_rows) {
    std::lock_guard<std::mutex> lk(mu_);
    cap_bytes_ = std::min<size_t>(cap_bytes ? cap_bytes : kDefaultCapBytes, kMaxCapBytes);
    cap_rows_ = cap_rows ? cap_rows : kDefaultCapRows;
    space_cv_.notify_all();
  }
  uint64_t backpressure_waits() {
    std::lock_guard<std::mutex> lk(mu_);
    return backpressure_waits_;
  }
  void drain(ibft_host_queue_stats *out) {
    std::unique_lock<std::mutex> lk(mu_);
    const uint64_t target = pushed_;
    done_cv_.wait(lk, [&] { return stats_.ingested >= target || stop_; });
    if (out) {
      *out = stats_;
      out->pushed = pushed_;
    }
  }
  void set_signal(ibft_host_signal_fn fn, void *user) {
    std::lock_guard<std::mutex> lk(mu_);
    on_signal_ = fn;
    user_ = user;
  }

 private:
  void loop();
  ibft_host *h_;
  size_t max_rows_;
  uint32_t linger_us_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_, space_cv_;
  static constexpr size_t kDefaultCapBytes = (size_t)256 << 20, kMaxCapBytes = 0xFFFF0000u, kDefaultCapRows = (size_t)1 << 20;
  size_t cap_bytes_ = kDefaultCapBytes, cap_rows_ = kDefaultCapRows;  // pending, not yet taken by the worker
  uint64_t backpressure_waits_ = 0;
  std::vector<uint8_t> wire_;
  std::vector<uint32_t> off_{0};
  uint64_t pushed_ = 0;
  std::chrono::steady_clock::time_point last_push_;
  ibft_host_queue_stats stats_{};
  ibft_host_signal_fn on_signal_ = nullptr;
  void *user_ = nullptr;
  bool stop_ = false;
  std::thread th_;
};
}  // namespace

struct ibft_host {
  // One mirror = one IBFT instance.  The reference calls AddMessage from transport goroutines while the round goroutine
  // walks the store (core/ibft.go:335-347); every entry point below takes this mutex, so any thread may call any of them
  // (the ingest queue's worker included) and each call sees the mirror between two other calls, never inside one.
  std::recursive_mutex mu;
  HotPath hp;
  CallbackVerifier cbv;
  std::unique_ptr<GpuBackend> gpu;
  std::unique_ptr<LoopBatch> loop;
  size_t min_device_rows = 0;        // ibft_host_set_min_device_rows (the environment's IBFT_MIN_DEVICE_ROWS otherwise)
  bool min_device_rows_set = false;
  size_t last_set_rows = 0;
  std::unique_ptr<IngestQueue> queue;  // declared last: its worker is joined before anything above is destroyed
};

namespace {
void IngestQueue::loop() {
  std::vector<uint32_t> off;
  std::vector<int8_t> res;
  std::vector<uint8_t> types;
  for (;;) {
    // the bytes that arrived become the buffer the stored messages point into: handed to the mirror, not copied again
    auto buf = std::make_shared<std::vector<uint8_t>>();
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_ || off_.size() > 1; });
      if (stop_) return;
      if (linger_us_ && off_.size() - 1 < max_rows_) {
        // a short wait for a burst to finish arriving, as long as messages keep coming
        const auto dl = std::chrono::microseconds(linger_us_);
        while (!stop_ && off_.size() - 1 < max_rows_ && std::chrono::steady_clock::now() - last_push_ < dl)
          // (a system_clock deadline: pthread_cond_timedwait.  wait_for goes through pthread_cond_clockwait, which GCC 11's
          // ThreadSanitizer does not intercept — it then misses the unlock / relock inside the wait and reports "double lock"
          // and races on everything this mutex guards: tests/test_host_sanitize.py, flaky exactly when this line waited.  The
          // loop re-reads the steady clock, so a wall-clock step only ends one 13 µs nap early or late.)
          cv_.wait_until(lk, std::chrono::system_clock::now() + dl / 4 + std::chrono::microseconds(1));
        if (stop_) return;
      }
      off.swap(off_);
      off_.assign(1, 0);
      buf->swap(wire_);
      wire_.reserve(buf->size());  // (the next burst is about as large: no regrowth from nothing under the pushers' feet)
      space_cv_.notify_all();      // pushers that waited for room
    }
    const std::vector<uint8_t> &wire = *buf;
    const std::shared_ptr<const void> shared(buf, buf->data());
    const size_t n = off.size() - 1;
    res.assign(n, -1);
    types.assign(n, 0xFF);
    ibft_host_queue_stats d{};
    uint64_t sig_height = 0, sig_round = 0;
    for (size_t lo = 0; lo < n; lo += max_rows_) {
      const size_t k = std::min(max_rows_, n - lo);
      std::lock_guard<std::recursive_mutex> hk(h_->mu);
      // rows [lo, lo + k): offsets rebased by IngestFlat's contract (off[0] may be non-zero: it reads wire[off[i] .. off[i+1]))
      HotPath::IngestStats st;
      const auto t0 = std::chrono::steady_clock::now();
      h_->hp.IngestFlat(wire.data(), off.data() + lo, k, res.data() + lo, &st, types.data() + lo, shared);
      d.ingest_us += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      d.device_us += (uint64_t)(st.device_ms * 1e3);
      d.batches++;
      d.device_calls += st.device_calls;
      d.cache_hits += st.cache_hits;
      if (k > d.max_batch_rows) d.max_batch_rows = k;
      sig_height = h_->hp.height;
      sig_round = h_->hp.round;
    }
    for (size_t i = 0; i < n; i++) {
      if (res[i] < 0)
        d.undecodable++;
      else if (res[i] == 0)
        d.rejected++;
      else
        d.stored++;
      if (res[i] == 2 && types[i] < 4) d.signals[types[i]]++;
    }
    ibft_host_signal_fn fn;
    void *user;
    {
      std::lock_guard<std::mutex> lk(mu_);
      stats_.ingested += n;
      stats_.stored += d.stored;
      stats_.rejected += d.rejected;
      stats_.undecodable += d.undecodable;
      stats_.batches += d.batches;
      stats_.device_calls += d.device_calls;
      stats_.cache_hits += d.cache_hits;
      stats_.ingest_us += d.ingest_us;
      stats_.device_us += d.device_us;
      if (d.max_batch_rows > stats_.max_batch_rows) stats_.max_batch_rows = d.max_batch_rows;
      for (int t = 0; t < 4; t++) stats_.signals[t] += d.signals[t];
      fn = on_signal_;
      user = user_;
    }
    // SignalEvent (core/ibft.go:1119), coalesced per type like the 1-deep subscription channels (event_subscription.go:80-83);
    // called WITHOUT the mirror's lock: the callee may call ibft_host_handle_* right away
    if (fn)
      for (uint32_t t = 0; t < 4; t++)
        if (d.signals[t]) fn(user, t, sig_height, sig_round);
    done_cv_.notify_all();
  }
}
}  // namespace

extern "C" {

ibft_host *ibft_host_new(void) {
  auto *h = new ibft_host();
  h->hp.verifier = &h->cbv;
  return h;
}
void ibft_host_free(ibft_host *h) {
  ibft_host_queue_stop(h);
  delete h;
}
void ibft_host_buf_free(ibft_host_buf *b) {
  if (b && b->data) free(b->data);
  if (b) *b = ibft_host_buf{nullptr, 0, 0};
}

int ibft_host_verify_senders_wire(ibft_ctx *ctx, const uint8_t *wire, const uint32_t *off, size_t n, int mode,
                                  uint8_t *verdict, double *host_ms, size_t *host_rows) {
  if (!ctx || (n && (!off || !verdict))) return IBFT_E_INVAL;
  GpuBackend gb(ctx);
  std::vector<uint8_t> v;
  GpuBackend::WireStats st;
  if (mode == 0) {
    if (!gb.VerifySendersWire(wire, off, n, v, &st)) return gb.last_rc ? gb.last_rc : IBFT_E_INVAL;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<size_t> idx;
    std::vector<MsgPtr> msgs;
    for (size_t i = 0; i < n; i++) {
      auto m = std::make_shared<IbftMessage>();
      if (!decode(wire + off[i], off[i + 1] - off[i], *m)) continue;
      idx.push_back(i);
      msgs.push_back(std::move(m));
    }
    SenderColumns c;
    flatten_senders(msgs, c);
    st.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    st.host_rows = idx.size();
    std::vector<uint8_t> v2(msgs.size(), 0);
    if (!msgs.empty()) {
      std::vector<uint64_t> mask((c.n + 63) / 64, 0);
      const int rc = ibft_verify_senders(ctx, c.payload.data(), c.off.data(), c.sig65.data(), c.from20.data(),
                                         c.pre_flags.data(), c.n, mask.data(), nullptr);
      if (rc != IBFT_OK) return rc;
      for (size_t j = 0; j < c.n; j++) v2[j] = (mask[j >> 6] >> (j & 63)) & 1;
    }
    v.assign(n, 0);
    for (size_t j = 0; j < idx.size(); j++) v[idx[j]] = v2[j];
  }
  for (size_t i = 0; i < n; i++) verdict[i] = v[i];
  if (host_ms) *host_ms = st.host_ms;
  if (host_rows) *host_rows = st.host_rows;
  return 0;
}

// Measurement aid (tools/cert_from_wire.py): every signature of a batch of certificate-carrying messages, two ways.
// route 0: the transport's bytes straight to ibft_verify_certificates_wire.  route 1: the route of f2 before it — decode the
// messages, collect every nested message, re-marshal PayloadNoSig of each, flatten into columns (host_ms) and
// ibft_verify_senders.  rows = messages judged, valid = those whose sender check passed.
int ibft_host_cert_routes(ibft_ctx *ctx, const uint8_t *wire, const uint32_t *off, size_t n, int route, size_t rows_cap, size_t *rows,
                          size_t *valid, double *host_ms, double *total_ms) {
  if (!ctx || !off || !rows || !valid) return IBFT_E_INVAL;
  const auto t0 = std::chrono::steady_clock::now();
  double hms = 0.0;
  *rows = *valid = 0;
  GpuBackend gb(ctx);
  if (rows_cap) gb.cert_rows_cap = rows_cap;
  if (route == 0) {
    CertVerdicts cv;
    if (!gb.VerifyCertificatesWire(wire, off, n, cv)) return gb.last_rc ? gb.last_rc : IBFT_E_INVAL;
    *rows = cv.n_rows;
    for (uint8_t b : cv.sender) *valid += b;
  } else {
    std::vector<MsgPtr> all;
    for (size_t i = 0; i < n; i++) {
      auto m = std::make_shared<IbftMessage>();
      if (!decode(wire + off[i], off[i + 1] - off[i], *m)) continue;
      all.push_back(std::move(m));
    }
    for (size_t k = 0; k < all.size(); k++) {  // breadth first, like the device
      const IbftMessage &m = *all[k];
      if (m.kind == PayloadKind::PREPREPARE && m.preprepare().realise_certificate() && m.preprepare().certificate) {
        for (auto &c : m.preprepare().certificate->round_change_messages) all.push_back(c);
      } else if (m.kind == PayloadKind::ROUND_CHANGE && m.round_change().realise_certificate() && m.round_change().latest_prepared_certificate) {
        const PreparedCertificate &pc = *m.round_change().latest_prepared_certificate;
        if (pc.proposal_message) all.push_back(pc.proposal_message);
        for (auto &c : pc.prepare_messages) all.push_back(c);
      }
    }
    std::vector<uint8_t> v;
    SenderColumns c;
    flatten_senders(all, c);
    hms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint64_t> mask((c.n + 63) / 64, 0);
    const int rc = ibft_verify_senders(ctx, c.payload.data(), c.off.data(), c.sig65.data(), c.from20.data(), c.pre_flags.data(), c.n,
                                       mask.data(), nullptr);
    if (rc != IBFT_OK) return rc;
    *rows = c.n;
    for (size_t j = 0; j < c.n; j++) *valid += (mask[j >> 6] >> (j & 63)) & 1;
  }
  if (host_ms) *host_ms = hms;
  if (total_ms) *total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int ibft_host_payload_no_sig(const uint8_t *wire, size_t len, ibft_host_buf *out) {
  IbftMessage m;
  if (!decode(wire, len, m)) return -1;
  to_buf(payload_no_sig(m), 1, out);
  return 0;
}
int ibft_host_reencode(const uint8_t *wire, size_t len, ibft_host_buf *out) {
  IbftMessage m;
  if (!decode(wire, len, m)) return -1;
  to_buf(encode(m), 1, out);
  return 0;
}

int ibft_host_store_add(ibft_host *h, const uint8_t *wire, size_t len) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  auto m = std::make_shared<IbftMessage>();
  if (!decode(wire, len, *m)) return -1;
  h->hp.messages.AddMessage(std::move(m));
  return 0;
}
size_t ibft_host_store_num(ibft_host *h, uint64_t height, uint64_t round, uint32_t type) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  return h->hp.messages.numMessages(View{height, round, {}}, (MessageType)type);
}
void ibft_host_store_prune(ibft_host *h, uint64_t height) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.messages.PruneByHeight(height); }

int ibft_host_store_get_valid(ibft_host *h, uint64_t height, uint64_t round, uint32_t type,
                              ibft_host_msg_pred pred, void *user, ibft_host_buf *out) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  auto msgs = h->hp.messages.GetValidMessages(View{height, round, {}}, (MessageType)type, [&](const IbftMessage &m) {
    if (!pred) return true;
    bytes w = encode(m);
    return pred(user, (const uint8_t *)w.data(), w.size()) != 0;
  });
  msgs_to_buf(msgs, out);
  return 0;
}
int ibft_host_store_get_extended_rcc(ibft_host *h, uint64_t height, ibft_host_msg_pred pred,
                                     ibft_host_rcc_pred rcc_pred, void *user, ibft_host_buf *out) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  auto msgs = h->hp.messages.GetExtendedRCC(
      height,
      [&](const IbftMessage &m) {
        if (!pred) return true;
        bytes w = encode(m);
        return pred(user, (const uint8_t *)w.data(), w.size()) != 0;
      },
      [&](uint64_t round, const std::vector<MsgPtr> &v) { return rcc_pred ? rcc_pred(user, round, v.size()) != 0 : true; });
  msgs_to_buf(msgs, out);
  return 0;
}
int ibft_host_store_get_extended_rcc_msgs(ibft_host *h, uint64_t height, ibft_host_msg_pred pred,
                                          ibft_host_rcc_msgs_pred rcc_pred, void *user, ibft_host_buf *out) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  auto msgs = h->hp.messages.GetExtendedRCC(
      height,
      [&](const IbftMessage &m) {
        if (!pred) return true;
        bytes w = encode(m);
        return pred(user, (const uint8_t *)w.data(), w.size()) != 0;
      },
      [&](uint64_t round, const std::vector<MsgPtr> &v) {
        if (!rcc_pred) return true;
        ibft_host_buf b{};
        msgs_to_buf(v, &b);
        const int r = rcc_pred(user, round, b.data, b.len, b.count);
        ibft_host_buf_free(&b);
        return r != 0;
      });
  msgs_to_buf(msgs, out);
  return 0;
}
int ibft_host_store_get_most_rc(ibft_host *h, uint64_t min_round, uint64_t height, ibft_host_buf *out) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  msgs_to_buf(h->hp.messages.GetMostRoundChangeMessages(min_round, height), out);
  return 0;
}

int ibft_host_has_unique_senders(const uint8_t *packed, size_t len) {
  std::vector<MsgPtr> msgs;
  if (!unpack_msgs(packed, len, msgs)) return -1;
  return has_unique_senders(msgs) ? 1 : 0;
}
int ibft_host_are_valid_pc_messages(const uint8_t *packed, size_t len, uint64_t height, uint64_t round_limit) {
  std::vector<MsgPtr> msgs;
  if (!unpack_msgs(packed, len, msgs)) return -1;
  return are_valid_pc_messages(msgs, height, round_limit) ? 1 : 0;
}
int ibft_host_extract_committed_seals(const uint8_t *packed, size_t len, ibft_host_buf *out) {
  std::vector<MsgPtr> msgs;
  if (!unpack_msgs(packed, len, msgs)) return -2;
  std::vector<std::optional<CommittedSeal>> seals;
  if (!extract_committed_seals(msgs, seals)) return -1;
  seals_to_buf(seals, out);
  return 0;
}

int ibft_host_vm_init(ibft_host *h, const uint8_t *packed_addrs, size_t len, const uint64_t *power, size_t n) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  std::vector<bytes> addrs;
  if (!unpack_list(packed_addrs, len, addrs) || addrs.size() != n) return -2;
  std::vector<std::pair<bytes, uint64_t>> p;
  for (size_t i = 0; i < n; i++) p.emplace_back(addrs[i], power[i]);
  bool ok = h->hp.validatorManager.Init(p);
  if (ok) h->hp.NotifyValidatorSetChanged();
  return ok ? 0 : -1;
}
int ibft_host_vm_has_quorum(ibft_host *h, const uint8_t *packed_senders, size_t len) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  std::vector<bytes> s;
  if (!unpack_list(packed_senders, len, s)) return -1;
  return h->hp.validatorManager.HasQuorum(std::set<bytes>(s.begin(), s.end())) ? 1 : 0;
}
int ibft_host_vm_has_prepare_quorum(ibft_host *h, const uint8_t *proposal_wire, size_t proposal_len,
                                    const uint8_t *packed_msgs, size_t len) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  std::vector<MsgPtr> msgs;
  if (!unpack_msgs(packed_msgs, len, msgs)) return -1;
  IbftMessage pm;
  const IbftMessage *pp = nullptr;
  if (proposal_wire) {
    if (!decode(proposal_wire, proposal_len, pm)) return -1;
    pp = &pm;
  }
  return h->hp.validatorManager.HasPrepareQuorum(pp, msgs) ? 1 : 0;
}
void ibft_host_vm_quorum(ibft_host *h, uint64_t *lo, uint64_t *hi) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  unsigned __int128 q = h->hp.validatorManager.quorum();
  *lo = (uint64_t)q;
  *hi = (uint64_t)(q >> 64);
}

int ibft_host_set_state(ibft_host *h, uint64_t height, uint64_t round, const uint8_t *proposal_wire,
                        size_t proposal_len) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->hp.height = height;
  h->hp.round = round;
  h->hp.proposalMessage.reset();
  if (proposal_wire) {
    auto m = std::make_shared<IbftMessage>();
    if (!decode(proposal_wire, proposal_len, *m)) return -1;
    h->hp.proposalMessage = std::move(m);
  }
  return 0;
}
void ibft_host_set_verifier(ibft_host *h, const ibft_host_verifier *v) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->cbv.cb = v ? *v : ibft_host_verifier{};
}
void ibft_host_attach_gpu(ibft_host *h, ibft_ctx *ctx) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->gpu = ctx ? std::make_unique<GpuBackend>(ctx) : nullptr;
  if (h->gpu && h->min_device_rows_set) h->gpu->min_device_rows = h->min_device_rows;
  h->hp.batch = h->gpu.get();
}
void ibft_host_use_batch(ibft_host *h, int on) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.use_batch = on != 0; }

int ibft_host_add_message(ibft_host *h, const uint8_t *wire, size_t len) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  auto m = std::make_shared<IbftMessage>();
  if (!decode(wire, len, *m)) return -1;
  return h->hp.AddMessage(std::move(m));
}

void ibft_host_enable_quorum_index(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.EnableQuorumIndex(); }
int ibft_host_add_message_fast(ibft_host *h, const uint8_t *wire, size_t len) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  auto m = std::make_shared<IbftMessage>();
  if (!decode(wire, len, *m)) return -1;
  return h->hp.AddMessageFast(std::move(m));
}

int ibft_host_add_messages_batch(ibft_host *h, const uint8_t *packed, size_t len, uint8_t *results, size_t n) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  std::vector<MsgPtr> msgs;
  if (!unpack_msgs(packed, len, msgs) || msgs.size() != n) return -1;
  if (!h->hp.batch) return -2;
  std::vector<uint8_t> ok;
  if (!h->hp.batch->VerifySenderBatch(msgs, ok)) return -3;
  // IBFT.AddMessage per message with IsValidValidator answered from the verdict table; the O(1) quorum probe
  // (AddMessageFast) when the quorum index is enabled
  for (size_t i = 0; i < n; i++) results[i] = (uint8_t)h->hp.addWithVerdict(msgs[i], ok[i] != 0);
  return 0;
}

static int ingest_flat(ibft_host *h, const uint8_t *wire, const uint32_t *off, size_t n, int8_t *results, size_t *device_rows,
                       size_t *cache_hits, size_t *device_calls) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  HotPath::IngestStats st;
  if (!h->hp.IngestFlat(wire, off, n, results, &st)) return -3;
  if (device_rows) *device_rows = st.device_rows;
  if (cache_hits) *cache_hits = st.cache_hits;
  if (device_calls) *device_calls = st.device_calls;
  h->last_set_rows = st.set_rows;
  return 0;
}
int ibft_host_ingest_wire(ibft_host *h, const uint8_t *packed, size_t len, int8_t *results, size_t n, size_t *device_rows,
                          size_t *cache_hits, size_t *device_calls) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  // repeated {u32 length, bytes} → rows back to back + offsets (one pass, one buffer)
  std::vector<uint8_t> wire;
  std::vector<uint32_t> off{0};
  wire.reserve(len);
  off.reserve(n + 1);
  size_t pos = 0;
  while (pos < len) {
    if (len - pos < 4) return -1;
    uint32_t l;
    memcpy(&l, packed + pos, 4);
    pos += 4;
    if (l > len - pos) return -1;
    wire.insert(wire.end(), packed + pos, packed + pos + l);
    off.push_back((uint32_t)wire.size());
    pos += l;
  }
  if (off.size() != n + 1) return -1;
  return ingest_flat(h, wire.data(), off.data(), n, results, device_rows, cache_hits, device_calls);
}
int ibft_host_ingest_flat(ibft_host *h, const uint8_t *wire, const uint32_t *off, size_t n, int8_t *results,
                          size_t *device_rows, size_t *cache_hits, size_t *device_calls) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (!h || (n && (!wire || !off || !results))) return -1;
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return -1;
  return ingest_flat(h, wire, off, n, results, device_rows, cache_hits, device_calls);
}
int ibft_host_queue_start(ibft_host *h, size_t max_rows, uint32_t linger_us) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (h->queue) return -1;
  h->queue.reset(new IngestQueue(h, max_rows, linger_us));
  return 0;
}
void ibft_host_queue_on_signal(ibft_host *h, ibft_host_signal_fn fn, void *user) {
  if (h->queue) h->queue->set_signal(fn, user);
}
int ibft_host_queue_push(ibft_host *h, const uint8_t *wire, const uint32_t *off, size_t n) {
  if (!h->queue || (n && (!wire || !off))) return -1;  // (no mirror lock: pushing never waits for an ingest in progress)
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return -1;
  return h->queue->push(wire, off, n);
}
void ibft_host_queue_set_caps(ibft_host *h, size_t max_pending_bytes, size_t max_pending_rows) {
  if (h->queue) h->queue->set_caps(max_pending_bytes, max_pending_rows);
}
uint64_t ibft_host_queue_backpressure_waits(ibft_host *h) { return h->queue ? h->queue->backpressure_waits() : 0; }
int ibft_host_queue_drain(ibft_host *h, ibft_host_queue_stats *out) {
  if (!h->queue) return -1;
  h->queue->drain(out);
  return 0;
}
void ibft_host_queue_stop(ibft_host *h) {
  std::unique_ptr<IngestQueue> q;
  {
    std::lock_guard<std::recursive_mutex> lk_(h->mu);
    q = std::move(h->queue);
  }
  q.reset();  // joins the worker (which may be waiting for the mirror's lock) outside the lock
}
size_t ibft_host_seen_entries(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.seen_entries(); }
void ibft_host_set_seen_caps(ibft_host *h, size_t stored_cap, size_t rejected_cap) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->hp.seen_cap = stored_cap;
  h->hp.rejected_cap = rejected_cap;
}
void ibft_host_use_sets(ibft_host *h, int on) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.use_sets = on != 0; }
void ibft_host_use_rc_rows(ibft_host *h, int on) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.use_rc_rows = on != 0; }
size_t ibft_host_repacked_bytes(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.repacked_bytes; }
void ibft_host_set_repack_min_bytes(ibft_host *h, size_t bytes) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.repack_min_bytes = bytes; }
size_t ibft_host_pp_from_rows(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.pp_from_rows; }
size_t ibft_host_rc_from_rows(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.rc_from_rows; }
void ibft_host_cert_roots_first(ibft_host *h, int mode) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.cert_roots_first = mode; }
size_t ibft_host_roots_first_calls(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.roots_first_calls; }
double ibft_host_last_ingest_device_ms(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.last_ingest_device_ms; }
int ibft_host_retain_heap(size_t bytes) {
  // glibc: freed memory at the top of the heap above M_TRIM_THRESHOLD goes back to the kernel, blocks above
  // M_MMAP_THRESHOLD are mapped and unmapped one by one — either way the next height's buffers are fresh pages again
  const size_t cap = bytes > ((size_t)1 << 30) ? ((size_t)1 << 30) : bytes;
  int ok = mallopt(M_TRIM_THRESHOLD, (int)cap);
  ok &= mallopt(M_TOP_PAD, (int)(cap / 8 > ((size_t)64 << 20) ? ((size_t)64 << 20) : cap / 8));
  ok &= mallopt(M_MMAP_THRESHOLD, (int)(cap > ((size_t)32 << 20) ? ((size_t)32 << 20) : cap));
  return ok ? 0 : -1;
}
void ibft_host_use_device_quorum(ibft_host *h, int on) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.device_quorum = on != 0; }
void ibft_host_device_quorum_stats(ibft_host *h, size_t *calls, size_t *mismatches) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (calls) *calls = h->hp.device_quorum_calls;
  if (mismatches) *mismatches = h->hp.device_quorum_mismatches;
}
void ibft_host_use_rows(ibft_host *h, int on) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.use_lean = on != 0; }
void ibft_host_lean_stats(ibft_host *h, uint64_t height, uint64_t round, uint32_t type, size_t *live, size_t *slots, size_t *buffers) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->hp.messages.LeanStats(View{height, round, {}}, (MessageType)type, live, slots, buffers);
}
size_t ibft_host_rows_kept(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.lean_rows; }
void ibft_host_use_certs(ibft_host *h, int on) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->hp.use_certs = on != 0; }
void ibft_host_cert_stats(ibft_host *h, size_t *calls, size_t *rows, size_t *hits) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (calls) *calls = h->hp.cert_calls;
  if (rows) *rows = h->hp.cert_rows;
  if (hits) *hits = h->hp.cert_hits;
}
size_t ibft_host_loop_batch_cert_calls(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->loop ? h->loop->cert_calls : 0; }
int ibft_host_handle_preprepare(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *msg) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  MsgPtr m = h->hp.handlePrePrepare(View{height, round, {}});
  if (msg) msgs_to_buf(m ? std::vector<MsgPtr>{m} : std::vector<MsgPtr>{}, msg);
  return m ? 1 : 0;
}
size_t ibft_host_last_set_rows(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->last_set_rows; }
size_t ibft_host_closure_hits(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.closure_hits; }
size_t ibft_host_loop_batch_set_calls(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->loop ? h->loop->set_calls : 0; }

void ibft_host_use_loop_batch(ibft_host *h, int fail_mask) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->loop.reset(new LoopBatch(&h->cbv));
  h->loop->quorum_vm = &h->hp.validatorManager;
  h->loop->fail_hashes = (fail_mask & 1) != 0;
  h->loop->fail_seals = (fail_mask & 2) != 0;
  h->loop->fail_senders = (fail_mask & 4) != 0;
  h->loop->fail_sets = (fail_mask & 8) != 0;
  h->loop->fail_certs = (fail_mask & 16) != 0;
  h->loop->fail_quorum = (fail_mask & 32) != 0;
  h->loop->wrong_quorum = (fail_mask & 64) != 0;
  if (h->min_device_rows_set) h->loop->min_device_rows = h->min_device_rows;
  h->hp.batch = h->loop.get();
}
// SURVEY §5 "min batch for GPU": batches below `rows` rows are declined by the batch verifier (device or loop) and the stock
// closures run — backend.hpp: BatchVerifier::min_device_rows.  Applies to the verifier attached NOW and to any attached later.
void ibft_host_set_min_device_rows(ibft_host *h, size_t rows) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  h->min_device_rows = rows;
  h->min_device_rows_set = true;
  if (h->gpu) h->gpu->min_device_rows = rows;
  if (h->loop) h->loop->min_device_rows = rows;
}
size_t ibft_host_declined_batches(ibft_host *h) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  return (h->gpu ? h->gpu->declined : 0) + (h->loop ? h->loop->declined : 0);
}
size_t ibft_host_loop_batch_calls(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->loop ? h->loop->calls : 0; }
size_t ibft_host_fallbacks(ibft_host *h) { std::lock_guard<std::recursive_mutex> lk_(h->mu); return h->hp.fallbacks; }

int ibft_host_handle_round_change(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *rcc) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  std::vector<MsgPtr> out = h->hp.handleRoundChangeMessage(View{height, round, {}});
  if (rcc) msgs_to_buf(out, rcc);
  return out.empty() ? 0 : 1;
}

// test hook: the receive side's look at a message before it is decoded, next to what the decoder finds.
// out[0..5] = peek {ok, has_view, height, round, type, kind}; out[6..11] = the same six read off the decoded message.
int ibft_host_peek_vs_decode(const uint8_t *wire, size_t len, uint64_t out[12]) {
  const Peek pk = peek(wire, len);
  out[0] = pk.ok; out[1] = pk.has_view; out[2] = pk.height; out[3] = pk.round; out[4] = pk.type; out[5] = (uint64_t)pk.kind;
  IbftMessage m;
  const bool ok = decode(wire, len, m);
  out[6] = ok; out[7] = ok && m.view.has_value(); out[8] = (ok && m.view) ? m.view->height : 0;
  out[9] = (ok && m.view) ? m.view->round : 0; out[10] = ok ? m.type : 0; out[11] = ok ? (uint64_t)m.kind : 0;
  return 0;
}
int ibft_host_peek_shortcut_agrees(const uint8_t *wire, size_t len) {
  const Peek a = peek(wire, len), b = peek_general(wire, len);
  const bool same = a.ok == b.ok && (!a.ok || (a.has_view == b.has_view && a.height == b.height && a.round == b.round && a.type == b.type &&
                                               a.kind == b.kind && a.simple == b.simple &&
                                               (!a.simple || (a.from_off == b.from_off && a.from_len == b.from_len && a.sig_off == b.sig_off &&
                                                              a.sig_len == b.sig_len && a.hash_off == b.hash_off && a.hash_len == b.hash_len &&
                                                              a.seal_off == b.seal_off && a.seal_len == b.seal_len))));
  return same ? (a.ok && a.simple ? 2 : 1) : 0;
}
int ibft_host_set_round_robin_proposer(ibft_host *h, const uint8_t *packed_addrs, size_t len, int use_height) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  std::vector<bytes> a;
  if (!unpack_list(packed_addrs, len, a)) return -1;
  h->cbv.rr_addrs = std::move(a);
  h->cbv.rr_height_weight = use_height ? 1 : 0;
  return 0;
}
void ibft_host_set_id(ibft_host *h, const uint8_t *id, size_t len) { std::lock_guard<std::recursive_mutex> lk_(h->mu); h->cbv.id.assign((const char *)id, len); }
int ibft_host_valid_pc(ibft_host *h, const uint8_t *pc_wire, size_t len, uint64_t round_limit, uint64_t height) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (!pc_wire) return h->hp.validPC(nullptr, round_limit, height) ? 1 : 0;
  PreparedCertificate pc;
  if (!decode(pc_wire, len, pc)) return -1;
  return h->hp.validPC(&pc, round_limit, height) ? 1 : 0;
}
int ibft_host_proposal_matches_certificate(ibft_host *h, const uint8_t *proposal_wire, size_t plen,
                                           const uint8_t *pc_wire, size_t clen) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  Proposal p;
  PreparedCertificate pc;
  if (proposal_wire && !decode(proposal_wire, plen, p)) return -1;
  if (pc_wire && !decode(pc_wire, clen, pc)) return -1;
  return h->hp.proposalMatchesCertificate(proposal_wire ? &p : nullptr, pc_wire ? &pc : nullptr) ? 1 : 0;
}
int ibft_host_validate_proposal0(ibft_host *h, const uint8_t *msg_wire, size_t len, uint64_t height, uint64_t round) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  IbftMessage m;
  if (!decode(msg_wire, len, m)) return -1;
  return h->hp.validateProposal0(m, View{height, round, {}}) ? 1 : 0;
}
int ibft_host_validate_proposal(ibft_host *h, const uint8_t *msg_wire, size_t len, uint64_t height, uint64_t round) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  IbftMessage m;
  if (!decode(msg_wire, len, m)) return -1;
  return h->hp.validateProposal(m, View{height, round, {}}) ? 1 : 0;
}
void ibft_host_last_cert_batch(ibft_host *h, size_t *senders, size_t *hashes) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  if (senders) *senders = h->hp.last_cert_senders;
  if (hashes) *hashes = h->hp.last_cert_hashes;
}

int ibft_host_handle_prepare(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *prepared) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  bool q = h->hp.handlePrepare(View{height, round, {}});
  if (prepared) {  // PC.PrepareMessages: the stored bytes when the view is held as rows, the objects' encoding otherwise
    bytes o;
    std::vector<bytes> w = q ? h->hp.PreparedWire() : std::vector<bytes>{};
    for (const bytes &b : w) pack_bytes(o, b);
    to_buf(o, w.size(), prepared);
  }
  return q ? 1 : 0;
}
int ibft_host_handle_commit(ibft_host *h, uint64_t height, uint64_t round, ibft_host_buf *seals) {
  std::lock_guard<std::recursive_mutex> lk_(h->mu);
  bool q = h->hp.handleCommit(View{height, round, {}});
  if (seals) {
    bytes o;
    const size_t count = q ? h->hp.PackCommittedSeals(o) : 0;
    to_buf(o, count, seals);
  }
  return q ? 1 : 0;
}

}  // extern "C"
