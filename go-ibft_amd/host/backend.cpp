// backend.cpp — see backend.hpp.
#include "backend.hpp"

#include <algorithm>

#include <chrono>

#include <cstring>

namespace ibft {

static void put_fixed(std::vector<uint8_t> &col, const bytes *src, size_t width, bool &bad) {
  size_t base = col.size();
  col.resize(base + width, 0);
  if (src && src->size() == width)
    memcpy(col.data() + base, src->data(), width);
  else
    bad = true;
}

void flatten_commits(const std::vector<MsgPtr> &msgs, SealColumns &c) {
  c = SealColumns{};
  c.n = msgs.size();
  for (const auto &m : msgs) {
    const bytes *hash = extract_commit_hash(*m);             // nil if wrong type/payload
    std::optional<CommittedSeal> seal = extract_committed_seal(*m);
    uint8_t pre = 0;
    bool bad = false;
    put_fixed(c.hash32, hash, 32, bad);
    c.hash_len.push_back(hash ? (uint8_t)(hash->size() > 255 ? 255 : hash->size()) : 0);
    if (!hash || !seal) pre |= IBFT_ROW_NIL;
    bad = false;
    put_fixed(c.sig65, seal ? &seal->signature : nullptr, 65, bad);
    if (bad) pre |= IBFT_ROW_BADLEN;
    bad = false;
    put_fixed(c.signer20, seal ? &seal->signer : nullptr, 20, bad);
    if (bad) pre |= IBFT_ROW_BADLEN;
    c.pre_flags.push_back(pre);
  }
}

void flatten_prepares(const std::vector<MsgPtr> &msgs, SealColumns &c) {
  c = SealColumns{};
  c.n = msgs.size();
  for (const auto &m : msgs) {
    const bytes *hash = extract_prepare_hash(*m);
    bool bad = false;
    put_fixed(c.hash32, hash, 32, bad);
    c.hash_len.push_back(hash ? (uint8_t)(hash->size() > 255 ? 255 : hash->size()) : 0);
  }
}

void flatten_senders(const std::vector<MsgPtr> &msgs, SenderColumns &c) {
  c = SenderColumns{};
  c.n = msgs.size();
  c.off.push_back(0);
  for (const auto &m : msgs) {
    bytes pns = payload_no_sig(*m);
    c.payload.insert(c.payload.end(), pns.begin(), pns.end());
    c.off.push_back((uint32_t)c.payload.size());
    uint8_t pre = 0;
    bool bad = false;
    put_fixed(c.sig65, &m->signature, 65, bad);
    put_fixed(c.from20, &m->from, 20, bad);
    if (bad) pre |= IBFT_ROW_BADLEN;
    c.pre_flags.push_back(pre);
  }
}

static void unpack_mask(const std::vector<uint64_t> &mask, size_t n, std::vector<uint8_t> &v) {
  v.assign(n, 0);
  for (size_t i = 0; i < n; i++) v[i] = (mask[i >> 6] >> (i & 63)) & 1;
}

bool GpuBackend::VerifyPrepareBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs,
                                    std::vector<uint8_t> &verdict) {
  verdict.assign(msgs.size(), 0);
  if (!proposal || msgs.empty()) return true;  // nil proposal: every hash check is false
  SealColumns c;
  flatten_prepares(msgs, c);
  std::vector<uint64_t> mask((c.n + 63) / 64, 0);
  last_rc = ibft_verify_hashes(ctx_, (const uint8_t *)proposal->raw_proposal.data(), proposal->raw_proposal.size(),
                               proposal->round, c.hash32.data(), c.hash_len.data(), c.n, mask.data());
  if (last_rc != IBFT_OK) return false;
  unpack_mask(mask, c.n, verdict);
  return true;
}

bool GpuBackend::VerifyCommitBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs,
                                   std::vector<uint8_t> &verdict) {
  verdict.assign(msgs.size(), 0);
  if (!proposal || msgs.empty()) return true;
  SealColumns c;
  flatten_commits(msgs, c);
  std::vector<uint64_t> m1((c.n + 63) / 64, 0), m2((c.n + 63) / 64, 0);
  last_rc = ibft_verify_hashes(ctx_, (const uint8_t *)proposal->raw_proposal.data(), proposal->raw_proposal.size(),
                               proposal->round, c.hash32.data(), c.hash_len.data(), c.n, m1.data());
  if (last_rc != IBFT_OK) return false;
  // a2 is skipped where a1 failed (ibft.go:938-943): mark those rows so the device does no work
  for (size_t i = 0; i < c.n; i++)
    if (!((m1[i >> 6] >> (i & 63)) & 1)) c.pre_flags[i] |= IBFT_ROW_HASH_BAD;
  last_rc = ibft_verify_seals(ctx_, c.hash32.data(), c.sig65.data(), c.signer20.data(), c.pre_flags.data(), c.n,
                              m2.data(), nullptr);
  if (last_rc != IBFT_OK) return false;
  unpack_mask(m2, c.n, verdict);
  return true;
}

bool GpuBackend::VerifySenderBatch(const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &verdict) {
  verdict.assign(msgs.size(), 0);
  if (msgs.empty()) return true;
  SenderColumns c;
  flatten_senders(msgs, c);
  std::vector<uint64_t> mask((c.n + 63) / 64, 0);
  last_rc = ibft_verify_senders(ctx_, c.payload.data(), c.off.data(), c.sig65.data(), c.from20.data(),
                                c.pre_flags.data(), c.n, mask.data(), nullptr);
  if (last_rc != IBFT_OK) return false;
  unpack_mask(mask, c.n, verdict);
  return true;
}

bool GpuBackend::VerifyMessageSet(const Proposal *proposal, MessageType type, const std::vector<MsgPtr> &msgs,
                                  std::vector<uint8_t> &sender, std::vector<uint8_t> &closure) {
  sender.assign(msgs.size(), 0);
  closure.assign(msgs.size(), 0);
  if ((type != PREPARE && type != COMMIT) || !proposal) return false;  // nothing to check the hashes against
  if (msgs.empty()) return true;
  SenderColumns sc;
  flatten_senders(msgs, sc);
  SealColumns cc;
  if (type == COMMIT) flatten_commits(msgs, cc); else flatten_prepares(msgs, cc);
  std::vector<uint64_t> ms((sc.n + 63) / 64, 0), mv((sc.n + 63) / 64, 0);
  last_rc = ibft_verify_messages(ctx_, sc.payload.data(), sc.off.data(), sc.sig65.data(), sc.from20.data(), cc.hash32.data(),
                                 cc.hash_len.data(), type == COMMIT ? cc.sig65.data() : nullptr, sc.pre_flags.data(),
                                 type == COMMIT ? cc.pre_flags.data() : nullptr, sc.n,
                                 (const uint8_t *)proposal->raw_proposal.data(), proposal->raw_proposal.size(),
                                 proposal->round, nullptr, ms.data(), mv.data(), nullptr);
  if (last_rc != IBFT_OK) return false;
  unpack_mask(ms, sc.n, sender);
  unpack_mask(mv, sc.n, closure);
  return true;
}

bool GpuBackend::VerifySendersWire(const uint8_t *wire, const uint32_t *off, size_t n, std::vector<uint8_t> &verdict,
                                   WireStats *stats) {
  verdict.assign(n, 0);
  if (n == 0) return true;
  std::vector<uint64_t> mask((n + 63) / 64, 0);
  std::vector<ibft_wire_row_t> rows(n);
  last_rc = ibft_verify_senders_wire(ctx_, wire, off, n, mask.data(), rows.data(), nullptr);
  if (last_rc != IBFT_OK) return false;
  unpack_mask(mask, n, verdict);
  // the rows the device would not vouch for: stock route
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<size_t> idx;
  std::vector<MsgPtr> msgs;
  for (size_t i = 0; i < n; i++) {
    if (rows[i].status == IBFT_WIRE_OK) continue;
    auto m = std::make_shared<IbftMessage>();
    if (!decode(wire + off[i], off[i + 1] - off[i], *m)) continue;  // proto.Unmarshal error: dropped
    idx.push_back(i);
    msgs.push_back(std::move(m));
  }
  if (stats) {
    stats->host_rows = idx.size();
    stats->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  if (msgs.empty()) return true;
  std::vector<uint8_t> v2;
  if (!VerifySenderBatch(msgs, v2)) return false;
  for (size_t j = 0; j < idx.size(); j++) verdict[idx[j]] = v2[j];
  return true;
}

bool GpuBackend::VerifyMessagesWire(const uint8_t *wire, const uint32_t *off, size_t n, uint64_t height, uint64_t round,
                                    const Proposal &proposal, std::vector<uint8_t> &sender, std::vector<uint8_t> &closure,
                                    std::vector<uint8_t> &judged) {
  sender.assign(n, 0);
  closure.assign(n, 0);
  judged.assign(n, 0);
  if (n == 0) return true;
  std::vector<uint64_t> ms((n + 63) / 64, 0), mv((n + 63) / 64, 0);
  std::vector<uint8_t> cls(n, 0);
  last_rc = ibft_verify_messages_wire(ctx_, wire, off, n, height, round, (const uint8_t *)proposal.raw_proposal.data(),
                                      proposal.raw_proposal.size(), proposal.round, nullptr, ms.data(), mv.data(), cls.data(),
                                      nullptr, nullptr);
  if (last_rc != IBFT_OK) return false;
  unpack_mask(ms, n, sender);
  unpack_mask(mv, n, closure);
  std::vector<size_t> idx;
  std::vector<MsgPtr> msgs;
  for (size_t i = 0; i < n; i++) {
    if (!(cls[i] & IBFT_WIRE_CLASS_NEEDS_HOST)) {
      // the closure of a PREPARE / COMMIT of this view is settled by the device, whatever the verdict
      judged[i] = (cls[i] & IBFT_WIRE_CLASS_CLOSURE) != 0;
      continue;
    }
    auto m = std::make_shared<IbftMessage>();
    if (!decode(wire + off[i], off[i + 1] - off[i], *m)) continue;  // proto.Unmarshal error: dropped
    idx.push_back(i);
    msgs.push_back(std::move(m));
  }
  if (msgs.empty()) return true;
  std::vector<uint8_t> v2;
  if (!VerifySenderBatch(msgs, v2)) return false;
  for (size_t j = 0; j < idx.size(); j++) sender[idx[j]] = v2[j];
  return true;
}

// the messages nested directly inside m, in the order the device lists them (wire order): the RoundChangeCertificate's
// messages of a PREPREPARE payload; proposalMessage, then prepareMessages, of a ROUND_CHANGE payload's PreparedCertificate
static void nested_messages(const IbftMessage &m, std::vector<MsgPtr> &out) {
  out.clear();
  if (m.kind == PayloadKind::PREPREPARE && m.preprepare.certificate) {
    out = m.preprepare.certificate->round_change_messages;
  } else if (m.kind == PayloadKind::ROUND_CHANGE && m.round_change.latest_prepared_certificate) {
    const PreparedCertificate &pc = *m.round_change.latest_prepared_certificate;
    if (pc.proposal_message) out.push_back(pc.proposal_message);
    out.insert(out.end(), pc.prepare_messages.begin(), pc.prepare_messages.end());
  }
}
// the proposal hash a message carries, by payload kind (what the device reports in ibft_wire_row_t.proposal_hash)
static const bytes *carried_hash(const IbftMessage &m) {
  switch (m.kind) {
    case PayloadKind::PREPREPARE: return &m.preprepare.proposal_hash;
    case PayloadKind::PREPARE: return &m.prepare.proposal_hash;
    case PayloadKind::COMMIT: return &m.commit.proposal_hash;
    default: return nullptr;
  }
}

bool GpuBackend::VerifyCertificatesWire(const uint8_t *wire, const uint32_t *off, size_t n, CertVerdicts &out) {
  out = CertVerdicts();
  if (n == 0) return true;
  const size_t cap = cert_rows_cap, words = (cap + 63) / 64;
  out.nodes.resize(cap);
  out.cls.assign(cap, 0);
  std::vector<uint64_t> ms(words, 0), mh(words, 0), mself(words, 0);
  size_t rows = 0;
  last_rc = ibft_verify_certificates_wire(ctx_, wire, off, n, cap, &rows, out.nodes.data(), nullptr, out.cls.data(), ms.data(),
                                          mh.data(), mself.data());
  if (last_rc != IBFT_OK) return false;  // IBFT_E_TOOBIG included: the caller's stock route handles the batch
  out.n_rows = rows;
  out.nodes.resize(rows);
  out.cls.resize(rows);
  unpack_mask(ms, rows, out.sender);
  unpack_mask(mh, rows, out.hash);
  unpack_mask(mself, rows, out.self);
  return true;
}

bool LoopBatch::VerifyCertificatesWire(const uint8_t *wire, const uint32_t *off, size_t n, CertVerdicts &out) {
  out = CertVerdicts();
  if (fail_certs) return false;
  calls++;
  cert_calls++;
  struct Item {
    MsgPtr m;  // null: did not decode
    const IbftMessage *parent;
  };
  std::vector<Item> items;
  for (size_t i = 0; i < n; i++) {
    auto m = std::make_shared<IbftMessage>();
    const bool ok = decode(wire + off[i], off[i + 1] - off[i], *m);
    ibft_cert_node_t nd{};
    nd.off = off[i];
    nd.len = off[i + 1] - off[i];
    nd.parent = IBFT_CERT_NO_PARENT;
    nd.ordinal = (uint32_t)i;
    out.nodes.push_back(nd);
    items.push_back(Item{ok ? m : nullptr, nullptr});
  }
  std::vector<MsgPtr> kids;
  for (size_t lo = 0, hi = items.size(); lo < hi;) {  // one level per turn
    size_t base = hi;
    for (size_t r = lo; r < hi; r++) {
      out.nodes[r].first_child = (uint32_t)base;
      if (!items[r].m) continue;
      nested_messages(*items[r].m, kids);
      out.nodes[r].n_children = (uint32_t)kids.size();
      const bool pc = items[r].m->kind == PayloadKind::ROUND_CHANGE;
      for (size_t k = 0; k < kids.size(); k++) {
        ibft_cert_node_t c{};
        c.parent = (uint32_t)r;
        c.ordinal = (uint32_t)k;
        c.level = (uint8_t)(out.nodes[r].level + 1);
        c.role = (uint8_t)(!pc ? IBFT_CERT_ROLE_RCC_MESSAGE
                               : (k == 0 && items[r].m->round_change.latest_prepared_certificate->proposal_message
                                      ? IBFT_CERT_ROLE_PC_PROPOSAL
                                      : IBFT_CERT_ROLE_PC_PREPARE));
        out.nodes.push_back(c);
        items.push_back(Item{kids[k], items[r].m.get()});
      }
      base += kids.size();
    }
    lo = hi;
    hi = items.size();
  }
  const size_t rows = items.size();
  out.n_rows = rows;
  out.cls.assign(rows, 0);
  out.sender.assign(rows, 0);
  out.hash.assign(rows, 0);
  out.self.assign(rows, 0);
  for (size_t r = 0; r < rows; r++) {
    if (!items[r].m) {
      out.cls[r] = IBFT_CERT_CLASS_NEEDS_HOST;
      continue;
    }
    const IbftMessage &m = *items[r].m;
    out.sender[r] = v_->IsValidValidator(m);
    const bytes *h = carried_hash(m);
    const IbftMessage *p = items[r].parent;
    if (h && p && p->kind == PayloadKind::ROUND_CHANGE && p->round_change.last_prepared_proposal)
      out.hash[r] = v_->IsValidProposalHash(&*p->round_change.last_prepared_proposal, h);
    if (m.kind == PayloadKind::PREPREPARE && m.preprepare.proposal)
      out.self[r] = v_->IsValidProposalHash(&*m.preprepare.proposal, &m.preprepare.proposal_hash);
  }
  return true;
}

bool LoopBatch::VerifyPrepareBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &v) {
  if (fail_hashes) return false;
  calls++;
  v.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) v[i] = v_->IsValidProposalHash(proposal, extract_prepare_hash(*msgs[i]));
  return true;
}
bool LoopBatch::VerifyCommitBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &v) {
  if (fail_hashes || fail_seals) return false;
  calls++;
  v.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) {
    const bytes *h = extract_commit_hash(*msgs[i]);
    std::optional<CommittedSeal> seal = extract_committed_seal(*msgs[i]);
    v[i] = v_->IsValidProposalHash(proposal, h) && v_->IsValidCommittedSeal(h, seal ? &*seal : nullptr);
  }
  return true;
}
bool LoopBatch::VerifySenderBatch(const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &v) {
  if (fail_senders) return false;
  calls++;
  v.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) v[i] = v_->IsValidValidator(*msgs[i]);
  return true;
}

bool LoopBatch::VerifyMessageSet(const Proposal *proposal, MessageType type, const std::vector<MsgPtr> &msgs,
                                 std::vector<uint8_t> &sender, std::vector<uint8_t> &closure) {
  if (fail_sets || !proposal || (type != PREPARE && type != COMMIT)) return false;
  calls++;
  set_calls++;
  sender.assign(msgs.size(), 0);
  closure.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) {
    sender[i] = v_->IsValidValidator(*msgs[i]);
    if (type == PREPARE) {
      closure[i] = v_->IsValidProposalHash(proposal, extract_prepare_hash(*msgs[i]));
    } else {
      const bytes *h = extract_commit_hash(*msgs[i]);
      std::optional<CommittedSeal> seal = extract_committed_seal(*msgs[i]);
      closure[i] = v_->IsValidProposalHash(proposal, h) && v_->IsValidCommittedSeal(h, seal ? &*seal : nullptr);
    }
  }
  return true;
}

bool HotPath::isAcceptableMessage(const IbftMessage &m) {
  if (!verifier || !verifier->IsValidValidator(m)) return false;  // ibft.go:1128
  if (!m.view) return false;                                       // :1133
  if (height > m.view->height) return false;                       // :1139
  if (height == m.view->height) return m.view->round >= round;     // :1144
  return true;
}

bool HotPath::hasQuorumByMsgType(const std::vector<MsgPtr> &msgs, uint32_t type) {
  switch (type) {
    case PREPREPARE: return msgs.size() >= 1;
    case PREPARE: return validatorManager.HasPrepareQuorum(proposalMessage.get(), msgs);
    case ROUND_CHANGE:
    case COMMIT: return validatorManager.HasQuorum(convertMessageToAddressSet(msgs));
    default: return false;
  }
}

int HotPath::AddMessage(MsgPtr m) {
  if (!m) return 0;
  if (!isAcceptableMessage(*m)) return 0;
  View view = *m->view;
  uint32_t type = m->type;
  messages.AddMessage(m);
  if (view.height == height) {  // ibft.go:1113-1120
    auto msgs = messages.GetValidMessages(view, (MessageType)type, [](const IbftMessage &) { return true; });
    if (hasQuorumByMsgType(msgs, type)) return 2;
  }
  return 1;
}

void HotPath::EnableQuorumIndex() {
  index_enabled_ = true;
  messages.SetHooks(
      [this](uint32_t type, uint64_t h, uint64_t r, const bytes &from, int delta) {
        quorumIndex.OnSender(type, h, r, from, delta, validatorManager);
      },
      [this](uint64_t below) {
        quorumIndex.OnPrune(below);
        PruneVerdictCache(below);
      });
}

void HotPath::PruneVerdictCache(uint64_t below_height) {
  for (auto it = verdict_cache_.begin(); it != verdict_cache_.end();)
    it = it->second.height < below_height ? verdict_cache_.erase(it) : std::next(it);
  for (auto it = closure_cache_.begin(); it != closure_cache_.end();) {
    const IbftMessage &m = *it->second.keep;
    it = (!m.view || m.view->height < below_height) ? closure_cache_.erase(it) : std::next(it);
  }
  for (auto it = cert_roots_.begin(); it != cert_roots_.end();) {
    if (it->second.height >= below_height) {
      ++it;
      continue;
    }
    for (const IbftMessage *k : it->second.senders) cert_sender_.erase(k);
    for (const auto &k : it->second.hashes) cert_hash_.erase(k);
    it = cert_roots_.erase(it);
  }
}

// The verdicts of one root row of a certificate call and of everything below it go into the arrival-time tables,
// matched to the decoded objects by position: a decoded message lists its nested messages in the order the device lists
// them.  A subtree whose row count differs from the decoded count (the device refused the wrapper as non-canonical) is left
// to the stock route.
void HotPath::noteCertificateTree(const CertVerdicts &cv, size_t row, const MsgPtr &root) {
  CertRoot &cr = cert_roots_[root.get()];
  cr.keep = root;
  cr.height = root->view ? root->view->height : 0;
  std::vector<std::pair<size_t, MsgPtr>> todo{{row, root}};
  std::vector<MsgPtr> kids;
  while (!todo.empty()) {
    const size_t r = todo.back().first;
    const MsgPtr m = todo.back().second;
    todo.pop_back();
    const uint8_t cls = cv.cls[r];
    // a PREPREPARE's own (proposal, proposalHash): validateProposalCommon's IsValidProposalHash
    if (!(cls & (IBFT_CERT_CLASS_NEEDS_HOST | IBFT_CERT_CLASS_PROPOSAL_BY_HOST))) {
      const Proposal *own = extract_proposal(*m);
      const bytes *oh = extract_proposal_hash(*m);
      if (own && oh) {
        cert_hash_[{own, oh}] = cv.self[r] != 0;
        cr.hashes.push_back({own, oh});
      }
    }
    nested_messages(*m, kids);
    const ibft_cert_node_t &nd = cv.nodes[r];
    if (nd.n_children != kids.size() || (size_t)nd.first_child + nd.n_children > cv.n_rows) continue;
    const Proposal *last = extract_last_prepared_proposal(*m);
    const bool hashes_decided = last && !(cls & (IBFT_CERT_CLASS_NEEDS_HOST | IBFT_CERT_CLASS_PROPOSAL_BY_HOST));
    for (size_t k = 0; k < kids.size(); k++) {
      const size_t c = nd.first_child + k;
      if (!kids[k]) continue;
      if (cv.cls[c] == 0) {
        cert_sender_[kids[k].get()] = cv.sender[c] != 0;
        cr.senders.push_back(kids[k].get());
        cert_rows++;
      }
      if (hashes_decided && !(cv.cls[c] & IBFT_CERT_CLASS_NEEDS_HOST)) {
        // the very pointers proposalMatchesCertificate will ask about (nil when type and payload disagree: not cached)
        const bytes *h = cv.nodes[c].role == IBFT_CERT_ROLE_PC_PROPOSAL ? extract_proposal_hash(*kids[k]) : extract_prepare_hash(*kids[k]);
        if (h) {
          cert_hash_[{last, h}] = cv.hash[c] != 0;
          cr.hashes.push_back({last, h});
        }
      }
      todo.push_back({c, kids[k]});
    }
  }
}

bool HotPath::lookupHashVerdict(const Proposal *proposal, const bytes *hash, bool &ok) const {
  auto it = hash_verdict_.find({proposal, hash});
  if (it != hash_verdict_.end()) {
    ok = it->second;
    return true;
  }
  auto jt = cert_hash_.find({proposal, hash});
  if (jt != cert_hash_.end()) {
    ok = jt->second;
    return true;
  }
  return false;
}

// IBFT.AddMessage with IsValidValidator already answered (by the device batch or the cache)
int HotPath::addWithVerdict(MsgPtr m, bool sender_ok) {
  struct TableVerifier : Verifier {
    Verifier *inner;
    bool verdict;
    bool IsValidProposalHash(const Proposal *p, const bytes *h) override { return inner->IsValidProposalHash(p, h); }
    bool IsValidCommittedSeal(const bytes *h, const CommittedSeal *s) override { return inner->IsValidCommittedSeal(h, s); }
    bool IsValidValidator(const IbftMessage &) override { return verdict; }
    bool IsProposer(const bytes &id, uint64_t hh, uint64_t rr) override { return inner->IsProposer(id, hh, rr); }
    bool IsValidProposal(const bytes &raw) override { return inner->IsValidProposal(raw); }
    bytes ID() override { return inner->ID(); }
  } tv;
  tv.inner = verifier;
  tv.verdict = sender_ok;
  Verifier *saved = verifier;
  verifier = &tv;
  const int rc = index_enabled_ ? AddMessageFast(std::move(m)) : AddMessage(std::move(m));
  verifier = saved;
  return rc;
}

void HotPath::syncClosureKey(const Proposal *proposal) {
  bytes key;
  if (proposal) {
    key = proposal->raw_proposal;
    for (int i = 7; i >= 0; i--) key.push_back((char)(proposal->round >> (8 * i)));
  }
  if (key != closure_key_) {  // another proposal (or none): what the table says no longer applies
    closure_key_ = std::move(key);
    closure_cache_.clear();
    closure_epoch_++;
  }
}

bool HotPath::IngestWire(const std::vector<bytes> &raw, std::vector<int> &results, IngestStats *stats) {
  results.assign(raw.size(), -1);
  IngestStats st;
  const Proposal *proposal = getProposal();
  syncClosureKey(proposal);
  std::vector<MsgPtr> msgs(raw.size());
  std::vector<int> verdict(raw.size(), -1);  // −1 unknown, 0 / 1 decided
  std::vector<int> closure(raw.size(), -1);  // the handle* closure where it is already known
  std::vector<size_t> ask;                   // rows the device has to judge: first occurrence of each distinct message
  std::map<bytes, size_t> first_in_batch;
  for (size_t i = 0; i < raw.size(); i++) {
    auto m = std::make_shared<IbftMessage>();
    if (!decode((const uint8_t *)raw[i].data(), raw[i].size(), *m)) continue;  // proto.Unmarshal error: dropped
    msgs[i] = std::move(m);
    auto hit = verdict_cache_.find(raw[i]);
    if (hit != verdict_cache_.end()) {
      verdict[i] = hit->second.ok ? 1 : 0;
      if (hit->second.carrier) msgs[i] = hit->second.carrier;  // the object the certificate tables know
      if (hit->second.closure >= 0 && hit->second.closure_epoch == closure_epoch_) closure[i] = hit->second.closure;
      st.cache_hits++;
    } else if (first_in_batch.emplace(raw[i], i).second) {
      ask.push_back(i);
    }
  }
  const std::vector<size_t> asked = ask;  // every distinct undecided message of the batch (cached at the end)
  // (0) messages that carry certificates: the whole tree — their own envelope and every message nested in them — in ONE
  // device call, from the bytes as they arrived
  if (use_batch && batch && use_certs) {
    std::vector<size_t> carriers;
    for (size_t i : ask)
      if (msgs[i]->kind == PayloadKind::PREPREPARE || msgs[i]->kind == PayloadKind::ROUND_CHANGE) carriers.push_back(i);
    if (!carriers.empty()) {
      bytes wire;
      std::vector<uint32_t> off{0};
      for (size_t i : carriers) {
        wire += raw[i];
        off.push_back((uint32_t)wire.size());
      }
      CertVerdicts cv;
      if (batch->VerifyCertificatesWire((const uint8_t *)wire.data(), off.data(), carriers.size(), cv) && cv.n_rows >= carriers.size()) {
        st.device_calls++;
        cert_calls++;
        for (size_t j = 0; j < carriers.size(); j++) {
          if (cv.cls[j] == 0) {
            verdict[carriers[j]] = cv.sender[j] ? 1 : 0;
            cert_rows++;
          }
          noteCertificateTree(cv, j, msgs[carriers[j]]);
        }
        std::vector<size_t> left;
        for (size_t i : ask)
          if (verdict[i] < 0) left.push_back(i);
        ask.swap(left);
      }
    }
  }
  // (1) messages of the current view with the proposal at hand: judged completely, one set call per type
  std::vector<size_t> rest;
  GpuBackend *gpu_sets = (use_batch && use_sets && proposal) ? dynamic_cast<GpuBackend *>(batch) : nullptr;
  bool wire_sets_done = false;
  if (gpu_sets && !ask.empty()) {
    // the device walks the bytes AND judges every PREPARE / COMMIT of this view completely: one call for the micro-batch
    bytes wire;
    std::vector<uint32_t> off{0};
    for (size_t i : ask) {
      wire += raw[i];
      off.push_back((uint32_t)wire.size());
    }
    std::vector<uint8_t> vs, vc, judged;
    if (gpu_sets->VerifyMessagesWire((const uint8_t *)wire.data(), off.data(), ask.size(), height, round, *proposal, vs, vc,
                                     judged)) {
      st.device_calls++;
      for (size_t j = 0; j < ask.size(); j++) {
        verdict[ask[j]] = vs[j] ? 1 : 0;
        if (judged[j]) {
          closure[ask[j]] = vc[j] ? 1 : 0;
          st.set_rows++;
        }
      }
      wire_sets_done = true;
    }
  }
  if (wire_sets_done) {
    // nothing left
  } else if (use_batch && batch && use_sets && proposal && !gpu_sets) {
    std::vector<size_t> of_type[2];
    for (size_t i : ask) {
      const IbftMessage &m = *msgs[i];
      const bool here = m.view && m.view->height == height && m.view->round == round;
      if (here && (m.type == PREPARE || m.type == COMMIT))
        of_type[m.type == COMMIT].push_back(i);
      else
        rest.push_back(i);
    }
    for (int t = 0; t < 2; t++) {
      if (of_type[t].empty()) continue;
      std::vector<MsgPtr> sub;
      for (size_t i : of_type[t]) sub.push_back(msgs[i]);
      std::vector<uint8_t> vs, vc;
      if (batch->VerifyMessageSet(proposal, t ? COMMIT : PREPARE, sub, vs, vc) && vs.size() == sub.size() &&
          vc.size() == sub.size()) {
        st.device_calls++;
        st.set_rows += sub.size();
        for (size_t k = 0; k < sub.size(); k++) {
          verdict[of_type[t][k]] = vs[k] ? 1 : 0;
          closure[of_type[t][k]] = vc[k] ? 1 : 0;
        }
      } else {  // not offered, or the device call failed: these rows take the sender route below
        rest.insert(rest.end(), of_type[t].begin(), of_type[t].end());
      }
    }
    std::sort(rest.begin(), rest.end());
  } else {
    rest = ask;
  }
  // (2) everything else: IsValidValidator only
  if (!rest.empty()) {
    std::vector<uint8_t> v;
    bool ok = false;
    if (use_batch && batch) {
      st.device_calls++;
      if (auto *gpu = dynamic_cast<GpuBackend *>(batch)) {  // the device walks the bytes themselves (§8f rank 3)
        bytes wire;
        std::vector<uint32_t> off{0};
        for (size_t i : rest) {
          wire += raw[i];
          off.push_back((uint32_t)wire.size());
        }
        ok = gpu->VerifySendersWire((const uint8_t *)wire.data(), off.data(), rest.size(), v);
      } else {
        std::vector<MsgPtr> sub;
        for (size_t i : rest) sub.push_back(msgs[i]);
        ok = batch->VerifySenderBatch(sub, v);
      }
      ok = ok && v.size() == rest.size();
    }
    if (!ok) {  // no batch backend, or the device call failed: the per-message verifier
      if (use_batch && batch) fallbacks++;
      v.assign(rest.size(), 0);
      for (size_t j = 0; j < rest.size(); j++) v[j] = verifier && verifier->IsValidValidator(*msgs[rest[j]]);
    }
    for (size_t j = 0; j < rest.size(); j++) verdict[rest[j]] = v[j] ? 1 : 0;
  }
  st.device_rows = asked.size();
  for (size_t i : asked) {
    CachedVerdict cv{verdict[i] == 1, msgs[i]->view ? msgs[i]->view->height : 0};
    cv.closure = closure[i];
    cv.closure_epoch = closure_epoch_;
    if (cert_roots_.count(msgs[i].get())) cv.carrier = msgs[i];
    verdict_cache_[raw[i]] = cv;
  }
  for (size_t i = 0; i < raw.size(); i++) {
    if (!msgs[i]) continue;
    if (verdict[i] < 0) {  // a repeat inside this batch
      const size_t first = first_in_batch[raw[i]];
      verdict[i] = verdict[first];
      closure[i] = closure[first];
      if (cert_roots_.count(msgs[first].get())) msgs[i] = msgs[first];
    }
    if (closure[i] >= 0 && verdict[i] == 1) closure_cache_[msgs[i].get()] = ClosureVerdict{msgs[i], closure[i] == 1};
    results[i] = addWithVerdict(msgs[i], verdict[i] == 1);
  }
  if (stats) *stats = st;
  return true;
}


int HotPath::AddMessageFast(MsgPtr m) {
  if (!m) return 0;
  if (!isAcceptableMessage(*m)) return 0;
  const View view = *m->view;
  const uint32_t type = m->type;
  messages.AddMessage(m);
  if (view.height != height) return 1;
  // hasQuorumByMsgType over the stored (unverified) messages of the view, without walking them
  auto rebuild = [&]() {
    std::vector<bytes> senders;
    for (auto &x : messages.GetValidMessages(view, (MessageType)type, [](const IbftMessage &) { return true; }))
      senders.push_back(x->from);
    return senders;
  };
  auto pc = quorumIndex.Get(type, view.height, view.round, rebuild, validatorManager);
  bool q = false;
  switch (type) {
    case PREPREPARE: q = pc.second >= 1; break;
    case PREPARE: {  // HasPrepareQuorum: proposer joins the set; a PREPARE from the proposer voids it
      if (!proposalMessage) break;
      if (messages.Has(view, PREPARE, proposalMessage->from)) break;
      auto p = validatorManager.powers().find(proposalMessage->from);
      unsigned __int128 w = p == validatorManager.powers().end() ? 0 : p->second;
      q = validatorManager.initialized() && pc.first + w >= validatorManager.quorum();
      break;
    }
    case ROUND_CHANGE:
    case COMMIT: q = validatorManager.initialized() && pc.first >= validatorManager.quorum(); break;
    default: break;
  }
  return q ? 2 : 1;
}

// The verdicts of a handle* walk in batch mode: what arrived through IngestWire's set calls is already in the table;
// the rest goes to the device in one batch; if that fails, the per-message closure answers, same lock held
//  — never "nothing".
std::vector<uint8_t> HotPath::closureVerdicts(const Proposal *proposal, MessageType type, const std::vector<MsgPtr> &all) {
  syncClosureKey(proposal);
  std::vector<uint8_t> v(all.size(), 0);
  std::vector<MsgPtr> rest;
  std::vector<size_t> rest_idx;
  for (size_t k = 0; k < all.size(); k++) {
    auto it = use_sets ? closure_cache_.find(all[k].get()) : closure_cache_.end();
    if (it != closure_cache_.end()) {
      v[k] = it->second.ok;
      closure_hits++;
    } else {
      rest.push_back(all[k]);
      rest_idx.push_back(k);
    }
  }
  if (rest.empty()) return v;
  std::vector<uint8_t> vr;
  const bool ok = (type == PREPARE ? batch->VerifyPrepareBatch(proposal, rest, vr) : batch->VerifyCommitBatch(proposal, rest, vr)) &&
                  vr.size() == rest.size();
  if (!ok) {
    fallbacks++;
    vr.assign(rest.size(), 0);
    for (size_t k = 0; k < rest.size(); k++) {
      if (type == PREPARE) {
        vr[k] = verifier->IsValidProposalHash(proposal, extract_prepare_hash(*rest[k]));
      } else {
        const bytes *proposalHash = extract_commit_hash(*rest[k]);
        std::optional<CommittedSeal> seal = extract_committed_seal(*rest[k]);
        vr[k] = verifier->IsValidProposalHash(proposal, proposalHash) &&
                verifier->IsValidCommittedSeal(proposalHash, seal ? &*seal : nullptr);
      }
    }
  }
  for (size_t k = 0; k < rest.size(); k++) v[rest_idx[k]] = vr[k];
  return v;
}

bool HotPath::handlePrepare(const View &view) {
  std::vector<MsgPtr> prepareMessages;
  const Proposal *proposal = getProposal();
  closure_hits = 0;
  if (use_batch && batch) {
    prepareMessages = messages.GetValidMessagesBatch(
        view, PREPARE, [&](const std::vector<MsgPtr> &all) { return closureVerdicts(proposal, PREPARE, all); });
  } else {
    prepareMessages = messages.GetValidMessages(view, PREPARE, [&](const IbftMessage &m) {
      return verifier->IsValidProposalHash(proposal, extract_prepare_hash(m));
    });
  }
  if (!hasQuorumByMsgType(prepareMessages, PREPARE)) return false;
  // sendCommitMessage(view) is the Go side's business; finalizePrepare: state.go
  preparedMessages = prepareMessages;
  stateName = StateName::commit;
  return true;
}

bool HotPath::handleCommit(const View &view) {
  std::vector<MsgPtr> commitMessages;
  const Proposal *proposal = getProposal();
  closure_hits = 0;
  if (use_batch && batch) {
    commitMessages = messages.GetValidMessagesBatch(
        view, COMMIT, [&](const std::vector<MsgPtr> &all) { return closureVerdicts(proposal, COMMIT, all); });
  } else {
    commitMessages = messages.GetValidMessages(view, COMMIT, [&](const IbftMessage &m) {
      const bytes *proposalHash = extract_commit_hash(m);
      std::optional<CommittedSeal> seal = extract_committed_seal(m);
      if (!verifier->IsValidProposalHash(proposal, proposalHash)) return false;
      return verifier->IsValidCommittedSeal(proposalHash, seal ? &*seal : nullptr);
    });
  }
  if (!hasQuorumByMsgType(commitMessages, COMMIT)) return false;
  std::vector<std::optional<CommittedSeal>> seals;
  if (!extract_committed_seals(commitMessages, seals)) return false;  // safe check, ibft.go:952-958
  committedSeals = std::move(seals);
  stateName = StateName::fin;
  return true;
}

}  // namespace ibft

// ---- certificate checks ------------------------------------------------------------------------
namespace ibft {

bool HotPath::isValidValidatorCached(const IbftMessage &m) {
  auto it = sender_verdict_.find(&m);
  if (it != sender_verdict_.end()) return it->second;
  auto jt = cert_sender_.find(&m);  // judged when the message that carries it arrived (IngestWire, use_certs)
  if (jt != cert_sender_.end()) return jt->second;
  return verifier->IsValidValidator(m);
}

void HotPath::prefetchSenders(const std::vector<const IbftMessage *> &all) {
  sender_verdict_.clear();
  last_cert_senders = 0;
  cert_hits = 0;
  std::vector<const IbftMessage *> msgs;  // what the arrival-time tables cannot answer
  for (const IbftMessage *m : all) {
    if (cert_sender_.count(m))
      cert_hits++;
    else
      msgs.push_back(m);
  }
  if (!(use_batch && batch) || msgs.empty()) return;
  std::vector<MsgPtr> owned;
  owned.reserve(msgs.size());
  for (const IbftMessage *m : msgs) owned.push_back(MsgPtr(MsgPtr(), const_cast<IbftMessage *>(m)));  // non-owning alias
  std::vector<uint8_t> v;
  if (!batch->VerifySenderBatch(owned, v) || v.size() != msgs.size()) return;  // device unavailable: stock path
  for (size_t i = 0; i < msgs.size(); i++) sender_verdict_[msgs[i]] = v[i] != 0;
  last_cert_senders = msgs.size();
}

static void collect_pc(const PreparedCertificate *pc, std::vector<const IbftMessage *> &out) {
  if (!pc) return;
  if (pc->proposal_message) out.push_back(pc->proposal_message.get());
  for (auto &m : pc->prepare_messages)
    if (m) out.push_back(m.get());
}

bool HotPath::validPC(const PreparedCertificate *certificate, uint64_t roundLimit, uint64_t height) {
  std::vector<const IbftMessage *> need;
  collect_pc(certificate, need);
  prefetchSenders(need);
  bool ok = validPCImpl(certificate, roundLimit, height);
  sender_verdict_.clear();
  return ok;
}

bool HotPath::validPCImpl(const PreparedCertificate *certificate, uint64_t roundLimit, uint64_t height) {
  if (!certificate) return true;  // PCs that are not set are valid by default
  // "either both the proposal message and the prepare messages are set together": a repeated
  // field that is empty on the wire decodes to nil, so empty == nil here
  if (!certificate->proposal_message || certificate->prepare_messages.empty()) return false;
  std::vector<MsgPtr> all;
  all.push_back(certificate->proposal_message);
  for (auto &m : certificate->prepare_messages) all.push_back(m);
  if (!validatorManager.HasQuorum(convertMessageToAddressSet(all))) return false;
  if (certificate->proposal_message->type != PREPREPARE) return false;
  for (auto &m : certificate->prepare_messages)
    if (m->type != PREPARE) return false;
  if (!are_valid_pc_messages(all, height, roundLimit)) return false;
  const IbftMessage &proposal = *certificate->proposal_message;
  if (!verifier->IsProposer(proposal.from, proposal.view->height, proposal.view->round)) return false;
  if (!isValidValidatorCached(proposal)) return false;
  for (auto &m : certificate->prepare_messages) {
    if (!isValidValidatorCached(*m)) return false;
    if (verifier->IsProposer(m->from, m->view->height, m->view->round)) return false;
  }
  return true;
}

bool HotPath::proposalMatchesCertificate(const Proposal *proposal, const PreparedCertificate *certificate) {
  if (!proposal && !certificate) return true;
  if (!certificate) return false;
  std::vector<const bytes *> hashes;
  // ExtractProposalHash on a nil message would panic in the reference; treat as a nil hash
  hashes.push_back(certificate->proposal_message ? extract_proposal_hash(*certificate->proposal_message) : nullptr);
  for (auto &m : certificate->prepare_messages) hashes.push_back(m ? extract_prepare_hash(*m) : nullptr);
  if (!hash_verdict_.empty() || !cert_hash_.empty()) {  // answered by handleRoundChangeMessage's pre-pass, or on arrival
    bool all_known = true, all_ok = true;
    for (const bytes *h : hashes) {
      bool ok = false;
      all_known = all_known && lookupHashVerdict(proposal, h, ok);
      all_ok = all_ok && ok;
    }
    if (all_known) return all_ok;
  }
  last_cert_hashes = 0;
  if (use_batch && batch && proposal) {
    // one hash batch: reuse VerifyPrepareBatch's column path through synthetic PREPARE messages
    std::vector<MsgPtr> synth;
    for (const bytes *h : hashes) {
      auto m = std::make_shared<IbftMessage>();
      m->type = PREPARE;
      if (h) {
        m->kind = PayloadKind::PREPARE;
        m->prepare.proposal_hash = *h;
      }
      synth.push_back(std::move(m));
    }
    std::vector<uint8_t> v;
    if (batch->VerifyPrepareBatch(proposal, synth, v) && v.size() == synth.size()) {
      last_cert_hashes = synth.size();
      for (uint8_t ok : v)
        if (!ok) return false;
      return true;
    }
  }
  for (const bytes *h : hashes)
    if (!verifier->IsValidProposalHash(proposal, h)) return false;
  return true;
}

bool HotPath::validateProposalCommon(const IbftMessage &msg, const View &view) {
  const Proposal *proposal = extract_proposal(msg);
  const bytes *proposalHash = extract_proposal_hash(msg);
  if (!proposal) return false;  // the reference dereferences it; a nil proposal cannot be valid
  if (proposal->round != view.round) return false;
  if (!verifier->IsProposer(msg.from, view.height, view.round)) return false;
  if (!isValidProposalHashCached(proposal, proposalHash)) return false;
  return verifier->IsValidProposal(proposal->raw_proposal);
}

bool HotPath::validateProposal0(const IbftMessage &msg, const View &view) {
  if (!msg.view || msg.view->round != 0) return false;
  if (!validateProposalCommon(msg, view)) return false;
  if (verifier->IsProposer(verifier->ID(), view.height, view.round)) return false;
  return true;
}

MsgPtr HotPath::handlePrePrepare(const View &view) {
  std::vector<MsgPtr> msgs = messages.GetValidMessages(view, PREPREPARE, [&](const IbftMessage &m) {
    return view.round == 0 ? validateProposal0(m, view) : validateProposal(m, view);
  });
  return msgs.empty() ? nullptr : msgs[0];
}

bool HotPath::validateProposal(const IbftMessage &msg, const View &view) {
  const uint64_t height = view.height, round = view.round;
  const Proposal *proposal = extract_proposal(msg);
  const RoundChangeCertificate *rcc = extract_round_change_certificate(msg);
  if (!validateProposalCommon(msg, view)) return false;
  if (!rcc) return false;
  std::vector<MsgPtr> rcs;
  for (auto &m : rcc->round_change_messages)
    if (m) rcs.push_back(m);
  if (!has_unique_senders(rcs)) return false;
  if (!hasQuorumByMsgType(rcs, ROUND_CHANGE)) return false;
  if (verifier->IsProposer(verifier->ID(), height, round)) return false;
  // batch pre-pass: every signature this walk can ask about — the RC envelopes and the messages
  // of their prepared certificates (worst case O(N²) signatures per round change)
  {
    std::vector<const IbftMessage *> need;
    for (auto &rc : rcs) {
      need.push_back(rc.get());
      collect_pc(extract_latest_pc(*rc), need);
    }
    prefetchSenders(need);
  }
  for (auto &rc : rcs) {
    if (rc->type != ROUND_CHANGE) return false;
    if (!rc->view || rc->view->height != height) return false;
    if (rc->view->round != round) return false;
    if (!isValidValidatorCached(*rc)) return false;
  }
  struct RoundHash {
    uint64_t round;
    const bytes *hash;
  };
  std::vector<RoundHash> tuples;
  for (auto &rc : rcs) {
    const PreparedCertificate *cert = extract_latest_pc(*rc);
    if (cert && msg.view && validPCImpl(cert, msg.view->round, height)) {
      tuples.push_back({cert->proposal_message->view->round, extract_proposal_hash(*cert->proposal_message)});
    }
  }
  sender_verdict_.clear();
  if (tuples.empty()) return true;
  uint64_t maxRound = 0;
  const bytes *expected = nullptr;
  for (auto &t : tuples)
    if (t.round >= maxRound) {
      maxRound = t.round;
      expected = t.hash;
    }
  Proposal p2;
  p2.raw_proposal = proposal->raw_proposal;
  p2.round = maxRound;
  return verifier->IsValidProposalHash(&p2, expected);
}

}  // namespace ibft

// ---- handleRoundChangeMessage (core/ibft.go:470-512) ------------------------------------------------
namespace ibft {

bool HotPath::isValidProposalHashCached(const Proposal *proposal, const bytes *hash) {
  bool ok = false;
  if (lookupHashVerdict(proposal, hash, ok)) return ok;
  return verifier->IsValidProposalHash(proposal, hash);
}

// One hash batch per DISTINCT (raw proposal, round) referenced by the ROUND-CHANGE messages: in a round change
// every honest node carries the same last prepared proposal, so this is normally a single device call for all
// the N·(Q+1) hashes of all certificates.
void HotPath::prefetchCertificateHashes(const std::vector<MsgPtr> &rcs) {
  hash_verdict_.clear();
  last_cert_hashes = 0;
  if (!(use_batch && batch)) return;
  struct Group {
    const Proposal *rep;
    std::vector<std::pair<const Proposal *, const bytes *>> keys;
    std::vector<MsgPtr> synth;
  };
  std::map<std::pair<bytes, uint64_t>, Group> groups;
  for (auto &rc : rcs) {
    const Proposal *proposal = extract_last_prepared_proposal(*rc);
    const PreparedCertificate *cert = extract_latest_pc(*rc);
    if (!proposal || !cert) continue;  // proposalMatchesCertificate decides these without the backend
    {  // everything about this certificate already settled when the message arrived?
      bool known = true, ok = false;
      known = lookupHashVerdict(proposal, cert->proposal_message ? extract_proposal_hash(*cert->proposal_message) : nullptr, ok);
      for (auto &m : cert->prepare_messages) known = known && lookupHashVerdict(proposal, m ? extract_prepare_hash(*m) : nullptr, ok);
      if (known) continue;
    }
    Group &g = groups[{proposal->raw_proposal, proposal->round}];
    if (g.keys.empty()) g.rep = proposal;
    auto push = [&](const bytes *h) {
      auto m = std::make_shared<IbftMessage>();
      m->type = PREPARE;
      if (h) {
        m->kind = PayloadKind::PREPARE;
        m->prepare.proposal_hash = *h;
      }
      g.synth.push_back(std::move(m));
      g.keys.push_back({proposal, h});
    };
    push(cert->proposal_message ? extract_proposal_hash(*cert->proposal_message) : nullptr);
    for (auto &m : cert->prepare_messages) push(m ? extract_prepare_hash(*m) : nullptr);
  }
  for (auto &kv : groups) {
    Group &g = kv.second;
    std::vector<uint8_t> v;
    if (!batch->VerifyPrepareBatch(g.rep, g.synth, v) || v.size() != g.synth.size()) {
      fallbacks++;  // this group stays on the per-message path (the table simply has no entry)
      continue;
    }
    for (size_t i = 0; i < g.keys.size(); i++) hash_verdict_[g.keys[i]] = v[i] != 0;
    last_cert_hashes += g.keys.size();
  }
}

std::vector<MsgPtr> HotPath::handleRoundChangeMessage(const View &view) {
  const uint64_t h = view.height;
  const bool hasAcceptedProposal = getProposal() != nullptr;
  auto isValidMsgFn = [&](const IbftMessage &msg) {
    const Proposal *proposal = extract_last_prepared_proposal(msg);
    const PreparedCertificate *certificate = extract_latest_pc(msg);
    if (!msg.view) return false;  // the reference dereferences msg.View; a stored message always has one
    if (!validPCImpl(certificate, msg.view->round, h)) return false;
    return proposalMatchesCertificate(proposal, certificate);
  };
  auto isValidRCCFn = [&](uint64_t round, const std::vector<MsgPtr> &msgs) {
    if (round == view.round && hasAcceptedProposal) return false;
    return hasQuorumByMsgType(msgs, ROUND_CHANGE);
  };
  std::function<void(const std::vector<MsgPtr> &)> prepass;
  if (use_batch && batch)
    prepass = [&](const std::vector<MsgPtr> &all) {
      std::vector<const IbftMessage *> need;
      for (auto &rc : all) collect_pc(extract_latest_pc(*rc), need);
      prefetchSenders(need);
      if (cert_hits < need.size() && sender_verdict_.empty()) fallbacks++;  // something was asked and the batch failed
      prefetchCertificateHashes(all);
    };
  else
    sender_verdict_.clear();
  std::vector<MsgPtr> out = messages.GetExtendedRCC(h, isValidMsgFn, isValidRCCFn, prepass);
  sender_verdict_.clear();
  hash_verdict_.clear();
  return out;
}

}  // namespace ibft
