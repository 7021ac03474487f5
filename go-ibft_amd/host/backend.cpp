// backend.cpp — see backend.hpp.
#include "backend.hpp"

#include <algorithm>

#include <chrono>

#include <cstring>
#include <random>
#include <thread>
#include <unordered_set>

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

GpuBackend::GpuBackend(ibft_ctx *ctx) : ctx_(ctx) {
  if (const char *e = getenv("IBFT_MIN_DEVICE_ROWS")) {
    char *end = nullptr;
    const unsigned long long v = strtoull(e, &end, 10);
    if (end != e && *end == '\0') min_device_rows = (size_t)v;
  }
}

bool GpuBackend::VerifyPrepareBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs,
                                    std::vector<uint8_t> &verdict) {
  verdict.assign(msgs.size(), 0);
  if (declines(msgs.size())) return false;
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
  if (declines(msgs.size())) return false;
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

int GpuBackend::QuorumOfSenders(const std::vector<bytes> &senders, const bytes *proposer) {
  if (declines(senders.size())) return -1;  // (a map walk over a handful of senders beats a launch)
  if (proposer && proposer->size() != 20) return -1;
  std::vector<uint8_t> col(senders.size() * 20);
  for (size_t i = 0; i < senders.size(); i++) {
    if (senders[i].size() != 20) return -1;  // no validator address has another length; the host's byte compare decides
    memcpy(&col[20 * i], senders[i].data(), 20);
  }
  std::vector<uint64_t> mask((senders.size() + 63) / 64 + 1, ~0ull);
  ibft_tally_t t{};
  last_rc = proposer ? ibft_tally_prepare(ctx_, col.data(), mask.data(), senders.size(), (const uint8_t *)proposer->data(), &t)
                     : ibft_tally(ctx_, col.data(), mask.data(), senders.size(), &t);
  if (last_rc != IBFT_OK) return -1;
  return t.has_quorum ? 1 : 0;
}

bool GpuBackend::VerifySenderBatch(const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &verdict) {
  verdict.assign(msgs.size(), 0);
  if (declines(msgs.size())) return false;
  return sender_batch(msgs, verdict);
}
bool GpuBackend::sender_batch(const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &verdict) {
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
  if (declines(msgs.size())) return false;
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
                                 proposal->round, nullptr, nullptr, ms.data(), mv.data(), nullptr);
  if (last_rc != IBFT_OK) return false;
  unpack_mask(ms, sc.n, sender);
  unpack_mask(mv, sc.n, closure);
  return true;
}

bool GpuBackend::VerifySendersWire(const uint8_t *wire, const uint32_t *off, size_t n, std::vector<uint8_t> &verdict,
                                   WireStats *stats) {
  verdict.assign(n, 0);
  if (declines(n)) return false;
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
  if (!sender_batch(msgs, v2)) return false;  // (the few rows of an accepted batch that take the stock route: no min-rows rule)
  for (size_t j = 0; j < idx.size(); j++) verdict[idx[j]] = v2[j];
  return true;
}

bool GpuBackend::VerifyMessagesWire(const uint8_t *wire, const uint32_t *off, size_t n, uint64_t height, uint64_t round,
                                    const Proposal &proposal, std::vector<uint8_t> &sender, std::vector<uint8_t> &closure,
                                    std::vector<uint8_t> &judged) {
  sender.assign(n, 0);
  closure.assign(n, 0);
  judged.assign(n, 0);
  if (declines(n)) return false;
  if (n == 0) return true;
  std::vector<uint64_t> ms((n + 63) / 64, 0), mv((n + 63) / 64, 0);
  std::vector<uint8_t> cls(n, 0);
  last_rc = ibft_verify_messages_wire(ctx_, wire, off, n, height, round, (const uint8_t *)proposal.raw_proposal.data(),
                                      proposal.raw_proposal.size(), proposal.round, nullptr, ms.data(), mv.data(), cls.data(),
                                      nullptr, nullptr, nullptr);
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
  if (!sender_batch(msgs, v2)) return false;
  for (size_t j = 0; j < idx.size(); j++) sender[idx[j]] = v2[j];
  return true;
}

// the messages nested directly inside m, in the order the device lists them (wire order): the RoundChangeCertificate's
// messages of a PREPREPARE payload; proposalMessage, then prepareMessages, of a ROUND_CHANGE payload's PreparedCertificate
static void nested_messages(const IbftMessage &m, std::vector<MsgPtr> &out) {
  out.clear();
  if (m.kind == PayloadKind::PREPREPARE && m.preprepare().realise_certificate() && m.preprepare().certificate) {
    out = m.preprepare().certificate->round_change_messages;
  } else if (m.kind == PayloadKind::ROUND_CHANGE && m.round_change().realise_certificate() && m.round_change().latest_prepared_certificate) {
    const PreparedCertificate &pc = *m.round_change().latest_prepared_certificate;
    if (pc.proposal_message) out.push_back(pc.proposal_message);
    out.insert(out.end(), pc.prepare_messages.begin(), pc.prepare_messages.end());
  }
}
// the proposal hash a message carries, by payload kind (what the device reports in ibft_wire_row_t.proposal_hash)
static const bytes *carried_hash(const IbftMessage &m) {
  switch (m.kind) {
    case PayloadKind::PREPREPARE: return &m.preprepare().proposal_hash;
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
  out.rows.resize(cap);
  out.cls.assign(cap, 0);
  std::vector<uint64_t> ms(words, 0), mh(words, 0), mself(words, 0);
  size_t rows = 0;
  // A message beyond IBFT_CERT_DIGEST_MAX_BYTES comes back with its envelope digest left to the host (below).  That hash —
  // ≈3 ms per MiB on one core — does not have to wait for the device: where the signature field lies is a look at the
  // top-level fields, so a helper thread hashes the long call messages WHILE the device works on the tree (a re-proposal
  // at N = 256 is one message of 4.3 MB: 13 ms of Keccak next to 3 ms of device work instead of behind it).
  struct Early {
    size_t row;
    uint32_t cut0, cut1;
    uint8_t digest[32];
  };
  std::vector<Early> early;
  for (size_t i = 0; i < n; i++) {
    const uint32_t len = off[i + 1] - off[i];
    if (len <= IBFT_CERT_DIGEST_MAX_BYTES) continue;
    const Peek pk = peek(wire + off[i], len);
    if (!pk.ok || pk.sig_len == 0) continue;
    uint32_t lv = 1;
    for (uint32_t x = pk.sig_len; x >= 0x80; x >>= 7) lv++;
    if (pk.sig_off < 1 + lv) continue;
    early.push_back(Early{i, pk.sig_off - 1 - lv, pk.sig_off + pk.sig_len, {0}});
  }
  std::thread hasher;
  if (!early.empty())
    hasher = std::thread([&]() {
      for (Early &e : early) {
        const uint8_t *m = wire + off[e.row];
        ibft_keccak256(m, e.cut0, m + e.cut1, (off[e.row + 1] - off[e.row]) - e.cut1, e.digest);
      }
    });
  last_rc = ibft_verify_certificates_wire(ctx_, wire, off, n, cap, &rows, out.nodes.data(), out.rows.data(), out.cls.data(),
                                          ms.data(), mh.data(), mself.data());
  if (hasher.joinable()) hasher.join();
  if (last_rc != IBFT_OK) return false;  // IBFT_E_TOOBIG included: the caller's stock route handles the batch
  out.n_rows = rows;
  out.nodes.resize(rows);
  out.rows.resize(rows);
  out.cls.resize(rows);
  unpack_mask(ms, rows, out.sender);
  unpack_mask(mh, rows, out.hash);
  unpack_mask(mself, rows, out.self);
  // What the device hands back because ONE sponge would have to absorb it (a message, or a Proposal, beyond
  // IBFT_CERT_DIGEST_MAX_BYTES): hashed here with the library's host Keccak (≈340 MB/s against the 25 MB/s of a wavefront),
  // the envelope then judged as a (digest, signature, From) row of ibft_verify_seals.  Rare — blocks above 1 MiB.
  bool by_host = false;
  for (size_t r = 0; r < rows; r++)
    by_host = by_host || ((out.cls[r] & (IBFT_CERT_CLASS_DIGEST_BY_HOST | IBFT_CERT_CLASS_PROPOSAL_BY_HOST)) &&
                          !(out.cls[r] & IBFT_CERT_CLASS_NEEDS_HOST));
  if (!by_host) return true;
  const std::vector<ibft_wire_row_t> &wr = out.rows;
  std::vector<size_t> drows;
  std::vector<uint8_t> dg, sg, fr;
  for (size_t r = 0; r < rows; r++) {
    const ibft_cert_node_t &nd = out.nodes[r];
    const uint8_t *m = wire + nd.off;
    if ((out.cls[r] & IBFT_CERT_CLASS_PROPOSAL_BY_HOST) && !(out.cls[r] & IBFT_CERT_CLASS_NEEDS_HOST) &&
        (nd.flags & IBFT_CERT_HAS_PROPOSAL)) {
      uint8_t be[8], H[32];
      for (int i = 0; i < 8; i++) be[i] = (uint8_t)(nd.proposal_round >> (8 * (7 - i)));
      ibft_keccak256(wire + nd.raw_off, nd.raw_len, be, 8, H);
      if (wr[r].payload_kind == 5)  // a PREPREPARE's own (proposal, proposalHash)
        out.self[r] = wr[r].hash_len == 32 && memcmp(wr[r].proposal_hash, H, 32) == 0;
      if (wr[r].payload_kind == 8)  // a ROUND_CHANGE's lastPreparedProposal against the hashes its certificate carries
        for (uint32_t k = 0; k < nd.n_children && (size_t)nd.first_child + k < rows; k++) {
          const size_t c = nd.first_child + k;
          out.hash[c] = wr[c].hash_len == 32 && memcmp(wr[c].proposal_hash, H, 32) == 0;
        }
      out.cls[r] &= (uint8_t)~IBFT_CERT_CLASS_PROPOSAL_BY_HOST;
    }
    if ((out.cls[r] & IBFT_CERT_CLASS_DIGEST_BY_HOST) && !(out.cls[r] & IBFT_CERT_CLASS_NEEDS_HOST)) {
      // PayloadNoSig = the bytes without the signature field [cut0, cut1) (the device vouched that they are canonical)
      if (nd.cut1 < nd.cut0 || nd.cut1 > nd.len || wr[r].from_len != 20 || wr[r].sig_len != 65 || nd.cut1 - nd.cut0 < 65) {
        out.sender[r] = 0;  // no 65-byte signature / 20-byte From: IsValidValidator is false without any arithmetic
        out.cls[r] &= (uint8_t)~IBFT_CERT_CLASS_DIGEST_BY_HOST;
        continue;
      }
      uint8_t d[32];
      const Early *done = nullptr;
      for (const Early &e : early)
        if (e.row == r && e.cut0 == nd.cut0 && e.cut1 == nd.cut1) done = &e;
      if (done)
        memcpy(d, done->digest, 32);  // hashed while the device worked
      else
        ibft_keccak256(m, nd.cut0, m + nd.cut1, nd.len - nd.cut1, d);
      drows.push_back(r);
      dg.insert(dg.end(), d, d + 32);
      sg.insert(sg.end(), m + nd.cut1 - 65, m + nd.cut1);
      fr.insert(fr.end(), wr[r].from, wr[r].from + 20);
    }
  }
  if (!drows.empty()) {
    std::vector<uint64_t> dm((drows.size() + 63) / 64, 0);
    if (ibft_verify_seals(ctx_, dg.data(), sg.data(), fr.data(), nullptr, drows.size(), dm.data(), nullptr) == IBFT_OK)
      for (size_t j = 0; j < drows.size(); j++) {
        out.sender[drows[j]] = (dm[j >> 6] >> (j & 63)) & 1;
        out.cls[drows[j]] &= (uint8_t)~IBFT_CERT_CLASS_DIGEST_BY_HOST;
      }
  }
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
                               : (k == 0 && items[r].m->round_change().latest_prepared_certificate->proposal_message
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
  out.rows.assign(rows, ibft_wire_row_t{});
  // like the device, this backend vouches (class 0) only for bytes that ARE the canonical encoding — of the message and of
  // everything below it: a root whose re-encoding differs from its bytes is handed back, with its whole subtree
  std::vector<uint8_t> handed_back(rows, 0);
  for (size_t r = 0; r < rows; r++) {
    if (r < n && items[r].m) {
      const bytes again = encode(*items[r].m);
      handed_back[r] = again.size() != off[r + 1] - off[r] || memcmp(again.data(), wire + off[r], again.size()) != 0;
    } else if (r >= n) {
      handed_back[r] = handed_back[out.nodes[r].parent];
    }
  }
  for (size_t r = 0; r < rows; r++) {
    if (!items[r].m) {
      out.cls[r] = IBFT_CERT_CLASS_NEEDS_HOST;
      continue;
    }
    const IbftMessage &m = *items[r].m;
    {  // the parsed fields, as the device reports them
      ibft_wire_row_t &w = out.rows[r];
      w.status = IBFT_WIRE_OK;
      w.has_view = m.view ? 1 : 0;
      w.height = m.view ? m.view->height : 0;
      w.round = m.view ? m.view->round : 0;
      w.type = (uint8_t)(m.type <= 255 ? m.type : 255);
      w.payload_kind = m.kind == PayloadKind::PREPREPARE ? 5 : m.kind == PayloadKind::PREPARE ? 6 : m.kind == PayloadKind::COMMIT ? 7
                       : m.kind == PayloadKind::ROUND_CHANGE ? 8 : 0;
      w.from_len = (uint8_t)(m.from.size() <= 20 ? m.from.size() : 255);
      if (m.from.size() <= 20) memcpy(w.from, m.from.data(), m.from.size());
      w.sig_len = (uint8_t)(m.signature.size() < 255 ? m.signature.size() : 255);
      if (const bytes *ch = carried_hash(m)) {
        w.hash_len = (uint8_t)(ch->size() <= 32 ? ch->size() : 255);
        if (ch->size() <= 32) memcpy(w.proposal_hash, ch->data(), ch->size());
      }
      if (m.kind == PayloadKind::ROUND_CHANGE) {
        if (m.round_change().last_prepared_proposal) out.nodes[r].flags |= IBFT_CERT_HAS_PROPOSAL;
        if (m.round_change().latest_prepared_certificate) out.nodes[r].flags |= IBFT_CERT_HAS_CERTIFICATE;
      } else if (m.kind == PayloadKind::PREPREPARE) {
        if (m.preprepare().proposal) out.nodes[r].flags |= IBFT_CERT_HAS_PROPOSAL;
        if (m.preprepare().certificate) out.nodes[r].flags |= IBFT_CERT_HAS_CERTIFICATE;
      }
    }
    if (handed_back[r]) {
      out.cls[r] = IBFT_CERT_CLASS_NEEDS_HOST;  // (the sender bit stays 0: the stock route decides this message)
      continue;
    }
    out.sender[r] = v_->IsValidValidator(m);
    const bytes *h = carried_hash(m);
    const IbftMessage *p = items[r].parent;
    if (h && p && p->kind == PayloadKind::ROUND_CHANGE && p->round_change().last_prepared_proposal)
      out.hash[r] = v_->IsValidProposalHash(&*p->round_change().last_prepared_proposal, h);
    if (m.kind == PayloadKind::PREPREPARE && m.preprepare().proposal)
      out.self[r] = v_->IsValidProposalHash(&*m.preprepare().proposal, &m.preprepare().proposal_hash);
  }
  return true;
}

bool LoopBatch::VerifyPrepareBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &v) {
  if (fail_hashes || declines(msgs.size())) return false;
  calls++;
  v.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) v[i] = v_->IsValidProposalHash(proposal, extract_prepare_hash(*msgs[i]));
  return true;
}
bool LoopBatch::VerifyCommitBatch(const Proposal *proposal, const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &v) {
  if (fail_hashes || fail_seals || declines(msgs.size())) return false;
  calls++;
  v.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) {
    const bytes *h = extract_commit_hash(*msgs[i]);
    std::optional<CommittedSeal> seal = extract_committed_seal(*msgs[i]);
    v[i] = v_->IsValidProposalHash(proposal, h) && v_->IsValidCommittedSeal(h, seal ? &*seal : nullptr);
  }
  return true;
}
int LoopBatch::QuorumOfSenders(const std::vector<bytes> &senders, const bytes *proposer) {
  if (!quorum_vm || fail_quorum || !quorum_vm->initialized() || declines(senders.size())) return -1;
  quorum_calls++;
  std::unordered_set<std::string_view, sv_hash> set;
  if (proposer) set.emplace(proposer->data(), proposer->size());
  bool voided = false;
  for (const bytes &s : senders) {
    if (proposer && s.size() == proposer->size() && memcmp(s.data(), proposer->data(), s.size()) == 0) voided = true;
    set.emplace(s.data(), s.size());
  }
  unsigned __int128 sum = 0;
  for (std::string_view a : set) sum += quorum_vm->powerOf(a);
  const bool q = !voided && sum >= quorum_vm->quorum();
  return (q != wrong_quorum) ? 1 : 0;
}

bool LoopBatch::VerifySenderBatch(const std::vector<MsgPtr> &msgs, std::vector<uint8_t> &v) {
  if (fail_senders || declines(msgs.size())) return false;
  calls++;
  v.assign(msgs.size(), 0);
  for (size_t i = 0; i < msgs.size(); i++) v[i] = v_->IsValidValidator(*msgs[i]);
  return true;
}

bool LoopBatch::VerifyMessageSet(const Proposal *proposal, MessageType type, const std::vector<MsgPtr> &msgs,
                                 std::vector<uint8_t> &sender, std::vector<uint8_t> &closure) {
  if (fail_sets || !proposal || (type != PREPARE && type != COMMIT) || declines(msgs.size())) return false;
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

bool LoopBatch::VerifyMessagesWire(const uint8_t *wire, const uint32_t *off, size_t n, uint64_t height, uint64_t round,
                                   const Proposal &proposal, std::vector<uint8_t> &sender, std::vector<uint8_t> &closure,
                                   std::vector<uint8_t> &judged) {
  if (fail_sets || declines(n)) return false;
  calls++;
  set_calls++;
  sender.assign(n, 0);
  closure.assign(n, 0);
  judged.assign(n, 0);
  for (size_t i = 0; i < n; i++) {
    const uint8_t *row = wire + off[i];
    const size_t len = off[i + 1] - off[i];
    IbftMessage m;
    if (!decode(row, len, m)) continue;  // dropped by the caller as well
    sender[i] = v_->IsValidValidator(m);
    const bool here = m.view && m.view->height == height && m.view->round == round;
    const bool kind_ok = (m.type == PREPARE && m.kind == PayloadKind::PREPARE) || (m.type == COMMIT && m.kind == PayloadKind::COMMIT);
    if (!here || !kind_ok || m.from.empty()) continue;
    const bytes again = encode(m);
    if (again.size() != len || memcmp(again.data(), row, len) != 0) continue;  // not the canonical bytes: not vouched for
    judged[i] = 1;
    if (m.type == PREPARE) {
      closure[i] = v_->IsValidProposalHash(&proposal, extract_prepare_hash(m));
    } else {
      const bytes *h = extract_commit_hash(m);
      std::optional<CommittedSeal> seal = extract_committed_seal(m);
      closure[i] = v_->IsValidProposalHash(&proposal, h) && v_->IsValidCommittedSeal(h, seal ? &*seal : nullptr);
    }
  }
  return true;
}

bool HotPath::isAcceptableMessage(const IbftMessage &m, const bool *sender_ok) {
  if (sender_ok ? !*sender_ok : (!verifier || !verifier->IsValidValidator(m))) return false;  // ibft.go:1128
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
    case COMMIT: return validatorManager.HasQuorumOf(msgs);
    default: return false;
  }
}

// hasQuorumByMsgType for the messages a GetValidMessages walk just returned: they are ALL the stored messages of the view
// (the walk pruned the rest), one per sender — so with the quorum index on, Σ power of the view is already known (the
// store's hooks kept it through the prunes) and no sender set has to be rebuilt.
// The decision itself, optionally from the device: `host_answer` is what the mirror's own rule says for these senders
bool HotPath::quorumDecision(uint32_t type, const std::vector<bytes> &senders, bool host_answer) {
  if (!device_quorum || !batch || (type != PREPARE && type != COMMIT && type != ROUND_CHANGE)) return host_answer;
  if (type == PREPARE && !proposalMessage) return host_answer;  // HasPrepareQuorum: nil proposal → false without asking (:101-110)
  const int q = batch->QuorumOfSenders(senders, type == PREPARE ? &proposalMessage->from : nullptr);
  if (q < 0) return host_answer;
  device_quorum_calls++;
  if ((q != 0) != host_answer) device_quorum_mismatches++;
  return q != 0;
}

bool HotPath::hasQuorumOfStoredView(const View &view, uint32_t type, const std::vector<MsgPtr> &msgs) {
  if (device_quorum && batch && validatorManager.initialized()) {
    std::vector<bytes> senders;
    for (auto &x : msgs) senders.push_back(x->from);
    return quorumDecision(type, senders, hasQuorumByMsgType(msgs, type));
  }
  if (!index_enabled_ || !validatorManager.initialized()) return hasQuorumByMsgType(msgs, type);
  auto rebuild = [&]() {
    std::vector<bytes> senders;
    for (auto &x : msgs) senders.push_back(x->from);
    return senders;
  };
  auto pc = quorumIndex.Get(type, view.height, view.round, rebuild, validatorManager);
  if (pc.second != msgs.size()) return hasQuorumByMsgType(msgs, type);  // (cannot happen; the walk is the authority)
  if (type == PREPARE) {
    if (!proposalMessage) return false;
    if (messages.Has(view, PREPARE, proposalMessage->from)) return false;  // proposer among PREPARE signers (:117-121)
    return pc.first + validatorManager.powerOf(proposalMessage->from) >= validatorManager.quorum();
  }
  return pc.first >= validatorManager.quorum();
}

int HotPath::AddMessage(MsgPtr m, bool accepted) {
  if (!m) return 0;
  if (!accepted && !isAcceptableMessage(*m)) return 0;
  View view = *m->view;
  uint32_t type = m->type;
  messages.AddMessage(m);
  if (view.height == height) {  // ibft.go:1113-1120
    auto msgs = messages.GetValidMessages(view, (MessageType)type, [](const IbftMessage &) { return true; });
    if (hasQuorumByMsgType(msgs, type)) return 2;
  }
  return 1;
}

// The store's hooks are installed once, for every mirror (the prune of the receive-side memory must not depend
// on the quorum index being enabled).
HotPath::HotPath() {
  std::random_device rd;
  fp_seed_ = ((uint64_t)rd() << 32) ^ rd() ^ (uint64_t)(uintptr_t)this;
  messages.SetHooks(
      [this](uint32_t type, uint64_t h, uint64_t r, const bytes &from, int delta) {
        if (index_enabled_) quorumIndex.OnSender(type, h, r, from, delta, validatorManager);
      },
      [this](uint64_t below) {
        quorumIndex.OnPrune(below);
        PruneVerdictCache(below);
      });
}

void HotPath::EnableQuorumIndex() { index_enabled_ = true; }

void HotPath::PruneVerdictCache(uint64_t below_height) {
  for (auto it = seen_.begin(); it != seen_.end();) it = it->second.height < below_height ? seen_.erase(it) : std::next(it);
  if (seen_.empty()) seen_has_votes_ = false;
}

// The verdicts of one root row of a certificate call and of everything below it are noted IN the decoded objects, matched
// by position: a decoded message lists its nested messages in the order the device lists them.  A subtree whose row count
// differs from the decoded count (the device refused the wrapper as non-canonical) is left to the stock route.
// validPC (core/ibft.go:1162-1231) for the PreparedCertificate of the ROUND_CHANGE message at `row`, read off the backend's
// rows: view / type / payload kind / From / carried hash of every nested message (ibft_wire_row_t) and its IsValidValidator
// bit; with match_proposal also proposalMatchesCertificate (:516-551) against the message's own lastPreparedProposal (the
// IsValidProposalHash bits).  2 = the message carries no certificate, 1 / 0 = the verdict, −1 = not decided here.  Only the
// REGULAR shape is decided — every nested message judged (class 0), with a view, a From of at most 20 bytes and a 32-byte
// hash under the payload its type announces; anything else is left to the walk over the decoded objects, which is the
// authority for the corner cases.
int HotPath::pcVerdictFromRows(const CertVerdicts &cv, size_t row, uint64_t limit, uint64_t height, bool match_proposal) {
  if (row >= cv.n_rows || cv.rows.size() != cv.n_rows || cv.cls[row] != 0) return -1;
  const ibft_wire_row_t &rc = cv.rows[row];
  const ibft_cert_node_t &nd = cv.nodes[row];
  if (rc.status != IBFT_WIRE_OK || !rc.has_view || rc.type != ROUND_CHANGE || rc.payload_kind != 8) return -1;
  const bool has_proposal = nd.flags & IBFT_CERT_HAS_PROPOSAL, has_cert = nd.flags & IBFT_CERT_HAS_CERTIFICATE;
  if (!has_cert) return 2;
  if (match_proposal && !has_proposal) return -1;  // IsValidProposalHash(nil, hash): the backend's business
  const size_t lo = nd.first_child, n = nd.n_children;
  if (n == 0) return 0;  // ProposalMessage == nil (and no PREPARE either) — before the bounds: a leaf's first_child means nothing
  if (lo + n > cv.n_rows) return -1;
  if (cv.nodes[lo].role != IBFT_CERT_ROLE_PC_PROPOSAL) return 0;  // ProposalMessage == nil
  if (n == 1) return 0;                                            // PrepareMessages == nil
  for (size_t c = lo; c < lo + n; c++) {
    const ibft_wire_row_t &w = cv.rows[c];
    if (cv.cls[c] != 0 || w.status != IBFT_WIRE_OK || !w.has_view || w.from_len > 20 || cv.nodes[c].n_children != 0) return -1;
    if (c > lo && cv.nodes[c].role != IBFT_CERT_ROLE_PC_PREPARE) return -1;
  }
  if (!validatorManager.initialized()) return 0;  // HasQuorum of anything is false
  // the proposal message is a PREPREPARE, the others are PREPAREs (:1189-1199)
  if (cv.rows[lo].type != PREPREPARE) return 0;
  for (size_t c = lo + 1; c < lo + n; c++)
    if (cv.rows[c].type != PREPARE) return 0;
  // (type and payload agree from here on, or the hash extraction rules of the helpers apply: not decided here)
  if (cv.rows[lo].payload_kind != 5 || cv.rows[lo].hash_len != 32) return -1;
  for (size_t c = lo + 1; c < lo + n; c++)
    if (cv.rows[c].payload_kind != 6 || cv.rows[c].hash_len != 32) return -1;
  // AreValidPCMessages (messages/helpers.go:167-214): one height, one round below the limit, one hash, unique senders;
  // HasQuorum over the sender set (:1184)
  const uint64_t round = cv.rows[lo].round;
  size_t slots = 64;
  while (slots < 4 * n) slots <<= 1;
  rc_set_.assign(slots, 0);
  unsigned __int128 power = 0;
  bool ok = true;
  for (size_t c = lo; c < lo + n && ok; c++) {
    const ibft_wire_row_t &w = cv.rows[c];
    ok = w.height == height && w.round == round && w.round < limit && memcmp(w.proposal_hash, cv.rows[lo].proposal_hash, 32) == 0;
    if (!ok) break;
    const std::string_view from((const char *)w.from, w.from_len);
    const uint64_t hk = hash_key(from.data(), from.size());
    for (size_t sl = hk & (slots - 1);; sl = (sl + 1) & (slots - 1)) {
      const uint64_t e = rc_set_[sl];
      if (e == 0) {
        rc_set_[sl] = (hk & 0xFFFFFFFF00000000ull) | (uint64_t)(c - lo + 1);
        break;
      }
      if ((e & 0xFFFFFFFF00000000ull) != (hk & 0xFFFFFFFF00000000ull)) continue;
      const ibft_wire_row_t &o = cv.rows[lo + (size_t)(e & 0xFFFFFFFFull) - 1];
      if (o.from_len == w.from_len && memcmp(o.from, w.from, w.from_len) == 0) {
        ok = false;  // the same sender twice
        break;
      }
    }
    power += validatorManager.powerOf(from);
  }
  if (!ok || power < validatorManager.quorum()) return 0;
  // the proposal message comes from the proposer of its view and is validly signed; the PREPAREs are validly signed and
  // none of them comes from the proposer (:1207-1228); with match_proposal every hash is the hash of the last prepared
  // proposal (:516-551)
  for (size_t c = lo; c < lo + n; c++) {
    const ibft_wire_row_t &w = cv.rows[c];
    if (!cv.sender[c] || (match_proposal && !cv.hash[c])) return 0;
    const bool proposer = verifier && verifier->IsProposer(bytes::view((const char *)w.from, w.from_len), w.height, w.round);
    if (proposer != (c == lo)) return 0;
  }
  return 1;
}

// handleRoundChangeMessage's isValidMsgFn (core/ibft.go:478-489) for the ROUND_CHANGE message at `row`:
// validPC(latestPC, view.round, view.height) ∧ proposalMatchesCertificate(lastPreparedProposal, latestPC)
int HotPath::roundChangeVerdictFromRows(const CertVerdicts &cv, size_t row) {
  if (row >= cv.n_rows || cv.rows.size() != cv.n_rows) return -1;
  const int v = pcVerdictFromRows(cv, row, cv.rows[row].round, cv.rows[row].height, true);
  if (v != 2) return v;
  // no certificate: validPC(nil) is true; a proposal without a certificate does not match, no proposal either does
  return (cv.nodes[row].flags & IBFT_CERT_HAS_PROPOSAL) ? 0 : 1;
}

// validateProposal (core/ibft.go:683-788) as far as it is a function of the PREPREPARE message at `row` and of the
// validator set: the RoundChangeCertificate has unique senders and a quorum of them, every ROUND_CHANGE message in it is of
// the proposal's view and validly signed (ok), and the prepared certificates that are valid name a highest round and the
// hash prepared in it (has_prepared, max_round, hash).  What depends on the node and on the application — the common
// checks, IsProposer(ID), IsValidProposal, the final IsValidProposalHash — stays with validateProposal.  false = not
// decided here.
bool HotPath::proposalVerdictFromRows(const CertVerdicts &cv, size_t row, ProposalVerdict &out) {
  out = ProposalVerdict();
  if (row >= cv.n_rows || cv.rows.size() != cv.n_rows || cv.cls[row] != 0) return false;
  const ibft_wire_row_t &pp = cv.rows[row];
  const ibft_cert_node_t &nd = cv.nodes[row];
  if (pp.status != IBFT_WIRE_OK || !pp.has_view || pp.type != PREPREPARE || pp.payload_kind != 5) return false;
  out.rows = 0;
  if (!(nd.flags & IBFT_CERT_HAS_CERTIFICATE)) return true;  // rcc == nil: not ok (for a round above 0)
  const size_t lo = nd.first_child, n = nd.n_children;
  if (n && lo + n > cv.n_rows) return false;
  for (size_t c = lo; c < lo + n; c++) {
    const ibft_wire_row_t &w = cv.rows[c];
    if (cv.cls[c] != 0 || w.status != IBFT_WIRE_OK || !w.has_view || w.from_len > 20 || cv.nodes[c].role != IBFT_CERT_ROLE_RCC_MESSAGE)
      return false;
    if (w.type == ROUND_CHANGE && w.payload_kind != 8) return false;  // (ExtractLatestPC's type / payload rule: not decided here)
    out.rows += 1 + cv.nodes[c].n_children;
  }
  if (n == 0 || !validatorManager.initialized()) return true;  // HasUniqueSenders of nothing / HasQuorum of anything: false
  // HasUniqueSenders, hasQuorumByMsgType(ROUND_CHANGE) = HasQuorum over the sender set (:707-716)
  size_t slots = 64;
  while (slots < 4 * n) slots <<= 1;
  std::vector<uint64_t> set(slots, 0);  // (rc_set_ is pcVerdictFromRows' scratch)
  unsigned __int128 power = 0;
  for (size_t c = lo; c < lo + n; c++) {
    const ibft_wire_row_t &w = cv.rows[c];
    const std::string_view from((const char *)w.from, w.from_len);
    const uint64_t hk = hash_key(from.data(), from.size());
    for (size_t sl = hk & (slots - 1);; sl = (sl + 1) & (slots - 1)) {
      const uint64_t e = set[sl];
      if (e == 0) {
        set[sl] = (hk & 0xFFFFFFFF00000000ull) | (uint64_t)(c - lo + 1);
        break;
      }
      if ((e & 0xFFFFFFFF00000000ull) != (hk & 0xFFFFFFFF00000000ull)) continue;
      const ibft_wire_row_t &o = cv.rows[lo + (size_t)(e & 0xFFFFFFFFull) - 1];
      if (o.from_len == w.from_len && memcmp(o.from, w.from, w.from_len) == 0) return true;  // the same sender twice: not ok
    }
    power += validatorManager.powerOf(from);
  }
  if (power < validatorManager.quorum()) return true;
  // every message is a ROUND_CHANGE of the proposal's view, validly signed (:724-744)
  for (size_t c = lo; c < lo + n; c++) {
    const ibft_wire_row_t &w = cv.rows[c];
    if (w.type != ROUND_CHANGE || w.height != pp.height || w.round != pp.round || !cv.sender[c]) return true;
  }
  // the valid prepared certificates: (round, hash) of their proposal messages; the highest round wins, the last one among equals (:746-778)
  for (size_t c = lo; c < lo + n; c++) {
    const int v = pcVerdictFromRows(cv, c, pp.round, pp.height, false);
    if (v < 0) return false;
    if (v != 1) continue;
    const ibft_wire_row_t &first = cv.rows[cv.nodes[c].first_child];
    if (!out.has_prepared || first.round >= out.max_round) {
      out.max_round = first.round;
      memcpy(out.hash, first.proposal_hash, 32);
    }
    out.has_prepared = true;
  }
  out.ok = true;
  return true;
}

void HotPath::noteCertificateTree(const CertVerdicts &cv, size_t row, const MsgPtr &root, bool note) {
  std::vector<std::pair<size_t, const IbftMessage *>> todo{{row, root.get()}};
  std::vector<MsgPtr> kids;
  while (!todo.empty()) {
    const size_t r = todo.back().first;
    const IbftMessage *m = todo.back().second;
    todo.pop_back();
    const uint8_t cls = cv.cls[r];
    // a PREPREPARE's own (proposal, proposalHash): validateProposalCommon's IsValidProposalHash
    if (!(cls & (IBFT_CERT_CLASS_NEEDS_HOST | IBFT_CERT_CLASS_PROPOSAL_BY_HOST))) {
      const Proposal *own = extract_proposal(*m);
      if (note && own && extract_proposal_hash(*m)) {
        m->verdicts.self = cv.self[r] != 0;
        m->verdicts.self_of = own;
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
        if (note) noteSender(*kids[k], cv.sender[c] != 0);
        cert_rows++;  // rows the device judged (counted whether or not the carrier turns out to be storable)
      }
      if (note && hashes_decided && !(cv.cls[c] & IBFT_CERT_CLASS_NEEDS_HOST)) {
        // about the very hash proposalMatchesCertificate will ask about (nil when type and payload disagree: not noted)
        const bytes *h = cv.nodes[c].role == IBFT_CERT_ROLE_PC_PROPOSAL ? extract_proposal_hash(*kids[k]) : extract_prepare_hash(*kids[k]);
        if (h) {
          kids[k]->verdicts.hash = cv.hash[c] != 0;
          kids[k]->verdicts.hash_of = last;
        }
      }
      todo.push_back({c, kids[k].get()});
    }
  }
}

bool HotPath::lookupHashVerdict(const IbftMessage *m, const Proposal *proposal, const bytes *hash, bool &ok) const {
  if (!hash_verdict_.empty()) {
    auto it = hash_verdict_.find({proposal, hash});
    if (it != hash_verdict_.end()) {
      ok = it->second;
      return true;
    }
  }
  if (m && proposal && hash) {
    if (m->verdicts.hash_of == proposal && (hash == extract_proposal_hash(*m) || hash == extract_prepare_hash(*m))) {
      ok = m->verdicts.hash != 0;
      return true;
    }
    if (m->verdicts.self_of == proposal && hash == extract_proposal_hash(*m)) {
      ok = m->verdicts.self != 0;
      return true;
    }
  }
  return false;
}

// IBFT.AddMessage with IsValidValidator already answered (by the device batch or the message's arrival-time verdict)
int HotPath::addWithVerdict(MsgPtr m, bool sender_ok) {
  if (!m) return 0;
  if (!isAcceptableMessage(*m, &sender_ok)) return 0;
  return index_enabled_ ? AddMessageFast(std::move(m), true) : AddMessage(std::move(m), true);
}

void HotPath::syncClosureKey(const Proposal *proposal) {
  // same proposal as last time? (raw ‖ BE64(round), compared without building the key)
  bool same = false;
  if (!proposal) {
    same = closure_key_.empty();
  } else if (closure_key_.size() == proposal->raw_proposal.size() + 8) {
    uint8_t be[8];
    for (int i = 0; i < 8; i++) be[i] = (uint8_t)(proposal->round >> (8 * (7 - i)));
    same = memcmp(closure_key_.data(), proposal->raw_proposal.data(), proposal->raw_proposal.size()) == 0 &&
           memcmp(closure_key_.data() + proposal->raw_proposal.size(), be, 8) == 0;
  }
  if (same) return;
  bytes key;
  if (proposal) {
    key = proposal->raw_proposal;
    for (int i = 7; i >= 0; i--) key.push_back((char)(proposal->round >> (8 * i)));
  }
  closure_key_ = std::move(key);  // another proposal (or none): the closure verdicts noted so far no longer apply
  closure_epoch_++;
}

namespace {
inline uint64_t mix64(uint64_t a, uint64_t b) {
  const unsigned __int128 r = (unsigned __int128)a * b;
  return (uint64_t)r ^ (uint64_t)(r >> 64);
}
// seeded 128-bit fingerprint of a message's bytes: a fast filter in front of a byte compare (stored messages) or a Keccak
// compare (rejected ones).  NOT collision resistant whatever the seed — mix64(x, 0) == 0, so a block that equals one of
// the multiplier constants zeroes a lane — and never the only evidence for a decision.
inline void fingerprint(const uint8_t *p, size_t n, uint64_t seed, uint64_t &f1, uint64_t &f2) {
  uint64_t a = seed ^ 0x9E3779B97F4A7C15ull, b = (seed * 0xD6E8FEB86659FD93ull) ^ (uint64_t)n;
  if (n >= 256) {  // a long message (certificates inside): eight independent chains over 64-byte strides — the multiplies of
                   // one chain wait for each other, and a round change brings megabytes
    uint64_t c = a ^ 0x2D358DCCAA6C78A5ull, d = b ^ 0x8BB84B93962EACC9ull, e = a ^ 0x4B33A62ED433D4A3ull,
             f = b ^ 0x4D5A2DA51DE1AA47ull, g = a ^ 0x9FB21C651E98DF25ull, h = b ^ 0xC3A5C85C97CB3127ull;
    while (n >= 64) {
      uint64_t w[8];
      memcpy(w, p, 64);
      a = mix64(a ^ w[0], 0xA0761D6478BD642Full ^ w[1]);
      b = mix64(b ^ w[1], 0xE7037ED1A0B428DBull ^ w[0]);
      c = mix64(c ^ w[2], 0xA0761D6478BD642Full ^ w[3]);
      d = mix64(d ^ w[3], 0xE7037ED1A0B428DBull ^ w[2]);
      e = mix64(e ^ w[4], 0xA0761D6478BD642Full ^ w[5]);
      f = mix64(f ^ w[5], 0xE7037ED1A0B428DBull ^ w[4]);
      g = mix64(g ^ w[6], 0xA0761D6478BD642Full ^ w[7]);
      h = mix64(h ^ w[7], 0xE7037ED1A0B428DBull ^ w[6]);
      p += 64;
      n -= 64;
    }
    a = mix64(a ^ c, 0x8EBC6AF09C88C6E3ull ^ e) ^ mix64(g, 0x589965CC75374CC3ull ^ c);
    b = mix64(b ^ d, 0x1D8E4E27C47D124Full ^ f) ^ mix64(h, 0xEB44ACCAB455D165ull ^ d);
  }
  while (n >= 16) {
    uint64_t x, y;
    memcpy(&x, p, 8);
    memcpy(&y, p + 8, 8);
    a = mix64(a ^ x, 0xA0761D6478BD642Full ^ y);
    b = mix64(b ^ y, 0xE7037ED1A0B428DBull ^ x);
    p += 16;
    n -= 16;
  }
  uint8_t tail[16] = {0};
  if (n) memcpy(tail, p, n);
  uint64_t x, y;
  memcpy(&x, tail, 8);
  memcpy(&y, tail + 8, 8);
  a = mix64(a ^ x, 0x8EBC6AF09C88C6E3ull ^ y);
  b = mix64(b ^ y, 0x589965CC75374CC3ull ^ x);
  f1 = mix64(a, b ^ 0x1D8E4E27C47D124Full);
  f2 = mix64(b, a ^ 0xEB44ACCAB455D165ull);
}
}  // namespace

bool HotPath::IngestWire(const std::vector<bytes> &raw, std::vector<int> &results, IngestStats *stats) {
  bytes wire;
  std::vector<uint32_t> off{0};
  size_t total = 0;
  for (const bytes &r : raw) total += r.size();
  wire.reserve(total);
  for (const bytes &r : raw) {
    wire += r;
    off.push_back((uint32_t)wire.size());
  }
  std::vector<int8_t> res(raw.size(), -1);
  const bool ok = IngestFlat((const uint8_t *)wire.data(), off.data(), raw.size(), res.data(), stats);
  results.assign(res.begin(), res.end());
  return ok;
}

// IngestFlat, stage by stage (go-ibft_amd/host/DESIGN.md has the reasons):
//   look     every row: field walk (peek), fingerprint where a table could hold it → re-deliveries of stored objects /
//            stored rows / rejected messages answered at once, repeats inside the batch folded, stale views rejected,
//            PREPARE / COMMIT of the current view marked as row candidates (not decoded), the rest queued for decoding
//   decode   the non-candidates, top level only for certificate carriers — on a helper thread while the device works
//   (0)      certificate carriers → ONE VerifyCertificatesWire (after a roots-first call while forged carriers keep arriving):
//            envelope verdicts, ROUND_CHANGE / PREPREPARE certificate rules from the rows, certificates left undecoded
//   (1)      everything else that is undecided → ONE VerifyMessagesWire: sender verdicts, closures of the current view's
//            PREPARE / COMMIT; judged candidates become rows, the others are decoded after all
//   (2)      what no byte-judging call covered → the sender batch / the per-message verifier
//   store    IBFT.AddMessage per message in arrival order (rows in runs under one lock), results 0 / 1 / 2 / −1, memory of
//            stored objects and of rejected fingerprints updated
bool HotPath::IngestFlat(const uint8_t *wire_in, const uint32_t *off, size_t n, int8_t *results, IngestStats *stats,
                         uint8_t *types, const std::shared_ptr<const void> &owned) {
  for (size_t i = 0; i < n; i++) results[i] = -1;
  if (types) memset(types, 0xFF, n);
  IngestStats st;
  if (n == 0) {
    if (stats) *stats = st;
    return true;
  }
  std::vector<uint32_t> rebased;  // rows of a larger buffer: offsets from the first row's first byte
  if (off[0] != 0) {
    rebased.resize(n + 1);
    for (size_t i = 0; i <= n; i++) rebased[i] = off[i] - off[0];
    wire_in += off[0];
    off = rebased.data();
  }
  const Proposal *proposal = getProposal();
  syncClosureKey(proposal);
  // ONE copy of the batch: the buffer every message decoded below points into (and keeps alive) — or none, when the caller
  // hands over a buffer it shares (`owned`: the receive queue's)
  const uint8_t *wire = wire_in;
  const std::shared_ptr<const void> backing = owned ? owned : make_backing(wire_in, off[n], &wire);
  std::vector<MsgPtr> msgs(n);
  std::vector<int8_t> verdict(n, -1);       // −1 unknown, 0 / 1 decided
  std::vector<uint64_t> fp1(n), fp2(n);
  std::vector<int32_t> dup_of(n, -1);       // a repeat inside this batch: the row that is asked instead
  std::vector<uint8_t> stale(n, 0);         // rejected without any arithmetic (a view that cannot be accepted)
  std::vector<uint8_t> kinds(n, 0);         // PayloadKind of each new row (from the peek)
  std::vector<size_t> to_decode;            // new rows whose top-level walk succeeded
  // Rows that may never become objects (backend.hpp: use_lean): a PREPARE / COMMIT of the current view whose fields occur
  // once.  Decoding them is put off until the backend has spoken: what it vouches for is stored as a row.
  const bool lean_mode = use_lean && use_batch && batch && use_sets && proposal && index_enabled_;
  std::vector<uint8_t> cand(lean_mode ? n : 0, 0), as_row(lean_mode ? n : 0, 0);
  std::vector<LeanRow> lrow(lean_mode ? n : 0);
  std::vector<uint8_t> ptype(n, 0xFF);
  struct Again {
    uint32_t type;
    LeanRow row;
    std::shared_ptr<const void> backing;
  };
  std::vector<Again> again;                 // stored rows delivered again
  std::vector<int32_t> again_at(n, -1);
  std::vector<size_t> ask;                  // rows the device has to judge: first occurrence of each distinct new message
  // repeats inside the batch: an open-addressing table over the fingerprints (entry = row + 1)
  size_t fib_mask = 63;
  while (fib_mask + 1 < 2 * n) fib_mask = fib_mask * 2 + 1;
  std::vector<uint32_t> fib(fib_mask + 1, 0);
  auto first_in_batch = [&](size_t i, const uint8_t *row, size_t len) -> int32_t {  // the earlier row with these bytes, or −1 (and i is entered)
    for (size_t sl = fp1[i] & fib_mask;; sl = (sl + 1) & fib_mask) {
      const uint32_t e = fib[sl];
      if (e == 0) {
        fib[sl] = (uint32_t)i + 1;
        return -1;
      }
      const size_t f = e - 1;
      if (fp1[f] == fp1[i] && fp2[f] == fp2[i] && off[f + 1] - off[f] == len && memcmp(wire + off[f], row, len) == 0) return (int32_t)f;
    }
  };
  // the rows already stored for the current view, per type: the store itself remembers them (a re-delivery is the sender's
  // row with the same bytes), so rows need no entry in seen_
  LeanView *stored_rows[4] = {nullptr, nullptr, nullptr, nullptr};
  if (lean_mode) {
    View cur;
    cur.height = height;
    cur.round = round;
    stored_rows[PREPARE] = messages.LeanFor(cur, PREPARE, closure_epoch_, valset_epoch_);
    stored_rows[COMMIT] = messages.LeanFor(cur, COMMIT, closure_epoch_, valset_epoch_);
  }
  // A row candidate needs its fingerprint only for the two tables it could be in: the rejected ones, and seen_ when that holds
  // PREPARE / COMMIT objects at all (its own re-deliveries are found in the store, its repeats inside the batch by sender)
  const bool cand_fp = !seen_rejected_.empty() || seen_has_votes_;
  std::vector<uint8_t> have_fp(n, 0);
  auto ensure_fp = [&](size_t i) {
    if (!have_fp[i]) {
      fingerprint(wire + off[i], off[i + 1] - off[i], fp_seed_, fp1[i], fp2[i]);
      have_fp[i] = 1;
    }
  };
  std::vector<uint32_t> cand_tab;  // repeats of row candidates inside the batch: by sender, then byte for byte
  size_t cand_mask = 0;
  auto first_cand_in_batch = [&](size_t i, const uint8_t *row, size_t len, uint64_t from_hash) -> int32_t {
    if (cand_tab.empty()) {
      cand_mask = fib_mask;
      cand_tab.assign(cand_mask + 1, 0);
    }
    for (size_t sl = from_hash & cand_mask;; sl = (sl + 1) & cand_mask) {
      const uint32_t e = cand_tab[sl];
      if (e == 0) {
        cand_tab[sl] = (uint32_t)i + 1;
        return -1;
      }
      const size_t f = e - 1;
      if (off[f + 1] - off[f] == len && memcmp(wire + off[f], row, len) == 0) return (int32_t)f;
    }
  };
  for (size_t i = 0; i < n; i++) {
    const uint8_t *row = wire + off[i];
    const size_t len = off[i + 1] - off[i];
    // A look at the message without decoding it (see below)
    const Peek pk = peek(row, len);
    const bool is_cand = pk.ok && lean_mode && pk.simple && pk.from_len && pk.has_view && pk.height == height && pk.round == round &&
                         ((pk.type == PREPARE && pk.kind == PayloadKind::PREPARE) || (pk.type == COMMIT && pk.kind == PayloadKind::COMMIT));
    if (!is_cand || cand_fp) {
      ensure_fp(i);
      if (!seen_.empty()) {
        auto hit = seen_.find(fp1[i]);
        if (hit != seen_.end() && hit->second.fp2 == fp2[i] && hit->second.len == len && memcmp(hit->second.wire, row, len) == 0) {
          msgs[i] = hit->second.msg;  // the stored object, with everything noted in it: no decode, nothing to ask
          verdict[i] = 1;
          st.cache_hits++;
          continue;
        }
      }
      if (!seen_rejected_.empty()) {
        auto rej = seen_rejected_.find(fp1[i]);
        if (rej != seen_rejected_.end() && rej->second.fp2 == fp2[i]) {
          uint8_t dg[32];
          (void)ibft_keccak256(row, len, nullptr, 0, dg);
          if (memcmp(dg, rej->second.digest, 32) == 0) {
            verdict[i] = 0;  // THESE bytes were rejected before (the reference would judge them again — and reject them again)
            st.cache_hits++;
            results[i] = 0;
            continue;
          }
          // another message under the same fingerprint (an honest message somebody prepared a colliding forgery for):
          // judged like any other
        }
      }
    }
    // A look at the message without decoding it: rows whose top-level walk fails are dropped (proto.Unmarshal error:
    // results −1).  What AddMessage rejects whatever the signature says costs no device work (and, for a PREPREPARE /
    // ROUND_CHANGE, no expansion of its certificates): a nil view or a view below the state's cannot pass
    // isAcceptableMessage (core/ibft.go:1133-1148).  (Membership of From is the Backend's business — a mock accepts anybody
    // — so it is NOT pre-judged here; the device rejects a non-member like any other bad signature.)
    if (is_cand) {
      kinds[i] = (uint8_t)pk.kind;
      ptype[i] = (uint8_t)pk.type;
      const std::string_view from((const char *)row + pk.from_off, pk.from_len);
      const uint64_t from_hash = hash_key(from.data(), from.size());
      if (const LeanView *lv = stored_rows[pk.type]) {
        const LeanRow *was = lv->find(from, from_hash);
        if (was && was->len == len && memcmp(was->wire, row, len) == 0) {
          // a stored row delivered again: stored again below, in arrival order (its sender's row is overwritten by itself)
          st.cache_hits++;
          verdict[i] = 2;
          again_at[i] = (int32_t)again.size();
          again.push_back(Again{pk.type, *was, lv->buffers[was->buf]});
          continue;
        }
      }
      if ((dup_of[i] = have_fp[i] ? first_in_batch(i, row, len) : first_cand_in_batch(i, row, len, from_hash)) >= 0) continue;
      cand[i] = 1;
      LeanRow &lr = lrow[i];
      lr.wire = row;
      lr.len = (uint32_t)len;
      lr.from_off = pk.from_off; lr.from_len = pk.from_len;
      lr.hash_off = pk.hash_off; lr.hash_len = pk.hash_len;
      lr.seal_off = pk.seal_off; lr.seal_len = pk.seal_len;
      lr.sender_hash = from_hash;
      ask.push_back(i);
      continue;
    }
    if ((dup_of[i] = first_in_batch(i, row, len)) >= 0) continue;
    if (!pk.ok) continue;
    kinds[i] = (uint8_t)pk.kind;
    ptype[i] = (uint8_t)(pk.type <= 3 ? pk.type : 0xFE);
    to_decode.push_back(i);
    const bool view_ok = pk.has_view && !(height > pk.height) && !(height == pk.height && pk.round < round);
    if (use_batch && batch && !view_ok) {
      stale[i] = 1;
      verdict[i] = 0;
      continue;
    }
    ask.push_back(i);
  }
  // Decoding (one object per new message) does not depend on the device and the device does not depend on it: with the
  // GPU backend — which takes the BYTES — a helper thread decodes while this thread is inside the device calls.
  // (the PreparedCertificate of a ROUND_CHANGE message waits for the backend's word: what the backend vouches for is
  // judged from its rows and stays undecoded — roundChangeVerdictFromRows — everything else is decoded right after the call)
  const bool defer_certificates = use_rc_rows && use_batch && batch && use_certs;
  std::vector<uint8_t> vouched(defer_certificates ? n : 0, 0);
  auto decode_all = [&]() {
    for (size_t i : to_decode) {
      auto m = std::make_shared<IbftMessage>();
      const bool defer = defer_certificates && (kinds[i] == (uint8_t)PayloadKind::ROUND_CHANGE || kinds[i] == (uint8_t)PayloadKind::PREPREPARE);
      if (decode_in(backing, wire + off[i], off[i + 1] - off[i], *m, defer)) msgs[i] = std::move(m);  // else: dropped (results −1)
    }
  };
  auto decode_candidate = [&](size_t i) {  // a candidate that needs an object after all
    cand[i] = 0;
    auto m = std::make_shared<IbftMessage>();
    if (decode_in(backing, wire + off[i], off[i + 1] - off[i], *m)) msgs[i] = std::move(m);
  };
  const bool bytes_backend = use_batch && batch && dynamic_cast<GpuBackend *>(batch) != nullptr;
  std::thread decoder;
  if (bytes_backend && to_decode.size() >= 512)
    decoder = std::thread(decode_all);
  else
    decode_all();
  auto decoded = [&]() {  // from here on msgs[] is needed
    if (decoder.joinable()) decoder.join();
  };
  struct JoinOnExit {
    std::thread &t;
    ~JoinOnExit() {
      if (t.joinable()) t.join();
    }
  } join_on_exit{decoder};
  if (!decoder.joinable()) {  // decoded already: rows that did not decode leave the batch here, as before
    std::vector<size_t> keep;
    for (size_t i : ask)
      if (msgs[i] || (lean_mode && cand[i])) keep.push_back(i);
    ask.swap(keep);
  }
  const std::vector<size_t> asked = ask;  // every distinct undecided message of the batch
  std::vector<uint8_t> was_asked(n, 0);
  for (size_t i : asked) was_asked[i] = 1;
  // rows [lo, hi) of the batch as one wire + offsets: the batch's own buffer when they are ALL its rows, a copy otherwise
  bytes sub_wire;
  std::vector<uint32_t> sub_off;
  auto rows_as_wire = [&](const std::vector<size_t> &rows, const uint8_t *&w, const uint32_t *&o) {
    if (rows.size() == n) {
      w = wire;
      o = off;
      return;
    }
    sub_wire.clear();
    sub_off.assign(1, 0);
    for (size_t i : rows) {
      sub_wire.append((const char *)wire + off[i], off[i + 1] - off[i]);
      sub_off.push_back((uint32_t)sub_wire.size());
    }
    w = (const uint8_t *)sub_wire.data();
    o = sub_off.data();
  };
  // (0) messages that carry certificates: the whole tree — their own envelope and every message nested in them — in ONE
  // device call, from the bytes as they arrived
  size_t forged_now = 0;  // carriers of this batch whose own envelope failed
  if (use_batch && batch && use_certs) {
    std::vector<size_t> carriers;
    for (size_t i : ask)
      if (kinds[i] == (uint8_t)PayloadKind::PREPREPARE || kinds[i] == (uint8_t)PayloadKind::ROUND_CHANGE) carriers.push_back(i);
    // Expanding a tree before its carrier is authenticated lets anybody buy up to N² signature checks with one message (the
    // reference spends ONE IsValidValidator on a stranger's message).  While forged carriers keep arriving, the envelopes
    // of the carriers are judged first — one more backend call — and only the trees of authenticated ones are expanded.
    const bool roots_first = !carriers.empty() && (cert_roots_first == 1 || (cert_roots_first == 2 && forged_carriers_ >= 4.0));
    if (roots_first) {
      decoded();
      std::vector<uint8_t> v;
      bool ok = false;
      const auto td = std::chrono::steady_clock::now();
      if (auto *gpu = dynamic_cast<GpuBackend *>(batch)) {
        const uint8_t *w;
        const uint32_t *o;
        rows_as_wire(carriers, w, o);
        ok = gpu->VerifySendersWire(w, o, carriers.size(), v);
      } else {
        std::vector<MsgPtr> sub;
        std::vector<size_t> at;
        for (size_t j = 0; j < carriers.size(); j++)
          if (msgs[carriers[j]]) {
            sub.push_back(msgs[carriers[j]]);
            at.push_back(j);
          }
        std::vector<uint8_t> vs;
        ok = batch->VerifySenderBatch(sub, vs) && vs.size() == sub.size();
        v.assign(carriers.size(), 1);  // (what did not decode is dropped anyway)
        for (size_t k = 0; ok && k < at.size(); k++) v[at[k]] = vs[k];
      }
      st.device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
      if (ok && v.size() == carriers.size()) {
        st.device_calls++;
        roots_first_calls++;
        std::vector<size_t> authentic;
        for (size_t j = 0; j < carriers.size(); j++) {
          if (v[j]) {
            authentic.push_back(carriers[j]);
          } else {
            verdict[carriers[j]] = 0;  // IsValidValidator failed: nothing below it is looked at
            forged_now++;
          }
        }
        carriers.swap(authentic);
        std::vector<size_t> left;
        for (size_t i : ask)
          if (verdict[i] < 0) left.push_back(i);
        ask.swap(left);
      }
    }
    if (!carriers.empty()) {
      const uint8_t *w;
      const uint32_t *o;
      rows_as_wire(carriers, w, o);
      CertVerdicts cv;
      const auto td = std::chrono::steady_clock::now();
      const bool cert_ok = batch->VerifyCertificatesWire(w, o, carriers.size(), cv);
      st.device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
      decoded();
      if (cert_ok && cv.n_rows >= carriers.size()) {
        st.device_calls++;
        cert_calls++;
        for (size_t j = 0; j < carriers.size(); j++) {
          if (cv.cls[j] == 0) {
            verdict[carriers[j]] = cv.sender[j] ? 1 : 0;
            cert_rows++;
            if (!cv.sender[j]) forged_now++;
          }
          MsgPtr &m = msgs[carriers[j]];
          if (!m) continue;
          const bool deferred = (m->kind == PayloadKind::ROUND_CHANGE && m->round_change().certificate_deferred) ||
                                (m->kind == PayloadKind::PREPREPARE && m->preprepare().certificate_deferred);
          if (defer_certificates && cv.cls[j] == 0 && verdict[carriers[j]] == 0 && deferred) {
            // a forged carrier: rejected whatever its certificate says — nothing below it is looked at, nothing decoded
            vouched[carriers[j]] = 1;
            size_t below = cv.nodes[j].n_children;
            if (m->kind == PayloadKind::PREPREPARE)
              for (size_t c = cv.nodes[j].first_child, e = c + cv.nodes[j].n_children; c < e && c < cv.n_rows; c++)
                below += cv.nodes[c].n_children;
            cert_rows += below;  // rows the device judged (counted whether or not the carrier turns out to be storable)
            continue;
          }
          if (defer_certificates && cv.cls[j] == 0 && m->kind == PayloadKind::ROUND_CHANGE) {
            vouched[carriers[j]] = 1;  // well-formed and canonical down to the last nested message: may stay undecoded
            const int rc_ok = roundChangeVerdictFromRows(cv, j);
            if (rc_ok >= 0) {
              const uint32_t nested = cv.nodes[j].n_children;
              if (verdict[carriers[j]] != 0) {
                m->verdicts.rc_ok = (uint8_t)rc_ok;
                m->verdicts.rc_rows = nested;
                m->verdicts.rc_epoch = valset_epoch_;
                rc_from_rows++;
              }
              if (deferred || nested == 0) {
                cert_rows += nested;  // rows the device judged (counted whether or not the carrier turns out to be storable)
                continue;
              }
            }
          }
          if (defer_certificates && cv.cls[j] == 0 && m->kind == PayloadKind::PREPREPARE) {
            vouched[carriers[j]] = 1;
            ProposalVerdict pv;
            if (proposalVerdictFromRows(cv, j, pv)) {
              if (verdict[carriers[j]] != 0) {
                m->proposal_verdict.mut() = pv;
                m->verdicts.pp_epoch = valset_epoch_;
                // (its own (proposal, proposalHash) bit — validateProposalCommon's question — is a row verdict as well)
                if (const Proposal *own = extract_proposal(*m)) {
                  if (extract_proposal_hash(*m) && !(cv.cls[j] & IBFT_CERT_CLASS_PROPOSAL_BY_HOST)) {
                    m->verdicts.self = cv.self[j] != 0;
                    m->verdicts.self_of = own;
                  }
                }
                pp_from_rows++;
              }
              if (deferred || cv.nodes[j].n_children == 0) {
                cert_rows += pv.rows;
                continue;
              }
            }
          }
          if (deferred && !(m->kind == PayloadKind::ROUND_CHANGE ? m->round_change().realise_certificate()
                                                                 : m->preprepare().realise_certificate())) {
            m.reset();  // proto.Unmarshal would have failed on this message: dropped (results −1)
            continue;
          }
          // what the device said about the nested messages is only worth noting for a carrier that can be stored
          noteCertificateTree(cv, j, m, verdict[carriers[j]] != 0);
        }
        std::vector<size_t> left;
        for (size_t i : ask)
          if (verdict[i] < 0) left.push_back(i);
        ask.swap(left);
      }
    }
  }
  forged_carriers_ = forged_carriers_ * 0.5 + (double)forged_now;  // (half-life: one batch)
  // (1) messages of the current view with the proposal at hand: judged completely, one set call per type
  std::vector<size_t> rest;
  // (a backend that judges bytes: the GPU always; the loop backend when rows are kept, so that the row path runs without a device)
  BatchVerifier *gpu_sets = (use_batch && use_sets && proposal && (bytes_backend || lean_mode)) ? batch : nullptr;
  bool wire_sets_done = false;
  bool wire_sets_offered = gpu_sets != nullptr;
  if (gpu_sets && !ask.empty()) {
    // the device walks the bytes AND judges every PREPARE / COMMIT of this view completely: one call for the micro-batch
    const uint8_t *w;
    const uint32_t *o;
    rows_as_wire(ask, w, o);
    std::vector<uint8_t> vs, vc, judged;
    const auto td = std::chrono::steady_clock::now();
    const bool sets_ok = gpu_sets->VerifyMessagesWire(w, o, ask.size(), height, round, *proposal, vs, vc, judged);
    st.device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
    decoded();
    if (sets_ok && vs.size() == ask.size() && vc.size() == ask.size() && judged.size() == ask.size()) {
      st.device_calls++;
      for (size_t j = 0; j < ask.size(); j++) {
        const size_t i = ask[j];
        verdict[i] = vs[j] ? 1 : 0;
        if (lean_mode && cand[i]) {
          if (judged[j]) {  // vouched canonical: a row if its sender is valid, rejected (without ever being decoded) if not
            st.set_rows++;
            if (vs[j]) {
              as_row[i] = 1;
              lrow[i].closure = vc[j] ? 1 : 0;
            }
            continue;
          }
          decode_candidate(i);
        }
        if (judged[j] && msgs[i]) {
          noteClosure(*msgs[i], vc[j] != 0);
          st.set_rows++;
        }
      }
      wire_sets_done = true;
    } else if (!bytes_backend) {
      wire_sets_offered = false;  // (the loop backend's set call failed: the object routes below, as before)
    }
  }
  decoded();
  if (defer_certificates)  // certificates nobody vouched for are decoded now; a message whose certificate does not decode is dropped
    for (size_t i : to_decode)
      if (msgs[i] && !vouched[i] &&
          ((msgs[i]->kind == PayloadKind::ROUND_CHANGE && msgs[i]->round_change().certificate_deferred &&
            !msgs[i]->round_change().realise_certificate()) ||
           (msgs[i]->kind == PayloadKind::PREPREPARE && msgs[i]->preprepare().certificate_deferred &&
            !msgs[i]->preprepare().realise_certificate())))
        msgs[i].reset();
  if (lean_mode && !wire_sets_done)
    for (size_t i : ask)
      if (cand[i]) decode_candidate(i);
  {  // rows that do not decode leave the batch (results −1), whichever thread decoded them
    std::vector<size_t> keep;
    for (size_t i : ask)
      if (msgs[i] || (lean_mode && cand[i])) keep.push_back(i);
    ask.swap(keep);
  }
  if (wire_sets_done) {
    // nothing left
  } else if (use_batch && batch && use_sets && proposal && !wire_sets_offered) {
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
      const auto td = std::chrono::steady_clock::now();
      const bool set_ok = batch->VerifyMessageSet(proposal, t ? COMMIT : PREPARE, sub, vs, vc);
      st.device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
      if (set_ok && vs.size() == sub.size() && vc.size() == sub.size()) {
        st.device_calls++;
        st.set_rows += sub.size();
        for (size_t k = 0; k < sub.size(); k++) {
          verdict[of_type[t][k]] = vs[k] ? 1 : 0;
          noteClosure(*sub[k], vc[k] != 0);
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
      const auto td = std::chrono::steady_clock::now();
      struct AddTime {
        IngestStats &st;
        std::chrono::steady_clock::time_point t0;
        ~AddTime() { st.device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
      } add_time{st, td};
      if (auto *gpu = dynamic_cast<GpuBackend *>(batch)) {  // the device walks the bytes themselves (§8f rank 3)
        const uint8_t *w;
        const uint32_t *o;
        rows_as_wire(rest, w, o);
        ok = gpu->VerifySendersWire(w, o, rest.size(), v);
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
      for (size_t j = 0; j < rest.size(); j++) v[j] = msgs[rest[j]] && verifier && verifier->IsValidValidator(*msgs[rest[j]]);
    }
    for (size_t j = 0; j < rest.size(); j++) verdict[rest[j]] = v[j] ? 1 : 0;
  }
  st.device_rows = asked.size();
  auto remember_rejected = [&](size_t i) {
    ensure_fp(i);
    if (rejected_fifo_.size() < rejected_cap) {
      rejected_fifo_.push_back(fp1[i]);
    } else if (rejected_cap) {
      seen_rejected_.erase(rejected_fifo_[rejected_head_]);
      rejected_fifo_[rejected_head_] = fp1[i];
      rejected_head_ = (rejected_head_ + 1) % rejected_cap;
    }
    if (rejected_cap) {
      Rejected &rj = seen_rejected_[fp1[i]];
      rj.fp2 = fp2[i];
      (void)ibft_keccak256(wire + off[i], off[i + 1] - off[i], nullptr, 0, rj.digest);
    }
  };
  // IBFT.AddMessage per message, in arrival order, with the verdict attached; what was stored is remembered for its
  // re-deliveries, what was rejected by its fingerprint
  // (consecutive rows of one type are stored as a run: one lock, one look at the counters — addLeanRun)
  std::vector<const LeanRow *> run_rows;
  std::vector<size_t> run_at;
  uint32_t run_type = 0;
  auto flush_run = [&]() {
    if (run_rows.empty()) return;
    addLeanRun(run_type, run_rows, run_at, backing, results);
    run_rows.clear();
    run_at.clear();
  };
  for (size_t i = 0; i < n; i++) {
    if (verdict[i] == 2) {  // a stored row of the current view delivered again
      flush_run();
      if (types) types[i] = ptype[i];
      const Again &sn = again[(size_t)again_at[i]];
      results[i] = (int8_t)addLeanRow(sn.type, height, round, sn.row, sn.backing);
      continue;
    }
    size_t src = i;
    if (dup_of[i] >= 0) {  // a repeat inside this batch: the first occurrence's object and verdict
      src = (size_t)dup_of[i];
      msgs[i] = msgs[src];
      verdict[i] = verdict[src];
    }
    if (lean_mode && cand[src]) {  // judged from its bytes, never decoded
      if (types) types[i] = ptype[src];
      if (!as_row[src]) {
        results[i] = 0;  // IsValidValidator failed (isAcceptableMessage, core/ibft.go:1128): the store is not touched
        if (src == i) remember_rejected(i);
      } else {
        if (!run_rows.empty() && run_type != ptype[src]) flush_run();
        run_type = ptype[src];
        run_rows.push_back(&lrow[src]);
        run_at.push_back(i);
      }
      continue;
    }
    flush_run();
    if (!msgs[i]) continue;
    if (types) types[i] = (uint8_t)(msgs[i]->type <= 3 ? msgs[i]->type : 0xFE);
    if (verdict[i] == 1) noteSender(*msgs[i], true);
    const bool fresh = was_asked[i] != 0;
    results[i] = (int8_t)addWithVerdict(msgs[i], verdict[i] == 1);
    if (!fresh) continue;
    if (results[i] > 0) {
      if (seen_.size() >= seen_cap) {
        seen_.clear();
        seen_has_votes_ = false;
      }
      const IbftMessage &m = *msgs[i];
      ensure_fp(i);
      if (m.type == PREPARE || m.type == COMMIT) seen_has_votes_ = true;
      seen_[fp1[i]] = Seen{fp2[i], msgs[i], wire + off[i], off[i + 1] - off[i], m.view ? m.view->height : 0};
    } else if (verdict[i] == 0) {
      remember_rejected(i);
    }
  }
  flush_run();
  // A large batch of which little was stored (a flood of rejected messages around a few honest ones): the stored rows must
  // not keep the whole buffer — everybody's rejected bytes — alive until the height is pruned.  They move into a buffer of
  // their own; objects stored from such a batch (few: carriers, odd encodings) still hold it.
  if (lean_mode && off[n] >= repack_min_bytes) {
    size_t kept = 0, objects = 0;
    for (size_t i = 0; i < n; i++) {
      if (results[i] <= 0) continue;
      kept += off[i + 1] - off[i];
      if (msgs[i]) objects++;
    }
    if (objects == 0 && kept * 4 < off[n]) {
      View cur;
      cur.height = height;
      cur.round = round;
      repacked_bytes += messages.RepackLean(cur, PREPARE, backing) + messages.RepackLean(cur, COMMIT, backing);
    }
  }
  last_ingest_device_ms = st.device_ms;
  if (stats) *stats = st;
  return true;
}


// hasQuorumByMsgType over the stored (unverified) messages of a view, without walking them: 2 = quorum (SignalEvent), 1 = not yet
int HotPath::quorumProbe(uint32_t type, const View &view) {
  auto rebuild = [&]() { return messages.SendersOf(view, (MessageType)type); };
  auto pc = quorumIndex.Get(type, view.height, view.round, rebuild, validatorManager);
  bool q = false;
  switch (type) {
    case PREPREPARE: q = pc.second >= 1; break;
    case PREPARE: {  // HasPrepareQuorum: proposer joins the set; a PREPARE from the proposer voids it
      if (!proposalMessage) break;
      const unsigned __int128 w = validatorManager.powerOf(proposalMessage->from);
      q = validatorManager.initialized() && pc.first + w >= validatorManager.quorum() &&
          !messages.Has(view, PREPARE, proposalMessage->from);  // (the store is only asked once the power suffices)
      break;
    }
    case ROUND_CHANGE:
    case COMMIT: q = validatorManager.initialized() && pc.first >= validatorManager.quorum(); break;
    default: break;
  }
  return q ? 2 : 1;
}

int HotPath::AddMessageFast(MsgPtr m, bool accepted) {
  if (!m) return 0;
  if (!accepted && !isAcceptableMessage(*m)) return 0;
  const View view = *m->view;
  const uint32_t type = m->type;
  messages.AddMessage(m);
  if (view.height != height) return 1;
  return quorumProbe(type, view);
}

// IBFT.AddMessage for a message that is kept as a row: IsValidValidator and the view checks of isAcceptableMessage were
// answered before (by the backend that judged the bytes, by the look at the view); store, then the quorum probe.
int HotPath::addLeanRow(uint32_t type, uint64_t h, uint64_t r, const LeanRow &row, const std::shared_ptr<const void> &backing) {
  if (!messages.AddLean(type, h, r, row, backing, closure_epoch_, valset_epoch_)) {
    // the view's rows were judged against another proposal / validator set (they are objects now): an object too
    auto m = std::make_shared<IbftMessage>();
    if (!decode_in(backing, row.wire, row.len, *m)) return -1;
    noteSender(*m, true);
    noteClosure(*m, row.closure != 0);
    return AddMessageFast(std::move(m), true);
  }
  lean_rows++;
  if (h != height) return 1;
  View v;
  v.height = h;
  v.round = r;
  return quorumProbe(type, v);
}

// addLeanRow for a run of rows of the current view: the counters of the view are read once, moved per row while the store
// takes the rows under one lock, and written back once; results[at[k]] = 1 / 2 as AddMessage would have signalled after row k.
void HotPath::addLeanRun(uint32_t type, const std::vector<const LeanRow *> &rows, const std::vector<size_t> &at,
                         const std::shared_ptr<const void> &backing, int8_t *results) {
  View v;
  v.height = height;
  v.round = round;
  auto rebuild = [&]() { return messages.SendersOf(v, (MessageType)type); };
  auto pc = quorumIndex.Get(type, height, round, rebuild, validatorManager);
  unsigned __int128 power = pc.first, dpower = 0;
  size_t dcount = 0;
  const bool vm_ok = validatorManager.initialized();
  const unsigned __int128 quorum = vm_ok ? validatorManager.quorum() : 0;
  // HasPrepareQuorum: the proposer's power joins the set; a PREPARE from the proposer voids it
  std::string_view proposer;
  unsigned __int128 w_proposer = 0;
  if (type == PREPARE && proposalMessage) {
    proposer = std::string_view(proposalMessage->from.data(), proposalMessage->from.size());
    w_proposer = validatorManager.powerOf(proposer);
  }
  int proposer_prepared = -1;  // not looked at yet
  const size_t taken = messages.AddLeanRun(
      type, height, round, rows.data(), rows.size(), backing, closure_epoch_, valset_epoch_,
      [&](size_t k, bool fresh, const LeanView &lv, const SenderMap *objs) {
        if (fresh) {
          const std::string_view from = rows[k]->from();
          const uint64_t w = rows[k]->sender_hash ? validatorManager.powerOf(from, rows[k]->sender_hash) : validatorManager.powerOf(from);
          power += w;
          dpower += w;
          dcount++;
          if (proposer_prepared == 0 && from == proposer) proposer_prepared = 1;
        }
        bool q = false;
        if (type == COMMIT) {
          q = vm_ok && power >= quorum;
        } else if (proposalMessage && vm_ok && power + w_proposer >= quorum) {
          // messages.Has: the proposer's PREPARE may be held as a row or as an object (round-3 advice: rows only gave a
          // spurious quorum signal, which handlePrepare then took back)
          if (proposer_prepared < 0)
            proposer_prepared = (lv.contains(proposer) || (objs && objs->contains(bytes::view(proposer.data(), proposer.size())))) ? 1 : 0;
          q = proposer_prepared == 0;
        }
        results[at[k]] = q ? 2 : 1;
      });
  if (taken == 0) {  // the view's rows were judged against other epochs: message by message
    for (size_t k = 0; k < rows.size(); k++) results[at[k]] = (int8_t)addLeanRow(type, height, round, *rows[k], backing);
    return;
  }
  quorumIndex.Add(type, height, round, dpower, dcount);
  lean_rows += taken;
}

// handlePrepare / handleCommit over a view that holds rows: the rows' closure verdicts came with them, so their walk is a
// filter; the few objects the view may hold besides (messages the backend did not vouch for byte by byte) take the batch
// walk; the quorum is the index's (the store's hooks followed the prunes).
bool HotPath::handleLean(const View &view, MessageType type, bool &quorum) {
  if (!(use_lean && use_sets && index_enabled_ && use_batch && batch)) {
    (void)messages.LeanFor(view, type, 0, 0);  // (epoch 0 never matches: any rows become objects for the generic walk)
    return false;
  }
  const Proposal *proposal = getProposal();
  syncClosureKey(proposal);
  LeanView *lv = messages.LeanFor(view, type, closure_epoch_, valset_epoch_);
  if (!lv) return false;
  size_t row_hits = 0;
  const size_t left = messages.FilterLean(view, type, [&](const LeanRow &row) {
    row_hits++;
    return row.closure != 0;
  });
  closure_hits = 0;
  std::vector<MsgPtr> objects = messages.GetValidMessagesBatch(
      view, type, [&](const std::vector<MsgPtr> &all) { return closureVerdicts(proposal, type, all); }, /*objects_only=*/true);
  closure_hits += row_hits;
  quorum = false;
  if (validatorManager.initialized()) {
    auto rebuild = [&]() { return messages.SendersOf(view, type); };
    auto pc = quorumIndex.Get(type, view.height, view.round, rebuild, validatorManager);
    if (pc.second != left + objects.size()) {  // (cannot happen: the hooks follow every change) — the generic walk is the authority
      (void)messages.LeanFor(view, type, 0, 0);
      return false;
    }
    if (type == PREPARE) {
      quorum = proposalMessage && !messages.Has(view, PREPARE, proposalMessage->from) &&
               pc.first + validatorManager.powerOf(proposalMessage->from) >= validatorManager.quorum();
    } else {
      quorum = pc.first >= validatorManager.quorum();
    }
    if (device_quorum) quorum = quorumDecision(type, messages.SendersOf(view, type), quorum);
  }
  if (!quorum) return true;
  if (type == PREPARE) {
    preparedMessages = std::move(objects);  // (+ the rows: PreparedWire())
    prepared_as_rows = true;
    prepared_view = view;
    stateName = StateName::commit;
  } else {
    // ExtractCommittedSeals (messages/helpers.go:22-35): {Signer: From, Signature: committedSeal} of every survivor — read
    // off the rows' bytes (the seal list keeps the buffers alive), extracted from the objects
    std::vector<std::optional<CommittedSeal>> seals;
    if (!extract_committed_seals(objects, seals)) {  // safe check, ibft.go:952-958
      quorum = false;
      return true;
    }
    committedSeals = std::move(seals);  // (+ the rows': CommittedSeals() / PackCommittedSeals())
    committed_as_rows = true;
    committed_view = view;
    stateName = StateName::fin;
  }
  return true;
}

// the seals of the last successful handleCommit: the objects' (extracted then) and, when the view held rows, the rows' — read
// off their bytes now (the seal list keeps the buffers alive)
const std::vector<std::optional<CommittedSeal>> &HotPath::CommittedSeals() {
  if (committed_as_rows) {
    committed_as_rows = false;
    if (LeanView *lv = messages.LeanFor(committed_view, COMMIT, closure_epoch_, valset_epoch_)) {
      committedSeals.reserve(committedSeals.size() + lv->size());
      lv->for_each([&](const LeanRow &row) {
        CommittedSeal cs{bytes::view((const char *)row.wire + row.from_off, row.from_len),
                         bytes::view((const char *)row.wire + row.seal_off, row.seal_len), nullptr, lv->buffers[row.buf]};
        committedSeals.emplace_back(std::move(cs));
      });
    }
  }
  return committedSeals;
}
// the same list in the C API's packing ({u8 present, u32 length + signer, u32 length + signature} per seal), rows straight
// from their bytes
size_t HotPath::PackCommittedSeals(bytes &out) {
  size_t count = 0;
  auto put = [&](const char *signer, size_t ls, const char *sig, size_t lg) {
    const uint32_t a = (uint32_t)ls, b = (uint32_t)lg;
    out.push_back(1);
    out.append((const char *)&a, 4);
    out.append(signer, ls);
    out.append((const char *)&b, 4);
    out.append(sig, lg);
    count++;
  };
  LeanView *lv = committed_as_rows ? messages.LeanFor(committed_view, COMMIT, closure_epoch_, valset_epoch_) : nullptr;
  size_t total = 0;
  for (auto &s : committedSeals) total += 9 + (s ? s->signer.size() + s->signature.size() : 0);
  if (lv) total += lv->size() * (9 + 20 + 65);
  out.reserve(out.size() + total);
  for (auto &s : committedSeals) {
    if (s) {
      put(s->signer.data(), s->signer.size(), s->signature.data(), s->signature.size());
    } else {
      const uint32_t z = 0;
      out.push_back(0);
      out.append((const char *)&z, 4);
      out.append((const char *)&z, 4);
      count++;
    }
  }
  if (lv)
    lv->for_each([&](const LeanRow &row) {
      put((const char *)row.wire + row.from_off, row.from_len, (const char *)row.wire + row.seal_off, row.seal_len);
    });
  return count;
}

std::vector<bytes> HotPath::PreparedWire() {
  std::vector<bytes> out;
  if (prepared_as_rows) {
    if (LeanView *lv = messages.LeanFor(prepared_view, PREPARE, closure_epoch_, valset_epoch_)) {
      lv->for_each([&](const LeanRow &row) { out.emplace_back((const char *)row.wire, row.len); });
      for (auto &m : preparedMessages) out.push_back(encode(*m));
      return out;
    }
    // the rows were turned into objects meanwhile: the objects of the view are the prepared messages
    for (auto &m : messages.GetValidMessages(prepared_view, PREPARE, [](const IbftMessage &) { return true; })) out.push_back(encode(*m));
    return out;
  }
  for (auto &m : preparedMessages) out.push_back(encode(*m));
  return out;
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
    if (use_sets && closureKnown(*all[k])) {  // judged completely when it arrived
      v[k] = all[k]->verdicts.closure;
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
  prepared_as_rows = false;
  {
    bool q = false;
    if (handleLean(view, PREPARE, q)) return q;
  }
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
  if (!hasQuorumOfStoredView(view, PREPARE, prepareMessages)) return false;
  // sendCommitMessage(view) is the Go side's business; finalizePrepare: state.go
  preparedMessages = prepareMessages;
  stateName = StateName::commit;
  return true;
}

bool HotPath::handleCommit(const View &view) {
  {
    bool q = false;
    if (handleLean(view, COMMIT, q)) return q;
  }
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
  if (!hasQuorumOfStoredView(view, COMMIT, commitMessages)) return false;
  std::vector<std::optional<CommittedSeal>> seals;
  if (!extract_committed_seals(commitMessages, seals)) return false;  // safe check, ibft.go:952-958
  committedSeals = std::move(seals);
  committed_as_rows = false;
  stateName = StateName::fin;
  return true;
}

}  // namespace ibft

// ---- certificate checks ------------------------------------------------------------------------
namespace ibft {

bool HotPath::isValidValidatorCached(const IbftMessage &m) {
  if (!sender_verdict_.empty()) {
    auto it = sender_verdict_.find(&m);
    if (it != sender_verdict_.end()) return it->second;
  }
  if (senderKnown(m)) return m.verdicts.sender;  // judged when the message that carries it arrived (IngestWire, use_certs)
  return verifier->IsValidValidator(m);
}

void HotPath::prefetchSenders(const std::vector<const IbftMessage *> &all) {
  sender_verdict_.clear();
  last_cert_senders = 0;
  cert_hits = 0;
  std::vector<const IbftMessage *> msgs;  // what the arrival-time verdicts cannot answer
  for (const IbftMessage *m : all) {
    if (senderKnown(*m))
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
  sender_verdict_.reserve(msgs.size() * 2);
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
  if (!validatorManager.HasQuorumOf(all)) return false;
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
  std::vector<const IbftMessage *> carriers;  // the message each hash comes from
  // ExtractProposalHash on a nil message would panic in the reference; treat as a nil hash
  hashes.push_back(certificate->proposal_message ? extract_proposal_hash(*certificate->proposal_message) : nullptr);
  carriers.push_back(certificate->proposal_message.get());
  for (auto &m : certificate->prepare_messages) {
    hashes.push_back(m ? extract_prepare_hash(*m) : nullptr);
    carriers.push_back(m.get());
  }
  {  // answered by handleRoundChangeMessage's pre-pass, or when the carrying message arrived
    bool all_known = true, all_ok = true;
    for (size_t k = 0; k < hashes.size() && all_known; k++) {
      bool ok = false;
      all_known = lookupHashVerdict(carriers[k], proposal, hashes[k], ok);
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
  if (!isValidProposalHashCached(msg, proposal, proposalHash)) return false;
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
  if (use_rc_rows && msg.verdicts.pp_epoch == valset_epoch_ && msg.view && msg.view->height == height && msg.view->round == round) {
    // everything about the RoundChangeCertificate was decided from the backend's rows when the message arrived
    // (proposalVerdictFromRows) — the certificate itself was never decoded; what depends on this node and on the
    // application is asked now
    const ProposalVerdict &pv = msg.proposal_verdict.get();
    if (!validateProposalCommon(msg, view)) return false;
    if (!pv.ok) return false;
    if (verifier->IsProposer(verifier->ID(), height, round)) return false;
    cert_hits = pv.rows;
    if (!pv.has_prepared) return true;
    Proposal p2;
    p2.raw_proposal = proposal->raw_proposal;
    p2.round = pv.max_round;
    const bytes expected = bytes::view((const char *)pv.hash, 32);
    return verifier->IsValidProposalHash(&p2, &expected);
  }
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

bool HotPath::isValidProposalHashCached(const IbftMessage &m, const Proposal *proposal, const bytes *hash) {
  bool ok = false;
  if (lookupHashVerdict(&m, proposal, hash, ok)) return ok;
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
      known = lookupHashVerdict(cert->proposal_message.get(), proposal,
                                cert->proposal_message ? extract_proposal_hash(*cert->proposal_message) : nullptr, ok);
      for (auto &m : cert->prepare_messages)
        known = known && lookupHashVerdict(m.get(), proposal, m ? extract_prepare_hash(*m) : nullptr, ok);
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
  size_t row_hits = 0;
  auto rowsDecided = [&](const IbftMessage &msg) { return use_rc_rows && msg.verdicts.rc_epoch == valset_epoch_; };
  auto isValidMsgFn = [&](const IbftMessage &msg) {
    if (rowsDecided(msg)) return msg.verdicts.rc_ok != 0;  // decided from the backend's rows when the message arrived
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
      std::vector<MsgPtr> undecided;
      for (auto &rc : all) {
        if (rowsDecided(*rc)) {
          row_hits += rc->verdicts.rc_rows;  // (its nested messages stay undecoded)
          continue;
        }
        collect_pc(extract_latest_pc(*rc), need);
        undecided.push_back(rc);
      }
      prefetchSenders(need);
      if (cert_hits < need.size() && sender_verdict_.empty()) fallbacks++;  // something was asked and the batch failed
      cert_hits += row_hits;  // sender verdicts the walk does not ask for at all
      prefetchCertificateHashes(undecided);
    };
  else
    sender_verdict_.clear();
  std::vector<MsgPtr> out = messages.GetExtendedRCC(h, isValidMsgFn, isValidRCCFn, prepass);
  sender_verdict_.clear();
  hash_verdict_.clear();
  return out;
}

}  // namespace ibft
