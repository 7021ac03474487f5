// proto.cpp — wire encode/decode + Extract* helpers (see proto.hpp for the reference lines).
#include "proto.hpp"

#include <cstdlib>
#include <cstring>
#include <set>
#include <string_view>
#include <unordered_set>

namespace ibft {

namespace {

void put_varint(bytes &o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back((char)(uint8_t)(v | 0x80));
    v >>= 7;
  }
  o.push_back((char)(uint8_t)v);
}
void put_len_field(bytes &o, uint32_t num, const bytes &data) {
  put_varint(o, ((uint64_t)num << 3) | 2);
  put_varint(o, data.size());
  o += data;
}
void put_bytes_field(bytes &o, uint32_t num, const bytes &data) {  // proto3: empty omitted
  if (!data.empty()) put_len_field(o, num, data);
}
void put_varint_field(bytes &o, uint32_t num, uint64_t v) {  // proto3: zero omitted
  if (v) {
    put_varint(o, (uint64_t)num << 3);
    put_varint(o, v);
  }
}

// decode mode of the current thread: byte fields become views into tl_backing (which every decoded message keeps), or
// owned copies (standalone Proposal / values with nothing to keep a buffer alive)
thread_local const std::shared_ptr<const void> *tl_backing = nullptr;
thread_local bool tl_defer_certificate = false;  // decode_in(..., defer_certificate)
inline void set_bytes(bytes &dst, const uint8_t *q, size_t l) {
  if (tl_backing)
    dst = bytes::view((const char *)q, l);
  else
    dst.assign((const char *)q, l);
}

struct Reader {
  const uint8_t *p, *end;
  bool varint(uint64_t &v) {
    v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) return false;
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return true;
    }
    return false;
  }
  bool len_delim(const uint8_t *&q, size_t &n) {
    uint64_t l;
    if (!varint(l) || l > (uint64_t)(end - p)) return false;
    q = p;
    n = (size_t)l;
    p += l;
    return true;
  }
};

// Skip (and copy verbatim into `unknown`) one field whose tag has just been read.
bool keep_unknown(Reader &r, uint64_t tag, const uint8_t *field_start, bytes &unknown) {
  uint64_t v;
  const uint8_t *q;
  size_t n;
  switch (tag & 7) {
    case 0:
      if (!r.varint(v)) return false;
      break;
    case 1:
      if (r.end - r.p < 8) return false;
      r.p += 8;
      break;
    case 2:
      if (!r.len_delim(q, n)) return false;
      break;
    case 5:
      if (r.end - r.p < 4) return false;
      r.p += 4;
      break;
    default:
      return false;  // groups / invalid wire types
  }
  unknown.append((const char *)field_start, (size_t)(r.p - field_start));
  return true;
}

bool decode_view(const uint8_t *p, size_t n, View &v) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    if (tag == ((1u << 3) | 0)) {
      if (!r.varint(v.height)) return false;
    } else if (tag == ((2u << 3) | 0)) {
      if (!r.varint(v.round)) return false;
    } else if ((tag >> 3) == 1 || (tag >> 3) == 2) {
      return false;  // known field, wrong wire type
    } else if (!keep_unknown(r, tag, start, v.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_proposal(const uint8_t *p, size_t n, Proposal &o) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(o.raw_proposal, q, l);
    } else if (tag == ((2u << 3) | 0)) {
      if (!r.varint(o.round)) return false;
    } else if ((tag >> 3) == 1 || (tag >> 3) == 2) {
      return false;
    } else if (!keep_unknown(r, tag, start, o.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_msg(const uint8_t *p, size_t n, IbftMessage &m, int depth);

bool decode_pc(const uint8_t *p, size_t n, PreparedCertificate &pc, int depth) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      if (!pc.proposal_message) pc.proposal_message = std::make_shared<IbftMessage>();
      if (!decode_msg(q, l, *pc.proposal_message, depth + 1)) return false;
    } else if (tag == ((2u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      auto m = std::make_shared<IbftMessage>();
      if (!decode_msg(q, l, *m, depth + 1)) return false;
      pc.prepare_messages.push_back(std::move(m));
    } else if ((tag >> 3) == 1 || (tag >> 3) == 2) {
      return false;
    } else if (!keep_unknown(r, tag, start, pc.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_rcc(const uint8_t *p, size_t n, RoundChangeCertificate &rcc, int depth) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      auto m = std::make_shared<IbftMessage>();
      if (!decode_msg(q, l, *m, depth + 1)) return false;
      rcc.round_change_messages.push_back(std::move(m));
    } else if ((tag >> 3) == 1) {
      return false;
    } else if (!keep_unknown(r, tag, start, rcc.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_preprepare(const uint8_t *p, size_t n, PrePrepareMessage &o, int depth) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      if (!o.proposal) o.proposal.emplace();
      if (!decode_proposal(q, l, *o.proposal)) return false;
    } else if (tag == ((2u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(o.proposal_hash, q, l);
    } else if (tag == ((3u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      if (tl_defer_certificate && depth == 0 && tl_backing && !o.certificate && !o.certificate_deferred) {
        o.certificate_deferred = true;  // (a second occurrence of the field — never canonical — is merged below)
        o.certificate_wire = bytes::view((const char *)q, l);
        o.certificate_backing = *tl_backing;
        continue;
      }
      if (o.certificate_deferred && !o.realise_certificate()) return false;
      if (!o.certificate) o.certificate.emplace();
      if (!decode_rcc(q, l, *o.certificate, depth)) return false;
    } else if ((tag >> 3) >= 1 && (tag >> 3) <= 3) {
      return false;
    } else if (!keep_unknown(r, tag, start, o.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_prepare(const uint8_t *p, size_t n, PrepareMessage &o) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(o.proposal_hash, q, l);
    } else if ((tag >> 3) == 1) {
      return false;
    } else if (!keep_unknown(r, tag, start, o.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_commit(const uint8_t *p, size_t n, CommitMessage &o) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(o.proposal_hash, q, l);
    } else if (tag == ((2u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(o.committed_seal, q, l);
    } else if ((tag >> 3) == 1 || (tag >> 3) == 2) {
      return false;
    } else if (!keep_unknown(r, tag, start, o.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_round_change(const uint8_t *p, size_t n, RoundChangeMessage &o, int depth) {
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    if (tag == ((1u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      if (!o.last_prepared_proposal) o.last_prepared_proposal.emplace();
      if (!decode_proposal(q, l, *o.last_prepared_proposal)) return false;
    } else if (tag == ((2u << 3) | 2)) {
      if (!r.len_delim(q, l)) return false;
      if (tl_defer_certificate && depth == 0 && tl_backing && !o.latest_prepared_certificate && !o.certificate_deferred) {
        o.certificate_deferred = true;  // (a second occurrence of the field — never canonical — is merged below)
        o.certificate_wire = bytes::view((const char *)q, l);
        o.certificate_backing = *tl_backing;
        continue;
      }
      if (o.certificate_deferred && !o.realise_certificate()) return false;
      if (!o.latest_prepared_certificate) o.latest_prepared_certificate.emplace();
      if (!decode_pc(q, l, *o.latest_prepared_certificate, depth)) return false;
    } else if ((tag >> 3) == 1 || (tag >> 3) == 2) {
      return false;
    } else if (!keep_unknown(r, tag, start, o.unknown)) {
      return false;
    }
  }
  return true;
}

bool decode_msg(const uint8_t *p, size_t n, IbftMessage &m, int depth) {
  if (depth > 64) return false;  // protobuf-go's default recursion limit is far above any real nesting
  if (tl_backing) m.backing = *tl_backing;
  Reader r{p, p + n};
  while (r.p < r.end) {
    const uint8_t *start = r.p;
    uint64_t tag;
    if (!r.varint(tag)) return false;
    const uint8_t *q;
    size_t l;
    const uint32_t num = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (num == 1 && wt == 2) {
      if (!r.len_delim(q, l)) return false;
      if (!m.view) m.view.emplace();
      if (!decode_view(q, l, *m.view)) return false;
    } else if (num == 2 && wt == 2) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(m.from, q, l);
    } else if (num == 3 && wt == 2) {
      if (!r.len_delim(q, l)) return false;
      set_bytes(m.signature, q, l);
    } else if (num == 4 && wt == 0) {
      uint64_t v;
      if (!r.varint(v)) return false;
      m.type = (uint32_t)v;
    } else if (num >= 5 && num <= 8 && wt == 2) {
      if (!r.len_delim(q, l)) return false;
      // oneof: the last member on the wire wins; a repeated member merges
      PayloadKind k = num == 5 ? PayloadKind::PREPREPARE : num == 6 ? PayloadKind::PREPARE
                      : num == 7 ? PayloadKind::COMMIT : PayloadKind::ROUND_CHANGE;
      if (m.kind != k) {
        if (m.kind != PayloadKind::NONE) {  // another member was set before: the last one on the wire wins
          m.preprepare_.reset();
          m.prepare = {};
          m.commit = {};
          m.round_change_.reset();
        }
        m.kind = k;
      }
      bool ok = k == PayloadKind::PREPREPARE  ? decode_preprepare(q, l, m.preprepare_mut(), depth)
                : k == PayloadKind::PREPARE   ? decode_prepare(q, l, m.prepare)
                : k == PayloadKind::COMMIT    ? decode_commit(q, l, m.commit)
                                              : decode_round_change(q, l, m.round_change_mut(), depth);
      if (!ok) return false;
    } else if (num >= 1 && num <= 8) {
      return false;  // known field with the wrong wire type
    } else if (!keep_unknown(r, tag, start, m.unknown)) {
      return false;
    }
  }
  return true;
}

}  // namespace

bytes encode(const View &v) {
  bytes o;
  put_varint_field(o, 1, v.height);
  put_varint_field(o, 2, v.round);
  o += v.unknown;
  return o;
}
bytes encode(const Proposal &p) {
  bytes o;
  put_bytes_field(o, 1, p.raw_proposal);
  put_varint_field(o, 2, p.round);
  o += p.unknown;
  return o;
}
bytes encode(const PreparedCertificate &pc) {
  bytes o;
  if (pc.proposal_message) put_len_field(o, 1, encode(*pc.proposal_message));
  for (const auto &m : pc.prepare_messages) put_len_field(o, 2, m ? encode(*m) : bytes());
  o += pc.unknown;
  return o;
}
bytes encode(const RoundChangeCertificate &rcc) {
  bytes o;
  for (const auto &m : rcc.round_change_messages) put_len_field(o, 1, m ? encode(*m) : bytes());
  o += rcc.unknown;
  return o;
}
static bytes encode(const PrePrepareMessage &p) {
  bytes o;
  if (p.proposal) put_len_field(o, 1, encode(*p.proposal));
  put_bytes_field(o, 2, p.proposal_hash);
  if (p.certificate_deferred)
    put_len_field(o, 3, p.certificate_wire);  // (canonical by the condition under which a certificate stays deferred)
  else if (p.certificate)
    put_len_field(o, 3, encode(*p.certificate));
  o += p.unknown;
  return o;
}
static bytes encode(const PrepareMessage &p) {
  bytes o;
  put_bytes_field(o, 1, p.proposal_hash);
  o += p.unknown;
  return o;
}
static bytes encode(const CommitMessage &c) {
  bytes o;
  put_bytes_field(o, 1, c.proposal_hash);
  put_bytes_field(o, 2, c.committed_seal);
  o += c.unknown;
  return o;
}
static bytes encode(const RoundChangeMessage &r) {
  bytes o;
  if (r.last_prepared_proposal) put_len_field(o, 1, encode(*r.last_prepared_proposal));
  if (r.certificate_deferred)
    put_len_field(o, 2, r.certificate_wire);  // (canonical by the condition under which a certificate stays deferred)
  else if (r.latest_prepared_certificate)
    put_len_field(o, 2, encode(*r.latest_prepared_certificate));
  o += r.unknown;
  return o;
}

bytes encode(const IbftMessage &m, bool with_signature) {
  bytes o;
  if (m.view) put_len_field(o, 1, encode(*m.view));
  put_bytes_field(o, 2, m.from);
  if (with_signature) put_bytes_field(o, 3, m.signature);
  put_varint_field(o, 4, m.type);
  switch (m.kind) {
    case PayloadKind::PREPREPARE: put_len_field(o, 5, encode(m.preprepare())); break;
    case PayloadKind::PREPARE: put_len_field(o, 6, encode(m.prepare)); break;
    case PayloadKind::COMMIT: put_len_field(o, 7, encode(m.commit)); break;
    case PayloadKind::ROUND_CHANGE: put_len_field(o, 8, encode(m.round_change())); break;
    case PayloadKind::NONE: break;
  }
  o += m.unknown;
  return o;
}

// The shape every honest PREPARE / COMMIT has — View{height, round}, From, Signature, type, PrepareMessage{hash} or
// CommitMessage{hash, seal}, each once, in field order, one-byte lengths except the payload's — recognised without the
// general walk; anything else (false) goes to it.  Same Peek as peek_general() on what it accepts
// (tests/test_host_semantics.py).
static bool peek_regular(const uint8_t *p, size_t n, Peek &out) {
  size_t i = 0;
  auto varint = [&](uint64_t &v) {
    v = 0;
    for (int sh = 0; sh < 64 && i < n; sh += 7) {
      const uint8_t b = p[i++];
      v |= (uint64_t)(b & 0x7F) << sh;
      if (!(b & 0x80)) return sh < 63 || b <= 1;
    }
    return false;
  };
  if (n < 8 || p[0] != 0x0A || p[1] >= 0x80) return false;
  const size_t view_end = 2 + p[1];
  if (view_end > n) return false;
  i = 2;
  uint64_t height = 0, round = 0;
  if (i < view_end && p[i] == 0x08) {
    i++;
    if (!varint(height) || i > view_end) return false;
  }
  if (i < view_end && p[i] == 0x10) {
    i++;
    if (!varint(round) || i > view_end) return false;
  }
  if (i != view_end) return false;
  if (i + 2 > n || p[i] != 0x12 || p[i + 1] >= 0x80 || p[i + 1] == 0) return false;
  const uint32_t from_off = (uint32_t)i + 2, from_len = p[i + 1];
  i += 2 + from_len;
  if (i + 2 > n || p[i] != 0x1A || p[i + 1] >= 0x80 || p[i + 1] == 0) return false;
  const uint32_t sig_off = (uint32_t)i + 2, sig_len = p[i + 1];
  i += 2 + sig_len;
  if (i + 2 > n || p[i] != 0x20 || (p[i + 1] != PREPARE && p[i + 1] != COMMIT)) return false;
  const uint32_t type = p[i + 1];
  i += 2;
  if (i + 2 > n || p[i] != (type == PREPARE ? 0x32 : 0x3A)) return false;
  i++;
  uint64_t plen;
  if (!varint(plen) || plen != n - i) return false;
  // PrepareMessage {1: proposalHash} / CommitMessage {1: proposalHash, 2: committedSeal}
  uint32_t hash_off = 0, hash_len = 0, seal_off = 0, seal_len = 0;
  if (i < n && p[i] == 0x0A) {
    if (i + 2 > n || p[i + 1] >= 0x80 || p[i + 1] == 0) return false;
    hash_off = (uint32_t)i + 2;
    hash_len = p[i + 1];
    i += 2 + hash_len;
  }
  if (type == COMMIT && i < n && p[i] == 0x12) {
    if (i + 2 > n || p[i + 1] >= 0x80 || p[i + 1] == 0) return false;
    seal_off = (uint32_t)i + 2;
    seal_len = p[i + 1];
    i += 2 + seal_len;
  }
  if (i != n) return false;
  out.ok = true;
  out.has_view = true;
  out.height = height;
  out.round = round;
  out.type = type;
  out.kind = type == PREPARE ? PayloadKind::PREPARE : PayloadKind::COMMIT;
  out.from_off = from_off; out.from_len = from_len;
  out.sig_off = sig_off; out.sig_len = sig_len;
  out.hash_off = hash_off; out.hash_len = hash_len;
  out.seal_off = seal_off; out.seal_len = seal_len;
  out.simple = true;
  return true;
}

Peek peek(const uint8_t *p, size_t n) {
  {
    Peek fast;
    if (peek_regular(p, n, fast)) return fast;
  }
  return peek_general(p, n);
}

Peek peek_general(const uint8_t *p, size_t n) {
  Peek out;
  bool repeated = false, payload_plain = true, seen_twice[9] = {false};
  int views = 0;
  Reader r{p, p + n};
  while (r.p < r.end) {
    uint64_t tag;
    if (!r.varint(tag)) return out;
    const uint32_t num = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    const uint8_t *q;
    size_t l;
    uint64_t v;
    if (num == 1 && wt == 2) {
      if (!r.len_delim(q, l)) return out;
      out.has_view = true;
      views++;
      Reader vr{q, q + l};
      while (vr.p < vr.end) {
        uint64_t vt;
        if (!vr.varint(vt)) return out;
        if (vt == ((1u << 3) | 0)) {
          if (!vr.varint(out.height)) return out;
        } else if (vt == ((2u << 3) | 0)) {
          if (!vr.varint(out.round)) return out;
        } else if ((vt >> 3) == 1 || (vt >> 3) == 2) {
          return out;
        } else {  // unknown field of View: skipped
          const uint8_t *uq;
          size_t ul;
          switch (vt & 7) {
            case 0: if (!vr.varint(v)) return out; break;
            case 1: if (vr.end - vr.p < 8) return out; vr.p += 8; break;
            case 2: if (!vr.len_delim(uq, ul)) return out; break;
            case 5: if (vr.end - vr.p < 4) return out; vr.p += 4; break;
            default: return out;
          }
        }
      }
    } else if ((num == 2 || num == 3) && wt == 2) {
      if (!r.len_delim(q, l)) return out;
      uint32_t &o = num == 2 ? out.from_off : out.sig_off, &n_ = num == 2 ? out.from_len : out.sig_len;
      if (n_ || seen_twice[num]) repeated = true;
      seen_twice[num] = true;
      o = (uint32_t)(q - p);
      n_ = (uint32_t)l;
    } else if (num == 4 && wt == 0) {
      if (!r.varint(v)) return out;
      out.type = (uint32_t)v;
    } else if (num >= 5 && num <= 8 && wt == 2) {
      if (!r.len_delim(q, l)) return out;
      if (out.kind != PayloadKind::NONE) repeated = true;
      out.kind = num == 5 ? PayloadKind::PREPREPARE : num == 6 ? PayloadKind::PREPARE
                 : num == 7 ? PayloadKind::COMMIT : PayloadKind::ROUND_CHANGE;
      if (num == 6 || num == 7) {  // PrepareMessage {1: proposalHash} / CommitMessage {1: proposalHash, 2: committedSeal}
        Reader pr{q, q + l};
        bool h_seen = false, s_seen = false;
        while (pr.p < pr.end) {
          uint64_t pt;
          const uint8_t *fq;
          size_t fl;
          if (!pr.varint(pt)) return out;
          if (pt == ((1u << 3) | 2)) {
            if (!pr.len_delim(fq, fl)) return out;
            if (h_seen) repeated = true;
            h_seen = true;
            out.hash_off = (uint32_t)(fq - p);
            out.hash_len = (uint32_t)fl;
          } else if (num == 7 && pt == ((2u << 3) | 2)) {
            if (!pr.len_delim(fq, fl)) return out;
            if (s_seen) repeated = true;
            s_seen = true;
            out.seal_off = (uint32_t)(fq - p);
            out.seal_len = (uint32_t)fl;
          } else {
            payload_plain = false;  // a wrong wire type or an unknown field: the decoder decides
            break;
          }
        }
      }
    } else if (num >= 1 && num <= 8) {
      return out;  // known field with the wrong wire type
    } else {
      switch (wt) {
        case 0: if (!r.varint(v)) return out; break;
        case 1: if (r.end - r.p < 8) return out; r.p += 8; break;
        case 2: if (!r.len_delim(q, l)) return out; break;
        case 5: if (r.end - r.p < 4) return out; r.p += 4; break;
        default: return out;
      }
    }
  }
  out.ok = true;
  out.simple = !repeated && payload_plain && views <= 1;
  return out;
}

std::shared_ptr<const void> make_backing(const uint8_t *p, size_t n, const uint8_t **copy) {
  uint8_t *b = static_cast<uint8_t *>(malloc(n ? n : 1));
  if (n) memcpy(b, p, n);
  *copy = b;
  return std::shared_ptr<const void>(b, free);
}
namespace {
struct BackingScope {
  const std::shared_ptr<const void> *saved;
  explicit BackingScope(const std::shared_ptr<const void> *b) : saved(tl_backing) { tl_backing = b; }
  ~BackingScope() { tl_backing = saved; }
};
}  // namespace
bool RoundChangeMessage::realise_certificate() const {
  if (!certificate_deferred) return true;
  certificate_deferred = false;
  const std::shared_ptr<const void> keep = std::move(certificate_backing);
  const bytes w = std::move(certificate_wire);
  certificate_backing.reset();
  certificate_wire = bytes();
  BackingScope scope(&keep);
  const bool saved = tl_defer_certificate;
  tl_defer_certificate = false;
  PreparedCertificate pc;
  const bool ok = decode_pc((const uint8_t *)w.data(), w.size(), pc, 0);
  tl_defer_certificate = saved;
  if (ok) const_cast<RoundChangeMessage *>(this)->latest_prepared_certificate = std::move(pc);
  return ok;
}
bool PrePrepareMessage::realise_certificate() const {
  if (!certificate_deferred) return true;
  certificate_deferred = false;
  const std::shared_ptr<const void> keep = std::move(certificate_backing);
  const bytes w = std::move(certificate_wire);
  certificate_backing.reset();
  certificate_wire = bytes();
  BackingScope scope(&keep);
  const bool saved = tl_defer_certificate;
  tl_defer_certificate = false;
  RoundChangeCertificate rcc;
  const bool ok = decode_rcc((const uint8_t *)w.data(), w.size(), rcc, 0);
  tl_defer_certificate = saved;
  if (ok) const_cast<PrePrepareMessage *>(this)->certificate = std::move(rcc);
  return ok;
}
bool decode_in(const std::shared_ptr<const void> &backing, const uint8_t *p, size_t n, IbftMessage &out, bool defer_certificate) {
  BackingScope scope(&backing);
  struct DeferScope {
    bool saved;
    explicit DeferScope(bool d) : saved(tl_defer_certificate) { tl_defer_certificate = d; }
    ~DeferScope() { tl_defer_certificate = saved; }
  } defer_scope(defer_certificate);
  return decode_msg(p, n, out, 0);
}
bool decode(const uint8_t *p, size_t n, IbftMessage &out) {
  out = IbftMessage{};
  const uint8_t *q;
  std::shared_ptr<const void> backing = make_backing(p, n, &q);
  return decode_in(backing, q, n, out);
}
bool decode(const uint8_t *p, size_t n, PreparedCertificate &out) {
  out = PreparedCertificate{};
  const uint8_t *q;
  std::shared_ptr<const void> backing = make_backing(p, n, &q);  // kept by the messages of the certificate
  BackingScope scope(&backing);
  return decode_pc(q, n, out, 0);
}
bool decode(const uint8_t *p, size_t n, Proposal &out) {
  out = Proposal{};
  BackingScope scope(nullptr);  // nothing would keep a buffer alive: owned copies
  return decode_proposal(p, n, out);
}

// ---- Extract* ------------------------------------------------------------------------------
const bytes *extract_commit_hash(const IbftMessage &m) {
  if (m.type != COMMIT || m.kind != PayloadKind::COMMIT) return nullptr;
  return &m.commit.proposal_hash;
}
std::optional<CommittedSeal> extract_committed_seal(const IbftMessage &m) {
  if (m.kind != PayloadKind::COMMIT) return std::nullopt;  // only the payload is checked (helpers.go:39-42)
  // views: the seal is read while the message is alive (the list form below keeps the message)
  return CommittedSeal{bytes::view(m.from.data(), m.from.size()),
                       bytes::view(m.commit.committed_seal.data(), m.commit.committed_seal.size()), nullptr};
}
const bytes *extract_prepare_hash(const IbftMessage &m) {
  if (m.type != PREPARE || m.kind != PayloadKind::PREPARE) return nullptr;
  return &m.prepare.proposal_hash;
}
const Proposal *extract_proposal(const IbftMessage &m) {
  if (m.type != PREPREPARE || m.kind != PayloadKind::PREPREPARE) return nullptr;
  return m.preprepare().proposal ? &*m.preprepare().proposal : nullptr;
}
const bytes *extract_proposal_hash(const IbftMessage &m) {
  if (m.type != PREPREPARE || m.kind != PayloadKind::PREPREPARE) return nullptr;
  return &m.preprepare().proposal_hash;
}
const RoundChangeCertificate *extract_round_change_certificate(const IbftMessage &m) {
  if (m.type != PREPREPARE || m.kind != PayloadKind::PREPREPARE) return nullptr;
  if (m.preprepare().certificate_deferred) (void)m.preprepare().realise_certificate();
  return m.preprepare().certificate ? &*m.preprepare().certificate : nullptr;
}
const PreparedCertificate *extract_latest_pc(const IbftMessage &m) {
  if (m.type != ROUND_CHANGE || m.kind != PayloadKind::ROUND_CHANGE) return nullptr;
  if (m.round_change().certificate_deferred) (void)m.round_change().realise_certificate();
  return m.round_change().latest_prepared_certificate ? &*m.round_change().latest_prepared_certificate : nullptr;
}
const Proposal *extract_last_prepared_proposal(const IbftMessage &m) {
  if (m.type != ROUND_CHANGE || m.kind != PayloadKind::ROUND_CHANGE) return nullptr;
  return m.round_change().last_prepared_proposal ? &*m.round_change().last_prepared_proposal : nullptr;
}

bool extract_committed_seals(const std::vector<MsgPtr> &msgs, std::vector<std::optional<CommittedSeal>> &out) {
  out.clear();
  for (const auto &m : msgs) {
    if (m->type != COMMIT) {  // safe check, helpers.go:26-29
      out.clear();
      return false;
    }
    out.push_back(extract_committed_seal(*m));
    if (out.back()) out.back()->keep = m;  // the views stay valid for as long as the seal list lives
  }
  return true;
}

bool has_unique_senders(const std::vector<MsgPtr> &msgs) {
  if (msgs.empty()) return false;
  std::unordered_set<std::string_view, sv_hash> seen;
  seen.reserve(msgs.size() * 2);
  for (const auto &m : msgs)
    if (!seen.insert(std::string_view(m->from.data(), m->from.size())).second) return false;
  return true;
}

static bool bytes_equal(const bytes *a, const bytes *b) {  // bytes.Equal: nil == empty
  static const bytes empty;
  return (a ? *a : empty) == (b ? *b : empty);
}

bool are_valid_pc_messages(const std::vector<MsgPtr> &msgs, uint64_t height, uint64_t round_limit) {
  if (msgs.empty()) return false;
  // messages[0].View.Round: a nil View would panic in the reference; treat as invalid
  if (!msgs[0]->view) return false;
  const uint64_t round = msgs[0]->view->round;
  std::unordered_set<std::string_view, sv_hash> senders;
  senders.reserve(msgs.size() * 2);
  const bytes *hash = nullptr;
  for (const auto &m : msgs) {
    if (!m->view) return false;
    if (m->view->height != height) return false;
    if (m->view->round != round || m->view->round >= round_limit) return false;
    const bytes *extracted = nullptr;
    bool ok = false;
    if (m->type == PREPREPARE) {
      extracted = extract_proposal_hash(*m);
      ok = true;
    } else if (m->type == PREPARE) {
      extracted = extract_prepare_hash(*m);
      ok = true;
    }
    // `if hash == nil { hash = extractedHash }` — an absent proto3 bytes field is nil in Go;
    // an explicitly-encoded empty one is indistinguishable here and treated as nil too
    if (hash == nullptr || hash->empty()) hash = extracted;
    if (!ok || !bytes_equal(hash, extracted)) return false;
    if (!senders.insert(std::string_view(m->from.data(), m->from.size())).second) return false;
  }
  return true;
}

}  // namespace ibft
