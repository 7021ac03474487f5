// wire_dev.h — IbftMessage wire bytes → verifier columns, on the device.
//
// Product code.  The transport hands go-ibft protobuf bytes; before a signature can be checked the
// reference unmarshals the message and marshals it again without the signature
// (/root/reference/messages/proto/helper.go:12-27, PayloadNoSig) — per message, on the host.  For the
// two message kinds that make up the bulk of a round (PREPARE, COMMIT: N of each per round,
// /root/reference/messages/proto/messages.proto:24-44, 59-71) that work is a flat walk over ≈150
// bytes, so one lane does it:
//   * walk the top-level fields, the View and the Prepare/Commit body;
//   * vouch that the bytes are exactly what proto.Marshal would emit for the decoded message
//     (known fields only, ascending field numbers, one-byte tags, minimal varints, no explicit zero
//     scalars or empty byte strings, at most one member of the payload oneof) — only then is
//     "wire bytes minus the signature field" equal to PayloadNoSig;
//   * hash those bytes (Keccak-256) and copy From / Signature / proposal hash / committed seal into
//     the columns the recover kernels read.
// Anything else — PREPREPARE / ROUND_CHANGE payloads (nested certificates), unknown fields, any
// deviation from the canonical encoding — is NOT judged here: the row is marked NEEDS_HOST, gets
// verdict 0, and the caller sends it through the protobuf runtime and ibft_verify_senders.
//
// Certificates (the second half of this file): a PREPREPARE carries a
// RoundChangeCertificate, a ROUND_CHANGE a PreparedCertificate (messages.proto:46-57, 73-101) — messages
// inside messages, O(N²) signatures per round change (core/ibft.go:470-551, 683-788).  The DEEP form of the
// walk also accepts those two payloads: it checks the wrappers between the message and its nested
// messages (Proposal, PreparedCertificate, RoundChangeCertificate), records where the nested messages
// lie, and leaves every nested message to the lane that owns it on the next level
// (ibft_verify_certificates_wire, kernels.hip.h: cert_*_kernel).  A message is canonical iff its own
// fields are and every message below it is; only then is PayloadNoSig "its bytes minus field 3".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keccak_dev.h"

namespace wire {

constexpr uint8_t STATUS_OK = 0, STATUS_NEEDS_HOST = 1;
constexpr uint8_t KIND_NONE = 0, KIND_PREPREPARE = 5, KIND_PREPARE = 6, KIND_COMMIT = 7, KIND_ROUND_CHANGE = 8;  // oneof field numbers
// what a message of the certificate tree is to the message that contains it
constexpr uint8_t ROLE_ROOT = 0, ROLE_PC_PROPOSAL = 1, ROLE_PC_PREPARE = 2, ROLE_RCC_MESSAGE = 3;
constexpr uint8_t TREE_HAS_PROPOSAL = 1;  // a Proposal sub-message is present (PREPREPARE: proposal; ROUND_CHANGE: lastPreparedProposal)
constexpr uint8_t TREE_HAS_CERT = 2;      // a certificate wrapper is present (PREPREPARE: RCC; ROUND_CHANGE: PC)
constexpr uint8_t TREE_TOO_BIG = 4;       // canonical, but longer than the device hashes with one lane: digest left to the host
constexpr uint8_t TREE_PROPOSAL_TOO_BIG = 8;  // the Proposal it carries likewise: the hash bits that depend on it are not decided

// mirrors ibft_wire_row_t (include/ibftgpu.h), 80 bytes
struct row_info {
  uint64_t height, round;  // View (0 when absent or omitted)
  uint8_t status;          // STATUS_*
  uint8_t type;            // IbftMessage.type
  uint8_t payload_kind;    // KIND_*
  uint8_t has_view;
  uint8_t hash_len;        // proposal hash bytes present (≤ 32)
  uint8_t seal_len;        // committed seal bytes present (COMMIT; 255 = longer)
  uint8_t from_len;        // 255 = longer
  uint8_t sig_len;         // 255 = longer
  uint8_t from[20];
  uint8_t proposal_hash[32];
  uint8_t pad[4];
};
static_assert(sizeof(row_info) == 80, "ABI");

// minimal-form varint at m[pos..end); false: truncated, longer than 10 bytes, overflow, or padded
HD bool read_varint(const uint8_t *m, uint32_t end, uint32_t &pos, uint64_t &v) {
  v = 0;
  for (int i = 0; i < 10; i++) {
    if (pos >= end) return false;
    const uint8_t b = m[pos++];
    v |= (uint64_t)(b & 0x7Fu) << (7 * i);
    if (!(b & 0x80u)) return !(i > 0 && b == 0) && !(i == 9 && b > 1);
  }
  return false;
}
// length prefix of a length-delimited field whose tag byte was just consumed: body = m[pos..pos+len)
HD bool read_len(const uint8_t *m, uint32_t end, uint32_t &pos, uint32_t &len) {
  uint64_t l;
  if (!read_varint(m, end, pos, l)) return false;
  if (l > (uint64_t)(end - pos)) return false;
  len = (uint32_t)l;
  return true;
}

// what a PREPREPARE / ROUND_CHANGE message carries besides its own scalar fields (DEEP walk only); offsets
// relative to the message's first byte
struct tree_part {
  uint32_t raw_off, raw_len;    // Proposal.rawProposal
  uint32_t cert_off, cert_len;  // body of the certificate wrapper: the nested messages, length-prefixed, nothing else
  uint64_t proposal_round;      // Proposal.round
  uint8_t flags;                // TREE_*
};

struct parsed {
  row_info ri;
  uint32_t sig_field_start, sig_field_end;  // the whole field 3 (tag, length, bytes); equal when absent
  uint32_t sig_pos, seal_pos;               // payload bytes of Signature / CommittedSeal
  tree_part t;
};

// View { uint64 height = 1; uint64 round = 2; }
HD bool parse_view(const uint8_t *m, uint32_t pos, uint32_t end, row_info &ri) {
  uint32_t last = 0;
  while (pos < end) {
    const uint8_t tag = m[pos++];
    const uint32_t f = tag >> 3;
    if ((tag & 0x80u) || (tag & 7u) != 0 || f <= last || f > 2) return false;
    last = f;
    uint64_t v;
    if (!read_varint(m, end, pos, v) || v == 0) return false;  // a zero scalar is never emitted
    if (f == 1) ri.height = v; else ri.round = v;
  }
  return true;
}
// PrepareMessage { bytes proposal_hash = 1; }   CommitMessage { bytes proposal_hash = 1; bytes committed_seal = 2; }
HD bool parse_body(const uint8_t *m, uint32_t pos, uint32_t end, bool commit, parsed &p) {
  uint32_t last = 0;
  while (pos < end) {
    const uint8_t tag = m[pos++];
    const uint32_t f = tag >> 3;
    if ((tag & 0x80u) || (tag & 7u) != 2 || f <= last || f > (commit ? 2u : 1u)) return false;
    last = f;
    uint32_t len;
    if (!read_len(m, end, pos, len) || len == 0) return false;  // empty bytes are never emitted
    if (f == 1) {
      if (len > 32) return false;  // column width; the host path copes with odd lengths
      p.ri.hash_len = (uint8_t)len;
      for (uint32_t i = 0; i < len; i++) p.ri.proposal_hash[i] = m[pos + i];
    } else {
      p.ri.seal_len = len > 255 ? 255 : (uint8_t)len;
      p.seal_pos = pos;
    }
    pos += len;
  }
  return true;
}

// Proposal { bytes rawProposal = 1; uint64 round = 2; }   (messages.proto:103-110)
HD bool parse_proposal(const uint8_t *m, uint32_t pos, uint32_t end, tree_part &t) {
  uint32_t last = 0;
  while (pos < end) {
    const uint8_t tag = m[pos++];
    const uint32_t f = tag >> 3;
    if ((tag & 0x80u) || f <= last || f > 2) return false;
    last = f;
    if (f == 1) {
      uint32_t len;
      if ((tag & 7u) != 2 || !read_len(m, end, pos, len) || len == 0) return false;
      t.raw_off = pos;
      t.raw_len = len;
      pos += len;
    } else {
      uint64_t v;
      if ((tag & 7u) != 0 || !read_varint(m, end, pos, v) || v == 0) return false;
      t.proposal_round = v;
    }
  }
  return true;
}
// One element of a certificate wrapper —
//   PreparedCertificate    { IbftMessage proposalMessage = 1; repeated IbftMessage prepareMessages = 2; }  (messages.proto:84-92)
//   RoundChangeCertificate { repeated IbftMessage roundChangeMessages = 1; }                                (:96-99)
// — at pos: its tag and length prefix are checked (known field, ascending, proposalMessage at most once, minimal
// varint, inside the wrapper) and consumed; body = [pos, pos + len) afterwards.  `last` carries the previous field.
// BYTES is anything indexable by absolute position (the message itself, or a window of it held in LDS).
template <typename BYTES>
HD bool cert_child_header(const BYTES &m, uint32_t end, bool pc, uint32_t &last, uint32_t &pos, uint32_t &len, uint8_t &role) {
  // tag + length prefix: at most 6 bytes, fetched together (independent loads: one LDS round trip on the device) and decoded
  // from registers
  uint8_t h[6];
#pragma unroll
  for (int k = 0; k < 6; k++) h[k] = pos + (uint32_t)k < end ? m[pos + (uint32_t)k] : (uint8_t)0;
  const uint32_t avail = end - pos;  // ≥ 1
  const uint8_t tag = h[0];
  const uint32_t f = tag >> 3;
  if ((tag & 0x80u) || (tag & 7u) != 2 || f == 0 || f < last || f > (pc ? 2u : 1u)) return false;
  if (pc && f == 1 && last == 1) return false;
  last = f;
  uint64_t l = 0;
  uint32_t used = 1;
#pragma unroll
  for (int i = 0; i < 5; i++) {  // minimal varint, at most 5 bytes (a length ≥ 2^32 cannot lie inside the buffer)
    if (used >= avail) return false;
    const uint8_t b = h[1 + i];
    used++;
    l |= (uint64_t)(b & 0x7Fu) << (7 * i);
    if (!(b & 0x80u)) {
      if (i > 0 && b == 0) return false;
      break;
    }
    if (i == 4) return false;
  }
  pos += used;
  if (l > (uint64_t)(end - pos)) return false;
  len = (uint32_t)l;
  role = pc ? (f == 1 ? ROLE_PC_PROPOSAL : ROLE_PC_PREPARE) : ROLE_RCC_MESSAGE;
  return true;
}
// PrePrepareMessage  { Proposal proposal = 1; bytes proposalHash = 2; RoundChangeCertificate certificate = 3; }   (:46-57)
// RoundChangeMessage { Proposal lastPreparedProposal = 1; PreparedCertificate latestPreparedCertificate = 2; }   (:73-81)
// The certificate's nested messages are not walked here: cert_off / cert_len say where they are.
HD bool parse_tree_body(const uint8_t *m, uint32_t pos, uint32_t end, bool preprepare, parsed &p) {
  uint32_t last = 0;
  while (pos < end) {
    const uint8_t tag = m[pos++];
    const uint32_t f = tag >> 3;
    if ((tag & 0x80u) || (tag & 7u) != 2 || f <= last || f > (preprepare ? 3u : 2u)) return false;
    last = f;
    uint32_t len;
    if (!read_len(m, end, pos, len)) return false;  // a present sub-message is emitted even when it is empty
    if (f == 1) {
      p.t.flags |= TREE_HAS_PROPOSAL;
      if (!parse_proposal(m, pos, pos + len, p.t)) return false;
    } else if (preprepare && f == 2) {
      if (len == 0 || len > 32) return false;  // empty bytes are never emitted; the column is 32 wide
      p.ri.hash_len = (uint8_t)len;
      for (uint32_t i = 0; i < len; i++) p.ri.proposal_hash[i] = m[pos + i];
    } else {
      p.t.flags |= TREE_HAS_CERT;
      p.t.cert_off = pos;
      p.t.cert_len = len;
    }
    pos += len;
  }
  return true;
}

// IbftMessage { View view = 1; bytes from = 2; bytes signature = 3; MessageType type = 4;
//               oneof payload { PrePrepare = 5; Prepare = 6; Commit = 7; RoundChange = 8 } }
// DEEP = false: PREPREPARE / ROUND_CHANGE payloads are left to the host (the flat PREPARE / COMMIT routes).
template <bool DEEP>
HD parsed parse_message_t(const uint8_t *m, uint32_t n) {
  parsed p;
  p.t.raw_off = p.t.raw_len = p.t.cert_off = p.t.cert_len = 0;
  p.t.proposal_round = 0;
  p.t.flags = 0;
  p.ri.height = p.ri.round = 0;
  p.ri.status = STATUS_NEEDS_HOST;
  p.ri.type = 0;
  p.ri.payload_kind = KIND_NONE;
  p.ri.has_view = p.ri.hash_len = p.ri.seal_len = p.ri.from_len = p.ri.sig_len = 0;
  for (int i = 0; i < 20; i++) p.ri.from[i] = 0;
  for (int i = 0; i < 32; i++) p.ri.proposal_hash[i] = 0;
  for (int i = 0; i < 4; i++) p.ri.pad[i] = 0;
  p.sig_field_start = p.sig_field_end = 0;
  p.sig_pos = p.seal_pos = 0;
  uint32_t pos = 0, last = 0;
  bool sig_seen = false;
  while (pos < n) {
    const uint32_t field_start = pos;
    const uint8_t tag = m[pos++];
    const uint32_t f = tag >> 3, wt = tag & 7u;
    if ((tag & 0x80u) || f <= last || f == 0 || f > 8) return p;  // unknown / out of order / repeated
    if (last >= 5) return p;                                      // second member of the oneof
    last = f;
    if (f == 4) {
      uint64_t v;
      if (wt != 0 || !read_varint(m, n, pos, v) || v == 0 || v > 255) return p;
      p.ri.type = (uint8_t)v;
      continue;
    }
    uint32_t len;
    if (wt != 2 || !read_len(m, n, pos, len)) return p;
    switch (f) {
      case 1:
        p.ri.has_view = 1;
        if (!parse_view(m, pos, pos + len, p.ri)) return p;
        break;
      case 2:
        if (len == 0) return p;
        p.ri.from_len = len > 255 ? 255 : (uint8_t)len;
        for (uint32_t i = 0; i < (len < 20 ? len : 20u); i++) p.ri.from[i] = m[pos + i];
        break;
      case 3:
        if (len == 0) return p;
        sig_seen = true;
        p.ri.sig_len = len > 255 ? 255 : (uint8_t)len;
        p.sig_field_start = field_start;
        p.sig_field_end = pos + len;
        p.sig_pos = pos;
        break;
      case 6:
      case 7:
        p.ri.payload_kind = (uint8_t)f;
        if (!parse_body(m, pos, pos + len, f == 7, p)) return p;
        break;
      default:  // 5 PREPREPARE, 8 ROUND_CHANGE: nested proposals and certificates
        if (!DEEP) return p;  // host
        p.ri.payload_kind = (uint8_t)f;
        if (!parse_tree_body(m, pos, pos + len, f == 5, p)) return p;
        break;
    }
    pos += len;
  }
  if (!sig_seen) p.sig_field_start = p.sig_field_end = n;  // nothing to cut out
  p.ri.status = STATUS_OK;
  return p;
}
HD parsed parse_message(const uint8_t *m, uint32_t n) { return parse_message_t<false>(m, n); }

// Keccak-256 of m[0..cut0) ‖ m[cut1..n): PayloadNoSig of a canonical message
HD void hash_without(const uint8_t *m, uint32_t n, uint32_t cut0, uint32_t cut1, uint64_t out4[4]) {
  const uint32_t gap = cut1 - cut0, total = n - gap;
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  uint32_t done = 0;
  for (;;) {
    const uint32_t left = total - done;
    const bool last = left < 136;
    for (int i = 0; i < 17; i++) {
      uint64_t w = 0;
      for (int b = 0; b < 8; b++) {
        const uint32_t off = 8 * i + b;
        uint64_t byte = 0;
        if (off < left) {
          const uint32_t v = done + off;  // index in the virtual concatenation
          byte = m[v < cut0 ? v : v + gap];
        }
        if (last && off == left) byte ^= 0x01u;
        if (last && off == 135) byte ^= 0x80u;
        w |= byte << (8 * b);
      }
      s[i] ^= w;
    }
    keccak::f1600(s);
    if (last) break;
    done += 136;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out4[i] = s[i];
}

// one row: parse, hash, fill the verifier columns.  pre_flag ≠ 0 ⇒ the sender check must answer 0
// for this row (not canonical here, or From / Signature of a length no signature check can pass).
HD void process_row(const uint8_t *m, uint32_t n, row_info *ri_out, uint8_t *digest32, uint8_t *sig65, uint8_t *from20,
                    uint8_t *seal65, uint8_t *pre_flag) {
  const parsed p = parse_message(m, n);
  *ri_out = p.ri;
  const bool ok = p.ri.status == STATUS_OK;
  uint64_t d[4] = {0, 0, 0, 0};
  if (ok) hash_without(m, n, p.sig_field_start, p.sig_field_end, d);
  for (int j = 0; j < 4; j++)
    for (int b = 0; b < 8; b++) digest32[8 * j + b] = (uint8_t)(d[j] >> (8 * b));
  const bool sig_ok = ok && p.ri.sig_len == 65, from_ok = ok && p.ri.from_len == 20;
  for (int i = 0; i < 65; i++) sig65[i] = sig_ok ? m[p.sig_pos + i] : 0;
  for (int i = 0; i < 20; i++) from20[i] = from_ok ? p.ri.from[i] : 0;
  const bool seal_ok = ok && p.ri.payload_kind == KIND_COMMIT && p.ri.seal_len == 65;
  for (int i = 0; i < 65; i++) seal65[i] = seal_ok ? m[p.seal_pos + i] : 0;
  *pre_flag = (uint8_t)((ok ? 0 : 1) | (sig_ok ? 0 : 2) | (from_ok ? 0 : 2));
}

// ---- certificates: messages inside messages ------------------------------------------------------------------
// One row of ibft_verify_certificates_wire per IbftMessage of the tree, breadth first: the call's n messages are
// level 0, the messages nested directly inside level-k rows are level k + 1 (children of row i before children of
// row j > i, each row's children in wire order).  mirrors ibft_cert_node_t (include/ibftgpu.h), 56 bytes.
struct node_info {
  uint32_t off, len;                  // the message's bytes in the call's buffer
  uint32_t parent, ordinal;           // containing row (0xFFFFFFFF: none) and position among its children
  uint32_t first_child, n_children;   // rows [first_child, first_child + n_children)
  uint32_t raw_off, raw_len;          // Proposal.rawProposal carried by this message, in the call's buffer
  uint64_t proposal_round;            // Proposal.round
  uint32_t cut0, cut1;                // the signature field (tag, length, bytes) relative to off: PayloadNoSig = bytes minus [cut0, cut1)
  uint8_t level, role, flags, pad[5];
};
static_assert(sizeof(node_info) == 56, "ABI");
constexpr uint32_t NO_PARENT = 0xFFFFFFFFu;
// Longest message the device hashes (one lane absorbs 136 bytes in ≈10 µs: 1 MiB ≈ 75 ms, next to the thousands of
// signatures such a message carries).  Longer canonical messages come back TREE_TOO_BIG with cut0 / cut1 for the host.
constexpr uint32_t TREE_DIGEST_MAX_BYTES = 1u << 20;

// 136 message bytes at src → state (aligned dword loads + funnel shifts; reads at most 3 bytes past src + 136)
HD void absorb_full_block(uint64_t s[25], const uint8_t *src) {
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u), sh = 8u * mis;
  const uint32_t *p = reinterpret_cast<const uint32_t *>(src - mis);
  uint32_t raw[35];
#pragma unroll
  for (int j = 0; j < 34; j++) raw[j] = p[j];
  raw[34] = mis ? p[34] : 0u;
#pragma unroll
  for (int i = 0; i < 17; i++) {
    const uint32_t lo = (uint32_t)(((uint64_t)raw[2 * i + 1] << 32 | raw[2 * i]) >> sh);
    const uint32_t hi = (uint32_t)(((uint64_t)raw[2 * i + 2] << 32 | raw[2 * i + 1]) >> sh);
    s[i] ^= (uint64_t)lo | ((uint64_t)hi << 32);
  }
}
// Keccak-256 of a[0..na) ‖ b[0..nb) ‖ tail[0..nt) (nt ≤ 8): whole blocks that lie inside one piece go through
// absorb_full_block, the blocks on a seam and the last one byte by byte.
HD void hash_pieces(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb, const uint8_t *tail, uint32_t nt, uint64_t out4[4]) {
  const uint64_t total = (uint64_t)na + nb + nt;
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  uint64_t done = 0;
  for (;;) {
    const uint64_t left = total - done;
    const bool last = left < 136;
    if (!last && done + 136 <= na) {
      absorb_full_block(s, a + done);
    } else if (!last && done >= na && done + 136 <= (uint64_t)na + nb) {
      absorb_full_block(s, b + (done - na));
    } else {
      for (int i = 0; i < 17; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) {
          const uint32_t o = 8 * i + k;
          uint64_t byte = 0;
          if (o < left) {
            const uint64_t v = done + o;
            byte = v < na ? a[v] : (v < (uint64_t)na + nb ? b[v - na] : tail[v - na - nb]);
          }
          if (last && o == left) byte ^= 0x01u;
          if (last && o == 135) byte ^= 0x80u;
          w |= byte << (8 * k);
        }
        s[i] ^= w;
      }
    }
    keccak::f1600(s);
    if (last) break;
    done += 136;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out4[i] = s[i];
}
// PayloadNoSig digest of a canonical message of any length
HD void hash_without_long(const uint8_t *m, uint32_t n, uint32_t cut0, uint32_t cut1, uint64_t out4[4]) {
  hash_pieces(m, cut0, m + cut1, n - cut1, m, 0, out4);
}
// keccak256(rawProposal ‖ BE64(round)): the proposal hash convention (include/ibftgpu.h)
HD void hash_proposal(const uint8_t *raw, uint32_t raw_len, uint64_t round, uint64_t out4[4]) {
  uint8_t be[8];
  for (int i = 0; i < 8; i++) be[i] = (uint8_t)(round >> (8 * (7 - i)));
  hash_pieces(raw, raw_len, raw, 0, be, 8, out4);
}

// A message's PayloadNoSig digest is DEFERRED when it cannot be final after the walk of its own fields, or would hold up
// the 63 other lanes of its wavefront: it carries a certificate (canonical only if every message below it is), or it is
// longer than a PREPARE / COMMIT (a PREPREPARE with its proposal: several Keccak blocks).  Deferred digests are computed
// after the whole tree is known, a wavefront per message (kernels.hip.h: cert_digest_wave_kernel).
constexpr uint32_t TREE_DEFER_BYTES = 256;
HD bool tree_deferred(const node_info &nd) { return (nd.flags & TREE_HAS_CERT) || nd.len > TREE_DEFER_BYTES; }

// One row of the tree: the DEEP walk, the columns of the sender check, the node's own facts.  The digest and the pre-flag
// of a row that is not deferred are final here; a deferred row's pre-flag says "verdict 0" until tree_digest_row.
HD void process_tree_row(const uint8_t *m, uint32_t n, row_info *ri_out, node_info *node, uint32_t cert_span[2],
                         uint8_t *digest32, uint8_t *sig65, uint8_t *from20, uint8_t *pre_flag) {
  const parsed p = parse_message_t<true>(m, n);
  *ri_out = p.ri;
  const bool ok = p.ri.status == STATUS_OK;
  const bool has_cert = ok && (p.t.flags & TREE_HAS_CERT);
  node->raw_off = ok ? node->off + p.t.raw_off : 0;
  node->raw_len = ok ? p.t.raw_len : 0;
  node->proposal_round = ok ? p.t.proposal_round : 0;
  node->cut0 = ok ? p.sig_field_start : 0;
  node->cut1 = ok ? p.sig_field_end : 0;
  node->flags = ok ? p.t.flags : 0;
  node->first_child = node->n_children = 0;
  // where the nested messages lie (relative to the message): the walk of the next step counts and lists them
  cert_span[0] = has_cert ? p.t.cert_off : 0;
  cert_span[1] = has_cert ? p.t.cert_len : 0;
  const bool now = ok && !tree_deferred(*node);
  uint64_t d[4] = {0, 0, 0, 0};
  if (now) hash_without(m, n, p.sig_field_start, p.sig_field_end, d);
  for (int j = 0; j < 4; j++)
    for (int b = 0; b < 8; b++) digest32[8 * j + b] = (uint8_t)(d[j] >> (8 * b));
  const bool sig_ok = ok && p.ri.sig_len == 65, from_ok = ok && p.ri.from_len == 20;
  for (int i = 0; i < 65; i++) sig65[i] = sig_ok ? m[p.sig_pos + i] : 0;
  for (int i = 0; i < 20; i++) from20[i] = from_ok ? p.ri.from[i] : 0;
  *pre_flag = (uint8_t)((now ? 0 : 1) | (sig_ok ? 0 : 2) | (from_ok ? 0 : 2));
}
// After the levels below have been judged (a non-canonical child has already turned this row's status to NEEDS_HOST): the
// pre-flag and class of a deferred row — and, with hash_here, its digest (the CPU harness; on the device a wavefront per
// deferred row has written it, cert_digest_wave_kernel) — and the hash of the Proposal a row carries.
HD void tree_digest_row(const uint8_t *wire, const row_info *ri, node_info *node, uint8_t *digest32, uint8_t *prop_digest32,
                        uint8_t *pre_flag, bool hash_here) {
  const bool ok = ri->status == STATUS_OK;
  if (tree_deferred(*node)) {
    bool judged = false;
    if (ok && node->len <= TREE_DIGEST_MAX_BYTES) {
      if (hash_here) {
        uint64_t d[4];
        hash_without_long(wire + node->off, node->len, node->cut0, node->cut1, d);
        for (int j = 0; j < 4; j++)
          for (int b = 0; b < 8; b++) digest32[8 * j + b] = (uint8_t)(d[j] >> (8 * b));
      }
      judged = true;
    } else if (ok) {
      node->flags |= TREE_TOO_BIG;
    }
    *pre_flag = (uint8_t)((judged ? 0 : 1) | (ri->sig_len == 65 && ri->from_len == 20 ? 0 : 2));
  }
  // the Proposal's hash: here for a row that is not deferred (a short message, a short proposal) or on the CPU; the device's
  // wavefront of a deferred row has written it next to the row's own digest
  const bool prop_ok = ok && (node->flags & TREE_HAS_PROPOSAL) && node->raw_len <= TREE_DIGEST_MAX_BYTES;
  if (ok && (node->flags & TREE_HAS_PROPOSAL) && !prop_ok) node->flags |= TREE_PROPOSAL_TOO_BIG;
  if (prop_ok && !hash_here && tree_deferred(*node)) return;
  uint64_t h[4] = {0, 0, 0, 0};
  if (prop_ok) hash_proposal(wire + node->raw_off, node->raw_len, node->proposal_round, h);
  for (int j = 0; j < 4; j++)
    for (int b = 0; b < 8; b++) prop_digest32[8 * j + b] = (uint8_t)(h[j] >> (8 * b));
}

// hash bit: the proposal hash this message carries = keccak(the lastPreparedProposal of the ROUND_CHANGE message whose
// PreparedCertificate contains it) — proposalMatchesCertificate, /root/reference/core/ibft.go:516-551; self bit: a
// PREPREPARE's proposalHash = keccak(its own Proposal) — IsValidProposalHash of validateProposalCommon, :640-651.
// cls: the row's routing byte (IBFT_CERT_CLASS_*).
HD void tree_compare_row(const node_info *nodes, const row_info *rows, const uint8_t *prop_digest32, uint32_t row, bool &hash_bit,
                         bool &self_bit, uint8_t &cls) {
  const node_info &nd = nodes[row];
  const row_info &ri = rows[row];
  const bool ok = ri.status == STATUS_OK;
  hash_bit = self_bit = false;
  if (ok && ri.hash_len == 32) {
    const uint32_t p = nd.parent;
    if (p != NO_PARENT && (nd.role == ROLE_PC_PROPOSAL || nd.role == ROLE_PC_PREPARE)) {
      const uint8_t pf = nodes[p].flags;
      if (rows[p].status == STATUS_OK && (pf & TREE_HAS_PROPOSAL) && !(pf & TREE_PROPOSAL_TOO_BIG)) {
        const uint8_t *d = prop_digest32 + 32ull * p;
        bool eq = true;
        for (int i = 0; i < 32; i++) eq = eq && d[i] == ri.proposal_hash[i];
        hash_bit = eq;
      }
    }
    if (ri.payload_kind == KIND_PREPREPARE && (nd.flags & TREE_HAS_PROPOSAL) && !(nd.flags & TREE_PROPOSAL_TOO_BIG)) {
      const uint8_t *d = prop_digest32 + 32ull * row;
      bool eq = true;
      for (int i = 0; i < 32; i++) eq = eq && d[i] == ri.proposal_hash[i];
      self_bit = eq;
    }
  }
  cls = (uint8_t)((ok ? 0 : 1) | ((nd.flags & TREE_TOO_BIG) ? 2 : 0) | ((nd.flags & TREE_PROPOSAL_TOO_BIG) ? 4 : 0));
}

}  // namespace wire
