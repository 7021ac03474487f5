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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keccak_dev.h"

namespace wire {

constexpr uint8_t STATUS_OK = 0, STATUS_NEEDS_HOST = 1;
constexpr uint8_t KIND_NONE = 0, KIND_PREPARE = 6, KIND_COMMIT = 7;  // oneof field numbers

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

struct parsed {
  row_info ri;
  uint32_t sig_field_start, sig_field_end;  // the whole field 3 (tag, length, bytes); equal when absent
  uint32_t sig_pos, seal_pos;               // payload bytes of Signature / CommittedSeal
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

// IbftMessage { View view = 1; bytes from = 2; bytes signature = 3; MessageType type = 4;
//               oneof payload { PrePrepare = 5; Prepare = 6; Commit = 7; RoundChange = 8 } }
HD parsed parse_message(const uint8_t *m, uint32_t n) {
  parsed p;
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
      default:  // 5 PREPREPARE, 8 ROUND_CHANGE: nested proposals and certificates — host
        return p;
    }
    pos += len;
  }
  if (!sig_seen) p.sig_field_start = p.sig_field_end = n;  // nothing to cut out
  p.ri.status = STATUS_OK;
  return p;
}

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

}  // namespace wire
