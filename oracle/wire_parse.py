"""oracle/wire_parse.py — what the device wire walker (csrc/wire_dev.h) must answer, restated
independently (TEST INFRASTRUCTURE ONLY).

Approach: decode the bytes the way the protobuf runtime does (generic tag/value walk, last scalar
wins, oneof = last member seen, unknown fields kept), re-marshal with the canonical encoder of
oracle/wire.py, and call a row OK exactly when

  * it decodes, carries no unknown fields and re-marshals to the very same bytes
    (then PayloadNoSig — /root/reference/messages/proto/helper.go:12-27 — is those bytes without
    the signature field), and
  * its payload is absent, a PrepareMessage or a CommitMessage with flat, canonical bodies
    (messages.proto:59-71), proposal hash ≤ 32 bytes, type ≤ 255.

Everything else is NEEDS_HOST.  The decode here is a different algorithm from the device's
single-pass canonical walk, which is the point.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import binding as B
from . import wire

OK, NEEDS_HOST = 0, 1


class Malformed(Exception):
    pass


def _varint(buf: bytes, pos: int) -> tuple[int, int]:
    v = 0
    for i in range(10):
        if pos >= len(buf):
            raise Malformed("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << (7 * i)
        if not b & 0x80:
            return v & 0xFFFFFFFFFFFFFFFF, pos
    raise Malformed("varint too long")


def fields(buf: bytes):
    """generic walk: yields (field number, wire type, value) — value is int or bytes"""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if f == 0:
            raise Malformed("field 0")
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > len(buf):
                raise Malformed("truncated bytes")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 1:
            if pos + 8 > len(buf):
                raise Malformed("truncated fixed64")
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 5:
            if pos + 4 > len(buf):
                raise Malformed("truncated fixed32")
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise Malformed("group / reserved wire type")
        yield f, wt, v


@dataclass
class Row:
    status: int = NEEDS_HOST
    height: int = 0
    round: int = 0
    type: int = 0
    payload_kind: int = 0
    has_view: int = 0
    sender: bytes = b""
    signature: bytes = b""
    proposal_hash: bytes = b""
    committed_seal: bytes = b""
    digest: bytes = bytes(32)  # keccak256(PayloadNoSig)

    @property
    def pre_flag(self) -> bool:
        """the sender check must answer 0 without looking at the curve"""
        return self.status != OK or len(self.sender) != 20 or len(self.signature) != 65


def _decode_view(buf: bytes) -> tuple[wire.View, bool]:
    v, unknown = wire.View(), False
    for f, wt, val in fields(buf):
        if f == 1 and wt == 0:
            v.height = val
        elif f == 2 and wt == 0:
            v.round = val
        else:
            unknown = True
    return v, unknown


def _decode_body(buf: bytes, commit: bool) -> tuple[bytes, bytes, bool]:
    h, seal, unknown = b"", b"", False
    for f, wt, val in fields(buf):
        if f == 1 and wt == 2:
            h = val
        elif commit and f == 2 and wt == 2:
            seal = val
        else:
            unknown = True
    return h, seal, unknown


def expected(buf: bytes) -> Row:
    row = Row()
    try:
        view, sender, sig, typ, kind, body, unknown = None, b"", b"", 0, 0, None, False
        for f, wt, val in fields(buf):
            if f == 1 and wt == 2:
                nv, u = _decode_view(val)
                unknown |= u
                if view is None:
                    view = nv
                else:  # embedded messages merge
                    view = wire.View(nv.height or view.height, nv.round or view.round)
            elif f == 2 and wt == 2:
                sender = val
            elif f == 3 and wt == 2:
                sig = val
            elif f == 4 and wt == 0:
                typ = val
            elif f in (5, 6, 7, 8) and wt == 2:
                kind, body = f, val  # oneof: the last member seen
            else:
                unknown = True
        if unknown or kind in (5, 8) or typ > 255:
            return row
        h = seal = b""
        if kind in (6, 7):
            h, seal, u = _decode_body(body, kind == 7)
            if u:
                return row
        # canonical re-marshal (oracle/wire.py); the payload field number is the oneof member, not `type`
        out = b""
        if view is not None:
            out += wire._len_field(1, view.encode(), emit_empty=True)
        out += wire._len_field(2, sender)
        cut0 = len(out)
        out += wire._len_field(3, sig)
        cut1 = len(out)
        out += wire._varint_field(4, typ)
        if kind:
            out += wire._len_field(kind, wire.commit_body(h, seal) if kind == 7 else wire.prepare_body(h),
                                   emit_empty=True)
        if out != buf or len(h) > 32:
            return row
        row.status = OK
        row.height, row.round = (view.height, view.round) if view is not None else (0, 0)
        row.has_view = 1 if view is not None else 0
        row.type, row.payload_kind = typ, kind
        row.sender, row.signature, row.proposal_hash, row.committed_seal = sender, sig, h, seal
        row.digest = B.keccak256(out[:cut0] + out[cut1:])
        return row
    except Malformed:
        return row
