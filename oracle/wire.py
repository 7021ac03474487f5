"""oracle/wire.py — restatement of the protobuf wire bytes of go-ibft messages
(TEST INFRASTRUCTURE ONLY).

``payload_no_sig`` follows /root/reference/messages/proto/helper.go:12-27
(PayloadNoSig = proto.Marshal of the message with Signature cleared) using the field
numbers of /root/reference/messages/proto/messages.proto:24-110: proto3, fields in
field-number order, minimal varints, zero scalars and empty bytes omitted, a present
sub-message emitted even when empty.  Pinned against the google.protobuf runtime by
tests/golden/make_wire_fixtures.py → tests/golden/wire_vectors.json.
"""
from __future__ import annotations

from dataclasses import dataclass, field

PREPREPARE, PREPARE, COMMIT, ROUND_CHANGE = 0, 1, 2, 3


def varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _len_field(num: int, data: bytes | None, emit_empty: bool = False) -> bytes:
    if data is None or (not data and not emit_empty):
        return b""
    return varint((num << 3) | 2) + varint(len(data)) + data


def _varint_field(num: int, v: int) -> bytes:
    return b"" if v == 0 else varint(num << 3) + varint(v)


@dataclass
class View:
    height: int = 0
    round: int = 0

    def encode(self) -> bytes:
        return _varint_field(1, self.height) + _varint_field(2, self.round)


@dataclass
class Proposal:
    raw_proposal: bytes = b""
    round: int = 0

    def encode(self) -> bytes:
        return _len_field(1, self.raw_proposal) + _varint_field(2, self.round)


@dataclass
class IbftMessage:
    """Mirror of proto.IbftMessage (messages.proto:24-44). ``payload`` is the already
    encoded oneof body (or None when the oneof is unset)."""
    view: View | None = None
    sender: bytes = b""          # `from`
    signature: bytes = b""
    type: int = PREPREPARE
    payload: bytes | None = None  # encoded Pre/Prepare/Commit/RoundChange message

    def encode(self, with_signature: bool = True) -> bytes:
        out = b""
        if self.view is not None:
            out += _len_field(1, self.view.encode(), emit_empty=True)
        out += _len_field(2, self.sender)
        if with_signature:
            out += _len_field(3, self.signature)
        out += _varint_field(4, self.type)
        if self.payload is not None:
            out += _len_field(5 + self.type, self.payload, emit_empty=True)
        return out

    def payload_no_sig(self) -> bytes:
        return self.encode(with_signature=False)


def prepare_body(proposal_hash: bytes) -> bytes:
    return _len_field(1, proposal_hash)


def commit_body(proposal_hash: bytes, committed_seal: bytes) -> bytes:
    return _len_field(1, proposal_hash) + _len_field(2, committed_seal)


def preprepare_body(proposal: Proposal | None, proposal_hash: bytes, certificate: bytes | None) -> bytes:
    out = b""
    if proposal is not None:
        out += _len_field(1, proposal.encode(), emit_empty=True)
    out += _len_field(2, proposal_hash)
    if certificate is not None:
        out += _len_field(3, certificate, emit_empty=True)
    return out


def round_change_body(last_prepared: Proposal | None, latest_pc: bytes | None) -> bytes:
    out = b""
    if last_prepared is not None:
        out += _len_field(1, last_prepared.encode(), emit_empty=True)
    if latest_pc is not None:
        out += _len_field(2, latest_pc, emit_empty=True)
    return out


def prepared_certificate(proposal_message: IbftMessage | None, prepare_messages: list[IbftMessage]) -> bytes:
    out = b""
    if proposal_message is not None:
        out += _len_field(1, proposal_message.encode(), emit_empty=True)
    for m in prepare_messages:
        out += _len_field(2, m.encode(), emit_empty=True)
    return out


def round_change_certificate(messages: list[IbftMessage]) -> bytes:
    return b"".join(_len_field(1, m.encode(), emit_empty=True) for m in messages)
