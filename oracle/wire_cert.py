"""oracle/wire_cert.py — what ibft_verify_certificates_wire (include/ibftgpu.h) must answer for a batch of
IbftMessages that carry certificates, restated independently (TEST INFRASTRUCTURE ONLY).

The reference verifies the messages nested in PREPREPARE / ROUND_CHANGE messages one by one after
proto.Unmarshal: /root/reference/core/ibft.go:516-551 (proposalMatchesCertificate), :683-788 (validPC,
validateProposal), :470-512 (handleRoundChangeMessage), each through IsValidValidator — recover over
keccak(PayloadNoSig), /root/reference/messages/proto/helper.go:12-27 — and IsValidProposalHash.

Approach (a different algorithm from the device's single-pass canonical walk, csrc/wire_dev.h): decode every
message the way the protobuf runtime does (generic tag / value walk), re-marshal what was decoded with the
canonical encoder of oracle/wire.py — nested messages embedded as the bytes they came in — and call the
message's OWN encoding canonical iff that reproduces its bytes; a message is judged iff its own encoding and
the encoding of every message below it are canonical.  Rows are numbered breadth first (the call's messages,
then their nested messages in order, …), as the header specifies.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import binding as B
from . import wire
from .wire_parse import Malformed, _varint, NEEDS_HOST, OK

NO_PARENT = 0xFFFFFFFF
ROLE_ROOT, ROLE_PC_PROPOSAL, ROLE_PC_PREPARE, ROLE_RCC_MESSAGE = 0, 1, 2, 3
HAS_PROPOSAL, HAS_CERT, TOO_BIG, PROPOSAL_TOO_BIG = 1, 2, 4, 8
CLASS_NEEDS_HOST, CLASS_DIGEST_BY_HOST, CLASS_PROPOSAL_BY_HOST = 1, 2, 4
DIGEST_MAX_BYTES = 1 << 20


def fields_pos(buf: bytes, base: int = 0):
    """generic walk: (field, wire type, value, position of the value's first byte in the enclosing buffer)"""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if f == 0:
            raise Malformed("field 0")
        if wt == 0:
            v, pos2 = _varint(buf, pos)
            yield f, wt, v, base + pos
            pos = pos2
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > len(buf):
                raise Malformed("truncated bytes")
            yield f, wt, buf[pos:pos + ln], base + pos
            pos += ln
        elif wt in (1, 5):
            w = 8 if wt == 1 else 4
            if pos + w > len(buf):
                raise Malformed("truncated fixed")
            yield f, wt, buf[pos:pos + w], base + pos
            pos += w
        else:
            raise Malformed("group / reserved wire type")


@dataclass
class Own:
    """one message, its own fields only"""
    height: int = 0
    round: int = 0
    has_view: int = 0
    type: int = 0
    kind: int = 0
    sender: bytes = b""
    signature: bytes = b""
    proposal_hash: bytes = b""
    committed_seal: bytes = b""
    flags: int = 0
    raw_off: int = 0          # relative to the message
    raw_len: int = 0
    proposal_round: int = 0
    cut0: int = 0
    cut1: int = 0
    children: list = field(default_factory=list)   # (relative offset, length, role)


def _decode_proposal(buf: bytes, base: int):
    raw, raw_pos, rnd, unknown = b"", 0, 0, False
    for f, wt, v, p in fields_pos(buf, base):
        if f == 1 and wt == 2:
            raw, raw_pos = v, p
        elif f == 2 and wt == 0:
            rnd = v
        else:
            unknown = True
    return raw, raw_pos, rnd, unknown


def own(buf: bytes) -> Own | None:
    """decode + canonical re-marshal of ONE message (nested messages kept as opaque bytes); None = not canonical"""
    try:
        view = None
        sender = sig = b""
        typ = kind = 0
        body, body_pos, unknown = None, 0, False
        for f, wt, v, p in fields_pos(buf):
            if f == 1 and wt == 2:
                nv = wire.View()
                for f2, wt2, v2, _ in fields_pos(v):
                    if f2 == 1 and wt2 == 0:
                        nv.height = v2
                    elif f2 == 2 and wt2 == 0:
                        nv.round = v2
                    else:
                        unknown = True
                view = nv if view is None else wire.View(nv.height or view.height, nv.round or view.round)
            elif f == 2 and wt == 2:
                sender = v
            elif f == 3 and wt == 2:
                sig = v
            elif f == 4 and wt == 0:
                typ = v
            elif f in (5, 6, 7, 8) and wt == 2:
                kind, body, body_pos = f, v, p
            else:
                unknown = True
        if unknown or typ > 255:
            return None
        o = Own(type=typ, kind=kind, sender=sender, signature=sig)
        payload = None
        if kind in (6, 7):
            h = seal = b""
            for f, wt, v, p in fields_pos(body, body_pos):
                if f == 1 and wt == 2:
                    h = v
                elif kind == 7 and f == 2 and wt == 2:
                    seal = v
                else:
                    return None
            o.proposal_hash, o.committed_seal = h, seal
            payload = wire.commit_body(h, seal) if kind == 7 else wire.prepare_body(h)
        elif kind in (5, 8):
            proposal = cert = None
            h = b""
            kids = []
            n_proposal_messages = 0
            for f, wt, v, p in fields_pos(body, body_pos):
                if f == 1 and wt == 2:
                    raw, raw_pos, rnd, u = _decode_proposal(v, p)
                    if u:
                        return None
                    if proposal is not None:
                        return None  # a second copy would be merged by the runtime: not what Marshal emits
                    proposal = wire.Proposal(raw, rnd)
                    o.flags |= HAS_PROPOSAL
                    o.raw_off, o.raw_len, o.proposal_round = (raw_pos if raw else 0), len(raw), rnd
                elif kind == 5 and f == 2 and wt == 2:
                    h = v
                elif ((kind == 5 and f == 3) or (kind == 8 and f == 2)) and wt == 2:
                    if cert is not None:
                        return None
                    cert = v
                    o.flags |= HAS_CERT
                    for f2, wt2, v2, p2 in fields_pos(v, p):
                        if wt2 != 2:
                            return None
                        if kind == 5 and f2 == 1:
                            kids.append((p2, len(v2), ROLE_RCC_MESSAGE, v2))
                        elif kind == 8 and f2 == 1:
                            n_proposal_messages += 1
                            kids.append((p2, len(v2), ROLE_PC_PROPOSAL, v2))
                        elif kind == 8 and f2 == 2:
                            kids.append((p2, len(v2), ROLE_PC_PREPARE, v2))
                        else:
                            return None
                else:
                    return None
            if n_proposal_messages > 1:
                return None
            o.proposal_hash = h
            if kind == 5:
                cert_bytes = None if cert is None else b"".join(wire._len_field(1, k[3], emit_empty=True) for k in kids)
                payload = wire.preprepare_body(proposal, h, cert_bytes)
            else:
                cert_bytes = None
                if cert is not None:
                    cert_bytes = b"".join(wire._len_field(1 if k[2] == ROLE_PC_PROPOSAL else 2, k[3], emit_empty=True)
                                          for k in sorted(kids, key=lambda k: k[2]))  # stable: proposalMessage first
                payload = wire.round_change_body(proposal, cert_bytes)
            o.children = [(k[0], k[1], k[2]) for k in kids]
        out = b""
        if view is not None:
            out += wire._len_field(1, view.encode(), emit_empty=True)
            o.height, o.round, o.has_view = view.height, view.round, 1
        out += wire._len_field(2, sender)
        o.cut0 = len(out)
        out += wire._len_field(3, sig)
        o.cut1 = len(out)
        out += wire._varint_field(4, typ)
        if kind:
            out += wire._len_field(kind, payload, emit_empty=True)
        if out != buf or len(o.proposal_hash) > 32:
            return None
        if not sig:
            o.cut0 = o.cut1 = len(buf)
        return o
    except Malformed:
        return None


@dataclass
class Tree:
    wire: bytes
    n_rows: int
    nodes: list     # dicts: off len parent ordinal first_child n_children raw_off raw_len proposal_round cut0 cut1 level role flags
    rows: list      # Own | None (None: own encoding not canonical)
    status: list    # OK / NEEDS_HOST after propagation
    cls: list
    digest: list    # keccak256(PayloadNoSig) or None
    prop_digest: list
    sender_ok: list
    hash_bit: list
    self_bit: list


def expected_tree(msgs: list, valset_addrs, rows_cap: int = 1 << 30, digest_max: int = DIGEST_MAX_BYTES) -> Tree | None:
    """valset_addrs: iterable of 20-byte validator addresses.  None when the tree has more than rows_cap rows."""
    members = {bytes(a) for a in valset_addrs}
    buf = b"".join(msgs)
    nodes, rows = [], []
    pos = 0
    for i, m in enumerate(msgs):
        nodes.append(dict(off=pos, len=len(m), parent=NO_PARENT, ordinal=i, level=0, role=ROLE_ROOT))
        pos += len(m)
    if len(nodes) > rows_cap:
        return None
    lo, hi = 0, len(nodes)
    levels = []
    while True:
        levels.append((lo, hi))
        base = hi
        nxt = []
        for r in range(lo, hi):
            nd = nodes[r]
            o = own(buf[nd["off"]:nd["off"] + nd["len"]])
            rows.append(o)
            kids = o.children if o is not None else []
            nd.update(first_child=base, n_children=len(kids),
                      raw_off=(nd["off"] + o.raw_off) if o is not None else 0,
                      raw_len=o.raw_len if o is not None else 0,
                      proposal_round=o.proposal_round if o is not None else 0,
                      cut0=o.cut0 if o is not None else 0, cut1=o.cut1 if o is not None else 0,
                      flags=o.flags if o is not None else 0)
            for k, (rel, ln, role) in enumerate(kids):
                nxt.append(dict(off=nd["off"] + rel, len=ln, parent=r, ordinal=k, level=nd["level"] + 1, role=role))
            base += len(kids)
        if not nxt:
            break
        if hi + len(nxt) > rows_cap:
            return None
        nodes.extend(nxt)
        lo, hi = hi, hi + len(nxt)
    n = len(nodes)
    status = [OK if o is not None else NEEDS_HOST for o in rows]
    for lo, hi in reversed(levels[1:]):
        for r in range(lo, hi):
            if status[r] != OK:
                status[nodes[r]["parent"]] = NEEDS_HOST
    digest, prop, cls, sender_ok = [None] * n, [None] * n, [0] * n, [False] * n
    for r in range(n):
        nd, o = nodes[r], rows[r]
        if status[r] != OK:
            cls[r] |= CLASS_NEEDS_HOST
            continue
        m = buf[nd["off"]:nd["off"] + nd["len"]]
        if nd["len"] > digest_max:
            nd["flags"] |= TOO_BIG
            cls[r] |= CLASS_DIGEST_BY_HOST
        else:
            digest[r] = B.keccak256(m[:o.cut0] + m[o.cut1:])
        if o.flags & HAS_PROPOSAL:
            if o.raw_len > digest_max:
                nd["flags"] |= PROPOSAL_TOO_BIG
                cls[r] |= CLASS_PROPOSAL_BY_HOST
            else:
                raw = buf[nd["raw_off"]:nd["raw_off"] + nd["raw_len"]]
                prop[r] = B.keccak256(raw + o.proposal_round.to_bytes(8, "big"))
        if digest[r] is not None and len(o.signature) == 65 and len(o.sender) == 20:
            a = B.recover_address(digest[r], o.signature)
            sender_ok[r] = a is not None and a == o.sender and o.sender in members
    hash_bit, self_bit = [False] * n, [False] * n
    for r in range(n):
        nd, o = nodes[r], rows[r]
        if status[r] != OK or len(o.proposal_hash) != 32:
            continue
        p = nd["parent"]
        if p != NO_PARENT and nd["role"] in (ROLE_PC_PROPOSAL, ROLE_PC_PREPARE) and status[p] == OK and prop[p] is not None:
            hash_bit[r] = o.proposal_hash == prop[p]
        if o.kind == 5 and prop[r] is not None:
            self_bit[r] = o.proposal_hash == prop[r]
    return Tree(buf, n, nodes, rows, status, cls, digest, prop, sender_ok, hash_bit, self_bit)
