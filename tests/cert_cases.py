"""Certificate-carrying messages for the §8f rank 2 "from bytes" tests: ROUND_CHANGE messages with
PreparedCertificates, PREPREPARE messages with RoundChangeCertificates (messages.proto:46-101), their
Byzantine and non-canonical variants, and the comparison of an implementation's answer with
oracle/wire_cert.py.  Test infrastructure only (uses the oracle's encoder and signer)."""
import random

import numpy as np

from oracle import binding as B
from oracle import wire
from oracle import wire_cert as WC
from wire_cases import signed


def prepare(r, i, height, round_, h=None, sk=None):
    m = wire.IbftMessage(view=wire.View(height, round_), sender=r.addrs[i].tobytes(), type=wire.PREPARE,
                         payload=wire.prepare_body(r.proposal_hash if h is None else h))
    return sm(m, r.sks[i] if sk is None else sk)


def sm(m: wire.IbftMessage, sk: bytes) -> wire.IbftMessage:
    """sign in place, return the message object (wire.prepared_certificate wants objects)"""
    m.signature = B.sign(sk, B.keccak256(m.payload_no_sig()))
    return m


def preprepare(r, i, height, round_, raw=None, proposal_round=None, h=None, rcc=None, sk=None):
    raw = r.raw if raw is None else raw
    pr = round_ if proposal_round is None else proposal_round
    h = B.proposal_hash(raw, pr) if h is None else h
    m = wire.IbftMessage(view=wire.View(height, round_), sender=r.addrs[i].tobytes(), type=wire.PREPREPARE,
                         payload=wire.preprepare_body(wire.Proposal(raw, pr), h, rcc))
    return sm(m, r.sks[i] if sk is None else sk)


def round_change(r, i, height, round_, last=None, pc=None, sk=None):
    m = wire.IbftMessage(view=wire.View(height, round_), sender=r.addrs[i].tobytes(), type=wire.ROUND_CHANGE,
                         payload=wire.round_change_body(last, pc))
    return sm(m, r.sks[i] if sk is None else sk)


def pc_bytes(r, height, round_, proposer, preparers, raw=None, **kw):
    """PreparedCertificate of (height, round_): the proposer's PREPREPARE + PREPAREs of `preparers`"""
    raw = r.raw if raw is None else raw
    h = B.proposal_hash(raw, round_)
    pm = preprepare(r, proposer, height, round_, raw=raw, **kw)
    return wire.prepared_certificate(pm, [prepare(r, i, height, round_, h=h) for i in preparers])


def honest_round_change_set(r, height=5, new_round=2, prepared_round=1, senders=None, raw=None):
    """every sender moves to new_round carrying the proposal prepared in prepared_round and its certificate"""
    n = r.n
    raw = r.raw if raw is None else raw
    senders = list(range(n)) if senders is None else senders
    proposer = prepared_round % n
    out = []
    for i in senders:
        preparers = [j for j in range(n) if j != proposer][: max(1, (2 * n) // 3)]
        pc = pc_bytes(r, height, prepared_round, proposer, preparers, raw=raw)
        out.append(round_change(r, i, height, new_round, wire.Proposal(raw, prepared_round), pc))
    return out


def preprepare_with_rcc(r, height=5, new_round=2, rcs=None, **kw):
    rcs = honest_round_change_set(r, height, new_round) if rcs is None else rcs
    return preprepare(r, new_round % r.n, height, new_round, rcc=wire.round_change_certificate(rcs), **kw)


def _raw_msg(view, sender, sig, typ, kind, body):
    """hand-assembled IbftMessage bytes (kind = oneof field number, body bytes or None)"""
    out = b""
    if view is not None:
        out += wire._len_field(1, view.encode(), emit_empty=True)
    out += wire._len_field(2, sender) + wire._len_field(3, sig) + wire._varint_field(4, typ)
    if body is not None:
        out += wire._len_field(kind, body, emit_empty=True)
    return out


def handmade(r):
    """(label, [message bytes …]) — each entry is one call's batch"""
    H, RND = 5, 2
    n = r.n
    addr = [r.addrs[i].tobytes() for i in range(n)]
    good_rcs = honest_round_change_set(r, H, RND)
    good = [m.encode() for m in good_rcs]
    outsider = B.keccak256(b"outsider")  # a key that is no validator's
    cases = [
        ("honest round-change set", good),
        ("preprepare with rcc", [preprepare_with_rcc(r, H, RND, good_rcs).encode()]),
        ("mixed batch: rc, prepare, commit-less, preprepare(rcc)", [good[0], prepare(r, 1, H, RND).encode(),
                                                                    preprepare_with_rcc(r, H, RND, good_rcs[:3]).encode(), good[1], b""]),
        ("rc without proposal and certificate", [round_change(r, 0, H, RND).encode()]),
        ("rc with proposal, no certificate", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1), None).encode()]),
        ("rc with empty certificate", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1), b"").encode()]),
        ("rc with empty proposal wrapper", [round_change(r, 0, H, RND, wire.Proposal(b"", 0), pc_bytes(r, H, 1, 1, [2, 3])).encode()]),
        ("rc with certificate, no proposal", [round_change(r, 0, H, RND, None, pc_bytes(r, H, 1, 1, [2, 3])).encode()]),
        ("preprepare with empty rcc", [preprepare(r, 2, H, RND, rcc=b"").encode()]),
        ("preprepare round 0, no rcc", [preprepare(r, 0, H, 0).encode()]),
    ]
    # Byzantine content, canonical bytes
    pm = preprepare(r, 1, H, 1)
    pr = [prepare(r, i, H, 1, h=B.proposal_hash(r.raw, 1)) for i in (2, 3, 4)]
    forged = prepare(r, 3, H, 1, h=B.proposal_hash(r.raw, 1), sk=outsider)                 # From = validator 3, signed by someone else
    stranger = wire.IbftMessage(view=wire.View(H, 1), sender=B.address(B.pubkey(outsider)), type=wire.PREPARE,
                                payload=wire.prepare_body(B.proposal_hash(r.raw, 1)))
    sm(stranger, outsider)                                                                   # valid signature, not a validator
    wrong_hash = prepare(r, 4, H, 1, h=B.keccak256(b"other"))
    short_hash = prepare(r, 4, H, 1, h=B.proposal_hash(r.raw, 1)[:31])
    commit_in_pc = wire.IbftMessage(view=wire.View(H, 1), sender=addr[2], type=wire.COMMIT,
                                    payload=wire.commit_body(B.proposal_hash(r.raw, 1), r.seal65[2].tobytes()))
    sm(commit_in_pc, r.sks[2])
    bad_pm_hash = preprepare(r, 1, H, 1, h=B.keccak256(b"not the proposal"))
    cases += [
        ("forged prepare inside a certificate", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                              wire.prepared_certificate(pm, [pr[0], forged, pr[2]])).encode()]),
        ("non-validator prepare", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                wire.prepared_certificate(pm, [pr[0], stranger])).encode()]),
        ("prepare for another proposal", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                       wire.prepared_certificate(pm, [wrong_hash, pr[1], short_hash])).encode()]),
        ("commit message inside a certificate", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                              wire.prepared_certificate(pm, [commit_in_pc, pr[1]])).encode()]),
        ("proposal message with a wrong own hash", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                                 wire.prepared_certificate(bad_pm_hash, pr)).encode()]),
        ("certificate for another proposal than lastPreparedProposal", [round_change(r, 0, H, RND, wire.Proposal(r.raw[:-1] + b"\x00", 1),
                                                                                     wire.prepared_certificate(pm, pr)).encode()]),
        ("certificate of another round than the proposal says", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 0),
                                                                              wire.prepared_certificate(pm, pr)).encode()]),
        ("rc envelope forged, certificate honest", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                                 wire.prepared_certificate(pm, pr), sk=outsider).encode()]),
        ("no proposal message, prepares only", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1),
                                                             wire.prepared_certificate(None, pr)).encode()]),
        ("proposal message only", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1), wire.prepared_certificate(pm, [])).encode()]),
        ("empty nested messages", [round_change(r, 0, H, RND, wire.Proposal(r.raw, 1), b"\x0a\x00\x12\x00\x12\x00").encode()]),
    ]
    # deeper nesting: the proposal message inside a certificate carries its own RCC
    inner_rcs = honest_round_change_set(r, H, 1, prepared_round=0, senders=[0, 1, 2])
    deep_pm = preprepare(r, 1, H, 1, rcc=wire.round_change_certificate(inner_rcs))
    deep_rc = round_change(r, 0, H, RND, wire.Proposal(r.raw, 1), wire.prepared_certificate(deep_pm, pr))
    cases += [
        ("four levels", [deep_rc.encode()]),
        ("five levels", [preprepare(r, 2, H, RND, rcc=wire.round_change_certificate([deep_rc, good_rcs[1]])).encode()]),
    ]
    # non-canonical encodings somewhere in the tree
    pcb = wire.prepared_certificate(pm, pr)
    padded = bytearray(pr[1].encode())
    assert padded[0] == 0x0A and padded[2] == 0x08
    padded_prepare = bytes(padded[:1]) + bytes([padded[1] + 1]) + b"\x08" + bytes([padded[3] | 0x80, 0x00]) + bytes(padded[4:])
    pc_with_padded = wire._len_field(1, pm.encode(), True) + wire._len_field(2, pr[0].encode(), True) + \
        wire._len_field(2, padded_prepare, True) + wire._len_field(2, pr[2].encode(), True)
    sig = B.sign(r.sks[0], b"\x22" * 32)

    def rc_raw(body):
        return _raw_msg(wire.View(H, RND), addr[0], sig, wire.ROUND_CHANGE, 8, body)
    prop = wire._len_field(1, wire.Proposal(r.raw, 1).encode(), True)
    cases += [
        ("non-canonical prepare inside a certificate", [rc_raw(prop + wire._len_field(2, pc_with_padded, True)), good[2]]),
        ("non-canonical prepare three levels down", [_raw_msg(wire.View(H, RND), addr[2], sig, wire.PREPREPARE, 5,
                                                              wire._len_field(1, wire.Proposal(r.raw, RND).encode(), True) +
                                                              wire._len_field(2, B.proposal_hash(r.raw, RND)) +
                                                              wire._len_field(3, wire._len_field(1, rc_raw(prop + wire._len_field(2, pc_with_padded, True)), True) +
                                                                              wire._len_field(1, good[3], True), True))]),
        ("proposal message after the prepares", [rc_raw(prop + wire._len_field(2, wire._len_field(2, pr[0].encode(), True) +
                                                                               wire._len_field(1, pm.encode(), True), True))]),
        ("two proposal messages", [rc_raw(prop + wire._len_field(2, wire._len_field(1, pm.encode(), True) * 2, True))]),
        ("unknown field in the certificate", [rc_raw(prop + wire._len_field(2, pcb + b"\x18\x01", True))]),
        ("varint where a message should be", [rc_raw(prop + wire._len_field(2, b"\x10\x05", True))]),
        ("nested message runs past the certificate", [rc_raw(prop + wire._len_field(2, b"\x12\x7f\x00", True))]),
        ("padded nested length", [rc_raw(prop + wire._len_field(2, b"\x12\x80\x00", True))]),
        ("certificate before the proposal", [rc_raw(wire._len_field(2, pcb, True) + prop)]),
        ("two certificates", [rc_raw(prop + wire._len_field(2, pcb, True) * 2)]),
        ("unknown field in the round-change body", [rc_raw(prop + wire._len_field(2, pcb, True) + b"\x18\x01")]),
        ("unknown field in the proposal", [rc_raw(wire._len_field(1, wire.Proposal(r.raw, 1).encode() + b"\x18\x01", True))]),
        ("explicit zero proposal round", [rc_raw(wire._len_field(1, wire._len_field(1, r.raw) + b"\x10\x00", True))]),
        ("empty raw proposal emitted", [rc_raw(wire._len_field(1, b"\x0a\x00\x10\x01", True))]),
        ("proposal round before raw", [rc_raw(wire._len_field(1, b"\x10\x01" + wire._len_field(1, r.raw), True))]),
        ("preprepare: hash before proposal", [_raw_msg(wire.View(H, RND), addr[2], sig, wire.PREPREPARE, 5,
                                                       wire._len_field(2, b"\x07" * 32) + wire._len_field(1, wire.Proposal(r.raw, RND).encode(), True))]),
        ("preprepare: 33-byte hash", [_raw_msg(wire.View(H, RND), addr[2], sig, wire.PREPREPARE, 5,
                                               wire._len_field(1, wire.Proposal(r.raw, RND).encode(), True) + wire._len_field(2, b"\x07" * 33))]),
        ("preprepare: empty hash emitted", [_raw_msg(wire.View(H, RND), addr[2], sig, wire.PREPREPARE, 5, b"\x12\x00")]),
        ("preprepare: pc-style field 2 in the rcc", [_raw_msg(wire.View(H, RND), addr[2], sig, wire.PREPREPARE, 5,
                                                              wire._len_field(3, wire._len_field(2, good[0], True), True))]),
        ("type and payload disagree (type PREPARE, round-change payload)", [_raw_msg(wire.View(H, RND), addr[0], sig, wire.PREPARE, 8,
                                                                                    prop + wire._len_field(2, pcb, True))]),
        ("round-change payload, then commit payload", [rc_raw(prop) + wire._len_field(7, wire.commit_body(b"\x01" * 32, b"\x02" * 65), True)]),
    ]
    return cases


def fuzz_batches(r, count, seed):
    """byte-level damage somewhere inside canonical certificate-carrying messages"""
    rng = random.Random(seed)
    H, RND = 5, 2
    base = [m.encode() for m in honest_round_change_set(r, H, RND, senders=[0, 1])]
    base.append(preprepare_with_rcc(r, H, RND, honest_round_change_set(r, H, RND, senders=[0, 1, 2])).encode())
    out = []
    for _ in range(count):
        b = bytearray(rng.choice(base))
        for _ in range(rng.choice([1, 1, 2, 3])):
            op = rng.randrange(4)
            pos = rng.randrange(len(b)) if b else 0
            if op == 0 and b:
                b[pos] ^= 1 << rng.randrange(8)
            elif op == 1:
                b.insert(pos, rng.randrange(256))
            elif op == 2 and b:
                del b[pos]
            elif b:
                del b[pos:]
        out.append([bytes(b), rng.choice(base)])
    return out


# ---- comparison ------------------------------------------------------------------------------------------------
def compare(label, exp: WC.Tree, n_rows, nodes, rows, cls, sender, hashb, selfb):
    """an implementation's answer (numpy structured arrays / bool arrays) against oracle/wire_cert.Tree"""
    assert n_rows == exp.n_rows, (label, n_rows, exp.n_rows)
    for k in range(n_rows):
        nd, e = nodes[k], exp.nodes[k]
        where = (label, k)
        for f in ("off", "len", "parent", "ordinal", "level", "role", "n_children"):
            assert int(nd[f]) == e[f], (where, f, int(nd[f]), e[f])
        if e["n_children"]:
            assert int(nd["first_child"]) == e["first_child"], where
        assert int(cls[k]) == exp.cls[k], (where, "class", int(cls[k]), exp.cls[k])
        assert (int(rows[k]["status"]) == 0) == (exp.status[k] == WC.OK), where
        assert bool(sender[k]) == exp.sender_ok[k], (where, "sender")
        assert bool(hashb[k]) == exp.hash_bit[k], (where, "hash")
        assert bool(selfb[k]) == exp.self_bit[k], (where, "self")
        if exp.status[k] != WC.OK:
            continue
        o, ri = exp.rows[k], rows[k]
        assert (int(ri["height"]), int(ri["round"]), int(ri["type"]), int(ri["payload_kind"]), int(ri["has_view"])) == \
            (o.height, o.round, o.type, o.kind, o.has_view), where
        assert int(ri["hash_len"]) == len(o.proposal_hash) and ri["proposal_hash"].tobytes()[:len(o.proposal_hash)] == o.proposal_hash, where
        assert int(ri["from_len"]) == min(len(o.sender), 255) and int(ri["sig_len"]) == min(len(o.signature), 255), where
        assert ri["from"].tobytes()[:min(20, len(o.sender))] == o.sender[:20], where
        assert int(nd["flags"]) == e["flags"], (where, "flags", int(nd["flags"]), e["flags"])
        assert (int(nd["cut0"]), int(nd["cut1"])) == (e["cut0"], e["cut1"]), where
        if e["flags"] & WC.HAS_PROPOSAL:
            assert (int(nd["raw_off"]), int(nd["raw_len"]), int(nd["proposal_round"])) == (e["raw_off"], e["raw_len"], e["proposal_round"]), where


def pack(msgs):
    off = np.zeros(len(msgs) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) for x in msgs])
    return b"".join(msgs), off


# ---- golden digests (tests/golden/cert_trees.json, made by tests/golden/make_cert_golden.py) --------------------
def _row_bytes(struct_fields, ok_fields):
    import struct
    out = struct.pack("<7I5B", *struct_fields)
    if ok_fields is not None:
        out += struct.pack("<2Q9B4IQ", *ok_fields)
    return out


def digest_expected(exp: WC.Tree) -> str:
    """what an answer must contain, hashed: tree shape, classes and bits of every row, parsed fields of the judged rows"""
    import hashlib
    h = hashlib.sha256()
    for k in range(exp.n_rows):
        e = exp.nodes[k]
        ok = exp.status[k] == WC.OK
        okf = None
        if ok:
            o = exp.rows[k]
            okf = (o.height, o.round, o.type, o.kind, o.has_view, len(o.proposal_hash), min(len(o.sender), 255), min(len(o.signature), 255),
                   min(len(o.committed_seal), 255) if o.kind == 7 else 0, e["flags"], 0, e["cut0"], e["cut1"],
                   e["raw_off"] if e["flags"] & WC.HAS_PROPOSAL else 0, e["raw_len"] if e["flags"] & WC.HAS_PROPOSAL else 0,
                   e["proposal_round"] if e["flags"] & WC.HAS_PROPOSAL else 0)
        h.update(_row_bytes((e["off"], e["len"], e["parent"], e["ordinal"], e["level"], e["role"], e["n_children"],
                             exp.cls[k], int(ok), int(exp.sender_ok[k]), int(exp.hash_bit[k]), int(exp.self_bit[k])), okf))
    return h.hexdigest()[:16]


def digest_actual(n_rows, nodes, rows, cls, sender, hashb, selfb) -> str:
    import hashlib
    h = hashlib.sha256()
    for k in range(n_rows):
        nd, ri = nodes[k], rows[k]
        ok = int(ri["status"]) == 0
        okf = None
        if ok:
            fl = int(nd["flags"])
            hp = fl & WC.HAS_PROPOSAL
            okf = (int(ri["height"]), int(ri["round"]), int(ri["type"]), int(ri["payload_kind"]), int(ri["has_view"]), int(ri["hash_len"]),
                   int(ri["from_len"]), int(ri["sig_len"]), int(ri["seal_len"]) if int(ri["payload_kind"]) == 7 else 0, fl, 0,
                   int(nd["cut0"]), int(nd["cut1"]), int(nd["raw_off"]) if hp else 0, int(nd["raw_len"]) if hp else 0,
                   int(nd["proposal_round"]) if hp else 0)
        h.update(_row_bytes((int(nd["off"]), int(nd["len"]), int(nd["parent"]), int(nd["ordinal"]), int(nd["level"]), int(nd["role"]),
                             int(nd["n_children"]), int(cls[k]), int(ok), int(bool(sender[k])), int(bool(hashb[k])), int(bool(selfb[k]))), okf))
    return h.hexdigest()[:16]


GOLDEN_SEED, GOLDEN_VALIDATORS, GOLDEN_ROUND_SEED = 31337, 8, 811


def golden_batches(count):
    """the deterministic batches the golden digests refer to: handmade cases first, then byte-level fuzz"""
    from oracle import workload as W
    r = W.make_round(GOLDEN_VALIDATORS, GOLDEN_ROUND_SEED, height=5, round_=1)
    hm = [m for _, m in handmade(r)]
    return r, (hm + fuzz_batches(r, max(0, count - len(hm)), GOLDEN_SEED))[:count]
