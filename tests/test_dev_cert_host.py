"""§8f rank 2 from bytes, on the CPU: the certificate-tree code of csrc/wire_dev.h (compiled for the host and driven in
the order the cert_* kernels apply it, csrc/host_arith_harness.hip: dev_cert_tree) against the independent
decode-and-re-marshal oracle (oracle/wire_cert.py): tree shape, parsed fields, canonical / NEEDS_HOST classes, digests,
hash bits — and the sender verdicts the recover kernels must then give, from the digests and signatures it extracts."""
import ctypes as C

import numpy as np
import pytest

import cert_cases as CC
from go_ibft_amd import build as B
from go_ibft_amd.verifier import CERT_NODE, WIRE_ROW
from oracle import binding as OB
from oracle import wire
from oracle import wire_cert as WC
from oracle import workload as W


@pytest.fixture(scope="module")
def dev():
    L = C.CDLL(B.build_host_harness())
    L.dev_cert_tree.restype = C.c_int64
    return L


@pytest.fixture(scope="module")
def rnd():
    return W.make_round(8, 811, height=5, round_=1)


def run_dev(dev, msgs, cap=4096):
    buf, off = CC.pack(msgs)
    wb = np.frombuffer(buf + b"\0" * 16, dtype=np.uint8)
    nodes, rows = np.zeros(cap, dtype=CERT_NODE), np.zeros(cap, dtype=WIRE_ROW)
    digest, sig, frm = np.zeros((cap, 32), np.uint8), np.zeros((cap, 65), np.uint8), np.zeros((cap, 20), np.uint8)
    pre, prop, cls = np.zeros(cap, np.uint8), np.zeros((cap, 32), np.uint8), np.zeros(cap, np.uint8)
    hb, sb = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = dev.dev_cert_tree(p(wb), p(off), len(msgs), cap, p(nodes), p(rows), p(digest), p(sig), p(frm), p(pre), p(prop), p(cls), p(hb), p(sb))
    return int(n), nodes, rows, digest, sig, frm, pre, prop, cls, hb, sb


def check(dev, r, label, msgs, cap=4096):
    exp = WC.expected_tree(msgs, r.addrs, rows_cap=cap)
    n, nodes, rows, digest, sig, frm, pre, prop, cls, hb, sb = run_dev(dev, msgs, cap)
    if exp is None:
        assert n == -1, label
        return None
    assert n == exp.n_rows, (label, n, exp.n_rows)
    # the sender verdict the verdict launch gives: pre-flag, else recover over the extracted digest / signature / From
    members = {bytes(a) for a in r.addrs}
    sender = np.zeros(n, dtype=bool)
    for k in range(n):
        if pre[k]:
            continue
        a = OB.recover_address(digest[k].tobytes(), sig[k].tobytes())
        sender[k] = a is not None and a == frm[k].tobytes() and a in members
    CC.compare(label, exp, n, nodes, rows, cls, sender, hb, sb)
    for k in range(n):
        if exp.digest[k] is not None:
            assert digest[k].tobytes() == exp.digest[k], (label, k, "digest")
        if exp.prop_digest[k] is not None:
            assert prop[k].tobytes() == exp.prop_digest[k], (label, k, "proposal digest")
        if exp.status[k] == WC.OK and len(exp.rows[k].signature) == 65 and len(exp.rows[k].sender) == 20:
            assert sig[k].tobytes() == exp.rows[k].signature and frm[k].tobytes() == exp.rows[k].sender
    return exp


def test_handmade_cases(dev, rnd):
    seen = {}
    for label, msgs in CC.handmade(rnd):
        exp = check(dev, rnd, label, msgs)
        seen[label] = exp
    honest = seen["honest round-change set"]
    assert honest.n_rows == 8 + 8 * 6 and all(honest.sender_ok) and all(honest.hash_bit[8:]) and not any(honest.cls)
    deep = seen["five levels"]
    assert max(nd["level"] for nd in deep.nodes) == 4 and all(deep.sender_ok)
    bad = seen["non-canonical prepare three levels down"]
    assert bad.cls[0] == WC.CLASS_NEEDS_HOST and bad.cls[1] == WC.CLASS_NEEDS_HOST and bad.cls[2] == 0  # the honest sibling is judged
    forged = seen["forged prepare inside a certificate"]
    assert [bool(x) for x in forged.sender_ok] == [True, True, True, False, True]
    assert seen["proposal message after the prepares"].n_rows == 1
    assert seen["certificate for another proposal than lastPreparedProposal"].hash_bit == [False] * 5


def test_fuzzed_trees(dev, rnd):
    stats = {"ok": 0, "host": 0}
    for i, msgs in enumerate(CC.fuzz_batches(rnd, 400, 90125)):
        exp = check(dev, rnd, f"fuzz {i}", msgs)
        stats["host" if exp.cls[0] & 1 else "ok"] += 1
    assert stats["ok"] > 20 and stats["host"] > 100, stats


def test_rows_cap(dev, rnd):
    msgs = [m.encode() for m in CC.honest_round_change_set(rnd)]
    assert check(dev, rnd, "cap exactly", msgs, cap=56) is not None
    assert check(dev, rnd, "cap one short", msgs, cap=55) is None
    assert check(dev, rnd, "cap below the call's messages", msgs, cap=7) is None


def test_longer_than_the_device_hashes(dev):
    r = W.make_round(4, 812, height=5, round_=1, raw_len=(1 << 20) + 5)
    rc = CC.round_change(r, 0, 5, 2, wire.Proposal(r.raw, 1), CC.pc_bytes(r, 5, 1, 1, [2, 3]))
    exp = check(dev, r, "1 MiB proposal", [rc.encode(), CC.prepare(r, 1, 5, 2).encode()])
    # the ROUND_CHANGE and the PREPREPARE inside its certificate both carry the long proposal and are that long themselves;
    # the PREPAREs next to them are judged
    both = WC.CLASS_DIGEST_BY_HOST | WC.CLASS_PROPOSAL_BY_HOST
    assert exp.cls[0] == both and exp.cls[1] == 0 and exp.cls[2] == both and exp.cls[3] == 0
    assert not exp.sender_ok[0] and not exp.sender_ok[2] and exp.sender_ok[1] and exp.sender_ok[3] and exp.sender_ok[4]


def test_many_children_and_window_sized_certificates(dev):
    """a PreparedCertificate much longer than the walk's 16 KiB window (the GPU test repeats this through the kernel)"""
    r = W.make_round(200, 813, height=5, round_=1, raw_len=64)
    pc = CC.pc_bytes(r, 5, 1, 1, [i for i in range(200) if i != 1])
    rc = CC.round_change(r, 0, 5, 2, wire.Proposal(r.raw, 1), pc)
    exp = check(dev, r, "200 validators", [rc.encode()])
    assert exp.n_rows == 201 and all(exp.sender_ok) and all(exp.hash_bit[1:])


def test_golden_trees(dev):
    """tests/golden/cert_trees.json (oracle/wire_cert.py cross-checked message by message against the google.protobuf runtime,
    tests/golden/make_cert_golden.py): one digest per batch over tree shape, classes, verdict bits and parsed fields"""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cert_trees.json")))
    count = 700
    r, batches = CC.golden_batches(count)
    members = {bytes(a) for a in r.addrs}
    for bi, msgs in enumerate(batches):
        n, nodes, rows, digest, sig, frm, pre, prop, cls, hb, sb = run_dev(dev, msgs)
        sender = np.zeros(n, dtype=bool)
        for k in range(n):
            if not pre[k]:
                a = OB.recover_address(digest[k].tobytes(), sig[k].tobytes())
                sender[k] = a is not None and a == frm[k].tobytes() and a in members
        assert CC.digest_actual(n, nodes, rows, cls, sender, hb, sb) == gold["digests"][bi], bi
