"""§8f rank 3 on the CPU: csrc/wire_dev.h (the device wire walker, compiled for the host) against the
independent decode-and-re-marshal oracle (oracle/wire_parse.py) and, where the protobuf runtime is
importable, against google.protobuf itself."""
import ctypes as C

import numpy as np
import pytest

import wire_cases as WCASE
from go_ibft_amd import build as B
from go_ibft_amd.verifier import WIRE_ROW
from oracle import wire_parse as WP
from oracle import workload as W


@pytest.fixture(scope="module")
def dev():
    return C.CDLL(B.build_host_harness())


def dev_row(dev, m: bytes):
    out = np.zeros(263, dtype=np.uint8)
    buf = np.frombuffer(m + b"\0", dtype=np.uint8)  # never a null pointer
    dev.dev_wire_row(buf.ctypes.data_as(C.c_void_p), len(m), out.ctypes.data_as(C.c_void_p))
    ri = out[:80].view(WIRE_ROW)[0]
    return ri, out[80:112].tobytes(), out[112:177].tobytes(), out[177:197].tobytes(), out[197:262].tobytes(), int(out[262])


def check(dev, m: bytes, label=""):
    exp = WP.expected(m)
    ri, digest, sig, frm, seal, pre = dev_row(dev, m)
    assert int(ri["status"]) == exp.status, (label, m.hex())
    assert (pre != 0) == exp.pre_flag, (label, m.hex())
    if exp.status != WP.OK:
        return exp
    assert digest == exp.digest, label
    assert (int(ri["height"]), int(ri["round"]), int(ri["type"]), int(ri["payload_kind"]), int(ri["has_view"])) == \
        (exp.height, exp.round, exp.type, exp.payload_kind, exp.has_view), label
    assert int(ri["hash_len"]) == len(exp.proposal_hash) and \
        ri["proposal_hash"].tobytes()[:len(exp.proposal_hash)] == exp.proposal_hash, label
    assert int(ri["from_len"]) == min(len(exp.sender), 255) and int(ri["sig_len"]) == min(len(exp.signature), 255)
    assert int(ri["seal_len"]) == min(len(exp.committed_seal), 255)
    assert ri["from"].tobytes()[:min(20, len(exp.sender))] == exp.sender[:20]
    if len(exp.signature) == 65 and len(exp.sender) == 20:
        assert sig == exp.signature and frm == exp.sender
    if exp.payload_kind == 7 and len(exp.committed_seal) == 65:
        assert seal == exp.committed_seal
    return exp


def test_canonical_round_all_kinds(dev):
    r = W.make_round(24, 501, height=11, round_=2, byzantine=True)
    rows = WCASE.canonical_round(r, ("commit", "prepare", "preprepare", "roundchange"))
    stats = [check(dev, m).status for m in rows]
    assert stats == [WP.OK, WP.OK, WP.NEEDS_HOST, WP.NEEDS_HOST] * 6


def test_handmade_encodings(dev):
    r = W.make_round(4, 502)
    seen = {}
    for label, m in WCASE.handmade(r):
        seen[label] = check(dev, m, label).status
    assert seen["canonical commit"] == WP.OK and seen["no view"] == WP.OK and seen["empty payload"] == WP.OK
    assert seen["64-byte signature"] == WP.OK and seen["21-byte from"] == WP.OK and seen["type 200 with a commit payload"] == WP.OK
    for label in ("padded varint height", "explicit zero type", "fields out of order", "two oneof members",
                  "unknown field 9", "truncated", "33-byte hash", "type 300", "empty from emitted",
                  "unknown field in the body", "seal before hash", "view round before height", "padded length"):
        assert seen[label] == WP.NEEDS_HOST, label


def test_fuzzed_rows_agree_with_the_oracle(dev):
    r = W.make_round(16, 503, byzantine=True)
    base = WCASE.canonical_round(r)
    ok = 0
    for m in WCASE.fuzz(base, 3000, 9):
        ok += check(dev, m).status == WP.OK
    assert 100 < ok < 2900  # both outcomes are exercised


def test_against_the_protobuf_runtime(dev):
    """status OK ⇒ google.protobuf decodes the row and re-serialises it, signature cleared, to exactly
    the bytes the device hashed (PayloadNoSig, helper.go:12-27)."""
    pytest.importorskip("google.protobuf")
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "make_wire_fixtures", os.path.join(os.path.dirname(__file__), "golden", "make_wire_fixtures.py"))
    mwf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mwf)
    from oracle import binding as OB
    msgs = mwf.build_messages()
    r = W.make_round(12, 504)
    rows = WCASE.canonical_round(r, ("commit", "prepare")) + [m for _, m in WCASE.handmade(r)] + \
        WCASE.fuzz(WCASE.canonical_round(r), 400, 10)
    checked = 0
    for m in rows:
        ri, digest, *_ = dev_row(dev, m)
        if int(ri["status"]) != WP.OK:
            continue
        g = msgs["IbftMessage"]()
        g.ParseFromString(m)
        g.signature = b""
        assert digest == OB.keccak256(g.SerializeToString(deterministic=True)), m.hex()
        checked += 1
    assert checked > 50


def test_golden_wire_rows(dev):
    """tests/golden/wire_rows.json (make_wire_rows.py: oracle expectation cross-checked with the protobuf
    runtime at generation time) — the walker alone against committed vectors."""
    import json
    import os
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wire_rows.json")))
    assert len(rows) > 150
    for o in rows:
        m = bytes.fromhex(o["wire"])
        ri, digest, sig, frm, seal, pre = dev_row(dev, m)
        assert int(ri["status"]) == o["status"] and (pre != 0) == bool(o["pre_flag"]), o["label"]
        if o["status"] != WP.OK:
            continue
        assert digest.hex() == o["digest"], o["label"]
        assert (int(ri["height"]), int(ri["round"]), int(ri["type"]), int(ri["payload_kind"]), int(ri["has_view"])) == \
            (o["height"], o["round"], o["type"], o["payload_kind"], o["has_view"]), o["label"]
        ph = bytes.fromhex(o["proposal_hash"])
        assert int(ri["hash_len"]) == len(ph) and ri["proposal_hash"].tobytes()[:len(ph)] == ph
        if len(o["signature"]) == 130 and len(o["from"]) == 40:
            assert sig.hex() == o["signature"] and frm.hex() == o["from"]
        if o["payload_kind"] == 7 and len(o["committed_seal"]) == 130:
            assert seal.hex() == o["committed_seal"]
