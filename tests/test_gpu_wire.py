"""§8f rank 3 on the GPU: ibft_verify_senders_wire (wire bytes → parse + canonical check + Keccak +
sender recover on the device) and ibft_wire_stage_seals, against the oracle."""
import numpy as np
import pytest

import wire_cases as WCASE
from oracle import wire_parse as WP
from oracle import workload as W

pytestmark = pytest.mark.gpu


def _expected(oracle, vs, rows_bytes):
    exps = [WP.expected(m) for m in rows_bytes]
    verdict = np.zeros(len(exps), dtype=bool)
    for i, e in enumerate(exps):
        if e.pre_flag:
            continue
        got = oracle.recover_address(e.digest, e.signature)
        verdict[i] = got is not None and got == e.sender and vs.index(e.sender) >= 0
    return exps, verdict


@pytest.mark.parametrize("cache", [False, True])
def test_wire_rows_vs_oracle(oracle, cache):
    import go_ibft_amd.verifier as V
    r = W.make_round(300, 601, height=9, round_=1, byzantine=True, weighted=True)
    rows_bytes = WCASE.canonical_round(r, ("commit", "prepare", "commit", "prepare", "preprepare", "roundchange")) + \
        [m for _, m in WCASE.handmade(r)] + WCASE.fuzz(WCASE.canonical_round(r)[:40], 500, 21)
    wire, off = WCASE.pack(rows_bytes)
    vs = oracle.ValSet(r.addrs, r.power)
    exps, want = _expected(oracle, vs, rows_bytes)
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if cache else 0, max_rows=4096)
    try:
        bv.set_validators(1, r.addrs, r.power)
        for _ in range(2 if cache else 1):  # second pass runs on the warm kernels
            got, rows, t = bv.is_valid_validator_wire(wire, off)
            assert (got == want).all(), np.nonzero(got != want)[0][:10]
            assert [int(s) for s in rows["status"]] == [e.status for e in exps]
            for ri, e in zip(rows, exps):
                if e.status != WP.OK:
                    continue
                assert (int(ri["height"]), int(ri["round"]), int(ri["type"]), int(ri["payload_kind"])) == \
                    (e.height, e.round, e.type, e.payload_kind)
                assert ri["proposal_hash"].tobytes()[:int(ri["hash_len"])] == e.proposal_hash
                assert ri["from"].tobytes()[:min(20, len(e.sender))] == e.sender[:20]
            senders = np.zeros((len(exps), 20), dtype=np.uint8)
            for i, e in enumerate(exps):
                if len(e.sender) == 20 and e.status == WP.OK:
                    senders[i] = np.frombuffer(e.sender, dtype=np.uint8)
            ot = oracle.tally(vs, senders, want.astype(np.uint8))
            assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == \
                (ot.power, ot.valid_rows, ot.distinct_senders, ot.has_quorum)
        assert want.sum() > 100 and (~want).sum() > 100
        # a2 on the seals the walker found: no second upload
        bv.wire_stage_seals()
        bv.seals_launch(1)
        seals, _ = bv.seals_fetch()
        hash32 = np.zeros((len(exps), 32), dtype=np.uint8)
        seal65 = np.zeros((len(exps), 65), dtype=np.uint8)
        pre = np.ones(len(exps), dtype=np.uint8)
        for i, e in enumerate(exps):
            if e.status == WP.OK and e.payload_kind == 7 and e.type == 2 and len(e.proposal_hash) == 32 and \
                    len(e.committed_seal) == 65 and len(e.sender) == 20:
                hash32[i] = np.frombuffer(e.proposal_hash, dtype=np.uint8)
                seal65[i] = np.frombuffer(e.committed_seal, dtype=np.uint8)
                pre[i] = 0
        exp_seals = oracle.verify_seals(vs, hash32, seal65, senders, pre).astype(bool)
        assert (seals == exp_seals).all()
        assert exp_seals.sum() > 50
    finally:
        bv.close()


def test_wire_path_equals_host_flattened_path(gpu_verifier, oracle):
    """for canonical rows the verdicts are those of ibft_verify_senders on PayloadNoSig + From + Signature"""
    r = W.make_round(200, 602, byzantine=True)
    rows_bytes = WCASE.canonical_round(r)
    wire, off = WCASE.pack(rows_bytes)
    gpu_verifier.set_validators(1, r.addrs, r.power)
    got, rows, _ = gpu_verifier.is_valid_validator_wire(wire, off)
    assert (rows["status"] == 0).all()
    pns, sigs, frm = [], [], []
    for m in rows_bytes:
        e = WP.expected(m)
        cut = m.index(b"\x1a\x41" + e.signature)
        pns.append(m[:cut] + m[cut + 67:])
        sigs.append(np.frombuffer(e.signature, dtype=np.uint8))
        frm.append(np.frombuffer(e.sender, dtype=np.uint8))
    payload, poff = WCASE.pack(pns)
    ref, _ = gpu_verifier.is_valid_validator(payload, poff, np.array(sigs), np.array(frm))
    assert (got == ref).all() and got.sum() > 100


def test_stage_seals_needs_a_resident_wire_batch(gpu_verifier):
    import go_ibft_amd.verifier as V
    r = W.make_round(8, 603)
    gpu_verifier.set_validators(1, r.addrs, r.power)
    gpu_verifier.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)  # restages the columns
    with pytest.raises(RuntimeError):
        gpu_verifier.wire_stage_seals()
    wire, off = WCASE.pack([])
    got, rows, t = gpu_verifier.is_valid_validator_wire(wire, off)   # empty batch
    assert len(got) == 0 and len(rows) == 0 and t.has_quorum == 0


def test_host_mirror_wire_route_equals_stock_route(gpu_verifier, oracle):
    """GpuBackend::VerifySendersWire (device walk + stock route for flagged rows) gives the verdicts of
    the stock route (decode, PayloadNoSig, flatten, ibft_verify_senders) on every row, whatever its
    encoding; undecodable rows are 0 on both."""
    import go_ibft_amd.hostlib as H
    r = W.make_round(120, 604, byzantine=True)
    rows_bytes = WCASE.canonical_round(r, ("commit", "prepare", "preprepare", "roundchange")) + \
        [m for _, m in WCASE.handmade(r)] + WCASE.fuzz(WCASE.canonical_round(r)[:30], 300, 22)
    wire, off = WCASE.pack(rows_bytes)
    gpu_verifier.set_validators(1, r.addrs, r.power)
    fast, _, host_rows = H.verify_senders_wire(gpu_verifier, wire, off)
    stock, _, stock_rows = H.verify_senders_wire(gpu_verifier, wire, off, stock=True)
    assert (fast == stock).all(), np.nonzero(fast != stock)[0][:10]
    assert 0 < host_rows < stock_rows <= len(rows_bytes)
    assert fast.sum() > 50


def test_golden_wire_rows_on_gpu(oracle):
    """the committed wire vectors through ibft_verify_senders_wire: status, extracted fields and — via the
    verdicts — the digest of every accepted row (a wrong PayloadNoSig hash cannot recover the sender)."""
    import json
    import os
    import go_ibft_amd.verifier as V
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wire_rows.json")))
    msgs = [bytes.fromhex(o["wire"]) for o in rows]
    r = W.make_round(12, 9001, height=5, round_=1, byzantine=True)  # the round the generator used
    vs = oracle.ValSet(r.addrs, r.power)
    want = np.zeros(len(rows), dtype=bool)
    for i, o in enumerate(rows):
        if o["status"] == 0 and not o["pre_flag"]:
            frm = bytes.fromhex(o["from"])
            got = oracle.recover_address(bytes.fromhex(o["digest"]), bytes.fromhex(o["signature"]))
            want[i] = got == frm and vs.index(frm) >= 0
    wire, off = WCASE.pack(msgs)
    bv = V.BatchVerifier(max_rows=1024)
    try:
        bv.set_validators(1, r.addrs, r.power)
        got, info, _ = bv.is_valid_validator_wire(wire, off)
    finally:
        bv.close()
    assert [int(s) for s in info["status"]] == [o["status"] for o in rows]
    assert (got == want).all() and want.sum() >= 10


@pytest.mark.parametrize("cache", [False, True])
def test_messages_judged_completely_from_wire_bytes(oracle, cache):
    """ibft_verify_messages_wire: sender bits ≡ ibft_verify_senders_wire; valid bits ≡ the handlePrepare / handleCommit
    closure evaluated by the oracle on the decoded message — for canonical PREPARE / COMMIT messages of the asked view
    only; other views, other kinds, odd encodings and fuzzed bytes have valid bit 0."""
    import go_ibft_amd.verifier as V
    height, rnd = 9, 1
    r = W.make_round(300, 611, height=height, round_=rnd, byzantine=True, weighted=True)
    other = W.make_round(40, 612, height=height, round_=rnd + 1)               # same height, another round
    rows_bytes = WCASE.canonical_round(r, ("commit", "prepare", "commit", "prepare", "preprepare", "roundchange")) + \
        WCASE.canonical_round(other) + [m for _, m in WCASE.handmade(r)] + WCASE.fuzz(WCASE.canonical_round(r)[:40], 400, 23)
    wire, off = WCASE.pack(rows_bytes)
    vs = oracle.ValSet(r.addrs, r.power)
    exps, want_sender = _expected(oracle, vs, rows_bytes)
    H = oracle.proposal_hash(r.raw, rnd)
    want_valid = np.zeros(len(exps), dtype=bool)
    for i, e in enumerate(exps):
        if e.status != WP.OK or (e.height, e.round) != (height, rnd) or len(e.sender) != 20 or e.proposal_hash != H:
            continue
        if e.type == 1 and e.payload_kind == 6:
            want_valid[i] = True
        elif e.type == 2 and e.payload_kind == 7 and len(e.committed_seal) == 65:
            got = oracle.recover_address(e.proposal_hash, e.committed_seal)
            want_valid[i] = got is not None and got == e.sender and vs.index(e.sender) >= 0
    assert want_valid.sum() > 100 and (want_sender & ~want_valid).sum() > 20
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if cache else 0, max_rows=4096)
    try:
        bv.set_validators(1, r.addrs, r.power)
        for _ in range(3 if cache else 1):
            s, v, rows, t = bv.verify_messages_wire(wire, off, height, rnd, raw=r.raw)
            assert (s == want_sender).all(), np.nonzero(s != want_sender)[0][:10]
            assert (v == want_valid).all(), np.nonzero(v != want_valid)[0][:10]
            assert [int(x) for x in rows["status"]] == [e.status for e in exps]
        # the one routing byte per row instead of the 80-byte parse results
        s, v, cls, _ = bv.verify_messages_wire(wire, off, height, rnd, raw=r.raw, want_rows=False)
        assert (s == want_sender).all() and (v == want_valid).all()
        for c, e in zip(cls, exps):
            assert bool(c & V.WIRE_CLASS_NEEDS_HOST) == (e.status != WP.OK)
            in_view = e.status == WP.OK and (e.height, e.round) == (height, rnd) and e.type in (1, 2)
            assert bool(c & V.WIRE_CLASS_CLOSURE) == in_view
            if e.status == WP.OK:
                assert c >> 4 == (e.type & 15)
        # the digest form of the proposal, and a pinned buffer for the bytes
        s, v, _, _ = bv.verify_messages_wire(V.pinned_copy(wire), V.pinned_copy(off), height, rnd, digest32=H)
        assert (s == want_sender).all() and (v == want_valid).all() and bv.gather_batches() >= 1
        # asked about another view: the same sender bits, the other round's messages do not match THIS proposal
        s, v, _, _ = bv.verify_messages_wire(wire, off, height, rnd + 1, raw=r.raw, proposal_round=rnd)
        assert (s == want_sender).all() and not v.any()
        # the older two-step route still answers the same afterwards (columns restaged cleanly)
        got, _, _ = bv.is_valid_validator_wire(wire, off)
        assert (got == want_sender).all()
        s, v, rows, t = bv.verify_messages_wire(*WCASE.pack([]), height, rnd, raw=r.raw)
        assert len(s) == 0 and len(v) == 0 and t.has_quorum == 0
    finally:
        bv.close()
