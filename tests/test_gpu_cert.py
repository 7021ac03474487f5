"""§8f rank 2 from the transport's bytes, on the GPU: ibft_verify_certificates_wire — the certificate tree of PREPREPARE /
ROUND_CHANGE messages expanded level by level on the device, every nested message a row of ONE verdict launch — against
the independent oracle (oracle/wire_cert.py): tree shape, canonical / NEEDS_HOST classes, sender, hash and self bits."""
import numpy as np
import pytest

import cert_cases as CC
from oracle import wire
from oracle import wire_cert as WC
from oracle import workload as W

pytestmark = pytest.mark.gpu


def run(bv, r, label, msgs, cap=None):
    buf, off = CC.pack(msgs)
    exp = WC.expected_tree(msgs, r.addrs, rows_cap=min(cap or bv.max_rows, bv.max_rows))
    if exp is None:
        with pytest.raises(RuntimeError, match="-7"):
            bv.verify_certificates_wire(buf, off, rows_cap=cap)
        return None
    n, nodes, rows, cls, sender, hb, sb = bv.verify_certificates_wire(buf, off, rows_cap=cap)
    CC.compare(label, exp, n, nodes, rows, cls, sender, hb, sb)
    return exp


@pytest.mark.parametrize("cache", [False, True])
def test_handmade_and_fuzzed_trees(cache):
    import go_ibft_amd.verifier as V
    r = W.make_round(8, 811, height=5, round_=1)
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if cache else 0, max_rows=4096)
    try:
        bv.set_validators(5, r.addrs, r.power)
        for _ in range(2 if cache else 1):  # second pass on the warm kernels
            for label, msgs in CC.handmade(r):
                run(bv, r, label, msgs)
        for i, msgs in enumerate(CC.fuzz_batches(r, 150, 4242)):
            run(bv, r, f"fuzz {i}", msgs)
        msgs = [m.encode() for m in CC.honest_round_change_set(r)]
        assert run(bv, r, "cap exactly", msgs, cap=56) is not None
        assert run(bv, r, "cap one short", msgs, cap=55) is None
        assert run(bv, r, "cap below the call's messages", msgs, cap=7) is None
    finally:
        bv.close()


def test_longer_than_the_device_hashes(gpu_verifier):
    r = W.make_round(4, 812, height=5, round_=1, raw_len=(1 << 20) + 5)
    gpu_verifier.set_validators(5, r.addrs, r.power)
    rc = CC.round_change(r, 0, 5, 2, wire.Proposal(r.raw, 1), CC.pc_bytes(r, 5, 1, 1, [2, 3]))
    exp = run(gpu_verifier, r, "1 MiB proposal", [rc.encode(), CC.prepare(r, 1, 5, 2).encode()])
    both = WC.CLASS_DIGEST_BY_HOST | WC.CLASS_PROPOSAL_BY_HOST
    assert exp.cls[0] == both and exp.cls[2] == both and exp.cls[1] == 0 and exp.sender_ok[3]


def test_round_change_certificate_n256(gpu_verifier):
    """BASELINE-style worst case at N = 256: a PREPREPARE whose RoundChangeCertificate holds 171 ROUND_CHANGE messages, each
    with a PreparedCertificate of 1 + 170 messages — 29 412 nested signatures from one 4.1 MB message, one call; then the
    same with a fifth of the nested PREPAREs corrupted."""
    n = 256
    r = W.make_round(n, 815, height=5, round_=1, raw_len=256)
    gpu_verifier.set_validators(5, r.addrs, r.power)
    q = (2 * n) // 3 + 1
    proposer = 1
    h1 = r.proposal_hash
    pm = CC.preprepare(r, proposer, 5, 1)
    prepares = [CC.prepare(r, j, 5, 1, h=h1) for j in range(n) if j != proposer][: q - 1]
    pcb = wire.prepared_certificate(pm, prepares)
    rcs = [CC.round_change(r, i, 5, 2, wire.Proposal(r.raw, 1), pcb) for i in range(q)]
    top = CC.preprepare(r, 2, 5, 2, rcc=wire.round_change_certificate(rcs))
    exp = run(gpu_verifier, r, "rcc n=256", [top.encode()])
    assert exp.n_rows == 1 + q + q * q and all(exp.sender_ok[1:]) and all(exp.hash_bit[1 + q:]) and exp.self_bit[0]
    assert exp.cls[0] == WC.CLASS_DIGEST_BY_HOST and not any(exp.cls[1:])  # the 4 MB envelope itself: its digest is the host's
    # Byzantine: every fifth nested PREPARE of every certificate is forged, for another proposal, or from a stranger
    outsider = b"\x07" * 32
    bad = []
    for k, m in enumerate(prepares):
        if k % 5 == 0:
            m = CC.prepare(r, (k + 3) % n, 5, 1, h=h1, sk=outsider)
        elif k % 5 == 1 and k % 2:
            m = CC.prepare(r, (k + 3) % n, 5, 1, h=b"\x00" * 32)
        bad.append(m)
    pcb2 = wire.prepared_certificate(pm, bad)
    rcs2 = [CC.round_change(r, i, 5, 2, wire.Proposal(r.raw, 1), pcb2 if i % 2 else pcb) for i in range(q)]
    top2 = CC.preprepare(r, 2, 5, 2, rcc=wire.round_change_certificate(rcs2))
    exp = run(gpu_verifier, r, "rcc n=256 byzantine", [top2.encode()])
    assert 1000 < sum(1 for x in exp.sender_ok if not x) < exp.n_rows // 4


def test_certificate_longer_than_the_walk_window(gpu_verifier):
    """children on either side of the 16 KiB LDS windows of cert_walk_kernel, more than 64 per window, a batch of containers"""
    r = W.make_round(600, 816, height=5, round_=1, raw_len=40)
    gpu_verifier.set_validators(5, r.addrs, r.power)
    msgs = []
    for k, cnt in enumerate((599, 130, 64, 65, 1, 0, 257)):
        pc = CC.pc_bytes(r, 5, 1, 1, [i for i in range(600) if i != 1][:cnt])
        msgs.append(CC.round_change(r, k, 5, 2, wire.Proposal(r.raw, 1), pc).encode())
    exp = run(gpu_verifier, r, "long certificates", msgs)
    assert exp.n_rows == 7 + 7 + 599 + 130 + 64 + 65 + 1 + 257 and all(exp.sender_ok)


def test_golden_trees(gpu_verifier):
    """tests/golden/cert_trees.json: 2 000 batches (hand-made cases, then byte-level fuzz), one digest per batch over tree shape,
    classes, sender / hash / self bits and parsed fields; the expectation was cross-checked against the google.protobuf runtime when
    the fixture was made (tests/golden/make_cert_golden.py)"""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cert_trees.json")))
    r, batches = CC.golden_batches(gold["count"])
    gpu_verifier.set_validators(5, r.addrs, r.power)
    bad = []
    for bi, msgs in enumerate(batches):
        buf, off = CC.pack(msgs)
        n, nodes, rows, cls, sender, hb, sb = gpu_verifier.verify_certificates_wire(buf, off, rows_cap=4096)
        if CC.digest_actual(n, nodes, rows, cls, sender, hb, sb) != gold["digests"][bi]:
            bad.append(bi)
    assert not bad, bad[:20]


def test_certificate_calls_feed_the_key_cache():
    """The certificate path has no tally kernel to carry the "keys learned" counter to the host: a key-caching context must
    still notice what its certificate calls taught the device, build the tables and serve the next call warm (a regression
    of round 3's device-wide key cache showed up as warm = cold in the bench line's certificates leg)."""
    import go_ibft_amd.verifier as V
    n = 64
    r = W.make_round(n, 871, height=5, round_=1, raw_len=64)
    q = (2 * n) // 3 + 1
    pm = CC.preprepare(r, 1, 5, 1)
    pcb = wire.prepared_certificate(pm, [CC.prepare(r, j, 5, 1) for j in range(n) if j != 1][: q - 1])
    msgs = [CC.round_change(r, i, 5, 2, wire.Proposal(r.raw, 1), pcb).encode() for i in range(q)]
    buf, off = CC.pack(msgs)
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=4096)
    try:
        bv.set_validators(5, r.addrs, r.power)
        for _ in range(3):
            k, _, _, cls, snd, _, _ = bv.verify_certificates_wire(buf, off, rows_cap=4096, want_rows=False)
            assert k == q * (q + 1) and snd.all() and not cls.any()
        tables, warm, cold = bv.cache_stats()
        assert tables >= q and warm >= 1, (tables, warm, cold)       # every signer of the tree has a table; a warm pass ran
    finally:
        bv.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_certificates_sharded_by_carrier_equal_one_call(world):
    """ibft_group_verify_certificates_wire: the trees of a batch split by carrier over `world` contexts (here all on device 0,
    each with its own stream), judged at once, and renumbered — n_rows, every node field (offsets into the WHOLE wire, parents,
    first children, ordinals), the parsed rows, classes and the three masks equal those of ONE call over the whole batch;
    and the oracle's tree.  Batches with fewer carriers than contexts, deep trees (PREPREPARE → ROUND_CHANGE → PREPARE) and
    non-canonical messages included."""
    import go_ibft_amd.verifier as V
    r = W.make_round(8, 811, height=5, round_=1)
    bv = V.BatchVerifier(max_rows=4096)
    g = V.DeviceGroup([0] * world, max_rows_total=4096 * world)
    try:
        bv.set_validators(5, r.addrs, r.power)
        g.set_validators(5, r.addrs, r.power)
        batches = [msgs for _, msgs in CC.handmade(r)] + list(CC.fuzz_batches(r, 60, 77))
        honest = [m.encode() for m in CC.honest_round_change_set(r)]
        batches += [honest, honest[:1], honest + [CC.preprepare_with_rcc(r).encode()] + honest[:3]]
        checked = 0
        for k, msgs in enumerate(batches):
            buf, off = CC.pack(msgs)
            if WC.expected_tree(msgs, r.addrs, rows_cap=4096) is None:
                continue
            one = bv.verify_certificates_wire(buf, off, rows_cap=4096)
            many = g.verify_certificates_wire(buf, off, rows_cap=4096)
            assert one[0] == many[0], k
            for name in one[1].dtype.names:
                if name == "first_child":       # means nothing for a leaf (include/ibftgpu.h): compared where there are children
                    has = one[1]["n_children"] > 0
                    assert (one[1][name][has] == many[1][name][has]).all(), (k, name)
                    assert (many[1][name][~has] < max(many[0], 1)).all(), (k, "a leaf's first_child must stay inside the tree")
                elif name != "pad":
                    assert (one[1][name] == many[1][name]).all(), (k, name)
            assert one[2].tobytes() == many[2].tobytes(), k
            for a, b in zip(one[3:], many[3:]):
                assert (a == b).all(), k
            CC.compare(f"sharded {k}", WC.expected_tree(msgs, r.addrs, rows_cap=4096), *many)
            checked += 1
        assert checked > 40
        with pytest.raises(RuntimeError, match="-7"):              # the merged tree must fit rows_cap
            g.verify_certificates_wire(*CC.pack(honest), rows_cap=55)
    finally:
        g.close()
        bv.close()
