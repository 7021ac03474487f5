"""GPU + host mirror: handlePrepare / handleCommit / batched ingest through the C++ host
(libibft_host.so) with the GPU BatchVerifier must give the same surviving-sender set,
quorum decision and seal list as the stock per-message path whose Verifier is the CPU
oracle (what an application's crypto Backend would answer)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build_round(oracle, n, seed, byzantine):
    from oracle import wire as W, workload as WL
    r = WL.make_round(n, seed, byzantine=byzantine, with_envelopes=False)
    proposer = r.addrs[0].tobytes()
    proposal = W.IbftMessage(view=W.View(r.height, r.round), sender=proposer, type=W.PREPREPARE,
                             payload=W.preprepare_body(W.Proposal(r.raw, r.round), r.proposal_hash, None))
    proposal.signature = oracle.sign(r.sks[0], oracle.keccak256(proposal.payload_no_sig()))
    commits, prepares = [], []
    for i in range(n):
        kind = r.kinds[i]
        hsh = r.hash32[i].tobytes()[: int(r.hash_len[i])]
        seal = r.seal65[i].tobytes()
        if kind == "len64":
            seal = seal[:64]
        body = None if kind == "nil_payload" else W.commit_body(hsh, seal)
        m = W.IbftMessage(view=W.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=W.COMMIT, payload=body)
        m.signature = oracle.sign(r.sks[i], oracle.keccak256(m.payload_no_sig()))
        if kind == "stolen_seal":           # also make the envelope signature wrong for the ingest test
            m.signature = oracle.sign(r.sks[(i + 1) % n], oracle.keccak256(m.payload_no_sig()))
        commits.append(m)
        if i:                               # PREPAREs come from everyone but the proposer
            p = W.IbftMessage(view=W.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=W.PREPARE,
                              payload=W.prepare_body(hsh if kind != "nil_payload" else r.proposal_hash))
            p.signature = oracle.sign(r.sks[i], oracle.keccak256(p.payload_no_sig()))
            prepares.append(p)
    return r, proposal, prepares, commits


def _oracle_verifier(oracle, r):
    """The per-message Verifier an application would implement, answered by the CPU oracle."""
    vs = oracle.ValSet(r.addrs, r.power)

    def is_valid_proposal_hash(prop, hsh):
        return prop is not None and hsh is not None and hsh == oracle.proposal_hash(prop[0], prop[1])

    def is_valid_committed_seal(hsh, seal):
        if hsh is None or seal is None or len(hsh) != 32 or len(seal[1]) != 65 or len(seal[0]) != 20:
            return False
        a = oracle.recover_address(hsh, seal[1])
        return a is not None and a == seal[0] and vs.index(a) >= 0

    def is_valid_validator(wire):
        import go_ibft_amd.hostlib as H
        from oracle import pyref  # noqa: F401  (kept for parity with the CPU tests' imports)
        pns = H.payload_no_sig(wire)
        sig, frm = _sig_from(wire)
        if len(sig) != 65 or len(frm) != 20:
            return False
        a = oracle.recover_address(oracle.keccak256(pns), sig)
        return a is not None and a == frm and vs.index(a) >= 0
    return is_valid_proposal_hash, is_valid_committed_seal, is_valid_validator


def _sig_from(wire):
    """Pull `from` (field 2) and `signature` (field 3) out of top-level wire bytes."""
    pos, frm, sig = 0, b"", b""
    while pos < len(wire):
        tag = wire[pos]; pos += 1
        if tag & 7 == 2:
            ln, shift = 0, 0
            while True:
                b = wire[pos]; pos += 1
                ln |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            val = wire[pos:pos + ln]; pos += ln
            if tag >> 3 == 2: frm = val
            if tag >> 3 == 3: sig = val
        else:
            while wire[pos] & 0x80: pos += 1
            pos += 1
    return sig, frm


@pytest.mark.parametrize("n,byz", [(16, False), (100, True), (300, True)])
def test_handle_prepare_commit_batch_equals_stock(gpu_verifier, oracle, n, byz):
    import go_ibft_amd.hostlib as H
    r, proposal, prepares, commits = _build_round(oracle, n, 900 + n, byz)
    gpu_verifier.set_validators(r.height, r.addrs, r.power)
    powers = {r.addrs[i].tobytes(): int(r.power[i]) for i in range(n)}
    f1, f2, f3 = _oracle_verifier(oracle, r)
    results = []
    for batch in (False, True):
        h = H.Host()
        assert h.vm_init(powers)
        h.set_state(r.height, r.round, proposal.encode())
        h.set_verifier(f1, f2, f3)
        h.attach_gpu(gpu_verifier)
        h.use_batch(batch)
        for m in prepares + commits:
            assert h.store_add(m.encode()) == 0
        okp, prepared = h.handle_prepare(r.height, r.round)
        okc, seals = h.handle_commit(r.height, r.round)
        results.append((okp, sorted(prepared), okc, sorted(seals), h.store_num(r.height, r.round, 1),
                        h.store_num(r.height, r.round, 2)))
        h.close()
    assert results[0] == results[1]
    okp, prepared, okc, seals, n_pr, n_cm = results[1]
    assert okc and okp
    honest = [i for i in range(n) if r.kinds[i] == ""]
    assert sorted(seals) == sorted((r.addrs[i].tobytes(), r.seal65[i].tobytes()) for i in honest)
    assert n_cm == len(honest)


def test_batched_ingest_equals_per_message(gpu_verifier, oracle):
    """IsValidValidator for a burst of incoming messages in one device call."""
    import go_ibft_amd.hostlib as H
    n = 64
    r, proposal, prepares, commits = _build_round(oracle, n, 4242, True)
    gpu_verifier.set_validators(r.height, r.addrs, r.power)
    powers = {r.addrs[i].tobytes(): int(r.power[i]) for i in range(n)}
    f1, f2, f3 = _oracle_verifier(oracle, r)
    wires = [m.encode() for m in commits]
    h1, h2 = H.Host(), H.Host()
    for h in (h1, h2):
        h.vm_init(powers)
        h.set_state(r.height, r.round, proposal.encode())
        h.set_verifier(f1, f2, f3)
    h2.attach_gpu(gpu_verifier)
    stock = [h1.add_message(w) for w in wires]
    batched = h2.add_messages_batch(wires)
    assert stock == batched
    assert 0 in stock and 2 in stock          # some rejected senders, and the quorum signal fired
    assert h1.store_num(r.height, r.round, 2) == h2.store_num(r.height, r.round, 2)


def test_round_change_certificate_batch_equals_stock(gpu_verifier, oracle):
    """§8f rank 2: validateProposal on a PREPREPARE for round 1 whose RoundChangeCertificate holds a
    quorum of ROUND-CHANGE messages, each carrying a PreparedCertificate (proposal + quorum of
    PREPAREs) — O(N²) nested signatures.  Batch mode sends all of them to the device in ONE call and
    must decide like the stock per-message walk (oracle-backed Verifier), for a valid certificate
    and for single corrupted signatures at every nesting level."""
    import go_ibft_amd.hostlib as H
    from oracle import wire as W, workload as WL
    n = 40
    r = WL.make_round(n, 2024)
    addrs = [r.addrs[i].tobytes() for i in range(n)]
    quorum = 2 * n // 3 + 1
    raw = r.raw
    proposer = lambda rnd: addrs[rnd % n]

    def signed(m, i):
        m.signature = oracle.sign(r.sks[i], oracle.keccak256(m.payload_no_sig()))
        return m
    h0 = oracle.proposal_hash(raw, 0)
    pp0 = signed(W.IbftMessage(view=W.View(1, 0), sender=addrs[0], type=W.PREPREPARE,
                               payload=W.preprepare_body(W.Proposal(raw, 0), h0, None)), 0)
    prepares0 = [signed(W.IbftMessage(view=W.View(1, 0), sender=addrs[i], type=W.PREPARE, payload=W.prepare_body(h0)), i)
                 for i in range(1, quorum)]

    def build(corrupt=None):
        prs = list(prepares0)
        pp = pp0
        if corrupt == "nested_prepare":
            bad = W.IbftMessage(view=W.View(1, 0), sender=addrs[5], type=W.PREPARE, payload=W.prepare_body(h0))
            bad.signature = prepares0[5].signature      # someone else's signature
            prs[4] = bad
        if corrupt == "nested_proposal":
            pp = W.IbftMessage(view=W.View(1, 0), sender=addrs[0], type=W.PREPREPARE,
                               payload=W.preprepare_body(W.Proposal(raw, 0), h0, None), signature=prepares0[0].signature)
        pc = W.prepared_certificate(pp, prs)
        rcs = []
        for i in range(quorum):
            m = W.IbftMessage(view=W.View(1, 1), sender=addrs[i + 2], type=W.ROUND_CHANGE,
                              payload=W.round_change_body(W.Proposal(raw, 0), pc))
            signed(m, i + 2)
            if corrupt == "rc_envelope" and i == 7:
                m.signature = bytes(65)
            rcs.append(m)
        h1 = oracle.proposal_hash(raw, 0)            # hash of (raw, maxRound = 0) is what round 1 must carry
        top = W.IbftMessage(view=W.View(1, 1), sender=proposer(1), type=W.PREPREPARE,
                            payload=W.preprepare_body(W.Proposal(raw, 1), oracle.proposal_hash(raw, 1),
                                                      W.round_change_certificate(rcs)))
        return signed(top, 1).encode(), h1

    gpu_verifier.set_validators(1, r.addrs, r.power)
    f1, f2, f3 = _oracle_verifier(oracle, r)
    powers = {a: 1 for a in addrs}
    for corrupt, expect in ((None, True), ("rc_envelope", False), ("nested_prepare", True), ("nested_proposal", True)):
        # a bad signature inside ONE rc's PC only invalidates that PC (validPC false -> PC ignored); with all
        # PCs identical every PC is dropped, no (round, hash) tuple remains and the proposal is accepted
        wire, _ = build(corrupt)
        out = []
        for batch in (False, True):
            h = H.Host()
            assert h.vm_init(powers)
            h.set_id(addrs[n - 1])
            h.set_verifier(f1, f2, f3, is_proposer=lambda who, hh, rr: who == proposer(rr))
            h.attach_gpu(gpu_verifier)
            h.use_batch(batch)
            out.append(h.validate_proposal(wire, 1, 1))
            if batch:
                senders, _ = h.last_cert_batch()
                assert senders == quorum + quorum * (1 + len(prepares0))      # every nested signature, one batch
            h.close()
        assert out[0] == out[1] == expect, (corrupt, out)


def test_extended_rcc_with_quorum_sized_certificates_batch_equals_stock(gpu_verifier, oracle):
    """§8f rank 2, handleRoundChangeMessage (core/ibft.go:470-512) at N = 256: the store holds ROUND-CHANGE
    messages of round 1 from more than a quorum of validators, each carrying a PreparedCertificate with a PREPREPARE
    and Q − 1 PREPAREs — ≈ Q² nested signatures and as many proposal-hash checks.  Batch mode answers all of them
    from ONE device sender batch and one hash batch (all carry the same last prepared proposal) and must build the
    same extended RCC as the per-message walk backed by the CPU oracle: the message whose certificate holds a bad
    PREPARE signature and the one whose certificate belongs to another proposal are left out by both."""
    import go_ibft_amd.hostlib as H
    from oracle import wire as W, workload as WL
    n = 256
    r = WL.make_round(n, 4096)
    addrs = [r.addrs[i].tobytes() for i in range(n)]
    idx = {a: i for i, a in enumerate(addrs)}
    quorum = 2 * n // 3 + 1
    raw = r.raw
    proposer = lambda hh, rr: addrs[(hh + rr) % n]

    def signed(m):
        m.signature = oracle.sign(r.sks[idx[m.sender]], oracle.keccak256(m.payload_no_sig()))
        return m

    def certificate(raw_block, bad_prepare=False):
        h0 = oracle.proposal_hash(raw_block, 0)
        pp = signed(W.IbftMessage(view=W.View(1, 0), sender=proposer(1, 0), type=W.PREPREPARE,
                                  payload=W.preprepare_body(W.Proposal(raw_block, 0), h0, None)))
        prs = [signed(W.IbftMessage(view=W.View(1, 0), sender=a, type=W.PREPARE, payload=W.prepare_body(h0)))
               for a in addrs if a != pp.sender][: quorum - 1]
        if bad_prepare:
            prs[17].signature = prs[18].signature
        return pp, prs
    good, bad_sig, other = certificate(raw), certificate(raw, bad_prepare=True), certificate(b"another block")
    senders = addrs[5: 5 + quorum + 4]
    rcs = []
    for k, a in enumerate(senders):
        cert = bad_sig if k == 3 else other if k == 9 else good
        rcs.append(signed(W.IbftMessage(view=W.View(1, 1), sender=a, type=W.ROUND_CHANGE,
                                        payload=W.round_change_body(W.Proposal(raw, 0), W.prepared_certificate(*cert)))))
    wires = [m.encode() for m in rcs]
    gpu_verifier.set_validators(1, r.addrs, r.power)
    f1, f2, f3 = _oracle_verifier(oracle, r)
    out = {}
    for batch in (False, True):
        h = H.Host()
        assert h.vm_init({a: 1 for a in addrs})
        h.set_verifier(f1, f2, f3, is_proposer=lambda who, hh, rr: who == proposer(hh, rr))
        h.set_state(1, 1, None)
        h.attach_gpu(gpu_verifier)
        h.use_batch(batch)
        for wv in wires:
            assert h.store_add(wv) == 0
        out[batch] = sorted(h.handle_round_change(1, 1))
        if batch:
            nsend, nhash = h.last_cert_batch()
            assert nsend == len(rcs) * quorum and nhash == len(rcs) * quorum     # every nested signature and hash
            assert h.fallbacks() == 0
        h.close()
    assert out[False] == out[True]
    assert len(out[True]) == len(rcs) - 2 and wires[3] not in out[True] and wires[9] not in out[True]


@pytest.mark.parametrize("sets,rows", [(False, False), (True, False), (True, True)])
def test_ingest_wire_through_the_device(gpu_verifier, oracle, sets, rows):
    """§8f rank 1 with the real backend.  sets off: a micro-batch of raw messages → one ibft_verify_senders_wire call.
    sets on (default): the same bytes → one ibft_verify_messages_wire call that also settles the handlePrepare /
    handleCommit closure of every PREPARE / COMMIT of the current view (both signatures of a COMMIT in one verdict
    launch), so handlePrepare / handleCommit then decide without another device call.  rows on (default): what the device
    judged completely is stored as rows and never decoded (include/ibft_host.h: ibft_host_use_rows).  Re-delivery → the
    verdict cache / the stored row.  Everything equals per-message IBFT.AddMessage and the stock walks with the
    oracle-backed verifier."""
    import go_ibft_amd.hostlib as H
    r, proposal, prepares, commits = _build_round(oracle, 300, 77, byzantine=True)
    gpu_verifier.set_validators(r.height, r.addrs, r.power)
    powers = {r.addrs[i].tobytes(): int(r.power[i]) for i in range(r.n)}
    f1, f2, f3 = _oracle_verifier(oracle, r)
    wires = [m.encode() for m in prepares + commits]
    ref, ing = H.Host(), H.Host()
    for h in (ref, ing):
        assert h.vm_init(powers)
        h.set_state(r.height, r.round, proposal.encode())
        h.set_verifier(f1, f2, f3)
    ing.attach_gpu(gpu_verifier)
    ing.use_batch(True)
    ing.use_sets(sets)
    ing.use_rows(rows)
    ing.enable_quorum_index()
    expect = [ref.add_message(x) for x in wires]
    got, asked, hits, calls = ing.ingest_wire(wires)
    assert got == expect and (asked, hits, calls) == (len(wires), 0, 1)   # ONE device call for the micro-batch either way
    assert ing.last_set_rows() >= 0.9 * len(wires) if sets else ing.last_set_rows() == 0
    assert ing.rows_kept >= 0.8 * sum(1 for x in got if x > 0) if rows else ing.rows_kept == 0
    assert 0 in got and 2 in got
    again, asked, hits, calls = ing.ingest_wire(wires)
    assert (asked, hits, calls) == (0, len(wires), 0) and [x != 0 for x in again] == [x != 0 for x in expect]
    for t in (1, 2):
        assert ref.store_num(r.height, r.round, t) == ing.store_num(r.height, r.round, t)
    okp, prepared = ref.handle_prepare(r.height, r.round)
    okp2, prepared2 = ing.handle_prepare(r.height, r.round)
    assert (okp, sorted(prepared)) == (okp2, sorted(prepared2))
    hits_p = ing.closure_hits()
    okc, seals = ref.handle_commit(r.height, r.round)
    okc2, seals2 = ing.handle_commit(r.height, r.round)
    assert okc and (okc, sorted(seals)) == (okc2, sorted(seals2))
    for t in (1, 2):
        assert ref.store_num(r.height, r.round, t) == ing.store_num(r.height, r.round, t)
    for t in (1, 2):
        assert sorted(ref.store_get_valid(r.height, r.round, t)) == sorted(ing.store_get_valid(r.height, r.round, t))
    if sets:    # every stored message had its closure verdict waiting
        assert hits_p >= len(prepared2) and ing.closure_hits() >= len(seals2)
    else:
        assert hits_p == 0 and ing.closure_hits() == 0
    assert ing.fallbacks() == 0
    ref.close(); ing.close()


@pytest.mark.parametrize("rc_rows,roots_first", [(True, 2), (False, 2), (True, 1)])
def test_certificates_judged_on_arrival_through_the_device(gpu_verifier, oracle, rc_rows, roots_first):
    """§8f ranks 1 + 2 with the real backend at N = 128: ROUND-CHANGE messages with quorum-sized PreparedCertificates and the
    PREPREPARE whose RoundChangeCertificate is made of them arrive as wire bytes; IngestWire settles every nested signature
    and proposal-hash check in ONE ibft_verify_certificates_wire call per micro-batch (no PayloadNoSig re-marshal of nested
    messages on the host), handleRoundChangeMessage / handlePrePrepare then ask the device nothing — and decide exactly like
    the per-message walks backed by the CPU oracle, Byzantine certificates included.  rc_rows (default): the certificate of a
    ROUND_CHANGE message is judged from the device's rows on arrival (validPC + proposalMatchesCertificate in row form) and
    never decoded; off: decoded, verdicts noted in the nested objects, the walk over them.  roots_first = 1: the carriers'
    envelopes are judged before any tree is expanded (ibft_host_cert_roots_first: one more device call, a forged carrier's
    tree is never looked at); 2 = the adaptive default, which stays with the single call here."""
    import go_ibft_amd.hostlib as H
    from oracle import wire as W, workload as WL
    n = 128
    r = WL.make_round(n, 4242)
    addrs = [r.addrs[i].tobytes() for i in range(n)]
    idx = {a: i for i, a in enumerate(addrs)}
    quorum = 2 * n // 3 + 1
    raw = r.raw
    proposer = lambda hh, rr: addrs[(hh + rr) % n]

    def signed(m):
        m.signature = oracle.sign(r.sks[idx[m.sender]], oracle.keccak256(m.payload_no_sig()))
        return m

    def certificate(raw_block, bad_prepare=False):
        h0 = oracle.proposal_hash(raw_block, 0)
        pp = signed(W.IbftMessage(view=W.View(1, 0), sender=proposer(1, 0), type=W.PREPREPARE,
                                  payload=W.preprepare_body(W.Proposal(raw_block, 0), h0, None)))
        prs = [signed(W.IbftMessage(view=W.View(1, 0), sender=a, type=W.PREPARE, payload=W.prepare_body(h0)))
               for a in addrs if a != pp.sender][: quorum - 1]
        if bad_prepare:
            prs[17].signature = prs[18].signature
        return pp, prs
    good, bad_sig, other = certificate(raw), certificate(raw, bad_prepare=True), certificate(b"another block")
    senders = addrs[5: 5 + quorum + 4]
    rcs = []
    for k, a in enumerate(senders):
        cert = bad_sig if k == 3 else other if k == 9 else good
        rcs.append(signed(W.IbftMessage(view=W.View(1, 1), sender=a, type=W.ROUND_CHANGE,
                                        payload=W.round_change_body(W.Proposal(raw, 0), W.prepared_certificate(*cert)))))
    forged = W.IbftMessage(view=W.View(1, 1), sender=addrs[2], type=W.ROUND_CHANGE, signature=rcs[0].signature,
                           payload=W.round_change_body(W.Proposal(raw, 0), W.prepared_certificate(*good)))  # envelope not signed by From
    wires = [m.encode() for m in rcs] + [forged.encode()]
    # the proposer of round 1 re-proposes the prepared block with the honest ROUND-CHANGE messages as its certificate
    honest = [m for k, m in enumerate(rcs) if k not in (3, 9)][:quorum]
    pp1 = signed(W.IbftMessage(view=W.View(1, 1), sender=proposer(1, 1), type=W.PREPREPARE,
                               payload=W.preprepare_body(W.Proposal(raw, 1), oracle.proposal_hash(raw, 1),
                                                         W.round_change_certificate(honest))))
    pp_bad = signed(W.IbftMessage(view=W.View(1, 1), sender=proposer(1, 1), type=W.PREPREPARE,
                                  payload=W.preprepare_body(W.Proposal(raw, 1), oracle.proposal_hash(raw, 1),
                                                            W.round_change_certificate(honest[:-1] + [forged]))))  # one envelope inside is forged
    gpu_verifier.set_validators(1, r.addrs, r.power)
    f1, f2, f3 = _oracle_verifier(oracle, r)
    ref, ing = H.Host(), H.Host()
    for h in (ref, ing):
        assert h.vm_init({a: 1 for a in addrs})
        h.set_verifier(f1, f2, f3, is_proposer=lambda who, hh, rr: who == proposer(hh, rr))
        h.set_id(addrs[0])
        h.set_state(1, 0, None)
    ing.attach_gpu(gpu_verifier)
    ing.use_batch(True)
    ing.use_rc_rows(rc_rows)
    ing.cert_roots_first(roots_first)
    expect = [ref.add_message(x) for x in wires]
    got, rows, hits, calls = ing.ingest_wire(wires)
    extra = 1 if roots_first == 1 else 0
    assert [x != 0 for x in got] == [x != 0 for x in expect] and got[-1] == 0 and calls == 1 + extra   # ONE device call for the micro-batch
    assert ing.rc_from_rows == (len(rcs) if rc_rows else 0) and ing.roots_first_calls == extra  # (the forged envelope is not stored)
    c_calls, c_rows, _ = ing.cert_stats()
    assert c_calls == 1 and c_rows == (len(wires) - extra) * (1 + quorum)                  # every message of every (authenticated) tree judged
    for h in (ref, ing):
        h.set_state(1, 1, None)
    a = sorted(ref.handle_round_change(1, 1))
    b = sorted(ing.handle_round_change(1, 1))
    assert a == b and len(b) == len(rcs) - 2 and set(b) <= set(wires)                      # handed out as the bytes that came in
    nsend, nhash = ing.last_cert_batch()
    assert nsend == 0 and nhash == 0 and ing.cert_stats()[2] == len(rcs) * quorum          # nothing left to ask: the tables answered
    # the PREPREPAREs of round 1: 1.5 MB envelopes (the host hashes those itself, by the stock route), trees judged on arrival
    for variant, ok in ((pp_bad, False), (pp1, True)):
        for h in (ref, ing):
            h.set_state(1, 1, None)
        e = ref.add_message(variant.encode())
        g, _, _, _ = ing.ingest_wire([variant.encode()])
        assert (g[0] != 0) == (e != 0) and e != 0
        pa, pb = ref.handle_preprepare(1, 1), ing.handle_preprepare(1, 1)
        assert pa == pb and (pb is not None) == ok
        assert ing.last_cert_batch()[0] == 0
    assert ing.pp_from_rows == (2 if rc_rows else 0)       # validateProposal's certificate rules: from the rows, or by the object walk
    assert ing.fallbacks() == 0
    ref.close(); ing.close()


@pytest.mark.parametrize("rows", [False, True])
@pytest.mark.parametrize("scenario", ["all", "short_by_one", "proposer_prepares", "weighted"])
def test_quorum_decision_taken_from_the_device(gpu_verifier, oracle, rows, scenario):
    """a8 through the mirror (ibft_host_use_device_quorum): handlePrepare asks ibft_tally_prepare — HasPrepareQuorum with
    proposalMessage.From, core/validator_manager.go:99-127 — and handleCommit ibft_tally over the senders that survived the
    walk; the decision equals the stock mirror's (hasQuorumByMsgType with the oracle-backed verifier), object mode and row
    mode, and the mirror's own quorum index never disagrees."""
    import go_ibft_amd.hostlib as H
    from oracle import wire as W
    n = 40
    r, proposal, prepares, commits = _build_round(oracle, n, 5150, False)
    power = r.power.copy()
    if scenario == "weighted":
        power[:] = 1
        power[3] = 60                              # one heavy validator …
    gpu_verifier.set_validators(r.height, r.addrs, power)
    powers = {r.addrs[i].tobytes(): int(power[i]) for i in range(n)}
    quorum = 2 * int(power.sum()) // 3 + 1
    if scenario == "short_by_one":                # PREPAREs + the proposer's seat = quorum − 1; COMMITs = quorum − 1
        prepares, commits = prepares[: quorum - 2], commits[: quorum - 1]
    elif scenario == "proposer_prepares":         # a (correctly signed) PREPARE of the proposer among the others
        p = W.IbftMessage(view=W.View(r.height, r.round), sender=r.addrs[0].tobytes(), type=W.PREPARE,
                          payload=W.prepare_body(r.proposal_hash))
        p.signature = oracle.sign(r.sks[0], oracle.keccak256(p.payload_no_sig()))
        prepares = prepares + [p]
    elif scenario == "weighted":                  # … who stays silent: 39 of 99 < 67
        prepares = [m for m in prepares if m.sender != r.addrs[3].tobytes()]
        commits = [m for m in commits if m.sender != r.addrs[3].tobytes()]
    f1, f2, f3 = _oracle_verifier(oracle, r)
    wires = [m.encode() for m in prepares + commits]
    ref, dev = H.Host(), H.Host()
    for h in (ref, dev):
        assert h.vm_init(powers)
        h.set_state(r.height, r.round, proposal.encode())
        h.set_verifier(f1, f2, f3)
    dev.attach_gpu(gpu_verifier)
    dev.use_batch(True)
    dev.use_rows(rows)
    dev.enable_quorum_index()
    dev.use_device_quorum(True)
    for x in wires:
        ref.add_message(x)
    dev.ingest_wire(wires)
    assert (dev.rows_kept > 0) == rows
    okp, prepared = ref.handle_prepare(r.height, r.round)
    okp2, prepared2 = dev.handle_prepare(r.height, r.round)
    okc, seals = ref.handle_commit(r.height, r.round)
    okc2, seals2 = dev.handle_commit(r.height, r.round)
    assert (okp, sorted(prepared), okc, sorted(seals)) == (okp2, sorted(prepared2), okc2, sorted(seals2))
    assert (okp, okc) == {"all": (True, True), "short_by_one": (False, False), "proposer_prepares": (False, True),
                          "weighted": (False, False)}[scenario]
    assert dev.device_quorum_stats() == (2, 0)
    ref.close(); dev.close()


@pytest.mark.parametrize("device_quorum", [False, True])
def test_go_hoststore_call_sequence_with_the_device_attached(gpu_verifier, oracle, device_quorum):
    """shim/go/hoststore's C call sequence (read from the Go file, tests/test_hoststore_sequence.py) replayed with the real
    backend: New (ibft_host_attach_gpu) → SetValidators → SetState → AddWireMessages × k → Drain → HandlePrepare →
    HandleCommit ≡ the stock mirror with the oracle-backed per-message Verifier, on a Byzantine round with real signatures."""
    import go_ibft_amd.hostlib as H
    from test_hoststore_sequence import Replay
    r, proposal, prepares, commits = _build_round(oracle, 200, 606, True)
    gpu_verifier.set_validators(r.height, r.addrs, r.power)
    powers = {r.addrs[i].tobytes(): int(r.power[i]) for i in range(r.n)}
    f1, f2, f3 = _oracle_verifier(oracle, r)
    stock = H.Host()
    assert stock.vm_init(powers)
    stock.set_state(r.height, r.round, proposal.encode())
    stock.set_verifier(f1, f2, f3)
    wires = [m.encode() for m in prepares + commits]
    expect = [stock.add_message(x) for x in wires]
    rp = Replay(None, None, gpu=gpu_verifier)
    rp.run("New", device_quorum=device_quorum, max_rows=0, linger_us=50)
    rp.run("SetValidators", powers=powers)
    rp.run("SetState", height=r.height, round=r.round, proposal=proposal.encode())
    for k in range(0, len(wires), 64):
        rp.run("AddWireMessages", raw=wires[k:k + 64])
    rp.run("Drain")
    assert rp.stats.ingested == len(wires) and rp.stats.stored == sum(1 for x in expect if x > 0)
    assert rp.stats.device_calls >= 1 and rp.stats.device_calls <= rp.stats.batches + 2
    rp.run("HandlePrepare", height=r.height, round=r.round)
    okp, prepared = stock.handle_prepare(r.height, r.round)
    assert rp.results[("HandlePrepare", "ibft_host_handle_prepare")] == int(okp) == 1
    assert sorted(H.unpack(rp.taken)) == sorted(prepared)
    rp.run("HandleCommit", height=r.height, round=r.round)
    okc, seals = stock.handle_commit(r.height, r.round)
    assert rp.results[("HandleCommit", "ibft_host_handle_commit")] == int(okc) == 1
    assert sorted(H.unpack_seals(rp.taken)) == sorted(seals)
    rp.run("RowsKept")
    assert rp.results[("RowsKept", "ibft_host_rows_kept")] > 0
    rp.run("DeviceQuorumStats")
    assert rp.dq == ((2, 0) if device_quorum else (0, 0))
    rp.run("Close")
    stock.close()
