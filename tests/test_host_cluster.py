"""BASELINE config #1 (BASELINE.json configs[0]): "core/consensus_test.go 4-validator cluster, mock Backend, pure-Go CPU
path (plumbing, no GPU)" — restated over the host mirror, because Go cannot run here.

N mirror nodes in one process, Multicast = loop AddMessage over every node (core/mock_test.go:546-550,
core/helpers_test.go:227-231), round-robin proposer (core/helpers_test.go:214-225), one height PREPREPARE → PREPARE →
COMMIT through handlePrePrepare / handlePrepare / handleCommit (tests/cluster_sim.py plays runStates' transitions).
Asserted, per TestConsensus_ValidFlow (core/consensus_test.go:133-248) and TestRunCommit (core/ibft_test.go:1088-1092):
every node inserts the proposer's block and hands InsertProposal the seals {Signer: From, Signature: CommittedSeal} of
the COMMITs that survived its handleCommit — in stock (per-message Verifier), batch (one call per walk) and ingest
(micro-batches of wire bytes) mode, with the reference's always-true mock AND with a Verifier that really rejects:
the reference's "bad commit seal" / "bad hash in PREPARE" scenarios (core/byzantine_test.go:258-362) pass there only
because the byzantine nodes are ≤ f (its mock never returns false from IsValidCommittedSeal); here the
bad messages are rejected, pruned, and absent from the inserted seals."""
import pytest

import cluster_sim as CS
from oracle import wire as W

PP, PR, CM, RC = W.PREPREPARE, W.PREPARE, W.COMMIT, W.ROUND_CHANGE
MODES = ["stock", "batch", "ingest"]

RAW = b"valid block"                      # correctRoundMessage.proposal (core/helpers_test.go:18-37)
HASH = b"proposal hash"
SEAL = b"seal"


def node_addresses(n):                    # generateNodeAddresses, core/consensus_test.go:17-25
    return [b"node %d" % i for i in range(n)]


def mock_build(bad_prepare_hash=False, bad_seal=False):
    """buildBasicPreprepareMessage / …PrepareMessage / …CommitMessage (core/consensus_test.go:28-90) with the
    byzantine flags of core/byzantine_test.go:330-390"""
    def build(nd, type_, view):
        v = W.View(*view)
        if type_ == PP:
            body = W.preprepare_body(W.Proposal(RAW, view[1]), HASH, None)
        elif type_ == PR:
            body = W.prepare_body(b"invalid proposal hash" if (nd.byzantine and bad_prepare_hash) else HASH)
        else:
            body = W.commit_body(HASH, b"invalid committed seal" if (nd.byzantine and bad_seal) else SEAL + b" %d" % nd.index)
        return W.IbftMessage(view=v, sender=nd.address, type=type_, payload=body).encode()
    return build


def mock_verifier(cluster_ref, reject_bad_seal=False):
    """the Backend of TestConsensus_ValidFlow (core/consensus_test.go:160-178); reject_bad_seal adds the
    IsValidCommittedSeal the reference's cluster tests never install"""
    def make(nd):
        def is_valid_proposal_hash(prop, hsh):
            return prop is not None and prop[0] == RAW and prop[1] == 0 and hsh == HASH

        def is_valid_committed_seal(hsh, seal):
            return not reject_bad_seal or (seal is not None and seal[1].startswith(SEAL))

        def is_valid_validator(wire):
            return True

        def is_proposer(frm, height, round_):
            c = cluster_ref[0]
            return frm == c.nodes[c.proposer(height, round_)].address

        def is_valid_proposal(raw):
            return raw == RAW
        return is_valid_proposal_hash, is_valid_committed_seal, is_valid_validator, is_proposer, is_valid_proposal
    return make


def run_cluster(n, mode, *, lockstep=True, byzantine=0, build=None, verifier=None, height=1):
    ref = [None]
    c = CS.Cluster(node_addresses(n), build or mock_build(), (verifier or mock_verifier)(ref), mode=mode, lockstep=lockstep)
    ref[0] = c
    for nd in c.nodes[:byzantine]:        # makeNByzantine, core/helpers_test.go:237-241
        nd.byzantine = True
    try:
        inserted = c.run_height(height)
        return c, inserted
    except Exception:
        c.close()
        raise


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", [4, 7])
def test_valid_flow_every_node_inserts_the_same_block_and_seals(n, mode):
    """TestConsensus_ValidFlow + the seal hand-over of TestRunCommit: with every message delivered before anybody acts,
    every node's InsertProposal receives the proposer's block and the seals of ALL N COMMITs"""
    c, inserted = run_cluster(n, mode)
    try:
        want = frozenset((a, SEAL + b" %d" % i) for i, a in enumerate(node_addresses(n)))
        assert all(x is not None for x in inserted)
        assert {x[0] for x in inserted} == {RAW}
        assert all(x[1] == want for x in inserted), [sorted(x[1]) for x in inserted]
        for nd in c.nodes:                # the PC every node finalised: PREPAREs of everyone but the proposer
            assert len(nd.prepared) == n - 1
            assert nd.host.store_num(1, 0, CM) == n and nd.host.store_num(1, 0, PR) == n - 1
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_valid_flow_eager_wakeups(mode):
    """the same with the reference's timing: a node acts on the SignalEvent of the message that completed its quorum
    (core/ibft.go:1113-1120) — it may finish with any quorum of seals, every one of them a COMMIT somebody sent"""
    n = 7
    c, inserted = run_cluster(n, mode, lockstep=False)
    try:
        sent = {(a, SEAL + b" %d" % i) for i, a in enumerate(node_addresses(n))}
        quorum = c.nodes[0].host.vm_quorum()
        assert quorum == 2 * n // 3 + 1
        for nd, x in zip(c.nodes, inserted):
            assert x is not None and x[0] == RAW
            assert x[1] <= sent and len(x[1]) >= quorum
            assert nd.signals > 0
    finally:
        c.close()


def test_modes_agree_node_by_node():
    """stock ≡ batch ≡ ingest: what each node inserted, its PC and its store, over three heights with the proposer moving
    round-robin (core/helpers_test.go:214-225)"""
    n = 5
    per_mode = {}
    for mode in MODES:
        ref = [None]
        c = CS.Cluster(node_addresses(n), mock_build(), mock_verifier(ref), mode=mode)
        ref[0] = c
        try:
            hist = []
            for height in (1, 2, 3):
                inserted = c.run_height(height)
                assert all(x is not None for x in inserted) and c.proposer(height, 0) == height % n
                hist.append((inserted, [nd.prepared for nd in c.nodes],
                             [nd.host.store_num(height, 0, CM) for nd in c.nodes],
                             [nd.host.store_num(height - 1, 0, CM) for nd in c.nodes]))   # pruned by RunSequence's start
            per_mode[mode] = hist
        finally:
            c.close()
    assert per_mode["stock"] == per_mode["batch"] == per_mode["ingest"]
    assert per_mode["stock"][2][3] == [0] * n


@pytest.mark.parametrize("mode", MODES)
def test_bad_commit_seal_is_rejected_not_just_outvoted(mode):
    """core/byzantine_test.go:258-290 ("bad commit seal", N = 6, f = 1) with an IsValidCommittedSeal that returns false:
    the byzantine node's COMMIT is pruned by handleCommit (messages/messages.go:193-196) and its seal never reaches
    InsertProposal; quorum ⌊2·6/3⌋+1 = 5 is met by the five honest seals"""
    n = 6
    verifier = lambda ref: mock_verifier(ref, reject_bad_seal=True)
    c, inserted = run_cluster(n, mode, byzantine=1, build=mock_build(bad_seal=True), verifier=verifier)
    try:
        honest = frozenset((a, SEAL + b" %d" % i) for i, a in enumerate(node_addresses(n)) if i >= 1)
        assert all(x is not None and x[0] == RAW and x[1] == honest for x in inserted)
        for nd in c.nodes:
            assert nd.host.store_num(1, 0, CM) == n - 1          # the rejected COMMIT was deleted from the store
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_too_many_bad_seals_stall_the_round(mode):
    """f + 1 = 2 rejected seals of 6: four valid COMMITs < quorum 5 — handleCommit never succeeds in round 0 (the round
    timer, out of scope, would take over); with the reference's always-true mock the same cluster would insert"""
    n = 6
    verifier = lambda ref: mock_verifier(ref, reject_bad_seal=True)
    c, inserted = run_cluster(n, mode, byzantine=2, build=mock_build(bad_seal=True), verifier=verifier)
    try:
        assert inserted == [None] * n
        assert all(nd.state == "commit" for nd in c.nodes)
    finally:
        c.close()
    c, inserted = run_cluster(n, mode, byzantine=2, build=mock_build(bad_seal=True))   # the reference's mock: accepts
    try:
        assert all(x is not None and len(x[1]) == n for x in inserted)
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_bad_hash_in_prepare(mode):
    """core/byzantine_test.go:330-362 ("malicious hash in prepare", N = 6, f = 1): IsValidProposalHash rejects the
    byzantine PREPARE, handlePrepare prunes it, the prepared certificate holds the honest ones only; the byzantine
    node's COMMIT (valid) still counts"""
    n = 6
    c, inserted = run_cluster(n, mode, byzantine=2, build=mock_build(bad_prepare_hash=True))
    try:
        # proposer of height 1 is node 1 (byzantine, but a proposer sends no PREPARE); node 0's PREPARE is the bad one
        assert c.proposer(1, 0) == 1
        assert all(x is not None and x[0] == RAW and len(x[1]) == n for x in inserted)
        for nd in c.nodes:
            assert len(nd.prepared) == n - 2                      # n − 1 PREPAREs sent, one rejected
            assert nd.host.store_num(1, 0, PR) == n - 2
    finally:
        c.close()


def _real_crypto(n, seed=5):
    """validators with secp256k1 keys; messages signed by the oracle's signer (test infrastructure), judged by a
    per-message Verifier that the oracle answers — what an application's crypto Backend would be"""
    from oracle import binding as B, workload as WL
    sks = [WL.validator_key(seed, i) for i in range(n)]
    addrs = [B.address(B.pubkey(sk)) for sk in sks]
    raw = bytes(range(200)) * 3
    index = {a: i for i, a in enumerate(addrs)}

    def signed(m, sk):
        m.signature = B.sign(sk, B.keccak256(m.payload_no_sig()))
        return m.encode()

    def build(bad_seal=False, bad_prepare_hash=False, forged_envelope=False):
        def f(nd, type_, view):
            sk = sks[nd.index]
            H0 = B.proposal_hash(raw, view[1])
            v = W.View(*view)
            if type_ == PP:
                body = W.preprepare_body(W.Proposal(raw, view[1]), H0, None)
            elif type_ == PR:
                body = W.prepare_body(B.keccak256(b"other") if (nd.byzantine and bad_prepare_hash) else H0)
            else:
                seal = B.sign(sk, B.keccak256(b"other block")) if (nd.byzantine and bad_seal) else B.sign(sk, H0)
                body = W.commit_body(H0, seal)
            m = W.IbftMessage(view=v, sender=nd.address, type=type_, payload=body)
            if nd.byzantine and forged_envelope and type_ == CM:
                return signed(m, sks[(nd.index + 1) % n])         # somebody else's key under this From
            return signed(m, sk)
        return f

    def verifier(ref):
        def make(nd):
            import go_ibft_amd.hostlib as H

            def is_valid_proposal_hash(prop, hsh):
                return prop is not None and hsh is not None and hsh == B.proposal_hash(prop[0], prop[1])

            def is_valid_committed_seal(hsh, seal):
                if hsh is None or seal is None or len(hsh) != 32 or len(seal[1]) != 65 or len(seal[0]) != 20:
                    return False
                a = B.recover_address(hsh, seal[1])
                return a is not None and a == seal[0] and a in index

            def is_valid_validator(wire):
                from oracle.wire_parse import fields
                f = {num: val for num, wt, val in fields(wire) if wt == 2}
                sig, frm = f.get(3, b""), f.get(2, b"")
                if len(sig) != 65 or len(frm) != 20:
                    return False
                a = B.recover_address(B.keccak256(H.payload_no_sig(wire)), sig)
                return a is not None and a == frm and a in index

            def is_proposer(frm, height, round_):
                c = ref[0]
                return frm == c.nodes[c.proposer(height, round_)].address

            return is_valid_proposal_hash, is_valid_committed_seal, is_valid_validator, is_proposer, lambda r: r == raw
        return make
    return sks, addrs, raw, build, verifier


@pytest.mark.parametrize("mode", MODES)
def test_real_signatures_one_height_with_a_byzantine_node(mode):
    """N = 7 (f = 2, quorum 5) with real keys: node 0 seals another block, node 1 signs its COMMIT envelope with a
    foreign key (rejected by IsValidValidator on arrival, core/ibft.go:1128, never stored).  Five honest seals insert."""
    from oracle import binding as B
    n = 7
    sks, addrs, raw, build, verifier = _real_crypto(n)
    per_node_build = build(bad_seal=True)
    forged = build(forged_envelope=True)

    def mixed(nd, type_, view):
        return forged(nd, type_, view) if nd.index == 1 else per_node_build(nd, type_, view)
    ref = [None]
    c = CS.Cluster(addrs, mixed, verifier(ref), mode=mode)
    ref[0] = c
    c.nodes[0].byzantine = c.nodes[1].byzantine = True
    try:
        inserted = c.run_height(2)                                 # proposer = node 2
        H0 = B.proposal_hash(raw, 0)
        honest = frozenset((addrs[i], B.sign(sks[i], H0)) for i in range(2, n))
        assert all(x is not None and x[0] == raw and x[1] == honest for x in inserted)
        for nd in c.nodes:
            assert nd.host.store_num(2, 0, CM) == n - 2            # node 1's never stored, node 0's pruned
    finally:
        c.close()


# ---- rounds that change (round 4): the timer is a tick(), everything else is the reference's ------------------------------

def fh(raw: bytes, rnd: int) -> bytes:
    """the mock Backend's proposal hash: a function of (RawProposal, Round), like the tuple ibft_test.go:2679-2701 pins"""
    return (b"H|" + raw[:20] + b"|%d" % rnd).ljust(32, b".")[:32]


def rc_build(proposals, lie_about_prepared=None, hash_fn=None, seal_fn=None, finish=None):
    """the Build*Message callbacks of core/consensus_test.go:28-128 over wire bytes: PREPREPARE of a round > 0 carries the
    RoundChangeCertificate the proposer collected, ROUND_CHANGE the node's latest prepared proposal + certificate.
    hash_fn / seal_fn / finish: the mock's (fh, a tagged seal, unsigned) or real ones (keccak, ECDSA seal, signed envelope)"""
    hash_fn = hash_fn or fh
    seal_fn = seal_fn or (lambda nd, hsh: SEAL + b" %d" % nd.index)
    finish = finish or (lambda nd, m: m.encode())

    def build(nd, type_, view):
        v = W.View(*view)
        if type_ == PP:
            raw = nd.proposal_raw if nd.proposal_raw is not None else proposals[nd.index]
            cert = CS.round_change_certificate_bytes(nd.rcc) if view[1] > 0 else None
            body = W.preprepare_body(W.Proposal(raw, view[1]), hash_fn(raw, view[1]), cert)
        elif type_ == PR:
            body = W.prepare_body(hash_fn(CS.raw_proposal_of(nd.accepted), view[1]))
        elif type_ == CM:
            hsh = hash_fn(CS.raw_proposal_of(nd.accepted), view[1])
            body = W.commit_body(hsh, seal_fn(nd, hsh))
        else:
            last = W.Proposal(*nd.latest_prepared) if nd.latest_prepared else None
            pc = CS.prepared_certificate_bytes(*nd.latest_pc) if nd.latest_pc else None
            if lie_about_prepared and nd.index in lie_about_prepared and last is not None:
                last = W.Proposal(b"another block", last.round)      # lastPreparedProposal that does not match the certificate
            body = W.round_change_body(last, pc)
        return finish(nd, W.IbftMessage(view=v, sender=nd.address, type=type_, payload=body))
    return build


def rc_verifier(ref, valid_proposals, proposer_of):
    def make(nd):
        def is_valid_proposal_hash(prop, hsh):
            return prop is not None and hsh == fh(prop[0], prop[1])

        def is_proposer(frm, height, round_):
            return frm == ref[0].nodes[proposer_of(height, round_)].address
        return (is_valid_proposal_hash, lambda hsh, seal: seal is not None and seal[1].startswith(SEAL), lambda wire: True,
                is_proposer, lambda raw: raw in valid_proposals)
    return make


def rc_cluster(n, mode, proposals, valid, proposer_of, lockstep=True, **kw):
    ref = [None]
    c = CS.Cluster(node_addresses(n), rc_build(proposals, **kw), rc_verifier(ref, valid, proposer_of), mode=mode,
                   lockstep=lockstep, build_proposal=lambda nd: proposals[nd.index])
    c.proposer = proposer_of
    ref[0] = c
    return c


def real_rc_cluster(n, mode, seed=21, gpu=None, lockstep=True):
    """the same cluster with secp256k1 keys: envelopes and seals signed by the oracle's signer (input generation), hashes =
    keccak(raw ‖ BE64(round)); the per-message Verifier is the oracle's (what an application's crypto Backend answers); with
    gpu: the device behind every node's batch calls"""
    import numpy as np
    from oracle import binding as B
    sks, addrs, _, _, verifier = _real_crypto(n, seed=seed)
    proposals = [b"block %d of a real cluster " % i + bytes(range(64)) for i in range(n)]

    def finish(nd, m):
        m.signature = B.sign(sks[nd.index], B.keccak256(m.payload_no_sig()))
        return m.encode()
    build = rc_build(proposals, hash_fn=B.proposal_hash, seal_fn=lambda nd, hsh: B.sign(sks[nd.index], hsh), finish=finish)
    ref = [None]

    def make(nd):
        f = list(verifier(ref)(nd))
        f[4] = lambda raw: raw in proposals                       # IsValidProposal
        return tuple(f)
    if gpu is not None:
        gpu.set_validators(1, np.frombuffer(b"".join(addrs), dtype=np.uint8).reshape(n, 20), np.ones(n, dtype=np.uint64))
    c = CS.Cluster(addrs, build, make, mode=mode, gpu=gpu, lockstep=lockstep, build_proposal=lambda nd: proposals[nd.index])
    ref[0] = c
    return c, proposals, sks, addrs


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("lockstep", [True, False])
def test_invalid_block_moves_everybody_to_round_1(mode, lockstep):
    """TestConsensus_InvalidBlock (core/consensus_test.go:260-394): node 0 proposes a block nobody accepts (IsValidProposal
    false), the round expires, every node multicasts ROUND_CHANGE(round 1) without a certificate, node 1 — proposer of round
    1 — collects the RoundChangeCertificate (handleRoundChangeMessage), proposes its own block with it, the others validate
    the PREPREPARE's certificate (validateProposal, core/ibft.go:683-788) and the round commits: every node inserts
    "proposal 2" and is on round 1."""
    n = 4
    proposals = [b"proposal 1", b"proposal 2", b"proposal 3", b"proposal 4"]
    c = rc_cluster(n, mode, proposals, {proposals[1]}, lambda h, r: r % n, lockstep=lockstep)
    try:
        assert c.run_height(1) == [None] * n
        assert [nd.state for nd in c.nodes] == ["prepare", "newRound", "newRound", "newRound"]   # node 0 accepted its own block
        inserted = c.tick()
        assert all(nd.round == 1 for nd in c.nodes)
        assert all(x is not None and x[0] == proposals[1] for x in inserted), inserted
        quorum = c.nodes[0].host.vm_quorum()
        for nd, x in zip(c.nodes, inserted):
            assert len(x[1]) >= quorum and all(sig.startswith(SEAL) for _, sig in x[1])
            assert nd.host.store_num(1, 1, RC) >= quorum and nd.host.store_num(1, 1, PP) == 1
        assert len(c.nodes[1].rcc) >= quorum
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_silent_proposer_then_round_1(mode):
    """core/byzantine_test.go:60-130 in its plainest form: the proposer of round 0 sends nothing; N = 6 so that the
    certificate is a real quorum (5 of 6 ROUND_CHANGE messages suffice; all six arrive)"""
    n = 6
    proposals = [b"block of %d" % i for i in range(n)]
    c = rc_cluster(n, mode, proposals, set(proposals), lambda h, r: (h + r) % n)
    try:
        c.nodes[c.proposer(3, 0)].silent = True
        assert c.run_height(3) == [None] * n and all(nd.state == "newRound" for nd in c.nodes)
        inserted = c.tick()
        p1 = c.proposer(3, 1)
        assert all(x is not None and x[0] == proposals[p1] for x in inserted)
        assert all(x[1] == inserted[0][1] and len(x[1]) == n for x in inserted)      # lockstep: everybody saw all six COMMITs
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_prepared_in_round_0_is_reproposed_in_round_1(mode):
    """The certificate-carrying half (f2 end to end): round 0 reaches PREPARE quorum everywhere — every node finalises a
    PreparedCertificate — but no COMMIT is sent; the timer fires; every ROUND_CHANGE(round 1) carries the node's
    lastPreparedProposal and its PC (PREPREPARE + PREPAREs of round 0, nested messages whose senders / hashes / rounds
    validPC checks, core/ibft.go:1162-1231); the round-1 proposer must re-propose THE SAME raw proposal
    (buildProposal, :1037-1088), and validateProposal accepts it only because hash(raw, maxRound = 0) equals the
    certificates' hash (:746-787).  Nobody inserts the round-1 proposer's own block."""
    n = 7
    proposals = [b"block of %d" % i for i in range(n)]
    c = rc_cluster(n, mode, proposals, set(proposals), lambda h, r: (h + r) % n)
    try:
        for nd in c.nodes:
            nd.withhold_commit = True
        assert c.run_height(1) == [None] * n
        assert all(nd.state == "commit" and nd.latest_pc is not None for nd in c.nodes)
        p0, p1 = c.proposer(1, 0), c.proposer(1, 1)
        for nd in c.nodes:
            nd.withhold_commit = False
        inserted = c.tick()
        assert all(x is not None and x[0] == proposals[p0] for x in inserted), [x and x[0] for x in inserted]
        assert proposals[p1] not in {x[0] for x in inserted}
        assert CS.raw_proposal_of(c.nodes[p1].accepted) == proposals[p0]
        # the accepted PREPREPARE of round 1 carries a certificate whose messages carry certificates (two levels of nesting)
        rcc = c.nodes[p1].rcc
        assert len(rcc) >= c.nodes[0].host.vm_quorum() and all(CS._first(CS._first(m, 8), 2) for m in rcc)
        if mode == "ingest":     # the certificates were judged when their carriers ARRIVED, from the batch backend's rows
            for nd in c.nodes:
                calls, rows, hits = nd.host.cert_stats()
                assert calls > 0 and rows > 0 and nd.host.fallbacks() == 0
            assert any(nd.host.rc_from_rows > 0 for nd in c.nodes) and any(nd.host.pp_from_rows > 0 for nd in c.nodes)
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_real_signatures_through_a_round_change(mode):
    """the re-proposal scenario with real keys: every nested message of every certificate carries a real envelope signature
    (IsValidValidator recovers it) and real keccak proposal hashes"""
    n = 4
    c, proposals, sks, addrs = real_rc_cluster(n, mode)
    try:
        for nd in c.nodes:
            nd.withhold_commit = True
        assert c.run_height(1) == [None] * n and all(nd.latest_pc is not None for nd in c.nodes)
        for nd in c.nodes:
            nd.withhold_commit = False
        inserted = c.tick()
        p0 = c.proposer(1, 0)
        assert all(x is not None and x[0] == proposals[p0] and len(x[1]) == n for x in inserted)
        from oracle import binding as B
        H1 = B.proposal_hash(proposals[p0], 1)
        assert inserted[0][1] == frozenset((addrs[i], B.sign(sks[i], H1)) for i in range(n))
    finally:
        c.close()


@pytest.mark.parametrize("mode", MODES)
def test_a_round_change_message_that_lies_about_its_prepared_proposal_is_left_out(mode):
    """proposalMatchesCertificate (core/ibft.go:516-551): two nodes claim a lastPreparedProposal their own certificate does
    not vouch for; handleRoundChangeMessage's isValidMsgFn rejects their messages (and prunes nothing: GetExtendedRCC does not
    prune), the certificate is built from the five honest ones — still a quorum of 7 — and the round commits the prepared
    block"""
    n = 7
    proposals = [b"block of %d" % i for i in range(n)]
    c = rc_cluster(n, mode, proposals, set(proposals), lambda h, r: (h + r) % n, lie_about_prepared={4, 5})
    try:
        for nd in c.nodes:
            nd.withhold_commit = True
        assert c.run_height(1) == [None] * n
        for nd in c.nodes:
            nd.withhold_commit = False
        inserted = c.tick()
        p0, p1 = c.proposer(1, 0), c.proposer(1, 1)
        assert all(x is not None and x[0] == proposals[p0] for x in inserted)
        liars = {c.nodes[4].address, c.nodes[5].address}
        senders = {dict((f, v) for f, wt, v in __import__("oracle.wire_parse", fromlist=["fields"]).fields(m) if wt == 2)[2]
                   for m in c.nodes[p1].rcc}
        assert not (senders & liars) and len(senders) == 5
    finally:
        c.close()


def test_round_change_modes_agree_node_by_node():
    """stock ≡ batch ≡ ingest through a round change: what each node inserted, the round-1 proposer's certificate, every
    node's stores"""
    n = 7
    proposals = [b"block of %d" % i for i in range(n)]
    per_mode = {}
    for mode in MODES:
        c = rc_cluster(n, mode, proposals, set(proposals), lambda h, r: (h + r) % n)
        try:
            for nd in c.nodes:
                nd.withhold_commit = True
            c.run_height(2)
            for nd in c.nodes:
                nd.withhold_commit = False
            inserted = c.tick()
            per_mode[mode] = (inserted, sorted(c.nodes[c.proposer(2, 1)].rcc),
                              [[nd.host.store_num(2, r, t) for r in (0, 1) for t in (PP, PR, CM, RC)] for nd in c.nodes])
        finally:
            c.close()
    assert per_mode["stock"] == per_mode["batch"] == per_mode["ingest"]
    assert all(x is not None for x in per_mode["stock"][0])
