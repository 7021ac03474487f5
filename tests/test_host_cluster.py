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

PP, PR, CM = W.PREPREPARE, W.PREPARE, W.COMMIT
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
