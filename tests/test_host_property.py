"""TestProperty (core/rapid_test.go:208-399) over the host mirror: a random cluster (size, heights), and for every round
of every height a random number of Byzantine nodes ≤ maxFaulty — the first `silent` of them drop everything they would
multicast, the rest send the bad proposal / hash / seal (badRoundMessage, core/helpers_test.go:28-37) — with rounds
generated until the round-robin proposer is an honest node (generatePropertyTestEvent, :150-203).

Asserted per height, as the reference does (:362-397): after the generated rounds every node that was honest in the last
round inserted exactly the correct proposal, no Byzantine node of that round inserted anything, at least a quorum
inserted; and before the last round nobody inserted.  On top of the reference: the three ways the mirror runs the walks
(per-message Verifier, one batch call per walk, micro-batched wire ingest + quorum index) agree node by node — inserted
block AND the seal set handed to InsertProposal — and a rejected Byzantine seal never appears in that set.

The reference draws 4..30 nodes and 5..20 heights under rapid's shrinker; the heights are cut (4..30 nodes as there, 1..5
heights) so that the CPU suite stays inside its few minutes; the round timer is cluster_sim's tick()."""
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import cluster_sim as CS
from oracle import wire as W
from test_host_cluster import MODES, SEAL, fh, node_addresses, rc_build

PP, PR, CM, RC = W.PREPREPARE, W.PREPARE, W.COMMIT, W.ROUND_CHANGE
GOOD, BAD = b"valid block", b"invalid block"          # correctRoundMessage / badRoundMessage (core/helpers_test.go:18-37)
BAD_SEAL = b"invalid seal"


def max_faulty(n):                                     # core/helpers_test.go:243-245
    return (n - 1) // 3


def quorum(n):                                         # core/consensus_test.go:118-125 (how many nodes the test awaits)
    return n if max_faulty(n) == 0 else -(-2 * n // 3)


@st.composite
def setups(draw):
    """generatePropertyTestEvent: {nodes, events[height] = [(silent, bad) per round]}, heights numbered from 1"""
    n = draw(st.integers(4, 30))
    heights = draw(st.integers(1, 5))
    events = []
    for h in range(1, heights + 1):
        rounds, r = [], 0
        while True:
            byz = draw(st.integers(0, max_faulty(n)))
            silent = draw(st.integers(0, byz))
            rounds.append((silent, byz - silent))
            if (h + r) % n >= byz or r >= 6:
                # r ≥ 6: end the draw with an honest round (rapid's generator is unbounded; ours is not)
                if (h + r) % n < byz:
                    rounds[-1] = (0, 0)
                break
            r += 1
        events.append(rounds)
    return n, events


def event_of(events, nd):
    rounds = events[nd.height - 1]
    return rounds[min(nd.round, len(rounds) - 1)]      # propertyTestSetup.getEvent (:128-146)


def property_cluster(n, events, mode):
    def is_bad(nd):
        s, b = event_of(events, nd)
        return nd.index < s + b

    def is_silent(nd):
        return nd.index < event_of(events, nd)[0]

    def expected(nd):                                  # propertyTestEvent.getMessage (:84-93)
        return BAD if is_bad(nd) else GOOD

    build = rc_build([None] * n, seal_fn=lambda nd, hsh: BAD_SEAL if is_bad(nd) else SEAL + b" %d" % nd.index)
    ref = [None]

    def verifier(nd):
        def is_valid_proposal_hash(prop, hsh):         # isValidProposalHashFn (:277-283)
            return prop is not None and prop[0] == expected(nd) and hsh == fh(prop[0], prop[1])

        def is_valid_committed_seal(hsh, seal):        # the reference's mock answers true; an honest Backend does not
            return seal is not None and (seal[1] == BAD_SEAL if is_bad(nd) else seal[1].startswith(SEAL))

        def is_proposer(frm, height, round_):          # isProposerFn (:263-268)
            return frm == ref[0].nodes[(height + round_) % n].address
        return (is_valid_proposal_hash, is_valid_committed_seal, lambda wire: True, is_proposer,
                lambda raw: raw == expected(nd))       # isValidProposalFn (:271-275)

    c = CS.Cluster(node_addresses(n), build, verifier, mode=mode, build_proposal=expected)
    c.censor = lambda nd, type_, view: is_silent(nd)   # commonTransportCallback (:222-237)
    ref[0] = c
    return c


def run_setup(n, events, mode):
    """→ per height, per node: what InsertProposal was handed (or None); asserts the reference's per-height properties"""
    c = property_cluster(n, events, mode)
    out = []
    try:
        for h, rounds in enumerate(events, start=1):
            inserted = c.run_height(h)
            for r in range(1, len(rounds)):
                assert inserted == [None] * n, (h, r, inserted)          # the proposer of every earlier round was Byzantine
                inserted = c.tick()
            bad_nodes = sum(rounds[-1])
            for i, x in enumerate(inserted):
                if i < bad_nodes:
                    assert x is None, (h, i, x)                         # :389-392
                else:
                    assert x is not None and x[0] == GOOD, (h, i, x)    # :381-388
                    assert len(x[1]) >= c.nodes[i].host.vm_quorum()          # the ValidatorManager's quorum of seals
                    assert all(sig.startswith(SEAL) for _, sig in x[1])  # no rejected seal reaches InsertProposal
                    assert c.nodes[i].round == len(rounds) - 1
            assert n - bad_nodes >= quorum(n)                           # :395
            out.append(inserted)
    finally:
        c.close()
    return out


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(setups())
def test_property_any_byzantine_schedule_commits_the_correct_block_in_every_mode(setup):
    n, events = setup
    per_mode = {mode: run_setup(n, events, mode) for mode in MODES}
    assert per_mode["stock"] == per_mode["batch"] == per_mode["ingest"]


@pytest.mark.parametrize("mode", MODES)
def test_property_worst_case_schedule(mode):
    """the deepest draw by hand: N = 10 (f = 3), heights whose first rounds all have a Byzantine proposer — silent, then
    lying, then silent again — before an honest one"""
    n = 10
    events = [[(3, 0), (1, 2), (0, 0)],               # height 1: proposers 1, 2 Byzantine (index < 3); round 2's is 3
              [(0, 3), (2, 1), (3, 0), (0, 3), (1, 1)],   # height 2: proposers 2 (bad), 3 ≥ 3 …
              [(0, 0)]]
    # trim each height at its first honest proposer, as the generator would
    trimmed = []
    for h, rounds in enumerate(events, start=1):
        keep = []
        for r, (s, b) in enumerate(rounds):
            keep.append((s, b))
            if (h + r) % n >= s + b:
                break
        trimmed.append(keep)
    run_setup(n, trimmed, mode)
