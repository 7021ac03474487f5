"""The mirror's quorum decision taken from the batch backend (ibft_host_use_device_quorum → BatchVerifier::QuorumOfSenders →
ibft_tally_prepare / ibft_tally on a device; here the backend that loops on the host stands in): handlePrepare /
handleCommit decide exactly like hasQuorumByMsgType (/root/reference/core/ibft.go:1273-1284, HasPrepareQuorum
core/validator_manager.go:99-127) in object mode and in row mode; a backend that cannot answer leaves the decision to the
mirror; a backend that answers wrongly is counted.  The device behind the same switch: tests/test_gpu_host.py."""
import pytest

import go_ibft_amd.hostlib as H
from oracle import wire as W
from test_host_roundchange import World, fake_hash, PP, PR, CM


def _world(n, silent=(), proposer_prepares=False, powers=None):
    w = World(n, 3)
    raw = b"block"
    hsh = fake_hash(raw, 0)
    proposal = W.IbftMessage(view=W.View(1, 0), sender=w.proposer(1, 0), type=PP,
                             payload=W.preprepare_body(W.Proposal(raw, 0), hsh, None))
    prepares = [W.IbftMessage(view=W.View(1, 0), sender=a, type=PR, payload=W.prepare_body(hsh))
                for i, a in enumerate(w.addrs) if i not in silent and (a != proposal.sender or proposer_prepares)]
    commits = [W.IbftMessage(view=W.View(1, 0), sender=a, type=CM, payload=W.commit_body(hsh, b"seal-" + a))
               for i, a in enumerate(w.addrs) if i not in silent]
    return w, proposal, prepares, commits


def _host(w, proposal, fail_mask, device_quorum, powers=None, index=True):
    h = H.Host()
    assert h.vm_init(powers or {a: 1 for a in w.addrs})
    h.set_verifier(**w.verifier())
    h.set_state(1, 0, proposal.encode())
    if index:
        h.enable_quorum_index()
    h.use_loop_batch(fail_mask)
    h.use_batch(True)
    h.use_device_quorum(device_quorum)
    return h


CASES = [
    dict(n=4, silent=()),                        # everybody: quorum
    dict(n=4, silent=(2, 3)),                    # proposer + one PREPARE of four: none
    dict(n=7, silent=(5, 6)),                    # exactly ⌊2·7/3⌋+1 = 5 with the proposer's seat
    dict(n=7, silent=(4, 5, 6)),                 # one short
    dict(n=7, silent=(), proposer_prepares=True),  # the proposer's own PREPARE among them: void (:114-121)
    dict(n=6, silent=(0,), powers=[5, 1, 1, 1, 1, 1]),   # weighted: the heavy validator is silent
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("rows", [False, True])
def test_device_quorum_decides_like_the_mirror(case, rows):
    c = dict(CASES[case])
    pw = c.pop("powers", None)
    w, proposal, prepares, commits = _world(**c)
    powers = {a: pw[i] for i, a in enumerate(w.addrs)} if pw else None
    ref = _host(w, proposal, 0, False, powers)
    dev = _host(w, proposal, 0, True, powers)
    for h in (ref, dev):
        h.use_rows(rows)
        wires = [m.encode() for m in prepares + commits]
        if rows:
            h.ingest_wire(wires)                 # judged from their bytes: kept as rows where the backend vouches
            assert h.rows_kept > 0
        else:
            for x in wires:
                h.store_add(x)
    a, b = ref.handle_prepare(1, 0), dev.handle_prepare(1, 0)
    assert a[0] == b[0] and sorted(a[1]) == sorted(b[1])
    qa, sa = ref.handle_commit(1, 0)
    qb, sb = dev.handle_commit(1, 0)
    assert qa == qb and sorted(sa) == sorted(sb)
    calls, mism = dev.device_quorum_stats()
    assert calls == 2 and mism == 0
    assert ref.device_quorum_stats() == (0, 0)
    if c.get("proposer_prepares"):
        assert not a[0]
    for h in (ref, dev):
        h.close()


def test_backend_without_an_answer_and_backend_with_a_wrong_one():
    w, proposal, prepares, commits = _world(7, silent=(5, 6))
    ref = _host(w, proposal, 0, False)
    mute = _host(w, proposal, 32, True)          # "device unavailable" for the quorum call: the mirror decides
    liar = _host(w, proposal, 64, True)          # answers the opposite: taken (the device is the authority), counted
    for h in (ref, mute, liar):
        for m in prepares + commits:
            h.store_add(m.encode())
    assert ref.handle_prepare(1, 0)[0] and mute.handle_prepare(1, 0)[0]
    assert mute.device_quorum_stats() == (0, 0)
    assert not liar.handle_prepare(1, 0)[0]
    assert liar.device_quorum_stats() == (1, 1)
    for h in (ref, mute, liar):
        h.close()


def test_no_proposal_message_is_decided_without_asking():
    """HasPrepareQuorum with proposalMessage == nil is false before any set is built (validator_manager.go:101-110)"""
    w, proposal, prepares, _ = _world(4)
    h = H.Host()
    assert h.vm_init({a: 1 for a in w.addrs})
    h.set_verifier(**w.verifier())
    h.set_state(1, 0, None)
    h.use_loop_batch(0)
    h.use_batch(True)
    h.use_device_quorum(True)
    for m in prepares:
        h.store_add(m.encode())
    assert not h.handle_prepare(1, 0)[0]
    assert h.device_quorum_stats() == (0, 0)
    h.close()


@pytest.mark.parametrize("n,threshold,declined", [(4, 8, True), (6, 8, True), (7, 7, True), (30, 8, False), (4, 0, False)])
@pytest.mark.parametrize("rows", [False, True])
def test_min_device_rows_declines_small_batches_and_the_stock_closures_decide(n, threshold, declined, rows):
    """SURVEY §5 "min batch for GPU" (ibft_host_set_min_device_rows / IBFT_MIN_DEVICE_ROWS): a batch below the threshold is
    answered "not offered" by the backend — what it answers when the device is unavailable — and handlePrepare / handleCommit
    run the per-message closures of /root/reference/core/ibft.go:856-862, 932-944 instead: same prepared set, same seals, same
    quorum decision as a mirror without the knob, at the validator counts the reference itself tests (4: core/consensus_test.go:139,
    6: core/byzantine_test.go:21, ≤ 30: core/rapid_test.go:156).  At or above the threshold the backend answers as before."""
    w, proposal, prepares, commits = _world(n, silent=(n - 1,))
    plain = _host(w, proposal, 0, True)
    knob = _host(w, proposal, 0, True)
    knob.set_min_device_rows(threshold)
    for h in (plain, knob):
        h.use_rows(rows)
        wires = [m.encode() for m in prepares + commits]
        if rows:
            h.ingest_wire(wires)
        else:
            for x in wires:
                h.store_add(x)
    before = knob.loop_batch_calls()
    a, b = plain.handle_prepare(1, 0), knob.handle_prepare(1, 0)
    assert a[0] == b[0] and sorted(a[1]) == sorted(b[1])
    qa, sa = plain.handle_commit(1, 0)
    qb, sb = knob.handle_commit(1, 0)
    assert qa == qb and sorted(sa) == sorted(sb) and qa == (n - 1 >= 2 * n // 3 + 1)
    assert plain.declined_batches() == 0
    if declined:
        assert knob.declined_batches() > 0
        if not rows:                                  # object mode: the two walks went to the callbacks, not to the backend
            assert knob.loop_batch_calls() == before and knob.fallbacks() >= 2
    else:
        assert knob.declined_batches() == 0 and knob.fallbacks() == 0
    for h in (plain, knob):
        h.close()
