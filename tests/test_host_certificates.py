"""Certificate checks of the host mirror (core/ibft.go validPC :1162-1231, validateProposal
:683-788, proposalMatchesCertificate :516-551) replayed from the reference's tables
TestIBFT_ValidPC (core/ibft_test.go:1510-2015) and TestIBFT_ValidateProposal (:2017-2797).
Many reference cases never initialise the ValidatorManager, so they return false at the quorum
check whatever rule they name; each rule is therefore ALSO replayed with an initialised manager so
that it is the rule under test that decides."""
import pytest

import go_ibft_amd.hostlib as H
from oracle import wire as W

PP, PR, CM, RC = W.PREPREPARE, W.PREPARE, W.COMMIT, W.ROUND_CHANGE
HASH = b"proposal hash"


def nodes(n):
    return [f"node {i}".encode() for i in range(n)]


def msg(t, frm=b"", view=(0, 0), hsh=None, payload="auto"):
    if payload == "auto":
        payload = {PP: W.preprepare_body(None, hsh or b"", None), PR: W.prepare_body(hsh or b""),
                   CM: W.commit_body(hsh or b"", b""), RC: None}[t]
    return W.IbftMessage(view=W.View(*view) if view is not None else None, sender=frm, type=t, payload=payload)


def pc_wire(proposal, prepares):
    return W.prepared_certificate(proposal, prepares)


def host(quorum_nodes=4, init=True, **verifier):
    h = H.Host()
    if init:
        assert h.vm_init({a: 1 for a in nodes(quorum_nodes)})
    h.set_verifier(**verifier)
    return h


def valid_parts(rlimit=1, sender=b"unique node"):
    proposal = msg(PP, sender, (0, rlimit - 1), HASH)
    prepares = [msg(PR, a, (0, rlimit - 1), HASH) for a in nodes(3)]
    return proposal, prepares


# ------------------------------------------------------------------ TestIBFT_ValidPC
def test_valid_pc_no_certificate_and_mismatch():
    h = host(init=False)
    assert h.valid_pc(None, 0, 0)                                                   # "no certificate"
    assert not h.valid_pc(pc_wire(None, []), 0, 0)                                   # proposal nil
    assert not h.valid_pc(pc_wire(W.IbftMessage(), []), 0, 0)                        # prepares nil


@pytest.mark.parametrize("init", [False, True])
def test_valid_pc_rules(init):
    sender = b"unique node"
    is_proposer = lambda who, hh, rr: who == sender
    h = host(init=init, is_proposer=is_proposer)
    ok_p, ok_pr = valid_parts()
    # completely valid PC (needs the manager: ibft_test.go:1973-2015 calls validatorManager.Init)
    assert h.valid_pc(pc_wire(ok_p, ok_pr), 1, 0) == init
    # no Quorum PP + P messages
    assert not h.valid_pc(pc_wire(ok_p, ok_pr[:1]), 1, 0)
    # invalid proposal message type
    assert not h.valid_pc(pc_wire(msg(PR, sender, (0, 0), HASH), ok_pr), 1, 0)
    # invalid prepare message type
    bad = list(ok_pr)
    bad[0] = msg(RC, bad[0].sender, (0, 0))
    assert not h.valid_pc(pc_wire(ok_p, bad), 1, 0)
    # non unique senders
    assert not h.valid_pc(pc_wire(msg(PP, b"node 0", (0, 0), HASH), [msg(PR, b"node 0", (0, 0), HASH)] * 3), 1, 0)
    dup = [ok_pr[0], ok_pr[0], ok_pr[1], ok_pr[2]]
    assert not h.valid_pc(pc_wire(ok_p, dup), 1, 0)
    # differing proposal hashes
    assert not h.valid_pc(pc_wire(msg(PP, sender, (0, 0), b"proposal hash 1"),
                                  [msg(PR, a, (0, 0), b"proposal hash 2") for a in nodes(3)]), 1, 0)
    # rounds not lower than rLimit
    p2, pr2 = valid_parts(rlimit=3)      # messages at round 2
    assert not h.valid_pc(pc_wire(p2, pr2), 2, 0) and not h.valid_pc(pc_wire(p2, pr2), 1, 0)
    assert h.valid_pc(pc_wire(p2, pr2), 3, 0) == init
    # heights are not the same / not the certificate's height
    mixed = [ok_pr[0], msg(PR, b"node 1", (5, 0), HASH), ok_pr[2]]
    assert not h.valid_pc(pc_wire(ok_p, mixed), 1, 0)
    assert not h.valid_pc(pc_wire(ok_p, ok_pr), 1, 7)
    # rounds are not the same
    mixed = [ok_pr[0], msg(PR, b"node 1", (0, 1), HASH), ok_pr[2]]
    assert not h.valid_pc(pc_wire(ok_p, mixed), 3, 0)


def test_valid_pc_proposer_and_sender_rules():
    sender = b"unique node"
    ok_p, ok_pr = valid_parts()
    cert = pc_wire(ok_p, ok_pr)
    # proposal not from proposer
    assert not host(is_proposer=lambda who, hh, rr: who != sender).valid_pc(cert, 1, 0)
    # prepare is from an invalid sender / proposal is from an invalid sender
    assert not host(is_proposer=lambda who, hh, rr: who == sender,
                    is_valid_validator=lambda w: b"node 1" not in w).valid_pc(cert, 1, 0)
    assert not host(is_proposer=lambda who, hh, rr: who == sender,
                    is_valid_validator=lambda w: sender not in w).valid_pc(cert, 1, 0)
    # prepare from proposer (IsProposer true for everyone)
    assert not host(is_proposer=lambda who, hh, rr: True).valid_pc(cert, 1, 0)
    # completely valid
    assert host(is_proposer=lambda who, hh, rr: who == sender).valid_pc(cert, 1, 0)


# ------------------------------------------------------------------ proposalMatchesCertificate
def test_proposal_matches_certificate():
    ok_p, ok_pr = valid_parts()
    h = host(is_valid_proposal_hash=lambda prop, hsh: hsh == HASH and prop == (b"block", 1))
    prop = W.Proposal(b"block", 1).encode()
    assert h.proposal_matches_certificate(None, None)                  # both nil
    assert not h.proposal_matches_certificate(prop, None)              # proposal without certificate
    assert h.proposal_matches_certificate(prop, pc_wire(ok_p, ok_pr))
    bad = [ok_pr[0], msg(PR, b"node 1", (0, 0), b"other"), ok_pr[2]]
    assert not h.proposal_matches_certificate(prop, pc_wire(ok_p, bad))
    assert not h.proposal_matches_certificate(W.Proposal(b"block", 2).encode(), pc_wire(ok_p, ok_pr))


# ------------------------------------------------------------------ TestIBFT_ValidateProposal
def proposal_msg(view, frm=b"", raw=b"", prop_round=None, hsh=b"", rcs=None, with_cert=True):
    cert = W.round_change_certificate(rcs) if (with_cert and rcs is not None) else None
    body = W.preprepare_body(W.Proposal(raw, view[1] if prop_round is None else prop_round), hsh, cert)
    return W.IbftMessage(view=W.View(*view), sender=frm, type=PP, payload=body)


def rc_msg(frm, view, pc=None, t=RC):
    return W.IbftMessage(view=W.View(*view), sender=frm, type=t, payload=W.round_change_body(None, pc))


def test_validate_proposal_common_rules():
    view = (0, 0)
    p = proposal_msg(view).encode()
    # proposer is not valid / block is not valid / proposal hash is not valid
    assert not host(is_proposer=lambda *a: False).validate_proposal(p, *view)
    assert not host(is_proposer=lambda *a: True, is_valid_proposal=lambda raw: False).validate_proposal(p, *view)
    assert not host(is_proposer=lambda *a: True, is_valid_proposal_hash=lambda pr, hs: False).validate_proposal(p, *view)
    # certificate is not present
    assert not host(is_proposer=lambda *a: True).validate_proposal(proposal_msg(view, with_cert=False).encode(), *view)
    # round is not correct (proposal.Round != view.Round)
    assert not host(is_proposer=lambda *a: True).validate_proposal(
        proposal_msg(view, prop_round=5, rcs=[rc_msg(a, view) for a in nodes(4)]).encode(), *view)
    # validateProposal0: proposal must be for round 0, this node must not be the proposer
    h = host(is_proposer=lambda who, hh, rr: who == b"proposer")
    h.set_id(b"me")
    assert h.validate_proposal0(proposal_msg((0, 0), b"proposer").encode(), 0, 0)
    assert not h.validate_proposal0(proposal_msg((0, 1), b"proposer", prop_round=0).encode(), 0, 0)
    h.set_id(b"proposer")
    assert not h.validate_proposal0(proposal_msg((0, 0), b"proposer").encode(), 0, 0)


def test_validate_proposal_rcc_rules():
    me, prop_id, view = b"node id", b"unique node", (0, 1)
    is_proposer = lambda who, hh, rr: who == prop_id
    good_rcs = [rc_msg(a, view) for a in nodes(4)]

    def run(rcs, init=True, **kw):
        h = host(init=init, is_proposer=kw.pop("is_proposer", is_proposer), **kw)
        h.set_id(kw.get("_id", me))
        return h.validate_proposal(proposal_msg(view, prop_id, rcs=rcs).encode(), *view)
    assert run(good_rcs)                                               # valid: quorum of plain RC messages
    assert not run(good_rcs, init=False)                               # manager not initialised -> no quorum
    assert not run([rc_msg(b"non unique node id", view)] * 4)          # non unique senders
    assert not run(good_rcs[:2])                                       # < quorum RC messages
    # current node should not be the proposer
    h = host(is_proposer=lambda who, hh, rr: True)
    h.set_id(me)
    assert not h.validate_proposal(proposal_msg(view, prop_id, rcs=good_rcs).encode(), *view)
    # sender is not the correct proposer
    assert not run(good_rcs, is_proposer=lambda who, hh, rr: False)
    # a message in the RCC is not a ROUND-CHANGE message / wrong height / wrong round / non-validator
    assert not run(good_rcs[:3] + [rc_msg(b"node 3", view, t=PR)])
    assert not run(good_rcs[:3] + [rc_msg(b"node 3", (9, 1))])
    assert not run(good_rcs[:3] + [rc_msg(b"node 3", (0, 2))])
    assert not run(good_rcs, is_valid_validator=lambda w: b"node 2" not in w)


def test_validate_proposal_max_round_hash_rule():
    """ibft_test.go:2663-2796: the hash of (RawProposal, maxRound) must equal the hash carried by the
    highest-round valid PC.  The reference case never initialises the manager; here it is."""
    raw, round_ = b"raw proposal", 2
    proposers = [b"proposer 0", b"proposer 1", b"proposer 2"]
    hash_fn = lambda r, rd: r + b"_" + str(rd).encode()
    vals = [f"node{i}".encode() for i in range(4)]

    def build(pc_hash):
        prev_prop = W.IbftMessage(view=W.View(0, 0), sender=proposers[0], type=PP,
                                  payload=W.preprepare_body(W.Proposal(raw, 0), pc_hash, None))
        prev_prep = [W.IbftMessage(view=W.View(0, 0), sender=a, type=PR, payload=W.prepare_body(pc_hash)) for a in vals]
        pc = W.prepared_certificate(prev_prop, prev_prep)
        rcs = [rc_msg(a, (0, round_), pc) for a in vals]
        return proposal_msg((0, round_), proposers[2], raw=raw, hsh=hash_fn(raw, round_), rcs=rcs).encode()

    def mk(init):
        h = H.Host()
        if init:
            assert h.vm_init({a: 1 for a in vals})
        h.set_id(b"node id")
        h.set_verifier(is_proposer=lambda who, hh, rr: who == proposers[rr],
                       is_valid_validator=lambda w: b"non validator" not in w,
                       is_valid_proposal_hash=lambda prop, hsh: prop is not None and hsh == hash_fn(prop[0], prop[1]))
        return h
    wrong = build(hash_fn(raw, round_))      # PC built at round 0 but carrying the round-2 hash
    right = build(hash_fn(raw, 0))
    assert not mk(False).validate_proposal(wrong, 0, round_)      # the reference's (uninitialised) outcome
    assert not mk(True).validate_proposal(wrong, 0, round_)       # ... and for the reason the case names
    assert mk(True).validate_proposal(right, 0, round_)


# ------------------------------------------------------------------ handlePrePrepare (core/ibft.go:792-813)
RAW_BLOCK = b"valid ethereum block"


def _filled_rc_messages(quorum, proposal, proposal_hash):
    """generateFilledRCMessages (core/ibft_test.go:158-216)"""
    prepares = [W.IbftMessage(view=W.View(0, 1), sender=b"node %d" % (k + 1), type=PR, payload=W.prepare_body(proposal_hash))
                for k in range(quorum - 1)]
    pm = W.IbftMessage(view=W.View(0, 1), sender=b"unique node", type=PP, payload=W.preprepare_body(proposal, proposal_hash, None))
    pc = W.prepared_certificate(pm, prepares)
    return [W.IbftMessage(view=W.View(0, 1), sender=b"node %d" % k, type=RC, payload=W.round_change_body(proposal, pc))
            for k in range(quorum)]


@pytest.mark.parametrize("mode", ["stock", "batch", "arrival"])
def test_handle_preprepare_replays_run_new_round_validator(mode):
    """The handlePrePrepare slice of TestRunNewRound_Validator_Zero (core/ibft_test.go:603-696) and
    TestRunNewRound_Validator_NonZero (:701-866): a non-proposer accepts the proposer's PREPREPARE of round 0, and of round 1 with a
    RoundChangeCertificate of plain ROUND-CHANGE messages ("new block") or of ROUND-CHANGE messages carrying the previously prepared
    proposal and its certificate ("old block").  stock = per-message Verifier; batch = certificate batches gathered by the walk;
    arrival = the messages come in through IngestWire and their certificates are judged then."""
    def mk(cnt, view):
        h = H.Host()
        assert h.vm_init({b"node %d" % k: 1 for k in range(cnt)})
        h.set_id(b"non proposer")
        h.set_verifier(is_proposer=lambda who, hh, rr: who == b"proposer")
        h.set_state(0, view[1], None)
        if mode != "stock":
            h.use_loop_batch(0)
            h.use_batch(True)
        return h

    def deliver(h, wire):
        if mode == "arrival":
            assert h.ingest_wire([wire])[0][0] in (1, 2)
        else:
            assert h.store_add(wire) == 0
    # round 0 (Validator_Zero): Proposal present, no hash, no certificate
    h = mk(1, (0, 0))
    p0 = W.IbftMessage(view=W.View(0, 0), sender=b"proposer", type=PP, payload=W.preprepare_body(W.Proposal(RAW_BLOCK, 0), b"", None))
    deliver(h, p0.encode())
    assert h.handle_preprepare(0, 0) == p0.encode()
    h.close()
    # round 1 (Validator_NonZero), both certificates
    quorum = 4
    proposal = W.Proposal(RAW_BLOCK, 1)
    plain = [W.IbftMessage(view=W.View(0, 1), sender=b"node %d" % k, type=RC, payload=W.round_change_body(None, None)) for k in range(quorum)]
    for rcs in (plain, _filled_rc_messages(quorum, proposal, b"proposal hash")):
        h = mk(quorum, (0, 1))
        p1 = W.IbftMessage(view=W.View(0, 1), sender=b"proposer", type=PP,
                           payload=W.preprepare_body(proposal, b"proposal hash", W.round_change_certificate(rcs)))
        deliver(h, p1.encode())
        assert h.handle_preprepare(0, 1) == p1.encode()
        if mode == "arrival":
            assert h.cert_stats()[0] == 1 and h.loop_batch_cert_calls() == 1
        # a PREPREPARE from somebody who is not the proposer is pruned and nothing is accepted
        h2 = mk(quorum, (0, 1))
        bad = W.IbftMessage(view=W.View(0, 1), sender=b"node 2", type=PP,
                            payload=W.preprepare_body(proposal, b"proposal hash", W.round_change_certificate(rcs)))
        deliver(h2, bad.encode())
        assert h2.handle_preprepare(0, 1) is None and h2.store_num(0, 1, PP) == 0
        h.close(); h2.close()


@pytest.mark.parametrize("mode", ["stock", "batch", "arrival"])
def test_future_proposal_and_future_rcc_replays(mode):
    """The certificate slices of TestIBFT_FutureProposal (core/ibft_test.go:1328-1506: watchForFutureProposal → handlePrePrepare at a
    future round, new block / old block) and of TestIBFT_WatchForFutureRCC (:2801-2896: watchForRoundChangeCertificates →
    handleRoundChangeMessage over ROUND-CHANGE messages of round 10 that carry filled certificates)."""
    quorum, node_id = 4, b"node ID"
    good_hash = b"proposal hash"

    def mk(**verifier):
        h = H.Host()
        assert h.vm_init({b"node %d" % k: 1 for k in range(quorum)})
        h.set_id(node_id)
        h.set_verifier(**verifier)
        h.set_state(0, 0, None)
        if mode != "stock":
            h.use_loop_batch(0)
            h.use_batch(True)
        return h

    def deliver(h, wires):
        if mode == "arrival":
            assert all(x in (1, 2) for x in h.ingest_wire(wires)[0])
        else:
            for wv in wires:
                assert h.store_add(wv) == 0
    # --- TestIBFT_FutureProposal
    fp = dict(is_proposer=lambda who, hh, rr: who != node_id,
              is_valid_proposal_hash=lambda p, hsh: p is not None and p[0] == RAW_BLOCK and hsh == good_hash)
    empty_rcs = [W.IbftMessage(view=W.View(0, 1), sender=b"node %d" % k, type=RC, payload=W.round_change_body(None, None)) for k in range(quorum)]
    filled = _filled_rc_messages(quorum, W.Proposal(RAW_BLOCK, 0), good_hash)
    for m in filled:
        m.view = W.View(0, 2)   # setRoundForMessages(messages, 2)
    for view, rcs in (((0, 1), empty_rcs), ((0, 2), filled)):
        h = mk(**fp)
        p = W.IbftMessage(view=W.View(*view), sender=b"proposer", type=PP,
                          payload=W.preprepare_body(W.Proposal(RAW_BLOCK, view[1]), good_hash, W.round_change_certificate(rcs)))
        deliver(h, [p.encode()])
        assert h.handle_preprepare(*view) == p.encode()          # the future proposal is signalled for its round
        assert h.handle_preprepare(view[0], view[1] + 1) is None   # nothing for another round
        h.close()
    # --- TestIBFT_WatchForFutureRCC
    rcc_round = 10
    rcs = _filled_rc_messages(quorum, W.Proposal(RAW_BLOCK, 0), good_hash)
    for m in rcs:
        m.view = W.View(0, rcc_round)
    h = mk(is_proposer=lambda who, hh, rr: who == b"unique node")
    deliver(h, [m.encode() for m in rcs])
    got = h.handle_round_change(0, rcc_round)
    assert sorted(got) == sorted(m.encode() for m in rcs)
    if mode == "arrival":
        assert h.cert_stats()[2] == quorum * quorum and h.last_cert_batch() == (0, 0)   # every nested verdict came from the tables
    h.close()
