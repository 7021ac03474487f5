"""Host-side semantics (store, helpers, quorum, hot-path callers) replayed from the
reference's own unit tables, against the C++ host mirror (libibft_host.so) and the
pure-Python restatement (oracle/semantics.py).  CPU only: the Verifier is the mock
backend (callbacks), exactly like /root/reference/core/mock_test.go:105-151."""
import json
import os
import random

import pytest

import go_ibft_amd.hostlib as H
from oracle import semantics as S
from oracle import wire as W

HERE = os.path.dirname(os.path.abspath(__file__))
PP, PR, CM, RC = W.PREPREPARE, W.PREPARE, W.COMMIT, W.ROUND_CHANGE


def gen_random_messages(count, view, *types):
    """generateRandomMessages, /root/reference/messages/messages_test.go:14-57."""
    out = []
    for i in range(count):
        for t in types:
            body = {PP: W.preprepare_body(None, b"", None), PR: W.prepare_body(b""),
                    CM: W.commit_body(b"", b"")}.get(t)
            out.append(W.IbftMessage(view=W.View(*view), sender=str(i).encode(), type=t, payload=body))
    return out


# ---------------------------------------------------------------- wire format
def test_wire_vectors_python_and_cpp():
    vecs = json.load(open(os.path.join(HERE, "golden", "wire_vectors.json")))
    for v in vecs:
        w, ns = bytes.fromhex(v["wire"]), bytes.fromhex(v["payload_no_sig"])
        assert H.reencode(w) == w, v["name"]
        assert H.payload_no_sig(w) == ns, v["name"]
    # the Python restatement (oracle/wire.py) produces the same bytes for the synthetic shapes
    addr, h32 = bytes(range(1, 21)), bytes(range(32, 64))
    seal, sig = bytes(range(100, 165)), bytes(range(7, 72))
    by = {v["name"]: v for v in vecs}
    m = W.IbftMessage(view=W.View(1, 0), sender=addr, signature=sig, type=PR, payload=W.prepare_body(h32))
    assert m.encode().hex() == by["prepare_h1_r0"]["wire"] and len(m.payload_no_sig()) == 64
    m = W.IbftMessage(view=W.View(0, 0), sender=addr, signature=sig, type=PR, payload=W.prepare_body(h32))
    assert m.encode().hex() == by["prepare_h0_r0_empty_view"]["wire"]
    m = W.IbftMessage(view=W.View(1, 0), sender=addr, signature=sig, type=CM, payload=W.commit_body(h32, seal))
    assert m.payload_no_sig().hex() == by["commit_h1_r0"]["payload_no_sig"] and len(m.payload_no_sig()) == 131
    m = W.IbftMessage(view=W.View(2**63 + 5, 2**32), sender=addr, signature=sig, type=CM,
                      payload=W.commit_body(h32, seal))
    assert m.encode().hex() == by["commit_big_height"]["wire"]
    m = W.IbftMessage(view=None, sender=addr, type=CM, payload=None)
    assert m.encode().hex() == by["commit_type_nil_payload_nil_view"]["wire"]


def test_decoder_rejects_malformed_and_keeps_unknown_fields():
    m = W.IbftMessage(view=W.View(3, 1), sender=b"x" * 20, signature=b"s" * 65, type=PR, payload=W.prepare_body(b"h" * 32))
    w = m.encode()
    assert H.reencode(w[:-3]) is None                      # truncated length-delimited field
    assert H.reencode(b"\x0a\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01") is None   # over-long varint
    unknown = b"\x78\x05" + b"\x82\x01\x02hi"            # field 15 varint, field 16 bytes
    assert H.reencode(w + unknown) == w + unknown          # preserved, emitted last (protobuf-go)
    assert H.payload_no_sig(w + unknown) == m.payload_no_sig() + unknown
    # non-canonical input (fields out of order) is re-marshalled canonically, as Go would
    shuffled = W._varint_field(4, PR) + W._len_field(2, b"x" * 20) + W._len_field(1, W.View(3, 1).encode(), True) + \
        W._len_field(6, W.prepare_body(b"h" * 32), True) + W._len_field(3, b"s" * 65)
    assert H.reencode(shuffled) == w


# ---------------------------------------------------------------- messages.Messages
def test_add_message():  # TestMessages_AddMessage, messages_test.go:65-93
    h = H.Host()
    for m in gen_random_messages(5, (1, 1), PR, CM, RC):
        assert h.store_add(m.encode()) == 0
    for t in (PR, CM, RC):
        assert h.store_num(1, 1, t) == 5


def test_add_duplicates():  # TestMessages_AddDuplicates, messages_test.go:100-128
    h = H.Host()
    for m in gen_random_messages(5, (1, 1), PR):
        m.sender = b"1"
        h.store_add(m.encode())
    assert h.store_num(1, 1, PR) == 1


def test_prune():  # TestMessages_Prune, messages_test.go:131-178
    h = H.Host()
    for r in (1, 2, 3):
        for m in gen_random_messages(5, (1, r), PR):
            h.store_add(m.encode())
    h.store_prune(2)
    for r in (1, 2, 3):
        assert h.store_num(1, r, PR) == 0
    # heights >= the prune height survive (messages.go:140-144: `msgHeight < height`)
    for m in gen_random_messages(3, (2, 0), PR) + gen_random_messages(2, (5, 0), PR):
        h.store_add(m.encode())
    h.store_prune(2)
    assert h.store_num(2, 0, PR) == 3 and h.store_num(5, 0, PR) == 2
    h.store_prune(3)
    assert h.store_num(2, 0, PR) == 0 and h.store_num(5, 0, PR) == 2


@pytest.mark.parametrize("t", [PP, PR, CM, RC])
def test_get_valid_messages_prunes_invalid(t):  # TestMessages_GetValidMessagesMessage :183-268
    h = H.Host()
    for m in gen_random_messages(5, (1, 0), t):
        h.store_add(m.encode())
    assert h.store_num(1, 0, t) == 5
    assert h.store_get_valid(1, 0, t, lambda w: False) == []
    assert h.store_num(1, 0, t) == 0


def test_get_valid_messages_keeps_valid_and_is_per_view():
    h = H.Host()
    msgs = gen_random_messages(6, (1, 0), CM)
    for m in msgs + gen_random_messages(2, (1, 1), CM):
        h.store_add(m.encode())
    keep = {m.encode() for m in msgs[::2]}
    got = h.store_get_valid(1, 0, CM, lambda w: w in keep)
    assert set(got) == keep and h.store_num(1, 0, CM) == 3 and h.store_num(1, 1, CM) == 2


def test_get_extended_rcc():  # TestMessages_GetExtendedRCC :273-329
    h = H.Host()
    rounds = {0: gen_random_messages(4, (0, 0), RC), 1: gen_random_messages(5, (0, 1), RC),
              2: gen_random_messages(5, (0, 2), RC), 3: gen_random_messages(4, (0, 3), RC)}
    for ms in rounds.values():
        for m in ms:
            h.store_add(m.encode())
    got = h.store_get_extended_rcc(0, lambda w: True, lambda r, n: n >= 5)
    assert sorted(got) == sorted(m.encode() for m in rounds[2])
    assert h.store_num(0, 1, RC) == 5                      # GetExtendedRCC does not prune
    # round 0 can never be returned (`round <= highestRound`, messages.go:222)
    h2 = H.Host()
    for m in gen_random_messages(7, (0, 0), RC):
        h2.store_add(m.encode())
    assert h2.store_get_extended_rcc(0, lambda w: True, lambda r, n: True) == []


def test_get_most_round_change_messages():  # TestMessages_GetMostRoundChangeMessages :334-373
    h = H.Host()
    for r, c in ((0, 1), (1, 2), (2, 3)):
        for m in gen_random_messages(c, (0, r), RC):
            h.store_add(m.encode())
    got = h.store_get_most_rc(0, 0)
    assert len(got) == 3 and all(b"\x10\x02" in w[:8] for w in got)   # View.Round == 2
    assert h.store_get_most_rc(3, 0) == []                             # nothing at round >= 3


def test_store_random_ops_vs_python_restatement():
    rng = random.Random(7)
    h, ref = H.Host(), S.Messages()
    for step in range(400):
        op = rng.random()
        view = (rng.randrange(1, 4), rng.randrange(0, 3))
        t = rng.choice([PP, PR, CM, RC])
        if op < 0.6:
            m = gen_random_messages(1, view, t)[0]
            m.sender = str(rng.randrange(6)).encode()
            m.signature = bytes([rng.randrange(256)])       # distinguishes overwrites
            h.store_add(m.encode())
            ref.add_message(m)
        elif op < 0.8:
            salt = rng.randrange(4)
            pred = lambda w, salt=salt: (sum(w) + salt) % 3 != 0
            got = h.store_get_valid(view[0], view[1], t, pred)
            exp = ref.get_valid_messages(view[0], view[1], t, lambda m: pred(m.encode()))
            assert sorted(got) == sorted(m.encode() for m in exp)
        elif op < 0.9:
            hh = rng.randrange(1, 5)
            h.store_prune(hh)
            ref.prune_by_height(hh)
        else:
            q = rng.randrange(1, 4)
            got = h.store_get_extended_rcc(view[0], lambda w: sum(w) % 5 != 0, lambda r, n: n >= q)
            exp = ref.get_extended_rcc(view[0], lambda m: sum(m.encode()) % 5 != 0, lambda r, ms: len(ms) >= q)
            assert sorted(got) == sorted(m.encode() for m in exp)
        for tt in range(4):
            assert h.store_num(view[0], view[1], tt) == ref.num_messages(view[0], view[1], tt)


# ---------------------------------------------------------------- messages/helpers.go
def test_extract_committed_seals():  # TestMessages_ExtractCommittedSeals, helpers_test.go:13-86
    seal = b"committed seal"
    cm = lambda s: W.IbftMessage(sender=s, type=CM, payload=W.commit_body(b"", seal)).encode()
    assert H.extract_committed_seals([cm(b"signer1"), cm(b"signer2")]) == [(b"signer1", seal), (b"signer2", seal)]
    wrong = W.IbftMessage(type=PP).encode()
    assert H.extract_committed_seals([cm(b"signer1"), wrong]) is None       # ErrWrongCommitMessageType
    # COMMIT type with a nil payload -> a nil seal in the list (helpers.go:39-42)
    assert H.extract_committed_seals([W.IbftMessage(sender=b"a", type=CM).encode()]) == [None]


def test_has_unique_senders():  # TestMessages_HasUniqueSenders, helpers_test.go:413-465
    mk = lambda s: W.IbftMessage(sender=s).encode()
    assert not H.has_unique_senders([])
    assert not H.has_unique_senders([mk(b"node 1"), mk(b"node 1")])
    assert H.has_unique_senders([mk(b"node 1"), mk(b"node 2")])


def _pp(h, r, frm, hsh):
    return W.IbftMessage(view=W.View(h, r), sender=frm, type=PP, payload=W.preprepare_body(None, hsh, None)).encode()


def _pr(h, r, frm, hsh):
    return W.IbftMessage(view=W.View(h, r), sender=frm, type=PR, payload=W.prepare_body(hsh)).encode()


def test_are_valid_pc_messages_tables():
    ph = b"proposal hash"
    # TestMessages_HaveSameProposalHash, helpers_test.go:467-573 (height 1, roundLimit 2)
    assert not H.are_valid_pc_messages([], 1, 2)
    rc = W.IbftMessage(view=W.View(1, 1), sender=b"node 1", type=RC).encode()
    assert not H.are_valid_pc_messages([rc], 1, 2)
    assert not H.are_valid_pc_messages([_pp(1, 1, b"node 1", ph), _pr(1, 1, b"node 2", b"differing hash")], 1, 2)
    assert H.are_valid_pc_messages([_pp(1, 1, b"node 1", ph), _pr(1, 1, b"node 2", ph)], 1, 2)
    # TestMessages_AllHaveLowerRond, helpers_test.go:575-710 (round = 1)
    rnd = 1
    assert not H.are_valid_pc_messages([], 0, rnd)
    assert not H.are_valid_pc_messages([_pp(0, rnd, b"node 1", ph), _pr(0, rnd, b"node 2", ph)], 0, rnd)      # true == limit
    assert not H.are_valid_pc_messages([_pp(0, rnd + 1, b"node 1", ph), _pr(0, rnd + 1, b"node 2", ph)], 0, rnd)
    assert not H.are_valid_pc_messages([_pp(0, rnd, b"node 1", ph), _pr(0, rnd + 1, b"node 2", ph)], 0, rnd + 1)  # mismatch
    assert H.are_valid_pc_messages([_pp(0, rnd, b"node 1", ph), _pr(0, rnd, b"node 2", ph)], 0, 2)
    # TestMessages_AllHaveSameHeight, helpers_test.go:712-808
    assert not H.are_valid_pc_messages([_pp(1, 0, b"node 1", ph), _pr(2, 0, b"node 2", ph)], 1, 1)
    assert H.are_valid_pc_messages([_pp(1, 0, b"node 1", ph), _pr(1, 0, b"node 2", ph)], 1, 1)
    # duplicate senders (helpers.go:203-208)
    assert not H.are_valid_pc_messages([_pp(1, 0, b"node 1", ph), _pr(1, 0, b"node 1", ph)], 1, 1)


# ---------------------------------------------------------------- core.ValidatorManager
QUORUM_CASES = [  # Test_CalculateQuorum, /root/reference/core/validator_manager_test.go:18-187
    ({"A": 1, "B": 1, "C": 1, "D": 1}, "ABCD", True), ({"A": 1, "B": 1, "C": 1, "D": 1}, "AB", False),
    ({k: 1 for k in "ABCDEF"}, "ABCDE", True), ({k: 1 for k in "ABCDEF"}, "ABCD", False),
    ({"A": 2, "B": 2, "C": 2, "D": 3}, "ACD", True), ({"A": 2, "B": 2, "C": 2, "D": 3}, "AD", False),
    ({"A": 2, "B": 2, "C": 3, "D": 3}, "ABD", True), ({"A": 2, "B": 2, "C": 3, "D": 3}, "AD", False),
    ({"A": 2, "B": 7, "C": 7, "D": 5}, "ABC", True), ({"A": 2, "B": 7, "C": 7, "D": 5}, "CD", False),
]


@pytest.mark.parametrize("powers,signers,expect", QUORUM_CASES)
def test_calculate_quorum_table(powers, signers, expect):
    h = H.Host()
    pw = {k.encode(): v for k, v in powers.items()}
    assert h.vm_init(pw)
    assert h.vm_has_quorum([s.encode() for s in signers]) == expect
    ref = S.ValidatorManager()
    assert ref.init(pw) and ref.has_quorum([s.encode() for s in signers]) == expect
    assert h.vm_quorum() == ref.quorum == 2 * sum(powers.values()) // 3 + 1


def test_validator_manager_edges():
    h = H.Host()
    assert not h.vm_has_quorum([b"A"])                       # not initialised (validator_manager.go:82-84)
    assert not h.vm_init({b"A": 0, b"B": 0})                 # errVotingPowerNotCorrect
    assert h.vm_init({b"A": 1, b"B": 1, b"C": 1, b"D": 1})
    assert not h.vm_init({b"Z": 0})                          # failed re-init leaves the old table in place
    assert h.vm_has_quorum([b"A", b"B", b"C", b"unknown"])   # unknown senders contribute 0
    assert not h.vm_has_quorum([b"A", b"B", b"unknown", b"unknown2"])
    # HasPrepareQuorum (:99-127): proposer is added to the set; a PREPARE from the proposer voids it
    proposal = W.IbftMessage(view=W.View(1, 0), sender=b"A", type=PP, payload=W.preprepare_body(None, b"h", None)).encode()
    pr = lambda s: _pr(1, 0, s, b"h")
    assert h.vm_has_prepare_quorum(proposal, [pr(b"B"), pr(b"C")])
    assert not h.vm_has_prepare_quorum(proposal, [pr(b"B")])
    assert not h.vm_has_prepare_quorum(proposal, [pr(b"A"), pr(b"B"), pr(b"C")])
    assert not h.vm_has_prepare_quorum(None, [pr(b"B"), pr(b"C"), pr(b"D")])
    # u64 powers near 2^64 sum in 128 bits
    big = 2**64 - 1
    assert h.vm_init({b"A": big, b"B": big, b"C": big}) and h.vm_quorum() == 2 * 3 * big // 3 + 1
    assert h.vm_has_quorum([b"A", b"B", b"C"]) and not h.vm_has_quorum([b"A", b"B"])


# ---------------------------------------------------------------- core/ibft.go hot-path callers
def _node_set(n):
    return [f"node {i}".encode() for i in range(n)]


def test_is_acceptable_message_table():
    """TestIBFT_IsAcceptableMessage, /root/reference/core/ibft_test.go:1103-1216."""
    def run(valid_sender, state, msg_view):
        h = H.Host()
        h.vm_init({a: 1 for a in _node_set(4)})
        h.set_state(state[0], state[1], None)
        h.set_verifier(is_valid_validator=lambda w: valid_sender)
        m = W.IbftMessage(view=W.View(*msg_view) if msg_view else None, sender=b"node 1", type=PR,
                          payload=W.prepare_body(b"h"))
        return h.add_message(m.encode()) > 0
    # the seven rows of the reference table, in order (state view, message view)
    assert not run(False, (0, 0), None)            # invalid sender
    assert not run(True, (0, 0), None)             # malformed message (nil view)
    assert run(True, (0, 0), (100, 0))             # higher height, same round number
    assert run(True, (0, 1), (100, 0))             # higher height, lower round number
    assert run(True, (0, 0), (0, 1))               # same heights, higher round number
    assert not run(True, (0, 2), (0, 1))           # same heights, lower round number
    assert not run(True, (1, 0), (0, 0))           # lower height number


def test_add_message_signals_on_unverified_quorum():
    """IBFT.AddMessage (core/ibft.go:1101-1123; cases of ibft_test.go:3120-3246): the signal
    fires once the STORED (unverified) messages of the view reach quorum."""
    h = H.Host()
    nodes = _node_set(4)
    h.vm_init({a: 1 for a in nodes})
    h.set_state(1, 0, None)
    rcs = [h.add_message(W.IbftMessage(view=W.View(1, 0), sender=a, type=CM, payload=W.commit_body(b"h", b"s")).encode())
           for a in nodes]
    assert rcs == [1, 1, 2, 2]                      # quorum of 4 equal validators is 3
    assert h.add_message(W.IbftMessage(view=W.View(2, 0), sender=nodes[0], type=CM,
                                       payload=W.commit_body(b"h", b"s")).encode()) == 1   # future height: stored, no probe
    assert h.add_message(b"\xff\xff") == -1         # undecodable wire bytes
    # PREPARE quorum needs the proposal message (HasPrepareQuorum: nil proposal -> false)
    h2 = H.Host()
    h2.vm_init({a: 1 for a in nodes})
    h2.set_state(1, 0, None)
    pr = [W.IbftMessage(view=W.View(1, 0), sender=a, type=PR, payload=W.prepare_body(b"h")).encode() for a in nodes[1:]]
    assert [h2.add_message(w) for w in pr] == [1, 1, 1]
    proposal = W.IbftMessage(view=W.View(1, 0), sender=nodes[0], type=PP,
                             payload=W.preprepare_body(W.Proposal(b"block", 0), b"h", None)).encode()
    h3 = H.Host()
    h3.vm_init({a: 1 for a in nodes})
    h3.set_state(1, 0, proposal)
    assert [h3.add_message(w) for w in pr] == [1, 2, 2]     # proposer + 2 PREPAREs = 3


def _commit_round(nodes, good_hash, seals, hashes=None):
    return [W.IbftMessage(view=W.View(1, 0), sender=a, type=CM,
                          payload=W.commit_body((hashes or {}).get(a, good_hash), seals[a])).encode() for a in nodes]


def test_handle_commit_seals_and_pruning():
    """TestRunCommit, /root/reference/core/ibft_test.go:977-1099: on quorum the seals handed on
    are exactly {Signer: From, Signature: CommittedSeal} of the surviving COMMITs; invalid
    messages are pruned from the store and a2 is skipped when a1 fails (ibft.go:938-943)."""
    nodes = _node_set(4)
    good = b"proposal hash"
    seals = {a: b"seal of " + a for a in nodes}
    proposal = W.IbftMessage(view=W.View(1, 0), sender=nodes[0], type=PP,
                             payload=W.preprepare_body(W.Proposal(b"block", 0), good, None)).encode()
    calls = []

    def mk(bad_seal=(), bad_hash=()):
        h = H.Host()
        h.vm_init({a: 1 for a in nodes})
        h.set_state(1, 0, proposal)
        h.set_verifier(
            is_valid_proposal_hash=lambda prop, hsh: prop == (b"block", 0) and hsh == good,
            is_valid_committed_seal=lambda hsh, seal: calls.append(seal[0]) or (seal[0] not in bad_seal))
        for w in _commit_round(nodes, good, seals, {a: b"bad hash" for a in bad_hash}):
            h.store_add(w)
        return h
    h = mk()
    ok, out = h.handle_commit(1, 0)
    assert ok and sorted(out) == sorted((a, seals[a]) for a in nodes)
    calls.clear()
    h = mk(bad_seal={nodes[3]}, bad_hash={nodes[2]})
    ok, out = h.handle_commit(1, 0)
    assert not ok and out == [] and h.store_num(1, 0, CM) == 2     # the two invalid COMMITs were pruned
    assert nodes[2] not in calls                                    # a2 short-circuited after a1 failed
    h = mk(bad_seal={nodes[3]})
    ok, out = h.handle_commit(1, 0)
    assert ok and sorted(out) == sorted((a, seals[a]) for a in nodes[:3])
    # nil proposal in state (state.go:135-144) -> the hash check sees a nil proposal
    h = mk()
    h.set_state(1, 0, None)
    assert h.handle_commit(1, 0) == (False, [])


def test_handle_prepare():
    """TestRunPrepare, /root/reference/core/ibft_test.go:870-973."""
    nodes = _node_set(4)
    good = b"proposal hash"
    proposal = W.IbftMessage(view=W.View(1, 0), sender=nodes[0], type=PP,
                             payload=W.preprepare_body(W.Proposal(b"block", 0), good, None)).encode()
    h = H.Host()
    h.vm_init({a: 1 for a in nodes})
    h.set_state(1, 0, proposal)
    h.set_verifier(is_valid_proposal_hash=lambda prop, hsh: hsh == good)
    for a, hsh in zip(nodes[1:], (good, good, b"wrong")):
        h.store_add(W.IbftMessage(view=W.View(1, 0), sender=a, type=PR, payload=W.prepare_body(hsh)).encode())
    ok, prepared = h.handle_prepare(1, 0)
    assert ok and len(prepared) == 2 and h.store_num(1, 0, PR) == 2
    # a PREPARE from the proposer itself voids the quorum (validator_manager.go:117-121)
    h.store_add(W.IbftMessage(view=W.View(1, 0), sender=nodes[0], type=PR, payload=W.prepare_body(good)).encode())
    assert h.handle_prepare(1, 0)[0] is False


# ---------------------------------------------------------------- incremental quorum probe (§8f rank 1)
def test_add_message_fast_equals_stock_on_random_streams():
    """AddMessageFast (O(1) Σ-power index fed by the store's insert/prune hooks) returns the same
    0/1/2 as the stock AddMessage (re-walk of the view, core/ibft.go:1113-1120) on random streams
    with duplicates, overwrites, unknown senders, proposer PREPAREs, prunes, validator-set changes
    and interleaved handleCommit pruning."""
    rng = random.Random(11)
    nodes = _node_set(9) + [b"stranger 1", b"stranger 2"]
    for trial in range(6):
        powers = {a: rng.randrange(1, 6) for a in nodes[:9]}
        hosts = [H.Host(), H.Host()]
        hosts[1].enable_quorum_index()
        proposal = W.IbftMessage(view=W.View(1, 0), sender=nodes[0], type=PP,
                                 payload=W.preprepare_body(W.Proposal(b"blk", 0), b"h", None)).encode()
        for h in hosts:
            assert h.vm_init(powers)
            h.set_state(1, 0, proposal if trial % 2 == 0 else None)
            h.set_verifier(is_valid_validator=lambda w: b"bad-sender" not in w,
                           is_valid_committed_seal=lambda hsh, seal: seal is not None and seal[1] != b"bad seal",
                           is_valid_proposal_hash=lambda prop, hsh: True)
        for step in range(300):
            op = rng.random()
            if op < 0.85:
                t = rng.choice([PP, PR, CM, CM, RC])
                view = (rng.choice([1, 1, 1, 2]), rng.choice([0, 0, 1]))
                frm = rng.choice(nodes)
                body = {PP: W.preprepare_body(None, b"h", None), PR: W.prepare_body(b"h"),
                        CM: W.commit_body(b"h", rng.choice([b"seal", b"bad seal"])),
                        RC: W.round_change_body(None, None)}[t]
                m = W.IbftMessage(view=W.View(*view), sender=frm, type=t, payload=body,
                                  signature=rng.choice([b"ok", b"bad-sender"]))
                w = m.encode()
                assert hosts[0].add_message(w) == hosts[1].add_message_fast(w), (trial, step)
            elif op < 0.92:
                r0, r1 = hosts[0].handle_commit(1, 0), hosts[1].handle_commit(1, 0)
                assert r0[0] == r1[0] and sorted(r0[1]) == sorted(r1[1])
            elif op < 0.96:
                below = rng.choice([1, 2])
                for h in hosts:
                    h.store_prune(below)
            else:
                powers = {a: rng.randrange(1, 6) for a in rng.sample(nodes[:9], 7)}
                for h in hosts:
                    assert h.vm_init(powers)
            for t in range(4):
                assert hosts[0].store_num(1, 0, t) == hosts[1].store_num(1, 0, t)


def test_ingest_cost_quadratic_vs_incremental():
    """The reference's ingest is O(N²) per phase (every accepted message re-walks the view); the
    incremental probe is O(N log N).  Same signal sequence, measured ratio grows with N."""
    import time
    def run(n, fast):
        nodes = [i.to_bytes(4, "big") * 5 for i in range(n)]
        h = H.Host()
        if fast:
            h.enable_quorum_index()
        h.vm_init({a: 1 for a in nodes})
        h.set_state(1, 0, None)
        wires = [W.IbftMessage(view=W.View(1, 0), sender=a, type=CM, payload=W.commit_body(b"h" * 32, b"s" * 65)).encode()
                 for a in nodes]
        t0 = time.perf_counter()
        out = [h.add_message_fast(w) if fast else h.add_message(w) for w in wires]
        return out, time.perf_counter() - t0
    for n in (256, 2048):
        slow, ts = run(n, False)
        fast, tf = run(n, True)
        assert slow == fast and slow.count(2) == n - (2 * n // 3 + 1) + 1
    assert ts / tf > 5          # at N = 2048 the re-walk dominates by far more than this


def test_receive_side_peek_agrees_with_the_decoder():
    """The ingest path classifies a message by a look at its top-level fields (no allocation) while a helper thread
    decodes it: the look must succeed whenever the decoder does and report the same view, type and payload member —
    repeated fields merged the same way — on well-formed messages, on the wire fixtures and under byte-level fuzz."""
    import ctypes as C
    L = H.lib()
    L.ibft_host_peek_vs_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 12)()

    L.ibft_host_peek_shortcut_agrees.argtypes = [C.c_char_p, C.c_size_t]
    regular = [0]

    def check(w):
        L.ibft_host_peek_vs_decode(w, len(w), out)
        pk, dc = list(out[:6]), list(out[6:])
        if dc[0]:
            assert pk[0] == 1 and pk[1:] == dc[1:], (w.hex(), pk, dc)
        # the look's shortcut for the regular PREPARE / COMMIT shape reports what the general walk reports, field offsets included
        same = L.ibft_host_peek_shortcut_agrees(w, len(w))
        assert same != 0, w.hex()
        regular[0] += same == 2
        return pk[0], dc[0]
    base = []
    for t, body in ((PP, W.preprepare_body(W.Proposal(b"raw", 3), b"h" * 32, W.round_change_certificate([]))),
                    (PR, W.prepare_body(b"h" * 32)), (CM, W.commit_body(b"h" * 32, b"s" * 65)),
                    (RC, W.round_change_body(W.Proposal(b"raw", 1), W.prepared_certificate(None, []))), (CM, None)):
        for view in (W.View(7, 2), W.View(0, 0), W.View(2**63 + 1, 2**40), None):
            base.append(W.IbftMessage(view=view, sender=b"a" * 20, signature=b"g" * 65, type=t, payload=body).encode())
    for v in json.load(open(os.path.join(HERE, "golden", "wire_vectors.json"))):
        base.append(bytes.fromhex(v["wire"]))
    # repeated View / payload fields: the later one wins in both
    m = W.IbftMessage(view=W.View(5, 1), sender=b"a" * 20, type=PR, payload=W.prepare_body(b"h" * 32)).encode()
    base.append(m + W._len_field(1, W.View(9, 4).encode(), True) + W._len_field(7, W.commit_body(b"x" * 32, b"y" * 65), True))
    base.append(m + b"\x78\x05" + b"\x82\x01\x02hi")            # unknown fields
    for w in base:
        assert check(w) == (1, 1)
    rng = random.Random(11)
    agree = 0
    for _ in range(4000):
        w = bytearray(rng.choice(base))
        for _ in range(rng.randint(1, 3)):
            op = rng.random()
            pos = rng.randrange(len(w) + 1)
            if op < 0.4 and w:
                w[min(pos, len(w) - 1)] ^= 1 << rng.randrange(8)
            elif op < 0.7:
                w[pos:pos] = bytes([rng.randrange(256)])
            elif w:
                del w[min(pos, len(w) - 1)]
        p_ok, d_ok = check(bytes(w))
        agree += p_ok == d_ok
    assert agree > 3000          # (the look may accept what a nested field later breaks; never the other way round)
    assert regular[0] > 300      # (row candidates among the fuzzed messages: the shortcut's own territory was covered)
