"""Rows instead of objects (include/ibft_host.h: ibft_host_use_rows; go-ibft_amd/host/messages.hpp: LeanRow).

A PREPARE / COMMIT of the current view that a batch backend judged completely from its bytes is stored as a row and never
decoded; everything the reference's callers can observe — AddMessage's return and signal (core/ibft.go:1085-1131), the
store's counts and contents (messages/messages.go), handlePrepare / handleCommit (core/ibft.go:843-960), the prepared
messages of the certificate, the committed seals — must be what the object path answers.  These tests run the same traffic
through two mirrors, rows on and rows off (and a third with no batch backend at all: the reference's own control flow),
with the loop backend standing in for the device (its VerifyMessagesWire vouches for a message when its bytes are the
canonical encoding).  On an MI355X the same comparison runs against libibftgpu (tests/test_gpu_host.py)."""
import random

import pytest

import go_ibft_amd.hostlib as H
from oracle import wire as W
from test_host_roundchange import World, fake_hash, PP, PR, CM, RC


def _world(n, seed, bad_hash=(2, 5), forged=(4,), bad_seal=(7,)):
    w = World(n, seed)
    raw = b"the block %d" % seed
    hsh = fake_hash(raw, 0)
    proposal = W.IbftMessage(view=W.View(1, 0), sender=w.proposer(1, 0), type=PP,
                             payload=W.preprepare_body(W.Proposal(raw, 0), hsh, None))
    prepares = [W.IbftMessage(view=W.View(1, 0), sender=a, type=PR, signature=b"sig-pr-" + a,
                              payload=W.prepare_body(hsh if i not in bad_hash else b"x" * 32))
                for i, a in enumerate(w.addrs) if a != proposal.sender]
    commits = [W.IbftMessage(view=W.View(1, 0), sender=a, type=CM, signature=b"sig-cm-" + a,
                             payload=W.commit_body(hsh if i not in bad_hash else b"y" * 32,
                                                   b"seal-" + a + (b"-bad" if i in bad_seal else b"")))
               for i, a in enumerate(w.addrs)]
    for i in forged:
        w.bad_wires.add(prepares[i].encode())
        w.bad_wires.add(commits[i].encode())
    ver = w.verifier()
    ver["is_valid_committed_seal"] = lambda h_, seal: seal is not None and not seal[1].endswith(b"-bad")
    return w, ver, proposal, prepares, commits


def _host(w, ver, proposal, rows, batch=True, index=True):
    h = H.Host()
    assert h.vm_init({a: 1 for a in w.addrs})
    h.set_verifier(**ver)
    h.set_state(1, 0, proposal.encode())
    if index:
        h.enable_quorum_index()
    if batch:
        h.use_loop_batch(0)
        h.use_batch(True)
    h.use_rows(rows)
    return h


def _same_answers(a, b, view=(1, 0), fresh=True):
    for t in (PR, CM):
        assert not fresh or a.store_num(*view, t) == b.store_num(*view, t)
    pa, pb = a.handle_prepare(*view), b.handle_prepare(*view)
    assert pa[0] == pb[0] and sorted(pa[1]) == sorted(pb[1])
    ca, cb = a.handle_commit(*view), b.handle_commit(*view)
    assert ca[0] == cb[0] and sorted(ca[1]) == sorted(cb[1])
    for t in (PR, CM):
        assert a.store_num(*view, t) == b.store_num(*view, t)      # the same messages pruned
        assert sorted(a.store_get_valid(*view, t)) == sorted(b.store_get_valid(*view, t))
    return pa, ca


@pytest.mark.parametrize("seed", range(6))
def test_rows_answer_like_objects(seed):
    rng = random.Random(seed)
    n = rng.choice([4, 7, 13, 31])
    w, ver, proposal, prepares, commits = _world(n, seed, bad_hash=(2,) if n > 4 else (), forged=(1,), bad_seal=(3,) if n > 4 else ())
    others = [W.IbftMessage(view=W.View(1, 1), sender=a, type=CM, signature=b"s", payload=W.commit_body(b"f" * 32, b"seal-" + a))
              for a in w.addrs[:3]]                                          # another round: objects in every mode
    others += [W.IbftMessage(view=W.View(1, 0), sender=w.addrs[0], type=PR, signature=b"s2", payload=W.commit_body(b"f" * 32, b"z"))]  # type ≠ payload
    wires = [m.encode() for m in prepares + commits + others] + [b"\xff\xff", b""]
    rng.shuffle(wires)
    wires += rng.sample(wires, 5)                                            # gossip: repeats inside and across batches
    rows, objs, stock = _host(w, ver, proposal, True), _host(w, ver, proposal, False), _host(w, ver, proposal, False, batch=False)
    got = {id(h): [] for h in (rows, objs, stock)}
    k = 0
    while k < len(wires):
        step = rng.choice([1, 3, 8, 50])
        for h in (rows, objs):
            got[id(h)] += h.ingest_wire(wires[k:k + step])[0]
        got[id(stock)] += [stock.add_message(x) if x else -1 for x in wires[k:k + step]]
        k += step
    assert got[id(rows)] == got[id(objs)]
    assert [g if g >= 0 else 0 for g in got[id(rows)]] == [e if e >= 0 else 0 for e in got[id(stock)]]   # signal included
    assert rows.rows_kept > 0 and objs.rows_kept == 0
    before = rows.loop_batch_calls()
    pa, ca = _same_answers(rows, objs)
    assert rows.loop_batch_calls() == before                                # the walks asked the backend nothing
    _same_answers(rows, stock, fresh=False)                               # (stock has not walked — and pruned — yet)
    if n >= 13:
        assert pa[0] and ca[0]
    for h in (rows, objs, stock):
        h.close()


def test_rows_become_objects_when_somebody_asks_for_them():
    w, ver, proposal, prepares, commits = _world(13, 3)
    rows, objs = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    wires = [m.encode() for m in prepares + commits]
    for h in (rows, objs):
        h.ingest_wire(wires[:11])
    kept = rows.rows_kept
    assert kept > 0
    # an object-level access: the view's rows are decoded, verdicts noted — same contents as the object store
    assert sorted(rows.store_get_valid(1, 0, PR)) == sorted(objs.store_get_valid(1, 0, PR))
    # what arrives for that view afterwards is stored as objects (a view holds rows or objects, never both) …
    for h in (rows, objs):
        h.ingest_wire(wires[11:])
    # … and the walks give the same answers, still without asking the backend (the verdicts travelled with the objects)
    before = rows.loop_batch_calls()
    _same_answers(rows, objs)
    assert rows.loop_batch_calls() == before
    rows.close(); objs.close()


def test_rows_and_a_validator_set_or_proposal_change():
    """A row's verdicts are only as good as the validator set and the proposal they were computed against."""
    w, ver, proposal, prepares, commits = _world(13, 4, forged=())
    rows, objs = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    wires = [m.encode() for m in prepares + commits]
    for h in (rows, objs):
        h.ingest_wire(wires)
    assert rows.rows_kept > 0
    # the validator set changes (a sender loses its seat): every stored verdict is void, both mirrors ask again
    smaller = {a: 1 for a in w.addrs[:-1]}
    for h in (rows, objs):
        assert h.vm_init(smaller)
    _same_answers(rows, objs)
    # another proposal for the view: the closure verdicts are void
    raw2 = b"another block"
    proposal2 = W.IbftMessage(view=W.View(1, 0), sender=w.proposer(1, 0), type=PP,
                              payload=W.preprepare_body(W.Proposal(raw2, 0), fake_hash(raw2, 0), None))
    rows2, objs2 = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    for h in (rows2, objs2):
        h.ingest_wire(wires)
        h.set_state(1, 0, proposal2.encode())
    pa, ca = _same_answers(rows2, objs2)
    assert not pa[0] and not ca[0]                                           # nobody prepared / committed THAT proposal
    # re-delivery after the change is judged afresh (a remembered row is not stored with a stale verdict)
    for h in (rows2, objs2):
        h.set_state(1, 0, proposal.encode())
    ra, rb = rows2.ingest_wire(wires)[0], objs2.ingest_wire(wires)[0]
    assert ra == rb
    _same_answers(rows2, objs2)
    for h in (rows, objs, rows2, objs2):
        h.close()


def test_rows_follow_the_state():
    """Re-delivery of a stored row after the round moved on is rejected like any stale message; pruning drops rows."""
    w, ver, proposal, prepares, commits = _world(7, 5, bad_hash=(), forged=(), bad_seal=())
    rows, objs = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    wires = [m.encode() for m in commits]
    for h in (rows, objs):
        assert h.ingest_wire(wires)[0][-1] == 2                              # quorum signalled
    again = [h.ingest_wire(wires[:3])[0] for h in (rows, objs)]
    assert again[0] == again[1] == [2, 2, 2]                                 # byte-identical re-delivery: stored again, signalled again
    for h in (rows, objs):
        h.set_state(1, 1, None)
    stale = [h.ingest_wire(wires[:3])[0] for h in (rows, objs)]
    assert stale[0] == stale[1] == [0, 0, 0]
    assert rows.store_num(1, 0, CM) == objs.store_num(1, 0, CM) == 7
    for h in (rows, objs):
        h.store_prune(2)
        assert h.store_num(1, 0, CM) == 0
    rows.close(); objs.close()


def test_failing_set_call_leaves_the_object_routes():
    w, ver, proposal, prepares, commits = _world(13, 6)
    rows, objs = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    rows.use_loop_batch(8)                                                   # the message-set calls fail
    rows.use_batch(True)
    wires = [m.encode() for m in prepares + commits]
    assert rows.ingest_wire(wires)[0] == objs.ingest_wire(wires)[0]
    assert rows.rows_kept == 0
    _same_answers(rows, objs)
    rows.close(); objs.close()


@pytest.mark.parametrize("seed", range(4))
def test_a_view_may_hold_rows_and_objects(seed):
    """A message the backend does not vouch for byte by byte (here: the type field written twice — it decodes to the same
    message, it is acceptable, it is not the canonical encoding) is stored as an object in the SAME view as the rows; a sender is in at
    most one of the two (the last writer wins, whichever form it takes) and is counted once.  One such message does not
    turn the view's rows into objects."""
    rng = random.Random(100 + seed)
    w, ver, proposal, prepares, commits = _world(13, 20 + seed, bad_hash=(2,), forged=(), bad_seal=(3,))
    odd = lambda m: bytes([0x20, m.type]) + m.encode()                        # field 4 (type) twice: the last one wins
    traffic = []
    for k, m in enumerate(prepares + commits):
        form = rng.choice(["row", "row", "row", "odd", "row-then-odd", "odd-then-row"])
        traffic += {"row": [m.encode()], "odd": [odd(m)], "row-then-odd": [m.encode(), odd(m)],
                    "odd-then-row": [odd(m), m.encode()]}[form]
    rng.shuffle(traffic)
    rows, objs, stock = _host(w, ver, proposal, True), _host(w, ver, proposal, False), _host(w, ver, proposal, False, batch=False)
    got_r, got_o, got_s = [], [], []
    k = 0
    while k < len(traffic):
        step = rng.choice([1, 2, 5, 40])
        got_r += rows.ingest_wire(traffic[k:k + step])[0]
        got_o += objs.ingest_wire(traffic[k:k + step])[0]
        got_s += [stock.add_message(x) for x in traffic[k:k + step]]
        k += step
        for t in (PR, CM):
            assert rows.store_num(1, 0, t) == objs.store_num(1, 0, t) == stock.store_num(1, 0, t)
    assert got_r == got_o == got_s
    canonical = sum(1 for x in traffic if not x.startswith(b"\x20"))
    assert rows.rows_kept == canonical                                        # every canonical message went in as a row …
    _same_answers(rows, objs)
    _same_answers(rows, stock, fresh=False)
    for h in (rows, objs, stock):
        h.close()


def test_rows_do_not_pin_a_flood():
    """Rows point into the buffer of the batch they arrived in.  A large batch of which little is stored — rejected bulk around a
    few honest messages — must not stay alive for them: the stored rows move into a buffer of their own."""
    w, ver, proposal, prepares, commits = _world(13, 9, bad_hash=(), forged=(), bad_seal=())
    rows, objs = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    junk = []
    for k in range(40):                                     # forged COMMITs with 10 KB seals: rejected, 400 KB of them
        m = W.IbftMessage(view=W.View(1, 0), sender=w.addrs[k % 13], type=CM, signature=b"junk-%d" % k,
                          payload=W.commit_body(b"j" * 32, bytes([k]) * 10_000))
        w.bad_wires.add(m.encode())
        junk.append(m.encode())
    honest = [m.encode() for m in commits[:3]]
    batch = junk[:20] + honest + junk[20:]
    ra, rb = rows.ingest_wire(batch)[0], objs.ingest_wire(batch)[0]
    assert ra == rb and ra.count(0) == 40 and rows.rows_kept == 3
    assert rows.repacked_bytes == sum(len(x) for x in honest) and objs.repacked_bytes == 0
    # the rows live on in their own buffer: re-delivery is still recognised, the walks answer as before
    again = rows.ingest_wire(honest)
    assert again[0] == [1, 1, 1] and again[2] == 3          # three hits, nothing asked
    rest = [m.encode() for m in prepares + commits[3:]]
    assert rows.ingest_wire(rest)[0] == objs.ingest_wire(rest)[0]
    assert rows.repacked_bytes == sum(len(x) for x in honest)   # (an ordinary batch is left where it is)
    _same_answers(rows, objs)
    rows.close(); objs.close()


def test_a_sender_that_keeps_replacing_its_message_pins_one_batch_not_all_of_them():
    """Round-3 advice: a Byzantine validator alternates two encodings-that-verify of its COMMIT (here: two seals the mock
    accepts), one per batch, each batch padded with other people's rejected bytes.  The reference keeps ONE message per
    sender per view (messages/messages.go:54-65); the row store must not keep every batch buffer alive behind the one live
    row, nor grow a row slot per replacement."""
    w, ver, proposal, prepares, commits = _world(13, 21, bad_hash=(), forged=(), bad_seal=())
    rows, objs = _host(w, ver, proposal, True), _host(w, ver, proposal, False)
    for h in (rows, objs):
        h.set_repack_min_bytes(1 << 30)                     # (the low-yield repack would hide the effect: off)
    base = [m.encode() for m in commits[1:4]]
    assert rows.ingest_wire(base)[0] == objs.ingest_wire(base)[0]
    who = w.addrs[0]
    for k in range(200):
        variant = W.IbftMessage(view=W.View(1, 0), sender=who, type=CM, signature=b"sig-%d" % k,
                                payload=W.commit_body(fake_hash(b"the block 21", 0), b"seal-%d-" % (k % 2) + who)).encode()
        junk = W.IbftMessage(view=W.View(1, 0), sender=w.addrs[5], type=CM, signature=b"junk-%d" % k,
                             payload=W.commit_body(b"j" * 32, bytes([k % 251]) * 2000)).encode()
        w.bad_wires.add(junk)
        ra, rb = rows.ingest_wire([junk, variant])[0], objs.ingest_wire([junk, variant])[0]
        assert ra == rb == [0, 1]
    live, slots, buffers = rows.lean_stats(1, 0, CM)
    assert live == 4 and slots == 4 and buffers == 2        # the first batch (three rows) + the LAST variant's batch
    assert rows.store_num(1, 0, CM) == objs.store_num(1, 0, CM) == 4
    # senders that come and go (pruned by a walk, then back): dead slots are compacted away
    rest = [m.encode() for m in prepares + commits[4:]]
    assert rows.ingest_wire(rest)[0] == objs.ingest_wire(rest)[0]
    _same_answers(rows, objs)
    live, slots, buffers = rows.lean_stats(1, 0, CM)
    assert slots <= 2 * live + 32 and buffers <= 3
    rows.close(); objs.close()
