"""SURVEY.md §8f ranks 1 and 2, host side, WITHOUT a device: the batch control flow of the C++ mirror is driven
through a batch backend that loops over the callback Verifier (ibft_host_use_loop_batch), so that

  * handleRoundChangeMessage (core/ibft.go:470-512 → messages.GetExtendedRCC, messages/messages.go:202-245) with
    one sender batch + one hash batch per distinct proposal decides exactly like the per-message walk;
  * a batch backend that reports failure makes handlePrepare / handleCommit / handleRoundChangeMessage answer
    with the per-message verifier under the same lock (the mirror used to return "nothing");
  * the receive side (IngestWire): one batch per micro-batch, a verdict cache keyed by the full wire bytes —
    re-deliveries hit it, a replayed (from, signature) under a different payload does not.
"""
import random

import pytest

import go_ibft_amd.hostlib as H
from oracle import wire as W

PP, PR, CM, RC = W.PREPREPARE, W.PREPARE, W.COMMIT, W.ROUND_CHANGE


def nodes(n):
    return [f"node {i:02d}".encode() for i in range(n)]


def fake_hash(raw: bytes, rnd: int) -> bytes:
    return (b"H|" + raw[:8] + b"|%d" % rnd).ljust(32, b".")[:32]


class World:
    """a validator set, a round-robin proposer, and a mock backend whose verdicts are looked up in sets"""

    def __init__(self, n, seed):
        self.rng = random.Random(seed)
        self.addrs = nodes(n)
        self.n = n
        self.bad_wires = set()        # IsValidValidator == false for exactly these message bytes
        self.calls = {"validator": 0, "hash": 0}

    def proposer(self, h, r):
        return self.addrs[(h + r) % self.n]

    def verifier(self):
        def vv(wire):
            self.calls["validator"] += 1
            return wire not in self.bad_wires

        def ph(prop, hsh):
            self.calls["hash"] += 1
            return prop is not None and hsh == fake_hash(prop[0], prop[1])
        return dict(is_valid_validator=vv, is_valid_proposal_hash=ph,
                    is_proposer=lambda who, hh, rr: who == self.proposer(hh, rr))

    def host(self):
        h = H.Host()
        assert h.vm_init({a: 1 for a in self.addrs})
        h.set_verifier(**self.verifier())
        return h

    def certificate(self, height, round_, raw, corrupt=None):
        """PREPREPARE of `round_` by its proposer + a quorum of PREPAREs by others; signatures are opaque tags"""
        hsh = fake_hash(raw, round_)
        prop = self.proposer(height, round_)
        pp = W.IbftMessage(view=W.View(height, round_), sender=prop, type=PP, signature=b"sig-pp-" + prop,
                           payload=W.preprepare_body(W.Proposal(raw, round_), hsh, None))
        others = [a for a in self.addrs if a != prop]
        self.rng.shuffle(others)
        q = 2 * self.n // 3 + 1
        prs = [W.IbftMessage(view=W.View(height, round_), sender=a, type=PR, signature=b"sig-pr-" + a,
                             payload=W.prepare_body(hsh)) for a in others[:q - 1]]
        if corrupt == "bad_prepare_signature":
            self.bad_wires.add(prs[self.rng.randrange(len(prs))].encode())
        elif corrupt == "bad_proposal_signature":
            self.bad_wires.add(pp.encode())
        elif corrupt == "wrong_hash_in_prepare":
            k = self.rng.randrange(len(prs))
            prs[k] = W.IbftMessage(view=W.View(height, round_), sender=prs[k].sender, type=PR, signature=prs[k].signature,
                                   payload=W.prepare_body(fake_hash(b"other", round_)))
        elif corrupt == "too_few_prepares":
            prs = prs[: q // 2]
        # ---- the rest of validPC's / AreValidPCMessages' rules (core/ibft.go:1162-1231, messages/helpers.go:167-214)
        elif corrupt == "duplicate_sender":
            prs[-1] = prs[0]
        elif corrupt == "prepare_from_proposer":
            prs[0] = W.IbftMessage(view=W.View(height, round_), sender=prop, type=PR, signature=b"sig-pr-" + prop, payload=W.prepare_body(hsh))
        elif corrupt == "proposal_not_from_proposer":
            pp = W.IbftMessage(view=W.View(height, round_), sender=others[-1], type=PP, signature=b"sig-pp-" + others[-1],
                               payload=W.preprepare_body(W.Proposal(raw, round_), hsh, None))
        elif corrupt == "one_prepare_of_another_round":
            k = self.rng.randrange(len(prs))
            prs[k] = W.IbftMessage(view=W.View(height, round_ + 1), sender=prs[k].sender, type=PR, signature=prs[k].signature, payload=W.prepare_body(hsh))
        elif corrupt == "one_prepare_of_another_height":
            k = self.rng.randrange(len(prs))
            prs[k] = W.IbftMessage(view=W.View(height + 1, round_), sender=prs[k].sender, type=PR, signature=prs[k].signature, payload=W.prepare_body(hsh))
        elif corrupt == "commit_among_prepares":     # type COMMIT: "all messages in the PC are Prepare messages" fails
            k = self.rng.randrange(len(prs))
            prs[k] = W.IbftMessage(view=W.View(height, round_), sender=prs[k].sender, type=CM, signature=prs[k].signature,
                                   payload=W.commit_body(hsh, b"seal"))
        elif corrupt == "unknown_field_in_prepare":   # a PrepareMessage with a field 2 nobody knows: still a valid PREPARE (not a corruption)
            k = self.rng.randrange(len(prs))
            prs[k] = W.IbftMessage(view=W.View(height, round_), sender=prs[k].sender, type=PR, signature=prs[k].signature,
                                   payload=W.commit_body(hsh, b"seal"))
        elif corrupt == "short_hash_everywhere":     # 31-byte hashes that agree with each other (and with nothing the backend accepts)
            pp = W.IbftMessage(view=W.View(height, round_), sender=prop, type=PP, signature=b"sig-pp-" + prop,
                               payload=W.preprepare_body(W.Proposal(raw, round_), hsh[:31], None))
            prs = [W.IbftMessage(view=W.View(height, round_), sender=m.sender, type=PR, signature=m.signature, payload=W.prepare_body(hsh[:31]))
                   for m in prs]
        elif corrupt == "stranger_among_prepares":   # a sender outside the validator set: validly "signed" for the mock, no voting power
            prs[0] = W.IbftMessage(view=W.View(height, round_), sender=b"a stranger", type=PR, signature=b"sig", payload=W.prepare_body(hsh))
        elif corrupt == "quorum_minus_one_plus_stranger":
            prs = prs[: q - 2] + [W.IbftMessage(view=W.View(height, round_), sender=b"a stranger", type=PR, signature=b"sig",
                                                payload=W.prepare_body(hsh))]
        return pp, prs


def rc_message(w, height, round_, sender, raw=None, cert=None, cert_round=0):
    body = W.round_change_body(W.Proposal(raw, cert_round) if raw is not None else None,
                               W.prepared_certificate(*cert) if cert is not None else None)
    return W.IbftMessage(view=W.View(height, round_), sender=sender, type=RC, signature=b"sig-rc-" + sender, payload=body)


KINDS = [None, None, None, "bad_prepare_signature", "bad_proposal_signature", "wrong_hash_in_prepare", "too_few_prepares",
         "no_certificate", "proposal_without_certificate", "other_proposal", "duplicate_sender", "prepare_from_proposer",
         "proposal_not_from_proposer", "one_prepare_of_another_round", "one_prepare_of_another_height", "commit_among_prepares",
         "unknown_field_in_prepare", "short_hash_everywhere", "stranger_among_prepares", "quorum_minus_one_plus_stranger"]


@pytest.mark.parametrize("seed", range(12))
def test_round_change_batch_equals_stock(seed):
    w = World(n=7 + seed % 4, seed=seed)
    height = 3
    raw = b"block-%d" % seed
    stock, batch = w.host(), w.host()
    batch.use_loop_batch(0)
    batch.use_batch(True)
    for rnd in (1, 2, 3):
        senders = list(w.addrs)
        w.rng.shuffle(senders)
        for a in senders[: w.rng.randrange(2, w.n + 1)]:
            kind = w.rng.choice(KINDS)
            cert_round = w.rng.randrange(0, rnd)
            if kind == "no_certificate":
                m = rc_message(w, height, rnd, a)
            elif kind == "proposal_without_certificate":
                m = rc_message(w, height, rnd, a, raw=raw, cert=None, cert_round=cert_round)
            elif kind == "other_proposal":      # certificate for another block than the carried proposal
                m = rc_message(w, height, rnd, a, raw=raw, cert=w.certificate(height, cert_round, b"another"), cert_round=cert_round)
            else:
                m = rc_message(w, height, rnd, a, raw=raw, cert=w.certificate(height, cert_round, raw, kind), cert_round=cert_round)
            for h in (stock, batch):
                assert h.store_add(m.encode()) == 0
    for view_round in (1, 2, 3):
        for accepted in (None, W.IbftMessage(view=W.View(height, view_round), sender=w.proposer(height, view_round), type=PP,
                                             payload=W.preprepare_body(W.Proposal(raw, view_round), fake_hash(raw, view_round), None))):
            for h in (stock, batch):
                h.set_state(height, view_round, accepted.encode() if accepted else None)
            w.calls = {"validator": 0, "hash": 0}
            a = stock.handle_round_change(height, view_round)
            stock_calls = dict(w.calls)
            before = batch.loop_batch_calls()
            b = batch.handle_round_change(height, view_round)
            assert sorted(a) == sorted(b), (seed, view_round, accepted is not None)
            # one sender batch + one hash batch per distinct carried proposal (raw, round): rounds 0, 1, 2 here
            assert batch.loop_batch_calls() - before <= 1 + 3
            assert batch.fallbacks() == 0
            if stock_calls["validator"] > 20:
                senders_batched, hashes_batched = batch.last_cert_batch()
                assert senders_batched >= 1 and hashes_batched >= 1
    stock.close(); batch.close()


def _commit_world(n=9, bad=(2, 5)):
    w = World(n, 1)
    raw = b"the block"
    hsh = fake_hash(raw, 0)
    proposal = W.IbftMessage(view=W.View(1, 0), sender=w.proposer(1, 0), type=PP,
                             payload=W.preprepare_body(W.Proposal(raw, 0), hsh, None))
    prepares = [W.IbftMessage(view=W.View(1, 0), sender=a, type=PR, payload=W.prepare_body(hsh if i not in bad else b"x" * 32))
                for i, a in enumerate(w.addrs) if a != proposal.sender]
    commits = [W.IbftMessage(view=W.View(1, 0), sender=a, type=CM,
                             payload=W.commit_body(hsh if i not in bad else b"y" * 32, b"seal-" + a)) for i, a in enumerate(w.addrs)]
    return w, proposal, prepares, commits


@pytest.mark.parametrize("fail_mask", [1, 2, 4, 7])
def test_failing_batch_backend_falls_back_to_the_per_message_verifier(fail_mask):
    w, proposal, prepares, commits = _commit_world()
    ver = w.verifier()
    ver["is_valid_committed_seal"] = lambda hsh, seal: seal is not None and not seal[1].endswith(b"03")
    hosts = []
    for mode in ("stock", "failing-batch"):
        h = H.Host()
        assert h.vm_init({a: 1 for a in w.addrs})
        h.set_verifier(**ver)
        h.set_state(1, 0, proposal.encode())
        if mode != "stock":
            h.use_loop_batch(fail_mask)
            h.use_batch(True)
        for m in prepares + commits:
            h.store_add(m.encode())
        hosts.append(h)
    stock, failing = hosts
    assert stock.handle_prepare(1, 0) == failing.handle_prepare(1, 0)
    qs, seals_s = stock.handle_commit(1, 0)
    qf, seals_f = failing.handle_commit(1, 0)
    assert qs == qf and sorted(seals_s) == sorted(seals_f)
    for t in (PR, CM):                                  # and the same messages were pruned from the store
        assert stock.store_num(1, 0, t) == failing.store_num(1, 0, t)
    if fail_mask & 3:
        assert failing.fallbacks() >= 1
    for h in hosts:
        h.close()


def test_ingest_micro_batches_and_the_verdict_cache():
    w, proposal, prepares, commits = _commit_world(n=10, bad=())
    forged = W.IbftMessage(view=W.View(1, 0), sender=commits[3].sender, type=CM, signature=b"sig",
                           payload=W.commit_body(b"z" * 32, b"another seal"))
    genuine = W.IbftMessage(view=W.View(1, 0), sender=commits[3].sender, type=CM, signature=b"sig",
                            payload=W.commit_body(fake_hash(b"the block", 0), b"seal-" + commits[3].sender))
    commits[3] = genuine
    w.bad_wires.add(forged.encode())                      # same From and Signature, different payload: invalid
    w.bad_wires.add(commits[7].encode())
    wires = [m.encode() for m in commits]
    ref, ing = w.host(), w.host()
    for h in (ref, ing):
        h.set_state(1, 0, proposal.encode())
    ing.use_loop_batch(0)
    ing.use_batch(True)
    ing.enable_quorum_index()
    expect = [ref.add_message(x) for x in wires]
    got, rows, hits, calls = ing.ingest_wire(wires + [b"\xff\xff"])
    assert got == expect + [-1] and (rows, hits, calls) == (len(wires), 0, 1)
    assert 0 in expect and 2 in expect                   # a rejected sender, and the quorum signal
    # gossip re-delivery: answered from the cache, no batch call; the store decisions repeat (same sender overwrites)
    again, rows, hits, calls = ing.ingest_wire(wires)
    assert (rows, hits, calls) == (0, len(wires), 0)
    assert [r != 0 for r in again] == [r != 0 for r in expect]
    # a replay of (From, Signature) under another payload is a different message: not a cache hit, judged, rejected
    got, rows, hits, calls = ing.ingest_wire([forged.encode(), wires[0], forged.encode()])
    assert got[0] == 0 and got[2] == 0 and got[1] != 0 and (rows, hits, calls) == (1, 1, 1)
    # validator set change: every cached verdict is dropped
    assert ing.vm_init({a: 1 for a in w.addrs})
    _, rows, hits, _ = ing.ingest_wire(wires[:4])
    assert (rows, hits) == (4, 0)
    # pruning the store below a height drops the cached verdicts of older messages
    ing.store_prune(2)
    _, rows, hits, _ = ing.ingest_wire(wires[:4])
    assert (rows, hits) == (4, 0)
    # no batch backend at all: the per-message verifier answers, results unchanged
    plain = w.host()
    plain.set_state(1, 0, proposal.encode())
    got, rows, hits, calls = plain.ingest_wire(wires)
    assert got == expect and calls == 0
    for h in (ref, ing, plain):
        h.close()


@pytest.mark.parametrize("mode", ["sets", "sets-fail", "sets-off"])
def test_messages_judged_completely_on_arrival(mode):
    """Message sets (ibft_verify_messages behind BatchVerifier::VerifyMessageSet): with the proposal of the view accepted,
    a PREPARE / COMMIT that arrives is judged completely — IsValidValidator AND the handlePrepare / handleCommit closure —
    by one set call per type and micro-batch; handlePrepare / handleCommit then decide like the stock walk WITHOUT another
    batch call.  A failing set call, or sets switched off, gives the same decisions through the older batches."""
    w, proposal, prepares, commits = _commit_world(n=13, bad=(2, 5))
    ver = w.verifier()
    ver["is_valid_committed_seal"] = lambda hsh, seal: seal is not None and not seal[1].endswith(b"03")
    w.bad_wires.add(commits[7].encode())                # a forged envelope: never stored
    w.bad_wires.add(prepares[4].encode())
    future = [W.IbftMessage(view=W.View(1, 1), sender=a, type=CM, payload=W.commit_body(b"f" * 32, b"seal-" + a))
              for a in w.addrs[:3]]                     # another round: sender check only
    ref, ing = H.Host(), H.Host()
    for h in (ref, ing):
        assert h.vm_init({a: 1 for a in w.addrs})
        h.set_verifier(**ver)
        h.set_state(1, 0, proposal.encode())
    ing.use_loop_batch(8 if mode == "sets-fail" else 0)
    ing.use_batch(True)
    ing.use_sets(mode != "sets-off")
    wires = [m.encode() for m in prepares + commits + future]
    random.Random(5).shuffle(wires)
    expect = [ref.add_message(x) for x in wires]
    got = []
    for k in range(0, len(wires), 7):                   # micro-batches, as the transport would hand them over
        res, rows, hits, calls = ing.ingest_wire(wires[k:k + 7])
        got += res
        if mode == "sets":
            assert calls <= 3 and ing.last_set_rows() + 3 >= rows       # ≤ one call per type + one for other views
        else:
            assert ing.last_set_rows() == 0
    assert [g != 0 for g in got] == [e != 0 for e in expect]
    assert ing.loop_batch_set_calls() > 0 if mode == "sets" else ing.loop_batch_set_calls() == 0
    # gossip re-delivery of everything: verdicts (both kinds) come from the cache, no batch call at all
    before = ing.loop_batch_calls()
    res, rows, hits, calls = ing.ingest_wire(wires)
    assert (rows, calls) == (0, 0) and ing.loop_batch_calls() == before
    for handle in ("handle_prepare", "handle_commit"):
        before = ing.loop_batch_calls()
        a, b = getattr(ref, handle)(1, 0), getattr(ing, handle)(1, 0)
        assert (a[0], sorted(a[1])) == (b[0], sorted(b[1])) if isinstance(a, tuple) else a == b
        stored = ing.store_num(1, 0, PR if handle == "handle_prepare" else CM)
        assert ref.store_num(1, 0, PR if handle == "handle_prepare" else CM) == stored   # same messages pruned
        if mode == "sets":
            assert ing.loop_batch_calls() == before and ing.closure_hits() >= stored
        else:
            assert ing.loop_batch_calls() == before + 1 and ing.closure_hits() == 0
    assert ing.fallbacks() == 0
    # a new round with another proposal: what the table said no longer applies — the walk asks the batch backend again
    raw2 = b"block two"
    proposal2 = W.IbftMessage(view=W.View(1, 1), sender=w.proposer(1, 1), type=PP,
                              payload=W.preprepare_body(W.Proposal(raw2, 1), fake_hash(raw2, 1), None))
    for h in (ref, ing):
        h.set_state(1, 1, proposal2.encode())
    before = ing.loop_batch_calls()
    a, b = ref.handle_commit(1, 1), ing.handle_commit(1, 1)
    assert a[0] == b[0] and ing.closure_hits() == 0 and ing.loop_batch_calls() == before + 1
    ref.close(); ing.close()


def test_flat_ingest_receive_side_memory_is_bounded_and_follows_the_store():
    """the receive side remembers only messages that AddMessage STORED (the entry is the stored message),
    rejected ones leave a fingerprint in a bounded FIFO, pruning the store prunes the memory WITHOUT the quorum index being
    enabled, and the tables have caps.  Through ibft_host_ingest_flat (rows back to back + offsets) ≡ ibft_host_ingest_wire."""
    import numpy as np
    w, proposal, prepares, commits = _commit_world(n=12, bad=())
    w.bad_wires.add(commits[5].encode())
    w.bad_wires.add(commits[9].encode())
    wires = [m.encode() for m in commits]
    flat = np.frombuffer(b"".join(wires), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(x) for x in wires])]).astype(np.uint32)
    a, b = w.host(), w.host()
    for h in (a, b):
        h.set_state(1, 0, proposal.encode())
        h.use_loop_batch(0)
        h.use_batch(True)                                 # NOTE: no enable_quorum_index()
    ra, rows_a, hits_a, calls_a = a.ingest_wire(wires)
    rb, (rows_b, hits_b, calls_b) = b.ingest_flat(flat, off, want_stats=True)
    assert list(rb) == [x & 0xFF for x in ra] and (rows_a, hits_a, calls_a) == (rows_b, hits_b, calls_b) == (12, 0, 1)
    assert ra[5] == 0 and ra[9] == 0 and b.seen_entries() == 10          # the two rejected ones are not remembered as messages
    before = b.loop_batch_calls()
    rb2, (rows, hits, calls) = b.ingest_flat(flat, off, want_stats=True)  # re-delivery: stored ones AND rejected ones answered
    assert (rows, hits, calls) == (0, 12, 0) and b.loop_batch_calls() == before
    assert [x != 0 for x in rb2] == [x != 0 for x in rb]
    # the rejected-FIFO is bounded: with room for one fingerprint, the older rejected message is judged again
    c = w.host()
    c.set_state(1, 0, proposal.encode()); c.use_loop_batch(0); c.use_batch(True)
    c.set_seen_caps(1 << 18, 1)
    c.ingest_wire(wires)
    _, rows, hits, _ = c.ingest_wire([wires[5], wires[9]])
    assert (rows, hits) == (1, 1)
    # the stored-message table is capped: at the cap it is dropped, re-deliveries are judged again (never wrongly answered)
    d = w.host()
    d.set_state(1, 0, proposal.encode()); d.use_loop_batch(0); d.use_batch(True)
    d.set_seen_caps(4, 16)
    rd, _, _, _ = d.ingest_wire(wires)
    assert [x != 0 for x in rd] == [x != 0 for x in ra] and d.seen_entries() <= 4
    rd2, rows, hits, _ = d.ingest_wire(wires)
    assert [x != 0 for x in rd2] == [x != 0 for x in ra] and rows + hits == 12 and rows >= 6
    # pruning the store drops what was remembered about older heights — the height hook is always installed
    b.store_prune(2)
    assert b.seen_entries() == 0
    for h in (a, b, c, d):
        h.close()


def test_ingest_queue_adaptive_batches_equal_direct_ingest():
    """SURVEY §8f rank 1 "queue → micro-batches": transport threads push, one worker ingests whatever is pending as one
    batch.  Same store, same signals as ibft_host_ingest_wire on the same messages; pushes from several threads; the
    SignalEvent callback fires from the worker and may call handle_* at once."""
    import threading
    import numpy as np
    w, proposal, prepares, commits = _commit_world(n=40, bad=(3, 11))
    w.bad_wires.add(commits[7].encode())
    wires = [m.encode() for m in prepares + commits] + [b"\xff\xff\xff"]
    ref, q = w.host(), w.host()
    for h in (ref, q):
        h.set_state(1, 0, proposal.encode())
        h.use_loop_batch(0)
        h.use_batch(True)
        h.enable_quorum_index()
    expect, *_ = ref.ingest_wire(wires)
    q.queue_start(max_rows=16, linger_us=200)
    signalled = []
    done = threading.Event()

    def on_signal(t, hh, rr):
        signalled.append((t, hh, rr))
        if t == CM and q.store_num(hh, rr, CM) > 0:         # a mirror call from the worker thread: its lock is not held
            done.set()
    q.queue_on_signal(on_signal)

    def producer(chunk):
        for k in range(0, len(chunk), 5):
            part = chunk[k:k + 5]
            flat = np.frombuffer(b"".join(part), dtype=np.uint8)
            off = np.concatenate([[0], np.cumsum([len(x) for x in part])]).astype(np.uint32)
            q.queue_push(flat, off)
    threads = [threading.Thread(target=producer, args=(wires[i::3],)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    st = q.queue_drain()
    assert st.pushed == st.ingested == len(wires) and st.undecodable == 1
    assert st.stored == sum(1 for x in expect if x > 0) and st.rejected == sum(1 for x in expect if x == 0)
    assert st.max_batch_rows <= 16 and st.batches >= len(wires) // 16
    assert st.signals[PR] > 0 and st.signals[CM] > 0 and (CM, 1, 0) in signalled
    for t in (PR, CM):
        assert q.store_num(1, 0, t) == ref.store_num(1, 0, t)
    assert done.wait(5)
    a, b = ref.handle_prepare(1, 0), q.handle_prepare(1, 0)
    assert (a[0], sorted(a[1])) == (b[0], sorted(b[1]))
    a, b = ref.handle_commit(1, 0), q.handle_commit(1, 0)
    assert (a[0], sorted(a[1])) == (b[0], sorted(b[1]))
    q.queue_stop()
    q.queue_start()                                          # restartable
    q.queue_push(np.frombuffer(wires[0], dtype=np.uint8), np.array([0, len(wires[0])], dtype=np.uint32))
    assert q.queue_drain().ingested == 1
    ref.close(); q.close()                                   # close() stops the worker


def test_a_forgery_prepared_to_collide_with_an_honest_message_does_not_silence_it():
    """Round-3 advice (medium): the receive side used to drop a message as "rejected before" on a match of its 128-bit
    fingerprint alone, and the fingerprint's multiply-fold mixing has a seed-independent collision: a 16-byte block equal to
    the two multiplier constants zeroes both lanes, whatever came before it.  An honest message that carries that block (a
    transaction inside the raw proposal) and a forgery that differs from it in an EARLIER byte then share a fingerprint under
    every seed; the forgery, sent first, is rejected — and the honest message was then dropped without being judged (the
    reference judges every delivery: core/ibft.go:1101-1123).  Rejections are now identified by Keccak-256 of the bytes."""
    import struct
    w = World(4, 11)
    block = struct.pack("<QQ", 0xE7037ED1A0B428DB, 0xA0761D6478BD642F)
    for filler in range(0, 16):                          # put the block at a 16-aligned offset of the message's bytes
        raw = b"t" * filler + block + b"tail of the block"
        hsh = fake_hash(raw, 0)
        m = W.IbftMessage(view=W.View(1, 0), sender=w.proposer(1, 0), type=PP, signature=b"sig-pp",
                          payload=W.preprepare_body(W.Proposal(raw, 0), hsh, None))
        honest = m.encode()
        at = honest.index(block)
        if at % 16 == 0:
            break
    assert at % 16 == 0 and len(honest) < 256
    forged = bytearray(honest)
    forged[honest.index(b"sig-pp") + 1] ^= 0x01          # one byte in front of the block: another message, same fingerprint
    forged = bytes(forged)
    w.bad_wires.add(forged)                              # its signature does not verify
    for first_forged in (True, False):
        h = w.host()
        h.set_state(1, 0, None)
        h.use_loop_batch(0)
        h.use_batch(True)
        if first_forged:
            assert h.ingest_wire([forged])[0] == [0]     # rejected, remembered
            res, asked, hits, calls = h.ingest_wire([forged])
            assert res == [0] and hits == 1 and asked == 0   # the SAME bytes again: answered from the memory
        res, asked, hits, calls = h.ingest_wire([honest])
        assert res[0] > 0 and hits == 0, "the honest message was not judged"
        assert h.store_num(1, 0, PP) == 1
        h.close()


def test_ingest_queue_is_bounded_and_applies_back_pressure():
    """Round-3 advice: the receive queue grew for as long as the worker was inside a device call (32-bit offsets, memory).
    Now: a push that would pass the caps WAITS until the worker has taken what is pending; a push that can never fit is
    refused.  A slow verifier stands in for a stalled device."""
    import threading
    import time
    import numpy as np
    w, proposal, prepares, commits = _commit_world(n=30, bad=())
    ver = w.verifier()
    slow = ver["is_valid_validator"]

    def slow_validator(wire):
        time.sleep(0.002)
        return slow(wire)
    ver["is_valid_validator"] = slow_validator
    q = H.Host()
    assert q.vm_init({a: 1 for a in w.addrs})
    q.set_verifier(**ver)
    q.set_state(1, 0, proposal.encode())
    q.queue_start(max_rows=8)
    wires = [m.encode() for m in prepares + commits]
    one = max(len(x) for x in wires)
    q.queue_set_caps(max_pending_bytes=4 * one, max_pending_rows=4)
    big = np.frombuffer(b"".join(wires[:6]), dtype=np.uint8)
    assert q.queue_try_push(big, np.concatenate([[0], np.cumsum([len(x) for x in wires[:6]])]).astype(np.uint32)) == -2
    t0 = time.perf_counter()

    def producer(chunk):
        for x in chunk:
            q.queue_push(np.frombuffer(x, dtype=np.uint8), np.array([0, len(x)], dtype=np.uint32))
    threads = [threading.Thread(target=producer, args=(wires[i::2],)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    pushed_in = time.perf_counter() - t0
    st = q.queue_drain()
    assert st.pushed == st.ingested == len(wires) and st.stored == len(wires)
    assert st.max_batch_rows <= 4                            # never more pending than the cap
    assert q.queue_backpressure_waits > 0 and pushed_in > 0.002 * (len(wires) - 8)   # the pushers were held back
    q.close()
