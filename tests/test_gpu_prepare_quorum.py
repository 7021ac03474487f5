"""GPU: ValidatorManager.HasPrepareQuorum on the device (/root/reference/core/validator_manager.go:99-127 — the decision
hasQuorumByMsgType takes for PREPARE messages, core/ibft.go:1273-1284): the proposer's address joins the sender set, a
valid PREPARE sent BY the proposer voids the quorum.  ibft_tally_prepare, the proposer20 argument of ibft_verify_messages /
ibft_verify_messages_wire / ibft_group_verify_messages, and the sharded merge, each against oracle/semantics.py (Python big
ints restating the Go lines)."""
from collections import namedtuple

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Msg = namedtuple("Msg", "sender")


def _addr(name: str) -> bytes:
    return name.encode() * 20


def _col(names):
    return np.frombuffer(b"".join(_addr(n) for n in names) or b"", np.uint8).reshape(-1, 20)


def _expect(vm, proposer: bytes, senders, verdict):
    """HasPrepareQuorum(state, proposalMessage{From: proposer}, the messages whose verdict bit is set)"""
    msgs = [Msg(bytes(s)) for s, v in zip(senders, verdict) if v]
    members = set(vm.power)
    distinct = ({bytes(proposer)} | {m.sender for m in msgs}) & members
    return (vm.has_prepare_quorum(Msg(bytes(proposer)), msgs), sum(vm.power[a] for a in distinct), len(distinct),
            sum(1 for m in msgs if m.sender == bytes(proposer)))


# the proposer rule on the reference's own quorum table (core/validator_manager_test.go:18-187): signers minus the
# proposer send PREPAREs; then the proposer's own PREPARE joins and voids
CASES = [
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, 'A', ['B', 'C']),            # 3 of 4 with the proposer's seat: quorum
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, 'A', ['B']),                 # 2 of 4: none
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, 'A', ['A', 'B', 'C']),       # the proposer among the signers: void (:114-121)
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, 'A', ['B', 'C', 'D', 'B']),  # a sender twice: counted once
    ({'A': 2, 'B': 2, 'C': 2, 'D': 3}, 'D', ['A', 'C']),            # weighted: 7 ≥ ⌊2·9/3⌋+1 = 7
    ({'A': 2, 'B': 2, 'C': 2, 'D': 3}, 'A', ['B', 'C']),            # 6 < 7
    ({'A': 2, 'B': 7, 'C': 7, 'D': 5}, 'Z', ['A', 'B', 'C']),       # the proposer is no validator: adds nothing (:88-92)
    ({'A': 2, 'B': 7, 'C': 7, 'D': 5}, 'Z', ['B', 'C', 'Z']),       # … and his own PREPARE still voids (bytes.Equal, :115)
    ({'A': 1}, 'A', []),                                            # one validator: the proposer alone is the quorum
    ({'A': 1, 'B': 1}, 'A', []),                                    # … not with two
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, 'A', ['B', 'C', 'Q', 'R']),  # unknown senders contribute 0
]


@pytest.mark.parametrize("scale", [1, 10**18 * 2**64 + 7])
def test_tally_prepare_rule_table(scale):
    import go_ibft_amd.verifier as V
    from oracle.semantics import ValidatorManager
    bv = V.BatchVerifier(max_rows=1024)
    try:
        for powers, proposer, signers in CASES:
            names = sorted(powers)
            vm = ValidatorManager()
            assert vm.init({_addr(n): powers[n] * scale for n in names})
            (bv.set_validators if scale == 1 else bv.set_validators_u256)(1, _col(names), [powers[n] * scale for n in names])
            send = _col(signers)
            ones = np.ones(len(signers), bool)
            want, power, distinct, prows = _expect(vm, _addr(proposer), [bytes(s) for s in send], ones)
            t = bv.has_prepare_quorum(send, ones, _addr(proposer))
            w = bv.last_tally_wide()
            assert (bool(t.has_quorum), w.power, t.distinct_senders, t.proposer_rows, t.valid_rows) == \
                   (want, power, distinct, prows, len(signers)), (powers, proposer, signers)
            # plain HasQuorum on the same rows is a different question
            t0 = bv.has_quorum(send, ones)
            assert bool(t0.has_quorum) == vm.has_quorum([bytes(s) for s in send]) and t0.proposer_rows == 0
    finally:
        bv.close()


@pytest.mark.parametrize("n,wide", [(300, False), (4096, False), (5000, True), (20000, False), (70000, True)])
def test_tally_prepare_random_against_the_big_int_rule(n, wide):
    """one workgroup and the multi-workgroup ticket form (n > 4 096), u64 and 256-bit powers: random masks with repeated
    senders and non-members; the proposer a member / an outsider, his rows present and valid / present and masked out"""
    import go_ibft_amd.verifier as V
    from oracle.semantics import ValidatorManager
    rng = np.random.default_rng(n)
    nv = max(8, n // 2)
    addrs = rng.integers(0, 256, size=(nv, 20), dtype=np.uint8)
    outsiders = rng.integers(0, 256, size=(16, 20), dtype=np.uint8)
    powers = [int(x) * (10**20 * 2**70 + 3 if wide else 1) for x in rng.integers(1, 1000, size=nv)]
    vm = ValidatorManager()
    assert vm.init({bytes(addrs[i]): powers[i] for i in range(nv)})
    bv = V.BatchVerifier(max_rows=max(n, 1024))
    try:
        (bv.set_validators_u256 if wide else bv.set_validators)(1, addrs, powers)
        for trial in range(8):
            frac = [0.9, 0.75, 0.66, 0.5][trial % 4]
            pick = rng.integers(0, nv, size=n)
            send = addrs[pick].copy()
            out_rows = rng.choice(n, size=n // 50 + 1, replace=False)
            send[out_rows] = outsiders[rng.integers(0, 16, size=len(out_rows))]
            verdict = rng.random(n) < frac
            proposer = bytes(addrs[int(rng.integers(0, nv))]) if trial % 3 else bytes(outsiders[trial % 16])
            mine = np.flatnonzero((send == np.frombuffer(proposer, np.uint8)).all(axis=1))
            if trial % 2 == 0:
                verdict[mine] = False            # the proposer's PREPAREs were pruned by the walk: no void
            elif len(mine) == 0:
                send[trial] = np.frombuffer(proposer, np.uint8)
                verdict[trial] = True
            want, power, distinct, prows = _expect(vm, proposer, [bytes(s) for s in send], verdict)
            t = bv.has_prepare_quorum(send, verdict, proposer)
            w = bv.last_tally_wide()
            assert (bool(t.has_quorum), w.power, t.distinct_senders, t.proposer_rows, t.valid_rows) == \
                   (want, power, distinct, prows, int(verdict.sum())), (trial, frac)
            assert bool(t.has_quorum) == (w.power >= w.quorum and prows == 0)
    finally:
        bv.close()


def _prepare_set(n, seed, proposer_idx):
    """a PREPARE set of validators 0..n−1 except the proposer, real envelope signatures (the oracle's signer)"""
    from oracle import workload as W, wire, binding as B
    r = W.make_round(n, seed, weighted=True)
    rows = [i for i in range(n) if i != proposer_idx]
    chunks, sigs = [], []
    for i in rows:
        m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.PREPARE,
                             payload=wire.prepare_body(r.proposal_hash))
        pns = m.payload_no_sig()
        chunks.append(pns)
        sigs.append(np.frombuffer(B.sign(r.sks[i], B.keccak256(pns)), np.uint8))
    return r, rows, chunks, sigs


def _pack(chunks):
    off = np.concatenate([[0], np.cumsum([len(c) for c in chunks])]).astype(np.uint32)
    return b"".join(chunks), off


@pytest.mark.parametrize("n,flags", [(4, 0), (100, 0), (1000, 2), (4096, 0)])
def test_prepare_set_with_the_proposer_rule(oracle, n, flags):
    """ibft_verify_messages(seal65 = NULL, proposer20): the tally is HasPrepareQuorum over the rows both verdicts accept.
    Three sets: exactly short of the quorum without the proposer's seat; the proposer's own (correctly signed) PREPARE
    among them → void; the proposer's PREPARE with a forged envelope → pruned by the sender verdict, no void."""
    import go_ibft_amd.verifier as V
    from oracle import wire, binding as B
    from oracle.semantics import ValidatorManager
    p = n // 3
    r, rows, chunks, sigs = _prepare_set(n, 4100 + n, p)
    vm = ValidatorManager()
    assert vm.init({bytes(r.addrs[i]): int(r.power[i]) for i in range(n)})
    proposer = r.addrs[p].tobytes()
    # the proposer's own PREPARE (what a faulty proposer would multicast)
    pm = wire.IbftMessage(view=wire.View(r.height, r.round), sender=proposer, type=wire.PREPARE,
                          payload=wire.prepare_body(r.proposal_hash)).payload_no_sig()
    psig = np.frombuffer(B.sign(r.sks[p], B.keccak256(pm)), np.uint8)
    bv = V.BatchVerifier(max_rows=max(n, 256), flags=flags)
    try:
        bv.set_validators(r.height, r.addrs, r.power)

        def run(chs, sgs, froms):
            payload, off = _pack(chs)
            k = len(chs)
            h = np.tile(np.frombuffer(r.proposal_hash, np.uint8), (k, 1))
            for rep in range(3 if flags else 1):
                s, v, t = bv.verify_messages(payload, off, np.array(sgs).reshape(-1, 65), np.array(froms).reshape(-1, 20), h,
                                             np.full(k, 32, np.uint8), raw=r.raw, round_=r.round, proposer=proposer)
            s0, v0, t0 = bv.verify_messages(payload, off, np.array(sgs).reshape(-1, 65), np.array(froms).reshape(-1, 20), h,
                                            np.full(k, 32, np.uint8), raw=r.raw, round_=r.round)
            assert (s0 == s).all() and (v0 == v).all() and t0.proposer_rows == 0
            return s, v, t, t0
        froms = [r.addrs[i] for i in rows]
        # (1) everybody but the proposer: with his seat the whole power is there
        s, v, t, t0 = run(chunks, sigs, froms)
        assert s.all() and v.all()
        want, power, distinct, prows = _expect(vm, proposer, [bytes(f) for f in froms], s & v)
        assert (bool(t.has_quorum), t.power, t.distinct_senders, t.proposer_rows) == (want, power, distinct, prows) == \
               (True, int(r.power.sum()), n, 0)
        assert t0.power == power - int(r.power[p]) and t0.distinct_senders == n - 1
        # (2) exactly the senders for which the seat decides: drop senders until HasQuorum alone fails
        keep = len(rows)
        order = sorted(range(len(rows)), key=lambda j: -int(r.power[rows[j]]))
        acc = 0
        for cnt, j in enumerate(order):
            acc += int(r.power[rows[j]])
            if acc + int(r.power[p]) >= vm.quorum:
                keep = cnt + 1
                break
        sel = order[:keep]
        s, v, t, t0 = run([chunks[j] for j in sel], [sigs[j] for j in sel], [froms[j] for j in sel])
        want, power, distinct, prows = _expect(vm, proposer, [bytes(froms[j]) for j in sel], s & v)
        assert (bool(t.has_quorum), t.power, t.distinct_senders, t.proposer_rows) == (want, power, distinct, prows)
        assert want and (bool(t0.has_quorum) == (acc >= vm.quorum))
        # (3) the proposer's own PREPARE, correctly signed, among all the others: void
        s, v, t, _ = run(chunks + [pm], sigs + [psig], froms + [r.addrs[p]])
        assert s.all() and v.all()
        assert (t.has_quorum, t.proposer_rows, t.power, t.distinct_senders) == (0, 1, int(r.power.sum()), n)
        # (4) the same message with a forged envelope: IsValidValidator rejects it, the walk prunes it, nothing is voided
        bad = psig.copy()
        bad[5] ^= 0x40
        s, v, t, _ = run(chunks + [pm], sigs + [bad], froms + [r.addrs[p]])
        assert not s[-1] and s[:-1].all()
        assert (t.has_quorum, t.proposer_rows, t.power) == (1, 0, int(r.power.sum()))
    finally:
        bv.close()


def test_prepare_set_from_wire_bytes_with_the_proposer_rule(oracle):
    """ibft_verify_messages_wire(proposer20): raw PREPARE messages of the view → HasPrepareQuorum; an empty batch asks
    whether the proposer alone is a quorum"""
    import go_ibft_amd.verifier as V
    from oracle import wire, binding as B
    from oracle.semantics import ValidatorManager
    n, p = 64, 5
    r, rows, chunks, sigs = _prepare_set(n, 4300, p)
    vm = ValidatorManager()
    assert vm.init({bytes(r.addrs[i]): int(r.power[i]) for i in range(n)})
    proposer = r.addrs[p].tobytes()

    def raw_msg(i, sig=None):
        m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.PREPARE,
                             payload=wire.prepare_body(r.proposal_hash))
        m.signature = bytes(sig) if sig is not None else B.sign(r.sks[i], B.keccak256(m.payload_no_sig()))
        return m.encode()
    bv = V.BatchVerifier(max_rows=256)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        msgs = [raw_msg(i) for i in rows]
        buf, off = _pack(msgs)
        s, v, cls, t = bv.verify_messages_wire(buf, off, r.height, r.round, raw=r.raw, want_rows=False, proposer=proposer)
        assert s.all() and v.all() and (cls & V.WIRE_CLASS_CLOSURE).all()
        assert (t.has_quorum, t.power, t.distinct_senders, t.proposer_rows) == (1, int(r.power.sum()), n, 0)
        buf, off = _pack(msgs + [raw_msg(p)])
        s, v, cls, t = bv.verify_messages_wire(buf, off, r.height, r.round, raw=r.raw, want_rows=False, proposer=proposer)
        assert s.all() and v.all() and (t.has_quorum, t.proposer_rows) == (0, 1)
        # half of them: the seat counts, the quorum does not follow
        half = msgs[: n // 2]
        buf, off = _pack(half)
        s, v, cls, t = bv.verify_messages_wire(buf, off, r.height, r.round, raw=r.raw, want_rows=False, proposer=proposer)
        want, power, distinct, prows = _expect(vm, proposer, [bytes(r.addrs[i]) for i in rows[: n // 2]], s & v)
        assert (bool(t.has_quorum), t.power, t.distinct_senders, t.proposer_rows) == (want, power, distinct, prows)
        # no PREPARE at all
        s, v, cls, t = bv.verify_messages_wire(b"", np.zeros(1, np.uint32), r.height, r.round, raw=r.raw, want_rows=False,
                                               proposer=proposer)
        assert (t.has_quorum, t.power, t.distinct_senders, t.valid_rows) == (0, int(r.power[p]), 1, 0)
        bv.set_validators(r.height, r.addrs[p:p + 1], r.power[p:p + 1])    # a set of one: the proposer alone decides
        s, v, cls, t = bv.verify_messages_wire(b"", np.zeros(1, np.uint32), r.height, r.round, raw=r.raw, want_rows=False,
                                               proposer=proposer)
        assert (t.has_quorum, t.distinct_senders) == (1, 1)
    finally:
        bv.close()


@pytest.mark.parametrize("world,n", [(2, 130), (3, 500), (4, 1000), (8, 700), (8, 4096)])
def test_sharded_prepare_set_with_the_proposer_rule(oracle, world, n):
    """ibft_group_verify_messages(proposer20) on W contexts of one device: every rank counts the proposer's rows of its own
    shard, the seat joins the MERGED bitmap, a proposer row in any shard voids — ≡ the one-device call ≡ the big-int rule"""
    import go_ibft_amd.verifier as V
    import go_ibft_amd.shard as S
    from oracle import wire, binding as B
    from oracle.semantics import ValidatorManager
    p = 1
    r, rows, chunks, sigs = _prepare_set(n, 4500 + n + world, p)
    vm = ValidatorManager()
    assert vm.init({bytes(r.addrs[i]): int(r.power[i]) for i in range(n)})
    proposer = r.addrs[p].tobytes()
    pm = wire.IbftMessage(view=wire.View(r.height, r.round), sender=proposer, type=wire.PREPARE,
                          payload=wire.prepare_body(r.proposal_hash)).payload_no_sig()
    psig = np.frombuffer(B.sign(r.sks[p], B.keccak256(pm)), np.uint8)
    froms = [r.addrs[i] for i in rows]
    # rows repeated across the first seam (same sender in two shards), a third of the senders silent
    k = len(rows) * 2 // 3
    lo1 = S.shard_range(k + 8, 1, world)[0]
    chs, sgs, frs = chunks[:k], sigs[:k], froms[:k]
    for j in range(4):
        if 0 < lo1 < k:
            chs.insert(lo1, chunks[j]); sgs.insert(lo1, sigs[j]); frs.insert(lo1, froms[j])
    one = V.BatchVerifier(max_rows=max(n + 16, 256))
    g = V.DeviceGroup([0] * world, max_rows_total=max(n + 16, 64 * world))
    try:
        one.set_validators(r.height, r.addrs, r.power)
        g.set_validators(r.height, r.addrs, r.power)
        for variant in ("without", "proposer_in_last_shard", "proposer_in_first_shard"):
            c2, s2, f2 = list(chs), list(sgs), list(frs)
            if variant == "proposer_in_last_shard":
                c2.append(pm); s2.append(psig); f2.append(r.addrs[p])
            elif variant == "proposer_in_first_shard":
                c2.insert(2, pm); s2.insert(2, psig); f2.insert(2, r.addrs[p])
            payload, off = _pack(c2)
            m = len(c2)
            h = np.tile(np.frombuffer(r.proposal_hash, np.uint8), (m, 1))
            args = (payload, off, np.array(s2).reshape(-1, 65), np.array(f2).reshape(-1, 20), h, np.full(m, 32, np.uint8))
            s1, v1, t1 = one.verify_messages(*args, raw=r.raw, round_=r.round, proposer=proposer)
            want, power, distinct, prows = _expect(vm, proposer, [bytes(f) for f in f2], s1 & v1)
            assert s1.all() and v1.all()
            assert (bool(t1.has_quorum), t1.power, t1.distinct_senders, t1.proposer_rows) == (want, power, distinct, prows)
            for rep in range(2):
                s, v, t = g.verify_messages(*args, raw=r.raw, round_=r.round, proposer=proposer)
                assert (s == s1).all() and (v == v1).all()
                assert (t.has_quorum, t.power, t.distinct_senders, t.proposer_rows, t.valid_rows) == \
                       (t1.has_quorum, t1.power, t1.distinct_senders, t1.proposer_rows, t1.valid_rows), variant
            assert prows == (0 if variant == "without" else 1)
            # … and without the rule the same rows give plain HasQuorum on both routes
            s, v, t = g.verify_messages(*args, raw=r.raw, round_=r.round)
            _, _, t0 = one.verify_messages(*args, raw=r.raw, round_=r.round)
            assert (t.has_quorum, t.power, t.distinct_senders, t.proposer_rows) == \
                   (t0.has_quorum, t0.power, t0.distinct_senders, 0)
    finally:
        g.close()
        one.close()


def test_tally_prepare_on_a_group_member_context_keeps_the_seat():
    """round-4 advice (medium): a context that is a member of a group (or of a communicator) dropped the proposer's seat in
    EVERY tally, also in ibft_tally_prepare and the one-shot set calls that no exchange follows — HasPrepareQuorum then
    under-counted against validator_manager.go:111-126 (3f+1 equal validators, f silent: the PREPARE quorum never came).
    The seat is left to the merge only by the sharded call itself."""
    import ctypes as C
    import go_ibft_amd.verifier as V
    from oracle.semantics import ValidatorManager
    names = ['A', 'B', 'C', 'D']
    vm = ValidatorManager()
    assert vm.init({_addr(n): 1 for n in names})
    g = V.DeviceGroup([0, 0], max_rows_total=256)
    try:
        g.set_validators(1, _col(names), [1, 1, 1, 1])
        L = g._L
        for rank in range(2):
            ctx = L.ibft_group_ctx(g._g, rank)
            for signers in (['B', 'C'], ['B'], ['A', 'B', 'C'], []):
                send = _col(signers)
                ones = np.ones(len(signers), bool)
                want, power, distinct, prows = _expect(vm, _addr('A'), [bytes(s) for s in send], ones)
                m = V.bool_to_mask(ones)
                pr = np.frombuffer(_addr('A'), np.uint8)
                t = V.Tally()
                assert L.ibft_tally_prepare(ctx, V._p(np.ascontiguousarray(send)), V._p(m), len(signers), V._p(pr), C.byref(t)) == 0
                assert (bool(t.has_quorum), t.power, t.distinct_senders, t.proposer_rows) == (want, power, distinct, prows), \
                    (rank, signers)
        # the sharded call on the same group still merges the seat exactly once
        r, rows, chunks, sigs = _prepare_set(130, 4711, 1)
        g.set_validators(r.height, r.addrs, r.power)
        payload, off = _pack(chunks)
        m = len(chunks)
        h = np.tile(np.frombuffer(r.proposal_hash, np.uint8), (m, 1))
        s, v, t = g.verify_messages(payload, off, np.array(sigs).reshape(-1, 65), np.array([r.addrs[i] for i in rows]).reshape(-1, 20),
                                    h, np.full(m, 32, np.uint8), raw=r.raw, round_=r.round, proposer=r.addrs[1].tobytes())
        assert s.all() and v.all() and (t.has_quorum, t.power, t.distinct_senders, t.proposer_rows) == (1, int(r.power.sum()), 130, 0)
    finally:
        g.close()


def test_wire_batch_of_both_types_only_a_prepare_of_the_proposer_voids(oracle):
    """ibft_verify_messages_wire(proposer20) over raw PREPAREs AND the proposer's own COMMIT: HasPrepareQuorum walks PREPARE
    messages, so the COMMIT is no PREPARE of his — proposer_rows stays 0 (round-4 advice, low)"""
    import go_ibft_amd.verifier as V
    from oracle import wire, binding as B
    n, p = 16, 3
    r, rows, chunks, sigs = _prepare_set(n, 4800, p)
    proposer = r.addrs[p].tobytes()

    def raw(i, type_, body):
        m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=type_, payload=body)
        m.signature = B.sign(r.sks[i], B.keccak256(m.payload_no_sig()))
        return m.encode()
    prepares = [raw(i, wire.PREPARE, wire.prepare_body(r.proposal_hash)) for i in rows]
    commit_p = raw(p, wire.COMMIT, wire.commit_body(r.proposal_hash, B.sign(r.sks[p], r.proposal_hash)))
    prepare_p = raw(p, wire.PREPARE, wire.prepare_body(r.proposal_hash))
    bv = V.BatchVerifier(max_rows=256)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        buf, off = _pack(prepares + [commit_p])
        s, v, cls, t = bv.verify_messages_wire(buf, off, r.height, r.round, raw=r.raw, want_rows=False, proposer=proposer)
        assert s.all() and v.all()
        assert (t.has_quorum, t.proposer_rows, t.distinct_senders) == (1, 0, n)
        buf, off = _pack(prepares + [commit_p, prepare_p])
        s, v, cls, t = bv.verify_messages_wire(buf, off, r.height, r.round, raw=r.raw, want_rows=False, proposer=proposer)
        assert s.all() and v.all() and (t.has_quorum, t.proposer_rows) == (0, 1)
    finally:
        bv.close()
