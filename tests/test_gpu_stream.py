"""GPU: two staging slots per context (ibft_seals_stage_next / ibft_seals_swap, include/ibftgpu.h) — a sustained stream of COMMIT
sets, a new batch at every wake-up (/root/reference/core/ibft.go:931-946): the copy of batch k+1 overlaps the verdict
kernels of batch k.  Parity: alternating batches of different content and different sizes, every verdict / tally equal to the
CPU oracle's for THAT batch — nothing of batch k leaks into k+1, kernels already enqueued keep reading the columns they were
launched on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batches(n_validators):
    from oracle import workload as W, binding as B
    r = W.make_round(n_validators, 9100 + n_validators, byzantine=True, weighted=True)
    vs = B.ValSet(r.addrs, r.power)
    cols = (r.hash32, r.seal65, r.signer20, r.pre_flags)
    out = []
    for k, (lo, hi, rev) in enumerate([(0, n_validators, False), (0, n_validators // 2, False), (n_validators // 3, n_validators, True),
                                       (0, 1, False), (5, n_validators - 7, True)]):
        sel = np.arange(lo, hi)[::-1] if rev else np.arange(lo, hi)
        h, s, f, p = (np.ascontiguousarray(c[sel]) for c in cols)
        if k == 2:
            p = None                                        # a batch without pre-flags behind one with them
            e = B.verify_seals(vs, h, s, f, None)
        else:
            e = B.verify_seals(vs, h, s, f, p)
        out.append((h, s, f, p, e.astype(bool), B.tally(vs, f, e)))
    return r, out


@pytest.mark.parametrize("n,pinned", [(300, True), (4096, True), (4096, False), (9000, True)])
def test_stream_of_batches_through_two_slots_equals_the_oracle(oracle, n, pinned):
    import go_ibft_amd.verifier as V
    r, batches = _batches(n)
    col = (lambda a: None if a is None else V.pinned_copy(a)) if pinned else (lambda a: a)
    host = [(col(h), col(s), col(f), col(p)) for h, s, f, p, _, _ in batches]
    bv = V.BatchVerifier(max_rows=max(n, 1024))
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        bv.seals_stage(*host[0])
        order = [0, 1, 2, 3, 4, 0, 2, 1, 4, 4, 3, 0]
        for step, (cur, nxt) in enumerate(zip(order, order[1:] + [0])):
            bv.seals_launch(1)                    # batch `cur` is resident
            bv.seals_stage_next(*host[nxt])       # its successor travels meanwhile
            verdict, t = bv.seals_fetch()
            _, _, _, _, want, wt = batches[cur]
            assert len(verdict) == len(want) and (verdict == want).all(), (step, cur)
            assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == (wt.power, wt.valid_rows, wt.distinct_senders, wt.has_quorum)
            bv.seals_swap()
        # the resident batch after the last swap is batch 0 again; the one-shot call still works on the same context
        verdict, t = bv.seals_run()
        assert (verdict == batches[0][4]).all()
        h, s, f, p, want, wt = batches[1]
        v2, t2 = bv.is_valid_committed_seal(h, s, f, p)
        assert (v2 == want).all() and t2.power == wt.power
    finally:
        bv.close()


def test_a_kernel_enqueued_before_the_swap_keeps_its_columns_and_the_swap_needs_a_staged_batch(oracle):
    import go_ibft_amd.verifier as V
    r, batches = _batches(2048)
    a, b = batches[0], batches[2]
    bv = V.BatchVerifier(max_rows=2048)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        with pytest.raises(RuntimeError):
            bv.seals_swap()                       # nothing staged
        bv.seals_stage(*a[:4])
        pa = [V.pinned_copy(x) for x in b[:3]]
        for _ in range(5):
            bv.seals_launch(1)                    # reads slot A … (not fetched yet)
            bv.seals_stage_next(pa[0], pa[1], pa[2], None)
            bv.seals_swap(wait_for_copy=False)    # … while the swap already points the context at slot B
            va, ta = bv.seals_fetch()
            assert (va == a[4]).all() and ta.power == a[5].power
            vb, tb = bv.seals_run()
            assert (vb == b[4]).all() and tb.power == b[5].power
            # and back: A through the spare slot
            bv.seals_stage_next(*a[:4])
            bv.seals_swap()
    finally:
        bv.close()


def test_pipelined_passes_deliver_every_pass_in_order_through_two_result_slots(oracle):
    """ibft_seals_submit / ibft_seals_collect: one pass kept in flight, every collect returns the OLDEST pass — over batches that
    change under the pipeline (stage_next / swap between submits) every pass's verdict words and tally are those of the batch it
    ran over; three submits are refused, a collect with nothing submitted too."""
    import go_ibft_amd.verifier as V
    r, batches = _batches(4096)
    host = [tuple(None if a is None else V.pinned_copy(a) for a in b[:4]) for b in batches]
    bv = V.BatchVerifier(max_rows=4096)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        with pytest.raises(RuntimeError):
            bv.seals_collect()
        order = [0, 1, 2, 4, 3, 0, 2, 2, 1, 4]
        bv.seals_stage(*host[order[0]])
        expected = []
        for i, cur in enumerate(order):
            bv.seals_submit()                                  # pass over batch `cur`
            expected.append(cur)
            # ibft_seals_rows (round 6): the LIBRARY's own counts — the resident batch, and the oldest pass not yet collected —
            # are what a binding sizes its verdict buffers from, whatever was swapped in between
            assert bv.seals_rows() == (len(batches[cur][4]), len(batches[expected[0]][4]))
            if i + 1 < len(order):
                bv.seals_stage_next(*host[order[i + 1]])
            if len(expected) == 2:
                with pytest.raises(RuntimeError):
                    bv.seals_submit()                          # a third pass in flight is refused
                verdict, t = bv.seals_collect()
                want, wt = batches[expected.pop(0)][4:6]
                assert len(verdict) == len(want) and (verdict == want).all(), i
                assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == (wt.power, wt.valid_rows, wt.distinct_senders, wt.has_quorum)
            if i + 1 < len(order):
                bv.seals_swap()
        verdict, t = bv.seals_collect()
        want, wt = batches[expected.pop(0)][4:6]
        assert (verdict == want).all() and t.power == wt.power
        assert bv.seals_rows() == (len(batches[order[-1]][4]), 0)      # nothing in flight
        # the synchronous calls still work next to the pipeline, warm path included
        verdict, t = bv.seals_run()
        assert (verdict == batches[order[-1]][4]).all()
    finally:
        bv.close()
    wv = V.BatchVerifier(max_rows=4096, flags=V.FLAG_PUBKEY_CACHE)
    try:
        wv.set_validators(r.height, r.addrs, r.power)
        wv.seals_stage(*[x for x in batches[0][:4]])
        for _ in range(3):                                     # learn, build, warm — through the pipeline
            wv.seals_submit(); wv.seals_submit()
            for _ in range(2):
                verdict, t = wv.seals_collect()
                assert (verdict == batches[0][4]).all() and t.power == batches[0][5].power
        assert wv.cache_stats()[0] > 0
    finally:
        wv.close()


def test_device_canary_reads_a_healthy_issue_time():
    """ibft_issue_probe (round 6): wall ns per aligned 8-byte VALU instruction per SIMD with one wavefront per SIMD — 1.89 through
    this probe on a healthy MI355X (profiles/r06*_kernel_ab.txt); the bench line flags a device more than 5 % off.  Here: a sane
    number (the probe really ran one wavefront per SIMD: the first version, packed four workgroups to a compute unit, read 2.53),
    stable across calls, and the context still verifies afterwards."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    bv = V.BatchVerifier(max_rows=1024)
    try:
        ns = [bv.issue_probe()[0] for _ in range(3)]
        assert all(1.6 < x < 2.3 for x in ns), ns
        assert max(ns) - min(ns) < 0.1, ns
        r = W.make_round(64, 77)
        bv.set_validators(1, r.addrs, r.power)
        got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)
        assert got.all() and t.has_quorum == 1
    finally:
        bv.close()


@pytest.mark.parametrize("n", [300, 4096, 20000])
@pytest.mark.parametrize("side", ["1", "0"])
def test_side_stream_tally_delivers_the_same_passes_and_other_calls_join_it(oracle, monkeypatch, n, side):
    """Round 6: the tally of a submitted pass runs on a stream of its own, next to the verdict kernel of the next pass, on a second
    pair of work mask / validator-index buffers (ibftgpu.hip: ibft_seals_submit).  Batches of different content and size change
    under the pipeline, and between the pipeline's calls come calls that are NOT part of it — a synchronous pass, a one-shot
    ibft_verify_seals over other rows, a plain tally — which share the tally's buffers and must find them finished with: every
    verdict word and every tally equals the CPU oracle's for ITS batch (HasQuorum, /root/reference/core/validator_manager.go:77-96).
    IBFT_SIDE_TALLY=0 (one stream, as until round 5) is held to the same answers."""
    import go_ibft_amd.verifier as V
    from oracle import binding as B
    monkeypatch.setenv("IBFT_SIDE_TALLY", side)
    r, batches = _batches(n)
    vs = B.ValSet(r.addrs, r.power)
    host = [tuple(None if a is None else V.pinned_copy(a) for a in b[:4]) for b in batches]
    bv = V.BatchVerifier(max_rows=max(n, 1024))
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        order = [0, 2, 1, 4, 0, 3, 2, 4, 1, 0, 2, 2]
        bv.seals_stage(*host[order[0]])
        in_flight = []

        def check(got, idx):
            verdict, t = got
            want, wt = batches[idx][4:6]
            assert len(verdict) == len(want) and (verdict == want).all(), idx
            assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == (wt.power, wt.valid_rows, wt.distinct_senders, wt.has_quorum), idx

        for i, cur in enumerate(order):
            bv.seals_submit()
            in_flight.append(cur)
            if i % 4 == 1:      # a synchronous pass over the resident batch while two pipelined passes are in flight
                check(bv.seals_run(), cur)
            if i % 4 == 2:      # a one-shot call over OTHER rows: restages the columns, runs verdict kernel + tally on the main stream
                h, s, f, p = batches[(cur + 1) % 5][:4]
                check(bv.is_valid_committed_seal(h, s, f, p), (cur + 1) % 5)
                bv.seals_stage(*host[cur])
            if i % 4 == 3:      # a tally of its own over a caller's verdicts (ibft_tally)
                e = batches[cur][4]
                t = bv.has_quorum(batches[cur][2], e)
                te = B.tally(vs, batches[cur][2], e.astype(np.uint8))
                assert (t.power, t.distinct_senders, t.has_quorum) == (te.power, te.distinct_senders, te.has_quorum)
            if i + 1 < len(order):
                bv.seals_stage_next(*host[order[i + 1]])
            if len(in_flight) == 2:
                check(bv.seals_collect(), in_flight.pop(0))
            if i + 1 < len(order):
                bv.seals_swap()
        check(bv.seals_collect(), in_flight.pop(0))
        sides, _ = bv.pipeline_stats()
        assert sides == (len(order) if side == "1" else 0)
        check(bv.seals_run(), order[-1])
    finally:
        bv.close()


def test_side_stream_tally_serves_the_warm_path_once_every_table_is_built(oracle):
    """With the key cache on the tally stays on the main stream while keys are being learned (it hands the device's learned-key
    counter on) and moves to the side stream once every validator has its table; verdicts and tallies are the oracle's throughout."""
    import go_ibft_amd.verifier as V
    r, batches = _batches(1024)
    wv = V.BatchVerifier(max_rows=1024, flags=V.FLAG_PUBKEY_CACHE)
    try:
        wv.set_validators(r.height, r.addrs, r.power)
        wv.seals_stage(*[x for x in batches[0][:4]])
        seen = []
        for _ in range(6):
            wv.seals_submit(); wv.seals_submit()
            for _ in range(2):
                verdict, t = wv.seals_collect()
                assert (verdict == batches[0][4]).all() and (t.power, t.has_quorum) == (batches[0][5].power, batches[0][5].has_quorum)
            seen.append(wv.pipeline_stats()[0])
        assert seen[0] == 0, seen                       # learning passes: one stream
        # Byzantine rows never teach a key, so some validators stay without a table and the cold kernel keeps running behind the
        # warm one: the side stream is taken only when EVERY validator of the set has a table
        built, total = wv.cache_stats()[0], len(r.addrs)
        assert (seen[-1] > 0) == (built >= total), (seen, built, total)
    finally:
        wv.close()


@pytest.mark.parametrize("side", ["1", "0"])
def test_pipeline_at_ragged_and_empty_sizes(oracle, monkeypatch, side):
    """The pipelined pass at the sizes where word and workgroup boundaries sit — 1, 63, 64, 65 rows, one row short of and one row
    past the one-workgroup tally's 4 096, an EMPTY batch between two full ones (no verdict kernel at all: the tally alone reports
    zero rows and no quorum, /root/reference/core/validator_manager.go:77-96 with an empty message list) — every pass against the
    oracle, with the tally on its own stream and on the main one."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W, binding as B
    monkeypatch.setenv("IBFT_SIDE_TALLY", side)
    r = W.make_round(4200, 777, byzantine=True, weighted=True)
    vs = B.ValSet(r.addrs, r.power)
    bv = V.BatchVerifier(max_rows=8192)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        sizes = [1, 63, 0, 64, 65, 4095, 0, 4097, 4200, 1]
        cols = lambda n: tuple(np.ascontiguousarray(c[:n]) for c in (r.hash32, r.seal65, r.signer20, r.pre_flags))
        want = {}
        for n in set(sizes):
            h, s, f, p = cols(n)
            e = B.verify_seals(vs, h, s, f, p) if n else np.zeros(0, np.uint8)
            want[n] = (e.astype(bool), B.tally(vs, f, e))
        in_flight = []
        for n in sizes:
            bv.seals_stage(*cols(n))
            bv.seals_submit()
            in_flight.append(n)
            if len(in_flight) == 2:
                m = in_flight.pop(0)
                verdict, t = bv.seals_collect()
                assert len(verdict) == m and (verdict == want[m][0]).all(), m
                wt = want[m][1]
                assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == (wt.power, wt.valid_rows, wt.distinct_senders, wt.has_quorum), m
        m = in_flight.pop(0)
        verdict, t = bv.seals_collect()
        assert len(verdict) == m and (verdict == want[m][0]).all() and t.power == want[m][1].power
    finally:
        bv.close()
