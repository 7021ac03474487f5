"""The N>1 path on CPU: world_size-2 (and 3) `gloo` process groups run the same shard layout
and all-reduce merge that bench.py uses over RCCL; per-rank verdicts come from the CPU oracle
(test infrastructure) so the test checks sharding + exchange, not the GPU kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _round_with_seam_duplicates(n_total, seed, world, dup):
    """a Byzantine weighted round; with `dup`, the valid rows just before every shard seam are repeated just after
    it (same sender, same seal): one validator with valid rows in two shards"""
    import go_ibft_amd.shard as S
    from oracle import workload as W
    r = W.make_round(n_total, seed, byzantine=True, weighted=True)
    cols = {k: np.array(getattr(r, k)) for k in ("hash32", "seal65", "signer20", "pre_flags")}
    if dup:
        for rank in range(1, world):
            lo, hi = S.shard_range(n_total, rank, world)
            k = min(5, hi - lo, lo)
            for c in cols.values():
                c[lo:lo + k] = c[lo - k:lo]
    return r, cols


def _worker(rank, world, port, n_total, seed, out_dir, dup):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import go_ibft_amd.shard as S
        from oracle import binding as B
        r, c = _round_with_seam_duplicates(n_total, seed, world, dup)
        vs = B.ValSet(r.addrs, r.power)           # validator table replicated on every rank
        lo, hi = S.shard_range(n_total, rank, world)
        v = B.verify_seals(vs, c["hash32"][lo:hi], c["seal65"][lo:hi], c["signer20"][lo:hi], c["pre_flags"][lo:hi])
        t = B.tally(vs, c["signer20"][lo:hi], v)
        from go_ibft_amd.verifier import bool_to_mask
        nv = len(r.addrs)
        slots, _, _ = S.exchange_layout(n_total, world, nv)
        buf = np.zeros(slots, dtype=np.int64)
        index = {bytes(a): i for i, a in enumerate(r.addrs)}
        counted = {index[bytes(a)] for a, ok in zip(c["signer20"][lo:hi], v) if ok and bytes(a) in index}
        assert sum(int(r.power[i]) for i in counted) == t.power and len(counted) == t.distinct_senders
        S.fill_local(buf, rank, n_total, world, nv, bool_to_mask(v.astype(bool)), counted, t.valid_rows)
        tens = torch.from_numpy(buf)
        dist.all_reduce(tens)                      # disjoint word ranges and segments: SUM == OR
        verdict, power, valid, distinct, hq, overlap = S.merge(tens.numpy(), n_total, world, [int(p) for p in r.power], vs.quorum)
        # every rank must hold the same merged result, equal to the unsharded oracle — a sender on both sides of a
        # seam counted once (core/validator_manager.go:86-92, 147-155)
        full = B.verify_seals(vs, c["hash32"], c["seal65"], c["signer20"], c["pre_flags"])
        tf = B.tally(vs, c["signer20"], full)
        assert (verdict == full.astype(bool)).all()
        assert (power, valid, distinct, int(hq)) == (tf.power, tf.valid_rows, tf.distinct_senders, tf.has_quorum)
        open(os.path.join(out_dir, f"ok{rank}_{overlap}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,dup", [(2, 200, False), (2, 128, True), (3, 500, True), (2, 200, True), (4, 700, True),
                                                (8, 1100, True)])   # the driver's scaling run: 1, 2, 4, 8 ranks
def test_sharded_verify_and_allreduce_merge(tmp_path, world, n_total, dup):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, 77 + n_total, str(tmp_path), dup), nprocs=world, join=True)
    names = sorted(os.listdir(tmp_path))
    assert [n.split("_")[0] for n in names] == [f"ok{r}" for r in range(world)]
    overlaps = {int(n.split("_")[1]) for n in names}
    assert len(overlaps) == 1                       # every rank computed the same overlap ...
    if not dup:
        assert overlaps == {0}                      # ... and there is none when every sender has one row


def test_shard_ranges_cover_and_align():
    import go_ibft_amd.shard as S
    for n in (1, 63, 64, 65, 1000, 1024, 4096, 65536, 70001):
        for w in (1, 2, 3, 4, 8):
            ranges = [S.shard_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and (a % 64 == 0 or b == a) and (c % 64 == 0 or d == c)   # empty tail shards may be unaligned
            assert all(hi - lo <= S.words_per_rank(n, w) * 64 for lo, hi in ranges)


def test_layout_equals_the_librarys_pure_functions():
    """go_ibft_amd/shard.py restates ibft_shard_range / ibft_exchange_layout (no GPU needed for either)."""
    import go_ibft_amd.shard as S
    import go_ibft_amd.verifier as V
    for n in (0, 1, 63, 64, 65, 1000, 4096, 16384, 65536, 70001):
        for w in (1, 2, 3, 4, 8):
            for r in range(w):
                assert V.shard_range(n, r, w) == S.shard_range(n, r, w)
            for nv in (1, 64, 65, 4096, 65536, 100000):
                for k in (1, 2):
                    wpr, sw, slots = V.exchange_layout(n, w, nv, k)
                    assert (slots, k * wpr * w) == S.exchange_layout(n, w, nv, k)[:2]
                    assert wpr == S.words_per_rank(n, w) and sw == S.seen_words(nv)


def test_merge_counts_a_validator_once_whatever_the_shards():
    """ADVICE r2 (medium) / validator_manager.go:77-96: the same validators valid in every shard; huge and 256-bit
    powers (no carry may be lost — the power is recomputed from the merged bitmap with Python ints here,
    with 32-bit pieces in 64-bit accumulators on the device)."""
    import go_ibft_amd.shard as S
    world, n_total = 3, 500
    for powers in ([2**64 - 1, 2**64 - 5, 2**64 - 3, 2**63, 2**63, 7],
                   [2**256 - 1, 2**255 + 12345, 2**256 - 2**200, 10**30, 2**64 + 1, 5]):
        nv = len(powers)
        slots, _, _ = S.exchange_layout(n_total, world, nv)
        total = np.zeros(slots, dtype=np.int64)
        seen_by = [[0, 1, 2], [1, 2, 3], [0, 3, 4]]            # validator 5 never valid; 0-3 in two shards each
        for rank, counted in enumerate(seen_by):
            buf = np.zeros(slots, dtype=np.int64)
            lo, hi = S.shard_range(n_total, rank, world)
            mw = np.full(S.words_per_rank(n_total, world), 2**64 - 1, dtype=np.uint64)[: (hi - lo + 63) // 64]
            S.fill_local(buf, rank, n_total, world, nv, mw, counted, hi - lo)
            total += buf                      # what all-reduce(SUM) does
        want = sum(powers[:5])
        verdict, power, valid, distinct, hq, overlap = S.merge(total, n_total, world, powers, want)
        assert power == want and hq and valid == n_total and distinct == 5 and overlap == 4
        assert not S.merge(total, n_total, world, powers, want + 1)[4]
        # a message set: two verdict arrays in one buffer
        slots2, _, _ = S.exchange_layout(n_total, world, nv, 2)
        total2 = np.zeros(slots2, dtype=np.int64)
        for rank, counted in enumerate(seen_by):
            buf = np.zeros(slots2, dtype=np.int64)
            lo, hi = S.shard_range(n_total, rank, world)
            k = (hi - lo + 63) // 64
            S.fill_local(buf, rank, n_total, world, nv, [np.full(k, 2**64 - 1, dtype=np.uint64), np.full(k, 0x5555555555555555, dtype=np.uint64)],
                         counted, hi - lo)
            total2 += buf
        (vs_, vv_), power2, *_ = S.merge(total2, n_total, world, powers, want, 2)
        assert vs_.all() and (vv_ == (np.arange(n_total) % 2 == 0)).all() and power2 == want
