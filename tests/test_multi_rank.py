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


def _worker(rank, world, port, n_total, seed, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import go_ibft_amd.shard as S
        from oracle import binding as B, workload as W
        r = W.make_round(n_total, seed, byzantine=True, weighted=True)
        vs = B.ValSet(r.addrs, r.power)           # validator table replicated on every rank
        lo, hi = S.shard_range(n_total, rank, world)
        v = B.verify_seals(vs, r.hash32[lo:hi], r.seal65[lo:hi], r.signer20[lo:hi], r.pre_flags[lo:hi])
        t = B.tally(vs, r.signer20[lo:hi], v)
        from go_ibft_amd.verifier import bool_to_mask
        slots, _ = S.exchange_layout(n_total, world)
        buf = np.zeros(slots, dtype=np.int64)
        pmap = {bytes(a): int(p) for a, p in zip(r.addrs, r.power)}
        counted = {bytes(a) for a, ok in zip(r.signer20[lo:hi], v) if ok and bytes(a) in pmap}
        powers = [pmap[a] for a in counted]
        assert sum(powers) == t.power and len(counted) == t.distinct_senders
        S.fill_local(buf, rank, n_total, world, bool_to_mask(v.astype(bool)), powers, t.valid_rows,
                     t.distinct_senders)
        tens = torch.from_numpy(buf)
        dist.all_reduce(tens)                      # disjoint shards: SUM == OR
        verdict, power, valid, distinct, hq = S.merge(tens.numpy(), n_total, world, vs.quorum)
        # every rank must hold the same merged result, equal to the unsharded oracle
        full = B.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags)
        tf = B.tally(vs, r.signer20, full)
        assert (verdict == full.astype(bool)).all()
        assert (power, valid, distinct, int(hq)) == (tf.power, tf.valid_rows, tf.distinct_senders, tf.has_quorum)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 200), (2, 128), (3, 500)])
def test_sharded_verify_and_allreduce_merge(tmp_path, world, n_total):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, 77 + n_total, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


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
            for pw in (1, 4):
                wpr, slots = V.exchange_layout(n, w, pw)
                assert (slots, wpr * w) == S.exchange_layout(n, w, pw) and wpr == S.words_per_rank(n, w)


def test_merge_keeps_carries_with_huge_powers():
    """partial powers whose low words overflow 2^64 when added across ranks, and 256-bit
    stakes: the 32-bit pieces in 64-bit slots lose nothing."""
    import go_ibft_amd.shard as S
    for pw, powers_by_rank in ((1, [[2**64 - 1, 2**64 - 5], [2**64 - 3], [2**63, 2**63, 7]]),
                               (4, [[2**256 - 1, 2**255 + 12345], [2**256 - 2**200], [10**30, 2**64 + 1]])):
        world, n_total = 3, 500
        slots, _ = S.exchange_layout(n_total, world, pw)
        total = np.zeros(slots, dtype=np.int64)
        for rank, powers in enumerate(powers_by_rank):
            buf = np.zeros(slots, dtype=np.int64)
            lo, hi = S.shard_range(n_total, rank, world)
            mw = np.full(S.words_per_rank(n_total, world), 2**64 - 1, dtype=np.uint64)[: (hi - lo + 63) // 64]
            S.fill_local(buf, rank, n_total, world, mw, powers, hi - lo, len(powers), pw)
            total += buf                      # what all-reduce(SUM) does
        want = sum(sum(p) for p in powers_by_rank)
        verdict, power, valid, distinct, hq = S.merge(total, n_total, world, want, pw)
        assert power == want and hq and valid == n_total and distinct == sum(map(len, powers_by_rank))
        assert not S.merge(total, n_total, world, want + 1, pw)[4]
