"""GPU: the multi-GPU exchange inside libibftgpu.so (ibft_comm_* / ibft_group_*), on the single-GPU test box
with world size 1 — the RCCL communicator, the pack / all-reduce / unpack pipeline, the double-buffered
hand-off and the group entry point all run for real; N > 1 differs only in the rank offset of the word range
(covered on CPU by tests/test_multi_rank.py and by the layout tests in tests/test_cabi.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_exchange_equals_local_fetch(oracle):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(3000, 31, byzantine=True, weighted=True)
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
    te = oracle.tally(vs, r.signer20, exp.astype(np.uint8))
    bv = V.BatchVerifier(max_rows=4096)
    try:
        bv.set_validators(1, r.addrs, r.power)
        bv.comm_init(V.comm_unique_id(), 0, 1)
        bv.seals_stage(r.hash32, r.seal65, r.signer20, r.pre_flags)
        # pipelined as bench.py drives it: exchange k overlaps launch k+1, results consumed one pass later
        bv.seals_launch(1)
        for k in range(6):
            bv.seals_exchange(r.n)
            if k + 1 < 6:
                bv.seals_launch(1)
            if k >= 1:
                got, t = bv.seals_fetch_merged()
                assert (got == exp).all()
        got, t = bv.seals_fetch_merged()
        assert (got == exp).all()
        assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum, t.quorum) == \
               (te.power, te.valid_rows, te.distinct_senders, te.has_quorum, te.quorum)
        local, tl = bv.seals_fetch()     # the un-merged results are still the local ones
        assert (local == exp).all() and tl.power == te.power
        with pytest.raises(RuntimeError):
            bv.seals_fetch_merged() if bv._xq.append(r.n) is None else None   # nothing outstanding
        bv._xq.clear()
        with pytest.raises(RuntimeError, match="not this rank's shard"):
            bv.seals_exchange(r.n + 64)
        bv.comm_destroy()
    finally:
        bv.close()


@pytest.mark.parametrize("n", [1, 64, 1000, 4096, 9000])
def test_group_of_one_device_matches_oracle(oracle, n):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    from oracle.semantics import ValidatorManager
    r = W.make_round(n, 77 + n, byzantine=n > 64, weighted=True)
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
    te = oracle.tally(vs, r.signer20, exp.astype(np.uint8))
    g = V.DeviceGroup([0], max_rows_total=max(n, 1024))
    try:
        assert g.size == 1
        g.set_validators(1, r.addrs, r.power)
        for _ in range(2):
            got, t = g.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
            assert (got == exp).all()
            assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == \
                   (te.power, te.valid_rows, te.distinct_senders, te.has_quorum)
        # 256-bit powers through the group: the exchange carries 8 pieces instead of 2
        stakes = [(1 + int(p)) * 10**21 for p in r.power]
        vm = ValidatorManager()
        assert vm.init({bytes(r.addrs[i]): stakes[i] for i in range(n)})
        g.set_validators_u256(1, r.addrs, stakes)
        got, t = g.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
        assert (got == exp).all()
        w = g.last_tally_wide()
        senders = {bytes(r.signer20[i]) for i in np.nonzero(exp)[0]}
        assert w.power == sum(vm.power[a] for a in senders) and w.quorum == vm.quorum
        assert bool(t.has_quorum) == (w.power >= w.quorum)
    finally:
        g.close()


def test_bench_sharded_path_with_one_rank():
    """bench.py's N>1 code path (process group + the library's RCCL exchange, pipelined) forced with one rank on
    the single-GPU box; `python bench.py --gpus N` launches the same path under torch.distributed.run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IBFT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline", "--no-sequence", "--no-warm"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])   # the JSON line is the last thing on stdout
    assert rec["n_gpus"] == 1 and rec["config"]["validators"] == 4096 and rec["value"] > 1e5
    assert rec["roofline"]["kernel"].startswith("ecrecover_")
