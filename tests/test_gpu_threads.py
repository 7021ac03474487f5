"""Re-entrancy: the Verifier is called concurrently from several goroutines
(/root/reference/core/ibft.go:335-347, 1128); each caller holds its own ibft_ctx.  Four host
threads hammer four contexts (two cold, two warm) at once and every result must equal the oracle."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_contexts(oracle):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    rounds = [W.make_round(300 + 40 * i, 60 + i, byzantine=True, weighted=True) for i in range(4)]
    exps = []
    for r in rounds:
        vs = oracle.ValSet(r.addrs, r.power)
        exps.append(oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=4).astype(bool))
    errors = []

    def worker(i):
        try:
            r = rounds[i]
            bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if i % 2 else 0, max_rows=4096)
            bv.set_validators(1, r.addrs, r.power)
            for _ in range(12):
                got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
                if not (got == exps[i]).all():
                    errors.append((i, "verdict mismatch"))
                h = bv.is_valid_proposal_hash(r.raw, r.round, r.hash32, r.hash_len)
                if h.sum() != (r.hash_len == 32).sum() - sum(1 for k in r.kinds if k == "wrong_hash_field"):
                    errors.append((i, "hash mismatch"))
            bv.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_one_context_shared_by_threads(oracle):
    """A single ibft_ctx used from several threads: calls serialise on the context mutex
    (ibft_verify_seals is one critical section) and still return oracle-exact verdicts."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(500, 71, byzantine=True)
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=4).astype(bool)
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=4096)
    bv.set_validators(1, r.addrs, r.power)
    errors = []

    def worker(k):
        try:
            for j in range(10):
                lo = (37 * (k + j)) % 300
                got, _ = bv.is_valid_committed_seal(r.hash32[lo:lo + 200], r.seal65[lo:lo + 200],
                                                    r.signer20[lo:lo + 200], r.pre_flags[lo:lo + 200])
                if not (got == exp[lo:lo + 200]).all():
                    errors.append((k, j))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    bv.close()
    assert not errors, errors


_EXPORT_ON_SCRIPT = r"""
import sys
import numpy as np
import torch                                  # before libibftgpu.so: one HIP runtime per process, torch's
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import go_ibft_amd.verifier as V
from oracle import binding as oracle, workload as W
r = W.make_round(1500, 91, byzantine=True, weighted=True)
vs = oracle.ValSet(r.addrs, r.power)
exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=4).astype(bool)
et = oracle.tally(vs, r.signer20, exp.astype(np.uint8))
words = (r.n + 63) // 64
bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=4096)
side = torch.cuda.Stream()
bufs = [torch.zeros(words + 4, dtype=torch.int64, device="cuda") for _ in range(6)]
bv.set_validators(1, r.addrs, r.power)
bv.seals_stage(r.hash32, r.seal65, r.signer20, r.pre_flags)
bv.seals_launch(1)
for k, b in enumerate(bufs):                  # cold pass first, known-key passes after
    bv.seals_export_on(b.data_ptr(), b[words:].data_ptr(), side.cuda_stream)
    if k + 1 < len(bufs):
        bv.seals_launch(1)                    # enqueued before the copies of pass k have run
side.synchronize()
bv.sync()
for b in bufs:
    h = b.cpu().numpy().view(np.uint64)
    got = V.mask_to_bool(h[:words].copy(), r.n)
    assert (got == exp).all()
    assert int(h[words]) | (int(h[words + 1]) << 64) == et.power
    assert int(h[words + 2]) & 0xFFFFFFFF == et.valid_rows and int(h[words + 3]) == et.has_quorum
bv.close()
print("EXPORT_ON_OK")
"""


def test_export_on_a_consumer_stream():
    """ibft_seals_export_on: the verdict words and tally partials reach caller-owned device buffers through
    the caller's stream (no host wait inside the call), launches keep flowing on the context's own stream,
    and the next tally waits for the copies — what bench.py's sharded loop relies on.  Runs in its own
    process with torch imported first (as bench.py does): loading torch's bundled HIP runtime after
    libibftgpu.so has initialised the system one leaves torch without a device."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _EXPORT_ON_SCRIPT, root], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "EXPORT_ON_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_concurrent_contexts_share_the_key_cache(oracle):
    """The key tables are per device (round 3): six threads, six contexts with the key cache — four on ONE validator set,
    two on others — learn, build and verify at the same time.  A key published by one context's kernel while another
    context's build pass scans the slots must never yield a half-written table: every verdict of every pass equals the
    oracle's, and in the end every context counts a table for each of its honest validators."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    shared = [W.make_round(600, 2100, round_=k, byzantine=True, weighted=True) for k in range(3)]   # same keys, new seals
    others = [W.make_round(350 + 50 * i, 2200 + i, byzantine=True) for i in range(2)]
    vs = oracle.ValSet(shared[0].addrs, shared[0].power)
    exp_shared = [oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=4).astype(bool) for r in shared]
    exp_others = [oracle.verify_seals(oracle.ValSet(r.addrs, r.power), r.hash32, r.seal65, r.signer20, r.pre_flags,
                                      nthreads=4).astype(bool) for r in others]
    errors, tables = [], {}
    start = threading.Barrier(6)

    def worker(i):
        try:
            bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)
            if i < 4:
                bv.set_validators(1, shared[0].addrs, shared[0].power)
            else:
                bv.set_validators(1, others[i - 4].addrs, others[i - 4].power)
            start.wait()
            for it in range(15):
                if i < 4:
                    r, exp = shared[(it + i) % 3], exp_shared[(it + i) % 3]
                else:
                    r, exp = others[i - 4], exp_others[i - 4]
                got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
                if not (got == exp).all():
                    errors.append((i, it, "verdict mismatch", np.flatnonzero(got != exp)[:6].tolist()))
            tables[i] = bv.cache_stats()[0]
            bv.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    honest = int(np.logical_or.reduce(exp_shared).sum())           # validators with at least one valid seal in some round
    assert all(tables[i] == honest for i in range(4)), (tables, honest)
