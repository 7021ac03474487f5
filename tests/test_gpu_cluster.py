"""GPU: BASELINE config #1's cluster (tests/test_host_cluster.py) with the device behind every node — N mirror nodes, real
secp256k1 keys, every walk of handlePrePrepare / handlePrepare / handleCommit and every micro-batch of the receive side
answered by libibftgpu.so — must insert exactly what the same cluster inserts with the per-message Verifier answered by
the CPU oracle (the crypto Backend an application would write)."""
import pytest

import cluster_sim as CS
import test_host_cluster as TC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,byz", [(4, 0), (7, 2)])
@pytest.mark.parametrize("mode", ["batch", "ingest"])
def test_cluster_through_the_device_equals_the_oracle_backend(gpu_verifier, n, byz, mode):
    import numpy as np
    from oracle import binding as B
    sks, addrs, raw, build, verifier = TC._real_crypto(n, seed=11)
    bad_seal, forged = build(bad_seal=True), build(forged_envelope=True)

    def mixed(nd, type_, view):
        return forged(nd, type_, view) if (nd.byzantine and nd.index == 1) else bad_seal(nd, type_, view)
    results = {}
    for m, gpu in (("stock", None), (mode, gpu_verifier)):
        if gpu is not None:
            gpu.set_validators(1, np.frombuffer(b"".join(addrs), dtype=np.uint8).reshape(n, 20), np.ones(n, dtype=np.uint64))
        ref = [None]
        c = CS.Cluster(addrs, mixed, verifier(ref), mode=m, gpu=gpu)
        ref[0] = c
        for nd in c.nodes[:byz]:
            nd.byzantine = True
        try:
            hist = []
            for height in (2, 3):
                inserted = c.run_height(height)
                hist.append((inserted, [nd.prepared for nd in c.nodes], [nd.host.store_num(height, 0, 2) for nd in c.nodes]))
                if gpu is not None:
                    assert all(nd.host.fallbacks() == 0 for nd in c.nodes)      # the device answered, not the callbacks
            results[m] = hist
        finally:
            c.close()
    assert results["stock"] == results[mode]
    H0 = B.proposal_hash(raw, 0)
    honest = frozenset((addrs[i], B.sign(sks[i], H0)) for i in range(byz, n))
    for inserted, _, stored in results[mode]:
        assert all(x is not None and x[0] == raw and x[1] == honest for x in inserted)
        assert stored == [n - byz] * n
