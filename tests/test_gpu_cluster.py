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


@pytest.mark.parametrize("mode", ["batch", "ingest"])
def test_round_change_through_the_device_equals_the_oracle_backend(gpu_verifier, mode):
    """A round that CHANGES, the device behind every node (core/consensus_test.go:260-394, core/byzantine_test.go:60-130 with
    real signatures): round 0 reaches PREPARE quorum, no COMMIT is sent, the timer ticks; every ROUND_CHANGE(round 1) carries
    a PreparedCertificate; the round-1 proposer re-proposes with the RoundChangeCertificate; handleRoundChangeMessage and
    handlePrePrepare are answered from ibft_verify_certificates_wire's rows (ingest) or its batch calls (batch).  Every node
    inserts what the same cluster inserts with the oracle-backed per-message Verifier, and the device never fell back."""
    n = 7
    results = {}
    for m, gpu in (("stock", None), (mode, gpu_verifier)):
        c, proposals, sks, addrs = TC.real_rc_cluster(n, m, gpu=gpu)
        try:
            for nd in c.nodes:
                nd.withhold_commit = True
            assert c.run_height(1) == [None] * n and all(nd.latest_pc is not None for nd in c.nodes)
            for nd in c.nodes:
                nd.withhold_commit = False
            inserted = c.tick()
            p0, p1 = c.proposer(1, 0), c.proposer(1, 1)
            assert all(x is not None and x[0] == proposals[p0] for x in inserted)
            results[m] = (inserted, sorted(c.nodes[p1].rcc), [[nd.host.store_num(1, r, t) for r in (0, 1) for t in (0, 1, 2, 3)] for nd in c.nodes])
            if gpu is not None:
                assert all(nd.host.fallbacks() == 0 for nd in c.nodes)
                if m == "ingest":
                    assert all(nd.host.cert_stats()[0] > 0 for nd in c.nodes)
                    assert any(nd.host.rc_from_rows > 0 for nd in c.nodes) and any(nd.host.pp_from_rows > 0 for nd in c.nodes)
        finally:
            c.close()
    assert results["stock"] == results[mode]
