"""GPU: the bit-parity soak of SURVEY.md §8d in the driver-run suite (reduced: tools/soak.py runs the full 10 300 rounds).

1 000 Byzantine rounds at N = 64 (seeds 1…1000, odd seeds weighted, every third with a forged envelope) + 10 rounds at
N = 4 096: every round through ibft_verify_seals twice on a key-caching context (recover kernels, then known-key kernels)
and twice as a whole COMMIT set through ibft_verify_messages — every verdict bit, Σ power, valid rows, distinct senders
and the quorum flag against the CPU oracle.  Rounds and oracle answers are produced by worker processes (spawned: they
never touch the HIP runtime of this process) while the GPU consumes."""
import os
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return n


def test_reduced_soak_bit_identical_with_the_oracle():
    import go_ibft_amd.verifier as V
    from oracle.soak_job import make
    # the long jobs first; 20 000 rows: the lane-group kernels (round 4: signed windows over a common-Z table)
    jobs = [(20000, 1), (4096, 1)] + [(4096, s) for s in range(2, 11)] + [(64, s) for s in range(1, 1001)]
    stat = {"rounds": 0, "rows": 0, "bad_rows": 0, "mismatches": 0, "quorum_true": 0, "quorum_false": 0, "prepare_true": 0,
            "prepare_voided": 0}
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=20000)
    try:
        with ProcessPoolExecutor(max_workers=min(16, _cores()), mp_context=mp.get_context("spawn")) as ex:
            for (n, seed, addrs, power, h, s, f, pre, exp, et, env) in ex.map(make, jobs, chunksize=4):
                bv.set_validators(seed, addrs, power)
                for _ in range(2):
                    got, t = bv.is_valid_committed_seal(h, s, f, pre)
                    stat["mismatches"] += int((got != exp).sum())
                    stat["mismatches"] += int((t.power, t.quorum, t.valid_rows, t.distinct_senders, t.has_quorum) != et)
                payload, off, msig, hlen, raw, rnd, snd, clo, ets, hpq = env
                for _ in range(2):
                    gs, gv, t = bv.verify_messages(payload, off, msig, f, h, hlen, s, valid_pre=pre, raw=raw, round_=rnd)
                    stat["mismatches"] += int((gs != snd).sum()) + int((gv != clo).sum())
                    stat["mismatches"] += int((t.power, t.quorum, t.valid_rows, t.distinct_senders, t.has_quorum) != ets)
                # HasPrepareQuorum on the device (ibft_tally_prepare) over the rows both verdicts accept, validator seed mod n
                # as the proposer: decision, Σ power with the proposer's seat, rows sent by the proposer
                tp = bv.has_prepare_quorum(f, snd & clo, hpq[3])
                stat["mismatches"] += int((tp.has_quorum, tp.power, tp.proposer_rows) != hpq[:3])
                stat["prepare_true"] += hpq[0]
                stat["prepare_voided"] += int(hpq[2] > 0)
                stat["rounds"] += 1
                stat["rows"] += n
                stat["bad_rows"] += int((~exp).sum())
                stat["quorum_true" if et[4] else "quorum_false"] += 1
    finally:
        bv.close()
    assert stat["rounds"] == 1011 and stat["mismatches"] == 0, stat
    assert stat["prepare_true"] > 0 and stat["prepare_voided"] > 0, stat
    assert stat["bad_rows"] > 0.15 * stat["rows"] and stat["quorum_true"] > 0 and stat["quorum_false"] > 0, stat
