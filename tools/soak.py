"""SURVEY.md §8d soak (run on the GPU box): bit-identical verdicts and quorum decisions over many synthetic
rounds.  Default: 10 000 rounds at N = 64 with seeds 1…10000 (every validator key, proposal and the 20 %
Byzantine mix derive from the seed; odd seeds use weighted voting power) + 100 rounds at each larger N.
Every round goes through ibft_verify_seals twice on a key-caching context (first pass: recover kernels,
second pass: known-key kernels) and — round 2 — as a whole COMMIT set through ibft_verify_messages twice
(envelope signatures and committed seals in one verdict launch), and is compared with the CPU oracle: verdict
of every row (seal verdicts; sender and closure verdicts of the set), Σ power, valid rows, distinct senders,
quorum flag.  Workers generate rounds and oracle answers in parallel; the GPU
consumer is this process.  Prints one JSON object."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


from oracle.soak_job import make  # noqa: E402  (round generation + the oracle's answers, in worker processes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10000)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--large", type=str, default="256,1024,4096")
    ap.add_argument("--large-rounds", type=int, default=100)
    ap.add_argument("--huge", type=str, default="16384:8,40000:4",
                    help="size:rounds pairs beyond the row-per-signature kernel's range (the lane-group and lane kernels)")
    ap.add_argument("--procs", type=int, default=0)
    args = ap.parse_args()
    import go_ibft_amd.verifier as V
    jobs = [(args.n, s) for s in range(1, args.rounds + 1)]
    for n in [int(x) for x in args.large.split(",") if x]:
        jobs += [(n, s) for s in range(1, args.large_rounds + 1)]
    huge = [tuple(int(v) for v in x.split(":")) for x in args.huge.split(",") if x]
    for n, k in huge:
        jobs += [(n, s) for s in range(1, k + 1)]
    procs = args.procs or min(16, len(os.sched_getaffinity(0)))
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=max([8192] + [n for n, _ in huge]))
    stat = {}
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        for (n, seed, addrs, power, h, s, f, pre, exp, et, env) in pool.imap(make, jobs, chunksize=8):
            st = stat.setdefault(n, {"rounds": 0, "rows": 0, "bad_rows": 0, "verdict_row_mismatches": 0,
                                     "tally_mismatches": 0, "quorum_mismatches": 0, "quorum_true": 0,
                                     "passes": 0, "set_passes": 0, "set_row_mismatches": 0, "set_tally_mismatches": 0,
                                     "prepare_quorum_mismatches": 0, "prepare_quorum_true": 0, "proposer_voided": 0})
            bv.set_validators(seed, addrs, power)
            for _ in range(2):  # recover kernels, then known-key kernels for the keys just learned
                got, t = bv.is_valid_committed_seal(h, s, f, pre)
                st["verdict_row_mismatches"] += int((got != exp).sum())
                st["tally_mismatches"] += int((t.power, t.quorum, t.valid_rows, t.distinct_senders) != et[:4])
                st["quorum_mismatches"] += int(t.has_quorum != et[4])
                st["passes"] += 1
            payload, off, msig, hlen, raw, rnd, snd, clo, ets, hpq = env
            for _ in range(2):  # the COMMIT set in one call: the keys are known by now; a fresh set of validators next round
                gs, gv, t = bv.verify_messages(payload, off, msig, f, h, hlen, s, valid_pre=pre, raw=raw, round_=rnd)
                st["set_row_mismatches"] += int((gs != snd).sum()) + int((gv != clo).sum())
                st["set_tally_mismatches"] += int((t.power, t.quorum, t.valid_rows, t.distinct_senders, t.has_quorum) != ets)
                st["set_passes"] += 1
            # HasPrepareQuorum on the device over the rows that survived both verdicts (ibft_tally_prepare)
            tp = bv.has_prepare_quorum(f, snd & clo, hpq[3])
            st["prepare_quorum_mismatches"] += int((tp.has_quorum, tp.power, tp.proposer_rows) != hpq[:3])
            st["prepare_quorum_true"] += hpq[0]
            st["proposer_voided"] += int(hpq[2] > 0)
            st["rounds"] += 1
            st["rows"] += n
            st["bad_rows"] += int((~exp).sum())
            st["quorum_true"] += int(et[4])
    bv.close()
    out = {"seconds": round(time.time() - t0, 1), "procs": procs, "by_n": {str(k): v for k, v in sorted(stat.items())},
           "total_rounds": sum(v["rounds"] for v in stat.values()),
           "total_mismatches": sum(v["verdict_row_mismatches"] + v["tally_mismatches"] + v["quorum_mismatches"] +
                                   v["set_row_mismatches"] + v["set_tally_mismatches"] + v["prepare_quorum_mismatches"]
                                   for v in stat.values())}
    print(json.dumps(out))
    sys.exit(0 if out["total_mismatches"] == 0 else 1)


if __name__ == "__main__":
    main()
