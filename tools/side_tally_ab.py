#!/usr/bin/env python3
"""tools/side_tally_ab.py [rounds=3] [N …] — the pipelined pass (ibft_seals_submit / _collect, one pass kept in flight: the
bench's headline step) with the tally on a stream of its own (IBFT_SIDE_TALLY=1, round 6) against everything on one stream
(IBFT_SIDE_TALLY=0), ONE library, one lease, separate processes alternating.  Per size: ms per step over 400 delivered passes
behind 150 untimed ones, the verdict kernel's HIP-event time sampled on every fourth pass of a second series (the tally of the
pass before now runs NEXT TO it: does it slow it down?), and the synchronous step (launch → results on the host) that the side
stream must not make worse.  A size written wN is the warm path (keys known).  IBFT_AB_ENV=<NAME> switches another 0 / 1 knob of the library instead
(round 6 compared a build with IBFT_EXT_STOP_EVENTS that way — the event a side-stream tally waits for attached to the verdict
dispatch instead of recorded behind it: profiles/r06zi_ext_stop_ab.txt, not adopted, the knob is gone).

    python tools/side_tally_ab.py 3 > gpurun_out/profiles/r06v_side_tally_ab.txt"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import go_ibft_amd.numa as NUMA
NUMA.pin_to_device_node(0)
import numpy as np
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
out = {}
for key in sys.argv[1:]:
    warm = key.startswith("w")
    n = int(key.lstrip("w"))
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if warm else 0, max_rows=max(n, 1024))
    r = SIM.make_round(bv, n, 600 + n)
    bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
    for _ in range(150): v, t = bv.seals_run()
    assert v.all() and t.has_quorum == 1
    if warm: assert bv.cache_stats()[0] == n
    def series(k):
        bv.seals_submit()
        t0 = time.perf_counter()
        for _ in range(k):
            bv.seals_submit()
            v, t = bv.seals_collect()
        dt = time.perf_counter() - t0
        v, t = bv.seals_collect()
        assert v.all() and t.has_quorum == 1 and t.valid_rows == n
        return dt / k * 1e3
    series(50)
    step = min(series(400) for _ in range(3))
    bv.set_kernel_timing(4); bv.last_kernel_ms()
    series(400)
    ms, k = bv.last_kernel_ms()
    bv.set_kernel_timing(0)
    lat = []
    for _ in range(60):
        t0 = time.perf_counter(); bv.seals_run(); lat.append((time.perf_counter() - t0) * 1e3)
    out[key] = {"step_ms": round(step, 5), "kernel_ms": round(ms / max(k, 1), 5), "sync_ms": round(float(np.median(lat)), 5),
                "side": bv.pipeline_stats()[0]}
    bv.close()
print(json.dumps(out))
''' % ROOT
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sizes = sys.argv[2:] or ["1024", "4096", "16384", "65536", "w4096", "w65536"]
acc = {"0": {}, "1": {}}
for rd in range(rounds):
    for side in ("0", "1"):
        env = dict(os.environ, **{os.environ.get("IBFT_AB_ENV", "IBFT_SIDE_TALLY"): side})   # IBFT_AB_ENV: A/B another 0 / 1 knob the same way
        p = subprocess.run([sys.executable, "-c", CHILD] + sizes, env=env, capture_output=True, text=True, timeout=900)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        print(f"side={side}", line or p.stderr[-600:], flush=True)
        try:
            for k, v in json.loads(line).items():
                for q, x in v.items():
                    acc[side].setdefault(k, {}).setdefault(q, []).append(x)
        except ValueError:
            pass
print("# medians: N  | step ms one stream -> side stream (ratio) | verdict kernel ms | synchronous step ms")
for k in sizes:
    if k in acc["0"] and k in acc["1"]:
        m = {s: {q: float(np.median(acc[s][k][q])) for q in acc[s][k]} for s in ("0", "1")}
        print(f"# {k:>7s} | {m['0']['step_ms']:.5f} -> {m['1']['step_ms']:.5f} ({m['1']['step_ms'] / m['0']['step_ms']:.4f}) | "
              f"{m['0']['kernel_ms']:.5f} -> {m['1']['kernel_ms']:.5f} | {m['0']['sync_ms']:.5f} -> {m['1']['sync_ms']:.5f} | "
              f"side-stream tallies {int(m['0']['side'])} / {int(m['1']['side'])}")
