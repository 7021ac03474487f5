#!/usr/bin/env python3
"""tools/side_tally_trace.py N — 30 pipelined passes (ibft_seals_submit / _collect, one kept in flight) over N rows, cold, for a
rocprofv3 --kernel-trace run: which queue ran what when (tools/trace_timeline.py prints it; IBFT_SIDE_TALLY as set in the
environment).  profiles/r06v_side_tally_trace.txt came from this."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
n = int(sys.argv[1])
bv = V.BatchVerifier(max_rows=max(n, 1024))
r = SIM.make_round(bv, n, 600 + n)
bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
for _ in range(20): bv.seals_run()
bv.seals_submit()
t0 = time.perf_counter()
for _ in range(30):
    bv.seals_submit(); bv.seals_collect()
print("step ms", (time.perf_counter() - t0) / 30 * 1e3)
bv.seals_collect()
bv.close()
