"""§8f rank 3 timing (run on the GPU box): sender verification of one round's PREPARE+COMMIT messages from
their wire bytes — stock route (host: proto decode + PayloadNoSig re-marshal + flatten; device: hash +
recover) against the device wire walk (ibft_verify_senders_wire).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import go_ibft_amd.hostlib as H
import go_ibft_amd.verifier as V
import wire_cases as WCASE
from oracle import workload as W

out = {}
for n in (1024, 4096):
    r = W.make_round(n, 700 + n)
    rows = WCASE.canonical_round(r)           # n messages, COMMIT and PREPARE alternating
    wire, off = WCASE.pack(rows)
    for cache in (False, True):
        bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if cache else 0, max_rows=max(n, 1024))
        bv.set_validators(1, r.addrs, r.power)
        res = {}
        for name, stock in (("stock_route", True), ("wire_walk", False)):
            for _ in range(3):
                v, _, _ = H.verify_senders_wire(bv, wire, off, stock=stock)
            assert v.all()
            ts, hs = [], []
            for _ in range(15):
                t0 = time.perf_counter()
                v, host_ms, _ = H.verify_senders_wire(bv, wire, off, stock=stock)
                ts.append((time.perf_counter() - t0) * 1e3)
                hs.append(host_ms)
            res[name] = {"total_ms_p50": float(np.median(ts)), "host_decode_flatten_ms_p50": float(np.median(hs))}
        res["wire_bytes"] = len(wire)
        out[f"n{n}_{'warm' if cache else 'cold'}"] = res
        bv.close()
print(json.dumps(out, indent=1))
