#!/usr/bin/env python
"""tools/sign_rate.py — wall-clock rate of ibft_sign_seals (f4, simulators) at a few batch sizes, and of
sign → verify on the resident batch.  Prints one JSON object."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import go_ibft_amd.verifier as V  # noqa: E402

out = {}
rng = np.random.default_rng(1)
for n in (1024, 4096, 16384, 65536):
    sk = np.frombuffer(rng.bytes(32 * n), np.uint8).reshape(-1, 32).copy()
    sk[:, 0] &= 0x7F
    sk[:, 31] |= 1
    hs = np.tile(np.frombuffer(rng.bytes(32), np.uint8), (n, 1))
    bv = V.BatchVerifier(max_rows=n)
    bv.sign_seals(sk, hs)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        sig, signer, ok = bv.sign_seals(sk, hs)
        best = min(best, time.perf_counter() - t0)
    assert ok.all()
    bv.set_validators(1, signer, np.ones(n, np.uint64))
    t0 = time.perf_counter()
    verdict, t = bv.seals_run()
    tv = time.perf_counter() - t0
    assert verdict.all() and t.has_quorum
    out[str(n)] = {"sign_ms": round(best * 1e3, 3), "seals_per_s": round(n / best), "verify_resident_ms": round(tv * 1e3, 3)}
    bv.close()
print(json.dumps({"ibft_sign_seals": out, "note": "host→host wall clock incl. PCIe both ways"}))
