#!/usr/bin/env python
"""tools/seq_breakdown.py — where the config-#3 sequence latency goes: p50 of each of its five C-ABI calls
(host columns → host-visible verdicts) next to the device time of the verdict kernel inside the call.
GPU box only.  Usage: python tools/seq_breakdown.py [rounds=300] > profiles/<tag>_seq_breakdown.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("IBFT_TORCH_FIRST") == "1":   # the HIP runtime bundled with torch (ROCm 7.0.2) instead of the image's 7.2
    import torch
    torch.cuda.set_device(0)
import go_ibft_amd.verifier as V  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
fx = dict(np.load(os.path.join(ROOT, "tests/golden/bench_round_n4096.npz")))
n = len(fx["addrs"])
raw, rnd = fx["raw"].tobytes(), int(fx["round"])
ppayload, poff, psig = fx["prepare_payload"].tobytes(), fx["prepare_off"], fx["prepare_sig65"]
cpayload, coff, csig = fx["payload"].tobytes(), fx["off"], fx["msg_sig65"]
pfrom, phash = fx["addrs"][1:], fx["hash32"][1:]
plen, clen = np.full(n - 1, 32, np.uint8), np.full(n, 32, np.uint8)
if pinned:
    ppayload, poff, psig, cpayload, coff, csig, pfrom, phash, plen, clen = (
        V.pinned_copy(x) for x in (ppayload, poff, psig, cpayload, coff, csig, pfrom, phash, plen, clen))
    for k in ("signer20", "hash32", "seal65"):
        fx[k] = V.pinned_copy(fx[k])
out = {"rows": n, "rounds": rounds, "host_columns": "pinned" if pinned else "pageable", "payload_bytes": {"prepare": len(ppayload), "commit": len(cpayload)}, "proposal_bytes": len(raw)}
for name, flags in (("cold", 0), ("warm", V.FLAG_PUBKEY_CACHE)):
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    bv.set_validators(int(fx["height"]), fx["addrs"], fx["power"])
    calls = [
        ("senders_prepare", lambda: bv.is_valid_validator(ppayload, poff, psig, pfrom)),
        ("hashes_prepare", lambda: bv.is_valid_proposal_hash(raw, rnd, phash, plen)),
        ("senders_commit", lambda: bv.is_valid_validator(cpayload, coff, csig, fx["signer20"])),
        ("hashes_commit", lambda: bv.is_valid_proposal_hash(raw, rnd, fx["hash32"], clen)),
        ("seals", lambda: bv.is_valid_committed_seal(fx["hash32"], fx["seal65"], fx["signer20"])),
        ("set_prepare", lambda: bv.verify_messages(ppayload, poff, psig, pfrom, phash, plen, raw=raw, round_=rnd)),
        ("set_commit", lambda: bv.verify_messages(cpayload, coff, csig, fx["signer20"], fx["hash32"], clen, fx["seal65"],
                                                  raw=raw, round_=rnd)),
    ]
    for _ in range(3):
        for _, f in calls:
            f()
    lat = {k: [] for k, _ in calls}
    kms = {k: [] for k, _ in calls}
    tot = []
    for _ in range(rounds):
        t00 = time.perf_counter()
        for k, f in calls:
            if k in ("senders_prepare", "set_prepare"):
                bv.forget_proposal()   # a new height: its proposal is hashed once
            bv.last_kernel_ms()
            t0 = time.perf_counter()
            f()
            lat[k].append(time.perf_counter() - t0)
            ms, launches = bv.last_kernel_ms()
            kms[k].append(ms)
        tot.append(time.perf_counter() - t00)
    out[name] = {k: {"call_ms_p50": round(float(np.median(lat[k]) * 1e3), 4), "kernel_ms_p50": round(float(np.median(kms[k])), 4)}
                 for k, _ in calls}
    out[name]["sum_of_all_seven_ms_p50"] = round(float(np.median(tot) * 1e3), 4)
    bv.close()
print(json.dumps(out, indent=1))
