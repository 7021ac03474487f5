#!/usr/bin/env python3
"""tools/box_class.py — what distinguishes the two classes of boxes?  Kernel time by HIP events of
  * the warm N = 4 096 kernel, the cold N = 16 384 and N = 65 536 kernels and (control) the cold N = 4 096 kernel,
  * back-to-back (passes 1–10, 11–50, 51–200: a clock ramp shows as a falling series) and with a 3 ms pause between passes
    (a governor that lets clocks fall shows as a slower kernel after every pause),
next to what rocm-smi reports (performance level, power cap, clocks and power right after the load).
Usage (GPU box): python tools/box_class.py > gpurun_out/profiles/rNN_box_class.txt"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_ibft_amd.verifier as V
import go_ibft_amd.simulate as SIM


def smi(*args):
    try:
        out = subprocess.run(["rocm-smi", *args], capture_output=True, text=True, timeout=30).stdout
        return " | ".join(l.split(":", 1)[1].strip() if l.startswith("GPU[0]") else "" for l in out.splitlines() if l.startswith("GPU[0]"))
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"


def series(bv, passes, pause_s=0.0):
    ms = []
    for _ in range(passes):
        bv.last_kernel_ms()
        bv.seals_run()
        k, n = bv.last_kernel_ms()
        ms.append(k / max(n, 1))
        if pause_s:
            time.sleep(pause_s)
    return np.array(ms)


print("# before any load:", smi("--showperflevel"), "||", smi("--showmaxpower"), "||", smi("--showclocks"))
for path, flags, n in (("cold", 0, 4096), ("warm", V.FLAG_PUBKEY_CACHE, 4096), ("cold", 0, 16384), ("cold", 0, 65536),
                       ("warm", V.FLAG_PUBKEY_CACHE, 65536)):
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    try:
        r = SIM.make_round(bv, n, 100 + n)
        bv.set_validators(1, r.addrs, r.power)
        bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
        for _ in range(3):
            bv.seals_run()
        bv.set_kernel_timing(1)
        time.sleep(0.5)                                   # let the device go idle: the series starts from idle clocks
        a = series(bv, 200)
        after = smi("--showclocks") + " || " + smi("--showpower")
        time.sleep(0.5)
        b = series(bv, 60, pause_s=0.003)
        print(f"{path} N={n}: back-to-back kernel ms passes 1-10 {a[:10].mean():.4f}  11-50 {a[10:50].mean():.4f}  51-200 {a[50:].mean():.4f}  "
              f"min {a.min():.4f} | with 3 ms pauses: {b[5:].mean():.4f} (min {b.min():.4f}, max {b.max():.4f})")
        print(f"    right after the back-to-back series: {after}")
    finally:
        bv.close()
