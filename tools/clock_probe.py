#!/usr/bin/env python3
"""tools/clock_probe.py — what the device's clock and power do UNDER each verdict kernel on this lease (the
lane-layout kernels — a field element per lane, 34–51 % v_mad_u64_u32 — run 15–20 % slower on some leases while the row-layout
kernels — 24 % mad, 42 % DPP — and a probe of scattered table reads run the same on all of them).  A sampler thread reads sclk /
power / temperature (rocm-smi's sysfs sources) every 20 ms while one kernel runs back to back for ≈ 2.5 s."""
import glob
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import go_ibft_amd.numa as NUMA  # noqa: E402
print("#", NUMA.pin_to_device_node(0))
import go_ibft_amd.verifier as V  # noqa: E402
import go_ibft_amd.simulate as SIM  # noqa: E402


def sysfs_sources():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if not hw:
            continue
        for name, pat in (("sclk_hz", "freq1_input"), ("power_uw", "power1_average"), ("power_in_uw", "power1_input"), ("temp_mc", "temp1_input"),
                          ("cap_uw", "power1_cap")):
            p = os.path.join(hw[0], pat)
            if os.path.exists(p):
                out.setdefault(card, {})[name] = p
    return out


SRC = sysfs_sources()


def read_all():
    row = {}
    for card, d in SRC.items():
        for k, p in d.items():
            try:
                row[(card, k)] = float(open(p).read())
            except (OSError, ValueError):
                pass
    return row


def smi_once():
    try:
        return subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return repr(e)


def under_load(name, fn, seconds=2.5):
    stop, rows = threading.Event(), []

    def sampler():
        while not stop.is_set():
            rows.append(read_all())
            time.sleep(0.02)
    for _ in range(60):
        fn()
    th = threading.Thread(target=sampler)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
    el = time.perf_counter() - t0
    stop.set()
    th.join()
    keys = sorted({k for r in rows for k in r})
    busy = None
    for card in SRC:                      # the card whose power moved most is the one we drive
        p = [r.get((card, "power_uw"), r.get((card, "power_in_uw"), 0.0)) for r in rows]
        if p and (busy is None or np.mean(p) > busy[1]):
            busy = (card, float(np.mean(p)))
    card = busy[0] if busy else None
    summ = {k[1]: (float(np.mean([r[k] for r in rows if k in r])), float(np.min([r[k] for r in rows if k in r])), float(np.max([r[k] for r in rows if k in r])))
            for k in keys if k[0] == card}
    print(f"{name:34s} {el / n * 1e3:8.4f} ms/step  samples {len(rows)}  card {card}")
    for k, (m, lo, hi) in summ.items():
        scale = {"sclk_hz": 1e6, "power_uw": 1e6, "power_in_uw": 1e6, "cap_uw": 1e6, "temp_mc": 1e3}[k]
        print(f"    {k:12s} mean {m / scale:9.1f}  min {lo / scale:9.1f}  max {hi / scale:9.1f}")


print("# sysfs sources:", {c: sorted(d) for c, d in SRC.items()})
print(smi_once()[:1500])
ctxs = {}
for n, flags, lanes in ((4096, 0, None), (65536, 0, None), (16384, 0, None), (4096, V.FLAG_PUBKEY_CACHE, None)):
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    r = SIM.make_round(bv, n, 4000 + n)
    bv.set_validators(1, r.addrs, r.power)
    bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
    for _ in range(3):
        bv.seals_run()
    ctxs[(n, flags)] = bv
for (n, flags), bv in ctxs.items():
    cold, warm = bv.last_dispatch()
    under_load(f"N={n} {'warm' if flags else 'cold'} lanes {warm if flags else cold}", bv.seals_run)
print(smi_once()[:600])
for bv in ctxs.values():
    bv.close()
