#!/usr/bin/env python3
"""bench.py — committed-seal verifies/sec + quorum latency on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one resident batch: the COMMIT seals of a synthetic
round (ECDSA recover + address compare + validator-set membership → verdict mask, then the
weighted quorum tally), results delivered so the host sees the verdict mask and the quorum flag.
Inputs are resident in HBM when the timed region starts.

  N=1 : BASELINE config #3 — "N=4096 validators, full PREPARE+COMMIT sequence with Keccak
        proposal-hash check, 1×MI355X".  `value` = cold committed-seal verifies/s over the 4096
        resident COMMIT seals (every row recovered, key cache off); `quorum_latency_ms_p50` = p50
        over 1000 rounds of the WHOLE sequence of one height (IsValidValidator on 4095 PREPAREs +
        4096 COMMITs, IsValidProposalHash over both sets, IsValidCommittedSeal + tally: 12 287
        signature checks, host columns → host-visible verdicts, H2D/D2H included) through one
        ibft_verify_messages call per message set, no key known; `quorum_latency` also holds the
        same sequence through the five separate Verifier batches and both forms with the key cache.
  N>1 : weak scaling, 4096 rows per GPU (N=4 is BASELINE config #4: 16 384 validators sharded 4
        ways): every rank verifies its own validator shard, then the verdict-mask words and tally
        partials are all-reduced over RCCL (disjoint shards: sum ≡ OR).  The exchange of pass k runs
        on its own stream and overlaps with the kernels of pass k+1.  At N=8 the line also carries
        `config5`: 65 536 validators, 8192 rows per GPU, 20 % Byzantine seals, checked against the
        CPU oracle outside the timed region.  Round 6: every N > 1 line carries `sharded_sweep` — N_total ∈
        {16 384, 65 536} split over the ranks (config #4 at N = 4, config #5 at N = 8 fall out of it).
  `python bench.py --gpus N` from a bare shell re-launches itself under torch.distributed.run.
  torch is imported only for N > 1 and only as torch.distributed over gloo (communicator id, barriers, max over ranks): the
  process runs ONE HIP runtime and ONE RCCL, the image's ROCm 7.2 ones the library was built for (torch 2.10+rocm7.0 bundles
  its own 7.0.2 copies, and a process that imports torch first binds the library to those); the fences around the timed
  region are the library's own stream syncs (ibft_sync: verdict, exchange and copy streams) — torch's streams never carried
  any of this work.  Every rank pins itself to its GPU's NUMA node first (go_ibft_amd/numa.py: 0.333 vs 0.345 ms per kernel).

Rank 0 prints TWO JSON lines: first the detail record ({"bench_detail": …}: sweep with counters, sequence forms, host-mirror
legs, certificates — also written to gpurun_out/bench_detail.json), then, LAST on stdout and under 6 KB, the headline line the
driver parses (metric, value, config, quorum latency, roofline with the VALU-issue bound, cpu_baseline).
"""
from __future__ import annotations

import argparse
import gc
import glob
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_VERIFY = 118  # 32 hash + 65 sig + 20 signer in, 1 verdict out
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
ROWS_PER_GPU = 4096          # BASELINE configs #3 (1 GPU) and #4 (4 GPUs × 4096)
SEQ_ROUNDS = 1000            # latency p50 over ≥1000 rounds
SWEEP_PREWARM_MIN_PASSES, SWEEP_PREWARM_MIN_S = 100, 0.1   # untimed passes in front of every sweep entry (clock ramp)
PREWARM_STEPS = 150         # untimed passes in front of the W warm-up steps of the headline legs (see run_config)
CONFIG5_TIMEOUT_S = 240     # the N = 8 extra leg (config #5) is abandoned after this long; the headline line goes out regardless
CANARY_HEALTHY_NS = 1.89     # ibft_issue_probe on healthy devices (profiles/r06b_kernel_ab.txt: 1.886 / 1.887 / 1.889 ns in three processes; the
                             # bare instruction stream of tools/ubench_wave.hip measures 1.78 ns on the same lease: the probe's loop and LDS set-up are the 6 %)
KERNEL_TIMING_EVERY = 4      # HIP-event pair around the verdict kernel of every 4th timed pass (≥ 50 samples at --steps 200)
FIXTURE = os.path.join(ROOT, "tests", "golden", "bench_round_n4096.npz")


# IBFT_BENCH_DRYRUN=1 (tests/test_bench_line.py): the rank / world plumbing of this file on CPU ranks over gloo with
# tests/bench_stub.py in place of the verifier — no measurement comes out of it, the record says dry_run
DRY_RUN = os.environ.get("IBFT_BENCH_DRYRUN") == "1"
COMM_INIT_TIMEOUT_S = 180   # ncclCommInitRank blocks until every rank has joined: a rank that never arrives must not hang the job


def bv_module_round(n_total: int, byzantine: bool = False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench_stub
    return bench_stub.make_round(n_total, byzantine)


def comm_init_or_die(bv, uid, rank: int, world: int) -> None:
    """ibft_comm_init with a deadline: on a hang every rank prints what it was waiting for and the process ends non-zero
    (the driver then sees an error line instead of a job that never returns)."""
    import threading
    done = threading.Event()

    def give_up():
        if not done.is_set():
            print(f"bench: rank {rank}/{world}: ibft_comm_init (ncclCommInitRank) has not returned after {COMM_INIT_TIMEOUT_S} s — "
                  f"a rank is missing or the ranks cannot reach each other (HSA_ENABLE_IPC_MODE_LEGACY="
                  f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}); giving up",
                  file=sys.stderr, flush=True)
            os._exit(3)
    t = threading.Timer(COMM_INIT_TIMEOUT_S, give_up)
    t.daemon = True
    t.start()
    try:
        bv.comm_init(uid, rank, world)
    finally:
        done.set()
        t.cancel()


def kernel_name(path: str, cold_lanes: int, warm_lanes: int, cold_table: int = 1) -> str:
    """the kernel's name as rocprofv3 prints it (without spaces); cold_table: ibft_last_cold_table (1 LDS → template
    argument 0, 3 private → 1, 2 private + prefetch → 2)"""
    if path == "cold":
        tab = {1: 0, 3: 1, 2: 2}.get(cold_table, 0)
        return {1: f"ecrecover_lane_kernel<0,{tab}>", 64: "ecrecover_wave_kernel<0>", 128: "ecrecover_wave2_kernel<0>",
                16: "ecrecover_rows_kernel<0>"}.get(cold_lanes, f"ecrecover_group_kernel<0,{cold_lanes},{tab}>")
    return {64: "verify_known_wave_kernel<0>", 1: "verify_known_lane_kernel<0>"}.get(
        warm_lanes, f"verify_known_group_kernel<0,{warm_lanes}>")


def load_round(bv, n_total: int, lo: int, hi: int, byzantine: bool = False):
    """Synthetic round.  N = 4096 comes from the committed fixture; every other size is signed ON THIS
    RANK'S DEVICE by the library's batch signer (go_ibft_amd/simulate.py → ibft_sign_seals: the whole validator table of
    a 65 536-validator round in milliseconds — no rank signs with host code at bench time, nothing of the oracle is on the
    path).  Rows [lo, hi) are this rank's shard."""
    if DRY_RUN:
        r = bv_module_round(n_total, byzantine)
        return {"addrs": r.addrs, "power": r.power, "hash32": r.hash32[lo:hi], "seal65": r.seal65[lo:hi],
                "signer20": r.signer20[lo:hi], "pre": r.pre_flags[lo:hi] if byzantine else None,
                "src": "DRY RUN: unsigned rows, stub verifier", "fx": None, "expect": r.expect[lo:hi]}
    if n_total == 4096 and not byzantine and os.path.exists(FIXTURE):
        with np.load(FIXTURE) as z:
            g = {k: z[k] for k in z.files}   # materialised once: an NpzFile re-reads the archive on every g[k]
        return {"addrs": g["addrs"], "power": g["power"], "hash32": g["hash32"][lo:hi], "seal65": g["seal65"][lo:hi],
                "signer20": g["signer20"][lo:hi], "pre": None, "src": "fixture", "fx": g, "expect": None}
    import go_ibft_amd.simulate as SIM
    r = SIM.make_round(bv, n_total, 1, byzantine=byzantine)
    return {"addrs": r.addrs, "power": r.power, "hash32": r.hash32[lo:hi], "seal65": r.seal65[lo:hi],
            "signer20": r.signer20[lo:hi], "pre": r.pre_flags[lo:hi] if byzantine else None,
            "src": "signed on the device (ibft_sign_seals)", "fx": None, "expect": r.expect[lo:hi]}


def usable_cores() -> int:
    """affinity mask ∧ cgroup CPU quota (the GPU box exposes 256 hardware threads but grants this
    container a 16-CPU quota), not the socket's thread count"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(addrs, power, hash32, seal65, signer20, budget_s: float = 12.0):
    """The CPU oracle (a port — the reference has no implementation of this path and no Go toolchain exists here) timed on
    this box's host cores over a bounded sample: the same COMMIT rows tiled so that every pthread gets ≥512 rows per call.
    Timed through the TUNED recovery (oracle/recover_tuned.inc: 5×52-bit lazy field, endomorphism split + width-5 NAF over a
    common-Z table, safegcd inversions — libsecp256k1-class), after it has agreed with the plain checker path on this box
    over a sample with corrupted rows; otherwise (and always as a second figure) the plain path is timed."""
    from oracle import binding as B
    cores = usable_cores()
    vs = B.ValSet(addrs, power)
    n = len(seal65)
    reps = max(1, (512 * cores + n - 1) // n)   # ≥512 rows per thread so pthread spawn is amortised
    h, s, f = np.tile(hash32, (reps, 1)), np.tile(seal65, (reps, 1)), np.tile(signer20, (reps, 1))
    B.verify_seals(vs, h[:cores], s[:cores], f[:cores], nthreads=cores)  # warm tables / spawn once
    tuned, why = False, None
    try:
        k = min(n, 1024)
        bad = np.array(seal65[:k], copy=True)
        bad[::3, 40] ^= 0x10                     # every third seal corrupted: the agreement covers both verdicts
        want = B.verify_seals(vs, hash32[:k], bad, signer20[:k], nthreads=cores)
        got = B.verify_seals_tuned(vs, hash32[:k], bad, signer20[:k], nthreads=cores)
        tuned = bool((want == got).all()) and 0 < int(want.sum()) < k
        if not tuned:
            why = "verdicts of the tuned path differ from the checker's on this box"
    except Exception as e:  # noqa: BLE001 — the baseline leg must not take the bench line down
        why = repr(e)
    run = B.verify_seals_tuned if tuned else B.verify_seals
    done, t0 = 0, time.perf_counter()
    while True:
        v = run(vs, h, s, f, nthreads=cores)  # one call = reps × N rows
        done += len(v)
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    all_valid = bool(v.all())
    t1 = time.perf_counter()
    run(vs, hash32[:256], seal65[:256], signer20[:256], nthreads=1)
    single = 256 / (time.perf_counter() - t1)
    plain_rate = plain_single = None
    if tuned:                                    # the plain (checker) path beside it, ≈3 s
        d2, t2 = 0, time.perf_counter()
        while time.perf_counter() - t2 < 2.5:
            d2 += len(B.verify_seals(vs, h, s, f, nthreads=cores))
        plain_rate = d2 / (time.perf_counter() - t2)
        t3 = time.perf_counter()
        B.verify_seals(vs, hash32[:256], seal65[:256], signer20[:256], nthreads=1)
        plain_single = 256 / (time.perf_counter() - t3)
    # secondary, independent CPU number: OpenSSL's EC_POINT arithmetic doing the recover, 1 thread
    ossl_rate = None
    try:
        import ctypes
        odir = os.path.join(ROOT, "oracle")
        subprocess.run(["make", "-C", odir, "libopenssl_xcheck.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ol = ctypes.CDLL(os.path.join(odir, "libopenssl_xcheck.so"))
        pub = ctypes.create_string_buffer(64)
        t2, k2 = time.perf_counter(), 0
        while time.perf_counter() - t2 < 1.0:
            i = k2 % n
            assert ol.ossl_ecrecover(hash32[i].tobytes(), seal65[i].tobytes(), pub) == 1
            k2 += 1
        ossl_rate = k2 / (time.perf_counter() - t2)
    except (OSError, AssertionError, AttributeError):
        pass
    out = {"value": done / el, "unit": "verifies/s", "cores": cores, "os_cpu_count": os.cpu_count(), "kind": "port",
           "path": "tuned" if tuned else "plain",
           "tuning": ("tuned C (oracle/recover_tuned.inc): 5x52-bit lazy field, endomorphism split + width-5 NAF over a "
                      "common-Z table of odd multiples, addition-chain square root, safegcd inversions, 32x256 affine table "
                      "of G; agreed with the plain checker path on this box before it was timed") if tuned else
                     "plain C restatement (4x64 limbs, fixed 4-bit windows, Fermat inversions): the checker itself",
           "one_thread": single, "plain_path": plain_rate, "plain_path_one_thread": plain_single,
           "openssl_ec_recover_1thread": ossl_rate,
           "sample": f"{done} seal verifies (the N={n} COMMIT batch tiled x{reps} per call, repeated for "
                     f"{el:.1f} s, {cores} pthreads, {'tuned' if tuned else 'plain'} path); 1 thread: {single:.0f} verifies/s"}
    if why:
        out["tuned_path_refused"] = why
    if not all_valid:
        out["error"] = "a row of the all-valid sample was rejected"
    return out


# wall ns per wave-instruction per SIMD by instruction class at one / two / four resident wavefronts per SIMD, measured with
# tools/ubench_wave.hip — round 5's run (0.25 ms kernels behind 30 ms of untimed launches, median of 7, instructions on 8-byte
# boundaries as go-ibft_amd/phase_align.py leaves them): profiles/r05a_ubench_wave.txt.  Round 2's numbers (2.07 / 2.22 / 2.50 at
# one wavefront) came from 34 µs kernels and carried the launch ramp: the kernels BEAT the ceilings priced with them.
ISSUE_NS = {1: {"plain": 1.79, "dpp": 1.79, "mad": 1.80}, 2: {"plain": 0.89, "dpp": 1.74, "mad": 1.75},
            4: {"plain": 0.89, "dpp": 1.70, "mad": 1.73}}
ISSUE_NS_SOURCE = "built-in (profiles/r05a_ubench_wave.txt)"


def _load_issue_ns():
    """the class times of the newest committed profiles/r*_ubench_wave.txt that has the round-5 class lines"""
    global ISSUE_NS, ISSUE_NS_SOURCE
    names = {"class 4-byte": "plain", "class DPP aligned": "dpp", "class mad aligned": "mad"}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_wave.txt")), reverse=True):
        got = {}
        try:
            for line in open(path):
                for pre, cls in names.items():
                    if line.startswith(pre):
                        m = re.search(r"waves/SIMD=(\d+).*wall: ([0-9.]+) ns per inst per SIMD", line)
                        if m:
                            got.setdefault(int(m.group(1)), {})[cls] = float(m.group(2))
        except OSError:
            continue
        if all(len(got.get(w, {})) == 3 for w in (1, 2, 4)):
            ISSUE_NS, ISSUE_NS_SOURCE = got, os.path.relpath(path, ROOT)
            return


_load_issue_ns()
LANES_OF = {"ecrecover_wave2_kernel": 128, "ecrecover_lane_kernel": 1, "verify_known_lane_kernel": 1, "ecrecover_rows_kernel": 16, "ecrecover_wave_kernel": 64,
            "verify_known_wave_kernel": 64}
# wavefronts a SIMD can hold (512 registers per lane: go-ibft_amd/csrc resource usage, tools/occupancy.py)
RESIDENT_CAP = {"ecrecover_lane_kernel": 1, "verify_known_lane_kernel": 2, "ecrecover_rows_kernel": 2, "ecrecover_wave_kernel": 2,
                "verify_known_wave_kernel": 2, "ecrecover_group_kernel": 1, "verify_known_group_kernel": 2}


LIVE_TAG = None   # set by --profile: the counter files of THIS run's rocprofv3 sub-steps (gpurun_out/profiles/LIVE_TAG_*)


def _attachment_files(suffix: str):
    """counter summaries to read, best first: this run's own (--profile), then the committed ones, newest first"""
    live = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "profiles", f"{LIVE_TAG}_{suffix}"))) if LIVE_TAG else []
    return live + sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{suffix}")), reverse=True)


LIVE_BUDGET_S = 200   # the default run's counter sub-steps are abandoned after this long (the line then attaches the committed files)


def collect_live_counters(rows: int, lean: bool) -> tuple[str | None, dict]:
    """The rocprofv3 passes of tools/profile.sh (kernel stats, FETCH_SIZE, WRITE_SIZE: separate runs) and tools/pmc_wave.sh
    (instruction counters, one counter set per run) as sub-steps of THIS run, over the same command line they always use; their
    summaries land in gpurun_out/profiles/live_n<rows>_* and are what this line then attaches.  lean (the default run, round 6:
    the driver's own line must carry counters from the driver's own box): the headline stats pass, the two traffic passes and
    two counter sets — five short processes, ≈ 40 s; --profile: the full series (≈ 2 minutes).  Returns (tag or None, what happened)."""
    import shutil
    info = {"mode": "lean" if lean else "full"}
    if shutil.which("rocprofv3") is None:
        info["skipped"] = "rocprofv3 is not on PATH"
        return None, info
    tag = f"live_n{rows}"
    for old_file in glob.glob(os.path.join(ROOT, "gpurun_out", "profiles", tag + "_*")):   # never a previous run's files
        try:
            os.remove(old_file)
        except OSError:
            pass
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, LEAN="1" if lean else "0")
    t0 = time.time()
    for cmd in (["bash", os.path.join(ROOT, "tools", "profile.sh"), tag, str(rows)],
                ["bash", os.path.join(ROOT, "tools", "pmc_wave.sh"), str(rows), tag]):
        left = (LIVE_BUDGET_S if lean else 900) - (time.time() - t0)
        if left < 20:
            info["skipped_step"] = os.path.basename(cmd[1]) + ": time budget spent"
            break
        try:
            # (a process group of its own: on a timeout the whole sub-tree goes, not just the shell)
            pr = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                rc = -9
        except OSError as e:
            rc, info["error"] = -1, repr(e)
        if rc != 0:
            info[os.path.basename(cmd[1])] = f"returned {rc}"
            print(f"bench: {' '.join(cmd[1:])} returned {rc}; attaching the committed counters where this run has none", file=sys.stderr)
    info["seconds"] = round(time.time() - t0, 1)
    have = [os.path.basename(x) for x in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "profiles", tag + "_*")))]
    info["files"] = have
    return (tag if have else None), info


def rocprof_avg_kernel_ms(kname: str):
    """average duration of `kname` in THIS run's rocprofv3 --kernel-trace --stats summary (the contract: it must agree with
    the HIP-event average) → (ms, calls) or None"""
    import csv
    if not LIVE_TAG:
        return None
    path = os.path.join(ROOT, "gpurun_out", "profiles", f"{LIVE_TAG}_kernel_stats.csv")
    needle = kname.replace(",", ", ")
    try:
        for row in csv.DictReader(open(path)):
            if needle in row.get("Name", ""):
                return float(row["AverageNs"]) / 1e6, int(row["Calls"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def _static_mix(kname: str):
    """(v_mad_u64_u32 share, DPP share, s_nop per VALU) of a kernel's hot loops from the newest profiles/r*_static_mix.txt"""
    base = kname.split("<")[0]
    args = re.findall(r"\d+", kname.split("<")[1]) if "<" in kname else []
    mangled = base + "ILi" + "ELi".join(args) + "E" if args else base
    for sp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_static_mix.txt")), reverse=True):
        st = open(sp).read()
        for sec in st.split("# kernel ")[1:]:
            if mangled in sec.splitlines()[0]:
                try:
                    f_mad = float(re.search(r"v_mad_u64_u32 share of VALU[^:]*: ([0-9.]+)", sec).group(1))
                    f_dpp = float(re.search(r"DPP share of VALU[^:]*: ([0-9.]+)", sec).group(1))
                    nop = float(re.search(r"s_nop per VALU instruction[^:]*: ([0-9.]+)", sec).group(1))
                except AttributeError:
                    continue
                if f_mad > 0:
                    return f_mad, f_dpp, nop, os.path.relpath(sp, ROOT)
    if "wave_kernel" in kname or "wave2_kernel" in kname:      # the one- / two-wavefront-per-signature kernels run the row-layout code (wfe_mul) inside calls
        return _static_mix("ecrecover_rows_kernel<0>")
    return None


def valu_issue(kname: str, rows: int, avg_kernel_s: float):
    """The bound that applies to this path (integer VALU issue), per launch: VALU wave-instructions from the newest
    profiles/r*_pmc_instruction_mix.txt that holds this kernel (rocprofv3 --pmc SQ_INSTS_VALU of this same command on the
    builder's box — a file attachment: PMCs cannot be read in-process — scaled by rows when the file's batch size differs:
    the count per signature does not depend on the batch), priced against
      * the guide's issue peak: 1 024 SIMDs x one wave64 VALU instruction per 2 cycles at 2.4 GHz;
      * the mix-weighted ceiling at FULL occupancy (4 wavefronts per SIMD) for this kernel's instruction mix;
      * the mix-weighted ceiling at the occupancy THIS launch reaches (wavefronts offered per SIMD, capped by registers)."""
    base = kname.split("<")[0]
    kshort = re.escape(base) + r"<[^>]*>" if "<" in kname else re.escape(base)
    exact = re.escape(kname.replace(",", ", ")) if ("group" in kname or "lane_kernel" in kname) else kshort
    insts = salu = None
    src = scaled = None
    for path in _attachment_files("pmc_instruction_mix.txt"):
        try:
            text = open(path).read()
        except OSError:
            continue
        m = re.search(exact + r"\s+SQ_INSTS_VALU\s+([0-9.]+) per launch", text)
        if not m:
            continue
        m_rows = re.search(r"^# rows per launch: (\d+)", text, re.M)
        file_rows = int(m_rows.group(1)) if m_rows else 1024
        if insts is not None and file_rows != rows:
            continue                                  # an exact-size file wins over the newest one
        ms = re.search(exact + r"\s+SQ_INSTS_SALU\s+([0-9.]+) per launch", text)
        insts, salu = float(m.group(1)) * rows / file_rows, (float(ms.group(1)) if ms else 0.0) * rows / file_rows
        src, scaled = os.path.relpath(path, ROOT), (None if file_rows == rows else file_rows)
        if file_rows == rows:
            break
    mix = _static_mix(kname)
    if insts is None or mix is None:
        return None
    f_mad, f_dpp, nop, mix_src = mix
    lanes = LANES_OF.get(base) or int(re.findall(r"\d+", kname)[1])
    offered = rows * lanes / 64 / 1024
    cap = 2 if kname.replace(" ", "").startswith("ecrecover_lane_kernel<0,1") else RESIDENT_CAP.get(base, 2)   # (table in the private segment, no prefetch: 256 registers)
    resident = max(1, min(cap, int(offered + 0.999)))

    def ceiling(waves):
        t = ISSUE_NS[waves]
        return 1024 / ((f_mad * t["mad"] + f_dpp * t["dpp"] + (1.0 - f_mad - f_dpp) * t["plain"]) * 1e-9)
    ach = insts / avg_kernel_s
    peak_guide = 1024 * 2.4e9 / 2.0
    # fewer wavefronts than SIMDs (small sets): only the SIMDs that hold one can issue
    full, here = ceiling(4), ceiling(1 if resident < 2 else 2) * min(1.0, offered)
    return {"wave_insts_per_launch": insts, "salu_insts_per_launch": salu, "achieved_ginst_s": ach / 1e9,
            "peak_guide_ginst_s": peak_guide / 1e9, "frac_of_guide_peak": ach / peak_guide,
            "ceiling_full_occupancy_ginst_s": full / 1e9, "frac_of_full_occupancy_ceiling": ach / full,
            "ceiling_at_this_occupancy_ginst_s": here / 1e9, "frac_of_ceiling_at_this_occupancy": ach / here,
            "wavefronts_offered_per_simd": offered, "wavefronts_resident_per_simd": resident,
            "simds_with_a_wavefront": int(min(1024, round(offered * 1024))),
            "mad_share": f_mad, "dpp_share": f_dpp, "s_nop_per_valu": nop,
            "note": "peak_guide = 1024 SIMDs x one wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md); the "
                    "ceilings price THIS kernel's instruction mix with the issue times tools/ubench_wave.hip measured per class "
                    f"(ns per instruction per SIMD at 1 / 2 / 4 resident wavefronts: {ISSUE_NS}; {ISSUE_NS_SOURCE})",
            "issue_ns_source": ISSUE_NS_SOURCE,
            "source": src, "instruction_count_scaled_from_rows": scaled, "mix_source": mix_src}


def profile_attachments(kname: str, rows: int, avg_kernel_s: float):
    """PMC counters cannot be read from inside this process: the per-launch values measured with rocprofv3
    (separate passes of this same command: tools/profile.sh → profiles/r*_traffic.json, tools/pmc_wave.sh →
    profiles/r*_pmc_instruction_mix.txt) are attached when kernel and batch size match."""
    traffic = None
    try:
        needle = "ibftk::" + kname.replace(",", ", ")
        for path in _attachment_files("traffic.json"):
            ent = next((v for k, v in json.load(open(path)).items() if needle in k), None)
            if ent and ent.get("rows") == rows:
                traffic = ent["hbm_bytes_per_launch"]
                break
    except (OSError, ValueError):
        pass
    try:
        valu = valu_issue(kname, rows, avg_kernel_s)
    except (OSError, ValueError, AttributeError, IndexError):
        valu = None
    return traffic, valu


def sequence_latency(V, fx, flags: int, rounds: int, form: str = "calls", pinned: bool = True):
    """BASELINE config #3: the whole PREPARE + COMMIT sequence of one height, host columns → host-visible verdicts.
    form "calls": the five Verifier batches at the moments the reference evaluates them (call sites core/ibft.go:1128
    ×2 sets, :858-861, :938, :943 + validator_manager.go:77-96).  form "sets": the same verdicts from two
    ibft_verify_messages calls — one per message set, envelope signatures and committed seals in one verdict launch.
    Returns p50 / p10 / p90 in ms."""
    n = len(fx["addrs"])
    raw, rnd = fx["raw"].tobytes(), int(fx["round"])
    # the columns as the caller's flatten step leaves them: page-locked buffers from ibft_pinned_alloc (the integration's
    # prescription) or ordinary pageable memory (what round 1 measured)
    col = V.pinned_copy if pinned else (lambda a: np.frombuffer(a, np.uint8).copy() if isinstance(a, bytes) else np.ascontiguousarray(a).copy())
    ppayload, poff, psig = col(fx["prepare_payload"].tobytes()), col(fx["prepare_off"]), col(fx["prepare_sig65"])
    cpayload, coff, csig = col(fx["payload"].tobytes()), col(fx["off"]), col(fx["msg_sig65"])
    pfrom, phash = col(fx["addrs"][1:]), col(fx["hash32"][1:])
    chash, cseal, cfrom = col(fx["hash32"]), col(fx["seal65"]), col(fx["signer20"])
    plen, clen = col(np.full(n - 1, 32, np.uint8)), col(np.full(n, 32, np.uint8))
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    try:
        bv.set_validators(int(fx["height"]), fx["addrs"], fx["power"])

        def five_calls():
            bv.forget_proposal()                          # a new height has a new proposal: hashed once per sequence
            a, _ = bv.is_valid_validator(ppayload, poff, psig, pfrom)                 # PREPARE ingest
            b = bv.is_valid_proposal_hash(raw, rnd, phash, plen)                      # handlePrepare
            c, _ = bv.is_valid_validator(cpayload, coff, csig, cfrom)                 # COMMIT ingest
            d = bv.is_valid_proposal_hash(raw, rnd, chash, clen)                      # handleCommit a1
            e, t = bv.is_valid_committed_seal(chash, cseal, cfrom)                    # a2 + tally
            return a.all() and b.all() and c.all() and d.all() and e.all(), t

        # the two set calls with their argument marshalling done once (fixed column buffers, refilled every round, are what
        # the integration prescribes): a round is two C calls that return verdict WORDS; decoding them into one bool per row
        # for the assertion below is the harness's business, not the path's
        run_p = bv.prepare_messages(ppayload, poff, psig, pfrom, phash, plen, raw=raw, round_=rnd)             # PREPARE set
        run_c = bv.prepare_messages(cpayload, coff, csig, cfrom, chash, clen, cseal, raw=raw, round_=rnd)      # COMMIT set

        def two_sets():
            bv.forget_proposal()                          # a new height has a new proposal: hashed once per sequence
            run_p()
            sw, vw, t = run_c()
            return t.valid_rows == n and t.has_quorum == 1, t

        def two_sets_checked():
            ws, wv, _ = run_p()
            ok = V.mask_to_bool(ws, n - 1).all() and V.mask_to_bool(wv, n - 1).all()
            ws, wv, t = run_c()
            return ok and V.mask_to_bool(ws, n).all() and V.mask_to_bool(wv, n).all(), t
        sequence = five_calls if form == "calls" else two_sets
        if form != "calls":
            ok, t = two_sets_checked()
            assert ok and t.has_quorum == 1 and t.valid_rows == n
        for _ in range(3):                                # warm path: the second pass builds the tables
            ok, t = sequence()
        assert ok and t.has_quorum == 1 and t.valid_rows == n
        lat = np.empty(rounds)
        for i in range(rounds):
            t0 = time.perf_counter()
            sequence()
            lat[i] = time.perf_counter() - t0
        dispatch = bv.last_dispatch()
    finally:
        bv.close()
    q = np.percentile(lat * 1e3, [10, 50, 90])
    return {"p50_ms": float(q[1]), "p10_ms": float(q[0]), "p90_ms": float(q[2]), "rounds": rounds,
            "c_abi_calls": 5 if form == "calls" else 2, "host_columns": "pinned (ibft_pinned_alloc)" if pinned else "pageable",
            "signatures_per_sequence": 3 * n - 1, "sig_verifies_per_s": (3 * n - 1) / float(np.median(lat)),
            "dispatch_cold_warm_lanes": list(dispatch)}


def sweep_sizes(V, sizes=(64, 256, 1024, 4096, 16384, 65536), steps: int = 50, warmup: int = 5):
    """north star: "sig-verifies/sec on synthetic rounds of N ∈ {64 … 65 536} … as absolute numbers and as fraction of
    HBM roofline".  One resident COMMIT batch per size (signed on the device), `steps` synchronous passes each through the
    cold path (every row recovered) and the warm path (keys known), kernel time from HIP events around every pass."""
    import go_ibft_amd.simulate as SIM
    out = []
    for path, flags in (("cold", 0), ("warm", V.FLAG_PUBKEY_CACHE)):
        bv = V.BatchVerifier(flags=flags, max_rows=max(sizes))
        try:
            for i, n in enumerate(sizes):
                r = SIM.make_round(bv, n, 100 + n)
                bv.set_validators(1, r.addrs, r.power)
                bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
                for _ in range(max(warmup, 3)):          # warm: pass 1 learns the keys, pass 2 builds the tables
                    verdict, t = bv.seals_run()
                assert verdict.all() and t.has_quorum == 1 and t.distinct_senders == n
                # untimed passes until the clocks have settled (profiles/r04u_box_class.txt: on a box whose governor is on
                # "auto" the first ≈50 passes after a pause run up to 12 % slower — N = 16 384 cold 0.819 → 0.736 ms)
                t_pre, n_pre = time.perf_counter(), 0
                while n_pre < SWEEP_PREWARM_MIN_PASSES or time.perf_counter() - t_pre < SWEEP_PREWARM_MIN_S:
                    bv.seals_run()
                    n_pre += 1
                for _ in range(2):                       # the pipeline's own first-use work (result slots, events, the tally's
                    bv.seals_submit()                    # stream) belongs to no size: two untimed pipelined passes
                    bv.seals_collect()
                bv.set_kernel_timing(1)
                bv.last_kernel_ms()
                t0 = time.perf_counter()
                bv.seals_submit()                        # one pass kept in flight, as in the headline leg
                for _ in range(steps - 1):
                    bv.seals_submit()
                    bv.seals_collect()
                bv.seals_collect()
                el = time.perf_counter() - t0
                kms, kl = bv.last_kernel_ms()
                cold_l, warm_l = bv.last_dispatch()
                if path == "warm":
                    bv.cache_stats()
                    warm_l = bv.lanes_per_signature
                kern = kms / max(kl, 1) / 1e3
                ent = out[i] if path == "warm" else {"validators": n}
                kn = kernel_name(path, cold_l, warm_l, bv.last_cold_table())
                try:
                    vi = valu_issue(kn, n, kern)
                except (OSError, ValueError, AttributeError, IndexError):
                    vi = None
                if vi:   # the sweep keeps the numbers, the headline object the prose
                    vi = {k: v for k, v in vi.items() if k not in ("note",)}
                ent[path] = {"verifies_per_s": n * steps / el, "ms_per_step": el / steps * 1e3, "kernel_ms": kern * 1e3,
                             "kernel": kn,
                             "hbm_gb_s": n * ALGO_BYTES_PER_VERIFY / kern / 1e9,
                             "hbm_frac": n * ALGO_BYTES_PER_VERIFY / kern / 1e9 / HBM_PEAK_GBS,
                             "valu_issue": vi}
                if path == "cold":
                    out.append(ent)
        finally:
            bv.close()
    return {"steps_per_size": steps, "definition": "one resident COMMIT batch of N seals per step (recover/verify + tally, "
            "results host-visible, one pass kept in flight), inputs signed on the device; hbm_* = 118 B x N / verdict-kernel time (HIP events, every pass); "
            f"every entry behind >= {SWEEP_PREWARM_MIN_PASSES} untimed passes / {SWEEP_PREWARM_MIN_S} s (clock ramp)",
            "sizes": out}


def small_n_leg(V, sizes=(4, 6, 8, 12, 16, 30, 64, 128, 256), calls: int = 300):
    """What a node with the reference's OWN validator counts gets (round-5 review, item 6): one round of COMMIT seals at N = 4, 6,
    30 (core/consensus_test.go:139, core/byzantine_test.go:21, core/rapid_test.go:156) and the sizes around them, host columns →
    host-visible verdicts through ONE ibft_verify_seals call (H2D, launch, kernel, tally, D2H: what a BatchVerifier call costs),
    cold (every row recovered) and warm (keys known), p50 of `calls` calls behind 100 untimed ones — next to the TUNED CPU
    recovery (oracle/recover_tuned.inc, libsecp256k1-class) on ONE core for the same rows: what the Go closure of
    core/ibft.go:932-944 costs with go-ethereum's secp256k1 behind IsValidCommittedSeal.  crossover_* = the smallest N at which
    the device call is faster than that loop — the value for IBFT_MIN_DEVICE_ROWS (shim/go/core/backend_batch.go,
    go-ibft_amd/host/backend.hpp).  The CPU figure uses the oracle as the CPU baseline, like cpu_baseline."""
    from oracle import binding as B, workload as W
    out = {"definition": "p50 ms of ONE ibft_verify_seals call (pinned host columns -> host-visible verdict words + tally) on N rows; "
                         "cold = key cache off, warm = keys known; cpu_one_core_ms = the same rows through the tuned CPU recovery on one "
                         "thread (p50 of 30); crossover_* = smallest N with device < cpu (the IBFT_MIN_DEVICE_ROWS to set)",
           "sizes": []}
    ctx = {"cold": V.BatchVerifier(flags=0, max_rows=1024), "warm": V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)}
    try:
        for n in sizes:
            r = W.make_round(n, 4000 + n)
            vs = B.ValSet(r.addrs, r.power)
            cols = tuple(V.pinned_copy(x) for x in (r.hash32, r.seal65, r.signer20))
            ent = {"validators": n}
            for name, bv in ctx.items():
                bv.set_validators(n, r.addrs, r.power)
                for _ in range(100):
                    got, t = bv.is_valid_committed_seal(*cols)
                assert got.all() and t.has_quorum == 1 and t.distinct_senders == n
                lat = np.empty(calls)
                for i in range(calls):
                    t0 = time.perf_counter()
                    bv.is_valid_committed_seal(*cols)
                    lat[i] = time.perf_counter() - t0
                ent[name + "_ms_p50"] = float(np.median(lat) * 1e3)
            cpu = np.empty(30)
            for i in range(30):
                t0 = time.perf_counter()
                v = B.verify_seals_tuned(vs, r.hash32, r.seal65, r.signer20, nthreads=1)
                cpu[i] = time.perf_counter() - t0
            assert v.all()
            ent["cpu_one_core_ms"] = float(np.median(cpu) * 1e3)
            out["sizes"].append(ent)
    finally:
        for bv in ctx.values():
            bv.close()
    for name in ("cold", "warm"):
        out["crossover_" + name] = next((e["validators"] for e in out["sizes"] if e[name + "_ms_p50"] < e["cpu_one_core_ms"]), None)
    return out


def sustained_leg(V, rd, steps: int = 400, sizes=(4096, 65536)):
    """Round-4 review, item 7: the headline is "inputs resident in HBM"; a node gets NEW messages at every wake-up
    (core/ibft.go:931-946), so a sustained stream pays an upload per batch.  With the context's two staging slots
    (ibft_seals_stage_next / ibft_seals_swap) the copy of batch k+1 runs on a copy stream while the kernels of batch k work:
    per step launch(k) → stage_next(k+1) → fetch(k) → swap, a FRESH host batch (pinned columns) every step — two batches
    alternate, the second with one seal corrupted so that every step's tally proves whose results it is.  Reported next to
    the resident-batch rate of the SAME context measured right before it (same box, same clocks)."""
    import go_ibft_amd.simulate as SIM
    out = {"definition": "verifies/s with a fresh host batch every step (H2D of the 117 B/row columns included, overlapped "
                         "through the second staging slot); resident = the same context re-running one resident batch",
           "sizes": []}
    for n in sizes:
        bv = V.BatchVerifier(flags=0, max_rows=max(n, 1024))
        try:
            if n == len(rd["seal65"]) and rd.get("pre") is None:
                addrs, power, h, s, f = rd["addrs"], rd["power"], rd["hash32"], rd["seal65"], rd["signer20"]
            else:
                r = SIM.make_round(bv, n, 300 + n)
                addrs, power, h, s, f = r.addrs, r.power, r.hash32, r.seal65, r.signer20
            bv.set_validators(1, addrs, power)
            s_bad = np.array(s, copy=True)
            s_bad[n // 2, 5] ^= 0x40                                     # batch B: one forged seal
            A = tuple(V.pinned_copy(x) for x in (h, s, f))
            Bt = tuple(V.pinned_copy(x) for x in (np.roll(h, 1, axis=0), np.roll(s_bad, 1, axis=0), np.roll(f, 1, axis=0)))
            host, expect = (A, Bt), (n, n - 1)
            bv.seals_stage(*A)
            for _ in range(PREWARM_STEPS):
                bv.seals_run()
            gc.collect(); gc.disable()
            t0 = time.perf_counter()
            for _ in range(steps):
                bv.seals_run()
            resident = time.perf_counter() - t0
            # the stream: two untimed rounds to create the copy stream and the spare columns, then the timed steps
            k = 0
            for _ in range(4):
                bv.seals_launch(1); bv.seals_stage_next(*host[(k + 1) & 1]); _, t = bv.seals_fetch(); bv.seals_swap(); k += 1
            ok = True
            t0 = time.perf_counter()
            for _ in range(steps):
                bv.seals_launch(1)
                bv.seals_stage_next(*host[(k + 1) & 1])
                _, t = bv.seals_fetch()
                ok &= t.valid_rows == expect[k & 1] and t.has_quorum == 1
                bv.seals_swap()
                k += 1
            stream = time.perf_counter() - t0
            gc.enable()
            assert ok, "a step of the stream delivered another batch's tally"
            out["sizes"].append({"validators": n, "steps": steps, "value": n * steps / stream, "ms_per_step": stream / steps * 1e3,
                                 "resident_value": n * steps / resident, "resident_ms_per_step": resident / steps * 1e3,
                                 "vs_resident": resident / stream, "h2d_bytes_per_step": int(n * 117), "slots": 2})
        finally:
            bv.close()
    head = out["sizes"][0]
    out.update({k: head[k] for k in ("value", "ms_per_step", "vs_resident", "slots")})
    out["unit"] = "verifies/s"
    return out


def set_change_leg(V, n: int = 4096, reps: int = 3, passes: int = 5):
    """the key cache's cost model (warm numbers assume a learned set): wall time of the first `passes` synchronous COMMIT
    passes after ibft_set_validators installs n validators the device has never seen — pass 1 recovers every row and
    learns the keys, pass 2 builds the per-validator tables (qtab_build_kernel) and verifies with them, later passes are
    warm — and what the tables hold in HBM; median over `reps` fresh sets."""
    import go_ibft_amd.simulate as SIM
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=n)
    try:
        ms = np.empty((reps, passes))
        b0, u0 = bv.cache_memory()[:2]
        for k in range(reps):
            r = SIM.make_round(bv, n, 7000 + 13 * k)
            bv.set_validators(1, r.addrs, r.power)
            bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
            bv.sync()
            for j in range(passes):
                t0 = time.perf_counter()
                verdict, t = bv.seals_run()
                ms[k, j] = (time.perf_counter() - t0) * 1e3
                assert verdict.all() and t.has_quorum == 1
        held, used, alloc, _ = bv.cache_memory()
        tables, warm_p, cold_p = bv.cache_stats()
    finally:
        bv.close()
    return {"definition": "ms of each of the first passes over one resident COMMIT batch after a full validator-set change "
                          "(no key known): pass 1 = recover + learn, pass 2 = build tables + warm verify, then warm",
            "validators": n, "pass_ms": [float(x) for x in np.median(ms, axis=0)], "fresh_sets": reps,
            "table_bytes_per_validator": int((held - b0) // (used - u0)) if used > u0 and held > b0 else None,
            "cache_bytes_held": int(held), "slots_in_use": int(used), "slots_allocated": int(alloc),
            "warm_passes": int(warm_p), "cold_passes": int(cold_p)}


def certificates_leg(V, n: int = 256, reps: int = 30):
    """§8f rank 2: the ROUND-CHANGE messages of one round change at n validators — Q = ⌊2n/3⌋+1 messages, each with a
    PreparedCertificate of Q messages: Q·(Q+1) signatures — as the transport's bytes → the verdict of every nested
    message (ibft_verify_certificates_wire), host bytes → host-visible verdicts, p50 of `reps` calls."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cert_cases as CC                      # input generation: messages signed by the oracle's signer (≈ 2Q signatures)
    from oracle import wire, workload as W
    r = W.make_round(n, 900 + n, height=5, round_=1, raw_len=1024)
    q = (2 * n) // 3 + 1
    pm = CC.preprepare(r, 1, 5, 1)
    prepares = [CC.prepare(r, j, 5, 1) for j in range(n) if j != 1][: q - 1]
    pcb = wire.prepared_certificate(pm, prepares)
    rcs = [CC.round_change(r, i, 5, 2, wire.Proposal(r.raw, 1), pcb).encode() for i in range(q)]
    buf, off = CC.pack(rcs)
    rows_expected = q * (q + 1)
    res = {"validators": n, "round_change_messages": q, "signatures": rows_expected, "wire_bytes": len(buf)}
    for name, flags in (("cold", 0), ("warm", V.FLAG_PUBKEY_CACHE)):
        bv = V.BatchVerifier(flags=flags, max_rows=max(65536, rows_expected + 64))
        try:
            bv.set_validators(5, r.addrs, r.power)
            for _ in range(3):
                k, _, _, cls, snd, _, _ = bv.verify_certificates_wire(buf, off, rows_expected + 64, want_rows=False)
            assert k == rows_expected and snd.all() and not cls.any()
            t = []
            for _ in range(reps):
                t0 = time.perf_counter()
                bv.verify_certificates_wire(buf, off, rows_expected + 64, want_rows=False)
                t.append(time.perf_counter() - t0)
            p = np.percentile(np.array(t) * 1e3, [10, 50, 90])
            res[name] = {"p50_ms": float(p[1]), "p10_ms": float(p[0]), "p90_ms": float(p[2]),
                         "signatures_per_s": rows_expected / (p[1] * 1e-3)}
        finally:
            bv.close()
    return res


LINE_LIMIT = 6144   # the driver keeps the last 8 081 bytes of stdout: the LAST line must parse on its own (round-4 review)
DETAIL_PATH = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms", "kernel_samples",
                  "rocprof_avg_kernel_ms", "rocprof_calls", "algorithmic_bytes_per_launch", "counters", "counters_live")
_VALU_KEYS = ("wave_insts_per_launch", "achieved_ginst_s", "peak_guide_ginst_s", "frac_of_guide_peak",
              "frac_of_full_occupancy_ceiling", "frac_of_ceiling_at_this_occupancy", "wavefronts_resident_per_simd", "source")


def _r(x, nd=6):
    """floats of the headline line at 6 significant digits: the line is for reading and parsing, the detail file keeps all"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def headline_record(rec: dict) -> dict:
    """The driver's line: SURVEY §8d's quantities and nothing else — metric / value / config, quorum latency, the roofline of the
    dominant kernel with the VALU-issue bound that applies, the CPU baseline; one number per sweep size.  Everything else
    (sweep with counters, extended sample, sequence forms, set change, certificates, host-mirror legs, prose) is the DETAIL
    record: gpurun_out/bench_detail.json and an EARLIER stdout line."""
    out = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data") if k in rec}
    cfg = rec.get("config", {})
    out["config"] = {k: cfg[k] for k in ("workload", "validators", "rows_per_gpu", "path", "prewarm_steps", "kernel",
                                         "parallelism", "pipeline", "numa_pin") if k in cfg}
    for k in ("quorum_latency_ms_p50", "step_latency_ms_p50", "step_latency_ms_p50_incl_h2d", "rccl_nranks", "rccl_rank0_device"):
        if k in rec:
            out[k] = rec[k]
    rf = rec.get("roofline") or {}
    out["roofline"] = {k: rf[k] for k in _ROOFLINE_KEYS if k in rf}
    if rf.get("valu_issue"):
        out["roofline"]["valu_issue"] = {k: rf["valu_issue"][k] for k in _VALU_KEYS if k in rf["valu_issue"]}
    cb = rec.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "path", "sample", "os_cpu_count", "plain_path", "tuned_path_refused", "error") if k in cb}
    if rec.get("extended"):   # the 400-step sample of the same leg (the K-step leg above has K/4 kernel-time samples)
        out["extended"] = {k: rec["extended"][k] for k in ("steps", "value", "ms_per_step", "avg_kernel_ms", "kernel_samples")
                           if k in rec["extended"]}
    if rec.get("device_canary"):
        out["device_canary"] = {k: rec["device_canary"][k] for k in ("issue_ns", "healthy_ns", "off", "flag", "error")
                                if k in rec["device_canary"]}
    ql = rec.get("quorum_latency") or {}
    if "message_sets_warm" in ql:
        out["quorum_latency_warm_ms_p50"] = ql["message_sets_warm"].get("p50_ms")
    if "warm_path" in rec:
        out["warm_path"] = {k: rec["warm_path"][k] for k in ("value", "ms_per_step", "kernel_ms", "kernel") if k in rec["warm_path"]}
    if "sustained_incl_h2d" in rec:
        out["sustained_incl_h2d"] = {k: rec["sustained_incl_h2d"][k] for k in ("value", "ms_per_step", "slots", "vs_resident")
                                     if k in rec["sustained_incl_h2d"]}
    sw = (rec.get("sweep") or {}).get("sizes")
    if sw:   # one row per size: [N, cold verifies/s, cold kernel ms, warm verifies/s, warm kernel ms]
        out["sweep"] = [[e["validators"], e.get("cold", {}).get("verifies_per_s"), e.get("cold", {}).get("kernel_ms"),
                         e.get("warm", {}).get("verifies_per_s"), e.get("warm", {}).get("kernel_ms")] for e in sw]
        out["sweep_columns"] = "N, cold verifies/s, cold kernel ms, warm verifies/s, warm kernel ms"
    sn = rec.get("small_n") or {}
    if sn.get("sizes"):   # one row per size: [N, cold call ms, warm call ms, one CPU core ms] + the crossovers
        out["small_n"] = {"rows": [[e["validators"], e.get("cold_ms_p50"), e.get("warm_ms_p50"), e.get("cpu_one_core_ms")] for e in sn["sizes"]
                                   if e["validators"] in (4, 6, 30)],
                          "columns": "N, one device call cold ms, warm ms, the same rows on ONE CPU core ms",
                          "crossover_cold": sn.get("crossover_cold"), "crossover_warm": sn.get("crossover_warm")}
    if "config5" in rec:
        c5 = rec["config5"]
        out["config5"] = {k: c5[k] for k in ("validators", "rows_per_gpu", "byzantine_fraction", "rccl_nranks", "value",
                                             "ms_per_step", "kernel", "valid_fraction", "error") if k in c5}
    ss = (rec.get("sharded_sweep") or {}).get("sizes")
    if ss:   # one row per size: [N_total, rows per GPU, verifies/s, ms per step, synchronous step p50 ms, kernel] (or the error)
        out["sharded_sweep"] = [[e.get("validators"), e.get("rows_per_gpu"), e.get("value"), e.get("ms_per_step"),
                                 e.get("step_latency_ms_p50"), e.get("kernel")] if "error" not in e else
                                [e.get("validators"), "error", e["error"][:120]] for e in ss]
        out["sharded_sweep_columns"] = "N_total, rows/GPU, verifies/s, ms/step, sync step p50 ms, kernel"
    if "build" in rec:
        out["build"] = rec["build"]
    if rec.get("dry_run"):
        out["dry_run"] = True
    out["parity"] = "numerics unpinned by the reference (DESIGN.md §3): verdicts checked against the CPU oracle"
    out["detail"] = "gpurun_out/bench_detail.json (also the previous stdout line)"
    out = _r(out)
    # never let an extra take the line over the limit: drop optional objects, widest first
    for k in ("sweep", "sweep_columns", "small_n", "warm_path", "config5", "sustained_incl_h2d", "build", "parity", "sharded_sweep_columns",
              "sharded_sweep", "extended", "device_canary"):
        if len(json.dumps(out)) < LINE_LIMIT:
            break
        out.pop(k, None)
    return out


def emit(rec: dict) -> str:
    """detail record → gpurun_out/bench_detail.json + one stdout line; then the headline line, LAST on stdout"""
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, "w") as f:
            json.dump(rec, f)
    except OSError:
        pass
    line = json.dumps(headline_record(rec))
    assert len(line) < LINE_LIMIT, len(line)
    print(json.dumps({"bench_detail": rec}), flush=True)
    print(line, flush=True)
    return line


def relaunch(args) -> int:
    """`python bench.py --gpus N` from a bare shell: become N ranks under torch.distributed.run."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    global LIVE_TAG
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows (validators) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", action="store_true",
                    help="N = 1: the FULL series of rocprofv3 sub-steps (tools/profile.sh, tools/pmc_wave.sh; ≈2 minutes).  Without "
                         "it the run still collects roofline.traffic / valu_issue itself with a lean series (≈40 s)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="no rocprofv3 sub-steps at all: attach the committed profiles/ files (what the sub-steps themselves run with)")
    ap.add_argument("--no-sequence", action="store_true", help="skip the config-#3 sequence latency legs")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm-path leg")
    ap.add_argument("--seq-rounds", type=int, default=SEQ_ROUNDS)
    ap.add_argument("--no-sweep", action="store_true", help="skip the N = 64 … 65 536 sweep")
    ap.add_argument("--no-sustained", action="store_true", help="skip the fresh-host-batch-every-step leg (two staging slots)")
    ap.add_argument("--no-certificates", action="store_true", help="skip the round-change certificate leg")
    ap.add_argument("--no-host-mirror", action="store_true", help="skip the end-to-end legs through include/ibft_host.h")
    ap.add_argument("--extended-steps", type=int, default=400,
                    help="a second, longer sample of the headline leg (reported next to the K-step one)")
    ap.add_argument("--path", choices=["cold", "warm"], default="cold",
                    help="cold = ECDSA recover+compare for every row (headline); warm = keys already learned, "
                         "rows verified against per-validator tables (IBFT_FLAG_PUBKEY_CACHE)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))

    # WHICH HIP RUNTIME THE PROCESS RUNS: torch 2.10+rocm7.0 bundles its own libamdhip64 / libhsa-runtime64 / librccl (ROCm
    # 7.0.2), and a process that imports torch first binds libibftgpu.so — built against the image's ROCm 7.2 — to THOSE.  The
    # library is loaded FIRST instead, and torch comes in only where the contract needs torch.distributed (N > 1): as the carrier
    # of the 128-byte communicator id and of the timing fences over a gloo group — CPU tensors, torch.cuda never initialised,
    # no minute-long torch import at N = 1.  The data-path collective is the library's own ncclAllReduce either way.  (Speed is
    # NOT the reason: with the process held on one socket both runtimes run the kernel in the same time, profiles/r05g_*.)
    # torch_first = the previous order (torch's runtime, nccl process group, torch.cuda fences).
    torch_first = os.environ.get("IBFT_BENCH_TORCH_FIRST") == "1" and not DRY_RUN
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # … and WHERE THE PROCESS RUNS: on the GPU's own NUMA node the same kernel takes 0.333 ms, on the other socket 0.345 ms
    # (profiles/r05g_harness_ab.txt, r05h_numa_pin_ab.txt — the runtime makes no difference once the socket is held fixed; what
    # looked like one in r05b was the scheduler's choice of socket).  Every rank pins itself to its device's node before the
    # first HIP call (go_ibft_amd/numa.py: sysfs only); a Go host does the same with numactl.
    import go_ibft_amd.numa as NUMA
    numa_pin = {"pinned": False, "why": "dry run"} if DRY_RUN else NUMA.pin_to_device_node(local)
    want_dist = world > 1 or os.environ.get("IBFT_BENCH_FORCE_DIST") == "1"
    torch = None
    if torch_first:
        import torch
        os.environ.setdefault("IBFT_RCCL_LIB", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    if DRY_RUN:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_stub as V
    else:
        import go_ibft_amd.verifier as V
        V.load_library()                                  # binds the process to /opt/rocm's HIP runtime (unless torch_first)
        if want_dist and not torch_first:
            # … and to /opt/rocm's librccl, BEFORE torch comes in: libtorch_hip needs "librccl.so.1" too, and whichever copy is
            # mapped first serves both (same SONAME) — torch's RCCL (built for ROCm 7.0) on the 7.2 HIP runtime fails in
            # ncclCommInitRank ("unhandled cuda error", profiles/r05c: the first forced-dist run of this order)
            V.comm_preload()
    dist = None
    if want_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if torch_first:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")    # one node: the container's hostname may not resolve
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")    # RCCL's bootstrap likewise (data moves over xGMI / P2P)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    elif torch_first:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local) if torch_first else (torch.device("cpu") if torch is not None else None)

    import go_ibft_amd.shard as S

    def run_config(rows: int, byzantine: bool, steps: int, warmup: int, path: str, keep=None):
        """one timed leg: `steps` passes over this rank's resident shard (+ the exchange when sharded).
        keep: a dict that carries the context and the resident batch from one leg to the next (a node keeps its context for
        its lifetime: the K-step leg runs on the context the extended leg has been using, not on a minute-old one)"""
        n_total = rows * world
        lo, hi = rank * rows, (rank + 1) * rows
        if keep is not None and "bv" in keep:
            bv, rd, t_gen = keep["bv"], keep["rd"], 0.0
        else:
            bv = V.BatchVerifier(device=local, max_rows=max(rows, 1024),   # raises without the HIP lib / GPU
                                 flags=V.FLAG_PUBKEY_CACHE if path == "warm" else 0)
            t_gen = time.perf_counter()
            rd = load_round(bv, n_total, lo, hi, byzantine)
            t_gen = time.perf_counter() - t_gen
        addrs, power = rd["addrs"], rd["power"]
        bv.set_validators(1, addrs, power)
        bv.seals_stage(rd["hash32"], rd["seal65"], rd["signer20"], rd["pre"])  # H2D once: inputs resident in HBM
        if path == "warm":                                             # learn the keys, build the tables (untimed)
            bv.seals_launch(1); bv.seals_fetch(); bv.seals_launch(1); bv.seals_fetch()
            assert bv.cache_stats()[0] == len(np.unique(rd["signer20"], axis=0)) or byzantine
        assert V.shard_range(n_total, rank, world) == (lo, hi) == S.shard_range(n_total, rank, world)
        comm_info = None
        if dist:
            # the data-path collective lives in libibftgpu.so (RCCL all-reduce of verdict words + tally pieces);
            # torch.distributed only carries the 128-byte communicator id and the timing fences
            uid = [V.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0, device=dev)        # (gloo: dev = cpu)
            comm_init_or_die(bv, uid[0], rank, world)
            comm_info = bv.comm_info()   # what the communicator itself reports: (ranks, this rank, device)

        def step():  # one synchronous pass, results on the host when it returns (latency legs, warm-up)
            return bv.seals_run()

        def run_pipelined(k_steps, lat_out=None):
            """N = 1: k_steps passes with ONE kept in flight (ibft_seals_submit / ibft_seals_collect: two host-visible result
            slots, the collect waits for the oldest pass only) — every pass's verdict words and tally are delivered to the host
            and looked at; the launch of pass k+1 overlaps the completion and delivery of pass k, so the device runs back to
            back.  The same depth-1 pipeline as the sharded form below (exchange k overlaps the kernels of pass k+1): `value` has
            ONE definition at every N.  The synchronous round trip of a single pass is reported as step_latency_ms_p50."""
            bv.seals_submit()
            res, s0 = None, time.perf_counter()
            for _ in range(k_steps - 1):
                bv.seals_submit()
                res = bv.seals_collect()
                if lat_out is not None:
                    s1 = time.perf_counter()
                    lat_out.append(s1 - s0)
                    s0 = s1
            res = bv.seals_collect()
            if lat_out is not None:
                lat_out.append(time.perf_counter() - s0)
            return res

        def run_sharded(k_steps):
            """the exchange of pass k (own stream, behind the tally) overlaps with the kernels of pass k+1; the
            host consumes every merged result one pass later (bounded pipeline, depth 1)"""
            bv.seals_launch(1)
            res = None
            for k in range(k_steps):
                bv.seals_exchange(n_total)
                if k + 1 < k_steps:
                    bv.seals_launch(1)
                if k >= 1:
                    res = bv.seals_fetch_merged()
            return bv.seals_fetch_merged()

        def fence():
            if dist is not None:
                dist.barrier()
            if torch_first:
                torch.cuda.synchronize()      # (torch's streams carry none of this work: the library's own streams are what counts)
            bv.sync()                         # hipStreamSynchronize of the context's verdict and exchange streams

        # torch's import leaves ~10^6 tracked objects: a generation-2 collection in the middle of a timed loop costs ≈40 ms.
        # Collect BEFORE the warm-up (not between warm-up and timing: 40 ms of idle device is enough for its clocks to fall,
        # and the first tens of passes after a pause run ≈4 % slower — measured: the same kernel 0.460 ms in a 25-pass leg that
        # started behind the collection, 0.440 ms over the 400 passes of the leg next to it), keep the collector off while timing.
        gc.collect()
        gc.disable()
        if dist is None:
            # PREWARM untimed passes bring the device to the state a node under load is in; they are not part of the W warm-up
            # steps the caller asked for and are reported in the line (config.prewarm_steps).
            for _ in range(PREWARM_STEPS if (keep is not None or path == "warm") else 0):
                step()
            if warmup:
                run_pipelined(warmup)
        else:
            run_sharded(PREWARM_STEPS)                        # (the same untimed passes in front of a sharded leg)
            if warmup:
                run_sharded(warmup)
        fence()
        lat, kernel_ms, kernel_launches = [], 0.0, 0
        t0 = time.perf_counter()
        if dist is None:
            bv.set_kernel_timing(KERNEL_TIMING_EVERY)         # an event pair costs ≈5 µs of a step: sample the passes
            bv.last_kernel_ms()                               # reset: the HIP-event pairs of the timed passes accumulate
            intervals = []
            out = run_pipelined(steps, intervals)
        else:
            out = run_sharded(steps)
        fence()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if dist is None:
            kernel_ms, kernel_launches = bv.last_kernel_ms()  # summed over exactly the timed passes, read after them
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            kernel_ms, kernel_launches = bv.last_kernel_ms()  # the last pass's kernels (events are per launch)
            for k in range(min(steps, 20)):                   # latency of one synchronous pass incl. the exchange
                s0 = time.perf_counter()
                bv.seals_launch(1)
                bv.seals_exchange(n_total)
                bv.seals_fetch_merged()
                lat.append(time.perf_counter() - s0)
        # the synchronous round trip of ONE pass (launch → verdict kernel → tally → results on the host), outside the timed region
        if dist is None:
            for _ in range(min(max(steps, 20), 100)):
                s0 = time.perf_counter()
                step()
                lat.append(time.perf_counter() - s0)
        # quorum latency of this one call including the host→device copies
        lat_h2d = []
        if dist is None:
            for _ in range(min(steps, 200)):
                s0 = time.perf_counter()
                bv.is_valid_committed_seal(rd["hash32"], rd["seal65"], rd["signer20"], rd["pre"])
                lat_h2d.append(time.perf_counter() - s0)
        # correctness of what was timed (outside the timed region)
        expect = None
        if byzantine and DRY_RUN:
            expect = rd["expect"]                      # (the stub's decree: rows without a pre-flag are "valid")
        elif byzantine:
            from oracle import binding as B
            expect = B.verify_seals(B.ValSet(addrs, power), rd["hash32"], rd["seal65"], rd["signer20"], rd["pre"],
                                    nthreads=usable_cores()).astype(bool)
            assert (expect == rd["expect"]).all(), "the oracle disagrees with the generator's by-construction verdicts"
        if dist is None:
            verdict, tally = out
            if expect is None:
                assert verdict.all() and tally.has_quorum == 1 and tally.power == int(power.sum())
            else:
                assert (verdict == expect).all()
        else:
            verdict, tally = out                              # merged: global mask, summed tally
            if expect is None:
                assert verdict.all() and tally.valid_rows == n_total and tally.power == int(power.sum()) and tally.has_quorum
            else:
                assert (verdict[lo:hi] == expect).all(), "merged verdicts differ from the oracle on this shard"
                assert bool(tally.has_quorum) == (tally.power >= tally.quorum) and tally.valid_rows == int(verdict.sum())
        cold_lanes, warm_lanes = bv.last_dispatch()
        if path == "warm":
            bv.cache_stats()
            warm_lanes = bv.lanes_per_signature
        canary = None
        if not DRY_RUN and hasattr(bv, "issue_probe"):
            try:      # the device canary right behind the timed passes (clocks up): ns per aligned VALU instruction per SIMD
                canary = bv.issue_probe()
            except Exception as e:  # noqa: BLE001 — a diagnostic: its failure is recorded, nothing else
                canary = repr(e)
        res = {"n_total": n_total, "rows": rows, "elapsed": elapsed, "steps": steps, "lat": lat, "lat_h2d": lat_h2d, "canary": canary,
               "intervals": intervals if dist is None else [],
               "rccl": comm_info,
               "kernel_ms": kernel_ms, "kernel_launches": kernel_launches,
               "kname": kernel_name(path, cold_lanes, warm_lanes, bv.last_cold_table()),
               "src": rd["src"], "rd": rd, "tables": bv.cache_stats()[0] if path == "warm" else 0, "input_generation_s": t_gen,
               "valid_fraction": float(verdict.mean())}
        if dist:
            bv.comm_destroy()
        if keep is not None and "bv" not in keep:
            keep["bv"], keep["rd"] = bv, rd      # the next leg goes on with this context; its last leg closes it
        else:
            bv.close()
        return res

    # the driver's K may be small (20 steps = 5 kernel-time samples): a second, longer sample of the same leg, reported
    # NEXT TO the K-step one (value / ms_per_step stay the K-step numbers the contract asks for).  It runs FIRST: the first
    # launches after a process start run ≈4 % slower (the r03a line: kernel 0.459 ms in a 20-step leg right after start-up,
    # 0.440 ms over the following 400 steps), and a steady-state throughput is what the metric means.
    carry = {} if (world == 1 and args.extended_steps > 0) else None
    long_leg = run_config(args.rows, False, args.extended_steps, 10, args.path, keep=carry) if carry is not None else None
    main_leg = run_config(args.rows, False, args.steps, args.warmup, args.path, keep=carry)

    rec = None
    if rank == 0:
        m = main_leg
        rows, n_total = m["rows"], m["n_total"]
        value = n_total * m["steps"] / m["elapsed"]
        avg_kernel_s = (m["kernel_ms"] / 1e3) / max(m["kernel_launches"], 1)
        achieved = rows * ALGO_BYTES_PER_VERIFY / avg_kernel_s / 1e9  # GB/s, per launch on this rank
        live_info = None
        if world == 1 and dist is None and not DRY_RUN and not args.no_live_counters and args.path == "cold" and \
                os.environ.get("IBFT_BENCH_NO_LIVE") != "1":
            # (behind the timed legs: the sub-steps are processes of their own.  Round 6: the DEFAULT, so that the line the
            # driver parses carries counters collected on the driver's own box in the driver's own run)
            LIVE_TAG, live_info = collect_live_counters(rows, lean=not args.profile)
        traffic, valu = profile_attachments(m["kname"], rows, avg_kernel_s)
        live_traffic = live_valu = False
        if LIVE_TAG:   # which of the two really came from this run's files
            live_valu = bool(valu) and str(valu.get("source", "")).startswith("gpurun_out")
            try:
                lt = json.load(open(os.path.join(ROOT, "gpurun_out", "profiles", f"{LIVE_TAG}_traffic.json")))
                live_traffic = traffic is not None and any(v.get("hbm_bytes_per_launch") == traffic for v in lt.values())
            except (OSError, ValueError):
                pass
        workload = (f"BASELINE config #3: N={n_total} validators, 1xMI355X — value: one round of COMMIT seals per step "
                    f"(ECDSA recover+compare+membership+quorum tally); quorum_latency_ms_p50: the full PREPARE+COMMIT "
                    f"sequence with Keccak proposal-hash check") if world == 1 and rows == 4096 else \
                   (f"N={n_total} validators sharded x{world} ({rows} rows/GPU), COMMIT seals per step + RCCL all-reduce of "
                    f"the verdict-mask words and tally partials" + (" (BASELINE config #4)" if (world, rows) == (4, 4096) else ""))
        rec = {
            "metric": "committed_seal_verifies_per_sec", "value": value, "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / m["steps"] * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": f"synthetic ({m['src']})",
            "config": {"workload": workload, "validators": n_total, "rows_per_gpu": rows, "path": args.path,
                       "prewarm_steps": PREWARM_STEPS if (carry is not None or world > 1 or dist is not None) else 0,
                       "kernel": m["kname"], "parallelism": f"rows sharded x{world}" if world > 1 else "single GPU",
                       "pipeline": "one pass kept in flight, every pass's results delivered to the host (N = 1: ibft_seals_submit / "
                                   "_collect, the tally of pass k on a stream of its own next to the verdict kernel of pass k+1 (round 6; "
                                   "IBFT_SIDE_TALLY=0 turns that off); N > 1: exchange k overlaps the kernels of pass k+1); "
                                   "step_latency_ms_p50 = one synchronous pass",
                       "numa_pin": numa_pin},
            "step_latency_ms_p50": float(np.median(m["lat"]) * 1e3),
            "step_latency_ms_p50_incl_h2d": float(np.median(m["lat_h2d"]) * 1e3) if m["lat_h2d"] else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": m["kname"], "avg_kernel_ms": avg_kernel_s * 1e3, "kernel_samples": m["kernel_launches"],
                         "kernel_timing": f"HIP events on the library's stream around the verdict kernel of every "
                                          f"{KERNEL_TIMING_EVERY if world == 1 else 1}th timed pass ({m['kernel_launches']} samples)",
                         "algorithmic_bytes_per_launch": rows * ALGO_BYTES_PER_VERIFY,
                         "valu_issue": valu,
                         "counters": (f"sub-steps of this run (rocprofv3, {live_info['mode']} series, {live_info.get('seconds')} s): "
                                      f"gpurun_out/profiles/{LIVE_TAG}_*" +
                                      ("" if (live_traffic and live_valu) else
                                       f" — but {'traffic' if not live_traffic else ''}{' and ' if not (live_traffic or live_valu) else ''}"
                                       f"{'valu_issue' if not live_valu else ''} attached from the committed profiles/ (that pass failed)"))
                                     if LIVE_TAG else
                                     ("attached: the newest committed profiles/ file of this kernel and batch size (valu_issue.source "
                                      "names it)" + (f"; live collection: {json.dumps(live_info)}" if live_info else "")),
                         "counters_live": bool(LIVE_TAG and live_traffic and live_valu),
                         "note": "integer-VALU-bound path: HBM fraction is reported as required; valu_issue is "
                                 "the bound that applies (DESIGN.md §5)"},
        }
        rp = rocprof_avg_kernel_ms(m["kname"])
        if rp:   # the rocprofv3 --kernel-trace --stats average of the same kernel in this run's own sub-step
            rec["roofline"]["rocprof_avg_kernel_ms"], rec["roofline"]["rocprof_calls"] = rp
        rec["quorum_latency_ms_p50"] = rec["step_latency_ms_p50"]   # replaced by the sequence below at N=1
        if DRY_RUN:
            rec["dry_run"] = True
            rec["data"] = "DRY RUN (IBFT_BENCH_DRYRUN=1): CPU ranks, stub verifier — plumbing only, NOT a measurement"
        try:
            import go_ibft_amd.build as BLD
            bi = BLD.build_info()
            rec["build"] = {"libibftgpu": bi["flavour"], "intact": bi["intact"], "sha256_12": (bi["sha256"] or "")[:12]}
        except Exception:  # noqa: BLE001
            pass
        if m["rccl"] is not None:   # N > 1 (or forced): the collective is the library's ncclAllReduce over this communicator
            rec["rccl_nranks"], rec["rccl_rank0_device"] = m["rccl"][0], m["rccl"][2]
        if isinstance(m.get("canary"), tuple):
            ns = m["canary"][0]
            rec["device_canary"] = {"issue_ns": ns, "healthy_ns": CANARY_HEALTHY_NS, "off": ns / CANARY_HEALTHY_NS - 1.0,
                                    "flag": abs(ns / CANARY_HEALTHY_NS - 1.0) > 0.05, "probe_kernel_ms": m["canary"][1],
                                    "what": "ibft_issue_probe: wall ns per aligned 8-byte VALU instruction per SIMD, one wavefront per SIMD, "
                                            "median of 5 launches right behind the timed passes; flag = more than 5 % off — every "
                                            "throughput figure of this line is then off by about as much (DESIGN.md §5.8)"}
        elif m.get("canary") is not None:
            rec["device_canary"] = {"error": m["canary"]}
        if long_leg is not None:
            L = long_leg
            lk = (L["kernel_ms"] / 1e3) / max(L["kernel_launches"], 1)
            rec["extended"] = {"steps": L["steps"], "value": L["n_total"] * L["steps"] / L["elapsed"],
                               "ms_per_step": L["elapsed"] / L["steps"] * 1e3, "avg_kernel_ms": lk * 1e3,
                               "kernel_samples": L["kernel_launches"],
                               "hbm_frac": rows * ALGO_BYTES_PER_VERIFY / lk / 1e9 / HBM_PEAK_GBS,
                               "step_interval_ms_p10_p50_p90": ([float(x) for x in np.percentile(np.array(L["intervals"]) * 1e3, [10, 50, 90])]
                                                                if L["intervals"] else None),
                               "step_latency_ms_p10_p50_p90": [float(x) for x in np.percentile(np.array(L["lat"]) * 1e3, [10, 50, 90])]}

    if DRY_RUN:   # the extra legs all need the device
        args.no_warm = args.no_sequence = args.no_sweep = args.no_certificates = args.no_host_mirror = args.no_cpu_baseline = True
        args.no_sustained = True
    if world == 1 and args.path == "cold" and not args.no_warm:
        # extra, NOT the headline: the same batch once every validator's key is known (steady state)
        w = run_config(args.rows, False, args.steps, args.warmup, "warm")
        rec["warm_path"] = {"value": w["n_total"] * w["steps"] / w["elapsed"], "unit": "verifies/s",
                            "ms_per_step": w["elapsed"] / w["steps"] * 1e3,
                            "kernel_ms": w["kernel_ms"] / max(w["kernel_launches"], 1),
                            "step_latency_ms_p50": float(np.median(w["lat"]) * 1e3),
                            "tables_bytes": int(w["tables"]) * 32 * 256 * 80, "kernel": w["kname"], "prewarm_steps": PREWARM_STEPS,
                            "note": "keys learned by an earlier cold pass; identical verdicts (csrc/verify_dev.h)"}
    if world == 1 and not args.no_sequence and main_leg["rd"]["fx"] is not None:
        fx = main_leg["rd"]["fx"]
        cold = sequence_latency(V, fx, 0, args.seq_rounds)
        warm = sequence_latency(V, fx, V.FLAG_PUBKEY_CACHE, args.seq_rounds)
        sets_cold = sequence_latency(V, fx, 0, args.seq_rounds, "sets")
        sets_warm = sequence_latency(V, fx, V.FLAG_PUBKEY_CACHE, args.seq_rounds, "sets")
        sets_cold_pageable = sequence_latency(V, fx, 0, max(100, args.seq_rounds // 4), "sets", pinned=False)
        cold_pageable = sequence_latency(V, fx, 0, max(100, args.seq_rounds // 4), "calls", pinned=False)
        rec["quorum_latency_ms_p50"] = sets_cold["p50_ms"]
        rec["quorum_latency"] = {"definition": "p50 over rounds of the config-#3 sequence: host SoA columns -> verdict masks + "
                                               "quorum flag visible to the host, H2D + kernels + D2H included (SURVEY §8d). "
                                               "headline = message_sets_cold: no key known, one ibft_verify_messages call per "
                                               "message set (PREPARE, COMMIT), columns in ibft_pinned_alloc buffers; five_calls_* = the "
                                               "same verdicts through the five separate Verifier batches; *_pageable_columns = columns "
                                               "in ordinary host memory (five_calls_cold_pageable_columns is the form measured in round 1)",
                                 "message_sets_cold": sets_cold, "message_sets_warm": sets_warm,
                                 "five_calls_cold": cold, "five_calls_warm": warm,
                                 "message_sets_cold_pageable_columns": sets_cold_pageable,
                                 "five_calls_cold_pageable_columns": cold_pageable}
    if world == 1 and rank == 0 and not args.no_sweep:
        try:
            rec["sweep"] = sweep_sizes(V)
        except Exception as e:  # noqa: BLE001 — an extra leg must never take the headline line down
            rec["sweep"] = {"error": repr(e)}
    if world == 1 and rank == 0 and not args.no_sustained and args.path == "cold":
        try:
            rec["sustained_incl_h2d"] = sustained_leg(V, main_leg["rd"])
        except Exception as e:  # noqa: BLE001
            rec["sustained_incl_h2d"] = {"error": repr(e)}
    if world == 1 and rank == 0 and not args.no_sweep and not args.no_cpu_baseline:
        try:
            rec["small_n"] = small_n_leg(V)
        except Exception as e:  # noqa: BLE001
            rec["small_n"] = {"error": repr(e)}
    if world == 1 and rank == 0 and not args.no_sweep:
        try:
            rec["set_change"] = set_change_leg(V)
        except Exception as e:  # noqa: BLE001
            rec["set_change"] = {"error": repr(e)}
    if world == 1 and rank == 0 and not args.no_certificates:
        try:
            rec["certificates"] = certificates_leg(V)
        except Exception as e:  # noqa: BLE001
            rec["certificates"] = {"error": repr(e)}
    if world == 1 and rank == 0 and not args.no_host_mirror and main_leg["rd"]["fx"] is not None:
        # what a caller of include/ibft_host.h experiences: the same height as 8 191 wire messages through the mirror
        # (ingest → store → handlePrepare / handleCommit → seals), and the N = 256 round change (tools/host_e2e.py)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import host_e2e as E
            import go_ibft_amd.hostlib as HL
            fx = main_leg["rd"]["fx"]
            HL.retain_heap()        # ibft_host_retain_heap: what a node calls at start-up (freed C heap stays in the process)
            hm = {"definition": "BASELINE config #3 end to end through include/ibft_host.h: 4 095 PREPARE + 4 096 COMMIT wire "
                                "messages -> ibft_host_ingest_wire (micro-batches of 256) or the mirror's receive queue (adaptive "
                                "batches) -> handlePrepare -> handleCommit -> seals; a fresh mirror per repetition, p50; PREPARE / COMMIT "
                                "messages kept as rows (ibft_host_use_rows), ibft_host_retain_heap called",
                  "micro_batches_cold": E.host_mirror_from_wire(V, HL, fx, 0, 20),
                  "micro_batches_warm": E.host_mirror_from_wire(V, HL, fx, V.FLAG_PUBKEY_CACHE, 20),
                  "queue_cold": E.host_mirror_queue(V, HL, fx, 0, 20),
                  "queue_warm": E.host_mirror_queue(V, HL, fx, V.FLAG_PUBKEY_CACHE, 20),
                  "queue_cold_linger50us": E.host_mirror_queue(V, HL, fx, 0, 20, linger_us=50),
                  "queue_warm_linger50us": E.host_mirror_queue(V, HL, fx, V.FLAG_PUBKEY_CACHE, 20, linger_us=50),
                  "round_change_n256": E.round_change_through_the_mirror(V, HL, 256, 6),
                  "reproposal_n256": E.reproposal_through_the_mirror(V, HL, 256, 4)}
            rec.setdefault("quorum_latency", {})["host_mirror_from_wire"] = hm
        except Exception as e:  # noqa: BLE001
            rec.setdefault("quorum_latency", {})["host_mirror_from_wire"] = {"error": repr(e)}
    if dist is not None and (world > 1 or os.environ.get("IBFT_BENCH_CONFIG5") == "1" or os.environ.get("IBFT_BENCH_SHARDED_SWEEP") == "1") \
            and os.environ.get("IBFT_BENCH_SKIP_SHARDED_SWEEP") != "1" and os.environ.get("IBFT_BENCH_SKIP_CONFIG5") != "1":
        # north star: "sig-verifies/sec on synthetic rounds of N ∈ {64 … 65 536} validators reported at 1/2/4/8 GPUs".  Beyond
        # one GPU's share of the headline (4 096 rows per rank) the line carries a SHARDED SWEEP: N_total ∈ {16 384, 65 536}
        # split over the ranks in 64-aligned shards — BASELINE config #4 falls out of it at G = 4 (16 384 validators, all
        # valid) and config #5 at G = 8 (65 536 validators, 20 % Byzantine seals, every rank's shard of the merged mask held
        # against the CPU oracle outside the timed region).  With IBFT_BENCH_FORCE_DIST=1 on one GPU the same legs run as ONE
        # shard through the sharded code path and a real one-rank RCCL communicator.
        # These legs meet a multi-GPU node for the first time in the driver's run: should a rank stall (a collective one rank
        # never reaches), the headline line — complete by now — must still go out.  A timer on every rank ends the process;
        # rank 0 prints the line first, with the stall recorded.
        import threading
        sweep_out = []
        if rank == 0:
            rec["sharded_sweep"] = {"definition": "N_total validators split over the ranks in 64-aligned shards, one resident COMMIT "
                                                  "batch, recover + tally on every rank, ONE ncclAllReduce of verdict words + distinct-sender "
                                                  "bitmaps per step (exchange k overlaps the kernels of pass k+1); value = N_total x steps / "
                                                  "max-over-ranks time; 65 536: 20 % Byzantine seals, merged mask vs the CPU oracle",
                                    "sizes": sweep_out}

        def give_up():
            if rank == 0:
                sweep_out.append({"error": f"no result within {CONFIG5_TIMEOUT_S} s (leg abandoned, headline unaffected)"})
                sys.stdout.flush()
                emit(rec)
            os._exit(0)
        watchdog = threading.Timer(CONFIG5_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        for n_sw, byz in ((16384, False), (65536, True)):
            if n_sw % (64 * world) != 0:
                continue
            try:
                c5 = run_config(n_sw // world, byz, max(10, args.steps // 4), 3, "cold")
                if rank == 0:
                    ent = {"validators": c5["n_total"], "rows_per_gpu": n_sw // world, "byzantine_fraction": 0.2 if byz else 0.0,
                           "rccl_nranks": c5["rccl"][0] if c5["rccl"] else None,
                           "value": c5["n_total"] * c5["steps"] / c5["elapsed"], "unit": "verifies/s",
                           "ms_per_step": c5["elapsed"] / c5["steps"] * 1e3, "kernel": c5["kname"],
                           "step_latency_ms_p50": float(np.median(c5["lat"]) * 1e3) if c5["lat"] else None,
                           "valid_fraction": c5["valid_fraction"],
                           "parity": ("every rank's shard of the merged verdict mask equals the CPU oracle's verdicts; merged quorum "
                                      "flag recomputed from the merged power") if byz else
                                     "every merged verdict bit set, merged power = total power, quorum"}
                    if (world, n_sw) == (4, 16384):
                        ent["baseline_config"] = 4
                    if (world, n_sw) == (8, 65536):
                        ent["baseline_config"] = 5
                    sweep_out.append(ent)
                    if byz:
                        rec["config5"] = dict(ent)     # (the key earlier rounds' records carry)
            except Exception as e:  # noqa: BLE001 — an extra leg must never take the headline line down
                if rank == 0:
                    sweep_out.append({"validators": n_sw, "error": repr(e)})
                break                                  # (a failed collective: do not enter another one)
        watchdog.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rd = main_leg["rd"]
        try:
            rec["cpu_baseline"] = cpu_baseline(rd["addrs"], rd["power"], rd["hash32"], rd["seal65"], rd["signer20"])
        except Exception as e:  # noqa: BLE001 — a reported baseline: its failure is recorded, the measurement above stands
            rec["cpu_baseline"] = {"value": None, "unit": "verifies/s", "cores": 0, "kind": "port", "sample": "", "error": repr(e)}
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (block-buffered on a pipe): push it out first so that
        # the JSON line is the LAST thing on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        emit(rec)


if __name__ == "__main__":
    main()
