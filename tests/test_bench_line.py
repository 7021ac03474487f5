"""bench.py's output contract (round-4 review: the one JSON line had grown to 22.5 KB, the driver keeps the last 8 081 bytes of
stdout, BENCH_r04.json came back with parsed = null).  The LAST stdout line must be strict JSON under 6 KB with the quantities
SURVEY §8d names; everything else is the detail record, printed as an earlier line and written to gpurun_out/.  And the
`--gpus N` plumbing — relaunch under torch.distributed.run, process group, shard ranges, launch / exchange / fetch pipeline,
max-over-ranks timing, merged record — executes here on CPU ranks over gloo with a stub verifier (IBFT_BENCH_DRYRUN=1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "quorum_latency_ms_p50", "roofline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms", "algorithmic_bytes_per_launch")


def _strict(line: str) -> dict:
    def no_constants(x):
        raise ValueError(f"not strict JSON: {x}")
    return json.loads(line, parse_constant=no_constants)


def _check_headline(line: str, n_gpus: int):
    assert len(line) < 6144 and "\n" not in line
    rec = _strict(line)
    for k in REQUIRED:
        assert k in rec, k
    for k in ROOFLINE:
        assert k in rec["roofline"], k
    assert rec["n_gpus"] == n_gpus and rec["metric"] == "committed_seal_verifies_per_sec" and rec["unit"] == "verifies/s"
    assert "workload" in rec["config"] and rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-9
    return rec


def test_headline_line_of_a_full_detail_record_is_small_and_complete():
    import bench
    detail = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r04x.json")))   # round 4's 22.5 KB line
    assert len(json.dumps(detail)) > 20000
    line = json.dumps(bench.headline_record(detail))
    rec = _check_headline(line, 1)
    assert len(line) < 4096
    assert rec["value"] == pytest.approx(detail["value"], rel=1e-5) and rec["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-5)
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == detail["cpu_baseline"]["cores"]
    vi = rec["roofline"]["valu_issue"]
    assert {"wave_insts_per_launch", "achieved_ginst_s", "frac_of_guide_peak", "frac_of_full_occupancy_ceiling"} <= set(vi)
    assert [r[0] for r in rec["sweep"]] == [64, 256, 1024, 4096, 16384, 65536]
    assert "BASELINE config #3" in rec["config"]["workload"]


def test_headline_line_of_a_sharded_record_and_of_an_oversized_one():
    import bench
    detail = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r04x.json")))
    detail.update(n_gpus=8, rccl_nranks=8, rccl_rank0_device=0,
                  config5={"validators": 65536, "rows_per_gpu": 8192, "byzantine_fraction": 0.2, "rccl_nranks": 8, "value": 3.1e8,
                           "unit": "verifies/s", "ms_per_step": 0.21, "kernel": "ecrecover_rows_kernel<0>", "valid_fraction": 0.8,
                           "parity": "x" * 500})
    rec = _check_headline(json.dumps(bench.headline_record(detail)), 8)
    assert rec["rccl_nranks"] == 8 and rec["config5"]["validators"] == 65536 and "parity" not in rec["config5"]
    # extras can never push the line over the limit: optional objects go first, the required ones stay
    detail["sweep"]["sizes"] = detail["sweep"]["sizes"] * 40
    rec = _check_headline(json.dumps(bench.headline_record(detail)), 8)
    assert "sweep" not in rec and "cpu_baseline" in rec


def _run_bench(args, env_extra, timeout=600):
    env = dict(os.environ, IBFT_BENCH_DRYRUN="1", **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return lines


def test_dry_run_one_rank_prints_detail_then_headline():
    lines = _run_bench(["--steps", "6", "--warmup", "2", "--extended-steps", "8"], {})
    rec = _check_headline(lines[-1], 1)
    assert rec["dry_run"] is True and "DRY RUN" in rec["data"] and rec["steps"] == 6 and rec["warmup"] == 2
    detail = _strict(lines[-2])["bench_detail"]
    assert detail["extended"]["steps"] == 8 and detail["metric"] == rec["metric"]
    assert os.path.exists(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dry_run_gpus_n_relaunches_shards_exchanges_and_merges(world):
    """`python bench.py --gpus N` from a bare shell → N ranks under torch.distributed.run; every rank verifies its 64-aligned
    shard, the exchange buffer of go_ibft_amd/shard.py crosses a real (gloo) all-reduce, rank 0 prints the merged line"""
    env = {k: "" for k in ()}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        assert k not in os.environ
    lines = _run_bench(["--gpus", str(world), "--steps", "5", "--warmup", "1"], env)
    rec = _check_headline(lines[-1], world)
    assert rec["dry_run"] is True and rec["rccl_nranks"] == world
    assert rec["config"]["validators"] == 4096 * world and rec["config"]["rows_per_gpu"] == 4096
    assert rec["config"]["parallelism"] == f"rows sharded x{world}"
    if world == 4:
        assert "BASELINE config #4" in rec["config"]["workload"]
    # round 6: every N > 1 line carries the sharded sweep — N_total = 16 384 and 65 536 split over the ranks (config #4 at
    # G = 4 and config #5 at G = 8 fall out of it); one row per size in the headline, the full entries in the detail record
    sw = rec["sharded_sweep"]
    assert [(r[0], r[1]) for r in sw] == [(16384, 16384 // world), (65536, 65536 // world)] and "N_total" in rec["sharded_sweep_columns"]
    assert all(isinstance(r[2], float) and r[2] > 0 and r[3] > 0 and r[4] > 0 for r in sw)
    detail = _strict(lines[-2])["bench_detail"]["sharded_sweep"]["sizes"]
    assert [e["rccl_nranks"] for e in detail] == [world, world] and detail[0]["valid_fraction"] == 1.0
    assert 0.79 < detail[1]["valid_fraction"] < 0.81 and detail[1]["byzantine_fraction"] == 0.2
    assert detail[0].get("baseline_config") == (4 if world == 4 else None) and detail[1].get("baseline_config") == (5 if world == 8 else None)
    c5 = rec["config5"]                             # BASELINE config #5's shape (65 536 validators, 20 % bad) at this world size
    assert (c5["validators"], c5["rows_per_gpu"], c5["rccl_nranks"]) == (65536, 65536 // world, world) and 0.79 < c5["valid_fraction"] < 0.81
    # value = rows of ALL ranks per second of the slowest rank
    assert rec["value"] == pytest.approx(4096 * world * 5 / (rec["ms_per_step"] * 5e-3), rel=1e-3)


def test_cpu_baseline_leg_times_the_tuned_path_only_after_it_agreed(oracle, monkeypatch):
    """bench.py's CPU leg: the tuned recovery is timed when it agrees with the checker on a sample with corrupted rows; when it
    does not (here: a doctored tuned path), the plain path is timed instead and the record says why"""
    import importlib
    bench = importlib.import_module("bench")
    from oracle import workload as W
    rd = W.make_round(256, seed=77)
    out = bench.cpu_baseline(rd.addrs, rd.power, rd.hash32, rd.seal65, rd.signer20, budget_s=0.3)
    assert out["path"] == "tuned" and out["kind"] == "port" and out["value"] > 0 and out["plain_path"] > 0
    assert out["value"] > 1.5 * out["plain_path"] and "tuned_path_refused" not in out and "error" not in out
    rec = bench.headline_record({"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
                                 "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                                 "config": {"workload": "w"}, "roofline": {"bound": "hbm", "achieved": 1, "peak": 1, "unit": "GB/s", "frac": 1, "traffic": None},
                                 "cpu_baseline": out})
    assert rec["cpu_baseline"]["path"] == "tuned" and abs(rec["cpu_baseline"]["plain_path"] - out["plain_path"]) < 1.0
    real = oracle.verify_seals_tuned

    def doctored(vs, h, s, f, pre_flags=None, flags=0, nthreads=1):
        v = real(vs, h, s, f, pre_flags, flags, nthreads)
        v[0] ^= 1
        return v
    monkeypatch.setattr(oracle, "verify_seals_tuned", doctored)
    out = bench.cpu_baseline(rd.addrs, rd.power, rd.hash32, rd.seal65, rd.signer20, budget_s=0.3)
    assert out["path"] == "plain" and "differ" in out["tuned_path_refused"] and out["plain_path"] is None and "error" not in out

    def broken(*a, **k):
        raise OSError("no such symbol")
    monkeypatch.setattr(oracle, "verify_seals_tuned", broken)
    out = bench.cpu_baseline(rd.addrs, rd.power, rd.hash32, rd.seal65, rd.signer20, budget_s=0.3)
    assert out["path"] == "plain" and "no such symbol" in out["tuned_path_refused"] and out["value"] > 0
