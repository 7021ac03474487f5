"""Build libibftgpu.so for gfx950 with hipcc (in-tree, so it travels with gpurun)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libibftgpu.so")
SOURCES = ["ibftgpu.hip", "kernels.hip.h", "recover_dev.h", "verify_dev.h", "wave_fe_dev.h", "wire_dev.h", "cert_wave_dev.h", "modinv_dev.h",
           "sign_dev.h", "secp256k1_dev.h", "keccak_dev.h", os.path.join("..", "..", "include", "ibftgpu.h")]
# Code-generation flags of the product library (part of its build stamp).  max-ilp: the verdict kernels run ONE wavefront per
# SIMD at the sizes that matter (N ≤ 4 096), where every hazard s_nop is a lost issue slot — scheduling for instruction-level
# parallelism instead of register pressure takes the s_nops of ecrecover_rows_kernel from 399 to 114 (133 → 174 VGPRs, still
# two wavefronts per SIMD) and the kernels 3–4.6 % down at N = 64 … 4 096, nothing lost at 16 384 / 65 536
# (profiles/r03r_*).  IBFT_HIPCC_FLAGS adds to them for experiments, IBFT_NO_CODEGEN_FLAGS=1 drops them (the A/B of
# profiles/r03r_sched_ab.txt).
CODEGEN_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
EXTRA_FLAGS = ([] if os.environ.get("IBFT_NO_CODEGEN_FLAGS") == "1" else CODEGEN_FLAGS) + os.environ.get("IBFT_HIPCC_FLAGS", "").split()
HOST_HARNESS = os.path.join(CSRC, "libdev_arith_host.so")
WAVE_HARNESS = os.path.join(CSRC, "libdev_wave_host.so")
CERT_WAVE_HARNESS = os.path.join(CSRC, "libdev_cert_wave_host.so")


def _digest(deps: list[str], extra: str = "") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for d in deps:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _file_sha256(path: str) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _read_stamp(target: str) -> list[str]:
    try:
        with open(target + ".stamp") as f:
            return f.read().split("\n")
    except OSError:
        return []


def _stale(target: str, deps: list[str], extra: str = "") -> bool:
    """A target is current when (1) the digest of its sources (+ the compile command) equals the one recorded next to it
    when it was built — modification times say nothing after a checkout or a copy to another box — and (2) the target's OWN
    sha256 equals the one recorded: a binary that something rewrote after the build (round-4 review: objcopy, in place) is
    rebuilt, never shipped."""
    st = _read_stamp(target)
    if not os.path.exists(target) or len(st) < 2:
        return True
    return st[0].strip() != _digest(deps, extra) or st[1].strip() != _file_sha256(target)


def _mark(target: str, deps: list[str], extra: str = "", flavour: str = "") -> None:
    """stamp = source digest, the built file's sha256, how it was built (line 3: what bench.py / build_info() report)"""
    with open(target + ".stamp", "w") as f:
        f.write(_digest(deps, extra) + "\n" + _file_sha256(target) + "\n" + flavour + "\n")


def build_info(target: str | None = None) -> dict:
    """what the stamp says about a built target: {"flavour": "phase_align" | "phase_align_failed: …" | "one_step", "sha256": …,
    "intact": the file still is what the build wrote}"""
    target = target or LIB
    st = _read_stamp(target)
    if len(st) < 3 or not os.path.exists(target):
        return {"flavour": None, "sha256": None, "intact": False}
    return {"flavour": st[2].strip(), "sha256": st[1].strip(), "intact": st[1].strip() == _file_sha256(target)}


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
PHASE_ALIGN = os.environ.get("IBFT_NO_PHASE_ALIGN") != "1"


def _build_lib_phase_aligned(verbose: bool) -> None:
    """hipcc's own steps, taken apart so that the device assembly passes through go-ibft_amd/phase_align.py (8-byte
    instructions on 8-byte boundaries: profiles/r04o_*): device code → assembly → aligned assembly → code object →
    fat binary → host compile with that fat binary embedded → libibftgpu.so."""
    import tempfile
    try:
        from . import phase_align
    except ImportError:  # (run as a script: python go-ibft_amd/build.py)
        import importlib.util
        spec = importlib.util.spec_from_file_location("phase_align", os.path.join(HERE, "phase_align.py"))
        phase_align = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(phase_align)
    src = os.path.join(CSRC, "ibftgpu.hip")
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", *EXTRA_FLAGS]
    with tempfile.TemporaryDirectory(prefix="ibftgpu_build_") as d:
        dev_s, al_s, dev_o, dev_out, fb = (os.path.join(d, n) for n in ("dev.s", "dev.aligned.s", "dev.o", "dev.out", "dev.hipfb"))
        def run(cmd):   # stderr is kept: a failing sub-step must say why
            p = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
            if verbose and p.stderr:
                print(p.stderr, end="", flush=True)
            if p.returncode != 0:
                raise RuntimeError(f"{os.path.basename(cmd[0])} returned {p.returncode}: {p.stderr[-2000:]}")
        run(["hipcc", *common, "--cuda-device-only", "-S", "-o", dev_s, src])
        phase_align.align_file(dev_s, al_s, dev_o, verbose=verbose)
        run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o])
        run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={dev_out}", f"-output={fb}"])
        run(["hipcc", *common, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-shared", "-fPIC",
             "-o", LIB, src, "-ldl"])


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *EXTRA_FLAGS, "-shared", "-fPIC",
           "-o", LIB, os.path.join(CSRC, "ibftgpu.hip"), "-ldl"]
    stamp_extra = " ".join(cmd[:-3]) + (" +phase_align" if PHASE_ALIGN else "")
    deps = SOURCES + ([os.path.join("..", "phase_align.py")] if PHASE_ALIGN else [])
    # a stamp that records a FAILED alignment step is retried (the fallback library is the same code without the step)
    retry = PHASE_ALIGN and (build_info(LIB)["flavour"] or "").startswith("phase_align_failed")
    if force or retry or _stale(LIB, deps, stamp_extra):
        done = False
        flavour = "one_step"
        if PHASE_ALIGN:
            try:
                if verbose:
                    print(" ".join(cmd), "  (+ phase_align.py between device assembly and code object)", flush=True)
                _build_lib_phase_aligned(verbose)
                done = True
                flavour = "phase_align"
            except Exception as e:  # the one-shot build below is the same library without the alignment step
                print(f"build: phase-aligned build failed ({e}); building with hipcc in one step", flush=True)
                flavour = "phase_align_failed: " + " ".join(str(e).split())[:300]
        if not done:
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
        _mark(LIB, deps, stamp_extra, flavour)
    return LIB


DEVTEST = os.path.join(CSRC, "libibft_devtest.so")


def build_devtest(force: bool = False) -> str:
    """TEST-ONLY: single arithmetic primitives as gfx950 kernels (tests/test_gpu_arith.py)."""
    deps = ["devtest.hip", "recover_dev.h", "modinv_dev.h", "secp256k1_dev.h", "keccak_dev.h", "wave_fe_dev.h", "verify_dev.h"]
    if force or _stale(DEVTEST, deps):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-w",
                               "-o", DEVTEST, os.path.join(CSRC, "devtest.hip")], cwd=CSRC)
        _mark(DEVTEST, deps)
    return DEVTEST


def build_host_harness(force: bool = False) -> str:
    """TEST-ONLY: the device arithmetic headers compiled for the CPU (hipcc host pass)."""
    deps = ["host_arith_harness.hip", "sign_dev.h", "recover_dev.h", "modinv_dev.h", "secp256k1_dev.h", "keccak_dev.h", "wire_dev.h", "verify_dev.h"]
    if force or _stale(HOST_HARNESS, deps):
        subprocess.check_call(["hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC",
                               "-o", HOST_HARNESS, os.path.join(CSRC, "host_arith_harness.hip")], cwd=CSRC)
        _mark(HOST_HARNESS, deps)
    return HOST_HARNESS


def build_wave_harness(force: bool = False) -> str:
    """TEST-ONLY: wave_fe_dev.h (one wavefront per signature) on the CPU through wave_emul.h."""
    deps = ["host_wave_harness.hip", "wave_fe_dev.h", "wave_emul.h", "recover_dev.h", "verify_dev.h",
            "modinv_dev.h", "secp256k1_dev.h", "keccak_dev.h"]
    if force or _stale(WAVE_HARNESS, deps):
        subprocess.check_call(["hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC",
                               "-o", WAVE_HARNESS, os.path.join(CSRC, "host_wave_harness.hip")], cwd=CSRC)
        _mark(WAVE_HARNESS, deps)
    return WAVE_HARNESS


def build_cert_wave_harness(force: bool = False) -> str:
    """TEST-ONLY: cert_wave_dev.h (Keccak by one wavefront, the certificate walk) on the CPU through wave_emul.h."""
    deps = ["host_cert_wave_harness.hip", "cert_wave_dev.h", "wave_emul.h", "wire_dev.h", "keccak_dev.h"]
    if force or _stale(CERT_WAVE_HARNESS, deps):
        subprocess.check_call(["hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC",
                               "-o", CERT_WAVE_HARNESS, os.path.join(CSRC, "host_cert_wave_harness.hip")], cwd=CSRC)
        _mark(CERT_WAVE_HARNESS, deps)
    return CERT_WAVE_HARNESS


if __name__ == "__main__":
    print(build_lib(verbose=True))
