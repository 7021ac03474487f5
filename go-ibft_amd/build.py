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


def _stale(target: str, deps: list[str], extra: str = "") -> bool:
    """A target is current when the digest of its sources (+ the compile command) equals the one recorded
    next to it when it was built — modification times say nothing after a checkout or a copy to another box."""
    stamp = target + ".stamp"
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps, extra)


def _mark(target: str, deps: list[str], extra: str = "") -> None:
    with open(target + ".stamp", "w") as f:
        f.write(_digest(deps, extra) + "\n")


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *EXTRA_FLAGS, "-shared", "-fPIC",
           "-o", LIB, os.path.join(CSRC, "ibftgpu.hip"), "-ldl"]
    if force or _stale(LIB, SOURCES, " ".join(cmd[:-3])):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        _mark(LIB, SOURCES, " ".join(cmd[:-3]))
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
