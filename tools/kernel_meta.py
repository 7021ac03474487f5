#!/usr/bin/env python
"""tools/kernel_meta.py [LIB.so] [SUBSTR…] — code-object metadata of the verdict kernels of a built libibftgpu.so: registers,
private segment (scratch), LDS and code size per kernel (llvm-readelf on the gfx950 bundle; works on a private copy of the
library).  The sizes DESIGN.md §4 / §5.8 quote come from here."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_meta(lib):
    d = tempfile.mkdtemp()
    try:
        shutil.copyfile(lib, f"{d}/lib.so")
        subprocess.check_call(["objcopy", "--dump-section", f".hip_fatbin={d}/fat.bin", f"{d}/lib.so", "/dev/null"], stderr=subprocess.DEVNULL)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/k.co"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{d}/k.co"], text=True)
        syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", f"{d}/k.co"], text=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    size = {}
    for l in syms.splitlines():
        f = l.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m:
            continue
        def g(key):
            mm = re.search(r"\." + key + r":\s+(\d+)", blk)
            return int(mm.group(1)) if mm else 0
        out[m.group(1)] = {"agpr": int(re.match(r"\s*(\d+)", blk).group(1)), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                           "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                           "code_bytes": size.get(m.group(1), 0)}
    return out, size


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "go-ibft_amd", "csrc", "libibftgpu.so")
    subs = sys.argv[2:] or ["ecrecover", "verify_known"]
    meta, size = kernel_meta(lib)
    rows = []
    for n, v in meta.items():
        dem = subprocess.check_output(["c++filt", n], text=True).strip().split("(")[0].replace("void ibftk::", "")
        if any(s in dem for s in subs):
            rows.append((dem, v))
    for dem, v in sorted(rows):
        print(f"{dem:42s} vgpr {v['vgpr']:4d} agpr {v['agpr']:3d} sgpr {v['sgpr']:3d} scratch {v['scratch']:5d} B  lds {v['lds']:6d} B  code {v['code_bytes'] / 1024:6.1f} KB")
    fn = sorted(((s, n) for n, s in size.items() if n not in meta and s >= 1024), reverse=True)
    print("outlined functions ≥ 1 KB:", ", ".join(f"{subprocess.check_output(['c++filt', n], text=True).strip().split('(')[0][-40:]} {s / 1024:.1f} KB" for s, n in fn[:12]))
