#!/usr/bin/env python3
"""Static instruction mix of the hot loops of a verdict kernel (no GPU needed): compiles csrc/ibftgpu.hip to gfx950
assembly and counts, inside the loops that contain multiplications, the share of v_mad_u64_u32, of DPP-modified
instructions and of s_nop.  bench.py uses the shares to price the kernel against the measured per-instruction issue
times of one wavefront per SIMD (profiles/r02a_ubench_wave.txt).
Usage: python tools/static_mix.py [kernel-name-substring ...] > profiles/rNN_static_mix.txt  (default: ecrecover_rows_kernelILi0)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wants = sys.argv[1:] or ["ecrecover_rows_kernelILi0"]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    sys.path.insert(0, ROOT)
    from go_ibft_amd.build import EXTRA_FLAGS      # the product's own code-generation flags
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *EXTRA_FLAGS, "--cuda-device-only", "-S", "-w", "-o", out,
                           os.path.join(ROOT, "go-ibft_amd", "csrc", "ibftgpu.hip")], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")


def one(want):
    start = next(i for i, l in enumerate(lines) if re.match(r"^[^\s.;]\S*" + re.escape(want) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    L = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(L) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(L):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))

    def body(a, b):
        return [l.split()[0] + (" dpp" if ("row_" in l or "quad_perm" in l) else "") for l in L[a:b + 1] if re.match(r"\s+[vs]_|\s+ds_|\s+global_", l)]
    print(f"# kernel {L[0].split(':')[0]}")
    tot = {"valu": 0, "mad": 0, "dpp": 0, "nop": 0, "salu": 0, "all": 0}
    outer = [lp for lp in loops if not any(o[0] < lp[0] and lp[1] < o[1] for o in loops)]
    for a, b in sorted(outer):
        bd = body(a, b)
        mads = sum(1 for x in bd if x.startswith("v_mad_u64_u32"))
        if not mads:
            continue
        valu = sum(1 for x in bd if x.startswith("v_"))
        dpp = sum(1 for x in bd if x.endswith(" dpp"))
        nop = sum(1 for x in bd if x.startswith("s_nop"))
        salu = sum(1 for x in bd if x.startswith("s_")) - nop
        print(f"# loop at lines {a}-{b}: {len(bd)} instructions, VALU {valu}, v_mad_u64_u32 {mads}, DPP {dpp}, s_nop {nop}, other SALU {salu}")
        for k, v in (("valu", valu), ("mad", mads), ("dpp", dpp), ("nop", nop), ("salu", salu), ("all", len(bd))):
            tot[k] += v
    if not tot["valu"]:          # (a kernel whose multiplications all sit in called functions: count the whole body)
        bd = body(0, len(L) - 1)
        tot = {"valu": sum(1 for x in bd if x.startswith("v_")), "mad": sum(1 for x in bd if x.startswith("v_mad_u64_u32")),
               "dpp": sum(1 for x in bd if x.endswith(" dpp")), "nop": sum(1 for x in bd if x.startswith("s_nop")), "salu": 0, "all": len(bd)}
        print(f"# no loop with multiplications: whole body, {len(bd)} instructions")
    print(f"# v_mad_u64_u32 share of VALU (static, hot loops): {tot['mad'] / tot['valu']:.4f}")
    print(f"# DPP share of VALU (static, hot loops): {tot['dpp'] / tot['valu']:.4f}")
    print(f"# s_nop per VALU instruction (static, hot loops): {tot['nop'] / tot['valu']:.4f}")


for w in wants:
    try:
        one(w)
    except StopIteration:
        print(f"# kernel {w}: not found")
