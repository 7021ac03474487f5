#!/usr/bin/env python
"""tools/code_phase.py LIB.so [KERNEL_SUBSTR] — how many 8-byte instructions of a kernel's loops start at an address that is
4 mod 8.  Measured on MI355X (profiles/r04n_code_placement_ab.txt): the SAME instructions run ≈3 % slower when the main loop of
ecrecover_rows_kernel has 57 % of its 8-byte instructions on such addresses than when it has 39 % — one resident wavefront
per SIMD pays for every instruction that straddles two aligned 8-byte fetch units.  Prints, per loop (a backward branch), the
counts; the weighted total uses the trip counts of the row recover (33 digits, 4 doublings each, 267-step chain, 16 windows)."""
import re, subprocess, sys, os, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib, sym_substr):
    d = tempfile.mkdtemp()
    # objcopy with ONE file operand rewrites that file in place (round-4 review: the CPU suite replaced the product library
    # while it was mapped): work on a private copy and send objcopy's output to /dev/null
    import shutil
    shutil.copyfile(lib, f"{d}/lib.so")
    subprocess.check_call(["objcopy", "--dump-section", f".hip_fatbin={d}/fat.bin", f"{d}/lib.so", "/dev/null"], stderr=subprocess.DEVNULL)
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/k.co"])
    syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", f"{d}/k.co"], text=True)
    names = sorted({l.split()[-1] for l in syms.splitlines() if " FUNC " in l and sym_substr in l})
    out = {}
    for nme in names:
        txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={nme}", f"{d}/k.co"], text=True)
        ins = []
        for l in txt.splitlines():
            m = re.search(r"//\s*([0-9A-F]+):\s*([0-9A-F]{8})(\s+[0-9A-F]{8})?", l)
            if m:
                t = l.split()
                ins.append((int(m.group(1), 16), 8 if m.group(3) else 4, t[0], t[1] if len(t) > 1 else ""))
        out[nme] = ins
    return out


def loops(ins):
    res = []
    for a, sz, op, arg in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                off = int(arg)
            except ValueError:
                continue
            if off >= 32768:
                res.append((a + 4 + (off - 65536) * 4, a))
    return res


def stats(body):
    n8 = [x for x in body if x[1] == 8]
    return len(body), len(n8), sum(1 for x in n8 if x[0] % 8 == 4)


if __name__ == "__main__":
    lib = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "ecrecover_rows_kernelILi0"
    for nme, ins in disassemble(lib, sub).items():
        print(nme, f"{len(ins)} instructions, start {ins[0][0]:x}")
        tot = stats(ins)
        print(f"  whole function: {tot[1]} 8-byte instructions, {tot[2]} at 4 mod 8 ({100 * tot[2] / max(1, tot[1]):.0f} %)")
        seen = set()
        for lo, hi in loops(ins):
            if (lo, hi) in seen:
                continue
            seen.add((lo, hi))
            n, n8, mis = stats([x for x in ins if lo <= x[0] <= hi])
            if n >= 100:
                print(f"  loop {lo:x}..{hi:x}: {n:5d} instructions, {n8:5d} of 8 bytes, {mis:5d} at 4 mod 8 ({100 * mis / max(1, n8):.0f} %)")
