"""8-byte phase alignment of the device code (a build step of libibftgpu.so; go-ibft_amd/build.py drives it).

Measured on MI355X (profiles/r04n_code_placement_ab.txt, tools/code_phase.py): with ONE resident wavefront per SIMD — the
shape every verdict kernel runs in at the sizes that matter — an 8-byte instruction (DPP, VOP3: 90 % of the row layout's hot
loops) that starts at an address 4 mod 8 costs ≈1 ns more than one that starts at 0 mod 8: the same 165.0 M instructions per
launch took 0.3563 ms with 39 % of the main loop's 8-byte instructions on such addresses and 0.3664 ms with 57 % — and which
of the two a build gets was decided by whether an even or an odd number of 4-byte instructions happened to precede the loop.

This pass takes the compiler's assembly (hipcc --cuda-device-only -S) and, function by function, keeps the 8-byte
instructions on 8-byte boundaries:
  * a 4-byte VALU instruction met at phase 0 is re-encoded in its 8-byte VOP3 form (`_e32` → `_e64`: same operation, same
    operands, no extra instruction), so that it does not flip the phase; met at phase 4 it stays as it is and restores phase 0;
  * a 4-byte instruction that has no wider form (SALU, s_nop, s_waitcnt, branches, VALU with a literal or carry semantics)
    flips the phase; when at least MIN_RUN 8-byte instructions follow before the next 4-byte one, an `s_nop 0` is put behind
    it (one issue slot ≈ two misplaced instructions);
  * functions that are not kernels get 8-byte alignment (kernels are 256-byte aligned already).
Instruction sizes come from assembling the input once and reading the disassembly; the output is assembled by the caller.
Adding wait states and widening an encoding cannot break a hazard rule or change a result; the GPU suite runs on the output.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MIN_RUN = int(os.environ.get("IBFT_PHASE_MIN_RUN", "6"))  # an s_nop costs an issue slot (≈2.1 ns), a misplaced 8-byte instruction ≈0.4 ns
# VOP1 / VOP2 / VOPC mnemonics whose `_e64` form takes the same operand list (no implicit carry, no literal allowed → checked)
WIDENABLE = re.compile(
    r"^v_(mov_b32|and_b32|or_b32|xor_b32|not_b32|add_u32|sub_u32|subrev_u32|lshlrev_b32|lshrrev_b32|ashrrev_i32|"
    r"mul_u32_u24|mul_hi_u32_u24|mul_i32_i24|max_u32|min_u32|max_i32|min_i32|cndmask_b32|ffbl_b32|ffbh_u32|bfrev_b32|"
    r"cmp_(eq|ne|lt|le|gt|ge)_(u32|i32)|cmp_(eq|ne|lt|le|gt|ge)_(u64|i64))_e32$")
INLINE_CONST = re.compile(r"^(-?\d+|0x[0-9a-fA-F]+|-?\d+\.\d+)$")


def _is_literal(tok: str) -> bool:
    """an operand that is a number outside the inline-constant range needs a literal dword: VOP3 has none on gfx9"""
    tok = tok.strip()
    if not INLINE_CONST.match(tok):
        return False
    if "." in tok:
        return tok not in ("0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0")
    v = int(tok, 0)
    return not (-16 <= v <= 64)


_REG = re.compile(r"^(v\d+|s\d+|v\[\d+:\d+\]|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0)$")


def _plain_operand(tok: str) -> bool:
    """a register or an inline constant: what a VOP3 encoding can carry"""
    tok = tok.strip()
    return bool(_REG.match(tok)) or (bool(INLINE_CONST.match(tok)) and not _is_literal(tok))


def assemble(src: str, obj: str) -> None:
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj])


def instruction_sizes(obj: str) -> dict[str, list[tuple[int, str]]]:
    """symbol → [(size, mnemonic)] in layout order"""
    txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", obj], text=True)
    out: dict[str, list[tuple[int, str]]] = {}
    cur = None
    for l in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.search(r"//\s*([0-9A-F]+):\s*([0-9A-F]{8})(\s+[0-9A-F]{8})?", l)
        if m and cur is not None:
            cur.append((8 if m.group(3) else 4, l.split()[0]))
    return out


_LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")


HOT_ONLY = os.environ.get("IBFT_PHASE_HOT_ONLY", "1") != "0"
_BRANCH = re.compile(r"^s_(?:branch|cbranch_\w+)\s+([A-Za-z_.$][\w.$]*)")


def hot_lines(lines: list[str]) -> list[bool]:
    """Round 6: which lines are worth aligning.  Widening and padding make the code ≈ 5 % longer; inside a loop (or inside an
    outlined function, which the loops call) that buys issue slots, in code a launch runs through ONCE it only adds cache lines
    to fetch — and the lane / group cold kernels are measured against the 64 KB instruction cache.  A line is
    hot when it lies between a local label and a branch back to it, or in a function that is not a kernel."""
    kernels = {m.group(1) for l in lines for m in [re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", l)] if m}
    hot = [False] * len(lines)
    labels: dict[str, int] = {}
    in_kernel = None
    for i, line in enumerate(lines):
        s = line.strip()
        m = _LABEL.match(s)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                in_kernel = name in kernels
                labels = {}
            else:
                labels[name] = i
            continue
        if in_kernel is None:
            continue
        if not in_kernel:
            hot[i] = True          # an outlined function: called from the loops
            continue
        b = _BRANCH.match(s)
        if b and b.group(1) in labels:   # a branch to a label seen earlier in this function: a loop
            for j in range(labels[b.group(1)], i + 1):
                hot[j] = True
    return hot


def align_text(lines: list[str], sizes: dict[str, list[tuple[int, str]]], stats: dict, skip: set[str] = frozenset()) -> list[str]:
    out: list[str] = []
    cur, k, phase = None, 0, 0
    seq: list[tuple[int, str]] = []
    hot = hot_lines(lines) if HOT_ONLY else [True] * len(lines)
    for idx, line in enumerate(lines):
        s = line.strip()
        m = _LABEL.match(s)
        if m and m.group(1) in sizes and not m.group(1).startswith(".L") and m.group(1) not in skip:
            cur, k, phase, seq = m.group(1), 0, 0, sizes[m.group(1)]
            # a function that is not a kernel: .p2align 2 in front of it → 8 bytes (look back a few lines)
            for j in range(len(out) - 1, max(len(out) - 6, -1), -1):
                if re.match(r"^\s*\.p2align\s+2\b", out[j]):
                    out[j] = re.sub(r"(\.p2align\s+)2\b", r"\g<1>3", out[j], count=1)
                    break
            out.append(line)
            continue
        if cur is None or not s or s[0] in ".;#" or _LABEL.match(s) or k >= len(seq):
            if cur is not None and s.startswith(".Lfunc_end"):
                cur = None
            out.append(line)
            continue
        size, mnem = seq[k]
        tok = s.split()[0]
        if tok != mnem and not (tok.startswith(mnem) or mnem.startswith(tok)):
            raise RuntimeError(f"{cur}: instruction {k}: assembly line says {tok!r}, disassembly {mnem!r}")
        k += 1
        if size == 8:
            stats["n8"] += 1
            if phase == 4:
                stats["misplaced"] += 1
            out.append(line)
            continue
        # 4-byte instruction
        if phase == 4:
            phase = 0
            out.append(line)
            continue
        if not hot[idx]:           # code that runs once per launch: left as the compiler wrote it, only the phase is tracked
            phase = 4
            out.append(line)
            stats["cold_left_alone"] = stats.get("cold_left_alone", 0) + 1
            continue
        # an even run of 4-byte instructions restores the phase by itself: nothing to do for its members
        n4, j = 0, k - 1
        while j < len(seq) and seq[j][0] == 4:
            n4 += 1
            j += 1
        if n4 % 2 == 0:
            out.append(line)
            phase = 4
            continue
        ops = s[len(tok):].split(";")[0]
        if WIDENABLE.match(tok) and all(_plain_operand(o) for o in ops.split(",")):
            out.append(line.replace(tok, tok[:-4] + "_e64", 1))
            stats["widened"] += 1
            continue
        # the phase flips: how many 8-byte instructions follow before the next 4-byte one?
        run, j = 0, k
        while j < len(seq) and seq[j][0] == 8:
            run += 1
            j += 1
        pad = "\ts_nop 0                                  ; (phase)\n"
        if tok == "s_getpc_b64":
            # s_getpc_b64 / s_add_u32 sym@rel32@lo+4 / s_addc_u32 sym@rel32@hi+12 is ONE unit: the offsets count bytes from the
            # end of the first instruction — nothing may come between them; the pad goes in front
            out.append(pad)
            out.append(line)
            stats["nops"] += 1
        elif run >= MIN_RUN:  # (behind an unconditional branch the s_nop is never executed: alignment for free)
            out.append(line)
            out.append(pad)
            stats["nops"] += 1
        else:
            out.append(line)
            phase = 4
    return out


def align_file(src: str, dst: str, obj: str | None = None, verbose: bool = False) -> dict:
    """src → dst (aligned assembly); when `obj` is given the result is assembled into it.  A function whose aligned form does
    not assemble (a branch that no longer reaches: the widened encodings make the 128 KB group kernels a little longer) is
    left as the compiler wrote it."""
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "in.o")
        assemble(src, tmp)
        sizes = instruction_sizes(tmp)
    with open(src) as f:
        lines = f.readlines()
    skip: set[str] = set()
    for attempt in range(8):
        stats = {"n8": 0, "misplaced": 0, "widened": 0, "nops": 0}
        out = align_text(lines, sizes, stats, skip)
        with open(dst, "w") as f:
            f.writelines(out)
        if obj is None:
            break
        r = subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dst, "-o", obj],
                           capture_output=True, text=True)
        if r.returncode == 0:
            break
        bad = set()
        for m in re.finditer(r":(\d+):\d+: error:", r.stderr):
            ln = int(m.group(1)) - 1
            while ln >= 0:
                lm = _LABEL.match(out[ln].strip())
                if lm and lm.group(1) in sizes and not lm.group(1).startswith(".L"):
                    bad.add(lm.group(1))
                    break
                ln -= 1
        if not bad or bad <= skip:
            raise RuntimeError("phase_align: the aligned assembly does not assemble:\n" + r.stderr[:2000])
        skip |= bad
    stats["skipped_functions"] = sorted(skip)
    if verbose:
        print(f"phase_align: {stats['n8']} 8-byte instructions, {stats['widened']} 4-byte VALU instructions widened, "
              f"{stats['nops']} s_nop inserted, {stats['misplaced']} still at 4 mod 8; left alone: {sorted(skip)}", file=sys.stderr)
    return stats


if __name__ == "__main__":
    align_file(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, verbose=True)
