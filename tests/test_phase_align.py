"""go-ibft_amd/phase_align.py — the build step that keeps 8-byte instructions of the device code on 8-byte boundaries
(profiles/r04o_*).  CPU only: the rules of the pass on a hand-made instruction list, and the built library's hot loops."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import go_ibft_amd.phase_align as PA  # noqa: E402


def _run(body, sizes):
    lines = ["\t.p2align\t2\n", "\t.type\tf,@function\n", "f:\n"] + [f"\t{b}\n" for b in body] + [".Lfunc_end0:\n"]
    stats = {"n8": 0, "misplaced": 0, "widened": 0, "nops": 0}
    seq = [(sz, b.split()[0]) for b, sz in zip(body, sizes)]
    return [x.strip() for x in PA.align_text(lines, {"f": seq}, stats)], stats


def test_rules_of_the_pass(monkeypatch):
    monkeypatch.setattr(PA, "MIN_RUN", 3)
    dpp = "v_mov_b32_dpp v1, v2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    # a lone 4-byte VALU instruction in front of 8-byte ones is widened; one that carries a literal is not
    out, st = _run(["v_add_u32_e32 v0, v1, v2", dpp, dpp, "v_and_b32_e32 v3, 0x3ffffff, v4", dpp], [4, 8, 8, 8, 8])
    assert out[3] == "v_add_u32_e64 v0, v1, v2" and "v_and_b32_e32 v3, 0x3ffffff, v4" in out and st["widened"] == 1 and st["misplaced"] == 0
    assert out[0].split() == [".p2align", "3"]                       # a function that is no kernel gets 8-byte alignment
    # an even run of 4-byte instructions restores the phase by itself
    out, st = _run(["v_add_u32_e32 v0, v1, v2", "s_add_i32 s0, s0, 1", dpp], [4, 4, 8])
    assert st == {"n8": 1, "misplaced": 0, "widened": 0, "nops": 0}
    # no wider form (SALU): an s_nop behind it when a long run of 8-byte instructions follows, nothing for a short one
    out, st = _run(["s_add_i32 s0, s0, 1", dpp, dpp, dpp], [4, 8, 8, 8])
    assert out[4].startswith("s_nop 0") and st["nops"] == 1 and st["misplaced"] == 0
    out, st = _run(["s_add_i32 s0, s0, 1", dpp, "s_nop 1", dpp], [4, 8, 4, 8])
    assert st["nops"] == 0 and st["misplaced"] == 1
    # the pc-relative triple stays in one piece: the pad goes in FRONT of s_getpc_b64
    body = ["s_getpc_b64 s[0:1]", "s_add_u32 s0, s0, g@rel32@lo+4", "s_addc_u32 s1, s1, g@rel32@hi+12", "s_swappc_b64 s[30:31], s[0:1]"]
    out, st = _run(body, [4, 8, 8, 4])
    i = out.index("s_getpc_b64 s[0:1]")
    assert out[i - 1].startswith("s_nop 0") and out[i + 1].startswith("s_add_u32") and out[i + 2].startswith("s_addc_u32")
    # operands a VOP3 encoding cannot carry keep the instruction as it is
    assert not PA._plain_operand("0x3ffffff") and not PA._plain_operand("sym@rel32@lo") and PA._plain_operand("v12") \
        and PA._plain_operand("64") and not PA._plain_operand("65") and PA._plain_operand("vcc") and PA._plain_operand("s[2:3]")


def test_hot_loops_of_the_built_library_are_aligned():
    import go_ibft_amd.build as B
    if not B.PHASE_ALIGN or not os.path.exists(B.LIB):
        pytest.skip("library not built with the alignment step")
    import code_phase as CP
    before = B._file_sha256(B.LIB)
    ins = list(CP.disassemble(B.LIB, "ecrecover_rows_kernelILi0").values())[0]
    # round-4 review: objcopy with one file operand rewrote the product library in place (while it was mapped: the suite ended
    # in SIGSEGV, and the rewritten binary travelled to the GPU box) — the tool reads a private copy now
    assert B._file_sha256(B.LIB) == before and B.build_info()["intact"]
    big = [(lo, hi) for lo, hi in set(CP.loops(ins)) if sum(1 for x in ins if lo <= x[0] <= hi) > 400]
    assert big
    for lo, hi in big:
        n, n8, mis = CP.stats([x for x in ins if lo <= x[0] <= hi])
        assert n8 > 0.7 * n and mis < 0.15 * n8, (hex(lo), n, n8, mis)


def test_a_rewritten_library_is_stale(tmp_path, monkeypatch):
    """the stamp records the built file's own sha256: a binary that something modified after the build is rebuilt, not shipped"""
    import go_ibft_amd.build as B
    t = tmp_path / "lib.so"
    dep = "keccak_dev.h"
    t.write_bytes(b"built")
    B._mark(str(t), [dep], "cmd", "phase_align")
    assert not B._stale(str(t), [dep], "cmd") and B.build_info(str(t)) == {"flavour": "phase_align", "sha256": B._file_sha256(str(t)), "intact": True}
    t.write_bytes(b"rewritten")
    assert B._stale(str(t), [dep], "cmd") and not B.build_info(str(t))["intact"]
    t.write_bytes(b"built")
    assert not B._stale(str(t), [dep], "cmd") and B._stale(str(t), [dep], "other cmd")
    (tmp_path / "lib.so.stamp").write_text("0123\n")            # a stamp of the old one-line format: rebuilt
    assert B._stale(str(t), [dep], "cmd")
