"""shim/go cannot be compiled here (no Go toolchain): tools/check_go_shim.py at least proves that every method,
package symbol, C entry point and import path the Go files use is defined where a compiler would look —
VERDICT r1 found an undefined method and two missing exports by hand."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_go_shim.py")


def test_shim_is_self_consistent():
    out = subprocess.run([sys.executable, TOOL], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


def test_checker_catches_the_round1_defects(tmp_path):
    dst = tmp_path / "go"
    shutil.copytree(os.path.join(ROOT, "shim", "go"), dst)
    f = dst / "core" / "wire_ingest.go"
    src = f.read_text()
    f.write_text(src.replace("func (i *IBFT) addVerifiedMessage(", "func (i *IBFT) addVerifiedMessageRenamed(")
                    .replace('"github.com/0xPolygon/go-ibft/ibftgpu"', '"github.com/0xPolygon/go-ibft/shim/ibftgpu"'))
    g = dst / "ibftgpu" / "ibftgpu.go"
    g.write_text(g.read_text().replace("func (c *Ctx) SealsRun(", "func (c *Ctx) sealsRunHidden(")
                 .replace("C.ibft_seals_run(", "C.ibft_seals_runn("))
    out = subprocess.run([sys.executable, TOOL, str(dst)], capture_output=True, text=True)
    assert out.returncode == 1
    assert "import path" in out.stdout and "ibft_seals_runn" in out.stdout
    if os.path.isdir("/root/reference"):          # methods of core.IBFT are looked up in the reference too
        assert "addVerifiedMessage" in out.stdout


def test_checker_counts_arguments_and_knows_the_host_header(tmp_path):
    """round 4: C.ibft_host_* resolve against include/ibft_host.h, and a C call with the wrong number of arguments — what a
    changed prototype leaves behind in a file no compiler sees — is reported"""
    dst = tmp_path / "go"
    shutil.copytree(os.path.join(ROOT, "shim", "go"), dst)
    f = dst / "hoststore" / "hoststore.go"
    src = f.read_text()
    assert "C.ibft_host_use_batch(h, 1)" in src and "C.ibft_host_rows_kept(s.h)" in src
    f.write_text(src.replace("C.ibft_host_use_batch(h, 1)", "C.ibft_host_use_batch(h)")
                    .replace("C.ibft_host_rows_kept(s.h)", "C.ibft_host_rows_keptt(s.h)"))
    g = dst / "ibftgpu" / "ibftgpu.go"
    gs = g.read_text()
    assert "C.uint64_t(round), nil, ptr8(proposer20)," in gs
    g.write_text(gs.replace("C.uint64_t(round), nil, ptr8(proposer20),", "C.uint64_t(round), nil,", 1))   # the pre-round-4 call
    out = subprocess.run([sys.executable, TOOL, str(dst)], capture_output=True, text=True)
    assert out.returncode == 1
    assert "ibft_host_use_batch called with 1 arguments, the prototype has 2" in out.stdout
    assert "ibft_host_rows_keptt is not declared" in out.stdout
    assert "ibft_verify_messages called with 18 arguments, the prototype has 19" in out.stdout
    assert "documented sequence" in out.stdout                      # … and the method's documented C sequence no longer matches
