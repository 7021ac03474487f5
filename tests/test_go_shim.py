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
