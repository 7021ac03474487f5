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


def test_checker_knows_what_the_go_compiler_refuses_outright(tmp_path):
    """an import nothing uses and a local nothing reads are compile errors in Go"""
    dst = tmp_path / "go"
    shutil.copytree(os.path.join(ROOT, "shim", "go"), dst)
    f = dst / "core" / "hoststore_glue.go"
    src = f.read_text()
    assert "q, prepared := hs.HandlePrepare(view)" in src
    f.write_text(src.replace("q, prepared := hs.HandlePrepare(view)", "q, prepared := hs.HandlePrepare(view)\n\tleftover := len(prepared)")
                    .replace('import (', 'import (\n\t"sort"', 1))
    out = subprocess.run([sys.executable, TOOL, str(dst)], capture_output=True, text=True)
    assert out.returncode == 1
    assert '"sort" imported and not used' in out.stdout and "leftover declared and not used" in out.stdout


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


CGO_TOOL = os.path.join(ROOT, "tools", "cgo_typecheck.py")


def test_cgo_boundary_typechecks_under_a_c_compiler():
    """every argument of every C.f(...) call, every composite-literal field, every field read from a C struct and every
    //export signature of shim/go, as `_Static_assert(__builtin_types_compatible_p(…))` compiled by gcc against include/*.h"""
    out = subprocess.run([sys.executable, CGO_TOOL], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    last = out.stdout.strip().splitlines()[-1]
    n_checked, n_left = (int(x) for x in __import__("re").findall(r"(\d+) assertions compiled, (\d+) arguments", last)[0])
    assert n_checked >= 300 and n_left == 0, last           # nothing the tool could not type is hiding behind the pass


def test_cgo_typecheck_catches_seeded_boundary_defects(tmp_path):
    dst = tmp_path / "go"
    shutil.copytree(os.path.join(ROOT, "shim", "go"), dst)
    g = dst / "ibftgpu" / "ibftgpu.go"
    src = g.read_text()
    seeds = [
        # a pointer of the wrong width (uint32 offsets handed over as uint64)
        ("rc := C.ibft_verify_senders(c.h, ptr8(payload), (*C.uint32_t)(unsafe.Pointer(&off[0]))",
         "rc := C.ibft_verify_senders(c.h, ptr8(payload), (*C.uint64_t)(unsafe.Pointer(&off[0]))", "C.ibft_verify_senders argument 3"),
        # an integer of the wrong C type: Go has no implicit widening
        ("C.ibft_set_seal_digest(c.h, mode, ptr8(suffix), C.size_t(len(suffix)))",
         "C.ibft_set_seal_digest(c.h, mode, ptr8(suffix), C.uint32_t(len(suffix)))", "C.ibft_set_seal_digest argument 4"),
        # a field that the C struct does not have
        ("max_rows: C.uint32_t(o.MaxRows)}", "max_row: C.uint32_t(o.MaxRows)}", "max_row"),
        # nil where the prototype takes an integer
        ("err := c.check(C.ibft_seals_swap(c.h, 1))", "err := c.check(C.ibft_seals_swap(c.h, nil))", "nil passed"),
    ]
    for old, new, _ in seeds:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    g.write_text(src)
    cb = dst / "hoststore" / "callbacks.go"
    cs = cb.read_text()
    assert "func hoststoreSignal(user unsafe.Pointer, msgType C.uint32_t, height, round C.uint64_t)" in cs
    cb.write_text(cs.replace("func hoststoreSignal(user unsafe.Pointer, msgType C.uint32_t, height, round C.uint64_t)",
                             "func hoststoreSignal(user unsafe.Pointer, msgType C.uint64_t, height, round C.uint64_t)"))
    hs = dst / "hoststore" / "hoststore.go"
    hsrc = hs.read_text()
    assert "C.ibft_host_use_batch(h, 1)" in hsrc
    hs.write_text(hsrc.replace("C.ibft_host_use_batch(h, 1)", "C.ibft_host_use_batch(1, h)"))        # arguments swapped
    out = subprocess.run([sys.executable, CGO_TOOL, str(dst)], capture_output=True, text=True)
    assert out.returncode == 1, out.stdout
    for _, _, needle in seeds:
        assert needle in out.stdout, (needle, out.stdout)
    assert "//export hoststoreSignal parameter 2" in out.stdout
    assert "C.ibft_host_use_batch argument 1" in out.stdout and "C.ibft_host_use_batch argument 2" in out.stdout


ARITY_TOOL = os.path.join(ROOT, "tools", "go_arity_check.py")


def test_go_calls_fit_their_definitions():
    """every call of shim/go whose receiver the tool can type (parameters, receivers, type assertions, struct fields, first
    results; interfaces and methods of the overlay and of the reference) passes as many arguments as the definition takes and
    assigns as many names as it returns"""
    out = subprocess.run([sys.executable, ARITY_TOOL], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    judged = int(__import__("re").search(r"(\d+) calls judged", out.stdout).group(1))
    assert judged >= (200 if os.path.isdir("/root/reference") else 60), out.stdout


def test_go_arity_check_catches_seeded_defects(tmp_path):
    dst = tmp_path / "go"
    shutil.copytree(os.path.join(ROOT, "shim", "go"), dst)
    glue = dst / "core" / "hoststore_glue.go"
    gs = glue.read_text()
    assert "q, prepared := hs.HandlePrepare(view)" in gs
    glue.write_text(gs.replace("q, prepared := hs.HandlePrepare(view)", "q := hs.HandlePrepare(view)"))       # two results, one name
    wi = dst / "core" / "wire_ingest.go"
    ws = wi.read_text()
    assert "mask, rows, ok := wv.VerifySendersWire(wire, off)" in ws and "case !ibftgpu.Bit(mask, k):" in ws
    wi.write_text(ws.replace("mask, rows, ok := wv.VerifySendersWire(wire, off)", "mask, rows, ok := wv.VerifySendersWire(wire)")
                    .replace("case !ibftgpu.Bit(mask, k):", "case !ibftgpu.Bit(mask):")
                    .replace("i.addWireStock(stock, nil)", "i.addWireStockRenamed(stock, nil)", 1))
    gs2 = glue.read_text()
    assert "func (i *IBFT) AddWireMessagesQueued(raw [][]byte) bool {" in gs2 and "return hs.AddWireMessages(raw) == nil" in gs2
    glue.write_text(gs2.replace("func (i *IBFT) AddWireMessagesQueued(raw [][]byte) bool {", "func (i *IBFT) AddWireMessagesQueued(raw []byte) bool {"))
    hs = dst / "hoststore" / "hoststore.go"
    hsrc = hs.read_text()
    old_sig = "func (s *Store) HandleCommit(view *proto.View) (quorum bool, seals []*messages.CommittedSeal) {"
    assert old_sig in hsrc and "func (s *Store) PruneByHeight(height uint64) {" in hsrc
    hs.write_text(hsrc.replace(old_sig, "func (s *Store) HandleCommit(view *proto.View) (quorum bool, seals [][]byte) {")
                      .replace("func (s *Store) PruneByHeight(height uint64) {", "func (s *Store) PruneByHeight(height uint32) {"))
    out = subprocess.run([sys.executable, ARITY_TOOL, str(dst)], capture_output=True, text=True)
    assert out.returncode == 1, out.stdout
    assert "*hoststore.Store.HandleCommit has the signature" in out.stdout and "core.hostStore wants" in out.stdout
    if os.path.isdir("/root/reference"):
        assert "*hoststore.Store.PruneByHeight has the signature (uint32) ()" in out.stdout       # core.Messages is the reference's
    assert "hs.AddWireMessages argument 1: raw is []byte, the parameter is [][]byte" in out.stdout
    assert "1 names receive the 2 results of hs.HandlePrepare" in out.stdout
    assert "wv.VerifySendersWire called with 1 arguments, the definition takes 2" in out.stdout
    assert "ibftgpu.Bit called with 1 arguments, the definition takes 2" in out.stdout
    if os.path.isdir("/root/reference"):
        assert "i.addWireStockRenamed is not defined on core.IBFT" in out.stdout
