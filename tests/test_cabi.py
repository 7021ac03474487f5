"""The C-ABI library loads and exports every symbol include/ibftgpu.h declares (no GPU
compute here); and without a device it fails loudly instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import go_ibft_amd.build as build
    import go_ibft_amd.verifier as V
    build.build_lib()
    return V.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ibftgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ibft_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    import go_ibft_amd.verifier as V
    assert sorted(V.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_strerror(lib):
    assert lib.ibft_version() == 4   # 3: the staged / pipelined pass calls, ibft_seals_rows, ibft_issue_probe; 4: ibft_pipeline_stats
    assert lib.ibft_strerror(0) == b"ok"
    assert b"device" in lib.ibft_strerror(-2)


def test_no_cpu_fallback_without_device(lib):
    import torch
    import go_ibft_amd.verifier as V
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(V.GpuUnavailable):
        V.BatchVerifier()


def test_product_does_not_reference_oracle():
    """Nothing under go-ibft_amd/ may import, include or link the oracle."""
    pkg = os.path.join(ROOT, "go-ibft_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                for line in text.splitlines():
                    s = line.strip()
                    if s.startswith(("#include", "import ", "from ")):
                        assert "oracle" not in s, (f, s)


def test_null_context_is_rejected_everywhere(lib):
    """every entry point that takes a context answers IBFT_E_INVAL (-1) for NULL instead of touching the
    device — checkable without a GPU"""
    import ctypes as C
    import go_ibft_amd.verifier as V
    t = V.Tally()
    m = (C.c_uint64 * 4)()
    b = (C.c_uint8 * 256)()
    off = (C.c_uint32 * 2)(0, 10)
    null = C.c_void_p()
    calls = [
        lambda: lib.ibft_set_validators(null, 1, b, b, 1),
        lambda: lib.ibft_verify_hashes(null, b, 8, 0, b, b, 1, m),
        lambda: lib.ibft_proposal_hash(null, b, 8, 0, b),
        lambda: lib.ibft_verify_seals(null, b, b, b, None, 1, m, C.byref(t)),
        lambda: lib.ibft_verify_senders(null, b, off, b, b, None, 1, m, C.byref(t)),
        lambda: lib.ibft_verify_senders_wire(null, b, off, 1, m, None, C.byref(t)),
        lambda: lib.ibft_wire_stage_seals(null),
        lambda: lib.ibft_tally(null, b, m, 1, C.byref(t)),
        lambda: lib.ibft_seals_stage(null, b, b, b, None, 1),
        lambda: lib.ibft_sign_seals(null, b, b, 1, b, None, None),
        lambda: lib.ibft_column_stats(null, None),
        lambda: lib.ibft_forget_proposal(null),
        lambda: lib.ibft_verify_messages_wire(null, b, off, 1, 0, 0, b, 1, 0, None, m, m, None, None, None, C.byref(t)),
        lambda: lib.ibft_verify_messages(null, b, off, b, b, b, b, None, None, None, 1, b, 1, 0, None, None, m, m, C.byref(t)),
        lambda: lib.ibft_tally_prepare(null, b, m, 1, b, C.byref(t)),
        lambda: lib.ibft_comm_info(null, None, None, None),
        lambda: lib.ibft_set_seal_digest(null, 1, b, 1),
        lambda: lib.ibft_group_set_seal_digest(null, 1, b, 1),
        lambda: lib.ibft_verify_certificates_wire(null, b, off, 1, 8, C.byref(C.c_size_t()), None, None, None, m, m, m),
        lambda: lib.ibft_seals_launch(null, 1),
        lambda: lib.ibft_seals_fetch(null, m, C.byref(t)),
        lambda: lib.ibft_seals_export(null, None, None),
        lambda: lib.ibft_seals_export_on(null, None, None, None),
        lambda: lib.ibft_seals_exchange(null, 64),
        lambda: lib.ibft_seals_fetch_merged(null, m, C.byref(t)),
        lambda: lib.ibft_group_verify_seals(null, b, b, b, None, 1, m, C.byref(t)),
        lambda: lib.ibft_group_verify_senders(null, b, off, b, b, None, 1, m, C.byref(t)),
        lambda: lib.ibft_group_verify_messages(null, b, off, b, b, b, b, None, None, None, 1, b, 1, 0, None, None, m, m, C.byref(t)),
        lambda: lib.ibft_group_set_validators(null, 1, b, b, 1),
        lambda: lib.ibft_group_is_local(null),
    ]
    for i, call in enumerate(calls):
        assert call() == -1, i


def test_tally_struct_matches_the_header():
    """ibft_tally_t: 56 bytes, field order as in include/ibftgpu.h (the ctypes struct and the Go struct mirror it)"""
    import ctypes as C
    import go_ibft_amd.verifier as V
    hdr = open(os.path.join(ROOT, "include", "ibftgpu.h")).read()
    body = hdr[hdr.index("typedef struct {\n  uint64_t quorum_lo, quorum_hi;"):hdr.index("} ibft_tally_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(quorum_lo|quorum_hi|power_lo|power_hi|valid_rows|distinct_senders|has_quorum|shard_overlap|proposer_rows|reserved)\b(?=[,;])", body)
    assert names == [f[0] for f in V.Tally._fields_]
    assert C.sizeof(V.Tally) == 56 and V.Tally.proposer_rows.offset == 48
    go = open(os.path.join(ROOT, "shim", "go", "ibftgpu", "ibftgpu.go")).read()
    assert "uint32(t.proposer_rows)" in go and "ProposerRows" in go


def test_cert_node_struct_matches_the_header():
    """ibft_cert_node_t: 56 bytes, field order as in include/ibftgpu.h"""
    import go_ibft_amd.verifier as V
    hdr = open(os.path.join(ROOT, "include", "ibftgpu.h")).read()
    body = hdr[hdr.index("typedef struct {\n  uint32_t off, len;"):hdr.index("} ibft_cert_node_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(off|len|parent|ordinal|first_child|n_children|raw_off|raw_len|proposal_round|cut0|cut1|level|role|flags|pad)\b(?=[,;\[])", body)
    assert names == list(V.CERT_NODE.names)
    assert V.CERT_NODE.itemsize == 56 and V.CERT_NODE.fields["proposal_round"][1] == 32 and V.CERT_NODE.fields["level"][1] == 48


def test_wire_row_struct_matches_the_header():
    """ibft_wire_row_t: 80 bytes, field order as in include/ibftgpu.h (the numpy dtype the wrapper uses and
    the Go struct in shim/go/ibftgpu/ibftgpu.go mirror it)"""
    import go_ibft_amd.verifier as V
    hdr = open(os.path.join(ROOT, "include", "ibftgpu.h")).read()
    body = hdr[hdr.index("typedef struct {\n  uint64_t height, round;"):hdr.index("} ibft_wire_row_t;")]
    names = re.findall(r"\b(height|round|status|type|payload_kind|has_view|hash_len|seal_len|from_len|sig_len|from|proposal_hash|pad)\b(?=[,;\[])", body)
    assert names == list(V.WIRE_ROW.names)
    assert V.WIRE_ROW.itemsize == 80 and V.WIRE_ROW.fields["from"][1] == 24 and V.WIRE_ROW.fields["proposal_hash"][1] == 44


def test_host_keccak_entry_needs_no_device(lib):
    """ibft_keccak256(a ‖ b) — the library's HOST Keccak (what hashes proposals and messages too long for one wavefront's sponge;
    compiled for BMI1/2 where the core has them, plain otherwise): block boundaries, two-part inputs, against the oracle."""
    import ctypes as C
    import numpy as np
    from oracle import binding as B
    f = lib.ibft_keccak256
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(5)
    for n in list(range(0, 280)) + [407, 408, 409, 1000, 4096, 100_000]:
        a = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        b = rng.integers(0, 256, n % 11, dtype=np.uint8).tobytes()
        out = (C.c_uint8 * 32)()
        assert f(a, len(a), b, len(b), out) == 0
        assert bytes(out) == B.keccak256(a + b), n
