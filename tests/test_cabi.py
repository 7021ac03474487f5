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
    assert lib.ibft_version() == 1
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
