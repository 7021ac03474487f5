"""The host mirror under sanitizers (SURVEY §5 asks for ASan/UBSan; the reference's CI runs `go test -race`,
.github/workflows/main.yml:33-48).  5 000 lines of C++ with borrowed views into batch buffers, a helper decode thread, a
receive-queue worker and per-type store locks: libibft_host.so is rebuilt with -fsanitize=address,undefined (and
-fsanitize=thread) into a temporary directory and the mirror's own test files run against THAT library in a child
interpreter with the sanitizer runtime preloaded.  Skipped where the toolchain has no sanitizer runtimes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    try:
        p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    except OSError:
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run(lib, preload, extra_env, files, select=None):
    env = dict(os.environ, IBFT_HOST_LIB=lib, LD_PRELOAD=preload, PYTHONPATH=ROOT, **extra_env)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider", *files]
    if select:
        cmd += ["-k", select]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


def test_mirror_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    asan = _runtime("libasan.so")
    if not asan:
        pytest.skip("no libasan in this toolchain")
    import go_ibft_amd.hostlib as H
    lib = H.build_host_sanitized("asan", str(tmp_path))
    files = ["tests/test_host_rows.py", "tests/test_host_roundchange.py", "tests/test_host_cert_ingest.py",
             "tests/test_host_cluster.py", "tests/test_host_device_quorum.py", "tests/test_hoststore_sequence.py",
             "tests/test_host_certificates.py", "tests/test_host_semantics.py"]
    out = _run(lib, asan, {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=97:allocator_may_return_null=1",
                           "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=98"}, files)
    text = out.stdout + out.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert out.returncode == 0, text[-4000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_receive_queue_under_thread_sanitizer(tmp_path):
    """concurrent ibft_host_queue_push (several transport threads) + the worker's ingests + handle_* / store_* calls from
    the signal callback and from the main thread"""
    tsan = _runtime("libtsan.so")
    if not tsan:
        pytest.skip("no libtsan in this toolchain")
    import go_ibft_amd.hostlib as H
    lib = H.build_host_sanitized("tsan", str(tmp_path))
    files = ["tests/test_host_roundchange.py", "tests/test_hoststore_sequence.py", "tests/test_host_threads.py"]
    out = _run(lib, tsan, {"TSAN_OPTIONS": "halt_on_error=0:exitcode=96:report_signal_unsafe=0"}, files,
               select="queue or sequence or threads")
    text = out.stdout + out.stderr
    if "unexpected memory mapping" in text or "FATAL: ThreadSanitizer" in text:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container: " + text[-300:])
    assert "WARNING: ThreadSanitizer" not in text, text[-6000:]
    assert out.returncode == 0, text[-4000:]
