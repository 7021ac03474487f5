"""Op-by-op check of the device arithmetic ON THE GPU (gfx950 code object) against Python
big integers: field/scalar multiply, Fermat and safegcd inversions, sqrt chain, GLV split."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref as R

pytestmark = pytest.mark.gpu
P, N = R.P, R.N
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-ibft_amd", "csrc")


@pytest.fixture(scope="module")
def dt():
    import go_ibft_amd.build as build
    L = C.CDLL(build.build_devtest())
    L.devtest_run.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    return L


def run(dt, op, xs, ys=None, extra=0):
    n = len(xs)
    ys = ys or [0] * n
    a = b"".join(x.to_bytes(32, "big") for x in xs)
    b = b"".join(y.to_bytes(32, "big") for y in ys)
    out = C.create_string_buffer(32 * n + extra)
    assert dt.devtest_run(op, n, a, b, out, 32 * n + extra) == 0
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "big") for i in range(n)], out.raw


def values(seed, count, mod):
    rng = np.random.default_rng(seed)
    edge = [0, 1, 2, 3, mod - 1, mod - 2, 2**255 % mod, 2**128, 2**30, 2**30 - 1, (mod + 1) // 2, mod // 3]
    return edge + [int.from_bytes(rng.bytes(32), "big") % mod for _ in range(count)] + \
        [(int.from_bytes(rng.bytes(32), "big") >> s) % mod for s in range(1, 256, 7)]


def test_field_ops_on_gpu(dt):
    xs, ys = values(1, 400, P), values(2, 400, P)
    got, _ = run(dt, 0, xs, ys)
    assert got == [x * y % P for x, y in zip(xs, ys)]
    got, _ = run(dt, 1, xs)
    assert got == [x * x % P for x in xs]
    got, _ = run(dt, 7, xs)
    assert got == [pow(x, (P + 1) // 4, P) for x in xs]


def test_inversions_on_gpu(dt):
    xs = values(3, 600, P)
    exp = [pow(x, -1, P) if x else 0 for x in xs]
    assert run(dt, 2, xs)[0] == exp          # Fermat chain
    assert run(dt, 9, xs)[0] == exp          # safegcd, raw
    assert run(dt, 3, xs)[0] == exp          # safegcd through the fe wrapper
    xs = values(4, 600, N)
    exp = [pow(x, -1, N) if x else 0 for x in xs]
    assert run(dt, 5, xs)[0] == exp
    assert run(dt, 10, xs)[0] == exp
    assert run(dt, 6, xs)[0] == exp


def test_scalar_mul_and_glv_on_gpu(dt):
    xs, ys = values(5, 400, N), values(6, 400, N)
    assert run(dt, 4, xs, ys)[0] == [x * y % N for x, y in zip(xs, ys)]
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    n = len(xs)
    k1s, raw = run(dt, 8, xs, extra=33 * n)
    for i, k in enumerate(xs):
        k2 = int.from_bytes(raw[32 * (n + i):32 * (n + i) + 32], "big")
        f = raw[64 * n + i]
        assert k1s[i] < 2**128 and k2 < 2**128
        assert ((-k1s[i] if f & 1 else k1s[i]) + (-k2 if f & 2 else k2) * lam) % N == k


def test_gtab_and_full_recover_on_gpu(dt, oracle):
    """The fixed-base table built on the device equals e·2^(8w)·G, and recover_address (device
    function, outside the product kernel's LDS staging) matches the oracle."""
    rng = np.random.default_rng(9)
    n = 96
    dig, rr, ss, vv, exp = b"", b"", b"", b"", []
    for i in range(n):
        sk = (int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1).to_bytes(32, "big")
        d = rng.bytes(32)
        sig = oracle.sign(sk, d)
        if i % 7 == 3:
            sig = rng.bytes(64) + bytes([i & 1])
        dig += d; rr += sig[:32]; ss += sig[32:64]; vv += sig[64:]
        exp.append(oracle.recover_address(d, sig))
    out = C.create_string_buffer(32 * n)
    nw, ne, nb = C.c_int(), C.c_int(), C.c_int()
    dt.devtest_gtab_dims(C.byref(nw), C.byref(ne), C.byref(nb))
    nw, ne, nb = nw.value, ne.value, nb.value
    gt = np.zeros(nw * ne * 20, dtype=np.uint32)
    dt.devtest_recover.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p]
    assert dt.devtest_recover(n, dig, rr, ss, vv, out, gt.ctypes.data_as(C.c_void_p)) == 0
    g = gt.reshape(nw, ne, 20)
    for w, e in ((0, 1), (0, 2), (0, 255), (1, 1), (5, 77), (nw - 1, ne - 1), (nw - 1, 1), (nw // 2, ne // 2 + 1)):
        pt = R.pt_mul(e << (nb * w), R.G)
        x = sum(int(g[w, e, i]) << (26 * i) for i in range(10))
        y = sum(int(g[w, e, 10 + i]) << (26 * i) for i in range(10))
        assert (x, y) == pt, (w, e)
    for i in range(n):
        row = out.raw[32 * i:32 * i + 32]
        assert bool(row[20]) == (exp[i] is not None), i
        if exp[i] is not None:
            assert row[:20] == exp[i], i


def test_exceptional_point_additions_mixed_lanes_on_gpu(dt):
    """One wavefront holds lanes that hit P+P, P+(−P), ∞+P, P+∞ and ordinary additions at the
    same time: the exceptional paths are wave-uniform branches with per-lane selects."""
    rng = np.random.default_rng(22)
    k1s, k2s = [], []
    for i in range(128):
        k = int.from_bytes(rng.bytes(32), "big") % N
        k1s.append(k)
        k2s.append([k, N - k, 0, (k * 7 + 3) % N, (k * 7 + 3) % N, 1, (k * 5) % N, k][i % 8])
        if i % 16 == 9:
            k1s[-1] = 0
    exp = []
    for a, b in zip(k1s, k2s):
        pt = R.pt_mul((a + b) % N, R.G)
        exp.append(pt[0] if pt is not None else 0)
    assert run(dt, 15, k1s, k2s)[0] == exp
    assert run(dt, 16, k1s, k2s)[0] == exp


# ---- wave_fe_dev.h: one wavefront per signature (the same cases as tests/test_dev_wave_host.py) ----
def _wave_ops(dt):
    vp = C.c_void_p

    def fe_op(op, a_rows, b_rows=None):
        a = np.array(a_rows, dtype=np.uint32).reshape(4, 10)
        b = np.array(b_rows if b_rows is not None else a_rows, dtype=np.uint32).reshape(4, 10)
        out = np.zeros((4, 16), dtype=np.uint32)
        assert dt.devtest_wave_fe(op, 1, a.ctypes.data_as(vp), b.ctypes.data_as(vp), out.ctypes.data_as(vp)) == 0
        return out

    def pt_op(op, p, q):
        out = np.zeros((4, 31), dtype=np.uint32)
        assert dt.devtest_wave_pt(op, 1, p.ctypes.data_as(vp), q.ctypes.data_as(vp), out.ctypes.data_as(vp)) == 0
        return out
    return fe_op, pt_op


@pytest.mark.parametrize("case", ["mul_matches_bigint_at_every_magnitude", "weak_normalise_and_negate",
                                  "gather_scatter_and_is_zero", "zero_filter_never_misses", "sqrt_chain", "point_double_add_madd",
                                  "point_exceptional_cases_mixed_over_rows"])
def test_wave_arithmetic_on_gpu(dt, case):
    import wave_cases as WC
    getattr(WC, "check_" + case)(*_wave_ops(dt))


def test_wave_recover_on_gpu(dt, oracle):
    import wave_cases as WC

    def recover(h, sig, flags=0):
        out = np.zeros((64, 24), dtype=np.uint8)
        assert dt.devtest_wave_recover(1, h, sig, flags, out.ctypes.data_as(C.c_void_p)) == 0
        assert (out[:, :21] == out[0, :21]).all(), "lanes of the wavefront disagree"  # bytes 21..23 are padding
        return bool(out[0, 20]), out[0, :20].tobytes()
    WC.check_full_recover_matches_oracle(recover, oracle, rounds=4)
