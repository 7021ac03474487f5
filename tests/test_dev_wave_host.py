"""wave_fe_dev.h (one wavefront per signature, limbs spread over lanes) on the CPU.

The exact device source is compiled for the host with the cross-lane primitives replaced by a
64-coroutine lockstep emulator (csrc/wave_emul.h, csrc/host_wave_harness.hip) and compared with
Python big-int arithmetic and the oracle.  Test infrastructure only.
"""
import ctypes
import numpy as np
import pytest

from go_ibft_amd import build as B
import wave_cases as WC


@pytest.fixture(scope="module")
def wh():
    lib = ctypes.CDLL(B.build_wave_harness())
    lib.wvh_neg_limb.restype = ctypes.c_uint32
    return lib


def _ops(wh):
    def fe_op(op, a_rows, b_rows=None):
        a = np.array(a_rows, dtype=np.uint32).reshape(4, 10)
        b = np.array(b_rows if b_rows is not None else a_rows, dtype=np.uint32).reshape(4, 10)
        out = np.zeros((4, 16), dtype=np.uint32)
        wh.wvh_fe_op(op, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                     out.ctypes.data_as(ctypes.c_void_p))
        return out

    def pt_op(op, p, q):
        out = np.zeros((4, 31), dtype=np.uint32)
        wh.wvh_pt_op(op, p.ctypes.data_as(ctypes.c_void_p), q.ctypes.data_as(ctypes.c_void_p),
                     out.ctypes.data_as(ctypes.c_void_p))
        return out
    return fe_op, pt_op


def test_neg_constants(wh):
    WC.check_neg_constants(wh.wvh_neg_limb)


@pytest.mark.parametrize("case", ["mul_matches_bigint_at_every_magnitude", "weak_normalise_and_negate",
                                  "gather_scatter_and_is_zero", "sqrt_chain", "point_double_add_madd",
                                  "point_exceptional_cases_mixed_over_rows"])
def test_wave_arithmetic(wh, case):
    getattr(WC, "check_" + case)(*_ops(wh))


def test_full_recover_matches_oracle(wh, oracle):
    wh.wvh_init_gtab()

    def recover(h, sig, flags=0):
        addr = np.zeros((64, 20), dtype=np.uint8)
        ok = np.zeros(64, dtype=np.int32)
        wh.wvh_recover(h, sig, flags, addr.ctypes.data_as(ctypes.c_void_p), ok.ctypes.data_as(ctypes.c_void_p))
        assert (ok == ok[0]).all() and (addr == addr[0]).all(), "lanes of the wavefront disagree"
        return bool(ok[0]), addr[0].tobytes()
    WC.check_full_recover_matches_oracle(recover, oracle)
