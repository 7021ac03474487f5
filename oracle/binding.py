"""oracle/binding.py — ctypes binding of libibft_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libibft_oracle.so")
_SRCS = ["keccak.c", "sha256.c", "secp256k1.c", "recover_tuned.inc", "ibft_oracle.c", "ibft_oracle.h"]

FLAG_STRICT_LOW_S = 1
ROW_NIL, ROW_BADLEN, ROW_HASH_BAD = 1, 2, 4


def build(force: bool = False) -> str:
    stale = force or not os.path.exists(_LIB) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_LIB) for s in _SRCS)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libibft_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class Tally(C.Structure):
    _fields_ = [("quorum_lo", C.c_uint64), ("quorum_hi", C.c_uint64),
                ("power_lo", C.c_uint64), ("power_hi", C.c_uint64),
                ("valid_rows", C.c_uint32), ("distinct_senders", C.c_uint32),
                ("has_quorum", C.c_uint32), ("reserved", C.c_uint32)]

    @property
    def power(self) -> int:
        return self.power_lo | (self.power_hi << 64)

    @property
    def quorum(self) -> int:
        return self.quorum_lo | (self.quorum_hi << 64)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, u8p = C.c_void_p, C.c_char_p
        L.orc_keccak256.argtypes = [u8p, C.c_size_t, u8p]
        L.orc_pubkey.argtypes = [u8p, u8p]; L.orc_pubkey.restype = C.c_int
        L.orc_address.argtypes = [u8p, u8p]
        L.orc_sign.argtypes = [u8p, u8p, u8p]; L.orc_sign.restype = C.c_int
        L.orc_ecrecover.argtypes = [u8p, u8p, C.c_uint32, u8p]; L.orc_ecrecover.restype = C.c_int
        L.orc_recover_address.argtypes = [u8p, u8p, C.c_uint32, u8p]; L.orc_recover_address.restype = C.c_int
        for f in (L.orc_fe_mul, L.orc_sc_mul):
            f.argtypes = [u8p, u8p, u8p]
        for f in (L.orc_fe_inv, L.orc_sc_inv):
            f.argtypes = [u8p, u8p]
        L.orc_fe_sqrt.argtypes = [u8p, u8p]; L.orc_fe_sqrt.restype = C.c_int
        L.orc_ecmult2.argtypes = [u8p, u8p, u8p, u8p]; L.orc_ecmult2.restype = C.c_int
        L.orc_valset_new.argtypes = [vp, vp, C.c_size_t]; L.orc_valset_new.restype = vp
        L.orc_valset_free.argtypes = [vp]
        L.orc_valset_index.argtypes = [vp, u8p]; L.orc_valset_index.restype = C.c_int
        L.orc_valset_quorum.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_proposal_hash.argtypes = [u8p, C.c_size_t, C.c_uint64, u8p]
        L.orc_verify_hashes.argtypes = [u8p, C.c_size_t, C.c_uint64, vp, vp, C.c_size_t, vp]
        L.orc_verify_seals.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, vp]
        L.orc_verify_seals_mt.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, vp, C.c_int]
        L.orc_verify_senders.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, vp]
        L.orc_tally.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(Tally)]
        _lib = L
    return _lib


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape is not None:
        a = a.reshape(shape)
    return a


def keccak256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_keccak256(bytes(data), len(data), out)
    return out.raw


def sponge256(data: bytes, pad: int) -> bytes:
    """rate-136 sponge with the first padding byte given: 0x01 = keccak256, 0x06 = NIST SHA3-256"""
    out = C.create_string_buffer(32)
    lib().orc_sponge256(bytes(data), C.c_size_t(len(data)), C.c_uint8(pad), out)
    return out.raw


def pubkey(sk32: bytes) -> bytes | None:
    out = C.create_string_buffer(64)
    return out.raw if lib().orc_pubkey(sk32, out) else None


def address(pub64: bytes) -> bytes:
    out = C.create_string_buffer(20)
    lib().orc_address(pub64, out)
    return out.raw


def sign(sk32: bytes, digest32: bytes) -> bytes:
    out = C.create_string_buffer(65)
    if not lib().orc_sign(sk32, digest32, out):
        raise ValueError("orc_sign failed")
    return out.raw


def sign_rfc6979(sk32: bytes, digest32: bytes) -> bytes:
    """r ‖ s ‖ v with the deterministic nonce of RFC 6979 (HMAC-SHA-256): what the published secp256k1 vectors were made with"""
    out = C.create_string_buffer(65)
    if not lib().orc_sign_rfc6979(sk32, digest32, out):
        raise ValueError("orc_sign_rfc6979 failed")
    return out.raw


def sha256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_sha256(bytes(data), C.c_size_t(len(data)), out)
    return out.raw


def hmac_sha256(key: bytes, msg: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_hmac_sha256(bytes(key), C.c_size_t(len(key)), bytes(msg), C.c_size_t(len(msg)), None, C.c_size_t(0), out)
    return out.raw


def ecrecover(digest32: bytes, sig65: bytes, flags: int = 0) -> bytes | None:
    out = C.create_string_buffer(64)
    return out.raw if lib().orc_ecrecover(digest32, sig65, flags, out) else None


def recover_address(digest32: bytes, sig65: bytes, flags: int = 0) -> bytes | None:
    out = C.create_string_buffer(20)
    return out.raw if lib().orc_recover_address(digest32, sig65, flags, out) else None


def _bin2(fn, a: bytes, b: bytes) -> bytes:
    out = C.create_string_buffer(32)
    fn(a, b, out)
    return out.raw


def fe_mul(a, b): return _bin2(lib().orc_fe_mul, a, b)
def sc_mul(a, b): return _bin2(lib().orc_sc_mul, a, b)


def fe_inv(a):
    out = C.create_string_buffer(32); lib().orc_fe_inv(a, out); return out.raw


def sc_inv(a):
    out = C.create_string_buffer(32); lib().orc_sc_inv(a, out); return out.raw


def fe_sqrt(a):
    out = C.create_string_buffer(32)
    return out.raw if lib().orc_fe_sqrt(a, out) else None


def ecmult2(k1: bytes, k2: bytes, p64: bytes) -> bytes | None:
    out = C.create_string_buffer(64)
    return out.raw if lib().orc_ecmult2(k1, k2, p64, out) else None


def proposal_hash(raw: bytes, round_: int) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_proposal_hash(raw, len(raw), round_, out)
    return out.raw


class ValSet:
    """Validator set: addresses (n×20 uint8) and u64 voting powers."""

    def __init__(self, addrs20: np.ndarray, power: np.ndarray):
        self.addrs = _u8(addrs20, (-1, 20))
        self.power = np.ascontiguousarray(power, dtype=np.uint64)
        assert len(self.addrs) == len(self.power)
        self.h = lib().orc_valset_new(_p(self.addrs), _p(self.power), len(self.power))
        if not self.h:
            raise ValueError("total voting power is zero or less")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_valset_free(self.h)
            self.h = None

    def index(self, addr20: bytes) -> int:
        return lib().orc_valset_index(self.h, addr20)

    @property
    def quorum(self) -> int:
        lo, hi = C.c_uint64(), C.c_uint64()
        lib().orc_valset_quorum(self.h, C.byref(lo), C.byref(hi))
        return lo.value | (hi.value << 64)


def verify_hashes(raw: bytes, round_: int, hash32: np.ndarray, hash_len: np.ndarray) -> np.ndarray:
    hash32 = _u8(hash32, (-1, 32)); hash_len = _u8(hash_len)
    n = len(hash_len)
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_verify_hashes(raw, len(raw), round_, _p(hash32), _p(hash_len), n, _p(out))
    return out


# ---- the tuned recovery (recover_tuned.inc): bench.py's cpu_baseline leg and its own tests; never the checker ----
def ecrecover_tuned(digest32: bytes, sig65: bytes, flags: int = 0) -> bytes | None:
    out = C.create_string_buffer(64)
    return out.raw if lib().orc_ecrecover_tuned(digest32, sig65, C.c_uint32(flags), out) else None


def recover_address_tuned(digest32: bytes, sig65: bytes, flags: int = 0) -> bytes | None:
    out = C.create_string_buffer(20)
    return out.raw if lib().orc_recover_address_tuned(digest32, sig65, C.c_uint32(flags), out) else None


def verify_seals_tuned(vs: ValSet, hash32, sig65, signer20, pre_flags=None, flags: int = 0, nthreads: int = 1) -> np.ndarray:
    hash32 = _u8(hash32, (-1, 32)); sig65 = _u8(sig65, (-1, 65)); signer20 = _u8(signer20, (-1, 20))
    n = len(sig65)
    pre = None if pre_flags is None else _u8(pre_flags)
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_verify_seals_tuned_mt.argtypes = lib().orc_verify_seals_mt.argtypes
    lib().orc_verify_seals_tuned_mt(vs.h, _p(hash32), _p(sig65), _p(signer20), _p(pre), n, flags, _p(out), nthreads)
    return out


def tuned_fe_ops(a32: bytes, b32: bytes):
    """(a·b, a², a⁻¹ (0 for a = 0), √a or None) mod p from the tuned path's field code"""
    m, q, i, r = (C.create_string_buffer(32) for _ in range(4))
    ok = C.c_int(0)
    lib().orc_tuned_fe_ops(a32, b32, m, q, i, r, C.byref(ok))
    return m.raw, q.raw, i.raw, (r.raw if ok.value else None)


def tuned_fe_lazy(a32: bytes, b32: bytes) -> bytes:
    """(8·(a + b) − b)·(2·(b − 3a))² mod p through unreduced sums, negations and small multiples of the lazy field"""
    out = C.create_string_buffer(32)
    lib().orc_tuned_fe_lazy(a32, b32, out)
    return out.raw


def tuned_fe_inv_gcd(a32: bytes) -> bytes:
    """a⁻¹ mod p by divsteps (computed as 5·(5a)⁻¹ from an unreduced operand)"""
    out = C.create_string_buffer(32)
    lib().orc_tuned_fe_inv_gcd(a32, out)
    return out.raw


def tuned_sc_inv(a32: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_tuned_sc_inv(a32, out)
    return out.raw


def tuned_glv_split(k32: bytes):
    """(|k1|, negative?, |k2|, negative?) with k ≡ k1 + k2·λ (mod n)"""
    k1, k2 = C.create_string_buffer(32), C.create_string_buffer(32)
    n1, n2 = C.c_int(0), C.c_int(0)
    lib().orc_tuned_glv_split(k32, k1, k2, C.byref(n1), C.byref(n2))
    return int.from_bytes(k1.raw, "big"), bool(n1.value), int.from_bytes(k2.raw, "big"), bool(n2.value)


def tuned_wnaf5(k32: bytes) -> list[int]:
    d = (C.c_int8 * 132)()
    n = lib().orc_tuned_wnaf5(k32, d)
    return [int(d[i]) for i in range(n)]


def tuned_ecmult_var(k32: bytes, p64: bytes) -> bytes | None:
    out = C.create_string_buffer(64)
    return out.raw if lib().orc_tuned_ecmult_var(k32, p64, out) else None


def verify_seals(vs: ValSet, hash32, sig65, signer20, pre_flags=None, flags: int = 0,
                 nthreads: int = 1) -> np.ndarray:
    hash32 = _u8(hash32, (-1, 32)); sig65 = _u8(sig65, (-1, 65)); signer20 = _u8(signer20, (-1, 20))
    n = len(sig65)
    pre = None if pre_flags is None else _u8(pre_flags)
    out = np.zeros(n, dtype=np.uint8)
    if nthreads > 1:
        lib().orc_verify_seals_mt(vs.h, _p(hash32), _p(sig65), _p(signer20), _p(pre), n, flags, _p(out), nthreads)
    else:
        lib().orc_verify_seals(vs.h, _p(hash32), _p(sig65), _p(signer20), _p(pre), n, flags, _p(out))
    return out


def verify_senders(vs: ValSet, payload: bytes, off: np.ndarray, sig65, from20, pre_flags=None,
                   flags: int = 0, nthreads: int = 1) -> np.ndarray:
    pl = np.frombuffer(bytes(payload) or b"\0", dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    sig65 = _u8(sig65, (-1, 65)); from20 = _u8(from20, (-1, 20))
    n = len(sig65)
    pre = None if pre_flags is None else _u8(pre_flags)
    out = np.zeros(n, dtype=np.uint8)
    if nthreads > 1 and n >= 4 * nthreads:
        # rows are independent and the offsets are absolute: every thread takes a contiguous range of rows over the
        # same payload buffer (the C loop runs without the GIL)
        from concurrent.futures import ThreadPoolExecutor
        cuts = [n * k // nthreads for k in range(nthreads + 1)]

        def part(k):
            lo, hi = cuts[k], cuts[k + 1]
            lib().orc_verify_senders(vs.h, _p(pl), _p(off[lo:hi + 1]), _p(sig65[lo:hi]), _p(from20[lo:hi]),
                                     _p(pre[lo:hi]) if pre is not None else None, hi - lo, flags, _p(out[lo:hi]))
        with ThreadPoolExecutor(max_workers=nthreads) as ex:
            list(ex.map(part, range(nthreads)))
        return out
    lib().orc_verify_senders(vs.h, _p(pl), _p(off), _p(sig65), _p(from20), _p(pre), n, flags, _p(out))
    return out


def tally(vs: ValSet | None, sender20, verdict) -> Tally:
    sender20 = _u8(sender20, (-1, 20)); verdict = _u8(verdict)
    t = Tally()
    lib().orc_tally(vs.h if vs is not None else None, _p(sender20), _p(verdict), len(verdict), C.byref(t))
    return t
