"""ctypes mirror of include/ibftgpu.h — the batch form of go-ibft's Verifier.

Method names follow the reference interface they replace
(/root/reference/core/backend.go:37-56):

    IsValidProposalHash(proposal, hash)        -> BatchVerifier.is_valid_proposal_hash(raw, round, hashes)
    IsValidCommittedSeal(proposalHash, seal)   -> BatchVerifier.is_valid_committed_seal(hashes, seals, signers)
    IsValidValidator(msg)                      -> BatchVerifier.is_valid_validator(payloads, sigs, froms)
    ValidatorManager.HasQuorum(senders)        -> the Tally returned next to each verdict array

Every method returns a numpy bool array (one verdict per row) — the value the
per-message reference predicate would have returned.  No CPU fallback: a missing
library or device raises ``GpuUnavailable``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .build import LIB

FLAG_STRICT_LOW_S = 1
FLAG_PUBKEY_CACHE = 2   # warm path: verify against per-validator tables once a key has been recovered
ROW_NIL, ROW_BADLEN, ROW_HASH_BAD = 1, 2, 4
KERNEL_AUTO, KERNEL_LANE, KERNEL_WAVE = 0, 1, 2

EXPORTS = [
    "ibft_version", "ibft_strerror", "ibft_last_error", "ibft_ctx_create", "ibft_ctx_destroy",
    "ibft_set_validators", "ibft_verify_hashes", "ibft_proposal_hash", "ibft_verify_seals",
    "ibft_verify_senders", "ibft_tally", "ibft_seals_stage", "ibft_seals_launch", "ibft_seals_fetch",
    "ibft_seals_device_ptrs", "ibft_seals_export", "ibft_last_kernel_ms", "ibft_cache_stats", "ibft_last_dispatch", "ibft_sync",
    "ibft_verify_senders_wire", "ibft_wire_stage_seals", "ibft_seals_export_on",
    "ibft_set_validators_u256", "ibft_last_tally_wide",
    "ibft_shard_range", "ibft_exchange_layout", "ibft_comm_unique_id", "ibft_comm_init", "ibft_comm_destroy",
    "ibft_seals_exchange", "ibft_seals_fetch_merged", "ibft_seals_run", "ibft_verify_hashes_digest", "ibft_set_kernel_timing",
    "ibft_group_create", "ibft_group_destroy", "ibft_group_size", "ibft_group_ctx", "ibft_group_set_validators",
    "ibft_group_set_validators_u256", "ibft_group_verify_seals", "ibft_group_verify_senders", "ibft_group_verify_messages", "ibft_group_verify_certificates_wire",
    "ibft_group_is_local", "ibft_sign_seals", "ibft_verify_messages", "ibft_pinned_alloc", "ibft_pinned_free", "ibft_column_stats",
    "ibft_verify_messages_wire", "ibft_forget_proposal", "ibft_verify_certificates_wire", "ibft_keccak256",
    "ibft_cache_memory", "ibft_tally_prepare", "ibft_comm_info", "ibft_set_seal_digest", "ibft_group_set_seal_digest",
    "ibft_seals_stage_next", "ibft_seals_swap", "ibft_last_cold_table", "ibft_seals_submit", "ibft_seals_collect",
    "ibft_comm_preload", "ibft_issue_probe", "ibft_seals_rows", "ibft_pipeline_stats",
]
EXPORTS_SINCE = {"ibft_pipeline_stats": 4}   # exports younger than version 3: the version that brought them
COMM_ID_BYTES = 128
E_RCCL = -8

WIRE_OK, WIRE_NEEDS_HOST = 0, 1
WIRE_CLASS_NEEDS_HOST, WIRE_CLASS_CLOSURE = 1, 2
# ibft_wire_row_t (include/ibftgpu.h)
WIRE_ROW = np.dtype([("height", "<u8"), ("round", "<u8"), ("status", "u1"), ("type", "u1"), ("payload_kind", "u1"),
                     ("has_view", "u1"), ("hash_len", "u1"), ("seal_len", "u1"), ("from_len", "u1"), ("sig_len", "u1"),
                     ("from", "u1", 20), ("proposal_hash", "u1", 32), ("pad", "u1", 4)])
assert WIRE_ROW.itemsize == 80
# ibft_cert_node_t (include/ibftgpu.h)
CERT_NODE = np.dtype([("off", "<u4"), ("len", "<u4"), ("parent", "<u4"), ("ordinal", "<u4"), ("first_child", "<u4"),
                      ("n_children", "<u4"), ("raw_off", "<u4"), ("raw_len", "<u4"), ("proposal_round", "<u8"),
                      ("cut0", "<u4"), ("cut1", "<u4"), ("level", "u1"), ("role", "u1"), ("flags", "u1"), ("pad", "u1", 5)])
assert CERT_NODE.itemsize == 56
CERT_CLASS_NEEDS_HOST, CERT_CLASS_DIGEST_BY_HOST, CERT_CLASS_PROPOSAL_BY_HOST = 1, 2, 4
CERT_NO_PARENT = 0xFFFFFFFF


class GpuUnavailable(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("max_rows", C.c_uint32),
                ("kernel", C.c_uint32)]


class Tally(C.Structure):
    _fields_ = [("quorum_lo", C.c_uint64), ("quorum_hi", C.c_uint64),
                ("power_lo", C.c_uint64), ("power_hi", C.c_uint64),
                ("valid_rows", C.c_uint32), ("distinct_senders", C.c_uint32),
                ("has_quorum", C.c_uint32), ("shard_overlap", C.c_uint32),
                ("proposer_rows", C.c_uint32), ("reserved", C.c_uint32)]

    @property
    def power(self) -> int:
        return self.power_lo | (self.power_hi << 64)

    @property
    def quorum(self) -> int:
        return self.quorum_lo | (self.quorum_hi << 64)


class TallyWide(C.Structure):
    """ibft_tally_wide_t: full-width (320-bit) power and quorum of the last tally."""
    _fields_ = [("quorum_w", C.c_uint64 * 5), ("power_w", C.c_uint64 * 5), ("has_quorum", C.c_uint32),
                ("reserved", C.c_uint32)]

    @property
    def power(self) -> int:
        return sum(int(w) << (64 * i) for i, w in enumerate(self.power_w))

    @property
    def quorum(self) -> int:
        return sum(int(w) << (64 * i) for i, w in enumerate(self.quorum_w))


def powers_be32(powers) -> np.ndarray:
    """list of Python ints → n × 32 big-endian bytes (what big.Int.FillBytes(make([]byte, 32)) produces)"""
    return np.frombuffer(b"".join(int(p).to_bytes(32, "big") for p in powers), dtype=np.uint8).reshape(-1, 32).copy()


_lib = None
ABI_VERSION = 4   # include/ibftgpu.h: ibft_version()  (3: the round-5/6 exports; 4: ibft_pipeline_stats)


def load_library() -> C.CDLL:
    """dlopen libibftgpu.so and declare the prototypes (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = os.environ.get("IBFT_GPU_LIB") or LIB   # IBFT_GPU_LIB: A/B experiments with another build of the library
    if not os.path.exists(lib_path):
        raise GpuUnavailable(f"{lib_path} is not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(lib_path)
    vp = C.c_void_p
    L.ibft_version.restype = C.c_int
    # IBFT_MIN_ABI (A/B tools only: tools/kernel_ab.py against an OLDER build of the library): accept a library of that version and
    # do without the exports that came after it — never below 3 (the struct layouts changed there)
    need = max(3, min(ABI_VERSION, int(os.environ.get("IBFT_MIN_ABI", ABI_VERSION))))
    if L.ibft_version() < need:   # a stale .so writes past the caller's ibft_tally_t (48 → 56 bytes at version 2)
        raise GpuUnavailable(f"{lib_path}: ibft_version() = {L.ibft_version()}, this binding needs >= {need} — rebuild")
    later = {name for name, since in EXPORTS_SINCE.items() if since > need}
    missing = [name for name in EXPORTS if name not in later and not hasattr(L, name)]   # BEFORE the prototypes below touch any of them
    if missing:
        raise GpuUnavailable(f"{lib_path} lacks {', '.join(missing)} — a stale build: rebuild")
    L.ibft_strerror.argtypes = [C.c_int]; L.ibft_strerror.restype = C.c_char_p
    L.ibft_last_error.argtypes = [vp]; L.ibft_last_error.restype = C.c_char_p
    L.ibft_ctx_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
    L.ibft_ctx_destroy.argtypes = [vp]; L.ibft_ctx_destroy.restype = None
    L.ibft_set_validators.argtypes = [vp, C.c_uint64, vp, vp, C.c_size_t]
    L.ibft_verify_hashes.argtypes = [vp, vp, C.c_size_t, C.c_uint64, vp, vp, C.c_size_t, vp]
    L.ibft_proposal_hash.argtypes = [vp, vp, C.c_size_t, C.c_uint64, vp]
    L.ibft_verify_hashes_digest.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.ibft_verify_seals.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp, C.POINTER(Tally)]
    L.ibft_verify_senders.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.POINTER(Tally)]
    L.ibft_tally.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(Tally)]
    L.ibft_tally_prepare.argtypes = [vp, vp, vp, C.c_size_t, vp, C.POINTER(Tally)]
    L.ibft_set_seal_digest.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
    L.ibft_group_set_seal_digest.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
    L.ibft_comm_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.ibft_seals_stage.argtypes = [vp, vp, vp, vp, vp, C.c_size_t]
    L.ibft_seals_launch.argtypes = [vp, C.c_uint32]
    L.ibft_seals_fetch.argtypes = [vp, vp, C.POINTER(Tally)]
    L.ibft_seals_run.argtypes = [vp, vp, C.POINTER(Tally)]
    L.ibft_seals_stage_next.argtypes = [vp, vp, vp, vp, vp, C.c_size_t]
    L.ibft_seals_swap.argtypes = [vp, C.c_int]
    L.ibft_last_cold_table.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.ibft_seals_submit.argtypes = [vp]
    L.ibft_seals_collect.argtypes = [vp, vp, C.POINTER(Tally)]
    L.ibft_sign_seals.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.ibft_verify_messages_wire.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint64, C.c_uint64, vp, C.c_size_t, C.c_uint64, vp, vp, vp, vp, vp,
                                            vp, C.POINTER(Tally)]
    L.ibft_forget_proposal.argtypes = [vp]
    L.ibft_pinned_alloc.argtypes = [C.c_size_t]; L.ibft_pinned_alloc.restype = vp
    L.ibft_pinned_free.argtypes = [vp]; L.ibft_pinned_free.restype = None
    L.ibft_column_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.ibft_verify_messages.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, vp, vp, vp, vp,
                                       C.POINTER(Tally)]
    L.ibft_seals_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
    L.ibft_seals_export.argtypes = [vp, vp, vp]
    L.ibft_seals_export_on.argtypes = [vp, vp, vp, vp]
    L.ibft_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.ibft_set_kernel_timing.argtypes = [vp, C.c_uint32]
    L.ibft_cache_stats.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 4
    L.ibft_last_dispatch.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ibft_sync.argtypes = [vp]
    L.ibft_issue_probe.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ibft_seals_rows.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if hasattr(L, "ibft_pipeline_stats"):
        L.ibft_pipeline_stats.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ibft_verify_senders_wire.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.POINTER(Tally)]
    L.ibft_wire_stage_seals.argtypes = [vp]
    L.ibft_verify_certificates_wire.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), vp, vp, vp, vp, vp, vp]
    L.ibft_set_validators_u256.argtypes = [vp, C.c_uint64, vp, vp, C.c_size_t]
    L.ibft_last_tally_wide.argtypes = [vp, C.POINTER(TallyWide)]
    L.ibft_shard_range.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ibft_exchange_layout.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ibft_comm_unique_id.argtypes = [vp]
    L.ibft_comm_init.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.ibft_comm_destroy.argtypes = [vp]
    L.ibft_seals_exchange.argtypes = [vp, C.c_uint64]
    L.ibft_seals_fetch_merged.argtypes = [vp, vp, C.POINTER(Tally)]
    L.ibft_group_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.ibft_group_destroy.argtypes = [vp]; L.ibft_group_destroy.restype = None
    L.ibft_group_size.argtypes = [vp]; L.ibft_group_size.restype = C.c_uint32
    L.ibft_group_ctx.argtypes = [vp, C.c_uint32]; L.ibft_group_ctx.restype = vp
    L.ibft_group_set_validators.argtypes = [vp, C.c_uint64, vp, vp, C.c_size_t]
    L.ibft_group_set_validators_u256.argtypes = [vp, C.c_uint64, vp, vp, C.c_size_t]
    L.ibft_group_verify_seals.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp, C.POINTER(Tally)]
    L.ibft_group_verify_senders.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.POINTER(Tally)]
    L.ibft_group_verify_messages.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, vp,
                                             vp, vp, vp, C.POINTER(Tally)]
    L.ibft_group_verify_certificates_wire.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    L.ibft_group_is_local.argtypes = [vp]
    L.ibft_keccak256.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp]
    L.ibft_cache_memory.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for name in EXPORTS:  # fail loudly on a stale build that lacks a declared symbol
        if name not in later:
            getattr(L, name)
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _PinnedOwner:
    """keeps one ibft_pinned_alloc block alive for as long as a numpy view of it exists"""

    def __init__(self, nbytes: int):
        self._L = load_library()
        self.ptr = self._L.ibft_pinned_alloc(nbytes)
        if not self.ptr:
            raise GpuUnavailable("ibft_pinned_alloc failed")
        self.buf = (C.c_uint8 * max(nbytes, 1)).from_address(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            self._L.ibft_pinned_free(self.ptr)
            self.ptr = None


def pinned_copy(a) -> np.ndarray:
    """a copy of `a` (array or bytes) in page-locked memory from ibft_pinned_alloc: what a caller's flatten step
    should write its columns into (include/ibftgpu.h, "pinned column buffers")"""
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a)
    own = _PinnedOwner(a.nbytes)
    out = np.frombuffer(own.buf, dtype=a.dtype, count=a.size).reshape(a.shape)  # .base chain keeps `own.buf` alive
    out[...] = a
    own.buf._owner = own  # the view's base chain reaches the ctypes array, which now keeps the block's owner alive
    return out


def _bytes_col(x) -> np.ndarray:
    """a byte column as the C ABI wants it, without copying arrays that already are one (pinned buffers stay pinned)"""
    if isinstance(x, np.ndarray) and x.dtype == np.uint8 and x.flags["C_CONTIGUOUS"] and x.size:
        return x.reshape(-1)
    return np.frombuffer(bytes(x) or b"\0", dtype=np.uint8)


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """ibft_shard_range (pure): rows [lo, hi) of rank `rank`"""
    lo, hi = C.c_uint64(), C.c_uint64()
    rc = load_library().ibft_shard_range(n_total, rank, world, C.byref(lo), C.byref(hi))
    if rc:
        raise ValueError(f"ibft_shard_range: {rc}")
    return lo.value, hi.value


def exchange_layout(n_total: int, world: int, n_validators: int, n_masks: int = 1) -> tuple[int, int, int]:
    """ibft_exchange_layout (pure): (verdict words per rank, bitmap words per rank, u64 slots of the exchange buffer)"""
    w, b, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = load_library().ibft_exchange_layout(n_total, world, n_validators, n_masks, C.byref(w), C.byref(b), C.byref(s))
    if rc:
        raise ValueError(f"ibft_exchange_layout: {rc}")
    return w.value, b.value, s.value


def keccak256(a: bytes, b: bytes = b"") -> bytes:
    """ibft_keccak256: keccak256(a ‖ b) by the library's host routine (no device needed)"""
    a, b = bytes(a), bytes(b)
    out = C.create_string_buffer(32)
    rc = load_library().ibft_keccak256(a or None, len(a), b or None, len(b), out)
    if rc:
        raise ValueError(f"ibft_keccak256: {rc}")
    return out.raw


def comm_preload() -> None:
    """ibft_comm_preload: map librccl now (before anything else in the process maps another copy under the same SONAME)"""
    rc = load_library().ibft_comm_preload()
    if rc != 0:
        raise GpuUnavailable(f"ibft_comm_preload: {load_library().ibft_strerror(rc).decode()} ({rc})")


def comm_unique_id() -> bytes:
    """ibft_comm_unique_id: rank 0 creates it and hands it to the other ranks (loads librccl)"""
    buf = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
    L = load_library()
    rc = L.ibft_comm_unique_id(_p(buf))
    if rc:
        raise GpuUnavailable(f"ibft_comm_unique_id: {L.ibft_strerror(rc).decode()} ({rc})")
    return buf.tobytes()


def _u8(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a.reshape(shape) if shape is not None else a


def mask_to_bool(mask: np.ndarray, n: int) -> np.ndarray:
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
    return bits[:n].astype(bool)


def bool_to_mask(v: np.ndarray) -> np.ndarray:
    n = len(v)
    words = (n + 63) // 64
    bits = np.zeros(words * 64, dtype=np.uint8)
    bits[:n] = np.asarray(v, dtype=np.uint8)
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


class BatchVerifier:
    """One ibft_ctx: one HIP stream + resident columns on one MI355X."""

    def __init__(self, device: int = 0, flags: int = 0, max_rows: int = 65536, kernel: int = KERNEL_AUTO):
        self._L = load_library()
        self._h = C.c_void_p()
        cfg = Cfg(device, flags, max_rows, kernel)
        rc = self._L.ibft_ctx_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise GpuUnavailable(f"ibft_ctx_create: {self._L.ibft_strerror(rc).decode()} ({rc})")
        self.max_rows = max_rows

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.ibft_ctx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _chk(self, rc: int, what: str):
        if rc != 0:
            detail = self._L.ibft_last_error(self._h).decode()
            raise RuntimeError(f"{what}: {self._L.ibft_strerror(rc).decode()} ({rc}) {detail}")

    # ValidatorBackend.GetVotingPowers -> device table
    def set_validators(self, height: int, addrs20, power) -> None:
        a = _u8(addrs20, (-1, 20)); p = np.ascontiguousarray(power, dtype=np.uint64)
        assert len(a) == len(p)
        self._chk(self._L.ibft_set_validators(self._h, height, _p(a), _p(p), len(p)), "ibft_set_validators")

    def set_validators_u256(self, height: int, addrs20, powers) -> None:
        """powers: Python ints < 2^256 (GetVotingPowers' *big.Int values)"""
        a = _u8(addrs20, (-1, 20)); p = powers_be32(powers)
        assert len(a) == len(p)
        self._chk(self._L.ibft_set_validators_u256(self._h, height, _p(a), _p(p), len(p)), "ibft_set_validators_u256")

    def last_tally_wide(self) -> TallyWide:
        t = TallyWide()
        self._chk(self._L.ibft_last_tally_wide(self._h, C.byref(t)), "ibft_last_tally_wide")
        return t

    def set_seal_digest(self, suffix: bytes | None) -> None:
        """the embedding Backend's seal convention: None = the seal signs the proposalHash itself (default),
        bytes = it signs keccak256(proposalHash ‖ suffix)  (include/ibftgpu.h: ibft_set_seal_digest)"""
        b = None if not suffix else np.frombuffer(bytes(suffix), dtype=np.uint8)
        self._chk(self._L.ibft_set_seal_digest(self._h, 0 if suffix is None else 1, _p(b), len(suffix or b"")),
                  "ibft_set_seal_digest")
        self._staged = 0

    def try_set_validators(self, height: int, addrs20, power) -> int:
        a = _u8(addrs20, (-1, 20)); p = np.ascontiguousarray(power, dtype=np.uint64)
        return self._L.ibft_set_validators(self._h, height, _p(a), _p(p), len(p))

    def proposal_hash(self, raw: bytes, round_: int) -> bytes:
        out = np.zeros(32, dtype=np.uint8)
        r = np.frombuffer(bytes(raw) or b"\0", dtype=np.uint8)
        self._chk(self._L.ibft_proposal_hash(self._h, _p(r), len(raw), round_, _p(out)), "ibft_proposal_hash")
        return out.tobytes()

    # Verifier.IsValidProposalHash, batched
    def is_valid_proposal_hash(self, raw: bytes, round_: int, hash32, hash_len) -> np.ndarray:
        h = _u8(hash32, (-1, 32)); hl = _u8(hash_len)
        n = len(hl)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        r = np.frombuffer(bytes(raw) or b"\0", dtype=np.uint8)
        self._chk(self._L.ibft_verify_hashes(self._h, _p(r), len(raw), round_, _p(h), _p(hl), n, _p(mask)),
                  "ibft_verify_hashes")
        return mask_to_bool(mask, n)

    def is_valid_proposal_hash_digest(self, digest32: bytes, hash32, hash_len) -> np.ndarray:
        """IsValidProposalHash with keccak(proposal) already known to the caller: a compare per row"""
        h = _u8(hash32, (-1, 32)); hl = _u8(hash_len)
        n = len(hl)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        d = np.frombuffer(bytes(digest32), dtype=np.uint8).copy()
        assert len(d) == 32
        self._chk(self._L.ibft_verify_hashes_digest(self._h, _p(d), _p(h), _p(hl), n, _p(mask)), "ibft_verify_hashes_digest")
        return mask_to_bool(mask, n)

    # Verifier.IsValidCommittedSeal, batched
    def is_valid_committed_seal(self, hash32, sig65, signer20, pre_flags=None):
        h = _u8(hash32, (-1, 32)); s = _u8(sig65, (-1, 65)); f = _u8(signer20, (-1, 20))
        n = len(s)
        pre = None if pre_flags is None else _u8(pre_flags)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_verify_seals(self._h, _p(h), _p(s), _p(f), _p(pre), n, _p(mask), C.byref(t)),
                  "ibft_verify_seals")
        return mask_to_bool(mask, n), t

    # Verifier.IsValidValidator, batched
    def is_valid_validator(self, payload: bytes, off, sig65, from20, pre_flags=None):
        pl = _bytes_col(payload)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        s = _u8(sig65, (-1, 65)); f = _u8(from20, (-1, 20))
        n = len(s)
        pre = None if pre_flags is None else _u8(pre_flags)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_verify_senders(self._h, _p(pl), _p(off), _p(s), _p(f), _p(pre), n, _p(mask),
                                              C.byref(t)), "ibft_verify_senders")
        return mask_to_bool(mask, n), t

    # a whole PREPARE / COMMIT set in one call: IsValidValidator ∧ IsValidProposalHash (∧ IsValidCommittedSeal)
    def verify_messages(self, payload: bytes, off, msg_sig65, from20, hash32, hash_len, seal65=None, sender_pre=None, valid_pre=None,
                        raw: bytes | None = None, round_: int = 0, digest32: bytes | None = None, proposer: bytes | None = None):
        """→ (sender bool[n], valid bool[n], Tally over sender ∧ valid); seal65=None for a PREPARE set; proposer (20 bytes):
        the tally is ValidatorManager.HasPrepareQuorum with that proposer instead of HasQuorum"""
        pr = None if proposer is None else np.frombuffer(bytes(proposer), dtype=np.uint8)
        pl = _bytes_col(payload)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        s = _u8(msg_sig65, (-1, 65)); f = _u8(from20, (-1, 20)); h = _u8(hash32, (-1, 32)); hl = _u8(hash_len)
        n = len(s)
        sl = None if seal65 is None else _u8(seal65, (-1, 65))
        spre = None if sender_pre is None else _u8(sender_pre)
        vpre = None if valid_pre is None else _u8(valid_pre)
        rawb = None if raw is None else np.frombuffer(bytes(raw) or b"\0", dtype=np.uint8)
        dg = None if digest32 is None else np.frombuffer(bytes(digest32), dtype=np.uint8)
        ms = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        mv = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_verify_messages(self._h, _p(pl), _p(off), _p(s), _p(f), _p(h), _p(hl), _p(sl), _p(spre), _p(vpre), n,
                                               _p(rawb), 0 if raw is None else len(raw), round_, _p(dg), _p(pr), _p(ms), _p(mv),
                                               C.byref(t)), "ibft_verify_messages")
        return mask_to_bool(ms, n), mask_to_bool(mv, n), t

    def prepare_messages(self, payload, off, msg_sig65, from20, hash32, hash_len, seal65=None, sender_pre=None, valid_pre=None,
                         raw: bytes | None = None, round_: int = 0, digest32: bytes | None = None, proposer: bytes | None = None):
        """verify_messages with the argument marshalling done ONCE: for a caller whose columns live in fixed buffers that
        are refilled every round (what the integration prescribes).  Returns run() → (sender words u64[⌈n/64⌉], valid words,
        Tally); the word arrays are reused between calls.  Decode with mask_to_bool(words, n)."""
        pl = _bytes_col(payload)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        s = _u8(msg_sig65, (-1, 65)); f = _u8(from20, (-1, 20)); h = _u8(hash32, (-1, 32)); hl = _u8(hash_len)
        n = len(s)
        sl = None if seal65 is None else _u8(seal65, (-1, 65))
        spre = None if sender_pre is None else _u8(sender_pre)
        vpre = None if valid_pre is None else _u8(valid_pre)
        rawb = None if raw is None else np.frombuffer(bytes(raw) or b"\0", dtype=np.uint8)
        dg = None if digest32 is None else np.frombuffer(bytes(digest32), dtype=np.uint8)
        ms = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        mv = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        pr = None if proposer is None else np.frombuffer(bytes(proposer), dtype=np.uint8)
        keep = (pl, off, s, f, h, hl, sl, spre, vpre, rawb, dg, pr)  # the pointers below stay valid while these live
        args = (self._h, _p(pl), _p(off), _p(s), _p(f), _p(h), _p(hl), _p(sl), _p(spre), _p(vpre), n, _p(rawb),
                0 if raw is None else len(raw), round_, _p(dg), _p(pr), _p(ms), _p(mv), C.byref(t))
        fn, chk = self._L.ibft_verify_messages, self._chk

        def run(_keep=keep):
            chk(fn(*args), "ibft_verify_messages")
            return ms, mv, t
        run.n = n
        return run

    # §8f rank 3: a3 straight from the wire bytes (PREPARE / COMMIT); rows["status"] == WIRE_NEEDS_HOST
    # are not judged (verdict 0) and go through the protobuf runtime + is_valid_validator
    def is_valid_validator_wire(self, wire: bytes, off):
        wb = np.frombuffer(bytes(wire) or b"\0", dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = len(off) - 1
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        rows = np.zeros(n, dtype=WIRE_ROW)
        t = Tally()
        self._chk(self._L.ibft_verify_senders_wire(self._h, _p(wb), _p(off), n, _p(mask), _p(rows) if n else None,
                                                   C.byref(t)), "ibft_verify_senders_wire")
        self._staged = n
        return mask_to_bool(mask, n), rows, t

    def verify_messages_wire(self, wire, off, height: int, round_: int, raw: bytes | None = None, proposal_round: int | None = None,
                             digest32: bytes | None = None, want_rows: bool = True, proposer: bytes | None = None):
        """raw messages judged completely → (sender bool[n], valid bool[n], rows or class bytes, Tally): rows = the full
        parse results (want_rows) or the one routing byte per row (CLASS_* bits); see include/ibftgpu.h"""
        wb = _bytes_col(wire)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = len(off) - 1
        ms = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        mv = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        rows = np.zeros(n, dtype=WIRE_ROW) if want_rows else None
        cls = np.zeros(max(n, 1), dtype=np.uint8)
        rawb = None if raw is None else np.frombuffer(bytes(raw) or b"\0", dtype=np.uint8)
        dg = None if digest32 is None else np.frombuffer(bytes(digest32), dtype=np.uint8)
        t = Tally()
        self._chk(self._L.ibft_verify_messages_wire(self._h, _p(wb), _p(off), n, height, round_, _p(rawb),
                                                    0 if raw is None else len(raw),
                                                    round_ if proposal_round is None else proposal_round, _p(dg), _p(ms), _p(mv),
                                                    _p(cls), _p(rows) if (n and want_rows) else None,
                                                    _p(None if proposer is None else np.frombuffer(bytes(proposer), dtype=np.uint8)),
                                                    C.byref(t)),
                  "ibft_verify_messages_wire")
        self._last_class = cls[:n]
        return mask_to_bool(ms, n), mask_to_bool(mv, n), (rows if want_rows else cls[:n]), t

    def verify_certificates_wire(self, wire, off, rows_cap: int | None = None, want_rows: bool = True):
        """§8f rank 2 from bytes: the whole certificate tree of a batch of raw messages →
        (n_rows, nodes[CERT_NODE], rows[WIRE_ROW] or None, class u8[n_rows], sender bool[n_rows], hash bool[n_rows],
        self bool[n_rows]); see include/ibftgpu.h.  Raises RuntimeError (IBFT_E_TOOBIG) when the tree exceeds rows_cap."""
        wb = _bytes_col(wire)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = len(off) - 1
        cap = int(rows_cap if rows_cap is not None else self.max_rows)
        words = (cap + 63) // 64 or 1
        nodes = np.zeros(max(cap, 1), dtype=CERT_NODE)
        rows = np.zeros(max(cap, 1), dtype=WIRE_ROW) if want_rows else None
        cls = np.zeros(max(cap, 1), dtype=np.uint8)
        ms, mh, mself = (np.zeros(words, dtype=np.uint64) for _ in range(3))
        n_rows = C.c_size_t(0)
        self._chk(self._L.ibft_verify_certificates_wire(self._h, _p(wb), _p(off), n, cap, C.byref(n_rows), _p(nodes),
                                                        _p(rows) if want_rows else None, _p(cls), _p(ms), _p(mh), _p(mself)),
                  "ibft_verify_certificates_wire")
        k = int(n_rows.value)
        return (k, nodes[:k], rows[:k] if want_rows else None, cls[:k], mask_to_bool(ms, k), mask_to_bool(mh, k),
                mask_to_bool(mself, k))

    def wire_stage_seals(self):
        """the COMMIT seals of the last is_valid_validator_wire batch become the resident seal batch"""
        self._chk(self._L.ibft_wire_stage_seals(self._h), "ibft_wire_stage_seals")

    # ValidatorManager.HasQuorum over a caller-supplied verdict array
    def has_quorum(self, sender20, verdict) -> Tally:
        f = _u8(sender20, (-1, 20))
        m = bool_to_mask(np.asarray(verdict, dtype=bool))
        t = Tally()
        self._chk(self._L.ibft_tally(self._h, _p(f), _p(m), len(f), C.byref(t)), "ibft_tally")
        return t

    # ValidatorManager.HasPrepareQuorum (core/validator_manager.go:99-127) over a caller-supplied verdict array
    def has_prepare_quorum(self, sender20, verdict, proposer: bytes) -> Tally:
        f = _u8(sender20, (-1, 20))
        m = bool_to_mask(np.asarray(verdict, dtype=bool))
        pr = np.frombuffer(bytes(proposer), dtype=np.uint8)
        assert len(pr) == 20
        t = Tally()
        self._chk(self._L.ibft_tally_prepare(self._h, _p(f), _p(m), len(f), _p(pr), C.byref(t)), "ibft_tally_prepare")
        return t

    def comm_info(self):
        """(ranks, this rank, device) as the RCCL communicator itself reports them"""
        n, r, d = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
        self._chk(self._L.ibft_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)), "ibft_comm_info")
        return int(n.value), int(r.value), int(d.value)

    # staged / device-resident form of a2 (bench.py, multi-GPU)
    def seals_stage(self, hash32, sig65, signer20, pre_flags=None) -> int:
        h = _u8(hash32, (-1, 32)); s = _u8(sig65, (-1, 65)); f = _u8(signer20, (-1, 20))
        pre = None if pre_flags is None else _u8(pre_flags)
        self._chk(self._L.ibft_seals_stage(self._h, _p(h), _p(s), _p(f), _p(pre), len(s)), "ibft_seals_stage")
        self._staged = len(s)
        return len(s)

    def seals_stage_next(self, hash32, sig65, signer20, pre_flags=None) -> int:
        """the NEXT batch into the spare column set, asynchronously on the context's copy stream (the arrays are kept alive
        here until the swap; pass ibft_pinned_alloc memory — pinned_copy() — for the copy to overlap the kernels)"""
        h = _u8(hash32, (-1, 32)); s = _u8(sig65, (-1, 65)); f = _u8(signer20, (-1, 20))
        pre = None if pre_flags is None else _u8(pre_flags)
        self._next_cols = (h, s, f, pre)
        self._chk(self._L.ibft_seals_stage_next(self._h, _p(h), _p(s), _p(f), _p(pre), len(s)), "ibft_seals_stage_next")
        return len(s)

    def seals_swap(self, wait_for_copy: bool = True) -> None:
        """the staged batch becomes the resident one; per step of a stream: launch(k), stage_next(k+1), fetch(k), swap"""
        self._chk(self._L.ibft_seals_swap(self._h, 1 if wait_for_copy else 0), "ibft_seals_swap")
        self._staged = len(self._next_cols[1])
        self._held_cols, self._next_cols = self._next_cols, None   # (with wait_for_copy=0 the copy may still read them)

    def seals_submit(self) -> None:
        """one more pass over the resident batch, asynchronously (at most two in flight); results through seals_collect"""
        self._chk(self._L.ibft_seals_submit(self._h), "ibft_seals_submit")
        self._submitted = getattr(self, "_submitted", []) + [self._staged]

    def seals_collect(self):
        """the OLDEST submitted pass → (verdict bool[n], Tally); waits for that pass only"""
        n = self._submitted[0] if getattr(self, "_submitted", None) else 0
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_seals_collect(self._h, _p(mask), C.byref(t)), "ibft_seals_collect")
        self._submitted.pop(0)
        return mask_to_bool(mask, n), t

    def seals_rows(self):
        """ibft_seals_rows: (rows of the resident batch, rows of the oldest submitted pass or 0) as the LIBRARY counts them"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        self._chk(self._L.ibft_seals_rows(self._h, C.byref(a), C.byref(b)), "ibft_seals_rows")
        return int(a.value), int(b.value)

    def pipeline_stats(self):
        """ibft_pipeline_stats: (passes whose tally ran on the side stream, batches that went out as two launches)"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        if not hasattr(self._L, "ibft_pipeline_stats"):    # (an older build under IBFT_MIN_ABI: it has no side stream either)
            return 0, 0
        self._chk(self._L.ibft_pipeline_stats(self._h, C.byref(a), C.byref(b)), "ibft_pipeline_stats")
        return int(a.value), int(b.value)

    def seals_launch(self, repeat: int = 1) -> None:
        self._chk(self._L.ibft_seals_launch(self._h, repeat), "ibft_seals_launch")
        self._launched = self._staged

    def seals_fetch(self):
        n = getattr(self, "_launched", self._staged)   # the batch of the last launch (a swap may have come in between)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_seals_fetch(self._h, _p(mask), C.byref(t)), "ibft_seals_fetch")
        return mask_to_bool(mask, n), t

    def seals_run(self):
        """one more pass over the resident batch (launch + fetch in one C call) → (verdict bool[n], Tally)"""
        n = self._staged
        if getattr(self, "_run_mask", None) is None or len(self._run_mask) != ((n + 63) // 64 or 1):
            self._run_mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_seals_run(self._h, _p(self._run_mask), C.byref(t)), "ibft_seals_run")
        return mask_to_bool(self._run_mask, n), t

    def sign_seals(self, sk32, hash32):
        """ibft_sign_seals (simulators only): (sig65 u8[n,65], signer20 u8[n,20], ok bool[n]); leaves the batch staged"""
        sk = np.ascontiguousarray(sk32, dtype=np.uint8).reshape(-1, 32)
        hs = np.ascontiguousarray(hash32, dtype=np.uint8).reshape(-1, 32)
        n = len(sk)
        if len(hs) != n:
            raise ValueError("one hash per key")
        sig = np.zeros((n, 65), dtype=np.uint8)
        signer = np.zeros((n, 20), dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._L.ibft_sign_seals(self._h, _p(sk), _p(hs), n, _p(sig), _p(signer), _p(ok)), "ibft_sign_seals")
        self._staged = n
        return sig, signer, ok.astype(bool)

    def forget_proposal(self) -> None:
        """the next call that names a proposal hashes it again (a new height has a new proposal)"""
        self._chk(self._L.ibft_forget_proposal(self._h), "ibft_forget_proposal")

    def gather_batches(self) -> int:
        """batches whose (pinned) columns the device read itself in one gather launch"""
        g = C.c_uint32()
        self._chk(self._L.ibft_column_stats(self._h, C.byref(g)), "ibft_column_stats")
        return g.value

    def seals_device_ptrs(self):
        dm, dt, w = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._chk(self._L.ibft_seals_device_ptrs(self._h, C.byref(dm), C.byref(w), C.byref(dt)), "device_ptrs")
        return dm.value, w.value, dt.value

    def seals_export(self, d_mask_ptr: int | None, d_tally_ptr: int | None) -> None:
        """D2D copy of mask/tally into caller-owned device memory (torch data_ptr), then sync."""
        self._chk(self._L.ibft_seals_export(self._h, d_mask_ptr, d_tally_ptr), "ibft_seals_export")

    def seals_export_on(self, d_mask_ptr: int | None, d_tally_ptr: int | None, stream: int) -> None:
        """the same copies enqueued on the caller's stream (hipStream_t handle) behind a results-ready event:
        no host wait; the caller's collective overlaps with the next seals_launch"""
        self._chk(self._L.ibft_seals_export_on(self._h, d_mask_ptr, d_tally_ptr, stream), "ibft_seals_export_on")

    # ---- multi-GPU: this context is one rank of a sharded batch (include/ibftgpu.h, ibft_comm_*) ----
    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        uid = np.frombuffer(unique_id, dtype=np.uint8).copy()
        assert len(uid) == COMM_ID_BYTES
        self._chk(self._L.ibft_comm_init(self._h, _p(uid), rank, world), "ibft_comm_init")

    def comm_destroy(self) -> None:
        self._chk(self._L.ibft_comm_destroy(self._h), "ibft_comm_destroy")

    def seals_exchange(self, n_total: int) -> None:
        """enqueue pack + RCCL all-reduce + delivery of the merged result behind the last seals_launch"""
        self._chk(self._L.ibft_seals_exchange(self._h, n_total), "ibft_seals_exchange")
        self._xq = getattr(self, "_xq", [])
        self._xq.append(n_total)

    def seals_fetch_merged(self):
        """oldest outstanding exchange → (global verdict bool[n_total], merged Tally)"""
        n_total = self._xq.pop(0)
        mask = np.zeros((n_total + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_seals_fetch_merged(self._h, _p(mask), C.byref(t)), "ibft_seals_fetch_merged")
        return mask_to_bool(mask, n_total), t

    def set_kernel_timing(self, every_n: int) -> None:
        """HIP-event pair around the verdict kernels of every n-th staged pass (1 = all, 0 = none)"""
        self._chk(self._L.ibft_set_kernel_timing(self._h, every_n), "ibft_set_kernel_timing")

    def last_kernel_ms(self):
        ms, k = C.c_float(), C.c_uint32()
        self._chk(self._L.ibft_last_kernel_ms(self._h, C.byref(ms), C.byref(k)), "ibft_last_kernel_ms")
        return ms.value, k.value

    def cache_stats(self):
        """(validators with a built table, verdict passes that used the warm kernel, cold passes)."""
        t, w, c, g = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self._L.ibft_cache_stats(self._h, C.byref(t), C.byref(w), C.byref(c), C.byref(g)), "ibft_cache_stats")
        self.lanes_per_signature = g.value
        return t.value, w.value, c.value

    def cache_memory(self):
        """the device-wide key cache: (bytes held, slots in use, slots allocated, contexts sharing it)"""
        b, u, a, k = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self._L.ibft_cache_memory(self._h, C.byref(b), C.byref(u), C.byref(a), C.byref(k)), "ibft_cache_memory")
        return b.value, u.value, a.value, k.value

    def last_dispatch(self):
        """(cold lanes per signature, warm lanes per signature) of the last verdict pass."""
        a, b = C.c_uint32(), C.c_uint32()
        self._chk(self._L.ibft_last_dispatch(self._h, C.byref(a), C.byref(b)), "ibft_last_dispatch")
        return a.value, b.value

    def last_cold_table(self) -> int:
        """where the last lane / group cold kernel kept its window tables: 0 none ran, 1 LDS, 2 private + prefetch, 3 private"""
        t = C.c_uint32()
        self._chk(self._L.ibft_last_cold_table(self._h, C.byref(t)), "ibft_last_cold_table")
        return t.value

    def sync(self):
        self._chk(self._L.ibft_sync(self._h), "ibft_sync")

    def issue_probe(self):
        """ibft_issue_probe: (ns per aligned 8-byte VALU instruction per SIMD at one wavefront per SIMD, ms of the probe
        kernel) — 1.89 ns on a healthy MI355X"""
        ns, ms = C.c_float(0), C.c_float(0)
        self._chk(self._L.ibft_issue_probe(self._h, C.byref(ns), C.byref(ms)), "ibft_issue_probe")
        return float(ns.value), float(ms.value)


class DeviceGroup:
    """ibft_group: one process, several MI355X — rows sharded over the devices, one all-reduce inside the library merges
    verdict words and the ranks' distinct-sender bitmaps (what a Go Backend would call for N beyond one GPU).  A device
    listed several times gives several contexts on that device and the library's own sum kernel as the collective."""

    def __init__(self, devices, flags: int = 0, max_rows_total: int = 0):
        self._L = load_library()
        self._g = C.c_void_p()
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self._L.ibft_group_create(_p(dv), len(dv), flags, max_rows_total, C.byref(self._g))
        if rc != 0:
            self._g = C.c_void_p()
            raise GpuUnavailable(f"ibft_group_create: {self._L.ibft_strerror(rc).decode()} ({rc})")

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._L.ibft_group_destroy(self._g)
            self._g = C.c_void_p()

    __del__ = close

    def set_seal_digest(self, suffix: bytes | None) -> None:
        b = None if not suffix else np.frombuffer(bytes(suffix), dtype=np.uint8)
        self._chk(self._L.ibft_group_set_seal_digest(self._g, 0 if suffix is None else 1, _p(b), len(suffix or b"")),
                  "ibft_group_set_seal_digest")

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self._L.ibft_strerror(rc).decode()} ({rc})")

    @property
    def size(self) -> int:
        return self._L.ibft_group_size(self._g)

    def set_validators(self, height: int, addrs20, power) -> None:
        a = _u8(addrs20, (-1, 20)); p = np.ascontiguousarray(power, dtype=np.uint64)
        self._chk(self._L.ibft_group_set_validators(self._g, height, _p(a), _p(p), len(p)), "ibft_group_set_validators")

    def set_validators_u256(self, height: int, addrs20, powers) -> None:
        a = _u8(addrs20, (-1, 20)); p = powers_be32(powers)
        self._chk(self._L.ibft_group_set_validators_u256(self._g, height, _p(a), _p(p), len(p)),
                  "ibft_group_set_validators_u256")

    def is_valid_committed_seal(self, hash32, sig65, signer20, pre_flags=None):
        h = _u8(hash32, (-1, 32)); s = _u8(sig65, (-1, 65)); f = _u8(signer20, (-1, 20))
        n = len(s)
        pre = None if pre_flags is None else _u8(pre_flags)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_group_verify_seals(self._g, _p(h), _p(s), _p(f), _p(pre), n, _p(mask), C.byref(t)),
                  "ibft_group_verify_seals")
        return mask_to_bool(mask, n), t

    def is_valid_validator(self, payload: bytes, off, sig65, from20, pre_flags=None):
        pl = _bytes_col(payload)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        s = _u8(sig65, (-1, 65)); f = _u8(from20, (-1, 20))
        n = len(s)
        pre = None if pre_flags is None else _u8(pre_flags)
        mask = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_group_verify_senders(self._g, _p(pl), _p(off), _p(s), _p(f), _p(pre), n, _p(mask), C.byref(t)),
                  "ibft_group_verify_senders")
        return mask_to_bool(mask, n), t

    def verify_certificates_wire(self, wire, off, rows_cap: int = 65536, want_rows: bool = True):
        """ibft_group_verify_certificates_wire: the trees sharded by carrier over the group's devices, rows numbered as ONE
        call over all messages numbers them; same tuple as BatchVerifier.verify_certificates_wire"""
        wb = _bytes_col(wire)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = len(off) - 1
        cap = int(rows_cap)
        words = (cap + 63) // 64 or 1
        nodes = np.zeros(max(cap, 1), dtype=CERT_NODE)
        rows = np.zeros(max(cap, 1), dtype=WIRE_ROW) if want_rows else None
        cls = np.zeros(max(cap, 1), dtype=np.uint8)
        ms, mh, mself = (np.zeros(words, dtype=np.uint64) for _ in range(3))
        n_rows = C.c_size_t(0)
        self._chk(self._L.ibft_group_verify_certificates_wire(self._g, _p(wb), _p(off), n, cap, C.byref(n_rows), _p(nodes),
                                                              _p(rows) if want_rows else None, _p(cls), _p(ms), _p(mh), _p(mself)),
                  "ibft_group_verify_certificates_wire")
        k = int(n_rows.value)
        return (k, nodes[:k], rows[:k] if want_rows else None, cls[:k], mask_to_bool(ms, k), mask_to_bool(mh, k),
                mask_to_bool(mself, k))

    def verify_messages(self, payload: bytes, off, msg_sig65, from20, hash32, hash_len, seal65=None, sender_pre=None,
                        valid_pre=None, raw: bytes | None = None, round_: int = 0, digest32: bytes | None = None,
                        proposer: bytes | None = None):
        """a whole PREPARE / COMMIT set sharded by message → (sender bool[n], valid bool[n], merged Tally)"""
        pr = None if proposer is None else np.frombuffer(bytes(proposer), dtype=np.uint8)
        pl = _bytes_col(payload)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        s = _u8(msg_sig65, (-1, 65)); f = _u8(from20, (-1, 20)); h = _u8(hash32, (-1, 32)); hl = _u8(hash_len)
        n = len(s)
        sl = None if seal65 is None else _u8(seal65, (-1, 65))
        spre = None if sender_pre is None else _u8(sender_pre)
        vpre = None if valid_pre is None else _u8(valid_pre)
        rawb = None if raw is None else np.frombuffer(bytes(raw) or b"\0", dtype=np.uint8)
        dg = None if digest32 is None else np.frombuffer(bytes(digest32), dtype=np.uint8)
        ms = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        mv = np.zeros((n + 63) // 64 or 1, dtype=np.uint64)
        t = Tally()
        self._chk(self._L.ibft_group_verify_messages(self._g, _p(pl), _p(off), _p(s), _p(f), _p(h), _p(hl), _p(sl), _p(spre),
                                                     _p(vpre), n, _p(rawb), 0 if raw is None else len(raw), round_, _p(dg),
                                                     _p(pr), _p(ms), _p(mv), C.byref(t)), "ibft_group_verify_messages")
        return mask_to_bool(ms, n), mask_to_bool(mv, n), t

    @property
    def is_local(self) -> bool:
        """True: the collective is the library's own sum kernel (a device listed more than once), False: RCCL"""
        return self._L.ibft_group_is_local(self._g) == 1

    def last_tally_wide(self) -> TallyWide:
        t = TallyWide()
        self._chk(self._L.ibft_last_tally_wide(self._L.ibft_group_ctx(self._g, 0), C.byref(t)), "ibft_last_tally_wide")
        return t
