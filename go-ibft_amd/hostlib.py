"""ctypes mirror of include/ibft_host.h — the host-side store / quorum / hot-path callers.

Names follow the reference: ``Host.store_*`` = messages.Messages
(/root/reference/messages/messages.go), ``Host.vm_*`` = core.ValidatorManager,
``Host.add_message`` / ``handle_prepare`` / ``handle_commit`` = core/ibft.go:1101, :855, :931.
Messages cross the boundary as protobuf wire bytes.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

from .build import CSRC, HERE, build_lib, _mark, _stale

HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HOST_DIR, "libibft_host.so")
HOST_SRCS = ["proto.cpp", "messages.cpp", "backend.cpp", "host_capi.cpp"]
HOST_DEPS = HOST_SRCS + ["proto.hpp", "messages.hpp", "backend.hpp"]


def build_host(force: bool = False) -> str:
    build_lib()
    deps = [os.path.join("..", "host", d) for d in HOST_DEPS] + [os.path.join("..", "..", "include", "ibft_host.h"),
                                                                  os.path.join("..", "..", "include", "ibftgpu.h")]
    if force or _stale(HOST_LIB, deps):  # digest of the sources, not modification times (build.py)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB, *HOST_SRCS,
                               "-L" + CSRC, "-libftgpu", "-Wl,-rpath,$ORIGIN/../csrc"], cwd=HOST_DIR)
        _mark(HOST_LIB, deps)
    return HOST_LIB


SANITIZERS = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "tsan": ["-fsanitize=thread"]}


def build_host_sanitized(kind: str, out_dir: str) -> str:
    """TEST-ONLY: the mirror compiled with AddressSanitizer + UBSan ("asan") or ThreadSanitizer ("tsan") into out_dir
    (tests/test_host_sanitize.py loads it through IBFT_HOST_LIB under LD_PRELOAD of the sanitizer runtime)"""
    build_lib()
    out = os.path.join(out_dir, f"libibft_host_{kind}.so")
    subprocess.check_call(["g++", "-O1", "-g", "-fno-omit-frame-pointer", *SANITIZERS[kind], "-std=c++17", "-fPIC", "-shared",
                           "-o", out, *HOST_SRCS, "-L" + CSRC, "-libftgpu", "-Wl,-rpath," + CSRC], cwd=HOST_DIR)
    return out


class Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("count", C.c_size_t)]


PROP_HASH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint64, C.c_int,
                           C.POINTER(C.c_uint8), C.c_size_t)
SEAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_int,
                      C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t)
VALIDATOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
MSG_PRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
RCC_PRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_size_t)


PROPOSER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint64, C.c_uint64)
VALID_PROPOSAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)


class QueueStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("pushed", "ingested", "stored", "rejected", "undecodable", "batches",
                                           "device_calls", "cache_hits", "max_batch_rows")] + [("signals", C.c_uint64 * 4),
                                                                                              ("ingest_us", C.c_uint64), ("device_us", C.c_uint64)]


SIGNAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64)


class VerifierCB(C.Structure):
    _fields_ = [("is_valid_proposal_hash", PROP_HASH_FN), ("is_valid_committed_seal", SEAL_FN),
                ("is_valid_validator", VALIDATOR_FN), ("user", C.c_void_p),
                ("is_proposer", PROPOSER_FN), ("is_valid_proposal", VALID_PROPOSAL_FN)]


_lib = None


def retain_heap(nbytes: int = 256 << 20) -> bool:
    """Process-wide (include/ibft_host.h: ibft_host_retain_heap): keep freed C heap instead of returning it to the kernel."""
    return lib().ibft_host_retain_heap(nbytes) == 0


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(os.environ.get("IBFT_HOST_LIB") or build_host())   # IBFT_HOST_LIB: a sanitizer build (tests)
        vp, bp = C.c_void_p, C.POINTER(Buf)
        L.ibft_host_new.restype = vp
        L.ibft_host_free.argtypes = [vp]
        L.ibft_host_buf_free.argtypes = [bp]
        L.ibft_host_payload_no_sig.argtypes = [C.c_char_p, C.c_size_t, bp]
        L.ibft_host_reencode.argtypes = [C.c_char_p, C.c_size_t, bp]
        L.ibft_host_store_add.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ibft_host_store_num.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32]; L.ibft_host_store_num.restype = C.c_size_t
        L.ibft_host_store_prune.argtypes = [vp, C.c_uint64]
        L.ibft_host_store_get_valid.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, MSG_PRED, vp, bp]
        L.ibft_host_store_get_extended_rcc.argtypes = [vp, C.c_uint64, MSG_PRED, RCC_PRED, vp, bp]
        L.ibft_host_store_get_most_rc.argtypes = [vp, C.c_uint64, C.c_uint64, bp]
        L.ibft_host_has_unique_senders.argtypes = [C.c_char_p, C.c_size_t]
        L.ibft_host_are_valid_pc_messages.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64]
        L.ibft_host_extract_committed_seals.argtypes = [C.c_char_p, C.c_size_t, bp]
        L.ibft_host_vm_init.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t]
        L.ibft_host_vm_has_quorum.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ibft_host_vm_has_prepare_quorum.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.ibft_host_vm_quorum.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ibft_host_set_state.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_char_p, C.c_size_t]
        L.ibft_host_set_verifier.argtypes = [vp, C.POINTER(VerifierCB)]
        L.ibft_host_attach_gpu.argtypes = [vp, vp]
        L.ibft_host_verify_senders_wire.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.c_int, vp,
                                                    C.POINTER(C.c_double), C.POINTER(C.c_size_t)]
        L.ibft_host_use_batch.argtypes = [vp, C.c_int]
        L.ibft_host_add_message.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ibft_host_enable_quorum_index.argtypes = [vp]
        L.ibft_host_add_message_fast.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ibft_host_add_messages_batch.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.ibft_host_set_id.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ibft_host_set_round_robin_proposer.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_int]
        L.ibft_host_valid_pc.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64]
        L.ibft_host_proposal_matches_certificate.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.ibft_host_validate_proposal0.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64]
        L.ibft_host_validate_proposal.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64]
        L.ibft_host_last_cert_batch.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ibft_host_handle_prepare.argtypes = [vp, C.c_uint64, C.c_uint64, bp]
        L.ibft_host_handle_commit.argtypes = [vp, C.c_uint64, C.c_uint64, bp]
        L.ibft_host_handle_round_change.argtypes = [vp, C.c_uint64, C.c_uint64, bp]
        L.ibft_host_ingest_wire.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ibft_host_ingest_flat.argtypes = [vp, vp, vp, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ibft_host_queue_start.argtypes = [vp, C.c_size_t, C.c_uint32]
        L.ibft_host_queue_push.argtypes = [vp, vp, vp, C.c_size_t]
        L.ibft_host_queue_drain.argtypes = [vp, C.POINTER(QueueStats)]
        L.ibft_host_queue_set_caps.argtypes = [vp, C.c_size_t, C.c_size_t]; L.ibft_host_queue_set_caps.restype = None
        L.ibft_host_queue_backpressure_waits.argtypes = [vp]; L.ibft_host_queue_backpressure_waits.restype = C.c_uint64
        L.ibft_host_queue_stop.argtypes = [vp]; L.ibft_host_queue_stop.restype = None
        L.ibft_host_queue_on_signal.argtypes = [vp, SIGNAL_FN, vp]; L.ibft_host_queue_on_signal.restype = None
        L.ibft_host_seen_entries.argtypes = [vp]; L.ibft_host_seen_entries.restype = C.c_size_t
        L.ibft_host_set_seen_caps.argtypes = [vp, C.c_size_t, C.c_size_t]; L.ibft_host_set_seen_caps.restype = None
        L.ibft_host_use_loop_batch.argtypes = [vp, C.c_int]
        L.ibft_host_loop_batch_calls.argtypes = [vp]; L.ibft_host_loop_batch_calls.restype = C.c_size_t
        L.ibft_host_fallbacks.argtypes = [vp]; L.ibft_host_fallbacks.restype = C.c_size_t
        L.ibft_host_set_min_device_rows.argtypes = [vp, C.c_size_t]; L.ibft_host_set_min_device_rows.restype = None
        L.ibft_host_declined_batches.argtypes = [vp]; L.ibft_host_declined_batches.restype = C.c_size_t
        L.ibft_host_use_sets.argtypes = [vp, C.c_int]; L.ibft_host_use_sets.restype = None
        for nm in ('ibft_host_last_set_rows', 'ibft_host_closure_hits', 'ibft_host_loop_batch_set_calls',
                   'ibft_host_loop_batch_cert_calls'):
            getattr(L, nm).argtypes = [vp]; getattr(L, nm).restype = C.c_size_t
        L.ibft_host_use_certs.argtypes = [vp, C.c_int]; L.ibft_host_use_certs.restype = None
        L.ibft_host_use_rows.argtypes = [vp, C.c_int]; L.ibft_host_use_rows.restype = None
        L.ibft_host_use_device_quorum.argtypes = [vp, C.c_int]; L.ibft_host_use_device_quorum.restype = None
        L.ibft_host_device_quorum_stats.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ibft_host_device_quorum_stats.restype = None
        L.ibft_host_retain_heap.argtypes = [C.c_size_t]; L.ibft_host_retain_heap.restype = C.c_int
        L.ibft_host_use_rc_rows.argtypes = [vp, C.c_int]; L.ibft_host_use_rc_rows.restype = None
        L.ibft_host_cert_roots_first.argtypes = [vp, C.c_int]; L.ibft_host_cert_roots_first.restype = None
        L.ibft_host_roots_first_calls.argtypes = [vp]; L.ibft_host_roots_first_calls.restype = C.c_size_t
        L.ibft_host_last_ingest_device_ms.argtypes = [vp]; L.ibft_host_last_ingest_device_ms.restype = C.c_double
        L.ibft_host_rc_from_rows.argtypes = [vp]; L.ibft_host_rc_from_rows.restype = C.c_size_t
        L.ibft_host_pp_from_rows.argtypes = [vp]; L.ibft_host_pp_from_rows.restype = C.c_size_t
        L.ibft_host_repacked_bytes.argtypes = [vp]; L.ibft_host_repacked_bytes.restype = C.c_size_t
        L.ibft_host_set_repack_min_bytes.argtypes = [vp, C.c_size_t]; L.ibft_host_set_repack_min_bytes.restype = None
        L.ibft_host_rows_kept.argtypes = [vp]; L.ibft_host_rows_kept.restype = C.c_size_t
        L.ibft_host_lean_stats.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32] + [C.POINTER(C.c_size_t)] * 3
        L.ibft_host_lean_stats.restype = None
        L.ibft_host_cert_stats.argtypes = [vp] + [C.POINTER(C.c_size_t)] * 3; L.ibft_host_cert_stats.restype = None
        L.ibft_host_handle_preprepare.argtypes = [vp, C.c_uint64, C.c_uint64, bp]
        _lib = L
    return _lib


def pack(items) -> bytes:
    return b"".join(struct.pack("<I", len(x)) + bytes(x) for x in items)


def unpack(raw: bytes) -> list[bytes]:
    out, pos = [], 0
    while pos < len(raw):
        (ln,) = struct.unpack_from("<I", raw, pos)
        out.append(raw[pos + 4: pos + 4 + ln])
        pos += 4 + ln
    return out


def _take(buf: Buf) -> bytes:
    raw = C.string_at(buf.data, buf.len) if buf.len else b""
    lib().ibft_host_buf_free(C.byref(buf))
    return raw


def unpack_seals(raw: bytes):
    out, pos = [], 0
    while pos < len(raw):
        present = raw[pos]; pos += 1
        (l1,) = struct.unpack_from("<I", raw, pos); signer = raw[pos + 4: pos + 4 + l1]; pos += 4 + l1
        (l2,) = struct.unpack_from("<I", raw, pos); sig = raw[pos + 4: pos + 4 + l2]; pos += 4 + l2
        out.append((signer, sig) if present else None)
    return out


def payload_no_sig(wire: bytes) -> bytes | None:
    b = Buf()
    return _take(b) if lib().ibft_host_payload_no_sig(wire, len(wire), C.byref(b)) == 0 else None


def reencode(wire: bytes) -> bytes | None:
    b = Buf()
    return _take(b) if lib().ibft_host_reencode(wire, len(wire), C.byref(b)) == 0 else None


def verify_senders_wire(batch_verifier, wire: bytes, off, stock: bool = False):
    """§8f rank 3 through the host mirror: device wire walk with the stock route for the rows it flags
    (stock=True: every row takes the stock route).  Returns (verdict bool[n], host_ms, host_rows)."""
    import numpy as np
    off = np.ascontiguousarray(off, dtype=np.uint32)
    n = len(off) - 1
    verdict = np.zeros(max(n, 1), dtype=np.uint8)
    ms, rows = C.c_double(0.0), C.c_size_t(0)
    rc = lib().ibft_host_verify_senders_wire(batch_verifier._h, bytes(wire) or b"\0", off.ctypes.data_as(C.c_void_p), n,
                                             1 if stock else 0, verdict.ctypes.data_as(C.c_void_p), C.byref(ms),
                                             C.byref(rows))
    if rc != 0:
        raise RuntimeError(f"ibft_host_verify_senders_wire: {rc}")
    return verdict[:n].astype(bool), ms.value, rows.value


def cert_routes(batch_verifier, wire: bytes, off, route: int, rows_cap: int = 0):
    """measurement aid: every sender check of a batch of certificate trees; route 0 = from the bytes
    (ibft_verify_certificates_wire), 1 = host decode + PayloadNoSig re-marshal + flatten + ibft_verify_senders.
    Returns (rows, valid, host_ms, total_ms)."""
    import numpy as np
    off = np.ascontiguousarray(off, dtype=np.uint32)
    L = lib()
    L.ibft_host_cert_routes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if isinstance(wire, np.ndarray):  # e.g. a page-locked buffer (verifier.pinned_copy): handed over as it is
        wire_ptr, keep = wire.ctypes.data_as(C.c_void_p), wire
    else:
        keep = np.frombuffer(bytes(wire) or b"\0", dtype=np.uint8)
        wire_ptr = keep.ctypes.data_as(C.c_void_p)
    rows, valid, hms, tms = C.c_size_t(0), C.c_size_t(0), C.c_double(0.0), C.c_double(0.0)
    rc = L.ibft_host_cert_routes(batch_verifier._h, wire_ptr, off.ctypes.data_as(C.c_void_p), len(off) - 1, route, rows_cap,
                                 C.byref(rows), C.byref(valid), C.byref(hms), C.byref(tms))
    if rc != 0:
        raise RuntimeError(f"ibft_host_cert_routes: {rc}")
    return rows.value, valid.value, hms.value, tms.value


def has_unique_senders(msgs) -> bool:
    p = pack(msgs)
    return lib().ibft_host_has_unique_senders(p, len(p)) == 1


def are_valid_pc_messages(msgs, height: int, round_limit: int) -> bool:
    p = pack(msgs)
    return lib().ibft_host_are_valid_pc_messages(p, len(p), height, round_limit) == 1


def extract_committed_seals(msgs):
    p = pack(msgs)
    b = Buf()
    rc = lib().ibft_host_extract_committed_seals(p, len(p), C.byref(b))
    if rc != 0:
        return None  # ErrWrongCommitMessageType
    return unpack_seals(_take(b))


class Host:
    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.ibft_host_new())
        self._keep = []

    def close(self):
        if self.h:
            self.L.ibft_host_free(self.h)
            self.h = None

    __del__ = close

    # --- messages.Messages
    def store_add(self, wire: bytes) -> int:
        return self.L.ibft_host_store_add(self.h, wire, len(wire))

    def store_num(self, height, round_, type_) -> int:
        return self.L.ibft_host_store_num(self.h, height, round_, type_)

    def store_prune(self, height):
        self.L.ibft_host_store_prune(self.h, height)

    def store_get_valid(self, height, round_, type_, pred=None) -> list[bytes]:
        cb = MSG_PRED(lambda u, p, n: int(bool(pred(C.string_at(p, n))))) if pred else MSG_PRED()
        b = Buf()
        self.L.ibft_host_store_get_valid(self.h, height, round_, type_, cb, None, C.byref(b))
        return unpack(_take(b))

    def store_get_extended_rcc(self, height, pred, rcc_pred) -> list[bytes]:
        cb = MSG_PRED(lambda u, p, n: int(bool(pred(C.string_at(p, n)))))
        rcb = RCC_PRED(lambda u, r, n: int(bool(rcc_pred(r, n))))
        b = Buf()
        self.L.ibft_host_store_get_extended_rcc(self.h, height, cb, rcb, None, C.byref(b))
        return unpack(_take(b))

    def store_get_most_rc(self, min_round, height) -> list[bytes]:
        b = Buf()
        self.L.ibft_host_store_get_most_rc(self.h, min_round, height, C.byref(b))
        return unpack(_take(b))

    # --- core.ValidatorManager
    def vm_init(self, powers: dict) -> bool:
        addrs = list(powers.keys())
        p = pack(addrs)
        arr = (C.c_uint64 * max(len(addrs), 1))(*[powers[a] for a in addrs])
        return self.L.ibft_host_vm_init(self.h, p, len(p), arr, len(addrs)) == 0

    def vm_has_quorum(self, senders) -> bool:
        p = pack(list(senders))
        return self.L.ibft_host_vm_has_quorum(self.h, p, len(p)) == 1

    def vm_has_prepare_quorum(self, proposal_wire, msgs) -> bool:
        p = pack(msgs)
        return self.L.ibft_host_vm_has_prepare_quorum(self.h, proposal_wire, len(proposal_wire or b""), p, len(p)) == 1

    def vm_quorum(self) -> int:
        lo, hi = C.c_uint64(), C.c_uint64()
        self.L.ibft_host_vm_quorum(self.h, C.byref(lo), C.byref(hi))
        return lo.value | (hi.value << 64)

    # --- state + verifier
    def set_state(self, height, round_, proposal_wire: bytes | None):
        return self.L.ibft_host_set_state(self.h, height, round_, proposal_wire, len(proposal_wire or b""))

    def set_verifier(self, is_valid_proposal_hash=None, is_valid_committed_seal=None, is_valid_validator=None,
                     is_proposer=None, is_valid_proposal=None):
        """Callbacks receive Python values; None arguments mirror Go nils."""
        def ph(u, hp, raw, rl, rnd, hh, hsh, hl):
            prop = (C.string_at(raw, rl) if rl else b"", rnd) if hp else None
            return int(bool(is_valid_proposal_hash(prop, (C.string_at(hsh, hl) if hl else b"") if hh else None)))

        def sl(u, hh, hsh, hl, hs, sg, sgl, sig, sigl):
            seal = ((C.string_at(sg, sgl) if sgl else b""), (C.string_at(sig, sigl) if sigl else b"")) if hs else None
            return int(bool(is_valid_committed_seal((C.string_at(hsh, hl) if hl else b"") if hh else None, seal)))

        def vv(u, p, n):
            return int(bool(is_valid_validator(C.string_at(p, n))))

        def ip(u, p, n, hh, rr):
            return int(bool(is_proposer(C.string_at(p, n) if n else b"", hh, rr)))

        def vpr(u, p, n):
            return int(bool(is_valid_proposal(C.string_at(p, n) if n else b"")))

        cb = VerifierCB(PROP_HASH_FN(ph) if is_valid_proposal_hash else PROP_HASH_FN(),
                        SEAL_FN(sl) if is_valid_committed_seal else SEAL_FN(),
                        VALIDATOR_FN(vv) if is_valid_validator else VALIDATOR_FN(), None,
                        PROPOSER_FN(ip) if is_proposer else PROPOSER_FN(),
                        VALID_PROPOSAL_FN(vpr) if is_valid_proposal else VALID_PROPOSAL_FN())
        self._keep.append(cb)
        self.L.ibft_host_set_verifier(self.h, C.byref(cb))

    def attach_gpu(self, batch_verifier):
        """batch_verifier: go_ibft_amd.verifier.BatchVerifier (its ibft_ctx is borrowed)."""
        self._keep.append(batch_verifier)
        self.L.ibft_host_attach_gpu(self.h, batch_verifier._h)

    def use_batch(self, on: bool):
        self.L.ibft_host_use_batch(self.h, int(on))

    # --- core/ibft.go hot-path callers
    def add_message(self, wire: bytes) -> int:
        return self.L.ibft_host_add_message(self.h, wire, len(wire))

    def enable_quorum_index(self):
        self.L.ibft_host_enable_quorum_index(self.h)

    def add_message_fast(self, wire: bytes) -> int:
        return self.L.ibft_host_add_message_fast(self.h, wire, len(wire))

    def add_messages_batch(self, wires) -> list[int]:
        p = pack(wires)
        res = C.create_string_buffer(len(wires))
        rc = self.L.ibft_host_add_messages_batch(self.h, p, len(p), res, len(wires))
        if rc != 0:
            raise RuntimeError(f"ibft_host_add_messages_batch rc={rc}")
        return list(res.raw)

    # --- certificate checks (core/ibft.go: validPC, proposalMatchesCertificate, validateProposal*)
    def set_id(self, node_id: bytes):
        self.L.ibft_host_set_id(self.h, node_id, len(node_id))

    def set_round_robin_proposer(self, addrs, use_height: bool = True):
        """native IsProposer: addrs[(height·use_height + round) mod n] (core/helpers_test.go:214-225)"""
        p = pack(list(addrs))
        assert self.L.ibft_host_set_round_robin_proposer(self.h, p, len(p), 1 if use_height else 0) == 0

    def valid_pc(self, pc_wire, round_limit: int, height: int) -> bool:
        return self.L.ibft_host_valid_pc(self.h, pc_wire, len(pc_wire or b""), round_limit, height) == 1

    def proposal_matches_certificate(self, proposal_wire, pc_wire) -> bool:
        return self.L.ibft_host_proposal_matches_certificate(self.h, proposal_wire, len(proposal_wire or b""),
                                                             pc_wire, len(pc_wire or b"")) == 1

    def validate_proposal0(self, msg_wire: bytes, height: int, round_: int) -> bool:
        return self.L.ibft_host_validate_proposal0(self.h, msg_wire, len(msg_wire), height, round_) == 1

    def validate_proposal(self, msg_wire: bytes, height: int, round_: int) -> bool:
        return self.L.ibft_host_validate_proposal(self.h, msg_wire, len(msg_wire), height, round_) == 1

    def last_cert_batch(self):
        a, b = C.c_size_t(), C.c_size_t()
        self.L.ibft_host_last_cert_batch(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def handle_prepare(self, height, round_):
        b = Buf()
        q = self.L.ibft_host_handle_prepare(self.h, height, round_, C.byref(b))
        return bool(q), unpack(_take(b))

    def handle_commit(self, height, round_):
        b = Buf()
        q = self.L.ibft_host_handle_commit(self.h, height, round_, C.byref(b))
        return bool(q), unpack_seals(_take(b))

    def handle_round_change(self, height, round_):
        """handleRoundChangeMessage → the extended RCC's messages (wire), [] = nil"""
        b = Buf()
        self.L.ibft_host_handle_round_change(self.h, height, round_, C.byref(b))
        return unpack(_take(b))

    def ingest_wire(self, wires):
        """receive side: (results per message: -1 undecodable / 0 / 1 / 2, device rows, cache hits, device calls)"""
        p = pack(wires)
        res = C.create_string_buffer(max(len(wires), 1))
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        rc = self.L.ibft_host_ingest_wire(self.h, p, len(p), res, len(wires), C.byref(a), C.byref(b), C.byref(c))
        if rc != 0:
            raise RuntimeError(f"ibft_host_ingest_wire rc={rc}")
        return [x - 256 if x > 127 else x for x in res.raw[:len(wires)]], a.value, b.value, c.value

    def ingest_packed(self, packed: bytes, n: int) -> bytes:
        """ingest_wire for a micro-batch that is already packed (repeated {u32 length, bytes}) → the n result bytes as
        they are (0 / 1 / 2, 0xFF = undecodable): no per-message Python work"""
        res = C.create_string_buffer(max(n, 1))
        rc = self.L.ibft_host_ingest_wire(self.h, packed, len(packed), res, n, None, None, None)
        if rc != 0:
            raise RuntimeError(f"ibft_host_ingest_wire rc={rc}")
        return res.raw[:n]

    def ingest_flat(self, wire, off, want_stats: bool = False):
        """ibft_host_ingest_flat: wire = numpy uint8 (rows back to back), off = numpy uint32 (n + 1) → the n result bytes
        (and (device rows, cache hits, device calls) with want_stats)"""
        n = len(off) - 1
        res = C.create_string_buffer(max(n, 1))
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        rc = self.L.ibft_host_ingest_flat(self.h, wire.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), n, res,
                                          C.byref(a), C.byref(b), C.byref(c))
        if rc != 0:
            raise RuntimeError(f"ibft_host_ingest_flat rc={rc}")
        return (res.raw[:n], (a.value, b.value, c.value)) if want_stats else res.raw[:n]

    # --- the receive-side queue (adaptive batching)
    def queue_start(self, max_rows: int = 65536, linger_us: int = 0):
        assert self.L.ibft_host_queue_start(self.h, max_rows, linger_us) == 0

    def queue_push(self, wire, off):
        """rows back to back (numpy uint8) + n + 1 offsets (numpy uint32): copied, returns at once"""
        if self.L.ibft_host_queue_push(self.h, wire.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(off) - 1) != 0:
            raise RuntimeError("ibft_host_queue_push")

    def queue_set_caps(self, max_pending_bytes: int = 0, max_pending_rows: int = 0):
        self.L.ibft_host_queue_set_caps(self.h, max_pending_bytes, max_pending_rows)

    def queue_try_push(self, wire, off) -> int:
        """ibft_host_queue_push's own return code (0 queued, −2 larger than the queue's caps)"""
        return self.L.ibft_host_queue_push(self.h, wire.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(off) - 1)

    @property
    def queue_backpressure_waits(self) -> int:
        return int(self.L.ibft_host_queue_backpressure_waits(self.h))

    def queue_drain(self) -> QueueStats:
        st = QueueStats()
        if self.L.ibft_host_queue_drain(self.h, C.byref(st)) != 0:
            raise RuntimeError("ibft_host_queue_drain")
        return st

    def queue_on_signal(self, fn):
        """fn(type, height, round) from the queue's worker thread — the SignalEvent of core/ibft.go:1119"""
        cb = SIGNAL_FN(lambda u, t, hh, rr: fn(t, hh, rr))
        self._keep.append(cb)
        self.L.ibft_host_queue_on_signal(self.h, cb, None)

    def queue_stop(self):
        self.L.ibft_host_queue_stop(self.h)

    def seen_entries(self) -> int:
        return self.L.ibft_host_seen_entries(self.h)

    def set_seen_caps(self, stored_cap: int, rejected_cap: int):
        self.L.ibft_host_set_seen_caps(self.h, stored_cap, rejected_cap)

    def handle_commit_raw(self, height, round_):
        """handleCommit → (quorum, the packed seal list as bytes); unpack_seals() decodes it"""
        b = Buf()
        q = self.L.ibft_host_handle_commit(self.h, height, round_, C.byref(b))
        return bool(q), _take(b)

    def handle_prepare_quiet(self, height, round_) -> bool:
        """handlePrepare without marshalling the prepared messages back (a Go caller holds the objects)"""
        return self.L.ibft_host_handle_prepare(self.h, height, round_, None) == 1

    def handle_round_change_count(self, height, round_) -> int:
        """handleRoundChangeMessage → 1 when an extended RCC exists (0 = nil), nothing marshalled back"""
        return self.L.ibft_host_handle_round_change(self.h, height, round_, None)

    def use_certs(self, on: bool):
        self.L.ibft_host_use_certs(self.h, 1 if on else 0)

    def last_ingest_device_ms(self) -> float:
        return float(self.L.ibft_host_last_ingest_device_ms(self.h))

    def cert_roots_first(self, mode: int):
        """0 = never, 1 = always, 2 = while forged carriers keep arriving (default): authenticate carriers before expanding trees"""
        self.L.ibft_host_cert_roots_first(self.h, mode)

    @property
    def roots_first_calls(self) -> int:
        return int(self.L.ibft_host_roots_first_calls(self.h))

    def use_rc_rows(self, on: bool):
        """Judge a ROUND_CHANGE message's certificate from the backend's rows on arrival (default) or by the object walk."""
        self.L.ibft_host_use_rc_rows(self.h, 1 if on else 0)

    @property
    def rc_from_rows(self) -> int:
        return int(self.L.ibft_host_rc_from_rows(self.h))

    @property
    def pp_from_rows(self) -> int:
        return int(self.L.ibft_host_pp_from_rows(self.h))

    def set_repack_min_bytes(self, nbytes: int):
        self.L.ibft_host_set_repack_min_bytes(self.h, nbytes)

    @property
    def repacked_bytes(self) -> int:
        return int(self.L.ibft_host_repacked_bytes(self.h))

    def use_device_quorum(self, on: bool):
        """handlePrepare / handleCommit take the quorum decision from the batch backend (ibft_tally_prepare / ibft_tally)"""
        self.L.ibft_host_use_device_quorum(self.h, 1 if on else 0)

    def device_quorum_stats(self):
        """(decisions the backend took, decisions on which the mirror's quorum index disagreed — must be 0)"""
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.L.ibft_host_device_quorum_stats(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def use_rows(self, on: bool):
        """Keep the PREPARE / COMMIT messages a batch backend judged from their bytes as rows (default) or as objects."""
        self.L.ibft_host_use_rows(self.h, 1 if on else 0)

    def lean_stats(self, height, round_, type_):
        """(live rows, row slots, batch buffers still referenced) of one view's rows"""
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self.L.ibft_host_lean_stats(self.h, height, round_, type_, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    @property
    def rows_kept(self) -> int:
        return int(self.L.ibft_host_rows_kept(self.h))

    def cert_stats(self):
        """(certificate calls made by ingest, rows they judged, sender verdicts the last certificate walk took from the tables)"""
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.L.ibft_host_cert_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def loop_batch_cert_calls(self) -> int:
        return self.L.ibft_host_loop_batch_cert_calls(self.h)

    def handle_preprepare(self, height, round_):
        """handlePrePrepare → the accepted PREPREPARE's wire bytes, or None"""
        b = Buf()
        q = self.L.ibft_host_handle_preprepare(self.h, height, round_, C.byref(b))
        out = unpack(_take(b))
        return out[0] if q and out else None

    def handle_preprepare_quiet(self, height, round_) -> bool:
        """handlePrePrepare → accepted or not (the message itself stays in the mirror)"""
        return bool(self.L.ibft_host_handle_preprepare(self.h, height, round_, None))

    def use_sets(self, on: bool):
        self.L.ibft_host_use_sets(self.h, 1 if on else 0)

    def last_set_rows(self) -> int:
        return self.L.ibft_host_last_set_rows(self.h)

    def closure_hits(self) -> int:
        return self.L.ibft_host_closure_hits(self.h)

    def loop_batch_set_calls(self) -> int:
        return self.L.ibft_host_loop_batch_set_calls(self.h)

    def use_loop_batch(self, fail_mask: int = 0):
        self.L.ibft_host_use_loop_batch(self.h, fail_mask)

    def loop_batch_calls(self) -> int:
        return self.L.ibft_host_loop_batch_calls(self.h)

    def fallbacks(self) -> int:
        return self.L.ibft_host_fallbacks(self.h)

    def set_min_device_rows(self, rows: int):
        """SURVEY §5 "min batch for GPU": batches below `rows` rows are declined by the batch backend, the stock closures run"""
        self.L.ibft_host_set_min_device_rows(self.h, rows)

    def declined_batches(self) -> int:
        return self.L.ibft_host_declined_batches(self.h)
