"""Synthetic consensus rounds signed ON THE DEVICE (ibft_sign_seals, include/ibftgpu.h §f4) — for load generators,
simulators and bench.py.  A process that plays n validators needs n committed seals per height
(Backend.BuildCommitMessage, /root/reference/core/backend.go:12-34); the batch signer produces 65 536 of them in under a
millisecond, so a benchmark rank can build the WHOLE validator table of a sharded round on its own GPU in milliseconds
instead of signing with host code for minutes.

Keys are deterministic in (seed, validator index): sk = SplitMix64 stream of the seed, 32 bytes per validator, top bit
cleared (< 2^255 < n) and forced non-zero.  The Byzantine mix follows SURVEY.md §8d: every fifth row (by a SplitMix64
stream of seed ^ 0xB12) is corrupted, the kind cycling over the twelve kinds below; `expect` is what every verifier
must answer by construction (an honest row is valid, a corrupted one is not), and bench.py / the tests additionally
check the rows against the CPU oracle."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

CORRUPTIONS = ["random65", "non_validator", "other_hash", "stolen_seal", "r_zero", "s_zero", "r_ge_n", "s_ge_n", "v_two",
               "len64", "wrong_hash_field", "nil_payload"]
ROW_NIL, ROW_BADLEN, ROW_HASH_BAD = 1, 2, 4
_N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
_M64 = (1 << 64) - 1


def _splitmix(seed: int, count: int) -> np.ndarray:
    """count 64-bit outputs of SplitMix64(seed), vectorised"""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def secret_keys(seed: int, n: int, salt: int = 0) -> np.ndarray:
    sk = _splitmix(seed * 0x10001 + salt, 4 * n).view(np.uint8).reshape(n, 32).copy()
    sk[:, 0] &= 0x7F
    sk[:, 31] |= 1
    return sk


@dataclass
class DeviceRound:
    n: int
    raw: bytes
    round: int
    proposal_hash: bytes
    addrs: np.ndarray      # n × 20 (validator table, row i = validator i)
    power: np.ndarray      # n u64
    hash32: np.ndarray     # n × 32
    seal65: np.ndarray     # n × 65
    signer20: np.ndarray   # n × 20
    pre_flags: np.ndarray | None
    expect: np.ndarray     # n bool: the verdict every row must get
    kinds: list


def make_round(bv, n: int, seed: int = 1, *, byzantine: bool = False, weighted: bool = False, raw_len: int = 1024,
               round_: int = 0) -> DeviceRound:
    """n validators, one COMMIT seal each over keccak256(raw ‖ BE64(round)) — keys, proposal hash and signatures all
    computed by `bv` (a BatchVerifier: ibft_proposal_hash, ibft_sign_seals).  Leaves bv's staged batch undefined."""
    raw = _splitmix(seed, (raw_len + 7) // 8).tobytes()[:raw_len]
    H = bv.proposal_hash(raw, round_)
    sk = secret_keys(seed, n)
    hcol = np.tile(np.frombuffer(H, dtype=np.uint8), (n, 1))
    seal, addrs, ok = _sign(bv, sk, hcol)
    assert ok.all()
    power = (1 + (_splitmix(seed ^ 0x57A4E, n) % np.uint64(16))).astype(np.uint64) if weighted else np.ones(n, dtype=np.uint64)
    signer = addrs.copy()
    hash32 = hcol.copy()
    pre = np.zeros(n, dtype=np.uint8)
    expect = np.ones(n, dtype=bool)
    kinds = [""] * n
    if byzantine:
        bad = np.flatnonzero(_splitmix(seed ^ 0xB12, n) % np.uint64(5) == 0)
        H2 = bv.proposal_hash(b"other" + raw, round_)
        h2col = np.tile(np.frombuffer(H2, dtype=np.uint8), (len(bad), 1))
        outsider, _, _ = _sign(bv, secret_keys(seed, len(bad), salt=0x5EED), hcol[: len(bad)])   # keys of no validator
        other, _, _ = _sign(bv, sk[bad], h2col) if len(bad) else (np.zeros((0, 65), np.uint8), None, None)
        rnd = _splitmix(seed ^ 0xABCD, 9 * len(bad)).view(np.uint8).reshape(len(bad), 72)
        for j, i in enumerate(bad):
            kind = CORRUPTIONS[j % len(CORRUPTIONS)]
            kinds[i] = kind
            expect[i] = False
            if kind == "random65":
                seal[i, :64] = rnd[j, :64]
                seal[i, 64] = rnd[j, 64] & 1
                # a random 64-byte string is a valid signature of SOME key with probability ≈ 1/2, never of this validator's
            elif kind == "non_validator":
                seal[i] = outsider[j]
            elif kind == "other_hash":
                seal[i] = other[j]
            elif kind == "stolen_seal":
                seal[i] = seal[(i + 1) % n] if kinds[(i + 1) % n] == "" else outsider[j]
            elif kind == "r_zero":
                seal[i, :32] = 0
            elif kind == "s_zero":
                seal[i, 32:64] = 0
            elif kind == "r_ge_n":
                seal[i, :32] = np.frombuffer(_N_ORDER.to_bytes(32, "big"), dtype=np.uint8)
            elif kind == "s_ge_n":
                seal[i, 32:64] = np.frombuffer((_N_ORDER + 1).to_bytes(32, "big"), dtype=np.uint8)
            elif kind == "v_two":
                seal[i, 64] = 2
            elif kind == "len64":
                seal[i, 64] = 0
                pre[i] |= ROW_BADLEN
            elif kind == "wrong_hash_field":
                hash32[i] = h2col[0]
                pre[i] |= ROW_HASH_BAD
            elif kind == "nil_payload":
                hash32[i] = 0
                seal[i] = 0
                pre[i] |= ROW_NIL
    return DeviceRound(n, raw, round_, H, addrs, power, hash32, seal, signer, pre if byzantine else None, expect, kinds)


def _sign(bv, sk, hcol, chunk: int | None = None):
    """ibft_sign_seals in pieces of at most the context's max_rows"""
    chunk = chunk or int(bv.max_rows)
    sigs, signers, oks = [], [], []
    for lo in range(0, len(sk), chunk):
        s, a, ok = bv.sign_seals(sk[lo:lo + chunk], hcol[lo:lo + chunk])
        sigs.append(s); signers.append(a); oks.append(ok)
    if not sigs:
        return np.zeros((0, 65), np.uint8), np.zeros((0, 20), np.uint8), np.zeros(0, bool)
    return np.concatenate(sigs), np.concatenate(signers), np.concatenate(oks)
