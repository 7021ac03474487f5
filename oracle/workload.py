"""oracle/workload.py — synthetic consensus rounds for tests and bench.

TEST/BENCH INPUT GENERATION ONLY: uses the oracle's signer to produce signed COMMIT /
PREPARE rows.  Nothing here is on the measured or shipped path.

Validator i of seed S:  sk_i = keccak256("ibft-mi355x|" ‖ LE64(S) ‖ LE64(i)) mod n.
Proposal: L bytes of SplitMix64(S); round r; H = keccak256(raw ‖ BE64(r)).
COMMIT_i carries seal_i = sign(sk_i, H); every message carries
Signature = sign(sk_i, keccak256(PayloadNoSig)).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import binding as B
from . import wire

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
_M64 = (1 << 64) - 1

CORRUPTIONS = ["random65", "non_validator", "other_hash", "stolen_seal", "r_zero", "s_zero", "r_ge_n",
               "s_ge_n", "v_two", "len64", "wrong_hash_field", "nil_payload"]


def splitmix64(state: int):
    while True:
        state = (state + 0x9E3779B97F4A7C15) & _M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        yield z ^ (z >> 31)


def validator_key(seed: int, i: int) -> bytes:
    h = B.keccak256(b"ibft-mi355x|" + seed.to_bytes(8, "little") + i.to_bytes(8, "little"))
    while True:
        k = int.from_bytes(h, "big") % N_ORDER
        if k:
            return k.to_bytes(32, "big")
        h = B.keccak256(h)


@dataclass
class Round:
    seed: int
    n: int
    height: int
    round: int
    raw: bytes
    proposal_hash: bytes
    sks: list            # n × 32-byte secret keys
    addrs: np.ndarray    # n × 20
    power: np.ndarray    # n u64
    # COMMIT rows (a2)
    hash32: np.ndarray   # n × 32  per-row proposalHash (ExtractCommitHash)
    hash_len: np.ndarray  # n  u8
    seal65: np.ndarray   # n × 65  CommittedSeal
    signer20: np.ndarray  # n × 20  msg.From
    pre_flags: np.ndarray  # n u8
    kinds: list          # corruption kind per row ("" = honest)
    # message envelopes (a3) for the COMMIT set
    payload: bytes
    off: np.ndarray      # n+1 u32
    msg_sig65: np.ndarray  # n × 65


def _addresses(sks):
    """addresses of all validators (pubkey derivation dominates large sets: spread over host threads —
    ctypes releases the GIL)"""
    def one(sk):
        return np.frombuffer(B.address(B.pubkey(sk)), dtype=np.uint8)
    if len(sks) < 2048:
        return np.array([one(sk) for sk in sks], dtype=np.uint8).reshape(len(sks), 20)
    from concurrent.futures import ThreadPoolExecutor
    import os
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        return np.array(list(ex.map(one, sks, chunksize=256)), dtype=np.uint8).reshape(len(sks), 20)


def make_shard(n: int, seed: int, lo: int, hi: int, **kw) -> "Round":
    """rows [lo, hi) of make_round(n, seed, **kw): the validator table covers all n validators, only the
    shard's rows are signed (multi-GPU bench: every rank builds its own shard of one large round)"""
    return make_round(n, seed, rows=(lo, hi), **kw)


def make_round(n: int, seed: int = 1, *, height: int = 1, round_: int = 0, raw_len: int = 1024,
               weighted: bool = False, byzantine: bool = False, with_envelopes: bool = False,
               rows: tuple | None = None) -> Round:
    sm = splitmix64(seed)
    raw = b"".join(next(sm).to_bytes(8, "little") for _ in range((raw_len + 7) // 8))[:raw_len]
    H = B.proposal_hash(raw, round_)
    sks = [validator_key(seed, i) for i in range(n)]
    addrs = _addresses(sks)
    lo, hi = rows if rows is not None else (0, n)
    if weighted:
        power = np.array([1 + B.keccak256(i.to_bytes(8, "little"))[0] % 16 for i in range(n)], dtype=np.uint64)
    else:
        power = np.ones(n, dtype=np.uint64)

    hash32 = np.tile(np.frombuffer(H, dtype=np.uint8), (n, 1)).copy()
    hash_len = np.full(n, 32, dtype=np.uint8)
    seal = np.zeros((n, 65), dtype=np.uint8)
    signer = addrs.copy()
    pre = np.zeros(n, dtype=np.uint8)
    kinds = [""] * n
    bz = splitmix64(seed ^ 0xB12)
    outsider = validator_key(seed ^ 0x5EED, 1 << 40)
    other_h = B.keccak256(b"other" + H)
    kind_ctr = 0
    for i in range(n):
        bad = byzantine and (next(bz) % 5 == 0)
        mine = lo <= i < hi  # rows outside the shard only advance the corruption stream
        sig = B.sign(sks[i], H) if mine else bytes(65)
        if bad:
            kind = CORRUPTIONS[kind_ctr % len(CORRUPTIONS)]
            kind_ctr += 1
            kinds[i] = kind
            if kind == "random65":
                rb = b"".join(next(bz).to_bytes(8, "little") for _ in range(9))[:64]
                sig = rb + bytes([next(bz) & 1])
            elif kind == "non_validator":
                sig = B.sign(outsider, H) if mine else sig
            elif kind == "other_hash":
                sig = B.sign(sks[i], other_h) if mine else sig
            elif kind == "stolen_seal":
                sig = B.sign(sks[(i + 1) % n], H) if mine else sig
            elif kind == "r_zero":
                sig = bytes(32) + sig[32:]
            elif kind == "s_zero":
                sig = sig[:32] + bytes(32) + sig[64:]
            elif kind == "r_ge_n":
                sig = N_ORDER.to_bytes(32, "big") + sig[32:]
            elif kind == "s_ge_n":
                sig = sig[:32] + (N_ORDER + 1).to_bytes(32, "big") + sig[64:]
            elif kind == "v_two":
                sig = sig[:64] + b"\x02"
            elif kind == "len64":
                sig = sig[:64] + b"\x00"
                pre[i] |= B.ROW_BADLEN
            elif kind == "wrong_hash_field":
                hash32[i] = np.frombuffer(other_h, dtype=np.uint8)
                pre[i] |= B.ROW_HASH_BAD  # a1 fails → a2 short-circuited (core/ibft.go:938-943)
            elif kind == "nil_payload":
                hash_len[i] = 0
                hash32[i] = 0
                sig = bytes(65)
                pre[i] |= B.ROW_NIL
        seal[i] = np.frombuffer(sig, dtype=np.uint8)

    payload = b""
    off = np.zeros(n + 1, dtype=np.uint32)
    msg_sig = np.zeros((n, 65), dtype=np.uint8)
    if with_envelopes:
        chunks = []
        pos = 0
        for i in range(lo, hi):
            body = wire.commit_body(hash32[i].tobytes()[: int(hash_len[i])], seal[i].tobytes())
            m = wire.IbftMessage(view=wire.View(height, round_), sender=addrs[i].tobytes(), type=wire.COMMIT,
                                 payload=body)
            pns = m.payload_no_sig()
            chunks.append(pns)
            off[i] = pos
            pos += len(pns)
            msg_sig[i] = np.frombuffer(B.sign(sks[i], B.keccak256(pns)), dtype=np.uint8)
        off[hi:] = pos
        payload = b"".join(chunks)

    if rows is not None:  # the shard: row columns cut to [lo, hi), the validator table stays whole
        sl = slice(lo, hi)
        return Round(seed, hi - lo, height, round_, raw, H, sks, addrs, power, hash32[sl], hash_len[sl], seal[sl],
                     signer[sl], pre[sl], kinds[lo:hi], payload, off[lo:hi + 1] - off[lo], msg_sig[sl])
    return Round(seed, n, height, round_, raw, H, sks, addrs, power, hash32, hash_len, seal, signer, pre,
                 kinds, payload, off, msg_sig)
