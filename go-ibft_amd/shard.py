"""Validator-shard layout and the one exchange step of the multi-GPU path — the host-side
statement of what libibftgpu.so does in ibft_seals_exchange / ibft_group_* (include/ibftgpu.h), used by the
multi-process CPU tests (gloo) and as documentation of the buffer.

Rows (one per message) are split into contiguous per-rank ranges whose length is a multiple of 64, so that every
64-bit verdict word is owned by exactly one rank.  Each rank verifies only its rows; the exchange is a single
all-reduce(SUM) of u64 slots

    [ K × (verdict words of rank 0 | … | rank W-1) | W × ⌈n_validators/64⌉ distinct-sender bitmap words | valid rows ]

where a rank fills only its own word range of each of the K verdict arrays (disjoint ⇒ SUM ≡ OR), its own bitmap
segment (bit v: validator v has a valid row in this shard) and its count of valid rows.  The tally is NOT additive:
ValidatorManager.HasQuorum counts a validator once however many valid messages it has
(core/validator_manager.go:86-92, 147-155 — a set of addresses), and rows of one sender may lie in two shards.  So
after the all-reduce every rank ORs the W segments and recomputes power, distinct senders and has_quorum from the
merged bitmap, exactly what the single-device tally computes from its own bitmap.
"""
from __future__ import annotations

import numpy as np


def rows_per_rank(n_total: int, world: int) -> int:
    return ((n_total + world - 1) // world + 63) // 64 * 64


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, 64-aligned split of n_total rows; the last rank takes the remainder (≡ ibft_shard_range)."""
    per = rows_per_rank(n_total, world)
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def words_per_rank(n_total: int, world: int) -> int:
    return rows_per_rank(n_total, world) // 64


def seen_words(n_validators: int) -> int:
    return (n_validators + 63) // 64


def exchange_layout(n_total: int, world: int, n_validators: int, n_masks: int = 1) -> tuple[int, int, int]:
    """(u64 slots of the all-reduce buffer, offset of the bitmap segments, offset of the valid-rows slot)
    (≡ ibft_exchange_layout)."""
    w = words_per_rank(n_total, world)
    seen_off = n_masks * w * world
    cnt_off = seen_off + world * seen_words(n_validators)
    return cnt_off + 1, seen_off, cnt_off


def fill_local(buf: np.ndarray, rank: int, n_total: int, world: int, n_validators: int, mask_words, counted_validators,
               valid_rows: int) -> None:
    """Write this rank's contribution into a zeroed exchange buffer (int64 view).  `mask_words`: one array of this
    rank's verdict words per verdict array (K of them); `counted_validators`: indices of the member validators with
    at least one valid row in this shard."""
    w = words_per_rank(n_total, world)
    masks = mask_words if isinstance(mask_words, (list, tuple)) else [mask_words]
    _, seen_off, cnt_off = exchange_layout(n_total, world, n_validators, len(masks))
    for k, m in enumerate(masks):
        mw = np.asarray(m, dtype=np.uint64)[:w]
        base = k * w * world + rank * w
        buf[base: base + len(mw)] = mw.view(np.int64)
    sw = seen_words(n_validators)
    seg = np.zeros(sw, dtype=np.uint64)
    for v in counted_validators:
        seg[v >> 6] |= np.uint64(1) << np.uint64(v & 63)
    buf[seen_off + rank * sw: seen_off + (rank + 1) * sw] = seg.view(np.int64)
    buf[cnt_off] = valid_rows


def merge(buf: np.ndarray, n_total: int, world: int, powers, quorum: int, n_masks: int = 1):
    """Decode the all-reduced buffer → (verdict bool[n_total] per verdict array, power, valid_rows, distinct,
    has_quorum, shard_overlap).  `powers`: voting power (Python int) by validator index."""
    w = words_per_rank(n_total, world)
    nv = len(powers)
    _, seen_off, cnt_off = exchange_layout(n_total, world, nv, n_masks)
    verdicts = []
    for k in range(n_masks):
        words = np.ascontiguousarray(buf[k * w * world: (k + 1) * w * world]).view(np.uint64)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")
        verdicts.append(bits[:n_total].astype(bool))      # rank r's words start at row r·w·64 = its `lo`
    sw = seen_words(nv)
    segs = np.ascontiguousarray(buf[seen_off: seen_off + world * sw]).view(np.uint64).reshape(world, sw)
    merged = np.bitwise_or.reduce(segs, axis=0) if world else np.zeros(sw, dtype=np.uint64)
    mbits = np.unpackbits(merged.view(np.uint8), bitorder="little")[:nv].astype(bool)
    per_rank = int(np.unpackbits(segs.view(np.uint8)).sum())
    distinct = int(mbits.sum())
    power = sum(int(powers[v]) for v in np.nonzero(mbits)[0])
    valid_rows = int(buf[cnt_off])
    out = verdicts[0] if n_masks == 1 else verdicts
    return out, power, valid_rows, distinct, power >= quorum, per_rank - distinct
