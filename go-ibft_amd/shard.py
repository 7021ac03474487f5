"""Validator-shard layout and the one exchange step of the multi-GPU path.

Rows (one per stored message; the store is keyed by sender, messages/messages.go:64, so a
validator appears at most once per view) are split into contiguous per-rank ranges whose
length is a multiple of 64, so that every 64-bit verdict word is owned by exactly one rank.
Each rank verifies only its rows; the exchange is a single all-reduce(SUM) of

    [ mask words of rank 0 | … | mask words of rank W-1 | power_lo | power_hi | counts ]

where a rank fills only its own word range (disjoint ⇒ SUM ≡ OR) and its partial tally.
`has_quorum` is recomputed from the merged power — it is NOT additive.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, 64-aligned split of n_total rows; the last rank takes the remainder."""
    per = ((n_total + world - 1) // world + 63) // 64 * 64
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def words_per_rank(n_total: int, world: int) -> int:
    return ((n_total + world - 1) // world + 63) // 64


def exchange_layout(n_total: int, world: int) -> tuple[int, int]:
    """(number of int64 slots in the all-reduce buffer, offset of the 3 tally slots)."""
    w = words_per_rank(n_total, world)
    return w * world + 3, w * world


def fill_local(buf: np.ndarray, rank: int, n_total: int, world: int, mask_words: np.ndarray,
               power: int, valid_rows: int, distinct: int) -> None:
    """Write this rank's contribution into a zeroed exchange buffer (int64 view)."""
    w = words_per_rank(n_total, world)
    _, off = exchange_layout(n_total, world)
    mw = np.asarray(mask_words, dtype=np.uint64)[:w]
    buf[rank * w: rank * w + len(mw)] = mw.view(np.int64)
    buf[off] = np.array([power & (2**64 - 1)], dtype=np.uint64).view(np.int64)[0]
    buf[off + 1] = np.array([power >> 64], dtype=np.uint64).view(np.int64)[0]
    buf[off + 2] = np.int64(valid_rows | (distinct << 32))


def merge(buf: np.ndarray, n_total: int, world: int, quorum: int):
    """Decode the all-reduced buffer → (verdict bool[n_total], power, valid_rows, distinct, has_quorum)."""
    w = words_per_rank(n_total, world)
    _, off = exchange_layout(n_total, world)
    words = np.asarray(buf[: w * world]).view(np.uint64)
    verdict = np.zeros(n_total, dtype=bool)
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        bits = np.unpackbits(words[r * w:(r + 1) * w].view(np.uint8), bitorder="little")
        verdict[lo:hi] = bits[: hi - lo].astype(bool)
    tl = np.asarray(buf[off: off + 3]).view(np.uint64)
    # partial power sums are u64 pairs; low words may wrap when added as int64 — callers that
    # can exceed 2^63 total power pass partials that fit (powers are u64, sums are exact mod 2^64
    # in each word; the carry between words is reconstructed by the caller only for > 2^64 totals)
    power = int(tl[0]) + (int(tl[1]) << 64)
    valid_rows = int(tl[2]) & 0xFFFFFFFF
    distinct = int(tl[2]) >> 32
    return verdict, power, valid_rows, distinct, power >= quorum
