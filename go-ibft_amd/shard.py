"""Validator-shard layout and the one exchange step of the multi-GPU path — the host-side
statement of what libibftgpu.so does in ibft_seals_exchange (include/ibftgpu.h), used by the multi-process CPU
tests (gloo) and as documentation of the buffer.

Rows (one per stored message; the store is keyed by sender, messages/messages.go:64, so a validator appears at
most once per view) are split into contiguous per-rank ranges whose length is a multiple of 64, so that every
64-bit verdict word is owned by exactly one rank.  Each rank verifies only its rows; the exchange is a single
all-reduce(SUM) of u64 slots

    [ mask words of rank 0 | … | rank W-1 | 2·PW 32-bit pieces of the partial power | valid rows | distinct senders ]

where a rank fills only its own word range (disjoint ⇒ SUM ≡ OR) and its partial tally.  The voting power
travels as 32-BIT PIECES in 64-bit slots — piece k = Σ over the rank's counted validators of bits [32k, 32k+32) of
their power — so no carry can be lost whatever the total (summing two 64-bit halves dropped the carry
out of the low word).  PW = 1 for u64 powers, 4 for 256-bit powers.  `has_quorum` is recomputed from the merged
power — it is NOT additive.
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


def exchange_layout(n_total: int, world: int, power_words: int = 1) -> tuple[int, int]:
    """(number of u64 slots in the all-reduce buffer, offset of the tally slots) (≡ ibft_exchange_layout)."""
    w = words_per_rank(n_total, world)
    return w * world + 2 * power_words + 2, w * world


def fill_local(buf: np.ndarray, rank: int, n_total: int, world: int, mask_words: np.ndarray,
               powers, valid_rows: int, distinct: int, power_words: int = 1) -> None:
    """Write this rank's contribution into a zeroed exchange buffer (int64 view).  `powers`: the powers (Python
    ints) of the distinct member senders among this rank's valid rows."""
    w = words_per_rank(n_total, world)
    _, off = exchange_layout(n_total, world, power_words)
    mw = np.asarray(mask_words, dtype=np.uint64)[:w]
    buf[rank * w: rank * w + len(mw)] = mw.view(np.int64)
    for k in range(2 * power_words):
        buf[off + k] = sum((int(p) >> (32 * k)) & 0xFFFFFFFF for p in powers)
    buf[off + 2 * power_words] = valid_rows
    buf[off + 2 * power_words + 1] = distinct


def merge(buf: np.ndarray, n_total: int, world: int, quorum: int, power_words: int = 1):
    """Decode the all-reduced buffer → (verdict bool[n_total], power, valid_rows, distinct, has_quorum)."""
    w = words_per_rank(n_total, world)
    _, off = exchange_layout(n_total, world, power_words)
    words = np.ascontiguousarray(buf[: w * world]).view(np.uint64)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")
    verdict = bits[:n_total].astype(bool)          # rank k's words start at row k·w·64 = its `lo`
    power = sum(int(buf[off + k]) << (32 * k) for k in range(2 * power_words))
    valid_rows = int(buf[off + 2 * power_words])
    distinct = int(buf[off + 2 * power_words + 1])
    return verdict, power, valid_rows, distinct, power >= quorum
