"""TEST-ONLY stand-in for go_ibft_amd.verifier, loaded by bench.py when IBFT_BENCH_DRYRUN=1: the rank / world plumbing of
`bench.py --gpus N` (relaunch under torch.distributed.run → process group → shard ranges → launch / exchange / fetch pipeline →
max-over-ranks timing → merged record → the two output lines) runs on CPU ranks over `gloo`, because the builder leases one
GPU and the driver's 8-GPU run must not be the first execution of that path (round-4 review, item 8).

Nothing here verifies a signature: rows whose pre-flag is 0 are "valid" by decree, and the record bench.py prints in this mode
says `dry_run: true` and carries no measurement.  What IS real: the exchange buffer layout and merge of go_ibft_amd/shard.py
(the host-side statement of ibft_seals_exchange), moved by a real all-reduce between the ranks."""
from __future__ import annotations

import time
from types import SimpleNamespace

import numpy as np

import go_ibft_amd.shard as S

FLAG_PUBKEY_CACHE = 2
COMM_ID_BYTES = 128
shard_range = S.shard_range


def comm_unique_id() -> bytes:
    return bytes(COMM_ID_BYTES)


def bool_to_mask(v: np.ndarray) -> np.ndarray:
    bits = np.zeros(((len(v) + 63) // 64 or 1) * 64, dtype=np.uint8)
    bits[: len(v)] = v
    return np.packbits(bits, bitorder="little").view(np.uint64)


def make_round(n_total: int, byzantine: bool = False):
    """n_total validators with distinct addresses and power 1, one (unsigned) row each; byzantine: every fifth row carries a
    pre-flag (the stub's stand-in for a bad seal)"""
    idx = np.arange(n_total, dtype=np.uint64)
    addrs = np.zeros((n_total, 20), dtype=np.uint8)
    addrs[:, :8] = idx.view(np.uint8).reshape(-1, 8)
    pre = ((idx % 5) == 0).astype(np.uint8) if byzantine else np.zeros(n_total, np.uint8)
    return SimpleNamespace(addrs=addrs, power=np.ones(n_total, dtype=np.uint64), hash32=np.zeros((n_total, 32), np.uint8),
                           seal65=np.zeros((n_total, 65), np.uint8), signer20=addrs.copy(),
                           pre_flags=pre, expect=pre == 0)


class BatchVerifier:
    lanes_per_signature = 16

    def __init__(self, device: int = 0, max_rows: int = 4096, flags: int = 0):
        self.flags, self._comm, self._pending = flags, None, []

    def set_validators(self, height, addrs, power):
        self._addrs, self._power = np.asarray(addrs), np.asarray(power, dtype=np.uint64)
        self._index = {bytes(a): i for i, a in enumerate(self._addrs)}
        total = int(self._power.sum())
        self._quorum = 2 * total // 3 + 1

    def seals_stage(self, hash32, seal65, signer20, pre=None):
        self._signer = np.asarray(signer20)
        self._verdict = np.ones(len(self._signer), bool) if pre is None else (np.asarray(pre) == 0)
        self._staged = len(self._signer)
        self._cached = None
        return self._staged

    def _tally(self, verdict, signer):
        seen = {self._index[bytes(a)] for a, ok in zip(signer, verdict) if ok and bytes(a) in self._index}
        power = int(sum(int(self._power[i]) for i in seen))
        return seen, SimpleNamespace(power=power, quorum=self._quorum, has_quorum=int(power >= self._quorum),
                                     valid_rows=int(verdict.sum()), distinct_senders=len(seen), proposer_rows=0)

    def seals_launch(self, repeat: int = 1):
        time.sleep(2e-4)                        # stands for the verdict kernel
        if getattr(self, "_cached", None) is None:   # (the resident batch has not changed: neither has its tally)
            self._cached = self._tally(self._verdict, self._signer)
        self._seen, self._t = self._cached

    def seals_fetch(self):
        return self._verdict.copy(), self._t

    def seals_run(self):
        self.seals_launch()
        return self.seals_fetch()

    def seals_submit(self):
        self.seals_launch()
        self._inflight = getattr(self, "_inflight", []) + [(self._verdict.copy(), self._t)]
        assert len(self._inflight) <= 2

    def seals_collect(self):
        return self._inflight.pop(0)

    def is_valid_committed_seal(self, hash32, seal65, signer20, pre=None):
        self.seals_stage(hash32, seal65, signer20, pre)
        return self.seals_run()

    # ---- one rank of a sharded batch: the layout of go_ibft_amd/shard.py over the process group bench.py created ----
    def comm_init(self, uid, rank, world):
        assert len(uid) == COMM_ID_BYTES
        self._comm = (rank, world)

    def comm_info(self):
        return (self._comm[1], self._comm[0], 0)

    def comm_destroy(self):
        self._comm = None

    def seals_exchange(self, n_total: int):
        import torch
        import torch.distributed as dist
        rank, world = self._comm
        nv = len(self._addrs)
        assert S.shard_range(n_total, rank, world)[1] - S.shard_range(n_total, rank, world)[0] == self._staged
        assert len(self._pending) < 2, "two exchanges already in flight"
        slots, _, _ = S.exchange_layout(n_total, world, nv)
        buf = np.zeros(slots, dtype=np.int64)
        S.fill_local(buf, rank, n_total, world, nv, bool_to_mask(self._verdict), self._seen, self._t.valid_rows)
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        self._pending.append((n_total, t.numpy()))

    def seals_fetch_merged(self):
        n_total, buf = self._pending.pop(0)
        rank, world = self._comm
        verdict, power, valid, distinct, hq, overlap = S.merge(buf, n_total, world, [int(p) for p in self._power], self._quorum)
        return verdict, SimpleNamespace(power=power, quorum=self._quorum, has_quorum=int(hq), valid_rows=valid,
                                        distinct_senders=distinct, shard_overlap=overlap, proposer_rows=0)

    def set_kernel_timing(self, every):
        pass

    def last_kernel_ms(self):
        return 0.2, 1

    def last_dispatch(self):
        return 16, 16

    def last_cold_table(self):
        return 0

    def cache_stats(self):
        return 0, 0, 0

    def sync(self):
        pass

    def close(self):
        pass
