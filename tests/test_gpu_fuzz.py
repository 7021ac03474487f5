"""GPU fuzz: large batches of adversarial / random rows through every kernel family (cold lane,
cold group, warm group, warm lane) must agree with the oracle row by row.  Catches rare-path
divergences (exceptional point additions, range edges, junk that happens to decode)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fuzz_rows(oracle, r, rng, n_rows):
    """Rows built from an honest round: random validator, then one of many mutations."""
    from oracle import workload as W
    n = r.n
    idx = rng.integers(0, n, n_rows)
    hash32 = r.hash32[idx].copy()
    seal = r.seal65[idx].copy()
    signer = r.signer20[idx].copy()
    kind = rng.integers(0, 16, n_rows)
    N = W.N_ORDER
    for i in range(n_rows):
        k = kind[i]
        if k == 0: seal[i] = np.frombuffer(rng.bytes(65), np.uint8)                    # junk (v random byte)
        elif k == 1: seal[i, :64] = np.frombuffer(rng.bytes(64), np.uint8); seal[i, 64] &= 1
        elif k == 2: seal[i, rng.integers(0, 64)] ^= 1 << rng.integers(0, 8)          # single bit flip in r/s
        elif k == 3: hash32[i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)        # single bit flip in digest
        elif k == 4: signer[i] = r.signer20[(idx[i] + 1) % n]                          # someone else's address
        elif k == 5: seal[i, 64] ^= 1
        elif k == 6:
            s = int.from_bytes(seal[i, 32:64].tobytes(), "big")
            seal[i, 32:64] = np.frombuffer((N - s).to_bytes(32, "big"), np.uint8); seal[i, 64] ^= 1
        elif k == 7: seal[i, :32] = np.frombuffer((N - 1).to_bytes(32, "big"), np.uint8)   # r = n-1
        elif k == 8: seal[i, 32:64] = np.frombuffer((1).to_bytes(32, "big"), np.uint8)     # s = 1
        elif k == 9: hash32[i] = 0                                                         # z = 0
        elif k == 10: hash32[i] = 0xFF                                                     # z >= n
        elif k == 11: signer[i] = np.frombuffer(rng.bytes(20), np.uint8)                   # non-member
        # 12..15: untouched honest rows
    return hash32, seal, signer


@pytest.mark.parametrize("flags,kernel,n_rows", [(0, 0, 3000), (0, 1, 3000), (0, 0, 40000), (2, 0, 3000), (2, 1, 3000),
                                                 (2, 2, 700), (3, 0, 3000)])
def test_fuzz_rows_vs_oracle(oracle, flags, kernel, n_rows):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    rng = np.random.default_rng(1000 + flags * 10 + kernel)
    r = W.make_round(200, 8800)
    vs = oracle.ValSet(r.addrs, r.power)
    bv = V.BatchVerifier(flags=flags, kernel=kernel, max_rows=65536)
    try:
        bv.set_validators(1, r.addrs, r.power)
        if flags & V.FLAG_PUBKEY_CACHE:                      # learn keys + build tables first
            bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)
            bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)
            assert bv.cache_stats()[0] == 200
        for rep in range(2):
            hash32, seal, signer = _fuzz_rows(oracle, r, rng, n_rows)
            got, t = bv.is_valid_committed_seal(hash32, seal, signer)
            exp = oracle.verify_seals(vs, hash32, seal, signer, flags=flags & 1, nthreads=16).astype(bool)
            assert (got == exp).all(), np.nonzero(got != exp)[0][:10]
            te = oracle.tally(vs, signer, exp.astype(np.uint8))
            assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == \
                   (te.power, te.valid_rows, te.distinct_senders, te.has_quorum)
            assert 0.2 < got.mean() < 0.5          # the mix really contains both verdicts
    finally:
        bv.close()
