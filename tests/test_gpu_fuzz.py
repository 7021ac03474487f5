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


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 64, 128])
def test_crafted_scalars_all_variants(oracle, lanes, monkeypatch):
    """Signatures built so that u2 = s/r (and u1 = −z/r) take extreme shapes: tiny, ±1 around 2^64 and
    2^128 (the piece boundaries of the lane groups and of the one-wavefront kernel), n − small, all-ones
    windows, zero digest.  Most window digits are then zero, accumulators stay at infinity for long
    stretches, top digits and carries of the signed recoding fire — every kernel variant must still
    name the same signer as the oracle, cold and then warm."""
    import go_ibft_amd.verifier as V
    from oracle import pyref
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    n = pyref.N
    rng = np.random.default_rng(4242)
    ts = [1, 2, 3, 7, 8, 9, 15, 16, 17, 255, 256, 2**16 - 1, 2**32, 2**63, 2**64 - 1, 2**64, 2**64 + 1, 2**127,
          2**128 - 1, 2**128, 2**128 + 1, 2**192, n - 1, n - 2, n - 16, n - 2**64, (n - 1) // 2, (n + 1) // 2,
          int("8" * 64, 16) % n, int("7" * 64, 16), int("f" * 32, 16)]
    hs, sigs, addrs = [], [], []
    for i, t in enumerate(ts):
        k = int.from_bytes(rng.bytes(32), "big") % (n - 1) + 1
        x, y = pyref.pt_mul(k, pyref.G)
        r = x % n
        if r == 0 or r != x:
            continue
        z = 0 if i % 5 == 0 else (r * ts[(i * 7) % len(ts)]) % n if i % 5 == 1 else int.from_bytes(rng.bytes(32), "big")
        sig = r.to_bytes(32, "big") + ((t * r) % n).to_bytes(32, "big") + bytes([i & 1])
        h = z.to_bytes(32, "big")
        a = oracle.recover_address(h, sig)
        assert a is not None and a == pyref.recover_address(h, sig)
        hs.append(np.frombuffer(h, dtype=np.uint8)); sigs.append(np.frombuffer(sig, dtype=np.uint8))
        addrs.append(np.frombuffer(a, dtype=np.uint8))
    hs, sigs, addrs = np.array(hs), np.array(sigs), np.array(addrs)
    assert len(addrs) >= 28 and len(np.unique(addrs, axis=0)) == len(addrs)
    wrong = addrs.copy()
    wrong[:, 0] ^= 0x80
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)
    try:
        bv.set_validators(1, np.concatenate([addrs, wrong]), np.ones(2 * len(addrs), dtype=np.uint64))
        for _ in range(2):
            got, _ = bv.is_valid_committed_seal(np.concatenate([hs, hs]), np.concatenate([sigs, sigs]),
                                                np.concatenate([addrs, wrong]))
            assert got[:len(addrs)].all() and not got[len(addrs):].any()
        assert bv.cache_stats()[0] == len(addrs)
    finally:
        bv.close()


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 64, 128])
def test_last_addition_exceptional_all_variants(oracle, lanes, monkeypatch):
    """u1·G = ±u2·R: the addition that joins the fixed-base part and the variable-base part meets equal or opposite points
    (a doubling / the point at infinity — the latter is no key: rejected).  R = k·G, z = ±s·k; with the recovery id of the
    other root too (then it is an ordinary addition).  Mixed with ordinary rows so that the rare route of the
    row-per-signature kernel (which joins the two parts symbolically, before it knows √(x³+7)) runs next to common rows.
    Every kernel variant, cold then warm, must agree with the oracle row by row."""
    import go_ibft_amd.verifier as V
    from oracle import pyref
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    n = pyref.N
    rng = np.random.default_rng(909)
    hs, sigs, signer, want = [], [], [], []
    for i in range(24):
        k = int.from_bytes(rng.bytes(32), "big") % (n - 1) + 1
        x, y = pyref.pt_mul(k, pyref.G)
        r, s = x % n, int.from_bytes(rng.bytes(32), "big") % (n - 1) + 1
        if r != x:
            continue
        sign = 1 if i % 2 == 0 else -1
        h = ((sign * s * k) % n).to_bytes(32, "big")
        sig = r.to_bytes(32, "big") + s.to_bytes(32, "big") + bytes([(y & 1) ^ (1 if i % 3 == 2 else 0)])
        a = oracle.recover_address(h, sig)
        assert a == pyref.recover_address(h, sig)
        hs.append(h); sigs.append(sig); want.append(a is not None)
        signer.append(a if a is not None else bytes([i + 1]) * 20)
    for i in range(8):                                # ordinary rows in between
        ski = (int.from_bytes(rng.bytes(32), "big") % (n - 1) + 1).to_bytes(32, "big")
        h = rng.bytes(32)
        sg = oracle.sign(ski, h)
        hs.insert(3 * i, h); sigs.insert(3 * i, sg); want.insert(3 * i, True)
        signer.insert(3 * i, oracle.recover_address(h, sg))
    assert 4 <= want.count(False) <= 16
    hs = np.array([np.frombuffer(h, dtype=np.uint8) for h in hs])
    sigs = np.array([np.frombuffer(x, dtype=np.uint8) for x in sigs])
    signer = np.array([np.frombuffer(a, dtype=np.uint8) for a in signer])
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)
    try:
        bv.set_validators(1, np.unique(signer, axis=0), np.ones(len(np.unique(signer, axis=0)), dtype=np.uint64))
        for _ in range(2):
            got, _ = bv.is_valid_committed_seal(hs, sigs, signer)
            assert got.tolist() == want
    finally:
        bv.close()


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 64, 128])
def test_recovery_id_policy_r_plus_n_candidate_is_never_tried(oracle, lanes, monkeypatch):
    """include/ibftgpu.h, conventions: v is the parity of R.y and nothing else — R.x = r always.  SEC 1 §4.1.6's
    second candidate R.x = r + n exists only for r < p − n (≈ 2^128.4); a signature whose TRUE nonce point has
    x = r + n is 'recoverable' only with recovery id 2 / 3, which this interface (like go-ethereum's 65-byte
    [R‖S‖V], V ∈ {0, 1}) does not have.  Such signatures are constructed here on purpose: every kernel variant,
    cold and warm, must (a) reject v = 2 and v = 3, (b) with v = 0 / 1 name whatever signer x = r gives (never the
    x = r + n one) — i.e. agree with the oracle row by row."""
    import go_ibft_amd.verifier as V
    from oracle import pyref
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    n, p = pyref.N, pyref.P
    rng = np.random.default_rng(77)
    hs, sigs, want, true_signer = [], [], [], []
    r = 5
    while len(hs) < 4 * 12:
        r += int(rng.integers(1, 1 << 60))
        assert r < p - n
        x = r + n                                   # the nonce point really has x = r + n ≥ n
        y2 = (pow(x, 3, p) + 7) % p
        y = pow(y2, (p + 1) // 4, p)
        if y * y % p != y2:
            continue
        s = int.from_bytes(rng.bytes(32), "big") % (n - 1) + 1
        z = int.from_bytes(rng.bytes(32), "big")
        # the key this signature was "made with": Q = r^-1 (s·R − z·G) with R = (r + n, y)
        R = (x, y)
        rinv = pow(r, -1, n)
        Q = pyref.pt_add(pyref.pt_mul(s * rinv % n, R), pyref.pt_mul((-z * rinv) % n, pyref.G))
        signer_true = pyref.address(Q)
        for v in (0, 1, 2, 3):
            sig = r.to_bytes(32, "big") + s.to_bytes(32, "big") + bytes([v])
            h = z.to_bytes(32, "big")
            a = oracle.recover_address(h, sig)
            assert a == pyref.recover_address(h, sig)
            if v >= 2:
                assert a is None
            hs.append(np.frombuffer(h, np.uint8)); sigs.append(np.frombuffer(sig, np.uint8))
            want.append(a); true_signer.append(signer_true)
    hs, sigs = np.array(hs), np.array(sigs)
    # members: the x = r signers (where one exists) and the "true" x = r + n signers
    named = [np.frombuffer(a, np.uint8) for a in want if a is not None]
    valset = np.unique(np.array(named + [np.frombuffer(a, np.uint8) for a in true_signer]), axis=0)
    vs = oracle.ValSet(valset, np.ones(len(valset), np.uint64))
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)
    try:
        bv.set_validators(1, valset, np.ones(len(valset), np.uint64))
        for claim in ("x=r signer", "x=r+n signer"):
            signer = np.array([np.frombuffer((w if (claim == "x=r signer" and w is not None) else t), np.uint8)
                               for w, t in zip(want, true_signer)])
            exp = oracle.verify_seals(vs, hs, sigs, signer).astype(bool)
            for _ in range(2):                      # cold, then warm for the keys learned
                got, _ = bv.is_valid_committed_seal(hs, sigs, signer)
                assert (got == exp).all(), (claim, np.nonzero(got != exp)[0][:8])
            v_col = sigs[:, 64]
            assert not got[v_col >= 2].any()        # recovery ids 2, 3: rejected whoever is claimed
            if claim == "x=r+n signer":
                assert not got.any()                # the r + n candidate is never the one recovered
            else:
                assert got[v_col < 2].sum() >= 10   # x = r on the curve: those rows DO verify as the x = r signer
    finally:
        bv.close()
