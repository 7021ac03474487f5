"""GPU parity tests (run on the MI355X box with `pytest -m gpu`): the HIP path, called
through the C ABI (libibftgpu.so), must agree BIT-FOR-BIT with the CPU oracle and with
the committed golden fixtures.  Integer/byte work: exact equality, no tolerance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def _check_round(bv, oracle, r, flags=0):
    """r: oracle.workload.Round.  Compare every entry point with the oracle."""
    vs = oracle.ValSet(r.addrs, r.power)
    bv.set_validators(r.height, r.addrs, r.power)
    got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, flags=flags, nthreads=8)
    assert (got == exp.astype(bool)).all(), np.nonzero(got != exp.astype(bool))[0][:10]
    te = oracle.tally(vs, r.signer20, exp)
    assert (t.power, t.quorum, t.has_quorum, t.valid_rows, t.distinct_senders) == \
           (te.power, te.quorum, te.has_quorum, te.valid_rows, te.distinct_senders)
    return got, t


@pytest.mark.parametrize("name", ["round_n64_honest", "round_n100_byz_weighted", "round_n256_byz"])
def test_golden_fixtures(gpu_verifier, name):
    g = _load(name)
    bv = gpu_verifier
    bv.set_validators(int(g["height"]), g["addrs"], g["power"])
    raw = g["raw"].tobytes()
    assert bv.proposal_hash(raw, int(g["round"])) == g["proposal_hash"].tobytes()
    hashes = bv.is_valid_proposal_hash(raw, int(g["round"]), g["hash32"], g["hash_len"])
    assert (hashes == g["exp_hashes"].astype(bool)).all()
    seals, t = bv.is_valid_committed_seal(g["hash32"], g["seal65"], g["signer20"], g["pre_flags"])
    assert (seals == g["exp_seals"].astype(bool)).all()
    assert t.power == int(g["exp_power"][0]) | (int(g["exp_power"][1]) << 64)
    assert t.quorum == int(g["exp_quorum"][0]) | (int(g["exp_quorum"][1]) << 64)
    assert t.has_quorum == int(g["exp_has_quorum"]) and t.distinct_senders == int(g["exp_distinct"])
    senders, _ = bv.is_valid_validator(g["payload"].tobytes(), g["off"], g["msg_sig65"], g["signer20"])
    assert (senders == g["exp_senders"].astype(bool)).all()


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 1000])
def test_ragged_sizes_vs_oracle(gpu_verifier, gpu_verifier_lane, oracle, n):
    """Default context: 8 lanes per signature at these sizes (8 rows per wavefront, ragged tails);
    lane context: 64 rows per wavefront."""
    from oracle import workload as W
    r = W.make_round(n, 100 + n, byzantine=True, weighted=True, with_envelopes=True)
    vs = oracle.ValSet(r.addrs, r.power)
    for bv in (gpu_verifier, gpu_verifier_lane):
        _check_round(bv, oracle, r)
        senders, _ = bv.is_valid_validator(r.payload, r.off, r.msg_sig65, r.signer20)
        assert (senders == oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20).astype(bool)).all()


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 64, 128])
def test_every_cold_variant_pinned(oracle, lanes, monkeypatch):
    """IBFT_COLD_LANES pins the cold kernel: lane kernel, 2/4/8-lane groups, one DPP row per
    signature (16), one wavefront per signature (64), TWO wavefronts per signature (128: main + helper).  Same Byzantine
    round, seals and senders, strict-low-s on and off."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    r = W.make_round(333, 4100 + lanes, byzantine=True, weighted=True, with_envelopes=True)
    vs = oracle.ValSet(r.addrs, r.power)
    for flags in (0, V.FLAG_STRICT_LOW_S):
        bv = V.BatchVerifier(flags=flags, max_rows=4096)
        try:
            bv.set_validators(1, r.addrs, r.power)
            got, _ = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
            assert bv.last_dispatch() == (lanes, 0)
            exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, flags=flags).astype(bool)
            assert (got == exp).all(), np.nonzero(got != exp)[0][:10]
            senders, _ = bv.is_valid_validator(r.payload, r.off, r.msg_sig65, r.signer20)
            exp = oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20, flags=flags).astype(bool)
            assert (senders == exp).all()
        finally:
            bv.close()


@pytest.mark.parametrize("lanes,table", [(1, "lds"), (1, "private"), (1, "private2"), (2, "lds"), (2, "private"), (4, "lds"), (4, "private")])
def test_every_table_placement_of_the_lane_and_group_kernels(oracle, lanes, table, monkeypatch):
    """round 5: IBFT_COLD_TABLE pins where the lane / group cold kernels keep the window table of u2·R — the workgroup's LDS
    (default: no private segment), the private segment with the entries read in front of the doublings (round 4's form), or
    without that prefetch (two resident wavefronts per SIMD: the lane kernel's form beyond 65 536 rows).  Same Byzantine round,
    seals and senders, both low-s modes, every placement against the oracle; then the public recover vectors."""
    import json
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    monkeypatch.setenv("IBFT_COLD_TABLE", table)
    r = W.make_round(500, 5100 + lanes, byzantine=True, weighted=True, with_envelopes=True)
    vs = oracle.ValSet(r.addrs, r.power)
    for flags in (0, V.FLAG_STRICT_LOW_S):
        bv = V.BatchVerifier(flags=flags, max_rows=4096)
        try:
            bv.set_validators(1, r.addrs, r.power)
            got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
            assert bv.last_dispatch() == (lanes, 0)
            exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, flags=flags).astype(bool)
            assert (got == exp).all(), np.nonzero(got != exp)[0][:10]
            te = oracle.tally(vs, r.signer20, exp.astype(np.uint8))
            assert (t.power, t.has_quorum) == (te.power, te.has_quorum)
            senders, _ = bv.is_valid_validator(r.payload, r.off, r.msg_sig65, r.signer20)
            exp = oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20, flags=flags).astype(bool)
            assert (senders == exp).all()
        finally:
            bv.close()
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))["public_recover_vectors"]
    addrs = np.array([np.frombuffer(bytes.fromhex(v["address"]), dtype=np.uint8) for v in k])
    other = addrs.copy()
    other[:, 19] ^= 1
    h = np.array([np.frombuffer(bytes.fromhex(v["digest"]), dtype=np.uint8) for v in k] * 2)
    s = np.array([np.frombuffer(bytes.fromhex(v["sig65"]), dtype=np.uint8) for v in k] * 2)
    bv = V.BatchVerifier(max_rows=1024)
    try:
        bv.set_validators(1, np.concatenate([addrs, other]), np.ones(2 * len(k), dtype=np.uint64))
        got, _ = bv.is_valid_committed_seal(h, s, np.concatenate([addrs, other]))
        assert got.tolist() == [True] * len(k) + [False] * len(k)
    finally:
        bv.close()


@pytest.mark.parametrize("n,split,table", [(70000, "1", 1), (70000, "0", 3), (98304, "1", 1), (100000, "1", 3)])
def test_batches_beyond_one_wavefront_per_simd(oracle, monkeypatch, n, split, table):
    """n > 65 536.  Up to 98 304 rows the batch is TWO launches (round 6): 65 536 rows through the lane kernel with its tables in
    LDS, the rest through whatever the AUTO rule picks for that many rows (here: the row-per-signature kernel for 4 464 rows, the
    2-lane group kernel for 32 768) — ibft_last_cold_table says 1; beyond that, or with IBFT_SPLIT_LARGE=0, ONE launch of the lane
    kernel with the table in the private segment (co-Z build in place, two resident wavefronts per SIMD) — table 3.  Verdicts with
    20 % Byzantine seals against the oracle (16 host threads), every row of both launches."""
    import go_ibft_amd.verifier as V
    import go_ibft_amd.simulate as SIM
    monkeypatch.setenv("IBFT_SPLIT_LARGE", split)
    bv = V.BatchVerifier(max_rows=n)
    try:
        r = SIM.make_round(bv, n, 77, byzantine=True)
        bv.set_validators(1, r.addrs, r.power)
        # a hash batch first: it leaves its verdict bits in the context's work mask (no tally consumes them) — the second launch of
        # a split batch ORs its bits into that mask and must find it cleared (the first version of the split did not clear it)
        hv = bv.is_valid_proposal_hash(r.raw, r.round, r.hash32, np.full(n, 32, np.uint8))
        assert hv.sum() > 0.7 * n
        got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
        assert bv.last_dispatch() == (1, 0) and bv.last_cold_table() == table
        exp = oracle.verify_seals(oracle.ValSet(r.addrs, r.power), r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=16).astype(bool)
        assert (got == exp).all() and (exp == r.expect).all()
        te = oracle.tally(oracle.ValSet(r.addrs, r.power), r.signer20, exp.astype(np.uint8))
        assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == (te.power, te.valid_rows, te.distinct_senders, te.has_quorum)
        got2, _ = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)      # a second pass over a mask the tally left clean
        assert (got2 == exp).all()
    finally:
        bv.close()


_ENVELOPE_ROUND = {}


def _envelope_round(n, seed):
    """(19 s of oracle signing at 70 000 rows: made once for both forms)"""
    from oracle import workload as W
    if (n, seed) not in _ENVELOPE_ROUND:
        _ENVELOPE_ROUND[(n, seed)] = W.make_round(n, seed, with_envelopes=True)
    return _ENVELOPE_ROUND[(n, seed)]


@pytest.mark.parametrize("split", ["1", "0"])
def test_senders_beyond_one_wavefront_per_simd(oracle, monkeypatch, split):
    """IsValidValidator over 70 000 envelopes (Keccak of PayloadNoSig fused, MODE = 1) — as two launches and as one: the second
    launch of the split form reads its payload offsets from row 65 536 on (a row base that was never applied to the offset column
    before round 6: only seal batches had used one)."""
    import go_ibft_amd.verifier as V
    n = 70000
    monkeypatch.setenv("IBFT_SPLIT_LARGE", split)
    r = _envelope_round(n, 4242)
    sig = np.array(r.msg_sig65, copy=True)
    bad = np.arange(0, n, 7)
    sig[bad, 9] ^= 0x20                                   # every seventh envelope signature forged, beyond row 65 536 too
    bv = V.BatchVerifier(max_rows=n)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        got, _ = bv.is_valid_validator(r.payload, r.off, sig, r.signer20)
        assert bv.last_dispatch() == (1, 0) and bv.last_cold_table() == (1 if split == "1" else 3)
        exp = oracle.verify_senders(oracle.ValSet(r.addrs, r.power), r.payload, r.off, sig, r.signer20, nthreads=16).astype(bool)
        assert (got == exp).all() and not exp[bad].any() and exp.sum() == n - len(bad)
    finally:
        bv.close()


@pytest.mark.parametrize("n,expect_group", [(1, 128), (61, 128), (512, 128), (513, 64), (1500, 64), (3000, 16), (5000, 16), (8192, 16),
                                            (12000, 4), (20000, 2), (40000, 1)])
def test_cold_group_sizes(oracle, n, expect_group):
    """AUTO picks two wavefronts per signature up to 512 rows, one up to 2 048, one DPP row per signature up to 8 192,
    then 4/2/1 lanes per signature so
    that n·G/64 ≤ 1024 wavefronts; each choice is compared with the oracle on a Byzantine round."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    assert (128 if n <= 512 else 64 if n <= 2048 else 16 if n <= 8192 else 4 if n * 4 <= 65536 else 2 if n * 2 <= 65536 else 1) == expect_group
    r = W.make_round(n, 3000 + n, byzantine=True)
    bv = V.BatchVerifier(max_rows=65536)
    try:
        _check_round(bv, oracle, r)
        assert bv.last_dispatch() == (expect_group, 0)
    finally:
        bv.close()


def test_empty_batch(gpu_verifier, oracle):
    from oracle import workload as W
    r = W.make_round(4, 1)
    bv = gpu_verifier
    bv.set_validators(1, r.addrs, r.power)
    z = np.zeros((0,), np.uint8)
    got, t = bv.is_valid_committed_seal(z.reshape(0, 32), z.reshape(0, 65), z.reshape(0, 20))
    assert len(got) == 0 and t.has_quorum == 0 and t.power == 0 and t.quorum == 3
    assert len(bv.is_valid_proposal_hash(r.raw, 0, z.reshape(0, 32), z)) == 0


def test_config2_n1024_commit_round(gpu_verifier, oracle):
    """BASELINE config #2: N=1024 validators, single round of COMMIT seals."""
    g = _load("bench_commit_n1024")
    bv = gpu_verifier
    bv.set_validators(1, g["addrs"], g["power"])
    got, t = bv.is_valid_committed_seal(g["hash32"], g["seal65"], g["signer20"])
    assert got.all() and t.has_quorum == 1 and t.power == 1024 and t.quorum == 683
    vs = oracle.ValSet(g["addrs"], g["power"])
    exp = oracle.verify_seals(vs, g["hash32"], g["seal65"], g["signer20"], nthreads=8)
    assert (got == exp.astype(bool)).all()


def test_config3_n4096_prepare_commit_sequence(gpu_verifier, oracle):
    """BASELINE config #3: N=4096, PREPARE senders + hash check + COMMIT senders + seals."""
    from oracle import workload as W
    r = W.make_round(4096, 3, byzantine=True, with_envelopes=True)
    got, t = _check_round(gpu_verifier, oracle, r)
    vs = oracle.ValSet(r.addrs, r.power)
    bv = gpu_verifier
    hashes = bv.is_valid_proposal_hash(r.raw, r.round, r.hash32, r.hash_len)
    assert (hashes == oracle.verify_hashes(r.raw, r.round, r.hash32, r.hash_len).astype(bool)).all()
    senders, _ = bv.is_valid_validator(r.payload, r.off, r.msg_sig65, r.signer20)
    assert (senders == oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20).astype(bool)).all()
    # the a1∧a2 composition of handleCommit (core/ibft.go:932-944)
    assert ((hashes & got) == got).all()


def test_quorum_boundary(gpu_verifier, oracle):
    """Exactly quorum-1 and quorum valid seals (validator_manager.go:95: power >= quorum)."""
    from oracle import workload as W
    r = W.make_round(100, 77)
    bv = gpu_verifier
    vs = oracle.ValSet(r.addrs, r.power)
    q = vs.quorum  # 67
    for k, want in ((q - 1, 0), (q, 1)):
        seal = r.seal65.copy()
        seal[k:, 40] ^= 0xFF  # corrupt s of the rows beyond k
        bv.set_validators(1, r.addrs, r.power)
        got, t = bv.is_valid_committed_seal(r.hash32, seal, r.signer20)
        exp = oracle.verify_seals(vs, r.hash32, seal, r.signer20)
        assert (got == exp.astype(bool)).all() and int(got.sum()) == k
        assert t.has_quorum == want and t.power == k


def test_duplicates_nonmembers_and_tally_entry(gpu_verifier, oracle):
    from oracle import workload as W
    r = W.make_round(32, 5, weighted=True)
    bv = gpu_verifier
    # validator set = first 24 only: rows 24..31 are valid signatures by NON-members
    bv.set_validators(1, r.addrs[:24], r.power[:24])
    vs = oracle.ValSet(r.addrs[:24], r.power[:24])
    got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20)
    assert (got == exp.astype(bool)).all() and got[:24].all() and not got[24:].any()
    # duplicate senders in one batch count once (map semantics, validator_manager.go:147-155)
    idx = np.array([0, 1, 2, 0, 1, 2, 30, 31])
    verdict = np.ones(8, bool)
    tg = bv.has_quorum(r.signer20[idx], verdict)
    te = oracle.tally(vs, r.signer20[idx], verdict.astype(np.uint8))
    assert (tg.power, tg.distinct_senders, tg.valid_rows, tg.has_quorum) == \
           (te.power, te.distinct_senders, te.valid_rows, te.has_quorum)
    assert tg.distinct_senders == 3 and tg.power == int(r.power[:3].sum())


def test_zero_power_and_missing_valset():
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(4, 9)
    bv = V.BatchVerifier(max_rows=1024)
    try:
        with pytest.raises(RuntimeError, match="validator set not configured"):
            bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)
        assert bv.try_set_validators(1, r.addrs, np.zeros(4, np.uint64)) == -6  # errVotingPowerNotCorrect
        with pytest.raises(RuntimeError, match="exceeds max_rows"):
            bv.set_validators(1, r.addrs, r.power)
            big = W.make_round(4, 9)
            bv.is_valid_committed_seal(np.zeros((2000, 32), np.uint8), np.zeros((2000, 65), np.uint8),
                                       np.zeros((2000, 20), np.uint8))
    finally:
        bv.close()


def test_strict_low_s_flag(oracle):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    n_order = W.N_ORDER
    r = W.make_round(64, 21)
    seal = r.seal65.copy()
    for i in range(0, 64, 2):  # replace by the high-s twin
        s = int.from_bytes(seal[i, 32:64].tobytes(), "big")
        seal[i, 32:64] = np.frombuffer((n_order - s).to_bytes(32, "big"), np.uint8)
        seal[i, 64] ^= 1
    vs = oracle.ValSet(r.addrs, r.power)
    for flags in (0, V.FLAG_STRICT_LOW_S):
        bv = V.BatchVerifier(flags=flags, max_rows=1024)
        try:
            bv.set_validators(1, r.addrs, r.power)
            got, _ = bv.is_valid_committed_seal(r.hash32, seal, r.signer20)
            exp = oracle.verify_seals(vs, r.hash32, seal, r.signer20, flags=flags)
            assert (got == exp.astype(bool)).all()
            assert got[1::2].all() and got[0::2].all() == (flags == 0)
        finally:
            bv.close()


def test_large_n_properties(gpu_verifier, oracle):
    """N=16384 (one GPU's shard of configs 4/5): full oracle comparison is affordable with
    8 host threads; also size-independent properties — the verdict of a row is independent
    of its neighbours (permutation equivariance) and idempotent across repeated launches."""
    from oracle import workload as W
    r = W.make_round(16384, 4, byzantine=True)
    got, t = _check_round(gpu_verifier, oracle, r)
    bv = gpu_verifier
    perm = np.random.default_rng(0).permutation(r.n)
    got_p, t_p = bv.is_valid_committed_seal(r.hash32[perm], r.seal65[perm], r.signer20[perm], r.pre_flags[perm])
    assert (got_p == got[perm]).all() and t_p.power == t.power and t_p.has_quorum == t.has_quorum
    bv.seals_stage(r.hash32, r.seal65, r.signer20, r.pre_flags)
    bv.seals_launch(3)
    again, t3 = bv.seals_fetch()
    assert (again == got).all() and t3.power == t.power


@pytest.mark.parametrize("lanes", [1, 8, 16, 64, 128])
def test_public_recover_vectors_on_gpu(lanes, monkeypatch):
    """The third-party recover vectors of tests/golden/kats.json (go-ethereum's signature test triple, the
    ecrecover-precompile example, five RFC 6979 secp256k1 vectors of the bitcoin test suites) through the cold
    kernels (lane, 8-lane, a DPP row per signature, one wavefront) and then the warm path: the signer must verify
    as the published address and as no other."""
    import json
    import go_ibft_amd.verifier as V
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))["public_recover_vectors"]
    addrs = np.array([np.frombuffer(bytes.fromhex(v["address"]), dtype=np.uint8) for v in k])
    other = addrs.copy()
    other[:, 19] ^= 1
    valset = np.concatenate([addrs, other])
    h = np.array([np.frombuffer(bytes.fromhex(v["digest"]), dtype=np.uint8) for v in k] * 2)
    s = np.array([np.frombuffer(bytes.fromhex(v["sig65"]), dtype=np.uint8) for v in k] * 2)
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)
    try:
        bv.set_validators(1, valset, np.ones(len(valset), dtype=np.uint64))
        for _ in range(2):  # cold, then warm for the learned keys
            got, _ = bv.is_valid_committed_seal(h, s, valset)
            assert got.tolist() == [True] * len(k) + [False] * len(k)
    finally:
        bv.close()


def test_stale_work_mask_bits_do_not_leak(oracle):
    """The tally leaves the work mask zeroed only for the words it consumed.  A hash pass over many rows
    (ballot words, no tally) followed by a small ibft_tally must not let the next seal pass — whose
    kernels OR their verdicts into the work mask without a memset — inherit stale bits."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(3000, 811, byzantine=True)
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags).astype(bool)
    bv = V.BatchVerifier(max_rows=4096)
    try:
        bv.set_validators(1, r.addrs, r.power)
        for _ in range(2):
            hashes = bv.is_valid_proposal_hash(r.raw, r.round, r.hash32, r.hash_len)   # all-ones words, no tally
            assert hashes.sum() > 2000
            t = bv.has_quorum(r.signer20[:100], np.ones(100, dtype=bool))               # consumes two words only
            assert t.valid_rows == 100
            got, _ = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
            assert (got == exp).all(), np.nonzero(got != exp)[0][:10]
            assert (~exp).sum() > 100
    finally:
        bv.close()


def test_proposal_hash_is_remembered_and_digest_form_agrees(oracle):
    """a1: the context hashes a proposal once (same (raw, round) → only the compares run), notices a changed
    byte or round, and the digest-supplied form gives the same verdicts; nil / short / wrong hashes stay false."""
    import go_ibft_amd.verifier as V
    rng = np.random.default_rng(99)
    bv = V.BatchVerifier(max_rows=4096)
    try:
        for L in (0, 1, 135, 136, 137, 1024, 70000):
            raw = rng.bytes(L)
            for rnd in (0, 5):
                H = oracle.proposal_hash(raw, rnd)
                n = 300
                hashes = np.tile(np.frombuffer(H, np.uint8), (n, 1)).copy()
                hl = np.full(n, 32, np.uint8)
                hashes[::3, 31] ^= 0x80
                hl[5] = 0; hashes[5] = 0
                hl[6] = 31
                exp = oracle.verify_hashes(raw, rnd, hashes, hl).astype(bool)
                assert exp.sum() > 100 and (~exp).sum() > 100
                for _ in range(3):
                    assert (bv.is_valid_proposal_hash(raw, rnd, hashes, hl) == exp).all()
                assert (bv.is_valid_proposal_hash_digest(H, hashes, hl) == exp).all()
                assert bv.proposal_hash(raw, rnd) == H
                if L:                                   # one byte of the proposal changes: nothing may be reused
                    raw2 = bytes([raw[0] ^ 1]) + raw[1:]
                    assert not bv.is_valid_proposal_hash(raw2, rnd, hashes, hl).any()
                    assert (bv.is_valid_proposal_hash(raw, rnd, hashes, hl) == exp).all()
    finally:
        bv.close()


@pytest.mark.parametrize("route", ["host", "device"])
def test_proposal_hash_routes_agree_at_block_sizes(oracle, route, monkeypatch):
    """a1 (IsValidProposalHash): keccak256(raw ‖ BE64(round)) by the library's host routine (the default since round 3:
    a sponge is sequential, one host core absorbs 13× what the one wavefront that can work on a message absorbs) and by
    the wavefront kernel (IBFT_PROPOSAL_HASH=device) — sizes around the 136-byte rate, 64 KiB, 1 MiB — against the oracle,
    through ibft_proposal_hash, ibft_verify_hashes and a COMMIT set whose combine step waits for the digest."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    monkeypatch.setenv("IBFT_PROPOSAL_HASH", route)
    rng = np.random.default_rng(5)
    bv = V.BatchVerifier(max_rows=256)
    try:
        r = W.make_round(40, 77, byzantine=True, with_envelopes=True)
        bv.set_validators(r.height, r.addrs, r.power)
        for L in (0, 1, 127, 128, 135, 136, 137, 271, 272, 1024, 65536, 1 << 20):
            raw = rng.bytes(L)
            for rnd in (0, 7, 2**40 + 3):
                H = oracle.proposal_hash(raw, rnd)
                assert bv.proposal_hash(raw, rnd) == H, (route, L, rnd)
            hashes = np.tile(np.frombuffer(H, np.uint8), (100, 1)).copy()
            hashes[::7, 3] ^= 1
            hl = np.full(100, 32, np.uint8)
            bv.forget_proposal()
            got = bv.is_valid_proposal_hash(raw, rnd, hashes, hl)
            assert (got == oracle.verify_hashes(raw, rnd, hashes, hl).astype(bool)).all(), (route, L)
        # a COMMIT set against a 64 KiB proposal: the verdict launch runs while the proposal is hashed
        raw = rng.bytes(65536)
        H = np.frombuffer(oracle.proposal_hash(raw, r.round), np.uint8)
        h32 = np.tile(H, (r.n, 1))
        h32[5, 0] ^= 1
        bv.forget_proposal()
        s, v, _ = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, h32, r.hash_len, None, raw=raw, round_=r.round)
        assert (v == oracle.verify_hashes(raw, r.round, h32, r.hash_len).astype(bool)).all() and not v[5]
    finally:
        bv.close()


def test_host_keccak_entry_point(oracle):
    """ibft_keccak256(a ‖ b): what the mirror hashes DIGEST_BY_HOST rows with (no context, no device)"""
    import go_ibft_amd.verifier as V
    rng = np.random.default_rng(9)
    for n in (0, 1, 135, 136, 137, 1000, 300000):
        x = rng.bytes(n)
        assert V.keccak256(x) == oracle.keccak256(x)
        assert V.keccak256(x[: n // 3], x[n // 3:]) == oracle.keccak256(x)
