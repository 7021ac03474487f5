"""GPU: a whole PREPARE / COMMIT set in one call (ibft_verify_messages) ≡ the three separate Verifier batches
(IsValidValidator core/backend.go:41-45, IsValidProposalHash :50-51, IsValidCommittedSeal :53-55) and ≡ the CPU
oracle, bit for bit; its tally ≡ HasQuorum over the rows both verdicts accept (core/ibft.go:1273-1284 on what
handleCommit's GetValidMessages returns, :932-944)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_expect(oracle, r, with_seals):
    vs = oracle.ValSet(r.addrs, r.power)
    senders = oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20).astype(bool)
    hashes = oracle.verify_hashes(r.raw, r.round, r.hash32, r.hash_len).astype(bool)
    valid = hashes
    if with_seals:
        seals = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
        valid = hashes & seals
    return vs, senders, valid


def _tally_expect(oracle, vs, r, both):
    return oracle.tally(vs, r.signer20, both.astype(np.uint8))


@pytest.mark.parametrize("n,flags", [(1, 0), (63, 0), (64, 0), (65, 0), (333, 0), (1000, 2), (4096, 0), (4096, 2), (5000, 0)])
def test_commit_set_equals_separate_calls_and_oracle(oracle, n, flags):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(n, 100 + n, byzantine=True, with_envelopes=True, weighted=True)
    # some envelopes are forged too: a foreign signature, a flipped byte, a non-member sender
    sig = r.msg_sig65.copy()
    rng = np.random.default_rng(n)
    forged = rng.choice(n, size=max(1, n // 9), replace=False)
    for j, i in enumerate(forged):
        if j % 3 == 0:
            sig[i] = r.msg_sig65[(i + 1) % n]
        elif j % 3 == 1:
            sig[i, j % 64] ^= 0x10
        else:
            sig[i, 64] ^= 1
    r.msg_sig65 = sig
    vs, senders, valid = _oracle_expect(oracle, r, True)
    bv = V.BatchVerifier(max_rows=max(n, 256), flags=flags)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        for rep in range(3 if flags else 1):    # with the key cache: cold pass, table build, warm passes
            s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                         valid_pre=r.pre_flags, raw=r.raw, round_=r.round)
            assert (s == senders).all(), np.flatnonzero(s != senders)[:8]
            assert (v == valid).all(), [(i, r.kinds[i]) for i in np.flatnonzero(v != valid)[:8]]
            et = _tally_expect(oracle, vs, r, senders & valid)
            assert (t.power, t.valid_rows, t.distinct_senders, t.has_quorum) == \
                   (et.power, et.valid_rows, et.distinct_senders, et.has_quorum)
        # the separate calls on the same context agree (and the digest form of the proposal)
        s2, _ = bv.is_valid_validator(r.payload, r.off, r.msg_sig65, r.signer20)
        h2 = bv.is_valid_proposal_hash(r.raw, r.round, r.hash32, r.hash_len)
        a2, _ = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
        assert (s2 == senders).all() and ((h2 & a2) == valid).all()
        s3, v3, t3 = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                        valid_pre=r.pre_flags, digest32=r.proposal_hash)
        assert (s3 == senders).all() and (v3 == valid).all() and t3.power == et.power
    finally:
        bv.close()


@pytest.mark.parametrize("n", [7, 200, 4095])
def test_prepare_set_has_no_seals(oracle, n):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(n, 900 + n, byzantine=True, with_envelopes=True)
    vs, senders, valid = _oracle_expect(oracle, r, False)
    bv = V.BatchVerifier(max_rows=4096)
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len,
                                     raw=r.raw, round_=r.round)
        assert (s == senders).all() and (v == valid).all()
        et = _tally_expect(oracle, vs, r, senders & valid)
        assert (t.power, t.valid_rows, t.has_quorum) == (et.power, et.valid_rows, et.has_quorum)
        # a wrong round changes the proposal hash: nothing is valid, every sender still is
        s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len,
                                     raw=r.raw, round_=r.round + 1)
        assert (s == senders).all() and not v.any() and t.valid_rows == 0 and t.has_quorum == 0
    finally:
        bv.close()


def test_pre_columns_empty_set_and_interleaving_with_other_calls(oracle):
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    r = W.make_round(300, 77, with_envelopes=True)
    bv = V.BatchVerifier(max_rows=1024)
    try:
        with pytest.raises(Exception):
            bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65, raw=r.raw)
        bv.set_validators(r.height, r.addrs, r.power)
        spre = np.zeros(300, np.uint8); spre[[0, 64, 299]] = 1
        vpre = np.zeros(300, np.uint8); vpre[[1, 64, 128]] = 4
        s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                     sender_pre=spre, valid_pre=vpre, raw=r.raw, round_=r.round)
        assert (s == (spre == 0)).all() and (v == (vpre == 0)).all() and t.valid_rows == 300 - 5
        # a seal batch right after (the work mask must be clean), then the set again, then an empty set
        got, t2 = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)
        assert got.all() and t2.valid_rows == 300
        s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                     raw=r.raw, round_=r.round)
        assert s.all() and v.all() and t.has_quorum == 1 and t.power == 300
        z = np.zeros((0, 65), np.uint8)
        s, v, t = bv.verify_messages(b"", np.zeros(1, np.uint32), z, np.zeros((0, 20), np.uint8), np.zeros((0, 32), np.uint8),
                                     np.zeros(0, np.uint8), z, raw=r.raw, round_=r.round)
        assert len(s) == 0 and len(v) == 0 and t.has_quorum == 0 and t.quorum == 201
    finally:
        bv.close()


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 64, 128])
def test_payload_lengths_around_the_keccak_rate_at_every_alignment(oracle, lanes, monkeypatch):
    """PayloadNoSig rows of every length around the 136-byte Keccak rate (empty, one block exactly, one byte over,
    several blocks), packed back to back so that rows start at every byte alignment: the kernels hash them with
    aligned dword loads + funnel shifts (kernels.hip.h:hash_range_dwords), the oracle byte by byte."""
    import go_ibft_amd.verifier as V
    monkeypatch.setenv("IBFT_COLD_LANES", str(lanes))
    rng = np.random.default_rng(1360 + lanes)
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 9, 55, 131, 132, 133, 134, 135, 136, 137, 138, 139, 140, 271, 272, 273, 407, 408, 409,
            1000, 4093] + [int(x) for x in rng.integers(0, 300, size=100)]
    n = len(lens)
    from oracle import workload as W
    sks = [W.validator_key(99, i) for i in range(n)]
    addrs = np.array([np.frombuffer(oracle.address(oracle.pubkey(sk)), np.uint8) for sk in sks])
    rows = [rng.bytes(L) for L in lens]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(lens)
    sig = np.array([np.frombuffer(oracle.sign(sks[i], oracle.keccak256(rows[i])), np.uint8) for i in range(n)])
    sig[5, 3] ^= 1      # one forged row: the verdict is not trivially all-ones
    payload = b"".join(rows)
    vs = oracle.ValSet(addrs, np.ones(n, np.uint64))
    exp = oracle.verify_senders(vs, payload, off, sig, addrs).astype(bool)
    assert exp.sum() == n - 1
    bv = V.BatchVerifier(max_rows=256)
    try:
        bv.set_validators(1, addrs, np.ones(n, np.uint64))
        got, _ = bv.is_valid_validator(payload, off, sig, addrs)
        assert bv.last_dispatch() == (lanes, 0)
        assert (got == exp).all(), [lens[i] for i in np.flatnonzero(got != exp)]
        h = np.zeros((n, 32), np.uint8)
        s, v, _ = bv.verify_messages(payload, off, sig, addrs, h, np.full(n, 32, np.uint8), raw=b"x", round_=0)
        assert (s == exp).all() and not v.any()
        # pinned columns (ibft_pinned_alloc) give the same answer
        s, v, _ = bv.verify_messages(V.pinned_copy(payload), V.pinned_copy(off), V.pinned_copy(sig), V.pinned_copy(addrs),
                                     V.pinned_copy(h), V.pinned_copy(np.full(n, 32, np.uint8)), raw=b"x", round_=0)
        assert (s == exp).all() and not v.any()
        assert bv.gather_batches() == 1     # only that last batch had every column in pinned blocks
        pin = [V.pinned_copy(x) for x in (payload, off, sig, addrs)]
        got, _ = bv.is_valid_validator(*pin)
        assert (got == exp).all() and bv.gather_batches() == 2
        got, _ = bv.is_valid_validator(pin[0], pin[1], sig, pin[3])   # one pageable column: copy commands for all
        assert (got == exp).all() and bv.gather_batches() == 2
        # odd source alignments inside a pinned block take the gather's dword / byte paths
        big = V.pinned_copy(np.zeros(len(payload) + 64, np.uint8))
        for shift in (1, 2, 4, 7):
            big[shift:shift + len(payload)] = np.frombuffer(payload, np.uint8)
            got, _ = bv.is_valid_validator(big[shift:shift + len(payload)], pin[1], pin[2], pin[3])
            assert (got == exp).all()
        assert bv.gather_batches() == 6
    finally:
        bv.close()


def test_prepared_call_on_fixed_buffers_refilled_in_place(oracle):
    """prepare_messages binds the column buffers once; refilling them in place (the next round) and calling run() again
    judges the new content — pinned buffers, so the gather launch reads them."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    n = 500
    r1 = W.make_round(n, 31, with_envelopes=True)
    r2 = W.make_round(n, 31, byzantine=True, with_envelopes=True)      # same validators and proposal, some bad seals
    assert (r1.addrs == r2.addrs).all() and r1.raw == r2.raw
    size = max(len(r1.payload), len(r2.payload))
    cols = dict(payload=V.pinned_copy(np.zeros(size, np.uint8)), off=V.pinned_copy(r1.off), sig=V.pinned_copy(r1.msg_sig65),
                frm=V.pinned_copy(r1.signer20), h=V.pinned_copy(r1.hash32), hl=V.pinned_copy(r1.hash_len), seal=V.pinned_copy(r1.seal65),
                vpre=V.pinned_copy(r1.pre_flags))
    bv = V.BatchVerifier(max_rows=1024)
    try:
        bv.set_validators(r1.height, r1.addrs, r1.power)
        run = bv.prepare_messages(cols["payload"], cols["off"], cols["sig"], cols["frm"], cols["h"], cols["hl"], cols["seal"],
                                  valid_pre=cols["vpre"], raw=r1.raw, round_=r1.round)
        for r in (r1, r2, r1):
            cols["payload"][:len(r.payload)] = np.frombuffer(r.payload, np.uint8)
            for k, a in (("off", r.off), ("sig", r.msg_sig65), ("frm", r.signer20), ("h", r.hash32), ("hl", r.hash_len),
                         ("seal", r.seal65), ("vpre", r.pre_flags)):
                cols[k][...] = a
            ws, wv, t = run()
            vs, senders, valid = _oracle_expect(oracle, r, True)
            assert (V.mask_to_bool(ws, n) == senders).all() and (V.mask_to_bool(wv, n) == valid).all()
            assert t.valid_rows == int((senders & valid).sum())
        assert bv.gather_batches() == 3
    finally:
        bv.close()


def test_long_payload_rows_in_pinned_columns(oracle):
    """64 rows that do not fit the gather launch's LDS buffer (32 KiB) are hashed straight from the host column."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.integers(300, 1500, size=150)]
    n = len(lens)
    sks = [W.validator_key(5, i) for i in range(n)]
    addrs = np.array([np.frombuffer(oracle.address(oracle.pubkey(sk)), np.uint8) for sk in sks])
    rows = [rng.bytes(L) for L in lens]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(lens)
    sig = np.array([np.frombuffer(oracle.sign(sks[i], oracle.keccak256(rows[i])), np.uint8) for i in range(n)])
    sig[9, 40] ^= 2
    payload = b"".join(rows)
    vs = oracle.ValSet(addrs, np.ones(n, np.uint64))
    exp = oracle.verify_senders(vs, payload, off, sig, addrs).astype(bool)
    bv = V.BatchVerifier(max_rows=256)
    try:
        bv.set_validators(1, addrs, np.ones(n, np.uint64))
        h = np.zeros((n, 32), np.uint8)
        pin = [V.pinned_copy(x) for x in (payload, off, sig, addrs, h, np.full(n, 32, np.uint8))]
        s, v, _ = bv.verify_messages(*pin, raw=b"x", round_=0)
        assert (s == exp).all() and exp.sum() == n - 1 and not v.any() and bv.gather_batches() == 1
    finally:
        bv.close()


def test_small_context_set_after_a_plain_batch(oracle):
    """Regression (found by the sharded-set tests of round 3): a context for fewer than 2 048 rows keeps its verdict
    words in a 256-byte buffer that never needs reallocating, so the first message set — which doubles the verdict rows —
    must not trust the "mask is clean" bookkeeping of an earlier plain batch: the seal words [half/64, …) had never been
    zeroed and stale bits made forged seals valid.  A plain batch first, then a Byzantine COMMIT set, on one context."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    # dirty the allocator's free lists first (fresh process memory is zero and hides the bug): a context that has run a
    # batch full of valid rows leaves set verdict words behind in the memory it frees
    r0 = W.make_round(300, 7299, with_envelopes=True)
    junk = V.BatchVerifier(max_rows=300)
    junk.set_validators(r0.height, r0.addrs, r0.power)
    junk.verify_messages(r0.payload, r0.off, r0.msg_sig65, r0.signer20, r0.hash32, r0.hash_len, r0.seal65, raw=r0.raw, round_=r0.round)
    junk.close()
    for n in (200, 256, 700):
        r = W.make_round(n, 7300 + n, byzantine=True, with_envelopes=True, weighted=True)
        vs, senders, valid = _oracle_expect(oracle, r, True)
        bv = V.BatchVerifier(max_rows=n)
        try:
            bv.set_validators(r.height, r.addrs, r.power)
            a2, _ = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)   # plain batch: words [0, n/64)
            s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                         valid_pre=r.pre_flags, raw=r.raw, round_=r.round)
            assert (s == senders).all() and (v == valid).all(), [(i, r.kinds[i]) for i in np.flatnonzero(v != valid)[:8]]
        finally:
            bv.close()
