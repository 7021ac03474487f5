"""GPU: the seal-digest convention (include/ibftgpu.h: ibft_set_seal_digest).  core/backend.go:53-55 says a committed seal
is "the signature for the proposal hash"; WHICH bytes are signed is the embedding Backend's choice — the 32-byte hash itself
(the default everywhere else in this suite) or keccak256(hash ‖ suffix), e.g. a Backend that appends the COMMIT type byte
before hashing.  Under either convention every a2 route (seal batches, staged batches, message sets from columns and from
wire bytes, sharded groups, the batch signer) must equal the oracle fed with the digests computed here in Python, while a1
keeps comparing the CARRIED hash."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SUFFIXES = [b"\x02", b"", b"commit|" + bytes(range(57))]     # one byte, empty (a bare re-hash), the longest allowed (64)


def _round_under(oracle, n, seed, suffix, envelopes=False):
    """a Byzantine round whose HONEST seals sign keccak(hash ‖ suffix); the corrupted rows keep their corruption"""
    from oracle import workload as W
    r = W.make_round(n, seed, byzantine=True, weighted=True, with_envelopes=False)
    dig = np.array([np.frombuffer(oracle.keccak256(r.hash32[i].tobytes() + suffix), np.uint8) for i in range(n)])
    seal = r.seal65.copy()
    for i in range(n):
        if r.kinds[i] in ("", "len64", "v_two", "r_zero", "s_zero", "r_ge_n", "s_ge_n"):
            s = np.frombuffer(oracle.sign(r.sks[i], dig[i].tobytes()), np.uint8).copy()
            if r.kinds[i] == "v_two":
                s[64] = 2
            elif r.kinds[i] == "r_zero":
                s[:32] = 0
            elif r.kinds[i] == "s_zero":
                s[32:64] = 0
            elif r.kinds[i] in ("r_ge_n", "s_ge_n"):
                s[:] = r.seal65[i]
            seal[i] = s
        elif r.kinds[i] == "stolen_seal":
            seal[i] = np.frombuffer(oracle.sign(r.sks[(i + 1) % n], dig[i].tobytes()), np.uint8)
    r.seal65 = seal
    if envelopes:
        from oracle import wire
        chunks, sigs, pos = [], [], [0]
        for i in range(n):
            body = wire.commit_body(r.hash32[i].tobytes()[: int(r.hash_len[i])], r.seal65[i].tobytes())
            m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.COMMIT, payload=body)
            pns = m.payload_no_sig()
            chunks.append(pns)
            pos.append(pos[-1] + len(pns))
            sigs.append(np.frombuffer(oracle.sign(r.sks[i], oracle.keccak256(pns)), np.uint8))
        r.payload, r.off, r.msg_sig65 = b"".join(chunks), np.array(pos, np.uint32), np.array(sigs)
    return r, dig


@pytest.mark.parametrize("suffix", SUFFIXES)
@pytest.mark.parametrize("n", [100, 4096, 9000])
def test_seal_batches_under_the_convention(oracle, n, suffix):
    import go_ibft_amd.verifier as V
    r, dig = _round_under(oracle, n, 8800 + n, suffix)
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, dig, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
    assert 0.7 < exp.mean() < 0.9
    te = oracle.tally(vs, r.signer20, exp.astype(np.uint8))
    bv = V.BatchVerifier(max_rows=max(n, 256))
    try:
        bv.set_validators(1, r.addrs, r.power)
        bv.set_seal_digest(suffix)
        got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)   # the CARRIED hashes go in
        assert (got == exp).all(), [(int(i), r.kinds[i]) for i in np.flatnonzero(got != exp)[:8]]
        assert (t.power, t.valid_rows, t.has_quorum) == (te.power, te.valid_rows, te.has_quorum)
        bv.seals_stage(r.hash32, r.seal65, r.signer20, r.pre_flags)                          # staged once, run thrice
        for _ in range(3):
            got, _ = bv.seals_run()
            assert (got == exp).all()
        bv.set_seal_digest(None)                                                             # back to the default: other digests
        none, t_none = bv.seals_run()                                # a batch staged under the other convention is dropped
        assert len(none) == 0 and t_none.valid_rows == 0
        got0, _ = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
        exp0 = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
        assert (got0 == exp0).all() and not got0.any()
    finally:
        bv.close()


@pytest.mark.parametrize("n,flags", [(333, 0), (4096, 0), (1000, 2)])
def test_message_sets_keep_a1_on_the_carried_hash(oracle, n, flags):
    """COMMIT sets from columns and from wire bytes: the valid bit is (carried hash ≡ keccak(proposal)) ∧ seal over
    keccak(carried hash ‖ suffix)"""
    import go_ibft_amd.verifier as V
    from oracle import wire
    suffix = b"\x02"
    r, dig = _round_under(oracle, n, 9100 + n, suffix, envelopes=True)
    vs = oracle.ValSet(r.addrs, r.power)
    senders = oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20, nthreads=8).astype(bool)
    hashes = oracle.verify_hashes(r.raw, r.round, r.hash32, r.hash_len).astype(bool)
    seals = oracle.verify_seals(vs, dig, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
    te = oracle.tally(vs, r.signer20, (senders & hashes & seals).astype(np.uint8))
    bv = V.BatchVerifier(max_rows=max(n, 256), flags=flags)
    g = V.DeviceGroup([0, 0, 0], flags=flags, max_rows_total=max(n, 256))
    try:
        bv.set_validators(r.height, r.addrs, r.power)
        bv.set_seal_digest(suffix)
        g.set_validators(r.height, r.addrs, r.power)
        g.set_seal_digest(suffix)
        for rep in range(3 if flags else 1):
            s, v, t = bv.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                         valid_pre=r.pre_flags, raw=r.raw, round_=r.round)
            assert (s == senders).all() and (v == (hashes & seals)).all(), \
                [(int(i), r.kinds[i], bool(hashes[i]), bool(seals[i])) for i in np.flatnonzero(v != (hashes & seals))[:8]]
            assert (t.power, t.valid_rows, t.has_quorum) == (te.power, te.valid_rows, te.has_quorum)
            s, v, t = g.verify_messages(r.payload, r.off, r.msg_sig65, r.signer20, r.hash32, r.hash_len, r.seal65,
                                        valid_pre=r.pre_flags, raw=r.raw, round_=r.round)
            assert (s == senders).all() and (v == (hashes & seals)).all() and t.power == te.power
        # the same messages as the transport's bytes (rows whose seal has another length take the stock route: skipped here)
        ok = [i for i in range(n) if r.kinds[i] not in ("len64", "nil_payload")]
        msgs = []
        for i in ok:
            m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.COMMIT,
                                 payload=wire.commit_body(r.hash32[i].tobytes()[: int(r.hash_len[i])], r.seal65[i].tobytes()))
            m.signature = r.msg_sig65[i].tobytes()
            msgs.append(m.encode())
        off = np.concatenate([[0], np.cumsum([len(x) for x in msgs])]).astype(np.uint32)
        s, v, cls, t = bv.verify_messages_wire(b"".join(msgs), off, r.height, r.round, raw=r.raw, want_rows=False)
        assert (s == senders[ok]).all() and (v == (hashes & seals)[ok]).all()
    finally:
        g.close()
        bv.close()


def test_the_batch_signer_signs_what_its_verifiers_check(oracle):
    import go_ibft_amd.verifier as V
    import go_ibft_amd.simulate as SIM
    suffix = b"\x02"
    bv = V.BatchVerifier(max_rows=1024)
    try:
        bv.set_seal_digest(suffix)
        sk = SIM.secret_keys(77, 500)
        H = np.tile(np.frombuffer(oracle.keccak256(b"a proposal"), np.uint8), (500, 1))
        sig, addr, ok = bv.sign_seals(sk, H)
        assert ok.all()
        dig = oracle.keccak256(H[0].tobytes() + suffix)
        for i in (0, 17, 499):
            assert sig[i].tobytes() == oracle.sign(sk[i].tobytes(), dig)           # byte-identical to the oracle's signer
            assert oracle.recover_address(dig, sig[i].tobytes()) == addr[i].tobytes()
        bv.set_validators(1, addr, np.ones(500, np.uint64))
        got, t = bv.is_valid_committed_seal(H, sig, addr)
        assert got.all() and t.has_quorum == 1
        bv.set_seal_digest(None)
        got, _ = bv.is_valid_committed_seal(H, sig, addr)
        assert not got.any()
    finally:
        bv.close()
