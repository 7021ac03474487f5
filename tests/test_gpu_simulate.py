"""GPU: go_ibft_amd/simulate.py — synthetic rounds signed on the device (ibft_sign_seals) for bench.py and load
generators.  Its by-construction verdicts (honest row valid, corrupted row invalid) must be the CPU oracle's, the
addresses must be the oracle's for the same keys, and the device verifier must agree on every row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,byz,weighted", [(64, False, False), (1000, True, True), (5000, True, False)])
def test_device_signed_round_against_the_oracle(oracle, gpu_verifier, n, byz, weighted):
    import go_ibft_amd.simulate as SIM
    r = SIM.make_round(gpu_verifier, n, 31 + n, byzantine=byz, weighted=weighted)
    assert r.proposal_hash == oracle.proposal_hash(r.raw, r.round)
    sk = SIM.secret_keys(31 + n, n)
    for i in (0, 1, n // 2, n - 1):
        assert bytes(r.addrs[i]) == oracle.address(oracle.pubkey(bytes(sk[i])))
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=8).astype(bool)
    assert (exp == r.expect).all(), [(i, r.kinds[i]) for i in np.flatnonzero(exp != r.expect)[:8]]
    if byz:
        assert 0.1 * n < (~exp).sum() < 0.3 * n and len({k for k in r.kinds if k}) == len(SIM.CORRUPTIONS)
    gpu_verifier.set_validators(1, r.addrs, r.power)
    got, t = gpu_verifier.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
    te = oracle.tally(vs, r.signer20, exp.astype(np.uint8))
    assert (got == exp).all() and (t.power, t.has_quorum, t.distinct_senders) == (te.power, te.has_quorum, te.distinct_senders)
