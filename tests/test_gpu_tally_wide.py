"""GPU: the weighted quorum tally (a8) with arbitrary-precision voting powers and across several tally
workgroups.  Oracle = oracle/semantics.ValidatorManager (Python big ints restating
/root/reference/core/validator_manager.go:61-135 — GetVotingPowers returns *big.Int)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# /root/reference/core/validator_manager_test.go:18-187 (Test_CalculateQuorum): powers, signers, hasQuorum
REF_TABLE = [
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, ['A', 'B', 'C', 'D'], True),
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1}, ['A', 'B'], False),
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1, 'E': 1, 'F': 1}, ['A', 'B', 'C', 'D', 'E'], True),
    ({'A': 1, 'B': 1, 'C': 1, 'D': 1, 'E': 1, 'F': 1}, ['A', 'B', 'C', 'D'], False),
    ({'A': 2, 'B': 2, 'C': 2, 'D': 3}, ['A', 'C', 'D'], True),
    ({'A': 2, 'B': 2, 'C': 2, 'D': 3}, ['A', 'D'], False),
    ({'A': 2, 'B': 2, 'C': 3, 'D': 3}, ['A', 'B', 'D'], True),
    ({'A': 2, 'B': 2, 'C': 3, 'D': 3}, ['A', 'D'], False),
    ({'A': 2, 'B': 7, 'C': 7, 'D': 5}, ['A', 'B', 'C'], True),
    ({'A': 2, 'B': 7, 'C': 7, 'D': 5}, ['C', 'D'], False),
]


def _addr(name: str) -> bytes:
    return name.encode() * 20


@pytest.mark.parametrize("scale", [1, 10**18, 10**18 * 2**64 + 12345, (2**256 - 1) // 7])
def test_reference_quorum_table_with_scaled_powers(scale):
    """Test_CalculateQuorum with every power multiplied by `scale` (1: also through the u64 entry point):
    the decision of HasQuorum is scale-invariant only up to the floor in ⌊2·total/3⌋+1, so the expected value
    comes from the big-int oracle, and at scale 1 it must equal the reference's own table."""
    import go_ibft_amd.verifier as V
    from oracle.semantics import ValidatorManager
    bv = V.BatchVerifier(max_rows=1024)
    try:
        for powers, signers, want in REF_TABLE:
            names = sorted(powers)
            addrs = np.frombuffer(b"".join(_addr(n) for n in names), np.uint8).reshape(-1, 20)
            pw = [powers[n] * scale for n in names]
            vm = ValidatorManager()
            assert vm.init({_addr(n): powers[n] * scale for n in names})
            expect = vm.has_quorum([_addr(s) for s in signers])
            if scale == 1:
                assert expect == want
            entry = [bv.set_validators_u256] + ([bv.set_validators] if scale == 1 else [])
            for setv in entry:
                setv(1, addrs, pw)
                send = np.frombuffer(b"".join(_addr(s) for s in signers), np.uint8).reshape(-1, 20)
                t = bv.has_quorum(send, np.ones(len(signers), bool))
                w = bv.last_tally_wide()
                assert bool(t.has_quorum) == expect and bool(w.has_quorum) == expect
                assert w.quorum == vm.quorum and w.power == sum(powers[s] * scale for s in signers)
                assert t.power == w.power & (2**128 - 1) and t.quorum == w.quorum & (2**128 - 1)
    finally:
        bv.close()


@pytest.mark.parametrize("n,wide", [(300, False), (5000, False), (5000, True), (65536, False), (65536, True), (70000, True)])
def test_tally_over_many_workgroups_with_duplicates_and_nonmembers(n, wide):
    """ibft_tally on a caller-supplied mask: random verdict bits, every third sender duplicated somewhere else
    in the batch (counted once), a tenth of the senders not in the set; n > 4096 spans several tally workgroups
    (device-scope bitmap + ticket).  Run twice: the kernel must leave its bitmap / accumulators clean."""
    import go_ibft_amd.verifier as V
    from oracle.semantics import ValidatorManager
    rng = np.random.default_rng(n + wide)
    nv = n * 9 // 10
    addrs = rng.integers(0, 256, (n, 20), dtype=np.uint8)
    if wide:
        powers = [int.from_bytes(rng.bytes(32), "big") >> int(rng.integers(0, 200)) for _ in range(nv)]
        powers[0] = 2**256 - 1
    else:
        powers = [int(x) for x in rng.integers(1, 2**63, nv, dtype=np.uint64)]
        powers[0] = 2**64 - 1
    sender = addrs.copy()
    dup = rng.permutation(n)[: n // 3]
    sender[dup] = addrs[rng.integers(0, n, len(dup))]
    verdict = rng.random(n) < 0.7
    vm = ValidatorManager()
    assert vm.init({bytes(addrs[i]): powers[i] for i in range(nv)})
    counted = {bytes(sender[i]) for i in np.nonzero(verdict)[0]}
    exp_power = sum(vm.power.get(a, 0) for a in counted)
    exp_distinct = sum(1 for a in counted if a in vm.power)
    bv = V.BatchVerifier(max_rows=max(n, 1024))
    try:
        (bv.set_validators_u256 if wide else bv.set_validators)(1, addrs[:nv], powers)
        for _ in range(2):
            t = bv.has_quorum(sender, verdict)
            w = bv.last_tally_wide()
            assert (t.valid_rows, t.distinct_senders) == (int(verdict.sum()), exp_distinct)
            assert w.power == exp_power and w.quorum == vm.quorum
            assert bool(t.has_quorum) == (exp_power >= vm.quorum)
        # a mask that reaches the quorum exactly on the boundary: accumulate senders until power ≥ quorum
        order, acc, k = np.arange(nv), 0, 0
        while acc < vm.quorum:
            acc += powers[order[k]]
            k += 1
        for kk, want in ((k - 1, False), (k, True)):
            v2 = np.zeros(n, bool)
            v2[:kk] = True
            t = bv.has_quorum(addrs, v2)
            assert bool(t.has_quorum) == want and bv.last_tally_wide().power == sum(powers[:kk])
    finally:
        bv.close()


def test_wide_powers_through_the_seal_path(oracle):
    """IsValidCommittedSeal + HasQuorum in one call with 256-bit powers (wei-denominated stakes)."""
    import go_ibft_amd.verifier as V
    from oracle import workload as W
    from oracle.semantics import ValidatorManager
    r = W.make_round(200, 4242, byzantine=True)
    stakes = [(1 + (i * 7919) % 1000) * 10**18 * 32 for i in range(r.n)]   # 32 … 32 000 tokens in wei: > 2^64
    vm = ValidatorManager()
    assert vm.init({bytes(r.addrs[i]): stakes[i] for i in range(r.n)})
    vs = oracle.ValSet(r.addrs, r.power)
    exp = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags).astype(bool)
    bv = V.BatchVerifier(max_rows=1024)
    try:
        bv.set_validators_u256(1, r.addrs, stakes)
        got, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20, r.pre_flags)
        assert (got == exp).all()
        w = bv.last_tally_wide()
        ok_senders = [bytes(r.signer20[i]) for i in np.nonzero(exp)[0]]
        assert w.power == sum(vm.power[a] for a in set(ok_senders)) and w.quorum == vm.quorum
        assert bool(t.has_quorum) == vm.has_quorum(ok_senders)
    finally:
        bv.close()
