"""GPU: the signing side, ibft_sign_seals.  Oracle = oracle/secp256k1.c:orc_sign (same
deterministic nonce rule, so the 65 bytes must be identical) and the verify side of the library itself:
what the device signed must verify on the device, through every dispatch of a2.
Reference call being batched: Backend.BuildCommitMessage (/root/reference/core/backend.go:12-34)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def b32(x):
    return x.to_bytes(32, "big")


def _keys(rng, n):
    edge = [1, 2, 3, N - 1, N - 2, (N - 1) // 2, (N + 1) // 2, 2**255 % N, 2**128, 2**128 - 1]
    ks = edge + [int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1 for _ in range(n - len(edge))]
    return np.frombuffer(b"".join(b32(k) for k in ks[:n]), np.uint8).reshape(-1, 32)


def test_device_seals_are_the_oracles_byte_for_byte():
    import go_ibft_amd.verifier as V
    from oracle import binding as O
    rng = np.random.default_rng(2024)
    n = 333  # ragged: five full wavefronts and a partial one
    sk = _keys(rng, n)
    hs = np.frombuffer(rng.bytes(32 * n), np.uint8).reshape(-1, 32).copy()
    hs[0] = 0
    hs[1] = 0xFF
    hs[2] = np.frombuffer(b32(N), np.uint8)       # digest ≡ 0 (mod n)
    hs[3] = np.frombuffer(b32(N + 1), np.uint8)
    bv = V.BatchVerifier(max_rows=1024)
    try:
        sig, signer, ok = bv.sign_seals(sk, hs)
        assert ok.all()
        for i in range(n):
            assert sig[i].tobytes() == O.sign(sk[i].tobytes(), hs[i].tobytes()), i
            assert signer[i].tobytes() == O.address(O.pubkey(sk[i].tobytes())), i
        s_int = [int.from_bytes(sig[i, 32:64].tobytes(), "big") for i in range(n)]
        assert max(s_int) <= (N - 1) // 2 and set(sig[:, 64].tolist()) == {0, 1}
    finally:
        bv.close()


def test_refused_keys_and_empty_batch():
    import go_ibft_amd.verifier as V
    rng = np.random.default_rng(5)
    sk = _keys(rng, 70).copy()
    bad = {4: 0, 17: N, 40: N + 1, 69: 2**256 - 1}
    for i, k in bad.items():
        sk[i] = np.frombuffer(b32(k), np.uint8)
    hs = np.frombuffer(rng.bytes(32 * 70), np.uint8).reshape(-1, 32)
    bv = V.BatchVerifier(max_rows=256)
    try:
        sig, signer, ok = bv.sign_seals(sk, hs)
        for i in range(70):
            assert ok[i] == (i not in bad)
            if i in bad:
                assert not sig[i].any() and not signer[i].any()
        # the staged batch verifies row by row: refused rows (zero signature) are invalid, the rest valid
        bv.set_validators(1, signer[ok], np.ones(int(ok.sum()), np.uint64))
        verdict, t = bv.seals_run()
        assert (verdict == ok).all() and t.valid_rows == 66
        sig0, signer0, ok0 = bv.sign_seals(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8))
        assert sig0.shape == (0, 65) and ok0.shape == (0,)
    finally:
        bv.close()


@pytest.mark.parametrize("n", [64, 1000, 4096, 10000])
def test_sign_then_verify_resident_round_trip(n):
    """sign → (batch is staged) → verify, cold and warm, and the same seals through the one-shot entry point;
    flipping one bit of any output breaks exactly that row."""
    import go_ibft_amd.verifier as V
    rng = np.random.default_rng(n)
    sk = np.frombuffer(rng.bytes(32 * n), np.uint8).reshape(-1, 32).copy()
    sk[:, 0] &= 0x7F                       # < 2^255 < n: every key usable
    sk[:, 31] |= 1
    hs = np.tile(np.frombuffer(rng.bytes(32), np.uint8), (n, 1))   # one proposal hash, n committed seals
    bv = V.BatchVerifier(max_rows=max(n, 1024), flags=V.FLAG_PUBKEY_CACHE)
    try:
        sig, signer, ok = bv.sign_seals(sk, hs)
        assert ok.all() and len({s.tobytes() for s in signer}) == n
        bv.set_validators(7, signer, np.ones(n, np.uint64))
        for _ in range(3):                 # cold pass learns the keys, later passes run the warm kernels
            verdict, t = bv.seals_run()
            assert verdict.all() and t.valid_rows == n and t.distinct_senders == n and t.has_quorum
        sig2 = sig.copy()
        rows = rng.choice(n, size=min(n, 37), replace=False)
        for j, i in enumerate(rows):
            sig2[i, j % 65] ^= 1 << (j % 8) if j % 65 != 64 else 1
        verdict, t = bv.is_valid_committed_seal(hs, sig2, signer)
        want = np.ones(n, bool)
        want[rows] = False
        assert (verdict == want).all() and t.valid_rows == n - len(rows)
    finally:
        bv.close()
