"""Cross-check the C oracle against two independent derivations: pure-Python big ints
(oracle/pyref.py) and OpenSSL libcrypto's secp256k1 (oracle/openssl_xcheck.c)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref as R

ORC_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
N, P = R.N, R.P


def b32(x):
    return x.to_bytes(32, "big")


@pytest.fixture(scope="module")
def ossl():
    subprocess.check_call(["make", "-C", ORC_DIR, "libopenssl_xcheck.so"], stdout=subprocess.DEVNULL)
    path = os.path.join(ORC_DIR, "libopenssl_xcheck.so")
    if not os.path.exists(path):
        pytest.skip("libcrypto not available")
    return C.CDLL(path)


def test_keccak_vs_pyref(oracle):
    rng = np.random.default_rng(1)
    for ln in [0, 1, 31, 32, 64, 135, 136, 137, 271, 272, 273, 1032, 4096]:
        m = rng.bytes(ln)
        assert oracle.keccak256(m) == R.keccak256(m)


def test_field_scalar_vs_bigint(oracle):
    rng = np.random.default_rng(2)
    edge = [0, 1, 2, P - 1, P - 2, N - 1, N - 2, 2**255, 2**32 + 977, (1 << 128) - 1]
    vals = edge + [int.from_bytes(rng.bytes(32), "big") for _ in range(60)]
    for a in vals:
        for b in vals[::7]:
            assert int.from_bytes(oracle.fe_mul(b32(a % P), b32(b % P)), "big") == a * b % P
            assert int.from_bytes(oracle.sc_mul(b32(a % N), b32(b % N)), "big") == a * b % N
        if a % P:
            assert int.from_bytes(oracle.fe_inv(b32(a % P)), "big") == pow(a % P, -1, P)
        if a % N:
            assert int.from_bytes(oracle.sc_inv(b32(a % N)), "big") == pow(a % N, -1, N)
        y = pow(a % P, (P + 1) // 4, P)
        got = oracle.fe_sqrt(b32(a % P))
        assert (got is not None) == (y * y % P == a % P)


def test_ecmult_and_recover_vs_pyref(oracle):
    rng = np.random.default_rng(3)
    for i in range(12):
        sk = int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1
        pt = R.pubkey(sk)
        assert oracle.pubkey(b32(sk)) == R.pub_bytes(pt)
        k1 = int.from_bytes(rng.bytes(32), "big") % N
        k2 = int.from_bytes(rng.bytes(32), "big") % N
        exp = R.pt_add(R.pt_mul(k1, R.G), R.pt_mul(k2, pt))
        assert oracle.ecmult2(b32(k1), b32(k2), R.pub_bytes(pt)) == R.pub_bytes(exp)
        d = rng.bytes(32)
        k = int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1
        sig_py = R.sign(sk, d, k)                 # pyref signer, arbitrary nonce
        assert oracle.recover_address(d, sig_py) == R.address(pt)
        sig_c = oracle.sign(b32(sk), d)           # C signer, deterministic nonce
        assert R.recover_address(d, sig_c) == R.address(pt)
        junk = rng.bytes(64) + bytes([i & 1])
        assert oracle.recover_address(d, junk) == R.recover_address(d, junk)


def test_recover_vs_openssl(oracle, ossl):
    rng = np.random.default_rng(4)
    for i in range(60):
        sk = b32(int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1)
        d = rng.bytes(32)
        sig = oracle.sign(sk, d)
        pub = C.create_string_buffer(64)
        assert ossl.ossl_pubkey(sk, pub) == 1 and pub.raw == oracle.pubkey(sk)
        assert ossl.ossl_verify(d, sig, pub.raw) == 1           # C signer produces valid ECDSA
        rec = C.create_string_buffer(64)
        assert ossl.ossl_ecrecover(d, sig, rec) == 1 and rec.raw == oracle.ecrecover(d, sig)
        junk = rng.bytes(64) + bytes([i & 1])
        ok = ossl.ossl_ecrecover(d, junk, rec)
        mine = oracle.ecrecover(d, junk)
        assert (mine is not None) == bool(ok)
        if ok:
            assert rec.raw == mine


def test_batch_functions_vs_pyref(oracle):
    """orc_verify_seals / _senders / _hashes / orc_tally against per-row pure Python."""
    from oracle import workload as W
    r = W.make_round(48, 11, byzantine=True, weighted=True, with_envelopes=True)
    vs = oracle.ValSet(r.addrs, r.power)
    members = {bytes(a) for a in r.addrs}
    seals = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags)
    seals_mt = oracle.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=4)
    assert (seals == seals_mt).all()
    senders = oracle.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20)
    hashes = oracle.verify_hashes(r.raw, r.round, r.hash32, r.hash_len)
    H = R.proposal_hash(r.raw, r.round)
    assert H == r.proposal_hash
    power = 0
    for i in range(r.n):
        exp = 0
        if not r.pre_flags[i]:
            a = R.recover_address(r.hash32[i].tobytes(), r.seal65[i].tobytes())
            exp = int(a is not None and a == r.signer20[i].tobytes() and a in members)
        assert seals[i] == exp
        power += int(r.power[i]) * exp
        a = R.recover_address(R.keccak256(r.payload[r.off[i]:r.off[i + 1]]), r.msg_sig65[i].tobytes())
        assert senders[i] == int(a == r.signer20[i].tobytes() and a in members)
        assert hashes[i] == int(r.hash_len[i] == 32 and r.hash32[i].tobytes() == H)
    t = oracle.tally(vs, r.signer20, seals)
    assert t.power == power and t.quorum == 2 * int(r.power.sum()) // 3 + 1
    assert t.has_quorum == int(power >= t.quorum) and t.valid_rows == int(seals.sum())
    # a non-member with a perfectly valid signature is rejected (membership clause of backend.go:44)
    outsider = W.validator_key(999, 7)
    d = r.proposal_hash
    sig = oracle.sign(outsider, d)
    addr = oracle.address(oracle.pubkey(outsider))
    v = oracle.verify_seals(vs, np.frombuffer(d, np.uint8), np.frombuffer(sig, np.uint8), np.frombuffer(addr, np.uint8))
    assert v.tolist() == [0]
    # duplicate senders are counted once; uninitialised manager -> no quorum (validator_manager.go:82-84)
    dup_s = np.concatenate([r.signer20[:4], r.signer20[:4]])
    t2 = oracle.tally(vs, dup_s, np.ones(8, np.uint8))
    assert t2.distinct_senders == 4 and t2.power == int(r.power[:4].sum()) and t2.valid_rows == 8
    assert oracle.tally(None, dup_s, np.ones(8, np.uint8)).has_quorum == 0
    with pytest.raises(ValueError):
        oracle.ValSet(r.addrs[:3], np.zeros(3, np.uint64))
