"""The TUNED CPU recovery (oracle/recover_tuned.inc — bench.py's cpu_baseline leg) held against the plain oracle path, which
stays the checker: every piece against Python big integers (field product / square / inverse / square root, r⁻¹ mod n, the
endomorphism split, the width-5 NAF, k·P), then the whole recovery row by row on random, known-answer and adversarial inputs."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE


def b32(x: int) -> bytes:
    return x.to_bytes(32, "big")


EDGE_FE = [0, 1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2**255, 2**256 - 2**32 - 978, 2**128, 2**128 - 1, 2**64, 2**192 + 1,
           0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF0000000000000000, P - 0x1000003D1, 0x1000003D1, 0x1000003D0,
           2**52 - 1, 2**52, 2**104 - 1, 2**104, 2**156 - 1, 2**156, 2**208 - 1, 2**208, 2**256 - 2**208, (2**52 - 1) << 52,
           (2**52 - 1) << 104 | (2**52 - 1), P - 2**52, P - 2**208, sum(1 << (52 * i) for i in range(5)) % P,
           sum((2**52 - 1) << (52 * i) for i in (0, 2, 4)) % P]


def test_field_pieces_against_big_ints(oracle):
    rng = np.random.default_rng(21)
    vals = EDGE_FE + [int.from_bytes(rng.bytes(32), "big") % P for _ in range(400)]
    for k, a in enumerate(vals):
        b = vals[(7 * k + 3) % len(vals)]
        m, q, inv, root = oracle.tuned_fe_ops(b32(a), b32(b))
        assert int.from_bytes(m, "big") == a * b % P
        assert int.from_bytes(q, "big") == a * a % P
        assert int.from_bytes(inv, "big") == (pow(a, -1, P) if a else 0)
        assert oracle.tuned_fe_inv_gcd(b32(a)) == inv          # divsteps = addition chain
        if pow(a, (P - 1) // 2, P) in (0, 1):
            assert root is not None and int.from_bytes(root, "big") == pow(a, (P + 1) // 4, P)
        else:
            assert root is None
        lazy = oracle.tuned_fe_lazy(b32(a), b32(b))
        assert int.from_bytes(lazy, "big") == (8 * (a + b) - b) * (2 * (b - 3 * a)) ** 2 % P
    # the product of the largest operands, and operands that are not reduced below p never reach these functions (inputs < p)
    m, q, _, _ = oracle.tuned_fe_ops(b32(P - 1), b32(P - 1))
    assert int.from_bytes(m, "big") == 1 and int.from_bytes(q, "big") == 1


def test_scalar_inverse_against_big_ints(oracle):
    rng = np.random.default_rng(22)
    assert oracle.tuned_sc_inv(b32(0)) == b32(0)
    assert pow(P, -1, 2**62) == 0x27C7F6E22DDACACF and pow(N, -1, 2**62) == 0x34F20099AA774EC1   # the constants of the batches
    vals = [1, 2, 3, N - 1, N - 2, (N - 1) // 2, (N + 1) // 2, 2**255, 2**128, 2**64 - 1, 2**200 + 12345] + \
           [int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1 for _ in range(4000)] + \
           [(1 << k) % N for k in range(1, 256)] + [N - (1 << k) for k in range(0, 255)] + [(1 << k) - 1 for k in range(2, 256)]
    for a in vals:
        assert int.from_bytes(oracle.tuned_sc_inv(b32(a)), "big") == pow(a, -1, N), hex(a)
        assert oracle.tuned_sc_inv(b32(a)) == oracle.sc_inv(b32(a))


def test_endomorphism_constants_and_split(oracle):
    """λ³ ≡ 1 (mod n), β³ ≡ 1 (mod p), λ·G = (β·Gx, Gy); the split's halves recombine and stay below 2^128"""
    assert pow(LAMBDA, 3, N) == 1 and LAMBDA != 1 and pow(BETA, 3, P) == 1 and BETA != 1
    g = oracle.pubkey(b32(1))
    gx, gy = int.from_bytes(g[:32], "big"), int.from_bytes(g[32:], "big")
    assert oracle.pubkey(b32(LAMBDA)) == b32(BETA * gx % P) + b32(gy)
    rng = np.random.default_rng(23)
    ks = [0, 1, 2, N - 1, N - 2, LAMBDA, N - LAMBDA, LAMBDA - 1, LAMBDA + 1, (N - 1) // 2, 2**128, 2**128 - 1, 2**255] + \
         [int.from_bytes(rng.bytes(32), "big") % N for _ in range(3000)]
    for k in ks:
        k1, n1, k2, n2 = oracle.tuned_glv_split(b32(k))
        assert k1 < 2**128 + 2**64 and k2 < 2**128 + 2**64, hex(k)     # (the lattice bound is 2^128; the NAF takes 129 bits)
        s1, s2 = (-k1 if n1 else k1), (-k2 if n2 else k2)
        assert (s1 + s2 * LAMBDA - k) % N == 0, hex(k)


def test_wnaf_digits(oracle):
    rng = np.random.default_rng(24)
    ks = [0, 1, 15, 16, 17, 31, 32, 33, 2**128 - 1, 2**128, 2**129 - 1, 2**64 - 1, 2**64, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF1] + \
         [int.from_bytes(rng.bytes(17), "big") % 2**129 for _ in range(2000)]
    for k in ks:
        d = oracle.tuned_wnaf5(b32(k))
        assert sum(v << i for i, v in enumerate(d)) == k, hex(k)
        assert all(v == 0 or (v % 2 and -15 <= v <= 15) for v in d)
        nz = [i for i, v in enumerate(d) if v]
        assert all(b - a >= 5 for a, b in zip(nz, nz[1:]))              # non-adjacent, width 5
        assert len(d) <= 131 and (not d or d[-1] != 0)


def test_variable_base_multiple_against_plain_path(oracle):
    """k·P of the tuned path (common-Z table on the isomorphic curve, two NAFs) = k·P of the plain path (orc_ecmult2)"""
    rng = np.random.default_rng(25)
    pts = [oracle.pubkey(b32(int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1)) for _ in range(6)] + [oracle.pubkey(b32(1))]
    ks = [0, 1, 2, 3, 15, 16, 17, N - 1, N - 2, LAMBDA, N - LAMBDA, LAMBDA + 1, (N - 1) // 2, (N + 1) // 2, 2**128, 2**255] + \
         [int.from_bytes(rng.bytes(32), "big") % N for _ in range(150)]
    for i, k in enumerate(ks):
        p = pts[i % len(pts)]
        assert oracle.tuned_ecmult_var(b32(k), p) == oracle.ecmult2(b32(0), b32(k), p), hex(k)


def _rows(oracle, n, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        sk = b32(int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1)
        d = rng.bytes(32)
        rows.append((d, oracle.sign(sk, d)))
    return rows


def test_recover_matches_plain_path_on_every_kind_of_row(oracle):
    rng = np.random.default_rng(26)
    rows = _rows(oracle, 300, 27)
    cases = []
    for d, sig in rows:
        r, s, v = sig[:32], int.from_bytes(sig[32:64], "big"), sig[64]
        cases += [(d, sig), (d, sig[:64] + bytes([v ^ 1])), (d, r + b32(N - s) + bytes([v ^ 1])), (rng.bytes(32), sig)]
    d0, sig0 = rows[0]
    r0, s0 = sig0[:32], sig0[32:64]
    cases += [(d0, sig0[:64] + bytes([2])), (d0, sig0[:64] + bytes([27])), (d0, b32(0) + s0 + b"\0"), (d0, r0 + b32(0) + b"\0"),
              (d0, b32(N) + s0 + b"\0"), (d0, r0 + b32(N) + b"\0"), (d0, b32(N - 1) + s0 + b"\1"), (d0, b32(P - 1)[:32] + s0 + b"\0"),
              (b32(0), sig0), (b32(N), sig0), (b32(2**256 - 1), sig0), (d0, b32(1) + b32(1) + b"\0"), (d0, b32(1) + b32(1) + b"\1"),
              (d0, b32(5) + s0 + b"\0"), (d0, b32(7) + s0 + b"\1")]                    # (x = 5, 7: no point / a point, as it falls)
    cases += [(rng.bytes(32), rng.bytes(64) + bytes([int(rng.integers(0, 2))])) for _ in range(400)]   # random r: half have no point
    n_ok = 0
    for flags in (0, oracle.FLAG_STRICT_LOW_S):
        for d, sig in cases:
            want = oracle.ecrecover(d, sig, flags)
            assert oracle.ecrecover_tuned(d, sig, flags) == want, (d.hex(), sig.hex(), flags)
            assert oracle.recover_address_tuned(d, sig, flags) == oracle.recover_address(d, sig, flags)
            n_ok += want is not None
    assert n_ok > 1000


def test_recover_tuned_on_the_public_known_answers(oracle):
    k = json.load(open(os.path.join(HERE, "golden", "kats.json")))
    for v in k["public_recover_vectors"]:
        d, sig = bytes.fromhex(v["digest"]), bytes.fromhex(v["sig65"])
        assert oracle.recover_address_tuned(d, sig).hex() == v["address"], v["source"]
        if v["pub64"]:
            assert oracle.ecrecover_tuned(d, sig).hex() == v["pub64"]


def test_degenerate_sums_take_the_exception_branches(oracle):
    """u1·G + u2·R built to meet the doubling / infinity branches of the mixed addition: with R = k·G the sum is
    (u1 + u2·k)·G, so choosing u1 ≡ −u2·k (mod n) gives infinity (recover fails in both paths), and rows whose partial
    sums coincide with a table entry of G are found by brute force over small scalars."""
    # R = G (r = Gx, v = parity of Gy): Q = r⁻¹(s·G − z·G) = (s − z)/r · G;  s = z → infinity
    g = oracle.pubkey(b32(1))
    r, v = g[:32], g[63] & 1
    for z in (1, 2, 12345, N - 1):
        sig = r + b32(z) + bytes([v])
        assert oracle.ecrecover(b32(z), sig) is None and oracle.ecrecover_tuned(b32(z), sig) is None
    # s − z = r → Q = G; s − z = small multiples → small multiples of G (the accumulator meets low table entries of G)
    rr = int.from_bytes(r, "big")
    for m in list(range(1, 40)) + [255, 256, 257, 65535, 65536, N - 1, N - 2]:
        for z in (0, 1, 77):
            s = (z + m * rr) % N
            if s == 0:
                continue
            sig = r + b32(s) + bytes([v])
            want = oracle.ecrecover(b32(z), sig)
            assert want == oracle.pubkey(b32(m)) and oracle.ecrecover_tuned(b32(z), sig) == want, (m, z)


def test_tuned_batch_verdicts_match(oracle):
    from oracle import workload as W
    rd = W.make_round(256, seed=31, byzantine=True)
    vs = oracle.ValSet(rd.addrs, rd.power)
    want = oracle.verify_seals(vs, rd.hash32, rd.seal65, rd.signer20, rd.pre_flags)
    for nt in (1, 3):
        got = oracle.verify_seals_tuned(vs, rd.hash32, rd.seal65, rd.signer20, rd.pre_flags, nthreads=nt)
        assert (got == want).all() and 0 < int(want.sum()) < len(want)
