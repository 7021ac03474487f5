"""Run the EXACT device arithmetic source (secp256k1_dev.h, keccak_dev.h, recover_dev.h)
on the CPU through hipcc's host pass and compare with Python big ints and the oracle.
This is what lets kernel arithmetic be validated in the GPU-less build container; the
GPU parity tests (-m gpu) then confirm the compiled gfx950 code object."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref as R

P, N = R.P, R.N


def b32(x):
    return x.to_bytes(32, "big")


@pytest.fixture(scope="module")
def dev():
    import go_ibft_amd.build as build
    return C.CDLL(build.build_host_harness())


def _c2(fn, a, b):
    o = C.create_string_buffer(32)
    fn(a, b, o)
    return int.from_bytes(o.raw, "big")


def _c1(fn, a):
    o = C.create_string_buffer(32)
    fn(a, o)
    return int.from_bytes(o.raw, "big")


def test_field_and_scalar_ops(dev):
    rng = np.random.default_rng(7)
    edge = [0, 1, 2, P - 1, P - 2, N, N - 1, N + 1, 2**256 - 1, 2**255, 2**32 + 977, 2**256 - 2**32 - 978,
            (1 << 128) - 1, (1 << 224) - 1]
    vals = edge + [int.from_bytes(rng.bytes(32), "big") for _ in range(200)]
    for a in vals:
        for b in vals[::17] + edge[:5]:
            ap, bp = a % P, b % P
            assert _c2(dev.dev_fe_mul, b32(ap), b32(bp)) == ap * bp % P
            assert _c2(dev.dev_fe_add, b32(ap), b32(bp)) == (ap + bp) % P
            assert _c2(dev.dev_fe_sub, b32(ap), b32(bp)) == (ap - bp) % P
            assert _c2(dev.dev_sc_mul, b32(a), b32(b)) == a * b % N   # any 256-bit input
        assert _c1(dev.dev_fe_sqr, b32(a % P)) == (a % P) ** 2 % P
        assert _c1(dev.dev_sc_sqr, b32(a)) == a * a % N


def test_lazy_magnitudes_and_noncanonical_inputs(dev):
    """The 10x26 representation is lazily reduced: un-normalised intermediates of magnitude up
    to 2k+2 feed fe_sqr/fe_mul directly; inputs >= p are legal encodings of their residue."""
    rng = np.random.default_rng(12)
    edge = [0, 1, P - 1, P, P + 1, 2**256 - 1, 2**255, 2**234 - 1, 2**234, (1 << 26) - 1, 1 << 26]
    vals = edge + [int.from_bytes(rng.bytes(32), "big") for _ in range(60)]
    for a in vals:
        for b in vals[::9]:
            for k in (1, 2, 3):
                c = vals[(a + b + k) % len(vals)]
                o = C.create_string_buffer(32)
                dev.dev_fe_lazy(b32(a), b32(b), b32(c), k, o)
                assert int.from_bytes(o.raw, "big") == pow(((a + b) * k - c), 2, P) * (a - b) % P
            assert bool(dev.dev_fe_equal(b32(a), b32(b))) == ((a - b) % P == 0)
            assert _c2(dev.dev_fe_mul, b32(a), b32(b)) == a * b % P      # non-canonical inputs
        assert _c1(dev.dev_fe_sqr, b32(a)) == a * a % P


def test_inverse_and_sqrt_chains(dev):
    rng = np.random.default_rng(8)
    vals = [1, 2, P - 1, N - 1, 2**255] + [int.from_bytes(rng.bytes(32), "big") for _ in range(40)]
    for a in vals:
        ap, an = a % P, a % N
        if ap:
            assert _c1(dev.dev_fe_inv, b32(ap)) == pow(ap, -1, P)
        if an:
            assert _c1(dev.dev_sc_inv, b32(an)) == pow(an, -1, N)
        o = C.create_string_buffer(32)
        ok = dev.dev_fe_sqrt(b32(ap), o)
        y = pow(ap, (P + 1) // 4, P)
        assert bool(ok) == (y * y % P == ap)
        if ok:
            assert int.from_bytes(o.raw, "big") == y


def test_safegcd_inversion(dev):
    """modinv_dev.h (divsteps, 20 batches of 30) against pow(x, -1, M) for both moduli, and the
    radix-2^30 constants it hard-codes."""
    for M, limbs, inv in ((P, [0x3FFFFC2F, 0x3FFFFFFB] + [0x3FFFFFFF] * 6 + [0xFFFF], 0x2DDACACF),
                          (N, [0x10364141, 0x3F497A33, 0x348A03BB, 0x2BB739AB, 0x3FFFFEBA] + [0x3FFFFFFF] * 3 + [0xFFFF],
                           0x2A774EC1)):
        assert sum(l << (30 * i) for i, l in enumerate(limbs)) == M and pow(M, -1, 2**30) == inv
    rng = np.random.default_rng(13)
    edge = [0, 1, 2, 3, P - 1, P - 2, N - 1, N - 2, 2**255, 2**128, 2**30, 2**30 - 1, 2**60 + 1, (P + 1) // 2,
            (N + 1) // 2, P // 3, N // 3]
    vals = edge + [int.from_bytes(rng.bytes(32), "big") for _ in range(1500)] + \
        [int.from_bytes(rng.bytes(32), "big") >> s for s in range(1, 256, 5)]
    for a in vals:
        ap, an = a % P, a % N
        assert _c1(dev.dev_fe_inv_safegcd, b32(ap)) == (pow(ap, -1, P) if ap else 0)
        assert _c1(dev.dev_sc_inv_safegcd, b32(an)) == (pow(an, -1, N) if an else 0)


def test_glv_split_and_variable_base_mult(dev):
    """k ≡ k1 + k2·λ (mod n) with 128-bit halves; λ·G = (β·Gx, Gy); ecmult_var(k, P) == k·P."""
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    beta = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
    assert pow(lam, 3, N) == 1 and pow(beta, 3, P) == 1
    assert R.pt_mul(lam, R.G) == (beta * R.G[0] % P, R.G[1])
    rng = np.random.default_rng(11)
    ks = [0, 1, 2, N - 1, N - 2, N // 2, N // 2 + 1, lam, N - lam, 2**128, 2**128 - 1, 2**255] + \
        [int.from_bytes(rng.bytes(32), "big") % N for _ in range(1500)]
    for k in ks:
        o = C.create_string_buffer(64)
        f = dev.dev_glv_split(b32(k), o)
        k1, k2 = int.from_bytes(o.raw[:32], "big"), int.from_bytes(o.raw[32:], "big")
        assert k1 < 2**128 and k2 < 2**128
        assert ((-k1 if f & 1 else k1) + (-k2 if f & 2 else k2) * lam) % N == k
    for k in ks[:30]:
        pt = R.pt_mul(int.from_bytes(rng.bytes(24), "big") + 1, R.G)
        o = C.create_string_buffer(64)
        ok = dev.dev_ecmult_var(b32(k), R.pub_bytes(pt), o)
        exp = R.pt_mul(k, pt)
        assert bool(ok) == (exp is not None)
        if exp is not None:
            assert o.raw == R.pub_bytes(exp)


def test_signed_window_form_of_the_variable_base_multiplication(dev):
    """round 4: ecmult_var recodes each GLV half into 33 signed radix-16 digits (digit i = nibble_i(k + Σ 8·16^i) − 8, no carry
    chain), uses a table of the multiples 1…8 brought to ONE common Z (mixed additions on the isomorphic curve) and multiplies
    the common Z back in at the end.  (1) the digits reconstruct the scalar and stay in [−8, 7]; (2) the result equals round
    1's form (unsigned windows, 15 Jacobian multiples, full additions) and the big-int reference on random scalars and on
    scalars whose halves are all-0 / all-7 / all-8 / all-F nibbles, powers of 16 and their neighbours — every digit value,
    the carry into digit 32, additions that hit the point at infinity."""
    rng = np.random.default_rng(17)
    halves = [0, 1, 7, 8, 9, 15, 16, 2**128 - 1, 2**127, 2**127 - 1, int("7" * 32, 16), int("8" * 32, 16), int("f" * 32, 16),
              int("78" * 16, 16), int("87" * 16, 16), 8 * 16**31, 8 * 16**31 - 1, 16**31] + \
             [int.from_bytes(rng.bytes(16), "big") for _ in range(200)]
    for h in halves:
        o = C.create_string_buffer(33)
        dev.dev_window_digits(b32(h), o)
        digits = [b - 8 for b in o.raw]
        assert all(-8 <= d <= 7 for d in digits) and digits[32] in (0, 1)
        assert sum(d * 16**i for i, d in enumerate(digits)) == h
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    special = [0, 1, 2, 7, 8, 9, 15, 16, 17, 2**128 - 1, 2**128, 2**128 + 1, N - 1, N - 8, N - 9, lam, lam + 1, (8 * lam) % N, (lam * 15 + 8) % N,
               int("8" * 32, 16), (int("8" * 32, 16) * lam + int("7" * 32, 16)) % N, (int("f" * 32, 16) * lam + int("f" * 32, 16)) % N]
    ks = special + [int.from_bytes(rng.bytes(32), "big") % N for _ in range(40)]
    pts = [R.G] + [R.pt_mul(int.from_bytes(rng.bytes(24), "big") + 1, R.G) for _ in range(3)]
    for j, k in enumerate(ks):
        pt = pts[j % len(pts)]
        a, b = C.create_string_buffer(64), C.create_string_buffer(64)
        oa, ob = dev.dev_ecmult_var(b32(k), R.pub_bytes(pt), a), dev.dev_ecmult_var_v1(b32(k), R.pub_bytes(pt), b)
        exp = R.pt_mul(k, pt)
        assert bool(oa) == bool(ob) == (exp is not None), hex(k)
        if exp is not None:
            assert a.raw == b.raw == R.pub_bytes(exp), hex(k)


def test_window_table_in_lds_equals_the_private_segment_form(dev):
    """round 5: ecmult_var_lds keeps the eight (x, y) pairs of the common-Z table in the workgroup's LDS (640 B per lane, β·X
    multiplied at use, the build's Z's in registers) — same result as the private-segment form and the big-int reference on the
    special scalars of the signed-window test and random ones; every lane writes its own column only (a 3-lane "workgroup")."""
    rng = np.random.default_rng(29)
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    special = [0, 1, 2, 7, 8, 9, 15, 16, 17, 2**128 - 1, 2**128, 2**128 + 1, N - 1, N - 8, N - 9, lam, lam + 1, (8 * lam) % N, (lam * 15 + 8) % N,
               int("8" * 32, 16), (int("8" * 32, 16) * lam + int("7" * 32, 16)) % N, (int("f" * 32, 16) * lam + int("f" * 32, 16)) % N]
    ks = special + [int.from_bytes(rng.bytes(32), "big") % N for _ in range(60)]
    pts = [R.G] + [R.pt_mul(int.from_bytes(rng.bytes(24), "big") + 1, R.G) for _ in range(4)]
    for j, k in enumerate(ks):
        pt = pts[j % len(pts)]
        a, b = C.create_string_buffer(64), C.create_string_buffer(64)
        guard = C.c_int(0)
        oa = dev.dev_ecmult_var(b32(k), R.pub_bytes(pt), a)
        ob = dev.dev_ecmult_var_lds(b32(k), R.pub_bytes(pt), b, j % 3, C.byref(guard))
        exp = R.pt_mul(k, pt)
        assert guard.value == 1, "a lane wrote outside its column"
        assert bool(oa) == bool(ob) == (exp is not None), hex(k)
        if exp is not None:
            assert a.raw == b.raw == R.pub_bytes(exp), hex(k)


def _point_add_cases():
    rng = np.random.default_rng(21)
    ks = [int.from_bytes(rng.bytes(32), "big") % N for _ in range(6)] + [1, 2, N - 1]
    cases = []
    for k in ks:
        cases += [(k, k), (k, N - k), (k, 0), (0, k), (k, (k * 7 + 3) % N), (k, 1)]
    return cases + [(0, 0)]


def test_zero_filter_of_the_additions_never_misses(dev):
    """fe_z_maybe_zero(Z3) stands in for fe_is_zero(H) in jac_add / jac_add_aff: it must say "maybe" whenever H ≡ 0,
    whatever the encodings (a, a + p), and hardly ever otherwise."""
    rng = np.random.default_rng(23)
    xs = [0, 1, 2**32 + 976, 2**32 + 977, 5, P - 1] + [int.from_bytes(rng.bytes(32), "big") % P for _ in range(300)]
    for x in xs:
        z = int.from_bytes(rng.bytes(32), "big") % P or 1
        for a, b in ((x, x), (x + P, x), (x, x + P)):
            if max(a, b) >= 2**256:
                continue
            assert dev.dev_fe_zero_filter(b32(a), b32(b), b32(z)) == 7, (x, a, b)
    flagged = 0
    for _ in range(2000):
        a, b, z = (int.from_bytes(rng.bytes(32), "big") % P for _ in range(3))
        if a == b or z == 0:
            continue
        f = dev.dev_fe_zero_filter(b32(a), b32(b), b32(z))
        assert not f & 2
        flagged += f & 1
    assert flagged <= 1   # 2^-25 each


def test_exceptional_point_additions(dev):
    """P+P, P+(−P), ∞+P, P+∞ through jac_add and jac_add_aff (wave-uniform exceptional paths)."""
    for k1, k2 in _point_add_cases():
        exp = R.pt_mul((k1 + k2) % N, R.G)
        for via_aff in (0, 1):
            o = C.create_string_buffer(32)
            ok = dev.dev_point_add_case(b32(k1), b32(k2), via_aff, o)
            assert bool(ok) == (exp is not None), (k1, k2, via_aff)
            if exp is not None:
                assert int.from_bytes(o.raw, "big") == exp[0], (k1, k2, via_aff)


def test_warm_path_verify_equals_recover_and_compare(dev, oracle):
    """verify_dev.h: per-key table build + verify_known gives the verdict of recover-and-compare
    for valid, tampered, out-of-range, wrong-key and high-s signatures (both low-s policies)."""
    rng = np.random.default_rng(14)
    for key in range(2):
        sk = b32(int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1)
        pub, addr = oracle.pubkey(sk), oracle.address(oracle.pubkey(sk))
        for i in range(48):
            d = rng.bytes(32)
            sig = oracle.sign(sk, d)
            if i % 5 == 1: sig = sig[:64] + bytes([sig[64] ^ 1])
            if i % 7 == 2: sig = rng.bytes(64) + bytes([i & 1])
            if i % 11 == 3: sig = bytes(32) + sig[32:]
            if i % 13 == 4: sig = sig[:32] + b32(N) + sig[64:]
            if i % 17 == 5: sig = sig[:64] + b"\x02"
            if i % 19 == 6:
                s_ = int.from_bytes(sig[32:64], "big")
                sig = sig[:32] + b32(N - s_) + bytes([sig[64] ^ 1])
            if i % 23 == 7: sig = oracle.sign(b32(4242 + i), d)          # valid, but another key
            for fl in (0, 1):
                rec = oracle.recover_address(d, sig, fl)
                assert dev.dev_verify_known(d, sig, pub, fl) == int(rec is not None and rec == addr), (key, i, fl)
        P_ = (int.from_bytes(pub[:32], "big"), int.from_bytes(pub[32:], "big"))
        for w, e in ((0, 1), (0, 2), (0, 255), (1, 1), (31, 255), (17, 100)):
            o = C.create_string_buffer(64)
            dev.dev_qtab_entry(w, e, o)
            assert o.raw == R.pub_bytes(R.pt_mul(e << (8 * w), P_))


def test_keccak_streaming(dev, oracle):
    rng = np.random.default_rng(9)
    for ln in [0, 1, 55, 64, 131, 135, 136, 137, 200, 271, 272, 273, 1032]:
        m = rng.bytes(ln)
        o = C.create_string_buffer(32)
        dev.dev_keccak256(m, ln, o)
        assert o.raw == oracle.keccak256(m), ln
        dev.dev_digest_limbs_roundtrip(m, ln, o)
        assert o.raw == oracle.keccak256(m), ln


def test_recover_address_matches_oracle(dev, oracle):
    rng = np.random.default_rng(10)
    for i in range(150):
        sk = b32(int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1)
        d = rng.bytes(32)
        sig = oracle.sign(sk, d)
        if i % 5 == 1: sig = sig[:64] + bytes([sig[64] ^ 1])
        if i % 7 == 2: sig = rng.bytes(64) + bytes([i & 1])
        if i % 11 == 3: sig = bytes(32) + sig[32:]
        if i % 13 == 4: sig = sig[:32] + b32(N) + sig[64:]
        if i % 17 == 5: sig = sig[:64] + b"\x02"
        if i % 19 == 6:
            s = int.from_bytes(sig[32:64], "big")
            sig = sig[:32] + b32(N - s) + bytes([sig[64] ^ 1])
        if i % 23 == 7: d = b32(N + 5)          # digest >= n is reduced mod n
        o = C.create_string_buffer(20)
        for fl in (0, 1):
            ok = dev.dev_recover_address(d, sig, fl, o)
            ref = oracle.recover_address(d, sig, fl)
            assert (ref is not None) == bool(ok), (i, fl)
            if ref is not None:
                assert o.raw == ref, (i, fl)
            # round 5, the lane kernel's form: window table in "LDS"
            o2 = C.create_string_buffer(20)
            ok2 = dev.dev_recover_address_lds(d, sig, fl, o2)
            assert bool(ok2) == bool(ok), (i, fl)
            if ref is not None:
                assert o2.raw == ref, (i, fl)
            # round 6, the lane kernel's form beyond 65 536 rows: the co-Z table in the private segment, one loop
            o3 = C.create_string_buffer(20)
            ok3 = dev.dev_recover_address_private(d, sig, fl, o3)
            assert bool(ok3) == bool(ok), (i, fl)
            if ref is not None:
                assert o3.raw == ref, (i, fl)


def test_variable_time_divsteps_match_constant_time(dev):
    """divsteps_30_var strips runs of even g with one ctz, divsteps_30_lockstep also cancels up to six steps of an odd
    run with one multiplication; batch by batch both must produce the same ζ and transition matrix as the constant-time
    form (all the update code is shared)."""
    rng = np.random.default_rng(77)
    dev.dev_divsteps_agree.argtypes = [C.c_int32, C.c_uint32, C.c_uint32]
    cases = [(-1, 1, 0), (-1, 0xFFFFFC2F, 0), (5, 3, 2**31), (-7, 0xFFFFFFFF, 0xFFFFFFFF), (0, 1, 1), (-1, 1, 2**30)]
    for _ in range(4000):
        cases.append((int(rng.integers(-40, 40)), int(rng.integers(0, 2**32)) | 1, int(rng.integers(0, 2**32))))
    for _ in range(2000):   # long odd runs (g ≡ −f mod 2^k), long even runs, ζ around the swap
        f0 = int(rng.integers(0, 2**32)) | 1
        k = int(rng.integers(1, 31))
        g0 = ((-f0 * int(rng.integers(1, 2**16) | 1)) % 2**k + (int(rng.integers(0, 2**32)) << k)) % 2**32
        cases.append((int(rng.integers(-3, 4)), f0, g0))
        cases.append((int(rng.integers(-700, -500)), f0, g0 << int(rng.integers(0, 8)) & 0xFFFFFFFF))
    for zeta, f0, g0 in cases:
        assert dev.dev_divsteps_agree(zeta, f0, g0) == 1, (zeta, f0, g0)


def test_sign_row_matches_oracle_byte_for_byte(dev):
    """sign_dev.h on the CPU against oracle/secp256k1.c:orc_sign — same deterministic nonce, so the 65 bytes
    are identical; the address is the oracle's address of the oracle's public key; keys outside [1, n) fail."""
    from oracle import binding as O
    rng = np.random.default_rng(41)
    keys = [1, 2, N - 1, N - 2, (N - 1) // 2, (N + 1) // 2, 2**255 % N] + \
           [int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1 for _ in range(40)]
    digests = [bytes(32), b"\xff" * 32, b32(N), b32(N - 1), b32(N + 1)] + [rng.bytes(32) for _ in range(6)]
    high = 0
    for i, sk in enumerate(keys):
        for dg in (digests if i < 8 else digests[5 + i % 6:6 + i % 6]):
            sig, addr = C.create_string_buffer(65), C.create_string_buffer(20)
            assert dev.dev_sign(b32(sk), dg, sig, addr) == 1
            assert sig.raw == O.sign(b32(sk), dg)
            assert addr.raw == O.address(O.pubkey(b32(sk)))
            assert int.from_bytes(sig.raw[32:64], "big") <= (N - 1) // 2          # low-s, always
            assert O.ecrecover(dg, sig.raw, 1) == O.pubkey(b32(sk))               # strict-low-s recover agrees
            high += sig.raw[64]
    assert high > 0                                                               # both parities seen
    for bad in (0, N, N + 1, 2**256 - 1):
        sig, addr = C.create_string_buffer(65), C.create_string_buffer(20)
        assert dev.dev_sign(b32(bad), digests[5], sig, addr) == 0
        assert sig.raw == bytes(65) and addr.raw == bytes(20)
