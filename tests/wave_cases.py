"""Shared checks for wave_fe_dev.h (one wavefront per signature).

Each check takes callables so the SAME cases run against the host build (64-coroutine lockstep
emulator, tests/test_dev_wave_host.py) and against the gfx950 code object
(tests/test_gpu_arith.py through csrc/devtest.hip):
    fe_op(op, a[4][10], b[4][10]) -> out[4][16]      pt_op(op, p[4][31], q[4][31]) -> out[4][31]
    recover(hash32, sig65, flags) -> (ok, addr20)
Test infrastructure only.
"""
import random

import numpy as np

from oracle import pyref

P = 2**256 - 2**32 - 977
U = 2**26 + 2**20
M26 = 2**26 - 1


def limbs(x, n=10):
    return [(x >> (26 * i)) & M26 for i in range(n)]


def value(l):
    return sum(int(v) << (26 * i) for i, v in enumerate(l))


def rand_limbs(rng, mag):
    """limbs up to mag·U: random, or pinned to the extremes"""
    mode = rng.randrange(4)
    hi = mag * U
    if mode == 0:
        return [hi] * 10
    if mode == 1:
        return [rng.choice([0, hi, hi - 1, 1]) for _ in range(10)]
    return [rng.randrange(hi + 1) for _ in range(10)]


def check_neg_constants(neg_limb, fe_op=None):
    s = sum(1 << (26 * i) for i in range(10))
    for which, m in ((0, 1), (1, 2), (2, 8)):
        k = [neg_limb(which, i) for i in range(10)]
        assert value(k) % P == 0
        assert all(m * U <= v < m * U + 2**26 for v in k)
        e = (-m * U * s) % P
        assert k == [m * U + d for d in limbs(e)]


def check_mul_matches_bigint_at_every_magnitude(fe_op, pt_op):
    rng = random.Random(11)
    for it in range(120):
        ma, mb = rng.choice([(1, 1), (15, 15), (15, 1), (2, 8), (6, 6), (10, 3)])
        a = [rand_limbs(rng, ma) for _ in range(4)]
        b = [rand_limbs(rng, mb) for _ in range(4)]
        out = fe_op(0, a, b)
        for r in range(4):
            assert value(out[r, :10]) % P == value(a[r]) * value(b[r]) % P, (it, r)
            assert all(int(v) < 2**26 + 2**16 for v in out[r, :10])
            assert not out[r, 10:].any()


def check_weak_normalise_and_negate(fe_op, pt_op):
    rng = random.Random(12)
    for it in range(60):
        a = [[rng.choice([2**32 - 1, rng.randrange(2**32), 0]) for _ in range(10)] for _ in range(4)]
        out = fe_op(1, a)
        for r in range(4):
            assert value(out[r, :10]) % P == value(a[r]) % P
            assert all(int(v) <= U for v in out[r, :10]) and not out[r, 10:].any()
        for op, m in ((2, 1), (3, 2), (4, 8)):
            a = [rand_limbs(rng, m) for _ in range(4)]
            b = [rand_limbs(rng, 1) for _ in range(4)]
            out = fe_op(op, a, b)
            for r in range(4):
                assert value(out[r, :10]) % P == (value(b[r]) - value(a[r])) % P
                assert all(int(v) <= (m + 2) * U for v in out[r, :10]) and not out[r, 10:].any()


def check_gather_scatter_and_is_zero(fe_op, pt_op):
    rng = random.Random(13)
    a = [rand_limbs(rng, 3) for _ in range(4)]
    out = fe_op(6, a)
    assert (out[:, :10] == np.array(a, dtype=np.uint32)).all() and not out[:, 10:].any()
    zs = [limbs(0), limbs(P), None, limbs(5)]
    zs[2] = [2 * v for v in limbs(P)]  # 2p in unreduced limbs
    out = fe_op(7, zs)
    assert [int(out[r, 0]) for r in range(4)] == [1, 1, 1, 0]


def check_zero_filter_never_misses(fe_op, pt_op):
    """wfe_z_maybe_zero on a product (what the point additions test Z3 = 2·Z1·H with): a product ≡ 0 (mod p) is flagged
    whatever the representation of its zero factor (magnitudes up to 15: the lazily reduced H), and almost nothing else is."""
    rng = random.Random(16)
    s = sum(1 << (26 * i) for i in range(10))

    def noncanon(x, hi):
        """another limb pattern of the same integer: borrow 2^26 from limb i + 1 into limb i here and there (limbs ≤ hi)"""
        l = limbs(x)
        for i in range(9):
            if l[i + 1] >= 1 and l[i] + 2**26 <= hi and rng.randrange(2):
                l[i] += 2**26
                l[i + 1] -= 1
        return l

    def zero_rep(m):
        """limbs ≤ (m + 1)·U + 2^26 of a value ≡ 0 (mod p)"""
        mode = rng.randrange(4)
        if mode == 0:
            return limbs(0)
        if mode == 1:
            j = rng.randrange(1, m + 1)
            return [j * v for v in limbs(P)]                                   # j·p, unreduced limbs
        if mode == 2:                                                          # u2 + (K_m − x1), u2 ≡ x1: the additions' H
            x = rng.randrange(P)
            x1 = noncanon(x, m * U)
            u2 = noncanon(x + P if x + P < 2**260 and rng.randrange(2) else x, U + 2**16)
            k = [m * U + d for d in limbs((-m * U * s) % P)]
            return [a + (b - c) for a, b, c in zip(u2, k, x1)]
        return limbs(rng.randrange(1, 16) * P)                                  # m·p < 2^260, canonical limbs
    for it in range(150):
        m = rng.choice([1, 2, 3, 8, 12])
        za = [zero_rep(m) for _ in range(4)]
        assert all(value(z) % P == 0 and all(0 <= v <= 15 * U for v in z) for z in za)
        other = [rand_limbs(rng, rng.choice([1, 2, 6, 15])) for _ in range(4)]
        for a, b in ((za, other), (other, za)):
            out = fe_op(8, a, b)
            assert (out[:, :16] == 1).all(), (it, m)
    flagged = 0
    for it in range(100):
        a = [limbs(rng.randrange(1, P)) for _ in range(4)]
        b = [limbs(rng.randrange(1, P)) for _ in range(4)]
        out = fe_op(8, a, b)
        for r in range(4):
            assert len(set(int(v) for v in out[r, :16])) == 1   # row-uniform
            flagged += int(out[r, 0])
    assert flagged <= 2   # expected 400 · 2.3e-4


def check_sqrt_chain(fe_op, pt_op):
    rng = random.Random(14)
    xs = [rng.randrange(P) for _ in range(4)]
    out = fe_op(5, [limbs(x) for x in xs])
    for r in range(4):
        assert value(out[r, :10]) % P == pow(xs[r], (P + 1) // 4, P)


# ---- points ------------------------------------------------------------------------------------
def jac_rows(pts):
    """pts: list of 4 (X, Y, Z, inf) big-int Jacobian coordinates → [4][31] u32"""
    a = np.zeros((4, 31), dtype=np.uint32)
    for r, (x, y, z, inf) in enumerate(pts):
        a[r, 0:10], a[r, 10:20], a[r, 20:30], a[r, 30] = limbs(x), limbs(y), limbs(z), 1 if inf else 0
    return a


def to_affine(row):
    if row[30]:
        return None
    x, y, z = value(row[0:10]) % P, value(row[10:20]) % P, value(row[20:30]) % P
    zi = pow(z, -1, P)
    return (x * zi * zi % P, y * zi * zi * zi % P)


def rand_jac(rng, k=None):
    k = k or rng.randrange(1, pyref.N)
    x, y = pyref.pt_mul(k, pyref.G)
    z = rng.randrange(1, P)
    return (x * z * z % P, y * z * z * z % P, z, False), (x, y)


def check_point_double_add_madd(fe_op, pt_op):
    rng = random.Random(15)
    ps, pa = zip(*[rand_jac(rng) for _ in range(4)])
    qs, qa = zip(*[rand_jac(rng) for _ in range(4)])
    P4, Q4 = jac_rows(ps), jac_rows(qs)
    out = pt_op(0, P4, Q4)
    for r in range(4):
        assert to_affine(out[r]) == pyref.pt_add(pa[r], pa[r])
    out = pt_op(1, P4, Q4)
    for r in range(4):
        assert to_affine(out[r]) == pyref.pt_add(pa[r], qa[r])
    QA = jac_rows([(x, y, 1, False) for (x, y) in qa])
    out = pt_op(2, P4, QA)
    for r in range(4):
        assert to_affine(out[r]) == pyref.pt_add(pa[r], qa[r])
    out = pt_op(3, P4, Q4)  # rows joined pairwise by the lane-xor
    for r in range(4):
        assert to_affine(out[r]) == pyref.pt_add(pa[r], pa[r ^ 1])


def check_point_exceptional_cases_mixed_over_rows(fe_op, pt_op):
    """row 0: P + P, row 1: P + (−P), row 2: ∞ + Q, row 3: ordinary — in ONE wavefront"""
    rng = random.Random(16)
    (p0, a0), (p1, a1), (q2, b2), (p3, a3) = [rand_jac(rng) for _ in range(4)]
    _, b3 = rand_jac(rng)
    z = rng.randrange(1, P)
    same0 = (a0[0] * z * z % P, a0[1] * z * z * z % P, z, False)       # same point, other Z
    neg1 = (a1[0] * z * z % P, (P - a1[1]) * z * z * z % P, z, False)  # −P, other Z
    q3 = (b3[0], b3[1], 1, False)
    P4 = jac_rows([p0, p1, (0, 0, 0, True), p3])
    Q4 = jac_rows([same0, neg1, q2, q3])
    out = pt_op(1, P4, Q4)
    assert to_affine(out[0]) == pyref.pt_add(a0, a0)
    assert to_affine(out[1]) is None
    assert to_affine(out[2]) == b2
    assert to_affine(out[3]) == pyref.pt_add(a3, b3)
    # mixed addition: the affine operand equals P / −P / meets ∞
    QA = jac_rows([(a0[0], a0[1], 1, False), (a1[0], P - a1[1], 1, False), (b2[0], b2[1], 1, False), q3])
    out = pt_op(2, P4, QA)
    assert to_affine(out[0]) == pyref.pt_add(a0, a0)
    assert to_affine(out[1]) is None
    assert to_affine(out[2]) == b2
    assert to_affine(out[3]) == pyref.pt_add(a3, b3)


# ---- full recover ------------------------------------------------------------------------------
def check_full_recover_matches_oracle(recover, oracle, rounds=6):
    rng = random.Random(17)
    for it in range(rounds):
        sk = rng.randrange(1, pyref.N).to_bytes(32, "big")
        h = rng.randrange(2**256).to_bytes(32, "big")
        sig = oracle.sign(sk, h)
        want = oracle.recover_address(h, sig)
        ok, addr = recover(h, sig)
        assert ok and addr == want, it
        assert addr == oracle.address(oracle.pubkey(sk))
    # rejected inputs: r = 0, s ≥ n, v = 2, x with no square root
    bad = bytearray(sig)
    bad[0:32] = bytes(32)
    assert recover(h, bytes(bad))[0] is False
    bad = bytearray(sig)
    bad[64] = 2
    assert recover(h, bytes(bad))[0] is False
    bad = bytearray(sig)
    bad[0:32] = (5).to_bytes(32, "big")  # x = 5: 5³+7 = 132 is not a square mod p
    assert pow(132, (P - 1) // 2, P) != 1
    assert recover(h, bytes(bad))[0] is False
    assert oracle.recover_address(h, bytes(bad)) is None
