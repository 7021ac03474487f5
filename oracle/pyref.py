"""oracle/pyref.py — independent pure-Python re-derivation of the hot-path numerics.

TEST INFRASTRUCTURE ONLY (second, independent oracle used to pin oracle/*.c; the
reference pins none of this arithmetic — see oracle/ibft_oracle.h "PARITY STATUS").

Deliberately written differently from the C restatement: Python ``int`` big
numbers, affine coordinates, ``pow(x, -1, m)`` inverses, plain double-and-add.
Only suitable for small cases (≈5 ms per recover).

Conventions: Keccak-256 with 0x01 padding; proposal hash =
keccak256(raw ‖ BE64(round)); signature = r‖s‖v (65 B, v∈{0,1});
address = keccak256(X‖Y)[12:].
"""
from __future__ import annotations

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)

_MASK = (1 << 64) - 1


def _rol(v: int, n: int) -> int:
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _MASK if n else v


def _keccak_f(A: list[list[int]]) -> None:
    """A[x][y] lanes; FIPS-202 §3.2 step mappings written from the spec text."""
    # round constants from the LFSR of §3.2.5
    R = 1
    for _ in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        for x in range(5):
            for y in range(5):
                A[x][y] ^= D[x]
        # rho and pi, walking the (x,y) -> (y, 2x+3y) orbit
        x, y = 1, 0
        cur = A[x][y]
        for t in range(24):
            x, y = y, (2 * x + 3 * y) % 5
            cur, A[x][y] = A[x][y], _rol(cur, (t + 1) * (t + 2) // 2)
        for y in range(5):
            row = [A[x][y] for x in range(5)]
            for x in range(5):
                A[x][y] = row[x] ^ ((~row[(x + 1) % 5]) & _MASK & row[(x + 2) % 5])
        # iota
        for j in range(7):
            R = ((R << 1) ^ ((R >> 7) * 0x71)) % 256
            if R & 2:
                A[0][0] ^= 1 << ((1 << j) - 1)


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        _keccak_f(A)
    out = b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


# ---- secp256k1, affine, None = infinity ---------------------------------------

def pt_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def pt_mul(k: int, pt):
    acc = None
    while k:
        if k & 1:
            acc = pt_add(acc, pt)
        pt = pt_add(pt, pt)
        k >>= 1
    return acc


def pubkey(sk: int):
    return pt_mul(sk % N, G)


def pub_bytes(pt) -> bytes:
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def address(pt) -> bytes:
    return keccak256(pub_bytes(pt))[12:]


def sign(sk: int, digest: bytes, k: int) -> bytes:
    """Plain ECDSA with caller-supplied nonce; low-s normalised; 65-byte r‖s‖v."""
    z = int.from_bytes(digest, "big") % N
    R = pt_mul(k, G)
    r = R[0] % N
    assert r != 0 and R[0] < N
    s = pow(k, -1, N) * (z + r * sk) % N
    assert s != 0
    v = R[1] & 1
    if s > N // 2:
        s, v = N - s, v ^ 1
    return r.to_bytes(32, "big") + s.to_bytes(32, "big") + bytes([v])


def ecrecover(digest: bytes, sig: bytes, strict_low_s: bool = False):
    """Returns the affine public key or None (SEC 1 v2 §4.1.6)."""
    if len(sig) != 65:
        return None
    r = int.from_bytes(sig[:32], "big")
    s = int.from_bytes(sig[32:64], "big")
    v = sig[64]
    if v > 1 or not (0 < r < N) or not (0 < s < N):
        return None
    if strict_low_s and s > N // 2:
        return None
    rhs = (r * r * r + 7) % P
    y = pow(rhs, (P + 1) // 4, P)
    if y * y % P != rhs:
        return None
    if (y & 1) != v:
        y = P - y
    z = int.from_bytes(digest, "big") % N
    rinv = pow(r, -1, N)
    u1 = (-z * rinv) % N
    u2 = (s * rinv) % N
    return pt_add(pt_mul(u1, G), pt_mul(u2, (r, y)))


def recover_address(digest: bytes, sig: bytes, strict_low_s: bool = False):
    q = ecrecover(digest, sig, strict_low_s)
    return None if q is None else address(q)


def proposal_hash(raw: bytes, round_: int) -> bytes:
    return keccak256(raw + round_.to_bytes(8, "big"))


def calculate_quorum(total: int) -> int:
    """/root/reference/core/validator_manager.go:130-135."""
    return 2 * total // 3 + 1
