"""The wavefront-cooperative device code of the certificate path (csrc/cert_wave_dev.h: Keccak-256 by 25 lanes with the state words
meeting in LDS; the walk over a certificate with its proven runs of equal-sized messages), compiled for the host and run on the
64-coroutine lockstep wavefront emulator (csrc/wave_emul.h) — the exact source the gfx950 kernels wrap — against the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

import cert_cases as CC
from go_ibft_amd import build as B
from oracle import binding as OB
from oracle import wire
from oracle import wire_cert as WC
from oracle import workload as W


@pytest.fixture(scope="module")
def dev():
    L = C.CDLL(B.build_cert_wave_harness())
    L.cwh_walk.restype = C.c_int64
    return L


def _aligned(data: bytes, lead: int = 0):
    """data at a 16-byte aligned address + lead, 32 bytes of slack behind it"""
    raw = np.zeros(len(data) + 64 + lead, dtype=np.uint8)
    base = (-raw.ctypes.data) % 16 + lead
    raw[base:base + len(data)] = np.frombuffer(data, dtype=np.uint8)
    return raw, base


def keccak_without(dev, m: bytes, cut0: int, cut1: int, lead: int = 0) -> bytes:
    raw, base = _aligned(m, lead)
    out = np.zeros(32, dtype=np.uint8)
    dev.cwh_keccak_without(C.c_void_p(raw.ctypes.data + base), len(m), cut0, cut1, out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def test_sponge_lengths_cuts_and_alignments(dev):
    rng = random.Random(5)
    lens = [0, 1, 7, 8, 9, 66, 67, 68, 135, 136, 137, 203, 271, 272, 273, 407, 408, 1000, 1088, 1089, 4093, 25000]
    for n in lens:
        m = bytes(rng.randrange(256) for _ in range(n))
        for _ in range(4):
            cut0 = rng.randrange(0, n + 1)
            cut1 = rng.randrange(cut0, min(n, cut0 + 70) + 1)
            lead = rng.randrange(4)
            assert keccak_without(dev, m, cut0, cut1, lead) == OB.keccak256(m[:cut0] + m[cut1:]), (n, cut0, cut1, lead)
        assert keccak_without(dev, m, n, n) == OB.keccak256(m)  # nothing cut out
    # known answers: Keccak-256 of "" and of "abc"
    assert keccak_without(dev, b"", 0, 0).hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak_without(dev, b"abc", 3, 3).hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_sponge_proposal_hash(dev):
    rng = random.Random(6)
    for n in [0, 1, 127, 128, 129, 135, 136, 264, 1024, 5000]:
        raw = bytes(rng.randrange(256) for _ in range(n))
        for rnd in (0, 1, 2**40 + 7, 2**64 - 1):
            buf, base = _aligned(raw, rng.randrange(4))
            out = np.zeros(32, dtype=np.uint8)
            dev.cwh_keccak_proposal(C.c_void_p(buf.ctypes.data + base), n, C.c_uint64(rnd), out.ctypes.data_as(C.c_void_p))
            assert out.tobytes() == OB.proposal_hash(raw, rnd), (n, rnd)


def test_sponge_on_payload_no_sig_of_real_messages(dev):
    r = W.make_round(8, 811, height=5, round_=1)
    for m in CC.honest_round_change_set(r)[:3] + [CC.preprepare_with_rcc(r)]:
        data = m.encode()
        exp = WC.own(data)
        assert keccak_without(dev, data, exp.cut0, exp.cut1) == OB.keccak256(m.payload_no_sig())


def walk(dev, cert: bytes, pc: bool, lead: bytes = b"", cap: int = 8192):
    """the certificate `cert` somewhere inside a buffer (after `lead`), as the device sees it"""
    raw, base = _aligned(lead + cert)
    rec = np.zeros(3 * cap, dtype=np.uint32)
    n = dev.cwh_walk(C.c_void_p(raw.ctypes.data + base), len(lead), len(lead) + len(cert), 1 if pc else 0, rec.ctypes.data_as(C.c_void_p), cap)
    return int(n), rec.reshape(-1, 3)[:max(n, 0)]


def expected_children(cert: bytes, pc: bool, at: int):
    """sequential restatement: one-byte tags, minimal length varints of at most 5 bytes (the canonical form the device insists on),
    known fields in order, proposalMessage at most once; None = refused"""
    out, last, pos = [], 0, 0
    while pos < len(cert):
        tag = cert[pos]
        pos += 1
        f, wt = tag >> 3, tag & 7
        if tag & 0x80 or wt != 2 or f == 0 or f < last or f > (2 if pc else 1) or (pc and f == 1 and last == 1):
            return None
        last = f
        ln = 0
        for i in range(6):
            if i == 5 or pos >= len(cert):
                return None
            b = cert[pos]
            pos += 1
            ln |= (b & 0x7F) << (7 * i)
            if not b & 0x80:
                if i > 0 and b == 0:
                    return None
                break
        if ln > len(cert) - pos:
            return None
        out.append((at + pos, ln, (WC.ROLE_PC_PROPOSAL if f == 1 else WC.ROLE_PC_PREPARE) if pc else WC.ROLE_RCC_MESSAGE))
        pos += ln
    return out


def check_walk(dev, cert, pc, lead=b"x" * 37):
    exp = expected_children(cert, pc, len(lead))
    n, rec = walk(dev, cert, pc, lead)
    if exp is None:
        assert n == -1
        return None
    assert n == len(exp) and [tuple(int(x) for x in row) for row in rec] == exp
    return exp


def test_walk_uniform_runs_windows_and_seams(dev):
    r = W.make_round(300, 816, height=5, round_=1, raw_len=40)
    others = [i for i in range(300) if i != 1]
    for cnt in (0, 1, 2, 63, 64, 65, 127, 128, 129, 299):
        pc = CC.pc_bytes(r, 5, 1, 1, others[:cnt])          # 1 proposal message + cnt PREPAREs of one size: long runs
        for lead in (b"", b"y" * 5, b"z" * 16381):           # certificate start near the end of a 16 KiB window
            exp = check_walk(dev, pc, True, lead)
            assert exp is not None and len(exp) == cnt + 1
    # an RCC of round-change messages of one size, longer than a window each
    rcs = CC.honest_round_change_set(r, senders=list(range(5)))
    exp = check_walk(dev, wire.round_change_certificate(rcs), False)
    assert len(exp) == 5 and exp[0][1] > 16384


def test_walk_mixed_sizes_and_breaks_in_runs(dev):
    """runs interrupted by a message of another size, of another field, by a malformed header — at every position of the first 70"""
    r = W.make_round(80, 817, height=5, round_=1, raw_len=16)
    prepares = [CC.prepare(r, j, 5, 1).encode() for j in range(2, 78)]
    pm = CC.preprepare(r, 1, 5, 1).encode()
    base = [wire._len_field(1, pm, True)] + [wire._len_field(2, p, True) for p in prepares]
    assert check_walk(dev, b"".join(base), True) is not None
    odd = wire._len_field(2, prepares[0] + b"", True)[:-1]                     # one byte short: every later header is off by one
    longer = wire._len_field(2, CC.prepare(r, 79, 5, 7 ** 9).encode(), True)   # another length (a big round number)
    for k in list(range(1, 70, 3)) + [64, 65, 66]:
        for repl in (longer, wire._len_field(1, pm, True), b"\x10\x05", b"\x12\x80\x00", odd):
            parts = list(base)
            parts[k] = repl
            cert = b"".join(parts)
            exp = check_walk(dev, cert, True, b"q" * 11)
            if repl in (b"\x10\x05", b"\x12\x80\x00", odd):
                assert exp is None, (k, repl[:4])


def test_walk_fuzzed_certificates(dev):
    rng = random.Random(99)
    r = W.make_round(40, 818, height=5, round_=1, raw_len=24)
    pc = CC.pc_bytes(r, 5, 1, 1, [i for i in range(40) if i != 1])
    stats = {"ok": 0, "bad": 0}
    for _ in range(300):
        b = bytearray(pc)
        for _ in range(rng.choice([1, 1, 2])):
            op, pos = rng.randrange(3), rng.randrange(len(b))
            if op == 0:
                b[pos] ^= 1 << rng.randrange(8)
            elif op == 1:
                b.insert(pos, rng.randrange(256))
            else:
                del b[pos]
        cert = bytes(b)
        exp = check_walk(dev, cert, True, b"w" * 3)
        stats["ok" if exp is not None else "bad"] += 1
    assert stats["ok"] > 40 and stats["bad"] > 40, stats


def test_sponge_public_keccak_vectors(dev):
    """the third-party Keccak-256 known answers of tests/golden/kats.json through the device sponge source"""
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))
    for v in k["public_keccak_vectors"]:
        m = v["message"].encode()
        assert keccak_without(dev, m, len(m), len(m)).hex().startswith(v.get("digest", v.get("digest_prefix"))), v["source"]
