"""Wire-byte rows for the §8f rank 3 tests: canonical PREPARE/COMMIT/PREPREPARE/ROUND_CHANGE
messages of a round, hand-made non-canonical encodings, and byte-level fuzz of canonical ones.
Test infrastructure only (uses the oracle's encoder and signer)."""
import random

import numpy as np

from oracle import binding as B
from oracle import wire


def signed(m: wire.IbftMessage, sk: bytes) -> bytes:
    m.signature = B.sign(sk, B.keccak256(m.payload_no_sig()))
    return m.encode()


def canonical_round(r, kinds=("commit", "prepare")):
    """one signed message per validator of oracle.workload round r, cycling through kinds"""
    rows = []
    for i in range(r.n):
        kind = kinds[i % len(kinds)]
        h = r.hash32[i].tobytes()
        if kind == "commit":
            m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.COMMIT,
                                 payload=wire.commit_body(h, r.seal65[i].tobytes()))
        elif kind == "prepare":
            m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.PREPARE,
                                 payload=wire.prepare_body(h))
        elif kind == "preprepare":
            m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=wire.PREPREPARE,
                                 payload=wire.preprepare_body(wire.Proposal(r.raw, r.round), h, None))
        else:
            m = wire.IbftMessage(view=wire.View(r.height, r.round), sender=r.addrs[i].tobytes(),
                                 type=wire.ROUND_CHANGE, payload=wire.round_change_body(None, None))
        rows.append(signed(m, r.sks[i]))
    return rows


def handmade(r):
    """(label, bytes): encodings the canonical walk must refuse or flag, and odd-but-canonical ones"""
    sk, addr, h = r.sks[0], r.addrs[0].tobytes(), r.hash32[0].tobytes()
    seal = r.seal65[0].tobytes()

    def msg(**kw):
        d = dict(view=wire.View(7, 3), sender=addr, type=wire.COMMIT, payload=wire.commit_body(h, seal))
        d.update(kw)
        return wire.IbftMessage(**d)
    good = signed(msg(), sk)
    sig_field = wire._len_field(3, B.sign(sk, b"\x11" * 32))
    body = wire._len_field(7, wire.commit_body(h, seal), emit_empty=True)
    out = [
        ("canonical commit", good),
        ("no view", signed(msg(view=None), sk)),
        ("empty view", signed(msg(view=wire.View(0, 0)), sk)),
        ("round only", signed(msg(view=wire.View(0, 9)), sk)),
        ("type 0 + prepare payload", wire.IbftMessage(view=wire.View(1, 0), sender=addr, type=0,
                                                      signature=b"\x01" * 65).encode() + wire._len_field(6, wire.prepare_body(h), True)),
        ("no payload", signed(msg(payload=None), sk)),
        ("empty payload", signed(msg(payload=b""), sk)),
        ("no signature", msg().encode()),
        ("64-byte signature", msg(signature=b"\x05" * 64).encode()),
        ("21-byte from", signed(msg(sender=addr + b"\x00"), sk)),
        ("300-byte from", signed(msg(sender=addr * 15), sk)),
        ("31-byte hash", signed(msg(payload=wire.commit_body(h[:31], seal)), sk)),
        ("33-byte hash", signed(msg(payload=wire.commit_body(h + b"\x00", seal)), sk)),
        ("66-byte seal", signed(msg(payload=wire.commit_body(h, seal + b"\x01")), sk)),
        ("huge height", signed(msg(view=wire.View(2**64 - 1, 2**63)), sk)),
        ("type 200 with a commit payload", wire._len_field(2, addr) + sig_field + b"\x20\xc8\x01" + body),
        ("type 300", wire._len_field(2, addr) + sig_field + b"\x20\xac\x02" + body),
        ("negative enum (10-byte varint)", wire._len_field(2, addr) + sig_field + b"\x20" + b"\xff" * 9 + b"\x01" + body),
        # non-canonical encodings of the same message
        ("padded varint height", b"\x0a\x03\x08\x87\x00" + wire._len_field(2, addr) + sig_field + b"\x20\x02" + body),
        ("explicit zero type", wire._len_field(2, addr) + sig_field + b"\x20\x00" + body),
        ("explicit zero height", b"\x0a\x02\x08\x00" + wire._len_field(2, addr) + sig_field + b"\x20\x02" + body),
        ("empty from emitted", b"\x12\x00" + sig_field + b"\x20\x02" + body),
        ("fields out of order", sig_field + wire._len_field(2, addr) + b"\x20\x02" + body),
        ("duplicate from", wire._len_field(2, addr) + wire._len_field(2, addr) + sig_field + b"\x20\x02" + body),
        ("two oneof members", wire._len_field(2, addr) + sig_field + b"\x20\x02" +
         wire._len_field(6, wire.prepare_body(h), True) + body),
        ("unknown field 9", good + b"\x48\x01"),
        ("unknown field 15 bytes", good + b"\x7a\x01\x00"),
        ("two-byte tag", good + b"\x80\x01\x01"),
        ("fixed32 for type", wire._len_field(2, addr) + sig_field + b"\x25\x02\x00\x00\x00" + body),
        ("padded length", wire._len_field(2, addr) + b"\x1a\xc1\x00" + B.sign(sk, b"\x11" * 32) + b"\x20\x02" + body),
        ("truncated", good[:-3]),
        ("length past the end", good[:-1] + b"\x7f"),
        ("unknown field in the body", wire._len_field(2, addr) + sig_field + b"\x20\x02" +
         wire._len_field(7, wire.commit_body(h, seal) + b"\x18\x01", True)),
        ("hash twice in the body", wire._len_field(2, addr) + sig_field + b"\x20\x02" +
         wire._len_field(7, wire._len_field(1, h) + wire.commit_body(h, seal), True)),
        ("seal before hash", wire._len_field(2, addr) + sig_field + b"\x20\x02" +
         wire._len_field(7, wire._len_field(2, seal) + wire._len_field(1, h), True)),
        ("empty hash emitted", wire._len_field(2, addr) + sig_field + b"\x20\x01" + wire._len_field(6, b"\x0a\x00", True)),
        ("view with unknown field", b"\x0a\x04\x08\x07\x18\x01" + wire._len_field(2, addr) + sig_field + b"\x20\x02" + body),
        ("view round before height", b"\x0a\x04\x10\x03\x08\x07" + wire._len_field(2, addr) + sig_field + b"\x20\x02" + body),
        ("empty message", b""),
        ("one byte", b"\x0a"),
        ("field 0", b"\x02\x00"),
    ]
    return out


def fuzz(rows, count, seed):
    """byte flips / insertions / deletions / truncations of canonical rows"""
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        b = bytearray(rng.choice(rows))
        for _ in range(rng.choice([1, 1, 1, 2, 3])):
            op = rng.randrange(4)
            pos = rng.randrange(len(b)) if b else 0
            if op == 0 and b:
                b[pos] ^= 1 << rng.randrange(8)
            elif op == 1:
                b.insert(pos, rng.randrange(256))
            elif op == 2 and b:
                del b[pos]
            elif b:
                del b[pos:]
        out.append(bytes(b))
    return out


def pack(rows):
    off = np.zeros(len(rows) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) for x in rows])
    return b"".join(rows), off
