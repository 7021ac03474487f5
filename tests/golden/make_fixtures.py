"""Generate the committed golden fixtures under tests/golden/.

The reference (go-ibft) pins no numerics for this path and cannot be executed here
(no Go toolchain), so the fixtures are produced by the CPU oracle (oracle/*.c) after it
has been pinned by the public KATs / pyref / OpenSSL cross-checks, and every expected
verdict is ALSO recomputed with the independent pure-Python derivation (oracle/pyref.py)
before it is written.  Run from the repo root:  python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as B, pyref as R, workload as W  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def seals_fixture(name, n, seed, **kw):
    r = W.make_round(n, seed, with_envelopes=True, **kw)
    vs = B.ValSet(r.addrs, r.power)
    verdict = B.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags)
    senders = B.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20)
    hashes = B.verify_hashes(r.raw, r.round, r.hash32, r.hash_len)
    t = B.tally(vs, r.signer20, verdict)
    # independent re-derivation of every verdict with pure-Python big ints
    members = {bytes(a) for a in r.addrs}
    for i in range(n):
        exp = 0
        if not r.pre_flags[i]:
            a = R.recover_address(r.hash32[i].tobytes(), r.seal65[i].tobytes())
            exp = int(a is not None and a == r.signer20[i].tobytes() and a in members)
        assert exp == verdict[i], (name, i, r.kinds[i])
        pl = r.payload[r.off[i]:r.off[i + 1]]
        a = R.recover_address(R.keccak256(pl), r.msg_sig65[i].tobytes())
        assert int(a is not None and a == r.signer20[i].tobytes() and a in members) == senders[i]
        assert int(r.hash_len[i] == 32 and r.hash32[i].tobytes() == R.proposal_hash(r.raw, r.round)) == hashes[i]
    power = sum(int(r.power[i]) for i in range(n) if verdict[i])
    assert power == t.power and R.calculate_quorum(int(r.power.sum())) == t.quorum
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), raw=np.frombuffer(r.raw, dtype=np.uint8), round=np.uint64(r.round),
        height=np.uint64(r.height), addrs=r.addrs, power=r.power, hash32=r.hash32, hash_len=r.hash_len,
        seal65=r.seal65, signer20=r.signer20, pre_flags=r.pre_flags,
        payload=np.frombuffer(r.payload, dtype=np.uint8), off=r.off, msg_sig65=r.msg_sig65,
        proposal_hash=np.frombuffer(r.proposal_hash, dtype=np.uint8),
        exp_seals=verdict, exp_senders=senders, exp_hashes=hashes,
        exp_power=np.array([t.power & (2**64 - 1), t.power >> 64], dtype=np.uint64),
        exp_quorum=np.array([t.quorum & (2**64 - 1), t.quorum >> 64], dtype=np.uint64),
        exp_has_quorum=np.uint32(t.has_quorum), exp_distinct=np.uint32(t.distinct_senders))
    print(name, "rows", n, "valid", int(verdict.sum()), "quorum", t.has_quorum)


def bench_fixture(name, n, seed):
    """Honest COMMIT round at the bench size (inputs only; the GPU result is checked
    against the oracle at run time by tests, not by bench.py)."""
    r = W.make_round(n, seed)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), addrs=r.addrs, power=r.power, hash32=r.hash32,
                        seal65=r.seal65, signer20=r.signer20)
    print(name, "rows", n)


def bench_round_fixture(name, n, seed):
    """BASELINE config #3 inputs: one whole height at N validators — the honest COMMIT round (seals +
    message envelopes) and the PREPARE envelopes of everyone but the proposer (validator 0).  Inputs only
    (so that bench.py signs nothing at run time); the expected verdict is all-true by construction and is
    checked against the oracle here, once."""
    from oracle import wire as WR
    r = W.make_round(n, seed, with_envelopes=True)
    vs = B.ValSet(r.addrs, r.power)
    assert B.verify_seals(vs, r.hash32, r.seal65, r.signer20, nthreads=8).all()
    assert B.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20).all()
    pp, poff, psig = [], [0], np.zeros((n - 1, 65), np.uint8)
    for i in range(1, n):
        m = WR.IbftMessage(view=WR.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=WR.PREPARE,
                           payload=WR.prepare_body(r.proposal_hash))
        pns = m.payload_no_sig()
        pp.append(pns)
        poff.append(poff[-1] + len(pns))
        psig[i - 1] = np.frombuffer(B.sign(r.sks[i], B.keccak256(pns)), np.uint8)
    ppayload, poff = b"".join(pp), np.array(poff, np.uint32)
    assert B.verify_senders(vs, ppayload, poff, psig, r.addrs[1:]).all()
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), seed=np.uint64(seed), raw=np.frombuffer(r.raw, dtype=np.uint8),
        round=np.uint64(r.round), height=np.uint64(r.height), addrs=r.addrs, power=r.power, hash32=r.hash32,
        seal65=r.seal65, signer20=r.signer20, payload=np.frombuffer(r.payload, dtype=np.uint8), off=r.off,
        msg_sig65=r.msg_sig65, prepare_payload=np.frombuffer(ppayload, dtype=np.uint8), prepare_off=poff,
        prepare_sig65=psig, proposal_hash=np.frombuffer(r.proposal_hash, dtype=np.uint8))
    print(name, "rows", n)


def kat_fixture():
    sk1 = (1).to_bytes(32, "big")
    d = B.keccak256(b"go-ibft golden")
    kats = {
        "keccak256_empty": B.keccak256(b"").hex(),
        "keccak256_abc": B.keccak256(b"abc").hex(),
        "sk1_pub": B.pubkey(sk1).hex(),
        "sk1_addr": B.address(B.pubkey(sk1)).hex(),
        "digest": d.hex(),
        "sk1_sig": B.sign(sk1, d).hex(),
    }
    assert kats["keccak256_empty"] == R.keccak256(b"").hex()
    assert kats["sk1_addr"] == R.address(R.pubkey(1)).hex()
    json.dump(kats, open(os.path.join(HERE, "kats.json"), "w"), indent=1)


if __name__ == "__main__":
    kat_fixture()
    seals_fixture("round_n64_honest", 64, 1)
    seals_fixture("round_n100_byz_weighted", 100, 2, byzantine=True, weighted=True)
    seals_fixture("round_n256_byz", 256, 3, byzantine=True)
    bench_fixture("bench_commit_n1024", 1024, 1)
    bench_round_fixture("bench_round_n4096", 4096, 3)
