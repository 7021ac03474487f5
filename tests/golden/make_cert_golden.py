"""Generate tests/golden/cert_trees.json: what ibft_verify_certificates_wire must answer for 2 000 batches of
certificate-carrying messages (the hand-made cases of tests/cert_cases.py, then byte-level fuzz of canonical ROUND_CHANGE /
PREPREPARE messages), as one 64-bit digest per batch over tree shape, classes, verdict bits and parsed fields
(tests/cert_cases.py: digest_expected).  The expectation comes from oracle/wire_cert.py AND is cross-checked here against the
google.protobuf runtime, message by message:

  * a row the oracle calls judged (canonical, everything below it canonical) must parse, carry no unknown fields anywhere in
    its tree and re-serialise to the very same bytes; its PayloadNoSig digest must be keccak of the runtime's serialisation
    with `signature` cleared;
  * a row the runtime finds canonical in that sense (and within the device's column limits: proposal hashes ≤ 32 bytes,
    type ≤ 255, everywhere in its tree) must be judged by the oracle.

Run from the repo root:  python tests/golden/make_cert_golden.py
"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cert_cases as CC  # noqa: E402
from oracle import binding as B  # noqa: E402
from oracle import wire_cert as WC  # noqa: E402

COUNT = 2000


def pb_tree_facts(Msg, data: bytes):
    """(canonical?, within the device's limits?) of one message by the protobuf runtime, its whole tree included"""
    from google.protobuf import unknown_fields
    g = Msg()
    try:
        g.ParseFromString(data)
    except Exception:
        return False, True, None
    if g.SerializeToString(deterministic=True) != data:
        return False, True, g

    def clean(m):  # no unknown fields anywhere, limits respected
        ok, lim = len(unknown_fields.UnknownFieldSet(m)) == 0, True
        for fd, val in m.ListFields():
            if fd.name == "type" and val > 255:
                lim = False
            if fd.name == "proposalHash" and len(val) > 32:
                lim = False
            if fd.type == fd.TYPE_MESSAGE:
                for sub in (val if getattr(fd, 'is_repeated', None) or hasattr(val, '__len__') and not hasattr(val, 'ListFields') else [val]):
                    o, l = clean(sub)
                    ok, lim = ok and o, lim and l
        return ok, lim
    ok, lim = clean(g)
    return ok, lim, g


def main():
    spec = importlib.util.spec_from_file_location("mwf", os.path.join(HERE, "make_wire_fixtures.py"))
    mwf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mwf)
    Msg = mwf.build_messages()["IbftMessage"]
    r, batches = CC.golden_batches(COUNT)
    digests, stats = [], {"rows": 0, "judged": 0, "needs_host": 0, "sender_ok": 0, "pb_checked": 0}
    for bi, msgs in enumerate(batches):
        exp = WC.expected_tree(msgs, r.addrs)
        for k in range(exp.n_rows):
            nd = exp.nodes[k]
            data = exp.wire[nd["off"]:nd["off"] + nd["len"]]
            canonical, within, g = pb_tree_facts(Msg, data)
            judged = exp.status[k] == WC.OK
            if judged:
                assert canonical, (bi, k, "oracle judges a row the protobuf runtime does not find canonical")
                ns = Msg()
                ns.CopyFrom(g)
                ns.ClearField("signature")
                if exp.digest[k] is not None:
                    assert B.keccak256(ns.SerializeToString(deterministic=True)) == exp.digest[k], (bi, k, "PayloadNoSig digest")
            elif canonical and within:
                raise AssertionError((bi, k, "the protobuf runtime finds the row canonical, the oracle refuses it"))
            stats["pb_checked"] += 1
        stats["rows"] += exp.n_rows
        stats["judged"] += sum(1 for s in exp.status if s == WC.OK)
        stats["needs_host"] += sum(1 for c in exp.cls if c & WC.CLASS_NEEDS_HOST)
        stats["sender_ok"] += sum(exp.sender_ok)
        digests.append(CC.digest_expected(exp))
    out = {"count": COUNT, "seed": CC.GOLDEN_SEED, "validators": CC.GOLDEN_VALIDATORS, "round_seed": CC.GOLDEN_ROUND_SEED, "stats": stats,
           "digests": digests}
    with open(os.path.join(HERE, "cert_trees.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(stats)


if __name__ == "__main__":
    main()
