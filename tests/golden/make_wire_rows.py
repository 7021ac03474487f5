"""Generate tests/golden/wire_rows.json: IbftMessage wire rows (canonical PREPARE/COMMIT/PREPREPARE/
ROUND_CHANGE of a small round, hand-made non-canonical encodings, fuzzed rows) with what the device
wire walker must answer for each.  The expectation comes from oracle/wire_parse.py AND is cross-checked
here against the google.protobuf runtime: a row is written as OK only if protobuf parses it,
re-serialises it to the same bytes, and its signature-less serialisation hashes to the expected
digest.  Run from the repo root:  python tests/golden/make_wire_rows.py
"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import wire_cases as WCASE  # noqa: E402
from oracle import binding as B  # noqa: E402
from oracle import wire_parse as WP  # noqa: E402
from oracle import workload as W  # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location("mwf", os.path.join(HERE, "make_wire_fixtures.py"))
    mwf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mwf)
    Msg = mwf.build_messages()["IbftMessage"]
    r = W.make_round(12, 9001, height=5, round_=1, byzantine=True)
    rows = [("canonical " + k, m) for k, m in zip(["commit", "prepare", "preprepare", "roundchange"] * 3,
                                                  WCASE.canonical_round(r, ("commit", "prepare", "preprepare", "roundchange")))]
    rows += WCASE.handmade(r)
    rows += [("fuzz %d" % i, m) for i, m in enumerate(WCASE.fuzz(WCASE.canonical_round(r), 150, 77))]
    out = []
    for label, m in rows:
        e = WP.expected(m)
        if e.status == WP.OK:
            g = Msg()
            g.ParseFromString(m)
            assert g.SerializeToString(deterministic=True) == m, label
            g.signature = b""
            assert B.keccak256(g.SerializeToString(deterministic=True)) == e.digest, label
        out.append({"label": label, "wire": m.hex(), "status": e.status, "pre_flag": int(e.pre_flag),
                    "digest": e.digest.hex() if e.status == WP.OK else "", "height": e.height, "round": e.round,
                    "type": e.type, "payload_kind": e.payload_kind, "has_view": e.has_view,
                    "from": e.sender.hex(), "signature": e.signature.hex(),
                    "proposal_hash": e.proposal_hash.hex(), "committed_seal": e.committed_seal.hex()})
    with open(os.path.join(HERE, "wire_rows.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out), "rows,", sum(1 for o in out if o["status"] == 0), "OK")


if __name__ == "__main__":
    main()
