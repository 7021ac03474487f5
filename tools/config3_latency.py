#!/usr/bin/env python3
"""BASELINE config #3 on one MI355X: N = 4096 validators, the full PREPARE + COMMIT sequence a node
runs per height — IsValidValidator on the 4095 PREPAREs and 4096 COMMITs as they arrive (ingest),
IsValidProposalHash over both sets, IsValidCommittedSeal + weighted quorum tally — timed end to end
from host columns to host-visible verdicts (H2D and D2H included), cold (no key cache) and warm.
"*_from_wire": the same sequence starting from the messages' protobuf bytes (§8f rank 3): two
ibft_verify_senders_wire calls (the device walks, hashes and verifies; a1 is a compare on the extracted
proposal hashes), then the COMMIT seals staged from the same upload (ibft_wire_stage_seals) + a2 + tally —
no protobuf decoding, PayloadNoSig re-marshalling or flattening on the host at all.
Writes one JSON object; every result is checked against the CPU oracle first."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import go_ibft_amd.verifier as V  # noqa: E402
from oracle import binding as B, wire as W, workload as WL  # noqa: E402


def main(n=4096, reps=30):
    r = WL.make_round(n, 3, with_envelopes=True)           # COMMIT envelopes + seals
    vs = B.ValSet(r.addrs, r.power)
    # PREPARE envelopes from everyone but the proposer (validator 0)
    pp, poff, psig = [], [0], np.zeros((n - 1, 65), np.uint8)
    for i in range(1, n):
        m = W.IbftMessage(view=W.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=W.PREPARE,
                          payload=W.prepare_body(r.proposal_hash))
        pns = m.payload_no_sig()
        pp.append(pns)
        poff.append(poff[-1] + len(pns))
        psig[i - 1] = np.frombuffer(B.sign(r.sks[i], B.keccak256(pns)), np.uint8)
    ppayload, poff = b"".join(pp), np.array(poff, np.uint32)
    # the same messages as wire bytes (signature field included)
    pw, cw = [], []
    for i in range(1, n):
        m = W.IbftMessage(view=W.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=W.PREPARE,
                          payload=W.prepare_body(r.proposal_hash), signature=psig[i - 1].tobytes())
        pw.append(m.encode())
    for i in range(n):
        m = W.IbftMessage(view=W.View(r.height, r.round), sender=r.addrs[i].tobytes(), type=W.COMMIT,
                          payload=W.commit_body(r.hash32[i].tobytes(), r.seal65[i].tobytes()),
                          signature=r.msg_sig65[i].tobytes())
        cw.append(m.encode())
    pwire, pwoff = b"".join(pw), np.concatenate([[0], np.cumsum([len(x) for x in pw])]).astype(np.uint32)
    cwire, cwoff = b"".join(cw), np.concatenate([[0], np.cumsum([len(x) for x in cw])]).astype(np.uint32)
    H = np.frombuffer(r.proposal_hash, np.uint8)
    pfrom, phash = r.addrs[1:], r.hash32[1:]
    out = {"n_validators": n, "reps": reps}
    for label, flags in (("cold", 0), ("warm", V.FLAG_PUBKEY_CACHE)):
        bv = V.BatchVerifier(flags=flags, max_rows=n)
        bv.set_validators(r.height, r.addrs, r.power)

        def sequence():
            a, _ = bv.is_valid_validator(ppayload, poff, psig, pfrom)                      # PREPARE ingest
            b = bv.is_valid_proposal_hash(r.raw, r.round, phash, np.full(n - 1, 32, np.uint8))  # handlePrepare
            c, _ = bv.is_valid_validator(r.payload, r.off, r.msg_sig65, r.signer20)       # COMMIT ingest
            d = bv.is_valid_proposal_hash(r.raw, r.round, r.hash32, r.hash_len)           # handleCommit a1
            e, t = bv.is_valid_committed_seal(r.hash32, r.seal65, r.signer20)             # handleCommit a2 + tally
            return a, b, c, d, e, t
        a, b, c, d, e, t = sequence()
        assert a.all() and b.all() and c.all() and d.all() and e.all() and t.has_quorum == 1
        assert (e == B.verify_seals(vs, r.hash32, r.seal65, r.signer20, nthreads=8).astype(bool)).all()
        sequence()                                       # warm: second pass builds/uses the tables
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            sequence()
            lat.append(time.perf_counter() - t0)
        sigs = (n - 1) + n + n                           # signatures checked per sequence
        out[label] = {"sequence_ms_p50": float(np.median(lat) * 1e3), "sequence_ms_min": float(min(lat) * 1e3),
                      "signatures_per_sequence": sigs, "sig_verifies_per_s": sigs / float(np.median(lat)),
                      "dispatch_cold_warm_lanes": bv.last_dispatch()}

        def sequence_wire():
            a, prow, _ = bv.is_valid_validator_wire(pwire, pwoff)                       # PREPARE ingest from bytes
            b = (prow["proposal_hash"] == H).all(axis=1) & (prow["hash_len"] == 32)     # handlePrepare a1
            c, crow, _ = bv.is_valid_validator_wire(cwire, cwoff)                       # COMMIT ingest from bytes
            d = (crow["proposal_hash"] == H).all(axis=1) & (crow["hash_len"] == 32)     # handleCommit a1
            bv.wire_stage_seals()                                                       # seals of that upload
            bv.seals_launch(1)
            e, t = bv.seals_fetch()                                                     # handleCommit a2 + tally
            return a, b, c, d, e, t
        a, b, c, d, e, t = sequence_wire()
        assert a.all() and b.all() and c.all() and d.all() and e.all() and t.has_quorum == 1
        sequence_wire()
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            sequence_wire()
            lat.append(time.perf_counter() - t0)
        out[label + "_from_wire"] = {"sequence_ms_p50": float(np.median(lat) * 1e3), "sequence_ms_min": float(min(lat) * 1e3),
                                     "wire_bytes": len(pwire) + len(cwire)}
        bv.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
