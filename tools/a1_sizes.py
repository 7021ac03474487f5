"""a1 (IsValidProposalHash, core/backend.go:50-51) at realistic block sizes: N rows against one proposal of
L bytes.  First call = upload + Keccak of the proposal on the device + N compares; repeated call with the same
proposal = N compares (the context remembers the proposal it hashed); digest form = the caller supplies
keccak(proposal).  GPU box only.  Usage: python tools/a1_sizes.py [N=4096] > profiles/r02_a1_sizes.json"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import go_ibft_amd.verifier as V
from oracle import binding as B      # checker only: expected digests / verdicts

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(1)
bv = V.BatchVerifier(max_rows=n)      # IBFT_PROPOSAL_HASH=device in the environment selects the wavefront sponge (round 2's route)
out = {"rows": n, "proposal_hashed_by": os.environ.get("IBFT_PROPOSAL_HASH", "host"),
       "note": "ms per call, host columns -> host-visible verdict mask (H2D, kernels, D2H included); p50 of 20 (first call: of 5 distinct proposals)"}
for L in (1024, 65536, 1 << 20, 8 << 20):
    first, again, dig = [], [], []
    for rep in range(5):
        raw = rng.bytes(L)
        H = B.proposal_hash(raw, 7)
        hashes = np.tile(np.frombuffer(H, np.uint8), (n, 1)).copy()
        hashes[::7, 3] ^= 1
        hl = np.full(n, 32, np.uint8)
        exp = B.verify_hashes(raw, 7, hashes, hl).astype(bool)
        t0 = time.perf_counter(); got = bv.is_valid_proposal_hash(raw, 7, hashes, hl); first.append(time.perf_counter() - t0)
        assert (got == exp).all()
        for _ in range(4):
            t0 = time.perf_counter(); got = bv.is_valid_proposal_hash(raw, 7, hashes, hl); again.append(time.perf_counter() - t0)
            assert (got == exp).all()
            t0 = time.perf_counter(); got = bv.is_valid_proposal_hash_digest(H, hashes, hl); dig.append(time.perf_counter() - t0)
            assert (got == exp).all()
            assert bv.proposal_hash(raw, 7) == H
    out[f"L={L}"] = {"first_call_ms": float(np.median(first) * 1e3), "same_proposal_again_ms": float(np.median(again) * 1e3),
                    "digest_supplied_ms": float(np.median(dig) * 1e3), "keccak_blocks": (L + 8) // 136 + 1}
bv.close()
print(json.dumps(out, indent=1))
