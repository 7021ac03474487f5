"""oracle/soak_job.py — one job of the bit-parity soak: a synthetic Byzantine round and the CPU oracle's
answers for it.  TEST INFRASTRUCTURE ONLY (tools/soak.py, tests/test_gpu_soak.py run these in worker processes while
the GPU consumer compares).  A top-level function of an importable module so that spawn-ed workers can load it."""
import numpy as np


def make(job):
    n, seed = job
    from oracle import binding as B
    from oracle import workload as W
    r = W.make_round(n, seed, byzantine=True, weighted=bool(seed & 1), with_envelopes=True)
    if seed % 3 == 0:   # some forged envelopes too: a neighbour's signature
        k = seed % n
        r.msg_sig65[k] = r.msg_sig65[(k + 1) % n]
    vs = B.ValSet(r.addrs, r.power)
    exp = B.verify_seals(vs, r.hash32, r.seal65, r.signer20, r.pre_flags, nthreads=1)
    t = B.tally(vs, r.signer20, exp)
    snd = B.verify_senders(vs, r.payload, r.off, r.msg_sig65, r.signer20).astype(bool)
    clo = B.verify_hashes(r.raw, r.round, r.hash32, r.hash_len).astype(bool) & exp.astype(bool)
    ts = B.tally(vs, r.signer20, (snd & clo).astype(np.uint8))
    # HasPrepareQuorum over the same surviving rows with validator seed mod n as the proposer (core/validator_manager.go:99-127,
    # restated in oracle/semantics.py): (has_quorum, Σ power incl. the proposer's seat, rows sent by the proposer)
    from oracle.semantics import ValidatorManager
    from collections import namedtuple
    M = namedtuple("M", "sender")
    vm = ValidatorManager()
    vm.init({bytes(r.addrs[i]): int(r.power[i]) for i in range(n)})
    proposer = bytes(r.addrs[seed % n])
    both = snd & clo
    msgs = [M(bytes(r.signer20[i])) for i in np.flatnonzero(both)]
    senders = ({proposer} | {m.sender for m in msgs}) & set(vm.power)
    hpq = (int(vm.has_prepare_quorum(M(proposer), msgs)), sum(vm.power[a] for a in senders),
           sum(1 for m in msgs if m.sender == proposer), proposer)
    return (n, seed, r.addrs, r.power, r.hash32, r.seal65, r.signer20, r.pre_flags, exp.astype(bool),
            (t.power, t.quorum, t.valid_rows, t.distinct_senders, t.has_quorum),
            (r.payload, r.off, r.msg_sig65, r.hash_len, r.raw, r.round, snd, clo,
             (ts.power, ts.quorum, ts.valid_rows, ts.distinct_senders, ts.has_quorum), hpq))
