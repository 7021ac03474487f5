"""BASELINE config #1 — the reference's in-process "cluster": N IBFT instances in one process whose Transport.Multicast is a
loop of AddMessage over every node (/root/reference/core/mock_test.go:546-550, core/helpers_test.go:227-231), round-robin
proposer (height + round) % N (core/helpers_test.go:214-225), one height PREPREPARE → PREPARE → COMMIT
(core/consensus_test.go:133-248).

Every node is ONE host mirror (go_ibft_amd.hostlib.Host = messages.Messages + ValidatorManager + IBFT.AddMessage +
handlePrePrepare / handlePrepare / handleCommit).  What is NOT the hot path — timers, goroutines, round changes — is not
built (RunSequence stays Go); this driver plays exactly the transitions of runStates for a round that
succeeds (core/ibft.go:554-576: newRound → prepare → commit → fin), woken the way the reference wakes them: an
AddMessage that returns 2 is the SignalEvent of core/ibft.go:1118-1119, and entering a state probes what is already
stored (subscribe, core/ibft.go:1286-1298).  Message delivery is a FIFO: deterministic, no threads.

Used by tests/test_host_cluster.py (CPU: mock verifier like the reference's, per-message / batched / batched ingest) and
tests/test_gpu_cluster.py (real signatures, the device behind every node)."""
from __future__ import annotations

import go_ibft_amd.hostlib as H
from oracle import wire as W

PP, PR, CM = W.PREPREPARE, W.PREPARE, W.COMMIT


def raw_proposal_of(preprepare_wire: bytes) -> bytes:
    """Proposal.rawProposal of a PREPREPARE message (messages/helpers.go:65-76: ExtractProposal), read off the wire:
    IbftMessage.preprepareData (field 5) → PrePrepareMessage.proposal (field 1) → Proposal.rawProposal (field 1)"""
    from oracle.wire_parse import fields

    def first(buf, num):
        for f, wt, val in fields(buf):
            if f == num and wt == 2:
                return val
        return b""
    return first(first(first(preprepare_wire, 5), 1), 1)


class Node:
    """one IBFT instance: the mirror + the Backend callbacks that BUILD messages (core/backend.go:12-34 — out of the hot
    path, so they live here) + InsertProposal's record"""

    def __init__(self, index: int, address: bytes, cluster: "Cluster"):
        self.index, self.address, self.cluster = index, address, cluster
        self.host = H.Host()
        self.state = "newRound"
        self.inserted = None          # (raw proposal, frozenset of (signer, signature)) handed to InsertProposal
        self.byzantine = False
        self.offline = False
        self.inbox: list[tuple[int, bytes]] = []
        self.accepted = None          # the PREPREPARE accepted for this view (wire bytes)
        self.signals = 0              # AddMessage results of 2 seen (SignalEvent count)
        self.pending: set = set()     # lockstep: signalled types waiting for the wake-up
        self.prepared = None          # PC.PrepareMessages after finalizePrepare

    # -- RunSequence start (core/ibft.go:304-317): state.reset, validatorManager.Init, PruneByHeight
    def start_height(self, height: int):
        self.height, self.round = height, 0
        self.state, self.accepted, self.inserted, self.prepared = "newRound", None, None, None
        self.inbox, self.signals = [], 0
        self.host.store_prune(height)
        self.host.set_state(height, 0, None)


class Cluster:
    """mode: 'stock' = per-message Verifier inside the walks; 'batch' = one batch call per walk (LoopBatch over the same
    callbacks, or the GPU when a BatchVerifier is attached); 'ingest' = batch + messages delivered as micro-batches of wire
    bytes through ibft_host_ingest_wire (the receive side).
    lockstep: nodes are only woken once every message in flight has been delivered (a network that delivers a whole
    phase before anybody's timer-free state machine runs) — then every node sees the same stored set and hands the same
    seals to InsertProposal; eager (lockstep=False): a node acts on the SignalEvent of the message that completed its
    quorum, as the reference's goroutines do, and may finish with any quorum-sized subset."""

    def __init__(self, addresses, build, verifier, mode="stock", gpu=None, micro_batch=3, powers=None, lockstep=True):
        self.mode, self.build, self.micro_batch, self.lockstep = mode, build, micro_batch, lockstep
        self.nodes = [Node(i, a, self) for i, a in enumerate(addresses)]
        powers = powers or {a: 1 for a in addresses}
        for nd in self.nodes:
            h = nd.host
            assert h.vm_init(powers)
            h.set_id(nd.address)
            h.set_verifier(*verifier(nd))
            if gpu is not None:
                h.attach_gpu(gpu)
            elif mode != "stock":
                h.use_loop_batch(0)
            h.use_batch(mode != "stock")
            if mode == "ingest":
                h.enable_quorum_index()

    def close(self):
        for nd in self.nodes:
            nd.host.close()

    def proposer(self, height, round_) -> int:
        return (height + round_) % len(self.nodes)     # core/helpers_test.go:214-225

    # -- Transport.Multicast: every node, the sender included (core/helpers_test.go:227-231)
    def multicast(self, type_: int, wire: bytes):
        for nd in self.nodes:
            if not nd.offline:
                nd.inbox.append((type_, wire))

    def _deliver(self, nd: Node, batch):
        if self.mode == "ingest":
            res, *_ = nd.host.ingest_wire([w for _, w in batch])
        else:
            res = [nd.host.add_message(w) for _, w in batch]
        for (type_, _), rc in zip(batch, res):
            if rc == 2:
                nd.signals += 1
                if self.lockstep:
                    nd.pending.add(type_)
                else:
                    self._wake(nd, type_)

    # -- the state functions of a successful round (core/ibft.go:579-625, 816-851, 892-927, 970-991)
    def _wake(self, nd: Node, type_=None):
        h, view = nd.host, (nd.height, nd.round)
        if nd.state == "newRound" and type_ in (None, PP):
            msg = h.handle_preprepare(*view)                       # handlePrePrepare → validateProposal0
            if msg is None:
                return
            nd.accepted = msg
            h.set_state(nd.height, nd.round, msg)                  # acceptProposal
            self.multicast(PR, self.build(nd, PR, view))           # sendPrepareMessage
            nd.state = "prepare"
            if not self.lockstep:
                self._wake(nd)                                     # subscribe() probes what is already stored
        elif nd.state == "prepare" and type_ in (None, PR):
            ok, prepared = h.handle_prepare(*view)
            if not ok:
                return
            nd.prepared = frozenset(prepared)                      # finalizePrepare: PC.PrepareMessages
            self.multicast(CM, self.build(nd, CM, view))           # sendCommitMessage
            nd.state = "commit"
            if not self.lockstep:
                self._wake(nd)
        elif nd.state == "commit" and type_ in (None, CM):
            ok, seals = h.handle_commit(*view)
            if not ok:
                return
            nd.state = "fin"
            # runFin → insertBlock → Backend.InsertProposal(proposal, seals) (core/ibft.go:970-991)
            nd.inserted = (raw_proposal_of(nd.accepted), frozenset(seals))

    def run_height(self, height: int, max_rounds: int = 100000):
        for nd in self.nodes:
            nd.start_height(height)
        p = self.nodes[self.proposer(height, 0)]
        view = (height, 0)
        if not p.offline:
            # runNewRound, proposer side (core/ibft.go:584-607): build, accept, multicast, move to prepare
            pp = self.build(p, PP, view)
            p.accepted = pp
            p.host.set_state(height, 0, pp)
            self.multicast(PP, pp)
            p.state = "prepare"
        take = self.micro_batch if self.mode == "ingest" else 1
        for _ in range(max_rounds):
            busy = False
            for nd in self.nodes:
                if nd.inbox and not nd.offline:
                    batch, nd.inbox = nd.inbox[:take], nd.inbox[take:]
                    self._deliver(nd, batch)
                    busy = True
            if busy:
                continue
            if self.lockstep:      # everything in flight has arrived: every node probes its state once
                before = [nd.state for nd in self.nodes]
                for nd in self.nodes:
                    nd.pending.clear()
                    self._wake(nd)
                if [nd.state for nd in self.nodes] != before or any(nd.inbox for nd in self.nodes):
                    continue
            break
        return [nd.inserted for nd in self.nodes]
