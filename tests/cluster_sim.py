"""BASELINE config #1 — the reference's in-process "cluster": N IBFT instances in one process whose Transport.Multicast is a
loop of AddMessage over every node (/root/reference/core/mock_test.go:546-550, core/helpers_test.go:227-231), round-robin
proposer (height + round) % N (core/helpers_test.go:214-225), one height PREPREPARE → PREPARE → COMMIT
(core/consensus_test.go:133-248).

Every node is ONE host mirror (go_ibft_amd.hostlib.Host = messages.Messages + ValidatorManager + IBFT.AddMessage +
handlePrePrepare / handlePrepare / handleCommit / handleRoundChangeMessage).  What is NOT the hot path — timers, goroutines —
is not built (RunSequence stays Go); this driver plays the transitions of runStates (core/ibft.go:554-576:
newRound → prepare → commit → fin), woken the way the reference wakes them: an AddMessage that returns 2 is the SignalEvent
of core/ibft.go:1118-1119, and entering a state probes what is already stored (subscribe, core/ibft.go:1286-1298).  Message
delivery is a FIFO: deterministic, no threads.

Round changes (round 4): the round TIMER is driven by hand — Cluster.tick() is "roundExpired" at every node that has not
finished (core/ibft.go:371-378: moveToNewRound(round + 1) + sendRoundChangeMessage with the node's latest prepared proposal
and prepared certificate) — and the new round is the reference's: its proposer waits for a RoundChangeCertificate
(waitForRCC → handleRoundChangeMessage, :413-512), re-proposes the proposal of the highest-round prepared certificate or
builds a new block (buildProposal, :1005-1089), multicasts the PREPREPARE with the certificate; everybody else validates it
(handlePrePrepare → validateProposal, :683-813) — so the certificate paths (f2: validPC, proposalMatchesCertificate, the
RCC's sender checks) run end to end, in stock / batch / ingest mode (core/consensus_test.go:260-394 invalid block → round 1,
core/byzantine_test.go:60-130).

Used by tests/test_host_cluster.py (CPU: mock verifier like the reference's, per-message / batched / batched ingest) and
tests/test_gpu_cluster.py (real signatures, the device behind every node)."""
from __future__ import annotations

import go_ibft_amd.hostlib as H
from oracle import wire as W

PP, PR, CM, RC = W.PREPREPARE, W.PREPARE, W.COMMIT, W.ROUND_CHANGE


def _first(buf, num):
    from oracle.wire_parse import fields
    for f, wt, val in fields(buf):
        if f == num and wt == 2:
            return val
    return None


def _varint_field(buf, num) -> int:
    from oracle.wire_parse import fields
    for f, wt, val in fields(buf):
        if f == num and wt == 0:
            return val
    return 0


def prepared_certificate_bytes(proposal_wire: bytes, prepare_wires) -> bytes:
    """PreparedCertificate{proposalMessage = 1, prepareMessages = 2 (repeated)} from wire messages"""
    return W._len_field(1, proposal_wire, emit_empty=True) + b"".join(W._len_field(2, m, emit_empty=True) for m in prepare_wires)


def round_change_certificate_bytes(rc_wires) -> bytes:
    """RoundChangeCertificate{roundChangeMessages = 1 (repeated)} from wire messages"""
    return b"".join(W._len_field(1, m, emit_empty=True) for m in rc_wires)


def previous_proposal_of(rc_wires):
    """buildProposal's scan of the RCC (core/ibft.go:1037-1063): the lastPreparedProposal of the ROUND_CHANGE message whose
    prepared certificate has the highest proposal round (the first one among equals), or None"""
    previous, max_round = None, 0
    for m in rc_wires:
        body = _first(m, 8) or b""
        pc = _first(body, 2)                                   # ExtractLatestPC
        if pc is None:
            continue
        pp = _first(pc, 1)                                     # PC.ProposalMessage → ExtractProposal
        proposal = _first(_first(pp or b"", 5) or b"", 1) or b""
        pc_round = _varint_field(proposal, 2)
        if previous is not None and pc_round <= max_round:
            continue
        last = _first(body, 1)                                 # ExtractLastPreparedProposal
        if last is None:
            continue
        previous, max_round = (_first(last, 1) or b""), pc_round
    return previous


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
        self.latest_pc = None         # state.latestPC: (proposal message wire, [PREPARE wires]) — survives round changes
        self.latest_prepared = None   # state.latestPreparedProposal: (raw, round)
        self.rcc = None               # the certificate this node (as proposer of a round > 0) put into its PREPREPARE
        self.silent = False           # a proposer that builds nothing (drops its own PREPREPARE)
        self.withhold_commit = False  # sends no COMMIT (to strand a round after PREPARE)

    # -- RunSequence start (core/ibft.go:304-317): state.reset, validatorManager.Init, PruneByHeight
    def start_height(self, height: int):
        self.height, self.round = height, 0
        self.state, self.accepted, self.inserted, self.prepared = "newRound", None, None, None
        self.latest_pc, self.latest_prepared, self.rcc = None, None, None
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

    def __init__(self, addresses, build, verifier, mode="stock", gpu=None, micro_batch=3, powers=None, lockstep=True,
                 build_proposal=None):
        self.mode, self.build, self.micro_batch, self.lockstep = mode, build, micro_batch, lockstep
        # Backend.BuildProposal (core/backend.go:60-62) for a round > 0 without a previous proposal: node → raw bytes
        self.build_proposal = build_proposal
        self.censor = None
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
    def multicast(self, type_: int, wire: bytes, frm: "Node" = None):
        # censor(node, type, (height, round)) → True: the node's Transport drops the message (a silent Byzantine node,
        # core/rapid_test.go:227-236 — it still builds and accepts what it would have sent)
        if frm is not None and self.censor is not None and self.censor(frm, type_, (frm.height, frm.round)):
            return
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
        if nd.state == "newRound" and nd.round > 0 and nd.accepted is None and not nd.silent and \
                nd.index == self.proposer(nd.height, nd.round) and type_ in (None, RC):
            # startRound of a round > 0, proposer side: buildProposal → waitForRCC → handleRoundChangeMessage (:413-512, :1005-1089)
            rcc = h.handle_round_change(*view)
            if not rcc:
                return
            nd.rcc = list(rcc)
            previous = previous_proposal_of(rcc)
            nd.proposal_raw = previous if previous is not None else self.build_proposal(nd)
            pp = self.build(nd, PP, view)                          # BuildPrePrepareMessage(raw, rcc, view)
            nd.accepted = pp
            h.set_state(nd.height, nd.round, pp)                   # acceptProposal
            self.multicast(PP, pp, nd)
            nd.state = "prepare"
            if not self.lockstep:
                self._wake(nd)
            return
        if nd.state == "newRound" and type_ in (None, PP):
            msg = h.handle_preprepare(*view)                       # handlePrePrepare → validateProposal0
            if msg is None:
                return
            nd.accepted = msg
            h.set_state(nd.height, nd.round, msg)                  # acceptProposal
            self.multicast(PR, self.build(nd, PR, view), nd)           # sendPrepareMessage
            nd.state = "prepare"
            if not self.lockstep:
                self._wake(nd)                                     # subscribe() probes what is already stored
        elif nd.state == "prepare" and type_ in (None, PR):
            ok, prepared = h.handle_prepare(*view)
            if not ok:
                return
            nd.prepared = frozenset(prepared)                      # finalizePrepare: PC.PrepareMessages
            nd.latest_pc = (nd.accepted, sorted(prepared))         # … latestPC / latestPreparedProposal (core/state.go:209-222)
            nd.latest_prepared = (raw_proposal_of(nd.accepted), nd.round)
            if not nd.withhold_commit:
                self.multicast(CM, self.build(nd, CM, view), nd)       # sendCommitMessage
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

    # -- the round timer, by hand: roundExpired at every node that has not finished (core/ibft.go:371-378)
    def tick(self, max_rounds: int = 100000):
        for nd in self.nodes:
            if nd.offline or nd.state == "fin":
                continue
            nd.round += 1                                          # moveToNewRound: view, proposal message nil, newRound
            nd.state, nd.accepted, nd.prepared = "newRound", None, None
            nd.pending.clear()
            nd.host.set_state(nd.height, nd.round, None)
            self.multicast(RC, self.build(nd, RC, (nd.height, nd.round)), nd)   # sendRoundChangeMessage(latestPreparedProposal, latestPC)
        return self._pump(max_rounds)

    def run_height(self, height: int, max_rounds: int = 100000):
        for nd in self.nodes:
            nd.start_height(height)
        p = self.nodes[self.proposer(height, 0)]
        view = (height, 0)
        if not p.offline and not p.silent:
            # runNewRound, proposer side (core/ibft.go:584-607): build, accept, multicast, move to prepare
            p.proposal_raw = self.build_proposal(p) if self.build_proposal else None
            pp = self.build(p, PP, view)
            p.accepted = pp
            p.host.set_state(height, 0, pp)
            self.multicast(PP, pp, p)
            p.state = "prepare"
        return self._pump(max_rounds)

    def _pump(self, max_rounds: int = 100000):
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
