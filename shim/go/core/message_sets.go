//go:build ibftgpu

// message_sets.go — lives in package core.  A PREPARE / COMMIT message that arrives while the proposal of its
// view is already accepted can be judged COMPLETELY on arrival: IsValidValidator (core/ibft.go:1128) and the
// closure handlePrepare / handleCommit will later apply to it (:856-862, :932-944) are pure functions of the
// message, the proposal and the validator set.  A Backend that offers SetVerifier gets one call per message
// type and micro-batch (libibftgpu: ibft_verify_messages — both signatures of every COMMIT in one verdict
// launch); the closure verdicts wait in a table keyed by the stored message, and commitMessagesFor /
// prepareMessagesFor (backend_batch.go) only send the device what the table cannot answer.
//
// NOT COMPILED HERE (no Go toolchain in the build image).  The identical control flow is implemented and
// tested in C++: go-ibft_amd/host/backend.cpp (HotPath::IngestWire, HotPath::closureVerdicts),
// tests/test_host_roundchange.py (no device), tests/test_gpu_host.py (MI355X).
package core

import (
	"bytes"
	"encoding/binary"
	"sync"

	"github.com/0xPolygon/go-ibft/messages/proto"
)

// SetVerifier is the optional interface a Backend offers next to BatchVerifier.
type SetVerifier interface {
	// sender[k] == IsValidValidator(msgs[k]); closure[k] == the handlePrepare (t == PREPARE) or handleCommit
	// (t == COMMIT) closure for msgs[k] against proposal.  ok == false: not judged, use the older routes.
	VerifyMessageSet(proposal *proto.Proposal, t proto.MessageType, msgs []*proto.IbftMessage) (sender, closure []bool, ok bool)
}

// closureTable holds the closure verdicts of stored messages for ONE proposal (raw bytes + round); a verdict
// computed against another proposal is worthless, so the table is dropped when the key changes.  Entries hold
// the message pointer, so an address cannot be reused while its verdict is remembered.
type closureTable struct {
	mu      sync.Mutex
	key     []byte
	verdict map[*proto.IbftMessage]bool
}

var closureTables sync.Map // *IBFT → *closureTable  (a field of IBFT in a real merge)

func (i *IBFT) closures() *closureTable {
	t, _ := closureTables.LoadOrStore(i, &closureTable{verdict: map[*proto.IbftMessage]bool{}})
	return t.(*closureTable)
}

func proposalKey(p *proto.Proposal) []byte {
	if p == nil {
		return nil
	}
	key := append([]byte{}, p.RawProposal...)
	return binary.BigEndian.AppendUint64(key, p.Round)
}

// sync drops the table when the accepted proposal is not the one its verdicts refer to.  Caller holds t.mu.
func (t *closureTable) sync(p *proto.Proposal) {
	if key := proposalKey(p); !bytes.Equal(key, t.key) {
		t.key = key
		t.verdict = map[*proto.IbftMessage]bool{}
	}
}

// lookupClosures splits `all` into verdicts already known and the messages the batch backend still has to judge.
func (i *IBFT) lookupClosures(all []*proto.IbftMessage) (verdicts []bool, rest []*proto.IbftMessage, restIdx []int) {
	t := i.closures()
	t.mu.Lock()
	defer t.mu.Unlock()
	t.sync(i.state.getProposal())
	verdicts = make([]bool, len(all))
	for k, m := range all {
		if v, known := t.verdict[m]; known {
			verdicts[k] = v
		} else {
			rest = append(rest, m)
			restIdx = append(restIdx, k)
		}
	}
	return verdicts, rest, restIdx
}

// AddMessages is what the transport calls with a micro-batch of decoded messages instead of AddMessage once per
// message.  The PREPARE / COMMIT messages of the current view go through one SetVerifier call per type; every
// other message (and every message when the Backend offers no SetVerifier, no proposal is accepted yet, or the
// device is unavailable) takes AddMessage, i.e. the stock route.
func (i *IBFT) AddMessages(batch []*proto.IbftMessage) {
	sv, hasSets := i.backend.(SetVerifier)
	proposal := i.state.getProposal()
	if !hasSets || proposal == nil {
		for _, m := range batch {
			i.AddMessage(m)
		}
		return
	}
	var ofType [2][]*proto.IbftMessage
	for _, m := range batch {
		here := m != nil && m.View != nil && m.View.Height == i.state.getHeight() && m.View.Round == i.state.getRound()
		switch {
		case here && m.Type == proto.MessageType_PREPARE:
			ofType[0] = append(ofType[0], m)
		case here && m.Type == proto.MessageType_COMMIT:
			ofType[1] = append(ofType[1], m)
		default:
			i.AddMessage(m)
		}
	}
	for k, t := range []proto.MessageType{proto.MessageType_PREPARE, proto.MessageType_COMMIT} {
		if len(ofType[k]) == 0 {
			continue
		}
		var sender, closure []bool
		ok := false
		if !tooFewForDevice(len(ofType[k])) { // (MinDeviceRows, backend_batch.go)
			sender, closure, ok = sv.VerifyMessageSet(proposal, t, ofType[k])
		}
		if !ok || len(sender) != len(ofType[k]) || len(closure) != len(ofType[k]) {
			for _, m := range ofType[k] {
				i.AddMessage(m)
			}
			continue
		}
		tab := i.closures()
		tab.mu.Lock()
		tab.sync(proposal)
		for j, m := range ofType[k] {
			if sender[j] {
				tab.verdict[m] = closure[j]
			}
		}
		tab.mu.Unlock()
		for j, m := range ofType[k] {
			if sender[j] {
				i.addVerifiedMessage(m) // wire_ingest.go: AddMessage minus the IsValidValidator call
			}
		}
	}
}
