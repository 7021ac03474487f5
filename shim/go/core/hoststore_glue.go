//go:build ibftgpu

// hoststore_glue.go — lives in package core.  The call sites a maintainer changes so that the MEASURED path (receive queue →
// rows → handle* walks in libibft_host.so, DESIGN.md §5.5) is the one the Go node runs.  IBFT.messages is an interface
// (core/ibft.go:23-46); *hoststore.Store implements it, so NewIBFT's `messages: messages.NewMessages()` (core/ibft.go:119)
// becomes `messages: store` — nothing else of RunSequence changes.  The store's extra methods are reached by a type
// assertion, exactly like BatchVerifier on the Backend (backend_batch.go): a node built without the tag, or whose store is
// the reference's, keeps the stock code below every `if`.
//
// NOT COMPILED HERE (no Go toolchain in the build image); the same control flow is the C++ mirror's, driven by
// tests/test_hoststore_sequence.py through the C call sequence of shim/go/hoststore.
package core

import (
	"github.com/0xPolygon/go-ibft/messages"
	"github.com/0xPolygon/go-ibft/messages/proto"
	goproto "google.golang.org/protobuf/proto"
)

// hostStore is implemented by *hoststore.Store.
type hostStore interface {
	AddWireMessages(raw [][]byte) error
	SetState(height, round uint64, proposalMessage *proto.IbftMessage) error
	HandlePrepare(view *proto.View) (quorum bool, prepared [][]byte)
	HandleCommit(view *proto.View) (quorum bool, seals []*messages.CommittedSeal)
	HandleRoundChange(view *proto.View) []*proto.IbftMessage
	HandlePrePrepare(view *proto.View) *proto.IbftMessage
}

// AddWireMessagesQueued is what the transport's handler calls with the bytes it received — instead of proto.Unmarshal +
// IBFT.AddMessage per message (core/ibft.go:1101-1123).  false = the store is not a host store (or refused): the caller
// takes AddWireMessages (wire_ingest.go), which unmarshals and judges in Go.
func (i *IBFT) AddWireMessagesQueued(raw [][]byte) bool {
	hs, ok := i.messages.(hostStore)
	if !ok {
		return false
	}
	return hs.AddWireMessages(raw) == nil
}

// syncHostState follows every state change the hot path reads (startRound :304, acceptProposal :1061, moveToNewRound
// :1036): the mirror judges arriving PREPARE / COMMIT messages against the accepted proposal of the current view.
func (i *IBFT) syncHostState() {
	if hs, ok := i.messages.(hostStore); ok {
		_ = hs.SetState(i.state.getHeight(), i.state.getRound(), i.state.getProposalMessage())
	}
}

// handlePrepareHost replaces the body of handlePrepare (core/ibft.go:855-889) when the store is a host store: the
// quorum decision (HasPrepareQuorum over the PREPAREs whose hash matched) comes back with the surviving messages' bytes;
// they are decoded here, once, only because finalizePrepare keeps them for a future PreparedCertificate.
func (i *IBFT) handlePrepareHost(view *proto.View) (handled, quorum bool) {
	hs, ok := i.messages.(hostStore)
	if !ok {
		return false, false
	}
	q, prepared := hs.HandlePrepare(view)
	if !q {
		return true, false
	}
	msgs := make([]*proto.IbftMessage, 0, len(prepared))
	for _, raw := range prepared {
		m := &proto.IbftMessage{}
		if goproto.Unmarshal(raw, m) == nil {
			msgs = append(msgs, m)
		}
	}
	i.sendCommitMessage(view)
	i.state.finalizePrepare(
		&proto.PreparedCertificate{ProposalMessage: i.state.getProposalMessage(), PrepareMessages: msgs},
		i.state.getProposal(),
	)
	return true, true
}

// handleCommitHost replaces the body of handleCommit (core/ibft.go:931-967): quorum + the committed seals of exactly the
// surviving COMMITs (ExtractCommittedSeals, messages/helpers.go:22-35), read off the rows' bytes by the mirror.
func (i *IBFT) handleCommitHost(view *proto.View) (handled, quorum bool) {
	hs, ok := i.messages.(hostStore)
	if !ok {
		return false, false
	}
	q, seals := hs.HandleCommit(view)
	if !q {
		return true, false
	}
	i.state.setCommittedSeals(seals)
	i.state.changeState(fin)
	return true, true
}

// handleRoundChangeHost / handlePrePrepareHost: the certificate walks (core/ibft.go:470-512, 792-813) answered from the
// verdicts the device left when the carriers arrived.
func (i *IBFT) handleRoundChangeHost(view *proto.View) (*proto.RoundChangeCertificate, bool) {
	hs, ok := i.messages.(hostStore)
	if !ok {
		return nil, false
	}
	msgs := hs.HandleRoundChange(view)
	if msgs == nil {
		return nil, true
	}
	return &proto.RoundChangeCertificate{RoundChangeMessages: msgs}, true
}

func (i *IBFT) handlePrePrepareHost(view *proto.View) (*proto.IbftMessage, bool) {
	hs, ok := i.messages.(hostStore)
	if !ok {
		return nil, false
	}
	return hs.HandlePrePrepare(view), true
}
