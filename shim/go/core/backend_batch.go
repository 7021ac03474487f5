//go:build ibftgpu

// backend_batch.go — lives in package core next to backend.go.  Adds the OPTIONAL batch
// interface; a Backend that does not implement it keeps the stock per-message path, so
// RunSequence and every existing Backend keep working unchanged.
//
// NOT COMPILED HERE (no Go toolchain in the build image).  The identical control flow is
// implemented and tested in C++: go-ibft_amd/host/backend.cpp (HotPath::handlePrepare /
// handleCommit, GpuBackend).
package core

import (
	"os"
	"strconv"

	"github.com/0xPolygon/go-ibft/messages"
	"github.com/0xPolygon/go-ibft/messages/proto"
)

// MinDeviceRows is SURVEY §5's "min batch for GPU" knob (the environment's IBFT_MIN_DEVICE_ROWS, 0 = never decline).  A launch
// has a floor of ≈ 0.2 ms whatever the row count, one host core recovers a signature in 31–50 µs: below a handful of rows the
// stock closures are FASTER than the device, and every validator count the reference itself tests lives there (4:
// core/consensus_test.go:139, 6: core/byzantine_test.go:21, ≤ 30: core/rapid_test.go:156).  A batch with fewer rows is not
// offered to the BatchVerifier at all — the callers below treat it exactly like ok == false and run the per-message closure
// (same verdicts by construction).  The measured crossover and the recommended value: INTEGRATION.md §2.  The C++ mirror's
// statement of the same rule: go-ibft_amd/host/backend.hpp (BatchVerifier::declines), tests/test_host_device_quorum.py.
var MinDeviceRows = envRows("IBFT_MIN_DEVICE_ROWS")

func envRows(name string) int {
	if v, err := strconv.Atoi(os.Getenv(name)); err == nil && v > 0 {
		return v
	}
	return 0
}

// tooFewForDevice reports whether a batch of n rows stays on the host (MinDeviceRows).
func tooFewForDevice(n int) bool { return n < MinDeviceRows }

// BatchVerifier is type-asserted on the Backend passed to NewIBFT.
type BatchVerifier interface {
	// verdicts[i] == IsValidProposalHash(proposal, ExtractPrepareHash(msgs[i]))
	VerifyPrepareBatch(proposal *proto.Proposal, msgs []*proto.IbftMessage) (verdicts []bool, ok bool)
	// verdicts[i] == IsValidProposalHash(proposal, ExtractCommitHash(msgs[i])) &&
	//                IsValidCommittedSeal(ExtractCommitHash(msgs[i]), ExtractCommittedSeal(msgs[i]))
	VerifyCommitBatch(proposal *proto.Proposal, msgs []*proto.IbftMessage) (verdicts []bool, ok bool)
	// verdicts[i] == IsValidValidator(msgs[i])
	VerifySenderBatch(msgs []*proto.IbftMessage) (verdicts []bool, ok bool)
}

// batchStore is implemented by messages.Messages (shim/go/messages/soa.go).
type batchStore interface {
	GetValidMessagesBatch(view *proto.View, t proto.MessageType,
		verdicts func([]*proto.IbftMessage) []bool) []*proto.IbftMessage
}

// commitMessagesFor replaces the first statement of handleCommit (core/ibft.go:931-946):
//
//	commitMessages := i.commitMessagesFor(view)
//
// With a BatchVerifier the whole view is verified in one device call; without one (or when the
// device path reports !ok) the stock closure runs, message by message.
func (i *IBFT) commitMessagesFor(view *proto.View) []*proto.IbftMessage {
	isValidCommit := func(message *proto.IbftMessage) bool {
		proposalHash := messages.ExtractCommitHash(message)
		committedSeal := messages.ExtractCommittedSeal(message)
		if !i.backend.IsValidProposalHash(i.state.getProposal(), proposalHash) {
			return false
		}
		return i.backend.IsValidCommittedSeal(proposalHash, committedSeal)
	}
	bv, hasBatch := i.backend.(BatchVerifier)
	store, storeOK := i.messages.(batchStore)
	if hasBatch && storeOK {
		fellBack := false
		msgs := store.GetValidMessagesBatch(view, proto.MessageType_COMMIT,
			func(all []*proto.IbftMessage) []bool {
				// what AddMessages already judged against this proposal (message_sets.go) is not asked again
				verdicts, rest, restIdx := i.lookupClosures(all)
				if len(rest) == 0 {
					return verdicts
				}
				var vr []bool
				ok := false
				if !tooFewForDevice(len(rest)) {
					vr, ok = bv.VerifyCommitBatch(i.state.getProposal(), rest)
				}
				if !ok || len(vr) != len(rest) { // device unavailable or batch too small: the per-message verifier, same lock held
					fellBack = true
					vr = make([]bool, len(rest))
					for k, m := range rest {
						vr[k] = isValidCommit(m)
					}
				}
				for k, v := range vr {
					verdicts[restIdx[k]] = v
				}
				return verdicts
			})
		_ = fellBack // exported as a metric in a real integration
		return msgs
	}
	return i.messages.GetValidMessages(view, proto.MessageType_COMMIT, isValidCommit)
}

// prepareMessagesFor is the same change for handlePrepare (core/ibft.go:855-868).
func (i *IBFT) prepareMessagesFor(view *proto.View) []*proto.IbftMessage {
	isValidPrepare := func(message *proto.IbftMessage) bool {
		return i.backend.IsValidProposalHash(i.state.getProposal(), messages.ExtractPrepareHash(message))
	}
	bv, hasBatch := i.backend.(BatchVerifier)
	store, storeOK := i.messages.(batchStore)
	if hasBatch && storeOK {
		return store.GetValidMessagesBatch(view, proto.MessageType_PREPARE,
			func(all []*proto.IbftMessage) []bool {
				verdicts, rest, restIdx := i.lookupClosures(all)
				if len(rest) == 0 {
					return verdicts
				}
				var vr []bool
				ok := false
				if !tooFewForDevice(len(rest)) {
					vr, ok = bv.VerifyPrepareBatch(i.state.getProposal(), rest)
				}
				if !ok || len(vr) != len(rest) {
					vr = make([]bool, len(rest))
					for k, m := range rest {
						vr[k] = isValidPrepare(m)
					}
				}
				for k, v := range vr {
					verdicts[restIdx[k]] = v
				}
				return verdicts
			})
	}
	return i.messages.GetValidMessages(view, proto.MessageType_PREPARE, isValidPrepare)
}
