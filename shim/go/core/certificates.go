//go:build ibftgpu

// certificates.go — lives in package core.  SURVEY.md §8f rank 2 from the transport's bytes: a PREPREPARE carries a
// RoundChangeCertificate, a ROUND_CHANGE a PreparedCertificate (messages/proto/messages.proto:46-101), and the reference
// verifies every message nested in them one by one, long after arrival — validateProposal :683-788, validPC :1162-1231,
// proposalMatchesCertificate :516-551, handleRoundChangeMessage :470-512 (core/ibft.go) — each through IsValidValidator
// (PayloadNoSig re-marshalled per nested message) and IsValidProposalHash: O(N²) signatures per round change.  All of those
// verdicts are pure functions of the bytes and of the validator set, so a Backend that offers CertificateVerifier gets the
// raw PREPREPARE / ROUND_CHANGE messages of a micro-batch in ONE call when they arrive (libibftgpu:
// ibft_verify_certificates_wire — the device expands the tree, hashes and verifies every nested message); the verdicts
// wait in tables keyed by the decoded message objects, and the certificate walks consult the tables first.
//
// Edits in core/ibft.go for a real merge (three call sites):
//   validPC, validateProposal:          i.backend.IsValidValidator(m)            → i.isValidValidatorTabled(m)
//   proposalMatchesCertificate:         i.backend.IsValidProposalHash(p, hash)   → i.isValidProposalHashTabled(p, carrier, hash)
//   validateProposalCommon:             i.backend.IsValidProposalHash(p, hash)   → i.isValidProposalHashTabled(p, msg, hash)
//
// NOT COMPILED HERE (no Go toolchain in the build image).  The identical control flow is implemented and tested in C++:
// go-ibft_amd/host/backend.cpp (HotPath::IngestWire stage 0, HotPath::noteCertificateTree, isValidValidatorCached,
// lookupHashVerdict), tests/test_host_cert_ingest.py (no device), tests/test_gpu_host.py (MI355X).
package core

import (
	"sync"

	goproto "google.golang.org/protobuf/proto"

	"github.com/0xPolygon/go-ibft/ibftgpu"
	"github.com/0xPolygon/go-ibft/messages"
	"github.com/0xPolygon/go-ibft/messages/proto"
)

// CertificateVerifier is the optional interface a Backend offers next to BatchVerifier.
type CertificateVerifier interface {
	// rows breadth first (ibftgpu.Ctx.VerifyCertificatesWire); ok == false: not judged (device unavailable, tree too large)
	VerifyCertificatesWire(wire []byte, off []uint32) (nodes []ibftgpu.CertNode, class []byte, sender, hash, self []uint64, ok bool)
}

type certHashKey struct {
	proposal *proto.Proposal
	carrier  *proto.IbftMessage // the message whose proposal hash is compared with keccak(proposal)
}

// A verdict is only as good as the validator set it was computed against: judgedAt is the height the state (and therefore
// the validator table on the device) was at when the carrying message arrived.  A lookup at another height ignores the
// entry and asks the backend, exactly as the reference would (ADVICE r2: a verdict of height H must not answer at H+1).
type certVerdict struct {
	ok       bool
	judgedAt uint64
}

// certRoot remembers which keys one stored carrier contributed, so that pruning costs O(pruned).
type certRoot struct {
	senders []*proto.IbftMessage
	hashes  []certHashKey
}

// certTable holds arrival-time verdicts about nested messages.  Keys are the decoded objects of STORED carriers (a tree is
// noted only after addVerifiedMessage accepted its carrier); entries are filed per height and dropped when the state moves
// past it — AddWireMessages prunes on entry, so the tables never outlive the store's PruneByHeight by more than one batch.
type certTable struct {
	mu       sync.Mutex
	sender   map[*proto.IbftMessage]certVerdict
	hash     map[certHashKey]certVerdict
	byHeight map[uint64][]*certRoot
}

var certTables sync.Map // *IBFT → *certTable  (a field of IBFT in a real merge)

func (i *IBFT) certificates() *certTable {
	t, _ := certTables.LoadOrStore(i, &certTable{
		sender:   map[*proto.IbftMessage]certVerdict{},
		hash:     map[certHashKey]certVerdict{},
		byHeight: map[uint64][]*certRoot{},
	})
	return t.(*certTable)
}

// dropCertificateVerdicts forgets everything judged below `height` (the state's current height): called at the top of
// AddWireMessages, i.e. at most one batch after RunSequence's PruneByHeight (core/ibft.go:317).  O(entries dropped).
func (i *IBFT) dropCertificateVerdicts(height uint64) {
	t := i.certificates()
	t.mu.Lock()
	for h, roots := range t.byHeight {
		if h >= height {
			continue
		}
		for _, r := range roots {
			for _, k := range r.senders {
				delete(t.sender, k)
			}
			for _, k := range r.hashes {
				delete(t.hash, k)
			}
		}
		delete(t.byHeight, h)
	}
	t.mu.Unlock()
}

// nestedMessages lists the messages directly inside m in wire order — the order of the device's rows.
func nestedMessages(m *proto.IbftMessage) []*proto.IbftMessage {
	switch p := m.Payload.(type) {
	case *proto.IbftMessage_PreprepareData:
		if p.PreprepareData != nil && p.PreprepareData.Certificate != nil {
			return p.PreprepareData.Certificate.RoundChangeMessages
		}
	case *proto.IbftMessage_RoundChangeData:
		if p.RoundChangeData != nil && p.RoundChangeData.LatestPreparedCertificate != nil {
			pc := p.RoundChangeData.LatestPreparedCertificate
			out := make([]*proto.IbftMessage, 0, len(pc.PrepareMessages)+1)
			if pc.ProposalMessage != nil {
				out = append(out, pc.ProposalMessage)
			}
			return append(out, pc.PrepareMessages...)
		}
	}
	return nil
}

// addWireCertificates judges the PREPREPARE / ROUND_CHANGE messages among raw completely and adds those whose envelope
// the device vouched for; it returns the messages that still need one of the older routes (other types, messages the
// device did not judge).  AddWireMessages (wire_ingest.go) calls it first.
func (i *IBFT) addWireCertificates(cv CertificateVerifier, raw [][]byte) (rest [][]byte) {
	var carriers []*proto.IbftMessage
	var carrierRaw [][]byte
	for _, b := range raw {
		msg := new(proto.IbftMessage)
		if goproto.Unmarshal(b, msg) != nil {
			continue // dropped, as the transport's handler would
		}
		if msg.Type == proto.MessageType_PREPREPARE || msg.Type == proto.MessageType_ROUND_CHANGE {
			carriers, carrierRaw = append(carriers, msg), append(carrierRaw, b)
		} else {
			rest = append(rest, b)
		}
	}
	if len(carriers) == 0 {
		return rest
	}
	wire, off := concat(carrierRaw)
	nodes, class, sender, hash, self, ok := cv.VerifyCertificatesWire(wire, off)
	if !ok || len(nodes) < len(carriers) {
		return append(rest, carrierRaw...)
	}
	t := i.certificates()
	judgedAt := i.state.getHeight()
	for k, msg := range carriers {
		switch {
		case class[k] != 0:
			rest = append(rest, carrierRaw[k]) // not canonical here, or too long to hash on the device: stock route
		case ibftgpu.Bit(sender, k):
			// the tree is worth remembering only for a carrier that was actually stored (ADVICE r2): a forged envelope, a
			// stale view leave nothing behind
			if i.addVerifiedMessage(msg) {
				root := &certRoot{}
				t.mu.Lock()
				i.noteCertificateTree(t, root, judgedAt, nodes, class, sender, hash, self, k, msg)
				t.byHeight[judgedAt] = append(t.byHeight[judgedAt], root)
				t.mu.Unlock()
			}
		}
	}
	return rest
}

// noteCertificateTree files the verdicts of row `row` (= msg) and of everything below it, matching rows to decoded
// messages by position.  A subtree whose row count differs from the decoded count (the device refused the wrapper as
// non-canonical) is left to the stock route.  Caller holds t.mu.
func (i *IBFT) noteCertificateTree(t *certTable, root *certRoot, judgedAt uint64, nodes []ibftgpu.CertNode, class []byte, sender, hash, self []uint64, row int, msg *proto.IbftMessage) {
	undecided := ibftgpu.CertClassNeedsHost | ibftgpu.CertClassProposalHost
	if own := messages.ExtractProposal(msg); own != nil && class[row]&undecided == 0 {
		key := certHashKey{own, msg}
		t.hash[key] = certVerdict{ibftgpu.Bit(self, row), judgedAt} // validateProposalCommon's IsValidProposalHash
		root.hashes = append(root.hashes, key)
	}
	kids := nestedMessages(msg)
	nd := nodes[row]
	if int(nd.NChildren) != len(kids) || int(nd.FirstChild)+len(kids) > len(nodes) {
		return
	}
	last := messages.ExtractLastPreparedProposal(msg)
	for k, child := range kids {
		c := int(nd.FirstChild) + k
		if child == nil {
			continue
		}
		if class[c] == 0 {
			t.sender[child] = certVerdict{ibftgpu.Bit(sender, c), judgedAt}
			root.senders = append(root.senders, child)
		}
		if last != nil && class[row]&undecided == 0 && class[c]&ibftgpu.CertClassNeedsHost == 0 {
			key := certHashKey{last, child}
			t.hash[key] = certVerdict{ibftgpu.Bit(hash, c), judgedAt} // proposalMatchesCertificate
			root.hashes = append(root.hashes, key)
		}
		i.noteCertificateTree(t, root, judgedAt, nodes, class, sender, hash, self, c, child)
	}
}

// isValidValidatorTabled replaces i.backend.IsValidValidator(m) inside validPC / validateProposal.
func (i *IBFT) isValidValidatorTabled(m *proto.IbftMessage) bool {
	t := i.certificates()
	t.mu.Lock()
	v, known := t.sender[m]
	t.mu.Unlock()
	if known && v.judgedAt == i.state.getHeight() { // judged against the validator set that is in force now
		return v.ok
	}
	return i.backend.IsValidValidator(m)
}

// isValidProposalHashTabled replaces i.backend.IsValidProposalHash(proposal, hash) where hash was extracted from carrier
// (ExtractProposalHash / ExtractPrepareHash: nil when type and payload disagree — then the table is not consulted).
func (i *IBFT) isValidProposalHashTabled(proposal *proto.Proposal, carrier *proto.IbftMessage, hash []byte) bool {
	if hash != nil {
		t := i.certificates()
		t.mu.Lock()
		v, known := t.hash[certHashKey{proposal, carrier}]
		t.mu.Unlock()
		if known { // (a hash comparison does not depend on the validator set: valid at any height)
			return v.ok
		}
	}
	return i.backend.IsValidProposalHash(proposal, hash)
}
