//go:build ibftgpu

// wire_ingest.go — lives in package core.  The receive side of SURVEY.md §8f rank 3: the transport's
// handler collects the raw protobuf bytes that arrived in the last few hundred microseconds and has
// the device vouch for them (canonical-form walk, Keccak of PayloadNoSig, sender recover) before any
// message is unmarshalled; only survivors — and the rows the device declined to judge — reach
// proto.Unmarshal and IBFT.AddMessage.
//
// NOT COMPILED HERE (no Go toolchain in the build image).  The identical merge — device verdicts for
// the rows it vouches for, stock route for the rest — is implemented and tested in C++:
// go-ibft_amd/host/backend.cpp (GpuBackend::VerifySendersWire), tests/test_gpu_wire.py.
package core

import (
	goproto "google.golang.org/protobuf/proto"

	"github.com/0xPolygon/go-ibft/ibftgpu"
	"github.com/0xPolygon/go-ibft/messages/proto"
)

// WireVerifier is the optional interface a Backend offers next to BatchVerifier.
type WireVerifier interface {
	// VerifySendersWire answers IsValidValidator for raw messages; rows[i].Status tells whether row i
	// was judged at all.
	VerifySendersWire(wire []byte, off []uint32) (mask []uint64, rows []ibftgpu.WireRow, ok bool)
}

// WireSetVerifier judges raw messages COMPLETELY (libibftgpu: ibft_verify_messages_wire): sender bit i is
// IsValidValidator(message i); for the canonical PREPARE / COMMIT messages of the view (height, round) — class[i] has
// ibftgpu.WireClassClosure — valid bit i is the handlePrepare / handleCommit closure against proposal, both
// signatures of a COMMIT verified in the same launch.  class[i] with ibftgpu.WireClassNeedsHost: not judged, stock route.
type WireSetVerifier interface {
	VerifyMessagesWire(wire []byte, off []uint32, height, round uint64, proposal *proto.Proposal) (sender, valid []uint64, class []byte, ok bool)
}

// AddWireMessages is what the transport calls instead of unmarshalling and calling AddMessage once per
// message.  raw[i] is one message exactly as received.
func (i *IBFT) AddWireMessages(raw [][]byte) {
	if len(raw) == 0 {
		return
	}
	if cv, hasCerts := i.backend.(CertificateVerifier); hasCerts {
		i.dropCertificateVerdicts(i.state.getHeight()) // what was judged at earlier heights goes with the store's prune
		// PREPREPARE / ROUND_CHANGE messages first: their own envelope and every message nested in them, one call
		if raw = i.addWireCertificates(cv, raw); len(raw) == 0 {
			return
		}
	}
	if ws, hasSets := i.backend.(WireSetVerifier); hasSets && i.state.getProposal() != nil && !tooFewForDevice(len(raw)) {
		if i.addWireSets(ws, raw) {
			return
		}
	}
	wv, hasWire := i.backend.(WireVerifier)
	if !hasWire || tooFewForDevice(len(raw)) { // (MinDeviceRows, backend_batch.go: a handful of messages stays on the host)
		i.addWireStock(raw, nil)
		return
	}
	wire, off := concat(raw)
	mask, rows, ok := wv.VerifySendersWire(wire, off)
	if !ok { // device unavailable: the stock route for everything
		i.addWireStock(raw, nil)
		return
	}
	var stock [][]byte
	for k := range raw {
		switch {
		case rows[k].Status == ibftgpu.WireNeedsHost:
			stock = append(stock, raw[k]) // PREPREPARE / ROUND_CHANGE, unknown fields, odd encodings
		case !ibftgpu.Bit(mask, k):
			// IsValidValidator == false: dropped exactly where isAcceptableMessage would have dropped it
		default:
			msg := new(proto.IbftMessage)
			if goproto.Unmarshal(raw[k], msg) == nil {
				i.addVerifiedMessage(msg) // AddMessage minus the IsValidValidator call (core/ibft.go:1128)
			}
		}
	}
	i.addWireStock(stock, nil)
}

// addWireSets: one device call settles IsValidValidator for the batch AND the handle* closure of every PREPARE /
// COMMIT of the current view; the closure verdicts go into the table commitMessagesFor / prepareMessagesFor consult
// (message_sets.go).  false = device unavailable, nothing was added.
func (i *IBFT) addWireSets(ws WireSetVerifier, raw [][]byte) bool {
	proposal := i.state.getProposal()
	wire, off := concat(raw)
	sender, valid, class, ok := ws.VerifyMessagesWire(wire, off, i.state.getHeight(), i.state.getRound(), proposal)
	if !ok || len(class) != len(raw) {
		return false
	}
	var stock [][]byte
	tab := i.closures()
	for k := range raw {
		switch {
		case class[k]&ibftgpu.WireClassNeedsHost != 0:
			stock = append(stock, raw[k])
		case !ibftgpu.Bit(sender, k):
			// IsValidValidator == false: dropped
		default:
			msg := new(proto.IbftMessage)
			if goproto.Unmarshal(raw[k], msg) != nil {
				continue
			}
			if class[k]&ibftgpu.WireClassClosure != 0 {
				tab.mu.Lock()
				tab.sync(proposal)
				tab.verdict[msg] = ibftgpu.Bit(valid, k)
				tab.mu.Unlock()
			}
			i.addVerifiedMessage(msg)
		}
	}
	i.addWireStock(stock, nil)
	return true
}

// addVerifiedMessage is AddMessage (core/ibft.go:1101-1123) for a message whose sender the device has already
// vouched for: isAcceptableMessage (core/ibft.go:1126-1149) minus its first check, then the same store +
// quorum probe + signal.
func (i *IBFT) addVerifiedMessage(message *proto.IbftMessage) (stored bool) {
	if message == nil || !i.isAcceptableView(message) {
		return false
	}
	i.messages.AddMessage(message)
	if message.View.Height == i.state.getHeight() {
		msgs := i.messages.GetValidMessages(
			message.View,
			message.Type,
			func(_ *proto.IbftMessage) bool { return true })
		if i.hasQuorumByMsgType(msgs, message.Type) {
			i.messages.SignalEvent(message.Type, message.View)
		}
	}
	return true
}

// isAcceptableView is the part of isAcceptableMessage (core/ibft.go:1132-1148) that follows the sender check.
func (i *IBFT) isAcceptableView(message *proto.IbftMessage) bool {
	if message.View == nil {
		return false
	}
	if i.state.getHeight() > message.View.Height {
		return false
	}
	if i.state.getHeight() == message.View.Height {
		return message.View.Round >= i.state.getRound()
	}
	return true
}

// addWireStock is the unchanged path: unmarshal, then AddMessage (which calls IsValidValidator itself).
func (i *IBFT) addWireStock(raw [][]byte, _ any) {
	for _, b := range raw {
		msg := new(proto.IbftMessage)
		if goproto.Unmarshal(b, msg) == nil {
			i.AddMessage(msg)
		}
	}
}

func concat(raw [][]byte) ([]byte, []uint32) {
	off := make([]uint32, len(raw)+1)
	n := 0
	for k, b := range raw {
		off[k] = uint32(n)
		n += len(b)
	}
	off[len(raw)] = uint32(n)
	wire := make([]byte, 0, n)
	for _, b := range raw {
		wire = append(wire, b...)
	}
	return wire, off
}
