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

// AddWireMessages is what the transport calls instead of unmarshalling and calling AddMessage once per
// message.  raw[i] is one message exactly as received.
func (i *IBFT) AddWireMessages(raw [][]byte) {
	wv, hasWire := i.backend.(WireVerifier)
	if !hasWire || len(raw) == 0 {
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

// addVerifiedMessage is AddMessage (core/ibft.go:1101-1123) for a message whose sender the device has already
// vouched for: isAcceptableMessage (core/ibft.go:1126-1149) minus its first check, then the same store +
// quorum probe + signal.
func (i *IBFT) addVerifiedMessage(message *proto.IbftMessage) {
	if message == nil || !i.isAcceptableView(message) {
		return
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
