//go:build ibftgpu

// soa.go — lives in package messages next to messages.go.  Flattens the messages of one
// (view, type) into the byte columns the C ABI takes: the SoA mirror of the store.
//
// NOT COMPILED HERE (no Go toolchain in the build image); the same flattening is implemented
// and tested in C++: go-ibft_amd/host/backend.cpp (flatten_commits / flatten_senders).
package messages

import "github.com/0xPolygon/go-ibft/messages/proto"

// SealColumns are the inputs of ibft_verify_hashes + ibft_verify_seals for COMMIT messages.
type SealColumns struct {
	Hash32, HashLen, Sig65, Signer20, PreFlags []byte
}

const (
	rowNil    = 0x01
	rowBadLen = 0x02
)

func putFixed(col []byte, src []byte, width int) ([]byte, bool) {
	if len(src) == width {
		return append(col, src...), true
	}
	return append(col, make([]byte, width)...), false
}

// FlattenCommits mirrors the closure of handleCommit (core/ibft.go:932-944): the hash is
// ExtractCommitHash (nil for a wrong type/payload), the seal ExtractCommittedSeal (nil for a
// wrong payload).  Rows the crypto can never accept are pre-flagged.
func FlattenCommits(msgs []*proto.IbftMessage) *SealColumns {
	c := &SealColumns{}
	for _, m := range msgs {
		hash := ExtractCommitHash(m)
		seal := ExtractCommittedSeal(m)
		var pre byte
		c.Hash32, _ = putFixed(c.Hash32, hash, 32)
		hl := len(hash)
		if hl > 255 {
			hl = 255
		}
		c.HashLen = append(c.HashLen, byte(hl))
		if hash == nil || seal == nil {
			pre |= rowNil
		}
		var sig, signer []byte
		if seal != nil {
			sig, signer = seal.Signature, seal.Signer
		}
		var ok bool
		if c.Sig65, ok = putFixed(c.Sig65, sig, 65); !ok {
			pre |= rowBadLen
		}
		if c.Signer20, ok = putFixed(c.Signer20, signer, 20); !ok {
			pre |= rowBadLen
		}
		c.PreFlags = append(c.PreFlags, pre)
	}
	return c
}

// FlattenPrepares mirrors the closure of handlePrepare (core/ibft.go:856-862): only the hash ExtractPrepareHash
// returns (nil for a wrong type/payload) and its length; the other columns stay empty.
func FlattenPrepares(msgs []*proto.IbftMessage) *SealColumns {
	c := &SealColumns{}
	for _, m := range msgs {
		hash := ExtractPrepareHash(m)
		c.Hash32, _ = putFixed(c.Hash32, hash, 32)
		hl := len(hash)
		if hl > 255 {
			hl = 255
		}
		c.HashLen = append(c.HashLen, byte(hl))
	}
	return c
}

// SenderColumns are the inputs of ibft_verify_senders.
type SenderColumns struct {
	Payload  []byte
	Off      []uint32
	Sig65    []byte
	From20   []byte
	PreFlags []byte
}

func FlattenSenders(msgs []*proto.IbftMessage) (*SenderColumns, error) {
	c := &SenderColumns{Off: []uint32{0}}
	for _, m := range msgs {
		raw, err := m.PayloadNoSig() // messages/proto/helper.go:12-27
		if err != nil {
			return nil, err
		}
		c.Payload = append(c.Payload, raw...)
		c.Off = append(c.Off, uint32(len(c.Payload)))
		var pre byte
		var ok bool
		if c.Sig65, ok = putFixed(c.Sig65, m.Signature, 65); !ok {
			pre |= rowBadLen
		}
		if c.From20, ok = putFixed(c.From20, m.From, 20); !ok {
			pre |= rowBadLen
		}
		c.PreFlags = append(c.PreFlags, pre)
	}
	return c, nil
}

// GetValidMessagesBatch is GetValidMessages (messages.go:169-199) with ONE verdict callback for
// the whole view instead of one per message: same lock, same prune-on-invalid.
func (ms *Messages) GetValidMessagesBatch(
	view *proto.View,
	messageType proto.MessageType,
	verdicts func(msgs []*proto.IbftMessage) []bool,
) []*proto.IbftMessage {
	mux := ms.muxMap[messageType]
	mux.Lock()
	defer mux.Unlock()

	messages := ms.getProtoMessages(view, messageType)
	keys := make([]string, 0, len(messages))
	all := make([]*proto.IbftMessage, 0, len(messages))
	for key, message := range messages {
		keys = append(keys, key)
		all = append(all, message)
	}
	ok := verdicts(all)
	if len(ok) != len(all) { // backend failure: prune nothing, return nothing
		return nil
	}
	valid := make([]*proto.IbftMessage, 0, len(all))
	for i, message := range all {
		if !ok[i] {
			delete(messages, keys[i])
			continue
		}
		valid = append(valid, message)
	}
	return valid
}
