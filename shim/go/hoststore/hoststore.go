//go:build ibftgpu

// Package hoststore binds libibft_host.so (include/ibft_host.h): the message store, the receive queue, the row path and
// the handle* walks that were MEASURED (DESIGN.md §5.5: a height of 8 191 wire messages end to end in 3.5 ms) — so that
// Go holds a handle, not 8 191 objects.  *Store implements core.Messages (/root/reference/core/ibft.go:23-46): it can be
// assigned to IBFT.messages as it is; the subscription half (Subscribe / Unsubscribe / SignalEvent,
// messages/event_manager.go) stays the reference's own Go code, everything that touches messages delegates to C.
//
// What a maintainer patches (INTEGRATION.md §5): the transport handler calls AddWireMessages(raw) instead of
// proto.Unmarshal + IBFT.AddMessage (core/ibft.go:1101-1123); handlePrepare / handleCommit call HandlePrepare /
// HandleCommit (core/ibft.go:855-889, 931-967) and get the quorum decision and the prepared messages / committed seals
// back; handleRoundChangeMessage / handlePrePrepare likewise (:470-512, :792-813).  shim/go/core/hoststore_glue.go holds
// those call sites.
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image).  tools/check_go_shim.py resolves every
// C.ibft_host_* call against include/ibft_host.h AND checks its argument count; tests/test_hoststore_sequence.py
// replays, through ctypes, exactly the C call sequence each method below makes (the sequences are read from this file).
package hoststore

/*
#cgo CFLAGS: -I${SRCDIR}
#cgo LDFLAGS: -libft_host -libftgpu
#include <stdlib.h>
#include "ibft_host.h"

// trampolines defined in callbacks.go (cgo: a file with //export may only DECLARE in its preamble)
int hoststoreMsgPred(void *user, uint8_t *wire, size_t len);
int hoststoreRccPred(void *user, uint64_t round, uint8_t *packed, size_t len, size_t n);
void hoststoreSignal(void *user, uint32_t type, uint64_t height, uint64_t round);
static inline ibft_host_msg_pred hoststore_msg_pred(void) { return (ibft_host_msg_pred)hoststoreMsgPred; }
static inline ibft_host_rcc_msgs_pred hoststore_rcc_pred(void) { return (ibft_host_rcc_msgs_pred)hoststoreRccPred; }
static inline ibft_host_signal_fn hoststore_signal(void) { return (ibft_host_signal_fn)hoststoreSignal; }
// a cgo.Handle is an integer: it crosses as one (C must not keep Go POINTERS, and the queue keeps `user`)
static inline void *hoststore_user(uintptr_t handle) { return (void *)handle; }
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"math/big"
	"runtime/cgo"
	"sort"
	"unsafe"

	"github.com/0xPolygon/go-ibft/ibftgpu"
	"github.com/0xPolygon/go-ibft/messages"
	"github.com/0xPolygon/go-ibft/messages/proto"
	protobuf "google.golang.org/protobuf/proto"
)

// Options of a Store.  MaxBatchRows / LingerMicros are ibft_host_queue_start's arguments (0 / 0 = 65 536 rows, no linger;
// 50 µs turns a burst into one device call per phase, profiles/r03q_queue_linger.txt).
type Options struct {
	MaxBatchRows uint32
	LingerMicros uint32
	DeviceQuorum bool // take hasQuorumByMsgType from the device (ibft_tally_prepare / ibft_tally) instead of the quorum index
}

// Store is one IBFT instance's message store in C.
type Store struct {
	*messages.Messages // Subscribe / Unsubscribe / SignalEvent: the reference's event manager, untouched
	h                  *C.ibft_host
	self               cgo.Handle
}

var ErrUnavailable = errors.New("hoststore: libibft_host refused the call")

// New creates the mirror, attaches the device context and starts the receive queue.
// C call sequence: ibft_host_new, ibft_host_attach_gpu, ibft_host_use_batch, ibft_host_enable_quorum_index,
// ibft_host_use_device_quorum, ibft_host_queue_start, ibft_host_queue_on_signal (on failure: ibft_host_queue_stop,
// ibft_host_free).
func New(gpu *ibftgpu.Ctx, o Options) (*Store, error) {
	h := C.ibft_host_new()
	if h == nil {
		return nil, ErrUnavailable
	}
	s := &Store{Messages: messages.NewMessages(), h: h}
	s.self = cgo.NewHandle(s)
	C.ibft_host_attach_gpu(h, (*C.ibft_ctx)(gpu.Handle()))
	C.ibft_host_use_batch(h, 1)
	C.ibft_host_enable_quorum_index(h)
	dq := C.int(0)
	if o.DeviceQuorum {
		dq = 1
	}
	C.ibft_host_use_device_quorum(h, dq)
	if C.ibft_host_queue_start(h, C.size_t(o.MaxBatchRows), C.uint32_t(o.LingerMicros)) != 0 {
		s.Close()
		return nil, ErrUnavailable
	}
	// SignalEvent(type, view) of core/ibft.go:1118-1119, from the queue's worker
	C.ibft_host_queue_on_signal(h, C.hoststore_signal(), C.hoststore_user(C.uintptr_t(s.self)))
	return s, nil
}

// Close stops the worker, frees the mirror and closes the embedded reference store (messages/messages.go:75: its event
// manager and subscriptions — this method shadows the promoted one, so it has to call it).
// C call sequence: ibft_host_queue_stop, ibft_host_free.
func (s *Store) Close() {
	if s.h != nil {
		C.ibft_host_queue_stop(s.h)
		C.ibft_host_free(s.h)
		s.h = nil
		s.self.Delete()
		// once, with the handle: the reference's eventManager.close() closes every subscription's doneCh and leaves the
		// subscriptions in the map, so a second call (a deferred Close plus an explicit one) would close a closed channel
		if s.Messages != nil {
			s.Messages.Close()
		}
	}
}

func ptr8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// SetValidators = ValidatorManager.Init (core/validator_manager.go:50-75) for the mirror's quorum rule; the device's
// table is set through ibftgpu.Ctx.SetValidators[Big] by the same caller.  Powers beyond 64 bits keep the Go
// ValidatorManager as the authority (DeviceQuorum then answers from the device's 256-bit table).
// C call sequence: ibft_host_vm_init.
func (s *Store) SetValidators(powers map[string]*big.Int) error {
	addrs := make([]string, 0, len(powers))
	for a := range powers {
		addrs = append(addrs, a)
	}
	sort.Strings(addrs)
	packed := make([]byte, 0, 24*len(addrs))
	pw := make([]uint64, len(addrs))
	for i, a := range addrs {
		if !powers[a].IsUint64() {
			return ErrUnavailable
		}
		pw[i] = powers[a].Uint64()
		packed = binary.LittleEndian.AppendUint32(packed, uint32(len(a)))
		packed = append(packed, a...)
	}
	var pp *C.uint64_t
	if len(pw) > 0 {
		pp = (*C.uint64_t)(unsafe.Pointer(&pw[0]))
	}
	if C.ibft_host_vm_init(s.h, ptr8(packed), C.size_t(len(packed)), pp, C.size_t(len(pw))) != 0 {
		return ErrUnavailable // errVotingPowerNotCorrect
	}
	return nil
}

// SetState mirrors the slice of core/state.go the hot path reads: the view and the accepted proposal message
// (nil while there is none).  C call sequence: ibft_host_set_state.
func (s *Store) SetState(height, round uint64, proposalMessage *proto.IbftMessage) error {
	var wire []byte
	if proposalMessage != nil {
		var err error
		if wire, err = protobuf.Marshal(proposalMessage); err != nil {
			return err
		}
	}
	if C.ibft_host_set_state(s.h, C.uint64_t(height), C.uint64_t(round), ptr8(wire), C.size_t(len(wire))) != 0 {
		return ErrUnavailable
	}
	return nil
}

const maxPushBytes = 0xFFFF0000 // kMaxCapBytes of the queue (go-ibft_amd/host/host_capi.cpp)

// AddWireMessages is the transport's entry point: the messages as they arrived, never unmarshalled in Go.  The bytes
// are copied into the queue; one worker ingests everything pending as one batch (one ibft_verify_messages_wire call:
// IsValidValidator + the handle* closure of every PREPARE / COMMIT of the view), stores the survivors as rows and
// signals quorum probes through SignalEvent.  Blocks only when the queue's caps are reached (back-pressure).
// C call sequence: ibft_host_queue_push.
func (s *Store) AddWireMessages(raw [][]byte) error {
	if len(raw) == 0 {
		return nil
	}
	total := 0
	for _, m := range raw {
		total += len(m)
	}
	// offsets are 32 bits on the C side: a batch past that is split in two pushes.  A single message that large can
	// never be pushed (the queue's own byte cap, ibft_host_queue_set_caps, is far below it): every message is checked
	// BEFORE anything is queued, so that a refusal never leaves the first half of the batch ingested and the caller unable
	// to tell (ADVICE round 5).
	if uint64(total) > maxPushBytes {
		for _, m := range raw {
			if uint64(len(m)) > maxPushBytes {
				return ErrUnavailable
			}
		}
		if err := s.AddWireMessages(raw[:len(raw)/2]); err != nil {
			return err
		}
		return s.AddWireMessages(raw[len(raw)/2:])
	}
	wire := make([]byte, 0, total)
	off := make([]uint32, 1, len(raw)+1)
	for _, m := range raw {
		wire = append(wire, m...)
		off = append(off, uint32(len(wire)))
	}
	if C.ibft_host_queue_push(s.h, ptr8(wire), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.size_t(len(raw))) != 0 {
		return ErrUnavailable
	}
	return nil
}

// QueueStats mirrors ibft_host_queue_stats.
type QueueStats struct {
	Pushed, Ingested, Stored, Rejected, Undecodable uint64
	Batches, DeviceCalls, CacheHits, MaxBatchRows   uint64
}

// Drain waits until everything pushed before the call has been ingested.  C call sequence: ibft_host_queue_drain.
func (s *Store) Drain() (QueueStats, error) {
	var st C.ibft_host_queue_stats
	if C.ibft_host_queue_drain(s.h, &st) != 0 {
		return QueueStats{}, ErrUnavailable
	}
	return QueueStats{uint64(st.pushed), uint64(st.ingested), uint64(st.stored), uint64(st.rejected), uint64(st.undecodable),
		uint64(st.batches), uint64(st.device_calls), uint64(st.cache_hits), uint64(st.max_batch_rows)}, nil
}

// ---- core.Messages (core/ibft.go:23-46) -----------------------------------------------------------------------

// AddMessage stores an already verified message (messages/messages.go:54-65; IBFT.AddMessage has judged it).
// C call sequence: ibft_host_store_add.
func (s *Store) AddMessage(message *proto.IbftMessage) {
	wire, err := protobuf.Marshal(message)
	if err != nil {
		return
	}
	C.ibft_host_store_add(s.h, ptr8(wire), C.size_t(len(wire)))
}

// PruneByHeight = messages/messages.go:123-148.  C call sequence: ibft_host_store_prune.
func (s *Store) PruneByHeight(height uint64) {
	C.ibft_host_store_prune(s.h, C.uint64_t(height))
}

type predicates struct {
	msg func(*proto.IbftMessage) bool
	rcc func(round uint64, msgs []*proto.IbftMessage) bool
}

// C call sequence: ibft_host_buf_free.
func decodeList(b *C.ibft_host_buf) []*proto.IbftMessage {
	defer C.ibft_host_buf_free(b)
	return decodePacked(unsafe.Slice((*byte)(unsafe.Pointer(b.data)), int(b.len)), int(b.count))
}

// a message list as it crosses the ABI: repeated { u32 little-endian length, bytes }
func decodePacked(data []byte, count int) []*proto.IbftMessage {
	out := make([]*proto.IbftMessage, 0, count)
	for at := 0; at+4 <= len(data); {
		n := int(binary.LittleEndian.Uint32(data[at:]))
		at += 4
		m := &proto.IbftMessage{}
		if protobuf.Unmarshal(data[at:at+n], m) == nil {
			out = append(out, m)
		}
		at += n
	}
	return out
}

// GetValidMessages = messages/messages.go:169-199 for callers OUTSIDE the hot path (the walk decodes the view's rows;
// handlePrepare / handleCommit use HandlePrepare / HandleCommit below and never do).
// C call sequence: ibft_host_store_get_valid, ibft_host_buf_free.
func (s *Store) GetValidMessages(view *proto.View, messageType proto.MessageType,
	isValid func(*proto.IbftMessage) bool) []*proto.IbftMessage {
	p := cgo.NewHandle(&predicates{msg: isValid})
	defer p.Delete()
	var out C.ibft_host_buf
	if C.ibft_host_store_get_valid(s.h, C.uint64_t(view.Height), C.uint64_t(view.Round), C.uint32_t(messageType),
		C.hoststore_msg_pred(), C.hoststore_user(C.uintptr_t(p)), &out) != 0 {
		return nil
	}
	return decodeList(&out)
}

// GetExtendedRCC = messages/messages.go:202-245 (isValidRCC receives the candidate messages, decoded).
// C call sequence: ibft_host_store_get_extended_rcc_msgs, ibft_host_buf_free.
func (s *Store) GetExtendedRCC(height uint64, isValidMessage func(*proto.IbftMessage) bool,
	isValidRCC func(round uint64, msgs []*proto.IbftMessage) bool) []*proto.IbftMessage {
	p := cgo.NewHandle(&predicates{msg: isValidMessage, rcc: isValidRCC})
	defer p.Delete()
	var out C.ibft_host_buf
	if C.ibft_host_store_get_extended_rcc_msgs(s.h, C.uint64_t(height), C.hoststore_msg_pred(), C.hoststore_rcc_pred(),
		C.hoststore_user(C.uintptr_t(p)), &out) != 0 {
		return nil
	}
	return decodeList(&out)
}

// GetMostRoundChangeMessages = messages/messages.go:249-286.
// C call sequence: ibft_host_store_get_most_rc, ibft_host_buf_free.
func (s *Store) GetMostRoundChangeMessages(minRound, height uint64) []*proto.IbftMessage {
	var out C.ibft_host_buf
	if C.ibft_host_store_get_most_rc(s.h, C.uint64_t(minRound), C.uint64_t(height), &out) != 0 {
		return nil
	}
	return decodeList(&out)
}

// ---- the handle* walks (what was measured) ----------------------------------------------------------------------

// HandlePrepare = core/ibft.go:855-889 up to finalizePrepare: the closure of every stored PREPARE was settled when it
// arrived, the quorum is HasPrepareQuorum (core/validator_manager.go:99-127).  prepared = the wire bytes of the surviving
// messages (PreparedCertificate.PrepareMessages, decoded only if a ROUND_CHANGE is ever built from them).
// C call sequence: ibft_host_handle_prepare, ibft_host_buf_free.
func (s *Store) HandlePrepare(view *proto.View) (quorum bool, prepared [][]byte) {
	var out C.ibft_host_buf
	rc := C.ibft_host_handle_prepare(s.h, C.uint64_t(view.Height), C.uint64_t(view.Round), &out)
	defer C.ibft_host_buf_free(&out)
	if rc != 1 {
		return false, nil
	}
	data := unsafe.Slice((*byte)(unsafe.Pointer(out.data)), int(out.len))
	for at := 0; at+4 <= len(data); {
		n := int(binary.LittleEndian.Uint32(data[at:]))
		at += 4
		prepared = append(prepared, append([]byte(nil), data[at:at+n]...))
		at += n
	}
	return true, prepared
}

// HandleCommit = core/ibft.go:931-967: quorum over the COMMITs whose hash and committed seal verified (the device judged
// both when they arrived), seals = ExtractCommittedSeals of exactly the survivors (messages/helpers.go:22-35) — what
// Backend.InsertProposal receives (core/backend.go:78-81).
// C call sequence: ibft_host_handle_commit, ibft_host_buf_free.
func (s *Store) HandleCommit(view *proto.View) (quorum bool, seals []*messages.CommittedSeal) {
	var out C.ibft_host_buf
	rc := C.ibft_host_handle_commit(s.h, C.uint64_t(view.Height), C.uint64_t(view.Round), &out)
	defer C.ibft_host_buf_free(&out)
	if rc != 1 {
		return false, nil
	}
	// packed as repeated { u8 present, u32 signer_len, signer, u32 sig_len, sig } (include/ibft_host.h)
	data := unsafe.Slice((*byte)(unsafe.Pointer(out.data)), int(out.len))
	for at := 0; at < len(data); {
		present := data[at]
		at++
		if present == 0 {
			seals = append(seals, nil)
			continue
		}
		n := int(binary.LittleEndian.Uint32(data[at:]))
		signer := append([]byte(nil), data[at+4:at+4+n]...)
		at += 4 + n
		n = int(binary.LittleEndian.Uint32(data[at:]))
		sig := append([]byte(nil), data[at+4:at+4+n]...)
		at += 4 + n
		seals = append(seals, &messages.CommittedSeal{Signer: signer, Signature: sig})
	}
	return true, seals
}

// HandleRoundChange = handleRoundChangeMessage (core/ibft.go:470-512): the extended RCC for (height, round) or nil; every
// certificate was judged from the device's rows when its carrier arrived.
// C call sequence: ibft_host_handle_round_change, ibft_host_buf_free.
func (s *Store) HandleRoundChange(view *proto.View) []*proto.IbftMessage {
	var out C.ibft_host_buf
	rc := C.ibft_host_handle_round_change(s.h, C.uint64_t(view.Height), C.uint64_t(view.Round), &out)
	msgs := decodeList(&out) // (frees the buffer whatever the answer)
	if rc != 1 {
		return nil
	}
	return msgs
}

// HandlePrePrepare = handlePrePrepare (core/ibft.go:792-813): the first stored PREPREPARE of the view that passes
// validateProposal (:683-788), or nil.  C call sequence: ibft_host_handle_preprepare, ibft_host_buf_free.
func (s *Store) HandlePrePrepare(view *proto.View) *proto.IbftMessage {
	var out C.ibft_host_buf
	rc := C.ibft_host_handle_preprepare(s.h, C.uint64_t(view.Height), C.uint64_t(view.Round), &out)
	msgs := decodeList(&out) // (frees the buffer whatever the answer)
	if rc != 1 || len(msgs) == 0 {
		return nil
	}
	return msgs[0]
}

// RowsKept reports how many stored messages were never decoded (rows).  C call sequence: ibft_host_rows_kept.
func (s *Store) RowsKept() int { return int(C.ibft_host_rows_kept(s.h)) }

// DeviceQuorumStats: decisions the device took, decisions on which the mirror's quorum index disagreed (must stay 0).
// C call sequence: ibft_host_device_quorum_stats.
func (s *Store) DeviceQuorumStats() (calls, mismatches int) {
	var a, b C.size_t
	C.ibft_host_device_quorum_stats(s.h, &a, &b)
	return int(a), int(b)
}
