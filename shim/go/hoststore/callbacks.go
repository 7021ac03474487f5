//go:build ibftgpu

package hoststore

/*
#include <stddef.h>
#include <stdint.h>
*/
import "C"

import (
	"runtime/cgo"
	"unsafe"

	"github.com/0xPolygon/go-ibft/messages/proto"
	protobuf "google.golang.org/protobuf/proto"
)

// The C side calls back with the integer of a cgo.Handle in `user` (hoststore.go: hoststore_user).

//export hoststoreMsgPred
func hoststoreMsgPred(user unsafe.Pointer, wire *C.uint8_t, n C.size_t) C.int {
	p := cgo.Handle(uintptr(user)).Value().(*predicates)
	m := &proto.IbftMessage{}
	if protobuf.Unmarshal(unsafe.Slice((*byte)(unsafe.Pointer(wire)), int(n)), m) != nil {
		return 0
	}
	if p.msg == nil || p.msg(m) {
		return 1
	}
	return 0
}

//export hoststoreRccPred
func hoststoreRccPred(user unsafe.Pointer, round C.uint64_t, packed *C.uint8_t, length, n C.size_t) C.int {
	p := cgo.Handle(uintptr(user)).Value().(*predicates)
	if p.rcc == nil {
		return 1
	}
	if p.rcc(uint64(round), decodePacked(unsafe.Slice((*byte)(unsafe.Pointer(packed)), int(length)), int(n))) {
		return 1
	}
	return 0
}

// SignalEvent(type, view) of core/ibft.go:1118-1119, called by the queue's worker WITHOUT the mirror's lock: the
// subscriber (the round goroutine) may call HandlePrepare / HandleCommit at once.
//
//export hoststoreSignal
func hoststoreSignal(user unsafe.Pointer, msgType C.uint32_t, height, round C.uint64_t) {
	s := cgo.Handle(uintptr(user)).Value().(*Store)
	s.SignalEvent(proto.MessageType(msgType), &proto.View{Height: uint64(height), Round: uint64(round)})
}
