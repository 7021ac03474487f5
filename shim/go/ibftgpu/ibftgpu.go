//go:build ibftgpu

// Package ibftgpu is the thin cgo binding of libibftgpu.so (include/ibftgpu.h).
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain.  It is the
// reference-side stub a go-ibft maintainer adds; the C side it binds is exercised through the
// same C ABI by tests/ (ctypes) and by go-ibft_amd/host (C++).  tools/check_go_shim.py checks that
// every identifier these files use exists — in this tree, in include/ibftgpu.h or in the reference.
//
// Placement: shim/go/ is an OVERLAY on the go-ibft module root — shim/go/ibftgpu → <module>/ibftgpu
// (import path github.com/0xPolygon/go-ibft/ibftgpu), shim/go/core/*.go → core/, shim/go/messages/*.go →
// messages/ — with include/ibftgpu.h copied next to this file and libibftgpu.so on the linker /
// loader path (CGO_LDFLAGS=-L…, LD_LIBRARY_PATH).
//
// cgo rules honoured: every slice handed to C is a flat []byte/[]uint32/[]uint64 without Go
// pointers inside; C copies it to HBM before returning and retains nothing.
package ibftgpu

/*
#cgo CFLAGS: -I${SRCDIR}
#cgo LDFLAGS: -libftgpu
#include <stdlib.h>
#include "ibftgpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/big"
	"runtime"
	"unsafe"
)

// ErrFallback tells the caller to run its own per-message Verifier for this batch.
// The library never synthesises verdicts on failure (all-false stalls liveness,
// all-true breaks safety).
var ErrFallback = errors.New("ibftgpu: device path unavailable, use the CPU verifier")

const (
	RowNil     = C.IBFT_ROW_NIL
	RowBadLen  = C.IBFT_ROW_BADLEN
	RowHashBad = C.IBFT_ROW_HASH_BAD
)

// Tally mirrors ibft_tally_t (ValidatorManager.HasQuorum over the valid rows).
type Tally struct {
	QuorumLo, QuorumHi uint64
	PowerLo, PowerHi   uint64
	ValidRows          uint32
	DistinctSenders    uint32
	HasQuorum          bool
	// sharded calls: how often a validator had valid rows in more than one shard (it is counted once)
	ShardOverlap uint32
	// HasPrepareQuorum calls (a proposer was given): valid rows sent BY the proposer — any such row makes HasQuorum false
	// (core/validator_manager.go:114-121)
	ProposerRows uint32
}

// Ctx owns one ibft_ctx (one HIP stream, resident columns on one GPU).  Use one Ctx per
// concurrent caller — AddMessage goroutines, the round goroutine and the two watcher goroutines of
// core/ibft.go:335-347 — from a FIXED set (a buffered channel of contexts, not a sync.Pool: the validator
// table and the key cache live in the context, so SetValidators must reach every one of them and none
// may be dropped by the garbage collector; INTEGRATION.md §2).
type Ctx struct {
	h *C.ibft_ctx
	// the Go slices of a batch handed to SealsStageNext: the library's copy stream reads them AFTER the cgo call has
	// returned, so they are pinned (runtime.Pinner: the collector neither moves nor frees them) until SealsSwap has waited
	// for that copy
	next runtime.Pinner
}

// Options mirrors ibft_cfg.  KeyCache turns on the warm path (IBFT_FLAG_PUBKEY_CACHE): the
// first valid signature of a validator teaches the device its public key, later ones are verified
// against a per-validator table in HBM (655 KB per validator) — identical verdicts, 4-10x less work.
type Options struct {
	Device     int
	MaxRows    uint32
	StrictLowS bool
	KeyCache   bool
}

// abiVersion is what this binding was written against (include/ibftgpu.h: ibft_version; 2 = ibft_tally_t with
// proposer_rows, 56 bytes: an older library would write a shorter struct; 3 = the staged / pipelined pass calls,
// ibft_seals_rows, ibft_issue_probe: a version-2 library lacks symbols this file links against; 4 = ibft_pipeline_stats).
const abiVersion = 4

func New(o Options) (*Ctx, error) {
	if v := int(C.ibft_version()); v < abiVersion {
		return nil, fmt.Errorf("%w: libibftgpu ABI version %d, binding needs >= %d", ErrFallback, v, abiVersion)
	}
	cfg := C.ibft_cfg{device: C.int32_t(o.Device), max_rows: C.uint32_t(o.MaxRows)}
	if o.StrictLowS {
		cfg.flags |= C.IBFT_FLAG_STRICT_LOW_S
	}
	if o.KeyCache {
		cfg.flags |= C.IBFT_FLAG_PUBKEY_CACHE
	}
	var h *C.ibft_ctx
	if rc := C.ibft_ctx_create(&cfg, &h); rc != C.IBFT_OK {
		return nil, fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	c := &Ctx{h: h}
	runtime.SetFinalizer(c, (*Ctx).Close)
	return c, nil
}

func (c *Ctx) Close() {
	if c.h != nil {
		C.ibft_ctx_destroy(c.h) // (drains the context's streams: nothing reads the pinned slices any more)
		c.h = nil
	}
	c.next.Unpin()
}

// SetSealDigest tells the context WHAT a committed seal signs in this Backend (core/backend.go:53-55 leaves it open):
// suffix == nil: the 32-byte proposalHash itself (the default); otherwise keccak256(proposalHash ‖ suffix) — e.g. a Backend
// that appends byte(proto.MessageType_COMMIT) before hashing passes []byte{2}.  Every a2 route of the context follows
// (seal batches, the seal half of message sets — a1 keeps comparing the carried hash — and the batch signer).
func (c *Ctx) SetSealDigest(suffix []byte) error {
	mode := C.uint32_t(C.IBFT_SEAL_DIGEST_IDENTITY)
	if suffix != nil {
		mode = C.IBFT_SEAL_DIGEST_KECCAK_SUFFIX
	}
	return c.check(C.ibft_set_seal_digest(c.h, mode, ptr8(suffix), C.size_t(len(suffix))))
}

// Handle is the ibft_ctx pointer for another cgo package of this module (hoststore attaches it to the host mirror:
// ibft_host_attach_gpu); the context stays owned by c.
func (c *Ctx) Handle() unsafe.Pointer { return unsafe.Pointer(c.h) }

// HasQuorum = ValidatorManager.HasQuorum over the rows of `mask` (ibft_tally, core/validator_manager.go:77-96).
func (c *Ctx) HasQuorum(sender20 []byte, mask []uint64) (Tally, error) {
	var t C.ibft_tally_t
	rc := C.ibft_tally(c.h, ptr8(sender20), (*C.uint64_t)(unsafe.Pointer(&mask[0])), C.size_t(len(sender20)/20), &t)
	return tally(t), c.check(rc)
}

// HasPrepareQuorum = ValidatorManager.HasPrepareQuorum (ibft_tally_prepare, core/validator_manager.go:99-127): proposer20
// is proposalMessage.From; the caller answers false itself when there is no proposal message (:101-110).
func (c *Ctx) HasPrepareQuorum(sender20 []byte, mask []uint64, proposer20 []byte) (Tally, error) {
	var t C.ibft_tally_t
	rc := C.ibft_tally_prepare(c.h, ptr8(sender20), (*C.uint64_t)(unsafe.Pointer(&mask[0])), C.size_t(len(sender20)/20),
		ptr8(proposer20), &t)
	return tally(t), c.check(rc)
}

func ptr8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

func (c *Ctx) check(rc C.int) error {
	if rc == C.IBFT_OK {
		return nil
	}
	return fmt.Errorf("%w: %s %s", ErrFallback, C.GoString(C.ibft_strerror(rc)), C.GoString(C.ibft_last_error(c.h)))
}

// SetValidators uploads the table GetVotingPowers(height) returned (20-byte addresses, u64
// powers; SetValidatorsBig takes the *big.Int values as they are).
func (c *Ctx) SetValidators(height uint64, addrs20 []byte, power []uint64) error {
	n := len(power)
	var pp *C.uint64_t
	if n > 0 {
		pp = (*C.uint64_t)(unsafe.Pointer(&power[0]))
	}
	return c.check(C.ibft_set_validators(c.h, C.uint64_t(height), ptr8(addrs20), pp, C.size_t(n)))
}

// SetValidatorsBig is SetValidators for map[string]*big.Int as ValidatorBackend.GetVotingPowers returns it
// (core/validator_manager.go:17-31): every power must fit 256 bits (a larger one returns ErrFallback and the
// set stays on the Go path).  Sums and the quorum are kept in 320 bits on the device.
func (c *Ctx) SetValidatorsBig(height uint64, addrs20 []byte, power []*big.Int) error {
	be := make([]byte, 32*len(power))
	for i, p := range power {
		if p.Sign() < 0 || p.BitLen() > 256 {
			return fmt.Errorf("%w: voting power %d does not fit 256 bits", ErrFallback, i)
		}
		p.FillBytes(be[32*i : 32*i+32])
	}
	return c.check(C.ibft_set_validators_u256(c.h, C.uint64_t(height), ptr8(addrs20), ptr8(be), C.size_t(len(power))))
}

// TallyWide is the full-width result of the last tally (ibft_last_tally_wide).
func (c *Ctx) TallyWide() (power, quorum *big.Int, hasQuorum bool, err error) {
	var t C.ibft_tally_wide_t
	if err = c.check(C.ibft_last_tally_wide(c.h, &t)); err != nil {
		return nil, nil, false, err
	}
	return words(t.power[:]), words(t.quorum[:]), t.has_quorum != 0, nil
}

func words(w []C.uint64_t) *big.Int {
	v := new(big.Int)
	for i := len(w) - 1; i >= 0; i-- {
		v.Lsh(v, 64).Or(v, new(big.Int).SetUint64(uint64(w[i])))
	}
	return v
}

// VerifyHashes = IsValidProposalHash over a batch (core/ibft.go:858-861, 938).
func (c *Ctx) VerifyHashes(raw []byte, round uint64, hash32, hashLen []byte) ([]uint64, error) {
	n := len(hashLen)
	mask := make([]uint64, (n+63)/64+1)
	rc := C.ibft_verify_hashes(c.h, ptr8(raw), C.size_t(len(raw)), C.uint64_t(round), ptr8(hash32), ptr8(hashLen),
		C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&mask[0])))
	return mask, c.check(rc)
}

// VerifySeals = IsValidCommittedSeal over a batch (core/ibft.go:943).
func (c *Ctx) VerifySeals(hash32, sig65, signer20, preFlags []byte) ([]uint64, Tally, error) {
	n := len(sig65) / 65
	mask := make([]uint64, (n+63)/64+1)
	var t C.ibft_tally_t
	rc := C.ibft_verify_seals(c.h, ptr8(hash32), ptr8(sig65), ptr8(signer20), ptr8(preFlags), C.size_t(n),
		(*C.uint64_t)(unsafe.Pointer(&mask[0])), &t)
	return mask, tally(t), c.check(rc)
}

// VerifySenders = IsValidValidator over a batch (core/ibft.go:1128); payload is the
// concatenation of msg.PayloadNoSig(), off its n+1 offsets.
func (c *Ctx) VerifySenders(payload []byte, off []uint32, sig65, from20, preFlags []byte) ([]uint64, Tally, error) {
	n := len(off) - 1
	mask := make([]uint64, (n+63)/64+1)
	var t C.ibft_tally_t
	rc := C.ibft_verify_senders(c.h, ptr8(payload), (*C.uint32_t)(unsafe.Pointer(&off[0])), ptr8(sig65), ptr8(from20),
		ptr8(preFlags), C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&mask[0])), &t)
	return mask, tally(t), c.check(rc)
}

// VerifyMessages = a whole PREPARE / COMMIT set in one call (ibft_verify_messages): senderMask bit i is
// IsValidValidator(message i), validMask bit i the handlePrepare / handleCommit closure for it, the tally is
// HasQuorum over the rows with both bits — ValidatorManager.HasPrepareQuorum (core/validator_manager.go:99-127) when
// proposer20 (the accepted proposal message's From) is not nil.  seal65 == nil for a PREPARE set; senderPre / validPre
// may be nil.
func (c *Ctx) VerifyMessages(payload []byte, off []uint32, msgSig65, from20, hash32, hashLen, seal65, senderPre, validPre,
	raw []byte, round uint64, proposer20 []byte) (senderMask, validMask []uint64, t Tally, err error) {
	n := len(off) - 1
	senderMask, validMask = make([]uint64, (n+63)/64+1), make([]uint64, (n+63)/64+1)
	var ct C.ibft_tally_t
	rc := C.ibft_verify_messages(c.h, ptr8(payload), (*C.uint32_t)(unsafe.Pointer(&off[0])), ptr8(msgSig65), ptr8(from20),
		ptr8(hash32), ptr8(hashLen), ptr8(seal65), ptr8(senderPre), ptr8(validPre), C.size_t(n), ptr8(raw),
		C.size_t(len(raw)), C.uint64_t(round), nil, ptr8(proposer20), (*C.uint64_t)(unsafe.Pointer(&senderMask[0])),
		(*C.uint64_t)(unsafe.Pointer(&validMask[0])), &ct)
	return senderMask, validMask, tally(ct), c.check(rc)
}

// Routing byte per raw message (ibft_verify_messages_wire, out_class).
const (
	WireClassNeedsHost = byte(C.IBFT_WIRE_CLASS_NEEDS_HOST)
	WireClassClosure   = byte(C.IBFT_WIRE_CLASS_CLOSURE)
)

// VerifyMessagesWire = raw messages judged completely (ibft_verify_messages_wire): the device walks the bytes and
// verifies both signatures of every PREPARE / COMMIT of the view (height, round) in one launch.
// proposer20 != nil (a batch of PREPAREs of the view): the tally is HasPrepareQuorum.
func (c *Ctx) VerifyMessagesWire(wire []byte, off []uint32, height, round uint64, raw []byte, proposalRound uint64,
	proposer20 []byte) (senderMask, validMask []uint64, class []byte, t Tally, err error) {
	n := len(off) - 1
	senderMask, validMask = make([]uint64, (n+63)/64+1), make([]uint64, (n+63)/64+1)
	class = make([]byte, n+1)
	var ct C.ibft_tally_t
	rc := C.ibft_verify_messages_wire(c.h, ptr8(wire), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.size_t(n), C.uint64_t(height),
		C.uint64_t(round), ptr8(raw), C.size_t(len(raw)), C.uint64_t(proposalRound), nil,
		(*C.uint64_t)(unsafe.Pointer(&senderMask[0])), (*C.uint64_t)(unsafe.Pointer(&validMask[0])), ptr8(class), nil,
		ptr8(proposer20), &ct)
	return senderMask, validMask, class[:n], tally(ct), c.check(rc)
}

// CertNode mirrors ibft_cert_node_t: one IbftMessage of a certificate tree (VerifyCertificatesWire).
type CertNode struct {
	Off, Len              uint32 // the message's bytes in the call's buffer
	Parent, Ordinal       uint32 // containing row (CertNoParent for the call's own messages), position among its children
	FirstChild, NChildren uint32 // its nested messages are rows [FirstChild, FirstChild+NChildren)
	RawOff, RawLen        uint32 // Proposal.rawProposal it carries
	ProposalRound         uint64
	Cut0, Cut1            uint32 // its signature field, relative to Off: PayloadNoSig = bytes minus [Cut0, Cut1)
	Level, Role, Flags    uint8
	_                     [5]byte
}

const (
	CertNoParent         = uint32(C.IBFT_CERT_NO_PARENT)
	CertClassNeedsHost   = byte(C.IBFT_CERT_CLASS_NEEDS_HOST)       // not canonical here: the stock route decides this message
	CertClassDigestHost  = byte(C.IBFT_CERT_CLASS_DIGEST_BY_HOST)   // canonical, longer than the device hashes
	CertClassProposalHost = byte(C.IBFT_CERT_CLASS_PROPOSAL_BY_HOST) // the Proposal it carries is that long
	CertRolePCProposal   = uint8(C.IBFT_CERT_ROLE_PC_PROPOSAL)
	CertRolePCPrepare    = uint8(C.IBFT_CERT_ROLE_PC_PREPARE)
	CertRoleRCCMessage   = uint8(C.IBFT_CERT_ROLE_RCC_MESSAGE)
)

// VerifyCertificatesWire = every IsValidValidator / IsValidProposalHash that validateProposal, validPC and
// handleRoundChangeMessage (core/ibft.go:470-551, 683-788, 1162-1231) ask about the messages NESTED in the given raw
// PREPREPARE / ROUND_CHANGE messages, and about those messages themselves, in one call (ibft_verify_certificates_wire).
// Rows are breadth first: the call's messages, then their nested messages in wire order, and so on — the order in which
// a decoded message lists them.  sender / hash / self are bit masks over the rows, class one routing byte per row.
func (c *Ctx) VerifyCertificatesWire(wire []byte, off []uint32, rowsCap int) (nodes []CertNode, class []byte, sender, hash, self []uint64, err error) {
	n := len(off) - 1
	nodes = make([]CertNode, rowsCap+1)
	class = make([]byte, rowsCap+1)
	sender, hash, self = make([]uint64, (rowsCap+63)/64+1), make([]uint64, (rowsCap+63)/64+1), make([]uint64, (rowsCap+63)/64+1)
	var rows C.size_t
	rc := C.ibft_verify_certificates_wire(c.h, ptr8(wire), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.size_t(n), C.size_t(rowsCap), &rows,
		(*C.ibft_cert_node_t)(unsafe.Pointer(&nodes[0])), nil, ptr8(class), (*C.uint64_t)(unsafe.Pointer(&sender[0])),
		(*C.uint64_t)(unsafe.Pointer(&hash[0])), (*C.uint64_t)(unsafe.Pointer(&self[0])))
	return nodes[:int(rows)], class[:int(rows)], sender, hash, self, c.check(rc)
}

// PinnedBytes returns n bytes of page-locked memory (ibft_pinned_alloc) as a Go slice: column buffers the
// flatten step writes into once and reuses every round.  When every column of a call lies in such buffers the
// library reads them with one gather launch instead of one copy command per column.  C memory: invisible to
// the garbage collector, release with FreePinned.
func PinnedBytes(n int) []byte {
	p := C.ibft_pinned_alloc(C.size_t(n))
	if p == nil {
		return make([]byte, n) // ordinary memory works everywhere, only slower
	}
	return unsafe.Slice((*byte)(p), n)
}

// FreePinned releases a slice obtained from PinnedBytes (and only such a slice).
func FreePinned(b []byte) {
	if len(b) > 0 {
		C.ibft_pinned_free(unsafe.Pointer(&b[0]))
	}
}

// WireRow is what the device found in one IbftMessage (ibft_wire_row_t).
type WireRow struct {
	Height, Round                            uint64
	Status, Type, PayloadKind, HasView       uint8
	HashLen, SealLen, FromLen, SigLen        uint8
	From                                     [20]byte
	ProposalHash                             [32]byte
	_                                        [4]byte
}

// WireNeedsHost marks a row the device did not judge (PREPREPARE / ROUND_CHANGE payloads, unknown
// fields, non-canonical encodings): decode it with proto.Unmarshal and use VerifySenders.
const WireNeedsHost = 1

// VerifySendersWire = IsValidValidator (core/ibft.go:1128) for n messages given as the bytes the
// transport delivered; wire is their concatenation, off the n+1 offsets.  Replaces proto.Unmarshal +
// PayloadNoSig (messages/proto/helper.go:12-27) + flattening for PREPARE and COMMIT messages.
func (c *Ctx) VerifySendersWire(wire []byte, off []uint32) ([]uint64, []WireRow, Tally, error) {
	n := len(off) - 1
	mask := make([]uint64, (n+63)/64+1)
	rows := make([]WireRow, n+1)
	var t C.ibft_tally_t
	rc := C.ibft_verify_senders_wire(c.h, ptr8(wire), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.size_t(n),
		(*C.uint64_t)(unsafe.Pointer(&mask[0])), (*C.ibft_wire_row_t)(unsafe.Pointer(&rows[0])), &t)
	return mask, rows[:n], tally(t), c.check(rc)
}

// StageWireSeals makes the COMMIT seals found by the last VerifySendersWire the resident seal batch;
// follow with SealsRun (IsValidCommittedSeal without a second upload).
func (c *Ctx) StageWireSeals() error { return c.check(C.ibft_wire_stage_seals(c.h)) }

// sealRows asks the LIBRARY how many rows the resident batch and the oldest submitted pass have (ibft_seals_rows): verdict
// buffers are sized from these, never from a count the caller passes along — a count that is too small would let C write
// past the Go slice (ADVICE round 5).
func (c *Ctx) sealRows() (resident, oldest int, err error) {
	var a, b C.uint32_t
	if err = c.check(C.ibft_seals_rows(c.h, &a, &b)); err != nil {
		return 0, 0, err
	}
	return int(a), int(b), nil
}

// PipelineStats = how many submitted passes had their tally on the side stream (next to the following pass's verdict kernel,
// DESIGN.md §5.10) and how many batches of 65 537 … 98 304 rows went out as two launches, since the context was created.
func (c *Ctx) PipelineStats() (sideTallies, splitBatches int, err error) {
	var a, b C.uint32_t
	if err = c.check(C.ibft_pipeline_stats(c.h, &a, &b)); err != nil {
		return 0, 0, err
	}
	return int(a), int(b), nil
}

// SealsRows = rows of the resident seal batch as the library counts them.
func (c *Ctx) SealsRows() (int, error) {
	n, _, err := c.sealRows()
	return n, err
}

// SealsRun = one pass of IsValidCommittedSeal + HasQuorum over the RESIDENT seal batch
// (ibft_seals_run: launch + fetch); the mask has a word per 64 resident rows (SealsRows).
func (c *Ctx) SealsRun() ([]uint64, Tally, error) {
	n, _, err := c.sealRows()
	if err != nil {
		return nil, Tally{}, err
	}
	mask := make([]uint64, (n+63)/64+1)
	var t C.ibft_tally_t
	rc := C.ibft_seals_run(c.h, (*C.uint64_t)(unsafe.Pointer(&mask[0])), &t)
	return mask, tally(t), c.check(rc)
}

// SealsStageNext copies the NEXT seal batch into the context's spare column set on a copy stream of its own while the
// kernels of the resident batch run; SealsSwap makes it the resident batch.  Per step of a sustained stream of COMMIT
// sets (one GetValidMessages walk per wake-up, core/ibft.go:931-946): SealsLaunch(k) → SealsStageNext(k+1) →
// SealsFetch(k) → SealsSwap.  Every column must hold exactly n = len(sig65)/65 rows (ErrFallback otherwise: the copy would
// read past a slice); preFlags may be nil.  The copy is asynchronous — the library reads the slices after this call has
// returned — so they are pinned here until SealsSwap (cgo's pointer rule; the caller must not write to them before
// either).  Slices carved from PinnedBytes are page-locked as well, and only then does the copy overlap the kernels.
func (c *Ctx) SealsStageNext(hash32, sig65, signer20, preFlags []byte) error {
	n := len(sig65) / 65
	if n == 0 || len(sig65) != 65*n || len(hash32) != 32*n || len(signer20) != 20*n || (preFlags != nil && len(preFlags) != n) {
		return ErrFallback
	}
	c.next.Unpin() // (a batch staged and never swapped in is replaced: the library restarts the copy)
	c.next.Pin(&hash32[0])
	c.next.Pin(&sig65[0])
	c.next.Pin(&signer20[0])
	if preFlags != nil {
		c.next.Pin(&preFlags[0])
	}
	err := c.check(C.ibft_seals_stage_next(c.h, ptr8(hash32), ptr8(sig65), ptr8(signer20), ptr8(preFlags), C.size_t(n)))
	if err != nil {
		c.next.Unpin()
	}
	return err
}

// SealsSubmit enqueues one more pass over the resident seal batch and returns at once (at most two in flight);
// SealsCollect waits for the OLDEST submitted pass only and returns its verdict words and tally.  Keeping one pass in flight
// lets the device run back to back while Go handles the previous pass's result (ibft_seals_submit / ibft_seals_collect).
func (c *Ctx) SealsSubmit() error { return c.check(C.ibft_seals_submit(c.h)) }

// SealsCollect: see SealsSubmit.  The mask is sized for the rows of THAT pass as the library recorded them at submit time
// (a SealsSwap to a batch of another size may have come in between); rows = that count.
func (c *Ctx) SealsCollect() (mask []uint64, rows int, t Tally, err error) {
	_, rows, err = c.sealRows()
	if err != nil {
		return nil, 0, Tally{}, err
	}
	if rows == 0 {
		return nil, 0, Tally{}, ErrFallback // nothing submitted
	}
	mask = make([]uint64, (rows+63)/64+1)
	var ct C.ibft_tally_t
	rc := C.ibft_seals_collect(c.h, (*C.uint64_t)(unsafe.Pointer(&mask[0])), &ct)
	return mask, rows, tally(ct), c.check(rc)
}

// SealsSwap: see SealsStageNext.  Waits for the copy of the staged batch, after which its Go slices are released.
func (c *Ctx) SealsSwap() error {
	err := c.check(C.ibft_seals_swap(c.h, 1))
	c.next.Unpin()
	return err
}

// IssueProbe = the device canary (ibft_issue_probe): wall nanoseconds per aligned 8-byte VALU instruction per SIMD at one
// wavefront per SIMD — 1.89 on a healthy MI355X; a Backend logs it at start-up.
func (c *Ctx) IssueProbe() (nsPerInst float32, err error) {
	var ns, ms C.float
	err = c.check(C.ibft_issue_probe(c.h, &ns, &ms))
	return float32(ns), err
}

// SignSeals = n × Backend.BuildCommitMessage's committed seal (core/backend.go:12-34) for a SIMULATOR that plays
// n validators in one process: sk and hashes are n×32 bytes; returns the n×65 seals, the n×20 signer addresses
// and ok[i] == 0 for a key outside [1, n).  Leaves the batch resident: SealsRun() verifies it without an upload.
// Not for a production validator's key (include/ibftgpu.h, ibft_sign_seals).
func (c *Ctx) SignSeals(sk, hashes []byte) (seals, signers, ok []byte, err error) {
	n := len(sk) / 32
	if n == 0 || len(hashes) != 32*n {
		return nil, nil, nil, ErrFallback
	}
	seals, signers, ok = make([]byte, 65*n), make([]byte, 20*n), make([]byte, n)
	rc := C.ibft_sign_seals(c.h, (*C.uint8_t)(unsafe.Pointer(&sk[0])), (*C.uint8_t)(unsafe.Pointer(&hashes[0])),
		C.size_t(n), (*C.uint8_t)(unsafe.Pointer(&seals[0])), (*C.uint8_t)(unsafe.Pointer(&signers[0])),
		(*C.uint8_t)(unsafe.Pointer(&ok[0])))
	return seals, signers, ok, c.check(rc)
}

// Group is one process driving several MI355X (ibft_group_*): the rows of a batch are sharded over the
// devices in 64-aligned ranges and ONE RCCL all-reduce inside the library merges the verdict words and the
// ranks' distinct-sender bitmaps (a validator with valid rows in two shards is counted once, as HasQuorum's
// address set does) — for validator sets beyond a single GPU's batch (BASELINE configs #4 / #5).
type Group struct{ g *C.ibft_group }

func NewGroup(devices []int32, o Options) (*Group, error) {
	var flags C.uint32_t
	if o.StrictLowS {
		flags |= C.IBFT_FLAG_STRICT_LOW_S
	}
	if o.KeyCache {
		flags |= C.IBFT_FLAG_PUBKEY_CACHE
	}
	var g *C.ibft_group
	rc := C.ibft_group_create((*C.int32_t)(unsafe.Pointer(&devices[0])), C.uint32_t(len(devices)), flags,
		C.uint32_t(o.MaxRows), &g)
	if rc != C.IBFT_OK {
		return nil, fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	gr := &Group{g: g}
	runtime.SetFinalizer(gr, (*Group).Close)
	return gr, nil
}

func (g *Group) Close() {
	if g.g != nil {
		C.ibft_group_destroy(g.g)
		g.g = nil
	}
}

func (g *Group) SetValidators(height uint64, addrs20 []byte, power []uint64) error {
	var pp *C.uint64_t
	if len(power) > 0 {
		pp = (*C.uint64_t)(unsafe.Pointer(&power[0]))
	}
	if rc := C.ibft_group_set_validators(g.g, C.uint64_t(height), ptr8(addrs20), pp, C.size_t(len(power))); rc != C.IBFT_OK {
		return fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	return nil
}

// VerifySeals = IsValidCommittedSeal + HasQuorum over n rows sharded across the group's devices; mask and
// tally are the merged (global) ones.
func (g *Group) VerifySeals(hash32, sig65, signer20, preFlags []byte) ([]uint64, Tally, error) {
	n := len(sig65) / 65
	mask := make([]uint64, (n+63)/64+1)
	var t C.ibft_tally_t
	rc := C.ibft_group_verify_seals(g.g, ptr8(hash32), ptr8(sig65), ptr8(signer20), ptr8(preFlags), C.size_t(n),
		(*C.uint64_t)(unsafe.Pointer(&mask[0])), &t)
	if rc != C.IBFT_OK {
		return nil, Tally{}, fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	return mask, tally(t), nil
}

// VerifySenders = IsValidValidator + HasQuorum, sharded like VerifySeals (payload/off: concatenated
// PayloadNoSig bytes and their n+1 offsets).
func (g *Group) VerifySenders(payload []byte, off []uint32, sig65, from20, preFlags []byte) ([]uint64, Tally, error) {
	n := len(sig65) / 65
	mask := make([]uint64, (n+63)/64+1)
	var t C.ibft_tally_t
	rc := C.ibft_group_verify_senders(g.g, ptr8(payload), (*C.uint32_t)(unsafe.Pointer(&off[0])), ptr8(sig65),
		ptr8(from20), ptr8(preFlags), C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&mask[0])), &t)
	if rc != C.IBFT_OK {
		return nil, Tally{}, fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	return mask, tally(t), nil
}

// VerifyMessages judges a whole PREPARE / COMMIT set sharded by message (Ctx.VerifyMessages' columns and
// results; seal65 nil for a PREPARE set; digest32 nil to have raw ‖ BE64(round) hashed on the devices).
func (g *Group) VerifyMessages(payload []byte, off []uint32, msgSig65, from20, hash32, hashLen, seal65,
	senderPre, validPre, raw []byte, round uint64, digest32, proposer20 []byte) (senders, valid []uint64, t Tally, err error) {
	n := len(msgSig65) / 65
	senders = make([]uint64, (n+63)/64+1)
	valid = make([]uint64, (n+63)/64+1)
	var ct C.ibft_tally_t
	rc := C.ibft_group_verify_messages(g.g, ptr8(payload), (*C.uint32_t)(unsafe.Pointer(&off[0])), ptr8(msgSig65),
		ptr8(from20), ptr8(hash32), ptr8(hashLen), ptr8(seal65), ptr8(senderPre), ptr8(validPre), C.size_t(n),
		ptr8(raw), C.size_t(len(raw)), C.uint64_t(round), ptr8(digest32), ptr8(proposer20),
		(*C.uint64_t)(unsafe.Pointer(&senders[0])), (*C.uint64_t)(unsafe.Pointer(&valid[0])), &ct)
	if rc != C.IBFT_OK {
		return nil, nil, Tally{}, fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	return senders, valid, tally(ct), nil
}

// VerifyCertificatesWire judges the certificate trees of raw PREPREPARE / ROUND_CHANGE messages sharded by carrier over the
// group's devices (Ctx.VerifyCertificatesWire's arguments and results; the rows are numbered as one call over all messages
// numbers them).  No tally is taken and the devices exchange nothing.
func (g *Group) VerifyCertificatesWire(wire []byte, off []uint32, rowsCap int) (nodes []CertNode, class []byte, sender, hash, self []uint64, err error) {
	n := len(off) - 1
	nodes = make([]CertNode, rowsCap+1)
	class = make([]byte, rowsCap+1)
	sender, hash, self = make([]uint64, (rowsCap+63)/64+1), make([]uint64, (rowsCap+63)/64+1), make([]uint64, (rowsCap+63)/64+1)
	var rows C.size_t
	rc := C.ibft_group_verify_certificates_wire(g.g, ptr8(wire), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.size_t(n), C.size_t(rowsCap), &rows,
		(*C.ibft_cert_node_t)(unsafe.Pointer(&nodes[0])), nil, ptr8(class), (*C.uint64_t)(unsafe.Pointer(&sender[0])),
		(*C.uint64_t)(unsafe.Pointer(&hash[0])), (*C.uint64_t)(unsafe.Pointer(&self[0])))
	if rc != C.IBFT_OK {
		return nil, nil, nil, nil, nil, fmt.Errorf("%w: %s", ErrFallback, C.GoString(C.ibft_strerror(rc)))
	}
	return nodes[:int(rows)], class[:int(rows)], sender, hash, self, nil
}

func tally(t C.ibft_tally_t) Tally {
	return Tally{uint64(t.quorum_lo), uint64(t.quorum_hi), uint64(t.power_lo), uint64(t.power_hi),
		uint32(t.valid_rows), uint32(t.distinct_senders), t.has_quorum != 0, uint32(t.shard_overlap), uint32(t.proposer_rows)}
}

// Bit reports row i's verdict.
func Bit(mask []uint64, i int) bool { return mask[i>>6]>>(uint(i)&63)&1 == 1 }
