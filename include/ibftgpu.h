/*
 * include/ibftgpu.h — C ABI of libibftgpu.so, the MI355X (gfx950) batch verifier
 * that sits behind go-ibft's Verifier hooks.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * framework types.  The Go side keeps core/ibft.go's RunSequence untouched and
 * calls these entry points through the cgo shim in shim/go/ (see INTEGRATION.md).
 *
 * Each entry point replaces a per-message interface call of the reference with one
 * call per batch:
 *
 *   ibft_verify_hashes   <- Verifier.IsValidProposalHash(proposal, hash)
 *                           /root/reference/core/backend.go:50-51; call sites
 *                           /root/reference/core/ibft.go:545, 649, 781, 858, 938
 *   ibft_verify_seals    <- Verifier.IsValidCommittedSeal(proposalHash, seal)
 *                           /root/reference/core/backend.go:53-55; call site
 *                           /root/reference/core/ibft.go:943
 *   ibft_verify_senders  <- Verifier.IsValidValidator(msg)
 *                           /root/reference/core/backend.go:41-45; call sites
 *                           /root/reference/core/ibft.go:735, 1128, 1213, 1220
 *   ibft_set_validators  <- ValidatorBackend.GetVotingPowers(height)
 *                           /root/reference/core/validator_manager.go:17-20, 50-75
 *   ibft_tally (+ the ibft_tally_t filled by the verify calls)
 *                        <- ValidatorManager.HasQuorum
 *                           /root/reference/core/validator_manager.go:77-96
 *   ibft_tally_prepare (+ the proposer20 argument of the message-set calls)
 *                        <- ValidatorManager.HasPrepareQuorum — the decision hasQuorumByMsgType takes for PREPARE
 *                           /root/reference/core/validator_manager.go:99-127, core/ibft.go:1273-1284
 *   ibft_verify_messages, ibft_verify_messages_wire
 *                        <- all of the above for a whole PREPARE / COMMIT set in ONE call (from SoA columns / from the
 *                           transport's bytes): IsValidValidator on arrival (core/ibft.go:1128) and the closure of
 *                           handlePrepare / handleCommit (:856-862, :932-944) are pure, so a message is judged once,
 *                           completely — both signatures of a COMMIT in one verdict launch
 *   ibft_verify_senders_wire, ibft_wire_stage_seals
 *                        <- proto.Unmarshal + PayloadNoSig (messages/proto/helper.go:12-27) + IsValidValidator
 *   ibft_verify_certificates_wire
 *                        <- every IsValidValidator / IsValidProposalHash that validateProposal, validPC and
 *                           handleRoundChangeMessage ask about the messages NESTED in PREPREPARE / ROUND_CHANGE
 *                           messages (core/ibft.go:470-551, 683-788), from the transport's bytes
 *   ibft_comm_*, ibft_group_*  validator shards over several MI355X, one RCCL all-reduce inside the library
 *   ibft_sign_seals      <- n × Backend.BuildCommitMessage's seal (core/backend.go:12-34), simulators only
 *   ibft_pinned_alloc    page-locked column buffers the device reads itself (one gather launch per call)
 *
 * Conventions (the reference fixes none of the arithmetic; these are the
 * Ethereum-style ones an IBFT backend uses, stated once, obeyed by CPU oracle and
 * GPU alike):
 *   proposal hash = keccak256(RawProposal ‖ BE64(Round));  seal digest = the 32-byte
 *   proposalHash itself (or keccak256(proposalHash ‖ suffix): ibft_set_seal_digest);  sender digest = keccak256(PayloadNoSig)
 *   (/root/reference/messages/proto/helper.go:12-27);  signature = 65 B r‖s‖v with
 *   r,s big-endian in [1,n-1] and v∈{0,1};  address = keccak256(X‖Y)[12:32].
 *   Recovery-id policy: v is the parity of R.y and nothing else — R.x = r always.  The second candidate
 *   of SEC 1 §4.1.6 (R.x = r + n, possible only for r < p − n ≈ 2^128.4, i.e. with probability ≈ 2^−127.6,
 *   recovery ids 2 and 3) is NEVER tried: v ∈ {2,3} (and every other value) is rejected by every kernel
 *   variant, cold and warm, exactly as go-ethereum's crypto.Ecrecover / the precompile reject it for a
 *   65-byte [R‖S‖V] input with V > 1.  High-s signatures are accepted unless IBFT_FLAG_STRICT_LOW_S.
 *
 * Verdict masks: bit i of out_mask[i/64] is row i's verdict (1 = the reference
 * predicate would return true).  Rows the host already knows to be structurally
 * invalid (nil payload, wrong length, a1 failed → a2 short-circuited,
 * /root/reference/core/ibft.go:938-943) are passed with a non-zero pre_flags byte
 * and come back 0 without touching the device arithmetic.
 *
 * Error behaviour: every call returns IBFT_OK (0) or a negative IBFT_E_* code and
 * never synthesises verdicts on failure — the caller must then run its own CPU
 * Verifier for that batch (all-false would stall liveness, all-true would break
 * safety).  There is NO CPU fallback inside this library.
 *
 * Threading: a context is internally serialised (one mutex, one HIP stream + a side stream for the proposal hash); use
 * one context per concurrent caller (the four goroutines of
 * /root/reference/core/ibft.go:335-347 would each hold one).  Nothing is retained
 * from caller memory after a call returns (cgo pointer rules).
 */
#ifndef IBFTGPU_H
#define IBFTGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBFT_OK 0
#define IBFT_E_INVAL (-1)     /* bad argument                                   */
#define IBFT_E_NODEVICE (-2)  /* no usable gfx950 device / HIP runtime error     */
#define IBFT_E_NOMEM (-3)     /* host or device allocation failed                */
#define IBFT_E_HIP (-4)       /* a HIP call failed; ibft_last_error() has detail */
#define IBFT_E_NOVALSET (-5)  /* ibft_set_validators has not succeeded yet       */
#define IBFT_E_POWER (-6)     /* total voting power is zero (errVotingPowerNotCorrect) */
#define IBFT_E_TOOBIG (-7)    /* batch larger than cfg.max_rows                  */
#define IBFT_E_RCCL (-8)      /* librccl could not be loaded, or a collective / communicator call failed */

/* cfg.flags */
#define IBFT_FLAG_STRICT_LOW_S 1u /* also reject s > n/2 (off: go-ethereum Ecrecover semantics) */
/* Warm path: remember each validator's public key the first time it is recovered (and hashes to
 * its address), build per-validator fixed-base tables in HBM (655 KB per validator, budget
 * IBFT_QTAB_BUDGET_GB, default 64) and VERIFY later signatures of that validator against them
 * instead of recovering — identical verdicts (csrc/verify_dev.h), ~4-10x less work.  The tables are
 * per DEVICE and per ADDRESS (ibft_cache_memory below): shared by every context of the device, kept
 * for every validator that stays in the set when ibft_set_validators changes it.                */
#define IBFT_FLAG_PUBKEY_CACHE 2u

/* pre_flags bits */
#define IBFT_ROW_NIL 0x01u
#define IBFT_ROW_BADLEN 0x02u
#define IBFT_ROW_HASH_BAD 0x04u

/* cfg.kernel: how many lanes work on one signature.  The verdicts never depend on it.
 *   AUTO  cold path (recover): TWO wavefronts per signature up to 512 rows (a helper wavefront takes r⁻¹, the scalar split
 *         and u1·G off the critical path), one wavefront per signature up to 2048 rows, one DPP row (16 lanes) per
 *         signature up to 8192 rows, then 4 / 2 lanes per signature while rows*lanes <= 65536, one
 *         lane beyond (the 8-lane form remains selectable);
 *         warm path (known keys): G = 64,32,...,2 lanes per signature so that a batch gives about
 *         one wavefront per SIMD (64 up to 1024 rows), one lane from 65536 rows.
 *   LANE  always one lane per signature (throughput form, both paths).
 *   WAVE  warm path pinned to one wavefront per signature.
 * Experiments only: the environment variable IBFT_COLD_LANES = 1|2|4|8|16|64|128 pins the cold variant (128 = two
 * wavefronts per signature), IBFT_PAIR_ROWS_MAX / IBFT_WAVE_ROWS_MAX / IBFT_ROWS_KERNEL_MAX move the AUTO thresholds of
 * the two-wavefront, the one-wavefront and the row-per-signature forms (read at ibft_ctx_create). */
#define IBFT_KERNEL_AUTO 0u
#define IBFT_KERNEL_LANE 1u
#define IBFT_KERNEL_WAVE 2u

typedef struct ibft_ctx ibft_ctx;

typedef struct {
  int32_t device;    /* HIP device ordinal                                         */
  uint32_t flags;    /* IBFT_FLAG_*                                                 */
  uint32_t max_rows; /* largest batch this context will be asked for (0 = 65536)    */
  uint32_t kernel;   /* IBFT_KERNEL_*                                               */
} ibft_cfg;

typedef struct {
  uint64_t quorum_lo, quorum_hi; /* floor(2*total/3)+1 (validator_manager.go:130-135) */
  uint64_t power_lo, power_hi;   /* Σ power over distinct valid senders ∈ validator set */
  uint32_t valid_rows;           /* popcount of the verdict mask                        */
  uint32_t distinct_senders;     /* distinct member senders among the valid rows        */
  uint32_t has_quorum;           /* power >= quorum                                     */
  uint32_t shard_overlap;        /* sharded calls (ibft_seals_fetch_merged, ibft_group_*): Σ over validators of (shards in which
                                    the validator has a valid row) − 1.  Such a validator is counted ONCE in power and
                                    distinct_senders, like everywhere else; 0 from the single-device calls             */
  uint32_t proposer_rows;        /* HasPrepareQuorum calls (a proposer20 was given): valid rows whose sender IS the proposer —
                                    any such row forces has_quorum = 0 (validator_manager.go:114-121); 0 otherwise     */
  uint32_t reserved;
} ibft_tally_t;

int ibft_version(void);
const char *ibft_strerror(int code);
/* last HIP error string seen by this context (thread-unsafe diagnostic) */
const char *ibft_last_error(const ibft_ctx *ctx);

int ibft_ctx_create(const ibft_cfg *cfg, ibft_ctx **out);
void ibft_ctx_destroy(ibft_ctx *ctx);

/* Replace the validator table (addresses n×20, powers n×u64) for `height`.
 * Builds the device-side open-addressing table used for set membership and the
 * weighted tally.  Powers are u64 here (128-bit sums); a set whose powers do not fit
 * uses ibft_set_validators_u256 (validator_manager.go:19 uses *big.Int).           */
int ibft_set_validators(ibft_ctx *ctx, uint64_t height, const uint8_t *addrs20,
                        const uint64_t *power, size_t n);

/* The same with arbitrary-precision powers: ValidatorBackend.GetVotingPowers returns map[string]*big.Int
 * (/root/reference/core/validator_manager.go:17-31) and a stake-weighted set (wei-denominated) exceeds 2^64
 * with ~19 tokens.  power_be32 is n × 32 bytes, each a big-endian 256-bit integer (big.Int.FillBytes of a
 * 32-byte slice); a power that does not fit 256 bits cannot be passed (keep such a set on the Go path).
 * Sums and the quorum ⌊2·total/3⌋+1 are kept in 320 bits; ibft_tally_t then carries the LOW 128 bits of power
 * and quorum and the exact has_quorum flag — ibft_last_tally_wide returns the full-width numbers.          */
int ibft_set_validators_u256(ibft_ctx *ctx, uint64_t height, const uint8_t *addrs20,
                             const uint8_t *power_be32, size_t n);
typedef struct {
  uint64_t quorum[5];        /* little-endian 64-bit words                                     */
  uint64_t power[5];         /* Σ power over distinct valid member senders of the last tally    */
  uint32_t has_quorum, reserved;
} ibft_tally_wide_t;
/* full-width result of the last tally this context delivered (verify / fetch / fetch_merged call)  */
int ibft_last_tally_wide(ibft_ctx *ctx, ibft_tally_wide_t *out);

/* ---- the seal-digest convention ---------------------------------------------------------------------------
 * "IsValidCommittedSeal checks if signature for proposal hash in committed seal is signed by a validator"
 * (/root/reference/core/backend.go:53-55): WHICH bytes the seal signs is the embedding Backend's choice.  Default
 * (IDENTITY): the 32-byte proposalHash itself.  KECCAK_SUFFIX: keccak256(proposalHash ‖ suffix), suffix ≤ 64 bytes —
 * e.g. a Backend that appends the COMMIT message type before hashing passes suffix = {0x02}.  The convention applies to
 * every a2 evaluation of the context: ibft_verify_seals / ibft_seals_stage (the hash32 column still carries the
 * proposalHash each row's message carries; the digest is derived on the device), ibft_wire_stage_seals, the seal half of
 * ibft_verify_messages / ibft_verify_messages_wire (a1 keeps comparing the CARRIED hash with the proposal's), the group
 * calls (ibft_group_set_seal_digest), and ibft_sign_seals (a simulator's seals sign what its verifiers check).  It does
 * not touch a3: an envelope signs keccak256(PayloadNoSig).  Changing it drops a resident staged batch.            */
#define IBFT_SEAL_DIGEST_IDENTITY 0u
#define IBFT_SEAL_DIGEST_KECCAK_SUFFIX 1u
#define IBFT_SEAL_SUFFIX_MAX 64u
int ibft_set_seal_digest(ibft_ctx *ctx, uint32_t mode, const uint8_t *suffix, size_t suffix_len);

/* a1.  raw/raw_len/round: the proposal all rows are checked against.  hash32 is
 * n×32 (zero-filled where absent), hash_len[i] the real byte length (0 = nil).    */
int ibft_verify_hashes(ibft_ctx *ctx, const uint8_t *raw, size_t raw_len, uint64_t round,
                       const uint8_t *hash32, const uint8_t *hash_len, size_t n,
                       uint64_t *out_mask);
/* The same with the proposal's Keccak already in hand (an application Backend computes it when it builds or
 * validates the proposal): a pure 32-byte compare per row.  Recommended for proposals beyond ~64 KiB — Keccak is
 * a sequential sponge, one lane walks it at ≈13 µs per 136-byte block (profiles/r02_a1_sizes.json).         */
int ibft_verify_hashes_digest(ibft_ctx *ctx, const uint8_t digest32[32], const uint8_t *hash32,
                              const uint8_t *hash_len, size_t n, uint64_t *out_mask);
/* keccak256(raw ‖ BE64(round)) (what BuildPrepareMessage's caller would sign).  The context remembers the last
 * proposal it hashed: ibft_verify_hashes / ibft_proposal_hash with the same (raw, round) — every PREPARE and COMMIT set
 * of a round, every wake-up — skip the hash.  WHERE it is hashed: on the calling host thread, with the library's own
 * permutation (csrc/keccak_dev.h compiles for both sides).  Keccak is a sequential sponge — one message cannot be spread
 * over the chip — and one host core absorbs ≈340 MB/s (3.6 µs for 1 KiB, ≈3 ms for 1 MiB) where the one wavefront that
 * can work on it absorbs 25 MB/s (41 ms for 1 MiB, round 2); the 32-byte digest then reaches the device on the side
 * stream, behind the verdict launch it overlaps with.  IBFT_PROPOSAL_HASH=device selects the wavefront kernel (A/B,
 * tests).  The device keeps every hash of which there are MANY to do in parallel: PayloadNoSig of every message, the
 * address of every recovered key, the digests of a certificate tree.                                               */
int ibft_proposal_hash(ibft_ctx *ctx, const uint8_t *raw, size_t raw_len, uint64_t round,
                       uint8_t out32[32]);

/* keccak256(a ‖ b) on the host (either part may be empty) — the same routine: for callers that hash what the device
 * hands back (IBFT_CERT_CLASS_DIGEST_BY_HOST rows: a = bytes[0, cut0), b = bytes[cut1, len)).  No context, no device.  */
int ibft_keccak256(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, uint8_t out32[32]);

/* a2.  hash32 n×32 (per-row proposalHash from ExtractCommitHash), sig65 n×65,
 * signer20 n×20 (msg.From).  tally may be NULL.                                   */
int ibft_verify_seals(ibft_ctx *ctx, const uint8_t *hash32, const uint8_t *sig65,
                      const uint8_t *signer20, const uint8_t *pre_flags, size_t n,
                      uint64_t *out_mask, ibft_tally_t *tally);

/* a3.  payload = concatenated PayloadNoSig bytes; row i is payload[off[i]..off[i+1]);
 * off has n+1 entries.                                                             */
int ibft_verify_senders(ibft_ctx *ctx, const uint8_t *payload, const uint32_t *off,
                        const uint8_t *sig65, const uint8_t *from20, const uint8_t *pre_flags,
                        size_t n, uint64_t *out_mask, ibft_tally_t *tally);

/* ---- the wire bytes themselves ---------------------------------------
 * Rows are IbftMessage protobuf bytes as the transport delivers them
 * (/root/reference/messages/proto/messages.proto:24-44); row i is wire[off[i]..off[i+1]).
 * Replaces, for PREPARE and COMMIT messages, the host's proto.Unmarshal + PayloadNoSig re-marshal
 * (/root/reference/messages/proto/helper.go:12-27) + column flattening in front of a3: the device
 * walks the fields, vouches that the bytes are the canonical encoding (so that "wire minus the
 * signature field" IS PayloadNoSig), hashes them, extracts From / Signature / view / proposal hash
 * / committed seal and verifies the sender.  out_rows (n entries, may be NULL) tells the host what
 * was found.  A row with status IBFT_WIRE_NEEDS_HOST (PREPREPARE / ROUND_CHANGE payloads, unknown
 * fields, any non-canonical encoding, truncated input) is NOT judged: its verdict bit is 0 and the
 * caller must decode it with the protobuf runtime and use ibft_verify_senders.                  */
#define IBFT_WIRE_OK 0u
#define IBFT_WIRE_NEEDS_HOST 1u
typedef struct {
  uint64_t height, round;    /* View (0 when absent / omitted)                              */
  uint8_t status;            /* IBFT_WIRE_*                                                  */
  uint8_t type;              /* IbftMessage.type                                             */
  uint8_t payload_kind;      /* oneof member: 0 none, 6 PrepareMessage, 7 CommitMessage (5 PrePrepareMessage,
                                8 RoundChangeMessage: ibft_verify_certificates_wire only)    */
  uint8_t has_view;
  uint8_t hash_len;          /* bytes of proposal_hash present (<= 32)                       */
  uint8_t seal_len;          /* bytes of committed_seal present (255 = more)                 */
  uint8_t from_len, sig_len; /* 255 = more                                                   */
  uint8_t from[20];
  uint8_t proposal_hash[32];
  uint8_t pad[4];
} ibft_wire_row_t;
int ibft_verify_senders_wire(ibft_ctx *ctx, const uint8_t *wire, const uint32_t *off, size_t n,
                             uint64_t *out_mask, ibft_wire_row_t *out_rows, ibft_tally_t *tally);
/* Make the COMMIT seals found by the last ibft_verify_senders_wire the resident seal batch
 * (hash column <- proposal hash, signature column <- committed seal, signer <- From): follow with
 * ibft_seals_launch + ibft_seals_fetch — a2 without another upload.  Rows that are not canonical
 * COMMIT messages (type COMMIT, 32-byte hash, 65-byte seal, 20-byte From) are pre-flagged.      */
int ibft_wire_stage_seals(ibft_ctx *ctx);

/* a8 alone: HasQuorum over the rows whose bit is set in mask.                      */
int ibft_tally(ibft_ctx *ctx, const uint8_t *sender20, const uint64_t *mask, size_t n,
               ibft_tally_t *tally);
/* a8, the PREPARE form: ValidatorManager.HasPrepareQuorum(state, proposalMessage, msgs)
 * (/root/reference/core/validator_manager.go:99-127 — what hasQuorumByMsgType asks for PREPARE messages,
 * core/ibft.go:1273-1284).  proposer20 = proposalMessage.From.  The proposer's address JOINS the sender set before the
 * powers are summed (it counts once, and only if it is a validator: unknown addresses add nothing, :88-92), and a
 * row of the mask whose sender equals the proposer byte for byte — validator or not — voids the result:
 * has_quorum = 0 whatever the power, tally->proposer_rows = the number of such rows (:114-121 "proposer is among
 * signers but it is not expected to be").  distinct_senders then counts the proposer's seat too.  With n = 0 the
 * answer is whether the proposer alone is a quorum (one validator of all the power).  The caller keeps the one case
 * the device cannot see: proposalMessage == nil → false without asking (:101-110).  The same rule is applied by
 * ibft_verify_messages / ibft_verify_messages_wire / ibft_group_verify_messages when their proposer20 argument is
 * not NULL, and by the sharded merge (the seat joins the MERGED bitmap; a proposer row in any shard voids).     */
int ibft_tally_prepare(ibft_ctx *ctx, const uint8_t *sender20, const uint64_t *mask, size_t n,
                       const uint8_t proposer20[20], ibft_tally_t *tally);

/* The context remembers the last (raw proposal, round) it hashed: ibft_verify_hashes / ibft_verify_messages
 * with the same proposal do not hash it again.  ibft_forget_proposal drops that memory (the next call hashes);
 * bench.py calls it once per measured sequence so that every height pays for its own proposal hash.         */
int ibft_forget_proposal(ibft_ctx *ctx);

/* ---- staged (device-resident) form of a2, used by bench.py and by multi-GPU ----
 * stage:  copy the batch into the context's HBM columns (H2D, synchronous).
 * launch: enqueue unpack → recover → tally on the context's stream (asynchronous);
 *         `repeat` back-to-back passes over the resident batch (each pass is the
 *         full computation; used to time K steps without host round trips).
 * fetch:  wait for the stream, copy mask + tally back.                             */
int ibft_seals_stage(ibft_ctx *ctx, const uint8_t *hash32, const uint8_t *sig65,
                     const uint8_t *signer20, const uint8_t *pre_flags, size_t n);
int ibft_seals_launch(ibft_ctx *ctx, uint32_t repeat);
int ibft_seals_fetch(ibft_ctx *ctx, uint64_t *out_mask, ibft_tally_t *tally);
/* launch(1) + fetch in one call: one more pass over the resident batch, results on return.      */
int ibft_seals_run(ibft_ctx *ctx, uint64_t *out_mask, ibft_tally_t *tally);
/* Two staging slots (round 5): the reference hands handleCommit NEW messages at every wake-up
 * (core/ibft.go:931-946: one GetValidMessages walk per wake-up), so a sustained stream of batches pays an upload per
 * batch; with one column set the upload of batch k+1 could not start before the verdict of batch k.
 *   ibft_seals_stage_next: copy the NEXT batch into the context's spare column set on a copy stream of its own,
 *       asynchronously — the source buffers must stay untouched until ibft_seals_swap(ctx, 1) returns (or, with
 *       wait_for_copy = 0, until the next fetch of that batch); for the copy to overlap they must be page-locked
 *       (ibft_pinned_alloc);
 *   ibft_seals_swap: the staged batch becomes the resident one (launch / run / fetch then work on it), the previous one
 *       becomes the spare slot; kernels already enqueued keep reading the columns they were launched on, the ones
 *       enqueued afterwards wait ON THE DEVICE for the copy.  IBFT_E_INVAL without a staged batch.
 * Per step of a sustained stream:  launch(k) → stage_next(k+1) → fetch(k) → swap.                                   */
/* Pipelined passes (round 5): ibft_seals_submit enqueues ONE pass over the resident batch (verdict kernel + tally) whose
 * results go to one of two host-visible result slots, and returns at once; ibft_seals_collect waits for the OLDEST
 * submitted pass — the event behind its tally, not the whole stream — and delivers its verdict words and tally.  At most
 * two passes in flight (IBFT_E_INVAL beyond; IBFT_E_INVAL on a collect with nothing submitted).  With one pass kept in
 * flight the launch of pass k+1 overlaps the completion, result delivery and host handling of pass k: the device runs
 * back to back, as a node's does when batches arrive faster than one host round trip (and as a rank's does in the
 * sharded form, where exchange k overlaps the kernels of pass k+1).  Combines with the staging slots:
 * submit(k) → stage_next(k+1) → collect(k−1) → swap.
 * out_mask of a collect holds ⌈rows/64⌉ words for the rows of THAT pass (the batch resident when it was submitted —
 * ibft_seals_rows says how many; a swap in between may have changed the resident batch's size).  With the key cache on, a
 * collect whose pass taught the device new keys builds their tables before it returns, which drains the context's whole
 * stream — the newer pass in flight included: the cold-to-warm transition costs one full wait.  A failure of that table
 * build takes nothing from the pass just delivered (IBFT_OK, verdicts valid); the next ibft_seals_submit returns it.   */
int ibft_seals_submit(ibft_ctx *ctx);
int ibft_seals_collect(ibft_ctx *ctx, uint64_t *out_mask, ibft_tally_t *tally);
/* Rows of the resident seal batch, and of the oldest pass submitted and not yet collected (0: none in flight): what a
 * binding sizes the out_mask of ibft_seals_run / _fetch / _collect from.                                              */
int ibft_seals_rows(ibft_ctx *ctx, uint32_t *resident_rows, uint32_t *oldest_pass_rows);
/* Round 6: the TALLY of a submitted pass (HasQuorum over the rows that passed, core/validator_manager.go:77-96) runs on a
 * stream of its own, next to the verdict kernel of the pass submitted after it, on a second copy of the work buffers the
 * two would share — the device goes from verdict kernel to verdict kernel without the tally and its two dependency gaps in
 * between (9–15 µs per pass).  Same results, slot for slot; every other entry point first puts the context's main stream
 * behind the tallies still in flight.  Not for a rank of a sharded batch (its exchange follows the tally) and not while
 * the key cache is still learning.  IBFT_SIDE_TALLY=0 keeps everything on one stream (A/B).
 * ibft_pipeline_stats: passes whose tally took the side stream, and batches of 65 537 … 98 304 rows that went out as two
 * launches (both since the context was created; either pointer may be NULL).                                             */
int ibft_pipeline_stats(ibft_ctx *ctx, uint32_t *side_tallies, uint32_t *split_batches);
int ibft_seals_stage_next(ibft_ctx *ctx, const uint8_t *hash32, const uint8_t *sig65,
                          const uint8_t *signer20, const uint8_t *pre_flags, size_t n);
int ibft_seals_swap(ibft_ctx *ctx, int wait_for_copy);
/* Device address of the resident verdict mask (⌈n/64⌉ u64 words) and of the two
 * u64 tally accumulators {power, valid_rows|distinct<<32}: lets the caller run an
 * RCCL all-reduce over validator shards without a host round trip.                 */
int ibft_seals_device_ptrs(ibft_ctx *ctx, void **d_mask, size_t *mask_words, void **d_tally);
/* Copy the resident verdict mask (⌈n/64⌉ u64) and the 4 tally words {power_lo,
 * power_hi, valid_rows|distinct<<32, has_quorum} into caller-owned DEVICE buffers
 * (e.g. a torch tensor's data_ptr) and wait for the copy: the hand-off point to an
 * RCCL all-reduce issued by the caller.  Either pointer may be NULL.               */
int ibft_seals_export(ibft_ctx *ctx, void *d_mask_dst, void *d_tally_dst);
/* Same hand-off without a host round trip: the copies are enqueued on the CALLER's stream (a
 * hipStream_t, e.g. torch.cuda.current_stream().cuda_stream — the one the collective will run on)
 * behind an event that marks the last launch's results ready, and the context's next tally waits
 * (on the device) until they have been read.  The caller's collective then overlaps with the next
 * ibft_seals_launch on the context's own stream.                                                  */
int ibft_seals_export_on(ibft_ctx *ctx, void *d_mask_dst, void *d_tally_dst, void *consumer_stream);
/* HIP-event time (ms) of the verdict kernels, measured on the context's own stream, summed over the
 * launches since the previous call of this function (which resets the sum), and their count.      */
int ibft_last_kernel_ms(ibft_ctx *ctx, float *ms, uint32_t *launches);
/* Time the verdict kernels of every n-th staged pass only (default 1 = every pass, 0 = never): an event pair
 * costs ≈5 µs of a 0.45 ms step (profiles/r02_launch_gap.txt).                                            */
int ibft_set_kernel_timing(ibft_ctx *ctx, uint32_t every_n);
/* Warm-path statistics: validators whose table is built, how many verdict passes ran with /
 * without the warm kernel since the context was created, and the lanes-per-signature (64 = one
 * wavefront per signature … 1 = lane kernel) the last warm pass used.                         */
int ibft_cache_stats(ibft_ctx *ctx, uint32_t *tables, uint32_t *warm_passes, uint32_t *cold_passes,
                     uint32_t *lanes_per_signature);
/* The key tables belong to the DEVICE: every context created on one device shares ONE fixed-base table of G (84 MB) and ONE
 * pool of validator tables keyed by ADDRESS (655 KB per validator, IBFT_QTAB_BUDGET_GB, default 64) — the four contexts a
 * Backend keeps for its four goroutines cost one pool, not four; a key learned through one context is
 * known to all of them; a validator that stays in the set across ibft_set_validators keeps its table (a rotation of 1 % of
 * the validators relearns 1 %), a slot nobody's current set refers to is reused.  This reports the shared object of ctx's
 * device: bytes it holds, slots in use / allocated, and how many contexts share it.                                   */
int ibft_cache_memory(ibft_ctx *ctx, uint64_t *device_bytes, uint32_t *slots_in_use, uint32_t *slots_allocated,
                      uint32_t *contexts_sharing);
/* Lanes per signature used by the last verdict pass: cold kernel (1 = ecrecover_lane_kernel,
 * 2/4/8 = ecrecover_group_kernel, 16 = ecrecover_rows_kernel, 64 = ecrecover_wave_kernel, 128 = ecrecover_wave2_kernel)
 * and warm kernel (0 = none ran, 1 = lane, 2..64 = group).                                        */
int ibft_last_dispatch(ibft_ctx *ctx, uint32_t *cold_lanes, uint32_t *warm_lanes);
/* (A cold batch of 65 537 … 98 304 rows is two launches — 65 536 rows through the lane kernel with its tables in LDS, the rest
 * through the kernel the rule above picks for that many rows — and reports cold_lanes = 1, table 1; IBFT_SPLIT_LARGE=0 in the
 * environment pins one launch of the private-segment lane kernel.)                                                        */
/* Where the last lane / group cold kernel kept the per-lane window table of u2·R: 0 = no such kernel ran, 1 = the
 * workgroup's LDS (default up to one wavefront per SIMD: no private segment), 2 = private segment with the entries read in
 * front of the doublings (round 4's form; IBFT_COLD_TABLE=private), 3 = private segment without that prefetch — two
 * resident wavefronts per SIMD, the lane kernel's form beyond 65 536 rows.                                              */
int ibft_last_cold_table(ibft_ctx *ctx, uint32_t *table);
/* ---- pinned column buffers -------------------------------------------------------------------------
 * Every entry point accepts ordinary (pageable) host memory for its columns; the runtime then stages each
 * column through its own bounce buffer — measured at ≈8.5 GB/s, 0.15 ms for the 1.3 MB of a 4 096-message
 * COMMIT set (profiles/r02g_seq_*).  A caller that flattens messages into columns anyway (the cgo shim's
 * SoA batcher) should write them into buffers from ibft_pinned_alloc instead: page-locked
 * memory the device reads directly — when every column of a call lies in such buffers the library replaces
 * the per-column copy commands by ONE gather launch (≈8 µs of host time per column saved, the copies no
 * longer queue one behind the other).  Plain memory otherwise: valid until ibft_pinned_free, usable with
 * any context, any thread.  Returns NULL when the runtime refuses (no device, out of lockable memory).     */
void *ibft_pinned_alloc(size_t bytes);
void ibft_pinned_free(void *p);
/* Batches of this context whose columns ALL lay in ibft_pinned_alloc blocks and were therefore read by the
 * device in one gather launch instead of one copy command per column (environment IBFT_NO_GATHER=1 turns
 * the gather off: pinned columns then take the copy commands, still faster to issue than pageable ones).  */
int ibft_column_stats(ibft_ctx *ctx, uint32_t *gather_batches);

/* ---- a whole PREPARE / COMMIT set in one call --------------------------------------------------------
 * The reference judges a stored PREPARE / COMMIT message three times, at different moments: IsValidValidator
 * when it arrives (core/ibft.go:1128, core/backend.go:41-45), IsValidProposalHash and — COMMIT only —
 * IsValidCommittedSeal when the view is handled (core/ibft.go:856-862 handlePrepare's closure, :932-944
 * handleCommit's).  All three are pure functions of the message bytes, the proposal and the validator set, so
 * a Backend that receives messages in batches can have the whole set judged at once:
 *   out_sender_mask bit i = IsValidValidator(message i)            — what AddMessage would have decided;
 *   out_valid_mask  bit i = hash32[i] ≡ keccak(raw ‖ BE64(round))  (hash_len[i] == 32)
 *                           ∧ (seal65 == NULL ∨ IsValidCommittedSeal(hash32[i], {from20[i], seal65[i]}))
 *                         — what the handle* closure would return for it;
 *   tally                 = ValidatorManager.HasQuorum over the rows with both bits (hasQuorumByMsgType of
 *                           the messages that were stored AND survive the closure); with proposer20 != NULL
 *                           (a PREPARE set: proposer20 = the accepted proposal message's From) the tally is
 *                           ValidatorManager.HasPrepareQuorum — see ibft_tally_prepare.
 * payload / off / msg_sig65 / from20 / sender_pre are ibft_verify_senders' columns (sender_pre = its pre_flags),
 * hash32 / hash_len are ibft_verify_hashes', seal65 / valid_pre are ibft_verify_seals' sig65 / pre_flags
 * (seal65 NULL for a PREPARE set; the seal's signer is the message's From, messages/helpers.go:22-35).  The proposal is given as raw bytes + round (hashed on the
 * device once and remembered, like ibft_verify_hashes) or, when digest32 != NULL, as its 32-byte digest.
 * sender_pre[i] != 0 clears the sender bit of row i, valid_pre[i] != 0 its valid bit (either may be NULL).
 * Both signatures of every message go through ONE verdict
 * launch of 2n rows: at n = 4 096 that is two wavefronts per SIMD instead of one, 0.70 ms for 8 192
 * signatures against 2 × 0.45 ms (profiles/).  Verdicts are bit-identical to the three separate calls.     */
int ibft_verify_messages(ibft_ctx *ctx, const uint8_t *payload, const uint32_t *off, const uint8_t *msg_sig65,
                         const uint8_t *from20, const uint8_t *hash32, const uint8_t *hash_len,
                         const uint8_t *seal65, const uint8_t *sender_pre, const uint8_t *valid_pre, size_t n,
                         const uint8_t *raw, size_t raw_len, uint64_t round, const uint8_t *digest32,
                         const uint8_t *proposer20, uint64_t *out_sender_mask, uint64_t *out_valid_mask,
                         ibft_tally_t *tally);

/* The same for a batch of messages AS THE TRANSPORT DELIVERED THEM (ibft_verify_senders_wire's input): the
 * device walks the bytes, and every canonical PREPARE / COMMIT message of the view (height, round) is judged
 * completely in one verdict launch — no proto.Unmarshal, no PayloadNoSig re-marshal, no flattening on the host.
 *   out_sender_mask bit i = IsValidValidator(message i)        (0 for rows with out_rows[i].status ==
 *                           IBFT_WIRE_NEEDS_HOST: those take the stock route, exactly as after
 *                           ibft_verify_senders_wire);
 *   out_valid_mask  bit i = message i is a PREPARE or COMMIT of (height, round) — type and payload agree, 20-byte
 *                           From, 65-byte seal for a COMMIT — whose 32-byte proposal hash equals the proposal's
 *                           and, for a COMMIT, whose committed seal verifies for From: the handlePrepare /
 *                           handleCommit closure (core/ibft.go:856-862, :932-944).  Messages of other views or
 *                           kinds have bit 0 here and are judged when their view is handled;
 *   tally                 = HasQuorum over the rows with both bits (meaningful for a batch of one type) —
 *                           HasPrepareQuorum when proposer20 != NULL (a batch of PREPAREs of the asked view; in a
 *                           batch of both types only a PREPARE of the proposer counts in proposer_rows and voids
 *                           the quorum — his COMMIT is no PREPARE of his, validator_manager.go:114-121);
 *   out_class[i] (n bytes, may be NULL) = what the caller needs to route row i: IBFT_WIRE_CLASS_NEEDS_HOST — not
 *                           judged here, stock route; IBFT_WIRE_CLASS_CLOSURE — a PREPARE / COMMIT of the asked
 *                           view: its valid bit IS the closure's verdict; bits 4..7 = IbftMessage.type.  One byte
 *                           per row, delivered with the masks;
 *   out_rows (may be NULL) = the full parse results, 80 B per row through a copy command — only when the
 *                           fields themselves are wanted.
 * The proposal: raw + proposal_round (hashed once and remembered) or digest32, as in ibft_verify_messages.   */
#define IBFT_WIRE_CLASS_NEEDS_HOST 0x01u
#define IBFT_WIRE_CLASS_CLOSURE 0x02u
int ibft_verify_messages_wire(ibft_ctx *ctx, const uint8_t *wire_bytes, const uint32_t *off, size_t n,
                              uint64_t height, uint64_t round, const uint8_t *raw, size_t raw_len,
                              uint64_t proposal_round, const uint8_t *digest32, uint64_t *out_sender_mask,
                              uint64_t *out_valid_mask, uint8_t *out_class, ibft_wire_row_t *out_rows,
                              const uint8_t *proposer20, ibft_tally_t *tally);

/* ---- SURVEY.md §8f rank 2 from the transport's bytes: certificates --------------------------------------------
 * A PREPREPARE message carries a RoundChangeCertificate, a ROUND_CHANGE message a PreparedCertificate
 * (/root/reference/messages/proto/messages.proto:46-57, 73-101): IbftMessages inside IbftMessages.  The reference
 * verifies every one of them — validateProposal / validPC / handleRoundChangeMessage → GetExtendedRCC
 * (/root/reference/core/ibft.go:470-551, 683-788, 1162-1231; messages/messages.go:202-245): IsValidValidator on the
 * envelope of each nested message (PayloadNoSig re-marshalled per message, proto/helper.go:12-27) and
 * IsValidProposalHash on the hash each one carries — O(N²) signatures per round change, after proto.Unmarshal has
 * materialised the whole pointer graph.  This call takes the messages AS THE TRANSPORT DELIVERED THEM (row i of the call
 * is wire[off[i]..off[i+1]), any IbftMessage type) and judges the whole tree on the device:
 *
 *   rows      one per IbftMessage of the tree, breadth first: rows [0, n) are the call's messages (level 0), the
 *             messages nested directly inside level-k rows are level k + 1 — children of row i before children of row
 *             j > i, a row's children in wire order (PreparedCertificate: proposalMessage first, then prepareMessages;
 *             RoundChangeCertificate: roundChangeMessages).  *out_n_rows = their number.  Because a decoded message
 *             lists its nested messages in the same order, the host maps row → *proto.IbftMessage by position.
 *   out_nodes[row]  where the row lies and what it is to its parent (ibft_cert_node_t), may be NULL
 *   out_rows[row]   its parsed fields, as ibft_verify_senders_wire reports them; payload_kind 5 / 8 rows carry
 *                   PrePrepareMessage.proposalHash in proposal_hash.  May be NULL
 *   out_class[row]  0 = judged here.  IBFT_CERT_CLASS_NEEDS_HOST: the bytes of this message — or of a message below it —
 *                   are not the canonical encoding (unknown fields, non-minimal varints, …): PayloadNoSig is not
 *                   "bytes minus the signature field", the sender bit is 0 and the caller decides this message by the stock
 *                   route (rows below it that are canonical are still judged).  IBFT_CERT_CLASS_DIGEST_BY_HOST: canonical
 *                   but longer than IBFT_CERT_DIGEST_MAX_BYTES — Keccak is sequential: the wavefront that hashes a long
 *                   message absorbs ≈25 MB/s — so the sender bit is 0 and the host hashes bytes[0, cut0) ‖ bytes[cut1, len)
 *                   itself (out_nodes) and asks ibft_verify_seals with that digest.  IBFT_CERT_CLASS_PROPOSAL_BY_HOST: the Proposal this message
 *                   carries is that long: the self bit of this row and the hash bits of its children are undecided
 *   out_sender_mask bit row = IsValidValidator(message): its envelope signature recovers to From, From is a validator
 *   out_hash_mask   bit row = the 32-byte proposal hash this message carries equals keccak(lastPreparedProposal) of the
 *                   ROUND_CHANGE message whose PreparedCertificate contains it (proposalMatchesCertificate,
 *                   core/ibft.go:516-551: IsValidProposalHash(proposal, hash) for every message of the certificate); 0 for
 *                   rows that are not inside a PreparedCertificate
 *   out_self_mask   bit row = a PREPREPARE payload's proposalHash equals keccak(its own Proposal)
 *                   (validateProposalCommon, core/ibft.go:640-651)
 * All masks have room for ⌈rows_cap/64⌉ words (the first ⌈*out_n_rows/64⌉ are written); out_hash_mask, out_self_mask and out_class may
 * be NULL.  The tree must fit rows_cap and the context's max_rows: IBFT_E_TOOBIG otherwise
 * (no verdicts — split the call or take the stock route).  What stays with the caller is everything that is not
 * arithmetic: types, views, rounds, proposer and quorum rules over the From / type / view columns.               */
#define IBFT_CERT_CLASS_NEEDS_HOST 0x01u
#define IBFT_CERT_CLASS_DIGEST_BY_HOST 0x02u
#define IBFT_CERT_CLASS_PROPOSAL_BY_HOST 0x04u
#define IBFT_CERT_DIGEST_MAX_BYTES (1u << 20)
#define IBFT_CERT_ROLE_ROOT 0u          /* one of the call's n messages                          */
#define IBFT_CERT_ROLE_PC_PROPOSAL 1u   /* PreparedCertificate.proposalMessage                   */
#define IBFT_CERT_ROLE_PC_PREPARE 2u    /* PreparedCertificate.prepareMessages[k]                */
#define IBFT_CERT_ROLE_RCC_MESSAGE 3u   /* RoundChangeCertificate.roundChangeMessages[k]         */
#define IBFT_CERT_HAS_PROPOSAL 0x01u    /* flags: a Proposal sub-message is present              */
#define IBFT_CERT_HAS_CERTIFICATE 0x02u /* flags: a certificate wrapper is present               */
#define IBFT_CERT_NO_PARENT 0xFFFFFFFFu
typedef struct {
  uint32_t off, len;                /* the message's bytes in `wire`                                             */
  uint32_t parent, ordinal;         /* containing row (IBFT_CERT_NO_PARENT for level 0), position among its children */
  uint32_t first_child, n_children; /* its nested messages are rows [first_child, first_child + n_children);
                                       first_child means nothing when n_children == 0 (test n_children first)   */
  uint32_t raw_off, raw_len;        /* Proposal.rawProposal it carries (PREPREPARE: proposal; ROUND_CHANGE:
                                       lastPreparedProposal), in `wire`                                          */
  uint64_t proposal_round;          /* Proposal.round                                                            */
  uint32_t cut0, cut1;              /* its signature field (tag, length, bytes) relative to off                  */
  uint8_t level, role, flags;       /* IBFT_CERT_ROLE_*, IBFT_CERT_HAS_* (the class bits 0x04 / 0x08 mirror out_class) */
  uint8_t pad[5];
} ibft_cert_node_t;
int ibft_verify_certificates_wire(ibft_ctx *ctx, const uint8_t *wire, const uint32_t *off, size_t n, size_t rows_cap,
                                  size_t *out_n_rows, ibft_cert_node_t *out_nodes, ibft_wire_row_t *out_rows,
                                  uint8_t *out_class, uint64_t *out_sender_mask, uint64_t *out_hash_mask,
                                  uint64_t *out_self_mask);

/* ---- f4: the signing side, for SIMULATORS ----------------------------------
 * Replaces, for a process that plays n validators at once, the n calls of Backend.BuildCommitMessage
 * (/root/reference/core/backend.go:12-34; sendCommitMessage, core/ibft.go:898-909) that each produce one
 * committed seal: row i gets the 65-byte seal r ‖ s ‖ v of hash32[i] under the secp256k1 key sk32[i]
 * (32 bytes big-endian, must lie in [1, n)), always low-s, v = parity of R.y after the low-s flip — i.e. a
 * seal ibft_verify_seals accepts under every flag — and the signer address keccak256(X‖Y)[12..32) of that
 * key.  out_signer20 and out_ok may be NULL; out_ok[i] = 0 (zero signature, zero address) for a refused key.
 * The nonce is deterministic: k = keccak256(sk ‖ hash ‖ LE32(ctr)) mod n with the first usable ctr
 * (go-ibft_amd/csrc/sign_dev.h) — the CPU oracle's rule, so device seals are byte-identical to the
 * oracle's; it is not RFC 6979.  NOT for a production validator's key: keys cross PCIe in the clear and sit
 * in HBM for the duration of the call (the column is zeroed before the call returns), and the kernel is not
 * written to be constant-time.  On return the batch is STAGED (hash32 / sig65 / signer20 columns are resident
 * exactly as after ibft_seals_stage): ibft_seals_run verifies what was just signed without another upload.  */
int ibft_sign_seals(ibft_ctx *ctx, const uint8_t *sk32, const uint8_t *hash32, size_t n, uint8_t *out_sig65,
                    uint8_t *out_signer20, uint8_t *out_ok);
/* Block the host until the context's stream is idle.                               */
int ibft_sync(ibft_ctx *ctx);
/* Device canary (diagnostic; no reference counterpart — a Backend may log it at start-up and a bench line carries it):
 * the verdict kernels are bound by VALU issue, so a sick or shared device shows in one number.  Runs a 0.24 ms kernel
 * of independent 8-byte VALU instructions at one wavefront per SIMD (three untimed launches, median of five timed)
 * on the context's stream and reports the wall time per instruction per SIMD: 1.89 ns on a healthy MI355X
 * (profiles/r06b_kernel_ab.txt; the bare instruction stream: 1.78 ns, profiles/r06b_ubench_wave.txt); > 5 % off means every throughput figure of that device is off by as much.   */
int ibft_issue_probe(ibft_ctx *ctx, float *ns_per_inst, float *kernel_ms);

/* ---- multi-GPU (BASELINE configs #4 / #5) -------------------------------------------
 * Rows are independent, so a batch of n_total rows is split into `world` contiguous row ranges whose
 * length is a multiple of 64 (every 64-bit verdict word is owned by one rank; the last rank takes the
 * remainder): rank k verifies rows [lo, hi) of ibft_shard_range on its own device, with the validator
 * table replicated (ibft_set_validators on every context).  The one exchange step the north star names —
 * "RCCL all-reduce over xGMI of the per-message valid-bitmask" — happens INSIDE this library: ONE
 * ncclAllReduce(sum, u64) over
 *   [ K × verdict words of every shard | one distinct-sender bitmap segment per rank | valid rows ]
 * (K = 1 for a seal / sender batch, 2 for a message set: sender words and valid words).  The verdict words are
 * disjoint (sum ≡ OR).  The tally is NOT additive — ValidatorManager.HasQuorum counts a validator once however
 * many of its messages are valid (a set of addresses: /root/reference/core/validator_manager.go:86-92, 147-155),
 * and nothing stops rows of one sender from landing in two shards — so every rank ships the bitmap of the
 * validators its tally counted (⌈n_validators/64⌉ words in its own segment of the buffer; the segments of the
 * ranks are not disjoint as sets, hence one segment per rank instead of one summed bitmap), and after the
 * all-reduce every rank ORs the segments and recomputes power, distinct_senders and has_quorum from the MERGED
 * bitmap — bit for bit what ibft_verify_seals computes for the same rows on one device (tests/test_gpu_comm.py
 * feeds the same sender to both sides of a shard seam).  ibft_tally_t.shard_overlap reports how often that happened.
 * Buffer size: 8 B × (K·⌈n_total/64⌉ + world·⌈n_validators/64⌉ + 1), e.g. 72 KiB for 65 536 rows / validators on
 * 8 devices — latency-bound on any link.  librccl is dlopen()ed on first use (IBFT_RCCL_LIB overrides the name),
 * so single-GPU users never need it.
 *
 * Two ways to drive it:
 *  (1) one process, all GPUs — what a Go Backend would call: ibft_group_create + ibft_group_verify_seals /
 *      _senders / _messages (one host thread per device inside the library);
 *  (2) one process (or thread) per GPU: every rank creates its context, rank 0 calls ibft_comm_unique_id and
 *      hands the 128 bytes to the others (any side channel), everybody calls ibft_comm_init (collective);
 *      per batch: ibft_seals_stage + ibft_seals_launch on the rank's shard, ibft_seals_exchange (asynchronous:
 *      runs on its own stream behind the tally, so it overlaps with the next ibft_seals_launch; at most two
 *      exchanges in flight), ibft_seals_fetch_merged (oldest exchange first) → global mask + merged tally.  */
#define IBFT_COMM_ID_BYTES 128
int ibft_shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi); /* pure */
/* layout of the exchange buffer (pure): verdict words per rank, bitmap words per rank, u64 slots in all;
 * n_masks = K above.  Any out pointer may be NULL.                                                        */
int ibft_exchange_layout(uint64_t n_total, uint32_t world, uint32_t n_validators, uint32_t n_masks,
                         uint32_t *words_per_rank, uint32_t *seen_words, uint32_t *slots);
/* Load librccl now (IBFT_RCCL_LIB, librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1 — first that opens) instead of at the
 * first ibft_comm_* call: a process that will ALSO load another copy of RCCL under the same SONAME (importing torch does)
 * calls this first, so that the library and the HIP runtime it was built for stay a pair.  IBFT_E_RCCL if none opens.     */
int ibft_comm_preload(void);
int ibft_comm_unique_id(uint8_t id[IBFT_COMM_ID_BYTES]);
int ibft_comm_init(ibft_ctx *ctx, const uint8_t id[IBFT_COMM_ID_BYTES], uint32_t rank, uint32_t world);
int ibft_comm_destroy(ibft_ctx *ctx);
int ibft_seals_exchange(ibft_ctx *ctx, uint64_t n_total);
/* out_mask: ⌈n_total/64⌉ words (bit g = verdict of global row g); either pointer may be NULL */
int ibft_seals_fetch_merged(ibft_ctx *ctx, uint64_t *out_mask, ibft_tally_t *tally);
/* What the communicator itself says about this rank (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): lets a
 * benchmark line state that the collective really spans `nranks` devices.  IBFT_E_INVAL without a communicator.  */
int ibft_comm_info(ibft_ctx *ctx, uint32_t *rccl_nranks, uint32_t *rccl_rank, int32_t *rccl_device);

typedef struct ibft_group ibft_group;
/* One context per listed device + the collective over them; max_rows_total = largest n (0 = 65536 per device).
 * Distinct devices: an RCCL communicator.  A list that names a device more than once (several contexts sharing
 * one MI355X — how the sharded path is exercised with world > 1 on a one-GPU box) cannot be an RCCL communicator
 * (one rank per device); the all-reduce is then the library's own sum kernel over the ranks' buffers on rank 0's
 * exchange stream, everything around it — pack, unpack, streams, double buffering — being the same code.
 * IBFT_GROUP_COLLECTIVE=local selects that kernel for distinct devices too (needs peer access from device[0]).  */
int ibft_group_create(const int32_t *devices, uint32_t n_devices, uint32_t flags, uint32_t max_rows_total,
                      ibft_group **out);
void ibft_group_destroy(ibft_group *g);
uint32_t ibft_group_size(const ibft_group *g);
int ibft_group_is_local(const ibft_group *g); /* 1: the library's sum kernel is the collective, 0: RCCL */
ibft_ctx *ibft_group_ctx(ibft_group *g, uint32_t i); /* the i-th device's context (diagnostics, ibft_last_tally_wide) */
int ibft_group_set_validators(ibft_group *g, uint64_t height, const uint8_t *addrs20, const uint64_t *power, size_t n);
int ibft_group_set_validators_u256(ibft_group *g, uint64_t height, const uint8_t *addrs20, const uint8_t *power_be32,
                                   size_t n);
int ibft_group_set_seal_digest(ibft_group *g, uint32_t mode, const uint8_t *suffix, size_t suffix_len);
/* IsValidCommittedSeal + HasQuorum for n rows sharded over the group's devices: same arguments and results as
 * ibft_verify_seals, out_mask / tally are the merged (global) ones — identical to ibft_verify_seals on the whole
 * batch whatever the rows contain (duplicated senders across shards included).                              */
int ibft_group_verify_seals(ibft_group *g, const uint8_t *hash32, const uint8_t *sig65, const uint8_t *signer20,
                            const uint8_t *pre_flags, size_t n, uint64_t *out_mask, ibft_tally_t *tally);
/* IsValidValidator + HasQuorum, sharded: ibft_verify_senders' arguments and results.                        */
int ibft_group_verify_senders(ibft_group *g, const uint8_t *payload, const uint32_t *off, const uint8_t *sig65,
                              const uint8_t *from20, const uint8_t *pre_flags, size_t n, uint64_t *out_mask,
                              ibft_tally_t *tally);
/* A whole PREPARE / COMMIT set sharded by MESSAGE: ibft_verify_messages' arguments and results (BASELINE config #3's
 * sequence at validator counts beyond one device's batch); the exchange carries both verdict arrays (K = 2).  */
int ibft_group_verify_messages(ibft_group *g, const uint8_t *payload, const uint32_t *off, const uint8_t *msg_sig65,
                               const uint8_t *from20, const uint8_t *hash32, const uint8_t *hash_len,
                               const uint8_t *seal65, const uint8_t *sender_pre, const uint8_t *valid_pre, size_t n,
                               const uint8_t *raw, size_t raw_len, uint64_t round, const uint8_t *digest32,
                               const uint8_t *proposer20, uint64_t *out_sender_mask, uint64_t *out_valid_mask,
                               ibft_tally_t *tally);
/* Certificate trees sharded by CARRIER: device k expands and judges the trees of its own contiguous range of the call's n
 * messages, all devices at once; ibft_verify_certificates_wire's arguments and results, the rows numbered as ONE call over
 * all n messages numbers them (breadth first).  Verdicts are per row and no tally is taken, so the devices exchange
 * nothing.  rows_cap bounds the merged tree and each device's share of it.                                          */
int ibft_group_verify_certificates_wire(ibft_group *g, const uint8_t *wire, const uint32_t *off, size_t n, size_t rows_cap,
                                        size_t *out_n_rows, ibft_cert_node_t *out_nodes, ibft_wire_row_t *out_rows,
                                        uint8_t *out_class, uint64_t *out_sender_mask, uint64_t *out_hash_mask,
                                        uint64_t *out_self_mask);

#ifdef __cplusplus
}
#endif
#endif
