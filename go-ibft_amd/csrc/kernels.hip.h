// kernels.hip.h — gfx950 kernels of the batch verifier.
//
// Product code.  Rows are go-ibft messages flattened by the host into byte
// columns; each kernel replaces the per-message callback loop of
// /root/reference/messages/messages.go:183-191.
//
//   hash_eq_kernel              a1  IsValidProposalHash   (core/ibft.go:858-861, 938)
//   proposal_hash_kernel            keccak256(raw ‖ BE64(round)), once per batch
//   ecrecover_lane_kernel       a2  IsValidCommittedSeal  (core/ibft.go:943)      cold path,
//   ecrecover_group_kernel<G>   a3  IsValidValidator      (core/ibft.go:1128)     1 / 2,4,8 lanes per signature
//   ecrecover_wave_kernel           same, one wavefront per signature (limbs spread over lanes, wave_fe_dev.h)
//   ecrecover_rows_kernel           same, sixteen lanes (one DPP row) per signature, four signatures per wavefront
//   verify_known_lane_kernel    a2/a3 against the validator's known key (warm path), 1 lane per signature
//   verify_known_group_kernel<G>    same, G = 2..32 lanes per signature
//   verify_known_wave_kernel        same, one wavefront per signature in the row layout of wave_fe_dev.h
//   tally_kernel                a8  HasQuorum             (core/validator_manager.go:77-96)
//   gtab_build_kernel, qtab_build_kernel, qtab_commit_kernel   one-time fixed-base tables
//   lookup_kernel                   sender → validator index for ibft_tally()
//   wire_parse_kernel, wire_stage_seals_kernel   §8f rank 3: wire bytes → columns on the device (wire_dev.h)
//
// Layout in HBM: one contiguous byte column per field (hash N×32, sig N×65, signer N×20,
// pre_flags N) — structure-of-arrays at field granularity.  The lane kernels stage a block's
// 64 rows through LDS with coalesced dword loads and each lane unpacks its own row into limbs held
// in VGPRs; the group kernels read their row directly (the G lanes of a group share the address).
// Rule for every kernel here: no call to an outlined device function under a partial EXEC mask
// (see secp256k1_dev.h:wave_any) — idle lanes compute on dummy operands and skip only stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "recover_dev.h"
#include "verify_dev.h"
#include "sign_dev.h"
#include "wave_fe_dev.h"
#include "wire_dev.h"
#include "cert_wave_dev.h"

namespace ibftk {

using secp::aff;
using secp::jac;
using secp::u256;

constexpr int ROWS_PER_BLOCK = 64; // one wavefront per block in the lane kernel

// ---- validator table lookup (open addressing, linear probing) ----------------------
__device__ __forceinline__ int valset_lookup(const uint32_t *__restrict__ vtab, uint32_t slot_mask,
                                             const uint32_t a[5]) {
  uint32_t s = addr_hash(a) & slot_mask;
  for (uint32_t probe = 0; probe <= slot_mask; probe++) {
    const uint32_t *e = vtab + 6u * s;
    uint32_t tag = e[5];
    if (tag == 0) return -1;
    if (e[0] == a[0] && e[1] == a[1] && e[2] == a[2] && e[3] == a[3] && e[4] == a[4])
      return (int)tag - 1;
    s = (s + 1) & slot_mask;
  }
  return -1;
}

// ---- warm path: remember a recovered key, exactly once per validator ---------------------------
// Several valid rows of the SAME validator in one cold batch are normal (PREPARE + COMMIT of one
// sender, round-change envelopes + certificate messages): the 0→1 transition of pub_state is claimed
// with a compare-and-swap and only the winner stores the key and counts it, so that `learned` is the
// number of validators with a key — the host compares it with n_validators to drop the cold kernel
// (learn_key, below recover_args).
// ---- host columns → HBM in one launch --------------------------------------------------------------------
// A batch arrives as 4–9 separate byte columns.  One hipMemcpyAsync per column costs ≈8 µs of host time each and the
// copies run one behind the other (profiles/r02h_seq_*: 134 µs before the first kernel of a COMMIT set).  When the
// caller keeps its columns in ibft_pinned_alloc buffers the device can read them itself: one launch, every segment
// read with 16-byte loads straight over PCIe (≈1.3 MB: ≈25 µs).  Pageable columns keep the per-column copies.
constexpr int GATHER_MAX = 12;
constexpr int GATHER_BLOCK_BYTES = 256 * 16;
constexpr uint32_t GATHER_DIGEST_LDS = 32 * 1024;
struct gather_args {
  const uint8_t *src[GATHER_MAX];
  uint8_t *dst[GATHER_MAX];
  uint32_t bytes[GATHER_MAX];
  uint32_t first_block[GATHER_MAX + 1];  // segment s owns blocks [first_block[s], first_block[s+1])
  uint32_t n;
  // optional digest job (a message set): the PayloadNoSig column is NOT copied — the blocks after the copy blocks read 64
  // rows each from the host column into LDS (coalesced) and write keccak256(row) into the digest column
  const uint8_t *pay_src;   // host (pinned), or null
  const uint32_t *off_src;  // host (pinned): n_rows + 1 offsets
  uint8_t *digest_dst;      // HBM: n_rows × 32
  uint32_t n_rows, pay_bytes;
};
__device__ __forceinline__ void hash_range_dwords(const uint8_t *__restrict__ in, uint32_t len, uint64_t out4[4]);
// DIGEST = true adds the digest blocks (and their 32 KiB of LDS per workgroup) to the launch.
template <bool DIGEST>
__global__ void __launch_bounds__(256) gather_columns_kernel(gather_args a) {
  __shared__ __attribute__((aligned(16))) uint8_t lbuf[DIGEST ? GATHER_DIGEST_LDS + 16 : 16];
  if (DIGEST && blockIdx.x >= a.first_block[a.n]) {
    // ---- digest blocks: 64 rows; all four wavefronts bring the rows into LDS (PCIe reads are latency-bound: many
    // requests in flight), the first one hashes ----
    const uint32_t row0 = (blockIdx.x - a.first_block[a.n]) * 64u, lane = threadIdx.x & 63u;
    if (row0 >= a.n_rows) return;
    const uint32_t cnt = a.n_rows - row0 < 64u ? a.n_rows - row0 : 64u;
    const uint32_t row = row0 + (lane < cnt ? lane : cnt - 1);
    const uint32_t o0 = a.off_src[row], o1 = a.off_src[row + 1];
    const uint32_t first = __shfl(o0, 0, 64), last = __shfl(o1, (int)cnt - 1, 64);
    const uint32_t b0 = first & ~15u;
    const bool staged = last - b0 <= GATHER_DIGEST_LDS && ((reinterpret_cast<uintptr_t>(a.pay_src) & 15u) == 0);  // block-uniform
    if (staged) {
      for (uint32_t i = 16u * threadIdx.x; b0 + i < last; i += 4096u) {  // 16-byte loads over PCIe, never past the column's end
        if (b0 + i + 16u <= a.pay_bytes) {
          *reinterpret_cast<uint4 *>(lbuf + i) = *reinterpret_cast<const uint4 *>(a.pay_src + b0 + i);
        } else {
          for (uint32_t j = 0; b0 + i + j < a.pay_bytes; j++) lbuf[i + j] = a.pay_src[b0 + i + j];
        }
      }
      __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    uint64_t d[4];
    if (staged)
      hash_range_dwords(lbuf + (o0 - b0), o1 - o0, d);  // dword reads may run a few bytes past the row: inside the buffer
    else
      keccak::hash_bytes(a.pay_src + o0, o1 - o0, d);   // rows too long for the buffer: exact byte reads from the host column
    if (lane < cnt) {
      uint4 *o = reinterpret_cast<uint4 *>(a.digest_dst + 32ull * row);
      o[0] = make_uint4((uint32_t)d[0], (uint32_t)(d[0] >> 32), (uint32_t)d[1], (uint32_t)(d[1] >> 32));
      o[1] = make_uint4((uint32_t)d[2], (uint32_t)(d[2] >> 32), (uint32_t)d[3], (uint32_t)(d[3] >> 32));
    }
    return;
  }
  uint32_t s = 0;
#pragma unroll 1
  while (s + 1 < a.n && blockIdx.x >= a.first_block[s + 1]) s++;
  const uint32_t off = (blockIdx.x - a.first_block[s]) * (uint32_t)GATHER_BLOCK_BYTES + threadIdx.x * 16u;
  const uint32_t len = a.bytes[s];
  if (off >= len) return;
  const uint8_t *src = a.src[s] + off;
  uint8_t *dst = a.dst[s] + off;
  const uint32_t take = len - off < 16u ? len - off : 16u;
  if (take == 16u && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
  } else if (take == 16u && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3u) == 0) {
    const uint32_t *sp = reinterpret_cast<const uint32_t *>(src);
    uint32_t *dp = reinterpret_cast<uint32_t *>(dst);
    const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2], w3 = sp[3];
    dp[0] = w0; dp[1] = w1; dp[2] = w2; dp[3] = w3;
  } else {
    for (uint32_t i = 0; i < take; i++) dst[i] = src[i];
  }
}

// ---- Keccak-256 of a byte range in HBM, one lane, dword loads ------------------------------------------
// keccak::hash_bytes reads the message a byte at a time (it also runs on the CPU test harness): 136 dependent-
// looking byte loads per block cost more than the permutation itself (payload_digest_kernel: 41 µs for one block
// per lane, profiles/r02g_seq_kernel_stats.csv).  Here a block is 35 aligned dword loads issued together and
// realigned with funnel shifts.  Reads up to 7 bytes past the end of the range: the payload buffers carry 256
// bytes of slack (ibftgpu.hip).
__device__ __forceinline__ void hash_range_dwords(const uint8_t *__restrict__ in, uint32_t len, uint64_t out4[4]) {
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u), sh = 8u * mis;
  const uint32_t *p = reinterpret_cast<const uint32_t *>(in - mis);
  for (;;) {
    const uint32_t take = len < 136u ? len : 136u;  // message bytes in this block
    const uint32_t nd = (take + 3u) >> 2;           // dwords that hold them
    uint32_t raw[35];
#pragma unroll
    for (int j = 0; j < 35; j++) raw[j] = (uint32_t)j <= nd ? p[j] : 0u;
    uint32_t w[34];
#pragma unroll
    for (int j = 0; j < 34; j++) {
      const uint32_t v = (uint32_t)(((uint64_t)raw[j + 1] << 32 | raw[j]) >> sh);  // bytes 4j..4j+3 of the block
      const uint32_t have = take > 4u * (uint32_t)j ? take - 4u * (uint32_t)j : 0u;  // how many of them are message bytes
      w[j] = have >= 4u ? v : (have ? (v & ((1u << (8u * have)) - 1u)) : 0u);
      // pad10*1: first bit right after the message … (static register index: no scratch)
      w[j] |= (take < 136u && (take >> 2) == (uint32_t)j) ? 0x01u << (8u * (take & 3u)) : 0u;
    }
#pragma unroll
    for (int i = 0; i < 17; i++) s[i] ^= (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    if (take < 136u) s[16] ^= 0x8000000000000000ULL;               // … last bit at byte 135
    keccak::f1600(s);
    if (take < 136u) break;
    p += 34;
    len -= 136u;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out4[i] = s[i];
}

// ---- fixed-base table build (entries computed by recover_dev.h:gtab_entry) ----------
__global__ void gtab_build_kernel(uint32_t *__restrict__ gtab) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= GTAB_WINDOWS * GTAB_ENTRIES) return;
  gtab_entry(tid / GTAB_ENTRIES, tid % GTAB_ENTRIES, gtab + (size_t)GTAB_ENTRY_DWORDS * tid);
}

// ---- f4: signing side, one lane per seal (sign_dev.h) ------------------------------------------------
// Writes the same columns ibft_seals_stage fills (hash32 is already there; sig65 and signer20 are produced
// here), so the batch it signed is a staged batch: ibft_seals_run verifies it without another upload.
struct sign_args {
  const uint32_t *gtab;
  const uint8_t *sk32;     // n × 32, big-endian secret keys
  const uint8_t *hash32;   // n × 32, the proposal hash each row seals
  uint8_t *sig65;          // n × 65 out
  uint8_t *signer20;       // n × 20 out
  uint8_t *ok;             // n out: 1 = signed, 0 = key outside [1, n)
  uint32_t n;
};
__global__ void __launch_bounds__(ROWS_PER_BLOCK) sign_lane_kernel(sign_args a) {
  const uint32_t row = blockIdx.x * (uint32_t)ROWS_PER_BLOCK + threadIdx.x;
  const bool live = row < a.n;
  const uint32_t src = live ? row : a.n - 1;  // idle lanes sign the last row again and store nothing
  uint8_t sk[32], dg[32];
  const uint32_t *ks = reinterpret_cast<const uint32_t *>(a.sk32 + 32ull * src);
  const uint32_t *ds = reinterpret_cast<const uint32_t *>(a.hash32 + 32ull * src);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t kw = ks[i], dw = ds[i];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      sk[4 * i + b] = (uint8_t)(kw >> (8 * b));
      dg[4 * i + b] = (uint8_t)(dw >> (8 * b));
    }
  }
  u256 r, s;
  uint32_t v, addr[5];
  const bool ok = sign_row(a.gtab, sk, dg, r, s, v, addr);
  if (!live) return;
  uint8_t *o = a.sig65 + 65ull * row;  // 65-byte rows are not dword aligned: byte stores
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      o[4 * (7 - i) + b] = (uint8_t)(r.v[i] >> (8 * (3 - b)));
      o[32 + 4 * (7 - i) + b] = (uint8_t)(s.v[i] >> (8 * (3 - b)));
    }
  }
  o[64] = (uint8_t)v;
  uint32_t *ad = reinterpret_cast<uint32_t *>(a.signer20 + 20ull * row);
#pragma unroll
  for (int i = 0; i < 5; i++) ad[i] = addr[i];
  a.ok[row] = ok ? 1 : 0;
}

// ---- proposal hash + a1 ---------------------------------------------------------------
// One sponge is sequential; one wavefront walks it with the state spread over 25 lanes (wave_sponge: ≈5.3 µs per 136-byte
// block; a single lane — scalar 64-bit code, ≈188 operations per round — needed ≈9.4 µs).  The host hands the message over
// already padded (raw ‖ BE64(round) ‖ pad10*1 to a multiple of the rate, 8-byte aligned), so absorbing a block is one 64-bit
// load and XOR in lanes 0…16.
__global__ void __launch_bounds__(64) proposal_hash_kernel(const uint64_t *__restrict__ padded, uint32_t blocks, uint64_t *__restrict__ out4) {
  __shared__ uint64_t A[32], B[32];
  const uint32_t lane = threadIdx.x;
  cw::wave_sponge sp;
  sp.init(lane);
  for (uint32_t b = 0; b < blocks; b++) {
    if (lane < 17u) sp.s ^= padded[17u * b + lane];
    sp.permute(A, B);
  }
  if (lane < 4u) out4[lane] = sp.s;
}

// bit i of mask = (hash_len[i] == 32 && hash32[i] == H); rows are 32 B so each lane
// reads two 16-B vectors; the ballot packs 64 verdicts into one u64 per wavefront.
__global__ void hash_eq_kernel(const uint8_t *__restrict__ hash32, const uint8_t *__restrict__ hash_len,
                               const uint64_t *__restrict__ H4, uint32_t n,
                               uint64_t *__restrict__ mask) {
  uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (row < n) {
    const uint4 *p = reinterpret_cast<const uint4 *>(hash32 + 32ull * row);
    uint4 a = p[0], b = p[1];
    const uint32_t *h = reinterpret_cast<const uint32_t *>(H4);
    uint32_t diff = (a.x ^ h[0]) | (a.y ^ h[1]) | (a.z ^ h[2]) | (a.w ^ h[3]) | (b.x ^ h[4]) |
                    (b.y ^ h[5]) | (b.z ^ h[6]) | (b.w ^ h[7]);
    ok = diff == 0 && hash_len[row] == 32;
  }
  uint64_t bal = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && row < n) mask[row >> 6] = bal;
}

// ---- ECDSA recover (cold path), one lane per signature ------------------------------------
struct recover_args {
  const uint8_t *hash32;    // n×32   (seals: per-row proposalHash)
  const uint8_t *sig65;     // n×65
  const uint8_t *signer20;  // n×20
  const uint8_t *pre_flags; // n or null
  const uint8_t *payload;   // senders: concatenated PayloadNoSig
  const uint32_t *off;      // senders: n+1 offsets
  const uint32_t *gtab;     // GTAB_WINDOWS×GTAB_ENTRIES×20 dwords
  const uint32_t *vtab;     // validator table
  uint32_t vslot_mask;
  uint32_t n;
  uint32_t flags;           // IBFT_FLAG_*
  uint64_t *mask;           // ⌈n/64⌉ verdict words
  int32_t *vidx;            // n: validator index of the row's sender (or -1)
  // warm path (null / 0 when the key cache is off)
  uint32_t *pub;            // n_validators × 20 dwords: recovered public keys (affine, 10×26 limbs)
  uint32_t *pub_state;      // n_validators: 0 unknown, 1 key known (claimed by CAS), 2 table built
  uint32_t *learned;        // counter of keys learned (host reads it with the tally)
  const uint32_t *qtab;     // n_validators × 32 × 256 × 20 dwords
  uint8_t *warm_done;       // n: 1 = the warm kernel already produced this row's verdict
  uint32_t dummy_validator; // a SLOT whose table is built (operand for lanes with no work)
  // The key tables belong to the DEVICE, not to a context (every context of a device shares them, a validator keeps its
  // slot — and its table — across validator-set changes): vslot[validator index] = the slot of that validator's
  // address in pub / pub_state / qtab, 0xFFFFFFFF = none (the pool is full: the validator stays on the recover path).
  const uint32_t *vslot;
};
__device__ __forceinline__ int key_slot(const recover_args &a, int vi) { return (int)a.vslot[vi]; }
// states of a key-cache slot.  0 → WRITING (claimed by one row's compare-and-swap) → KNOWN (the key is in `pub`) → BUILDING
// (claimed by one build pass: keys that become KNOWN while the pass runs wait for the next one) → BUILT (the table is complete)
constexpr uint32_t KEY_UNKNOWN = 0, KEY_KNOWN = 1, KEY_BUILT = 2, KEY_WRITING = 3, KEY_BUILDING = 5;

__device__ __forceinline__ void learn_key(const recover_args &a, int vi, const aff &Qa) {
  if (!a.pub_state) return;
  const int sl = key_slot(a, vi);
  if (sl < 0 || a.pub_state[sl] != KEY_UNKNOWN) return;
  if (atomicCAS(a.pub_state + sl, KEY_UNKNOWN, KEY_WRITING) != KEY_UNKNOWN) return;  // another row of this validator won the claim
  // The tables are shared by every context of the device, so a build pass of ANOTHER context (another stream) may scan the
  // states while this kernel runs: the key is published (state KEY_KNOWN, release) only after it has been written.
  store_affine(a.pub + (size_t)GTAB_ENTRY_DWORDS * sl, Qa);
  __hip_atomic_store(a.pub_state + sl, KEY_KNOWN, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  atomicAdd(a.learned, 1u);
  a.learned[1] = (uint32_t)sl;  // any learned slot: operand for idle lanes of the warm kernel
}

// Stage the block's rows through LDS (coalesced dword loads) and unpack this lane's row.
struct row_regs {
  u256 r, s, z;
  uint32_t v;
  uint32_t want[5];
  bool live, pre;
  uint32_t row;
};
template <int MODE>
__device__ __forceinline__ row_regs stage_rows(const recover_args &a, uint8_t *lds) {
  uint8_t *l_sig = lds;
  uint8_t *l_hash = lds + ROWS_PER_BLOCK * 65;
  uint8_t *l_from = l_hash + ROWS_PER_BLOCK * 32;
  const uint32_t row0 = blockIdx.x * ROWS_PER_BLOCK;
  const uint32_t rows = min((uint32_t)ROWS_PER_BLOCK, a.n - row0);
  const uint32_t lane = threadIdx.x;
  {
    // sig: rows*65 bytes starting at 65*row0 (a multiple of 4 because row0 % 64 == 0)
    const uint32_t nb = rows * 65;
    const uint32_t *g = reinterpret_cast<const uint32_t *>(a.sig65 + 65ull * row0);
    for (uint32_t i = lane; i < nb / 4; i += ROWS_PER_BLOCK) reinterpret_cast<uint32_t *>(l_sig)[i] = g[i];
    for (uint32_t i = (nb & ~3u) + lane; i < nb; i += ROWS_PER_BLOCK) l_sig[i] = a.sig65[65ull * row0 + i];
    const uint32_t *gf = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row0);
    for (uint32_t i = lane; i < rows * 5; i += ROWS_PER_BLOCK) reinterpret_cast<uint32_t *>(l_from)[i] = gf[i];
    if (MODE == 0) {
      const uint4 *gh = reinterpret_cast<const uint4 *>(a.hash32 + 32ull * row0);
      for (uint32_t i = lane; i < rows * 2; i += ROWS_PER_BLOCK) reinterpret_cast<uint4 *>(l_hash)[i] = gh[i];
    }
  }
  __syncthreads();
  row_regs q;
  q.row = row0 + lane;
  q.live = lane < rows;
  const uint32_t lrow = q.live ? lane : 0;
  q.pre = q.live && a.pre_flags && a.pre_flags[q.row] != 0;
  q.r = secp::from_be32(l_sig + 65 * lrow);
  q.s = secp::from_be32(l_sig + 65 * lrow + 32);
  q.v = l_sig[65 * lrow + 64];
  if (MODE == 0) {
    q.z = secp::from_be32(l_hash + 32 * lrow);
  } else {
    uint64_t d[4];
    uint32_t o0 = q.live ? a.off[q.row] : 0u, o1 = q.live ? a.off[q.row + 1] : 0u;
    hash_range_dwords(a.payload + o0, o1 - o0, d);
    keccak::digest_to_limbs(d, q.z.v);
  }
#pragma unroll
  for (int i = 0; i < 5; i++) q.want[i] = reinterpret_cast<const uint32_t *>(l_from)[5 * lrow + i];
  return q;
}

// MODE 0: seals (digest = hash32 row).  MODE 1: senders (digest = keccak256(payload row)).
// TAB: where the per-lane window table of u2·R lives (recover_dev.h) —
//   TAB_LDS      in the workgroup's LDS (640 B per lane: one wavefront per SIMD, what a batch of ≤ 65 536 rows offers anyway);
//                no private segment at all: the launch moves the rows, the G-table lines and nothing else through HBM;
//   TAB_PRIVATE  in the private segment, read where it is used: 256 registers, TWO resident wavefronts per SIMD — the form for
//                batches that offer more than one wavefront per SIMD (n > 65 536: 15 instead of 18 ns per verify);
//   TAB_PRIVATE_PREFETCH  round 4's form (entries read in front of the doublings, one resident wavefront): kept for the A/B.
constexpr int TAB_LDS = 0, TAB_PRIVATE = 1, TAB_PRIVATE_PREFETCH = 2;
constexpr int STAGE_BYTES = ROWS_PER_BLOCK * (65 + 32 + 20);
template <int MODE, int TAB>
__global__ void __launch_bounds__(ROWS_PER_BLOCK) ecrecover_lane_kernel(recover_args a) {
  // one block of LDS words: first the staging area of the rows, then (TAB_LDS) the window tables — the block is ONE wavefront
  // (ROWS_PER_BLOCK = 64) and its LDS operations execute in order, so the table may overwrite rows already read into registers
  constexpr int LDS_WORDS = TAB == TAB_LDS ? LTAB_WORDS * ROWS_PER_BLOCK : (STAGE_BYTES + 3) / 4;
  static_assert(LTAB_WORDS * ROWS_PER_BLOCK * 4 >= STAGE_BYTES && ROWS_PER_BLOCK == 64, "the table area holds the staging area");
  __shared__ __attribute__((aligned(16))) uint32_t ldsw[LDS_WORDS];
  uint8_t *lds = reinterpret_cast<uint8_t *>(ldsw);
  row_regs q = stage_rows<MODE>(a, lds);
  const uint32_t lane = threadIdx.x;
  // rows the warm kernel already decided keep their bit; a wavefront with nothing left exits
  const bool done = a.warm_done && q.live && a.warm_done[q.row] != 0;
  const bool need = q.live && !done;
  if (!__any(need ? 1 : 0)) return;

  uint32_t got[5];
  aff Qa;
  bool rec;
  if (TAB == TAB_LDS)
    rec = recover_pubkey_with(a.gtab, q.z, q.r, q.s, q.v, a.flags, got, Qa, var_mult_lds<ROWS_PER_BLOCK>{ldsw + lane});
  else
    rec = recover_pubkey_with(a.gtab, q.z, q.r, q.s, q.v, a.flags, got, Qa, var_mult_private<TAB == TAB_PRIVATE_PREFETCH>{});
  bool ok = need && !q.pre && rec;
#pragma unroll
  for (int i = 0; i < 5; i++) ok = ok && (got[i] == q.want[i]);
  // membership: "the signer address is one of the validators" (backend.go:44, 53-54)
  int vi = valset_lookup(a.vtab, a.vslot_mask, q.want);
  ok = ok && vi >= 0;
  if (need) a.vidx[q.row] = vi;
  // a key that hashes to a member's address is remembered for the warm path
  if (ok) learn_key(a, vi, Qa);
  uint64_t bal = __ballot(ok);
  uint64_t keep = __ballot(done);
  if (lane == 0) {
    uint64_t *w = a.mask + (blockIdx.x * (uint32_t)ROWS_PER_BLOCK >> 6);
    *w = keep ? ((*w & keep) | (bal & ~keep)) : bal;
  }
}

// ---- warm path, one lane per signature ----------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(ROWS_PER_BLOCK) verify_known_lane_kernel(recover_args a) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[ROWS_PER_BLOCK * (65 + 32 + 20)];
  row_regs q = stage_rows<MODE>(a, lds);
  const uint32_t lane = threadIdx.x;
  int vi = valset_lookup(a.vtab, a.vslot_mask, q.want);
  const int sl = vi >= 0 ? key_slot(a, vi) : -1;
  const bool have_table = sl >= 0 && a.pub_state[sl] == KEY_BUILT;
  // decided here: pre-flagged rows and non-members (verdict false), and rows whose validator has a table
  const bool decided = q.live && (q.pre || vi < 0 || have_table);
  const bool crypto = q.live && !q.pre && have_table;
  bool ok = false;
  if (__any(crypto ? 1 : 0)) {  // wave-uniform: every lane runs the arithmetic, idle ones on a dummy table
    const uint32_t tv = crypto ? (uint32_t)sl : a.dummy_validator;
    ok = verify_known(a.gtab, a.qtab + QTAB_DWORDS_PER_VALIDATOR * tv, q.z, q.r, q.s, q.v, a.flags) && crypto;
  }
  if (q.live) {
    a.warm_done[q.row] = decided ? 1 : 0;
    if (decided) a.vidx[q.row] = vi;
  }
  uint64_t bal = __ballot(ok);
  if (lane == 0) a.mask[blockIdx.x * (uint32_t)ROWS_PER_BLOCK >> 6] = bal;
}

// ---- warm path, G LANES PER SIGNATURE (G = 64: one wavefront per signature) -------------------
// The 32 window points of u2·Q (validator's table) and the 16 window points of u1·G are dealt
// round-robin to the G lanes of a group; each lane sums its share with mixed additions, then a
// log2(G)-level xor-butterfly of Jacobian additions leaves the total in every lane.  The scalar
// work (s⁻¹ mod n, final Z⁻¹) is replicated across the group.  G is chosen by the host so that a
// batch of n rows yields about one wavefront per SIMD (n·G/64 ≈ 1024): G = 64 up to 1 024 rows,
// 16 at 4 096, 4 at 16 384, and the LDS-staged lane kernel (G = 1) beyond.
__device__ __forceinline__ jac shfl_xor_jac(const jac &p, int off) {
  jac r;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    r.x.n[i] = __shfl_xor(p.x.n[i], off, 64);
    r.y.n[i] = __shfl_xor(p.y.n[i], off, 64);
    r.z.n[i] = __shfl_xor(p.z.n[i], off, 64);
  }
  r.inf = __shfl_xor(p.inf ? 1 : 0, off, 64) != 0;
  return r;
}

template <int MODE, int G>
__global__ void __launch_bounds__(64) verify_known_group_kernel(recover_args a) {
  constexpr int ROWS = 64 / G;                    // signatures per wavefront
  constexpr int POINTS = QTAB_WINDOWS + GTAB_WINDOWS;
  // Round 5: the kernel's CODE has to fit the 64 KB instruction cache — on one lease in four a fetch past it costs 65 % more
  // and a 60 KB kernel ran 0.132 instead of 0.110 ms.  Both point additions inlined (11 + 16 multiplications
  // of 224 instructions) made 60.5 KB; only the one that dominates a lane's work is pasted now — the butterfly for G ≥ 16 (4–5
  // folds of 16 multiplications against 2–3 mixed additions of 11), the mixed addition for G ≤ 8 — and the other one calls the
  // outlined multiply (≈ 25 instructions per call: +2 % of this kernel's instructions).
  constexpr bool INL_FOLD = G >= 16, INL_MADD = !INL_FOLD;
  const uint32_t lane = threadIdx.x;
  const uint32_t sub = lane % G;
  const uint32_t row_raw = blockIdx.x * ROWS + lane / G;
  const bool live = row_raw < a.n;
  const uint32_t row = live ? row_raw : a.n - 1;  // idle groups recompute the last row, store nothing
  const bool pre = a.pre_flags && a.pre_flags[row] != 0;
  u256 r = secp::from_be32(a.sig65 + 65ull * row);
  u256 s = secp::from_be32(a.sig65 + 65ull * row + 32);
  const uint32_t v = a.sig65[65ull * row + 64];
  u256 z;
  if (MODE == 0) {
    z = secp::from_be32(a.hash32 + 32ull * row);
  } else {
    uint64_t d[4];
    hash_range_dwords(a.payload + a.off[row], a.off[row + 1] - a.off[row], d);
    keccak::digest_to_limbs(d, z.v);
  }
  uint32_t want[5];
#pragma unroll
  for (int i = 0; i < 5; i++) want[i] = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row)[i];
  const int vi = valset_lookup(a.vtab, a.vslot_mask, want);
  const int sl = vi >= 0 ? key_slot(a, vi) : -1;
  const bool have_table = sl >= 0 && a.pub_state[sl] == KEY_BUILT;
  const bool decided = pre || vi < 0 || have_table;
  const bool crypto = live && !pre && have_table;
  if (live && sub == 0) {
    a.warm_done[row] = decided ? 1 : 0;
    if (decided) a.vidx[row] = vi;
  }
  if (!__any(crypto ? 1 : 0)) return;  // wave-uniform: from here every lane runs the same stream

  bool ok = sig_in_range(r, s, v, a.flags);
  u256 u1, u2;
  // G ≥ 16: a signature owns whole DPP rows, and all their lanes hold the same (z, r, s) — the two inversions (s⁻¹ mod n
  // here, Z⁻¹ mod p at the end) run with the limbs of d, e, f, g over the lanes of the row (wave_fe_dev.h:modinv_wave,
  // ≈9.5 k issue slots) instead of every lane running the whole constant-time safegcd on its own copy (≈15 k): they were
  // 43 % of this kernel at N = 4 096.  Smaller groups share a row between signatures and keep the lane form.
  constexpr bool ROW_INV = G >= 16;
  const wv::wk wkc = wv::wk_init();
  if (ROW_INV) {
    const secp::sc sinv = secp::sc_from_u256(wv::modinv_wave<secp::ModN>(s, wkc));
    u1 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z), sinv));
    u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(r), sinv));
  } else {
    verify_scalars(z, r, s, u1, u2);
  }
  const uint32_t *qt = a.qtab + QTAB_DWORDS_PER_VALIDATOR * (crypto ? (uint32_t)sl : a.dummy_validator);
  jac acc = secp::jac_inf();
  // this lane's table point of step `it` (point it·G + sub of the 48): where it lies, its digit, whether it exists
  auto point_of = [&](int it, uint32_t &dgt, bool &has) -> const uint32_t * {
    const int p = it * G + (int)sub;
    has = p < POINTS;
    const int pp = has ? p : 0;
    if (pp < QTAB_WINDOWS) {
      dgt = (u2.v[pp >> 2] >> (8 * (pp & 3))) & 255u;
      return qt + (size_t)GTAB_ENTRY_DWORDS * (pp * QTAB_ENTRIES + dgt);
    }
    const int w = pp - QTAB_WINDOWS;
    dgt = (u1.v[(w * GTAB_BITS) >> 5] >> ((w * GTAB_BITS) & 31)) & (uint32_t)(GTAB_ENTRIES - 1);
    return a.gtab + (size_t)GTAB_ENTRY_DWORDS * ((size_t)w * GTAB_ENTRIES + dgt);
  };
  // Round 6: software-pipelined by one — the entry of step it + 1 is asked for before the addition of step it runs (a
  // dependent read of a validator's table, far beyond any cache, used to open every step: ≈ 2 µs in front of each addition)
  constexpr int STEPS = (POINTS + G - 1) / G;  // wave-uniform trip count
  uint32_t dgt;
  bool has;
  aff cur = load_affine(point_of(0, dgt, has));
#pragma unroll 1
  for (int it = 0; it < STEPS; it++) {
    uint32_t dn;
    bool hasn;
    const aff nxt = load_affine(point_of(it + 1 < STEPS ? it + 1 : it, dn, hasn));  // (the last step re-reads its own entry)
    jac sum = secp::jac_add_aff_t<INL_MADD>(acc, cur);  // (the one inlined copy of the mixed addition: secp256k1_dev.h)
    acc = secp::jac_select(has && dgt != 0, sum, acc);
    cur = nxt;
    dgt = dn;
    has = hasn;
  }
#pragma unroll 1
  for (int off = G / 2; off >= 1; off >>= 1) {
    jac other = shfl_xor_jac(acc, off);
    acc = secp::jac_add_t<INL_FOLD>(acc, other);  // (one inlined copy in the rolled butterfly)
  }
  if (ROW_INV) {
    // The butterfly left the SUM in every lane of the group, but not one REPRESENTATION of it: P + Q and Q + P come out
    // with Z of opposite sign, so the lanes of a group hold (λ²X, λ³Y, λZ) for different λ.  The row-layout inversion
    // takes its limbs from all lanes of a row: give every lane the Z of the group's first lane; that lane then holds a
    // consistent (X, Y, Z) and is the one that reports.
#pragma unroll
    for (int i = 0; i < 10; i++)
      acc.z.n[i] = G == 16 ? wv::row_bcast<0>(acc.z.n[i]) : (uint32_t)__shfl((int)acc.z.n[i], (int)(lane & ~(uint32_t)(G - 1)), 64);
    acc.inf = __shfl(acc.inf ? 1 : 0, (int)(lane & ~(uint32_t)(G - 1)), 64) != 0;
    aff A;
    const bool fin = wv::jac_to_aff_wave(A, acc, wkc);
    const secp::fe rx = secp::fe_from_u256(r);  // r < n < p: canonical limbs
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) diff |= A.x.n[i] ^ rx.n[i];
    ok = fin && diff == 0 && (A.y.n[0] & 1u) == v && ok && crypto;
  } else {
    ok = verify_finish(acc, r, v) && ok && crypto;
  }
  if (sub == 0 && ok) atomicOr(reinterpret_cast<unsigned long long *>(a.mask + (row >> 6)), 1ull << (row & 63));
}

// WAVE_KERNEL_WAVES wavefronts (= signatures) per workgroup: the four wavefronts of a workgroup land on
// the four SIMDs of one CU, so a 1 024-row launch is exactly one wavefront per SIMD by construction
// instead of by the dispatcher's choice.
constexpr int WAVE_KERNEL_WAVES = 4;
// ---- warm path, one wavefront per signature with the limbs spread over lanes (wave_fe_dev.h) ----
// Replaces verify_known_group_kernel<·,64>: same prologue, the point sum runs in the row layout.
template <int MODE>
__global__ void __launch_bounds__(64 * WAVE_KERNEL_WAVES) verify_known_wave_kernel(recover_args a) {
  const uint32_t row = blockIdx.x * WAVE_KERNEL_WAVES + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (row >= a.n) return;  // whole wavefront
  const bool pre = a.pre_flags && a.pre_flags[row] != 0;
  const u256 r = secp::from_be32(a.sig65 + 65ull * row);
  const u256 s = secp::from_be32(a.sig65 + 65ull * row + 32);
  const uint32_t v = a.sig65[65ull * row + 64];
  u256 z;
  if (MODE == 0) {
    z = secp::from_be32(a.hash32 + 32ull * row);
  } else {
    uint64_t d[4];
    hash_range_dwords(a.payload + a.off[row], a.off[row + 1] - a.off[row], d);
    keccak::digest_to_limbs(d, z.v);
  }
  uint32_t want[5];
#pragma unroll
  for (int i = 0; i < 5; i++) want[i] = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row)[i];
  const int vi = valset_lookup(a.vtab, a.vslot_mask, want);
  const int sl = vi >= 0 ? key_slot(a, vi) : -1;
  const bool have_table = sl >= 0 && a.pub_state[sl] == KEY_BUILT;
  const bool decided = pre || vi < 0 || have_table;
  if (lane == 0) {
    a.warm_done[row] = decided ? 1 : 0;
    if (decided) a.vidx[row] = vi;
  }
  if (pre || !have_table) return;  // wave-uniform: the wavefront holds one row
  const bool ok = wv::verify_known_wave(a.gtab, a.qtab + QTAB_DWORDS_PER_VALIDATOR * (uint32_t)sl, z, r, s, v, a.flags);
  if (lane == 0 && ok) atomicOr(reinterpret_cast<unsigned long long *>(a.mask + (row >> 6)), 1ull << (row & 63));
}

// ---- cold path with G = 2, 4 or 8 lanes per signature ----------------------------------------
// The GLV split already yields two independent halves (k1·R, k2·λR); each half is cut into
// P = G/2 pieces of 128/P bits with bases 2^(j·128/P)·R, so a lane does 128/P doublings instead of
// 128 and the pieces are joined by a log2(G)-level butterfly.  Everything that does not split
// (√, r⁻¹, the prefix doublings that produce the bases, the per-lane table, Z⁻¹, Keccak) is
// replicated across the group, so the critical path shrinks from ≈676 k to ≈520 k (G = 2),
// ≈440 k (G = 4), ≈410 k (G = 8) VALU instructions: worth it exactly when the batch is too small
// to fill the chip (host picks G so that n·G/64 ≤ 1024 wavefronts).
template <int MODE, int G, int TAB = TAB_LDS>
__global__ void __launch_bounds__(64) ecrecover_group_kernel(recover_args a) {
  __shared__ uint32_t ldsw[(TAB == TAB_LDS && G != 8) ? LTAB_WORDS * 64 : 1];  // the lanes' window tables (4-bit windows: G = 2, 4)
  constexpr int ROWS = 64 / G;
  constexpr int P = G / 2;            // pieces per GLV half
  constexpr int PIECE_BITS = 128 / P; // 128, 64 or 32 ... (G = 2, 4, 8)
  constexpr int NIBS = PIECE_BITS / 4;
  const uint32_t lane = threadIdx.x;
  const uint32_t sub = lane % G;
  const uint32_t half = sub & 1u;     // 0: k1 on ±R, 1: k2 on λ(±R)
  const uint32_t piece = sub >> 1;    // which 128/P-bit piece of that half
  const uint32_t row_raw = blockIdx.x * ROWS + lane / G;
  const bool live = row_raw < a.n;
  const uint32_t row = live ? row_raw : a.n - 1;
  const bool pre = a.pre_flags && a.pre_flags[row] != 0;
  const bool done = a.warm_done && a.warm_done[row] != 0;
  const bool need = live && !done;
  if (!__any(need ? 1 : 0)) return;

  u256 r = secp::from_be32(a.sig65 + 65ull * row);
  u256 s = secp::from_be32(a.sig65 + 65ull * row + 32);
  const uint32_t v = a.sig65[65ull * row + 64];
  u256 z;
  if (MODE == 0) {
    z = secp::from_be32(a.hash32 + 32ull * row);
  } else {
    uint64_t d[4];
    hash_range_dwords(a.payload + a.off[row], a.off[row + 1] - a.off[row], d);
    keccak::digest_to_limbs(d, z.v);
  }
  uint32_t want[5];
#pragma unroll
  for (int i = 0; i < 5; i++) want[i] = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row)[i];

  bool ok = sig_in_range(r, s, v, a.flags);
  // R = (r, y), y² = r³ + 7, parity(y) = v
  secp::fe rx = secp::fe_from_u256(r);
  secp::fe seven = secp::fe_zero();
  seven.n[0] = 7;
  secp::fe rhs = secp::fe_add(secp::fe_mul(secp::fe_sqr(rx), rx), seven);
  secp::fe y = secp::fe_sqrt_candidate(rhs);
  ok = ok && secp::fe_equal(secp::fe_sqr(y), rhs, 2);
  y = secp::fe_normalize(y);
  y = secp::l26_select((y.n[0] & 1u) != v, secp::fe_normalize_weak(secp::fe_neg(y, 1)), y);
  // u1 = −z/r, u2 = s/r, u2 = k1 + k2·λ
  secp::sc rinv = secp::sc_from_u256(secp::modinv_shared<secp::ModN>(r));
  u256 u1 = secp::sc_neg_canon(secp::sc_canon(secp::sc_mul(secp::sc_from_u256(z), rinv)));
  u256 u2 = secp::sc_canon(secp::sc_mul(secp::sc_from_u256(s), rinv));
  secp::glv_split sp = secp::sc_split_lambda(u2);
  // this lane's base: ±R doubled piece·PIECE_BITS times (uniform trip count: every lane runs the
  // longest prefix and keeps the intermediate it needs)
  aff R1;
  R1.x = rx;
  R1.y = secp::l26_select(sp.neg1, secp::fe_normalize_weak(secp::fe_neg(y, 1)), y);
  jac base = secp::jac_from_aff(R1);
  if (P > 1) {
    jac run = base;
#pragma unroll 1
    for (int j = 1; j < P; j++) {
#pragma unroll 1
      for (int d = 0; d < PIECE_BITS; d++) run = secp::jac_dbl(run);
      base = secp::jac_select(piece >= (uint32_t)j && piece == (uint32_t)j, run, base);
    }
  }
  // per-lane window table over the base (Jacobian).  G = 8 (32-bit pieces): 2-bit windows, the three
  // entries {B, 2B, 3B} live in VGPRs and are picked with selects — no scratch traffic at all
  // (the 4-bit table is 1.9 KB per lane in scratch: 16 MB written + re-read per launch at
  // N = 1024, G = 8).  G = 2, 4 (128/64-bit pieces): 4-bit windows, table in scratch.
  constexpr int WBITS = (G == 8) ? 2 : 4;
  constexpr int DIGITS = PIECE_BITS / WBITS;
  const u256 &kk = half ? sp.k2 : sp.k1;
  const bool flip = half && (sp.neg1 != sp.neg2);
  const secp::fe beta = secp::GLV_CONST(1);
  jac acc = secp::jac_inf();
  if (WBITS == 2) {
    const jac b1 = base, b2 = secp::jac_dbl(base), b3 = secp::jac_add(b2, base);
#pragma unroll 1
    for (int dgi = DIGITS - 1; dgi >= 0; dgi--) {
      acc = secp::jac_dbl(secp::jac_dbl(acc));
      const int bit = (int)(piece * PIECE_BITS) + 2 * dgi;
      const uint32_t dg = (kk.v[bit >> 5] >> (bit & 31)) & 3u;
      jac q = secp::jac_select(dg == 1, b1, secp::jac_select(dg == 2, b2, b3));
      secp::fe bx = secp::fe_mul(q.x, beta);
      q.x = secp::l26_select(half != 0, bx, q.x);
      q.y = secp::l26_select(flip, secp::fe_neg(q.y, 1), q.y);
      jac sum = secp::jac_add(acc, q);
      acc = secp::jac_select(dg != 0, sum, acc);
    }
  } else {
    // signed radix-16 windows over a table of the multiples 1…8 of this lane's base brought to one common Z (recover_dev.h:
    // ecmult_table).  The base is Jacobian (X, Y, Z) — the AFFINE point (X, Y) of the isomorphic curve y² = x³ + 7·Z⁶, on
    // which the table is built and the loop runs; Z·Zc goes back into the accumulator's Z at the end.  The λ half uses the
    // entries' β·X, the sign of a digit is a negation of Y.  Digit j of a piece = nibble(k + Σ 8·16^i, piece·NIBS + j) − 8;
    // the half's 33rd digit (the carry, 0 or 1) belongs to its top piece — the other pieces see a zero digit there.
    aff b1;
    b1.x = base.x;
    b1.y = base.y;
    const u256 kb = window_bias(kk);
    if (TAB == TAB_LDS) {
      // the table in LDS (recover_dev.h: ltab): 640 B per lane, β·X multiplied at use — for every lane, the λ half selects it
      // (half differs from lane to lane: no call under divergent control flow)
      ltab<64> lt;
      lt.col = ldsw + lane;
      ecmult_table_lds<64>(b1, lt);
      // this lane's digits — nibbles piece·NIBS … piece·NIBS + NIBS of kb — as a shift register whose current digit is the low
      // nibble of word NW (round 6: nothing indexed by `at`, no select chain per digit, no private segment)
      constexpr int NW = NIBS / 8;  // 4 words (G = 2), 2 (G = 4)
      u256 kr = secp::zero256();
#pragma unroll
      for (int i = 0; i <= NW; i++) {
        kr.v[i] = kb.v[i];
#pragma unroll
        for (int j = 1; j < P; j++) kr.v[i] = piece == (uint32_t)j ? kb.v[i + NW * j] : kr.v[i];
      }
      // Round 6: the window additions and this lane's share of the fixed-base additions are steps of ONE loop around ONE pasted
      // copy of the mixed addition (recover_dev.h: ecmult_var_gen_lds has the reasoning) — the G additions used to go through
      // the outlined multiply in a loop of their own.  u1 is a shift register (this lane's current window in the low bits of
      // word 0: shifted down by GTAB_BITS·sub once, by GTAB_BITS·G per step), the table entry of the next G step is asked for
      // before the addition of this one runs, the first one before the window loop starts.
      constexpr int GSTEPS = (GTAB_WINDOWS + G - 1) / G;
      constexpr int LOG2G = G == 8 ? 3 : (G == 4 ? 2 : 1);
      u256 ug = u1;
      secp::shr_units<GTAB_BITS, LOG2G>(ug, sub);
      uint32_t dg = ug.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
      bool has = (int)sub < GTAB_WINDOWS;
      gtab_raw cur = gtab_load(a.gtab, has ? (int)sub : 0, has ? dg : 0u);
#pragma unroll 1
      for (int st = 0; st <= NIBS + GSTEPS; st++) {
        aff q;
        bool take;
        gtab_raw nxt = cur;
        uint32_t dn = dg;
        bool hasn = has;
        if (st <= NIBS) {  // (wave-uniform) window step: digit NIBS − st of this lane's piece
          const bool mine = st > 0 || piece == (uint32_t)(P - 1);
          const int e = mine ? (int)secp::top_nibble<NW + 1>(kr) - 8 : 0;
          secp::shl4<NW + 1>(kr);
          if (st != 0) {
#pragma unroll 1
            for (int d = 0; d < 4; d++) acc = secp::jac_dbl_t<true>(acc);
          }
          q = window_operand_lds<64>(lt, e, flip);
          q.x = secp::l26_select(half != 0, secp::fe_mul(q.x, beta), q.x);
          take = e != 0;
        } else {           // fixed-base step: window (st − NIBS − 1)·G + sub of u1
          const int it = st - NIBS - 1;
          if (it == 0) acc.z = secp::fe_mul(acc.z, secp::fe_mul(lt.zc, base.z));  // back from the isomorphic curve first
          secp::shr_const<GTAB_BITS * G>(ug);
          const int wn = (it + 1) * G + (int)sub;
          hasn = it + 1 < GSTEPS && wn < GTAB_WINDOWS;
          dn = ug.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
          nxt = gtab_load(a.gtab, hasn ? wn : 0, hasn ? dn : 0u);  // (past the end: entry 0 of window 0, never added)
          q = gtab_point(cur);
          take = has && dg != 0;
        }
        const jac sum = secp::jac_add_aff_t<true>(acc, q);
        acc = secp::jac_select(take, sum, acc);
        if (st > NIBS) {  // (only the fixed-base steps move the pipeline)
          cur = nxt;
          dg = dn;
          has = hasn;
        }
      }
    } else {
      wtab wt;
      ecmult_table(b1, wt);
#pragma unroll 1
      for (int nib = NIBS; nib >= 0; nib--) {
        const int at = (int)(piece * NIBS) + nib;
        const bool mine = nib < NIBS || piece == (uint32_t)(P - 1);
        const int e = mine ? (int)secp::nibble(kb, at) - 8 : 0;
        const aff q = window_operand(wt, e, half != 0, flip);  // read in front of the doublings that cover its latency
        if (nib != NIBS) {
#pragma unroll 1
          for (int d = 0; d < 4; d++) acc = secp::jac_dbl_t<true>(acc);
        }
        acc = window_add_q(acc, q, e);
      }
      acc.z = secp::fe_mul(acc.z, secp::fe_mul(wt.zc, base.z));
    }
  }
  // u1·G: the fixed-base windows are dealt to the lanes of the group — lane `sub` takes windows sub, sub + G, … .  Round 6: u1 is
  // a shift register (this lane's current window in the low bits of word 0: shifted down by GTAB_BITS·sub once, by GTAB_BITS·G
  // per step — nothing indexed), and the table entry of the NEXT step is asked for before the addition of this one runs.
  if (!(WBITS == 4 && TAB == TAB_LDS)) {  // (the LDS-table form has run them as steps of its window loop)
    constexpr int STEPS = (GTAB_WINDOWS + G - 1) / G;
    constexpr int LOG2G = G == 8 ? 3 : (G == 4 ? 2 : 1);
    u256 ug = u1;
    secp::shr_units<GTAB_BITS, LOG2G>(ug, sub);
    uint32_t dg = ug.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
    bool has = (int)sub < GTAB_WINDOWS;
    gtab_raw cur = gtab_load(a.gtab, has ? (int)sub : 0, has ? dg : 0u);
#pragma unroll 1
    for (int it = 0; it < STEPS; it++) {
      secp::shr_const<GTAB_BITS * G>(ug);
      const int wn = (it + 1) * G + (int)sub;
      const bool hasn = it + 1 < STEPS && wn < GTAB_WINDOWS;
      const uint32_t dn = ug.v[0] & (uint32_t)(GTAB_ENTRIES - 1);
      const gtab_raw nxt = gtab_load(a.gtab, hasn ? wn : 0, hasn ? dn : 0u);  // (past the end: entry 0 of window 0, never added)
      const jac sum = secp::jac_add_aff(acc, gtab_point(cur));
      acc = secp::jac_select(has && dg != 0, sum, acc);
      cur = nxt;
      dg = dn;
      has = hasn;
    }
  }
#pragma unroll 1
  for (int off = G / 2; off >= 1; off >>= 1) {
    jac other = shfl_xor_jac(acc, off);
    // (round 5: through the outlined multiply — one to three folds per signature do not pay for 28 KB of pasted code in a
    // kernel that is far over the 64 KB instruction cache already: DESIGN.md §5.8)
    acc = secp::jac_add_t<false>(acc, other);
  }
  aff Qa;
  ok = secp::jac_to_aff_fast(Qa, acc) && ok;
  u256 qx = secp::l26_to_u256(Qa.x), qy = secp::l26_to_u256(Qa.y);
  uint32_t got[5];
  keccak::address_from_xy(qx.v, qy.v, got);
  ok = ok && need && !pre;
#pragma unroll
  for (int i = 0; i < 5; i++) ok = ok && (got[i] == want[i]);
  int vi = valset_lookup(a.vtab, a.vslot_mask, want);
  ok = ok && vi >= 0;
  if (sub == 0 && need) {
    a.vidx[row] = vi;
    if (ok) learn_key(a, vi, Qa);
    if (ok) atomicOr(reinterpret_cast<unsigned long long *>(a.mask + (row >> 6)), 1ull << (row & 63));
  }
}

// ---- cold path, ONE WAVEFRONT PER SIGNATURE (wave_fe_dev.h) --------------------------------------
// For batches that leave most of the chip idle (n ≤ 2 048: at most two wavefronts per SIMD).  A
// field element is one VGPR spread over the 16 lanes of a DPP row, a multiplication costs ≈86
// instructions for four independent products, and the four rows carry the four 64-bit pieces
// of the GLV-split scalar.  All control flow is uniform: the wavefront holds a single signature.
template <int MODE>
__global__ void __launch_bounds__(64 * WAVE_KERNEL_WAVES) ecrecover_wave_kernel(recover_args a) {
  const uint32_t row = blockIdx.x * WAVE_KERNEL_WAVES + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (row >= a.n) return;                            // whole wavefront
  if (a.warm_done && a.warm_done[row] != 0) return;  // decided by the warm kernel
  uint32_t want[5];
#pragma unroll
  for (int i = 0; i < 5; i++) want[i] = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row)[i];
  const int vi = valset_lookup(a.vtab, a.vslot_mask, want);
  const bool pre = a.pre_flags && a.pre_flags[row] != 0;
  if (lane == 0) a.vidx[row] = vi;
  if (pre || vi < 0) return;  // verdict stays 0 (mask was cleared by the host before the launch)
  const u256 r = secp::from_be32(a.sig65 + 65ull * row);
  const u256 s = secp::from_be32(a.sig65 + 65ull * row + 32);
  const uint32_t v = a.sig65[65ull * row + 64];
  u256 z;
  if (MODE == 0) {
    z = secp::from_be32(a.hash32 + 32ull * row);
  } else {
    uint64_t d[4];
    hash_range_dwords(a.payload + a.off[row], a.off[row + 1] - a.off[row], d);
    keccak::digest_to_limbs(d, z.v);
  }
  uint32_t got[5];
  aff Qa;
  bool ok = wv::recover_pubkey_wave(a.gtab, z, r, s, v, a.flags, got, Qa);
#pragma unroll
  for (int i = 0; i < 5; i++) ok = ok && (got[i] == want[i]);
  if (lane == 0 && ok) {
    learn_key(a, vi, Qa);
    atomicOr(reinterpret_cast<unsigned long long *>(a.mask + (row >> 6)), 1ull << (row & 63));
  }
}

// ---- cold path, TWO WAVEFRONTS PER SIGNATURE (n ≤ 512: round 4) ----------------------------------------------------
// Half of the chip's SIMDs idle when n ≤ 512 signatures run one wavefront each.  A workgroup here is PAIRS_PER_BLOCK = 2 pairs
// — four wavefronts, which land on the four SIMDs of ONE compute unit, so 512 rows are exactly one wavefront per SIMD of
// the chip by construction: wavefront w < 2 is the MAIN wavefront of signature slot w (prefix doublings with √, table, main
// loop, joins, Z⁻¹, Keccak), wavefront 2 + w its HELPER (r⁻¹ mod n, u₁, u₂, GLV split; then u₁·G) — the two meet in LDS at two
// workgroup barriers (wave_fe_dev.h: recover_pubkey_wave<…, PAIR>, recover_helper_wave).  Every wavefront of the workgroup
// passes both barriers, whatever its row turns out to be: a pair with nothing to do only synchronises.
constexpr int PAIRS_PER_BLOCK = 2;
template <int MODE>
__global__ void __launch_bounds__(128 * PAIRS_PER_BLOCK) ecrecover_wave2_kernel(recover_args a) {
  __shared__ wv::pair_shared sh[PAIRS_PER_BLOCK];
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t slot = w % PAIRS_PER_BLOCK;
  const bool helper = w >= PAIRS_PER_BLOCK;
  const uint32_t row_raw = blockIdx.x * PAIRS_PER_BLOCK + slot;
  const bool live = row_raw < a.n;
  const uint32_t row = live ? row_raw : a.n - 1;
  const bool done = a.warm_done && a.warm_done[row] != 0;
  uint32_t want[5];
#pragma unroll
  for (int i = 0; i < 5; i++) want[i] = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row)[i];
  const int vi = valset_lookup(a.vtab, a.vslot_mask, want);
  const bool pre = a.pre_flags && a.pre_flags[row] != 0;
  if (!helper && lane == 0 && live && !done) a.vidx[row] = vi;
  const bool active = live && !done && !pre && vi >= 0;  // the same for both wavefronts of the pair
  auto sync = [] { __syncthreads(); };
  if (!active) {  // (wave-uniform) nothing to recover: keep the workgroup's barrier count
    sync();
    sync();
    return;
  }
  const u256 r = secp::from_be32(a.sig65 + 65ull * row);
  const u256 s = secp::from_be32(a.sig65 + 65ull * row + 32);
  const uint32_t v = a.sig65[65ull * row + 64];
  u256 z;
  if (MODE == 0) {
    z = secp::from_be32(a.hash32 + 32ull * row);
  } else {
    uint64_t d[4];
    hash_range_dwords(a.payload + a.off[row], a.off[row + 1] - a.off[row], d);
    keccak::digest_to_limbs(d, z.v);
  }
  if (helper) {
    wv::recover_helper_wave(a.gtab, z, r, s, &sh[slot], sync);
    return;
  }
  uint32_t got[5];
  aff Qa;
  bool ok = wv::recover_pubkey_wave<99, true>(a.gtab, z, r, s, v, a.flags, got, Qa, &sh[slot], sync);
#pragma unroll
  for (int i = 0; i < 5; i++) ok = ok && (got[i] == want[i]);
  if (lane == 0 && ok) {
    learn_key(a, vi, Qa);
    atomicOr(reinterpret_cast<unsigned long long *>(a.mask + (row >> 6)), 1ull << (row & 63));
  }
}

// ---- cold path, SIXTEEN LANES PER SIGNATURE: every row of a wavefront recovers its own signature ------
// (wave_fe_dev.h:recover_pubkey_row).  n = 4 096 is one wavefront per SIMD again; used for
// 2 048 < n ≤ 8 192.  Rows beyond n recompute the last row and store nothing.
// waves_per_eu(1, 2): the scheduler may spend registers (up to 256) on interleaving the independent
// multiplications of a doubling — 42 → 9 s_nop per doubling in the main loop, 0.474 → 0.470 ms at 4 096 rows
// (profiles/r02c_sweeps.txt) — while two wavefronts per SIMD (8 192 rows) still fit.
#ifndef IBFT_ROWS_WAVES_PER_EU_ATTR
#define IBFT_ROWS_WAVES_PER_EU_ATTR __attribute__((amdgpu_waves_per_eu(1, 2)))
#endif
template <int MODE>
__global__ void __launch_bounds__(64 * WAVE_KERNEL_WAVES) IBFT_ROWS_WAVES_PER_EU_ATTR ecrecover_rows_kernel(recover_args a) {
  const uint32_t wave = blockIdx.x * WAVE_KERNEL_WAVES + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (wave * 4u >= a.n) return;  // whole wavefront
  const uint32_t row_raw = wave * 4u + (lane >> 4);
  const bool live = row_raw < a.n;
  const uint32_t row = live ? row_raw : a.n - 1;
  uint32_t want[5];
#pragma unroll
  for (int i = 0; i < 5; i++) want[i] = reinterpret_cast<const uint32_t *>(a.signer20 + 20ull * row)[i];
  const int vi = valset_lookup(a.vtab, a.vslot_mask, want);
  const bool pre = a.pre_flags && a.pre_flags[row] != 0;
  const bool done = a.warm_done && a.warm_done[row] != 0;
  const bool need = live && !done;
  if (need && (lane & 15u) == 0) a.vidx[row] = vi;
  if (!__any((need && !pre && vi >= 0) ? 1 : 0)) return;  // nothing in this wavefront needs the curve
  const u256 r = secp::from_be32(a.sig65 + 65ull * row);
  const u256 s = secp::from_be32(a.sig65 + 65ull * row + 32);
  const uint32_t v = a.sig65[65ull * row + 64];
  u256 z;
  if (MODE == 0) {
    z = secp::from_be32(a.hash32 + 32ull * row);
  } else {
    uint64_t d[4];
    hash_range_dwords(a.payload + a.off[row], a.off[row + 1] - a.off[row], d);
    keccak::digest_to_limbs(d, z.v);
  }
  uint32_t got[5];
  aff Qa;
  __shared__ uint32_t row_tab[WAVE_KERNEL_WAVES][wv::ROW_TAB_SLOTS * 64];  // the window tables: wave-private LDS
  bool ok = wv::recover_pubkey_row(a.gtab, z, r, s, v, a.flags, got, Qa, row_tab[threadIdx.x >> 6]);
  ok = ok && need && !pre && vi >= 0;
#pragma unroll
  for (int i = 0; i < 5; i++) ok = ok && (got[i] == want[i]);
  if ((lane & 15u) == 0 && ok) {
    learn_key(a, vi, Qa);
    atomicOr(reinterpret_cast<unsigned long long *>(a.mask + (row >> 6)), 1ull << (row & 63));
  }
}

// ---- warm path table build -------------------------------------------------------------------
// Lane ↔ (validator, window): a wavefront holds ONE window index for 64 consecutive validators so
// that the doubling loop's trip count is wave-uniform.
// A build pass is three launches on one stream: claim (KNOWN → BUILDING: the set of slots this pass builds is fixed here,
// whatever other contexts learn meanwhile), build, commit (BUILDING → BUILT).  Passes are serialised by the host (the
// device's mutex).
__global__ void qtab_claim_kernel(uint32_t *__restrict__ pub_state, uint32_t n_slots) {
  uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n_slots && __hip_atomic_load(pub_state + v, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == KEY_KNOWN)
    pub_state[v] = KEY_BUILDING;  // (the acquire pairs with learn_key's release: `pub` of this slot is complete)
}
__global__ void __launch_bounds__(64) qtab_build_kernel(const uint32_t *__restrict__ pub, const uint32_t *__restrict__ pub_state,
                                                        uint32_t *__restrict__ qtab, uint32_t n_validators) {
  const uint32_t w = blockIdx.x % QTAB_WINDOWS;
  const uint32_t v = (blockIdx.x / QTAB_WINDOWS) * 64 + threadIdx.x;
  const bool work = v < n_validators && pub_state[v] == KEY_BUILDING;
  if (!__any(work ? 1 : 0)) return;
  aff Q = work ? load_affine(pub + (size_t)GTAB_ENTRY_DWORDS * v) : secp::generator();
  uint32_t *out = qtab + QTAB_DWORDS_PER_VALIDATOR * (work ? v : 0u) + (size_t)GTAB_ENTRY_DWORDS * QTAB_ENTRIES * w;
  qtab_build_window(Q, (int)w, out, work);
}
__global__ void qtab_commit_kernel(uint32_t *__restrict__ pub_state, uint32_t n_validators) {
  uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n_validators && pub_state[v] == KEY_BUILDING) pub_state[v] = KEY_BUILT;
}

// ---- §8f rank 3: IbftMessage wire bytes → verifier columns (wire_dev.h) -----------------------------
// One lane per message: canonical-form walk, Keccak of PayloadNoSig, scatter into the columns.
// The 64 messages of a wavefront are contiguous in the batch: they are brought into LDS with coalesced dword loads
// first (a lane walking ≈200 bytes of HBM one dependent byte load at a time made this kernel 75 µs), and every lane
// walks and hashes its message from there.  A wavefront whose messages exceed the buffer reads HBM directly.
constexpr uint32_t WIRE_LDS_BYTES = 32 * 1024;
__global__ void __launch_bounds__(64) wire_parse_kernel(const uint8_t *__restrict__ wire_bytes, const uint32_t *__restrict__ off,
                                                        uint32_t n, wire::row_info *__restrict__ rows,
                                                        uint8_t *__restrict__ digest32, uint8_t *__restrict__ sig65,
                                                        uint8_t *__restrict__ from20, uint8_t *__restrict__ seal65,
                                                        uint8_t *__restrict__ pre_flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lbuf[WIRE_LDS_BYTES + 16];
  const uint32_t row0 = blockIdx.x * 64u, lane = threadIdx.x;
  const uint32_t cnt = n - row0 < 64u ? n - row0 : 64u;
  const uint32_t b0 = off[row0] & ~15u, b1 = off[row0 + cnt];
  const bool staged = b1 - b0 <= WIRE_LDS_BYTES;  // wave-uniform
  if (staged) {
    cw::stage_bytes(lbuf, wire_bytes + b0, b1 - b0, lane);
    __syncthreads();
  }
  const uint32_t row = row0 + lane;
  if (row >= n) return;
  const uint32_t o0 = off[row], o1 = off[row + 1];
  const uint8_t *m = staged ? lbuf + (o0 - b0) : wire_bytes + o0;
  wire::process_row(m, o1 - o0, rows + row, digest32 + 32ull * row, sig65 + 65ull * row, from20 + 20ull * row,
                    seal65 + 65ull * row, pre_flags + row);
}
// After the sender pass: make the COMMIT seals found by wire_parse_kernel the resident seal batch
// (hash column ← proposal hash, signature column ← committed seal, From stays).  Rows that are not
// canonical COMMIT messages with a 32-byte hash and a 65-byte seal are pre-flagged.
__global__ void wire_stage_seals_kernel(const wire::row_info *__restrict__ rows, const uint8_t *__restrict__ seal65,
                                        uint32_t n, uint8_t *__restrict__ hash32, uint8_t *__restrict__ sig65,
                                        uint8_t *__restrict__ pre_flags) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const wire::row_info &ri = rows[row];
  const bool ok = ri.status == wire::STATUS_OK && ri.payload_kind == wire::KIND_COMMIT && ri.type == 2 &&
                  ri.hash_len == 32 && ri.seal_len == 65 && ri.from_len == 20;
  for (int i = 0; i < 32; i++) hash32[32ull * row + i] = ri.proposal_hash[i];
  for (int i = 0; i < 65; i++) sig65[65ull * row + i] = seal65[65ull * row + i];
  pre_flags[row] = ok ? 0 : 1;
}

// ---- a message set in one pass: both signatures of every message in ONE verdict launch ------------------------
// A PREPARE / COMMIT message is checked three times by the reference, at different moments: IsValidValidator when it
// arrives (core/ibft.go:1128), IsValidProposalHash and — COMMIT only — IsValidCommittedSeal when the view is
// handled (:856-862, :932-944).  All three are pure functions of the message bytes and of the validator set, so a
// whole set can be judged at once: the envelope signatures are rows [0, n) of the batch (their digests, Keccak of
// PayloadNoSig, come from payload_digest_kernel), the committed seals are rows [half, half + n) with the hash each
// message carries as digest, half = ⌈n/64⌉·64 — 2n signatures in one launch fill the chip where n alone left
// every SIMD with a single wavefront (n = 4 096: 8 192 rows, two wavefronts per SIMD).
__global__ void payload_digest_kernel(const uint8_t *__restrict__ payload, const uint32_t *__restrict__ off, uint32_t n,
                                      uint8_t *__restrict__ digest32) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = row < n;
  const uint32_t o0 = live ? off[row] : 0u, o1 = live ? off[row + 1] : 0u;
  uint64_t d[4];
  hash_range_dwords(payload + o0, o1 - o0, d);
  if (live) {
    uint4 *o = reinterpret_cast<uint4 *>(digest32 + 32ull * row);
    o[0] = make_uint4((uint32_t)d[0], (uint32_t)(d[0] >> 32), (uint32_t)d[1], (uint32_t)(d[1] >> 32));
    o[1] = make_uint4((uint32_t)d[2], (uint32_t)(d[2] >> 32), (uint32_t)d[3], (uint32_t)(d[3] >> 32));
  }
}
// The same set judged straight from the transport's bytes: wire_parse_kernel has filled the sender rows [0, n)
// (digest of PayloadNoSig, Signature, From) and the seal column; this kernel lays the second group of verdict rows
// [half, half + n) — the hash each message carries as digest, its committed seal as signature, From as signer — and
// decides, per row, everything about the closure that needs no arithmetic: valid_pre ≠ 0 unless the message is a
// canonical COMMIT (type 2, CommitMessage, 65-byte seal) of the view (height, round) with a 20-byte From — a canonical
// PREPARE of the view (type 1, PrepareMessage) is flagged in is_prepare instead: its closure is the hash compare alone (ExtractPrepareHash / ExtractCommitHash / ExtractCommittedSeal return nil
// otherwise, messages/helpers.go).  A PREPARE row's seal row is a zero signature: its verdict is forced below.
__global__ void wire_set_stage_kernel(const wire::row_info *__restrict__ rows, const uint8_t *__restrict__ seal65, uint32_t n,
                                      uint32_t half, uint64_t height, uint64_t round, uint8_t *__restrict__ hash32,
                                      uint8_t *__restrict__ hash_len, uint8_t *__restrict__ sig65,
                                      uint8_t *__restrict__ signer20, uint8_t *__restrict__ pre_flags,
                                      uint8_t *__restrict__ is_prepare, uint8_t *__restrict__ row_class,
                                      uint8_t *__restrict__ host_row_class) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const wire::row_info &ri = rows[row];
  const bool here = ri.status == wire::STATUS_OK && ri.has_view && ri.height == height && ri.round == round && ri.from_len == 20;
  const bool prepare = here && ri.type == 1 && ri.payload_kind == wire::KIND_PREPARE;
  const bool commit = here && ri.type == 2 && ri.payload_kind == wire::KIND_COMMIT && ri.seal_len == 65;
  const uint32_t dst = half + row;
  for (int i = 0; i < 32; i++) hash32[32ull * dst + i] = ri.proposal_hash[i];
  for (int i = 0; i < 65; i++) sig65[65ull * dst + i] = commit ? seal65[65ull * row + i] : 0;
  for (int i = 0; i < 20; i++) signer20[20ull * dst + i] = ri.from[i];
  hash_len[row] = ri.hash_len;
  pre_flags[dst] = commit ? 0 : 1;  // the verdict launch skips every seal row that is not a COMMIT's (whole wavefronts exit)
  is_prepare[row] = prepare ? 1 : 0;
  // what the host needs to route the row, in one byte (include/ibftgpu.h: IBFT_WIRE_CLASS_*): bit 0 = not judged here
  // (stock route), bit 1 = a PREPARE / COMMIT of the asked view, i.e. its valid bit IS the closure's verdict, bits 4.. = type
  const bool in_view = ri.status == wire::STATUS_OK && ri.has_view && ri.height == height && ri.round == round;
  const uint8_t cls = (uint8_t)((ri.status == wire::STATUS_OK ? 0 : 1) | ((in_view && (ri.type == 1 || ri.type == 2)) ? 2 : 0) |
                                ((ri.type & 15u) << 4));
  row_class[row] = cls;
  if (host_row_class) host_row_class[row] = cls;
}

// ---- §8f rank 2 from the transport's bytes: the certificate tree (wire_dev.h, second half) -------------------------
// ibft_verify_certificates_wire: PREPREPARE / ROUND_CHANGE messages carry messages (RoundChangeCertificate,
// PreparedCertificate: core/ibft.go:470-551, 683-788 verify every one of them).  Level by level, rows [lo, hi) of one
// level at a time:
//   cert_parse_kernel       a lane per message: the DEEP walk of its own fields, sender columns, digest of a short message
//                           that has nothing below it; where its nested messages lie goes to cert_span
//   cert_walk_kernel<false> a wavefront per message with a certificate: counts (and checks) the nested messages
//   cert_scan_kernel        exclusive scan of the counts: every row's children become a contiguous row range of the next
//   (cert_scan_tiles/_offsets/_apply for a long level)  level, in order, and the rows whose digest is deferred are listed;
//                           the totals go to the host, which sizes the next level's launches
//   cert_walk_kernel<true>  the same walk again, writing the child rows
// then, bottom-up, cert_propagate_kernel (a message with a non-canonical message below it is not canonical either),
// cert_digest_wave_kernel (the deferred digests — messages that carry certificates, long messages — and the proposal hashes:
// one wavefront per sponge, wave_sponge), cert_finish_kernel (final pre-flags and class bits), the verdict launch over all
// rows — or, with the digests on a side stream, over the rows that are not deferred, and a second small one over the deferred
// rows (cert_carrier_stage_kernel, cert_scatter_kernel) —, cert_compare_kernel (hash bits).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  for (int o = 32; o; o >>= 1) {
    const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
    v = w < v ? w : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  for (int o = 32; o; o >>= 1) {
    const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}
__global__ void __launch_bounds__(64) cert_parse_kernel(const uint8_t *__restrict__ wire_bytes, wire::node_info *__restrict__ nodes,
                                                        uint32_t lo, uint32_t hi, wire::row_info *__restrict__ rows,
                                                        uint2 *__restrict__ cert_span, uint8_t *__restrict__ digest32,
                                                        uint8_t *__restrict__ sig65, uint8_t *__restrict__ from20,
                                                        uint8_t *__restrict__ pre_flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lbuf[WIRE_LDS_BYTES + 16];
  const uint32_t lane = threadIdx.x, row = lo + blockIdx.x * 64u + lane;
  const bool live = row < hi;
  wire::node_info nd{};
  if (live) nd = nodes[row];
  // the 64 messages of a wavefront are usually neighbours in the buffer (children of one certificate): staged in LDS
  // with coalesced dword loads when they span little enough, like wire_parse_kernel
  const uint32_t b0 = wave_min_u32(live ? nd.off : 0xFFFFFFFFu) & ~15u, b1 = wave_max_u32(live ? nd.off + nd.len : 0u);
  const bool staged = b1 >= b0 && b1 - b0 <= WIRE_LDS_BYTES;  // wave-uniform
  if (staged) {
    cw::stage_bytes(lbuf, wire_bytes + b0, b1 - b0, lane);
    __syncthreads();
  }
  if (!live) return;
  const uint8_t *m = staged ? lbuf + (nd.off - b0) : wire_bytes + nd.off;
  uint32_t span[2];
  wire::process_tree_row(m, nd.len, rows + row, &nd, span, digest32 + 32ull * row, sig65 + 65ull * row, from20 + 20ull * row,
                         pre_flags + row);
  nodes[row] = nd;
  cert_span[row] = make_uint2(span[0], span[1]);
}
// A wavefront per message with a certificate lists (FILL) or counts and checks (!FILL) its nested messages: cw::walk_certificate.
template <bool FILL>
__global__ void __launch_bounds__(64) cert_walk_kernel(const uint8_t *__restrict__ wire_bytes, wire::node_info *__restrict__ nodes,
                                                       wire::row_info *__restrict__ rows, const uint2 *__restrict__ cert_span,
                                                       uint32_t lo, uint32_t hi, uint32_t *__restrict__ child_count) {
  __shared__ __attribute__((aligned(16))) uint8_t win[cw::CERT_WIN_BYTES + 16];
  const uint32_t row = lo + blockIdx.x, lane = threadIdx.x;
  if (row >= hi) return;
  const wire::node_info nd = nodes[row];
  const bool has = rows[row].status == wire::STATUS_OK && (nd.flags & wire::TREE_HAS_CERT);
  const uint32_t deferred = rows[row].status == wire::STATUS_OK && wire::tree_deferred(nd) ? 0x80000000u : 0u;
  if (!has || (FILL && nd.n_children == 0)) {
    if (!FILL && lane == 0) child_count[row - lo] = deferred;
    return;
  }
  const bool pc = rows[row].payload_kind == wire::KIND_ROUND_CHANGE;
  const uint2 span = cert_span[row];
  const uint32_t pos = nd.off + span.x;
  bool ok = true;
  uint32_t count = cw::walk_certificate(wire_bytes, pos, pos + span.y, pc, win, lane, ok, [&](uint32_t ordinal, uint32_t off, uint32_t len, uint8_t role) {
    if (!FILL) return;
    wire::node_info c{};
    c.off = off;
    c.len = len;
    c.parent = row;
    c.ordinal = ordinal;
    c.level = (uint8_t)(nd.level + 1);
    c.role = role;
    nodes[nd.first_child + ordinal] = c;
  });
  if (!FILL && lane == 0) {
    if (!ok) {  // a malformed wrapper: the whole message goes the stock route, nothing below it is listed
      rows[row].status = wire::STATUS_NEEDS_HOST;
      count = 0;
    }
    nodes[row].n_children = count;
    child_count[row - lo] = count | (ok ? deferred : 0u);
  }
}
// first_child of rows [lo, hi) = base + exclusive prefix sum of their child counts; the sum → total[0] (device and host).
// The rows whose digest is deferred (wire::tree_deferred: they carry a certificate, or are long) are listed the same way:
// deferred_rows[slot_base + rank] = row, their count → total[1].  child_count[i] bit 31 = row lo + i is deferred.
__global__ void __launch_bounds__(1024) cert_scan_kernel(const uint32_t *__restrict__ child_count, wire::node_info *__restrict__ nodes,
                                                         uint32_t lo, uint32_t hi, uint32_t base, uint32_t slot_base,
                                                         uint32_t *__restrict__ deferred_rows, uint32_t *__restrict__ total_dev,
                                                         uint32_t *__restrict__ total_host) {
  __shared__ uint32_t part[1024], part2[1024];
  const uint32_t n = hi - lo, t = threadIdx.x, per = (n + 1023u) / 1024u;
  const uint32_t b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
  uint32_t sum = 0, sum2 = 0;
  for (uint32_t i = b; i < e; i++) {
    const uint32_t c = child_count[i];
    sum += c & 0x7FFFFFFFu;
    sum2 += c >> 31;
  }
  part[t] = sum;
  part2[t] = sum2;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1) {
    const uint32_t v = t >= o ? part[t - o] : 0u, v2 = t >= o ? part2[t - o] : 0u;
    __syncthreads();
    part[t] += v;
    part2[t] += v2;
    __syncthreads();
  }
  uint32_t run = base + part[t] - sum, run2 = slot_base + part2[t] - sum2;
  for (uint32_t i = b; i < e; i++) {
    const uint32_t c = child_count[i];
    nodes[lo + i].first_child = run;
    run += c & 0x7FFFFFFFu;
    if (c >> 31) deferred_rows[run2++] = lo + i;
  }
  if (t == 1023u) {
    total_dev[0] = part[1023];
    total_dev[1] = part2[1023];
    if (total_host) {
      total_host[0] = part[1023];
      total_host[1] = part2[1023];
    }
  }
}
// The same scan for a long level (the single workgroup above walks n/1024 rows per thread: 0.8 ms at 467 k rows): (A) per-tile
// sums of 1 024 rows, (B) one workgroup scans the tile sums (≤ 1 024 tiles per pass of its loop) and delivers the totals, (C) every
// tile scans its own rows and adds its offset.
__global__ void __launch_bounds__(1024) cert_scan_tiles_kernel(const uint32_t *__restrict__ child_count, uint32_t n, uint2 *__restrict__ tile_sum) {
  __shared__ uint32_t a[1024], b[1024];
  const uint32_t t = threadIdx.x, i = blockIdx.x * 1024u + t;
  const uint32_t c = i < n ? child_count[i] : 0u;
  a[t] = c & 0x7FFFFFFFu;
  b[t] = c >> 31;
  __syncthreads();
  for (uint32_t o = 512u; o; o >>= 1) {
    if (t < o) {
      a[t] += a[t + o];
      b[t] += b[t + o];
    }
    __syncthreads();
  }
  if (t == 0) tile_sum[blockIdx.x] = make_uint2(a[0], b[0]);
}
__global__ void __launch_bounds__(1024) cert_scan_offsets_kernel(uint2 *__restrict__ tile_sum, uint32_t tiles, uint32_t base, uint32_t slot_base,
                                                                 uint32_t *__restrict__ total_dev, uint32_t *__restrict__ total_host) {
  __shared__ uint32_t a[1024], b[1024];
  const uint32_t t = threadIdx.x;
  uint32_t run = 0, run2 = 0;  // sums of the passes before this one (the same in every thread)
  for (uint32_t t0 = 0; t0 < tiles; t0 += 1024u) {
    const uint2 v = t0 + t < tiles ? tile_sum[t0 + t] : make_uint2(0, 0);
    a[t] = v.x;
    b[t] = v.y;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
      const uint32_t x = t >= o ? a[t - o] : 0u, y = t >= o ? b[t - o] : 0u;
      __syncthreads();
      a[t] += x;
      b[t] += y;
      __syncthreads();
    }
    if (t0 + t < tiles) tile_sum[t0 + t] = make_uint2(base + run + a[t] - v.x, slot_base + run2 + b[t] - v.y);  // exclusive, with the bases
    run += a[1023];
    run2 += b[1023];
    __syncthreads();
  }
  if (t == 0) {
    total_dev[0] = run;
    total_dev[1] = run2;
    if (total_host) {
      total_host[0] = run;
      total_host[1] = run2;
    }
  }
}
__global__ void __launch_bounds__(1024) cert_scan_apply_kernel(const uint32_t *__restrict__ child_count, wire::node_info *__restrict__ nodes,
                                                               uint32_t lo, uint32_t n, const uint2 *__restrict__ tile_off,
                                                               uint32_t *__restrict__ deferred_rows) {
  __shared__ uint32_t a[1024], b[1024];
  const uint32_t t = threadIdx.x, i = blockIdx.x * 1024u + t;
  const uint32_t c = i < n ? child_count[i] : 0u, cnt = c & 0x7FFFFFFFu, def = c >> 31;
  a[t] = cnt;
  b[t] = def;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1) {
    const uint32_t x = t >= o ? a[t - o] : 0u, y = t >= o ? b[t - o] : 0u;
    __syncthreads();
    a[t] += x;
    b[t] += y;
    __syncthreads();
  }
  if (i >= n) return;
  const uint2 off = tile_off[blockIdx.x];
  nodes[lo + i].first_child = off.x + a[t] - cnt;
  if (def) deferred_rows[off.y + b[t] - 1u] = lo + i;
}
__global__ void cert_propagate_kernel(const wire::node_info *__restrict__ nodes, wire::row_info *__restrict__ rows, uint32_t lo,
                                      uint32_t hi) {
  const uint32_t row = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= hi) return;
  const uint32_t parent = nodes[row].parent;
  if (rows[row].status != wire::STATUS_OK && parent != wire::NO_PARENT) rows[parent].status = wire::STATUS_NEEDS_HOST;
}
// (Keccak by one wavefront: cert_wave_dev.h, cw::sponge_message.)
// One job per wavefront: job 2k hashes PayloadNoSig of deferred row k → digest32[row]; job 2k + 1 hashes the Proposal that row
// carries, keccak(rawProposal ‖ BE64(round)) → prop_digest32[row] (exits at once when there is none).  The message is
// piece A = [a0, a0 + na) followed by piece B = [b0, b0 + nb) of the buffer, then `tail` (≤ 8 bytes, by value).
__global__ void __launch_bounds__(64) cert_digest_wave_kernel(const uint8_t *__restrict__ wire_bytes, const wire::node_info *__restrict__ nodes,
                                                              const wire::row_info *__restrict__ rows,
                                                              const uint32_t *__restrict__ deferred_rows, uint32_t region,
                                                              uint8_t *__restrict__ digest32, uint8_t *__restrict__ prop_digest32) {
  __shared__ uint64_t A[32], B[32];
  const uint32_t row = deferred_rows[blockIdx.x >> 1], lane = threadIdx.x;
  // region ≠ 0: the deferred rows form a batch of their own at rows [region, region + deferred) of the columns (cert_carrier_stage_kernel)
  const uint32_t digest_row = region ? region + (blockIdx.x >> 1) : row;
  const bool proposal_job = blockIdx.x & 1u;
  const wire::node_info nd = nodes[row];
  if (rows[row].status != wire::STATUS_OK) return;  // block-uniform: not judged here
  const uint8_t *m;
  uint32_t cut0, gap, total;
  uint64_t tail = 0;
  if (!proposal_job) {
    if (nd.len > wire::TREE_DIGEST_MAX_BYTES) return;
    m = wire_bytes + nd.off;
    cut0 = nd.cut0;
    gap = nd.cut1 - nd.cut0;
    total = nd.len - gap;
  } else {
    if (!(nd.flags & wire::TREE_HAS_PROPOSAL) || nd.raw_len > wire::TREE_DIGEST_MAX_BYTES) return;
    m = wire_bytes + nd.raw_off;
    cut0 = nd.raw_len;  // everything before the "cut" is the raw proposal; behind it come the 8 bytes of the round
    gap = 0;
    total = nd.raw_len + 8u;
    for (int k = 0; k < 8; k++) tail |= (uint64_t)((nd.proposal_round >> (8 * (7 - k))) & 0xFFu) << (8 * k);  // BE64 as the bytes lie
  }
  const uint64_t word = cw::sponge_message(m, cut0, gap, total, tail, proposal_job, lane, A, B);
  if (lane < 4u) *reinterpret_cast<uint64_t *>((proposal_job ? prop_digest32 + 32ull * row : digest32 + 32ull * digest_row) + 8u * lane) = word;
}
// After the deferred digests: every row's final pre-flag and class bits, and the hash of the Proposal it carries (a lane per row;
// proposals are short next to the messages that carry certificates)
// two_launches: the deferred rows are judged as a batch of their own (cert_carrier_stage_kernel) while the verdict launch over the
// other rows is already running — their pre-flags in the tree's own rows must keep saying "verdict 0".
__global__ void __launch_bounds__(64) cert_finish_kernel(const uint8_t *__restrict__ wire_bytes, wire::node_info *__restrict__ nodes,
                                                         const wire::row_info *__restrict__ rows, uint32_t n,
                                                         uint8_t *__restrict__ digest32, uint8_t *__restrict__ prop_digest32,
                                                         uint8_t *__restrict__ pre_flags, uint32_t two_launches) {
  const uint32_t row = blockIdx.x * 64u + threadIdx.x;
  if (row >= n) return;
  wire::node_info nd = nodes[row];
  // almost every row of a big tree is a PREPARE: not deferred, no Proposal — nothing to do (the proposal-digest column was zeroed)
  if (!wire::tree_deferred(nd) && !(nd.flags & wire::TREE_HAS_PROPOSAL)) return;
  const uint8_t before = nd.flags;
  uint8_t unused = 0;
  wire::tree_digest_row(wire_bytes, rows + row, &nd, digest32 + 32ull * row, prop_digest32 + 32ull * row, two_launches ? &unused : pre_flags + row,
                        false);  // a deferred row's digests are there already
  if (nd.flags != before) nodes[row].flags = nd.flags;
}
// The batch of the deferred rows: signature, From and pre-flag of deferred row k → row region + k of the columns (its digest is
// written there by cert_digest_wave_kernel); after the second verdict launch cert_scatter_kernel brings the bits home.
__global__ void cert_carrier_stage_kernel(const wire::node_info *__restrict__ nodes, const wire::row_info *__restrict__ rows,
                                          const uint32_t *__restrict__ deferred_rows, uint32_t n_deferred, uint32_t region,
                                          uint8_t *__restrict__ sig65, uint8_t *__restrict__ from20, uint8_t *__restrict__ pre_flags) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_deferred) return;
  const uint32_t row = deferred_rows[k], dst = region + k;
  const wire::row_info &ri = rows[row];
  const bool judged = ri.status == wire::STATUS_OK && nodes[row].len <= wire::TREE_DIGEST_MAX_BYTES;
  for (int i = 0; i < 65; i++) sig65[65ull * dst + i] = sig65[65ull * row + i];
  for (int i = 0; i < 20; i++) from20[20ull * dst + i] = from20[20ull * row + i];
  pre_flags[dst] = (uint8_t)((judged ? 0 : 1) | (ri.sig_len == 65 && ri.from_len == 20 ? 0 : 2));
}
__global__ void cert_scatter_kernel(const uint32_t *__restrict__ deferred_rows, uint32_t n_deferred, uint32_t region, uint64_t *__restrict__ mask) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_deferred) return;
  const uint32_t src = region + k, row = deferred_rows[k];
  if ((mask[src >> 6] >> (src & 63u)) & 1ull) atomicOr((unsigned long long *)(mask + (row >> 6)), 1ull << (row & 63u));
}
// hash / self bits and the routing byte of every row (wire::tree_compare_row), one verdict word per wavefront
__global__ void __launch_bounds__(256) cert_compare_kernel(const wire::node_info *__restrict__ nodes, const wire::row_info *__restrict__ rows,
                                                           const uint8_t *__restrict__ prop_digest32, uint32_t n,
                                                           uint64_t *__restrict__ hash_mask, uint64_t *__restrict__ self_mask,
                                                           uint8_t *__restrict__ row_class) {
  const uint32_t row = blockIdx.x * 256u + threadIdx.x;
  bool hash_bit = false, self_bit = false;
  if (row < n) {
    uint8_t cls;
    wire::tree_compare_row(nodes, rows, prop_digest32, row, hash_bit, self_bit, cls);
    row_class[row] = cls;
  }
  const uint64_t hb = __ballot(hash_bit), sb = __ballot(self_bit);
  if ((threadIdx.x & 63u) == 0 && row < n) {
    hash_mask[row >> 6] = hb;
    self_mask[row >> 6] = sb;
  }
}

// ---- a8: weighted quorum tally ------------------------------------------------------------
// HasQuorum (core/validator_manager.go:77-96): Σ power over the DISTINCT member senders of the valid rows
// ≥ ⌊2·total/3⌋+1.  Several workgroups (4 096 rows each) walk the verdict words: a validator's power is
// counted once — a bitmap in HBM updated with device-scope atomicOr gives the Go map's set semantics across
// workgroups —, lanes reduce with wave shuffles, waves through LDS, workgroups with one device-scope
// atomicAdd per 32-bit piece; the workgroup that draws the last ticket recombines the pieces (carries),
// compares with the quorum, delivers the result (also into mapped host memory) and leaves bitmap,
// accumulators and ticket zeroed for the next launch.
//   PW = 64-bit words per voting power: 1 (u64, ibft_set_validators) or 4 (256-bit, ibft_set_validators_u256:
//   GetVotingPowers returns *big.Int, validator_manager.go:17-31).  Sums are TALLY_SUM_WORDS wide.
// out[] (u64 words): [0..1] low 128 bits of the power, [2] valid_rows | distinct << 32, [3] has_quorum,
//   [4] keys learned | a learned validator (written by the recover kernels), [5..9] the power in full,
//   [10] valid rows sent by the proposer (prop_on: any such row voids the quorum, validator_manager.go:114-121),
//   [16..16+2·PW) the raw 32-bit piece sums (what the multi-GPU exchange adds across ranks).
constexpr int TALLY_THREADS = 1024;
constexpr int TALLY_ROWS_PER_BLOCK = 4096;
constexpr int TALLY_SUM_WORDS = 5;       // 256-bit powers × up to 2^20 validators < 2^276
constexpr int TALLY_MAX_PIECES = 8;      // 2·PW 32-bit pieces
constexpr int TALLY_OUT_WIDE = 5;        // out[5..10)
constexpr int TALLY_OUT_PIECES = 16;     // out[16..24)
constexpr int TALLY_OUT_WORDS = 24;
constexpr int TALLY_ACC_WORDS = TALLY_MAX_PIECES + 4;  // pieces, valid, distinct, ticket, rows sent by the proposer
constexpr int TALLY_OUT_PROPOSER_ROWS = 10;            // out[10]: valid rows whose sender is the proposer (HasPrepareQuorum)
constexpr int32_t VIDX_PROPOSER_OUTSIDER = -2;         // lookup_kernel: sender == proposer and the proposer is no validator

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Σ piece_k·2^(32k) → TALLY_SUM_WORDS little-endian u64 words (pieces are sums of 32-bit values, < 2^60)
__device__ __forceinline__ void pieces_to_words(const uint64_t *piece, int n_pieces, uint64_t w[TALLY_SUM_WORDS]) {
  uint64_t carry = 0;  // running value >> 32 at each 32-bit position
  uint32_t half[2 * TALLY_SUM_WORDS];
#pragma unroll
  for (int k = 0; k < 2 * TALLY_SUM_WORDS; k++) {
    const uint64_t t = carry + (k < n_pieces ? piece[k] : 0ull);  // < 2^61
    half[k] = (uint32_t)t;
    carry = t >> 32;
  }
#pragma unroll
  for (int i = 0; i < TALLY_SUM_WORDS; i++) w[i] = (uint64_t)half[2 * i] | ((uint64_t)half[2 * i + 1] << 32);
}
__device__ __forceinline__ bool words_ge(const uint64_t a[TALLY_SUM_WORDS], const uint64_t *b) {
#pragma unroll
  for (int i = TALLY_SUM_WORDS - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}

// After the verdict launch of a message set: word w of the work mask holds the envelope verdicts of rows 64w.., word
// half/64 + w the seal verdicts of the same messages.  a1 is evaluated here (hash_eq_kernel's compare).  Delivered: sender
// words (IsValidValidator), valid words (a1 ∧ a2 — handlePrepare's / handleCommit's closure; a1 alone without seals);
// the tally goes on with sender ∧ valid, which answers hasQuorumByMsgType for the set.  This runs INSIDE the tally
// kernel (a wavefront per verdict word, lane l judging row 64w + l): as a kernel of its own it cost 4.4 µs plus a
// dependency gap per set.
struct set_args {
  const uint8_t *hash32;    // n × 32: the proposal hash each message carries
  const uint8_t *hash_len;  // n
  const uint64_t *H4;       // keccak256(raw proposal ‖ BE64(round))
  const uint8_t *sender_pre, *valid_pre;  // n each or null: rows the host rejected before any crypto
  const uint8_t *no_seal;   // n or null: rows (PREPAREs of a mixed wire batch) whose valid bit is a1 alone
  uint32_t n, half_words;   // half_words = 0: no seals (PREPARE)
  uint64_t *sender_out, *valid_out;    // device copies (⌈n/64⌉ words each)
  uint64_t *host_sender, *host_valid;  // mapped pinned host memory or null
};
// Per-row part (loads only, no stores: the caller issues several rows' loads before it waits for any of them)
struct set_row_flags {
  bool a1, spre, nos;
};
__device__ __forceinline__ set_row_flags set_row(const set_args &a, uint32_t row) {
  set_row_flags f{false, false, false};
  if (row < a.n) {
    f.spre = a.sender_pre && a.sender_pre[row] != 0;
    f.nos = a.no_seal && a.no_seal[row] != 0;
    const uint4 *p = reinterpret_cast<const uint4 *>(a.hash32 + 32ull * row);
    const uint4 x = p[0], y = p[1];
    const uint32_t *h = reinterpret_cast<const uint32_t *>(a.H4);
    const uint32_t diff = (x.x ^ h[0]) | (x.y ^ h[1]) | (x.z ^ h[2]) | (x.w ^ h[3]) | (y.x ^ h[4]) | (y.y ^ h[5]) |
                          (y.z ^ h[6]) | (y.w ^ h[7]);
    // valid_pre marks a dead seal side; a row flagged no_seal (a PREPARE among raw messages) has no seal side to be dead
    const bool dead = a.valid_pre && a.valid_pre[row] != 0 && !f.nos;
    f.a1 = diff == 0 && a.hash_len[row] == 32 && !dead;
  }
  return f;
}
// Per-word part, called by a whole wavefront with its lanes' flags and the two verdict words already loaded; returns
// sender ∧ valid on every lane, lane 0 delivers the two words
__device__ __forceinline__ uint64_t set_finish_word(const set_args &a, uint64_t *work_mask, uint32_t w, uint32_t lane,
                                                    const set_row_flags &f, uint64_t sender_word, uint64_t seal_word) {
  const uint64_t bal = __ballot(f.a1), sbad = __ballot(f.spre), skip_seal = __ballot(f.nos);
  const uint32_t first = w * 64u;
  const uint32_t left = a.n > first ? a.n - first : 0u;
  const uint64_t tail = left >= 64 ? ~0ull : (left ? (~0ull >> (64 - left)) : 0ull);
  const uint64_t S = sender_word & tail & ~sbad;
  uint64_t V = bal;
  if (a.half_words) V &= seal_word | skip_seal;
  if (lane == 0) {
    if (a.half_words) work_mask[a.half_words + w] = 0;
    a.sender_out[w] = S;
    a.valid_out[w] = V;
    if (a.host_sender) {
      a.host_sender[w] = S;
      a.host_valid[w] = V;
    }
  }
  return S & V;
}

struct tally_args {
  uint64_t *work_mask;      // verdict words accumulated by the verdict kernels; consumed (zeroed) here
  uint64_t *mask;           // verdict words as fetch / export / exchange read them
  const int32_t *vidx;      // n: validator index of the row's sender (−1 = not a member)
  const uint32_t *vpower32; // n_validators × 2·PW little-endian 32-bit pieces
  uint32_t n, n_validators;
  uint32_t *seen;           // ⌈n_validators/32⌉ words, zero between launches
  uint64_t *acc;            // TALLY_ACC_WORDS, zero between launches
  const uint64_t *quorum;   // TALLY_SUM_WORDS
  uint64_t *out;            // TALLY_OUT_WORDS
  uint64_t *host_mask, *host_tally;  // mapped pinned host memory (or null): no device-to-host copy commands
  uint32_t lds_bitmap;      // the launch carries ⌈n_validators/32⌉ words of dynamic LDS for the workgroup's bitmap
  const uint64_t *learned_src;  // or null: {keys learned, a learned slot} of the device's key cache, passed on to the host
  uint32_t *seen_out;       // or null: the launch's distinct-sender bitmap is left here (⌈n_validators/32⌉ words) — what a
                            // sharded batch exchanges, because the SET of senders merges across shards and sums do not
  uint32_t set_on;          // a message set: the verdict words are combined here first (set_combine_word)
  // HasPrepareQuorum (/root/reference/core/validator_manager.go:99-127): the proposer's address joins the sender set
  // (prop_vidx = its validator index, −1 when it is no validator: it then adds nothing, :88-92) and a valid row sent BY
  // the proposer voids the quorum (:114-121).  prop_seat = 0: a rank of a sharded batch only counts the proposer's rows —
  // the seat is added where the shards' bitmaps are merged (exchange_unpack_kernel).
  uint32_t prop_on, prop_seat;
  int32_t prop_vidx;
  set_args set;
};

// MULTI = false: ONE workgroup (n ≤ TALLY_ROWS_PER_BLOCK, the latency-critical sizes): the distinct-sender set is
// a bitmap in LDS (dynamic: ⌈n_validators/32⌉ words) and nothing leaves the workgroup before the result — no
// global atomics, no ticket.  MULTI = true: the bitmap lives in HBM (device-scope atomicOr), workgroups add their
// 32-bit pieces with device-scope atomicAdd and the one that draws the last ticket finishes.
template <int PW, bool MULTI>
__global__ void __launch_bounds__(TALLY_THREADS) tally_kernel(tally_args a) {
  constexpr int NP = 2 * PW;
  constexpr int RPT = TALLY_ROWS_PER_BLOCK / TALLY_THREADS;
  extern __shared__ uint32_t lseen[];  // ⌈n_validators/32⌉ words when a.lds_bitmap
  __shared__ uint64_t wds[TALLY_ROWS_PER_BLOCK / 64];
  __shared__ uint64_t part[NP + 2][TALLY_THREADS / 64];
  __shared__ uint32_t last_flag;
  const uint32_t tid = threadIdx.x;
  const uint32_t row0 = blockIdx.x * (uint32_t)TALLY_ROWS_PER_BLOCK;
  const uint32_t total_words = (a.n + 63) / 64;
  // Two dependent memory round trips instead of four: the validator index of every row of this thread is loaded
  // together with the verdict words, the voting powers right behind them — both speculatively (rows without a
  // verdict bit and repeated senders simply do not use what was loaded).
  int vi[RPT];
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const uint32_t row = row0 + (uint32_t)j * TALLY_THREADS + tid;
    vi[j] = row < a.n ? a.vidx[row] : -1;
  }
  // The verdict kernels accumulate into work_mask (atomicOr / ballot words).  The tally CONSUMES it: the words
  // move to `mask`, to host_mask when given, and work_mask is left zeroed for the next launch.
  if (a.set_on) {
    // every wavefront combines its share of the workgroup's verdict words; all loads first, then the ballots and stores
    constexpr int WPW = (TALLY_ROWS_PER_BLOCK / 64) / (TALLY_THREADS / 64);  // words per wavefront
    const uint32_t wv = tid >> 6, ln = tid & 63u;
    set_row_flags fl[WPW];
    uint64_t sw[WPW], lw[WPW];
#pragma unroll
    for (int q = 0; q < WPW; q++) {
      const uint32_t wi = row0 / 64 + wv + (uint32_t)q * (TALLY_THREADS / 64);
      const bool in = wi < total_words;
      fl[q] = set_row(a.set, in ? wi * 64u + ln : 0xFFFFFFFFu);
      sw[q] = in ? a.work_mask[wi] : 0ull;
      lw[q] = (in && a.set.half_words) ? a.work_mask[a.set.half_words + wi] : 0ull;
    }
#pragma unroll
    for (int q = 0; q < WPW; q++) {
      const uint32_t k = wv + (uint32_t)q * (TALLY_THREADS / 64);
      const uint32_t wi = row0 / 64 + k;
      uint64_t w = 0;
      if (wi < total_words) {  // wave-uniform
        w = set_finish_word(a.set, a.work_mask, wi, ln, fl[q], sw[q], lw[q]);
        if (ln == 0) {
          a.work_mask[wi] = 0;
          a.mask[wi] = w;
          if (a.host_mask) a.host_mask[wi] = w;
        }
      }
      if (ln == 0) wds[k] = w;
    }
  } else if (tid < TALLY_ROWS_PER_BLOCK / 64) {
    const uint32_t wi = row0 / 64 + tid;
    uint64_t w = 0;
    if (wi < total_words) {
      w = a.work_mask[wi];
      a.work_mask[wi] = 0;
      a.mask[wi] = w;
      if (a.host_mask) a.host_mask[wi] = w;
    }
    wds[tid] = w;
  }
  // the workgroup's own distinct-sender bitmap is always in LDS (a.lds_bitmap = 0 only when a huge validator set
  // does not fit: MULTI then sends every row's bit to the HBM bitmap directly)
  const uint32_t seen_words = (a.n_validators + 31) / 32;
  if (a.lds_bitmap)
    for (uint32_t i = tid; i < seen_words; i += TALLY_THREADS) lseen[i] = 0;
  uint32_t pw[RPT][NP];
#pragma unroll
  for (int j = 0; j < RPT; j++)
#pragma unroll
    for (int k = 0; k < NP; k++) pw[j][k] = vi[j] >= 0 ? a.vpower32[(size_t)vi[j] * NP + k] : 0u;
  __syncthreads();
  bool first[RPT];
  uint64_t valid = 0, distinct = 0;
  uint32_t by_proposer = 0;
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const uint32_t lr = (uint32_t)j * TALLY_THREADS + tid;
    const bool bit = row0 + lr < a.n && ((wds[lr >> 6] >> (lr & 63)) & 1ull);
    valid += bit;
    first[j] = false;
    if (a.prop_on && bit && ((a.prop_vidx >= 0 && vi[j] == a.prop_vidx) || vi[j] == VIDX_PROPOSER_OUTSIDER)) {
      // HasPrepareQuorum walks PREPARE messages (core/ibft.go:864-871 hands it the PREPAREs of the view): in a batch of raw
      // messages of both types (the wire form marks its PREPAREs: no_seal) a COMMIT of the proposer is no PREPARE of his
      const bool is_prepare = !(a.set_on && a.set.no_seal) || a.set.no_seal[row0 + lr] != 0;
      if (is_prepare) by_proposer++;
    }
    if (bit && vi[j] >= 0) {  // unknown senders contribute 0 (validator_manager.go:88-92)
      const uint32_t m = 1u << (vi[j] & 31);
      const uint32_t old = a.lds_bitmap ? atomicOr(&lseen[vi[j] >> 5], m)
                                        : atomicOr(&a.seen[vi[j] >> 5], m);  // device scope (huge sets only)
      first[j] = !(old & m);  // distinct-sender set (validator_manager.go:147-155)
    }
  }
  // the proposer's seat in the sender set: one more "row" of thread 0 of the first workgroup
  bool pfirst = false;
  if (a.prop_on && a.prop_seat && a.prop_vidx >= 0 && blockIdx.x == 0 && tid == 0) {
    const uint32_t m = 1u << (a.prop_vidx & 31);
    const uint32_t old = a.lds_bitmap ? atomicOr(&lseen[a.prop_vidx >> 5], m) : atomicOr(&a.seen[a.prop_vidx >> 5], m);
    pfirst = !(old & m);
  }
  if (MULTI && a.lds_bitmap) {
    // Several workgroups: the set is shared through a bitmap in HBM, but device-scope atomics on one cache line
    // retire at ≈12 ns each — one per ROW made this kernel 52 µs at 4 096 rows.  So the workgroup merges its LDS
    // bitmap WORD by word (one returning atomicOr per non-empty word) and keeps, per word, the bits it was the
    // first to set: a row counts iff it was first inside its workgroup AND its workgroup was first for the validator.
    __syncthreads();
    for (uint32_t i = tid; i < seen_words; i += TALLY_THREADS) {
      const uint32_t mine = lseen[i];
      if (mine) lseen[i] = mine & ~atomicOr(&a.seen[i], mine);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; j++)
    {
      const uint32_t v = first[j] ? (uint32_t)vi[j] : 0u;  // (first[j] implies vi[j] ≥ 0)
      first[j] = first[j] && ((lseen[v >> 5] >> (v & 31)) & 1u);
    }
    if (pfirst) pfirst = (lseen[a.prop_vidx >> 5] >> (a.prop_vidx & 31)) & 1u;
  }
  uint64_t p[NP];
#pragma unroll
  for (int k = 0; k < NP; k++) p[k] = 0;
  if (pfirst) {
    distinct++;
#pragma unroll
    for (int k = 0; k < NP; k++) p[k] += a.vpower32[(size_t)a.prop_vidx * NP + k];
  }
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    if (!first[j]) continue;
    distinct++;
#pragma unroll
    for (int k = 0; k < NP; k++) p[k] += pw[j][k];
  }
  const uint32_t wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const uint64_t sum = wave_sum_u64(p[k]);
    if (lane == 0) part[k][wave] = sum;
  }
  {
    const uint64_t cnt = wave_sum_u64(valid | (distinct << 32));
    if (lane == 0) part[NP][wave] = cnt;
    const uint64_t bp = a.prop_on ? wave_sum_u64((uint64_t)by_proposer) : 0ull;  // (prop_on is launch-uniform)
    if (lane == 0) part[NP + 1][wave] = bp;
  }
  __syncthreads();
  uint64_t piece[TALLY_MAX_PIECES];
  uint64_t v = 0, d = 0, proposer_rows = 0;
  if (MULTI) {
    if (tid <= NP + 1) {  // thread k adds piece k (thread NP: the two counters, NP + 1: the proposer's rows) of this workgroup to the launch-wide sums
      uint64_t sum = 0;
      for (int w = 0; w < TALLY_THREADS / 64; w++) sum += part[tid][w];
      if (tid < NP) {
        if (sum) atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + tid), (unsigned long long)sum);
      } else if (tid == NP + 1) {
        if (sum) atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES + 3), (unsigned long long)sum);
      } else {
        if (sum & 0xFFFFFFFFull) atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES), (unsigned long long)(sum & 0xFFFFFFFFull));
        if (sum >> 32) atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES + 1), (unsigned long long)(sum >> 32));
      }
      // the adds return nothing: wait until the memory side has acknowledged them before this workgroup's ticket
      // can be drawn (device-scope read-modify-writes are performed where they are acknowledged)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // the bitmap atomics returned values (awaited by their users), the adds were awaited above
    if (tid == 0) {
      const unsigned long long t = atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES + 2), 1ull);
      last_flag = (t == (unsigned long long)gridDim.x - 1ull) ? 1u : 0u;
    }
    __syncthreads();
    if (!last_flag) return;
    // ---- the last workgroup: every other one has added its pieces and set its bits ----
    for (uint32_t i = tid; i < (a.n_validators + 31) / 32; i += TALLY_THREADS) {
      if (a.seen_out) a.seen_out[i] = __hip_atomic_load(a.seen + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.seen + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid != 0) return;
#pragma unroll
    for (int k = 0; k < TALLY_MAX_PIECES; k++)
      piece[k] = k < NP ? (uint64_t)atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + k), 0ull) : 0ull;
    v = (uint64_t)atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES), 0ull);
    d = (uint64_t)atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES + 1), 0ull);
    proposer_rows = (uint64_t)atomicAdd(reinterpret_cast<unsigned long long *>(a.acc + TALLY_MAX_PIECES + 3), 0ull);
#pragma unroll
    for (int k = 0; k < TALLY_ACC_WORDS; k++)
      __hip_atomic_store(a.acc + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (a.seen_out)  // (the barrier above is behind every row's atomicOr)
      for (uint32_t i = tid; i < seen_words; i += TALLY_THREADS) a.seen_out[i] = lseen[i];
    if (tid != 0) return;
#pragma unroll
    for (int k = 0; k < TALLY_MAX_PIECES; k++) {
      piece[k] = 0;
      if (k < NP)
        for (int w = 0; w < TALLY_THREADS / 64; w++) piece[k] += part[k][w];
    }
    uint64_t c = 0;
    for (int w = 0; w < TALLY_THREADS / 64; w++) {
      c += part[NP][w];
      proposer_rows += part[NP + 1][w];
    }
    v = c & 0xFFFFFFFFull;
    d = c >> 32;
  }
  uint64_t w[TALLY_SUM_WORDS];
  pieces_to_words(piece, NP, w);
  // a PREPARE from the proposer among the valid ones: no quorum, whatever the power (validator_manager.go:114-121)
  const uint64_t hq = (words_ge(w, a.quorum) && proposer_rows == 0) ? 1 : 0;
  const uint64_t c = (v & 0xFFFFFFFFull) | (d << 32);
  a.out[0] = w[0];
  a.out[1] = w[1];
  a.out[2] = c;
  a.out[3] = hq;
#pragma unroll
  for (int i = 0; i < TALLY_SUM_WORDS; i++) a.out[TALLY_OUT_WIDE + i] = w[i];
#pragma unroll
  for (int k = 0; k < TALLY_MAX_PIECES; k++) a.out[TALLY_OUT_PIECES + k] = piece[k];
  if (a.learned_src) a.out[4] = *a.learned_src;
  a.out[TALLY_OUT_PROPOSER_ROWS] = proposer_rows;
  if (a.host_tally) {
    a.host_tally[TALLY_OUT_PROPOSER_ROWS] = proposer_rows;
    a.host_tally[0] = w[0];
    a.host_tally[1] = w[1];
    a.host_tally[2] = c;
    a.host_tally[3] = hq;
    a.host_tally[4] = a.out[4];  // keys learned | a learned slot (the recover kernels' counter in the device's key cache)
#pragma unroll
    for (int i = 0; i < TALLY_SUM_WORDS; i++) a.host_tally[TALLY_OUT_WIDE + i] = w[i];
  }
}

// ---- multi-GPU: the one exchange step -------------------------------------------------
// Exchange buffer of a sharded batch (u64 slots):
//   [ K × (verdict words of rank 0 | … | rank W−1) | W × ⌈n_validators/64⌉ distinct-sender bitmap words | valid rows ]
// A rank fills its own word range of each of the K verdict arrays (K = 1: a seal / sender batch, K = 2: the sender and
// the valid words of a message set), its own bitmap segment and its count of valid rows; everything else is zero.
// After the all-reduce(SUM) the words are the global verdict masks (disjoint shards: sum ≡ OR) and every rank holds
// every rank's bitmap.  The tally is NOT additive: HasQuorum counts a validator once however many rows it has
// (a map[string]struct{} in core/validator_manager.go:86-92, 147-155), and rows of one sender may lie in two shards —
// so the unpack step ORs the W segments and recomputes power, distinct senders and has_quorum from the merged
// bitmap, exactly what tally_kernel computes from its own bitmap on one device.
struct exchange_pack_args {
  const uint64_t *mask[2];   // this rank's verdict words (my_words each); mask[1] unused when K = 1
  const uint32_t *seen;      // this rank's distinct-sender bitmap (tally_args::seen_out), 2·seen_words 32-bit words
  const uint64_t *tally_out; // this rank's tally (word 2: valid rows | distinct << 32)
  uint64_t *xbuf;
  uint32_t K, world, rank, words_per_rank, my_words, seen_words;
};
__global__ void exchange_pack_kernel(exchange_pack_args a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total_words = a.words_per_rank * a.world;
  const uint32_t seen_base = a.K * total_words, cnt = seen_base + a.world * a.seen_words;
  if (i < seen_base) {
    const uint32_t k = i / total_words, j = i - k * total_words, off = a.rank * a.words_per_rank;
    a.xbuf[i] = (j >= off && j < off + a.my_words) ? a.mask[k][j - off] : 0ull;
  } else if (i < cnt) {
    const uint32_t r = (i - seen_base) / a.seen_words, j = (i - seen_base) - r * a.seen_words;
    a.xbuf[i] = r == a.rank ? ((uint64_t)a.seen[2 * j] | ((uint64_t)a.seen[2 * j + 1] << 32)) : 0ull;
  } else if (i == cnt) {  // valid rows | valid rows sent by the proposer << 32 (both sum across ranks)
    a.xbuf[i] = (a.tally_out[2] & 0xFFFFFFFFull) | (a.tally_out[TALLY_OUT_PROPOSER_ROWS] << 32);
  }
}

// The collective itself when every rank's buffer is addressable from one device (a group that lists one device
// several times, or peer-mapped devices): out[i] = Σ_r buf_r[i], written back into every rank's buffer — what
// ncclAllReduce(sum, u64) leaves there.
constexpr int XSUM_MAX_RANKS = 64;
struct exchange_sum_args {
  uint64_t *buf[XSUM_MAX_RANKS];
  uint32_t world, slots;
};
__global__ void exchange_sum_kernel(exchange_sum_args a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.slots) return;
  uint64_t s = 0;
  for (uint32_t r = 0; r < a.world; r++) s += a.buf[r][i];
  for (uint32_t r = 0; r < a.world; r++) a.buf[r][i] = s;
}

// merged buffer → [ K·total_words verdict words | TALLY_OUT_WORDS-style tally ] in device memory and, when given,
// mapped host memory.  The last workgroup is the tally over the merged bitmap.
constexpr int XUNPACK_THREADS = 1024;
struct exchange_unpack_args {
  const uint64_t *xbuf;
  const uint32_t *vpower32;  // n_validators × n_pieces 32-bit pieces
  const uint64_t *quorum;
  uint64_t *dst, *host_dst;
  uint32_t K, world, words_per_rank, seen_words, n_pieces, n_validators;
  uint32_t prop_on;    // HasPrepareQuorum: the proposer's seat joins the MERGED bitmap, a valid row of his in any shard voids
  int32_t prop_vidx;
};
__global__ void __launch_bounds__(XUNPACK_THREADS) exchange_unpack_kernel(exchange_unpack_args a) {
  const uint32_t total_words = a.words_per_rank * a.world;
  const uint32_t seen_base = a.K * total_words;
  if (blockIdx.x + 1 < gridDim.x) {
    const uint32_t i = blockIdx.x * XUNPACK_THREADS + threadIdx.x;
    if (i < seen_base) {
      const uint64_t w = a.xbuf[i];
      a.dst[i] = w;
      if (a.host_dst) a.host_dst[i] = w;
    }
    return;
  }
  __shared__ uint64_t part[TALLY_MAX_PIECES + 2][XUNPACK_THREADS / 64];
  const uint32_t tid = threadIdx.x;
  uint64_t p[TALLY_MAX_PIECES];
#pragma unroll
  for (int k = 0; k < TALLY_MAX_PIECES; k++) p[k] = 0;
  uint64_t distinct = 0, overlap = 0;
  for (uint32_t j = tid; j < a.seen_words; j += XUNPACK_THREADS) {
    uint64_t m = 0, per_rank = 0;
    for (uint32_t r = 0; r < a.world; r++) {
      const uint64_t s = a.xbuf[seen_base + r * a.seen_words + j];
      m |= s;
      per_rank += (uint64_t)__popcll(s);
    }
    if (a.prop_on && a.prop_vidx >= 0 && (uint32_t)a.prop_vidx / 64u == j) {  // the proposer's seat
      const uint64_t seat = 1ull << ((uint32_t)a.prop_vidx & 63u);
      if (!(m & seat)) per_rank++;  // (not an overlap: keep per_rank − popcount(m) what it was)
      m |= seat;
    }
    const uint64_t here = (uint64_t)__popcll(m);
    distinct += here;
    overlap += per_rank - here;  // senders with valid rows in more than one shard: counted once
    while (m) {
      const uint32_t b = (uint32_t)__ffsll((long long)m) - 1u;
      m &= m - 1;
      const uint32_t v = 64u * j + b;
      if (v < a.n_validators) {
#pragma unroll
        for (int k = 0; k < TALLY_MAX_PIECES; k++)
          if ((uint32_t)k < a.n_pieces) p[k] += a.vpower32[(size_t)v * a.n_pieces + k];
      }
    }
  }
  const uint32_t wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < TALLY_MAX_PIECES; k++) {
    const uint64_t sum = wave_sum_u64(p[k]);
    if (lane == 0) part[k][wave] = sum;
  }
  {
    const uint64_t sd = wave_sum_u64(distinct), so = wave_sum_u64(overlap);
    if (lane == 0) {
      part[TALLY_MAX_PIECES][wave] = sd;
      part[TALLY_MAX_PIECES + 1][wave] = so;
    }
  }
  __syncthreads();
  if (tid != 0) return;
  uint64_t piece[TALLY_MAX_PIECES];
#pragma unroll
  for (int k = 0; k < TALLY_MAX_PIECES; k++) {
    piece[k] = 0;
    for (int w = 0; w < XUNPACK_THREADS / 64; w++) piece[k] += part[k][w];
  }
  uint64_t d = 0, o = 0;
  for (int w = 0; w < XUNPACK_THREADS / 64; w++) {
    d += part[TALLY_MAX_PIECES][w];
    o += part[TALLY_MAX_PIECES + 1][w];
  }
  uint64_t w[TALLY_SUM_WORDS];
  pieces_to_words(piece, TALLY_MAX_PIECES, w);
  uint64_t t[6 + TALLY_SUM_WORDS];
  const uint64_t counts = a.xbuf[seen_base + a.world * a.seen_words];
  const uint64_t proposer_rows = a.prop_on ? counts >> 32 : 0ull;
  t[0] = w[0];
  t[1] = w[1];
  t[2] = (counts & 0xFFFFFFFFull) | (d << 32);
  t[3] = (words_ge(w, a.quorum) && proposer_rows == 0) ? 1 : 0;
#pragma unroll
  for (int k = 0; k < TALLY_SUM_WORDS; k++) t[4 + k] = w[k];
  t[4 + TALLY_SUM_WORDS] = o;
  t[5 + TALLY_SUM_WORDS] = proposer_rows;
#pragma unroll
  for (int k = 0; k < 6 + TALLY_SUM_WORDS; k++) {
    a.dst[seen_base + k] = t[k];
    if (a.host_dst) a.host_dst[seen_base + k] = t[k];
  }
}

// ---- the seal-digest convention (ibft_set_seal_digest) --------------------------------------------------
// Backend.IsValidCommittedSeal is "the signature for the proposal hash in the committed seal" (core/backend.go:53-55);
// WHAT is signed is the embedding Backend's business — the 32-byte proposalHash itself (default), or
// keccak256(proposalHash ‖ suffix) (e.g. a COMMIT-type byte appended before hashing).  One lane per row: the carried hash
// is kept (copy32: what a1 compares) and the row's digest replaces it in the column the verdict kernels read.
struct seal_digest_args {
  uint8_t *hash32;        // n × 32: carried hashes in, digests out
  uint8_t *copy32;        // or null: the carried hashes, unchanged
  uint32_t n;
  uint64_t suffix_words[9];  // suffix ‖ 0x01 ‖ 0… as little-endian words (bytes 32..103 of the one Keccak block)
};
__global__ void seal_digest_kernel(seal_digest_args a) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.n) return;
  uint4 *p = reinterpret_cast<uint4 *>(a.hash32 + 32ull * row);
  const uint4 x = p[0], y = p[1];
  if (a.copy32) {
    uint4 *q = reinterpret_cast<uint4 *>(a.copy32 + 32ull * row);
    q[0] = x;
    q[1] = y;
  }
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  s[0] = (uint64_t)x.x | ((uint64_t)x.y << 32);
  s[1] = (uint64_t)x.z | ((uint64_t)x.w << 32);
  s[2] = (uint64_t)y.x | ((uint64_t)y.y << 32);
  s[3] = (uint64_t)y.z | ((uint64_t)y.w << 32);
#pragma unroll
  for (int j = 0; j < 9; j++) s[4 + j] = a.suffix_words[j];
  s[16] ^= 0x8000000000000000ull;  // pad10*1: the last bit of the 136-byte rate
  keccak::f1600(s);
  p[0] = make_uint4((uint32_t)s[0], (uint32_t)(s[0] >> 32), (uint32_t)s[1], (uint32_t)(s[1] >> 32));
  p[1] = make_uint4((uint32_t)s[2], (uint32_t)(s[2] >> 32), (uint32_t)s[3], (uint32_t)(s[3] >> 32));
}

// sender -> validator index, for ibft_tally() on a caller-supplied mask
// proposer5 (or null): HasPrepareQuorum compares From with the proposer's address byte for byte, member or not
// (validator_manager.go:114-115) — a row of a proposer who is no validator is marked VIDX_PROPOSER_OUTSIDER
struct lookup_proposer {
  uint32_t on, a[5];
};
__global__ void lookup_kernel(const uint8_t *__restrict__ signer20, const uint32_t *__restrict__ vtab,
                              uint32_t slot_mask, uint32_t n, int32_t *__restrict__ vidx, lookup_proposer prop) {
  uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  uint32_t a[5];
  const uint32_t *p = reinterpret_cast<const uint32_t *>(signer20 + 20ull * row);
  for (int i = 0; i < 5; i++) a[i] = p[i];
  int v = valset_lookup(vtab, slot_mask, a);
  if (prop.on && v < 0 && a[0] == prop.a[0] && a[1] == prop.a[1] && a[2] == prop.a[2] && a[3] == prop.a[3] && a[4] == prop.a[4])
    v = VIDX_PROPOSER_OUTSIDER;
  vidx[row] = v;
}

// ---- device canary (ibft_issue_probe) ------------------------------------------------------------------------------
// The verdict kernels are bound by VALU issue, so a device that issues slower than its kind (round 5 met one lease in ≈ 35
// whose every throughput-bound kernel ran 1.3–1.45 × slower) shows in ONE number: the time of a plain
// 8-byte VALU instruction that starts on an 8-byte boundary with one resident wavefront per SIMD — 1.89 ns through this
// probe on a healthy MI355X (the bare stream: 1.78 ns = 4.16 cycles at 2.39 GHz, profiles/r06b_ubench_wave.txt).  ISSUE_PROBE_ITERS × 64 independent v_add_u32 per
// wavefront ≈ 0.24 ms per launch; the launch shape is the rows kernel's at N = 4 096 (256 workgroups of four wavefronts).
constexpr int ISSUE_PROBE_ITERS = 2048;
// 96 KB of LDS per workgroup: more than half of a compute unit's 160 KB, so the dispatcher cannot put two of these
// workgroups on one compute unit — 256 workgroups of four wavefronts are ONE wavefront per SIMD on 256 compute units by
// construction.  (A kernel this small — 9 registers, no LDS — is otherwise packed several workgroups to a compute unit, and
// four wavefronts on a SIMD issue a plain instruction every 6.2 cycles each: the first version of this probe read 2.53 ns on a
// healthy device, profiles/r06a_kernel_ab.txt.)
constexpr int ISSUE_PROBE_LDS_WORDS = 96 * 1024 / 4;
__global__ void __launch_bounds__(256) issue_probe_kernel(uint32_t *out, uint32_t seed) {
  __shared__ uint32_t hold[ISSUE_PROBE_LDS_WORDS];
  hold[threadIdx.x] = seed;
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3;
  uint32_t b0 = a0 ^ 0x1234567, b1 = a1 ^ 0x89abcde, b2 = a2 ^ 0x13579bd, b3 = a3 ^ 0x2468ace;
#pragma unroll 1
  for (int i = 0; i < ISSUE_PROBE_ITERS; i++) {
#define IBFT_P8(x) x x x x x x x x
    // (no .p2align here: go-ibft_amd/phase_align.py keeps 8-byte instructions on 8-byte boundaries itself, and an assembler
    // directive that emits padding breaks its instruction count)
    asm volatile(IBFT_P8("v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n"
                                        "v_add_u32_e64 %3, %3, %8\n v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n"
                                        "v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
                 : "v"(seed | 1));
#undef IBFT_P8
  }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3) == 0x9e3779b9u) out[0] = a0 + hold[(threadIdx.x * 97u) % ISSUE_PROBE_LDS_WORDS];  // keeps the chain (and the LDS block) alive
}

}  // namespace ibftk
