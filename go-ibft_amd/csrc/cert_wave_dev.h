// cert_wave_dev.h — the wavefront-cooperative pieces of the certificate path (ibft_verify_certificates_wire) and of the proposal
// hash: Keccak-256 of one long message by one wavefront, and the walk over the length-prefixed messages of a certificate.
//
// Product code (kernels.hip.h wraps these in kernels).  Written against three primitives — lane id, a wavefront barrier that orders
// the LDS accesses, a ballot — so that the very same source also runs in the build container, which has no GPU: with IBFT_WAVE_EMUL
// the lanes are the 64 lockstep coroutines of wave_emul.h (TEST ONLY: csrc/host_wave_harness.hip, tests/test_dev_cert_wave_host.py)
// and "LDS" is ordinary memory shared by them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keccak_dev.h"
#include "wire_dev.h"
#if !defined(__HIP_DEVICE_COMPILE__) && defined(IBFT_WAVE_EMUL)
#include "wave_emul.h"  // tests only
#endif

namespace cw {

HD uint32_t lane_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  return threadIdx.x & 63u;
#elif defined(IBFT_WAVE_EMUL)
  return (uint32_t)wave_emul::lane();
#else
  return 0;  // host pass of the product build: never called
#endif
}
// every lane's LDS accesses before it are visible to every lane after it.  One wavefront per workgroup on the device: no s_barrier is
// emitted for __syncthreads, only the ordering (and the wait for outstanding LDS operations).
HD void barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  __syncthreads();
#elif defined(IBFT_WAVE_EMUL)
  (void)wave_emul::ballot(true);
#endif
}
HD uint64_t ballot(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(c);
#elif defined(IBFT_WAVE_EMUL)
  return wave_emul::ballot(c);
#else
  return c ? 1ull : 0ull;
#endif
}

// nbytes of the buffer at src (16-byte aligned) → LDS, one wavefront: 16-byte loads, four in flight per lane.  (A dword per lane and
// trip — the first form — is a chain of ≈1 µs round trips: 16 KiB took ≈50 µs.)  Reads up to 15 bytes past src + nbytes: the
// payload buffer carries 256 bytes of slack; the LDS buffer must hold nbytes rounded up to 16.
HD void stage_bytes(uint8_t *lds, const uint8_t *src, uint32_t nbytes, uint32_t lane) {
  const uint32_t chunks = (nbytes + 15u) >> 4;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
  uint4 *d4 = reinterpret_cast<uint4 *>(lds);
  for (uint32_t c0 = 0; c0 < chunks; c0 += 256u) {
    uint4 v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      const uint32_t c = c0 + lane + 64u * k;
      v[k] = c < chunks ? s4[c] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      const uint32_t c = c0 + lane + 64u * k;
      if (c < chunks) d4[c] = v[k];
    }
  }
}

// ---- Keccak-256 of one long message by one wavefront -------------------------------------------------------------------
// A sponge is sequential, and one lane (or the scalar unit) spends ≈9–14 µs on a 136-byte block: ≈188 64-bit operations per round
// at one instruction per ≈4 ticks.  A message that carries a certificate is tens of kilobytes, so here 25 lanes hold one
// 64-bit word of the state each (lane i = x + 5y) and the words meet in LDS: per round every lane
//   θ   writes its word, reads the two neighbouring COLUMNS (10 words), forms D[x] = C[x−1] ^ rotl(C[x+1], 1) itself;
//   ρ,π rotates its word by its own offset and writes it to where π sends it;
//   χ,ι reads the two words to its right in its row, combines, lane 0 adds the round constant.
// Two dependent LDS round trips and ≈35 VALU instructions per round instead of ≈190 (or ≈380 32-bit ones): 5.3 µs per block.
HD uint32_t keccak_rho(uint32_t i) {
  const uint8_t RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  return RHO[i];
}
HD uint64_t rotl64_var(uint64_t v, uint32_t r) { return r ? (v << r) | (v >> (64u - r)) : v; }
// The lane's share of the state and where its neighbours are.  A and B are 32 × u64 of LDS each.
struct wave_sponge {
  uint64_t s;  // state word i = x + 5y of lane i < 25 (lanes 25…63 mirror lane 0 and never write)
  uint32_t i, cm, cp, pi, r1, r2, rho;
  bool act, first;
  HD void init(uint32_t lane) {
    act = lane < 25u;
    first = lane == 0;
    i = act ? lane : 0u;
    const uint32_t x = i % 5u, y = i / 5u;
    cm = (x + 4u) % 5u;                       // the columns on either side
    cp = (x + 1u) % 5u;
    pi = y + 5u * ((2u * x + 3u * y) % 5u);   // where π sends this lane's word
    r1 = (x + 1u) % 5u + 5u * y;
    r2 = (x + 2u) % 5u + 5u * y;
    rho = keccak_rho(i);
    s = 0;
  }
  HD void permute(uint64_t *A, uint64_t *B) {
#pragma unroll
    for (int round = 0; round < 24; round++) {  // unrolled: the round constants are literals, nothing is loaded inside the chain
      if (act) A[i] = s;
      barrier();
      const uint64_t c_minus = A[cm] ^ A[cm + 5] ^ A[cm + 10] ^ A[cm + 15] ^ A[cm + 20];
      const uint64_t c_plus = A[cp] ^ A[cp + 5] ^ A[cp + 10] ^ A[cp + 15] ^ A[cp + 20];
      s ^= c_minus ^ ((c_plus << 1) | (c_plus >> 63));
      if (act) B[pi] = rotl64_var(s, rho);
      barrier();
      s = B[i] ^ (~B[r1] & B[r2]);
      s ^= first ? keccak::rc(round) : 0ull;
    }
  }
};

// Keccak-256 of the message m[0, cut0) ‖ m[cut0 + gap, cut0 + gap + …) ‖ tail, `total` bytes in all, by the calling wavefront; every
// lane returns its state word (lanes 0…3: the digest).  with_tail = false: PayloadNoSig of a canonical message — its bytes minus the
// signature field [cut0, cut0 + gap) —, total = len − gap.  with_tail = true: a Proposal — gap = 0, m[0, cut0) is the raw proposal and
// the 8 bytes of `tail` (as they lie in memory) follow.  Lanes 0…16 fetch their 8 bytes of each block with aligned dword loads (up
// to 3 bytes past them: buffer slack), byte by byte only where a word straddles the cut or lies in the last block.
HD uint64_t sponge_message(const uint8_t *m, uint32_t cut0, uint32_t gap, uint32_t total, uint64_t tail, bool with_tail, uint32_t lane,
                           uint64_t *A, uint64_t *B) {
  wave_sponge sp;
  sp.init(lane);
  for (uint32_t done = 0;; done += 136u) {
    const uint32_t left = total - done;
    const bool last = left < 136u;
    if (lane < 17u) {
      const uint32_t v = done + 8u * lane;  // this lane's 8 bytes of the block, in the message
      uint64_t w = 0;
      const bool whole = with_tail ? v + 8u <= cut0 : (v + 8u <= cut0 || v >= cut0);
      if (!last && whole) {
        const uint8_t *p = m + (v < cut0 ? v : v + gap);
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u), sh = 8u * mis;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p - mis);
        const uint32_t d0 = q[0], d1 = q[1], d2 = mis ? q[2] : 0u;
        const uint32_t lo = (uint32_t)(((uint64_t)d1 << 32 | d0) >> sh), hi = (uint32_t)(((uint64_t)d2 << 32 | d1) >> sh);
        w = (uint64_t)lo | ((uint64_t)hi << 32);
      } else {
        for (uint32_t k = 0; k < 8u; k++) {
          const uint32_t o = v + k, in_block = 8u * lane + k;
          uint64_t byte = 0;
          if (o < total) {
            if (with_tail)
              byte = o < cut0 ? m[o] : (tail >> (8u * (o - cut0))) & 0xFFu;
            else
              byte = m[o < cut0 ? o : o + gap];
          }
          if (last && in_block == left) byte ^= 0x01u;
          if (last && in_block == 135u) byte ^= 0x80u;
          w |= byte << (8u * k);
        }
      }
      sp.s ^= w;
    }
    sp.permute(A, B);
    if (last) break;
  }
  return sp.s;
}

// ---- the walk over a certificate ------------------------------------------------------------------------------------------
// A certificate is a run of length-prefixed messages: finding message k needs the lengths of the k − 1 before it, a chain of dependent
// loads (≈1 µs each from HBM: a PreparedCertificate of 2 731 PREPAREs would take milliseconds).  So a wavefront brings the certificate
// through LDS in windows (coalesced) and hops from header to header there — one message at a time (≈0.6 µs each), or, where the
// messages are of one size, up to 64 at a time (RUN, below).
constexpr uint32_t CERT_WIN_BYTES = 16 * 1024;
// bytes [w0, …) of the buffer held in LDS, addressed by their position in the buffer
struct lds_window {
  const uint8_t *win;
  uint32_t w0;
  HD uint8_t operator[](uint32_t a) const { return win[a - w0]; }
};
// Walks the certificate [pos, end) of the buffer (pc: a PreparedCertificate, else a RoundChangeCertificate); emit(ordinal, offset, length,
// role) is called for every nested message — by lane 0 for a single hop, by lanes 0 … k−1 for a run; returns their number, ok = false
// when the wrapper is malformed.  win: CERT_WIN_BYTES + 16 bytes of LDS.  Everything is wave-uniform — every lane decodes the same
// header from LDS (broadcast reads) — except the RUN step.
template <typename EMIT>
HD uint32_t walk_certificate(const uint8_t *wire_bytes, uint32_t pos, uint32_t end, bool pc, uint8_t *win, uint32_t lane, bool &ok, EMIT emit) {
  uint32_t last = 0, count = 0;
  uint32_t H = 0, L = 0, S = 0;  // header bytes, body length and stride of the last message hopped over (S = 0: none yet)
  uint8_t role_s = 0;
  ok = true;
  while (pos < end && ok) {
    const uint32_t w0 = pos & ~15u;
    const uint32_t wend = end - w0 <= CERT_WIN_BYTES ? end : w0 + CERT_WIN_BYTES;
    stage_bytes(win, wire_bytes + w0, wend - w0, lane);
    barrier();
    const lds_window W{win, w0};
    while (pos < wend) {
      if (pos + 6u > wend && wend < end) break;  // tag + length prefix (≤ 6 bytes) may cross the window: the next window starts here
      // RUN: the messages of a certificate are mostly of one size (PREPAREs of one view).  Lane j checks that a message of the last
      // stride starts at pos + j·S — same field, same header, same length; if lanes 0 … k−1 all agree, the chain pos → pos + S → …
      // is proven link by link and k messages are hopped over at once (one LDS round trip instead of k dependent ones).
      if (S) {
        const uint64_t q64 = (uint64_t)pos + (uint64_t)lane * S;
        bool good = q64 + S <= end && (q64 + 6u <= wend || wend == end);  // fits the certificate; its header lies in the window
        const uint32_t q = (uint32_t)q64;
        if (good) {
          uint32_t qq = q, len_j = 0, l2 = last;
          uint8_t role_j = 0;
          good = wire::cert_child_header(W, end, pc, l2, qq, len_j, role_j) && len_j == L && qq - q == H && role_j == role_s && l2 == last;
        }
        const uint64_t m = ballot(good);
        const uint32_t k = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);  // lanes 0 … k−1 agree
        if (k) {
          if (lane < k) emit(count + lane, q + H, L, role_s);
          count += k;
          pos += k * S;
          continue;
        }
      }
      // one message
      uint32_t p2 = pos, len1 = 0, l2 = last;
      uint8_t role1 = 0;
      if (!wire::cert_child_header(W, end, pc, l2, p2, len1, role1)) {
        ok = false;
        break;
      }
      if (lane == 0) emit(count, p2, len1, role1);
      H = p2 - pos;
      L = len1;
      S = H + L;
      role_s = role1;
      last = l2;
      count++;
      pos = p2 + len1;  // the message itself is skipped: its own lane walks it on the next level
    }
    barrier();
  }
  return count;
}

}  // namespace cw
