// keccak_dev.h — Keccak-f[1600] / Keccak-256 (0x01 padding, rate 136) for gfx950.
//
// Product code.  Implements the "Keccak hash of the proposal" that go-ibft leaves to
// the application (/root/reference/messages/proto/messages.proto:51,61,67;
// /root/reference/core/ibft.go:648) and the address / sender-digest hashes of
// IsValidCommittedSeal / IsValidValidator (/root/reference/core/backend.go:41-55).
//
// One hash per lane: the 25-lane state lives in 50 VGPRs, rho rotations are compile-time
// constants (v_alignbit_b32 pairs).  The 24 rounds are a ROLLED loop: unrolled they are 54 KB of
// straight-line code per hash site — more than the 64 KB instruction cache once the curve
// arithmetic is next to it, and code that runs once per signature is fetch-bound (measured
// ≈20 cycles per instruction cold vs ≈6 from the cache).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef HD
#define HD __host__ __device__ __forceinline__
#endif

namespace keccak {

HD uint64_t rc(int round) {
  const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  return RC[round];
}

template <int N>
HD uint64_t rotl(uint64_t v) {
  if constexpr (N == 0)
    return v;
  else
    return (v << N) | (v >> (64 - N));
}

// state index = x + 5*y
HD void f1600(uint64_t s[25]) {
#pragma unroll 1
  for (int round = 0; round < 24; round++) {
    uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20];
    uint64_t c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];
    uint64_t c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22];
    uint64_t c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];
    uint64_t c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
    uint64_t d0 = c4 ^ rotl<1>(c1);
    uint64_t d1 = c0 ^ rotl<1>(c2);
    uint64_t d2 = c1 ^ rotl<1>(c3);
    uint64_t d3 = c2 ^ rotl<1>(c4);
    uint64_t d4 = c3 ^ rotl<1>(c0);
    // theta + rho + pi: b[y + 5*((2x+3y)%5)] = rotl(s[x+5y] ^ d[x], rho[x+5y])
    uint64_t b[25];
    b[0] = s[0] ^ d0;
    b[10] = rotl<1>(s[1] ^ d1);
    b[20] = rotl<62>(s[2] ^ d2);
    b[5] = rotl<28>(s[3] ^ d3);
    b[15] = rotl<27>(s[4] ^ d4);
    b[16] = rotl<36>(s[5] ^ d0);
    b[1] = rotl<44>(s[6] ^ d1);
    b[11] = rotl<6>(s[7] ^ d2);
    b[21] = rotl<55>(s[8] ^ d3);
    b[6] = rotl<20>(s[9] ^ d4);
    b[7] = rotl<3>(s[10] ^ d0);
    b[17] = rotl<10>(s[11] ^ d1);
    b[2] = rotl<43>(s[12] ^ d2);
    b[12] = rotl<25>(s[13] ^ d3);
    b[22] = rotl<39>(s[14] ^ d4);
    b[23] = rotl<41>(s[15] ^ d0);
    b[8] = rotl<45>(s[16] ^ d1);
    b[18] = rotl<15>(s[17] ^ d2);
    b[3] = rotl<21>(s[18] ^ d3);
    b[13] = rotl<8>(s[19] ^ d4);
    b[14] = rotl<18>(s[20] ^ d0);
    b[24] = rotl<2>(s[21] ^ d1);
    b[9] = rotl<61>(s[22] ^ d2);
    b[19] = rotl<56>(s[23] ^ d3);
    b[4] = rotl<14>(s[24] ^ d4);
    // chi
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
      s[y + 0] = b[y + 0] ^ (~b[y + 1] & b[y + 2]);
      s[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
      s[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
      s[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y + 0]);
      s[y + 4] = b[y + 4] ^ (~b[y + 0] & b[y + 1]);
    }
    s[0] ^= rc(round);
  }
}

HD uint32_t bswap32(uint32_t v) {
  return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
}
HD uint64_t bswap64(uint64_t v) {
  return ((uint64_t)bswap32((uint32_t)v) << 32) | bswap32((uint32_t)(v >> 32));
}

// Keccak-256 of exactly 64 bytes given as two big-endian 256-bit integers (8 LE limbs
// each, limb 7 = most significant): the public key X‖Y.  Returns the low 20 bytes of
// the digest (the address) as 5 little-endian-loaded dwords of digest[12..32).
HD void address_from_xy(const uint32_t x[8], const uint32_t y[8], uint32_t addr[5]) {
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  // message byte k of X is the big-endian byte; state lane j holds bytes 8j..8j+7 little-endian
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint64_t be = ((uint64_t)x[7 - 2 * j] << 32) | x[6 - 2 * j];  // bytes 8j..8j+7 as a BE number
    s[j] = bswap64(be);
    uint64_t be2 = ((uint64_t)y[7 - 2 * j] << 32) | y[6 - 2 * j];
    s[4 + j] = bswap64(be2);
  }
  s[8] = 0x01ULL;                 // pad start at byte 64
  s[16] ^= 0x8000000000000000ULL; // pad end at byte 135
  f1600(s);
  // digest bytes 12..31 = upper half of s[1], all of s[2], s[3]
  addr[0] = (uint32_t)(s[1] >> 32);
  addr[1] = (uint32_t)s[2];
  addr[2] = (uint32_t)(s[2] >> 32);
  addr[3] = (uint32_t)s[3];
  addr[4] = (uint32_t)(s[3] >> 32);
}

// Streaming Keccak-256 over a byte range (one lane, sequential blocks).
HD void hash_bytes(const uint8_t *in, uint32_t len, uint64_t out4[4]) {
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  while (len >= 136) {
    for (int i = 0; i < 17; i++) {
      uint64_t w = 0;
      for (int b = 0; b < 8; b++) w |= (uint64_t)in[8 * i + b] << (8 * b);
      s[i] ^= w;
    }
    f1600(s);
    in += 136;
    len -= 136;
  }
  // final block: absorb len remaining bytes + padding
  for (int i = 0; i < 17; i++) {
    uint64_t w = 0;
    for (int b = 0; b < 8; b++) {
      uint32_t pos = 8 * i + b;
      uint64_t byte = pos < len ? in[pos] : 0u;
      if (pos == len) byte ^= 0x01u;
      if (pos == 135) byte ^= 0x80u;
      w |= byte << (8 * b);
    }
    s[i] ^= w;
  }
  f1600(s);
#pragma unroll
  for (int i = 0; i < 4; i++) out4[i] = s[i];
}

// digest lanes (little-endian bytes) -> 256-bit big-endian integer in 8 LE limbs
HD void digest_to_limbs(const uint64_t d[4], uint32_t limbs[8]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint64_t be = bswap64(d[j]);  // bytes 8j..8j+7 as a BE number
    limbs[7 - 2 * j] = (uint32_t)(be >> 32);
    limbs[6 - 2 * j] = (uint32_t)be;
  }
}

}  // namespace keccak
