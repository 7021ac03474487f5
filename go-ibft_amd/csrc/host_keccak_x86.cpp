// host_keccak_x86.cpp — Keccak-f[1600] for the HOST side of libibftgpu.so on an x86-64 core with AVX-512F.
//
// Product host code (plain C++, no device pass).  Why it exists: a sponge is sequential, so what is too long for one
// wavefront (25 MB/s) is hashed on the host (ibftgpu.hip: host_keccak256 — proposals above a KiB, the multi-megabyte
// envelope of a re-proposal whose signature covers its whole RoundChangeCertificate), and there the permutation IS the
// latency: 4.3 MB at N = 256 were 7 of the 8 ms a node needs to validate the new proposal.
//
// Layout: a plane per register — A[y] holds the lanes (x = 0..4, y) in qword positions 0..4 (positions 5..7 are never
// read into 0..4).  θ and ρ are plane-wise (two in-register rotations of the parity plane, per-lane rotates).  π sends
// lane (x, y) to (y, 2x + 3y): a whole plane y becomes the COLUMN x' = y of the result, permuted inside the register
// (position p takes lane 3p + y), so after five in-register permutes the registers hold columns — and χ, which combines
// x, x+1, x+2 of one row, is then three-operand logic on whole registers without any shuffle.  One 5 × 5 transpose
// (nine two-source permutes, four single ones, five blends) brings the planes back for the next round's θ.  40 instructions
// a round, 20 of them shuffles, against ≈160 scalar operations.
//
// Checked against the scalar permutation of keccak_dev.h on every input of tests/test_cabi.py::test_host_keccak_entry…
// (block boundaries, two-part inputs) — the scalar code stays the reference and the fallback.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <immintrin.h>

namespace {

alignas(64) const uint64_t kRho[5][8] = {{0, 1, 62, 28, 27, 0, 0, 0},
                                         {36, 44, 6, 55, 20, 0, 0, 0},
                                         {3, 10, 43, 25, 39, 0, 0, 0},
                                         {41, 45, 15, 21, 8, 0, 0, 0},
                                         {18, 2, 61, 56, 14, 0, 0, 0}};
// π inside the register of old plane j: position p takes lane (3p + j) mod 5
alignas(64) const uint64_t kPi[5][8] = {{0, 3, 1, 4, 2, 5, 6, 7},
                                        {1, 4, 2, 0, 3, 5, 6, 7},
                                        {2, 0, 3, 1, 4, 5, 6, 7},
                                        {3, 1, 4, 2, 0, 5, 6, 7},
                                        {4, 2, 0, 3, 1, 5, 6, 7}};
alignas(64) const uint64_t kPrev[8] = {4, 0, 1, 2, 3, 5, 6, 7};  // C[x-1]
alignas(64) const uint64_t kNext[8] = {1, 2, 3, 4, 0, 5, 6, 7};  // C[x+1]
alignas(64) const uint64_t kZipLo[8] = {0, 8, 1, 9, 2, 10, 3, 11};   // a0 b0 a1 b1 a2 b2 a3 b3
alignas(64) const uint64_t kZipHi[8] = {4, 12, 4, 12, 4, 12, 4, 12};  // a4 b4 …
alignas(64) const uint64_t kPick[4][8] = {{0, 1, 8, 9, 4, 5, 6, 7},
                                          {2, 3, 10, 11, 4, 5, 6, 7},
                                          {4, 5, 12, 13, 4, 5, 6, 7},
                                          {6, 7, 14, 15, 4, 5, 6, 7}};
alignas(64) const uint64_t kFifth[4][8] = {{0, 1, 2, 3, 0, 5, 6, 7},   // position 4 takes lane y of column 4
                                           {0, 1, 2, 3, 1, 5, 6, 7},
                                           {0, 1, 2, 3, 2, 5, 6, 7},
                                           {0, 1, 2, 3, 3, 5, 6, 7}};
const uint64_t kRC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                          0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                          0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                          0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                          0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                          0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

#define IBFT_AVX512 __attribute__((target("avx512f")))

IBFT_AVX512 inline __m512i ld(const uint64_t *p) { return _mm512_load_si512((const void *)p); }

// absorb `blocks` whole 136-byte blocks (Keccak-256's rate) into the state, one permutation after each
IBFT_AVX512 void absorb_avx512(uint64_t s[25], const uint8_t *p, size_t blocks) {
  __m512i a0 = _mm512_maskz_loadu_epi64(0x1F, s + 0), a1 = _mm512_maskz_loadu_epi64(0x1F, s + 5),
          a2 = _mm512_maskz_loadu_epi64(0x1F, s + 10), a3 = _mm512_maskz_loadu_epi64(0x1F, s + 15),
          a4 = _mm512_maskz_loadu_epi64(0x1F, s + 20);
  const __m512i prev = ld(kPrev), next = ld(kNext), zip_lo = ld(kZipLo), zip_hi = ld(kZipHi);
  const __m512i rho0 = ld(kRho[0]), rho1 = ld(kRho[1]), rho2 = ld(kRho[2]), rho3 = ld(kRho[3]), rho4 = ld(kRho[4]);
  const __m512i pi0 = ld(kPi[0]), pi1 = ld(kPi[1]), pi2 = ld(kPi[2]), pi3 = ld(kPi[3]), pi4 = ld(kPi[4]);
  const __m512i pick0 = ld(kPick[0]), pick1 = ld(kPick[1]), pick2 = ld(kPick[2]), pick3 = ld(kPick[3]);
  const __m512i f0 = ld(kFifth[0]), f1 = ld(kFifth[1]), f2 = ld(kFifth[2]), f3 = ld(kFifth[3]);
  for (; blocks; blocks--, p += 136) {
    // the 17 words of the block: planes 0..2 whole, two lanes of plane 3
    a0 = _mm512_xor_si512(a0, _mm512_maskz_loadu_epi64(0x1F, p));
    a1 = _mm512_xor_si512(a1, _mm512_maskz_loadu_epi64(0x1F, p + 40));
    a2 = _mm512_xor_si512(a2, _mm512_maskz_loadu_epi64(0x1F, p + 80));
    a3 = _mm512_xor_si512(a3, _mm512_maskz_loadu_epi64(0x03, p + 120));
    for (int round = 0; round < 24; round++) {
      // θ
      __m512i c = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(a0, a1, a2, 0x96), a3, a4, 0x96);
      const __m512i dm = _mm512_permutexvar_epi64(prev, c), dp = _mm512_rol_epi64(_mm512_permutexvar_epi64(next, c), 1);  // D = dm ^ dp
      // ρ, then π inside each register: the planes become the columns of the result
      const __m512i r0 = _mm512_permutexvar_epi64(pi0, _mm512_rolv_epi64(_mm512_ternarylogic_epi64(a0, dm, dp, 0x96), rho0));
      const __m512i r1 = _mm512_permutexvar_epi64(pi1, _mm512_rolv_epi64(_mm512_ternarylogic_epi64(a1, dm, dp, 0x96), rho1));
      const __m512i r2 = _mm512_permutexvar_epi64(pi2, _mm512_rolv_epi64(_mm512_ternarylogic_epi64(a2, dm, dp, 0x96), rho2));
      const __m512i r3 = _mm512_permutexvar_epi64(pi3, _mm512_rolv_epi64(_mm512_ternarylogic_epi64(a3, dm, dp, 0x96), rho3));
      const __m512i r4 = _mm512_permutexvar_epi64(pi4, _mm512_rolv_epi64(_mm512_ternarylogic_epi64(a4, dm, dp, 0x96), rho4));
      // χ on columns: x, x + 1, x + 2 are whole registers;  ι on lane (0, 0)
      __m512i e0 = _mm512_ternarylogic_epi64(r0, r1, r2, 0xD2);
      const __m512i e1 = _mm512_ternarylogic_epi64(r1, r2, r3, 0xD2);
      const __m512i e2 = _mm512_ternarylogic_epi64(r2, r3, r4, 0xD2);
      const __m512i e3 = _mm512_ternarylogic_epi64(r3, r4, r0, 0xD2);
      const __m512i e4 = _mm512_ternarylogic_epi64(r4, r0, r1, 0xD2);
      e0 = _mm512_xor_si512(e0, _mm512_maskz_set1_epi64(0x01, (long long)kRC[round]));
      // columns back to planes: A[y][x] = E[x][y]
      const __m512i u01 = _mm512_permutex2var_epi64(e0, zip_lo, e1), u23 = _mm512_permutex2var_epi64(e2, zip_lo, e3);
      const __m512i v01 = _mm512_permutex2var_epi64(e0, zip_hi, e1), v23 = _mm512_permutex2var_epi64(e2, zip_hi, e3);
      // (the fifth lane of each plane comes from column 4 alone: moved to position 4 beside the zips, blended in last)
      a0 = _mm512_mask_blend_epi64(0x10, _mm512_permutex2var_epi64(u01, pick0, u23), _mm512_permutexvar_epi64(f0, e4));
      a1 = _mm512_mask_blend_epi64(0x10, _mm512_permutex2var_epi64(u01, pick1, u23), _mm512_permutexvar_epi64(f1, e4));
      a2 = _mm512_mask_blend_epi64(0x10, _mm512_permutex2var_epi64(u01, pick2, u23), _mm512_permutexvar_epi64(f2, e4));
      a3 = _mm512_mask_blend_epi64(0x10, _mm512_permutex2var_epi64(u01, pick3, u23), _mm512_permutexvar_epi64(f3, e4));
      a4 = _mm512_mask_blend_epi64(0x10, _mm512_permutex2var_epi64(v01, pick0, v23), e4);
    }
  }
  _mm512_mask_storeu_epi64(s + 0, 0x1F, a0);
  _mm512_mask_storeu_epi64(s + 5, 0x1F, a1);
  _mm512_mask_storeu_epi64(s + 10, 0x1F, a2);
  _mm512_mask_storeu_epi64(s + 15, 0x1F, a3);
  _mm512_mask_storeu_epi64(s + 20, 0x1F, a4);
}

}  // namespace

// 1 = absorbed (an AVX-512F core), 0 = not available here: the caller uses the scalar permutation
extern "C" int ibftk_host_keccak_absorb_x86(uint64_t s[25], const uint8_t *p, size_t blocks) {
  static const bool have = __builtin_cpu_supports("avx512f");
  if (!have) return 0;
  absorb_avx512(s, p, blocks);
  return 1;
}
