// host_cert_wave_harness.hip — TEST-ONLY: runs cert_wave_dev.h (Keccak by one wavefront, the walk over a certificate) on the CPU
// through the 64-coroutine lockstep emulator in wave_emul.h, so tests/test_dev_cert_wave_host.py can check the exact device source in
// this GPU-less container.  Built with hipcc's host pass; never linked into libibftgpu.so, never a fallback.
#define IBFT_WAVE_EMUL 1
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "cert_wave_dev.h"

namespace {

// what is LDS on the device: memory shared by the 64 coroutines
alignas(16) uint64_t g_A[32], g_B[32];
alignas(16) uint8_t g_win[cw::CERT_WIN_BYTES + 16];

struct sponge_job {
  const uint8_t *m;
  uint32_t cut0, gap, total;
  uint64_t tail;
  bool with_tail;
  uint64_t out[4];
};
void lane_sponge(void *p) {
  sponge_job *j = (sponge_job *)p;
  const uint32_t lane = cw::lane_id();
  const uint64_t w = cw::sponge_message(j->m, j->cut0, j->gap, j->total, j->tail, j->with_tail, lane, g_A, g_B);
  if (lane < 4) j->out[lane] = w;
}

struct walk_job {
  const uint8_t *wire;
  uint32_t pos, end;
  bool pc;
  uint32_t *rec;  // cap × 3: offset, length, role by ordinal
  uint32_t cap, count;
  bool ok, overflow;
};
void lane_walk(void *p) {
  walk_job *j = (walk_job *)p;
  const uint32_t lane = cw::lane_id();
  bool ok = true;
  const uint32_t count = cw::walk_certificate(j->wire, j->pos, j->end, j->pc, g_win, lane, ok, [&](uint32_t ordinal, uint32_t off, uint32_t len, uint8_t role) {
    if (ordinal >= j->cap) {
      j->overflow = true;
      return;
    }
    j->rec[3 * ordinal] = off;
    j->rec[3 * ordinal + 1] = len;
    j->rec[3 * ordinal + 2] = role;
  });
  if (lane == 0) {
    j->count = count;
    j->ok = ok;
  }
}

}  // namespace

extern "C" {

// keccak256(m[0, cut0) ‖ m[cut1, len)): PayloadNoSig of a canonical message whose signature field is [cut0, cut1)
void cwh_keccak_without(const uint8_t *m, uint32_t len, uint32_t cut0, uint32_t cut1, uint8_t *out32) {
  sponge_job j{m, cut0, cut1 - cut0, len - (cut1 - cut0), 0, false, {0, 0, 0, 0}};
  wave_emul::run(lane_sponge, &j);
  memcpy(out32, j.out, 32);
}
// keccak256(raw ‖ BE64(round)): the proposal hash
void cwh_keccak_proposal(const uint8_t *raw, uint32_t raw_len, uint64_t round, uint8_t *out32) {
  uint64_t tail = 0;
  for (int k = 0; k < 8; k++) tail |= (uint64_t)((round >> (8 * (7 - k))) & 0xFFu) << (8 * k);
  sponge_job j{raw, raw_len, 0, raw_len + 8u, tail, true, {0, 0, 0, 0}};
  wave_emul::run(lane_sponge, &j);
  memcpy(out32, j.out, 32);
}
// the nested messages of the certificate wire[pos, end): returns their number (rec filled by ordinal), −1 malformed, −2 more than cap.
// `wire` must be 16-byte aligned and carry 16 bytes of slack behind `end`.
int64_t cwh_walk(const uint8_t *wire, uint32_t pos, uint32_t end, int pc, uint32_t *rec, uint32_t cap) {
  walk_job j{wire, pos, end, pc != 0, rec, cap, 0, true, false};
  wave_emul::run(lane_walk, &j);
  if (!j.ok) return -1;
  if (j.overflow) return -2;
  return j.count;
}

}  // extern "C"
