// wave_emul.h — a 64-lane lockstep wavefront emulator for HOST builds of wave_fe_dev.h.
//
// TEST INFRASTRUCTURE ONLY: never part of libibftgpu.so.  wave_fe_dev.h spreads one field
// element over the lanes of a wavefront and moves limbs between lanes with DPP; to run that
// exact source on the CPU (tests/test_dev_wave_host.py) every "lane" becomes a ucontext
// coroutine and each cross-lane primitive becomes a rendezvous:
//   write my value → switch to the next lane … (all 64 have written) → read the source lane.
// Lanes are resumed strictly round-robin and all of them execute the same sequence of
// exchanges (the code under test has wave-uniform control flow), so two buffers alternate
// safely: a lane is never more than one exchange ahead of any other.
#pragma once
#if defined(__HIP_DEVICE_COMPILE__)
#error "wave_emul.h is for host builds"
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

namespace wave_emul {

constexpr int LANES = 64;
constexpr size_t STACK_BYTES = 1u << 20;

struct state {
  ucontext_t main_ctx, lane_ctx[LANES];
  char *stacks = nullptr;
  uint32_t buf[2][LANES];
  uint32_t tag[2][LANES];  // which primitive each lane thinks this rendezvous is
  uint64_t phase[LANES];
  bool alive[LANES];
  int cur = -1;
  void (*fn)(void *) = nullptr;
  void *arg = nullptr;
};
inline state &S() {
  static thread_local state s;
  return s;
}
inline int lane() { return S().cur; }

inline void yield_to_next() {
  state &s = S();
  const int me = s.cur;
  // strict round-robin over live lanes; control comes back here once every other lane has
  // reached its next rendezvous (or finished)
  int nxt = me;
  for (int k = 1; k <= LANES; k++) {
    int c = (me + k) % LANES;
    if (s.alive[c]) {
      nxt = c;
      break;
    }
  }
  if (nxt == me) return;
  s.cur = nxt;
  swapcontext(&s.lane_ctx[me], &s.lane_ctx[nxt]);
}

// value of `v` held by lane `src` at this rendezvous (src < 0: zero)
inline void check_same_primitive(state &s, int me, uint64_t ph, uint32_t tag) {
  // the code under test must be wave-uniform around cross-lane operations: every live lane has to
  // arrive at THIS rendezvous through the same primitive (a divergent `c ? f(dpp) : x` is caught here)
  for (int l = 0; l < LANES; l++)
    if (s.alive[l] && s.tag[ph & 1][l] != tag) {
      fprintf(stderr, "wave_emul: divergent cross-lane op (lane %d op %#x, lane %d op %#x)\n", me, tag, l,
              s.tag[ph & 1][l]);
      abort();
    }
}
inline uint32_t xchg(uint32_t v, int src, uint32_t tag) {
  state &s = S();
  const int me = s.cur;
  const uint64_t ph = s.phase[me]++;
  s.buf[ph & 1][me] = v;
  s.tag[ph & 1][me] = tag;
  yield_to_next();
  check_same_primitive(s, me, ph, tag);
  // strict lockstep: lanes resumed before me in this round are one rendezvous further, the others
  // have just written this one — anything else means the code under test diverged around a cross-lane op
  for (int l = 0; l < LANES; l++)
    if (s.alive[l] && s.phase[l] != (l < me ? ph + 2 : ph + 1)) {
      fprintf(stderr, "wave_emul: lanes out of lockstep (lane %d phase %llu, lane %d phase %llu)\n", me,
              (unsigned long long)ph, l, (unsigned long long)s.phase[l]);
      abort();
    }
  return src < 0 ? 0u : s.buf[ph & 1][src];
}
inline uint64_t ballot(bool c) {
  state &s = S();
  const int me = s.cur;
  const uint64_t ph = s.phase[me]++;
  s.buf[ph & 1][me] = c ? 1u : 0u;
  s.tag[ph & 1][me] = 0xBA110000u;
  yield_to_next();
  check_same_primitive(s, me, ph, 0xBA110000u);
  uint64_t m = 0;
  for (int l = 0; l < LANES; l++) m |= (uint64_t)(s.buf[ph & 1][l] & 1u) << l;
  return m;
}

inline void trampoline() {
  state &s = S();
  const int me = s.cur;
  s.fn(s.arg);
  s.alive[me] = false;
  // hand over to the next live lane, or back to main when this was the last one
  for (int k = 1; k <= LANES; k++) {
    int c = (me + k) % LANES;
    if (s.alive[c]) {
      s.cur = c;
      setcontext(&s.lane_ctx[c]);
    }
  }
  setcontext(&s.main_ctx);
}

// run fn(arg) on 64 lanes in lockstep; fn reads wave_emul::lane()
inline void run(void (*fn)(void *), void *arg) {
  state &s = S();
  if (!s.stacks) s.stacks = (char *)malloc(STACK_BYTES * LANES);
  s.fn = fn;
  s.arg = arg;
  for (int l = 0; l < LANES; l++) {
    getcontext(&s.lane_ctx[l]);
    s.lane_ctx[l].uc_stack.ss_sp = s.stacks + STACK_BYTES * l;
    s.lane_ctx[l].uc_stack.ss_size = STACK_BYTES;
    s.lane_ctx[l].uc_link = nullptr;
    makecontext(&s.lane_ctx[l], (void (*)())trampoline, 0);
    s.alive[l] = true;
    s.phase[l] = 0;
    s.buf[0][l] = s.buf[1][l] = 0;
  }
  s.cur = 0;
  swapcontext(&s.main_ctx, &s.lane_ctx[0]);
  s.cur = -1;
}

}  // namespace wave_emul
