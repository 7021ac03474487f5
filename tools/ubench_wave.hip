// tools/ubench_wave.hip — issue-rate microbenchmark for the instructions the row-layout field
// multiplication (csrc/wave_fe_dev.h:wfe_mul) is made of, at 1 / 2 / 4 wavefronts per SIMD.
// Not part of the product; its output (profiles/r02_ubench_wave.txt) is what DESIGN.md's VALU-issue
// ceiling for the one-wavefront-per-SIMD kernels is derived from.
//
// Every test issues long unrolled runs (loop overhead < 2 %), reports shader cycles per wave-instruction
// from s_memtime (clock64) and the wall time from HIP events, i.e. also the clock the chip really ran at.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I go-ibft_amd/csrc -o tools/ubench_wave tools/ubench_wave.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>

#include "wave_fe_dev.h"

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

constexpr int ITERS = 2048;  // × 64 unrolled instructions: ≈0.3 ms per launch at one wavefront per SIMD (round 5: was 256 — 34 µs kernels
                             // priced with HIP events carried the launch ramp; the ceilings bench.py derives were beaten by the kernel)

template <int OP>
__global__ void __launch_bounds__(256) k_inst(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3;
  uint32_t b0 = a0 ^ 0x1234567, b1 = a1 ^ 0x89abcde, b2 = a2 ^ 0x13579bd, b3 = a3 ^ 0x2468ace;
  uint64_t q0 = a0, q1 = a1;
  const uint64_t t0 = clock64();
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) {  // plain VALU, independent
      asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
    } else if (OP == 1) {  // plain VALU, ONE dependent chain
      asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(seed | 1));
    } else if (OP == 2) {  // DPP row_shr moves of an unchanging source (as in wfe_mul: ten shifts of b)
      asm volatile(REP8("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %1, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %8 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %8 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %4, %8 row_shr:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %5, %8 row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %6, %8 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %7, %8 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
    } else if (OP == 3) {  // DPP row_newbcast
      asm volatile(REP8("v_mov_b32_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
    } else if (OP == 4) {  // v_mad_u64_u32 accumulating into ONE register pair (wfe_mul's column sum)
      asm volatile(REP8("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n"
                        "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %1, %4, %0\n"
                        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n"
                        "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %1, %4, %0\n")
                   : "+v"(q0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
    } else if (OP == 5) {  // the inner pattern of wfe_mul: bcast, shr, mad — 24 instructions per group of 8 columns
      asm volatile(REP8("v_mov_b32_dpp %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                        "v_mov_b32_dpp %5, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %6, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n"
                        "v_mov_b32_dpp %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %4 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                   : "+v"(q0), "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3), "v"(b0), "v"(b1) : "vcc");
    } else if (OP == 6) {  // ds_bpermute_b32, independent
      asm volatile(REP8("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n"
                        "ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n"
                        "ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
                   : "v"((threadIdx.x * 4u + 4u) & 255u));
    } else if (OP == 7) {  // VALU op with a DPP operand (v_add_u32_dpp): is the DPP form itself dearer?
      asm volatile(REP8("v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %1, %8, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %2, %8, %2 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %3, %8, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %4, %8, %4 row_shr:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %5, %8, %5 row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %6, %8, %6 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_u32_dpp %7, %8, %7 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
    } else if (OP == 8) {  // v_mad_u64_u32, two independent accumulators alternating
      asm volatile(REP8("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %1, vcc, %2, %5, %1\n"
                        "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %1, vcc, %2, %5, %1\n")
                   : "+v"(q0), "+v"(q1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
    } else if (OP == 9) {  // SALU only
      uint32_t s0 = seed, s1 = seed + 1;
      asm volatile(REP64("s_add_u32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc");
      a0 ^= s0;
    } else if (OP >= 11 && OP <= 16) {
      // Round 5: the classes go-ibft_amd/phase_align.py distinguishes.  A run of 64 eight-byte instructions that starts on an
      // 8-byte boundary (ALN0) or 4 bytes past one (ALN4: every instruction straddles two aligned fetch units), for the three
      // eight-byte kinds of the kernels' hot loops: a plain VALU op in its VOP3 form, a DPP move, v_mad_u64_u32.
#define ALN0 ".p2align 3\n"
#define ALN4 ".p2align 3\n s_nop 0\n"
#define RUN_E64 REP8("v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n" \
                     "v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n")
#define RUN_DPP REP8("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"      \
                     "v_mov_b32_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                     "v_mov_b32_dpp %2, %8 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"      \
                     "v_mov_b32_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                     "v_mov_b32_dpp %4, %8 row_shr:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"      \
                     "v_mov_b32_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                     "v_mov_b32_dpp %6, %8 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"      \
                     "v_mov_b32_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
#define RUN_MAD REP8("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n" \
                     "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %1, vcc, %2, %5, %1\n" \
                     "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n" \
                     "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %1, vcc, %2, %5, %1\n")
      if (OP == 11)
        asm volatile(ALN0 RUN_E64 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
      else if (OP == 12)
        asm volatile(ALN4 RUN_E64 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
      else if (OP == 13)
        asm volatile(ALN0 RUN_DPP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
      else if (OP == 14)
        asm volatile(ALN4 RUN_DPP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
      else if (OP == 15)
        asm volatile(ALN0 RUN_MAD : "+v"(q0), "+v"(q1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
      else
        asm volatile(ALN4 RUN_MAD : "+v"(q0), "+v"(q1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
    } else if (OP == 17) {  // the 4-byte plain op with an explicit start on an 8-byte boundary (pairs share a fetch unit)
      asm volatile(".p2align 3\n" REP8("v_add_u32_e32 %0, %0, %8\n v_add_u32_e32 %1, %1, %8\n v_add_u32_e32 %2, %2, %8\n v_add_u32_e32 %3, %3, %8\n"
                        "v_add_u32_e32 %4, %4, %8\n v_add_u32_e32 %5, %5, %8\n v_add_u32_e32 %6, %6, %8\n v_add_u32_e32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
    } else if (OP == 10) {  // v_mul_u32_u24 + v_mad_u32_u24 pairs (a 24-bit limb alternative)
      asm volatile(REP8("v_mad_u32_u24 %0, %8, %1, %0\n v_mad_u32_u24 %1, %8, %2, %1\n v_mad_u32_u24 %2, %8, %3, %2\n"
                        "v_mad_u32_u24 %3, %8, %4, %3\n v_mad_u32_u24 %4, %8, %5, %4\n v_mad_u32_u24 %5, %8, %6, %5\n"
                        "v_mad_u32_u24 %6, %8, %7, %6\n v_mad_u32_u24 %7, %8, %0, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed | 1));
    }
  }
  const uint64_t t1 = clock64();
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  out[g] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (uint32_t)q0 ^ (uint32_t)(q0 >> 32) ^ (uint32_t)q1;
  if ((threadIdx.x & 63) == 0) cyc[g >> 6] = t1 - t0;
}

// the multiplication itself: CHAINS independent dependent-chains of wfe_mul<true> in one wavefront
template <int CHAINS>
__global__ void __launch_bounds__(256) k_mul(uint32_t *out, uint64_t *cyc, uint32_t seed, int iters) {
  const wv::wk k = wv::wk_init();
  uint32_t x[CHAINS], y = (seed * 2654435761u + threadIdx.x * 40503u) & 0x3FFFFFFu & k.act;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) x[c] = ((seed + c) * 2246822519u + threadIdx.x * 7919u) & 0x3FFFFFFu & k.act;
  const uint64_t t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = wv::wfe_mul<true>(x[c], y, k);
  }
  const uint64_t t1 = clock64();
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc ^= x[c];
  out[g] = acc;
  if ((threadIdx.x & 63) == 0) cyc[g >> 6] = t1 - t0;
}
// same through the outlined copy (call + return + argument moves per multiplication)
__global__ void __launch_bounds__(256) k_mul_call(uint32_t *out, uint64_t *cyc, uint32_t seed, int iters) {
  const wv::wk k = wv::wk_init();
  uint32_t x = (seed * 2246822519u + threadIdx.x * 7919u) & 0x3FFFFFFu & k.act;
  const uint32_t y = (seed * 2654435761u + threadIdx.x * 40503u) & 0x3FFFFFFu & k.act;
  const uint64_t t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < iters; i++) x = wv::wfe_mul<false>(x, y, k);
  const uint64_t t1 = clock64();
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  out[g] = x;
  if ((threadIdx.x & 63) == 0) cyc[g >> 6] = t1 - t0;
}

static uint32_t *d_out;
static uint64_t *d_cyc;

template <class F>
static void timed(const char *name, int w, double units_per_wave, const char *unit, F launch) {
  const int blocks = 256 * w;  // 256 threads = 4 wavefronts = one per SIMD of a CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // untimed launches until ≥ 30 ms of this kernel have run (the governor's ramp: DESIGN §5.1), then the MEDIAN of 7 launches
  {
    hipEventRecord(e0);
    float spent = 0;
    for (int i = 0; i < 400 && spent < 30.f; i++) {
      launch(blocks);
      if ((i & 7) == 7) {
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&spent, e0, e1);
      }
    }
    hipDeviceSynchronize();
  }
  float t[7];
  for (int i = 0; i < 7; i++) {
    hipEventRecord(e0);
    launch(blocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[i], e0, e1);
  }
  std::sort(t, t + 7);
  const float ms = t[3];
  std::vector<uint64_t> cyc((size_t)blocks * 4);
  hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (uint64_t c : cyc) sum += (double)c;
  const double wave_cycles = sum / cyc.size();
  // s_memtime ticks per wave ÷ units = ticks per unit as ONE wave sees it; × 1/w = per SIMD issue slot
  printf("%-44s waves/SIMD=%d  %8.3f ms  %9.0f ticks/wave  %7.2f ticks per %s per wave  %6.2f per SIMD  (wall: %.2f ns per %s per SIMD)\n",
         name, w, ms, wave_cycles, wave_cycles / units_per_wave, unit, wave_cycles / units_per_wave / w,
         ms * 1e6 / (units_per_wave * w), unit);
}

int main() {
  hipMalloc(&d_out, 256 * 8 * 256 * 4);
  hipMalloc(&d_cyc, 256 * 8 * 4 * 8);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s, clockRate %d kHz; ticks = s_memtime (clock64)\n", prop.gcnArchName, prop.clockRate);
  const double n_inst = (double)ITERS * 64;
  for (int w : {1, 2, 4}) {
#define RUN_INST(OP, NAME, UNITS) timed(NAME, w, UNITS, "inst", [&](int b) { k_inst<OP><<<b, 256>>>(d_out, d_cyc, 12345u); })
    RUN_INST(17, "class 4-byte: v_add_u32_e32", n_inst);
    RUN_INST(11, "class 8-byte plain aligned: v_add_u32_e64", n_inst);
    RUN_INST(12, "class 8-byte plain at 4 mod 8: v_add_u32_e64", n_inst);
    RUN_INST(13, "class DPP aligned: v_mov_b32_dpp", n_inst);
    RUN_INST(14, "class DPP at 4 mod 8: v_mov_b32_dpp", n_inst);
    RUN_INST(15, "class mad aligned: v_mad_u64_u32", n_inst);
    RUN_INST(16, "class mad at 4 mod 8: v_mad_u64_u32", n_inst);
    RUN_INST(0, "v_add_u32, 8 independent", n_inst);
    RUN_INST(1, "v_add_u32, one dependent chain", n_inst);
    RUN_INST(2, "v_mov_b32_dpp row_shr", n_inst);
    RUN_INST(3, "v_mov_b32_dpp row_newbcast", n_inst);
    RUN_INST(7, "v_add_u32_dpp row_shr", n_inst);
    RUN_INST(4, "v_mad_u64_u32, one accumulator", n_inst);
    RUN_INST(8, "v_mad_u64_u32, two accumulators", n_inst);
    RUN_INST(10, "v_mad_u32_u24", n_inst);
    RUN_INST(5, "bcast+shr+mad pattern (wfe_mul inner)", n_inst);
    RUN_INST(6, "ds_bpermute_b32 (8 in flight)", n_inst);
    RUN_INST(9, "s_add_u32 chain (SALU)", n_inst);
    const int iters = 4096;
    timed("wfe_mul<inline>, 1 chain", w, iters, "mul", [&](int b) { k_mul<1><<<b, 256>>>(d_out, d_cyc, 7u, iters); });
    timed("wfe_mul<inline>, 2 independent chains", w, 2.0 * iters, "mul", [&](int b) { k_mul<2><<<b, 256>>>(d_out, d_cyc, 7u, iters); });
    timed("wfe_mul<inline>, 4 independent chains", w, 4.0 * iters, "mul", [&](int b) { k_mul<4><<<b, 256>>>(d_out, d_cyc, 7u, iters); });
    timed("wfe_mul<call>, 1 chain", w, iters, "mul", [&](int b) { k_mul_call<<<b, 256>>>(d_out, d_cyc, 7u, iters); });
  }
  return 0;
}
