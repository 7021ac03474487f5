// tools/ubench_int.hip — per-instruction issue-rate microbenchmark for the integer /
// fp64 VALU ops a 256-bit modular multiply can be built from (gfx950).  Not part of the
// product; its output (profiles/ubench_int_r01.txt) justifies the limb representation
// chosen in csrc/secp256k1_dev.h and DESIGN.md's int-op ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define UNROLL 8

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3;
  uint32_t b0 = a0 ^ 0x1234567, b1 = a1 ^ 0x89abcde, b2 = a2 ^ 0x13579bd, b3 = a3 ^ 0x2468ace;
  uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = b0, q5 = b1, q6 = b2, q7 = b3;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = b0, d5 = b1, d6 = b2, d7 = b3;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) {  // v_mad_u64_u32: 8 independent chains
      asm volatile(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n"
          "v_mad_u64_u32 %2, vcc, %8, %11, %2\n v_mad_u64_u32 %3, vcc, %8, %12, %3\n"
          "v_mad_u64_u32 %4, vcc, %9, %10, %4\n v_mad_u64_u32 %5, vcc, %9, %11, %5\n"
          "v_mad_u64_u32 %6, vcc, %9, %12, %6\n v_mad_u64_u32 %7, vcc, %10, %11, %7\n"
          : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0)
          : "vcc");
    } else if (OP == 1) {  // v_mul_lo_u32
      asm volatile(
          "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
          "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
          : "v"(seed | 1));
    } else if (OP == 2) {  // v_mul_hi_u32
      asm volatile(
          "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
          "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
          : "v"(seed | 0xF0000001));
    } else if (OP == 3) {  // v_mad_u32_u24
      asm volatile(
          "v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n"
          "v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
          : "v"(seed | 1));
    } else if (OP == 4) {  // v_add_co_u32 + v_addc_co_u32 pairs (carry chain)
      asm volatile(
          "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n"
          "v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
          "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n"
          "v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
          : "v"(seed | 1)
          : "vcc");
    } else if (OP == 5) {  // v_fma_f64
      asm volatile(
          "v_fma_f64 %0, %0, %8, %1\n v_fma_f64 %1, %1, %8, %2\n v_fma_f64 %2, %2, %8, %3\n v_fma_f64 %3, %3, %8, %4\n"
          "v_fma_f64 %4, %4, %8, %5\n v_fma_f64 %5, %5, %8, %6\n v_fma_f64 %6, %6, %8, %7\n v_fma_f64 %7, %7, %8, %0\n"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
          : "v"(1.0000001));
    } else if (OP == 6) {  // v_add_u32 (plain full-rate reference)
      asm volatile(
          "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
          "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
          : "v"(seed | 1));
    } else if (OP == 7) {  // v_mul_u32_u24 + v_mul_hi_u32_u24
      asm volatile(
          "v_mul_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
          "v_mul_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
          : "v"(seed | 0xFFFFF1));
    } else if (OP == 8) {  // v_mad_u64_u32, ONE dependent chain (latency)
      asm volatile(
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
          : "+v"(q0)
          : "v"(a0), "v"(a1)
          : "vcc");
    } else if (OP == 9) {  // v_mul_f64
      asm volatile(
          "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
          "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
          : "v"(1.0000001));
    }
  }
  uint64_t acc = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
  double dd = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
  out[blockIdx.x * blockDim.x + threadIdx.x] =
      a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ (uint32_t)dd;
}

template <int OP>
void run(const char *name, int waves_per_simd, uint32_t *d_out) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d_out, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d_out, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double wave_instr_per_simd = (double)ITERS * UNROLL * waves_per_simd;  // each SIMD executes this many wave-instructions
  double ns_per = ms * 1e6 / wave_instr_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)  chip %.2f T lane-ops/s\n",
         name, waves_per_simd, ms, ns_per, ns_per * 2.4, 1024.0 * 64 / ns_per / 1e3);
}

int main() {
  uint32_t *d_out;
  hipMalloc(&d_out, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4}) {
    run<0>("v_mad_u64_u32 (8 chains)", w, d_out);
    run<8>("v_mad_u64_u32 (1 chain)", w, d_out);
    run<1>("v_mul_lo_u32", w, d_out);
    run<2>("v_mul_hi_u32", w, d_out);
    run<3>("v_mad_u32_u24", w, d_out);
    run<7>("v_mul(_hi)_u32_u24", w, d_out);
    run<4>("v_add_co/v_addc_co", w, d_out);
    run<6>("v_add_u32", w, d_out);
    run<5>("v_fma_f64", w, d_out);
    run<9>("v_mul_f64", w, d_out);
  }
  return 0;
}
