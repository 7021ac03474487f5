// tools/scatter_probe.hip — how fast does THIS lease serve scattered 80-byte reads from a large table?  (the
// kernels that differ 15–20 % between leases are the ones with tens of thousands of lanes each waiting on its own table read;
// the rows kernel, four signatures per wavefront, does not notice.)  Every lane performs `K` reads of 80 bytes (five 16-byte
// loads, like load_affine of a G-table entry) at pseudo-random 80-byte-aligned offsets of a table of `MB` megabytes, each
// address depending on the previous read (the way a window's table point is needed before the next addition can finish), with
// W wavefronts on the chip.  Prints ns per read as one lane sees it.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/scatter_probe tools/scatter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(64) probe(const uint4 *__restrict__ tab, uint64_t entries, int K, uint32_t seed, uint32_t *out) {
  uint64_t x = (uint64_t)(blockIdx.x * 64u + threadIdx.x) * 0x9E3779B97F4A7C15ull + seed;
  uint32_t acc = 0;
  for (int k = 0; k < K; k++) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint4 *e = tab + (x % entries) * 5;
    const uint4 a = e[0], b = e[1], c = e[2], d = e[3], f = e[4];
    acc += a.x ^ b.y ^ c.z ^ d.w ^ f.x;
    x += acc;  // the next address waits for this read
  }
  out[blockIdx.x * 64u + threadIdx.x] = acc;
}

// ---- instruction fetch: the same dependent-free VALU stream from a loop body that fits the 64 KB instruction cache (16 KB) and
// from one that does not (128 KB, 512 KB): a lone wavefront per SIMD pays for every fetch (the lane-layout kernels inline
// 224-instruction multiplications: their loop bodies are tens of KB, the row-layout kernels' 15 KB)
#define REP8(x) x x x x x x x x
#define BLK64 REP8(REP8("v_add_u32 %0, %0, %1\n"))   /* 64 instructions = 256 B */
#define K2 REP8(BLK64)                                  /* 2 KB */
#define K4 K2 K2
#define K8 K4 K4
#define K16 K8 K8
#define K32 K16 K16
#define K64 K32 K32
template <int KB>
__global__ void __launch_bounds__(64) ifetch(uint32_t *out, uint32_t seed, int iters) {
  uint32_t a = seed + threadIdx.x;
  for (int i = 0; i < iters; i++) {
    if (KB == 16) asm volatile(K16 : "+v"(a) : "v"(seed | 1));
    else if (KB == 24) asm volatile(K16 K8 : "+v"(a) : "v"(seed | 1));
    else if (KB == 32) asm volatile(K32 : "+v"(a) : "v"(seed | 1));
    else if (KB == 40) asm volatile(K32 K8 : "+v"(a) : "v"(seed | 1));
    else if (KB == 48) asm volatile(K32 K16 : "+v"(a) : "v"(seed | 1));
    else if (KB == 64) asm volatile(K64 : "+v"(a) : "v"(seed | 1));
    else if (KB == 96) asm volatile(K64 K32 : "+v"(a) : "v"(seed | 1));
    else if (KB == 128) asm volatile(K64 K64 : "+v"(a) : "v"(seed | 1));
    else asm volatile(K64 K64 K64 K64 K64 K64 K64 K64 : "+v"(a) : "v"(seed | 1));   // 512 KB
  }
  out[blockIdx.x * 64u + threadIdx.x] = a;
}
template <int KB>
static void run_ifetch(uint32_t *out, double instr_per_iter) {
  const int iters = (int)(4.0e6 / instr_per_iter) + 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {1024, 2048}) {
    for (int i = 0; i < 6; i++) ifetch<KB><<<waves, 64>>>(out, 3u + i, iters);
    hipDeviceSynchronize();
    float t[5];
    for (int i = 0; i < 5; i++) {
      hipEventRecord(e0);
      ifetch<KB><<<waves, 64>>>(out, 9u + i, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&t[i], e0, e1);
    }
    std::sort(t, t + 5);
    printf("loop body %7.0f KB   wavefronts %5d   %6.3f ns per wave-instruction per SIMD\n", instr_per_iter * 4 / 1024, waves,
           t[2] * 1e6 / (instr_per_iter * iters) / (waves / 1024.0));
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s; ns per 80-byte read as one lane sees it (dependent reads, K = 64 per lane), median of 9 launches behind 20 untimed ones\n", prop.gcnArchName);
  printf("# table MB   wavefronts   ns/read\n");
  uint32_t *out;
  hipMalloc(&out, 4096 * 64 * 4);
  printf("# instruction fetch: a v_add_u32 stream, one / two wavefronts per SIMD\n");
  run_ifetch<16>(out, 16 * 256.0);
  run_ifetch<24>(out, 24 * 256.0);
  run_ifetch<32>(out, 32 * 256.0);
  run_ifetch<40>(out, 40 * 256.0);
  run_ifetch<48>(out, 48 * 256.0);
  run_ifetch<64>(out, 64 * 256.0);
  run_ifetch<96>(out, 96 * 256.0);
  run_ifetch<128>(out, 128 * 256.0);
  run_ifetch<512>(out, 512 * 256.0);
  printf("# scattered reads\n");
  for (size_t mb : {84ull, 2700ull}) {
    const uint64_t entries = mb * 1000000ull / 80;
    uint4 *tab;
    if (hipMalloc(&tab, entries * 80) != hipSuccess) { printf("%zu MB: allocation failed\n", mb); continue; }
    hipMemset(tab, 1, entries * 80);
    for (int waves : {64, 256, 1024, 4096}) {
      const int K = 64;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 20; i++) probe<<<waves, 64>>>(tab, entries, K, 17u + i, out);
      hipDeviceSynchronize();
      float t[9];
      for (int i = 0; i < 9; i++) {
        hipEventRecord(e0);
        probe<<<waves, 64>>>(tab, entries, K, 99u + i, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t[i], e0, e1);
      }
      std::sort(t, t + 9);
      printf("%8zu   %8d   %8.1f\n", mb, waves, t[4] * 1e6 / K);
    }
    hipFree(tab);
  }
  return 0;
}
