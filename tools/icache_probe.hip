// tools/icache_probe.hip — what does a launch pay for walking code the instruction cache has not seen, and does ANY way of
// launching keep the cache warm from one launch to the next?  (the headline kernel walks 46 KB of code once per
// launch; `tools/rows_stages.py` prices a pass over warm code 0.04 ms below the first pass of a launch.)
// The kernel is KB kilobytes of straight-line, dependent-free `v_add_u32` (4-byte instructions), walked `passes` times by one
// wavefront per SIMD (256 workgroups of 256 lanes, 100 KB of LDS each so that a second workgroup does not fit a compute unit).
//   (a) one launch between two events, passes = 1, 2, 3: pass 2 and 3 find the code in the cache
//   (b) 40 launches back to back on one stream, nothing between them
//   (c) the same with a small other kernel between them (the product's tally kernel)
//   (d) the 40 launches as one hipGraph
//   (e) hipExtLaunchKernelGGL with hipExtAnyOrderLaunch (no barrier bit between the dispatches)
// each next to the same series of an EMPTY walk (passes = 0), so that launch overhead cancels.
//   build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/icache_probe tools/icache_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define BLK64 REP8(REP8("v_add_u32 %0, %0, %1\n")) /* 64 instructions = 256 B */
#define K2 REP8(BLK64)                                /* 2 KB */
#define K4 K2 K2
#define K8 K4 K4
#define K16 K8 K8
#define K32 K16 K16

template <int KB>
__global__ void __launch_bounds__(256) walk(uint32_t *out, uint32_t seed, int passes) {
  extern __shared__ uint32_t lds[];
  uint32_t a = seed + threadIdx.x;
  for (int p = 0; p < passes; p++) {
    if (KB == 16) asm volatile(K16 : "+v"(a) : "v"(seed | 1));
    else if (KB == 32) asm volatile(K32 : "+v"(a) : "v"(seed | 1));
    else asm volatile(K32 K16 : "+v"(a) : "v"(seed | 1));
  }
  if (a == 0x12345u) lds[threadIdx.x] = a;   // keeps the LDS allocation alive
  out[blockIdx.x * 256u + threadIdx.x] = a;
}
__global__ void small_other(uint32_t *out) { out[threadIdx.x] += 1; }

static const int LDS_BYTES = 100 * 1024, BLOCKS = 256, K = 40;
static hipStream_t st;
static uint32_t *out;

template <int KB>
static void launch(int passes, bool any_order = false) {
  if (any_order)
    hipExtLaunchKernelGGL((walk<KB>), dim3(BLOCKS), dim3(256), LDS_BYTES, st, nullptr, nullptr, hipExtAnyOrderLaunch, out, 7u, passes);
  else
    walk<KB><<<BLOCKS, 256, LDS_BYTES, st>>>(out, 7u, passes);
}
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int KB>
static double series(int passes, int mode) {   // µs per launch, median of 7 series of K launches
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  if (mode == 3) {
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < K; i++) launch<KB>(passes);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  }
  std::vector<double> us;
  for (int rep = 0; rep < 9; rep++) {
    hipEventRecord(e0, st);
    if (mode == 3) hipGraphLaunch(ge, st);
    else
      for (int i = 0; i < K; i++) {
        launch<KB>(passes, mode == 4);
        if (mode == 2) small_other<<<1, 64, 0, st>>>(out);
      }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) us.push_back(ms * 1e3 / K);
  }
  if (ge) hipGraphExecDestroy(ge);
  if (g) hipGraphDestroy(g);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return median(us);
}
template <int KB>
static double single(int passes) {             // µs of ONE launch between two events, median of 9
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> us;
  for (int rep = 0; rep < 11; rep++) {
    hipEventRecord(e0, st);
    launch<KB>(passes);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) us.push_back(ms * 1e3);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return median(us);
}

template <int KB>
static void run() {
  hipFuncSetAttribute(reinterpret_cast<const void *>(&walk<KB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const double n_inst = KB * 1024.0 / 4.0;
  double s0 = single<KB>(0), s1 = single<KB>(1), s2 = single<KB>(2), s3 = single<KB>(3);
  const double warm = s3 - s2, cold = s1 - s0;
  printf("code %2d KB (%5.0f instructions per pass)  one launch between events: empty %6.2f us, 1 pass %6.2f, 2 passes %6.2f, 3 passes %6.2f\n",
         KB, n_inst, s0, s1, s2, s3);
  printf("   -> a pass over warm code %6.2f us (%.2f ns per instruction), the first pass of a launch %6.2f us: cold walk costs %+6.2f us = %.0f ns per 64-byte line\n",
         warm, warm * 1e3 / n_inst, cold, cold - warm, (cold - warm) * 1e3 / (KB * 16.0));
  const char *names[] = {"", "40 launches back to back", "40 launches, a small other kernel between", "40 launches as one hipGraph", "40 launches, hipExtAnyOrderLaunch"};
  for (int mode = 1; mode <= 4; mode++) {
    double e = series<KB>(0, mode), w = series<KB>(1, mode);
    printf("   %-44s empty %6.2f us per launch, 1 pass %6.2f: pass = %6.2f us -> %s (%+.2f us over warm)\n", names[mode], e, w, w - e,
           (w - e) < warm + 0.35 * (cold - warm) ? "code stays WARM between launches" : "code is COLD at every launch", w - e - warm);
  }
}

int main() {
  hipStreamCreate(&st);
  hipMalloc(&out, BLOCKS * 256 * 4);
  hipMemset(out, 0, BLOCKS * 256 * 4);
  run<16>();
  run<32>();
  run<48>();
  hipDeviceSynchronize();
  return 0;
}
