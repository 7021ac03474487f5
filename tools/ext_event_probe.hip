// tools/ext_event_probe.hip — what does an event cost between two kernels of one stream, and are the start / stop events that
// hipExtLaunchKernelGGL attaches to a dispatch cheaper than hipEventRecord markers around it?  40 launches back to back of a small
// kernel (16 KB of v_add_u32, one wavefront per SIMD, ≈ 7 µs): (a) nothing between them, (b) hipEventRecord (timing disabled)
// after each, (c) a timing pair hipEventRecord before / after each, (d) hipExtLaunchKernelGGL with a stop event, (e) with start
// and stop events; and what (c) / (e) read as the kernel's duration.
//   build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/ext_event_probe tools/ext_event_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP8(x) x x x x x x x x
#define BLK64 REP8(REP8("v_add_u32 %0, %0, %1\n"))
#define K2 REP8(BLK64)
#define K16 K2 K2 K2 K2 K2 K2 K2 K2
__global__ void __launch_bounds__(256) walk(uint32_t *out, uint32_t seed) {
  extern __shared__ uint32_t lds[];
  uint32_t a = seed + threadIdx.x;
  asm volatile(K16 : "+v"(a) : "v"(seed | 1));
  if (a == 0x12345u) lds[threadIdx.x] = a;
  out[blockIdx.x * 256u + threadIdx.x] = a;
}
static const int LDS = 100 * 1024, BLOCKS = 256, K = 40;
int main() {
  hipStream_t st; hipStreamCreate(&st);
  uint32_t *out; hipMalloc(&out, BLOCKS * 256 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&walk), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  std::vector<hipEvent_t> ev(2 * K), evn(K);
  for (auto &e : ev) hipEventCreate(&e);
  for (auto &e : evn) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
  const char *names[] = {"nothing between the launches", "hipEventRecord (no timing) after each", "timing pair of hipEventRecord around each",
                         "hipExtLaunchKernelGGL, stop event", "hipExtLaunchKernelGGL, start + stop events"};
  for (int mode = 0; mode < 5; mode++) {
    std::vector<double> us; double kdur = 0;
    for (int rep = 0; rep < 9; rep++) {
      hipEventRecord(t0, st);
      for (int i = 0; i < K; i++) {
        if (mode == 2) hipEventRecord(ev[2 * i], st);
        if (mode >= 3) hipExtLaunchKernelGGL(walk, dim3(BLOCKS), dim3(256), LDS, st, mode == 4 ? ev[2 * i] : nullptr, ev[2 * i + 1], 0, out, 7u);
        else walk<<<BLOCKS, 256, LDS, st>>>(out, 7u);
        if (mode == 1) hipEventRecord(evn[i], st);
        if (mode == 2) hipEventRecord(ev[2 * i + 1], st);
      }
      hipEventRecord(t1, st); hipEventSynchronize(t1);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      if (rep >= 2) us.push_back(ms * 1e3 / K);
      if (mode == 2 || mode == 4) { float k; hipEventElapsedTime(&k, ev[20], ev[21]); kdur = k * 1e3; }
    }
    std::sort(us.begin(), us.end());
    printf("%-46s %6.2f us per launch", names[mode], us[us.size() / 2]);
    if (kdur > 0) printf("   (the pair reads %.2f us for one kernel)", kdur);
    printf("\n");
  }
  return 0;
}
