// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950 (ROCm 7.2)?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards".  Two single-workgroup kernels that each spin ~100 us:
// serial = ~200 us, overlapped = ~100 us.   hipcc --offload-arch=gfx950 -O2 any_order.hip -o any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
  if (out) *out = 1;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int *d; hipMalloc(&d, 8);
  const long long ticks = 10000;   // 100 MHz wall clock: 100 us
  for (int flags = 0; flags < 2; ++flags)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks, d);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, ticks, d + 1);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("flags=%d: two 100-us kernels back to back took %.1f us\n", flags, ms * 1e3);
    }
  return 0;
}
