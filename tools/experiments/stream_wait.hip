// Does a stream wait on a counter that the workgroups of a RUNNING kernel of another stream increment?
// (hipStreamWaitValue64 on signal memory; the slab driver's "edges first, exchange while the sweep goes on").
// Build: hipcc --offload-arch=gfx950 -O2 tools/experiments/stream_wait.hip -o /tmp/stream_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void worker(unsigned long long *signal, long long spin_early, long long spin_late, unsigned long long *t_signal) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_early) {}
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(signal, 1ull);
    t_signal[8 + blockIdx.x] = (unsigned long long)wall_clock64();
    if (blockIdx.x == 0) *t_signal = (unsigned long long)wall_clock64();
  }
  while (wall_clock64() - t0 < spin_late) {}
}
// one wave polling the counter (sleeping between polls), with a way out: 1 s
__global__ void spin_wait(const unsigned long long *signal, unsigned long long target, unsigned *timed_out) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(signal, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 100000000ll) { *timed_out = 1u; break; }
  }
}
__global__ void mark(const unsigned long long *signal, unsigned long long *seen, unsigned long long *t_mark) {
  *seen = *signal;
  *t_mark = (unsigned long long)wall_clock64();
}

int main(int argc, char **argv) {
  const long long early = argc > 1 ? atoll(argv[1]) : 20000ll;
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("{\"can_use_stream_wait_value\": %d", can);
  if (!can) { printf("}\n"); return 0; }
  unsigned long long *signal, *out;
  CK(hipExtMallocWithFlags((void **)&signal, 8, hipMallocSignalMemory));
  CK(hipMalloc((void **)&out, 4 * 8 + 8 * 8 + 256 * 8));
  CK(hipMemset(signal, 0, 8));
  CK(hipMemset(out, 0, 32));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  const int blocks = 256;
  // wall_clock64 ticks at 100 MHz: signal after 0.2 ms, leave after 2 ms
  for (int round = 1; round <= 3; ++round) {
    hipLaunchKernelGGL(worker, dim3(blocks), dim3(256), 0, a, signal, early, 200000ll, out + 0);
    CK(hipStreamWaitValue64(b, signal, (uint64_t)round * blocks, hipStreamWaitValueGte, ~0ull));
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, b, signal, out + 1, out + 2);
    CK(hipStreamSynchronize(b));
    unsigned long long h[3];
    CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
    CK(hipStreamSynchronize(a));
    unsigned long long t_end;
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, a, signal, out + 3, out + 3);
    CK(hipStreamSynchronize(a));
    CK(hipMemcpy(&t_end, out + 3, 8, hipMemcpyDeviceToHost));
    { unsigned long long ts[256]; CK(hipMemcpy(ts, out + 8, 256 * 8, hipMemcpyDeviceToHost));
      unsigned long long lo = ts[0], hi = ts[0]; for (int i = 0; i < 256; ++i) { if (ts[i] < lo) lo = ts[i]; if (ts[i] > hi) hi = ts[i]; }
      printf(", \"signals_spread_us_round%d\": %.1f, \"mark_after_latest_signal_us\": %.1f", round, ((double)hi - (double)lo) / 100.0, ((double)h[2] - (double)hi) / 100.0); }
    printf(", \"round%d\": {\"seen\": %llu, \"target\": %d, \"mark_after_last_signal_us\": %.1f, \"mark_before_kernel_end_us\": %.1f}",
           round, h[1], round * blocks, ((double)h[2] - (double)h[0]) / 100.0, ((double)t_end - (double)h[2]) / 100.0);
  }
  // the same with a polling kernel instead of the command processor's wait, counter in ordinary device memory
  unsigned long long *plain;
  CK(hipMalloc((void **)&plain, 8));
  CK(hipMemset(plain, 0, 8));
  for (int round = 1; round <= 3; ++round) {
    hipLaunchKernelGGL(worker, dim3(blocks), dim3(256), 0, a, plain, early, 200000ll, out + 0);
    hipLaunchKernelGGL(spin_wait, dim3(1), dim3(64), 0, b, plain, (unsigned long long)round * blocks, (unsigned *)(out + 4));
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, b, plain, out + 1, out + 2);
    CK(hipStreamSynchronize(b));
    unsigned long long h[3];
    CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
    CK(hipStreamSynchronize(a));
    printf(", \"plain_spin_round%d\": {\"seen\": %llu, \"target\": %d, \"mark_after_signal_us\": %.1f}",
           round, h[1], round * blocks, ((double)h[2] - (double)h[0]) / 100.0);
  }
  unsigned *flag;
  CK(hipMalloc((void **)&flag, 4));
  CK(hipMemset(flag, 0, 4));
  for (int round = 4; round <= 6; ++round) {
    hipLaunchKernelGGL(worker, dim3(blocks), dim3(256), 0, a, signal, early, 200000ll, out + 0);
    hipLaunchKernelGGL(spin_wait, dim3(1), dim3(64), 0, b, signal, (unsigned long long)round * blocks, flag);
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, b, signal, out + 1, out + 2);
    CK(hipStreamSynchronize(b));
    unsigned long long h[3];
    CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
    CK(hipStreamSynchronize(a));
    printf(", \"spin_round%d\": {\"seen\": %llu, \"target\": %d, \"mark_after_last_signal_us\": %.1f}",
           round, h[1], round * blocks, ((double)h[2] - (double)h[0]) / 100.0);
  }
  unsigned hf = 0;
  CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
  printf(", \"timed_out\": %u}\n", hf);
  return 0;
}
