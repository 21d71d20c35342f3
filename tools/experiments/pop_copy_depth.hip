// Development-only: the population copy as a sweep (persistent workgroups, `seg` planes each) with DEPTH planes of
// loads in flight per wave: is the sweep slower than short-lived workgroups because a wave's loads queue behind its
// own stores (in-order vmcnt retirement)?  512 threads = 2 rows of 256 nodes; nontemporal loads and stores.
#include <hip/hip_runtime.h>

template <int Q, int DEPTH, bool BAR>
__global__ void __launch_bounds__(512) pop_copy_depth(const float *__restrict__ in, float *__restrict__ out, int n1, int n2, int seg) {
  extern __shared__ float dummy[];
  constexpr int n0 = 256, ROWS = 2;
  const int tid = threadIdx.x;
  const int groups1 = n1 / ROWS;
  const int b = blockIdx.x;
  const int row0 = (b % groups1) * ROWS;
  const int s = (b / groups1) * seg;
  const size_t plane = (size_t)n0 * n1, N = plane * n2;
  const size_t own = (size_t)row0 * n0 + tid;
  float buf[DEPTH][Q];
  auto load = [&](int k, float (&r)[Q]) {
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = __builtin_nontemporal_load(in + q * N + k * plane + own);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (s + d < s + seg) load(s + d, buf[d]);
  // seg is a multiple of DEPTH: the loop is unrolled DEPTH times so that the buffers are static
  for (int k = s; k < s + seg; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int q = 0; q < Q; ++q) __builtin_nontemporal_store(buf[d][q], out + q * N + (k + d) * plane + own);
      if (k + d + DEPTH < s + seg) load(k + d + DEPTH, buf[d]);
    }
  }
}

extern "C" int lt_pop_copy_depth(int depth, int bar, const float *in, float *out, int n1, int n2, int seg, int lds, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((n1 / 2) * (n2 / seg));
#define V(D, B)                                                                                                        \
  if (depth == D && bar == B) {                                                                                        \
    (void)hipFuncSetAttribute((const void *)pop_copy_depth<19, D, (B != 0)>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((pop_copy_depth<19, D, (B != 0)>), dim3(grid), dim3(512), lds, st, in, out, n1, n2, seg);         \
    return (int)hipGetLastError();                                                                                     \
  }
  V(1, 0) V(1, 1) V(2, 0) V(2, 1) V(4, 0) V(4, 1)
  return -1;
}
