// Development-only: from the one-step kernel's streaming rate (6.15 TB/s) to the sweep's (5.2): one knob at a time.
// Every workgroup copies ROWS consecutive rows of n0 = 256 nodes of `seg` consecutive planes of Q populations;
// THREADS = 256 * ROWS.  NTL: nontemporal loads.  Stores are nontemporal.  No barrier unless BAR.
#include <hip/hip_runtime.h>

template <int Q, int ROWS, bool NTL, bool BAR, bool PREFETCH>
__global__ void __launch_bounds__(256 * ROWS) pop_copy(const float *__restrict__ in, float *__restrict__ out, int n1, int n2,
                                                       int seg, int plane_pad) {
  extern __shared__ float dummy[];
  constexpr int n0 = 256;
  const int tid = threadIdx.x;
  const int groups1 = n1 / ROWS;                       // workgroups per plane
  const int b = blockIdx.x;
  const int row0 = (b % groups1) * ROWS;
  const int s = (b / groups1) * seg;
  // plane_pad: floats between consecutive planes (a padded, engine-owned layout): does the fixed offset of a
  // workgroup's rows within every 256 KB plane pin it to a few memory channels?
  const size_t plane = (size_t)n0 * n1 + plane_pad, N = plane * n2;
  const size_t own = (size_t)row0 * n0 + tid;
  float cur[Q], nxt[Q];
  auto load = [&](int k, float (&r)[Q]) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float *p = in + q * N + k * plane + own;
      r[q] = NTL ? __builtin_nontemporal_load(p) : *p;
    }
  };
  load(s, cur);
  for (int k = s; k < s + seg; ++k) {
    if (PREFETCH && k + 1 < s + seg) load(k + 1, nxt);
    if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < Q; ++q) __builtin_nontemporal_store(cur[q], out + q * N + k * plane + own);
    if (PREFETCH) {
#pragma unroll
      for (int q = 0; q < Q; ++q) cur[q] = nxt[q];
    } else if (k + 1 < s + seg) {
      load(k + 1, cur);
    }
  }
}

template <int ROWS, bool NTL, bool BAR, bool PREFETCH>
static int go(const float *in, float *out, int n1, int n2, int seg, int lds, hipStream_t st, int plane_pad) {
  const unsigned grid = (unsigned)((n1 / ROWS) * (n2 / seg));
  (void)hipFuncSetAttribute((const void *)pop_copy<19, ROWS, NTL, BAR, PREFETCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((pop_copy<19, ROWS, NTL, BAR, PREFETCH>), dim3(grid), dim3(256 * ROWS), lds, st, in, out, n1, n2, seg, plane_pad);
  return (int)hipGetLastError();
}

// variant = rows * 100 + ntl * 10 + bar * 2 + prefetch
extern "C" int lt_pop_copy(int variant, const float *in, float *out, int n1, int n2, int seg, int lds, void *stream,
                           int plane_pad) {
  hipStream_t st = (hipStream_t)stream;
#define V(R, N, B, P) if (variant == R * 100 + N * 10 + B * 2 + P) return go<R, (N != 0), (B != 0), (P != 0)>(in, out, n1, n2, seg, lds, st, plane_pad);
  V(1, 0, 0, 0) V(1, 1, 0, 0) V(1, 0, 0, 1) V(1, 1, 0, 1) V(2, 0, 0, 0) V(2, 1, 0, 0) V(2, 0, 0, 1) V(2, 1, 0, 1) V(2, 1, 1, 1)
  V(4, 0, 0, 0) V(4, 1, 0, 0) V(4, 1, 0, 1) V(4, 1, 1, 1) V(4, 0, 1, 1)
  return -1;
}
