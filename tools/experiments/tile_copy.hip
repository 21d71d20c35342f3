// Development-only microbenchmark: what does the memory system make of the two-step kernel's access pattern?
// A workgroup owns a T0 x T1 tile in (a0, a1) and sweeps planes along a2, copying Q populations plane by plane
// (loads of plane k + 1 issued before the stores of plane k, optional barrier per plane).  MODE bit 0: read,
// bit 1: write.  TILED: the tile's T0*T1 nodes of a (population, plane) are contiguous (a tile-major layout)
// instead of T1 row segments of T0 nodes.  HALO: also read the two neighbouring rows (as the real kernel does).
#include <hip/hip_runtime.h>

template <int Q, int T0, int T1, int MODE, int TILED, bool BARRIER, bool HALO>
__global__ void __launch_bounds__(T0 * T1) tile_copy(const float *__restrict__ in, float *__restrict__ out, int n0, int n1,
                                                      int n2, int seg, int skew, long long pstride) {
  extern __shared__ float dummy[];
  const int tid = threadIdx.x;
  const int tiles0 = n0 / T0, tiles1 = n1 / T1;
  int b = blockIdx.x;
  if (gridDim.x % 8 == 0) b = (b % 8) * (gridDim.x / 8) + b / 8;
  const int tile = b % (tiles0 * tiles1);
  const int t0 = (tile % tiles0) * T0, t1 = (tile / tiles0) * T1;
  // skew: 0 all workgroups sweep in step; 1: every XCD patch starts 'skew' planes after the previous one; 2: every tile
  const int xcd = blockIdx.x % 8;
  const int s = (b / (tiles0 * tiles1)) * seg + (skew > 0 ? xcd * skew : (skew < 0 ? (tile * -skew) % n2 : 0));
  const int j1 = tid / T0, j0 = tid - j1 * T0;
  const size_t plane = (size_t)n0 * n1, N = (size_t)pstride;
  // TILED 2: [q][tile][plane][T0*T1]: a workgroup's sweep is one contiguous stream per population
  const size_t own = TILED == 1 ? (size_t)tile * (T0 * T1) + tid : (TILED == 2 ? (size_t)tile * n2 * (T0 * T1) + tid : (size_t)(t1 + j1) * n0 + t0 + j0);
  const size_t pstep = TILED == 2 ? (size_t)(T0 * T1) : plane;
  // halo rows: threads of the first two rows also fetch rows t1 - 1 and t1 + T1
  const int hy = j1 == 0 ? (t1 == 0 ? n1 - 1 : t1 - 1) : (t1 + T1 == n1 ? 0 : t1 + T1);
  const size_t halo = (size_t)hy * n0 + t0 + j0;
  float cur[Q], nxt[Q], h = 0.f;
  auto load = [&](int kk, float (&r)[Q]) {
    const int k = kk >= n2 ? kk - n2 : kk;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      r[q] = (MODE & 1) ? in[q * N + k * pstep + own] : (float)(q + tid);
      if (HALO && (MODE & 1) && j1 < 2 && TILED == 0) h += in[q * N + k * plane + halo];
    }
  };
  load(s, cur);
  for (int k = s; k < s + seg; ++k) {
    if (k + 1 < s + seg) load(k + 1, nxt);
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (MODE & 2) {
#pragma unroll
      for (int q = 0; q < Q; ++q) __builtin_nontemporal_store(cur[q] + (HALO ? h * 0.f : 0.f), out + q * N + (k >= n2 ? k - n2 : k) * pstep + own);
    } else {
      float acc = h;
#pragma unroll
      for (int q = 0; q < Q; ++q) acc += cur[q];
      if (acc == 12345.678f) out[own] = acc;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) cur[q] = nxt[q];
  }
}

template <int T0, int T1, int MODE, int TILED, bool BARRIER, bool HALO>
static int go(const float *in, float *out, int n0, int n1, int n2, int seg, int lds, hipStream_t st, int skew = 0, long long pstride = 0) {
  if (pstride == 0) pstride = (long long)n0 * n1 * n2;
  const unsigned grid = (unsigned)((n0 / T0) * (n1 / T1) * (n2 / seg));
  (void)hipFuncSetAttribute((const void *)tile_copy<19, T0, T1, MODE, TILED, BARRIER, HALO>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((tile_copy<19, T0, T1, MODE, TILED, BARRIER, HALO>), dim3(grid), dim3(T0 * T1), lds, st, in, out, n0, n1, n2, seg, skew, pstride);
  return (int)hipGetLastError();
}

// variant = shape * 100 + mode * 10 + flags; shape 0: 64x8, 1: 128x4, 2: 256x2, 3: 256x4, 4: 64x4, 5: 32x8; flags bit0 tiled, bit1 barrier, bit2 halo
extern "C" int lt_tile_copy(int variant, const float *in, float *out, int n0, int n1, int n2, int seg, int lds, void *stream,
                            int skew, long long pstride) {
  hipStream_t st = (hipStream_t)stream;
  const int shape = variant / 100, mode = (variant / 10) % 10, fl = variant % 10;
#define SH(S, A, B)                                                                                    \
  if (shape == S) {                                                                                    \
    if (mode == 3 && fl == 0) return go<A, B, 3, 0, false, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride); \
    if (mode == 3 && fl == 1) return go<A, B, 3, 1, false, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);  \
    if (mode == 3 && fl == 2) return go<A, B, 3, 0, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);  \
    if (mode == 3 && fl == 3) return go<A, B, 3, 1, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
    if (mode == 3 && fl == 6) return go<A, B, 3, 0, true, true>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
    if (mode == 3 && fl == 4) return go<A, B, 3, 2, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
    if (mode == 1 && fl == 4) return go<A, B, 1, 2, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
    if (mode == 2 && fl == 4) return go<A, B, 2, 2, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
    if (mode == 1 && fl == 2) return go<A, B, 1, 0, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);  \
    if (mode == 1 && fl == 3) return go<A, B, 1, 1, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
    if (mode == 2 && fl == 2) return go<A, B, 2, 0, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);  \
    if (mode == 2 && fl == 3) return go<A, B, 2, 1, true, false>(in, out, n0, n1, n2, seg, lds, st, skew, pstride);   \
  }
  SH(0, 64, 8) SH(1, 128, 4) SH(2, 256, 2) SH(3, 256, 4) SH(4, 64, 4) SH(5, 32, 8)
  return -1;
}
