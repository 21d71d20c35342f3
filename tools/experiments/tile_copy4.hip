// Development-only: the sweep access pattern of the two-step kernel with 16-byte accesses per lane (tile_copy.hip
// uses 4-byte ones).  A 512-thread workgroup owns a 64 x 8 tile; thread t moves float4 number t % 128 of the
// tile (row (t%128)/16, columns 4*(t%16)..+3) of the populations q = t/128, t/128 + 4, ...
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int Q, int MODE, bool BARRIER>
__global__ void __launch_bounds__(512) tile_copy4(const float *__restrict__ in, float *__restrict__ out, int n0, int n1,
                                                  int n2, int seg) {
  extern __shared__ float dummy[];
  constexpr int T0 = 64, T1 = 8;
  const int tid = threadIdx.x;
  const int tiles0 = n0 / T0, tiles1 = n1 / T1;
  int b = blockIdx.x;
  if (gridDim.x % 8 == 0) b = (b % 8) * (gridDim.x / 8) + b / 8;
  const int tile = b % (tiles0 * tiles1);
  const int t0 = (tile % tiles0) * T0, t1 = (tile / tiles0) * T1;
  const int s = (b / (tiles0 * tiles1)) * seg;
  const int pg = tid >> 7, i = tid & 127;
  const size_t plane = (size_t)n0 * n1, N = plane * n2;
  const size_t own = (size_t)(t1 + (i >> 4)) * n0 + t0 + 4 * (i & 15);
  constexpr int NQ = (Q + 3) / 4;
  f4 cur[NQ], nxt[NQ];
  auto load = [&](int k, f4 (&r)[NQ]) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = pg + 4 * j;
      if (q < Q) r[j] = (MODE & 1) ? *reinterpret_cast<const f4 *>(in + q * N + k * plane + own) : f4{1.f, 2.f, 3.f, (float)tid};
    }
  };
  load(s, cur);
  for (int k = s; k < s + seg; ++k) {
    if (k + 1 < s + seg) load(k + 1, nxt);
    if (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = pg + 4 * j;
      if (q < Q) {
        if (MODE & 2) __builtin_nontemporal_store(cur[j], reinterpret_cast<f4 *>(out + q * N + k * plane + own));
        else if (cur[j].x == 12345.678f) out[own] = cur[j].y;
      }
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) cur[j] = nxt[j];
  }
}

extern "C" int lt_tile_copy4(int mode, int barrier, const float *in, float *out, int n0, int n1, int n2, int seg, int lds,
                             void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((n0 / 64) * (n1 / 8) * (n2 / seg));
#define GO(M, B)                                                                                               \
  if (mode == M && barrier == B) {                                                                             \
    (void)hipFuncSetAttribute((const void *)tile_copy4<19, M, (B != 0)>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((tile_copy4<19, M, (B != 0)>), dim3(grid), dim3(512), lds, st, in, out, n0, n1, n2, seg);  \
    return (int)hipGetLastError();                                                                             \
  }
  GO(1, 0) GO(1, 1) GO(2, 0) GO(2, 1) GO(3, 0) GO(3, 1)
  return -1;
}
