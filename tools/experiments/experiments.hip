// Development-only variants of the hot kernel (D3Q19 / D3Q27, BGK, fused, periodic), built into
// tools/experiments/libexperiments.so and timed by tools/experiments/sweep.py.  Not part of
// the product library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "kernels.hpp"

using namespace lt;

template <typename T, class S, int VEC, int SHIFT, int TUNE, int MINW>
__global__ void __launch_bounds__(512, MINW) exp_kernel(const KParams<T> p) {
  lbm_body<T, S, 0, 1, true, true, false, VEC, SHIFT, TUNE>(p);
}

template <typename T, class S, int VEC, int SHIFT, int TUNE, int MINW>
static int go(const void *in, void *out, int n0, int n1, int n2, double tau, int cap, int tpb, hipStream_t st) {
  KParams<T> p;
  p.in = (const T *)in; p.out = (T *)out;
  p.n0 = n0; p.n1 = n1; p.n2 = n2; p.nv0 = n0 / VEC; p.p_begin = 0; p.p_stride = 1; p.wrap2 = 1;
  p.N = (long long)n0 * n1 * n2;
  p.nvec_total = (unsigned)((long long)p.nv0 * n1 * n2);
  p.tau_inv = (T)(1.0 / tau); p.beta = p.inv_beta = 0;
  p.node = nullptr; p.nsm_bits = nullptr; p.bt = nullptr; p.nb = 0;
  if (tpb <= 0) tpb = 256;
  unsigned grid = (p.nvec_total + tpb - 1) / tpb;
  (void)cap;   // the kernel body no longer has a grid-stride loop
  if (SHIFT == 3 && n0 != 64 * VEC) return -2;
  hipLaunchKernelGGL((exp_kernel<T, S, VEC, SHIFT, TUNE, MINW>), dim3(grid), dim3(tpb), 0, st, p);
  return (int)hipGetLastError();
}

// NODES independent nodes per thread, each `chunk` apart (not adjacent): more loads in flight per
// wave at scalar-access register cost.
template <typename T, class S, int NODES, int TUNE, int MINW>
__global__ void __launch_bounds__(512, MINW) multi_kernel(const KParams<T> p) {
  const unsigned chunk = gridDim.x * blockDim.x;
  const unsigned v0 = blockIdx.x * blockDim.x + threadIdx.x;
  T f[NODES][S::Q][1];
  unsigned own[NODES];
#pragma unroll
  for (int n = 0; n < NODES; ++n) {
    const unsigned v = v0 + n * chunk;
    if (v < p.nvec_total) {
      const unsigned rowid = v / (unsigned)p.nv0;
      const int c0 = (int)(v - rowid * (unsigned)p.nv0);
      const int r2 = (int)(rowid / (unsigned)p.n1);
      const int c1 = (int)(rowid - (unsigned)r2 * (unsigned)p.n1);
      const Coord c = make_coord(p, c0, c1, r2);
      own[n] = (unsigned)(r2 * p.n1 + c1) * (unsigned)p.n0 + (unsigned)c0;
      gather<T, S, 0, true, 1, 0, (TUNE & 1) != 0>(p, c, f[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < NODES; ++n) {
    const unsigned v = v0 + n * chunk;
    if (v < p.nvec_total) {
      collide_bgk<T, S, 0, 1, 0>(f[n], p.tau_inv);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        vstore<T, 1, (TUNE & 2) != 0>(p.out + (long long)q * p.N + own[n], f[n][q]);
      });
    }
  }
}

template <typename T, class S, int NODES, int TUNE, int MINW>
static int go_multi(const void *in, void *out, int n0, int n1, int n2, double tau, int tpb, hipStream_t st) {
  KParams<T> p;
  p.in = (const T *)in; p.out = (T *)out;
  p.n0 = n0; p.n1 = n1; p.n2 = n2; p.nv0 = n0; p.p_begin = 0; p.p_stride = 1; p.wrap2 = 1;
  p.N = (long long)n0 * n1 * n2;
  p.nvec_total = (unsigned)p.N;
  p.tau_inv = (T)(1.0 / tau); p.beta = p.inv_beta = 0;
  p.node = nullptr; p.nsm_bits = nullptr; p.bt = nullptr; p.nb = 0;
  if (tpb <= 0) tpb = 256;
  const unsigned per_block = tpb * NODES;
  const unsigned grid = (p.nvec_total + per_block - 1) / per_block;
  hipLaunchKernelGGL((multi_kernel<T, S, NODES, TUNE, MINW>), dim3(grid), dim3(tpb), 0, st, p);
  return (int)hipGetLastError();
}
#define M(ID, TT, SS, NODES, TUNE, MINW) \
  case ID: return go_multi<TT, SS, NODES, TUNE, MINW>(in, out, n0, n1, n2, tau, tpb, (hipStream_t)stream);

// KBC variants (periodic): LEAN on/off x launch-bounds min waves
template <typename T, class S, bool LEAN, int MINW>
__global__ void __launch_bounds__(256, MINW) kbc_kernel(const KParams<T> p) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= p.nvec_total) return;
  const unsigned rowid = v / (unsigned)p.nv0;
  const int c0 = (int)(v - rowid * (unsigned)p.nv0);
  const int r2 = (int)(rowid / (unsigned)p.n1);
  const int c1 = (int)(rowid - (unsigned)r2 * (unsigned)p.n1);
  const Coord c = make_coord(p, c0, c1, r2);
  const unsigned own = (unsigned)(r2 * p.n1 + c1) * (unsigned)p.n0 + (unsigned)c0;
  T f[S::Q][1];
  gather<T, S, 0, true, 1, 0, true>(p, c, f);
  collide_kbc<T, S, 0, 1, 0, LEAN>(f, p.beta, p.inv_beta);
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    vstore<T, 1, true>(p.out + (long long)q * p.N + own, f[q]);
  });
}
template <typename T, class S, bool LEAN, int MINW>
static int go_kbc(const void *in, void *out, int n0, int n1, int n2, double tau, hipStream_t st) {
  KParams<T> p;
  p.in = (const T *)in; p.out = (T *)out;
  p.n0 = n0; p.n1 = n1; p.n2 = n2; p.nv0 = n0; p.p_begin = 0; p.p_stride = 1; p.wrap2 = 1;
  p.N = (long long)n0 * n1 * n2; p.nvec_total = (unsigned)p.N;
  p.tau_inv = (T)(1.0 / tau); const double beta = 1. / (2 * tau); p.beta = (T)beta; p.inv_beta = (T)(1. / beta);
  p.node = nullptr; p.nsm_bits = nullptr; p.bt = nullptr; p.nb = 0;
  hipLaunchKernelGGL((kbc_kernel<T, S, LEAN, MINW>), dim3((p.nvec_total + 255) / 256), dim3(256), 0, st, p);
  return (int)hipGetLastError();
}
#define K(ID, TT, SS, LEAN, MINW) \
  case ID: return go_kbc<TT, SS, LEAN, MINW>(in, out, n0, n1, n2, tau, (hipStream_t)stream);

#define V(ID, TT, SS, VEC, SHIFT, TUNE, MINW) \
  case ID: return go<TT, SS, VEC, SHIFT, TUNE, MINW>(in, out, n0, n1, n2, tau, cap, tpb, (hipStream_t)stream);

extern "C" int lt_experiment(int id, const void *in, void *out, int n0, int n1, int n2, double tau,
                             int cap, int tpb, void *stream) {
  switch (id) {
    // D3Q19 fp32
    V(0, float, D3Q19, 4, 0, 0, 1)
    V(1, float, D3Q19, 4, 2, 3, 1)
    V(2, float, D3Q19, 4, 3, 3, 1)
    V(3, float, D3Q19, 4, 3, 2, 1)
    V(4, float, D3Q19, 4, 3, 0, 1)
    V(5, float, D3Q19, 4, 3, 1, 1)
    V(6, float, D3Q19, 4, 0, 0, 2)
    V(7, float, D3Q19, 4, 3, 3, 2)
    V(8, float, D3Q19, 4, 3, 3, 5)
    V(9, float, D3Q19, 4, 3, 3, 6)
    V(10, float, D3Q19, 2, 0, 0, 1)
    V(11, float, D3Q19, 2, 2, 3, 1)
    V(12, float, D3Q19, 2, 0, 2, 1)
    V(13, float, D3Q19, 4, 0, 2, 3)
    V(14, float, D3Q19, 4, 2, 2, 1)
    V(15, float, D3Q19, 4, 2, 1, 1)
    V(16, float, D3Q19, 4, 3, 3, 4)
    V(17, float, D3Q19, 4, 0, 0, 4)
    V(18, float, D3Q19, 4, 2, 3, 4)
    V(19, float, D3Q19, 4, 3, 2, 4)
    // D3Q19 fp64
    V(20, double, D3Q19, 2, 0, 0, 1)
    V(21, double, D3Q19, 2, 2, 3, 1)
    V(22, double, D3Q19, 2, 2, 2, 1)
    V(23, double, D3Q19, 2, 0, 2, 1)
    V(24, double, D3Q19, 2, 2, 3, 2)
    V(25, double, D3Q19, 1, 0, 0, 1)
    V(26, double, D3Q19, 1, 0, 2, 1)
    // D3Q27 fp32
    V(30, float, D3Q27, 4, 0, 0, 1)
    V(31, float, D3Q27, 4, 2, 3, 1)
    V(32, float, D3Q27, 4, 3, 3, 1)
    V(33, float, D3Q27, 4, 3, 2, 1)
    V(34, float, D3Q27, 4, 3, 3, 2)
    V(35, float, D3Q27, 2, 0, 2, 1)
    V(36, float, D3Q27, 4, 3, 3, 3)
    V(37, float, D3Q27, 4, 0, 2, 3)
    V(38, float, D3Q27, 2, 2, 3, 1)
    V(40, float, D3Q19, 1, 0, 0, 1)
    V(41, float, D3Q19, 1, 0, 2, 1)
    V(42, float, D3Q19, 2, 0, 2, 4)
    V(43, float, D3Q19, 2, 0, 3, 1)
    V(44, float, D3Q19, 2, 2, 2, 1)
    V(45, float, D3Q19, 2, 1, 2, 1)
    V(50, double, D3Q19, 1, 0, 2, 4)
    V(51, double, D3Q19, 1, 0, 3, 1)
    V(61, float, D3Q27, 1, 0, 2, 1)
    V(62, float, D3Q27, 2, 0, 2, 2)
    V(63, float, D3Q27, 2, 0, 0, 1)
    V(46, float, D3Q19, 1, 0, 3, 1)
    V(47, float, D3Q19, 1, 0, 1, 1)
    M(70, float, D3Q19, 2, 2, 1)
    M(71, float, D3Q19, 2, 3, 1)
    M(72, float, D3Q19, 3, 2, 1)
    M(73, float, D3Q19, 4, 2, 1)
    M(74, float, D3Q19, 2, 2, 8)
    M(75, float, D3Q19, 1, 2, 1)
    K(80, float, D3Q27, false, 1)
    K(81, float, D3Q27, false, 4)
    K(82, float, D3Q27, true, 1)
    K(83, float, D3Q27, true, 6)
    K(84, double, D3Q27, false, 1)
    K(85, double, D3Q27, true, 1)
    K(86, double, D3Q27, false, 2)
    K(87, float, D2Q9, false, 1)
    K(88, float, D2Q9, true, 1)
    default: return -1;
  }
}
