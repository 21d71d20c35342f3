// Development-only: timing ablations of the two-step kernel's skeleton (twostep.hpp, DBG mask: 1 no halo-column
// loads, 2 no LDS traffic, 4 no barriers, 8 no stores, 16 no loads).  Results are wrong by construction; only the
// launch time is of interest.  Built into tools/experiments/libtwostep_exp.so, driven by twostep_exp.py.
#include <hip/hip_runtime.h>
#include <string.h>
#include "kernels.hpp"
#include "twostep.hpp"   // tools/experiments/twostep.hpp

using namespace lt;

template <int COLL, int ORDER, int PF, int DBG>
static int go(const void *in, void *out, int n0, int n1, int n2, double tau, int seg, hipStream_t st) {
  using T = float; using S = D3Q19;
  using B = TwoStep<T, S, 64, 8>;
  KParams<T> p;
  memset(&p, 0, sizeof p);
  p.in = (const T *)in; p.out = (T *)out;
  p.n0 = n0; p.n1 = n1; p.n2 = n2; p.nv0 = n0; p.wrap2 = 1;
  p.p_begin = 0; p.p_end = n2;
  p.N = (long long)n0 * n1 * n2;
  p.tau_inv = (T)(1.0 / tau);
  const unsigned grid = (unsigned)((n0 / 64) * (n1 / 8) * ((n2 + seg - 1) / seg));
  hipLaunchKernelGGL((lbm2v_kernel<T, S, 0, COLL, 64, 8, ORDER, false, PF, DBG>), dim3(grid), dim3(B::THREADS), 0, st, p, seg);
  return (int)hipGetLastError();
}

// per-workgroup start / end stamps of one launch of the BGK kernel (DBG 256); stamps: 3 * grid uint64 on the device
extern "C" int lt_twostep_stamps(const void *in, void *out, int n0, int n1, int n2, double tau, int seg, void *stamps,
                                 void *stream) {
  using T = float; using S = D3Q19;
  using B = TwoStep<T, S, 64, 8>;
  KParams<T> p;
  memset(&p, 0, sizeof p);
  p.in = (const T *)in; p.out = (T *)out;
  p.n0 = n0; p.n1 = n1; p.n2 = n2; p.nv0 = n0; p.wrap2 = 1;
  p.p_begin = 0; p.p_end = n2;
  p.N = (long long)n0 * n1 * n2;
  p.tau_inv = (T)(1.0 / tau);
  p.nsm_bits = (const unsigned *)stamps;
  const unsigned grid = (unsigned)((n0 / 64) * (n1 / 8) * ((n2 + seg - 1) / seg));
  hipLaunchKernelGGL((lbm2v_kernel<T, S, 0, 1, 64, 8, 0, false, 1, 256>), dim3(grid), dim3(B::THREADS), 0, (hipStream_t)stream, p, seg);
  return (int)hipGetLastError();
}

extern "C" int lt_twostep_experiment(int coll, int dbg, const void *in, void *out, int n0, int n1, int n2, double tau,
                                     int seg, void *stream) {
  hipStream_t st = (hipStream_t)stream;
#define CASE(C, D) if (coll == C && dbg == D) return go<C, 0, 1, D>(in, out, n0, n1, n2, tau, seg, st);
  CASE(0, 0) CASE(0, 1) CASE(0, 2) CASE(0, 3) CASE(0, 4) CASE(0, 6) CASE(0, 7) CASE(0, 8) CASE(0, 16) CASE(0, 24) CASE(0, 10) CASE(0, 18)
  CASE(1, 0) CASE(1, 1) CASE(1, 8) CASE(1, 16) CASE(1, 24) CASE(1, 32) CASE(1, 64) CASE(1, 128) CASE(1, 160) CASE(0, 32) CASE(0, 64)
  return -1;
}
