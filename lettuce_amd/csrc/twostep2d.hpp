// Two lattice updates per launch on 2-D lattices: f*_out = (C S)^2 f*_in with the intermediate state in LDS,
// the scheme of lbm2_kernel (kernels.hpp) one dimension down.  Reference layout f[q][a1][a0] (a0 = y
// contiguous, a1 = x).  A workgroup owns W consecutive columns along a0 and sweeps seg_len rows along a1:
//   phase A(j):  thread t < W + 2 pulls node (column t0 - 1 + t, row j) of the halo'd strip from global
//                memory, collides it and writes its populations to LDS;
//   phase B(k):  thread t < W pulls output node (t0 + t, k) from the LDS rows k - 1, k, k + 1, collides and
//                stores.
// B(k) reads the populations moving up the sweep axis (e along a1 = +1) only from row k - 1, the ones that
// stay in the row only from row k, the downward ones only from row k + 1: 4 + 3 + 2 LDS row slots per
// class, one barrier per row, A two rows ahead of B -- as in three dimensions.  Only the two halo columns
// are redundant (W = 512: 0.4 %), and a workgroup needs 27 row slots of W + 2 values: 55 KB in fp32 (two
// workgroups per CU), 111 KB in fp64.  Same pull, same collide: bit for bit two lbm_kernel launches.
#pragma once
#include "kernels.hpp"
#include "twostep_masked.hpp"

namespace lt {

template <class S, int E1>
constexpr int count_a1() {                           // populations with e along memory axis a1 == E1
  int n = 0;
  for (int q = 0; q < S::Q; ++q) n += MemMap<S, 0>::e(q, 1) == E1 ? 1 : 0;
  return n;
}
template <class S, int q>
constexpr int rank_a1() {                            // populations q' < q with the same e along a1
  int r = 0;
  for (int k = 0; k < q; ++k)
    if (MemMap<S, 0>::e(k, 1) == MemMap<S, 0>::e(q, 1)) ++r;
  return r;
}

template <typename T, int W>
struct TwoStep2D {
  static constexpr int NI = W + 2;                   // intermediate nodes per row
  static constexpr int THREADS = (NI + 63) / 64 * 64;
};

template <typename T, class S, int COLL, int W>
__global__ void __launch_bounds__((TwoStep2D<T, W>::THREADS))
lbm2d2_kernel(const KParams<T> p, const int seg_len) {
  static_assert(S::D == 2, "2-D lattices");
  static_assert(COLL == 0 || COLL == 1, "two-step kernel: streaming only or BGK");
  using M = MemMap<S, 0>;
  constexpr int NI = TwoStep2D<T, W>::NI;
  constexpr int NU = count_a1<S, 1>(), NC = count_a1<S, 0>(), ND = count_a1<S, -1>();
  __shared__ T lds_u[4][NU][NI];
  __shared__ T lds_c[3][NC][NI];
  __shared__ T lds_d[2][ND][NI];

  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / W;
  const int t0 = ((int)blockIdx.x % tiles0) * W;
  const int s = ((int)blockIdx.x / tiles0) * seg_len;        // first output row of this workgroup
  const int rows = p.n1;
  const bool in_a = tid < NI, in_b = tid < W;
  // phase A: the W inner columns go to the first W threads (aligned row segments per wave), the two halo
  // columns to the next two threads
  const int i0 = tid < W ? 1 + tid : (tid == W ? 0 : NI - 1);
  int g0 = t0 + i0 - 1; g0 = g0 < 0 ? g0 + p.n0 : (g0 >= p.n0 ? g0 - p.n0 : g0);
  const int g0m = g0 == 0 ? p.n0 - 1 : g0 - 1, g0p = g0 == p.n0 - 1 ? 0 : g0 + 1;
  unsigned col[3];                                   // source column by e0 + 1: node - e
  col[0] = (unsigned)g0p; col[1] = (unsigned)g0; col[2] = (unsigned)g0m;
  const unsigned out_col = (unsigned)(t0 + tid);
  const unsigned n0 = (unsigned)p.n0;

  T pre[S::Q][1];
  auto load_a = [&](int row) {
    int g1 = row < 0 ? row + rows : (row >= rows ? row - rows : row);
    const int g1m = g1 == 0 ? rows - 1 : g1 - 1, g1p = g1 == rows - 1 ? 0 : g1 + 1;
    if (in_a) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
        const int z = e1 == 0 ? g1 : (e1 > 0 ? g1m : g1p);
        pre[q][0] = p.in[(long long)q * p.Ni + (long long)((unsigned)z * n0 + col[e0 + 1])];
      });
    }
  };
  // r = index of the row relative to s - 1; r3 = r % 3
  auto compute_a = [&](int r, int r3) {
    if (in_a) {
      if constexpr (COLL == 1) collide_bgk<T, S, 0, 1, 0>(pre, p.tau_inv);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e1 = M::e(q, 1), rank = rank_a1<S, q>();
        if constexpr (e1 > 0) lds_u[r & 3][rank][i0] = pre[q][0];
        else if constexpr (e1 == 0) lds_c[r3][rank][i0] = pre[q][0];
        else lds_d[r & 1][rank][i0] = pre[q][0];
      });
    }
  };
  T f[S::Q][1];
  auto read_b = [&](int r, int r3) {
    if (in_b) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), rank = rank_a1<S, q>();
        const int at = tid + 1 - e0;
        if constexpr (e1 > 0) f[q][0] = lds_u[(r - 1) & 3][rank][at];
        else if constexpr (e1 == 0) f[q][0] = lds_c[r3][rank][at];
        else f[q][0] = lds_d[(r + 1) & 1][rank][at];
      });
    }
  };
  auto finish_b = [&](int k) {
    if (in_b) {
      if constexpr (COLL == 1) collide_bgk<T, S, 0, 1, 0>(f, p.tau_inv);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        __builtin_nontemporal_store(f[q][0], p.out + ((long long)q * p.No + (long long)((unsigned)k * n0 + out_col)));
      });
    }
  };

  // intermediate rows s-1 .. s+seg_len are needed (relative indices 0 .. seg_len+1)
  const int last = s + seg_len < rows ? s + seg_len : rows;
  load_a(s - 1); compute_a(0, 0);
  load_a(s);     compute_a(1, 1);
  load_a(s + 1); compute_a(2, 2);
  if (s + 2 <= last) load_a(s + 2);
  int r = 1, r3 = 1;                                // output row k has relative index k - s + 1
  for (int k = s; k < last; ++k) {
    lds_barrier();                                  // rows up to k + 1 complete; reads of k - 1 done
    read_b(r, r3);
    if (k + 2 <= last) {
      compute_a(r + 2, r3 == 0 ? 2 : r3 - 1);       // (r + 2) % 3
      if (k + 3 <= last) load_a(k + 3);
    }
    finish_b(k);
    ++r;
    r3 = r3 == 2 ? 0 : r3 + 1;
  }
}

// ---- the same with boundaries ------------------------------------------------------------------------------
// lbm2m_kernel's rules (twostep_masked.hpp) one dimension down: bounce-back and equilibrium nodes anywhere; an
// anti-bounce-back outlet only at the LAST row of the sweep axis (memory axis a1 = x, side +1: the reference's
// 2-D Obstacle), where the moments of the node next to an outlet node were computed by the same thread one row
// earlier (both phases) and the outlet's no-streaming bits -- the populations moving down the sweep axis, on
// every node of that row -- are a test on the row index: phase A reads them from the node itself, phase B from
// the node's own intermediate populations, for which the downward populations get a third LDS slot.  The
// populations of the first two uniform equilibrium boundaries sit in LDS.  Bit for bit two masked lbm_kernel
// launches.
template <typename T, class S, int COLL, int W>
__global__ void __launch_bounds__((TwoStep2D<T, W>::THREADS))
lbm2d2m_kernel(const KParams<T> p, const int seg_len) {
  static_assert(S::D == 2, "2-D lattices");
  static_assert(COLL == 0 || COLL == 1, "two-step kernel: streaming only or BGK");
  using M = MemMap<S, 0>;
  constexpr int NI = TwoStep2D<T, W>::NI;
  constexpr int NU = count_a1<S, 1>(), NC = count_a1<S, 0>(), ND = count_a1<S, -1>();
  __shared__ T lds_u[4][NU][NI];
  __shared__ T lds_c[3][NC][NI];
  __shared__ T lds_d[3][ND][NI];
  __shared__ T lds_feq[kEqCached][S::Q];

  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / W;
  const int t0 = ((int)blockIdx.x % tiles0) * W;
  const int s = ((int)blockIdx.x / tiles0) * seg_len;        // first output row of this workgroup
  const int rows = p.n1;
  const bool in_a = tid < NI, in_b = tid < W;
  const int i0 = tid < W ? 1 + tid : (tid == W ? 0 : NI - 1);
  int g0 = t0 + i0 - 1; g0 = g0 < 0 ? g0 + p.n0 : (g0 >= p.n0 ? g0 - p.n0 : g0);
  const int g0m = g0 == 0 ? p.n0 - 1 : g0 - 1, g0p = g0 == p.n0 - 1 ? 0 : g0 + 1;
  unsigned col[3];                                   // source column by e0 + 1: node - e
  col[0] = (unsigned)g0p; col[1] = (unsigned)g0; col[2] = (unsigned)g0m;
  const unsigned out_col = (unsigned)(t0 + tid);
  const unsigned n0 = (unsigned)p.n0;

  const MaskedPlanInfo info = masked_plan_info(p);
  if (tid < kEqCached * S::Q) {
    const int c = tid / S::Q, slot = c == 0 ? info.eq0 : info.eq1;
    lds_feq[c][tid - c * S::Q] = slot ? p.bt->feq[slot][tid - c * S::Q] : T(0);
  }
  lds_barrier();
  const bool abb_on = info.abb_slot != 0;
  const int abb_nbr = abb_on ? info.abb_plane - info.abb_side : -2;     // the row next to the outlet row
  auto wrapped = [&](int row) { return row < 0 ? row + rows : (row >= rows ? row - rows : row); };

  auto collide_and_bound = [&](T (&g)[S::Q][1], int nd, bool on_outlet, unsigned own, T rn, const T (&jn)[3]) {
    const int bidx = nd & 0x7f;
    if (bidx == 0) {
      if constexpr (COLL == 1) collide_bgk<T, S, 0, 1, 0>(g, p.tau_inv);
    }
    if (on_outlet && (bidx == 0 || info.abb_slot <= bidx)) abb_apply_ax<T, S, 0, 1>(info.abb_side, rn, jn, g);
    if (bidx != 0) {
      const int kind = (int)((info.kinds >> (2 * bidx)) & 3u);
      if (kind == kBounceBack) {
        bounce_back<T, S, 1, 0>(g);
      } else if (kind == kEquilibrium) {
        if ((info.fields >> bidx) & 1u) {
          const T *fld = p.bt->field[bidx];
          static_for<S::Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            g[q][0] = fld[(long long)q * p.N + own];
          });
        } else if (bidx == info.eq0 || bidx == info.eq1) {
          const int c = bidx == info.eq0 ? 0 : 1;
          static_for<S::Q>([&](auto qc) { g[decltype(qc)::value][0] = lds_feq[c][decltype(qc)::value]; });
        } else {
          static_for<S::Q>([&](auto qc) { g[decltype(qc)::value][0] = p.bt->feq[bidx][decltype(qc)::value]; });
        }
      }
      if (on_outlet && info.abb_slot > bidx) abb_apply_ax<T, S, 0, 1>(info.abb_side, rn, jn, g);
    }
  };
  auto moments_for_outlet = [&](const T (&g)[S::Q][1], int nd, unsigned own, T &rho, T (&j)[3]) {
    moments<T, S, 0, 1, 0>(g, rho, j);
    lower_boundaries_on_moments<T, S, 0>(p, nd & 0x7f, info.abb_slot, own, rho, j);
  };

  T pre[S::Q][1];
  int nd_pre = 0;
  T sa_rho = T(1), sa_j[3] = {T(0), T(0), T(0)};      // phase A: moments of this column one row earlier
  auto load_a = [&](int row) {
    const int g1 = wrapped(row);
    const int g1m = g1 == 0 ? rows - 1 : g1 - 1, g1p = g1 == rows - 1 ? 0 : g1 + 1;
    const bool keep_down = abb_on && g1 == info.abb_plane;     // the outlet row keeps its downward populations
    if (in_a) {
      nd_pre = p.node[(unsigned)g1 * n0 + (unsigned)g0];
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
        const int z = e1 == 0 ? g1 : (e1 > 0 ? g1m : (keep_down ? g1 : g1p));
        const unsigned c = (e1 < 0 && keep_down) ? (unsigned)g0 : col[e0 + 1];
        pre[q][0] = p.in[(long long)q * p.Ni + (long long)((unsigned)z * n0 + c)];
      });
    }
  };
  auto compute_a = [&](int r, int r3, int row) {
    if (in_a) {
      const int g1 = wrapped(row);
      const unsigned own = (unsigned)g1 * n0 + (unsigned)g0;
      T keep_rho = T(1), keep_j[3] = {T(0), T(0), T(0)};
      if (g1 == abb_nbr) moments_for_outlet(pre, nd_pre, own, keep_rho, keep_j);
      collide_and_bound(pre, nd_pre, abb_on && g1 == info.abb_plane, own, sa_rho, sa_j);
      sa_rho = keep_rho; sa_j[0] = keep_j[0]; sa_j[1] = keep_j[1]; sa_j[2] = keep_j[2];
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e1 = M::e(q, 1), rank = rank_a1<S, q>();
        if constexpr (e1 > 0) lds_u[r & 3][rank][i0] = pre[q][0];
        else if constexpr (e1 == 0) lds_c[r3][rank][i0] = pre[q][0];
        else lds_d[r3][rank][i0] = pre[q][0];
      });
    }
  };
  T f[S::Q][1];
  int nd_b = 0;
  T sb_rho = T(1), sb_j[3] = {T(0), T(0), T(0)};      // phase B: moments of this column one row earlier
  auto read_b = [&](int r, int r3, int k) {
    if (in_b) {
      const bool keep_down = abb_on && k == info.abb_plane;
      const int dslot = keep_down ? r3 : (r3 == 2 ? 0 : r3 + 1);
      nd_b = p.node[(unsigned)k * n0 + out_col];
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), rank = rank_a1<S, q>();
        const int at = tid + 1 - e0;
        if constexpr (e1 > 0) f[q][0] = lds_u[(r - 1) & 3][rank][at];
        else if constexpr (e1 == 0) f[q][0] = lds_c[r3][rank][at];
        else f[q][0] = lds_d[dslot][rank][keep_down ? tid + 1 : at];
      });
    }
  };
  auto finish_b = [&](int k) {
    if (in_b) {
      const unsigned own = (unsigned)k * n0 + out_col;
      T keep_rho = T(1), keep_j[3] = {T(0), T(0), T(0)};
      if (k == abb_nbr) moments_for_outlet(f, nd_b, own, keep_rho, keep_j);
      collide_and_bound(f, nd_b, abb_on && k == info.abb_plane, own, sb_rho, sb_j);
      sb_rho = keep_rho; sb_j[0] = keep_j[0]; sb_j[1] = keep_j[1]; sb_j[2] = keep_j[2];
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        __builtin_nontemporal_store(f[q][0], p.out + ((long long)q * p.No + (long long)own));
      });
    }
  };

  const int last = s + seg_len < rows ? s + seg_len : rows;
  if (abb_on && wrapped(s - 1) == info.abb_plane) {
    // the sweep opens ON the outlet row (row -1 of the periodic grid): the moments of the row before it have
    // not been met yet
    load_a(s - 2);
    if (in_a) moments_for_outlet(pre, nd_pre, (unsigned)wrapped(s - 2) * n0 + (unsigned)g0, sa_rho, sa_j);
  }
  load_a(s - 1); compute_a(0, 0, s - 1);
  load_a(s);     compute_a(1, 1, s);
  load_a(s + 1); compute_a(2, 2, s + 1);
  if (s + 2 <= last) load_a(s + 2);
  int r = 1, r3 = 1;                                // output row k has relative index k - s + 1
  for (int k = s; k < last; ++k) {
    lds_barrier();                                  // rows up to k + 1 complete; reads of k - 1 done
    read_b(r, r3, k);
    if (k + 2 <= last) {
      compute_a(r + 2, r3 == 0 ? 2 : r3 - 1, k + 2);    // (r + 2) % 3
      if (k + 3 <= last) load_a(k + 3);
    }
    finish_b(k);
    ++r;
    r3 = r3 == 2 ? 0 : r3 + 1;
  }
}

}  // namespace lt
