// Three fused steps per launch (periodic, no masks): f*_out = (C S)^3 f*_in with BOTH intermediate states in LDS,
// so that HBM sees one read and one write of the populations per THREE lattice updates.
//
// A workgroup owns a T0 x T1 column of output nodes and sweeps along a2 like lbm2_kernel (kernels.hpp), with two
// intermediate levels instead of one:
//   level 1 on the (T0+4) x (T1+4) tile around the outputs: pulled from global memory and collided   (phase A)
//   level 2 on the (T0+2) x (T1+2) tile: pulled from level 1 in LDS and collided                     (phase B)
//   level 3 = the outputs: pulled from level 2 in LDS, collided, stored                              (phase C)
// In iteration i the workgroup produces level-1 plane a = s - 2 + i, level-2 plane a - 2 and output plane a - 4, each
// from planes completed in earlier iterations.  A level keeps 3 slots of the populations moving up, 2 of the
// in-plane ones and 1 of those moving down (38 population planes, two thirds of lbm2_kernel's 57), which needs TWO
// barriers per iteration: every thread first reads what its phases B and C need, then -- after the second barrier
// -- the phases write into the slots those reads have just freed:
//   slot of plane x in iteration i:   up: (x's iteration) % 3,  in-plane: & 1,  down: 0
//   B(i) reads level 1 written in iterations i-3 (up), i-2 (in-plane), i-1 (down) = slots i%3, i&1, 0 -- exactly
//   the slots A(i) writes; C(i) and B(i) share the level-2 slots the same way.
// Arithmetic per node is the one-step kernel's (same pull, same collide): bit for bit three lbm_kernel launches.
// Redundant work: (T0+4)(T1+4) + (T0+2)(T1+2) nodes for the first two steps of T0 T1 outputs, four extra planes per
// segment.  64 x 4 tiles: 143 KB of LDS, sixteen waves with one role (or B and C) each.
#pragma once

namespace lt {

template <typename T, class S, int T0_, int T1>
struct ThreeStep {
  static constexpr int T0 = T0_;
  static constexpr int A0 = T0 + 4, A1 = T1 + 4, NA = A0 * A1;   // level-1 tile
  static constexpr int B0 = T0 + 2, B1 = T1 + 2, NB = B0 * B1;   // level-2 tile
  static constexpr int NO = T0 * T1;                             // outputs
  // wave roles: the first WA waves do phase A, the next WB phase B, and the last WC of those also phase C -- no wave
  // does more than two phases per iteration and (64 x 4 fp32: 9 + 7 waves, C on the last 4) every SIMD gets five
  static constexpr int WA = (NA + 63) / 64, WB = (NB + 63) / 64, WC = (NO + 63) / 64;
  static constexpr int THREADS = (WA + WB) * 64;
  static_assert(WC <= WB, "phase C rides on phase-B waves");
  template <int LAYOUT, int E2>
  static constexpr int count() {
    int n = 0;
    for (int q = 0; q < S::Q; ++q) n += MemMap<S, LAYOUT>::e(q, 2) == E2 ? 1 : 0;
    return n;
  }
  template <int LAYOUT>
  static constexpr size_t lds_bytes() {
    return sizeof(T) * (size_t)(NA + NB) * (3 * count<LAYOUT, 1>() + 2 * count<LAYOUT, 0>() + count<LAYOUT, -1>());
  }
};

template <typename T, class S, int LAYOUT, int COLL, int T0_, int T1>
__global__ void __launch_bounds__((ThreeStep<T, S, T0_, T1>::THREADS))
lbm3_kernel(const KParams<T> p, const int seg_len) {
  using G = ThreeStep<T, S, T0_, T1>;
  using M = MemMap<S, LAYOUT>;
  constexpr int T0 = G::T0, A0 = G::A0, A1 = G::A1, NA = G::NA, B0 = G::B0, B1 = G::B1, NB = G::NB, NO = G::NO;
  constexpr int PU = G::template count<LAYOUT, 1>(), PC = G::template count<LAYOUT, 0>(),
                PD = G::template count<LAYOUT, -1>();
  static_assert(COLL == 0 || COLL == 1, "three-step kernel: streaming only or BGK");
  __shared__ T l1u[3][PU][NA];
  __shared__ T l1c[2][PC][NA];
  __shared__ T l1d[1][PD][NA];
  __shared__ T l2u[3][PU][NB];
  __shared__ T l2c[2][PC][NB];
  __shared__ T l2d[1][PD][NB];

  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / T0, tiles1 = p.n1 / T1;
  // an eighth of every layer of tiles per XCD, as a compact patch (kernels.hpp, lbm2_kernel)
  int b = blockIdx.x;
  {
    const int tiles = tiles0 * tiles1;
    if (tiles % 8 == 0) {
      const int layer = b / tiles, t = b - layer * tiles;
      b = layer * tiles + (t % 8) * (tiles / 8) + t / 8;
    }
  }
  const int t0 = (b % tiles0) * T0; b /= tiles0;
  const int t1 = (b % tiles1) * T1; b /= tiles1;
  const int s = p.p_begin + b * seg_len;
  const int len = s + seg_len < p.p_end ? seg_len : p.p_end - s;

  // roles are per wave (scalar branches: the register allocation of a wave is that of its own phases)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_a = wave < G::WA, role_b = !role_a, role_c = wave >= G::WA + G::WB - G::WC;
  const int tb = tid - G::WA * 64, tc = tid - (G::WA + G::WB - G::WC) * 64;
  const bool in_a = tid < NA, in_b = role_b && tb < NB, in_c = role_c && tc < NO;
  // phase A: node (i0, i1) of the level-1 tile; the T0 inner columns of a row go to T0 consecutive threads (one
  // aligned 256-byte row segment per wave and population), the four halo columns of all rows to the last threads
  int a_at;
  unsigned voff[3][3];                               // [e1 + 1][e0 + 1], bytes within a plane
  {
    constexpr int inner = T0 * A1;
    int i1, i0;
    if (tid < inner) {
      i1 = tid / T0; i0 = 2 + (tid - i1 * T0);
    } else {
      const int h = tid - inner;
      i1 = h >> 2;
      const int c = h & 3;
      i0 = c < 2 ? c : T0 + c;
    }
    if (i1 >= A1) i1 = A1 - 1;                       // threads beyond the tile (not in_a) stay in range
    a_at = i1 * A0 + i0;
    int g0 = t0 + i0 - 2; g0 = g0 < 0 ? g0 + p.n0 : (g0 >= p.n0 ? g0 - p.n0 : g0);
    int g1 = t1 + i1 - 2; g1 = g1 < 0 ? g1 + p.n1 : (g1 >= p.n1 ? g1 - p.n1 : g1);
    const int g0m = g0 == 0 ? p.n0 - 1 : g0 - 1, g0p = g0 == p.n0 - 1 ? 0 : g0 + 1;
    const int g1m = g1 == 0 ? p.n1 - 1 : g1 - 1, g1p = g1 == p.n1 - 1 ? 0 : g1 + 1;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int y = a == 0 ? g1p : (a == 1 ? g1 : g1m);     // source = node - e
        const int x = c == 0 ? g0p : (c == 1 ? g0 : g0m);
        voff[a][c] = ((unsigned)y * (unsigned)p.n0 + (unsigned)x) * (unsigned)sizeof(T);
      }
  }
  // phase B: node (j0, j1) of the level-2 tile = node (j0 + 1, j1 + 1) of the level-1 tile
  int b_src, b_at;
  {
    constexpr int inner = T0 * B1;
    const int t = tb < 0 ? 0 : tb;
    int j1, j0;
    if (t < inner) {
      j1 = t / T0; j0 = 1 + (t - j1 * T0);
    } else {
      const int h = t - inner;
      j1 = h >> 1;
      j0 = (h & 1) ? B0 - 1 : 0;
    }
    if (j1 >= B1) j1 = B1 - 1;
    b_at = j1 * B0 + j0;
    b_src = (j1 + 1) * A0 + (j0 + 1);
  }
  // phase C: output node (k0, k1) = node (k0 + 1, k1 + 1) of the level-2 tile
  int c_src;
  unsigned out_off;
  {
    const int t = tc < 0 ? 0 : tc;
    int k1 = t / T0;
    const int k0 = t - k1 * T0;
    if (k1 >= T1) k1 = T1 - 1;
    c_src = (k1 + 1) * B0 + (k0 + 1);
    out_off = ((unsigned)(t1 + k1) * (unsigned)p.n0 + (unsigned)(t0 + k0)) * (unsigned)sizeof(T);
  }
  const unsigned plane_nodes = (unsigned)p.n1 * (unsigned)p.n0;

  T pre[3][S::Q][1];                                 // phase A keeps two planes of loads in flight
  auto load_a = [&](int plane, T (&dst)[S::Q][1]) {  // periodic along a2
    int g2 = plane < 0 ? plane + p.n2 : (plane >= p.n2 ? plane - p.n2 : plane);
    g2 = g2 < 0 ? g2 + p.n2 : g2;
    const int g2m = g2 == 0 ? p.n2 - 1 : g2 - 1;
    const int g2p = g2 == p.n2 - 1 ? 0 : g2 + 1;
    if (in_a) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2);
        const int z = e2 == 0 ? g2 : (e2 > 0 ? g2m : g2p);
        const T *base = p.in + ((long long)q * p.Ni + (long long)((unsigned)z * plane_nodes));
        dst[q][0] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + voff[e1 + 1][e0 + 1]);
      });
    }
  };
  T fb[S::Q][1], fc[S::Q][1];
  auto read_b = [&](int i3, int i1_) {
    if (in_b) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        const int at = b_src - e1 * A0 - e0;
        if constexpr (e2 > 0) fb[q][0] = l1u[i3][rank][at];
        else if constexpr (e2 == 0) fb[q][0] = l1c[i1_][rank][at];
        else fb[q][0] = l1d[0][rank][at];
      });
    }
  };
  auto read_c = [&](int i3, int i1_) {
    if (in_c) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        const int at = c_src - e1 * B0 - e0;
        if constexpr (e2 > 0) fc[q][0] = l2u[i3][rank][at];
        else if constexpr (e2 == 0) fc[q][0] = l2c[i1_][rank][at];
        else fc[q][0] = l2d[0][rank][at];
      });
    }
  };
  auto compute_a = [&](int i3, int i1_, T (&src)[S::Q][1]) {
    if (in_a) {
      if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(src, p.tau_inv);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        if constexpr (e2 > 0) l1u[i3][rank][a_at] = src[q][0];
        else if constexpr (e2 == 0) l1c[i1_][rank][a_at] = src[q][0];
        else l1d[0][rank][a_at] = src[q][0];
      });
    }
  };
  auto compute_b = [&](int i3, int i1_) {
    if (in_b) {
      if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(fb, p.tau_inv);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        if constexpr (e2 > 0) l2u[i3][rank][b_at] = fb[q][0];
        else if constexpr (e2 == 0) l2c[i1_][rank][b_at] = fb[q][0];
        else l2d[0][rank][b_at] = fb[q][0];
      });
    }
  };
  auto compute_c = [&](int plane) {
    if (in_c) {
      if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(fc, p.tau_inv);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        T *base = p.out + ((long long)q * p.No + (long long)((unsigned)plane * plane_nodes));
        __builtin_nontemporal_store(fc[q][0], reinterpret_cast<T *>(reinterpret_cast<char *>(base) + out_off));
      });
    }
  };

  // iteration i: A makes level-1 plane s - 2 + i (i <= len + 3), B level-2 plane s - 4 + i (3 <= i <= len + 4),
  // C output plane s - 6 + i (6 <= i <= len + 5)
  // One loop per role -- the same two barriers per iteration in both -- so that a wave's registers are those of
  // its own phases (in one loop the prefetched plane of phase A stays allocated through phases B and C: spills)
  if (role_a) {
    // The loads of iteration i + 2 are issued before the collide of iteration i: one plane in flight bounds an
    // iteration from below by the memory latency under load (0.92 ms per launch at 256^3: 3.5 us per iteration)
    load_a(s - 2, pre[0]);
    if (1 <= len + 3) load_a(s - 1, pre[1]);
    auto step_a = [&](int i, auto kc) {
      constexpr int K = decltype(kc)::value;          // == i % 3: the buffer of this iteration and its up-slot
      lds_barrier();                                 // what earlier iterations wrote is complete
      lds_barrier();                                 // ... and phases B, C have read what this iteration overwrites
      if (i <= len + 3) {
        if (i + 2 <= len + 3) load_a(s + i, pre[(K + 2) % 3]);
        compute_a(K, i & 1, pre[K]);
      }
    };
    for (int i = 0; i <= len + 5; i += 3) {
      step_a(i, std::integral_constant<int, 0>{});
      if (i + 1 <= len + 5) step_a(i + 1, std::integral_constant<int, 1>{});
      if (i + 2 <= len + 5) step_a(i + 2, std::integral_constant<int, 2>{});
    }
  } else {
    int i3 = 0;
    for (int i = 0; i <= len + 5; ++i) {
      const bool do_b = i >= 3 && i <= len + 4, do_c = role_c && i >= 6;
      lds_barrier();
      if (do_b) read_b(i3, i & 1);
      if (do_c) read_c(i3, i & 1);
      lds_barrier();
      if (do_b) compute_b(i3, i & 1);
      if (do_c) compute_c(s - 6 + i);
      i3 = i3 == 2 ? 0 : i3 + 1;
    }
  }
}

}  // namespace lt
