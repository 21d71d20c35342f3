// Hand-written gfx950 kernels of the stream-and-collide hot path.
//
// One kernel template covers the three operators the host composes
// (see include/lettuce_hip.h):
//   STREAM && COLLIDE   f*_out = B(C(S(f*_in)))   fused pull scheme (the hot kernel)
//   !STREAM && COLLIDE  f*_out = B(C(f_in))       prologue of lt_run
//   STREAM && !COLLIDE  f_out  = S(f*_in)         epilogue of lt_run
// where S = Simulation._stream (lettuce/_simulation.py:160-175), C = the collision operator
// (bgk_collision.py:17-22 / kbc_collision.py:96-160 with quadratic_equilibrium.py:11-25 and
// Flow.rho/j/u, _flow.py:136-172) and B = the boundaries applied in index order
// (_simulation.py:177-189).
//
// Design for MI355X (HBM-bound: 2*q*sizeof(T) bytes per node, ~2-4 flop/byte, no MFMA):
//  * SoA per velocity; a thread owns VEC consecutive nodes along the contiguous axis a0
//    (VEC*sizeof(T) = 16 B), so every population is read and written with one 16-byte
//    access per lane = 1 KiB per wave instruction, fully coalesced.
//  * Pull scheme: each slot of f*_in is read by exactly one thread, each slot of the output
//    is written by exactly one thread (no atomics, no write races, no halo re-reads).  The
//    +-1 shift along a0 is resolved in registers from the aligned 16-byte load plus one
//    neighbour element (or a cross-lane shift), so HBM traffic stays at the algorithmic
//    2*q*sizeof(T) per node.
//  * Everything between the loads and the stores lives in VGPRs; the lattice is a template
//    parameter so e_q, w_q and the opposite table are folded into the instruction stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lattice.hpp"

namespace lt {

constexpr int kThreads = 256;
constexpr int kMaxB = 127;   // == LT_MAX_BOUNDARIES: what the node byte's seven index bits can name

// boundary kinds (== lt_boundary_kind)
constexpr int kBounceBack = 1, kEquilibrium = 2, kAbbOutlet = 3;

template <typename T>
struct BoundaryTable {
  int kind[kMaxB + 1];       // [1..nb]
  int mem_axis[kMaxB + 1];   // ABB: memory axis of the outlet normal
  int side[kMaxB + 1];       // ABB: +1 / -1 along that memory axis
  int plane[kMaxB + 1];      // ABB: memory coordinate of the outlet plane
  int nbr[kMaxB + 1];        // ABB: memory coordinate of the plane next to it (inside)
  T feq[kMaxB + 1][27];      // EQUILIBRIUM, uniform
  const T *field[kMaxB + 1]; // EQUILIBRIUM, per node [q][N] (or null)
};

template <typename T>
struct KParams {
  const T *in;
  T *out;
  int n0, n1, n2;            // memory extents; n2 includes ghost planes
  int nv0;                   // n0 / VEC
  int p_begin;               // first a2 plane of this launch
  int p_end;                 // two-step kernel: one past the last output plane
  int p_begin2, p_end2;      // two-step kernel: optional second range of output planes (slab edges)
  int p_stride;              // distance between consecutive planes of this launch (normally 1)
  int wrap2;                 // periodic wrap along a2 (0 with ghost planes)
  long long N;               // n0*n1*n2: nodes per population (and the stride of a boundary's per-node field)
  long long Ni, No;          // distance between consecutive populations of `in` / of `out`, in elements (>= N:
                             // engine-owned buffers are padded so that the q streams of a node do not meet in the
                             // same memory channels, DESIGN.md section 4 "population stride")
  unsigned nvec_total;       // threads doing work: nv0 * n1 * planes
  T tau_inv;                 // BGK: 1/tau
  T beta, inv_beta;          // KBC: 1/(2 tau), 1/beta
  const unsigned char *node; // [N] boundary index | 0x80 if any no-streaming bit (or null)
  const unsigned *nsm_bits;  // [N] bit q set: population q keeps its value (or null)
  const BoundaryTable<T> *bt;
  int nb;
  // slab boundary launch (PACK kernels): the crossing populations of plane pack_lo_plane
  // (e along a2 = -1) / pack_hi_plane (+1) are also written to contiguous send buffers
  // [k][n1*n0], k = rank of q among the populations with that e (ascending q)
  T *pack_lo, *pack_hi;
  int pack_lo_plane, pack_hi_plane;
  // one-step kernels: slot of the plan's only anti-bounce-back outlet when its normal is the contiguous axis
  // and the rows are whole waves (n0 % 64 == 0): the node next to an outlet node then sits in the next lane
  // (lbm_body); 0: none
  int abb0_slot;
  // two-step slab launch that covers the whole slab and releases the exchange while it runs (slab layout):
  // edge_first != 0: the workgroups of the second plane range (the upper edge) and of the first segment of
  // the first range (the lower edge) get the lowest block indices; each of them adds 1 to *signal once its
  // two planes next to the cut are in memory (wait_counter_kernel on the communication stream polls it)
  unsigned long long *signal;
  int edge_first;
  // two-step slab EDGE launch that takes the neighbours' planes straight from the receive buffers (MODE 1 of
  // lbm2_kernel): ghost_lo / ghost_hi = the halo message that arrived from the rank below / above (layout of
  // halo2_kernel), read wherever the pull reaches below plane `lo` / beyond plane `hi - 1`; null = the ghost planes
  // of the field hold them (they were unpacked)
  const T *ghost_lo, *ghost_hi;
  int lo, hi;                // first interior plane, one past the last interior plane
};

// ---- constants the reference builds from cs = 1/np.sqrt(3.0) (lettuce/_stencil.py:17) ----
// cs**2 evaluates to 0.33333333333333337 in double; keep that value, not 1/3.
constexpr double kCs = 0.57735026918962584;   // 1/sqrt(3) rounded to double
constexpr double kCs2 = kCs * kCs;
constexpr double kCs4 = kCs2 * kCs2;

// ---- 16-byte vector access ---------------------------------------------------------------
template <typename T, int VEC> struct Vec;
template <> struct Vec<float, 4> {
  typedef float type __attribute__((ext_vector_type(4)));
  typedef float utype __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned char mtype __attribute__((ext_vector_type(4)));
};
template <> struct Vec<float, 2> {
  typedef float type __attribute__((ext_vector_type(2)));
  typedef float utype __attribute__((ext_vector_type(2), aligned(4)));
  typedef unsigned char mtype __attribute__((ext_vector_type(2)));
};
template <> struct Vec<double, 2> {
  typedef double type __attribute__((ext_vector_type(2)));
  typedef double utype __attribute__((ext_vector_type(2), aligned(8)));
  typedef unsigned char mtype __attribute__((ext_vector_type(2)));
};

// NT = nontemporal hint: the populations are streamed once per step and the working set
// (2.5 GB at 256^3) is far beyond L2 + Infinity Cache, so nothing is gained by keeping lines.
template <typename T, int VEC, bool NT = false>
__device__ __forceinline__ void vload(const T *__restrict__ p, T (&r)[VEC]) {
  if constexpr (VEC == 1) {
    r[0] = NT ? __builtin_nontemporal_load(p) : *p;
  } else {
    using V = typename Vec<T, VEC>::type;
    const V *vp = reinterpret_cast<const V *>(p);
    const V v = NT ? __builtin_nontemporal_load(vp) : *vp;
#pragma unroll
    for (int k = 0; k < VEC; ++k) r[k] = v[k];
  }
}
template <typename T, int VEC>
__device__ __forceinline__ void vload_unaligned(const T *__restrict__ p, T (&r)[VEC]) {
  const typename Vec<T, VEC>::utype v = *reinterpret_cast<const typename Vec<T, VEC>::utype *>(p);
#pragma unroll
  for (int k = 0; k < VEC; ++k) r[k] = v[k];
}
template <typename T, int VEC, bool NT = false>
__device__ __forceinline__ void vstore(T *__restrict__ p, const T (&r)[VEC]) {
  if constexpr (VEC == 1) {
    if constexpr (NT) __builtin_nontemporal_store(r[0], p); else *p = r[0];
  } else {
    using V = typename Vec<T, VEC>::type;
    V v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = r[k];
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V *>(p));
    else *reinterpret_cast<V *>(p) = v;
  }
}

// ---- node coordinates -----------------------------------------------------------------------
struct Coord {
  int c0, c1, c2;      // own (first of VEC along a0)
  int c1m, c1p;        // periodic neighbours along a1
  int c2m, c2p;        // neighbours along a2 (periodic iff wrap2)
};

template <typename T>
__device__ __forceinline__ Coord make_coord(const KParams<T> &p, int c0, int c1, int c2) {
  Coord c;
  c.c0 = c0; c.c1 = c1; c.c2 = c2;
  c.c1m = c1 == 0 ? p.n1 - 1 : c1 - 1;
  c.c1p = c1 == p.n1 - 1 ? 0 : c1 + 1;
  c.c2m = c2 - 1;
  c.c2p = c2 + 1;
  if (p.wrap2) {
    if (c.c2m < 0) c.c2m = p.n2 - 1;
    if (c.c2p == p.n2) c.c2p = 0;
  }
  return c;
}

// ---- gather: post-streaming populations of VEC nodes ------------------------------------
// f_q(x) = f*_q(x - e_q), periodic (Simulation._stream: torch.roll by +e_q,
// lettuce/_simulation.py:156-158,164-175).  SHIFT selects how the a0 shift is resolved:
//   0: aligned 16-B load + one neighbour element load
//   1: unaligned 16-B load (row ends handled separately)
//   2: aligned 16-B load + cross-lane shift (ds_bpermute), neighbour element only at wave/row edges
//   3: aligned 16-B load + wave ROTATE: legal when a row is exactly one wave (n0 == 64 * VEC);
//      the periodic wrap is then the rotation itself and no neighbour element is ever loaded
template <typename T, class S, int LAYOUT, bool STREAM, int VEC, int SHIFT, bool NTL = false>
__device__ __forceinline__ void gather(const KParams<T> &p, const Coord &c, T (&f)[S::Q][VEC]) {
  using M = MemMap<S, LAYOUT>;
  const int n0 = p.n0, n1 = p.n1;
  const unsigned own = (unsigned)(c.c2 * n1 + c.c1) * (unsigned)n0 + (unsigned)c.c0;
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const T *__restrict__ src = p.in + (long long)q * p.Ni;
    if constexpr (!STREAM) {
      vload<T, VEC, NTL>(src + own, f[q]);
    } else {
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2);
      const int s1 = e1 == 0 ? c.c1 : (e1 > 0 ? c.c1m : c.c1p);
      const int s2 = e2 == 0 ? c.c2 : (e2 > 0 ? c.c2m : c.c2p);
      const unsigned row = (unsigned)(s2 * n1 + s1) * (unsigned)n0;
      if constexpr (e0 == 0) {
        vload<T, VEC, NTL>(src + row + c.c0, f[q]);
      } else if constexpr (VEC == 1) {
        const int s0 = e0 > 0 ? (c.c0 == 0 ? n0 - 1 : c.c0 - 1) : (c.c0 == n0 - 1 ? 0 : c.c0 + 1);
        f[q][0] = src[row + s0];
      } else if constexpr (e0 > 0) {
        // want src[c0-1], src[c0], ..., src[c0+VEC-2]
        if constexpr (SHIFT == 1) {
          if (c.c0 != 0) {
            vload_unaligned<T, VEC>(src + row + c.c0 - 1, f[q]);
          } else {
            T a[VEC];
            vload<T, VEC, NTL>(src + row, a);
            f[q][0] = src[row + n0 - 1];
#pragma unroll
            for (int k = 1; k < VEC; ++k) f[q][k] = a[k - 1];
          }
        } else {
          T a[VEC];
          vload<T, VEC, NTL>(src + row + c.c0, a);
          T nb;
          if constexpr (SHIFT == 3) {
            nb = __shfl(a[VEC - 1], (threadIdx.x + 63) & 63);
          } else if constexpr (SHIFT == 2) {
            nb = __shfl_up(a[VEC - 1], 1);
            if ((threadIdx.x & 63) == 0 || c.c0 == 0) nb = src[row + (c.c0 == 0 ? n0 - 1 : c.c0 - 1)];
          } else {
            nb = src[row + (c.c0 == 0 ? n0 - 1 : c.c0 - 1)];
          }
          f[q][0] = nb;
#pragma unroll
          for (int k = 1; k < VEC; ++k) f[q][k] = a[k - 1];
        }
      } else {
        // want src[c0+1], ..., src[c0+VEC]
        const bool last = c.c0 + VEC == n0;
        if constexpr (SHIFT == 1) {
          if (!last) {
            vload_unaligned<T, VEC>(src + row + c.c0 + 1, f[q]);
          } else {
            T a[VEC];
            vload<T, VEC, NTL>(src + row + c.c0, a);
#pragma unroll
            for (int k = 0; k < VEC - 1; ++k) f[q][k] = a[k + 1];
            f[q][VEC - 1] = src[row];
          }
        } else {
          T a[VEC];
          vload<T, VEC, NTL>(src + row + c.c0, a);
          T nb;
          if constexpr (SHIFT == 3) {
            nb = __shfl(a[0], (threadIdx.x + 1) & 63);
          } else if constexpr (SHIFT == 2) {
            nb = __shfl_down(a[0], 1);
            if ((threadIdx.x & 63) == 63 || last) nb = src[row + (last ? 0 : c.c0 + VEC)];
          } else {
            nb = src[row + (last ? 0 : c.c0 + VEC)];
          }
#pragma unroll
          for (int k = 0; k < VEC - 1; ++k) f[q][k] = a[k + 1];
          f[q][VEC - 1] = nb;
        }
      }
    }
  });
}

// destination-side no-streaming mask: slot (q, x) keeps the value it had before streaming
// (lettuce/_simulation.py:171-174)
template <typename T, class S, int VEC, int k>
__device__ __forceinline__ void keep_unstreamed(const KParams<T> &p, unsigned own,
                                                T (&f)[S::Q][VEC]) {
  const unsigned bits = p.nsm_bits[own + k];
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q > 0) {   // population 0 never moves (_simulation.py:165)
      if (bits & (1u << q)) f[q][k] = p.in[(long long)q * p.Ni + own + k];
    }
  });
}

// ---- moments (Flow.rho / Flow.j / Flow.u, lettuce/_flow.py:136-138,152-172) ---------------
// j += e_q * v along the three memory axes, all signs folded at compile time
template <class S, int LAYOUT, int q, typename T>
__device__ __forceinline__ void add_momentum(T (&j)[3], T v) {
  using M = MemMap<S, LAYOUT>;
  static_for<3>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    constexpr int e = M::e(q, m);
    if constexpr (e > 0) j[m] += v;
    else if constexpr (e < 0) j[m] -= v;
  });
}
// e_q . u, accumulated over the logical axes x, y, z from zero (the GEMM order of the reference's
// tensordot(e, u); matters for the three-component velocities of D3Q15 / D3Q27)
template <class S, int LAYOUT, int q, typename T>
__device__ __forceinline__ T dot_e(const T (&u)[3]) {
  using M = MemMap<S, LAYOUT>;
  T r = T(0);
  static_for<S::D>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    constexpr int e = S::E[q][a];
    if constexpr (e > 0) r += u[M::memory(a)];
    else if constexpr (e < 0) r -= u[M::memory(a)];
  });
  return r;
}

// Sum over q in the order of torch.sum(f, dim=0) on the CPU (ATen cascade_sum: the first 16
// terms are accumulated from zero, then the remaining terms from zero, then the two partial sums
// are added) -- bit-identical to the reference's Flow.rho() for every lattice; sum_q e_q f_q is a
// GEMM in the reference and accumulates sequentially in q, as add_momentum does.
template <int Q, typename T>
struct CascadeSum {
  T head = T(0), tail = T(0);
  template <int q>
  __device__ __forceinline__ void add(T v) {
#pragma clang fp contract(off)
    if constexpr (q < 16) head += v; else tail += v;      // (never fused with the product that made v)
  }
  __device__ __forceinline__ T result() const {
    if constexpr (Q > 16) return tail + head; else return head;
  }
};

template <typename T, class S, int LAYOUT, int VEC, int k>
__device__ __forceinline__ void moments(const T (&f)[S::Q][VEC], T &rho, T (&j)[3]) {
  CascadeSum<S::Q, T> mass;
  j[0] = j[1] = j[2] = T(0);
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const T v = f[q][k];
    mass.template add<q>(v);
    add_momentum<S, LAYOUT, q>(j, v);
  });
  rho = mass.result();
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// value barrier: the optimiser may not assume anything about x afterwards
__device__ __forceinline__ float launder(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double launder(double x) { asm volatile("" : "+v"(x)); return x; }

// u.u summed in the logical order x, y, z with separately rounded products and sums -- what
// torch's einsum("d...,d...->...") produces on the CPU (checked bit for bit against the reference's
// vectors), so that feq(rho, u) is reproduced exactly, not just to an ulp
template <class S, int LAYOUT, typename T>
__device__ __forceinline__ T square_norm(const T (&u)[3]) {
#pragma clang fp contract(off)
  using M = MemMap<S, LAYOUT>;
  T r = u[M::memory(0)] * u[M::memory(0)];
  if constexpr (S::D > 1) r = r + u[M::memory(1)] * u[M::memory(1)];
  if constexpr (S::D > 2) r = r + u[M::memory(2)] * u[M::memory(2)];
  return r;
}

// x / D for the constants D = 2 cs^2 and cs^2 of the equilibrium.  The reference divides the fp32
// field by the python double cast to fp32 (D_f = 0.66666669 / 0.33333334, 3e-8 above 2/3 and 1/3);
// that systematic 3e-8 is what makes its fp32 kinetic energy drift by -1.1e-7 per step against its
// fp64 run.  To track the reference's fp32 path (not just fp64 truth) the same IEEE quotient is
// formed, in three instructions instead of the ~10 of a division: q = RN(x r) with r = RN(1 / D),
// the exact remainder x - q D by FMA, one correction step (Markstein).  Equal to x / D for every
// fp32 x with |x / D| >= 1e-30 (exhaustive, tests/aux/exact_division_check.c) and in 6.4e9 random
// fp64 quotients.
// fp32 (round 3): TWO instructions -- x r_hi + RN(x r_lo) with r_hi + r_lo = 1 / D to twice the precision, one
// rounding of a sum that is x / D to 2^-47 -- equal to x / D for EVERY fp32 x with |x / D| >= 1e-30 for these two
// constants (exhaustive as well: exact_division_check f32two 1; the general argument does not exclude a quotient that
// close to a rounding boundary, so fp64 keeps the three-instruction form, which Markstein's theorem covers).  The
// equilibrium calls this 28 times per node: 7 % of the collision's issue slots.
template <int WHICH, typename T>   // 0: D = 2 cs^2, 1: D = cs^2
__device__ __forceinline__ T div_cs(T x) {
  constexpr T d = (T)(WHICH == 0 ? 2.0 * kCs2 : kCs2);
  constexpr T r = (T)(1.0 / (double)d);
  if constexpr (sizeof(T) == 4) {
    constexpr T r_lo = (T)(1.0 / (double)d - (double)r);
    const T low = x * r_lo;                           // rounded on its own: a product feeding an fma's addend
    return fma_t(x, r, low);
  } else {
    const T q = x * r;
    const T rem = fma_t(-q, d, x);
    return fma_t(rem, r, q);
  }
}

// QuadraticEquilibrium (lettuce/ext/_equilibrium/quadratic_equilibrium.py:15-24), u along
// memory axes (the dot products are invariant under the axis permutation)
template <typename T, class S, int LAYOUT, int q>
__device__ __forceinline__ T feq_q(T rho, const T (&u)[3], T uxu) {
  // every operation rounds separately, as the reference's whole-field torch ops do: with
  // mul+add fused the 1-ulp differences are correlated with the sign of e.u and halve the
  // (reference-inherent) fp32 momentum deficit of the equilibrium
#pragma clang fp contract(off)
  const T exu = dot_e<S, LAYOUT, q>(u);
  const T a = div_cs<0>(T(2) * exu - uxu);
  const T b = div_cs<1>(exu);
  return T(S::W[q]) * (rho * (a + T(0.5) * (b * b) + T(1)));
}

// the same for population q and its opposite o (q < o) together: e_o = -e_q, so e_o.u, 2 e_o.u
// and (e_o.u) / cs^2 are the exact negatives and the quadratic term is shared -- the two values
// are bit for bit what feq_q<q> and feq_q<o> return, for a third less arithmetic
template <typename T, class S, int LAYOUT, int q>
__device__ __forceinline__ void feq_pair(T rho, const T (&u)[3], T uxu, T &fq, T &fo) {
#pragma clang fp contract(off)
  const T exu = dot_e<S, LAYOUT, q>(u);
  const T b = div_cs<1>(exu);
  const T h = T(0.5) * (b * b);
  // 2 e.u is exact, so one fused multiply-add IS the reference's "2 * exu - uxu" (two roundings of which the first
  // does nothing): one instruction instead of two per population
  const T two = T(2) * exu;
  const T aq = div_cs<0>(two - uxu);
  const T ao = div_cs<0>(-two - uxu);
  fq = T(S::W[q]) * (rho * (aq + h + T(1)));
  fo = T(S::W[q]) * (rho * (ao + h + T(1)));
}

// fn(q, feq_q) for every q (opposite pairs back to back, not in index order)
template <typename T, class S, int LAYOUT, class F>
__device__ __forceinline__ void for_each_feq(T rho, const T (&u)[3], T uxu, F &&fn) {
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int o = S::OPP[q];
    if constexpr (q == o) {
      fn(qc, feq_q<T, S, LAYOUT, q>(rho, u, uxu));
    } else if constexpr (q < o) {
      T a, b;
      feq_pair<T, S, LAYOUT, q>(rho, u, uxu, a, b);
      fn(qc, a);
      fn(std::integral_constant<int, o>{}, b);
    }
  });
}

// ---- collisions ---------------------------------------------------------------------------
template <typename T, class S, int LAYOUT, int VEC, int k>
__device__ __forceinline__ void collide_bgk(T (&f)[S::Q][VEC], T tau_inv) {
  T rho, j[3], u[3];
  moments<T, S, LAYOUT, VEC, k>(f, rho, j);
  u[0] = j[0] / rho; u[1] = j[1] / rho; u[2] = j[2] / rho;
  const T uxu = square_norm<S, LAYOUT>(u);
  for_each_feq<T, S, LAYOUT>(rho, u, uxu, [&](auto qc, T feq) {
#pragma clang fp contract(off)
    constexpr int q = decltype(qc)::value;
    f[q][k] = f[q][k] - tau_inv * (f[q][k] - feq);
  });
}

// BGK in "fast" arithmetic (COLL = 3; lt_plan_set_arithmetic): the same collision to rounding level instead of bit
// for bit.  collide_bgk reproduces every rounding of the reference's whole-field torch operators (~290 vector
// instructions per node: three IEEE divisions by rho, the exact-division emulation 28 times, no fused multiply-adds,
// ATen's summation order); SURVEY.md 8(d) only asks for max |df| <= 1e-5 max |f| after 10 steps and the kinetic
// energy to 1e-6 (10 steps) / 5e-5 (100 steps) in fp32.  Here: moments over opposite pairs, one reciprocal of rho,
// cs^2 = 1/3, omega folded into the weights, everything contracted -- about half the instructions.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
template <typename T, class S, int LAYOUT, int VEC, int k>
__device__ __forceinline__ void collide_bgk_fast(T (&f)[S::Q][VEC], T omega) {
  // explicit fused multiply-adds and no contraction by the compiler: every kernel this is inlined into (one-step,
  // collide-only, two-step) then returns the same bits, whatever the surrounding code
#pragma clang fp contract(off)
  using M = MemMap<S, LAYOUT>;
  T rho = T(0), j[3] = {T(0), T(0), T(0)};
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int o = S::OPP[q];
    if constexpr (q == o) {
      rho += f[q][k];
    } else if constexpr (q < o) {
      rho += f[q][k] + f[o][k];
      add_momentum<S, LAYOUT, q>(j, f[q][k] - f[o][k]);
    }
  });
  const T inv = fast_rcp(rho);
  // The reference divides by 2 cs^2 ROUNDED to the working dtype (fp32: 0.6666667, 3e-8 too large), which gives its
  // fp32 equilibrium a momentum deficit of 3e-8 and its kinetic energy a drift of -1.1e-7 per step against its fp64
  // run.  To stay within 1e-6 of ITS fp32 energy series the velocity carries the same factor:
  // u = RN(j / rho (1 - delta)), formed inside one fused multiply-add so that the half-ulp correction acts through
  // the rounding of the product (on a rounded product it would always round away).  fp64: delta = 1e-16, no effect.
  constexpr double delta = 1.5 * (double)(T)(2.0 * kCs2) - 1.0;
  T u[3] = {T(0), T(0), T(0)};
  static_for<S::D>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    const T plain = j[a] * inv;
    u[a] = fma_t(j[a], inv, T(-delta) * plain);
  });
  T uu = u[M::memory(0)] * u[M::memory(0)];
  if constexpr (S::D > 1) uu = fma_t(u[M::memory(1)], u[M::memory(1)], uu);
  if constexpr (S::D > 2) uu = fma_t(u[M::memory(2)], u[M::memory(2)], uu);
  const T c0 = fma_t(T(-1.5), uu, T(1)), keep = T(1) - omega, wr = omega * rho;
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int o = S::OPP[q];
    if constexpr (q == o) {
      f[q][k] = fma_t(keep, f[q][k], (T(S::W[q]) * wr) * c0);
    } else if constexpr (q < o) {
      const T eu = dot_e<S, LAYOUT, q>(u);
      // the weights as literals: w h +- 3 w (e.u) with h = 1 - 1.5 u.u + 4.5 (e.u)^2, then omega rho once.  Measured on
      // the same buffers against three other arrangements of the same multiply-adds (weights times omega rho in
      // registers; sums fused the other way round; the compiler's own contraction): 0.4749 against 0.4771 - 0.4845 ms
      // per launch at 256^3, the exact arithmetic 0.496 - 0.516 (profiles/r04j_fast_arithmetic_variants.jsonl)
      const T h = fma_t(T(4.5) * eu, eu, c0);
      const T g = fma_t(T(3.0 * S::W[q]), eu, T(S::W[q]) * h), m = fma_t(T(-3.0 * S::W[q]), eu, T(S::W[q]) * h);
      f[q][k] = fma_t(keep, f[q][k], wr * g);
      f[o][k] = fma_t(keep, f[o][k], wr * m);
    }
  });
}

// KBC.  The reference forms s(f) and s(feq) from the second moments m/rho of f and of feq
// (lettuce/ext/_collision/kbc_collision.py:25-39 moments, :44-94 s_i), each s_i being rho times a
// linear combination of the normalised moments -- i.e. a linear function of the populations -- and
// then ds = s(f) - s(feq).  Here ds is evaluated as s(f - feq) from the raw second moments of
// x = f - feq: no division by rho and re-multiplication, one moment pass instead of two, and the
// opposite pairs (equal e_a e_b) are summed first.  Same value up to rounding; KBC is compared with
// the reference at rounding level, not bit for bit.
template <typename T, class S>
struct KbcS {
  T s0, sa, sb, sc, pxy, pxz, pyz;   // 3-D: s0, s1(=s2), s3(=s4), s5(=s6), s15, s11, s7
  template <int q>
  static constexpr int sign() {       // get<q>() == sign * component, 0: this s_q is zero
    if constexpr (S::D == 3) return q >= 19 ? 0 : ((q >= 9 && q <= 10) || (q >= 13 && q <= 14) || q >= 17) ? -1 : 1;
    else return (q == 6 || q == 8) ? -1 : 1;
  }
  template <int q>
  __device__ __forceinline__ T magnitude() const {
    if constexpr (S::D == 3) {
      if constexpr (q == 0) return s0;
      else if constexpr (q <= 2) return sa;
      else if constexpr (q <= 4) return sb;
      else if constexpr (q <= 6) return sc;
      else if constexpr (q <= 10) return pyz;
      else if constexpr (q <= 14) return pxz;
      else if constexpr (q <= 18) return pxy;
      else return T(0);
    } else {
      if constexpr (q == 0) return s0;
      else if constexpr (q == 1 || q == 3) return sa;
      else if constexpr (q == 2 || q == 4) return sb;
      else return pxy;
    }
  }
  __device__ __forceinline__ KbcS scaled(T c) const {
    return KbcS{c * s0, c * sa, c * sb, c * sc, c * pxy, c * pxz, c * pyz};
  }
};

// s(g) for a population set given as g(q), from raw second moments (logical axes x, y, z)
template <typename T, class S, class G>
__device__ __forceinline__ KbcS<T, S> kbc_s(const G &g) {
#pragma clang fp contract(off)
  T xx = T(0), yy = T(0), zz = T(0), xy = T(0), xz = T(0), yz = T(0);
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int o = S::OPP[q];
    if constexpr (q < o) {
      constexpr int ex = S::E[q][0], ey = S::E[q][1], ez = S::E[q][2];
      const T v = g(qc) + g(std::integral_constant<int, o>{});
      if constexpr (ex != 0) xx += v;
      if constexpr (ey != 0) yy += v;
      if constexpr (ez != 0) zz += v;
      if constexpr (ex * ey > 0) xy += v; else if constexpr (ex * ey < 0) xy -= v;
      if constexpr (ex * ez > 0) xz += v; else if constexpr (ex * ez < 0) xz -= v;
      if constexpr (ey * ez > 0) yz += v; else if constexpr (ey * ez < 0) yz -= v;
    }
  });
  KbcS<T, S> s;
  if constexpr (S::D == 3) {
    const T Tr = xx + yy + zz, nxz = xx - zz, nyz = yy - zz;
    s.s0 = -Tr;
    s.sa = T(1. / 6.) * (T(2) * nxz - nyz + Tr);
    s.sb = T(1. / 6.) * (T(2) * nyz - nxz + Tr);
    s.sc = T(1. / 6.) * (-nxz - nyz + Tr);
    s.pyz = T(0.25) * yz;
    s.pxz = T(0.25) * xz;
    s.pxy = T(0.25) * xy;
  } else {
    const T Tr = xx + yy, n = xx - yy;
    s.s0 = -Tr;
    s.sa = T(0.5) * (T(0.5) * (Tr + n));
    s.sb = T(0.5) * (T(0.5) * (Tr - n));
    s.sc = T(0);
    s.pxy = T(0.25) * xy;
    s.pxz = s.pyz = T(0);
  }
  return s;
}

// x / y for the two entropic sums of KBC.  The reference divides twice per population
// (ds*dh/feq and dh*dh/feq, kbc_collision.py:150-151); here dh/feq is formed once and, in fp32,
// with the hardware reciprocal (<= 1 ulp) instead of the ~10-instruction IEEE sequence: the sums
// feed only gamma, whose own rounding noise (dh is a difference of nearly equal numbers) is three
// orders of magnitude larger.
__device__ __forceinline__ float kbc_ratio(float x, float y) { return x * __builtin_amdgcn_rcpf(y); }
__device__ __forceinline__ double kbc_ratio(double x, double y) { return x / y; }

// f' = f - beta (2 ds + gamma dh), dh = f - feq - ds (kbc_collision.py:130-158), evaluated as
// f - (beta gamma) (f - feq) - (beta (2 - gamma)) ds with the seven distinct ds values scaled once.
// No contraction by the compiler anywhere in the collision (round 4): which multiply-adds hipcc fused depended on the
// kernel the function was inlined into, so the fused, the collide-only, the many-step and the two-step KBC kernels
// agreed at rounding level only.  The two multiply-adds worth an instruction are written as fma_t.
template <typename T, class S, int LAYOUT, int VEC, int k>
__device__ __forceinline__ void collide_kbc(T (&f)[S::Q][VEC], T beta, T inv_beta) {
#pragma clang fp contract(off)
  static_assert(S::Q == 9 || S::Q == 27, "KBC exists for D2Q9 and D3Q27 only (kbc_collision.py:100-128)");
  T rho, j[3], u[3];
  moments<T, S, LAYOUT, VEC, k>(f, rho, j);
  u[0] = j[0] / rho; u[1] = j[1] / rho; u[2] = j[2] / rho;
  const T uxu = square_norm<S, LAYOUT>(u);
  T feq[S::Q];
  for_each_feq<T, S, LAYOUT>(rho, u, uxu, [&](auto qc, T v) { feq[decltype(qc)::value] = v; });
  const KbcS<T, S> ds = kbc_s<T, S>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    return f[q][k] - feq[q];
  });
  CascadeSum<S::Q, T> acc_s, acc_h;
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int sg = KbcS<T, S>::template sign<q>();
    const T x = f[q][k] - feq[q];
    const T m = ds.template magnitude<q>();
    const T dh = sg == 0 ? x : (sg > 0 ? x - m : x + m);
    const T t = kbc_ratio(dh, feq[q]);
    if constexpr (sg > 0) acc_s.template add<q>(m * t);
    if constexpr (sg < 0) acc_s.template add<q>(-(m * t));
    acc_h.template add<q>(dh * t);
  });
  const T sum_s = acc_s.result(), sum_h = acc_h.result();
  T gamma = inv_beta - (T(2) - inv_beta) * sum_s / sum_h;
  if (gamma < T(1e-15)) gamma = T(2);
  if (gamma != gamma) gamma = T(2);
  const T bg = beta * gamma;
  const KbcS<T, S> dsc = ds.scaled(beta * (T(2) - gamma));
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int sg = KbcS<T, S>::template sign<q>();
    const T x = f[q][k] - feq[q];
    const T y = fma_t(-bg, x, f[q][k]);
    const T m = dsc.template magnitude<q>();
    f[q][k] = sg == 0 ? y : (sg > 0 ? y - m : y + m);
  });
}

// ---- boundaries -------------------------------------------------------------------------
// BounceBackBoundary: f <- f[opposite] (lettuce/ext/_boundary/bounce_back_boundary.py:17-18)
template <typename T, class S, int VEC, int k>
__device__ __forceinline__ void bounce_back(T (&f)[S::Q][VEC]) {
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int o = S::OPP[q];
    if constexpr (q < o) {
      const T t = f[q][k];
      f[q][k] = f[o][k];
      f[o][k] = t;
    }
  });
}

// What the boundaries with an index below `slot` do to (rho, j) of a node whose no_collision_mask index
// is b (collision conserves both): bounce-back negates j, an equilibrium boundary replaces the moments
// by those of its populations (neighbour_moments; also the two-step kernel's phase B)
template <typename T, class S, int LAYOUT>
__device__ __forceinline__ void lower_boundaries_on_moments(const KParams<T> &p, int b, int slot, unsigned own,
                                                            T &rho, T (&j)[3]) {
  if (b > 0 && b < slot) {
    const int kind = p.bt->kind[b];
    if (kind == kBounceBack) {
      j[0] = -j[0]; j[1] = -j[1]; j[2] = -j[2];
    } else if (kind == kEquilibrium) {
      const T *fld = p.bt->field[b];
      rho = T(0); j[0] = j[1] = j[2] = T(0);
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const T v = fld ? fld[(long long)q * p.N + own] : p.bt->feq[b][q];
        rho += v;
        add_momentum<S, LAYOUT, q>(j, v);
      });
    }
  }
}

// AntiBounceBackOutlet (lettuce/ext/_boundary/anti_bounce_back_outlet.py:72-91) on one node of the outlet
// plane, whose normal is memory axis AX, given (rho, j) of the node next to it (neighbour_moments, or the
// two-step kernels' saved / shuffled moments).  Only the populations leaving through the plane (e.n = +1)
// are read and only their opposites written, so nothing is computed for the others.  No FMA contraction
// here: the function is inlined into one-step and two-step kernels that must agree bit for bit, and which
// products the backend fuses depends on the surrounding code (the reference's CPU ops do not fuse either).
template <typename T, class S, int LAYOUT, int AX, int VEC = 1, int k = 0>
__device__ __forceinline__ void abb_apply_ax(int side, T rn, const T (&jn)[3], T (&f)[S::Q][VEC]) {
#pragma clang fp contract(off)
  using M = MemMap<S, LAYOUT>;
  T rho, j[3];
  moments<T, S, LAYOUT, VEC, k>(f, rho, j);
  T uw[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const T u = j[m] / rho, un = jn[m] / rn;
    uw[m] = u + T(0.5) * (u - un);
  }
  const T nrm = sqrt(uw[0] * uw[0] + uw[1] * uw[1] + uw[2] * uw[2]) / T(kCs);
  const T nrm2 = nrm * nrm;
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int en = AX < S::D ? M::e(q, AX) : 0;
    if constexpr (en != 0) {
      if (en * side == 1) {
        const T eu = dot_e<S, LAYOUT, q>(uw);
        f[S::OPP[q]][k] = -f[q][k] + T(S::W[q]) * rho * (T(2) + eu * eu / T(kCs4) - nrm2);
      }
    }
  });
}
// the same for the outlet with index `slot` of the plan
template <typename T, class S, int LAYOUT, int VEC, int k>
__device__ __forceinline__ void abb_apply(const KParams<T> &p, int slot, T rn, const T (&jn)[3],
                                          T (&f)[S::Q][VEC]) {
  const int ax = p.bt->mem_axis[slot], side = p.bt->side[slot];
  if (ax == 0) abb_apply_ax<T, S, LAYOUT, 0, VEC, k>(side, rn, jn, f);
  else if (ax == 1) abb_apply_ax<T, S, LAYOUT, 1, VEC, k>(side, rn, jn, f);
  else abb_apply_ax<T, S, LAYOUT, 2, VEC, k>(side, rn, jn, f);
}

// (rho, j) of the node (c0, c1, c2), as Flow.rho()/Flow.u() would see it when the AntiBounceBackOutlet with
// index `slot` is evaluated: after collision (which conserves both) and after the boundaries with a lower
// index (anti_bounce_back_outlet.py:77-80).  DEPTH = how many OTHER outlets with a lower index may touch
// this node (plans with several outlets whose planes meet in an edge): there the lower outlet has already
// rewritten some of this node's populations, so the node's state is rebuilt in full -- pull, collision,
// boundaries below `slot` in order, the lower outlet with ITS neighbour's moments at DEPTH - 1 -- instead
// of being read off the conserved moments.  DEPTH = 0 is the kernel of plans with one outlet.
template <typename T, class S, int LAYOUT, bool STREAM, bool MASKED, int COLL = 0, int DEPTH = 0>
__device__ __forceinline__ void neighbour_moments(const KParams<T> &p, int c0, int c1, int c2, int slot,
                                                  T &rho, T (&j)[3]) {
  const Coord c = make_coord(p, c0, c1, c2);
  T g[S::Q][1];
  gather<T, S, LAYOUT, STREAM, 1, 0>(p, c, g);
  const unsigned own = (unsigned)(c2 * p.n1 + c1) * (unsigned)p.n0 + (unsigned)c0;
  int b = 0;
  if constexpr (MASKED) {
    const unsigned char nd = p.node[own];
    b = nd & 0x7f;
    if constexpr (STREAM) {
      if (nd & 0x80) keep_unstreamed<T, S, 1, 0>(p, own, g);
    }
  }
  if constexpr (DEPTH > 0) {
    bool touched = false;                      // does an outlet with a lower index rewrite this node?
    for (int t = 1; t < slot; ++t)
      if (p.bt->kind[t] == kAbbOutlet) {
        const int ax = p.bt->mem_axis[t];
        touched = touched || (ax == 0 ? c0 : (ax == 1 ? c1 : c2)) == p.bt->plane[t];
      }
    if (touched) {
      if (b == 0) {
        if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(g, p.tau_inv);
        if constexpr (COLL == 2) collide_kbc<T, S, LAYOUT, 1, 0>(g, p.beta, p.inv_beta);
      }
      for (int t = 1; t < slot; ++t) {
        const int kind = p.bt->kind[t];
        if (kind == kAbbOutlet) {
          const int ax = p.bt->mem_axis[t], nbr = p.bt->nbr[t];
          if ((ax == 0 ? c0 : (ax == 1 ? c1 : c2)) == p.bt->plane[t]) {
            T rn, jn[3];
            neighbour_moments<T, S, LAYOUT, STREAM, MASKED, COLL, DEPTH - 1>(
                p, ax == 0 ? nbr : c0, ax == 1 ? nbr : c1, ax == 2 ? nbr : c2, t, rn, jn);
            abb_apply<T, S, LAYOUT, 1, 0>(p, t, rn, jn, g);
          }
        } else if (b == t) {
          if (kind == kBounceBack) {
            bounce_back<T, S, 1, 0>(g);
          } else if (kind == kEquilibrium) {
            const T *fld = p.bt->field[t];
            static_for<S::Q>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              g[q][0] = fld ? fld[(long long)q * p.N + own] : p.bt->feq[t][q];
            });
          }
        }
      }
      moments<T, S, LAYOUT, 1, 0>(g, rho, j);
      return;
    }
  }
  moments<T, S, LAYOUT, 1, 0>(g, rho, j);
  lower_boundaries_on_moments<T, S, LAYOUT>(p, b, slot, own, rho, j);
}

template <typename T, class S, int LAYOUT, bool STREAM, bool MASKED, int VEC, int k, int COLL = 0, int ABBD = 0>
__device__ __forceinline__ void abb_outlet(const KParams<T> &p, int slot, int c0k, int c1,
                                           int c2, T (&f)[S::Q][VEC]) {
  const int ax = p.bt->mem_axis[slot], nbr = p.bt->nbr[slot];
  T rn, jn[3];
  neighbour_moments<T, S, LAYOUT, STREAM, MASKED, COLL, ABBD>(p, ax == 0 ? nbr : c0k, ax == 1 ? nbr : c1,
                                                               ax == 2 ? nbr : c2, slot, rn, jn);
  abb_apply<T, S, LAYOUT, VEC, k>(p, slot, rn, jn, f);
}

// The boundaries of one node in index order (lettuce/_simulation.py:183-188); b = the node's index in
// no_collision_mask, (c0k, c1, c2) its memory coordinates, ownk its index within a population.
// lane_slot != 0: (lane_rho, lane_j) are the moments of the node next to this one along a0 as outlet `lane_slot`
// sees them, handed over by the neighbouring lane (lbm_body) instead of being gathered again.
template <typename T, class S, int LAYOUT, bool STREAM, int VEC, int k, int COLL = 0, int ABBD = 0>
__device__ __forceinline__ void apply_boundaries(const KParams<T> &p, int b, int c0k, int c1, int c2,
                                                 unsigned ownk, T (&f)[S::Q][VEC], int lane_slot = 0,
                                                 T lane_rho = T(1), const T *lane_j = nullptr) {
  for (int slot = 1; slot <= p.nb; ++slot) {
    const int kind = p.bt->kind[slot];
    if (kind == kAbbOutlet) {
      // applies on the whole outlet plane, whatever the node's index: the reference
      // mutates flow.f in place and the masked torch.where is then a no-op
      // (anti_bounce_back_outlet.py:81-91, _simulation.py:186-188)
      const int ax = p.bt->mem_axis[slot];
      const int coord = ax == 0 ? c0k : (ax == 1 ? c1 : c2);
      if (coord == p.bt->plane[slot]) {
        if (slot == lane_slot) {
          const T jn[3] = {lane_j[0], lane_j[1], lane_j[2]};
          abb_apply<T, S, LAYOUT, VEC, k>(p, slot, lane_rho, jn, f);
        } else {
          abb_outlet<T, S, LAYOUT, STREAM, true, VEC, k, COLL, ABBD>(p, slot, c0k, c1, c2, f);
        }
      }
    } else if (b == slot) {
      if (kind == kBounceBack) {
        bounce_back<T, S, VEC, k>(f);
      } else if (kind == kEquilibrium) {
        const T *fld = p.bt->field[slot];
        static_for<S::Q>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          f[q][k] = fld ? fld[(long long)q * p.N + ownk] : p.bt->feq[slot][q];
        });
      }
    }
  }
}

// ---- the kernel ---------------------------------------------------------------------------
// TUNE bit 0: nontemporal loads, bit 1: nontemporal stores.  One thread per VEC nodes, the grid
// covers the work exactly (a capped grid with a grid-stride loop measured 7 % slower and cost
// 20-30 VGPRs in the KBC kernels).
// number of populations q' < q with the same velocity component along memory axis a2
template <class S, int LAYOUT, int q>
constexpr int crossing_rank() {
  int r = 0;
  for (int k = 0; k < q; ++k)
    if (MemMap<S, LAYOUT>::e(k, 2) == MemMap<S, LAYOUT>::e(q, 2)) ++r;
  return r;
}

// ABBD: plans with ABBD + 1 anti-bounce-back outlets (neighbour_moments, DEPTH)
template <typename T, class S, int LAYOUT, int COLL, bool STREAM, bool COLLIDE, bool MASKED,
          int VEC, int SHIFT, int TUNE = 0, bool PACK = false, int ABBD = 0>
__device__ __forceinline__ void lbm_body(const KParams<T> &p) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= p.nvec_total) return;
  {
  const unsigned rowid = v / (unsigned)p.nv0;
  const int c0 = (int)(v - rowid * (unsigned)p.nv0) * VEC;
  const int r2 = (int)(rowid / (unsigned)p.n1);
  const int c1 = (int)(rowid - (unsigned)r2 * (unsigned)p.n1);
  const int c2 = p.p_begin + r2 * p.p_stride;
  const Coord c = make_coord(p, c0, c1, c2);
  const unsigned own = (unsigned)(c2 * p.n1 + c1) * (unsigned)p.n0 + (unsigned)c0;

  T f[S::Q][VEC];
  gather<T, S, LAYOUT, STREAM, VEC, SHIFT, (TUNE & 1) != 0>(p, c, f);

  unsigned char nd[VEC];
  if constexpr (MASKED) {
    if constexpr (VEC == 1) {
      nd[0] = p.node[own];
    } else {
      const typename Vec<T, VEC>::mtype m =
          *reinterpret_cast<const typename Vec<T, VEC>::mtype *>(p.node + own);
#pragma unroll
      for (int k = 0; k < VEC; ++k) nd[k] = m[k];
    }
    if constexpr (STREAM) {
      static_for<VEC>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (nd[k] & 0x80) keep_unstreamed<T, S, VEC, k>(p, own, f);
      });
    }
  }

  // An outlet whose normal is the contiguous axis: the node next to an outlet node is held by the next lane of
  // the same wave (rows are whole waves), whose populations at this point -- pulled, no-streaming slots kept,
  // not yet collided -- are exactly what neighbour_moments would gather again (19-27 single-lane loads and
  // their latency in every wave that ends a row: 0.48 -> 0.60 ms at 512 x 512 x 64 with the Obstacle's outlet)
  int lane_slot = 0;
  T lane_rho = T(1), lane_j[3] = {T(0), T(0), T(0)};
  if constexpr (COLLIDE && MASKED && VEC == 1 && ABBD == 0) {
    if (p.abb0_slot != 0) {
      const int slot = p.abb0_slot, plane = p.bt->plane[slot];
      const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int first = c0 - lane;                                 // a0 coordinate of lane 0: same in all lanes
      if (plane >= first && plane < first + 64) {
        moments<T, S, LAYOUT, 1, 0>(f, lane_rho, lane_j);
        lower_boundaries_on_moments<T, S, LAYOUT>(p, nd[0] & 0x7f, slot, own, lane_rho, lane_j);
        const int from = (lane - p.bt->side[slot]) & 63;
        lane_rho = __shfl(lane_rho, from);
        lane_j[0] = __shfl(lane_j[0], from); lane_j[1] = __shfl(lane_j[1], from); lane_j[2] = __shfl(lane_j[2], from);
        lane_slot = slot;
      }
    }
  }

  if constexpr (COLLIDE) {
    static_for<VEC>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      int b = 0;
      if constexpr (MASKED) b = nd[k] & 0x7f;
      if (b == 0) {
        if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, VEC, k>(f, p.tau_inv);
        if constexpr (COLL == 2) collide_kbc<T, S, LAYOUT, VEC, k>(f, p.beta, p.inv_beta);
        if constexpr (COLL == 3) collide_bgk_fast<T, S, LAYOUT, VEC, k>(f, p.tau_inv);
      }
      if constexpr (MASKED)
        apply_boundaries<T, S, LAYOUT, STREAM, VEC, k, COLL, ABBD>(p, b, c0 + k, c1, c2, own + k, f, lane_slot,
                                                                    lane_rho, lane_j);
    });
  }

  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    vstore<T, VEC, (TUNE & 2) != 0>(p.out + (long long)q * p.No + own, f[q]);
  });
  if constexpr (PACK) {
    // halo packing fused into the boundary-plane launch of the slab driver (saves two pack
    // launches on the critical path of the exchange)
    using M = MemMap<S, LAYOUT>;
    const unsigned in_plane = (unsigned)c1 * (unsigned)p.n0 + (unsigned)c0;
    const unsigned plane_nodes = (unsigned)p.n1 * (unsigned)p.n0;
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e2 = M::e(q, 2);
      if constexpr (e2 != 0) {
        constexpr int rank = crossing_rank<S, LAYOUT, q>();
        T *buf = e2 < 0 ? p.pack_lo : p.pack_hi;
        const int plane = e2 < 0 ? p.pack_lo_plane : p.pack_hi_plane;
        if (c2 == plane) vstore<T, VEC, false>(buf + (size_t)rank * plane_nodes + in_plane, f[q]);
      }
    });
  }
  }
}

template <typename T, class S, int LAYOUT, int COLL, bool STREAM, bool COLLIDE, bool MASKED,
          int VEC, int SHIFT, int TUNE = 0, bool PACK = false, int ABBD = 0>
__global__ void __launch_bounds__(kThreads) lbm_kernel(const KParams<T> p) {
  lbm_body<T, S, LAYOUT, COLL, STREAM, COLLIDE, MASKED, VEC, SHIFT, TUNE, PACK, ABBD>(p);
}

// same kernel with the register allocator told to fit 4 waves per SIMD (<= 128 VGPRs): the masked
// D3Q27-KBC kernel sits at 131 VGPRs otherwise and loses a wave per SIMD (cfg4: 0.735 -> ms below)
template <typename T, class S, int LAYOUT, int COLL, bool STREAM, bool COLLIDE, bool MASKED,
          int VEC, int SHIFT, int TUNE = 0, bool PACK = false>
__global__ void __launch_bounds__(kThreads, 4) lbm_kernel_occ4(const KParams<T> p) {
  lbm_body<T, S, LAYOUT, COLL, STREAM, COLLIDE, MASKED, VEC, SHIFT, TUNE, PACK>(p);
}

// ---- two fused steps per launch (periodic, no masks) -----------------------------------------
// f*_out = (C S)^2 f*_in with the intermediate state held in LDS, so that HBM sees one read and one
// write of the populations per TWO lattice updates.  A workgroup owns a T0 x T1 column of nodes in
// (a0, a1) and sweeps seg_len planes along a2:
//   phase A(j):   every thread pulls one node of the (T0+2) x (T1+2) halo'd tile of plane j from
//                 global memory, collides it and writes its populations to LDS;
//   phase B(k):   the first T0*T1 threads pull their node of plane k from the LDS planes k-1, k,
//                 k+1, collide and store to global memory.
// B(k) reads the populations moving up (e2 = +1, "U") only from plane k-1, the in-plane ones ("C")
// only from plane k and those moving down ("D") only from plane k+1.  With 4 LDS slots for U, 3 for
// C and 2 for D -- 57 population planes, as many as three whole planes -- A(k+2) can write while
// other waves still read for B(k), and ONE barrier per plane is enough:
//   barrier; issue the LDS reads of B(k); collide A(k+2) -> LDS (the reads fly behind it); issue
//   the global loads of A(k+3) (they land during the next plane); collide B(k); store B(k).
// The barrier waits for LDS traffic only (an ordinary __syncthreads() would drain the prefetch); the
// loads are issued as early as the registers allow and before the stores, so that the vmcnt waits
// in front of the next A never meet stores issued just before them.  The order was found by
// measurement: each of these placements is worth 3-10 %.
// Arithmetic per node is the one-step kernel's (same pull, same collide): results are bit for bit
// those of two lbm_kernel launches.  Redundant work: (T0+2)(T1+2)/(T0 T1) in the first step and
// two extra planes per segment.  HBM traffic (PMC, 256^3): reads 1.05x one pass, writes 1.00x.
// NPT / NPB: intermediate / output nodes per thread (A/B variants, 1 is the product setting);
// PACK: slab edge launches that also write the halo message.
template <typename T, class S, int T0_, int T1>
struct TwoStep {
  static constexpr int T0 = T0_, H0 = T0 + 2, H1 = T1 + 2;
  static constexpr int NI = H0 * H1;                    // intermediate nodes per plane
  static constexpr int NO = T0 * T1;                    // output nodes per plane
  static constexpr int THREADS = (NI + 63) / 64 * 64;
  template <int LAYOUT, int E2>
  static constexpr int count() {                        // populations with e along a2 == E2
    int n = 0;
    for (int q = 0; q < S::Q; ++q) n += MemMap<S, LAYOUT>::e(q, 2) == E2 ? 1 : 0;
    return n;
  }
};

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// MODE (slab layout): 0 = plain sweep; 1 = edge launch: every workgroup also writes the halo messages, and the
// planes beyond the cuts are read from the receive buffers when p.ghost_lo / p.ghost_hi are given; 2 = launch
// over the whole slab whose edge workgroups start first and count themselves done (p.signal).
template <typename T, class S, int LAYOUT, int COLL, int T0_, int T1, int NPT = 1, int MODE = 0,
          int NPB = NPT>
__global__ void __launch_bounds__(((TwoStep<T, S, T0_, T1>::NI / NPT + 63) / 64 * 64))
lbm2_kernel(const KParams<T> p, const int seg_len) {
  constexpr bool PACK = MODE == 1;
  static_assert(MODE == 0 || LAYOUT == 1, "edge / signalling launches exist in the slab layout");
  // NPT intermediate nodes and NPB output nodes per thread (1 or 2): thread t owns intermediate
  // nodes t + k NA, k < NPT, and output nodes t + k NB, k < NPB
  using B = TwoStep<T, S, T0_, T1>;
  using M = MemMap<S, LAYOUT>;
  constexpr int T0 = B::T0, H0 = B::H0, NI = B::NI, NO = B::NO;
  constexpr int NA = NI / NPT, NB = NO / NPB;
  static_assert(NI % NPT == 0 && NO % NPB == 0, "nodes per thread must divide the tile");
  constexpr int NU = B::template count<LAYOUT, 1>(), NC = B::template count<LAYOUT, 0>(),
                ND = B::template count<LAYOUT, -1>();
  static_assert(COLL == 0 || COLL == 1 || COLL == 3, "two-step kernel: streaming only or BGK (exact / fast arithmetic)");
  __shared__ T lds_u[4][NU][NI];
  __shared__ T lds_c[3][NC][NI];
  __shared__ T lds_d[2][ND][NI];

  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / T0, tiles1 = p.n1 / T1;
  // Workgroups go to the 8 XCDs round-robin (block b -> XCD b % 8) and every XCD has its own L2.
  // Renumber so that an XCD owns a compact patch of neighbouring tiles: the halo rows two tiles
  // share are then fetched into one L2 once instead of into two L2s.
  int b = blockIdx.x;
  const int segs_a = (p.p_end - p.p_begin + seg_len - 1) / seg_len;
  if (MODE == 2) {
    // edges first (the hardware starts workgroups in index order): upper edge = the one segment of the second
    // range, then the first segment of the first range, then the rest with the XCD-aware numbering
    // (XCD-aware within each layer of tiles)
    const int tiles = tiles0 * tiles1;
    const int layer = b / tiles, t = b - layer * tiles;
    const int tile = tiles % 8 == 0 ? (t % 8) * (tiles / 8) + t / 8 : t;
    b = (layer == 0 ? segs_a : layer - 1) * tiles + tile;
  } else if (p.nb == 0 && (tiles0 * tiles1) % 8 == 0) {
    // every XCD gets an eighth of EVERY segment layer -- a compact patch of tiles -- rather than an eighth of
    // the grid: slab launches cut their plane range into segments of unequal length (64 planes as 62 + 2: four
    // XCDs had all the long workgroups, 1.00 instead of 0.55 ms), and with equal segments it is as good or
    // better (256^3: 0.511 / 0.537 / 0.532 against 0.512 / 0.572 / 0.579 ms with 128 / 64 / 32 planes)
    const int tiles = tiles0 * tiles1;
    const int layer = b / tiles, t = b - layer * tiles;
    b = layer * tiles + (t % 8) * (tiles / 8) + t / 8;
  } else if (p.nb != 1 && gridDim.x % 8 == 0) {
    b = (b % 8) * (gridDim.x / 8) + b / 8;           // A/B (nb = 2), or tiles that do not divide by 8
  }
  const int t0 = (b % tiles0) * T0; b /= tiles0;
  const int t1 = (b % tiles1) * T1; b /= tiles1;
  // first output plane of this workgroup: segments of the first range, then of the second one
  const bool second = b >= segs_a;
  const int range_end = second ? p.p_end2 : p.p_end;
  const int s = second ? p.p_begin2 + (b - segs_a) * seg_len : p.p_begin + b * seg_len;

  const bool in_a = tid < NA, in_b = tid < NB;
  // Addresses: the plane part is uniform (scalar registers, recomputed per plane), the in-plane
  // part is a per-thread constant -- nine byte offsets for the nine (e0, e1) pairs of the lattice.
  unsigned voff[NPT][3][3];                          // [k][e1 + 1][e0 + 1], bytes within a plane
  unsigned out_off[NPB];
  int a_at[NPT];                                     // LDS index of the intermediate node
  int b_at[NPB];                                     // LDS index of the output node incl. halo offset
  static_for<NPT>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    // phase A: node (i0, i1) of the halo'd tile, global coordinates (g0, g1).  The T0 inner columns
    // of a row go to T0 consecutive threads (a wave reads one aligned 256-byte row segment per
    // population), the two halo columns of all rows to the last threads.
    const int ia = tid + k * NA;
    int i1, i0;
    if (NPT == 1) {
      constexpr int inner = T0 * B::H1;
      i1 = ia < inner ? ia / T0 : (ia - inner) >> 1;
      i0 = ia < inner ? 1 + (ia - i1 * T0) : (((ia - inner) & 1) ? H0 - 1 : 0);
    } else {
      i1 = ia / H0; i0 = ia - i1 * H0;
    }
    a_at[k] = i1 * H0 + i0;
    int g0 = t0 + i0 - 1; g0 = g0 < 0 ? g0 + p.n0 : (g0 >= p.n0 ? g0 - p.n0 : g0);
    int g1 = t1 + i1 - 1; g1 = g1 < 0 ? g1 + p.n1 : (g1 >= p.n1 ? g1 - p.n1 : g1);
    const int g0m = g0 == 0 ? p.n0 - 1 : g0 - 1, g0p = g0 == p.n0 - 1 ? 0 : g0 + 1;
    const int g1m = g1 == 0 ? p.n1 - 1 : g1 - 1, g1p = g1 == p.n1 - 1 ? 0 : g1 + 1;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int y = a == 0 ? g1p : (a == 1 ? g1 : g1m);     // source = node - e
        const int x = c == 0 ? g0p : (c == 1 ? g0 : g0m);
        voff[k][a][c] = ((unsigned)y * (unsigned)p.n0 + (unsigned)x) * (unsigned)sizeof(T);
      }
  });
  static_for<NPB>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    // phase B: output node (j0, j1) of the tile
    const int ib = tid + k * NB;
    const int j1 = ib / T0, j0 = ib - j1 * T0;
    out_off[k] = ((unsigned)(t1 + j1) * (unsigned)p.n0 + (unsigned)(t0 + j0)) * (unsigned)sizeof(T);
    b_at[k] = (j1 + 1) * H0 + (j0 + 1);
  });
  const unsigned plane_nodes = (unsigned)p.n1 * (unsigned)p.n0;

  T pre[S::Q][NPT];
  auto load_a = [&](int plane) {
    // periodic along a2, or a slab whose ghost planes (two per side) hold the neighbours' data
    int g2 = plane, g2m = plane - 1, g2p = plane + 1;
    if (p.wrap2) {
      g2 = plane < 0 ? plane + p.n2 : (plane >= p.n2 ? plane - p.n2 : plane);
      g2m = g2 == 0 ? p.n2 - 1 : g2 - 1;
      g2p = g2 == p.n2 - 1 ? 0 : g2 + 1;
    }
    if (in_a) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2);
        const int z = e2 == 0 ? g2 : (e2 > 0 ? g2m : g2p);
        // 32-bit scalar multiply (a plane's first node index fits: N < 2^31), 64-bit scalar add
        const T *base = p.in + ((long long)q * p.Ni + (long long)((unsigned)z * plane_nodes));
        if constexpr (MODE == 1) {
          // Planes beyond a cut: the neighbour's populations as they arrived (halo2_kernel's message: in-plane
          // populations of its plane next to the cut | the crossing ones of that plane | the crossing ones of the
          // plane behind it).  The plane index is uniform, so this is scalar work; the populations moving away
          // from a cut are never pulled across it.
          constexpr int rank = crossing_rank<S, LAYOUT, q>();
          if constexpr (e2 >= 0) {
            if (p.ghost_lo != nullptr && z < p.lo)
              base = p.ghost_lo + (size_t)(e2 == 0 ? rank : (z == p.lo - 1 ? NC + rank : NC + NU + rank)) * plane_nodes;
          }
          if constexpr (e2 <= 0) {
            if (p.ghost_hi != nullptr && z >= p.hi)
              base = p.ghost_hi + (size_t)(e2 == 0 ? rank : (z == p.hi ? NC + rank : NC + ND + rank)) * plane_nodes;
          }
        }
        static_for<NPT>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          pre[q][k] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + voff[k][e1 + 1][e0 + 1]);
        });
      });
    }
  };
  // r = index of the plane relative to s - 1; r3 = r % 3
  auto compute_a = [&](int r, int r3) {
    if (in_a) {
      if constexpr (COLL == 1)
        static_for<NPT>([&](auto kc) { collide_bgk<T, S, LAYOUT, NPT, decltype(kc)::value>(pre, p.tau_inv); });
      if constexpr (COLL == 3)
        static_for<NPT>([&](auto kc) { collide_bgk_fast<T, S, LAYOUT, NPT, decltype(kc)::value>(pre, p.tau_inv); });
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        static_for<NPT>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (e2 > 0) lds_u[r & 3][rank][a_at[k]] = pre[q][k];
          else if constexpr (e2 == 0) lds_c[r3][rank][a_at[k]] = pre[q][k];
          else lds_d[r & 1][rank][a_at[k]] = pre[q][k];
        });
      });
    }
  };
  T f[S::Q][NPB];
  auto read_b = [&](int r, int r3) {                 // output plane with relative index r
    if (in_b) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        static_for<NPB>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const int at = b_at[k] - e1 * H0 - e0;
          if constexpr (e2 > 0) f[q][k] = lds_u[(r - 1) & 3][rank][at];
          else if constexpr (e2 == 0) f[q][k] = lds_c[r3][rank][at];
          else f[q][k] = lds_d[(r + 1) & 1][rank][at];
        });
      });
    }
  };
  auto collide_b = [&]() {
    if (in_b) {
      if constexpr (COLL == 1)
        static_for<NPB>([&](auto kc) { collide_bgk<T, S, LAYOUT, NPB, decltype(kc)::value>(f, p.tau_inv); });
      if constexpr (COLL == 3)
        static_for<NPB>([&](auto kc) { collide_bgk_fast<T, S, LAYOUT, NPB, decltype(kc)::value>(f, p.tau_inv); });
    }
  };
  // packing: this workgroup writes halo messages (PACK kernels; a launch that covers a whole slab runs the
  // sweep of its other workgroups without that code -- with it in the loop they took twice as long)
  // how: 0 = nontemporal stores; 1 = also the halo messages (PACK kernels); 2 = stores that are performed at
  // device scope (write-through: another XCD's kernel may read them while this launch still runs)
  auto store_b = [&](int k2, auto how) {
    constexpr int HOW = decltype(how)::value;
    constexpr bool PACKING = HOW == 1;
    if (in_b) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        T *base = p.out + ((long long)q * p.No + (long long)((unsigned)k2 * plane_nodes));
        static_for<NPB>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          T *at = reinterpret_cast<T *>(reinterpret_cast<char *>(base) + out_off[k]);
          if constexpr (HOW == 2) {
            using Bits = std::conditional_t<sizeof(T) == 4, unsigned, unsigned long long>;
            __hip_atomic_store(reinterpret_cast<Bits *>(at), __builtin_bit_cast(Bits, f[q][k]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          } else {
            __builtin_nontemporal_store(f[q][k], at);
          }
        });
        // Slab edge launches (PACK) also write the two-step halo message (layout of halo2_kernel: in-plane
        // populations of the plane next to the cut | its crossing populations | the crossing
        // populations of the plane behind it), possibly straight into the neighbour's memory.
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        if constexpr (PACKING) {
        if (p.pack_lo != nullptr && e2 <= 0) {
          const int d = k2 - p.pack_lo_plane;                    // 0: near plane, 1: far plane
          if (d == 0 || (d == 1 && e2 < 0)) {
            const int slot = e2 == 0 ? rank : (d == 0 ? NC + rank : NC + ND + rank);
            T *msg = p.pack_lo + (size_t)slot * plane_nodes;
            static_for<NPB>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              *reinterpret_cast<T *>(reinterpret_cast<char *>(msg) + out_off[k]) = f[q][k];
            });
          }
        }
        if (p.pack_hi != nullptr && e2 >= 0) {
          const int d = p.pack_hi_plane - k2;
          if (d == 0 || (d == 1 && e2 > 0)) {
            const int slot = e2 == 0 ? rank : (d == 0 ? NC + rank : NC + NU + rank);
            T *msg = p.pack_hi + (size_t)slot * plane_nodes;
            static_for<NPB>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              *reinterpret_cast<T *>(reinterpret_cast<char *>(msg) + out_off[k]) = f[q][k];
            });
          }
        }
        }
      });
    }
  };

  // intermediate planes s-1 .. s+seg_len are needed (relative indices 0 .. seg_len+1)
  const int last = s + seg_len < range_end ? s + seg_len : range_end;
  if constexpr (PACK && NPT == 1 && NPB == 1) {
    if (last - s == 2) {
      // Slab edge launch: two output planes per workgroup, i.e. four intermediate planes and no sweep to amortise a
      // serial prologue over, with all 256 workgroups of a round in lock-step (memory idle while they collide, compute
      // units idle while they load).  Straight-line schedule on two register sets with the loads of plane j + 1 in
      // flight behind the collide of plane j -- what the steady state of the sweep does.  Every thread loads (the 44
      // of 704 without an intermediate node read the plane's first node): a load under `if (in_a)` leaves the other
      // lanes' registers undefined, the compiler zeroes them AFTER the loads and that write waits for the loads.
      if (!in_a) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) voff[0][a][c] = 0u;
      }
      auto load_into = [&](int plane, T (&dst)[S::Q][1]) {
        const int g2 = plane, g2m = plane - 1, g2p = plane + 1;      // slab layout: no wrap along a2
        static_for<S::Q>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
          const int z = e2 == 0 ? g2 : (e2 > 0 ? g2m : g2p);
          const T *base = p.in + ((long long)q * p.Ni + (long long)((unsigned)z * plane_nodes));
          if constexpr (e2 >= 0) {
            if (p.ghost_lo != nullptr && z < p.lo)
              base = p.ghost_lo + (size_t)(e2 == 0 ? rank : (z == p.lo - 1 ? NC + rank : NC + NU + rank)) * plane_nodes;
          }
          if constexpr (e2 <= 0) {
            if (p.ghost_hi != nullptr && z >= p.hi)
              base = p.ghost_hi + (size_t)(e2 == 0 ? rank : (z == p.hi ? NC + rank : NC + ND + rank)) * plane_nodes;
          }
          dst[q][0] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + voff[0][e1 + 1][e0 + 1]);
        });
      };
      // KEEP: which populations of the plane anybody reads -- bit 0: those moving up (the output plane above pulls
      // them), bit 1: in-plane, bit 2: moving down.  The first intermediate plane only feeds the output plane above
      // it, the last one only the plane below: 38 of the 76 post-collision populations of the four planes are needed,
      // the others are neither stored nor (dead code to the compiler) computed.
      auto collide_into_lds = [&](T (&src)[S::Q][1], int r, int r3, auto keep) {
        constexpr int KEEP = decltype(keep)::value;
        if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(src, p.tau_inv);
        if constexpr (COLL == 3) collide_bgk_fast<T, S, LAYOUT, 1, 0>(src, p.tau_inv);
        if (in_a) {
          static_for<S::Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
            if constexpr (e2 > 0) { if constexpr (KEEP & 1) lds_u[r & 3][rank][a_at[0]] = src[q][0]; }
            else if constexpr (e2 == 0) { if constexpr (KEEP & 2) lds_c[r3][rank][a_at[0]] = src[q][0]; }
            else { if constexpr (KEEP & 4) lds_d[r & 1][rank][a_at[0]] = src[q][0]; }
          });
        }
      };
      using Up = std::integral_constant<int, 1>;
      using UpIn = std::integral_constant<int, 3>;
      using InDown = std::integral_constant<int, 6>;
      using Down = std::integral_constant<int, 4>;
      // (sched_barrier: the register-minimising scheduler of this unit sinks the loads behind the collide otherwise)
      T pre2[S::Q][1];
      load_into(s - 1, pre); load_into(s, pre2);
      __builtin_amdgcn_sched_barrier(0);
      collide_into_lds(pre, 0, 0, Up{});
      __builtin_amdgcn_sched_barrier(0);
      load_into(s + 1, pre);
      __builtin_amdgcn_sched_barrier(0);
      collide_into_lds(pre2, 1, 1, UpIn{});
      __builtin_amdgcn_sched_barrier(0);
      load_into(s + 2, pre2);
      __builtin_amdgcn_sched_barrier(0);
      collide_into_lds(pre, 2, 2, InDown{});
      lds_barrier();
      read_b(1, 1);
      collide_into_lds(pre2, 3, 0, Down{});
      collide_b();
      store_b(s, std::integral_constant<int, 1>{});
      lds_barrier();
      read_b(2, 2);
      collide_b();
      store_b(s + 1, std::integral_constant<int, 1>{});
      return;
    }
  }
  load_a(s - 1); compute_a(0, 0);
  load_a(s);     compute_a(1, 1);
  load_a(s + 1); compute_a(2, 2);
  if (s + 2 <= last) load_a(s + 2);
  int r = 1, r3 = 1;                                  // output plane k has relative index k - s + 1
  auto interval = [&](int k, auto how) {
    lds_barrier();                                    // planes up to k + 1 complete; reads of k - 1 done
    read_b(r, r3);                                    // 19 LDS reads in flight ...
    if (k + 2 <= last) {
      compute_a(r + 2, r3 == 0 ? 2 : r3 - 1);         // ... behind the collide of plane k + 2; (r + 2) % 3
      if (k + 3 <= last) load_a(k + 3);
    }
    collide_b();
    store_b(k, how);
    ++r;
    r3 = r3 == 2 ? 0 : r3 + 1;
  };
  int k = s;
  using Plain = std::integral_constant<int, 0>;
  if constexpr (PACK) {
    // edge launches: every workgroup also writes the halo messages.  (This copy of the loop is slow -- 12 spilled
    // registers, message stores -- and even its presence slows the other copy: a launch over the whole slab
    // with packing first segments took 1.0 instead of 0.6 ms, so that launch does not pack.)
    for (; k < last; ++k) interval(k, std::integral_constant<int, 1>{});
  } else if constexpr (MODE == 2) {
    if (second || b == 0) {
      // Launch over the whole slab: the planes next to a cut are this workgroup's last two (upper edge) or first
      // two (lower edge).  They are stored at device scope, and once the stores have been performed the
      // workgroup counts itself done.  No release fence: at device scope that is a write-back of the XCD's
      // whole L2 -- 1024 of them made the launch take 0.99 instead of 0.66 ms.
      const int until = second ? last : (s + 2 < last ? s + 2 : last);
      for (; k < until; ++k) interval(k, std::integral_constant<int, 2>{});
      // every wave waits until ITS edge-plane stores have been acknowledged (they are write-through stores at
      // device scope, so the acknowledgement means "performed in memory"): a workgroup-scope release alone emits
      // no vmcnt wait on gfx950, and the counter below must not become visible before the planes are
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.signal, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (; k < last; ++k) interval(k, Plain{});
}

// One wave on the communication stream: returns when *counter has reached `target` (the edge workgroups of
// the launch running on the compute stream have stored the planes next to the cuts), or after about a second, setting
// *timed_out -- an exit every launch reaches.  17 us from the last increment to the next kernel of the stream
// (tools/experiments/stream_wait.hip; hipStreamWaitValue64 needs signal memory, where 256 device atomics
// drained in 280 us).
static __global__ void wait_counter_kernel(const unsigned long long *counter, unsigned long long target,
                                           unsigned *timed_out) {
  // relaxed polls (one uncached load each): an ACQUIRE per poll would invalidate this XCD's L2 every
  // microsecond under the sweep that runs beside it -- measured 0.63 instead of 0.37 ms per step
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > 100000000ll) {           // 100 MHz
      *timed_out = 1u;
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// ---- up to KMAX steps per launch on small 2-D grids ---------------------------------------------
// A 128^2 grid is launch-bound (3.5 us per step against < 1 us of work).  Here a workgroup loads the
// (TO0 + 2(K-1)) x (TO1 + 2(K-1)) neighbourhood of its TO0 x TO1 tile, performs K stream-collide
// steps on it in LDS (two buffers, one __syncthreads per step; after step s only the nodes at
// distance >= s-1 from the border of the neighbourhood are still valid, after step K exactly the
// tile) and stores the tile.  The neighbourhood is recomputed by every workgroup that needs it --
// free while the GPU waits for launches, which is why lt_run only uses this on small grids.
// Same pull and same collide as the one-step kernel: K launches of lbm_kernel give the same bits.
template <int TO0, int TO1, int KMAX>
struct ManyStep2D {
  static constexpr int R0 = TO0 + 2 * (KMAX - 1), R1 = TO1 + 2 * (KMAX - 1);
  static constexpr int NR = R0 * R1;
  static constexpr int THREADS = (NR + 63) / 64 * 64;
};

// MASKED: plans with boundaries.  A thread keeps its node for all K steps, so the node byte, the no-streaming
// bits and -- on an equilibrium node -- the populations the boundary writes are fetched once.  A slot with a
// no-streaming bit keeps the node's own value of the step before (global memory in step 1, the LDS buffer
// afterwards).  The anti-bounce-back outlet needs (rho, j) of the node next to it as the one-step kernel sees
// them: the moments of that node's pulled populations, which are rebuilt from the same source as the thread's
// own pull (neighbour_moments in step 1, the LDS buffer afterwards).  That neighbour must itself be valid, so
// plans with an outlet recompute one more ring (halo = K instead of K - 1: at most KMAX - 1 steps per launch).
// Same functions in the same order as lbm_body: K launches of the masked lbm_kernel give the same bits.
template <typename T, class S, int COLL, int TO0, int TO1, int KMAX, bool MASKED = false>
__global__ void __launch_bounds__((ManyStep2D<TO0, TO1, KMAX>::THREADS))
lbm_many_kernel(const KParams<T> p, const int K) {
  static_assert(S::D == 2, "2-D lattices");
  using M = MemMap<S, 0>;
  using G = ManyStep2D<TO0, TO1, KMAX>;
  __shared__ T lds[2][S::Q][G::NR];
  const int tid = threadIdx.x;
  const int halo = K - 1 + (MASKED ? p.abb0_slot : 0);        // abb0_slot: 1 = the plan has an outlet
  const int r0 = TO0 + 2 * halo, r1 = TO1 + 2 * halo;        // neighbourhood of this launch
  const int tiles0 = p.n0 / TO0;
  const int t0 = (blockIdx.x % tiles0) * TO0, t1 = (blockIdx.x / tiles0) * TO1;
  const bool in_region = tid < r0 * r1;
  const int i1 = tid / r0, i0 = tid - i1 * r0;
  auto wrap = [](int x, int n) { x %= n; return x < 0 ? x + n : x; };
  const int g0 = wrap(t0 - halo + i0, p.n0), g1 = wrap(t1 - halo + i1, p.n1);
  const unsigned own = (unsigned)g1 * (unsigned)p.n0 + (unsigned)g0;
  auto collide = [&](T (&f)[S::Q][1]) {
    if constexpr (COLL == 1) collide_bgk<T, S, 0, 1, 0>(f, p.tau_inv);
    if constexpr (COLL == 2) collide_kbc<T, S, 0, 1, 0>(f, p.beta, p.inv_beta);
  };
  // ---- boundaries (MASKED) ----
  int bidx = 0;
  unsigned bits = 0;
  T eqv[S::Q];                                          // what this node's equilibrium boundary writes
  if constexpr (MASKED) {
    if (in_region) {
      const unsigned char nd = p.node[own];
      bidx = nd & 0x7f;
      if (nd & 0x80) bits = p.nsm_bits[own];
      if (bidx != 0 && p.bt->kind[bidx] == kEquilibrium) {
        const T *fld = p.bt->field[bidx];
        static_for<S::Q>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          eqv[q] = fld ? fld[(long long)q * p.N + own] : p.bt->feq[bidx][q];
        });
      }
    }
  }
  // What the boundaries do to this node is the same in every step: its own boundary (bounce-back / equilibrium,
  // index bidx) and, on the outlet's plane, the outlet (index out_slot) -- in index order.  Everything the
  // plan's table says about them is fetched once; for an outlet node also the index, node byte and
  // no-streaming bits of the node next to it.
  int my_kind = 0, out_slot = 0, out_axis = 0, out_side = 1;
  unsigned nown = 0, nbits = 0;
  int nbidx = 0;
  if constexpr (MASKED) {
    if (in_region) {
      if (bidx != 0) my_kind = p.bt->kind[bidx];
      if (p.abb0_slot) {
        for (int slot = 1; slot <= p.nb; ++slot)
          if (p.bt->kind[slot] == kAbbOutlet) {
            const int ax = p.bt->mem_axis[slot];
            if ((ax == 0 ? g0 : g1) == p.bt->plane[slot]) {
              out_slot = slot; out_axis = ax; out_side = p.bt->side[slot];
              const int ng0 = ax == 0 ? p.bt->nbr[slot] : g0, ng1 = ax == 1 ? p.bt->nbr[slot] : g1;
              nown = (unsigned)ng1 * (unsigned)p.n0 + (unsigned)ng0;
              const unsigned char nnd = p.node[nown];
              nbidx = nnd & 0x7f;
              nbits = (nnd & 0x80) ? p.nsm_bits[nown] : 0u;
            }
          }
      }
    }
  }
  // collision and the boundaries in index order; nbr(rho, j): moments of the node next to an outlet node
  auto collide_and_bound = [&](T (&f)[S::Q][1], auto &&nbr) {
    if constexpr (!MASKED) {
      collide(f);
    } else {
      if (bidx == 0) collide(f);
      auto outlet = [&]() {
        T rn, jn[3];
        nbr(rn, jn);
        if (out_axis == 0) abb_apply_ax<T, S, 0, 0>(out_side, rn, jn, f);
        else abb_apply_ax<T, S, 0, 1>(out_side, rn, jn, f);
      };
      // the outlet rewrites its whole plane, whatever the node's own index (apply_boundaries)
      if (out_slot != 0 && (bidx == 0 || out_slot <= bidx)) outlet();
      if (my_kind == kBounceBack) {
        bounce_back<T, S, 1, 0>(f);
      } else if (my_kind == kEquilibrium) {
        static_for<S::Q>([&](auto qc) { f[decltype(qc)::value][0] = eqv[decltype(qc)::value]; });
      }
      if (out_slot != 0 && bidx != 0 && out_slot > bidx) outlet();
    }
  };
  T f[S::Q][1];
  // step 1: pull from global memory, every node of the neighbourhood
  if (in_region) {
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
      const int s0 = e0 == 0 ? g0 : wrap(g0 - e0, p.n0), s1 = e1 == 0 ? g1 : wrap(g1 - e1, p.n1);
      f[q][0] = p.in[(long long)q * p.Ni + (long long)s1 * p.n0 + s0];
      if constexpr (MASKED && q > 0) {
        if (bits & (1u << q)) f[q][0] = p.in[(long long)q * p.Ni + own];
      }
    });
    collide_and_bound(f, [&](T &rn, T (&jn)[3]) {
      const int nb = p.bt->nbr[out_slot];
      neighbour_moments<T, S, 0, true, true, COLL, 0>(p, out_axis == 0 ? nb : g0, out_axis == 1 ? nb : g1, 0, out_slot,
                                                     rn, jn);
    });
  }
  for (int s = 1; s < K; ++s) {                 // f holds the state after step s
    const int buf = s & 1;
    if (in_region) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        lds[buf][q][i1 * G::R0 + i0] = f[q][0];
      });
    }
    __syncthreads();
    // step s + 1 is valid for nodes at distance >= s from the border
    const bool valid = in_region && i0 >= s && i0 < r0 - s && i1 >= s && i1 < r1 - s;
    if (valid) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
        f[q][0] = lds[buf][q][(i1 - e1) * G::R0 + (i0 - e0)];
        if constexpr (MASKED && q > 0) {
          if (bits & (1u << q)) f[q][0] = lds[buf][q][i1 * G::R0 + i0];
        }
      });
      collide_and_bound(f, [&](T &rn, T (&jn)[3]) {
        // the node next to this outlet node, inside the domain: its pull from the same LDS state
        const int n0i = out_axis == 0 ? i0 - out_side : i0, n1i = out_axis == 1 ? i1 - out_side : i1;
        T g[S::Q][1];
        static_for<S::Q>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
          g[q][0] = lds[buf][q][(n1i - e1) * G::R0 + (n0i - e0)];
          if constexpr (q > 0) {
            if (nbits & (1u << q)) g[q][0] = lds[buf][q][n1i * G::R0 + n0i];
          }
        });
        moments<T, S, 0, 1, 0>(g, rn, jn);
        lower_boundaries_on_moments<T, S, 0>(p, nbidx, out_slot, nown, rn, jn);
      });
    }
  }
  // after K steps the valid nodes are the tile
  if (in_region && i0 >= halo && i0 < r0 - halo && i1 >= halo && i1 < r1 - halo) {
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      p.out[(long long)q * p.No + own] = f[q][0];
    });
  }
}

// ---- two steps per launch on small 3-D grids -----------------------------------------------------
// The 3-D counterpart of lbm_many_kernel for grids that are bound by launch latency (32^3: 3.9 us per step for
// well under 1 us of work).  A workgroup of 1000 threads loads the 10^3 neighbourhood of its 8^3 tile (one node per
// thread: the ordinary pull from global memory + collide), leaves the populations in LDS (q x 1000 values: 76 KB
// for D3Q19 fp32, two workgroups per CU), and the 512 threads of the inner tile pull their second step from there,
// collide and store.  The shell is recomputed by every workgroup that needs it (1.95 x the arithmetic of the first
// step) -- free while the chip waits for launches.  Same pull, same collide as the one-step kernel: two launches of
// lbm_kernel give the same bits (BGK / streaming); grids smaller than the neighbourhood wrap (8^3).
constexpr int kMany3dTile = 8, kMany3dEdge = kMany3dTile + 2, kMany3dNodes = kMany3dEdge * kMany3dEdge * kMany3dEdge;
template <typename T, class S, int COLL>
__global__ void __launch_bounds__(1024) lbm_many3d_kernel(const KParams<T> p) {
  static_assert(S::D == 3, "3-D lattices");
  using M = MemMap<S, 0>;
  constexpr int E = kMany3dEdge, NR = kMany3dNodes, TO = kMany3dTile;
  __shared__ T lds[S::Q][NR];
  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / TO, tiles1 = p.n1 / TO;
  int b = blockIdx.x;
  const int t0 = (b % tiles0) * TO; b /= tiles0;
  const int t1 = (b % tiles1) * TO; b /= tiles1;
  const int t2 = b * TO;
  const bool in_region = tid < NR;
  const int i2 = tid / (E * E), i1 = (tid / E) % E, i0 = tid % E;
  auto wrap = [](int x, int n) { x %= n; return x < 0 ? x + n : x; };
  const int g0 = wrap(t0 - 1 + i0, p.n0), g1 = wrap(t1 - 1 + i1, p.n1), g2 = wrap(t2 - 1 + i2, p.n2);
  T f[S::Q][1];
  if (in_region) {
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2);
      const int s0 = e0 == 0 ? g0 : wrap(g0 - e0, p.n0), s1 = e1 == 0 ? g1 : wrap(g1 - e1, p.n1),
                s2 = e2 == 0 ? g2 : wrap(g2 - e2, p.n2);
      f[q][0] = p.in[(long long)q * p.Ni + ((long long)s2 * p.n1 + s1) * p.n0 + s0];
    });
    if constexpr (COLL == 1) collide_bgk<T, S, 0, 1, 0>(f, p.tau_inv);
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      lds[q][tid] = f[q][0];
    });
  }
  __syncthreads();
  // second step: the nodes of the tile (distance >= 1 from the border of the neighbourhood)
  if (in_region && i0 >= 1 && i0 <= TO && i1 >= 1 && i1 <= TO && i2 >= 1 && i2 <= TO) {
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2);
      f[q][0] = lds[q][tid - (e2 * E + e1) * E - e0];
    });
    if constexpr (COLL == 1) collide_bgk<T, S, 0, 1, 0>(f, p.tau_inv);
    const long long own = ((long long)g2 * p.n1 + g1) * p.n0 + g0;
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      p.out[(long long)q * p.No + own] = f[q][0];
    });
  }
}

// ---- auxiliary kernels --------------------------------------------------------------------
// rho [N], u [d][N] (logical axis order; u_stride elements between components) from f  -- Flow.rho / Flow.u
template <typename T, class S, int LAYOUT>
__global__ void __launch_bounds__(kThreads) macroscopic_kernel(const T *__restrict__ f,
                                                               T *__restrict__ rho_out,
                                                               T *__restrict__ u_out,
                                                               long long N, long long stride,
                                                               long long u_stride) {
  using M = MemMap<S, LAYOUT>;
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  T g[S::Q][1];
  static_for<S::Q>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    g[q][0] = f[(long long)q * stride + i];
  });
  T rho, j[3];
  moments<T, S, LAYOUT, 1, 0>(g, rho, j);
  if (rho_out) rho_out[i] = rho;
  if (u_out) {
#pragma unroll
    for (int a = 0; a < S::D; ++a) u_out[(long long)a * u_stride + i] = j[M::memory(a)] / rho;
  }
}

// feq [q][N] from rho [N], u [d][N]  -- QuadraticEquilibrium.__call__
template <typename T, class S, int LAYOUT>
__global__ void __launch_bounds__(kThreads) equilibrium_kernel(const T *__restrict__ rho_in,
                                                               const T *__restrict__ u_in,
                                                               T *__restrict__ feq_out,
                                                               long long N) {
  using M = MemMap<S, LAYOUT>;
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  const T rho = rho_in[i];
  T u[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int a = 0; a < S::D; ++a) u[M::memory(a)] = u_in[(long long)a * N + i];
  const T uxu = square_norm<S, LAYOUT>(u);
  for_each_feq<T, S, LAYOUT>(rho, u, uxu, [&](auto qc, T v) {
    feq_out[(long long)decltype(qc)::value * N + i] = v;
  });
}

// f = feq(rho, u) - w_q Pi1:Q_q  -- initialize_f_neq (lettuce/_flow.py:309-336), reference layout,
// periodic.  S[a][b] = d u_a / d x_b: torch_gradient's 6th-order central differences (dx = 1), term
// order of the reference's expression; Pi1 = ((1.0 tau) rho) S / cs^2; Q_q,ab = e_qa e_qb - eye_cs2 d_ab.
template <typename T, class S>
__global__ void __launch_bounds__(kThreads) fneq_kernel(const T *__restrict__ rho_in, const T *__restrict__ u_in,
                                                       T *__restrict__ f_out, int n0, int n1, int n2, T tau,
                                                       T eye_cs2) {
#pragma clang fp contract(off)
  using M = MemMap<S, 0>;
  constexpr int D = S::D;
  const long long N = (long long)n0 * n1 * n2;
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  const int c0 = (int)(i % n0), c1 = (int)((i / n0) % n1), c2 = (int)(i / ((long long)n0 * n1));
  const T w6[6] = {T(-1. / 60.), T(3. / 20.), T(-3. / 4.), T(3. / 4.), T(-3. / 20.), T(1. / 60.)};
  const int sh[6] = {3, 2, 1, -1, -2, -3};
  T grad[D][D];                                   // [component a][logical axis b]
#pragma unroll
  for (int b = 0; b < D; ++b) {
    const int m = D - 1 - b;                      // memory axis of logical axis b
    long long at[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      int a0 = c0, a1 = c1, a2 = c2;
      if (m == 0) { a0 = c0 - sh[k]; a0 = a0 < 0 ? a0 + n0 : (a0 >= n0 ? a0 - n0 : a0); }
      if (m == 1) { a1 = c1 - sh[k]; a1 = a1 < 0 ? a1 + n1 : (a1 >= n1 ? a1 - n1 : a1); }
      if (m == 2) { a2 = c2 - sh[k]; a2 = a2 < 0 ? a2 + n2 : (a2 >= n2 ? a2 - n2 : a2); }
      at[k] = ((long long)a2 * n1 + a1) * n0 + a0;
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const T *uc = u_in + (long long)a * N;
      T r = w6[0] * uc[at[0]];
#pragma unroll
      for (int k = 1; k < 6; ++k) r = r + w6[k] * uc[at[k]];
      grad[a][b] = r;
    }
  }
  const T rho = rho_in[i];
  const T scale = (T(1.0) * tau) * rho;
  const T cs2 = (T)kCs2;
  T pi[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) pi[a][b] = scale * grad[a][b] / cs2;
  T u[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int a = 0; a < D; ++a) u[M::memory(a)] = u_in[(long long)a * N + i];
  const T uxu = square_norm<S, 0>(u);
  for_each_feq<T, S, 0>(rho, u, uxu, [&](auto qc, T feq) {
    constexpr int q = decltype(qc)::value;
    T acc = T(0);
    static_for<D>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      static_for<D>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        constexpr int ee = S::E[q][a] * S::E[q][b];
        const T qab = a == b ? T(ee) - eye_cs2 : T(ee);
        acc = acc + pi[a][b] * qab;
      });
    });
    f_out[(long long)q * N + i] = feq - T(S::W[q]) * acc;
  });
}

// wavefront (64-lane) + workgroup reduction of a double (sum, or max when MAX); result valid in
// thread 0
template <bool MAX = false>
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double part[kThreads / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off);
    v = MAX ? (o > v ? o : v) : v + o;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s = MAX ? (part[w] > s ? part[w] : s) : s + part[w];
  }
  return s;
}

// per-block partial sums of 0.5*u.u (MODE 0) or of sum_q f (MODE 1), or per-block maximum of |u|
// (MODE 2), over the planes
// [p_begin, p_begin + planes) of a2; fixed grid -> fixed summation order.
template <typename T, class S, int LAYOUT, int MODE>
__global__ void __launch_bounds__(kThreads) reduce_kernel(const T *__restrict__ f, long long N,
                                                          long long first, long long count,
                                                          double *__restrict__ partial) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count;
       i += (long long)gridDim.x * kThreads) {
    T g[S::Q][1];
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      g[q][0] = f[(long long)q * N + first + i];
    });
    T rho, j[3];
    moments<T, S, LAYOUT, 1, 0>(g, rho, j);
    if constexpr (MODE == 0) {
      const T uu[3] = {j[0] / rho, j[1] / rho, j[2] / rho};
      acc += (double)(T(0.5) * square_norm<S, LAYOUT>(uu));
    } else if constexpr (MODE == 1) {
      acc += (double)rho;
    } else {
      const T ux = j[0] / rho, uy = j[1] / rho, uz = j[2] / rho;
      const double m = (double)sqrt(ux * ux + uy * uy + uz * uz);
      acc = m > acc ? m : acc;
    }
  }
  const double s = block_sum<MODE == 2>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Enstrophy (observable_reporter.py:45-68) from the velocity field u [d][N] (lattice units, logical
// component order, reference layout): per node the 6th-order periodic central differences of
// torch_gradient (util/utility.py:37-99) of u_pu = scale * u, the squared vorticity, fp64 partial sums.
// d(u_c)/d(axis): sum_k w_k u_c(x - s_k e_axis), s = 3, 2, 1, -1, -2, -3, times 1 / dx -- the order of
// the reference's expression (roll by +s reads x - s).
//
// SLAB: u is a rank's velocity field in the slab layout, [3][n2][n1][n0] with x fastest (logical axis a on memory axis
// a) and n2 = the rank's planes + THREE planes of the neighbours on either side; the sum runs over the planes
// [3, n2 - 3) and nothing wraps along a2.
template <typename T, int D, bool SLAB = false>
__global__ void __launch_bounds__(kThreads) enstrophy_kernel(const T *__restrict__ u, int n0, int n1, int n2,
                                                            T scale, T inv_dx, double *__restrict__ partial) {
#pragma clang fp contract(off)
  const long long N = (long long)n0 * n1 * n2;
  const long long first = SLAB ? 3ll * n0 * n1 : 0ll, count = SLAB ? N - 2 * first : N;
  const T w[6] = {T(-1. / 60.), T(3. / 20.), T(-3. / 4.), T(3. / 4.), T(-3. / 20.), T(1. / 60.)};
  const int sh[6] = {3, 2, 1, -1, -2, -3};
  double acc = 0.0;
  for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < count; k += (long long)gridDim.x * kThreads) {
    const long long i = first + k;
    const int c0 = (int)(i % n0), c1 = (int)((i / n0) % n1), c2 = (int)(i / ((long long)n0 * n1));
    // derivative of component c along MEMORY axis m
    auto ddx = [&](int c, int m) -> T {
      const T *uc = u + (long long)c * N;
      T r = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        int a0 = c0, a1 = c1, a2 = c2;
        if (m == 0) { a0 = c0 - sh[k]; a0 = a0 < 0 ? a0 + n0 : (a0 >= n0 ? a0 - n0 : a0); }
        if (m == 1) { a1 = c1 - sh[k]; a1 = a1 < 0 ? a1 + n1 : (a1 >= n1 ? a1 - n1 : a1); }
        if (m == 2) { a2 = c2 - sh[k]; if (!SLAB) a2 = a2 < 0 ? a2 + n2 : (a2 >= n2 ? a2 - n2 : a2); }
        const T v = w[k] * (uc[((long long)a2 * n1 + a1) * n0 + a0] * scale);
        r = k == 0 ? v : r + v;
      }
      return r * inv_dx;
    };
    // logical axis a lives on memory axis D - 1 - a (reference layout) / a (slab layout)
    auto grad = [&](int c, int a) -> T { return ddx(c, SLAB ? a : D - 1 - a); };
    const T wz = grad(0, 1) - grad(1, 0);
    T node = wz * wz;
    if constexpr (D == 3) {
      const T wx = grad(2, 1) - grad(1, 2), wy = grad(0, 2) - grad(2, 0);
      node = node + (wx * wx + wy * wy);
    }
    acc += (double)node;
  }
  const double s = block_sum<false>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Mass observable (observable_reporter.py:140-158): sum of all populations over the nodes off the first /
// last index of the two fastest axes, minus the populations of the nodes flagged by `mask` (anywhere)
template <typename T, int Q>
__global__ void __launch_bounds__(kThreads) interior_mass_kernel(const T *__restrict__ f, long long N, int n0, int n1,
                                                                const unsigned char *__restrict__ mask,
                                                                double *__restrict__ partial) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < N; i += (long long)gridDim.x * kThreads) {
    const int c0 = (int)(i % n0), c1 = (int)((i / n0) % n1);
    const bool inner = c0 > 0 && c0 < n0 - 1 && c1 > 0 && c1 < n1 - 1;
    const bool masked = mask != nullptr && mask[i] != 0;
    if (!inner && !masked) continue;
    double node = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) node += (double)f[(long long)q * N + i];
    acc += (inner ? node : 0.0) - (masked ? node : 0.0);
  }
  const double s = block_sum<false>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// The same over a rank's slab (slab layout, `stride` elements between populations): nodes [first, first + count) are
// the rank's own planes, plane `first / (n0 n1)` is plane z_begin of the nz_global planes of the whole grid; the
// reference's two fastest axes are y and z, i.e. a1 and the GLOBAL a2 here.  `mask` is indexed like the nodes of f.
template <typename T, int Q>
__global__ void __launch_bounds__(kThreads) interior_mass_slab_kernel(const T *__restrict__ f, long long stride,
                                                                     long long first, long long count, int n0, int n1,
                                                                     int z_begin, int nz_global,
                                                                     const unsigned char *__restrict__ mask,
                                                                     double *__restrict__ partial) {
  double acc = 0.0;
  for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < count; k += (long long)gridDim.x * kThreads) {
    const long long i = first + k;
    const int c1 = (int)((i / n0) % n1), z = z_begin + (int)(k / ((long long)n0 * n1));
    const bool inner = c1 > 0 && c1 < n1 - 1 && z > 0 && z < nz_global - 1;
    const bool masked = mask != nullptr && mask[i] != 0;
    if (!inner && !masked) continue;
    double node = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) node += (double)f[(long long)q * stride + i];
    acc += (inner ? node : 0.0) - (masked ? node : 0.0);
  }
  const double s = block_sum<false>(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <bool MAX>
static __global__ void __launch_bounds__(kThreads) finish_sum_kernel(const double *__restrict__ partial,
                                                              int n, double *__restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) acc = MAX ? (partial[i] > acc ? partial[i] : acc) : acc + partial[i];
  const double s = block_sum<MAX>(acc);
  if (threadIdx.x == 0) *out = s;
}

// halo pack / unpack of the slab driver: the populations that cross a z cut (5 of 19, 9 of 27)
// of one a2 plane <-> one contiguous buffer [n][n1*n0], so that a ghost exchange is a single
// send and a single receive per direction
struct QList {
  int n;
  int q[9];
};
template <typename T, bool PACK>
__global__ void __launch_bounds__(kThreads) plane_pack_kernel(T *__restrict__ f, T *__restrict__ buf,
                                                              long long N, long long plane_off,
                                                              int plane_nodes, QList ql) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= plane_nodes) return;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    if (k < ql.n) {
      T *slot = f + (long long)ql.q[k] * N + plane_off + i;
      if (PACK) buf[(long long)k * plane_nodes + i] = *slot;
      else *slot = buf[(long long)k * plane_nodes + i];
    }
  }
}

// Halo message of the two-step slab driver: [in-plane populations of the plane next to the cut |
// crossing populations of that plane | crossing populations of the plane behind it | plans with masks: the
// populations of the near plane that move AWAY from the cut, which a no-streaming node of the ghost plane
// keeps], each a contiguous block of plane_nodes values.  PACK: f -> buf, else buf -> f.
template <typename T, bool PACK>
__global__ void __launch_bounds__(kThreads) halo2_kernel(T *__restrict__ f, T *__restrict__ buf, long long N,
                                                         long long off_near, long long off_far,
                                                         int plane_nodes, QList in_plane, QList cross, QList away) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= plane_nodes) return;
  auto move = [&](int slot, int q, long long off) {
    T *at = f + (long long)q * N + off + i;
    if (PACK) buf[(long long)slot * plane_nodes + i] = *at;
    else *at = buf[(long long)slot * plane_nodes + i];
  };
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    if (k < in_plane.n) move(k, in_plane.q[k], off_near);
    if (k < cross.n) {
      move(in_plane.n + k, cross.q[k], off_near);
      move(in_plane.n + cross.n + k, cross.q[k], off_far);
    }
    if (k < away.n) move(in_plane.n + 2 * cross.n + k, away.q[k], off_near);
  }
}

// node descriptor byte + sparse streaming-mask bits from the reference's two mask tensors
// The masked two-step kernel's admission tests (twostep_masked.hpp), collected in *mismatch:
//  bit 0: the no-streaming bits of a node differ from `expected` on the outlet -- a2 plane `plane` (axis = 2)
//         or a0 column `plane` (axis = 0) -- or from zero elsewhere (axis < 0: no bits anywhere);
//  bit 1: a node of a0 column `face` (>= 0: the face opposite an a0 outlet) is not an equilibrium node
//         (eq_slots: bit s set = boundary s is an EquilibriumBoundaryPU).
static __global__ void __launch_bounds__(kThreads) compile_masks_kernel(
    const unsigned char *__restrict__ ncm, const unsigned char *__restrict__ nsm, int q,
    long long N, unsigned char *__restrict__ node, unsigned *__restrict__ bits, long long plane_nodes, int n0,
    int axis, int plane, unsigned expected, int face, unsigned eq_slots, unsigned *__restrict__ mismatch) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  unsigned b = 0;
  if (nsm) {
    for (int k = 1; k < q; ++k)
      if (nsm[(long long)k * N + i] == 1) b |= 1u << k;
  }
  const int slot = ncm ? (ncm[i] & 0x7f) : 0;
  node[i] = (unsigned char)(slot | (b ? 0x80 : 0));
  if (bits) bits[i] = b;
  const int c0 = (int)(i % n0);
  const bool on_outlet = axis == 2 ? i / plane_nodes == plane : (axis == 0 ? c0 == plane : false);
  unsigned bad = b != (on_outlet ? expected : 0u) ? 1u : 0u;
  if (face >= 0 && c0 == face && !((eq_slots >> slot) & 1u)) bad |= 2u;
  if (bad) atomicOr(mismatch, bad);
}

}  // namespace lt
