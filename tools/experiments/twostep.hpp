// Two lattice updates per launch, second generation (see kernels.hpp, lbm2_kernel, for the scheme:
// a workgroup sweeps a T0 x T1 column of nodes along a2, phase A pulls + collides the halo'd tile of
// an intermediate plane into LDS, phase B pulls the output nodes from three LDS planes, collides and
// stores).  Same LDS slots, same thread -> node maps and the same arithmetic per node as lbm2_kernel
// -- results are bit for bit those of two lbm_kernel launches -- but the memory side is rebuilt:
//
//  * Buffer addressing.  Every global access is `buffer_load/store_dword v, voffset, s[desc], soffset`:
//    one descriptor per field, soffset = the plane (uniform), voffset = a per-thread loop constant
//    (population + in-plane offset, 32 bits).  No address arithmetic in the loop; lbm2_kernel kept 19
//    per-thread 64-bit addresses and spent one 64-bit vector add per load.  (The population cannot go
//    into soffset: 19 more live scalars and hipcc moves some of them to VGPRs and wraps the loads in
//    readfirstlane loops.)  Fields of 4 GiB and more keep lbm2_kernel.
//  * Roles per wave, decided on scalar registers: waves that hold output nodes run a loop with
//    phases A and B, the others a loop with phase A only, and phase B has no exec test (its waves are
//    full).  hipcc places `s_waitcnt vmcnt(N)` from the memory operations that may be pending on
//    ANY path into a block; an `if (thread has an output node)` around the stores made "no stores
//    issued" one of those paths, so the wait in front of the first collide assumed the loads were
//    the youngest operations: vmcnt(18), which in the waves that did store also waits for stores of
//    the previous plane, whose acknowledgements come late.
//  * For the same reason the steady-state loop is peeled once and has no conditions inside: the
//    prologue (loads pending, no stores) and the back edge (loads, then stores) would otherwise merge
//    at the loop header.  With the first iteration peeled both edges carry "19 loads, 19 stores" and
//    the waits name the loads only (vmcnt(37) ...).
//  * ORDER: which of the two jobs of a barrier interval a wave does first (A(k+2) may write its LDS
//    slots while other waves still read for B(k), so the order is free per wave):
//      0  all waves A first     1  all waves B first
//      2  waves 4..7 B first (the SIMD partners of waves 0..3), the others A first
//      3  odd waves B first
#pragma once
#include "kernels.hpp"

namespace lt {

typedef unsigned bufu2 __attribute__((ext_vector_type(2)));

template <typename T> struct BufIO;
template <> struct BufIO<float> {
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
  }
  static __device__ __forceinline__ void store_nt(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 2);
  }
};
template <> struct BufIO<double> {
  static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
  }
  static __device__ __forceinline__ void store_nt(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bufu2, v), r, (int)voff, (int)soff, 2);
  }
};

// descriptor of a whole population field [q][N]: raw buffer (stride 0).  The population and the plane go
// into the scalar offset of the access, so the host checks q * N * sizeof(T) < 2^32.
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t field_rsrc(const T *field, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(field), 0, (int)bytes, 0x00020000);
}

template <typename T, class S, int LAYOUT, int COLL, int T0_, int T1, int ORDER = 0, bool PACK = false, int PF = 1, int DBG = 0>
__global__ void __launch_bounds__((TwoStep<T, S, T0_, T1>::THREADS))
lbm2v_kernel(const KParams<T> p, const int seg_len) {
  using B = TwoStep<T, S, T0_, T1>;
  using M = MemMap<S, LAYOUT>;
  constexpr int T0 = B::T0, H0 = B::H0, NI = B::NI, NO = B::NO;
  constexpr int NU = B::template count<LAYOUT, 1>(), NC = B::template count<LAYOUT, 0>(),
                ND = B::template count<LAYOUT, -1>();
  static_assert(COLL == 0 || COLL == 1, "two-step kernel: streaming only or BGK");
  static_assert(NO % 64 == 0, "the output nodes of a tile fill whole waves");
  __shared__ T lds_u[4][NU][NI];
  __shared__ T lds_c[3][NC][NI];
  __shared__ T lds_d[2][ND][NI];

  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / T0, tiles1 = p.n1 / T1;
  // an XCD (blocks b, b + 8, ...) owns a compact patch of neighbouring tiles: shared halo rows are
  // fetched into one L2 once
  int b = blockIdx.x;
  if (p.nb == 0 && gridDim.x % 8 == 0) b = (b % 8) * (gridDim.x / 8) + b / 8;
  const int t0 = (b % tiles0) * T0; b /= tiles0;
  const int t1 = (b % tiles1) * T1; b /= tiles1;
  const int segs_a = (p.p_end - p.p_begin + seg_len - 1) / seg_len;
  const bool second = b >= segs_a;
  const int range_end = second ? p.p_end2 : p.p_end;
  const int s = second ? p.p_begin2 + (b - segs_a) * seg_len : p.p_begin + b * seg_len;

  const bool in_a = tid < NI;
  const unsigned pop_bytes = (unsigned)(p.N * (long long)sizeof(T));   // the host checked Q * N * sizeof(T) < 2^32
  // per-thread byte offsets of the source slot of every population (phase A) and of the output slot
  // (phase B) relative to the first node of the plane in population 0: loop constants in VGPRs, the
  // plane is the scalar offset of the access
  unsigned voff[S::Q], out_off[S::Q];
  int a_at, b_at;
  {
    // phase A: the T0 inner columns of a row go to T0 consecutive threads (one aligned 256-byte row
    // segment per wave and population), the two halo columns of all rows to the last threads
    constexpr int inner = T0 * B::H1;
    const int i1 = tid < inner ? tid / T0 : (tid - inner) >> 1;
    const int i0 = tid < inner ? 1 + (tid - i1 * T0) : (((tid - inner) & 1) ? H0 - 1 : 0);
    a_at = i1 * H0 + i0;
    int g0 = t0 + i0 - 1; g0 = g0 < 0 ? g0 + p.n0 : (g0 >= p.n0 ? g0 - p.n0 : g0);
    int g1 = t1 + i1 - 1; g1 = g1 < 0 ? g1 + p.n1 : (g1 >= p.n1 ? g1 - p.n1 : g1);
    const int g0m = g0 == 0 ? p.n0 - 1 : g0 - 1, g0p = g0 == p.n0 - 1 ? 0 : g0 + 1;
    const int g1m = g1 == 0 ? p.n1 - 1 : g1 - 1, g1p = g1 == p.n1 - 1 ? 0 : g1 + 1;
    // phase B (threads below NO): output node (j0, j1) of the tile
    const int j1 = tid / T0, j0 = tid - j1 * T0;
    const unsigned own = ((unsigned)(t1 + j1) * (unsigned)p.n0 + (unsigned)(t0 + j0)) * (unsigned)sizeof(T);
    b_at = (j1 + 1) * H0 + (j0 + 1);
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
      const int y = e1 == 0 ? g1 : (e1 > 0 ? g1m : g1p);      // source = node - e
      const int x = e0 == 0 ? g0 : (e0 > 0 ? g0m : g0p);
      voff[q] = ((unsigned)y * (unsigned)p.n0 + (unsigned)x) * (unsigned)sizeof(T) + (unsigned)q * pop_bytes;
      out_off[q] = own + (unsigned)q * pop_bytes;
    });
  }
  const unsigned plane_nodes = (unsigned)p.n1 * (unsigned)p.n0;
  const unsigned plane_bytes = plane_nodes * (unsigned)sizeof(T);
  const __amdgpu_buffer_rsrc_t in_r = field_rsrc(p.in, (unsigned)S::Q * pop_bytes),
                               out_r = field_rsrc(p.out, (unsigned)S::Q * pop_bytes);

  // PF = planes of loads in flight per wave: the loads of plane k + 2 + PF are issued in interval k
  // (PF = 2: two register sets, used alternately, the loop is unrolled by two)
  static_assert(PF == 1 || PF == 2, "prefetch depth");
  T pre[PF][S::Q][1];
  auto load_a = [&](int plane, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    // periodic along a2, or a slab whose ghost planes (two per side) hold the neighbours' data
    int g2 = plane, g2m = plane - 1, g2p = plane + 1;
    if (p.wrap2) {
      g2 = plane < 0 ? plane + p.n2 : (plane >= p.n2 ? plane - p.n2 : plane);
      g2m = g2 == 0 ? p.n2 - 1 : g2 - 1;
      g2p = g2 == p.n2 - 1 ? 0 : g2 + 1;
    }
    const unsigned off0 = (unsigned)g2 * plane_bytes, offm = (unsigned)g2m * plane_bytes,
                   offp = (unsigned)g2p * plane_bytes;
    if (DBG & 16) return;                            // (timing experiments: tools/experiments)
    if constexpr (DBG & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((DBG & 1) ? tid < T0 * B::H1 : in_a) {
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e2 = M::e(q, 2);
        pre[BUF][q][0] = BufIO<T>::load(in_r, voff[q], e2 == 0 ? off0 : (e2 > 0 ? offm : offp));
      });
    }
  };
  // r = index of the plane relative to s - 1; r3 = r % 3
  auto compute_a = [&](int r, int r3, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    if (in_a) {
      if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(pre[BUF], p.tau_inv);
      if constexpr ((DBG & 2) == 0)
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        if constexpr (e2 > 0) lds_u[r & 3][rank][a_at] = pre[BUF][q][0];
        else if constexpr (e2 == 0) lds_c[r3][rank][a_at] = pre[BUF][q][0];
        else lds_d[r & 1][rank][a_at] = pre[BUF][q][0];
      });
    }
  };
  // phase B runs in whole waves (every lane has an output node): no exec test around it, so every
  // wave of a role issues the same sequence of memory operations and the vmcnt waits can be exact
  T f[S::Q][1];
  auto read_b = [&](int r, int r3) {                 // output plane with relative index r
    if constexpr (DBG & 2) {
      static_for<S::Q>([&](auto qc) { f[decltype(qc)::value][0] = pre[0][decltype(qc)::value][0]; });
      return;
    }
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
      const int at = b_at - e1 * H0 - e0;
      if constexpr (e2 > 0) f[q][0] = lds_u[(r - 1) & 3][rank][at];
      else if constexpr (e2 == 0) f[q][0] = lds_c[r3][rank][at];
      else f[q][0] = lds_d[(r + 1) & 1][rank][at];
    });
  };
  auto collide_b = [&]() {
    if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(f, p.tau_inv);
  };
  auto store_b = [&](int k2) {
    const unsigned off = (unsigned)k2 * plane_bytes;
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr ((DBG & 8) == 0) BufIO<T>::store_nt(f[q][0], out_r, out_off[q], off);
      // Slab edge launches (PACK) also write the two-step halo message (layout of halo2_kernel: in-plane
      // populations of the plane next to the cut | its crossing populations | the crossing
      // populations of the plane behind it), possibly straight into the neighbour's memory.
      if constexpr (PACK) {
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        if (p.pack_lo != nullptr && e2 <= 0) {
          const int d = k2 - p.pack_lo_plane;                    // 0: near plane, 1: far plane
          if (d == 0 || (d == 1 && e2 < 0)) {
            const int slot = e2 == 0 ? rank : (d == 0 ? NC + rank : NC + ND + rank);
            T *msg = p.pack_lo + (size_t)slot * plane_nodes;
            *reinterpret_cast<T *>(reinterpret_cast<char *>(msg) + out_off[0]) = f[q][0];
          }
        }
        if (p.pack_hi != nullptr && e2 >= 0) {
          const int d = p.pack_hi_plane - k2;
          if (d == 0 || (d == 1 && e2 > 0)) {
            const int slot = e2 == 0 ? rank : (d == 0 ? NC + rank : NC + NU + rank);
            T *msg = p.pack_hi + (size_t)slot * plane_nodes;
            *reinterpret_cast<T *>(reinterpret_cast<char *>(msg) + out_off[0]) = f[q][0];
          }
        }
      }
    });
  };

  // intermediate planes s-1 .. s+seg_len are needed (relative indices 0 .. seg_len+1)
  const int last = s + seg_len < range_end ? s + seg_len : range_end;

  // The sweep of one wave.  HAS_B: the wave holds output nodes (waves below NO / 64); B_FIRST: it does
  // B(k) before A(k + 2) in a barrier interval.  Both are compile-time here, so that every copy of the
  // loop has ONE sequence of memory operations per interval.
  auto sweep = [&](auto has_b, auto b_first_c) {
    constexpr bool HAS_B = decltype(has_b)::value, B_FIRST = decltype(b_first_c)::value;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, PF - 1>;
    load_a(s - 1, B0{}); compute_a(0, 0, B0{});
    load_a(s, B0{});     compute_a(1, 1, B0{});
    load_a(s + 1, B0{}); compute_a(2, 2, B0{});
    if (s + 2 <= last) load_a(s + 2, B0{});
    if (PF == 2 && s + 3 <= last) load_a(s + 3, B1{});
    int r = 1, r3 = 1;                              // output plane k has relative index k - s + 1
    // one barrier interval: B(k) and, while planes are left, A(k + 2) and the loads of plane k + 2 + PF.
    // FULL: k + 2 + PF <= last is known (the steady state); else the two conditions are tested.
    // BUF: the register set that holds plane k + 2 and receives plane k + 2 + PF: (k - s) % PF.
    auto interval = [&](auto full, auto bufc, int k) {
      constexpr bool FULL = decltype(full)::value;
      const bool do_a = FULL || k + 2 <= last, do_l = FULL || k + 2 + PF <= last;
      if constexpr ((DBG & 4) == 0) lds_barrier();  // planes up to k + 1 complete; reads of k - 1 done
      if constexpr (DBG & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (DBG & 128) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
      if constexpr (HAS_B && B_FIRST) {
        read_b(r, r3);
        collide_b();
        store_b(k);
      }
      if constexpr (HAS_B && !B_FIRST) read_b(r, r3);   // the LDS reads are in flight behind the collide of A
      if (do_a) {
        compute_a(r + 2, r3 == 0 ? 2 : r3 - 1, bufc);   // (r + 2) % 3
        if (do_l) load_a(k + 2 + PF, bufc);
      }
      if constexpr (HAS_B && !B_FIRST) {
        collide_b();
        store_b(k);
      }
      ++r;
      r3 = r3 == 2 ? 0 : r3 + 1;
    };
    int k = s;
    if constexpr (PF == 1) {
      if (last - s >= 4) {
        interval(std::true_type{}, B0{}, k++);      // peeled: see the header of this file
        for (; k + 3 <= last; ++k) interval(std::true_type{}, B0{}, k);
      }
      for (; k < last; ++k) interval(std::false_type{}, B0{}, k);
    } else {
      if (last - s >= 7) {
        interval(std::true_type{}, B0{}, k++);      // peeled pair
        interval(std::true_type{}, B1{}, k++);
        for (; k + 5 <= last; k += 2) {
          interval(std::true_type{}, B0{}, k);
          interval(std::true_type{}, B1{}, k + 1);
        }
      }
      for (; k < last; k += 2) {
        interval(std::false_type{}, B0{}, k);
        if (k + 1 < last) interval(std::false_type{}, B1{}, k + 1);
      }
    }
  };

  // roles are uniform per wave: branch on scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // DBG 256: start / end time of every workgroup (100 MHz constant clock) into the buffer behind p.nsm_bits,
  // plus the XCC id: how evenly do the 256 workgroups of a launch finish?
  unsigned long long t_start = 0;
  if constexpr (DBG & 256) t_start = __builtin_amdgcn_s_memrealtime();
  struct Stamp {
    const KParams<T> &p; unsigned long long t0; int tid;
    __device__ ~Stamp() {
      if constexpr (DBG & 256) {
        if (tid == 0) {
          unsigned long long *buf = reinterpret_cast<unsigned long long *>(const_cast<unsigned *>(p.nsm_bits));
          unsigned xcc;
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
          buf[3 * blockIdx.x] = t0;
          buf[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
          buf[3 * blockIdx.x + 2] = xcc & 0xf;
        }
      }
    }
  } stamp{p, t_start, tid};
  if (wave >= NO / 64) {
    sweep(std::false_type{}, std::false_type{});
  } else {
    const bool b_first = ORDER == 1 || (ORDER == 2 && wave >= 4) || (ORDER == 3 && (wave & 1) != 0);
    if constexpr (ORDER == 0) sweep(std::true_type{}, std::false_type{});
    else if constexpr (ORDER == 1) sweep(std::true_type{}, std::true_type{});
    else if (b_first) sweep(std::true_type{}, std::true_type{});
    else sweep(std::true_type{}, std::false_type{});
  }
}

}  // namespace lt
