// Two lattice updates per launch for plans WITH boundaries (no_collision_mask / no_streaming_mask compiled
// to one byte per node; bounce-back, equilibrium and one anti-bounce-back outlet).  The scheme is
// lbm2_kernel's (kernels.hpp): a workgroup sweeps a T0 x T1 column of nodes along a2, phase A pulls +
// collides + applies the boundaries on the halo'd tile of an intermediate plane into LDS, phase B does the
// same for the output nodes from three LDS planes.  Results are bit for bit those of two masked
// lbm_kernel launches.
//
// Boundary work is kept to what is node-local or uniform per plane, so that the plane loop stays the
// unmasked kernel's plus a byte load and a few compares:
//  * bounce-back and equilibrium nodes: a register permutation / an overwrite from the table or field;
//  * the outlet is admitted only at the LAST plane of the sweep axis (memory axis a2, side +1: the
//    reference's Obstacle in the reference layout).  Then (rho, j) of the node next to the outlet plane
//    were computed by the same thread one plane earlier, in both phases, and are kept in four registers;
//    and the no-streaming bits of the outlet -- the downward populations of every node of that plane --
//    become a test on the plane index: at the outlet plane the downward populations are taken from the
//    node itself (phase A: the input field; phase B: the intermediate populations in LDS, which is why
//    the downward populations have a third LDS slot: plane k must survive interval k).
//  * or (AX = 0) at the first / last node of the rows, i.e. with its normal along the contiguous axis a0 (the
//    Obstacle in the slab layout, where x is contiguous): the node next to the outlet is then the
//    neighbouring LANE of the same wave, in both phases, so its (rho, j) arrive by a lane shuffle; the
//    outlet's no-streaming bits -- the populations entering through the outlet, on every node of it -- are
//    a per-thread constant: those populations are read from the node itself (phase A: the source offset of
//    the thread is set up that way; phase B: the node's own LDS slot).  The grid wraps along a0, so the
//    outlet column is also the halo column of the tiles at the other end of the rows, whose threads have no
//    neighbour lane: the host admits such a plan only if every node of the face opposite the outlet is an
//    equilibrium (inlet) node, which ignores what it pulls.
//  The host admits a plan only if its masks have exactly that shape (api.hip, masked_two_step_ok; the
//  no-streaming bits and the inlet face are checked on the device when the masks are compiled); anything
//  else -- outlets along a1, stray no-streaming bits, KBC -- keeps the one-step kernel.
//
// Memory side (differs from lbm2_kernel because boundary code needs registers: lbm2_kernel + masks spilled):
//  * buffer addressing: `buffer_load/store_dword v, voffset, s[desc], soffset` with one descriptor per
//    field, soffset = the plane (uniform), voffset = a per-thread loop constant (population + in-plane
//    offset, 32 bits): no address arithmetic in the loop.  Fields of 4 GiB and more run the BIG instantiation: one
//    descriptor per POPULATION, formed on the scalar unit next to each access (base + q * population stride), the
//    population term leaves voffset -- populations of 4 GiB and more are refused;
//  * roles per wave on scalar registers: waves that hold output nodes run a loop with phases A and B, the
//    others a loop with phase A only, and phase B has no exec test (its waves are full), so that every
//    copy of the loop issues ONE sequence of memory operations per interval and hipcc's `s_waitcnt
//    vmcnt` can name the loads without waiting for the stores of the previous plane; for the same reason
//    the steady-state loop is peeled once and has no conditions inside.
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

// descriptor of a whole population field [q][N]: raw buffer (stride 0); the host checks q * N * sizeof(T) < 2^32
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t field_rsrc(const T *field, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(field), 0, (int)bytes, 0x00020000);
}

// LDS bytes of lbm2m_kernel: 4 slots of the upward, 3 of the in-plane, 3 of the downward populations, and the
// populations of two uniform equilibrium boundaries
constexpr int kEqCached = 2;
template <typename T, class S, int LAYOUT, int T0, int T1>
constexpr size_t two_step_masked_lds() {
  using B = TwoStep<T, S, T0, T1>;
  return sizeof(T) * ((size_t)B::NI * (4 * B::template count<LAYOUT, 1>() + 3 * B::template count<LAYOUT, 0>() +
                                       3 * B::template count<LAYOUT, -1>()) + (size_t)kEqCached * S::Q);
}

// kinds of the plan's boundaries, two bits per slot, and the outlet's slot / plane (slot 0: no outlet)
struct MaskedPlanInfo {
  unsigned kinds;
  unsigned fields;                  // bit s: equilibrium boundary s has a per-node field
  int eq0, eq1;                     // the first two uniform equilibrium boundaries: their populations sit in LDS (0: none).
                                    // Two scalars, not an array: an array indexed by a variable makes the whole struct
                                    // an alloca, which hipcc "promotes" to 32 bytes of LDS per thread and reads back
                                    // into VECTOR registers -- every test on the plan then looks divergent
  int abb_slot, abb_side, abb_plane, abb_axis;
};
template <typename T>
__device__ __forceinline__ MaskedPlanInfo masked_plan_info(const KParams<T> &p) {
  MaskedPlanInfo m = {0u, 0u, 0, 0, 0, 1, -1, 2};
  int cached = 0;
  for (int slot = 1; slot <= p.nb; ++slot) {
    const int kind = p.bt->kind[slot];
    m.kinds |= (unsigned)kind << (2 * slot);
    if (kind == kEquilibrium) {
      if (p.bt->field[slot]) m.fields |= 1u << slot;
      else if (cached == 0) { m.eq0 = slot; cached = 1; }
      else if (cached == 1) { m.eq1 = slot; cached = 2; }
    }
    if (kind == kAbbOutlet) {
      m.abb_slot = slot; m.abb_side = p.bt->side[slot]; m.abb_plane = p.bt->plane[slot];
      m.abb_axis = p.bt->mem_axis[slot];
    }
  }
  return m;
}

// AX: memory axis of the outlet's normal, 2 or 0 (plans without an outlet run the AX = 2 kernel)
// BIG: fields of 4 GiB and more (see "Memory side" above)
template <typename T, class S, int LAYOUT, int COLL, int T0_, int T1, int AX = 2, bool BIG = false>
__global__ void __launch_bounds__((TwoStep<T, S, T0_, T1>::THREADS))
lbm2m_kernel(const KParams<T> p, const int seg_len) {
  static_assert(AX == 0 || AX == 2, "outlet along the sweep axis or along the rows");
  using B = TwoStep<T, S, T0_, T1>;
  using M = MemMap<S, LAYOUT>;
  constexpr int T0 = B::T0, H0 = B::H0, NI = B::NI, NO = B::NO;
  constexpr int NU = B::template count<LAYOUT, 1>(), NC = B::template count<LAYOUT, 0>(),
                ND = B::template count<LAYOUT, -1>();
  static_assert(COLL == 0 || COLL == 1 || COLL == 2, "two-step kernel: streaming only, BGK or (experiment) KBC");
  static_assert(NO % 64 == 0, "the output nodes of a tile fill whole waves");
  // (D3Q27: the collision + boundary code of one node is ~3000 instructions and the sweep below inlines it fifteen
  // times -- 300 KB of code against a 64 KB instruction cache.  A compact form with one copy of the prologue plane and
  // one of the interval per role, 105 KB, measured 5 % SLOWER in round 3: the peeled steady-state loop earns its size.)
  __shared__ T lds_u[4][NU][NI];
  __shared__ T lds_c[3][NC][NI];
  __shared__ T lds_d[3][ND][NI];
  // A uniform equilibrium boundary writes the same Q values on each of its nodes.  Read from the plan's table
  // in global memory they are a dependent load in the middle of every plane whose tile touches the boundary
  // -- with an inlet FACE along the rows (slab layout) that is every plane of an eighth of the workgroups,
  // measured 0.30 -> 0.41 ms per update at 512 x 512 x 64: the first two such boundaries are kept in LDS.
  __shared__ T lds_feq[kEqCached][S::Q];

  const int tid = threadIdx.x;
  const int tiles0 = p.n0 / T0, tiles1 = p.n1 / T1;
  // an XCD (blocks b, b + 8, ...) owns a compact patch of neighbouring tiles: shared halo rows are
  // fetched into one L2 once
  int b = blockIdx.x;
  if ((tiles0 * tiles1) % 8 == 0) {
    // an eighth of every segment layer per XCD, not an eighth of the grid: with segments of unequal length
    // (slab launches) some XCDs would get only the short last segments (see lbm2_kernel)
    const int tiles = tiles0 * tiles1;
    const int layer = b / tiles, t = b - layer * tiles;
    b = layer * tiles + (t % 8) * (tiles / 8) + t / 8;
  } else if (gridDim.x % 8 == 0) {
    b = (b % 8) * (gridDim.x / 8) + b / 8;
  }
  const int t0 = (b % tiles0) * T0; b /= tiles0;
  const int t1 = (b % tiles1) * T1; b /= tiles1;
  const int s = p.p_begin + b * seg_len;           // first output plane of this workgroup

  const bool in_a = tid < NI;
  // BIG: the population's share of an address is in its descriptor (in_of / out_of below), not in the 32-bit offsets
  const unsigned pop_bytes = BIG ? 0u : (unsigned)(p.Ni * (long long)sizeof(T)),
                 pop_bytes_out = BIG ? 0u : (unsigned)(p.No * (long long)sizeof(T));
  const MaskedPlanInfo info = masked_plan_info(p);
  if (tid < kEqCached * S::Q) {
    const int c = tid / S::Q, slot = c == 0 ? info.eq0 : info.eq1;
    lds_feq[c][tid - c * S::Q] = slot ? p.bt->feq[slot][tid - c * S::Q] : T(0);
  }
  lds_barrier();
  // AX = 0: is this thread's intermediate (a_out) / output (b_out) node on the outlet?  tile_out (uniform):
  // does the tile hold the outlet column among its inner columns?
  bool a_out = false, b_out = false;
  const bool tile_out = AX == 0 && info.abb_slot != 0 && info.abb_plane >= t0 && info.abb_plane < t0 + T0;
  // per-thread byte offsets of the source slot of every population (phase A) and of the output slot
  // (phase B) relative to the first node of the plane in population 0
  unsigned voff[S::Q], out_off[S::Q];
  unsigned a_own, b_own;                             // node index within a plane
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
    a_own = (unsigned)g1 * (unsigned)p.n0 + (unsigned)g0;
    const int g0m = g0 == 0 ? p.n0 - 1 : g0 - 1, g0p = g0 == p.n0 - 1 ? 0 : g0 + 1;
    const int g1m = g1 == 0 ? p.n1 - 1 : g1 - 1, g1p = g1 == p.n1 - 1 ? 0 : g1 + 1;
    // phase B (threads below NO): output node (j0, j1) of the tile
    const int j1 = tid / T0, j0 = tid - j1 * T0;
    b_own = (unsigned)(t1 + j1) * (unsigned)p.n0 + (unsigned)(t0 + j0);
    b_at = (j1 + 1) * H0 + (j0 + 1);
    if constexpr (AX == 0) {
      a_out = tile_out && tid < inner && g0 == info.abb_plane;
      b_out = tile_out && t0 + j0 == info.abb_plane;
    }
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1);
      // the populations entering through an a0 outlet are not streamed: their source is the node itself
      const bool keep = AX == 0 && e0 != 0 && a_out && e0 == -info.abb_side;
      const int y = e1 == 0 || keep ? g1 : (e1 > 0 ? g1m : g1p);      // source = node - e
      const int x = e0 == 0 || keep ? g0 : (e0 > 0 ? g0m : g0p);
      voff[q] = ((unsigned)y * (unsigned)p.n0 + (unsigned)x) * (unsigned)sizeof(T) + (unsigned)q * pop_bytes;
      out_off[q] = b_own * (unsigned)sizeof(T) + (unsigned)q * pop_bytes_out;
    });
  }
  const unsigned plane_nodes = (unsigned)p.n1 * (unsigned)p.n0;
  const unsigned plane_bytes = plane_nodes * (unsigned)sizeof(T);
  const __amdgpu_buffer_rsrc_t in_r = field_rsrc(p.in, BIG ? 0u : (unsigned)S::Q * pop_bytes),
                               out_r = field_rsrc(p.out, BIG ? 0u : (unsigned)S::Q * pop_bytes_out);
  // BIG: descriptor of population q AT the plane with byte offset `soff` (the plane goes into the base, soffset is 0:
  // a descriptor that does not depend on the plane is hoisted out of the sweep -- 2 x Q of them, four scalar registers
  // each, 520-570 scalar spills -- while this one is four scalar instructions next to its access)
  auto load_from = [&](auto qc, unsigned voffset, unsigned soff) __attribute__((always_inline)) {
    if constexpr (BIG) {
      const T *base = reinterpret_cast<const T *>(reinterpret_cast<const char *>(p.in + (long long)decltype(qc)::value * p.Ni) + soff);
      return BufIO<T>::load(field_rsrc(base, plane_bytes), voffset, 0u);
    } else {
      return BufIO<T>::load(in_r, voffset, soff);
    }
  };
  auto store_to = [&](auto qc, T v, unsigned voffset, unsigned soff) __attribute__((always_inline)) {
    if constexpr (BIG) {
      T *base = reinterpret_cast<T *>(reinterpret_cast<char *>(p.out + (long long)decltype(qc)::value * p.No) + soff);
      BufIO<T>::store_nt(v, field_rsrc(base, plane_bytes), voffset, 0u);
    } else {
      BufIO<T>::store_nt(v, out_r, voffset, soff);
    }
  };
  const bool abb_a2 = AX == 2 && info.abb_slot != 0;                  // outlet at a plane of the sweep axis
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int nbr_lane = (lane - info.abb_side) & 63;                    // AX = 0: who holds the node next to mine

  // The plane index is uniform, but hipcc does not always see it (in the D3Q27 instantiations it kept it in a vector
  // register, multiplied it there and wrapped every buffer access of a plane -- whose plane offset is the SCALAR
  // operand -- in a read-first-lane loop: 27 loops per phase).  Saying so keeps plane arithmetic on the scalar unit.
  auto uniform = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  auto wrapped = [&](int plane) __attribute__((always_inline)) {
    return uniform(p.wrap2 ? (plane < 0 ? plane + p.n2 : (plane >= p.n2 ? plane - p.n2 : plane)) : plane);
  };

  // collision, then the boundaries in index order, on the post-streaming populations g of the node `own`
  // of plane g2 whose node byte is nd; (rn, jn): moments of the same column one plane earlier
  auto collide_and_bound = [&](T (&g)[S::Q][1], int nd, bool on_outlet, unsigned own, T rn, const T (&jn)[3])
                               __attribute__((always_inline)) {
    const int bidx = nd & 0x7f;
    if (bidx == 0) {
      if constexpr (COLL == 1) collide_bgk<T, S, LAYOUT, 1, 0>(g, p.tau_inv);
      if constexpr (COLL == 2) collide_kbc<T, S, LAYOUT, 1, 0>(g, p.beta, p.inv_beta);
    }
    if (on_outlet && (bidx == 0 || info.abb_slot <= bidx)) abb_apply_ax<T, S, LAYOUT, AX>(info.abb_side, rn, jn, g);
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
          static_for<S::Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            g[q][0] = lds_feq[c][q];
          });
        } else {
          static_for<S::Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            g[q][0] = p.bt->feq[bidx][q];
          });
        }
      }
      if (on_outlet && info.abb_slot > bidx) abb_apply_ax<T, S, LAYOUT, AX>(info.abb_side, rn, jn, g);
    }
  };
  // (rho, j) of a node as the outlet one plane further sees it: the moments of its post-streaming
  // populations (collision conserves them) after the boundaries with an index below the outlet's
  auto moments_for_outlet = [&](const T (&g)[S::Q][1], int nd, unsigned own, T &rho, T (&j)[3])
                                __attribute__((always_inline)) {
    moments<T, S, LAYOUT, 1, 0>(g, rho, j);
    lower_boundaries_on_moments<T, S, LAYOUT>(p, nd & 0x7f, info.abb_slot, own, rho, j);
  };
  // the plane next to the outlet plane (uniform test); -2: never
  const int abb_nbr = abb_a2 ? info.abb_plane - info.abb_side : -2;
  // AX = 0: (rho, j) of the node next to the outlet node, from the lane that holds it
  auto from_neighbour_lane = [&](T &rho, T (&j)[3]) __attribute__((always_inline)) {
    rho = __shfl(rho, nbr_lane);
    j[0] = __shfl(j[0], nbr_lane); j[1] = __shfl(j[1], nbr_lane); j[2] = __shfl(j[2], nbr_lane);
  };

  T pre[S::Q][1];
  int nd_pre = 0;                                    // node byte of the intermediate node being loaded
  T sa_rho = T(1), sa_j[3] = {T(0), T(0), T(0)};      // phase A: moments of this column one plane earlier
  auto load_a = [&](int plane) __attribute__((always_inline)) {
    const int g2 = wrapped(plane);
    int g2m = uniform(plane - 1), g2p = uniform(plane + 1);
    if (p.wrap2) {
      g2m = g2 == 0 ? p.n2 - 1 : g2 - 1;
      g2p = g2 == p.n2 - 1 ? 0 : g2 + 1;
    }
    // (the products too: hipcc sometimes forms them on the vector unit, and a vector soffset means a loop per load)
    const unsigned off0 = (unsigned)uniform((int)((unsigned)g2 * plane_bytes)),
                   offm = (unsigned)uniform((int)((unsigned)g2m * plane_bytes));
    // at an a2 outlet plane the downward populations are not streamed: they come from the node itself
    const bool keep_down = abb_a2 && g2 == info.abb_plane;
    const unsigned offp = (unsigned)uniform((int)((unsigned)(keep_down ? g2 : g2p) * plane_bytes));
    if (in_a) {
      nd_pre = p.node[(unsigned)g2 * plane_nodes + a_own];
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e0 = M::e(q, 0), e2 = M::e(q, 2);
        if constexpr (AX == 0 && e0 != 0 && e2 != 0) {
          // an a0 outlet node keeps the populations entering through the outlet: voff[q] already points at
          // the node within a plane, and the plane is its own instead of the one below / above
          const bool keep = a_out && e0 == -info.abb_side;
          if constexpr (BIG) {
            // (the plane is a per-lane choice here: the one case where it has to ride in voffset -- of a descriptor
            // that spans the population)
            const T *base = p.in + (long long)q * p.Ni;
            pre[q][0] = BufIO<T>::load(field_rsrc(base, (unsigned)(p.Ni * (long long)sizeof(T))),
                                       voff[q] + (keep ? off0 : (e2 > 0 ? offm : offp)), 0u);
          } else {
            pre[q][0] = BufIO<T>::load(in_r, voff[q] + (keep ? off0 : (e2 > 0 ? offm : offp)), 0u);
          }
        } else if constexpr (e2 < 0) {
          const unsigned v = keep_down ? a_own * (unsigned)sizeof(T) + (unsigned)q * pop_bytes : voff[q];
          pre[q][0] = load_from(qc, v, offp);
        } else {
          pre[q][0] = load_from(qc, voff[q], e2 == 0 ? off0 : offm);
        }
      });
    }
  };
  // r = index of the plane relative to s - 1; r3 = r % 3; plane = its a2 index
  auto compute_a = [&](int r_, int r3_, int plane) __attribute__((always_inline)) {
    const int r = uniform(r_), r3 = uniform(r3_);
    if (in_a) {
      const int g2 = wrapped(plane);
      const unsigned own = (unsigned)g2 * plane_nodes + a_own;
      T keep_rho = T(1), keep_j[3] = {T(0), T(0), T(0)};
      if constexpr (AX == 0) {
        if (tile_out) {                               // uniform; the rows' waves are full
          moments_for_outlet(pre, nd_pre, own, keep_rho, keep_j);
          from_neighbour_lane(keep_rho, keep_j);
        }
        collide_and_bound(pre, nd_pre, a_out, own, keep_rho, keep_j);
      } else {
        if (g2 == abb_nbr) moments_for_outlet(pre, nd_pre, own, keep_rho, keep_j);
        collide_and_bound(pre, nd_pre, abb_a2 && g2 == info.abb_plane, own, sa_rho, sa_j);
        sa_rho = keep_rho; sa_j[0] = keep_j[0]; sa_j[1] = keep_j[1]; sa_j[2] = keep_j[2];
      }
      static_for<S::Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
        if constexpr (e2 > 0) lds_u[r & 3][rank][a_at] = pre[q][0];
        else if constexpr (e2 == 0) lds_c[r3][rank][a_at] = pre[q][0];
        else lds_d[r3][rank][a_at] = pre[q][0];
      });
    }
  };
  // phase B runs in whole waves (every lane has an output node): no exec test around it
  T f[S::Q][1];
  int nd_b = 0, nd_b_next = 0;                       // node byte of the output node of this / the next interval
  T sb_rho = T(1), sb_j[3] = {T(0), T(0), T(0)};      // phase B: moments of this column one plane earlier
  // post-streaming populations of the output node of plane k2 (relative index r) from the intermediate state:
  // upward populations from plane r - 1, in-plane ones from r, downward ones from r + 1 -- or, at the
  // outlet plane, from the node itself
  auto read_b = [&](int r_, int r3_, int k2_) __attribute__((always_inline)) {
    const int r = uniform(r_), r3 = uniform(r3_), k2 = uniform(k2_);
    const bool keep_down = abb_a2 && k2 == info.abb_plane;
    const int dslot = keep_down ? r3 : (r3 == 2 ? 0 : r3 + 1);
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int e0 = M::e(q, 0), e1 = M::e(q, 1), e2 = M::e(q, 2), rank = crossing_rank<S, LAYOUT, q>();
      const int at = b_at - e1 * H0 - e0;
      if constexpr (AX == 0 && e0 != 0) {
        // an a0 outlet node keeps the populations entering through the outlet: its own intermediate value
        const bool keep = b_out && e0 == -info.abb_side;
        if constexpr (e2 > 0) f[q][0] = lds_u[(keep ? r : r - 1) & 3][rank][keep ? b_at : at];
        else if constexpr (e2 == 0) f[q][0] = lds_c[r3][rank][keep ? b_at : at];
        else f[q][0] = lds_d[keep ? r3 : dslot][rank][keep ? b_at : at];
      } else {
        if constexpr (e2 > 0) f[q][0] = lds_u[(r - 1) & 3][rank][at];
        else if constexpr (e2 == 0) f[q][0] = lds_c[r3][rank][at];
        else f[q][0] = lds_d[dslot][rank][keep_down ? b_at : at];
      }
    });
  };
  auto finish_b = [&](int k2_) __attribute__((always_inline)) {
    const int k2 = uniform(k2_);
    const unsigned own = (unsigned)k2 * plane_nodes + b_own;
    T keep_rho = T(1), keep_j[3] = {T(0), T(0), T(0)};
    if constexpr (AX == 0) {
      if (tile_out) {
        moments_for_outlet(f, nd_b, own, keep_rho, keep_j);
        from_neighbour_lane(keep_rho, keep_j);
      }
      collide_and_bound(f, nd_b, b_out, own, keep_rho, keep_j);
    } else {
      if (k2 == abb_nbr) moments_for_outlet(f, nd_b, own, keep_rho, keep_j);
      collide_and_bound(f, nd_b, abb_a2 && k2 == info.abb_plane, own, sb_rho, sb_j);
      sb_rho = keep_rho; sb_j[0] = keep_j[0]; sb_j[1] = keep_j[1]; sb_j[2] = keep_j[2];
    }
    const unsigned off = (unsigned)uniform((int)((unsigned)k2 * plane_bytes));
    static_for<S::Q>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      store_to(qc, f[q][0], out_off[q], off);
    });
  };

  // intermediate planes s-1 .. s+seg_len are needed (relative indices 0 .. seg_len+1)
  const int last = s + seg_len < p.p_end ? s + seg_len : p.p_end;

  // The sweep of one wave.  HAS_B: the wave holds output nodes (waves below NO / 64).
  const bool b_wave = __builtin_amdgcn_readfirstlane(tid >> 6) < NO / 64;
  auto sweep = [&](auto has_b) __attribute__((always_inline)) {
    constexpr bool HAS_B = decltype(has_b)::value;
    if (abb_a2 && wrapped(s - 1) == info.abb_plane) {
      // the sweep opens ON the outlet plane (plane -1 of a periodic grid): the moments of the plane before
      // it, which the outlet needs, have not been met yet
      load_a(s - 2);
      if (in_a) moments_for_outlet(pre, nd_pre, (unsigned)wrapped(s - 2) * plane_nodes + a_own, sa_rho, sa_j);
    }
    load_a(s - 1); compute_a(0, 0, s - 1);
    load_a(s);     compute_a(1, 1, s);
    load_a(s + 1); compute_a(2, 2, s + 1);
    if constexpr (HAS_B) nd_b_next = p.node[(unsigned)s * plane_nodes + b_own];
    if (s + 2 <= last) load_a(s + 2);
    int r = 1, r3 = 1;                              // output plane k has relative index k - s + 1
    // one barrier interval: B(k) and, while planes are left, A(k + 2) and the loads of plane k + 3.
    // FULL: k + 3 <= last is known (the steady state); else the conditions are tested.
    auto interval = [&](auto full, int k) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full)::value;
      const bool do_a = FULL || k + 2 <= last, do_l = FULL || k + 3 <= last;
      lds_barrier();                                // planes up to k + 1 complete; reads of k - 1 done
      if constexpr (HAS_B) {
        nd_b = nd_b_next;
        read_b(r, r3, k);                           // the LDS reads are in flight behind the collide of A
      }
      if (do_a) compute_a(r + 2, r3 == 0 ? 2 : r3 - 1, k + 2);     // (r + 2) % 3
      if constexpr (HAS_B) {
        // fetched ahead and in front of the population loads: waiting for it never waits for a store
        if (FULL || k + 1 < last) nd_b_next = p.node[(unsigned)(k + 1) * plane_nodes + b_own];
      }
      if (do_a && do_l) load_a(k + 3);
      if constexpr (HAS_B) finish_b(k);
      ++r;
      r3 = r3 == 2 ? 0 : r3 + 1;
    };
    int k = s;
    if (last - s >= 4) {
      interval(std::true_type{}, k++);              // peeled: see the header of this file
      for (; k + 3 <= last; ++k) interval(std::true_type{}, k);
    }
#pragma clang loop unroll(disable)
    for (; k < last; ++k) interval(std::false_type{}, k);
  };

  // roles are uniform per wave: branch on a scalar register
  if (!b_wave) sweep(std::false_type{});
  else sweep(std::true_type{});
}

}  // namespace lt
