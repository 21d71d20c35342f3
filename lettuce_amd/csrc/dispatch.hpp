// Type-erased launch interface between api.hip (plan handling, C ABI) and the
// per-(stencil, dtype) translation units that instantiate the kernels.
#pragma once
#include <hip/hip_runtime.h>
#ifndef LT_EXPERIMENTS
#define LT_EXPERIMENTS 0        // make EXPERIMENTS=1: also the kernels that lost their A/B (unit.inc)
#endif
#include <stdint.h>

namespace lt {

enum StepMode { kFused = 0, kCollideOnly = 1, kStreamOnly = 2, kFusedTwice = 3, kFusedMany = 4, kFusedThrice = 5 };

struct StepArgs {
  const void *in;
  void *out;
  int n0, n1, n2;        // memory extents (n2 incl. ghost planes)
  long long stride_in, stride_out;   // elements between consecutive populations of in / out; 0 = dense (n0*n1*n2)
  int p_begin, planes;   // a2 planes of this launch: p_begin + i * p_stride, i < planes
  int p_stride;
  int p_begin2, planes2;  // kFusedTwice: optional second range of output planes
  int wrap2;
  double tau;
  const unsigned char *node;
  const unsigned *nsm_bits;
  const void *bt;        // BoundaryTable<T>* (device)
  int nb;
  int layout, coll, mode, masked, wide, shift, tune;
  int strip;             // kFusedTwice on 2-D lattices: columns per workgroup (512 / 256 / 128 / 64)
  int abb_axis;          // kFusedTwice with masks: memory axis of the plan's outlet (2 without one)
  int n_abb;             // anti-bounce-back outlets of the plan
  int abb0_slot;         // one-step kernels: 1-based index of the plan's only outlet if its normal is memory axis a0, else 0
  int abb_depth;         // anti-bounce-back outlets of the plan - 1 (0: at most one; the kernels exist for 0 and 1)
  int lds_bytes;         // unused dynamic LDS per workgroup (residency cap), 0 = none
  int seg_len;           // kFusedTwice: a2 planes per workgroup; kFusedMany: steps in this launch
  void *pack_lo, *pack_hi;         // fused halo packing (slab boundary launch) or null
  int pack_lo_plane, pack_hi_plane;
  unsigned long long *signal;      // kFusedTwice with both message buffers: edge workgroups first, each adds 1 here (or null)
  const void *ghost_lo, *ghost_hi; // kFusedTwice edge launch: received halo messages to read the planes beyond the cuts from (or null)
  int interior_begin, interior_end;   // first interior plane, one past the last
  hipStream_t stream;
};

struct AuxArgs {
  int what;              // 0 macroscopic, 1 equilibrium, 2 kinetic energy, 3 mass, 4 max |u|, 5 enstrophy, 6 interior mass,
                         // 7 f_neq initialisation (scale = tau, inv_dx = the identity's cs^2), 8 enstrophy of a slab's
                         // velocity field u [3][n2][n1][n0] (three neighbour planes per side), 9 interior mass of a slab
  int layout;
  const void *f;         // populations (what 0, 2, 3) / feq output (what 1; cast away const)
  void *rho;             // what 0: out, what 1: in
  void *u;               // what 0: out, what 1: in
  long long N;           // nodes per population (incl. ghost planes)
  long long stride;      // elements between consecutive populations of f (what 0, 2, 3, 4; dense = N elsewhere)
  long long first, count;  // node range reduced (what 2, 3)
  double *partial;       // plan scratch, >= reduce_blocks doubles
  int reduce_blocks;
  double *out;           // device scalar
  int n0, n1, n2;        // memory extents (what 5, 6)
  double scale, inv_dx;  // what 5: u_pu = scale * u_lu, 1 / dx_pu
  const unsigned char *mask;   // what 6, 9: no-mass mask or null
  int z_begin, nz_global;      // what 9: global index of the rank's first plane, planes of the whole grid
  long long u_stride;          // what 0: elements between the components of u (0 = N)
  hipStream_t stream;
};

typedef int (*StepFn)(const StepArgs &);
typedef int (*AuxFn)(const AuxArgs &);
typedef const char *(*NameFn)(const StepArgs &);

// one set per translation unit inst_<stencil>_<dtype>.hip
#define LT_DECLARE_UNIT(tag)              \
  int step_##tag(const StepArgs &);       \
  int aux_##tag(const AuxArgs &);         \
  const char *name_##tag(const StepArgs &);

// the unmasked 3-D two-step launches of a unit, instantiated by inst2_<tag>.hip (unit.inc, LT_PART)
#define LT_DECLARE_TWICE(tag) int twice_##tag(const StepArgs &, bool name_only, const char **name);
LT_DECLARE_TWICE(d3q15_f32)
LT_DECLARE_TWICE(d3q15_f64)
LT_DECLARE_TWICE(d3q19_f32)
LT_DECLARE_TWICE(d3q19_f64)
LT_DECLARE_TWICE(d3q27_f32)
LT_DECLARE_TWICE(d3q27_f64)

LT_DECLARE_UNIT(d1q3_f32)
LT_DECLARE_UNIT(d1q3_f64)
LT_DECLARE_UNIT(d3q15_f32)
LT_DECLARE_UNIT(d3q15_f64)
LT_DECLARE_UNIT(d2q9_f32)
LT_DECLARE_UNIT(d2q9_f64)
LT_DECLARE_UNIT(d3q19_f32)
LT_DECLARE_UNIT(d3q19_f64)
LT_DECLARE_UNIT(d3q27_f32)
LT_DECLARE_UNIT(d3q27_f64)

// returns -1 when the combination has no instantiated kernel, else the hipError_t of the launch
constexpr int kNoKernel = -1;

}  // namespace lt
