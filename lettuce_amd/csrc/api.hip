// C ABI of the engine (include/lettuce_hip.h): plan handling, argument checks, dispatch to
// the per-(stencil, dtype) kernel units.  No kernel code here.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "dispatch.hpp"
#include "kernels.hpp"
#include "lettuce_hip.h"

namespace {

// per calling thread: a caller must read lt_last_error() on the thread whose call failed (the ctypes
// binding does, right after the failing call, before anything can migrate the Python thread)
thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof g_error, fmt, ap);
  va_end(ap);
  return code;
}

#define LT_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(LT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));          \
  } while (0)

struct Unit {
  lt::StepFn step;
  lt::AuxFn aux;
  lt::NameFn name;
  int q, d;
};

const Unit kUnits[5][2] = {
    {{lt::step_d2q9_f32, lt::aux_d2q9_f32, lt::name_d2q9_f32, 9, 2},
     {lt::step_d2q9_f64, lt::aux_d2q9_f64, lt::name_d2q9_f64, 9, 2}},
    {{lt::step_d3q19_f32, lt::aux_d3q19_f32, lt::name_d3q19_f32, 19, 3},
     {lt::step_d3q19_f64, lt::aux_d3q19_f64, lt::name_d3q19_f64, 19, 3}},
    {{lt::step_d3q27_f32, lt::aux_d3q27_f32, lt::name_d3q27_f32, 27, 3},
     {lt::step_d3q27_f64, lt::aux_d3q27_f64, lt::name_d3q27_f64, 27, 3}},
    {{lt::step_d1q3_f32, lt::aux_d1q3_f32, lt::name_d1q3_f32, 3, 1},
     {lt::step_d1q3_f64, lt::aux_d1q3_f64, lt::name_d1q3_f64, 3, 1}},
    {{lt::step_d3q15_f32, lt::aux_d3q15_f32, lt::name_d3q15_f32, 15, 3},
     {lt::step_d3q15_f64, lt::aux_d3q15_f64, lt::name_d3q15_f64, 15, 3}},
};

constexpr int kReduceBlocks = 1024;

// plain streaming copy, 16 B per lane: the device-copy ceiling the roofline fraction is quoted
// against next to the 8 TB/s spec number
typedef float probe_f4 __attribute__((ext_vector_type(4)));
template <int TUNE>
__global__ void __launch_bounds__(256) probe_copy_kernel(const probe_f4 *__restrict__ src,
                                                         probe_f4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const probe_f4 v = (TUNE & 1) ? __builtin_nontemporal_load(src + i) : src[i];
    if (TUNE & 2) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

// One wave: returns when *flag >= at_least (a neighbour rank's copy engine has delivered its halo message and the
// flag write that follows it in that rank's stream has landed), or after about a second, setting *timed_out -- an exit
// every launch reaches.  Relaxed polls at SYSTEM scope (the writer is another device or a copy engine), one acquire
// at the end; no LDS, one wave: it shares a CU with a 150 KB sweep workgroup.
__global__ void wait_flag_kernel(const unsigned long long *flag, unsigned long long at_least, unsigned *timed_out) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < at_least) {
    __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > 100000000ll) {           // 100 MHz: one second
      __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
__global__ void write_flag_kernel(unsigned long long *flag, unsigned long long value) {
  __atomic_thread_fence(__ATOMIC_RELEASE);
  __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the equilibrium's exact division by its rounded constants (kernels.hpp, div_cs) on an array: the GPU-side pin of
// what tests/aux/exact_division_check.c proves on the host -- in particular that the device keeps fp32 denormals
// (x r_lo of the two-instruction form is denormal for |x / D| below ~1e-30)
template <typename T>
__global__ void __launch_bounds__(256) probe_div_cs_kernel(const T *x, T *out, long long n, int which) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = which == 0 ? lt::div_cs<0>(x[i]) : lt::div_cs<1>(x[i]);
}

// first-use check of the masked two-step kernels (run_canary): synthetic populations -- positive, near 1 / Q, a
// different value in nearly every slot -- and a bit-for-bit comparison of two population fields over a plane range
template <typename T>
__global__ void __launch_bounds__(256) canary_fill_kernel(T *f, long long stride, long long N, int Q) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N * Q; i += (long long)gridDim.x * 256) {
    const long long q = i / N, x = i - q * N;
    const unsigned h = (unsigned)(((unsigned long long)i * 2654435761ull) >> 9) & 4095u;
    f[q * stride + x] = (T)((1.0 / Q) * (1.0 + 0.2 * ((double)h / 4096.0 - 0.5)));
  }
}
template <typename U>
__global__ void __launch_bounds__(256) canary_compare_kernel(const U *a, const U *b, long long stride, long long first,
                                                             long long count, int Q, unsigned long long *mismatches) {
  unsigned long long mine = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count * Q; i += (long long)gridDim.x * 256) {
    const long long q = i / count, x = first + (i - q * count);
    mine += a[q * stride + x] != b[q * stride + x];
  }
  if (mine) atomicAdd(mismatches, mine);
}

}  // namespace

struct lt_plan {
  lt_plan_desc desc;
  Unit unit;
  int esize;                 // sizeof scalar
  int n0, n1, n2;            // memory extents incl. ghost planes
  int interior_begin, interior_end;   // a2 planes that are real nodes
  long long N;               // n0*n1*n2
  int wide_ok;               // n0 divisible by the 16-byte vector width
  int shift;
  int tune = -1;             // cache policy: -1 = automatic
  int residency = -1;        // workgroups per CU of the big launches: -1 = automatic, 0 = no cap
  int n_cu = 0;              // compute units of the plan's device
  int two_step = -1;         // lt_run: pair the fused steps (lbm2_kernel): -1 = automatic, 0 / 1
  int seg_len = 0;           // planes per workgroup of the two-step kernel, 0 = automatic
  int many = -1;             // lt_run: several steps per launch on small 2-D grids: -1 = automatic, 0 / 1
  int many_now = 1;          // steps of the kFusedMany launch being issued
  long long second_begin = 0, second_end = 0;   // second plane range of the kFusedTwice launch being issued
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // lt_plan_set_fused_events
  long long last_single = 0, last_twice = 0, last_many = 0;   // fused launches of the last lt_run
  int want_wide = 0;         // 16-byte accesses for the hot kernel (A/B experiments)
  // engine-owned device scratch
  unsigned char *node = nullptr;
  unsigned *nsm_bits = nullptr;
  void *bt = nullptr;        // BoundaryTable<T>
  double *partial = nullptr;
  int masked = 0;
  int n_abb = 0;             // anti-bounce-back outlets of the plan
  int abb_depth = 0;         // how deep an outlet's neighbour must be rebuilt (lt_plan_create)
  int nsm_confined = 1;      // the no-streaming bits are exactly those of the plan's outlet (lt_plan_set_masks)
  int inlet_faces_outlet = 1;  // outlet along a0: every node of the opposite face is an equilibrium node
  unsigned *mask_flag = nullptr;   // device word written by the mask compilation
  // lt_stream_collide_twice_slab: counter the edge workgroups increment, the value it reaches when the messages
  // of the last such launch are complete, and the word wait_counter_kernel sets when it gives up
  unsigned long long *signal = nullptr;
  unsigned long long signal_target = 0;
  unsigned *signal_timed_out = nullptr;
  unsigned long long *signal_now = nullptr;   // where step() finds the counter for the launch being issued (or null)
  const void *ghost_lo_now = nullptr, *ghost_hi_now = nullptr;   // received halo messages the edge launch being issued reads
  int arith = 0;             // 0 = the reference's arithmetic, operation for operation; 1 = fast (lt_plan_set_arithmetic)
  int defer_stream = 0;      // lt_run / lt_continue stop before their last (streaming) pass (lt_plan_set_deferred_stream)
  // distance between consecutive populations of the caller's buffers, in elements (lt_plan_set_population_stride);
  // 0 = dense (N).  stride_in_now / stride_out_now: what step() uses for the launch being issued when the two
  // differ from it (the resident entry points move between the caller's dense buffers and the padded ones)
  long long pop_stride = 0;
  long long stride_in_now = -1, stride_out_now = -1;
  // engine-owned padded population buffers (lt_resident_*): res[res_cur] holds the post-collision populations
  int resident = -1;         // -1 = automatic, 0 = off, 1 = on (lt_plan_set_resident)
  long long res_pad = -1;    // elements between the end of one population and the start of the next; -1 = automatic
  long long res_stride = 0;  // stride of the allocated buffers
  void *res[2] = {nullptr, nullptr};
  int res_cur = 0;
  int res_valid = 0;
  // first-use check of the masked two-step kernel (run_canary; lt_plan_set_canary / lt_plan_canary_status)
  int canary_mode = 1;       // 1 = check on first use, 0 = skip, 2 = test hook: report a mismatch without launching
  int canary = 0;            // 0 = not run for the present masks / settings, 1 = passed, 2 = skipped, -1 = failed
  int canary_running = 0;
  long long canary_mismatches = 0;
  char canary_msg[320] = "";
  hipStream_t cstream = nullptr;
  char kernel_name[192];
  // launch-bound grids: a captured hipGraph of kGraphChunk fused steps (ping-pong returns to the
  // starting buffer), replayed on a plan-owned stream that is forked from / joined to the caller's
  int graph_mode = -1;       // -1 automatic (small grids), 0 off, 1 always
  hipStream_t gstream = nullptr;
  hipEvent_t gev_in = nullptr, gev_out = nullptr;
  hipGraphExec_t gexec = nullptr;
  struct { void *a, *b; double tau; int masked, tune, wide, shift, residency; } gkey = {};
};

namespace {

// elements between consecutive populations of the caller's buffers
long long pop_stride_of(const lt_plan *p) { return p->pop_stride > 0 ? p->pop_stride : p->N; }

// the collision the kernels are asked for: the plan's, or 3 = BGK in fast arithmetic (lt_plan_set_arithmetic)
int coll_of(const lt_plan *p) {
  return (p->arith == 1 && p->desc.collision == LT_COLLISION_BGK) ? 3 : p->desc.collision;
}

int mem_axis_of(const lt_plan *p, int logical_axis) {
  if (p->unit.d == 1) return 0;
  if (p->unit.d == 2) return logical_axis == 0 ? 1 : 0;
  return p->desc.layout == LT_LAYOUT_REFERENCE ? 2 - logical_axis : logical_axis;
}

template <typename T>
int upload_boundaries(lt_plan *p, hipStream_t stream) {
  lt::BoundaryTable<T> h;
  memset(&h, 0, sizeof h);
  const int ext[3] = {p->n0, p->n1, p->n2};
  for (int i = 0; i < p->desc.n_boundaries; ++i) {
    const lt_boundary_desc &b = p->desc.boundaries[i];
    const int s = i + 1;
    h.kind[s] = b.kind;
    if (b.kind == LT_BOUNDARY_ABB_OUTLET) {
      const int ax = mem_axis_of(p, b.axis);
      h.mem_axis[s] = ax;
      h.side[s] = b.side;
      // along the decomposed axis of a slab the first / last plane of the grid is the first / last
      // interior plane of the rank that holds it; the other ranks have no outlet (a plane no node has)
      const int g = ax == 2 ? p->desc.ghost_planes : 0;
      h.plane[s] = b.side > 0 ? ext[ax] - 1 - g : g;
      h.nbr[s] = h.plane[s] - b.side;
      if (g && (b.flags & LT_BOUNDARY_ABSENT)) h.plane[s] = h.nbr[s] = -1000000;
    } else if (b.kind == LT_BOUNDARY_EQUILIBRIUM) {
      for (int q = 0; q < p->unit.q; ++q) h.feq[s][q] = (T)b.feq[q];
      h.field[s] = static_cast<const T *>(b.feq_field_dev);
    }
  }
  // staged through the (pageable) host struct: synchronous w.r.t. the host, ordered on stream
  LT_HIP(hipMemcpyAsync(p->bt, &h, sizeof h, hipMemcpyHostToDevice, stream));
  LT_HIP(hipStreamSynchronize(stream));
  return LT_OK;
}

int check_boundary(const lt_plan *p, const lt_boundary_desc &b, int n_abb_before) {
  switch (b.kind) {
    case LT_BOUNDARY_BOUNCE_BACK:
    case LT_BOUNDARY_EQUILIBRIUM:
      return LT_OK;
    case LT_BOUNDARY_ABB_OUTLET: {
      if (b.axis < 0 || b.axis >= p->unit.d || (b.side != 1 && b.side != -1))
        return fail(LT_ERR_INVALID, "anti-bounce-back outlet: axis %d side %d", b.axis, b.side);
      if (p->desc.shape[b.axis] < 2)
        return fail(LT_ERR_INVALID, "anti-bounce-back outlet needs >= 2 planes along its axis");
      if (p->desc.ghost_planes && b.axis == 2 && !(b.flags & LT_BOUNDARY_ABSENT) && p->desc.shape[2] < 2)
        return fail(LT_ERR_INVALID, "an outlet along the decomposed (z) axis needs >= 2 planes on its rank");
      return LT_OK;
    }
    default:
      return fail(LT_ERR_INVALID, "unknown boundary kind %d", b.kind);
  }
}

// Nontemporal accesses pay off when the two population buffers cannot stay in the caches
// (32 MiB L2 + 256 MiB Infinity Cache); small grids keep cached stores.
// populations whose velocity component along the slowest memory axis equals `dir`
template <class S>
lt::QList crossing_of(int logical_axis, int dir) {
  lt::QList l;
  l.n = 0;
  for (int q = 0; q < S::Q; ++q)
    if (S::E[q][logical_axis] == dir && l.n < 9) l.q[l.n++] = q;
  return l;
}

lt::QList crossing(const lt_plan *p, int dir) {
  const int axis = p->unit.d == 2 ? 0 : (p->desc.layout == LT_LAYOUT_REFERENCE ? 0 : 2);
  switch (p->desc.stencil) {
    case LT_D1Q3: return crossing_of<lt::D1Q3>(0, dir);
    case LT_D2Q9: return crossing_of<lt::D2Q9>(axis, dir);
    case LT_D3Q15: return crossing_of<lt::D3Q15>(axis, dir);
    case LT_D3Q19: return crossing_of<lt::D3Q19>(axis, dir);
    default: return crossing_of<lt::D3Q27>(axis, dir);
  }
}

// populations whose velocity component along MEMORY axis `mem_axis` equals `dir` (the axis maps are
// involutions: mem_axis_of also takes a memory axis to its logical one)
lt::QList crossing_axis(const lt_plan *p, int mem_axis, int dir) {
  const int axis = mem_axis_of(p, mem_axis);
  switch (p->desc.stencil) {
    case LT_D1Q3: return crossing_of<lt::D1Q3>(0, dir);
    case LT_D2Q9: return crossing_of<lt::D2Q9>(axis, dir);
    case LT_D3Q15: return crossing_of<lt::D3Q15>(axis, dir);
    case LT_D3Q19: return crossing_of<lt::D3Q19>(axis, dir);
    default: return crossing_of<lt::D3Q27>(axis, dir);
  }
}

int pack(lt_plan *p, bool do_pack, void *f, long long plane, int dir, void *buf, void *stream) {
  if (!p || !f || !buf) return fail(LT_ERR_INVALID, "null argument");
  if (dir != 1 && dir != -1) return fail(LT_ERR_INVALID, "direction %d (must be +1 or -1)", dir);
  if (plane < 0 || plane >= p->n2) return fail(LT_ERR_INVALID, "plane %lld outside [0, %d)", plane, p->n2);
  const lt::QList ql = crossing(p, dir);
  const int plane_nodes = p->n0 * p->n1;
  const long long off = plane * plane_nodes;
  const unsigned grid = (plane_nodes + lt::kThreads - 1) / lt::kThreads;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->desc.dtype == LT_F32) {
    if (do_pack) hipLaunchKernelGGL((lt::plane_pack_kernel<float, true>), dim3(grid), dim3(lt::kThreads), 0, s,
                                    (float *)f, (float *)buf, pop_stride_of(p), off, plane_nodes, ql);
    else hipLaunchKernelGGL((lt::plane_pack_kernel<float, false>), dim3(grid), dim3(lt::kThreads), 0, s,
                            (float *)f, (float *)buf, pop_stride_of(p), off, plane_nodes, ql);
  } else {
    if (do_pack) hipLaunchKernelGGL((lt::plane_pack_kernel<double, true>), dim3(grid), dim3(lt::kThreads), 0, s,
                                    (double *)f, (double *)buf, pop_stride_of(p), off, plane_nodes, ql);
    else hipLaunchKernelGGL((lt::plane_pack_kernel<double, false>), dim3(grid), dim3(lt::kThreads), 0, s,
                            (double *)f, (double *)buf, pop_stride_of(p), off, plane_nodes, ql);
  }
  LT_HIP(hipGetLastError());
  return LT_OK;
}

int resolve_tune(const lt_plan *p, int wide) {
  if (wide) return p->tune < 0 ? 0 : p->tune;
  if (p->tune == 0 || p->tune == 3) return p->tune;
  const long long bytes = 2ll * p->unit.q * p->N * p->esize;
  return bytes > (128ll << 20) ? 3 : 0;
}

// Fewer resident waves stream better: with 19-27 read and as many write streams per workgroup,
// 3 workgroups per CU (3 waves per SIMD) instead of the 8 the registers allow ran 1-4 % faster on
// MI355X (tools/occupancy_probe.py: D3Q19 fp32 -1.3..-2.7 %, D3Q27 fp32 -1..-4 %, D3Q19 fp64
// -2..-4 %; 2 per CU is 15 % slower in fp32).  fp32 KBC, the one kernel with real arithmetic, gets 4.  The cap is an unused dynamic-LDS
// allocation (160 KB per CU).  Only for launches that fill the chip several times over and stream
// from HBM; small launches (boundary planes beside the interior launch, small grids) stay uncapped.
int resolve_lds(const lt_plan *p, long long workgroups) {
  int n = p->residency;
  if (n < 0) {
    const long long bytes = 2ll * p->unit.q * p->N * p->esize;
    const bool kbc32 = p->desc.collision == LT_COLLISION_KBC && p->esize == 4;
    n = (bytes > (128ll << 20) && workgroups >= 8192) ? (kbc32 ? 4 : 3) : 0;
  }
  if (n <= 0 || n >= 8) return 0;
  const int per_wg = (160 * 1024 / n) & ~2047;   // the LDS allocator rounds up: stay below 160 KB / n
  return per_wg > 65536 ? 65536 : per_wg;
}

// ---- engine-owned padded population buffers (lt_resident_*) ----
// Distance added between consecutive populations of the engine's own buffers.  The q read streams and q write
// streams of a node advance in lockstep; with the populations a power of two apart (256^3 fp32: exactly 64 MiB)
// they meet in the same memory channels.  Measured on MI355X (tools/pad_sweep_probe.py).
// Of the pads tried on five 3-D grids and three lattices (0, 128, 320, 2112, 2368, 32832 elements, three runs
// each) 32768 + 64 elements was the best or within a per cent of the best everywhere: two-step launch -7.0 % at
// 256^3 fp32 (populations exactly 64 MiB apart when dense), -2.3 % at 512 x 512 x 64, -5.1 % at 256 x 256 x 512,
// -4.9 % at 384^3, -5.2 % at 256^3 fp64, -3.9 % for D3Q15 (profiles/r03_pad_sweep_robust.jsonl).
long long default_pad(const lt_plan *) { return 32768 + 64; }
long long resident_stride(const lt_plan *p) {
  const long long pad = p->res_pad >= 0 ? p->res_pad : default_pad(p);
  const long long unit = 256 / p->esize;                      // keep every population 256-byte aligned
  return (p->N + pad + unit - 1) / unit * unit;
}
bool two_step_wanted(lt_plan *p);
bool resident_wanted(const lt_plan *p) {
  if (p->desc.ghost_planes || p->unit.d < 2) return false;
  if (p->resident >= 0) return p->resident != 0;
  // automatic: 3-D plans whose fused steps run as two-step launches (which implies the streaming regime).  The
  // one-step kernels stream best from DENSE populations (padding costs them 2-4 %: 256^3 D3Q19 fp32 0.411 ->
  // 0.42-0.44 ms) and D2Q9's two-step kernel gains nothing (4096^2: -2 % .. +4 %), so those keep the caller's buffers.
  return p->unit.d == 3 && two_step_wanted(const_cast<lt_plan *>(p));
}
int resident_alloc(lt_plan *p) {
  const long long stride = resident_stride(p);
  if (p->res[0] && p->res_stride == stride) return LT_OK;
  for (void *&r : p->res) { if (r) (void)hipFree(r); r = nullptr; }
  p->res_valid = 0;
  const size_t bytes = (size_t)stride * p->unit.q * p->esize;
  for (void *&r : p->res)
    if (hipMalloc(&r, bytes) != hipSuccess) {
      for (void *&x : p->res) { if (x) (void)hipFree(x); x = nullptr; }
      return fail(LT_ERR_ALLOC, "hipMalloc of the resident population buffers (2 x %zu bytes) failed", bytes);
    }
  p->res_stride = stride;
  return LT_OK;
}

// the two-step tile (the rule of unit.inc): rows of 256 bytes, 8 or 4 of them; rows = 0: no kernel
struct TwoStepTile { int width, rows; };
TwoStepTile two_step_tile_of(int d, int q, int esize, int n0, bool masked = false) {
  if (d == 2) {
    // 2-D (twostep2d.hpp): strips of `width` columns, the widest that divides the contiguous extent; one "row"
    for (int w = 512; w >= 64; w /= 2)
      if (n0 % w == 0) return {w, 1};
    return {0, 0};
  }
  const long long per_node = (long long)esize * 3 * q;
  const int width = 256 / esize;
  int rows = per_node * (width + 2) * 10 <= 160 * 1024
                 ? 8 : (esize == 4 && per_node * (width + 2) * 6 <= 160 * 1024 ? 4 : 0);
  if (masked && rows == 8) {
    // with boundaries the downward populations have a third LDS slot (lbm2m_kernel): where 8 rows then exceed
    // the 160 KB, 4 rows are used (D3Q19 fp64: 168.6 KB -> 32 x 4 tiles, 101 KB); unit.inc applies the same rule
    const int in_plane = q == 15 ? 5 : 9, crossing = (q - in_plane) / 2;
    const long long lds8 = (long long)esize * ((width + 2) * 10 * (4 * crossing + 3 * in_plane + 3 * crossing) + 2 * q);
    if (lds8 > 160 * 1024) rows = LT_EXPERIMENTS ? 4 : 0;   // (product build: no masked two-step kernel there; it lost its A/B)
  }
  return {width, rows};
}
TwoStepTile two_step_tile(const lt_plan *p) {
  return two_step_tile_of(p->unit.d, p->unit.q, p->esize, p->n0, p->masked != 0);
}

// The 3-D two-step kernels address with 32-bit byte offsets: lbm2_kernel within a plane, lbm2m_kernel (buffer
// instructions) within the whole field -- the rule of unit.inc's launch_twice / launch_twice_masked, which return
// kNoKernel beyond it.  Asked BEFORE a plan is put on the two-step path, so that lt_run falls back to the one-step
// kernel instead of failing at the first launch (Obstacle D3Q19 fp32 at 384^3 is 4.0 GiB).  With BGK in the reference
// layout lbm2m_kernel has, for D3Q15 / D3Q19, a second instantiation with one buffer descriptor per population (BIG):
// there a POPULATION must stay below 4 GiB, not the field.
bool two_step_addressable(int d, int q, int esize, long long n0, long long n1, long long n2, bool masked,
                          bool per_population = false) {
  if (d != 3) return true;
  if (masked) return (per_population ? 1ll : (long long)q) * n0 * n1 * n2 * esize < (1ll << 32);
  return n0 * n1 * esize < (1ll << 32);
}
bool masked_big_exists(int layout, int collision, int q) {
  return layout == LT_LAYOUT_REFERENCE && collision == LT_COLLISION_BGK && q <= 19;      // unit.inc, kHasBig
}

// planes per workgroup of the two-step kernel.  One workgroup occupies a CU (150 KB of LDS), so the
// launch runs in rounds of n_cu workgroups; a segment of L planes computes L + 2 intermediate
// planes.  Pick the divisor L of n2 with the best (L / (L + 2)) * (blocks / blocks rounded up to
// whole rounds): 128 planes = 256 workgroups at 256^3 (measured 0.3345 ms per step against 0.3412
// with 32 planes and 0.52 with 256, which leaves half the CUs idle).
int resolve_seg_len(const lt_plan *p, int planes) {
  if (p->seg_len > (p->masked ? 1 : 0)) return p->seg_len;
  const TwoStepTile tile = two_step_tile(p);
  long long tiles = (long long)(p->n0 / tile.width) * (p->n1 / (tile.rows ? tile.rows : 8));
  long long cus = p->n_cu > 0 ? p->n_cu : 256;
  if (p->unit.d == 2) {
    // the sweep runs along a1.  Two workgroups would fit a CU in fp32 (27 row slots of width + 2 values =
    // 55 KB), but one per CU streams better: 4096^2 fp32 0.118 ms per update with 256 workgroups of 128 rows
    // against 0.159 with 512 of 64 (tools/two_step_2d_probe.py)
    tiles = p->n0 / tile.width;
  }
  int best = 1;
  double best_score = -1.0;
  for (int len = p->masked ? 2 : 1; len <= planes; ++len) {   // masked: the outlet's plane never opens a segment
    const long long segs = (planes + len - 1) / len;
    if (!p->desc.ghost_planes && planes % len) continue;   // keep whole-grid launches evenly cut
    const long long blocks = tiles * segs;
    const long long rounds = (blocks + cus - 1) / cus;
    const double score = ((double)planes / (double)(planes + 2 * segs)) *
                         ((double)blocks / (double)(rounds * cus));
    if (score > best_score) { best_score = score; best = len; }
  }
  return best;
}

// planes per workgroup of the three-step kernel (lbm3_kernel: tiles of 64 / 32 x 4 nodes, one workgroup per CU, a
// segment of L planes computes L + 4 level-1 and L + 2 level-2 planes)
int resolve_seg_len3(const lt_plan *p, int planes) {
  const int width = 256 / p->esize;
  const long long tiles = (long long)(p->n0 / width) * (p->n1 / 4);
  const long long cus = p->n_cu > 0 ? p->n_cu : 256;
  int best = 1;
  double best_score = -1.0;
  for (int len = 1; len <= planes; ++len) {
    if (planes % len) continue;
    const long long blocks = tiles * (planes / len);
    const long long rounds = (blocks + cus - 1) / cus;
    const double score = ((double)len / (double)(len + 3)) * ((double)blocks / (double)(rounds * cus));
    if (score > best_score) { best_score = score; best = len; }
  }
  return best;
}

// Two-step slab halo: side -1 = lower cut, +1 = upper cut.  Packing reads the two interior planes
// next to the cut, unpacking fills the two ghost planes beyond it.
int halo2(lt_plan *p, bool do_pack, void *f, int side, void *buf, void *stream) {
  if (!p || !f || !buf) return fail(LT_ERR_INVALID, "null argument");
  if (side != 1 && side != -1) return fail(LT_ERR_INVALID, "side %d (must be +1 or -1)", side);
  if (p->desc.ghost_planes != 2) return fail(LT_ERR_INVALID, "the two-step halo needs ghost_planes = 2");
  const int g = 2, n2 = p->n2;
  long long near, far;
  int dir;
  if (do_pack) {     // what the neighbour beyond `side` needs: populations staying in / leaving through the cut
    near = side < 0 ? g : n2 - g - 1;
    far = side < 0 ? g + 1 : n2 - g - 2;
    dir = side;
  } else {           // what arrived from the neighbour beyond `side`: populations entering through the cut
    near = side < 0 ? g - 1 : n2 - g;
    far = side < 0 ? g - 2 : n2 - g + 1;
    dir = -side;
  }
  const lt::QList in_plane = crossing(p, 0), cross = crossing(p, dir);
  lt::QList away = crossing(p, -dir);
  if (!p->masked) away.n = 0;                      // only a no-streaming node ever reads them
  const int plane_nodes = p->n0 * p->n1;
  const unsigned grid = (plane_nodes + lt::kThreads - 1) / lt::kThreads;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->desc.dtype == LT_F32) {
    if (do_pack) hipLaunchKernelGGL((lt::halo2_kernel<float, true>), dim3(grid), dim3(lt::kThreads), 0, s, (float *)f,
                                    (float *)buf, pop_stride_of(p), near * plane_nodes, far * plane_nodes, plane_nodes, in_plane, cross, away);
    else hipLaunchKernelGGL((lt::halo2_kernel<float, false>), dim3(grid), dim3(lt::kThreads), 0, s, (float *)f,
                            (float *)buf, pop_stride_of(p), near * plane_nodes, far * plane_nodes, plane_nodes, in_plane, cross, away);
  } else {
    if (do_pack) hipLaunchKernelGGL((lt::halo2_kernel<double, true>), dim3(grid), dim3(lt::kThreads), 0, s, (double *)f,
                                    (double *)buf, pop_stride_of(p), near * plane_nodes, far * plane_nodes, plane_nodes, in_plane, cross, away);
    else hipLaunchKernelGGL((lt::halo2_kernel<double, false>), dim3(grid), dim3(lt::kThreads), 0, s, (double *)f,
                            (double *)buf, pop_stride_of(p), near * plane_nodes, far * plane_nodes, plane_nodes, in_plane, cross, away);
  }
  LT_HIP(hipGetLastError());
  return LT_OK;
}

// Two updates per launch on a plan with boundaries (lbm2m_kernel, twostep_masked.hpp): bounce-back and
// equilibrium nodes anywhere; at most one anti-bounce-back outlet, either at the LAST plane of the sweep axis
// (memory axis a2, side +1 -- the reference's Obstacle in the reference layout), where its neighbour's moments
// and its no-streaming bits reduce to tests on the plane index, or at an end of the rows (memory axis a0 --
// the Obstacle in the slab layout), where the neighbour is the next lane and the face opposite the outlet
// must consist of equilibrium nodes; and no-streaming bits exactly where that outlet puts them (both checked
// on the device when the masks are compiled).  Anything else keeps the one-step kernel.
// Returns the memory axis of the outlet (2 without one), or -1: not admitted.
int masked_two_step_axis(const lt_plan *p) {
  if (!p->nsm_confined) return -1;
  if (p->desc.n_boundaries > 15) return -1;          // two bits of kind per slot in one 32-bit register (MaskedPlanInfo)
  const int sweep = p->unit.d == 2 ? 1 : 2;          // the slowest memory axis
  if (p->unit.d < 2) return -1;
  int n_abb = 0, axis = 2;
  for (int i = 0; i < p->desc.n_boundaries; ++i) {
    const lt_boundary_desc &b = p->desc.boundaries[i];
    if (b.kind != LT_BOUNDARY_ABB_OUTLET || (b.flags & LT_BOUNDARY_ABSENT)) continue;
    if (++n_abb > 1) return -1;
    const int ax = mem_axis_of(p, b.axis);
    if (ax == sweep) {
      if (b.side != 1 || (sweep == 2 ? p->n2 : p->n1) < 3) return -1;
      axis = 2;                                      // "at the last plane / row of the sweep"
    } else if (ax == 0 && p->unit.d == 3) {
      if (!p->inlet_faces_outlet) return -1;
      axis = 0;
    } else {
      return -1;
    }
  }
  return axis;
}
bool masked_two_step_ok(const lt_plan *p) { return masked_two_step_axis(p) >= 0; }
bool canary_ok(lt_plan *p);

int step(lt_plan *p, int mode, const void *in, void *out, double tau, long long pb, long long pe,
         void *stream, long long stride = 1, void *pack_lo = nullptr, void *pack_hi = nullptr) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (!in || !out) return fail(LT_ERR_INVALID, "null population buffer");
  if (in == out) return fail(LT_ERR_INVALID, "in-place operation is not supported (in == out)");
  if (pb < 0 || pe > p->n2 || pb > pe)
    return fail(LT_ERR_INVALID, "plane range [%lld, %lld) outside [0, %d)", pb, pe, p->n2);
  if (p->desc.ghost_planes && (pb < 1 || pe > p->n2 - 1) && mode != lt::kCollideOnly && pe > pb)
    return fail(LT_ERR_INVALID, "streaming from ghost planes: range [%lld, %lld) must stay in [1, %d)",
                pb, pe, p->n2 - 1);
  if (mode == lt::kFusedMany && (p->desc.ghost_planes || (p->masked && p->n_abb > 1)))
    return fail(LT_ERR_UNSUPPORTED, "several steps per launch: no slabs, at most one anti-bounce-back outlet");
  if (mode == lt::kFusedTwice) {
    if (p->desc.ghost_planes == 1)
      return fail(LT_ERR_INVALID, "two steps per launch read two planes beyond the range: the plan needs "
                                  "ghost_planes = 2");
    if (p->desc.ghost_planes == 2 && (pb < 2 || pe > p->n2 - 2) && pe > pb)
      return fail(LT_ERR_INVALID, "two-step range [%lld, %lld) must stay in [2, %d)", pb, pe, p->n2 - 2);
    if (!p->desc.ghost_planes && (pb != 0 || pe != p->n2))
      return fail(LT_ERR_INVALID, "periodic plan: the two-step launch covers all planes");
    if (p->masked && (!masked_two_step_ok(p) || (p->desc.ghost_planes && p->n_abb > 0 && masked_two_step_axis(p) != 0)))
      return fail(LT_ERR_UNSUPPORTED, "two steps per launch with boundaries: at most one anti-bounce-back outlet, at the "
                                      "last plane of the slowest memory axis (periodic plans only) or at an end of "
                                      "the contiguous axis opposite a face of equilibrium nodes; no-streaming bits "
                                      "exactly that outlet's");
    if (p->masked && !p->canary_running && !canary_ok(p)) return fail(LT_ERR_UNSUPPORTED, "%s", p->canary_msg);
  }
  if (p->desc.n_boundaries > 0 && !p->masked)
    return fail(LT_ERR_INVALID, "plan has boundaries but lt_plan_set_masks was not called");
  if (mode != lt::kStreamOnly && p->desc.collision != LT_COLLISION_NONE && !(tau > 0.0))
    return fail(LT_ERR_INVALID, "relaxation time tau = %g", tau);
  lt::StepArgs a;
  memset(&a, 0, sizeof a);
  a.in = in; a.out = out;
  a.n0 = p->n0; a.n1 = p->n1; a.n2 = p->n2;
  a.p_begin = (int)pb; a.planes = (int)((pe - pb + stride - 1) / stride); a.p_stride = (int)stride;
  a.wrap2 = p->desc.ghost_planes ? 0 : 1;
  a.tau = tau > 0.0 ? tau : 1.0;
  a.node = p->node; a.nsm_bits = p->nsm_bits; a.bt = p->bt; a.nb = p->desc.n_boundaries;
  a.layout = p->desc.layout; a.coll = coll_of(p); a.mode = mode;
  a.masked = p->masked;
  a.abb_depth = p->abb_depth;
  a.abb_axis = p->masked ? masked_two_step_axis(p) : 2;
  a.n_abb = p->masked ? p->n_abb : 0;
  a.abb0_slot = 0;
  if (p->masked && p->n_abb == 1)
    for (int i = 0; i < p->desc.n_boundaries; ++i) {
      const lt_boundary_desc &b = p->desc.boundaries[i];
      if (b.kind == LT_BOUNDARY_ABB_OUTLET && !(b.flags & LT_BOUNDARY_ABSENT) && mem_axis_of(p, b.axis) == 0)
        a.abb0_slot = i + 1;
    }
  const bool aligned = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                       (p->pop_stride * p->esize) % 16 == 0 && p->stride_in_now < 0 && p->stride_out_now < 0;
  const bool hot = mode == lt::kFused && !a.masked && a.coll == LT_COLLISION_BGK;
  a.wide = (p->want_wide && hot && p->wide_ok && aligned) ? 1 : 0;
  a.shift = a.wide ? p->shift : 0;
  a.tune = resolve_tune(p, a.wide);
  a.lds_bytes = resolve_lds(p, ((long long)a.planes * a.n1 * a.n0 + 255) / 256);
  a.seg_len = mode == lt::kFusedTwice ? resolve_seg_len(p, p->unit.d == 2 ? p->n1 : a.planes) : 0;
  a.strip = p->unit.d == 2 ? two_step_tile(p).width : 0;
  if (mode == lt::kFusedTwice) a.shift = p->shift;      // tile-shape A/B variant
  if (mode == lt::kFusedThrice) a.seg_len = p->seg_len > 0 && a.planes % p->seg_len == 0 ? p->seg_len : resolve_seg_len3(p, a.planes);
  if (mode == lt::kFusedMany) a.seg_len = p->many_now;
  a.stream = static_cast<hipStream_t>(stream);
  a.stride_in = p->stride_in_now >= 0 ? p->stride_in_now : p->pop_stride;
  a.stride_out = p->stride_out_now >= 0 ? p->stride_out_now : p->pop_stride;
  a.pack_lo = pack_lo; a.pack_hi = pack_hi;
  a.signal = p->signal_now;
  a.ghost_lo = p->ghost_lo_now; a.ghost_hi = p->ghost_hi_now;
  a.interior_begin = p->interior_begin; a.interior_end = p->interior_end;
  a.pack_lo_plane = (int)pb; a.pack_hi_plane = (int)(pb + stride * (a.planes - 1));
  if (mode == lt::kFusedTwice) {
    a.p_begin2 = (int)p->second_begin;
    a.planes2 = (int)(p->second_end - p->second_begin);
    if (a.planes2 > 0) a.pack_hi_plane = (int)p->second_end - 1;   // upper message: last plane of the second range
  }
  const int r = p->unit.step(a);
  if (r == lt::kNoKernel)
    return fail(LT_ERR_UNSUPPORTED, "no kernel for layout %d collision %d mode %d masked %d",
                a.layout, a.coll, a.mode, a.masked);
  if (r != 0) return fail(LT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)r));
  return LT_OK;
}

constexpr int kGraphChunk = 32;            // fused steps per graph (even: ends in the start buffer)

// Measured on MI355X (profiles/r01_small_grid_graph.jsonl): the eager C loop of lt_run already
// runs at 3.5 us per step on 128^2 (dependent-kernel boundary ~1.5 us + kernel), and graph replay
// was slower on three of four small grids (4.1 vs 3.5 us at 128^2; 4.6 vs 5.1 us at 256^2), so
// "automatic" currently means eager; the graph path stays available with mode 1.
bool graph_wanted(const lt_plan *p, long long fused) {
  if (fused < 2 * kGraphChunk) return false;
  return p->graph_mode == 1;
}

// Replays floor(fused / kGraphChunk) chunks of fused steps starting (and ending) in `cur`; returns
// the number of steps done through the graph, or a negative status on failure.
long long run_graph(lt_plan *p, void *cur, void *other, double tau, long long fused, void *stream) {
  hipStream_t user = static_cast<hipStream_t>(stream);
  if (!p->gstream) {
    if (hipStreamCreateWithFlags(&p->gstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&p->gev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->gev_out, hipEventDisableTiming) != hipSuccess)
      return -fail(LT_ERR_HIP, "cannot create the graph stream/events");
  }
  const bool same = p->gexec && p->gkey.a == cur && p->gkey.b == other && p->gkey.tau == tau &&
                    p->gkey.masked == p->masked && p->gkey.tune == p->tune && p->gkey.residency == p->residency &&
                    p->gkey.wide == p->want_wide && p->gkey.shift == p->shift;
  if (!same) {
    if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(p->gstream, hipStreamCaptureModeThreadLocal) != hipSuccess)
      return -fail(LT_ERR_HIP, "hipStreamBeginCapture failed");
    void *c = cur, *o = other;
    int rc = LT_OK;
    for (int i = 0; i < kGraphChunk && rc == LT_OK; ++i) {
      rc = step(p, lt::kFused, c, o, tau, 0, p->n2, p->gstream);
      void *t = c; c = o; o = t;
    }
    const hipError_t e = hipStreamEndCapture(p->gstream, &graph);
    if (rc != LT_OK) { if (graph) (void)hipGraphDestroy(graph); return -rc; }
    if (e != hipSuccess || !graph) return -fail(LT_ERR_HIP, "hipStreamEndCapture failed");
    const hipError_t ei = hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { p->gexec = nullptr; return -fail(LT_ERR_HIP, "hipGraphInstantiate failed"); }
    p->gkey = {cur, other, tau, p->masked, p->tune, p->want_wide, p->shift, p->residency};
  }
  const long long reps = fused / kGraphChunk;
  if (hipEventRecord(p->gev_in, user) != hipSuccess ||
      hipStreamWaitEvent(p->gstream, p->gev_in, 0) != hipSuccess)
    return -fail(LT_ERR_HIP, "fork to the graph stream failed");
  for (long long r = 0; r < reps; ++r)
    if (hipGraphLaunch(p->gexec, p->gstream) != hipSuccess)
      return -fail(LT_ERR_HIP, "hipGraphLaunch failed");
  if (hipEventRecord(p->gev_out, p->gstream) != hipSuccess ||
      hipStreamWaitEvent(user, p->gev_out, 0) != hipSuccess)
    return -fail(LT_ERR_HIP, "join from the graph stream failed");
  return reps * kGraphChunk;
}

// Is there a two-step launch for this plan?  Needs the kernel for this lattice / dtype / collision (asked of
// the unit by name), a grid that tiles (a0 % 64, a1 % 8) and, with masks, their admission
// (masked_two_step_axis).  why != nullptr: the reason it is not.
bool two_step_possible(lt_plan *p, const char **why) {
  const char *dummy;
  if (!why) why = &dummy;
  const TwoStepTile tile = two_step_tile(p);
  if (tile.rows == 0 || p->n0 % tile.width != 0 || p->n1 % tile.rows != 0) {
    *why = "the grid does not tile (contiguous extent % 64 (fp32) / 32 (fp64), middle extent % 8 or 4)";
    return false;
  }
  // with masks the whole (padded) field is addressed with 32-bit offsets
  const long long widest = std::max(pop_stride_of(p), p->resident != 0 ? resident_stride(p) : 0ll);
  const bool big = masked_big_exists(p->desc.layout, p->desc.collision, p->unit.q);
  if (!two_step_addressable(p->unit.d, p->unit.q, p->esize, p->n0, p->n1, p->n2, p->masked != 0, big) ||
      (p->masked && p->unit.d == 3 && (big ? 1ll : (long long)p->unit.q) * widest * p->esize >= (1ll << 32))) {
    *why = p->masked ? "with boundaries the two-step kernel addresses with 32-bit offsets: q * nodes * sizeof(scalar) (BGK, "
                       "reference layout: nodes * sizeof(scalar)) must stay below 4 GiB"
                     : "a plane of the grid exceeds the two-step kernel's 32-bit in-plane offsets";
    return false;
  }
  if (p->masked && (!masked_two_step_ok(p) || (p->desc.ghost_planes && p->n_abb > 0 && masked_two_step_axis(p) != 0))) {
    *why = "boundaries: at most one anti-bounce-back outlet, at the last plane of the slowest memory axis (periodic "
           "plans only) or at an end of the contiguous axis opposite a face of equilibrium nodes; no-streaming bits "
           "exactly that outlet's";
    return false;
  }
  lt::StepArgs a;
  memset(&a, 0, sizeof a);
  a.layout = p->desc.layout; a.coll = coll_of(p); a.mode = lt::kFusedTwice;
  a.masked = p->masked;
  a.abb_axis = p->masked ? masked_two_step_axis(p) : 2;
  a.strip = p->unit.d == 2 ? tile.width : 0;
  if (!p->unit.name(a)) {
    *why = "no two-step kernel for this lattice / dtype / collision";
    return false;
  }
  // with boundaries: the kernel is held against two one-step launches once per plan and set of masks (run_canary)
  if (p->masked && !p->canary_running && !canary_ok(p)) {
    *why = p->canary_msg;
    return false;
  }
  return true;
}

// Does lt_run pair its fused steps?  "automatic" also asks for the streaming regime (populations beyond the
// caches), where halving the HBM passes pays.
bool two_step_wanted(lt_plan *p) {
  if (p->two_step == 0 || p->desc.ghost_planes) return false;
  if (!two_step_possible(p, nullptr)) return false;
  if (p->two_step == 1) return true;
  // KBC inside the masked two-step kernel agrees with the one-step kernel at rounding level only (hipcc contracts
  // its multiply-adds differently in the two inlining contexts): never automatic, so that the result of n steps
  // does not depend on how the caller splits them into batches
  if (p->desc.collision == LT_COLLISION_KBC) return false;
  // fp64 plans with boundaries whose tile has only 4 rows (D3Q19: the third downward slot does not fit beside 8) are
  // slower than one update per launch: Obstacle D3Q19 256^3 fp64 1.009 against 0.880 ms per update
  // (tools/fp64_masked_two_step_probe.py; 4 waves per workgroup, 1.6 x redundant first step) -- opt-in only
  if (p->masked && p->esize == 8 && p->unit.d == 3 && two_step_tile(p).rows == 4) return false;
  const long long bytes = 2ll * p->unit.q * p->N * p->esize;
  return bytes > (128ll << 20);
}

// ---- first-use canary of the masked two-step kernels -----------------------------------------------------------
// The boundary dispatch of these kernels is the code hipcc has miscompiled before (Makefile header, DESIGN.md
// section 6: a register copy dropped at the join of the equilibrium branch -- wrong populations on inlet / outlet
// nodes of one instantiation, right everywhere else, and different from build to build).  The build no longer runs
// the pass at fault, but a green test on one compiler build does not protect the next one: the FIRST time a plan
// with masks is about to use a two-step kernel, one double step over all its planes is held against two one-step
// launches -- synthetic populations on scratch buffers, the plan's own masks, boundary table, tile and segment
// length, bit-for-bit comparison on the device.  A mismatch (or scratch memory that cannot be had) keeps the plan on
// the one-step kernel: two_step_possible() then says why, lt_run falls back by itself and leaves the reason in
// lt_last_error(), the explicit two-step entry points fail with it.  Run again after lt_plan_set_masks and after a
// change of the segment length or tile variant.  Costs four scratch fields and five launches, once.
// Reference semantics being protected: lettuce/_simulation.py:177-189 (collision, then the boundaries in index order).
void canary_verdict(lt_plan *p, int verdict, long long mismatches, const char *fmt, ...) {
  p->canary = verdict;
  p->canary_mismatches = mismatches;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(p->canary_msg, sizeof p->canary_msg, fmt, ap);
  va_end(ap);
  if (verdict < 0) snprintf(g_error, sizeof g_error, "%s", p->canary_msg);
}

template <typename T, typename U>
void run_canary(lt_plan *p) {
  const int g = p->desc.ghost_planes;
  if (g == 1) { canary_verdict(p, 2, 0, "no two-step launch on a plan with one ghost plane"); return; }
  if (!p->cstream && hipStreamCreateWithFlags(&p->cstream, hipStreamNonBlocking) != hipSuccess) {
    canary_verdict(p, -1, 0, "two steps per launch with boundaries: cannot create the canary's stream; the plan keeps "
                             "the one-step kernel");
    return;
  }
  const size_t bytes = (size_t)p->unit.q * p->N * p->esize;
  void *buf[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long *count = nullptr;
  bool ok = hipMalloc((void **)&count, sizeof *count) == hipSuccess;
  for (void *&b : buf) ok = ok && hipMalloc(&b, bytes) == hipSuccess;
  auto release = [&]() { for (void *b : buf) if (b) (void)hipFree(b); if (count) (void)hipFree(count); };
  if (!ok) {
    release();
    canary_verdict(p, -1, 0, "two steps per launch with boundaries: no memory for the first-use check (4 x %zu "
                             "bytes); the plan keeps the one-step kernel", bytes);
    return;
  }
  // the plan as the caller set it up, but on dense scratch buffers and without the slab drivers' per-launch state
  const long long s_in = p->stride_in_now, s_out = p->stride_out_now, sb = p->second_begin, se = p->second_end;
  unsigned long long *const sig = p->signal_now;
  const void *const glo = p->ghost_lo_now, *const ghi = p->ghost_hi_now;
  p->stride_in_now = p->stride_out_now = p->N;
  p->second_begin = p->second_end = 0;
  p->signal_now = nullptr;
  p->ghost_lo_now = p->ghost_hi_now = nullptr;
  p->canary_running = 1;
  char saved_error[sizeof g_error];
  memcpy(saved_error, g_error, sizeof g_error);
  const double tau = 0.8;
  const long long b1 = g ? 1 : 0, e1 = g ? p->n2 - 1 : p->n2, b2 = g, e2 = p->n2 - g;
  hipLaunchKernelGGL((canary_fill_kernel<T>), dim3(1024), dim3(256), 0, p->cstream, (T *)buf[0], p->N, p->N, p->unit.q);
  if (g) for (int i = 1; i < 4; ++i) (void)hipMemsetAsync(buf[i], 0, bytes, p->cstream);   // planes a slab launch never writes
  (void)hipMemsetAsync(count, 0, sizeof *count, p->cstream);
  int rc = step(p, lt::kFused, buf[0], buf[1], tau, b1, e1, p->cstream);
  if (rc == LT_OK) rc = step(p, lt::kFused, buf[1], buf[2], tau, b2, e2, p->cstream);
  if (rc == LT_OK) rc = step(p, lt::kFusedTwice, buf[0], buf[3], tau, b2, e2, p->cstream);
  unsigned long long bad = 0;
  hipError_t e = hipSuccess;
  if (rc == LT_OK) {
    const long long plane = (long long)p->n0 * p->n1;
    hipLaunchKernelGGL((canary_compare_kernel<U>), dim3(1024), dim3(256), 0, p->cstream, (const U *)buf[2],
                       (const U *)buf[3], p->N, plane * b2, plane * (e2 - b2), p->unit.q, count);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, count, sizeof bad, hipMemcpyDeviceToHost, p->cstream);
  }
  const hipError_t es = hipStreamSynchronize(p->cstream);
  if (e == hipSuccess) e = es;
  p->canary_running = 0;
  p->stride_in_now = s_in; p->stride_out_now = s_out;
  p->second_begin = sb; p->second_end = se;
  p->signal_now = sig;
  p->ghost_lo_now = glo; p->ghost_hi_now = ghi;
  release();
  if (rc != LT_OK) {
    char why[sizeof g_error];
    memcpy(why, g_error, sizeof why);
    canary_verdict(p, -1, 0, "two steps per launch with boundaries: the first-use check could not run (%.160s); the "
                             "plan keeps the one-step kernel", why);
  } else if (e != hipSuccess) {
    canary_verdict(p, -1, 0, "two steps per launch with boundaries: the first-use check failed with a HIP error (%s); "
                             "the plan keeps the one-step kernel", hipGetErrorString(e));
  } else if (bad != 0) {
    canary_verdict(p, -1, (long long)bad, "two steps per launch with boundaries: the two-step kernel disagrees with two "
                   "one-step launches in %llu of %lld populations (first-use check on this plan's masks); the plan "
                   "keeps the one-step kernel", bad, (long long)p->unit.q * p->n0 * p->n1 * (e2 - b2));
  } else {
    memcpy(g_error, saved_error, sizeof g_error);
    canary_verdict(p, 1, 0, "");
  }
}

// may this plan's masked two-step kernel be used?  Runs the check if it has not been run for the present masks
bool canary_ok(lt_plan *p) {
  if (!p->masked) return true;
  if (p->canary == 0) {
    if (p->canary_mode == 0) canary_verdict(p, 2, 0, "");
    else if (p->canary_mode == 2)
      canary_verdict(p, -1, -1, "two steps per launch with boundaries: first-use check forced to fail "
                                "(lt_plan_set_canary(plan, 2)); the plan keeps the one-step kernel");
    else if (p->esize == 4) run_canary<float, unsigned>(p);
    else run_canary<double, unsigned long long>(p);
  }
  return p->canary > 0;
}

constexpr int kManyMax = 8;        // == kManyMax of unit.inc
// steps per many-step launch: a plan with an outlet recomputes one more ring of nodes around each tile; the 3-D
// kernel does two (its neighbourhood of an 8^3 tile is 10^3 nodes: q x 1000 values of LDS)
int many_max(const lt_plan *p) { return p->unit.d == 3 ? 2 : kManyMax - ((p->masked && p->n_abb > 0) ? 1 : 0); }

// Several steps per launch (lbm_many_kernel): 2-D, tiles of 8 x 8, with masks at most one outlet.  Every
// workgroup recomputes a halo of K - 1 nodes around its tile (K with an outlet), so this only pays while the
// grid is launch-bound; "automatic" stops at 256 x 256 nodes, 256 x 128 with masks (measured,
// tools/small_grid_bench.py, tools/small_masked_bench.py).
bool many_step_wanted(lt_plan *p) {
  if (p->many == 0 || p->desc.ghost_planes || p->unit.d < 2 || p->arith != 0) return false;
  if (p->unit.d == 3 && !LT_EXPERIMENTS) return false;       // lbm_many3d_kernel: slower than two launches on every grid
  if (p->masked && (p->n_abb > 1 || p->unit.d == 3)) return false;
  if (p->n0 % 8 != 0 || p->n1 % 8 != 0 || (p->unit.d == 3 && p->n2 % 8 != 0)) return false;
  if (p->unit.d == 3 && p->desc.layout != LT_LAYOUT_REFERENCE) return false;
  lt::StepArgs a;
  memset(&a, 0, sizeof a);
  a.layout = p->desc.layout; a.coll = coll_of(p); a.mode = lt::kFusedMany;
  a.masked = p->masked;
  if (!p->unit.name(a)) return false;
  if (p->many == 1) return true;
  // automatic only where the many-step kernel is bit-identical to the one-step kernel, so that the
  // result of n steps does not depend on how the caller splits them into batches (KBC agrees at
  // rounding level only)
  if (p->desc.collision == LT_COLLISION_KBC) return false;
  // with masks the launch is bound by its own latency chain (node byte, boundary values, a barrier per step) and
  // recomputes a wider ring: 128 x 64 nodes 4.2 -> 2.4 us per step, 128 x 128 4.3 -> 2.5, 256 x 128 4.5 -> 3.3,
  // 256 x 256 5.6 -> 5.7 (tools/small_masked_bench.py, fp64)
  if (p->masked) return p->N <= 256ll * 128ll;
  // 3-D (lbm_many3d_kernel, two steps per launch): never automatic.  Measured slower than one launch per step on
  // every grid (tools/small_grid_bench.py, D3Q19 fp32, us per step: 16^3 3.66 -> 4.29, 32^3 4.21 -> 5.79, 48^3 6.5 ->
  // 7.9, 64^3 8.3 -> 15.4): two steps amortise one launch gap (~2 us) but the launch is a chain of two dependent
  // gather -> collide phases with a barrier in between, on 1000-thread workgroups that gather 10-node rows; the 2-D
  // kernel wins because it amortises EIGHT steps per launch, which the LDS does not allow in 3-D (K = 3 needs the
  // 12^3 neighbourhood: 131 KB for D3Q19 fp32 and 3.4 x the arithmetic).
  if (p->unit.d == 3) return false;
  return p->N <= 256ll * 256ll;
}

// `fused` stream-collide steps starting from the post-collision populations in *cur, ping-ponging with *other (both
// pointers are updated: on return *cur holds the result): pairs of steps through the two-step kernel where it
// applies, the rest one by one; several steps per launch on small 2-D grids.  The optional events bracket the
// dominant kind of launch only.
int fused_section(lt_plan *p, void *&cur, void *&other, double tau, long long fused, void *stream) {
  int rc;
  if (graph_wanted(p, fused) && p->stride_in_now < 0) {
    const long long done = run_graph(p, cur, other, tau, fused, stream);
    if (done < 0) return (int)-done;
    fused -= done;                       // an even number of steps: still in `cur`
  }
  hipStream_t hs = static_cast<hipStream_t>(stream);
  p->last_many = 0;
  if (many_step_wanted(p) && fused >= 2) {
    // launches of up to kManyMax steps each; the events bracket them
    long long odd = 0;
    if (p->ev_start) (void)hipEventRecord(p->ev_start, hs);
    while (fused > 0) {
      p->many_now = fused < many_max(p) ? (int)fused : many_max(p);
      // the 3-D kernel does exactly two steps: an odd one at the end is an ordinary launch
      const int mode = (p->unit.d == 3 && p->many_now == 1) ? lt::kFused : lt::kFusedMany;
      rc = step(p, mode, cur, other, tau, 0, p->n2, stream);
      if (rc) return rc;
      void *t = cur; cur = other; other = t;
      fused -= p->many_now;
      if (mode == lt::kFusedMany) ++p->last_many; else ++odd;
    }
    if (p->ev_stop) (void)hipEventRecord(p->ev_stop, hs);
    p->last_twice = 0;
    p->last_single = odd;
    return LT_OK;
  }
  const long long twice = two_step_wanted(p) ? fused / 2 : 0;
  const long long single = fused - 2 * twice;
  p->last_twice = twice; p->last_single = single;
  if (p->ev_start && twice > 0) (void)hipEventRecord(p->ev_start, hs);
  for (long long i = 0; i < twice; ++i) {
    rc = step(p, lt::kFusedTwice, cur, other, tau, 0, p->n2, stream);
    if (rc) return rc;
    void *t = cur; cur = other; other = t;
  }
  if (p->ev_stop && twice > 0) (void)hipEventRecord(p->ev_stop, hs);
  if (p->ev_start && twice == 0) (void)hipEventRecord(p->ev_start, hs);
  for (long long i = 0; i < single; ++i) {
    rc = step(p, lt::kFused, cur, other, tau, 0, p->n2, stream);
    if (rc) return rc;
    void *t = cur; cur = other; other = t;
  }
  if (p->ev_stop && twice == 0) (void)hipEventRecord(p->ev_stop, hs);
  return LT_OK;
}

int run(lt_plan *p, bool from_fstar, void *a, void *b, double tau, long long n, void *stream,
        int32_t *result_in_b) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (n < 1) return fail(LT_ERR_INVALID, "n_steps = %lld", n);
  if (!result_in_b) return fail(LT_ERR_INVALID, "null result_in_b");
  if (p->desc.ghost_planes)
    return fail(LT_ERR_UNSUPPORTED, "lt_run/lt_continue need a periodic plan (no ghost planes); "
                                    "drive slabs with the *_planes entry points");
  void *cur = a, *other = b;
  int rc;
  long long fused = n;
  if (!from_fstar) {
    rc = step(p, lt::kCollideOnly, cur, other, tau, 0, p->n2, stream);
    if (rc) return rc;
    void *t = cur; cur = other; other = t;
    fused = n - 1;
  }
  rc = fused_section(p, cur, other, tau, fused, stream);
  if (rc) return rc;
  if (p->defer_stream) {               // the caller streams when somebody wants to see the populations
    *result_in_b = (cur == b) ? 1 : 0;
    return LT_OK;
  }
  rc = step(p, lt::kStreamOnly, cur, other, tau, 0, p->n2, stream);
  if (rc) return rc;
  *result_in_b = (other == b) ? 1 : 0;
  return LT_OK;
}

int aux(lt_plan *p, int what, const void *f, void *rho, void *u, double *out, void *stream,
        double scale = 1.0, double inv_dx = 1.0, const unsigned char *mask = nullptr, long long u_stride = 0,
        int z_begin = 0, int nz_global = 0) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (!f && what != 8) return fail(LT_ERR_INVALID, "null population buffer");
  lt::AuxArgs a;
  memset(&a, 0, sizeof a);
  a.what = what; a.layout = p->desc.layout;
  a.f = f; a.rho = rho; a.u = u;
  a.N = p->N;
  a.stride = pop_stride_of(p);
  a.u_stride = u_stride; a.z_begin = z_begin; a.nz_global = nz_global;
  if (a.stride != a.N && what != 0 && what != 2 && what != 3 && what != 4 && what != 8 && what != 9)
    return fail(LT_ERR_UNSUPPORTED, "auxiliary kernel %d reads dense population buffers only (the plan has a "
                                    "population stride of %lld)", what, a.stride);
  const long long plane = (long long)p->n0 * p->n1;
  a.first = plane * p->interior_begin;
  a.count = plane * (p->interior_end - p->interior_begin);
  a.partial = p->partial; a.reduce_blocks = kReduceBlocks; a.out = out;
  a.n0 = p->n0; a.n1 = p->n1; a.n2 = what == 8 ? (p->interior_end - p->interior_begin) + 6 : p->n2;
  a.scale = scale; a.inv_dx = inv_dx; a.mask = mask;
  a.stream = static_cast<hipStream_t>(stream);
  const int r = p->unit.aux(a);
  if (r == lt::kNoKernel) return fail(LT_ERR_UNSUPPORTED, "no auxiliary kernel %d", what);
  if (r != 0) return fail(LT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)r));
  return LT_OK;
}

}  // namespace

extern "C" {

int lt_abi_version(void) { return LT_ABI_VERSION; }
int lt_build_flags(void) { return LT_EXPERIMENTS ? 1 : 0; }
const char *lt_last_error(void) { return g_error; }

int lt_plan_create(const lt_plan_desc *d, lt_plan **out) {
  if (!d || !out) return fail(LT_ERR_INVALID, "null argument");
  *out = nullptr;
  if (d->abi_version != LT_ABI_VERSION)
    return fail(LT_ERR_INVALID, "ABI version %d, library is %d", d->abi_version, LT_ABI_VERSION);
  if (d->stencil < 0 || d->stencil > 4) return fail(LT_ERR_UNSUPPORTED, "stencil %d", d->stencil);
  if (d->dtype < 0 || d->dtype > 1) return fail(LT_ERR_UNSUPPORTED, "dtype %d (fp32/fp64 only)", d->dtype);
  if (d->collision < 0 || d->collision > 2) return fail(LT_ERR_UNSUPPORTED, "collision %d", d->collision);
  const Unit unit = kUnits[d->stencil][d->dtype];
  if (d->dims != unit.d) return fail(LT_ERR_INVALID, "stencil is %d-dimensional, dims = %d", unit.d, d->dims);
  if (d->collision == LT_COLLISION_KBC && d->stencil != LT_D2Q9 && d->stencil != LT_D3Q27)
    return fail(LT_ERR_UNSUPPORTED, "KBC collision exists for D2Q9 and D3Q27 only");
  if (d->layout != LT_LAYOUT_REFERENCE && d->layout != LT_LAYOUT_SLAB)
    return fail(LT_ERR_INVALID, "layout %d", d->layout);
  if (d->layout == LT_LAYOUT_SLAB && unit.d != 3) return fail(LT_ERR_UNSUPPORTED, "slab layout is 3-D only");
  if (d->ghost_planes != 0 &&
      !((d->ghost_planes == 1 || d->ghost_planes == 2) && d->layout == LT_LAYOUT_SLAB))
    return fail(LT_ERR_INVALID, "ghost_planes = %d (0, or 1 / 2 with the slab layout)", d->ghost_planes);
  if (d->n_boundaries < 0 || d->n_boundaries > LT_MAX_BOUNDARIES)
    return fail(LT_ERR_UNSUPPORTED, "%d boundaries (max %d)", d->n_boundaries, LT_MAX_BOUNDARIES);
  for (int a = 0; a < unit.d; ++a)
    if (d->shape[a] < 1) return fail(LT_ERR_INVALID, "shape[%d] = %lld", a, (long long)d->shape[a]);

  lt_plan *p = new (std::nothrow) lt_plan;
  if (!p) return fail(LT_ERR_ALLOC, "out of host memory");
  p->desc = *d;
  p->unit = unit;
  p->esize = d->dtype == LT_F32 ? 4 : 8;
  long long e0, e1, e2;
  if (unit.d == 1) { e0 = d->shape[0]; e1 = 1; e2 = 1; }
  else if (unit.d == 2) { e0 = d->shape[1]; e1 = d->shape[0]; e2 = 1; }
  else if (d->layout == LT_LAYOUT_REFERENCE) { e0 = d->shape[2]; e1 = d->shape[1]; e2 = d->shape[0]; }
  else { e0 = d->shape[0]; e1 = d->shape[1]; e2 = d->shape[2] + 2 * d->ghost_planes; }
  if (e0 * e1 * e2 >= (1ll << 31)) {
    delete p;
    return fail(LT_ERR_UNSUPPORTED, "%lld nodes per rank exceed the 2^31 index range", e0 * e1 * e2);
  }
  p->n0 = (int)e0; p->n1 = (int)e1; p->n2 = (int)e2;
  p->N = e0 * e1 * e2;
  p->interior_begin = d->ghost_planes;
  p->interior_end = p->n2 - d->ghost_planes;
  p->wide_ok = (e0 % (16 / p->esize)) == 0;
  p->shift = 0;
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      p->n_cu = cus;
  }
  int n_abb = 0;
  for (int i = 0; i < d->n_boundaries; ++i) {
    const int rc = check_boundary(p, d->boundaries[i], n_abb);
    if (rc) { delete p; return rc; }
    if (d->boundaries[i].kind == LT_BOUNDARY_ABB_OUTLET) ++n_abb;
  }
  p->n_abb = n_abb;
  // How deep the kernels must rebuild an outlet's neighbour (neighbour_moments, DEPTH): where the planes of outlets
  // meet, an outlet's neighbour node has been rewritten by the outlets with a lower index -- and their neighbours by
  // still lower ones.  Outlets on opposite faces of one axis never meet, so the chain is as long as the number of
  // distinct AXES the outlets lie on, minus one.  Kernels exist for depth 0 (one outlet), 1 (any list of outlets on at
  // most two axes: every 2-D flow) and, in the reference layout, 2 (outlets on all three axes, whose planes meet in
  // corners) -- the reference takes any list, lettuce/_simulation.py:57-86.
  {
    int axes = 0;
    for (int i = 0; i < d->n_boundaries; ++i)
      if (d->boundaries[i].kind == LT_BOUNDARY_ABB_OUTLET) axes |= 1 << d->boundaries[i].axis;
    const int n_axes = (axes & 1) + ((axes >> 1) & 1) + ((axes >> 2) & 1);
    p->abb_depth = n_abb <= 1 ? 0 : (n_axes <= 2 ? 1 : n_axes - 1);
    if (p->abb_depth > 1 && d->layout != LT_LAYOUT_REFERENCE) {
      lt_plan_destroy(p);
      return fail(LT_ERR_UNSUPPORTED, "AntiBounceBackOutlets on all three axes (their planes meet in corners, where an "
                                      "outlet's neighbour depends on two earlier outlets) exist for the reference "
                                      "layout only; on slabs outlets on one or two axes run, in any number");
    }
  }
  const size_t bt_size = d->dtype == LT_F32 ? sizeof(lt::BoundaryTable<float>)
                                            : sizeof(lt::BoundaryTable<double>);
  if (hipMalloc(&p->bt, bt_size) != hipSuccess ||
      hipMalloc((void **)&p->partial, kReduceBlocks * sizeof(double)) != hipSuccess) {
    lt_plan_destroy(p);
    return fail(LT_ERR_ALLOC, "hipMalloc of plan scratch failed");
  }
  const int rc = d->dtype == LT_F32 ? upload_boundaries<float>(p, nullptr)
                                    : upload_boundaries<double>(p, nullptr);
  if (rc) { lt_plan_destroy(p); return rc; }
  *out = p;
  return LT_OK;
}

int lt_plan_destroy(lt_plan *p) {
  if (!p) return LT_OK;
  if (p->node) (void)hipFree(p->node);
  if (p->nsm_bits) (void)hipFree(p->nsm_bits);
  if (p->bt) (void)hipFree(p->bt);
  if (p->mask_flag) (void)hipFree(p->mask_flag);
  if (p->signal) (void)hipFree(p->signal);
  if (p->signal_timed_out) (void)hipFree(p->signal_timed_out);
  if (p->partial) (void)hipFree(p->partial);
  for (void *r : p->res) if (r) (void)hipFree(r);
  if (p->gexec) (void)hipGraphExecDestroy(p->gexec);
  if (p->gev_in) (void)hipEventDestroy(p->gev_in);
  if (p->gev_out) (void)hipEventDestroy(p->gev_out);
  if (p->gstream) (void)hipStreamDestroy(p->gstream);
  if (p->cstream) (void)hipStreamDestroy(p->cstream);
  delete p;
  return LT_OK;
}

int lt_plan_set_masks(lt_plan *p, const uint8_t *ncm, const uint8_t *nsm, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (!ncm && !nsm) {
    if (p->desc.n_boundaries > 0)
      return fail(LT_ERR_INVALID, "a plan with boundaries needs a no_collision_mask");
    p->masked = 0;
    return LT_OK;
  }
  if (!p->node) LT_HIP(hipMalloc((void **)&p->node, (size_t)p->N));
  if (nsm && !p->nsm_bits) LT_HIP(hipMalloc((void **)&p->nsm_bits, (size_t)p->N * sizeof(unsigned)));
  if (!p->mask_flag) LT_HIP(hipMalloc((void **)&p->mask_flag, sizeof(unsigned)));
  hipStream_t hs = static_cast<hipStream_t>(stream);
  LT_HIP(hipMemsetAsync(p->mask_flag, 0, sizeof(unsigned), hs));
  // the only no-streaming bits the two-step kernel can take: those of ONE outlet at the last plane of the
  // slowest memory axis or at an end of the rows -- the populations entering through it, on every node of it
  int axis = -1, plane = -1, face = -1, outlets = 0;
  unsigned expected = 0, eq_slots = 0;
  const int sweep = p->unit.d == 2 ? 1 : 2;          // the slowest memory axis: rows in 2-D, planes in 3-D
  for (int i = 0; i < p->desc.n_boundaries; ++i) {
    const lt_boundary_desc &b = p->desc.boundaries[i];
    if (b.kind == LT_BOUNDARY_EQUILIBRIUM && i + 1 < 32) eq_slots |= 1u << (i + 1);   // (only read for plans the two-step kernels admit: <= 15 boundaries)
    if (b.kind != LT_BOUNDARY_ABB_OUTLET || (b.flags & LT_BOUNDARY_ABSENT)) continue;
    ++outlets;
    const int ax = mem_axis_of(p, b.axis);
    if ((ax == sweep && b.side == 1) || (ax == 0 && p->unit.d == 3)) {
      axis = ax == sweep ? 2 : 0;                    // compile_masks_kernel: 2 = "index / plane_nodes", 0 = a0 column
      plane = ax == sweep ? (sweep == 2 ? p->n2 : p->n1) - 1 - p->desc.ghost_planes : (b.side == 1 ? p->n0 - 1 : 0);
      if (ax == 0) face = b.side == 1 ? 0 : p->n0 - 1;
      const lt::QList in = crossing_axis(p, ax, -b.side);
      for (int k = 0; k < in.n; ++k) expected |= 1u << in.q[k];
    }
  }
  if (outlets != 1) { axis = -1; plane = -1; face = -1; expected = 0; }
  const unsigned grid = (unsigned)((p->N + lt::kThreads - 1) / lt::kThreads);
  hipLaunchKernelGGL(lt::compile_masks_kernel, dim3(grid), dim3(lt::kThreads), 0, hs, ncm, nsm, p->unit.q, p->N,
                     p->node, nsm ? p->nsm_bits : nullptr, sweep == 1 ? (long long)p->n0 : (long long)p->n0 * p->n1, p->n0,
                     axis, plane, expected,
                     face, eq_slots, p->mask_flag);
  LT_HIP(hipGetLastError());
  unsigned flags = 0;
  LT_HIP(hipMemcpyAsync(&flags, p->mask_flag, sizeof flags, hipMemcpyDeviceToHost, hs));
  LT_HIP(hipStreamSynchronize(hs));
  p->nsm_confined = (flags & 1u) == 0;
  p->inlet_faces_outlet = (flags & 2u) == 0;
  p->masked = 1;
  p->canary = 0;                                     // new masks: the first-use check runs again
  return LT_OK;
}

int lt_plan_update_boundary(lt_plan *p, int32_t index, const lt_boundary_desc *b, void *stream) {
  if (!p || !b) return fail(LT_ERR_INVALID, "null argument");
  if (index < 0 || index >= p->desc.n_boundaries) return fail(LT_ERR_INVALID, "boundary index %d", index);
  if (b->kind != p->desc.boundaries[index].kind)
    return fail(LT_ERR_INVALID, "boundary %d: kind cannot change (%d -> %d)", index,
                p->desc.boundaries[index].kind, b->kind);
  p->desc.boundaries[index] = *b;
  return p->desc.dtype == LT_F32 ? upload_boundaries<float>(p, static_cast<hipStream_t>(stream))
                                 : upload_boundaries<double>(p, static_cast<hipStream_t>(stream));
}

int lt_collide(lt_plan *p, const void *f, void *o, double tau, void *s) {
  return step(p, lt::kCollideOnly, f, o, tau, p ? p->interior_begin : 0, p ? p->interior_end : 0, s);
}
int lt_stream(lt_plan *p, const void *f, void *o, void *s) {
  return step(p, lt::kStreamOnly, f, o, 1.0, p ? p->interior_begin : 0, p ? p->interior_end : 0, s);
}
int lt_stream_collide(lt_plan *p, const void *f, void *o, double tau, void *s) {
  return step(p, lt::kFused, f, o, tau, p ? p->interior_begin : 0, p ? p->interior_end : 0, s);
}
int lt_collide_planes(lt_plan *p, const void *f, void *o, double tau, int64_t b, int64_t e, void *s) {
  return step(p, lt::kCollideOnly, f, o, tau, b, e, s);
}
int lt_stream_planes(lt_plan *p, const void *f, void *o, int64_t b, int64_t e, void *s) {
  return step(p, lt::kStreamOnly, f, o, 1.0, b, e, s);
}
int lt_stream_collide_planes(lt_plan *p, const void *f, void *o, double tau, int64_t b, int64_t e,
                             void *s) {
  return step(p, lt::kFused, f, o, tau, b, e, s);
}

int lt_stream_collide_plane_pair(lt_plan *p, const void *f, void *o, double tau, int64_t first,
                                 int64_t second, void *s) {
  if (second <= first) return fail(LT_ERR_INVALID, "plane pair (%lld, %lld)", (long long)first, (long long)second);
  return step(p, lt::kFused, f, o, tau, first, second + 1, s, second - first);
}

int lt_stream_collide_plane_pair_packed(lt_plan *p, const void *f, void *o, double tau,
                                        int64_t first, int64_t second, void *pack_first,
                                        void *pack_second, void *s) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (second < first) return fail(LT_ERR_INVALID, "plane pair (%lld, %lld)", (long long)first, (long long)second);
  if (!pack_first || !pack_second) return fail(LT_ERR_INVALID, "null pack buffer");
  if (p->desc.layout != LT_LAYOUT_SLAB) return fail(LT_ERR_UNSUPPORTED, "fused packing needs the slab layout");
  const long long stride = second > first ? second - first : 1;
  return step(p, lt::kFused, f, o, tau, first, second + 1, s, stride, pack_first, pack_second);
}

int lt_run(lt_plan *p, void *a, void *b, double tau, int64_t n, void *s, int32_t *r) {
  return run(p, false, a, b, tau, n, s, r);
}
int lt_continue(lt_plan *p, void *a, void *b, double tau, int64_t n, void *s, int32_t *r) {
  return run(p, true, a, b, tau, n, s, r);
}

int lt_macroscopic(lt_plan *p, const void *f, void *rho, void *u, void *s) {
  if (!rho && !u) return fail(LT_ERR_INVALID, "both outputs null");
  return aux(p, 0, f, rho, u, nullptr, s);
}
int lt_equilibrium(lt_plan *p, const void *rho, const void *u, void *feq, void *s) {
  if (!rho || !u) return fail(LT_ERR_INVALID, "null rho/u");
  return aux(p, 1, feq, const_cast<void *>(rho), const_cast<void *>(u), nullptr, s);
}
int lt_kinetic_energy(lt_plan *p, const void *f, double *out, void *s) {
  if (!out) return fail(LT_ERR_INVALID, "null output");
  return aux(p, 2, f, nullptr, nullptr, out, s);
}
int lt_mass(lt_plan *p, const void *f, double *out, void *s) {
  if (!out) return fail(LT_ERR_INVALID, "null output");
  return aux(p, 3, f, nullptr, nullptr, out, s);
}
int lt_max_velocity(lt_plan *p, const void *f, double *out, void *s) {
  if (!out) return fail(LT_ERR_INVALID, "null output");
  return aux(p, 4, f, nullptr, nullptr, out, s);
}

int lt_init_fneq(lt_plan *p, const void *rho, const void *u, double tau, double identity_cs2, void *f, void *s) {
  if (!rho || !u) return fail(LT_ERR_INVALID, "null rho/u");
  if (p && (p->unit.d < 2 || p->desc.layout != LT_LAYOUT_REFERENCE || p->desc.ghost_planes))
    return fail(LT_ERR_UNSUPPORTED, "f_neq initialisation: 2-D / 3-D grids in the reference layout");
  if (!(tau > 0.0)) return fail(LT_ERR_INVALID, "relaxation time tau = %g", tau);
  return aux(p, 7, f, const_cast<void *>(rho), const_cast<void *>(u), nullptr, s, tau, identity_cs2);
}
int lt_enstrophy(lt_plan *p, const void *f, void *u_scratch, double u_scale, double inv_dx, double *out, void *s) {
  if (!out || !u_scratch) return fail(LT_ERR_INVALID, "null output / scratch");
  if (p && (p->unit.d < 2 || p->desc.layout != LT_LAYOUT_REFERENCE || p->desc.ghost_planes))
    return fail(LT_ERR_UNSUPPORTED, "enstrophy: 2-D / 3-D periodic grids in the reference layout");
  return aux(p, 5, f, nullptr, u_scratch, out, s, u_scale, inv_dx);
}
int lt_mass_interior(lt_plan *p, const void *f, const uint8_t *mask, double *out, void *s) {
  if (!out) return fail(LT_ERR_INVALID, "null output");
  if (p && (p->unit.d < 2 || p->desc.layout != LT_LAYOUT_REFERENCE || p->desc.ghost_planes))
    return fail(LT_ERR_UNSUPPORTED, "interior mass: 2-D / 3-D grids in the reference layout");
  return aux(p, 6, f, nullptr, nullptr, out, s, 1.0, 1.0, mask);
}

namespace {
int slab_plan_ok(lt_plan *p, const char *who) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (p->desc.layout != LT_LAYOUT_SLAB || p->desc.ghost_planes < 1)
    return fail(LT_ERR_UNSUPPORTED, "%s: for plans in the slab layout with ghost planes (a rank's share of the "
                                    "observable); single-domain plans have lt_enstrophy / lt_mass_interior", who);
  return LT_OK;
}
}  // namespace

int lt_slab_velocity(lt_plan *p, const void *f, void *u_ext, void *s) {
  if (const int rc = slab_plan_ok(p, "lt_slab_velocity")) return rc;
  if (!u_ext) return fail(LT_ERR_INVALID, "null velocity field");
  const long long plane = (long long)p->n0 * p->n1;
  const long long owned = p->interior_end - p->interior_begin;
  // node i of the plan (its ghost planes included) is node i + (3 - g) planes of the extended field
  char *u0 = static_cast<char *>(u_ext) + (3 - p->desc.ghost_planes) * plane * p->esize;
  return aux(p, 0, f, nullptr, u0, nullptr, s, 1.0, 1.0, nullptr, (owned + 6) * plane);
}
int lt_slab_enstrophy(lt_plan *p, const void *u_ext, double u_scale, double inv_dx, double *out, void *s) {
  if (const int rc = slab_plan_ok(p, "lt_slab_enstrophy")) return rc;
  if (!out || !u_ext) return fail(LT_ERR_INVALID, "null output / velocity field");
  return aux(p, 8, nullptr, nullptr, const_cast<void *>(u_ext), out, s, u_scale, inv_dx);
}
int lt_slab_mass_interior(lt_plan *p, const void *f, const uint8_t *mask, int32_t z_begin, int32_t nz_global,
                          double *out, void *s) {
  if (const int rc = slab_plan_ok(p, "lt_slab_mass_interior")) return rc;
  if (!out) return fail(LT_ERR_INVALID, "null output");
  const int owned = p->interior_end - p->interior_begin;
  if (z_begin < 0 || nz_global < 1 || z_begin + owned > nz_global)
    return fail(LT_ERR_INVALID, "planes [%d, %d) of %d", z_begin, z_begin + owned, nz_global);
  return aux(p, 9, f, nullptr, nullptr, out, s, 1.0, 1.0, mask, 0, z_begin, nz_global);
}

int lt_plan_kernel_info(lt_plan *p, int32_t *vec, int32_t *tpb, int64_t *blocks) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  const int v = (p->want_wide && p->wide_ok) ? 16 / p->esize : 1;
  if (vec) *vec = v;
  if (tpb) *tpb = lt::kThreads;
  if (blocks) {
    const long long nodes = (long long)p->n0 * p->n1 * (p->interior_end - p->interior_begin);
    *blocks = (nodes / v + lt::kThreads - 1) / lt::kThreads;
  }
  return LT_OK;
}

const char *lt_plan_kernel_name(lt_plan *p) {
  if (!p) return "";
  lt::StepArgs a;
  memset(&a, 0, sizeof a);
  a.layout = p->desc.layout; a.coll = coll_of(p);
  // what the fused section launches: lt_run's pairs, or a two-step slab driver on a plan with two
  // ghost planes
  a.mode = (two_step_wanted(p) || p->desc.ghost_planes == 2) ? lt::kFusedTwice : lt::kFused;
  if (many_step_wanted(p)) a.mode = lt::kFusedMany;
  a.masked = p->masked;
  a.abb_depth = p->abb_depth;
  a.abb_axis = p->masked ? masked_two_step_axis(p) : 2;
  a.strip = p->unit.d == 2 ? two_step_tile(p).width : 0;
  if (a.mode == lt::kFusedTwice && !p->unit.name(a)) a.mode = lt::kFused;
  a.wide = (p->want_wide && p->wide_ok && !a.masked && a.coll == LT_COLLISION_BGK) ? 1 : 0;
  a.shift = a.wide ? p->shift : 0;
  a.tune = resolve_tune(p, a.wide);
  const char *n = p->unit.name(a);
  snprintf(p->kernel_name, sizeof p->kernel_name, "%s", n ? n : "");
  return p->kernel_name;
}

int lt_slab_crossing(lt_plan *p, int32_t direction, int32_t *q_out, int32_t *n_out) {
  if (!p || !n_out) return fail(LT_ERR_INVALID, "null argument");
  const lt::QList ql = crossing(p, direction);
  *n_out = ql.n;
  if (q_out) for (int k = 0; k < ql.n; ++k) q_out[k] = ql.q[k];
  return LT_OK;
}
int lt_slab_pack(lt_plan *p, const void *f, int64_t plane, int32_t direction, void *buf, void *s) {
  return pack(p, true, const_cast<void *>(f), plane, direction, buf, s);
}
int lt_slab_unpack(lt_plan *p, void *f, int64_t plane, int32_t direction, const void *buf, void *s) {
  return pack(p, false, f, plane, direction, const_cast<void *>(buf), s);
}

int lt_probe_copy(void *dst, const void *src, int64_t n_bytes, int32_t cache_policy,
                  int32_t max_blocks, void *stream) {
  if (!dst || !src || n_bytes < 16 || n_bytes % 16 != 0 || (uintptr_t)dst % 16 || (uintptr_t)src % 16)
    return fail(LT_ERR_INVALID, "probe copy needs 16-byte aligned buffers and size");
  const size_t n = (size_t)n_bytes / 16;
  size_t grid = (n + 255) / 256;
  if (max_blocks > 0 && grid > (size_t)max_blocks) grid = (size_t)max_blocks;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const probe_f4 *sp = static_cast<const probe_f4 *>(src);
  probe_f4 *dp = static_cast<probe_f4 *>(dst);
  switch (cache_policy) {
    case 0: hipLaunchKernelGGL(probe_copy_kernel<0>, dim3((unsigned)grid), dim3(256), 0, s, sp, dp, n); break;
    case 1: hipLaunchKernelGGL(probe_copy_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, sp, dp, n); break;
    case 2: hipLaunchKernelGGL(probe_copy_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, sp, dp, n); break;
    case 3: hipLaunchKernelGGL(probe_copy_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, sp, dp, n); break;
    default: return fail(LT_ERR_INVALID, "cache policy %d", cache_policy);
  }
  LT_HIP(hipGetLastError());
  return LT_OK;
}

int lt_probe_div_cs(const void *x, void *out, int64_t n, int32_t dtype, int32_t which, void *stream) {
  if (!x || !out || n < 1 || (which != 0 && which != 1) || (dtype != LT_F32 && dtype != LT_F64))
    return fail(LT_ERR_INVALID, "probe_div_cs: null buffer, n = %lld, which = %d, dtype = %d", (long long)n, which, dtype);
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == LT_F32)
    hipLaunchKernelGGL(probe_div_cs_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)x, (float *)out, (long long)n, which);
  else
    hipLaunchKernelGGL(probe_div_cs_kernel<double>, dim3(grid), dim3(256), 0, s, (const double *)x, (double *)out, (long long)n, which);
  LT_HIP(hipGetLastError());
  return LT_OK;
}

int lt_plan_set_shift_policy(lt_plan *p, int32_t policy) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (policy < 0 || policy > 5) return fail(LT_ERR_INVALID, "shift policy %d", policy);
  if (!LT_EXPERIMENTS && (policy == 1 || policy == 2 || policy == 5))
    return fail(LT_ERR_UNSUPPORTED, "shift policy %d is a tile variant of the experiments build (make EXPERIMENTS=1)", policy);
  if (p->shift != policy) p->canary = 0;
  p->shift = policy;
  return LT_OK;
}

int lt_plan_set_deferred_stream(lt_plan *p, int32_t on) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  p->defer_stream = on ? 1 : 0;
  return LT_OK;
}

int lt_plan_set_population_stride(lt_plan *p, int64_t stride) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (stride != 0 && (stride < p->N || (stride * p->esize) % 256 != 0))
    return fail(LT_ERR_INVALID, "population stride %lld: 0 (dense) or >= %lld nodes and a multiple of 256 bytes",
                (long long)stride, p->N);
  p->pop_stride = stride == p->N ? 0 : stride;
  return LT_OK;
}
int lt_plan_population_stride(lt_plan *p, int64_t *stride) {
  if (!p || !stride) return fail(LT_ERR_INVALID, "null argument");
  *stride = pop_stride_of(p);
  return LT_OK;
}

// ---- engine-owned padded population buffers ----------------------------------------------------------------
int lt_plan_set_resident(lt_plan *p, int32_t mode, int64_t pad_elements) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (mode < -1 || mode > 1) return fail(LT_ERR_INVALID, "resident mode %d", mode);
  if (pad_elements < -1) return fail(LT_ERR_INVALID, "pad %lld", (long long)pad_elements);
  if (mode == 1 && (p->desc.ghost_planes || p->unit.d < 2))
    return fail(LT_ERR_UNSUPPORTED, "resident populations: periodic 2-D / 3-D plans (slab ranks own their buffers: "
                                    "lt_plan_set_population_stride)");
  p->resident = mode;
  if (pad_elements != p->res_pad) p->res_valid = 0;
  p->res_pad = pad_elements;
  return LT_OK;
}
int lt_resident_enabled(lt_plan *p, int32_t *enabled, int64_t *stride) {
  if (!p || !enabled) return fail(LT_ERR_INVALID, "null argument");
  *enabled = resident_wanted(p) ? 1 : 0;
  if (stride) *stride = resident_stride(p);
  return LT_OK;
}
int lt_resident_load(lt_plan *p, const void *f, double tau, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (p->desc.ghost_planes) return fail(LT_ERR_UNSUPPORTED, "resident populations need a periodic plan");
  int rc = resident_alloc(p);
  if (rc) return rc;
  p->res_valid = 0;
  p->stride_in_now = pop_stride_of(p); p->stride_out_now = p->res_stride;
  rc = step(p, lt::kCollideOnly, f, p->res[0], tau, 0, p->n2, stream);
  p->stride_in_now = p->stride_out_now = -1;
  if (rc) return rc;
  p->res_cur = 0;
  p->res_valid = 1;
  return LT_OK;
}
int lt_resident_advance(lt_plan *p, double tau, int64_t n, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (n < 0) return fail(LT_ERR_INVALID, "n_steps = %lld", (long long)n);
  if (!p->res_valid) return fail(LT_ERR_INVALID, "no resident populations: call lt_resident_load first");
  p->last_single = p->last_twice = p->last_many = 0;
  if (n == 0) return LT_OK;
  void *cur = p->res[p->res_cur], *other = p->res[1 - p->res_cur];
  p->stride_in_now = p->stride_out_now = p->res_stride;
  const int rc = fused_section(p, cur, other, tau, n, stream);
  p->stride_in_now = p->stride_out_now = -1;
  if (rc) { p->res_valid = 0; return rc; }
  p->res_cur = cur == p->res[0] ? 0 : 1;
  return LT_OK;
}
int lt_resident_store(lt_plan *p, void *out, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (!p->res_valid) return fail(LT_ERR_INVALID, "no resident populations: call lt_resident_load first");
  p->stride_in_now = p->res_stride; p->stride_out_now = pop_stride_of(p);
  const int rc = step(p, lt::kStreamOnly, p->res[p->res_cur], out, 1.0, 0, p->n2, stream);
  p->stride_in_now = p->stride_out_now = -1;
  return rc;
}
int lt_resident_free(lt_plan *p) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  for (void *&r : p->res) { if (r) (void)hipFree(r); r = nullptr; }
  p->res_valid = 0;
  p->res_stride = 0;
  return LT_OK;
}

int lt_plan_set_graph_mode(lt_plan *p, int32_t mode) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (mode < -1 || mode > 1) return fail(LT_ERR_INVALID, "graph mode %d", mode);
  p->graph_mode = mode;
  return LT_OK;
}

#if LT_EXPERIMENTS
int lt_stream_collide_thrice(lt_plan *p, const void *f, void *out, double tau, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (p->unit.d != 3 || p->desc.ghost_planes || p->masked || p->desc.layout != LT_LAYOUT_REFERENCE)
    return fail(LT_ERR_UNSUPPORTED, "three steps per launch: periodic 3-D plans without boundaries, reference layout");
  return step(p, lt::kFusedThrice, f, out, tau, 0, p->n2, stream);
}
#endif
int lt_stream_collide_twice(lt_plan *p, const void *f, void *out, double tau, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  const int g = p->desc.ghost_planes;
  return step(p, lt::kFusedTwice, f, out, tau, g, p->n2 - g, stream);
}
int lt_stream_collide_twice_planes(lt_plan *p, const void *f, void *out, double tau, int64_t begin,
                                   int64_t end, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  return step(p, lt::kFusedTwice, f, out, tau, begin, end, stream);
}
int lt_stream_collide_twice_planes_packed(lt_plan *p, const void *f, void *out, double tau, int64_t begin,
                                          int64_t end, void *pack_lower, void *pack_upper, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (p->desc.ghost_planes != 2) return fail(LT_ERR_INVALID, "fused two-step packing needs ghost_planes = 2");
  if (!pack_lower && !pack_upper) return fail(LT_ERR_INVALID, "null pack buffers");
  if (end - begin < 2) return fail(LT_ERR_INVALID, "the halo message reads two planes: range [%lld, %lld)",
                                   (long long)begin, (long long)end);
  if (pack_lower && begin != 2)
    return fail(LT_ERR_INVALID, "lower message: the range must start at the first interior plane");
  if (pack_upper && end != p->n2 - 2)
    return fail(LT_ERR_INVALID, "upper message: the range must end at the last interior plane");
  return step(p, lt::kFusedTwice, f, out, tau, begin, end, stream, 1, pack_lower, pack_upper);
}
int lt_stream_collide_twice_edges(lt_plan *p, const void *f, void *out, double tau, int32_t edge_planes,
                                  void *pack_lower, void *pack_upper, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (p->desc.ghost_planes != 2) return fail(LT_ERR_INVALID, "edge launch needs ghost_planes = 2");
  const long long lo = 2, hi = p->n2 - 2;
  if (edge_planes < 2 || 2ll * edge_planes > hi - lo)
    return fail(LT_ERR_INVALID, "edge_planes = %d (2 .. %lld)", edge_planes, (hi - lo) / 2);
  if ((pack_lower == nullptr) != (pack_upper == nullptr))
    return fail(LT_ERR_INVALID, "give both message buffers or neither");
  p->second_begin = hi - edge_planes; p->second_end = hi;
  const int saved = p->seg_len;
  p->seg_len = edge_planes;                       // one segment per range and tile
  const int rc = step(p, lt::kFusedTwice, f, out, tau, lo, lo + edge_planes, stream, 1, pack_lower, pack_upper);
  p->seg_len = saved;
  p->second_begin = p->second_end = 0;
  return rc;
}
int lt_stream_collide_twice_edges_direct(lt_plan *p, const void *f, void *out, double tau, int32_t edge_planes,
                                         const void *recv_lower, const void *recv_upper, void *pack_lower,
                                         void *pack_upper, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (!pack_lower || !pack_upper || (recv_lower == nullptr) != (recv_upper == nullptr))
    return fail(LT_ERR_INVALID, "the direct edge launch needs both send buffers, and both received messages or "
                                "neither (then the ghost planes of f_dev are read)");
  if (p->masked) return fail(LT_ERR_UNSUPPORTED, "the direct edge launch exists for plans without masks");
  p->ghost_lo_now = recv_lower; p->ghost_hi_now = recv_upper;
  const int rc = lt_stream_collide_twice_edges(p, f, out, tau, edge_planes, pack_lower, pack_upper, stream);
  p->ghost_lo_now = p->ghost_hi_now = nullptr;
  return rc;
}
// The whole slab in one launch that feeds the exchange while it runs: see include/lettuce_hip.h
int lt_stream_collide_twice_slab(lt_plan *p, const void *f, void *out, double tau, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (p->desc.ghost_planes != 2) return fail(LT_ERR_INVALID, "the two-step slab launch needs ghost_planes = 2");
  if (p->masked) return fail(LT_ERR_UNSUPPORTED, "the signalling slab launch exists for plans without masks");
  const long long lo = 2, hi = p->n2 - 2;
  if (hi - lo < 4) return fail(LT_ERR_INVALID, "the slab needs at least 4 interior planes");
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (!p->signal) {
    LT_HIP(hipMalloc((void **)&p->signal, sizeof(unsigned long long)));
    LT_HIP(hipMalloc((void **)&p->signal_timed_out, sizeof(unsigned)));
    LT_HIP(hipMemsetAsync(p->signal, 0, sizeof(unsigned long long), hs));
    LT_HIP(hipMemsetAsync(p->signal_timed_out, 0, sizeof(unsigned), hs));
    // once: the polling wave of ANOTHER stream must never see the counter before it is zero
    LT_HIP(hipStreamSynchronize(hs));
    p->signal_target = 0;
  }
  const TwoStepTile tile = two_step_tile(p);
  if (tile.rows == 0 || p->n0 % tile.width != 0 || p->n1 % tile.rows != 0)
    return fail(LT_ERR_UNSUPPORTED, "two steps per launch: the grid does not tile");
  // upper edge = second range [hi - 2, hi): one short segment; first range [lo, hi - 2) in segments of at least
  // two planes
  p->second_begin = hi - 2; p->second_end = hi;
  const int saved = p->seg_len;
  if (p->seg_len == 1) p->seg_len = 2;
  if (p->seg_len == 0) {
    const int len = resolve_seg_len(p, (int)(hi - 2 - lo));
    p->seg_len = len < 2 ? 2 : len;
  }
  p->signal_now = p->signal;
  const int rc = step(p, lt::kFusedTwice, f, out, tau, lo, hi - 2, stream);
  p->signal_now = nullptr;
  p->seg_len = saved;
  p->second_begin = p->second_end = 0;
  if (rc == LT_OK) p->signal_target += 2ull * (unsigned long long)((p->n0 / tile.width) * (p->n1 / tile.rows));
  return rc;
}
int lt_slab_wait_edges(lt_plan *p, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (!p->signal) return fail(LT_ERR_INVALID, "no lt_stream_collide_twice_slab launch to wait for");
  hipLaunchKernelGGL(lt::wait_counter_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), p->signal,
                     p->signal_target, p->signal_timed_out);
  LT_HIP(hipGetLastError());
  return LT_OK;
}
int lt_slab_wait_timed_out(lt_plan *p, int32_t *timed_out, void *stream) {
  if (!p || !timed_out) return fail(LT_ERR_INVALID, "null argument");
  unsigned flag = 0;
  if (p->signal_timed_out) {
    hipStream_t hs = static_cast<hipStream_t>(stream);
    LT_HIP(hipMemcpyAsync(&flag, p->signal_timed_out, sizeof flag, hipMemcpyDeviceToHost, hs));
    LT_HIP(hipStreamSynchronize(hs));
    // reported once: the next batch starts with a clean word (the counter itself stays consistent -- every edge
    // workgroup of the launch that was waited for still adds its 1)
    if (flag) LT_HIP(hipMemsetAsync(p->signal_timed_out, 0, sizeof(unsigned), hs));
  }
  *timed_out = (int32_t)flag;
  return LT_OK;
}
int lt_plan_two_step_admitted(lt_plan *p) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  const char *why = "";
  if (!two_step_possible(p, &why)) return fail(LT_ERR_UNSUPPORTED, "two steps per launch: %s", why);
  return LT_OK;
}
int lt_two_step_limits(const lt_plan_desc *d, int32_t masked, int32_t *tile_width, int32_t *tile_rows,
                       int32_t *addressable) {
  if (!d) return fail(LT_ERR_INVALID, "null descriptor");
  if (d->stencil < 0 || d->stencil > 4) return fail(LT_ERR_UNSUPPORTED, "stencil %d", d->stencil);
  if (d->dtype < 0 || d->dtype > 1) return fail(LT_ERR_UNSUPPORTED, "dtype %d (fp32/fp64 only)", d->dtype);
  const Unit &unit = kUnits[d->stencil][d->dtype];
  if (d->dims != unit.d) return fail(LT_ERR_INVALID, "stencil is %d-dimensional, dims = %d", unit.d, d->dims);
  const int esize = d->dtype == LT_F32 ? 4 : 8;
  long long e0, e1, e2;
  if (unit.d == 1) { e0 = d->shape[0]; e1 = 1; e2 = 1; }
  else if (unit.d == 2) { e0 = d->shape[1]; e1 = d->shape[0]; e2 = 1; }
  else if (d->layout == LT_LAYOUT_REFERENCE) { e0 = d->shape[2]; e1 = d->shape[1]; e2 = d->shape[0]; }
  else { e0 = d->shape[0]; e1 = d->shape[1]; e2 = d->shape[2] + 2 * d->ghost_planes; }
  const TwoStepTile tile = unit.d >= 2 ? two_step_tile_of(unit.d, unit.q, esize, (int)e0, masked != 0) : TwoStepTile{0, 0};
  if (tile_width) *tile_width = tile.width;
  if (tile_rows) *tile_rows = tile.rows;
  if (addressable)
    *addressable = two_step_addressable(unit.d, unit.q, esize, e0, e1, e2, masked != 0,
                                        masked_big_exists(d->layout, d->collision, unit.q)) ? 1 : 0;
  return LT_OK;
}
int lt_slab_two_step_message_blocks(lt_plan *p, int32_t *blocks) {
  if (!p || !blocks) return fail(LT_ERR_INVALID, "null argument");
  *blocks = crossing(p, 0).n + (p->masked ? 3 : 2) * crossing(p, 1).n;
  return LT_OK;
}
int lt_slab_pack_two_step(lt_plan *p, const void *f, int32_t side, void *buf, void *s) {
  return halo2(p, true, const_cast<void *>(f), side, buf, s);
}
int lt_slab_unpack_two_step(lt_plan *p, void *f, int32_t side, const void *buf, void *s) {
  return halo2(p, false, f, side, const_cast<void *>(buf), s);
}

int lt_plan_set_fused_events(lt_plan *p, void *start_event, void *stop_event) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if ((start_event == nullptr) != (stop_event == nullptr))
    return fail(LT_ERR_INVALID, "give both events or neither");
  p->ev_start = static_cast<hipEvent_t>(start_event);
  p->ev_stop = static_cast<hipEvent_t>(stop_event);
  return LT_OK;
}

int lt_plan_last_run_info(lt_plan *p, int64_t *single_step_launches, int64_t *two_step_launches,
                          int64_t *many_step_launches) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (single_step_launches) *single_step_launches = p->last_single;
  if (two_step_launches) *two_step_launches = p->last_twice;
  if (many_step_launches) *many_step_launches = p->last_many;
  return LT_OK;
}

int lt_stream_collide_many(lt_plan *p, const void *f, void *out, double tau, int32_t n_steps, void *stream) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (n_steps < 1 || n_steps > many_max(p))
    return fail(LT_ERR_INVALID, "n_steps = %d (1..%d per launch for this plan)", n_steps, many_max(p));
  p->many_now = n_steps;
  return step(p, lt::kFusedMany, f, out, tau, 0, p->n2, stream);
}

int lt_plan_set_many_step(lt_plan *p, int32_t mode) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (mode < -1 || mode > 1) return fail(LT_ERR_INVALID, "many-step mode %d", mode);
  p->many = mode;
  return LT_OK;
}

int lt_plan_set_two_step(lt_plan *p, int32_t mode, int32_t planes_per_workgroup) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (mode < -1 || mode > 1) return fail(LT_ERR_INVALID, "two-step mode %d", mode);
  const int sweep = p->unit.d == 2 ? p->n1 : p->n2;          // extent of the sweep axis
  if (planes_per_workgroup < 0 ||
      (planes_per_workgroup > 0 && !p->desc.ghost_planes && sweep % planes_per_workgroup != 0))
    return fail(LT_ERR_INVALID, "planes per workgroup %d does not divide %d", planes_per_workgroup, sweep);
  if (p->seg_len != planes_per_workgroup) p->canary = 0;   // another segment length takes other paths of the sweep
  p->two_step = mode;
  p->seg_len = planes_per_workgroup;
  return LT_OK;
}

// ---- halo transport without compute units (include/lettuce_hip.h) -------------------------------------------------
int lt_halo_copy(void *dst, const void *src, int64_t n_bytes, int32_t engine, void *stream, int32_t *engine_used) {
  if (!dst || !src || n_bytes <= 0) return fail(LT_ERR_INVALID, "halo copy: null buffer or %lld bytes", (long long)n_bytes);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (engine_used) *engine_used = 0;
  if (engine == 1) {
    // device-to-device WITHOUT compute units: the runtime hands the copy to an SDMA engine (over xGMI when dst is a
    // peer's mapped buffer) instead of launching a blit kernel
    if (hipMemcpyAsync(dst, src, (size_t)n_bytes, hipMemcpyDeviceToDeviceNoCU, hs) == hipSuccess) {
      if (engine_used) *engine_used = 1;
      return LT_OK;
    }
    (void)hipGetLastError();                 // not offered for this pair of buffers: the ordinary copy below
  }
  LT_HIP(hipMemcpyAsync(dst, src, (size_t)n_bytes, hipMemcpyDeviceToDevice, hs));
  return LT_OK;
}
int lt_flag_write(uint64_t *flag, uint64_t value, int32_t how, void *stream) {
  if (!flag) return fail(LT_ERR_INVALID, "null flag");
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (how == 1) {
    // a stream memory operation: the command processor writes the word when the stream gets there (no kernel)
    if (hipStreamWriteValue64(hs, flag, value, 0) == hipSuccess) return LT_OK;
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(write_flag_kernel, dim3(1), dim3(1), 0, hs, reinterpret_cast<unsigned long long *>(flag),
                     (unsigned long long)value);
  LT_HIP(hipGetLastError());
  return LT_OK;
}
int lt_flag_wait(const uint64_t *flag, uint64_t at_least, uint32_t *timed_out, void *stream) {
  if (!flag || !timed_out) return fail(LT_ERR_INVALID, "null flag / time-out word");
  hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long *>(flag), (unsigned long long)at_least, timed_out);
  LT_HIP(hipGetLastError());
  return LT_OK;
}
// Device memory other processes of the node can map (hipIpc*): the receive windows of the copy transport
int lt_ipc_alloc(int64_t n_bytes, int32_t fine_grained, void **dev_out, void *handle_out_64_bytes) {
  if (!dev_out || !handle_out_64_bytes || n_bytes <= 0) return fail(LT_ERR_INVALID, "ipc alloc: null argument or %lld bytes", (long long)n_bytes);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
  void *p = nullptr;
  // fine_grained: memory that stays coherent with writers outside the running kernel (another device's copy engine or
  // command processor) -- for the arrival counters a polling wave reads while it runs; the messages themselves are
  // read by the NEXT launch and live in ordinary (cached) device memory
  const hipError_t ea = fine_grained ? hipExtMallocWithFlags(&p, (size_t)n_bytes, hipDeviceMallocFinegrained)
                                     : hipMalloc(&p, (size_t)n_bytes);
  if (ea != hipSuccess) { (void)hipGetLastError(); return fail(LT_ERR_ALLOC, "allocation of %lld bytes (ipc window%s) failed", (long long)n_bytes, fine_grained ? ", fine-grained" : ""); }
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); return fail(LT_ERR_HIP, "hipIpcGetMemHandle failed: %s", hipGetErrorString(e)); }
  LT_HIP(hipMemset(p, 0, (size_t)n_bytes));
  memcpy(handle_out_64_bytes, &h, sizeof h);
  *dev_out = p;
  return LT_OK;
}
int lt_ipc_open(const void *handle_64_bytes, void **dev_out) {
  if (!handle_64_bytes || !dev_out) return fail(LT_ERR_INVALID, "ipc open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle_64_bytes, sizeof h);
  LT_HIP(hipIpcOpenMemHandle(dev_out, h, hipIpcMemLazyEnablePeerAccess));
  return LT_OK;
}
int lt_ipc_close(void *mapped) {
  if (!mapped) return LT_OK;
  LT_HIP(hipIpcCloseMemHandle(mapped));
  return LT_OK;
}
int lt_ipc_free(void *dev) {
  if (!dev) return LT_OK;
  LT_HIP(hipFree(dev));
  return LT_OK;
}

#if LT_EXPERIMENTS
int lt_plan_set_arithmetic(lt_plan *p, int32_t mode) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (mode != 0 && mode != 1) return fail(LT_ERR_INVALID, "arithmetic %d (0 = the reference's, 1 = fast)", mode);
  if (mode == 1 && (p->desc.collision != LT_COLLISION_BGK || p->unit.d != 3 || p->desc.n_boundaries > 0 || p->masked ||
                    p->desc.layout != LT_LAYOUT_REFERENCE || p->desc.ghost_planes))
    return fail(LT_ERR_UNSUPPORTED, "fast arithmetic exists for BGK on periodic 3-D plans without boundaries in the "
                                    "reference layout");
  p->arith = mode;
  return LT_OK;
}

#endif

int lt_plan_set_canary(lt_plan *p, int32_t mode) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (mode < 0 || mode > 2) return fail(LT_ERR_INVALID, "canary mode %d (0 skip, 1 check on first use, 2 report a mismatch)", mode);
  if (mode != p->canary_mode) p->canary = 0;
  p->canary_mode = mode;
  return LT_OK;
}
int lt_plan_canary_status(lt_plan *p, int32_t *status, int64_t *mismatches, const char **message) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (status) *status = p->canary;
  if (mismatches) *mismatches = p->canary_mismatches;
  if (message) *message = p->canary_msg;
  return LT_OK;
}

int lt_plan_set_residency(lt_plan *p, int32_t workgroups_per_cu) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (workgroups_per_cu < -1 || workgroups_per_cu == 1 || workgroups_per_cu > 8)
    return fail(LT_ERR_INVALID, "workgroups per CU %d (use -1, 0 or 2..8)", workgroups_per_cu);
  p->residency = workgroups_per_cu;
  return LT_OK;
}

int lt_plan_set_tuning(lt_plan *p, int32_t cache_policy, int32_t wide) {
  if (!p) return fail(LT_ERR_INVALID, "null plan");
  if (cache_policy < -1 || cache_policy > 3) return fail(LT_ERR_INVALID, "cache policy %d", cache_policy);
  if (wide && !LT_EXPERIMENTS)
    return fail(LT_ERR_UNSUPPORTED, "the 16-byte variants of the one-step kernel belong to the experiments build "
                                    "(make EXPERIMENTS=1; they measured 8-13 %% slower)");
  p->tune = cache_policy;
  p->want_wide = wide ? 1 : 0;
  return LT_OK;
}

}  // extern "C"
