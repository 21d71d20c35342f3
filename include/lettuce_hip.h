/* lettuce_hip.h -- C ABI of the MI355X (gfx950) lattice-Boltzmann stream-and-collide engine.
 *
 * This is the drop-in boundary of the build: a plain-C shared library
 * (liblettuce_hip.so) with raw device pointers and sizes, no torch types.  It
 * takes the place of the module the reference JIT-generates and binds with
 * pybind11 for the same slot:
 *
 *   reference entry point  void collide_and_stream_<hash>(at::Tensor f, at::Tensor f_next
 *                              [, at::Tensor no_collision_mask, at::Tensor no_streaming_mask]
 *                              [, double tau_inv] [, at::Tensor velocityN, at::Tensor densityN ...])
 *                          lettuce/cuda_native/_template.py:58-86 (binding), :297-367 (launcher),
 *                          :175-295 (kernel), :31-44 (python invoke, f/f_next swap)
 *   reference swap point   Simulation._collide_and_stream, lettuce/_simulation.py:92-96,148,202
 *
 * All pointers named *_dev are device pointers (tensor.data_ptr()); buffers are
 * owned by the caller (lettuce/_flow.py:90,124-134) and only borrowed for the
 * call.  Engine-owned scratch (compressed masks, boundary tables, reduction
 * partials) lives in the plan and is released by lt_plan_destroy.  Every launch
 * goes to the caller's HIP stream and returns without synchronising.
 *
 * Memory layout of a population field ("f"): [q][a2][a1][a0], a0 fastest.
 *   LT_LAYOUT_REFERENCE  a0 = last logical axis: f[q][x][y][z] (3-D), f[q][x][y] (2-D)
 *                        -- exactly the reference's C-contiguous tensor (lettuce/_flow.py:90).
 *   LT_LAYOUT_SLAB       3-D only, a0 = x, a2 = z: f[q][z][y][x]; used by the multi-GPU
 *                        z-slab driver so that ghost planes are contiguous.
 * ghost_planes = g (1, or 2 for the two-step slab driver) adds g planes below and g above along
 * a2 (slab exchange); then a2 has shape_a2 + 2 g planes and no periodic wrap is applied along it.
 *
 * Error convention: every function returns LT_OK (0) or a positive LT_ERR_* code
 * and never aborts; lt_last_error() gives the message of the calling thread's
 * last failure (the reference asserts / TORCH_CHECKs: _template.py:53-56,317-323).
 */
#ifndef LETTUCE_HIP_H
#define LETTUCE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ABI_VERSION 2
/* boundaries per plan: the compiled node byte has seven index bits (the reference's uint8 no_collision_mask could
 * name 255, lettuce/_simulation.py:63-86); plans with more than 15 keep the one-step kernels (the two-step kernels
 * with boundaries carry the kinds as two bits per slot in one register) */
#define LT_MAX_BOUNDARIES 127
#define LT_MAX_Q 27

enum lt_status {
  LT_OK = 0,
  LT_ERR_INVALID = 1,      /* bad argument (null pointer, shape, alignment, in == out ...) */
  LT_ERR_UNSUPPORTED = 2,  /* valid request the engine has no kernel for */
  LT_ERR_HIP = 3,          /* a HIP runtime call failed */
  LT_ERR_ALLOC = 4
};

/* lettuce/ext/_stencil/d2q9.py:8-10, d3q19.py:8-13, d3q27.py:8-12, d1q3.py:8-10, d3q15.py:8-13
 * (same velocity order) */
enum lt_stencil { LT_D2Q9 = 0, LT_D3Q19 = 1, LT_D3Q27 = 2, LT_D1Q3 = 3, LT_D3Q15 = 4 };
/* AT_DISPATCH_FLOATING_TYPES, lettuce/cuda_native/_template.py:357 */
enum lt_dtype { LT_F32 = 0, LT_F64 = 1 };
/* lettuce/ext/_collision/no_collision.py:9-17, bgk_collision.py:12-35, kbc_collision.py:11-166 */
enum lt_collision { LT_COLLISION_NONE = 0, LT_COLLISION_BGK = 1, LT_COLLISION_KBC = 2 };
/* lettuce/ext/_boundary/bounce_back_boundary.py:10-32, equilibrium_boundary_pu.py:13-46,
 * anti_bounce_back_outlet.py:13-109 */
enum lt_boundary_kind {
  LT_BOUNDARY_BOUNCE_BACK = 1,
  LT_BOUNDARY_EQUILIBRIUM = 2,
  LT_BOUNDARY_ABB_OUTLET = 3
};
enum lt_layout { LT_LAYOUT_REFERENCE = 0, LT_LAYOUT_SLAB = 1 };
#define LT_BOUNDARY_ABSENT 1

/* One entry of Simulation.boundaries[1:] (lettuce/_simulation.py:57-58); entry i (0-based)
 * is the boundary whose index in no_collision_mask is i + 1. */
typedef struct lt_boundary_desc {
  int32_t kind;            /* lt_boundary_kind */
  int32_t axis;            /* ABB outlet: logical axis of `direction` (0 = x) */
  int32_t side;            /* ABB outlet: +1 or -1 */
  int32_t flags;           /* ABB outlet on a slab plan (ghost_planes > 0) whose normal is the decomposed axis:
                            * bit 0 (LT_BOUNDARY_ABSENT) = another rank holds the first / last plane of the
                            * global grid, this rank has no outlet; else 0 */
  /* EQUILIBRIUM with uniform velocity/pressure: the populations written on the masked
   * nodes, i.e. feq(rho(p_pu), u_lu(v_pu)) as the host evaluated it in the flow's dtype. */
  double feq[LT_MAX_Q];
  /* EQUILIBRIUM with per-node velocity/pressure: device pointer to a [q, *res] field of the
   * plan's dtype (borrowed; must outlive the plan or be re-set), or NULL. */
  const void *feq_field_dev;
} lt_boundary_desc;

typedef struct lt_plan_desc {
  int32_t abi_version;     /* LT_ABI_VERSION */
  int32_t stencil;         /* lt_stencil */
  int32_t dtype;           /* lt_dtype */
  int32_t collision;       /* lt_collision */
  int32_t layout;          /* lt_layout */
  int32_t ghost_planes;    /* 0; or 1 / 2 (LT_LAYOUT_SLAB only; 2 for the two-step slab entry points) */
  int32_t dims;            /* 1, 2 or 3; must match the stencil */
  int32_t n_boundaries;    /* 0 .. LT_MAX_BOUNDARIES */
  int64_t shape[3];        /* logical resolution (nx, ny, nz); unused trailing entries = 1.  With ghost
                              planes this is the rank-local slab WITHOUT the ghosts. */
  lt_boundary_desc boundaries[LT_MAX_BOUNDARIES];
} lt_plan_desc;

typedef struct lt_plan lt_plan;

int lt_abi_version(void);
/* bit 0: the library was built with the experiment kernels (make EXPERIMENTS=1; the entry points under
 * LT_EXPERIMENTS below) */
int lt_build_flags(void);
const char *lt_last_error(void);

int lt_plan_create(const lt_plan_desc *desc, lt_plan **out_plan);
int lt_plan_destroy(lt_plan *plan);

/* Compile Simulation.no_collision_mask (uint8 [*res], value = boundary index,
 * lettuce/_simulation.py:73-82) and Simulation.no_streaming_mask (uint8 [q, *res], 0/1,
 * :83-86; may be NULL) into the plan's node-descriptor byte (+ a sparse per-node bit set for
 * the streaming mask).  Both NULL removes the masks.  Required before stepping a plan that
 * has boundaries.  The masks are indexed like the populations of the plan's layout: in the slab
 * layout uint8 [nz + 2][ny][nx] and [q][nz + 2][ny][nx] (ghost-plane entries are ignored). */
int lt_plan_set_masks(lt_plan *plan, const uint8_t *no_collision_mask_dev,
                      const uint8_t *no_streaming_mask_dev, void *stream);

/* Replace boundary i's parameters (e.g. inlet velocity changed between calls; the reference
 * re-reads them every step, cuda_native/ext/_boundary/equilibrium_pu.py:40-48). */
int lt_plan_update_boundary(lt_plan *plan, int32_t index, const lt_boundary_desc *desc,
                            void *stream);

/* out = B(C(in)): collision on nodes with no_collision_mask == 0, then the boundaries in
 * index order; no streaming (Simulation._collide, lettuce/_simulation.py:177-189).
 * in != out. */
int lt_collide(lt_plan *plan, const void *f_dev, void *f_out_dev, double tau, void *stream);

/* out = S(in): periodic streaming, destination-side no-streaming mask
 * (Simulation._stream, lettuce/_simulation.py:160-175).  in != out. */
int lt_stream(lt_plan *plan, const void *f_dev, void *f_out_dev, void *stream);

/* out = B(C(S(in))): the fused pull-scheme kernel.  `in` and `out` hold post-collision
 * populations ("f*").  in != out. */
int lt_stream_collide(lt_plan *plan, const void *fstar_dev, void *fstar_out_dev, double tau,
                      void *stream);

/* Same three operators restricted to memory planes [plane_begin, plane_end) of the slowest
 * axis a2 (ghost-plane numbering: 0 is the lower ghost).  Used by the slab driver to do the
 * two boundary planes first and the interior while the halo exchange is in flight. */
int lt_collide_planes(lt_plan *plan, const void *f_dev, void *f_out_dev, double tau,
                      int64_t plane_begin, int64_t plane_end, void *stream);
int lt_stream_planes(lt_plan *plan, const void *f_dev, void *f_out_dev,
                     int64_t plane_begin, int64_t plane_end, void *stream);
int lt_stream_collide_planes(lt_plan *plan, const void *fstar_dev, void *fstar_out_dev,
                             double tau, int64_t plane_begin, int64_t plane_end, void *stream);

/* Fused stream-collide of exactly the two planes `first` < `second` in one launch (the slab's
 * two boundary planes). */
int lt_stream_collide_plane_pair(lt_plan *plan, const void *fstar_dev, void *fstar_out_dev,
                                 double tau, int64_t first, int64_t second, void *stream);

/* As lt_stream_collide_plane_pair, and in the same launch the crossing populations of plane
 * `first` (velocity component -1 along the slowest axis) are packed into pack_first_dev and those
 * of plane `second` (+1) into pack_second_dev, in the layout of lt_slab_pack (first == second is
 * allowed: a one-plane slab). */
int lt_stream_collide_plane_pair_packed(lt_plan *plan, const void *fstar_dev, void *fstar_out_dev,
                                        double tau, int64_t first, int64_t second,
                                        void *pack_first_dev, void *pack_second_dev, void *stream);

/* Halo packing for the slab driver (no reference counterpart: the reference is single-GPU).
 * direction = +1 / -1 selects the populations whose velocity component along the slowest memory
 * axis is +1 / -1 (5 of 19 for D3Q19, 9 of 27 for D3Q27), in ascending q.
 * lt_slab_crossing reports them; lt_slab_pack copies them from memory plane `plane` of f into the
 * contiguous buffer buf[n][n1*n0] (one message per direction instead of n); lt_slab_unpack is the
 * inverse. */
int lt_slab_crossing(lt_plan *plan, int32_t direction, int32_t *q_out, int32_t *n_out);
int lt_slab_pack(lt_plan *plan, const void *f_dev, int64_t plane, int32_t direction, void *buf_dev,
                 void *stream);
int lt_slab_unpack(lt_plan *plan, void *f_dev, int64_t plane, int32_t direction, const void *buf_dev,
                   void *stream);

/* Two-step slabs (ghost_planes = 2 in the plan).  lt_stream_collide_twice_planes is
 * lt_stream_collide_twice for the output planes [begin, end) of a slab: it reads planes
 * [begin - 2, end + 2), so begin >= 2 and end <= n2 - 2.  Before the launch the ghost planes must hold,
 * on the lower side, the in-plane and upward populations of the neighbour's top plane (ghost plane 1)
 * and the upward populations of the plane below that (ghost plane 0); mirrored on the upper side.
 * lt_slab_pack_two_step gathers exactly that message for the neighbour beyond `side` (-1 lower,
 * +1 upper) from this rank's two interior planes next to the cut, as
 *   buf[n_in_plane + 2 n_crossing][n1*n0] = [in-plane of the near plane | crossing of the near
 *   plane | crossing of the far plane]           (9 + 5 + 5 = 19 blocks for D3Q19);
 * lt_slab_unpack_two_step scatters a received message into the two ghost planes beyond `side`.
 * lt_slab_crossing(plan, 0, ...) reports the in-plane populations.
 * Plans with masks (lt_plan_set_masks called) carry one more group of n_crossing blocks: the populations of
 * the near plane that move away from the cut, which a no-streaming node of the ghost plane keeps (24 blocks
 * for D3Q19); lt_slab_two_step_message_blocks reports the count.  With boundaries the two-step launches take
 * bounce-back and equilibrium nodes and one anti-bounce-back outlet whose normal is x (the contiguous axis
 * of the slab layout) opposite an inlet face of equilibrium nodes -- lt_plan_two_step_admitted returns
 * LT_ERR_UNSUPPORTED with the reason otherwise; the fused packing of the entry points below exists for
 * plans without masks only. */
int lt_stream_collide_twice_planes(lt_plan *plan, const void *f_dev, void *out_dev, double tau,
                                   int64_t begin, int64_t end, void *stream);
/* As lt_stream_collide_twice_planes, and in the same launch the halo message for the lower neighbour
 * (pack_lower_dev != NULL; begin must be the first interior plane) and / or for the upper neighbour
 * (pack_upper_dev != NULL; end must be one past the last interior plane) is written in the layout of
 * lt_slab_pack_two_step -- the buffers may be peer-mapped memory of the neighbour. */
int lt_stream_collide_twice_planes_packed(lt_plan *plan, const void *f_dev, void *out_dev, double tau,
                                          int64_t begin, int64_t end, void *pack_lower_dev,
                                          void *pack_upper_dev, void *stream);
/* The two edges of a slab in ONE launch: output planes [2, 2 + edge_planes) and
 * [n2 - 2 - edge_planes, n2 - 2), optionally writing both halo messages (both buffers or neither). */
int lt_stream_collide_twice_edges(lt_plan *plan, const void *f_dev, void *out_dev, double tau,
                                  int32_t edge_planes, void *pack_lower_dev, void *pack_upper_dev,
                                  void *stream);
/* The same edge launch fed straight from the receive buffers: wherever its pull reaches beyond a cut it reads the
 * halo message that arrived from the rank below (recv_lower_dev) / above (recv_upper_dev) -- layout of
 * lt_slab_pack_two_step -- instead of the ghost planes of f_dev, which need not be filled (no
 * lt_slab_unpack_two_step); and it writes both outgoing messages.  With lt_stream_collide_twice_planes for the
 * planes [2 + edge_planes, n2 - 2 - edge_planes), which never reads a ghost plane, a double step of a slab is two
 * launches on one stream and one exchange that starts after the first of them: no pack launch, no unpack launch,
 * no launches competing for the compute units.  Both recv pointers NULL: the ghost planes of f_dev are read (the
 * first double step after a collide pass, whose exchange ended in them).  Plans without masks. */
int lt_stream_collide_twice_edges_direct(lt_plan *plan, const void *f_dev, void *out_dev, double tau,
                                         int32_t edge_planes, const void *recv_lower_dev,
                                         const void *recv_upper_dev, void *pack_lower_dev, void *pack_upper_dev,
                                         void *stream);
/* The whole slab -- output planes [2, n2 - 2) -- in ONE launch that releases the exchange while it runs (plans
 * without masks).  The workgroups that produce the two planes next to each cut start first; each adds 1 to a
 * counter of the plan when those planes are in memory.  lt_slab_wait_edges enqueues, on ANOTHER stream, one
 * polling wave that returns when the counter has reached the value that completes the edges of the last such
 * launch -- work enqueued behind it (lt_slab_pack_two_step, the transfers, lt_slab_unpack_two_step) then runs
 * beside the rest of the sweep -- or after about one second (lt_slab_wait_timed_out then reports 1; it
 * synchronises the stream).  Replaces two edge launches + one interior launch: no planes are computed twice at
 * the lower cut, the upper edge costs its prologue only once, and three launches do not compete for the CUs. */
int lt_stream_collide_twice_slab(lt_plan *plan, const void *f_dev, void *out_dev, double tau, void *stream);
int lt_slab_wait_edges(lt_plan *plan, void *stream);
int lt_slab_wait_timed_out(lt_plan *plan, int32_t *timed_out, void *stream);
int lt_slab_two_step_message_blocks(lt_plan *plan, int32_t *blocks_out);
/* LT_OK when lt_stream_collide_twice / _planes has a kernel for this plan as it stands (lattice, dtype,
 * collision, grid extents, masks); LT_ERR_UNSUPPORTED (and the reason in lt_last_error) otherwise. */
int lt_plan_two_step_admitted(lt_plan *plan);
/* Descriptor-only query (no device, no plan): the tile of the two-step kernels for this stencil / dtype
 * (tile_width x tile_rows nodes; rows = 0: no kernel) and whether a field of these extents stays within the
 * kernels' 32-bit byte offsets (masked != 0: the whole field, q * nodes * sizeof(scalar) < 4 GiB; else one plane).
 * lt_run / lt_plan_two_step_admitted apply the same rule, so a plan beyond it keeps the one-step kernel. */
int lt_two_step_limits(const lt_plan_desc *desc, int32_t masked, int32_t *tile_width, int32_t *tile_rows,
                       int32_t *addressable);
int lt_slab_pack_two_step(lt_plan *plan, const void *f_dev, int32_t side, void *buf_dev, void *stream);
int lt_slab_unpack_two_step(lt_plan *plan, void *f_dev, int32_t side, const void *buf_dev, void *stream);

/* n whole lettuce steps (collide, boundaries, stream) starting from post-streaming
 * populations in buf_a (lettuce/_simulation.py:201-203): one collide launch, n-1 fused
 * launches, one stream launch, ping-ponging between the two buffers.  On return
 * *result_in_b tells which buffer holds the new post-streaming populations (1 = buf_b);
 * the other buffer then holds the post-collision populations of the last step, from which
 * lt_continue can carry on without the extra collide pass. */
int lt_run(lt_plan *plan, void *buf_a_dev, void *buf_b_dev, double tau, int64_t n_steps,
           void *stream, int32_t *result_in_b);
/* As lt_run, but buf_a holds post-collision populations f* of the step before (what lt_run
 * leaves in the non-result buffer): n fused launches + one stream launch. */
int lt_continue(lt_plan *plan, void *fstar_a_dev, void *buf_b_dev, double tau, int64_t n_steps,
                void *stream, int32_t *result_in_b);
/* on != 0: lt_run / lt_continue stop before their last (streaming) launch: *result_in_b then tells which
 * buffer holds the POST-COLLISION populations of the last step, the other buffer is scratch; lt_stream of
 * that buffer gives the post-streaming populations whenever somebody wants to see them, and lt_continue
 * carries on from it without them.  A caller that advances in batches and looks at the populations only
 * now and then (lettuce's Simulation with reporters every k steps) saves one pass over memory per batch. */
int lt_plan_set_deferred_stream(lt_plan *plan, int32_t on);

/* rho [*res] and u [d, *res] (either may be NULL) from post-streaming populations
 * (Flow.rho / Flow.j / Flow.u, lettuce/_flow.py:136-138,152-172). */
int lt_macroscopic(lt_plan *plan, const void *f_dev, void *rho_dev, void *u_dev, void *stream);

/* feq [q, *res] from rho [*res] and u [d, *res]
 * (QuadraticEquilibrium.__call__, lettuce/ext/_equilibrium/quadratic_equilibrium.py:11-25). */
int lt_equilibrium(lt_plan *plan, const void *rho_dev, const void *u_dev, void *feq_dev,
                   void *stream);

/* *out_dev (one double, device memory) = sum over nodes of 0.5 * u.u in lattice units,
 * accumulated in fp64 with a fixed reduction order (Flow.incompressible_energy summed as in
 * IncompressibleKineticEnergy, lettuce/_flow.py:178-181,
 * lettuce/ext/_reporter/observable_reporter.py:34-42; the pu scaling stays on the host).
 * Ghost planes are excluded. */
int lt_kinetic_energy(lt_plan *plan, const void *f_dev, double *out_dev, void *stream);

/* *out_dev = sum over nodes and populations of f (total mass, fp64 accumulation). */
int lt_mass(lt_plan *plan, const void *f_dev, double *out_dev, void *stream);

/* *out_dev = max over nodes of |u| in lattice units (MaximumVelocity observable,
 * lettuce/ext/_reporter/observable_reporter.py:27-31; the pu scaling stays on the host). */
int lt_max_velocity(lt_plan *plan, const void *f_dev, double *out_dev, void *stream);

/* Non-equilibrium initialisation (lettuce/_flow.py:309-336, Krueger et al. 2017): f_dev [q][N] =
 * feq(rho, u) - w_q Pi1:Q_q with Pi1_ab = tau rho d_b u_a / cs^2 from the 6th-order periodic central
 * differences of torch_gradient (util/utility.py:37-99, dx = 1) and Q_q,ab = e_qa e_qb - identity_cs2
 * delta_ab.  identity_cs2 is passed by the caller because the reference builds the identity in torch's
 * DEFAULT dtype (an fp32-rounded cs^2 even in an fp64 run, _flow.py:328-330).  rho_dev [N] and u_dev
 * [d][N] (lattice units, logical component order) are the moments of the equilibrium populations, as
 * Flow.rho() / Flow.u() give them; one launch, nothing else is materialised (the reference builds
 * [d][d][N] gradients, Pi1 and two [q][N] fields).  Reference layout, periodic along every axis. */
int lt_init_fneq(lt_plan *plan, const void *rho_dev, const void *u_dev, double tau, double identity_cs2,
                 void *f_dev, void *stream);

/* Enstrophy observable (lettuce/ext/_reporter/observable_reporter.py:45-68): *out_dev = sum over nodes
 * of |curl(u_scale * u)|^2 with the 6th-order periodic central differences of torch_gradient
 * (lettuce/util/utility.py:37-99; weights -1/60, 3/20, -3/4, 3/4, -3/20, 1/60 times inv_dx), evaluated
 * per node in the working dtype as the reference's whole-field expression and accumulated in fp64
 * with a fixed reduction order; the factor dx^d stays on the host.  Two launches: u [d][N] into
 * u_scratch_dev (d * N scalars of the plan's dtype, caller-owned), then the stencil reduction over
 * it.  2-D / 3-D, reference layout, periodic domains only (as the reference). */
int lt_enstrophy(lt_plan *plan, const void *f_dev, void *u_scratch_dev, double u_scale, double inv_dx,
                 double *out_dev, void *stream);

/* Mass observable (observable_reporter.py:140-158): *out_dev = sum over all populations of the nodes
 * that are not on the first / last index of the two fastest axes (the reference's f[..., 1:-1, 1:-1])
 * minus, if no_mass_mask_dev (uint8 [N], 0 / 1) is given, sum_q f over the nodes it flags (borders
 * included, as in the reference).  fp64 accumulation.  2-D / 3-D, reference layout. */
int lt_mass_interior(lt_plan *plan, const void *f_dev, const uint8_t *no_mass_mask_dev, double *out_dev,
                     void *stream);

/* The same two observables for one rank of the z-slab decomposition (plans with LT_LAYOUT_SLAB and
 * ghost planes), so that a slab reporter needs no gather (observable_reporter.py:45-68, 140-158):
 *  lt_slab_velocity   u_ext_dev [3][nz + 6][ny][nx] (nz = the rank's own planes, logical component order, lattice
 *                     units) <- u of the plan's nz + 2 g planes, written at planes [3 - g, 3 + nz + g); the caller
 *                     then fills planes [0, 3) and [nz + 3, nz + 6) with the neighbours' planes (6th-order
 *                     differences reach three planes: a velocity halo exchange, host side);
 *  lt_slab_enstrophy  *out_dev = sum over the rank's own nodes of |curl(u_scale * u)|^2, as lt_enstrophy;
 *  lt_slab_mass_interior  *out_dev = the rank's share of lt_mass_interior: own planes only, y and the GLOBAL z index
 *                     (z_begin + local plane, of nz_global) off their first / last value; no_mass_mask_dev is uint8 per
 *                     node of the plan (ghost planes included, as f) or null.
 * The caller sums the ranks' results (all-reduce). */
int lt_slab_velocity(lt_plan *plan, const void *f_dev, void *u_ext_dev, void *stream);
int lt_slab_enstrophy(lt_plan *plan, const void *u_ext_dev, double u_scale, double inv_dx, double *out_dev,
                      void *stream);
int lt_slab_mass_interior(lt_plan *plan, const void *f_dev, const uint8_t *no_mass_mask_dev, int32_t z_begin,
                          int32_t nz_global, double *out_dev, void *stream);

/* Introspection for tests and benchmarks. */
int lt_plan_kernel_info(lt_plan *plan, int32_t *vec_width, int32_t *threads_per_block,
                        int64_t *blocks_per_launch);
/* Name of the fused kernel variant this plan launches (for matching rocprof rows). */
const char *lt_plan_kernel_name(lt_plan *plan);
/* A/B selector.  16-byte variant of the one-step kernel: 0 = aligned vector load + one neighbour
 * element, 1 = unaligned vector load, 2 = aligned vector load + cross-lane shift.  Two-step kernel
 * (D3Q19 / D3Q15 fp32, BGK, reference layout): 0 = product variant, 1 = two nodes per thread in both
 * phases, 2 = two output nodes per thread, 3 = no XCD-aware renumbering of the workgroups, 4 = the round-1
 * renumbering (an eighth of the grid per XCD instead of an eighth of every segment layer), 5 = 32 x 8 tiles for
 * the slab edge launch (two workgroups per CU; measured slower: 99 against 77 us).  1, 2 and 5 are tile variants
 * that lost their A/B: they exist in the experiments build only (LT_ERR_UNSUPPORTED otherwise). */
int lt_plan_set_shift_policy(lt_plan *plan, int32_t policy);
/* out[i] = x[i] / D as the kernels' equilibrium forms it (div_cs: two or three instructions that return the IEEE
 * quotient by the constant D = 2 cs^2 (which 0) or cs^2 (which 1) rounded to dtype, the reference's divisors:
 * lettuce/ext/_equilibrium/quadratic_equilibrium.py:15-24) -- a test hook that pins the emulation on the device. */
int lt_probe_div_cs(const void *x_dev, void *out_dev, int64_t n, int32_t dtype, int32_t which, void *stream);
/* Diagnostic: dst[0:n_bytes] = src[0:n_bytes] with 16-byte accesses and the cache-policy bits of
 * lt_plan_set_tuning; max_blocks > 0 caps its grid (grid-stride loop).  bench.py uses it to
 * measure this device's copy ceiling. */
int lt_probe_copy(void *dst_dev, const void *src_dev, int64_t n_bytes, int32_t cache_policy,
                  int32_t max_blocks, void *stream);
/* hipGraph replay inside lt_run / lt_continue: the fused launches are captured 32 at a time into a
 * graph and replayed.  mode 1 = always, 0 = never, -1 = automatic, which is "never" at present:
 * on MI355X the eager launch loop was as fast or faster on small grids (3.5 vs 4.1 us per step at
 * 128^2), see DESIGN.md. */
int lt_plan_set_graph_mode(lt_plan *plan, int32_t mode);
/* Tuning knobs.  cache_policy: -1 = automatic (nontemporal accesses when the populations exceed
 * the caches), else bit 0 = nontemporal loads, bit 1 = nontemporal stores (the one-node-per-
 * thread kernels exist for 0 and 3).
 * wide != 0 switches the hot kernel (fused, BGK, no masks) to its 16-byte-per-lane A/B variant,
 * whose shift handling lt_plan_set_shift_policy selects. */
int lt_plan_set_tuning(lt_plan *plan, int32_t cache_policy, int32_t wide);
#ifdef LT_EXPERIMENTS
/* (experiments build only: 20 % slower per update than two updates per launch)
 * THREE fused stream-collide steps in one launch (lbm3_kernel): out = (C S)^3 f with both intermediate states in
 * LDS -- one HBM read and one write of the populations per three lattice updates.  Periodic 3-D plans without
 * boundaries in the reference layout whose grid tiles (contiguous extent % 64 (fp32) / 32 (fp64), middle extent % 4);
 * LT_ERR_UNSUPPORTED otherwise.  Bit-identical to three lt_stream_collide calls.  Same contract as the reference's
 * native step applied three times (lettuce/cuda_native/_template.py:58-86). */
int lt_stream_collide_thrice(lt_plan *plan, const void *f_dev, void *out_dev, double tau, void *stream);
#endif

/* Two fused steps in one launch: out = (C S)^2 f for the whole periodic grid, the intermediate
 * state staged through LDS (one HBM read and one write of the populations per two lattice updates).
 * Bit-identical to two lt_stream_collide calls.  Exists with BGK / no collision for the 3-D lattices
 * (D3Q15 and D3Q19 in fp32 and fp64, D3Q27 in fp32) on grids whose contiguous extent is a multiple
 * of 64 (fp32) / 32 (fp64) and whose middle extent is a multiple of 8 (D3Q27: of 4), for D2Q9 (fp32 and
 * fp64) on grids whose contiguous extent is a multiple of 64, and -- D2Q9; 3-D in fp32 and D3Q15 fp64 -- for
 * plans with bounce-back / equilibrium boundaries and at most one anti-bounce-back outlet: at the last plane
 * (2-D: row) of the slowest memory axis or, in 3-D, at an end of the contiguous axis opposite a face of
 * equilibrium nodes (no-streaming bits exactly that outlet's); LT_ERR_UNSUPPORTED otherwise
 * (lt_plan_two_step_admitted tells). */
int lt_stream_collide_twice(lt_plan *plan, const void *f_dev, void *out_dev, double tau, void *stream);
/* Small 2-D grids (launch-bound): n_steps <= 8 stream-collide steps in one launch.  Every workgroup
 * keeps the neighbourhood of its 8 x 8 tile in LDS and recomputes the halo, so the launch does
 * redundant arithmetic but replaces n_steps launches; bit-identical to n_steps lt_stream_collide
 * calls.  2-D lattices, extents multiples of 8; plans with masks too (bounce-back, equilibrium and at most
 * one anti-bounce-back outlet, for which one more ring of nodes is recomputed: n_steps <= 7);
 * LT_ERR_UNSUPPORTED otherwise.
 * Small 3-D grids: n_steps = 2 exactly (the 10^3 neighbourhood of an 8^3 tile in LDS): periodic plans in the
 * reference layout without masks, extents multiples of 8, BGK / no collision, every 3-D lattice and dtype whose
 * q x 1000 values fit the LDS (not D3Q27 fp64); bit-identical to two lt_stream_collide calls.
 * lt_plan_set_many_step: lt_run / lt_continue use it for their fused steps: -1 = automatic (2-D grids up
 * to 256 x 256 nodes, with masks up to 256 x 128; BGK / no collision, where it is bit-identical to the one-step
 * kernel; never for 3-D grids, where two steps per launch measured slower than two launches), 0 = never,
 * 1 = whenever supported. */
int lt_stream_collide_many(lt_plan *plan, const void *f_dev, void *out_dev, double tau, int32_t n_steps,
                           void *stream);
int lt_plan_set_many_step(lt_plan *plan, int32_t mode);
/* lt_run / lt_continue pair their fused steps with lt_stream_collide_twice where it exists:
 * mode -1 = automatic, 0 = never, 1 = whenever supported.  planes_per_workgroup: segment length of
 * the sweep along the slowest axis (0 = automatic). */
int lt_plan_set_two_step(lt_plan *plan, int32_t mode, int32_t planes_per_workgroup);
/* Population stride.  By default a population buffer is dense: population q starts q * nodes elements after
 * population 0 (the reference's [q, *res] tensor, lettuce/_flow.py:90).  With the q populations a power of two
 * apart (256^3 fp32: exactly 64 MiB) the q read and q write streams of a node meet in the same memory channels;
 * buffers the CALLER owns and the reference never sees -- the slab tensors of a multi-GPU rank -- may therefore
 * be padded: every population buffer handed to this plan's entry points (steps, pack / unpack, reductions,
 * lt_macroscopic) then has `stride_elements` (>= nodes incl. ghost planes, a multiple of 256 bytes) between
 * consecutive populations; 0 = dense again.  Per-node boundary fields and masks stay dense. */
int lt_plan_set_population_stride(lt_plan *plan, int64_t stride_elements);
int lt_plan_population_stride(lt_plan *plan, int64_t *stride_elements_out);
/* Resident populations: the engine's own padded ping-pong buffers for the fused steps of a periodic plan, so that
 * the caller's tensors stay the reference's plain [q, *res] (lettuce/_flow.py:90,124-134,226-236) and are touched
 * only when somebody looks:
 *   lt_resident_load     f* <- B(C(f)): one collide pass from the caller's post-streaming populations (the
 *                        reference's flow.f) into the resident buffer;
 *   lt_resident_advance  n fused stream-collide steps on the resident f* (lt_run's fused section: two updates per
 *                        launch where the kernel exists, several on small 2-D grids; lt_plan_set_fused_events and
 *                        lt_plan_last_run_info apply);
 *   lt_resident_store    out <- S(f*): the streaming pass that presents the populations in lettuce's convention;
 *                        the resident state stays valid, a later lt_resident_advance carries on from it;
 *   lt_resident_free     releases the two buffers (lt_plan_destroy does too).
 * k whole lettuce steps from flow.f are load + advance(k - 1) (+ store when flow.f is read); a batch that carries
 * on is advance(k).  lt_plan_set_resident: mode -1 = automatic (lt_resident_enabled reports 1 when the populations
 * exceed the caches), 0 = off, 1 = on; pad_elements = distance added between populations, -1 = the engine's choice.
 * lt_run / lt_continue keep working on caller-owned dense buffers whatever the mode. */
int lt_plan_set_resident(lt_plan *plan, int32_t mode, int64_t pad_elements);
int lt_resident_enabled(lt_plan *plan, int32_t *enabled_out, int64_t *stride_elements_out);
int lt_resident_load(lt_plan *plan, const void *f_dev, double tau, void *stream);
int lt_resident_advance(lt_plan *plan, double tau, int64_t n_steps, void *stream);
int lt_resident_store(lt_plan *plan, void *out_dev, void *stream);
int lt_resident_free(lt_plan *plan);
/* Measurement hooks.  With two hipEvent_t set, lt_run / lt_continue record them on the launch stream
 * around their fused launches (around the two-step launches when there are any, else around the
 * single-step ones, or around the many-step launches); null, null removes them.
 * lt_plan_last_run_info reports how many fused launches of each kind the last lt_run / lt_continue
 * issued. */
int lt_plan_set_fused_events(lt_plan *plan, void *start_event, void *stop_event);
int lt_plan_last_run_info(lt_plan *plan, int64_t *single_step_launches, int64_t *two_step_launches,
                          int64_t *many_step_launches);
/* Workgroups resident per CU for the chip-filling launches (an unused dynamic-LDS allocation caps
 * them): -1 = automatic (3, or 4 for fp32 KBC, once the populations stream from HBM and the launch
 * fills the chip several times over; no cap otherwise), 0 = no cap, 2..8 = that many. */
int lt_plan_set_residency(lt_plan *plan, int32_t workgroups_per_cu);
#ifdef LT_EXPERIMENTS
/* (experiments build only: 4-7 % faster than the exact arithmetic where <= 0.45 ms per launch had been the bar, and
 * 1.0-1.5e-6 off the reference's fp32 kinetic energy after 10 steps where SURVEY 8(d) states 1e-6 -- DESIGN.md section 4)
 * Arithmetic of the BGK collision.  0 (default) = the reference's, operation for operation: every rounding of
 * lettuce's whole-field torch operators is reproduced (ATen's summation order, u = j / rho by IEEE division, the
 * division by the rounded constants 2 cs^2 and cs^2, no fused multiply-adds; lettuce/_flow.py:136-172,
 * ext/_equilibrium/quadratic_equilibrium.py:11-25, ext/_collision/bgk_collision.py:17-22), so periodic BGK flows are
 * bit-identical to the reference's CPU path.  1 = fast: the same collision to rounding level in about half the
 * instructions (moments over opposite pairs, one reciprocal of rho, cs^2 = 1/3, contracted multiply-adds), inside
 * the tolerances SURVEY.md 8(d) states (fp32: max |df| <= 1e-5 max |f| after 10 steps, kinetic energy 1e-6 / 5e-5
 * over 10 / 100 steps) but NOT bit-identical to the reference; exists for BGK on periodic 3-D plans without
 * boundaries in the reference layout, never chosen by the engine itself. */
int lt_plan_set_arithmetic(lt_plan *plan, int32_t mode);
#endif
/* First-use check of the two-step kernels of a plan WITH masks.  Before such a plan uses a two-step kernel for the
 * first time (lt_run's pairs, lt_resident_advance, every lt_stream_collide_twice* entry point,
 * lt_plan_two_step_admitted), one double step over all its planes is held against two one-step launches: synthetic
 * populations on engine-owned scratch buffers, the plan's own masks, boundaries, tile and segment length, bit-for-bit
 * comparison on the device (four scratch fields, five launches, one synchronisation of an engine-owned stream; again
 * after lt_plan_set_masks or a change of the segment length).  If the two disagree -- hipcc has miscompiled exactly
 * this code once, see csrc/Makefile -- or the check cannot run, the plan keeps the one-step kernel: lt_run /
 * lt_continue / lt_resident_advance fall back by themselves and leave the reason in lt_last_error(); the explicit
 * two-step entry points and lt_plan_two_step_admitted return LT_ERR_UNSUPPORTED with it.  What is protected:
 * lettuce/_simulation.py:177-189 (collision, then every boundary in index order, on the nodes of its mask).
 * lt_plan_set_canary: 1 = check on first use (default), 0 = trust the kernel, 2 = report a mismatch without
 * launching (test hook).  lt_plan_canary_status: 0 = not run yet, 1 = passed, 2 = skipped, -1 = failed (then
 * *mismatches_out holds the number of differing populations, or 0 when the check could not run, and *message_out
 * the reason; the string lives as long as the plan). */
int lt_plan_set_canary(lt_plan *plan, int32_t mode);
int lt_plan_canary_status(lt_plan *plan, int32_t *status_out, int64_t *mismatches_out, const char **message_out);

/* Halo transport without compute units (slab drivers, transport "copy"; SURVEY.md 8(e): the reference has no
 * multi-GPU path, north_star asks for send/recv over xGMI overlapped with the interior).  RCCL moves a message with a
 * kernel of 64 workgroups and 20 KB of LDS each, which cannot share a compute unit with a 150 KB sweep workgroup and
 * costs the sweep beside it about 5 %; a device-to-device copy handed to a copy (SDMA) engine needs neither.
 *   lt_ipc_alloc / lt_ipc_open / lt_ipc_close / lt_ipc_free   device memory the other processes of the node can map:
 *       every rank allocates its receive window (halo messages) and, as fine-grained memory (fine_grained = 1:
 *       coherent with writers outside the running kernel), one 64-bit arrival counter per direction; it sends the
 *       64-byte handles to its two z-neighbours through the process group and opens theirs;
 *   lt_halo_copy   dst <- src on `stream`; engine 1 = without compute units (hipMemcpyDeviceToDeviceNoCU: an SDMA
 *       engine, over xGMI when dst is a neighbour's window), 0 = the runtime's choice; *engine_used reports which;
 *   lt_flag_write  *flag <- value in stream order after the copies: how 1 = stream memory operation (command
 *       processor, no kernel), 0 = a one-thread kernel storing at system scope;
 *   lt_flag_wait   one wave that returns when *flag >= at_least or, after about a second, sets *timed_out_dev and
 *       returns all the same (the driver raises at the end of the batch; nothing traps, nothing spins for ever). */
int lt_ipc_alloc(int64_t n_bytes, int32_t fine_grained, void **dev_out, void *handle_out_64_bytes);
int lt_ipc_open(const void *handle_64_bytes, void **dev_out);
int lt_ipc_close(void *mapped_dev);
int lt_ipc_free(void *dev);
int lt_halo_copy(void *dst_dev, const void *src_dev, int64_t n_bytes, int32_t engine, void *stream, int32_t *engine_used);
int lt_flag_write(uint64_t *flag_dev, uint64_t value, int32_t how, void *stream);
int lt_flag_wait(const uint64_t *flag_dev, uint64_t at_least, uint32_t *timed_out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LETTUCE_HIP_H */
