/*
 * ludwig_hip.h - C ABI of libludwig_hip.so, the MI355X (gfx950) collide-and-stream engine
 * that drops in under OPEN_Ludwig's per-level time-step API.
 *
 * Every entry point names the reference interface it replaces (paths are relative to the
 * reference repository root). The reference is Julia; the binding a maintainer adds is a
 * `ccall` wrapper, shown in INTEGRATION.md and julia/LudwigHIP.jl.
 *
 * Conventions
 *   - plain pointers and sizes only; host arrays are borrowed for the duration of the call.
 *   - host arrays use the reference's memory layout (src/blocks.jl:118-150): Julia
 *     column-major A[x,y,z,b,k] -> linear (x-1) + 8(y-1) + 64(z-1) + 512(b-1) + 512*n_blocks*(k-1);
 *     index tables keep the reference's 1-based values with 0 = absent.
 *   - every function returns LUDWIG_OK (0) or a negative LUDWIG_ERR_* code and never throws;
 *     the message is available from ludwig_last_error() (thread-local).
 *   - a LudwigLevel is not re-entrant; calls are asynchronous on the level's HIP stream and
 *     ordered by it; ludwig_sync() is the reference's KernelAbstractions.synchronize.
 *   - lattice tables (c, w, opp, mirror_y, mirror_z of src/physics_v2.jl:99-117) are compile-time
 *     constants inside the library: k = (cx+1) + 3(cy+1) + 9(cz+1) (+1 in the reference).
 */
#ifndef LUDWIG_HIP_H
#define LUDWIG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUDWIG_ABI_VERSION 1

#define LUDWIG_OK              0
#define LUDWIG_ERR_INVALID    -1   /* bad argument / inconsistent sizes          */
#define LUDWIG_ERR_HIP        -2   /* a HIP runtime call failed                  */
#define LUDWIG_ERR_NO_DEVICE  -3   /* no usable gfx950 device                    */
#define LUDWIG_ERR_ALLOC      -4   /* device or host allocation failed           */
#define LUDWIG_ERR_STATE      -5   /* call not valid for this level's storage    */

/* Fields of BlockLevel (src/blocks.jl:16-65) addressable through upload/download/field_ptr. */
enum LudwigField {
    LUDWIG_F = 0,            /* f                 [8,8,8,nb,27] f32 */
    LUDWIG_F_TEMP = 1,       /* f_temp            [8,8,8,nb,27] f32 */
    LUDWIG_F_POST = 2,       /* f_post_collision  [8,8,8,nb,27] f32 (only if n_boundary_cells > 0) */
    LUDWIG_F_OLD = 3,        /* f_old             [8,8,8,nb,27] f32 (only if temporal storage)     */
    LUDWIG_RHO = 4,          /* rho               [8,8,8,nb]    f32 */
    LUDWIG_RHO_OLD = 5,
    LUDWIG_VEL = 6,          /* vel               [8,8,8,nb,3]  f32 */
    LUDWIG_VEL_TEMP = 7,
    LUDWIG_VEL_OLD = 8,
    LUDWIG_OBSTACLE = 9,     /* obstacle          [8,8,8,nb]    u8 (Bool) */
    LUDWIG_SPONGE = 10,      /* sponge            [8,8,8,nb]    f32 */
    LUDWIG_WALL_DIST = 11,   /* wall_dist         [8,8,8,nb]    f32 */
    LUDWIG_FIELD_COUNT = 12
};

/* Which blocks of a level a launch covers (multi-GPU overlap of halo exchange and interior work). */
enum LudwigPart {
    LUDWIG_PART_ALL = 0,
    LUDWIG_PART_BOUNDARY = 1,   /* owned blocks flagged in LudwigLevelHost.comm_boundary */
    LUDWIG_PART_INTERIOR = 2    /* the other owned blocks                                  */
};

typedef struct LudwigLevel LudwigLevel;   /* opaque; owns all device memory of one BlockLevel */

/*
 * Host description of one BlockLevel, consumed by ludwig_level_create, which replaces
 * `adapt(backend, level)` (src/blocks.jl:67-87, called at src/main.jl:98).
 * Optional pointers may be NULL: the constructor defaults of src/blocks.jl:118-150 apply
 * (rho = 1, vel = 0, f = 0, obstacle = false, sponge = 0, wall_dist = 100).
 */
typedef struct LudwigLevelHost {
    int32_t level_id;              /* 1-based, BlockLevel.level_id                                  */
    int32_t n_blocks;              /* blocks in the arrays = owned blocks followed by ghost blocks  */
    int32_t n_owned;               /* blocks this device steps; 0 means n_blocks (single device),
                                      < 0 means none (the level only holds ghost copies here)        */
    float   tau;                   /* BlockLevel.tau                                                */
    int32_t grid_dim_x, grid_dim_y, grid_dim_z;   /* size(block_pointer)                            */
    const int32_t *block_pointer;  /* [dim_x,dim_y,dim_z], 1-based, 0 = absent (src/blocks.jl:111-114) */
    const int32_t *neighbor_table; /* [n_blocks,27], 1-based, 0 = absent (src/domain_topology.jl:135-160) */
    const int32_t *map_x, *map_y, *map_z;         /* [n_blocks] 1-based block coords                */
    const uint8_t *obstacle;       /* optional */
    const float   *sponge;         /* optional */
    const float   *wall_dist;      /* optional */
    int32_t enable_temporal_interpolation;        /* allocate f_old/rho_old/vel_old (src/blocks.jl:123-142) */
    int32_t n_boundary_cells;      /* > 0 with a q map enables Bouzidi (src/blocks.jl:152); < 0: no cells here, but allocate and
                                      store f_post_collision anyway (multi-GPU: a peer's Bouzidi cells read this rank's face layer) -
                                      in every block, until ludwig_level_add_post_collision_readers says where it is read */
    const uint16_t *bouzidi_q_map; /* Float16 bits [8,8,8,nb,27]; optional                          */
    const int32_t  *bouzidi_cell_block;           /* [n_boundary_cells] 1-based                     */
    const int8_t   *bouzidi_cell_x, *bouzidi_cell_y, *bouzidi_cell_z;   /* 1-based local coords     */
    const uint8_t  *comm_boundary; /* optional [n_blocks]: 1 = owned block adjacent to a ghost block */
    int32_t store_post_collision_everywhere;      /* 0: f_post_collision is written only where it has a reader (the reference writes
                                      it for every cell, src/physics_kernels.jl:350-352, and reads it only in the Bouzidi kernel,
                                      src/bouzidi_kernel.jl:44-77): the x-rows of 8 cells that hold a listed Bouzidi cell with a link
                                      q > 0 or the cell one step behind such a link; whole blocks that hold or touch a listed cell
                                      when the step's q_min_threshold is negative (every direction is a link then) or with
                                      LUDWIG_POST_ROWS=0. The rest of the array keeps what it held (zeros after create).
                                      1: every block, as the reference (multi-GPU: a peer's cells read this rank's blocks)  */
} LudwigLevelHost;

/*
 * Scalar arguments of perform_timestep_v2! (src/physics_v2.jl:26-38) that are constant over a run,
 * plus the globals it reads (SYMMETRIC_ANALYSIS src/physics_v2.jl:71, Q_MIN_THRESHOLD :93).
 */
typedef struct LudwigStepFlags {
    int32_t domain_nx, domain_ny, domain_nz;   /* coarse (level-1) cell dims                 */
    int32_t is_symmetric;
    int32_t wall_model_active;
    int32_t use_temporal_interp;
    int32_t sponge_blend_distributions;
    float   c_wale;
    float   nu_sgs_background;
    float   inlet_turbulence;
    float   q_min_threshold;
} LudwigStepFlags;

/* ---- library ---- */
int         ludwig_abi_version(void);
const char *ludwig_last_error(void);
int         ludwig_device_count(int *count);

/* ---- level life cycle: adapt(backend, BlockLevel) src/blocks.jl:67-87 ---- */
int  ludwig_level_create(const LudwigLevelHost *host, int device, LudwigLevel **out);
void ludwig_level_destroy(LudwigLevel *level);

/* HIP stream (hipStream_t) all later calls on this level are queued on; NULL = the null stream. */
int  ludwig_level_set_stream(LudwigLevel *level, void *hip_stream);

/* Multi-GPU, Bouzidi levels: name the f_post_collision elements somebody OUTSIDE this level's own cell list reads - a peer rank's
 * Bouzidi links reach one cell across a cut (src/bouzidi_kernel.jl:47-58), i.e. exactly the elements of this rank's group-2 send
 * lists (ludwig_halo_plan_create). offsets: element offsets into f_post_collision in the reference layout [8,8,8,n_blocks,27], like the
 * halo index lists. The library adds their x-rows to the rows the stream-collide step stores (store_post_collision_everywhere above);
 * a level created with n_boundary_cells < 0 stops storing every block from the first call on (n = 0 is a valid call: nobody reads).
 * No effect on a level created with store_post_collision_everywhere = 1. May be called again; the sets add up. Synchronizes the stream. */
int  ludwig_level_add_post_collision_readers(LudwigLevel *level, const int64_t *offsets, int64_t n);

/*
 * A HIP stream whose kernels may use every compute unit of `device` except `reserved_cus` of them (0 = an ordinary stream).
 * For the multi-GPU schedule: the interior blocks of step t + 1 run while the halo of step t travels (ludwig_halo_pack, RCCL
 * send/recv, ludwig_halo_unpack on other streams). A stream-collide launch fills every CU, and a send/recv kernel queued beside
 * it - even on a high-priority stream - is handed its workgroups only as the launch drains (measured: 20 us alone, 470 us beside
 * it). Compute units the stepping stream never uses are free the moment the exchange needs them; the step is HBM-bound and does
 * not miss them. The reserved CUs are spread evenly over the XCDs AND over the four shader engines of every XCD (an unbalanced mask costs the
 * step more than a larger balanced one): on a 256-CU device the request is rounded up to a multiple of 32. No reference counterpart (single GPU);
 * hipStream_t in *stream_out.
 */
int  ludwig_stream_create(int device, int reserved_cus, void **stream_out);
int  ludwig_stream_destroy(int device, void *hip_stream);

/*
 * Launch order of the stream-collide kernel. One item per WAVE: items[i] = (block0 << 3) | z with block0 0-based
 * and z in 0..7 = the 8x8 z-plane of that block the wave steps; a negative item is an idle wave. Every
 * (block, plane) of the part must appear exactly once. Purely a performance knob (L2 / Infinity-Cache
 * locality); results do not depend on it. Default: x-runs, plane-per-XCD, see DESIGN.md.
 * What the library makes of the list: 4 consecutive items form one workgroup, and workgroup
 * g is expected on XCD g % 8. A group of 4 items whose blocks are all of one kind (all 26 neighbours present, or all
 * with a missing neighbour) keeps its composition and its slot; items of mixed or incomplete groups are re-packed behind
 * them. Neighbouring items of a group that hold x-adjacent blocks at the same plane exchange their face column through
 * LDS. Levels below 8 192 owned blocks step both kinds in one launch (LUDWIG_MERGE_CLASSES overrides).
 */
int  ludwig_level_set_order(LudwigLevel *level, int part, const int32_t *items, int64_t n_items);

/* Array(level.field) / copyto!(level.field, host) */
int  ludwig_level_upload(LudwigLevel *level, int field, const void *host, size_t bytes);
int  ludwig_level_download(const LudwigLevel *level, int field, void *host, size_t bytes);
/* The device arrays hold the blocks in the library's own order (x-consecutive blocks consecutive in memory), not in the
 * reference's. Every entry point that takes or returns arrays, block ids or element offsets speaks the REFERENCE order and
 * translates; only raw pointers (below) expose the internal one: element (cell, block b, component k) of a field lives at
 * cell + 512 * ref_to_internal[b] + 512 * n_blocks * k. ref_to_internal: [n_blocks], filled by this call (identity when the
 * environment variable LUDWIG_REFERENCE_BLOCK_ORDER is set). */
int  ludwig_level_block_order(const LudwigLevel *level, int32_t *ref_to_internal);

/* Raw device pointer of a field (e.g. to let RCCL receive straight into it); blocks in the internal order, see above. The pointer stays valid for the life of the
 * level and the caller may write through it whenever the level's stream is idle. Because the library cannot see such writes,
 * a level that has handed out a pointer to a state field gives up three internal shortcuts from then on (results are the
 * same, it is only slower): copy_to_old! really copies, the interface values of its children are no longer computed one
 * sub-step ahead, rho is stored by every step. Geometry fields (obstacle, sponge, wall_dist) must be changed with
 * ludwig_level_upload, which also refreshes the per-block flags derived from them. */
int  ludwig_level_field_ptr(const LudwigLevel *level, int field, void **device_ptr, size_t *bytes);
/* Layout of a device array, for raw pointers: the device arrays are BLOCK-major - element (cell, block b, component k) of a field
 * with K components lives at cell + component_stride * k + block_stride * ref_to_internal[b], with component_stride = 512 and
 * block_stride = 512 K (the 27 populations of a block are one contiguous 54-KiB piece), whereas every array passed through the ABI
 * keeps the reference's [8,8,8,n_blocks,K] (src/blocks.jl:118-150: components 512 n_blocks apart). Why: 27 + 27 concurrent streams
 * n_blocks x 2 KiB apart load MI355X's memory system unevenly at distances that depend on nothing but n_blocks.
 * Strides in ELEMENTS; any output pointer may be NULL. */
int  ludwig_level_field_layout(const LudwigLevel *level, int field, int32_t *components, int64_t *block_stride, int64_t *component_stride);

/* init_eq! (src/main.jl:109-134): f = f_temp = (f_old) = w_k, rho_old = 1, vel_old = 0 */
int  ludwig_init_equilibrium(LudwigLevel *level);

/* ---- stepping ---- */
/*
 * perform_timestep_v2! (src/physics_v2.jl:26-97) with the A/B roles of src/solver_control.jl:35-41
 * derived from t_sub: iseven(t_sub) -> in = f/vel, out = f_temp/vel_temp, else swapped.
 * parent == NULL <=> level 1 (parent_f === nothing). For a child, the parent's newest state is the
 * output of the parent's step t_sub >> 1 (src/solver_control.jl:63-83) and is selected the same way.
 * Runs stream-collide, then the Bouzidi correction if the level has boundary cells.
 */
int  ludwig_step(LudwigLevel *level, const LudwigLevel *parent, int64_t t_sub,
                 float u_curr, float parent_tau, float temporal_weight,
                 const LudwigStepFlags *flags);

/* The two kernels of perform_timestep_v2! separately (src/physics_v2.jl:58-83 and :87-96), so that a
 * multi-GPU caller can exchange halos in between and overlap interior work. `part` is a LudwigPart. */
int  ludwig_stream_collide(LudwigLevel *level, const LudwigLevel *parent, int64_t t_sub,
                           float u_curr, float parent_tau, float temporal_weight,
                           const LudwigStepFlags *flags, int part);
/* apply_bouzidi_correction! (src/bouzidi_kernel.jl:99-123) on the output buffer of step t_sub */
int  ludwig_bouzidi_correction(LudwigLevel *level, int64_t t_sub, float q_min_threshold);

/* copy_to_old!(level, f_in, vel_in) (src/blocks.jl:199-205) for the step t_sub about to run */
int  ludwig_save_old(LudwigLevel *level, int64_t t_sub);

/*
 * execute_timestep_batch! (src/solver_control.jl:145-165) in one call: for t = t_start .. t_start+batch_size-1 run
 * recursive_step!(grids, 1, t, ...) - per level: A/B roles from t_sub parity, copy_to_old! when the level has children
 * and temporal interpolation is on, perform_timestep_v2!, then the child level twice (2 t_sub with temporal weight 0.0,
 * 2 t_sub + 1 with 0.5) - and synchronize at the end like the reference. levels[0] is level 1; all on one device.
 */
int  ludwig_execute_timestep_batch(LudwigLevel *const *levels, int32_t n_levels, int64_t t_start, int32_t batch_size,
                                   float u_curr, const LudwigStepFlags *flags);

/* KernelAbstractions.synchronize(backend) (src/solver_control.jl:164) */
int  ludwig_sync(const LudwigLevel *level);

/* ---- surface stresses for the force diagnostics (the caller of the path on the output side) ----
 * map_stresses_kernel! (src/forces/surface.jl:138-266, launched at :406-420): for every triangle the nearest fluid cell
 * of the level (shells of radius 0..search_radius around the cell holding the triangle centre, scan order dz, dy, dx, the
 * first strictly smaller distance wins, the search stops after a shell of radius > 1 once a cell was found), then
 * p = (rho - 1) / 3 * pressure_scale and tau = rho * nu * u_t / d * stress_scale (compute_stress_from_cell :32-76).
 * centers / normals: HOST [n_triangles * 3] Float32 (x, y, z per triangle, mesh coordinates: the offset is added here);
 * outputs: HOST [n_triangles] Float32 each. vel_field: LUDWIG_VEL (what the reference reads, :412) or LUDWIG_VEL_TEMP.
 * The integration (integrate_forces_kernel! :282-366) stays with the caller: it is a sum over these four arrays. */
typedef struct LudwigSurfaceParams {
    float   dx;                 /* level.dx                                  */
    float   tau;                /* level.tau                                 */
    float   offset_x, offset_y, offset_z;   /* params.mesh_offset            */
    float   pressure_scale, stress_scale;
    int32_t search_radius;      /* reference: 5 (src/main.jl:197)            */
} LudwigSurfaceParams;

int  ludwig_map_surface_stresses(const LudwigLevel *level, int vel_field, int32_t n_triangles, const float *centers,
                                 const float *normals, const LudwigSurfaceParams *sp,
                                 float *pressure, float *shear_x, float *shear_y, float *shear_z);

/* compute_flow_stats (src/diagnostics.jl:56-94, CUDA branch), the rho_min column of the run log: minimum of rho over the
 * non-obstacle cells of the blocks this device owns, reduced on the device (+inf for a level without owned blocks). A
 * multi-GPU caller takes the minimum over ranks (one all-reduce MIN). */
int  ludwig_level_rho_min(const LudwigLevel *level, float *rho_min);

/* rho store policy. perform_timestep_v2!'s kernel writes rho for every cell on every step (src/physics_kernels.jl:243-246). By
 * default a level nobody reads rho of between two steps skips that store and reproduces the array on demand, bit for bit (every
 * reader inside the library asks for it). every_step = 1 restores the reference's store pattern on this level
 * (LUDWIG_EAGER_RHO=1 in the environment does it for every level), 0 lets the library elide again. Results never depend on it. */
int  ludwig_level_set_rho_store(LudwigLevel *level, int every_step);

/* ---- halo exchange helpers (no reference counterpart: the reference is single-device) ---- */
/* dst[i] = field[index[i]] / field[index[i]] = src[i]; index, dst, src are DEVICE pointers, index holds element
 * offsets into the field in the reference layout. hip_stream: the stream to queue on (hipStream_t), NULL = the
 * level's stream - the exchange normally runs on its own stream so that it overlaps the interior update. */
int  ludwig_halo_pack(const LudwigLevel *level, int field, const int64_t *index_dev, int64_t n,
                      float *dst_dev, void *hip_stream);
int  ludwig_halo_unpack(LudwigLevel *level, int field, const int64_t *index_dev, int64_t n,
                        const float *src_dev, void *hip_stream);

/* ---- multi-GPU: communicator, halo plan, exchange, distributed step (no reference counterpart: the reference is single-device,
 * src/main.jl:75; its caller for this path is execute_timestep_batch!, src/solver_control.jl:145-165, which a multi-GPU host calls
 * once per rank) ----
 * One process per GPU. RCCL (ncclSend / ncclRecv, point-to-point over xGMI) is resolved at run time from the librccl already mapped
 * into the process (e.g. PyTorch's) so that two copies never coexist; otherwise librccl.so(.1) is opened from the loader's path
 * (LUDWIG_RCCL_LIB overrides). The library does not link against it: single-GPU users never load it. */
typedef struct LudwigComm LudwigComm;
#define LUDWIG_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: called by ONE rank; the host carries the 128 bytes to the others by its own means (MPI, a file, a socket). */
int  ludwig_comm_unique_id(void *id_out);
/* ncclCommInitRank on `device` (collective over the `world` ranks). */
int  ludwig_comm_create(const void *unique_id, int rank, int world, int device, LudwigComm **out);
void ludwig_comm_destroy(LudwigComm *comm);
/* in-place all-reduce of a few Float32 scalars over the communicator (diagnostics: rho_min, force sums); op: 0 sum, 2 max, 3 min;
 * buf is a HOST array. NaN in any rank's value gives NaN for min / max (the reference's minimum() propagates it). */
int  ludwig_comm_allreduce_f32(LudwigComm *comm, float *buf, int32_t n, int32_t op);

/* The ghost elements of one level this rank receives after a step and the owned elements it sends, per peer and per logical field
 * GROUP: 0 = populations (f or f_temp), 1 = velocity (vel or vel_temp), 2 = f_post_collision, 3 = rho. For every group the lists of
 * all peers are concatenated in peer order: index[count[0] + ... + count[p-1] ...] belongs to peer p. Element offsets are in the
 * REFERENCE layout of the level (cell + 512 b + 512 n_blocks k); a peer's receive list and the matching send list of that peer name
 * the same elements in the same order. Everything is translated and uploaded here, once: a step then costs three enqueue calls. */
#define LUDWIG_HALO_GROUPS 4
typedef struct LudwigHaloPlanDesc {
    int32_t n_peers;
    const int32_t *peer_ranks;                          /* [n_peers] rank in the communicator; this rank itself = a device copy   */
    const int64_t *send_count[LUDWIG_HALO_GROUPS];      /* [n_peers] each; NULL = the group is empty                              */
    const int64_t *recv_count[LUDWIG_HALO_GROUPS];
    const int64_t *send_index[LUDWIG_HALO_GROUPS];      /* HOST arrays, concatenated over peers                                   */
    const int64_t *recv_index[LUDWIG_HALO_GROUPS];
} LudwigHaloPlanDesc;
typedef struct LudwigHaloPlan LudwigHaloPlan;
/* comm may be NULL when every peer is this rank itself (periodic wrap onto the own brick; tests). */
int  ludwig_halo_plan_create(LudwigLevel *level, LudwigComm *comm, const LudwigHaloPlanDesc *desc, LudwigHaloPlan **out);
void ludwig_halo_plan_destroy(LudwigHaloPlan *plan);

/* One exchange: for i < n: group groups[i] of the plan moves field fields[i] (a LudwigField with as many components as the group).
 * Queued on the plan's own HIGH-priority stream behind everything queued on the level's stream so far: pack (one kernel per group)
 * -> ncclGroupStart, one ncclSend / ncclRecv per peer and group, ncclGroupEnd -> unpack. Returns at once; nothing queued on the
 * level's stream LATER waits for it until ludwig_halo_wait, so work that reads no ghost can run under it. */
int  ludwig_halo_exchange(LudwigHaloPlan *plan, int32_t n, const int32_t *groups, const int32_t *fields);
/* everything queued on the level's stream after this call runs after the plan's last exchange */
int  ludwig_halo_wait(LudwigHaloPlan *plan);
/* For hosts that carry the messages themselves (a transport other than RCCL; the one-GPU rehearsals over gloo): the two halves of an
 * exchange on `hip_stream` (NULL = the level's), and the message buffers (DEVICE pointers, Float32, peers concatenated as in the plan). */
int  ludwig_halo_plan_pack(LudwigHaloPlan *plan, int32_t group, int32_t field, void *hip_stream);
int  ludwig_halo_plan_unpack(LudwigHaloPlan *plan, int32_t group, int32_t field, void *hip_stream);
int  ludwig_halo_plan_buffers(const LudwigHaloPlan *plan, int32_t group, void **send_dev, int64_t *n_send, void **recv_dev, int64_t *n_recv);
/* In-stream mode: ludwig_halo_exchange queues pack, transfer and unpack on the LEVEL's stream instead of the plan's own - no overlap
 * with what the level launches next, and no cross-stream hand-over either (each costs the device tens of idle microseconds: more than
 * the whole exchange of a small level; nested levels take 2^(l-1) steps per coarse step). ludwig_halo_wait is then a no-op. The Python
 * host chooses it for levels below 8 192 owned blocks (LUDWIG_HALO_IN_STREAM_BELOW). Same messages, same bits. */
int  ludwig_halo_plan_in_stream(LudwigHaloPlan *plan, int enable);
/* Benchmarks: with timing on, every ludwig_halo_exchange is bracketed by events on the plan's stream (the span includes waiting beside
 * whatever the device is busy with); ludwig_halo_plan_exchange_ms returns the spans of the exchanges finished since the last call
 * (at most `max`, oldest first; call after a device synchronize). */
int  ludwig_halo_plan_timing(LudwigHaloPlan *plan, int enable);
int  ludwig_halo_plan_exchange_ms(LudwigHaloPlan *plan, float *ms_out, int32_t max, int32_t *n_out);

/* perform_timestep_v2! (src/physics_v2.jl:26-97) of a level whose blocks are spread over ranks, with the exchange hidden behind
 * compute: interior blocks of step t_sub (they read no ghost) -> wait for the exchange of the previous step -> boundary blocks
 * -> [f_post_collision halo of the links that reach across a cut, Bouzidi correction] -> exchange of this step's output (groups 0
 * and 1: the populations and velocity buffer step t_sub wrote), left in flight under the next call's interior blocks.
 * ludwig_halo_wait (or the next call) joins it; ludwig_sync waits for the device. Same arguments as ludwig_step. */
int  ludwig_step_distributed(LudwigLevel *level, LudwigHaloPlan *plan, const LudwigLevel *parent, int64_t t_sub, float u_curr,
                             float parent_tau, float temporal_weight, const LudwigStepFlags *flags);

/* ---- introspection for benchmarks ---- */
typedef struct LudwigLevelInfo {
    int32_t n_blocks, n_owned;
    int32_t n_fast_blocks;        /* owned blocks with all 26 neighbours present (no edge / interface patching)   */
    int32_t n_general_blocks;
    int32_t n_boundary_cells;
    int32_t has_temporal_storage, has_post_collision;
    int32_t n_xrun_blocks;        /* of the fast blocks: how many the current LUDWIG_PART_ALL order steps next to an x
                                     neighbour in the same workgroup (face column through LDS instead of global memory) */
    int64_t device_bytes;
} LudwigLevelInfo;
int  ludwig_level_info(const LudwigLevel *level, LudwigLevelInfo *info);

#ifdef __cplusplus
}
#endif
#endif /* LUDWIG_HIP_H */
