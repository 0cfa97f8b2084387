/*
 * sphhip.h -- C ABI of the MI355X-native SPH step (libsphhip.so).
 *
 * Drop-in boundary for the step interface of the alphazm/SPH reference.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference tree).  Plain C:
 * POD structs, raw pointers and sizes, int error codes.  No HIP / torch types appear here on
 * purpose: the reference's host header defines its own global `struct float2` (SPH/sph.h:22-25)
 * which collides with HIP's, so host code that includes sph.h can include this file safely.
 *
 * Particle record (SPH/sph.h:27-33, duplicated SPH/sph.cuh:25-31): 28 bytes,
 *   pos.x pos.y @0, vel.x vel.y @8, density @16, pressure @20, valid(bool) @24, 3 pad bytes.
 * Host arrays handed to this library use exactly that layout and the reference's index order;
 * the library's internal (cell-sorted, SoA) order is never visible through upload/download.
 *
 * Error convention: every function returns 0 on success, a negative SPHHIP_E* code otherwise
 * (the reference's CUDA_CHECK prints and calls exit(1), SPH/sph.cuh:17-23; a library must not).
 * sphhip_last_error() returns a human-readable message for the calling thread's last failure.
 */
#ifndef SPHHIP_H
#define SPHHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHHIP_PARTICLE_BYTES 28
#define SPHHIP_HALO_COLS 2 /* ghost columns a slab needs on each side (see sphhip_pack_halo) */
/* owned columns next to a slab boundary that the halo messages of the next step are packed from (a particle
 * crosses less than one column per step): the edge widths sphhip_step_begin must be given on a side with a
 * neighbour */
#define SPHHIP_PACK_REACH (SPHHIP_HALO_COLS + 1)

enum {
    SPHHIP_OK = 0,
    SPHHIP_EINVAL = -1,   /* bad argument (null pointer, n out of range, bad column range ...) */
    SPHHIP_EHIP = -2,     /* a HIP runtime call failed; see sphhip_last_error() */
    SPHHIP_ESTATE = -3,   /* call order violated (e.g. forces before any upload) */
    SPHHIP_ECAPACITY = -4 /* more particles / cells than the context was created for */
};

/*
 * Simulation constants.  The defaults (sphhip_default_params) reproduce the reference's
 * compile-time #defines bit for bit (SPH/sph.h:11-20) together with the literals that are
 * hard-coded in its step (SPH/sph.cpp:77 gravity 10.0f, :81-84 mouse radius 0.25f,
 * :91-92 force cap 1000.0f, :113-128 walls 0.05f / 1.95f).  Making the walls parameters is what
 * lets the 1M..64M-particle scenes keep ~50 neighbours per particle: the reference's fixed
 * 1.9 x 1.9 box makes the neighbour count grow linearly with N.
 */
typedef struct SphParams {
    float h;            /* H            0.15f   smoothing radius                     */
    float dt;           /* DT           0.0005f                                      */
    float mass;         /* MASS         0.05f                                        */
    float viscosity;    /* VISCOSITY    0.3f                                         */
    float stiffness;    /* STIFFNESS    1000.0f                                      */
    float rest_density; /* REST_DENSITY 1000.0f                                      */
    float damping;      /* DAMPING      0.9f    wall restitution                     */
    float max_vel;      /* MAX_VEL      5.0f                                         */
    float sigma;        /* SIGMA        0.05f   Lennard-Jones                        */
    float epsilon;      /* EPSILON      1.0f    Lennard-Jones                        */
    float gravity;      /* 10.0f   (sph.cpp:77)                                      */
    float force_cap;    /* 1000.0f (sph.cpp:91-92), multiplied by the density        */
    float mouse_radius; /* 0.25f   (sph.cpp:81-84)                                   */
    float xmin, xmax;   /* walls, 0.05f / 1.95f (sph.cpp:113-120)                    */
    float ymin, ymax;   /* walls, 0.05f / 1.95f (sph.cpp:121-128)                    */
    float cell_size;    /* uniform-grid cell width; 0 => h * (1 + 1/256).  Must be   */
                        /* > h so that every pair with r2 < h*h lies in the 3x3      */
                        /* stencil despite fp32 rounding of the cell coordinate.     */
} SphParams;

/* Values derived from SphParams on the host with the reference's own expression order
 * (powf on constants only: sph.cpp:11,18,25).  Exposed so tests can pin them. */
typedef struct SphDerived {
    float h2;          /* h*h                                     0x3cb851ec at defaults */
    float poly6;       /* 315/(64*3.14159f*powf(h,9))             0x4c1b75da            */
    float spiky;       /* -45/(3.14159f*powf(h,6))                0xc999818a            */
    float visc_lap;    /* 45/(3.14159f*powf(h,6))                 0x4999818a            */
    float visc_mass;   /* viscosity*mass                          0x3c75c290            */
    float lj_cut;      /* 2.5f*sigma                              0x3e000000            */
    float lj_24eps;    /* 24.0f*epsilon                                                 */
    float neg_mass;    /* -mass                                                         */
    float mouse_r2;    /* mouse_radius*mouse_radius                                     */
    float cell;        /* effective cell width                                          */
    float inv_cell;    /* 1.0f / cell                                                   */
    int   nx, ny;      /* grid columns (x) and rows (y)                                 */
} SphDerived;

typedef struct SphCtx SphCtx; /* opaque; owns all device memory (no hidden globals, cf. sph.cu:10) */

/* ---- parameters ------------------------------------------------------------------------- */

/* Constants of SPH/sph.h:11-20 + the literals listed above. */
int sphhip_default_params(SphParams* out);
/* Host-side derivation (no GPU needed). */
int sphhip_derive(const SphParams* p, SphDerived* out);

/* ---- context ---------------------------------------------------------------------------- */

/* Creates a context on HIP device `device` able to hold n_max particles.
 * Replaces the hidden global `d_particles` + cudaMalloc of initSimulation (SPH/sph.cu:10,166-169). */
int sphhip_create(const SphParams* p, int n_max, int device, SphCtx** out);
/* Replaces cleanupSimulation (SPH/sph.cu:275-280). */
int sphhip_destroy(SphCtx* ctx);
/* Run on a caller-owned HIP stream (hipStream_t as void*; NULL => the context's own stream).  The context's side
 * stream (interior force pass of a split slab step) is re-created at that stream's priority: the two wait for
 * each other, and a stream that another one waits for must never have the lower priority. */
int sphhip_set_stream(SphCtx* ctx, void* hip_stream);
/* Run on the device's default (null) stream, e.g. to be ordered with a framework that works on
 * it (the context's own stream is non-blocking and does not synchronise with the null stream). */
int sphhip_use_default_stream(SphCtx* ctx);
int sphhip_get_stream(SphCtx* ctx, void** hip_stream);
int sphhip_synchronize(SphCtx* ctx);
const char* sphhip_last_error(void);
/* Pair-kernel tuning: cells of one grid column per workgroup (1..64) and neighbour candidates
 * staged in LDS per workgroup (64..4608); 0 keeps the current value.  Results do not depend on
 * either (the per-particle summation order is fixed).  No counterpart in the reference
 * (its launch geometry is hard-coded, SPH/sph.cu:219-220). */
int sphhip_set_tuning(SphCtx* ctx, int tile_cells, int pair_cap);

/* ---- state transfer --------------------------------------------------------------------- */

/* Host AoS (28-byte records, reference index order) -> device.  Replaces the H2D copy of
 * initSimulation / CUDA_performance_test (SPH/sph.cu:202,243).  Particle i gets global id
 * id_base + i.  Invalidates the cell list. */
int sphhip_upload(SphCtx* ctx, const void* aos28, int n, int id_base);
/* Device -> host AoS, un-permuted to ascending id (record k = k-th smallest id; with
 * id_base=0 that is the reference's index order).  Replaces the full D2H of stepSimulation
 * (SPH/sph.cu:235). */
int sphhip_download(SphCtx* ctx, void* aos28, int n);
/* Same, and also returns the id of every record (ids[k], ascending): what a slab rank needs to
 * place its particles in the global array. */
int sphhip_download_ids(SphCtx* ctx, void* aos28, int* ids, int n);
int sphhip_count(SphCtx* ctx, int* n);

/* ---- phases of the step (SPH/sph.h:35-39; device flavour SPH/sph.cuh:33-38) ------------ */

/* Neighbour search: cell hash, sort by cell (one counting pass with LDS/wave-level run detection by default;
 * SPHHIP_GRID=radix selects the LDS-local LSD radix sort), cell start/end.  No counterpart in the
 * reference beyond the dead `int* grid, int* cellStart, int* cellEnd` parameters of its kernels
 * (SPH/sph.cuh:33-34,38; always nullptr, SPH/sph.cu:222-223).  Reorders the device state into
 * (cell key, id) order. */
int sphhip_build_grid(SphCtx* ctx);
/* serial_computeDensityPressure (SPH/sph.cpp:41-55) / computeDensityPressure (SPH/sph.cu:48-66). */
int sphhip_density(SphCtx* ctx);
/* serial_computeForces (SPH/sph.cpp:57-99) / computeForces (SPH/sph.cu:68-129): pair forces,
 * gravity, mouse force, force clamp, velocity update.  Neighbour velocities are read from the
 * state as it was when the call started (snapshot / Jacobi semantics, as mpi_computeStep does,
 * SPH/MPI_sph.cpp:144-147,181); the serial loop's in-place update (sph.cpp:96-97) is inherently
 * sequential. */
int sphhip_forces(SphCtx* ctx, float mouse_x, float mouse_y, float strength);
/* serial_integrate (SPH/sph.cpp:101-130) / integrate (SPH/sph.cu:131-164). */
int sphhip_integrate(SphCtx* ctx);
/* serial_stepSimulation (SPH/sph.cpp:166-171) x nsteps with the state resident on the device:
 * grid, density, then forces + velocity update + integrate + wall clamp fused in one kernel.
 * Replaces the 3-kernel loop of CUDA_performance_test (SPH/sph.cu:254-258).  Asynchronous, but the host is kept
 * less than 64 steps ahead of the GPU (it waits on an event every 32 steps): what the library decides from its
 * kernels' reports through mapped memory must not be hundreds of steps old. */
int sphhip_step(SphCtx* ctx, float mouse_x, float mouse_y, float strength, int nsteps);
/* NO COUNTERPART IN THE REFERENCE.  The same steps as sphhip_step for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): no context, no hipMalloc /
 * hipFree / hipHostMalloc / stream or event creation inside.  The caller passes the input and the output, both DEVICE
 * arrays of n 28-byte Particle records in the reference's index order (they may be the same array); every working
 * buffer (sorted SoA state, cell tables, masks, tile list, counters) lives in a static device array of the library
 * (16 MiB per GPU the library runs on, held until the process ends).  Limits: n <= SPHHIP_STATIC_MAX_N and a grid
 * of at most SPHHIP_STATIC_MAX_CELLS cells (SphDerived nx * ny), SPHHIP_ECAPACITY beyond.  Enqueues on `hip_stream`
 * (hipStream_t as void*; NULL = the null stream) of HIP device `device` and returns without synchronising; the
 * output is complete when the stream reaches the end of the call's work.  All calls on one GPU share that GPU's
 * array, which keeps the last call's contents (the call clears what must start at zero): calls that overlap on one
 * GPU -- two streams, two host threads -- race.  Results equal sphhip_upload + sphhip_step + sphhip_download bit for
 * bit. */
#define SPHHIP_STATIC_MAX_N 65536
#define SPHHIP_STATIC_MAX_CELLS 65536
int sphhip_step_static(const SphParams* p, int device, const void* d_aos_in, void* d_aos_out, int n, float mouse_x,
                       float mouse_y, float strength, int nsteps, void* hip_stream);
/* One step in two halves, for a slab driver that overlaps the halo exchange with the step (replaces the
 * Allgather-then-compute order of mpi_computeStep, SPH/MPI_sph.cpp:135-243, where every rank waits for the
 * full state before it computes).  sphhip_step_begin: grid, density, and the force + integrate pass of the
 * ghost columns and of the first edge_left / last edge_right owned columns -- everything the next halo
 * messages are packed from.  sphhip_step_finish: the same pass over the remaining owned columns.  Between
 * the two only sphhip_pack_halo_async (and stream / error queries) may be called, and only with send ranges
 * inside the halo and edge widths >= SPHHIP_PACK_REACH on every side that has a message (SPHHIP_ESTATE
 * otherwise); begin + finish give exactly the state of sphhip_step(..., 1). */
int sphhip_step_begin(SphCtx* ctx, float mouse_x, float mouse_y, float strength, int edge_left, int edge_right);
int sphhip_step_finish(SphCtx* ctx);

/* ---- cell-list and neighbour diagnostics (parity: bit-exact integers) ------------------- */

/* Any pointer may be NULL.  keys[n]: cell key per sorted slot; order[n]: particle id per sorted
 * slot; cell_start/cell_end[nx*ny]: half-open slot range of each cell (empty: start==end).
 * Requires a valid grid (sphhip_build_grid or a step). */
int sphhip_get_cells(SphCtx* ctx, int* keys, int* order, int* cell_start, int* cell_end);
/* counts[k] = #{j : r2 < h*h} including self (the reference predicate, sph.cpp:46-48), for the
 * k-th smallest id.  Written by the most recent density pass. */
int sphhip_get_neighbor_counts(SphCtx* ctx, int* counts, int n);

/* serial_stepSimulation (SPH/sph.cpp:166-171) in the reference's own evaluation order: all pairs in
 * ascending j, and serial_computeForces' in-place velocity update (sph.cpp:96-97: particle i sees the
 * updated velocity of every j < i).  Inherently sequential in i -- a parity tool for n <= 4096
 * (SPHHIP_ECAPACITY above), O(n^2); particle ids must be 0..n-1.  Results equal the reference's serial
 * step bit for bit (sphhip_step computes the snapshot semantics of SPH/MPI_sph.cpp instead). */
int sphhip_step_reference_order(SphCtx* ctx, float mouse_x, float mouse_y, float strength, int nsteps);

/* ---- slab decomposition hooks (replace the index-block + Allgather scheme of
 *      mpi_computeStep, SPH/MPI_sph.cpp:135-166,255-257) --------------------------------- */

/* Restrict density / forces / integrate to particles whose cell column is in [col_lo, col_hi)
 * (density is evaluated one column wider on each side so that ghost densities are recomputed
 * locally instead of exchanged, cf. the dens/pres Allgathers MPI_sph.cpp:163-166).
 * Default: all columns.  With a restricted range the context is expected to hold only particles of
 * the owned columns and of SPHHIP_HALO_COLS ghost columns on each side (what sphhip_pack_halo /
 * sphhip_append_message produce); position and velocity of particles farther out are undefined
 * after a step. */
int sphhip_set_owned_columns(SphCtx* ctx, int col_lo, int col_hi);
/* Classifies every particle by the grid column of its CURRENT position.  Copies it, as a 20-byte
 * record {x,y,vx,vy,idv}, into d_left if column < send_hi_left and into d_right if column >=
 * send_lo_right (device buffers, capacity in records; migrants and ghosts for the neighbour
 * slabs).  Ghost copies left from the last step are dropped; a particle that has just left the
 * owned columns (set by sphhip_set_owned_columns) is sent AND kept for one more step as a ghost
 * while it is within SPHHIP_HALO_COLS columns of the slab, because it is still a neighbour of
 * the particles that stay; anything farther out is dropped.  Returns the two counts after a
 * stream sync.  The grid is not required. */
int sphhip_pack_halo(SphCtx* ctx, int send_hi_left, int send_lo_right,
                     void* d_left, int cap_left, int* n_left,
                     void* d_right, int cap_right, int* n_right);
/* The same split WITHOUT a host synchronisation, for an exchange whose size the host fixes beforehand:
 * d_msg_left / d_msg_right are device buffers of (cap_left + 1) / (cap_right + 1) 20-byte records whose
 * record 0 is a header (word 0 = number of records that follow); either may be NULL (no neighbour on
 * that side).  A message can be sent as it is -- its first (cap + 1) records -- stream-ordered behind
 * this call; sender and receiver agree on cap per slab boundary (both derive it from the global column
 * histogram), so no count has to travel first.  From here on the particle count is
 * known on the device only (kernels read it there; launches are sized by the capacity) until a
 * call that needs it on the host synchronises, e.g. sphhip_sync_counts, sphhip_download. */
int sphhip_pack_halo_async(SphCtx* ctx, int send_hi_left, int send_lo_right, void* d_msg_left, int cap_left,
                           void* d_msg_right, int cap_right);
/* Appends the records of a received fixed-size message (header + up to cap records, device
 * memory), reading the count from its header on the device: no host synchronisation. */
int sphhip_append_message(SphCtx* ctx, const void* d_msg, int cap);
/* The messages from both neighbours in one launch: a first, b behind it (the same result as two
 * sphhip_append_message calls in that order); either may be NULL. */
int sphhip_append_messages(SphCtx* ctx, const void* d_msg_a, int cap_a, const void* d_msg_b, int cap_b);
/* Synchronises the stream, returns the particle count (n may be NULL) and reports, as
 * SPHHIP_ECAPACITY, any overflow (halo message buffer, context capacity) that happened in the
 * asynchronous calls since the last synchronisation. */
int sphhip_sync_counts(SphCtx* ctx, int* n);
/* Drops every particle whose column is outside [col_lo, col_hi); sends nothing. */
int sphhip_keep_columns(SphCtx* ctx, int col_lo, int col_hi);
/* Drops the ghost copies left over from the last step on a slab (particles that were outside the
 * owned columns when that step sorted them); sphhip_pack_halo does the same implicitly.  Needed
 * before a download that must return owned particles only. */
int sphhip_drop_ghosts(SphCtx* ctx);
/* Appends n 20-byte records (device memory) to the state; density/pressure of the new
 * particles are undefined until the next density pass. */
int sphhip_append(SphCtx* ctx, const void* d_records, int n);
/* Per-column particle histogram (nx ints, host) of the current positions: input of the
 * count-balanced slab boundaries. */
int sphhip_column_histogram(SphCtx* ctx, int* hist);

/* Self-test of the library's own correctly rounded reciprocal / division sequences
 * (sph_amd/csrc/exact_div.h) against the compiler's IEEE `/` on the GPU: every float of the safe
 * range for 1/x, 2^31 random pairs for a/b.  No counterpart in the reference (it divides with `/`,
 * SPH/sph.cpp:18,31,69,72); this is what licenses the cheaper sequence to claim the same bits. */
int sphhip_selftest_division(int device, unsigned long long* checked, unsigned long long* mismatches);

/* ---- measurement ------------------------------------------------------------------------ */

enum {
    SPHHIP_PH_HASH = 0,
    SPHHIP_PH_SORT,
    SPHHIP_PH_CELLS,
    SPHHIP_PH_REORDER,
    SPHHIP_PH_DENSITY,
    SPHHIP_PH_FORCE,      /* force (+ fused integrate inside sphhip_step) */
    SPHHIP_PH_INTEGRATE,  /* stand-alone integrate */
    SPHHIP_PH_FORCE_INTERIOR, /* split slab step: the interior columns' force pass, which runs on the context's side
                               * stream beside the boundary columns' pass (that one is counted under SPHHIP_PH_FORCE) */
    SPHHIP_PH_COUNT
};
/* When enabled, every phase is bracketed by HIP events on the context's stream. */
int sphhip_set_profiling(SphCtx* ctx, int enable);
/* Same, for the phases whose bit (1u << SPHHIP_PH_*) is set only: event records between
 * back-to-back small kernels cost ~10 % of a step at 1M particles, one pair around the dominant
 * kernel costs 4-5 % (see sphhip_set_profiling_period). */
int sphhip_set_profiling_mask(SphCtx* ctx, unsigned phase_mask);
/* Bracket the selected phases in every period-th step only (default 1: every step).  An event pair leaves ~11 us of
 * gaps on the stream (rocprofv3 kernel trace: 5.6 us before and after the bracketed kernel): sampling one step in
 * eight keeps a live measurement of the kernel inside a timed region without slowing that region by 4 %. */
int sphhip_set_profiling_period(SphCtx* ctx, int period);
/* Sums since the last reset: milliseconds and launches per phase (arrays of SPHHIP_PH_COUNT). */
int sphhip_get_phase_times(SphCtx* ctx, float* ms, int* launches);
int sphhip_reset_phase_times(SphCtx* ctx);
/* What the last pair-kernel launches reported about their work list (after a synchronisation): stats[0] entries of
 * the list (workgroup-sized tiles), [1] rows of a base tile, [2] entries with a long stencil range (257..448
 * candidates: handled in place by a second pass over the mask words), [3] / [4] entries of the dense sub-tile
 * list of the last density / force pass; -1 = not reported yet. */
int sphhip_get_tile_stats(SphCtx* ctx, int* stats5);
/* Total in-range pairs (j != i, r2 < h*h) counted by the most recent density pass. */
int sphhip_get_pair_count(SphCtx* ctx, long long* pairs);

#ifdef __cplusplus
}
#endif
#endif /* SPHHIP_H */
