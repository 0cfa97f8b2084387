// sphhip_internal.h -- shared declarations of libsphhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/sphhip.h"

// ---------------------------------------------------------------------------------------------
// Constants handed to every kernel by value.  All folded on the host with the reference's
// expression order (sphhip_derive); kernels never call powf.
// ---------------------------------------------------------------------------------------------
struct DevConsts {
    float h, h2, poly6, spiky, visc_lap, visc_mass, neg_mass, lj_cut, lj24, sigma, mass;
    float stiffness, rest_density, gravity, force_cap, mouse_radius, mouse_r2;
    float dt, max_vel, damping;
    float xmin, xmax, ymin, ymax;
    float inv_cell;
    int nx, ny;
};

// idv word: particle id in bits 0..30, bit 31 set when the reference's `valid` flag is false.
#define SPH_INVALID_BIT 0x80000000u
#define SPH_ID_MASK 0x7fffffffu

// ---- radix sort geometry --------------------------------------------------------------------
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 2048 keys per workgroup
constexpr int SORT_RADIX = 256;

// ---- pair-kernel geometry -------------------------------------------------------------------
// workgroup size of the pair kernels: 128 or 256 (runtime choice, SphCtx::pair_threads)
constexpr int PAIR_MAX_TILE_CELLS = 64;  // cells of one column per workgroup (runtime <= this)

struct PhaseTimer {
    std::vector<hipEvent_t> start[SPHHIP_PH_COUNT], stop[SPHHIP_PH_COUNT];
    size_t used[SPHHIP_PH_COUNT] = {0};
    double ms[SPHHIP_PH_COUNT] = {0};
    long launches[SPHHIP_PH_COUNT] = {0};
};

struct SphCtx {
    SphParams params;
    SphDerived drv;
    DevConsts dc;
    int device = 0;
    int n_max = 0;   // particle capacity
    int n = 0;       // particles currently held (exact only while n_exact; see ctx.hip)
    bool n_exact = true;
    int* n_ring = nullptr;  // device: ring of 8 particle-count words
    int n_slot = 0;
    int* d_n = nullptr;     // device: current particle count (= n_ring + n_slot)
    int* d_err = nullptr;   // device: sticky error bits of asynchronous steps
    int ncell = 0;   // nx * ny
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool have_state = false;
    bool grid_valid = false;
    bool counts_valid = false;
    bool slice_valid = false;    // slots of [slice_lo, slice_hi) columns (at the last sort) = owned particles
    bool tail_appended = false;  // records were appended behind the sorted slots since the last sort
    bool pending_drop = false;   // the stale ghosts outside that slice are still in the arrays: the next grid
                                 // build drops them (sphhip_pack_halo_async's fast path does not compact)
    int slice_lo = 0, slice_hi = 0;
    int col_lo = 0, col_hi = 0;  // owned columns (default: all)
    int tile_cells = 12;         // cells of a column per pair-kernel workgroup
    int tile_cells_max = 12;     // ... its upper bound; sph_adapt_tile_cells picks tile_cells below it for dense scenes
    bool tile_cells_auto = true; // false once SPHHIP_TILE_CELLS / sphhip_set_tuning fixed the tiling
    int pair_cap = 960;          // neighbour candidates staged in LDS per workgroup (4 workgroups of k_force per CU)
    int pair_threads = 256;      // 128 or 256
    bool fast_div = true;        // exact_div.h allowed (constants in its safe range; SPHHIP_FAST_DIV=0 disables)         // neighbour candidates staged in LDS per workgroup

    // state, SoA, all in the same (slot) order
    float2 *pos = nullptr, *vel = nullptr, *pos2 = nullptr, *vel2 = nullptr;
    float *rho = nullptr, *prs = nullptr, *rho2 = nullptr, *prs2 = nullptr;
    uint32_t *idv = nullptr, *idv2 = nullptr;
    int* ncount = nullptr;  // neighbour counts of the last density pass (slot order)
    uint32_t* nmask = nullptr;  // in-range bit masks of the last density pass: word w of slot s at [w * n_max + s]
    int mask_tile_cells = 0, mask_cap = 0;  // tiling the masks were built with
    bool grid_counting = true;  // counting sort by cell (SPHHIP_GRID=radix: the LSD radix sort)
    bool small_grid = true;     // SPHHIP_SMALL_GRID=0: never the one-workgroup grid build (grid.hip: k_grid_small)
    bool lazy_pack = true;      // SPHHIP_LAZY_PACK=0: sphhip_pack_halo_async always compacts
    bool use_masks = true;      // SPHHIP_USE_MASKS=0: k_force always scans

    // grid
    uint32_t *keys = nullptr, *keys2 = nullptr;  // cell key per slot (ping-pong during the sort)
    uint32_t *perm = nullptr, *perm2 = nullptr;  // source slot per sorted slot
    uint32_t* ids_sorted = nullptr;              // id per sorted slot before the in-cell id ranking
    int* cell_start = nullptr;                   // ncell + 1, monotone (= cell_buf[cell_cur])
    int* cell_buf[2] = {nullptr, nullptr};       // two cell tables (+ scan partials): the idle one is zeroed ahead
    int cell_cur = 0;
    bool cell_next_zeroed = false;               // cell_buf[1 - cell_cur] is all zero
    int tab_clo = -1, tab_chi = -1;              // grid columns the cell tables are maintained for (slab + reach)
    int* block_hist = nullptr;                   // SORT_RADIX * nblocks
    int* digit_total = nullptr;                  // SORT_RADIX (+ SORT_RADIX bases)
    int* scan_tmp = nullptr;                     // block partials of the suffix-min scan
    int* counters = nullptr;                     // small device scratch (compaction counters ...)
    int* tile_list = nullptr;                    // non-empty tiles of the pair kernels (pairs.hip), <= ncell entries
    int tile_list_cap = 0;                       // allocated entries (a pair-kernel grid never exceeds it: for_each_tile)
    bool tiles_valid = false;                    // ... current for this grid, tiles_cells, [tiles_lo, tiles_hi)
    int tiles_cells = 0, tiles_lo = 0, tiles_hi = 0;
    int num_cus = 256;
    int* h_host_tiles = nullptr;                 // mapped host ints: tile count of a recent launch (-1: none yet), its tile_cells
    int* d_host_tiles = nullptr;                 // its device address
    int4* dense_list = nullptr;                  // dense sub-tiles of the running pair kernel (pairs.hip: DenseSink)
    int dense_cap = 0;                           // its capacity in entries
    int* h_dense = nullptr;                      // mapped host ints: entries of the last dense density / force kernel (-1: unknown)
    int* d_dense = nullptr;                      // their device address
    int dense_force_part = 256;                  // own particles per entry of the last force-pass list
    bool dense_tp = true;                        // SPHHIP_DENSE_TP=0: never the term-parallel dense force kernel
    void* staging = nullptr;                     // AoS staging for upload / download (28 B * n_max)

    // sphhip_step replays GRAPH_STEPS captured steps as one hipGraph when the scene is small enough to be
    // launch-bound (ctx.hip)
    hipGraphExec_t step_graph = nullptr;
    std::vector<uint64_t> step_graph_sig;        // host state the graph was captured in (and returns to)
    bool graph_off = false;                      // SPHHIP_GRAPH=0, or a capture failed

    // a fused force pass split in two launches (sphhip_step_begin / sphhip_step_finish): the interior is still to run
    struct {
        bool pending = false, masks_ok = false;
        int il = 0, ih = 0;  // interior columns
        int edge_l = 0, edge_r = 0;  // owned columns sphhip_step_begin finished on each side
        float mx = 0, my = 0, strength = 0;
    } split;

    bool profiling = false;
    unsigned profile_mask = ~0u;  // bit ph: bracket that phase with events
    PhaseTimer timer;
};

// ---- error plumbing -------------------------------------------------------------------------
void sph_set_error(const std::string& msg);
int sph_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define HIP_TRY(call)                                                        \
    do {                                                                     \
        hipError_t _e = (call);                                              \
        if (_e != hipSuccess) return sph_hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define LAUNCH_CHECK()                                                       \
    do {                                                                     \
        hipError_t _e = hipGetLastError();                                   \
        if (_e != hipSuccess) return sph_hip_fail(_e, "kernel launch", __FILE__, __LINE__); \
    } while (0)

// ---- phase implementations (each enqueues on ctx->stream and returns) -----------------------
int sph_grid_build(SphCtx* c, bool carry_density);
int sph_density_launch(SphCtx* c);
void sph_adapt_tile_cells(SphCtx* c, int n);  // pairs.hip: tile height for the coming grid build
int sph_force_launch(SphCtx* c, float mx, float my, float strength, bool fuse_integrate, int part = 0, int edge_l = 0,
                     int edge_r = 0);
int sph_integrate_launch(SphCtx* c);
int sph_step_reference_order(SphCtx* c, float mx, float my, float strength, int nsteps);
int sph_set_n_host(SphCtx* c, int n);
int sph_sync_n(SphCtx* c);
// launch bound for per-particle kernels: the exact count, or the capacity while only the device knows it
static inline int sph_n_grid(const SphCtx* c) { return c->n_exact ? c->n : c->n_max; }
int sph_selftest_division(int device, unsigned long long* checked, unsigned long long* mismatches);
int sph_unpack_aos(SphCtx* c, int n, int id_base);
int sph_pack_aos(SphCtx* c, int n, uint32_t* d_ids_out);

// phase timing helpers
void sph_phase_begin(SphCtx* c, int ph);
void sph_phase_end(SphCtx* c, int ph);

struct PhaseScope {
    SphCtx* c;
    int ph;
    PhaseScope(SphCtx* c_, int ph_) : c(c_), ph(ph_) { sph_phase_begin(c, ph); }
    ~PhaseScope() { sph_phase_end(c, ph); }
};

// ---- small device helpers -------------------------------------------------------------------
__device__ __forceinline__ int sph_lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Cell coordinate of x: clamp((int)floorf((x - x0) * inv_cell), 0, n - 1); NaN -> 0.
__device__ __forceinline__ int sph_cell_coord(float x, float x0, float inv_cell, int n) {
    float f = floorf((x - x0) * inv_cell);
    int c;
    if (!(f >= 0.0f))
        c = 0;
    else if (f >= (float)n)
        c = n - 1;
    else
        c = (int)f;
    return c;
}

// Tiles of the pair kernels are listed in bands of PAIR_BAND_COLS adjacent grid columns (pairs.hip).
constexpr int PAIR_BAND_COLS = 4;

// ---- tile list of the pair kernels (pairs.hip: for_each_tile) --------------------------------
// Thread i of a compaction pass looks at one tile of columns [col0, col0 + ncols), in band-major order,
// and appends it to list[] if it holds particles; a wave appends its hits with one atomic.
__device__ __forceinline__ void sph_tile_list_thread(int i, const int* __restrict__ cell_start, int ny,
                                                     int tile_cells, int tiles_per_col, int col0, int ncols,
                                                     int* __restrict__ list, int* __restrict__ count) {
    const int per_band = PAIR_BAND_COLS * tiles_per_col;
    const int band = i / per_band, r = i % per_band;
    const int trow = r / PAIR_BAND_COLS, col = col0 + band * PAIR_BAND_COLS + r % PAIR_BAND_COLS;
    bool hit = false;
    if (col < col0 + ncols) {
        const int cy0 = trow * tile_cells, cy1 = min(cy0 + tile_cells, ny);
        hit = cell_start[col * ny + cy1] > cell_start[col * ny + cy0];
    }
    const unsigned long long hits = __ballot(hit);
    const int lane = sph_lane_id();
    int base = 0;
    if (lane == 0 && hits) base = atomicAdd(count, __popcll(hits));
    base = __shfl(base, 0, 64);
    if (hit) list[base + __popcll(hits & ((1ull << lane) - 1ull))] = col * tiles_per_col + trow;
}

struct TileListArgs {
    int tile_cells, tiles_per_col, col0, ncols, nthreads;
    int* list;
    int* count;
};

// sticky error bits of asynchronous steps (SphCtx::d_err)
enum { SPH_ERR_HALO_OVERFLOW = 1, SPH_ERR_MSG_BAD = 2, SPH_ERR_CAPACITY = 4, SPH_ERR_OUT_OF_SLAB = 8 };

constexpr int SPH_TILE_COUNT_SLOT = 56;  // SphCtx::counters[56]: number of listed tiles
constexpr int SPH_N_SORTED_SLOT = 57;    // counters[57]: particle count after the last grid build
constexpr int SPH_MSG_COUNT_SLOT = 58;   // counters[58..59]: record counts of the halo messages being packed
constexpr int SPH_DENSE_COUNT_SLOT = 61; // counters[61]: entries of the dense sub-tile list
constexpr int SPH_MSG_TICKET_SLOT = 60;  // counters[60]: finished workgroups of the pack kernel

// ring of device-side particle counts (see ctx.hip)
static inline int* sph_n_slot_next(SphCtx* c) { return c->n_ring + ((c->n_slot + 1) & 7); }
static inline void sph_n_advance(SphCtx* c) {
    c->n_slot = (c->n_slot + 1) & 7;
    c->d_n = c->n_ring + c->n_slot;
}

// Tile list parameters for the current tiling and column range: owned columns plus the halo (the force
// pass's range; the density pass skips what it does not need).
static inline void sph_tile_list_args(const SphCtx* c, TileListArgs& tl, int& lo, int& hi) {
    lo = c->col_lo - SPHHIP_HALO_COLS > 0 ? c->col_lo - SPHHIP_HALO_COLS : 0;
    hi = c->col_hi + SPHHIP_HALO_COLS < c->dc.nx ? c->col_hi + SPHHIP_HALO_COLS : c->dc.nx;
    tl.tile_cells = c->tile_cells;
    tl.tiles_per_col = (c->dc.ny + c->tile_cells - 1) / c->tile_cells;
    tl.col0 = lo;
    tl.ncols = hi - lo;
    const int nbands = (hi - lo + PAIR_BAND_COLS - 1) / PAIR_BAND_COLS;
    tl.nthreads = nbands * PAIR_BAND_COLS * tl.tiles_per_col;
    tl.list = c->tile_list;
    tl.count = c->counters + SPH_TILE_COUNT_SLOT;
}
