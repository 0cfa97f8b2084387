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
// The constant part of a pair kernel's dense-sub-tile sink (pairs.hip: DenseSink), one per pass, in device memory.
struct DenseSinkDev {
    int4* list;      // (col, first row, rows, part)
    int* count;
    int capacity;
    int* host_flag;  // mapped host memory
    int* err;
};

struct DevConsts {
    float h, h2, poly6, spiky, visc_lap, visc_mass, neg_mass, lj_cut, lj24, sigma, mass;
    float stiffness, rest_density, gravity, force_cap, mouse_radius, mouse_r2;
    float dt, max_vel, damping;
    float xmin, xmax, ymin, ymax;
    float inv_cell;
    int nx, ny;
};

// What k_force needs rarely, in device memory instead of kernel arguments (scalar registers): the constants again, for
// the epilogue, and the array of mask words 8..13 (long tiles only).
struct ForceExtraDev {
    DevConsts c;
    const uint32_t* nmask_hi;
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
    // slab steps: the interior force pass runs on this stream beside the boundary columns' pass
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Bound on how far the host runs ahead of the GPU (sph_throttle): an event every THROTTLE_STEPS steps, two in rotation
    hipEvent_t ev_pace[2] = {nullptr, nullptr};
    long pace_steps = 0;
    bool pace_armed[2] = {false, false};
    bool split_concurrent = true;  // SPHHIP_SPLIT_CONCURRENT=0: the two passes one after the other on the context's stream
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
    int tile_cells = 15;         // rows of a column per base tile of the pair kernels (TileDesc)
    int tile_cells_max = 16;     // ... its upper bound; sph_adapt_tile_cells picks tile_cells from the scene's cell population
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
    uint32_t* nmask_hi = nullptr;  // words 8..13 per particle (long stencil ranges), same word-major layout
    int mask_tile_cells = 0, mask_cap = 0;  // tiling the masks were built with
    bool grid_counting = true;  // counting sort by cell (SPHHIP_GRID=radix: the LSD radix sort)
    bool small_grid = true;     // SPHHIP_SMALL_GRID=0: never the one-workgroup grid build (grid.hip: k_grid_small)
    bool lazy_pack = true;      // SPHHIP_LAZY_PACK=0: sphhip_pack_halo_async always compacts
    bool use_masks = true;      // SPHHIP_USE_MASKS=0: k_force always scans
    bool zero_p = true;         // SPHHIP_ZERO_P=0: k_force never drops the pressure term of an all-zero-pressure tile

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
    struct TileDesc* tile_list = nullptr;        // work list of the pair kernels: one entry per workgroup pass (see TileDesc)
    int tile_list_cap = 0;                       // allocated entries (a pair-kernel grid never exceeds it: for_each_tile)
    bool tiles_valid = false;                    // ... current for this grid, tiles_cells, [tiles_lo, tiles_hi)
    int tiles_cells = 0, tiles_lo = 0, tiles_hi = 0, tiles_cap = 0, tiles_threads = 0;
    int num_cus = 256;
    int* h_host_tiles = nullptr;                 // mapped host ints: tile count of a recent launch (-1: none yet), its tile_cells, long entries (-1: unknown)
    int* d_host_tiles = nullptr;                 // its device address
    DenseSinkDev* d_sinks = nullptr;             // [0] density pass, [1] force pass
    ForceExtraDev* d_consts = nullptr;           // device copy of dc (+ rarely used pointers): k_force reads its epilogue constants from here
    int4* dense_list = nullptr;                  // dense sub-tiles of the running pair kernel (pairs.hip: DenseSink)
    int dense_cap = 0;                           // its capacity in entries
    int* big_list = nullptr;                     // crowded cells of the current grid build (k_scan_apply -> k_rank_big)
    int big_cap = 0;
    bool tile_list_fused = true;                 // the work list is built by head workgroups of k_cell_scatter (SPHHIP_TILE_LIST_FUSED=0: own launch)
    bool rank_big = true;                        // SPHHIP_RANK_BIG=0: in-cell ranking always by k_rank_reorder's loop
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
        bool interior_enqueued = false;  // sphhip_step_begin already started the interior pass on the side stream
        float mx = 0, my = 0, strength = 0;
    } split;

    bool profiling = false;
    unsigned profile_mask = ~0u;  // bit ph: bracket that phase with events
    int profile_period = 1;       // ... in every profile_period-th step only (an event pair costs the stream ~11 us of gaps)
    long profile_step = 0;        // steps begun since profiling was last configured
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
int sph_rank_big_setup();  // grid.hip: LDS opt-in of k_rank_big
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
void sph_phase_begin(SphCtx* c, int ph, hipStream_t on);  // ... on another stream than the context's (the side stream)
void sph_phase_end(SphCtx* c, int ph, hipStream_t on);

struct PhaseScope {
    SphCtx* c;
    int ph;
    PhaseScope(SphCtx* c_, int ph_) : c(c_), ph(ph_) { sph_phase_begin(c, ph); }
    ~PhaseScope() { sph_phase_end(c, ph); }
};

// sticky error bits of asynchronous steps (SphCtx::d_err)
enum { SPH_ERR_HALO_OVERFLOW = 1, SPH_ERR_MSG_BAD = 2, SPH_ERR_CAPACITY = 4, SPH_ERR_OUT_OF_SLAB = 8, SPH_ERR_DEBUG_BOUNDS = 16 };

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

constexpr int PAIR_ROW_PAD = 3;       // sentinel records behind every candidate row of a tile's LDS image (pairs.hip)
constexpr int FORCE_MAX_RANGE = 256;  // stencil range (LDS records) one pass of the force kernels handles: 8 mask words / byte queue
// Longest 3x3 stencil range of the LDS path: 14 mask words per particle.  A tile whose longest range is above
// FORCE_MAX_RANGE ("long": 29-48 particles per cell) is walked by k_force in two passes over the mask words; without
// usable masks it goes the way of the dense sub-tiles.  (14, not 16: the list builder packs the word count in 4 bits.)
constexpr int FORCE_LONG_RANGE = 448;
constexpr int PAIR_MASK_WORDS_MAX = FORCE_LONG_RANGE / 32;

// ---- tile list of the pair kernels ----------------------------------------------------------------
// One entry = one workgroup's work: the particles of rows [cy0, cy1) of grid column `col` (one contiguous slot
// range, the keys are column-major) and their neighbour candidates, rows [cy0-1, cy1+1) of columns col-1..col+1
// (one contiguous slot range per column).  Everything a pair kernel needs to start its global loads is in the
// entry, so the loads of the particle data do not wait for a round trip through the cell table first.
struct __align__(16) TileDesc {
    int col, cy0, cy1;
    int info;            // bits 0..7: mask words the longest stencil range of the tile needs (<= PAIR_MASK_WORDS_MAX); bit 8: dense (see below)
    int own_lo, own_hi;  // slots of the own particles
    int c_lo[3], c_hi[3];  // slots of the candidates in columns col-1, col, col+1 (empty outside the grid)
};
static_assert(sizeof(TileDesc) == 48, "TileDesc");
constexpr int TILE_DENSE = 1 << 8;
constexpr int SPH_LONG_COUNT_OFFSET = 7;  // the count of long entries sits 7 ints behind the entry count (counters[63])
constexpr int TILE_FAST_ROWS = 16;  // base tiles up to this height are checked with one batch of loads (sph_tile_list_thread)

// The list is built on the device right after the cell table (tail of the grid build).  Thread i looks at one BASE
// tile -- `tile_cells` rows of one column, column-major order -- and cuts it into entries such that every entry
//   * holds at most `max_own` own particles (one lane each, a single pass over the workgroup),
//   * stages at most `cap` LDS records (candidates + PAIR_ROW_PAD sentinels per row),
//   * has no stencil range longer than FORCE_LONG_RANGE,
// skipping rows without particles at its ends; a single row that violates one of these becomes an entry of its own
// with TILE_DENSE set (the dense sub-tile kernels take those).  Base tiles that fit as a whole -- the normal case,
// the host adapts `tile_cells` to the scene -- give one entry.  A wave appends its entries with one atomic.
struct TileListArgs {
    int tile_cells, tiles_per_col, col0, ncols, nthreads, cap, max_own, list_cap;
    TileDesc* list;
    int* count;
};

struct TileRow {  // one grid row of a base tile's neighbourhood
    int own;      // particles in (col, y)
    int cand;     // particles in (col-1..col+1, y) + PAIR_ROW_PAD; 0 outside the grid
};

__device__ __forceinline__ TileRow sph_tile_row(const int* __restrict__ cell_start, int nx, int ny, int col, int y) {
    TileRow r = {0, 0};
    if (y < 0 || y >= ny) return r;
    const int k = col * ny + y;
    r.own = cell_start[k + 1] - cell_start[k];
    r.cand = r.own + PAIR_ROW_PAD;
    if (col > 0) r.cand += cell_start[k - ny + 1] - cell_start[k - ny];
    if (col + 1 < nx) r.cand += cell_start[k + ny + 1] - cell_start[k + ny];
    return r;
}

__device__ __forceinline__ TileDesc sph_tile_desc(const int* __restrict__ cell_start, int nx, int ny, int col, int a, int b,
                                                  int info) {
    TileDesc d;
    d.col = col;
    d.cy0 = a;
    d.cy1 = b;
    d.info = info;
    d.own_lo = cell_start[col * ny + a];
    d.own_hi = cell_start[col * ny + b];
    const int ylo = a > 0 ? a - 1 : 0, yhi = b < ny ? b + 1 : ny;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int cc = col - 1 + k;
        const bool in = cc >= 0 && cc < nx;
        d.c_lo[k] = in ? cell_start[cc * ny + ylo] : 0;
        d.c_hi[k] = in ? cell_start[cc * ny + yhi] : 0;
    }
    return d;
}

// Cutting a base tile into entries, one row at a time (rows in ascending order): the state of the entry being grown
// and the step that looks at row y given the candidate counts of rows y-1, y, y+1.  emit(a, b, info) is called for
// every finished entry [a, b).
struct TileCut {
    int a = -1;     // first row of the entry being grown (-1: none)
    int last = -1;  // its last row with particles
    int own = 0;    // own particles in [a, last]
    int total = 0;  // LDS records of rows [a-1, y-1] (y: the row being looked at)
    int nw = 0;     // mask words of the longest stencil range among rows [a, last]
    int n_out = 0;
    template <class Emit>
    __device__ __forceinline__ void row(int y, int prev_cand, int cur_own, int cur_cand, int next_cand, int cap, int max_own,
                                        Emit&& emit) {
        const int len = prev_cand + cur_cand + next_cand - PAIR_ROW_PAD;  // stencil range of the particles of row y
        const bool row_fits = cur_own <= max_own && prev_cand + cur_cand + next_cand <= cap && len <= FORCE_LONG_RANGE;
        if (cur_own > 0) {
            // can row y join the entry [a, last]?  (rows between last and y are empty but are staged as well)
            if (a >= 0 && (!row_fits || own + cur_own > max_own || total + cur_cand + next_cand > cap)) {
                emit(a, last + 1, nw);
                n_out++;
                a = -1;
            }
            if (!row_fits) {
                emit(y, y + 1, TILE_DENSE);
                n_out++;
            } else {
                if (a < 0) {
                    a = y;
                    own = 0;
                    nw = 0;
                    total = prev_cand;
                }
                own += cur_own;
                last = y;
                nw = max(nw, (len + 31) >> 5);
            }
        }
        if (a >= 0) total += cur_cand;  // rows [a-1, y] are staged if the entry goes on beyond y
    }
    template <class Emit>
    __device__ __forceinline__ int finish(Emit&& emit) {
        if (a >= 0) {
            emit(a, last + 1, nw);
            n_out++;
        }
        return n_out;
    }
};

// Walks the rows of base tile (col, rows [cy0, cy1)) reading the cell table row by row (one dependent memory round
// trip per row: only for base tiles higher than TILE_FAST_ROWS).  Returns the number of entries.
template <class Emit>
__device__ __forceinline__ int sph_tile_walk(const int* __restrict__ cell_start, int nx, int ny, int col, int cy0, int cy1,
                                             int cap, int max_own, Emit&& emit) {
    TileCut cut;
    TileRow prev = sph_tile_row(cell_start, nx, ny, col, cy0 - 1), cur = sph_tile_row(cell_start, nx, ny, col, cy0);
    for (int y = cy0; y < cy1; y++) {
        const TileRow next = sph_tile_row(cell_start, nx, ny, col, y + 1);
        cut.row(y, prev.cand, cur.own, cur.cand, next.cand, cap, max_own, emit);
        prev = cur;
        cur = next;
    }
    return cut.finish(emit);
}

// The same walk over per-row counts held in registers (row j <-> grid row cy0 - 1 + j; fully unrolled, static indices).
template <int R, class Emit>
__device__ __forceinline__ int sph_tile_walk_regs(const int (&own_r)[R + 2], const int (&cand)[R + 2], int rows, int cy0,
                                                  int cap, int max_own, Emit&& emit) {
    TileCut cut;
#pragma unroll
    for (int j = 1; j <= R; j++)
        if (j <= rows) cut.row(cy0 - 1 + j, cand[j - 1], own_r[j], cand[j], cand[j + 1], cap, max_own, emit);
    return cut.finish(emit);
}

__device__ __forceinline__ void sph_tile_list_thread(int i, const int* __restrict__ cell_start, int nx, int ny,
                                                     const TileListArgs& tl, int* __restrict__ err) {
    // Column-major: consecutive lanes take vertically adjacent base tiles of one column, whose cell table entries
    // are contiguous (a wave's loads below touch ~30 cache lines each instead of 64 with one column per lane).
    // The list inherits that order: an XCD's chunk of it is a block of adjacent columns.
    const int trow = i % tl.tiles_per_col, col = tl.col0 + i / tl.tiles_per_col;
    int k = 0, cy0 = 0, cy1 = 0;
    TileDesc first;
    int nlong = 0;                           // entries with a stencil range above FORCE_MAX_RANGE (not dense)
    unsigned cut_starts = 0, cut_ends = 0;  // a cut base tile (fast path): first / last row of every entry
    unsigned long long cut_info = 0;        // ... 4 bits per first row: mask words, 15 = dense
    if (i < tl.nthreads && col < tl.col0 + tl.ncols) {
        cy0 = trow * tl.tile_cells;
        cy1 = min(cy0 + tl.tile_cells, ny);
        bool walk = true;
        // (an empty base tile -- three quarters of a dam break's grid, nearly all of the last slab's in a wide domain --
        // is recognised from the two ends of its slot range before anything else is loaded)
        if (cell_start[col * ny + cy1] == cell_start[col * ny + cy0]) {
            walk = false;
        } else if (tl.tile_cells <= TILE_FAST_ROWS) {
            // Fast path (the normal case: the base tile fits as a whole).  All boundaries of the tile's neighbourhood
            // are fetched with independent loads -- one memory round trip -- and the three limits are checked in
            // registers; only a base tile that has to be cut takes the row-by-row walk below.
            const int rows = cy1 - cy0;
            int b[3][TILE_FAST_ROWS + 3];  // b[k][j] = cell_start of (col-1+k, cy0-1+j), rows clamped to [0, ny]
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                const int cc = col - 1 + kk;
                const bool in = cc >= 0 && cc < nx;
#pragma unroll
                for (int j = 0; j < TILE_FAST_ROWS + 3; j++) {
                    const int y = min(max(cy0 - 1 + j, 0), ny);
                    b[kk][j] = (in && j <= rows + 2) ? cell_start[cc * ny + y] : 0;
                }
            }
            int cand[TILE_FAST_ROWS + 2], own_r[TILE_FAST_ROWS + 2];  // row j <-> y = cy0-1+j
            int own = 0, total = 0;
#pragma unroll
            for (int j = 0; j < TILE_FAST_ROWS + 2; j++) {
                const int y = cy0 - 1 + j;
                const bool row_in = j <= rows + 1 && y >= 0 && y < ny;
                own_r[j] = (j >= 1 && j <= rows) ? b[1][j + 1] - b[1][j] : 0;
                cand[j] = row_in ? (b[0][j + 1] - b[0][j]) + (b[1][j + 1] - b[1][j]) + (b[2][j + 1] - b[2][j]) + PAIR_ROW_PAD : 0;
                own += own_r[j];
                total += cand[j];
            }
            int maxlen = 0;
#pragma unroll
            for (int j = 1; j <= TILE_FAST_ROWS; j++)
                if (own_r[j] > 0) maxlen = max(maxlen, cand[j - 1] + cand[j] + cand[j + 1] - PAIR_ROW_PAD);
            walk = false;
            if (own == 0) {
                // empty base tile: no entry
            } else if (!(own <= tl.max_own && total <= tl.cap && maxlen <= FORCE_LONG_RANGE)) {
                // to be cut: the cuts are noted in bit masks over the tile's rows (written out below)
                k = sph_tile_walk_regs<TILE_FAST_ROWS>(own_r, cand, rows, cy0, tl.cap, tl.max_own, [&](int a, int e, int info) {
                    cut_starts |= 1u << (a - cy0);
                    cut_ends |= 1u << (e - 1 - cy0);
                    cut_info |= (unsigned long long)((info & TILE_DENSE) ? 15 : info) << (4 * (a - cy0));
                    nlong += (!(info & TILE_DENSE) && info > FORCE_MAX_RANGE / 32) ? 1 : 0;
                });
            } else {
                k = 1;
                first.col = col;
                first.cy0 = cy0;
                first.cy1 = cy1;
                first.info = (maxlen + 31) >> 5;
                nlong = first.info > FORCE_MAX_RANGE / 32 ? 1 : 0;
                first.own_lo = b[1][1];
                int own_hi = 0, hi0 = 0, hi1 = 0, hi2 = 0;
#pragma unroll
                for (int j = 0; j < TILE_FAST_ROWS + 3; j++) {
                    if (j == rows + 1) own_hi = b[1][j];
                    if (j == rows + 2) {  // (row cy1 + 1, clamped to ny: the end of the candidate rows)
                        hi0 = b[0][j];
                        hi1 = b[1][j];
                        hi2 = b[2][j];
                    }
                }
                first.own_hi = own_hi;
                first.c_lo[0] = b[0][0];
                first.c_lo[1] = b[1][0];
                first.c_lo[2] = b[2][0];
                first.c_hi[0] = hi0;
                first.c_hi[1] = hi1;
                first.c_hi[2] = hi2;
            }
        }
        if (walk && !(cell_start[col * ny + cy1] > cell_start[col * ny + cy0])) walk = false;  // empty base tile
        if (walk) {
            bool have_first = false;  // (only the first entry is kept in this pass)
            k = sph_tile_walk(cell_start, nx, ny, col, cy0, cy1, tl.cap, tl.max_own, [&](int a, int b, int info) {
                if (!have_first) first = sph_tile_desc(cell_start, nx, ny, col, a, b, info);
                have_first = true;
                nlong += (!(info & TILE_DENSE) && info > FORCE_MAX_RANGE / 32) ? 1 : 0;
            });
        }
    }
    // exclusive prefix of k over the wave, one atomic for the wave's total
    const int lane = sph_lane_id();
    int incl = k;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    const int wave_total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && wave_total) base = atomicAdd(tl.count, wave_total);
    base = __shfl(base, 63, 64) + incl - k;
    if (k == 0) return;
    // (how many entries are "long": the host picks the force kernel that handles them in place only while there are any)
    if (nlong) atomicAdd(tl.count + SPH_LONG_COUNT_OFFSET, nlong);
    if (base + k > tl.list_cap) {  // (cannot happen: the list holds an entry per particle)
        atomicOr(err, SPH_ERR_CAPACITY);
        return;
    }
    if (cut_starts) {  // a base tile that was cut (fast path): one entry per noted cut
        for (int j = 0; cut_starts; j++) {
            const int ra = __ffs(cut_starts) - 1, re = __ffs(cut_ends) - 1;
            cut_starts &= cut_starts - 1;
            cut_ends &= cut_ends - 1;
            const int inf4 = (int)((cut_info >> (4 * ra)) & 15);
            tl.list[base + j] = sph_tile_desc(cell_start, nx, ny, col, cy0 + ra, cy0 + re + 1, inf4 == 15 ? TILE_DENSE : inf4);
        }
    } else if (k == 1) {
        tl.list[base] = first;
    } else {  // a base tile higher than TILE_FAST_ROWS that was cut: walk it again and write the entries
        int j = 0;
        sph_tile_walk(cell_start, nx, ny, col, cy0, cy1, tl.cap, tl.max_own, [&](int a, int b, int info) {
            tl.list[base + j] = sph_tile_desc(cell_start, nx, ny, col, a, b, info);
            j++;
        });
    }
}


constexpr int SPH_TILE_COUNT_SLOT = 56;  // SphCtx::counters[56]: number of listed tiles (counters[63]: the long ones among them)
constexpr int SPH_N_SORTED_SLOT = 57;    // counters[57]: particle count after the last grid build
constexpr int SPH_MSG_COUNT_SLOT = 58;   // counters[58..59]: record counts of the halo messages being packed
constexpr int SPH_DENSE_COUNT_SLOT = 61; // counters[61]: entries of the dense sub-tile list
constexpr int SPH_BIG_COUNT_SLOT = 62;   // counters[62]: cells listed for k_rank_big (more than RANK_BIG_MIN particles)
constexpr int RANK_BIG_MIN = 2048;       // cells above this population are ranked by sorting their ids in LDS
constexpr int RANK_BIG_MAX = 32768;      // ... in chunks of this many ids (128 KB of LDS)
constexpr int RANK_BIG_THREADS = 1024;
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
    tl.nthreads = (hi - lo) * tl.tiles_per_col;
    tl.cap = c->pair_cap;
    tl.max_own = c->pair_threads;
    tl.list_cap = c->tile_list_cap;
    tl.list = c->tile_list;
    tl.count = c->counters + SPH_TILE_COUNT_SLOT;
}

// The tile list in c->tile_list is current for this grid if it was built with these settings.
static inline bool sph_tiles_current(const SphCtx* c, int lo, int hi) {
    return c->tiles_valid && c->tiles_cells == c->tile_cells && c->tiles_lo == lo && c->tiles_hi == hi &&
           c->tiles_cap == c->pair_cap && c->tiles_threads == c->pair_threads;
}
static inline void sph_tiles_mark(SphCtx* c, int lo, int hi) {
    c->tiles_valid = true;
    c->tiles_cells = c->tile_cells;
    c->tiles_lo = lo;
    c->tiles_hi = hi;
    c->tiles_cap = c->pair_cap;
    c->tiles_threads = c->pair_threads;
}
