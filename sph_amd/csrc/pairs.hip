// pairs.hip -- density/pressure and force pair sums over the cell list, plus integrate, for gfx950.
//
// Arithmetic follows the reference's serial step expression by expression in IEEE binary32
// (built with -ffp-contract=off; hipcc's `/` and sqrtf are correctly rounded by default):
//   density/pressure  serial_computeDensityPressure  SPH/sph.cpp:41-55  (kernels :7-12)
//   forces            serial_computeForces           SPH/sph.cpp:57-99  (kernels :15-39)
//   integrate         serial_integrate               SPH/sph.cpp:101-130
// The only difference from the serial loop is the one every parallel variant of the reference has
// (SPH/MPI_sph.cpp:144-147,181): neighbour velocities come from the state before the force pass.
// The j-sum of each particle runs sequentially in the canonical neighbour order (cell rows cy-1, cy,
// cy+1; inside a row the cells of columns cx-1, cx, cx+1; inside a cell ascending slot), which is
// also the order of oracle/sph_oracle.c's cell-list functions: the two agree bit for bit.
//
// Work decomposition: one workgroup per tile = `tile_cells` consecutive cells of one grid column.
// With column-major keys the tile's particles are one contiguous slot range (coalesced SoA loads);
// its neighbour candidates (three columns, rows +-1) are staged once in LDS in
// row-major cell order, which makes every particle's 3x3 stencil one contiguous LDS range.  Lanes of
// one cell scan the same LDS addresses (broadcast).  If a tile's candidates exceed the LDS budget
// the same sums read global memory instead.
// No MFMA: the pair sums are irregular, distance-predicated gathers, not a contraction.
#include "sphhip_internal.h"
#include "exact_div.h"

// Diagnostic build only (-DSPH_STAMPS, tools/stamps.py): every wave of k_force writes s_memtime stamps of its
// tile phases into g_stamps (8 words per wave and tile-list entry).  Not compiled into the product library.
#ifdef SPH_STAMPS
constexpr int STAMP_TILES = 1 << 16;
__device__ unsigned long long g_stamps[STAMP_TILES * 4 * 8];
#define STAMP_PARAM , int stamp_slot
#define STAMP_ARG , stamp_slot
#define STAMP(k)                                                                                              \
    do {                                                                                                      \
        if (stamp_slot >= 0 && (tix & 63) == 0)                                                               \
            g_stamps[((size_t)stamp_slot * 4 + (tix >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime();         \
    } while (0)
#define STAMP_VAL(k, v)                                                                                       \
    do {                                                                                                      \
        if (stamp_slot >= 0 && (tix & 63) == 0) g_stamps[((size_t)stamp_slot * 4 + (tix >> 6)) * 8 + (k)] = (v); \
    } while (0)
extern "C" int sphhip_debug_read_stamps(unsigned long long* out, int tiles) {
    if (tiles > STAMP_TILES) tiles = STAMP_TILES;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)tiles * 32) == hipSuccess ? 0 : -2;
}
#else
#define STAMP_PARAM
#define STAMP_ARG
#define STAMP(k) do { } while (0)
#define STAMP_VAL(k, v) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Tile = `tile_cells` consecutive cells [cy0, cy1) of grid column `col`.  Its neighbour candidates
// are the cells of rows [ylo, yhi) in columns col-1, col, col+1.  They are staged in LDS in
// ROW-MAJOR cell order (row y: cells (col-1,y), (col,y), (col+1,y); ascending slot inside a cell),
// so the 3x3 stencil of a particle in row cy is ONE contiguous LDS range: rows cy-1..cy+1.
// That order -- rows cy-1, cy, cy+1; inside a row columns cx-1, cx, cx+1; inside a cell ascending
// slot -- is the canonical summation order, shared with oracle/sph_oracle.c (orc_stencil_ranges).
//
// All LDS of the pair kernels is one dynamic array (16-byte aligned base, no static __shared__ in
// front of it): [32 * cap bytes of staged records: PairRec rec[cap] in the dense kernels; hot[cap] (16 B), vel[cap] (8 B),
// prs[cap] (4 B) in k_force, see rec_load] (force) or [float2 pos[cap]] (density) [int off[MAXC+1]] [int gs[MAXC]]
// [int wave_tmp[8]] [u8 queue[QCAP*256] | u32 masks[8*256]] (force only).
// ---------------------------------------------------------------------------------------------
constexpr int PAIR_MAXC = 3 * (PAIR_MAX_TILE_CELLS + 2);  // candidate cells of a tile (<= 256)
#ifndef FORCE_Q
#define FORCE_Q 8  // in-range neighbours gathered per round (scan path only)
#endif
#ifndef FORCE_SCAN_UNROLL
#define FORCE_SCAN_UNROLL 4
#endif
#ifndef FORCE_HEAVY_UNROLL
#define FORCE_HEAVY_UNROLL 2  // pairs in flight per lane (2: 5 % faster than 4 with the mask path -- less padding of the last group)
#endif
constexpr int FORCE_QCAP = FORCE_Q + FORCE_SCAN_UNROLL + FORCE_HEAVY_UNROLL;  // pushes overshoot by < SCAN_UNROLL, reads by < HEAVY_UNROLL
// After the three cells of every candidate row the LDS image holds PAIR_ROW_PAD sentinel records at
// +inf: a lane may read up to PAIR_ROW_PAD records past the end of its range without a bounds test
// (r2 = +inf fails the distance test); the sentinels inside its range fail the same way.
// (PAIR_ROW_PAD, and FORCE_MAX_RANGE -- the longest 3x3 stencil range the masks / the byte queue can index -- are in
// sphhip_internal.h: the tile list builder needs them)
constexpr int PAIR_STAGE_BATCH = 4;  // candidates per thread whose global loads are issued up front (4 x 256 >= the default LDS budget)
constexpr int PAIR_MASK_WORDS = FORCE_MAX_RANGE / 32;  // in-range bit masks handed from k_density to k_force
static_assert(FORCE_SCAN_UNROLL - 1 <= PAIR_ROW_PAD, "scan unroll reads past the range end");

// One staged neighbour candidate of the dense force kernels: 32 bytes, so that one shifted index addresses both halves
// (k_force itself keeps the same fields in three arrays: rec_load / rec_store below).
struct __align__(16) PairRec {
    float4 aux;  // vx, vy, rho, P
    float2 pos;  // +inf for sentinels and invalid particles
    float yd;    // refined 1 / rho (sph_rcp_refined), used by FAST tiles
    float pad;
};

// Two LDS layouts of the staged records.  !SPLIT: an array of PairRec (the dense kernels).  SPLIT (the tuned force
// kernel): three arrays, hot[cap] = {x, y, 1/rho~, rho} (16 bytes: everything the distance, the root and the
// divisions by rho_j need), vel[cap] = {vx, vy} (8 bytes) and prs[cap] (4 bytes, not read at all on the zero-pressure
// path); `hot_bytes` = 16 * cap.  A popped candidate is read at a lane-varying record index: with 32-byte records
// either of the two 16-byte loads could only ever touch every other 16-byte bank group, a two-way conflict built
// into the layout (SQ_LDS_BANK_CONFLICT was 65 % of the LDS cycles and the LDS pipe 70 % busy once the zero-pressure
// shortcut had thinned the arithmetic out); apart, consecutive records fall into consecutive bank groups.
// tools/ubench/lds_gather.hip, LDS cycles per wave and candidate at 4 workgroups per CU: 32-byte records 30.7, two
// arrays of 16-byte halves 22.6, 16 + 8 bytes 18.5, no conflicts at all 13.1.
struct RecVals {
    float4 aux;  // vx, vy, rho, P (P = 0 where it was not read)
    float2 pos;
    float yd;
};
template <bool SPLIT, bool NEED_P>
__device__ __forceinline__ RecVals rec_load(const PairRec* base, int r, int hot_bytes) {
    RecVals v;
    const char* b = reinterpret_cast<const char*>(base);
    if (SPLIT) {
        const float4 h = *reinterpret_cast<const float4*>(b + r * 16);
        const float2 w = *reinterpret_cast<const float2*>(b + hot_bytes + r * 8);
        const float p = NEED_P ? *reinterpret_cast<const float*>(b + hot_bytes + (hot_bytes >> 1) + r * 4) : 0.0f;
        v.aux = make_float4(w.x, w.y, h.w, p);
        v.pos = make_float2(h.x, h.y);
        v.yd = h.z;
    } else {
        const float4* q = reinterpret_cast<const float4*>(b + r * 32);
        v.aux = q[0];
        const float4 t = q[1];
        v.pos = make_float2(t.x, t.y);
        v.yd = t.z;
    }
    return v;
}
template <bool SPLIT>
__device__ __forceinline__ float2 rec_pos(const PairRec* base, int r) {
    const char* b = reinterpret_cast<const char*>(base);
    return *reinterpret_cast<const float2*>(b + (SPLIT ? r * 16 : r * 32 + 16));
}
template <bool SPLIT>
__device__ __forceinline__ void rec_store(PairRec* base, int r, int hot_bytes, float4 aux, float2 pos, float yd) {
    char* b = reinterpret_cast<char*>(base);
    if (SPLIT) {
        *reinterpret_cast<float4*>(b + r * 16) = make_float4(pos.x, pos.y, yd, aux.z);
        *reinterpret_cast<float2*>(b + hot_bytes + r * 8) = make_float2(aux.x, aux.y);
        *reinterpret_cast<float*>(b + hot_bytes + (hot_bytes >> 1) + r * 4) = aux.w;
    } else {
        *reinterpret_cast<float4*>(b + r * 32) = aux;
        *reinterpret_cast<float4*>(b + r * 32 + 16) = make_float4(pos.x, pos.y, yd, 0.0f);
    }
}

struct TileGeom {
    int col, cy0, cy1, ylo, yhi, own_lo, own_hi, ncells, total;
};

static size_t pair_lds_bytes(int cap, bool force, int threads) {
    size_t b = sizeof(int) * (2 * PAIR_MAXC + 1 + 8);  // off, gs, wave_tmp
    // (the scan path's byte queue and the mask path's mask words share one region)
    const size_t qbytes = sizeof(uint8_t) * FORCE_QCAP * threads, mbytes = sizeof(uint32_t) * PAIR_MASK_WORDS * threads;
    b += force ? sizeof(PairRec) * (size_t)cap + (qbytes > mbytes ? qbytes : mbytes) : sizeof(float2) * (size_t)cap;
    return b;
}

// Exclusive scan of one int per thread over the workgroup of THREADS threads (two barriers).
template <int THREADS>
__device__ __forceinline__ int wg_excl_scan(int v, int* lds_wave, int& total, int tix) {
    int lane = tix & 63, wave = tix >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        int x = lds_wave[w];
        off += (w < wave) ? x : 0;
        total += x;
    }
    __syncthreads();
    return off + incl - v;
}

__device__ __forceinline__ int cand_count(const int* s_off, int cc) {  // real particles of candidate cell cc
    return s_off[cc + 1] - s_off[cc] - ((cc % 3) == 2 ? PAIR_ROW_PAD : 0);
}

// Fills g, s_off[0..ncells] (LDS start of every candidate cell, row-major, PAIR_ROW_PAD sentinel slots
// after every third cell) and s_gs[0..ncells) (global slot start of every candidate cell).
// Returns false for an empty tile (uniform).

template <int THREADS>
__device__ __forceinline__ bool tile_setup(const DevConsts& c, const int* __restrict__ cell_start, int col, int cy0,
                                           int cy1, TileGeom& g, int* s_off, int* s_gs, int* s_wave, int tix) {
    g.col = col;
    g.cy0 = cy0;
    g.cy1 = cy1;
    g.ylo = max(g.cy0 - 1, 0);
    g.yhi = min(g.cy1 + 1, c.ny);
    // (all cell_start loads of the tile are issued together: one memory round trip; tiles come from the
    // list of non-empty tiles, so there is no early exit to wait for)
    g.own_lo = cell_start[col * c.ny + g.cy0];
    g.own_hi = cell_start[col * c.ny + g.cy1];
    g.ncells = 3 * (g.yhi - g.ylo);
    int t = tix;
    int cnt = 0, g0 = 0;
    if (t < g.ncells) {
        int row = t / 3, d = t - 3 * row;
        int cc = col - 1 + d;
        if (cc >= 0 && cc < c.nx) {
            int key = cc * c.ny + g.ylo + row;
            g0 = cell_start[key];
            cnt = cell_start[key + 1] - g0;
        }
    }
    int padded = cnt + ((t < g.ncells && (t % 3) == 2) ? PAIR_ROW_PAD : 0);
    int ex = wg_excl_scan<THREADS>(padded, s_wave, g.total, tix);
    if (t < g.ncells) {
        s_off[t] = ex;
        s_gs[t] = g0;
    }
    if (t == 0) s_off[g.ncells] = g.total;
    return g.own_lo != g.own_hi;
}

// ---------------------------------------------------------------------------------------------
// density + pressure + neighbour count.   Algorithmic HBM bytes / particle: R pos 8, W rho 4 + P 4
// (+ 4 for the neighbour count) = 16 (+4).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void density_acc(const DevConsts& c, float2 pi, float2 q, float& dens, int& cnt) {
    float rx = pi.x - q.x, ry = pi.y - q.y;
    float r2 = rx * rx + ry * ry;
    float term = c.h2 - r2;
    float w = c.mass * (c.poly6 * term * term * term);  // MASS * poly6(r2, H), sph.cpp:11,49
    bool in = r2 < c.h2;                                // sph.cpp:48
    dens += in ? w : 0.0f;                              // x + 0.0f == x: same bits as the branch
    cnt += in ? 1 : 0;
}

// The same sum without a compare or select (those cost twice an fp32 add on gfx950), and with the
// in-range bit shifted into m:
//   t = r2 - h2 = -(h2 - r2) exactly, so max(-t, 0) is the reference's (h2 - r2) when r2 < h2 and +0
//   otherwise (also for the +inf sentinels and for NaN); a term of mass * (poly6 * 0 * 0 * 0) = +0 leaves
//   the sum unchanged, like the reference's skipped iteration;
//   r2 < h2  <=>  t < 0  <=>  sign bit of t (t is never -0; for NaN positions the bit is unspecified --
//   k_force does not use the masks of a tile that holds a non-finite position, and the neighbour count
//   of such a scene is unspecified).
__device__ __forceinline__ void density_acc_mask(const DevConsts& c, float2 pi, float2 q, float& dens, uint32_t& m) {
    float rx = pi.x - q.x, ry = pi.y - q.y;
    float r2 = rx * rx + ry * ry;
    float t = r2 - c.h2;
    // (v_max is half rate; a subtract with the clamp output modifier instead made no difference: the loop is bound by
    // instruction issue, not by VALU cycles)
    float term = fmaxf(-t, 0.0f);
    dens += c.mass * (c.poly6 * term * term * term);  // MASS * poly6(r2, H), sph.cpp:11,49
    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);  // (m << 1) | sign(t)
}

// ---------------------------------------------------------------------------------------------
// Work distribution of the pair kernels: a list of entries (sphhip_internal.h: TileDesc), one workgroup pass each,
// built on the device right after the cell table (grid.hip: k_tile_list_build) in column-major order and split
// into 8 contiguous chunks, one per XCD (workgroup b runs on XCD b % 8 under round-robin dispatch): an XCD works
// on a block of adjacent columns, so horizontally adjacent tiles -- which share two of their three candidate
// columns -- are served by the same L2.  A dam break fills an eighth of the grid: launching one workgroup
// per tile of the whole grid spent a fifth of both pair kernels on empty workgroups.
// The number of listed tiles is only known on the device.  The host sizes the grid from the count a
// previous launch wrote into mapped host memory (one workgroup per tile, a little slack), or, while it
// has none, for one resident wave of workgroups; workgroups stride over their chunk, so any grid size
// is correct.  (Fully persistent workgroups were 20 % slower: they run in lock-step, every workgroup
// in its latency-bound tile prologue at the same time.)
// ---------------------------------------------------------------------------------------------
struct TileQueue {
    const TileDesc* list;  // the work list (sphhip_internal.h: TileDesc)
    const int* count;      // number of entries
    int* host_count;       // mapped host memory: the count and the tiling it belongs to, for sizing later launches
    int tile_cells;
};

// Chunk k = entries [k * ceil(total / 8), (k + 1) * ceil(total / 8)) of the list goes to XCD k: workgroup b (on XCD
// b & 7; the grid is a multiple of 8) takes entry (b >> 3) of chunk b & 7, then strides by gridDim / 8 INSIDE its
// chunk.  The chunks follow the list's true length, which only the device knows, not the grid the host chose from
// a count that is 32 to 64 steps old (sph_throttle): while the chunks were cut by gridDim, every entry beyond a too
// small grid fell to the first workgroups of chunk 0 -- when a settling dam break's tile count jumped from 4423 to
// 4676 between two steps, XCD 0 got 838 entries instead of 585 and both pair kernels ran 40 % longer until the news
// reached the host (tools/interval_probe.py: 0.228 instead of 0.193 ms per step over 50 steps), and a too large grid
// starved XCD 7.  Now any grid is balanced: too small costs a second entry for some workgroups of every XCD alike,
// too large costs workgroups that leave at once.
// The entry is read with wave-uniform (scalar) loads.  `tix` is threadIdx.x behind an opaque barrier: without it the
// compiler hoists ~20 VGPRs of thread-index arithmetic out of the (rarely repeated) loop and the force kernel drops
// from 4 to 3 workgroups per CU.
template <class Body>
__device__ __forceinline__ void for_each_tile(const TileQueue& q, Body&& body) {
    const int total = *q.count;
    const int per_chunk = (total + 7) >> 3;
    const int stride = max((int)(gridDim.x >> 3), 1);  // workgroups per chunk (pair_grid: a multiple of 8, >= 8)
    const int first = (int)(blockIdx.x & 7) * per_chunk;
    const int last = min(first + per_chunk, total);  // (< first for the chunks behind the end of a short list)
    int t = first + (int)(blockIdx.x >> 3);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ((volatile int*)q.host_count)[0] = total;
        ((volatile int*)q.host_count)[1] = q.tile_cells;
        ((volatile int*)q.host_count)[2] = q.count[SPH_LONG_COUNT_OFFSET];
    }
    while (t < last) {
        const TileDesc d = q.list[t];
        __syncthreads();  // the previous entry is done with the LDS image
        int tix = threadIdx.x;
        asm volatile("" : "+v"(tix));
        body(d, tix, t);
        t += stride;
    }
}

// Geometry of a list entry (no memory access).
__device__ __forceinline__ TileGeom entry_geom(const DevConsts& c, const TileDesc& d) {
    TileGeom g;
    g.col = d.col;
    g.cy0 = d.cy0;
    g.cy1 = d.cy1;
    g.ylo = max(d.cy0 - 1, 0);
    g.yhi = min(d.cy1 + 1, c.ny);
    g.own_lo = d.own_lo;
    g.own_hi = d.own_hi;
    g.ncells = 3 * (g.yhi - g.ylo);
    g.total = 0;
    return g;
}

// The candidate-cell table of an entry, in two halves so that the global loads of the first half are in flight
// together with everything else the prologue of a pair kernel loads:
//   entry_cells_load   thread t < ncells fetches the slot range [g0, g1) of candidate cell t (row-major: t = 3 row + d)
//   entry_cells_table  exclusive scan of the padded counts -> s_off[0..ncells], s_gs[0..ncells), g.total (one barrier
//                      inside for entries of more than 64 candidate cells; the caller's barrier publishes the table)
__device__ __forceinline__ void entry_cells_load(const DevConsts& c, const int* __restrict__ cell_start, const TileGeom& g,
                                                 int tix, int& g0, int& g1) {
    g0 = g1 = 0;
    if (tix < g.ncells) {
        const int row = tix / 3, dd = tix - 3 * row;
        const int cc = g.col - 1 + dd;
        if (cc >= 0 && cc < c.nx) {
            const int key = cc * c.ny + g.ylo + row;
            g0 = cell_start[key];
            g1 = cell_start[key + 1];
        }
    }
}

template <int THREADS>
__device__ __forceinline__ void entry_cells_table(TileGeom& g, int g0, int g1, int* s_off, int* s_gs, int* s_wave, int tix) {
    const int lane = tix & 63, wave = tix >> 6;
    const bool mine = tix < g.ncells;
    const int padded = mine ? (g1 - g0) + ((tix % 3) == 2 ? PAIR_ROW_PAD : 0) : 0;
    int incl = padded;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const int t = __shfl_up(incl, k, 64);
        if (lane >= k) incl += t;
    }
    int before = 0;
    if (g.ncells > 64) {  // (uniform; tiles of more than 19 rows)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < THREADS / 64; w++) before += (w < wave) ? s_wave[w] : 0;
    }
    if (mine) {
        s_off[tix] = before + incl - padded;
        s_gs[tix] = g0;
        if (tix == g.ncells - 1) s_off[g.ncells] = before + incl;
    }
}

// Candidate i of an entry's concatenated candidate sequence (column col-1, then col, then col+1; ascending slot):
// its global slot and its column 0..2.
__device__ __forceinline__ int entry_cand_slot(const TileDesc& d, int i, int& dd) {
    const int n0 = d.c_hi[0] - d.c_lo[0], n1 = d.c_hi[1] - d.c_lo[1];
    dd = (i >= n0 ? 1 : 0) + (i >= n0 + n1 ? 1 : 0);
    return i + (dd == 0 ? d.c_lo[0] : dd == 1 ? d.c_lo[1] - n0 : d.c_lo[2] - n0 - n1);
}
__device__ __forceinline__ int entry_cand_count(const TileDesc& d) {
    return (d.c_hi[0] - d.c_lo[0]) + (d.c_hi[1] - d.c_lo[1]) + (d.c_hi[2] - d.c_lo[2]);
}
// LDS slot of the candidate at global slot `slot`, column dd, cell key `key` (clamped: a stale key must not become
// an address outside the image).
__device__ __forceinline__ int entry_cand_dst(const DevConsts& c, const TileGeom& g, const int* s_off, const int* s_gs,
                                              int slot, int dd, uint32_t key, int cap) {
    int cc = 3 * ((int)key - (g.col - 1 + dd) * c.ny - g.ylo) + dd;
    cc = min(max(cc, 0), g.ncells - 1);
    const int dst = s_off[cc] + (slot - s_gs[cc]);
    return min(max(dst, 0), cap - 1);
}

// ---------------------------------------------------------------------------------------------
// Dense sub-tiles.  A sub-tile the LDS path cannot take -- its candidates exceed the budget even at one
// row of cells, or (force pass) a stencil range is too long for the masks / byte queue: ~30+ particles
// per cell, e.g. the reference's fixed 1.9 x 1.9 box with tens of thousands of particles -- is not
// processed by the tile's workgroup.  It is appended to a list, one entry per DENSE_THREADS own
// particles, and k_density_dense / k_force_dense, launched right after, give every entry a workgroup
// of its own: the dense code stays out of the tuned kernels, and dense regions spread over the GPU
// instead of serialising in the few workgroups whose tiles they fall into.
// Whether the follow-up kernel is needed is only known on the device.  The host launches it (and
// passes defer = 1) unless the last dense kernel reported an empty list through mapped host memory;
// with defer = 0 a workgroup that does meet a dense sub-tile evaluates it in place from global memory
// (slow, correct) and raises the mapped flag, so that a later launch defers again.
// ---------------------------------------------------------------------------------------------
constexpr int DENSE_THREADS = 256;
constexpr int DENSE_U = 8, DENSE_Q = 24;  // pairs in flight / queued per round in k_force_dense
// its byte queue (rows of one byte per thread) lives in the mask region of the force kernels' LDS layout: a
// scan trip pushes at most FORCE_SCAN_UNROLL entries past DENSE_Q - 1, the evaluation reads whole groups of DENSE_U
static_assert((DENSE_Q - 1 + FORCE_SCAN_UNROLL + DENSE_U - 1) / DENSE_U * DENSE_U <= (int)sizeof(uint32_t) * PAIR_MASK_WORDS,
              "k_force_dense: queue rows exceed the LDS region");

// Where a pair kernel puts the dense sub-tiles it meets.  The five fields that never change live in device memory
// (SphCtx::d_sinks, one per pass) and are read only by the workgroup that has something to append: as kernel arguments
// they occupied eleven scalar registers for the whole kernel, and k_force has none to spare -- without them it has 35
// fewer SGPR reloads in its code and runs 3.5 % faster, k_density 3 % (A/B at 1 M and 16 M particles).
struct DenseSink {
    const DenseSinkDev* dev;
    int mode;  // bit 0: the follow-up kernel is scheduled (append); bits 1..: own particles per entry
};

#ifdef SPH_DENSE_DEFER_NOINLINE
// Experiment ("out-of-line dense hand-off"): the append as a real function call with scalar
// arguments, every index checked before use (a bad one raises SPH_ERR_DEBUG_BOUNDS instead of touching memory).
__device__ __noinline__ void dense_defer_call(int4* list, int* count, int capacity, int* host_flag, int* err, int defer,
                                              int part, int col, int cy0, int rows, int own, int nx, int ny) {
    if (!defer) {
        *(volatile int*)host_flag = 1;
        return;
    }
    if (part <= 0 || own < 0 || col < 0 || col >= nx || cy0 < 0 || rows <= 0 || cy0 + rows > ny) {
        atomicOr(err, SPH_ERR_DEBUG_BOUNDS);
        return;
    }
    const int nparts = (own + part - 1) / part;
    const int e = atomicAdd(count, nparts);
    if (e < 0) {
        atomicOr(err, SPH_ERR_DEBUG_BOUNDS);
    } else if (e + nparts <= capacity) {
        for (int p = 0; p < nparts; p++) list[e + p] = make_int4(col, cy0, rows, p);
    } else {
        atomicOr(err, SPH_ERR_CAPACITY);
    }
}
#endif

// true: the sub-tile went to the list (uniform over the workgroup).
__device__ __forceinline__ bool dense_defer(const DenseSink& d, const TileGeom& g, int tix, int nx = 1 << 30, int ny = 1 << 30) {
    const int defer = d.mode & 1, part = d.mode >> 1;
#ifdef SPH_DENSE_DEFER_NOINLINE
    if (tix == 0) {
        const DenseSinkDev s = *d.dev;
        dense_defer_call(s.list, s.count, s.capacity, s.host_flag, s.err, defer, part, g.col, g.cy0, g.cy1 - g.cy0,
                         g.own_hi - g.own_lo, nx, ny);
    }
    return defer != 0;
#endif
    if (tix == 0) {
        const DenseSinkDev s = *d.dev;
        if (!defer) {
            *(volatile int*)s.host_flag = 1;
        } else {
            const int nparts = (g.own_hi - g.own_lo + part - 1) / part;
            const int e = atomicAdd(s.count, nparts);
            if (e + nparts <= s.capacity) {
                for (int p = 0; p < nparts; p++) s.list[e + p] = make_int4(g.col, g.cy0, g.cy1 - g.cy0, p);
            } else {
                atomicOr(s.err, SPH_ERR_CAPACITY);
            }
        }
    }
    return defer != 0;
}

// One list entry of the density pass.  Prologue: every global load it needs -- candidate-cell table, own particles,
// candidate positions and keys (contiguous slot ranges straight from the entry, PAIR_STAGE_BATCH per thread) -- is
// issued before the first one is waited for: one memory round trip and two barriers, where walking from the cell
// table to the particle data took three dependent round trips (stamps: 9 of 24 us per tile).
template <int THREADS>
__device__ __forceinline__ void density_entry(const DevConsts& c, const float2* __restrict__ pos,
                                              const uint32_t* __restrict__ skeys, const int* __restrict__ cell_start,
                                              float* __restrict__ rho, float* __restrict__ prs,
                                              int* __restrict__ ncount, uint32_t* __restrict__ nmask, uint32_t* __restrict__ nmask_hi, int n_max,
                                              const TileDesc& d, int cap, const DenseSink& sink, float2* s_pos,
                                              int* s_off, int* s_gs, int* s_wave, int tix) {
    TileGeom g = entry_geom(c, d);
    const bool use_lds = (d.info & TILE_DENSE) == 0;
    if (!use_lds && dense_defer(sink, g, tix, c.nx, c.ny)) return;
    int g0, g1;
    entry_cells_load(c, cell_start, g, tix, g0, g1);
    float2 pi0 = make_float2(0.0f, 0.0f);
    uint32_t key0 = 0;
    if (g.own_lo + tix < g.own_hi) {
        pi0 = pos[g.own_lo + tix];
        key0 = skeys[g.own_lo + tix];
    }
    const int n_cand = use_lds ? entry_cand_count(d) : 0;
    float2 cq[PAIR_STAGE_BATCH];
    uint32_t ck[PAIR_STAGE_BATCH];
    int cslot[PAIR_STAGE_BATCH], cdd[PAIR_STAGE_BATCH];
#pragma unroll
    for (int b = 0; b < PAIR_STAGE_BATCH; b++) {
        const int i = tix + b * THREADS;
        cslot[b] = -1;
        cdd[b] = 0;
        if (i < n_cand) {
            cslot[b] = entry_cand_slot(d, i, cdd[b]);
            cq[b] = pos[cslot[b]];
            ck[b] = skeys[cslot[b]];
        }
    }
    entry_cells_table<THREADS>(g, g0, g1, s_off, s_gs, s_wave, tix);
    __syncthreads();
    if (use_lds) {
#pragma unroll
        for (int b = 0; b < PAIR_STAGE_BATCH; b++)
            if (cslot[b] >= 0) s_pos[entry_cand_dst(c, g, s_off, s_gs, cslot[b], cdd[b], ck[b], cap)] = cq[b];
        for (int i = tix + PAIR_STAGE_BATCH * THREADS; i < n_cand; i += THREADS) {  // (LDS budgets beyond 4 x THREADS)
            int dd;
            const int slot = entry_cand_slot(d, i, dd);
            s_pos[entry_cand_dst(c, g, s_off, s_gs, slot, dd, skeys[slot], cap)] = pos[slot];
        }
        for (int t = tix; t < (g.ncells / 3) * PAIR_ROW_PAD; t += THREADS)
            s_pos[s_off[3 * (t / PAIR_ROW_PAD) + 3] - PAIR_ROW_PAD + (t % PAIR_ROW_PAD)] =
                make_float2(__builtin_inff(), __builtin_inff());
        __syncthreads();
    }
    // (one trip per lane on the LDS path: an entry holds at most THREADS own particles)
    for (int s = g.own_lo + tix; s < g.own_hi; s += THREADS) {
        const bool first = s == g.own_lo + tix;
        const float2 pi = first ? pi0 : pos[s];
        const uint32_t key = first ? key0 : skeys[s];
        const int cy = (int)key - g.col * c.ny;
        const int r0 = max(cy - 1, 0) - g.ylo, r1 = min(cy + 1, c.ny - 1) - g.ylo;
        float dens = 0.0f;
        int cnt = 0;
        if (use_lds) {
            // Four candidates per trip (the row sentinels absorb the overshoot).  The in-range tests are
            // also shifted into one bit mask per 32 candidates, first candidate in the top bit, and
            // stored for the force pass (k_force), which then needs no distance scan of its own.
            // (Measured dead end: no branch per group -- lanes past their range reading a block of sentinels
            // instead -- was 7 % slower.)
            const int lo = s_off[3 * r0], hi = s_off[3 * r1 + 3] - PAIR_ROW_PAD;
            int j = lo;
            for (int w = 0; j < hi; w++) {
                uint32_t m = 0;
#pragma unroll
                for (int grp = 0; grp < 8; grp++) {
                    if (j < hi) {
                        float2 q[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) q[u] = s_pos[j + u];
#pragma unroll
                        for (int u = 0; u < 4; u++) density_acc_mask(c, pi, q[u], dens, m);
                        j += 4;
                    } else {
                        m <<= 4;
                    }
                }
                cnt += __popc(m);
                if (w < PAIR_MASK_WORDS)
                    nmask[(size_t)w * n_max + s] = m;
                else if (w < PAIR_MASK_WORDS_MAX)  // (long stencil ranges: the words beyond the first 8 have an array of their own)
                    nmask_hi[(size_t)(w - PAIR_MASK_WORDS) * n_max + s] = m;
            }
        } else {
            for (int cc = 3 * r0; cc < 3 * r1 + 3; cc++) {
                int gs = s_gs[cc], n = cand_count(s_off, cc);
                for (int j = gs; j < gs + n; j++) density_acc(c, pi, pos[j], dens, cnt);
            }
        }
        float pr = c.stiffness * (dens - c.rest_density);  // sph.cpp:52-53
        if (pr < 0.0f) pr = 0.0f;
        rho[s] = dens;
        prs[s] = pr;
        ncount[s] = cnt;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_density(DevConsts c, const float2* __restrict__ pos,
                                                           const uint32_t* __restrict__ skeys,
                                                           const int* __restrict__ cell_start,
                                                           float* __restrict__ rho, float* __restrict__ prs,
                                                           int* __restrict__ ncount, uint32_t* __restrict__ nmask,
                                                           uint32_t* __restrict__ nmask_hi, int n_max, TileQueue q, int col_lo, int col_hi, int cap,
                                                           DenseSink sink) {
    extern __shared__ __align__(16) float4 s_dyn[];
    float2* s_pos = reinterpret_cast<float2*>(s_dyn);
    int* s_off = reinterpret_cast<int*>(s_pos + cap);
    int* s_gs = s_off + PAIR_MAXC + 1;
    int* s_wave = s_gs + PAIR_MAXC;
    for_each_tile(q, [&](const TileDesc& d, int tix, int) {
        if (d.col < col_lo || d.col >= col_hi) return;  // (the list may cover the force pass's wider range)
        density_entry<THREADS>(c, pos, skeys, cell_start, rho, prs, ncount, nmask, nmask_hi, n_max, d, cap, sink, s_pos, s_off, s_gs,
                               s_wave, tix);
    });
}

// ---------------------------------------------------------------------------------------------
// U independent j-terms of the force sum, SPH/sph.cpp:64-75, written statement by statement over
// the U pairs so that their long dependent chains (sqrt, four divisions) interleave in the
// instruction stream.  (rx, ry) = p_i - p_j.  Every operation is the reference's, in its order.
// ---------------------------------------------------------------------------------------------
// (*) The reference's `r_len > h` guards (sph.cpp:16,24) are dead here: a pair is evaluated only if
// r2 < RN(h*h), hence r2 <= h*h exactly or r2 is the float just below RN(h*h) <= h*h(1+2^-24), and the
// correctly rounded sqrt of either is <= h.  (A NaN r_len fails `>` as well.)
// FAST: divisions through exact_div.h (operands checked to be in the safe range when the tile was
// staged); otherwise hipcc's generic correctly rounded `/`.  Same results bit for bit.
// ZEROP (only with FAST): the tile was staged with every pressure -- the candidates', hence also the own particles' --
// equal to +0, and the pressure term is left out: pt = (+0 + +0) / (2 rho_j) = +0 (rho_j > 0 in the safe range), so
// pressureForce = ((-m) * pt) * grad = +-0 in both components (grad is finite: zeroed for r_len <= 1e-6, at most
// spiky * h^2 / 1e-6 otherwise), and (+-0 + a) + b == a + b unless a and b are both zeros, when only the sign of the
// zero term can differ -- which cannot change the sum, by the same argument as for the guards below (the sum starts
// at +0 and is never -0).  The spiky gradient (sph.cpp:15-20) is used by the pressure term alone and goes with it:
// one division by r_len, one by rho_j and ten multiplications / additions fewer per pair.  Every BASELINE scene runs
// here (densities far below REST_DENSITY: P = max(0, K (rho - rho0)) = +0 throughout); golden F5 (dense block, P > 0)
// and the reference box keep the general instantiation busy.  SPHHIP_ZERO_P=0 switches the shortcut off.
template <int U, bool FAST, bool ZEROP = false>
__device__ __forceinline__ void force_terms(const DevConsts& c, const float (&rx)[U], const float (&ry)[U],
                                            const float (&r2)[U], float pres_i, float vix, float viy,
                                            const float4 (&aj)[U] /* vx, vy, rho, P of j */,
                                            const float (&yd)[U] /* FAST: refined 1 / rho_j */, float (&fx)[U],
                                            float (&fy)[U]) {
    float r_len[U], gcoeff[U], pt[U], visc[U], inv_r[U], lcoeff[U];
    if (FAST) {
        // r2 is clamped to SPH_SQRT_LO = 2^-60 from below, i.e. r_len to 2^-30, for the rsq-based square
        // root (no zero / denormal handling).  That changes nothing downstream: any r_len <= 1e-6 has its
        // gradient and LJ terms zeroed by the guards below, and h - r_len == h for every r_len <= 2^-30
        // because h >= 2^-5 (checked when fast_div is enabled, ctx.hip).  The divisions by such an r_len
        // may produce inf / NaN: those feed only the two guarded coefficients.
        float yr[U];
#pragma unroll
        for (int u = 0; u < U; u++) r_len[u] = sph_sqrt_core(fmaxf(r2[u], SPH_SQRT_LO), yr[u]);  // sph.cpp:67
#pragma unroll
        for (int u = 0; u < U; u++) inv_r[u] = sph_rcp_core(r_len[u], yr[u]);  // 1.0f / r_len, sph.cpp:31
        if (!ZEROP) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                float term = c.h - r_len[u];
                gcoeff[u] = sph_div_core(c.spiky * term * term, r_len[u], yr[u]);  // sph.cpp:17-18
            }
#pragma unroll
            for (int u = 0; u < U; u++)  // (P_i + P_j) / (2 rho_j) == ((P_i + P_j) / rho_j) / 2 exactly in the safe range;
                pt[u] = sph_div_core(pres_i + aj[u].w, aj[u].z, yd[u]);  // the 1/2 is folded into neg_mass below, sph.cpp:69
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float lap = c.visc_lap * (c.h - r_len[u]);               // sph.cpp:23-26, see note (*)
            visc[u] = sph_div_core(c.visc_mass * lap, aj[u].z, yd[u]);  // sph.cpp:72
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; u++) r_len[u] = sqrtf(r2[u]);  // sph.cpp:67
#pragma unroll
        for (int u = 0; u < U; u++) inv_r[u] = 1.0f / r_len[u];  // sph.cpp:31
#pragma unroll
        for (int u = 0; u < U; u++) {
            float term = c.h - r_len[u];
            gcoeff[u] = c.spiky * term * term / r_len[u];  // sph.cpp:17-18
        }
#pragma unroll
        for (int u = 0; u < U; u++) pt[u] = (pres_i + aj[u].w) / (2.0f * aj[u].z);  // sph.cpp:69
#pragma unroll
        for (int u = 0; u < U; u++) {
            float lap = c.visc_lap * (c.h - r_len[u]);  // sph.cpp:23-26, see note (*)
            visc[u] = c.visc_mass * lap / aj[u].z;      // sph.cpp:72
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {  // Lennard-Jones, sph.cpp:32-37
        float sr = c.sigma * inv_r[u];
        float sr2 = sr * sr;
        float sr6 = sr2 * sr2 * sr2;
        float sr12 = sr6 * sr6;
        float inv_r2 = inv_r[u] * inv_r[u];
        lcoeff[u] = c.lj24 * (2.0f * sr12 - sr6) * inv_r2;
    }
    if (ZEROP) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            float relx = aj[u].x - vix, rely = aj[u].y - viy;  // sph.cpp:71
            bool zero_lj = (r_len[u] >= c.lj_cut) || (r_len[u] < 1e-6f);  // sph.cpp:30
            float lc = zero_lj ? 0.0f : lcoeff[u];
            fx[u] = visc[u] * relx + lc * rx[u];  // sph.cpp:74-75 without the (+-0) pressure term
            fy[u] = visc[u] * rely + lc * ry[u];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        // spiky gradient guard, sph.cpp:16.  `r_len < 1e-6` there is a double compare: true for
        // r_len <= 1e-6f (the float just below 1e-6), false from the next float up.
        bool zero_grad = r_len[u] <= 1e-6f;  // note (*)
        // The guards zero the COEFFICIENT (one select each) instead of both vector components: the
        // reference returns float2(0, 0); here a component may come out as -0 instead of +0.  That
        // cannot change the sum: a term of +-0 leaves `force` unchanged (force is never -0: it starts
        // at +0 and x + (-x) = +0), and next to a non-zero addend the sign of a zero is irrelevant.
        float gc = zero_grad ? 0.0f : gcoeff[u];
        float gx = gc * rx[u];
        float gy = gc * ry[u];
        // FAST: pt lacks its factor 1/2; (neg_mass / 2) * pt == neg_mass * (pt / 2) exactly (power of two,
        // no underflow in the safe range)
        const float nm = FAST ? 0.5f * c.neg_mass : c.neg_mass;
        float pfx = nm * pt[u] * gx;  // sph.cpp:70
        float pfy = nm * pt[u] * gy;
        float relx = aj[u].x - vix, rely = aj[u].y - viy;  // sph.cpp:71
        bool zero_lj = (r_len[u] >= c.lj_cut) || (r_len[u] < 1e-6f);  // sph.cpp:30
        float lc = zero_lj ? 0.0f : lcoeff[u];
        float ljx = lc * rx[u];
        float ljy = lc * ry[u];
        fx[u] = pfx + visc[u] * relx + ljx;  // sph.cpp:74-75 (the caller adds it to force)
        fy[u] = pfy + visc[u] * rely + ljy;
    }
}

__device__ __forceinline__ void force_term(const DevConsts& c, float rx, float ry, float r2, float pres_i,
                                           float vix, float viy, float4 aj, float& fx, float& fy) {
    float a[1] = {rx}, b[1] = {ry}, d[1] = {r2}, ox[1], oy[1], yd[1] = {0.0f};
    float4 e[1] = {aj};
    force_terms<1, false>(c, a, b, d, pres_i, vix, viy, e, yd, ox, oy);
    fx = ox[0];
    fy = oy[0];
}

// Everything after the j loop of serial_computeForces (sph.cpp:77-97) and, when FUSE, the body of
// serial_integrate (sph.cpp:105-128).
template <bool FUSE>
__device__ __forceinline__ void force_epilogue(const DevConsts& c, float fx, float fy, float2& p, float2& v,
                                               float dens, float mx, float my, float strength) {
    fy -= c.gravity * dens;
    if (strength != 0.0f) {
        float rx = p.x - mx, ry = p.y - my;
        float r2 = rx * rx + ry * ry;
        if (r2 < c.mouse_r2) {
            float r_len = sqrtf(r2);
            if (r_len > 1e-6f) {
                float forceMag = strength * (c.mouse_radius - r_len) / c.mouse_radius;
                fx += forceMag * rx / r_len;
                fy += forceMag * ry / r_len;
            }
        }
    }
    float forceMag = sqrtf(fx * fx + fy * fy);
    if (forceMag > c.force_cap * dens) {
        float scale = c.force_cap * dens / forceMag;
        fx *= scale;
        fy *= scale;
    }
    v.x += c.dt * fx / dens;
    v.y += c.dt * fy / dens;
}

__device__ __forceinline__ void integrate_one(const DevConsts& c, float2& p, float2& v) {
    p.x += c.dt * v.x;  // sph.cpp:105-106
    p.y += c.dt * v.y;
    float velMag = sqrtf(v.x * v.x + v.y * v.y);
    if (velMag > c.max_vel) {
        float scale = c.max_vel / velMag;
        v.x *= scale;
        v.y *= scale;
    }
    if (p.x < c.xmin) {
        p.x = c.xmin;
        v.x = fabsf(v.x) * c.damping;
    }
    if (p.x > c.xmax) {
        p.x = c.xmax;
        v.x = -fabsf(v.x) * c.damping;
    }
    if (p.y < c.ymin) {
        p.y = c.ymin;
        v.y = fabsf(v.y) * c.damping;
    }
    if (p.y > c.ymax) {
        p.y = c.ymax;
        v.y = -fabsf(v.y) * c.damping;
    }
}

// ---------------------------------------------------------------------------------------------
// force (+ velocity update, + fused integrate and wall clamp).
// Algorithmic HBM bytes / particle: R pos 8, vel 8, rho 4, P 4; W vel 8 = 32; fused: + W pos 8 = 40.
// Tiles whose column is outside [own_lo_col, own_hi_col) only pass their particles through
// (ghost columns of a slab).
// ---------------------------------------------------------------------------------------------
// Heavy phase of one round: evaluates the qn queued pairs of every lane (byte offsets from `lo` in
// s_q), FORCE_HEAVY_UNROLL at a time, in queue order.
template <int THREADS, bool FAST, bool WIN, int U = FORCE_HEAVY_UNROLL, bool SPLIT = false>
__device__ __forceinline__ void force_heavy(const DevConsts& c, const PairRec* s_rec,
                                            uint8_t* s_q, int qn, int lo, int self_off, float2 pi, float2 vi,
                                            float pres_i, float& fx, float& fy, int tix, int pq_off = 0) {
    const int tid = tix;
    // WIN: the range is one window of a longer one and the particle itself may lie outside it, so there is no
    // +0 term to pad with: exhausted slots evaluate the window's first record and are left out of the sum.
    if (!WIN) {
#pragma unroll
        for (int u = 0; u < U - 1; u++) s_q[(qn + u) * THREADS + tid] = (uint8_t)self_off;
    }
    for (int t = 0; __any(t < qn); t += U) {
        int jj[U];
        float tx[U], ty[U];
        float rx[U], ry[U], r2[U];
        float4 aj[U];
        float yd[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int e = s_q[(t + u) * THREADS + tid];
            if (WIN)
                jj[u] = lo + ((t + u < qn) ? e : 0);
            else
                jj[u] = lo + ((t < qn) ? e : self_off);  // a lane whose queue is exhausted re-evaluates itself
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const RecVals q = rec_load<SPLIT, true>(s_rec, jj[u], pq_off);
            aj[u] = q.aux;
            yd[u] = q.yd;
            rx[u] = pi.x - q.pos.x;
            ry[u] = pi.y - q.pos.y;
            r2[u] = rx[u] * rx[u] + ry[u] * ry[u];
        }
        force_terms<U, FAST>(c, rx, ry, r2, pres_i, vi.x, vi.y, aj, yd, tx, ty);
        // same order as the sequential sum: force += term(t); force += term(t+1); ...
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (FAST && !WIN) {
                fx += tx[u];
                fy += ty[u];
            } else {
                fx = (t + u < qn) ? fx + tx[u] : fx;
                fy = (t + u < qn) ? fy + ty[u] : fy;
            }
        }
    }
}

// The per-particle neighbour sum of the force pass over the LDS image, for the whole wave at once.
// Rounds: (1) every lane scans its candidate range until it has queued FORCE_Q in-range neighbours
// (cheap distance test only; indices go to a per-lane LDS queue in scan order), (2) all lanes
// evaluate their queued pairs together, FORCE_HEAVY_UNROLL independent pairs in flight per lane.
// The expensive pair arithmetic thus runs with (nearly) all lanes active instead of being predicated
// on a per-candidate distance test, and it stays in the canonical order.
//
// FAST (tile operands proven safe for exact_div.h): the particle itself is NOT filtered out of the
// scan -- its term is exactly +0 (r_len = 0: gradient and LJ are zeroed by their guards, relVel is
// +0, the pressure factor is finite because rho_i is in the safe range) and force + 0 == force --
// and the queue tail is padded with the particle itself, so the accumulation needs no predicate.
// !FAST keeps the reference's `j == i` test (sph.cpp:63) and a predicated accumulation.
// (U pairs in flight per lane, Q queued per round: the tuned kernels run many waves per SIMD and do best with
// 2 and 8; the dense kernel runs one wave per SIMD and wants the instruction-level parallelism of 8 and 24)
template <int THREADS, bool FAST, bool WIN = false, int U = FORCE_HEAVY_UNROLL, int Q = FORCE_Q, bool SPLIT = false>
__device__ __forceinline__ void force_rounds(const DevConsts& c, const PairRec* s_rec,
                                             uint8_t* s_q, int j, int hi, int self, float2 pi, float2 vi,
                                             float pres_i, float& fx, float& fy, int tix, int pq_off = 0) {
    const int tid = tix;
    // the queue holds byte offsets from the start of the lane's range (ranges are <= FORCE_MAX_RANGE long)
    const int lo = j;
    const int self_off = self - lo;
    while (__any(j < hi)) {
        int qn = 0;
        while (__any((j < hi) && (qn < Q))) {
            if ((j < hi) && (qn < Q)) {
                float2 q[FORCE_SCAN_UNROLL];
#pragma unroll
                for (int u = 0; u < FORCE_SCAN_UNROLL; u++) q[u] = rec_pos<SPLIT>(s_rec, j + u);  // may run into the row sentinels
#pragma unroll
                for (int u = 0; u < FORCE_SCAN_UNROLL; u++) {
                    float rx = pi.x - q[u].x, ry = pi.y - q[u].y;
                    float r2 = rx * rx + ry * ry;
                    bool in = !(r2 >= c.h2);  // sph.cpp:66
                    if (!FAST || WIN) in = in && (j + u != self);  // sph.cpp:63
                    if (WIN) in = in && (j + u < hi);  // (no sentinels behind a window: the range goes on)
                    // branch-free push: the slot is rewritten until a candidate is accepted
                    s_q[qn * THREADS + tid] = (uint8_t)(j + u - lo);
                    qn += in ? 1 : 0;
                }
                j += FORCE_SCAN_UNROLL;
            }
        }
        force_heavy<THREADS, FAST, WIN, U, SPLIT>(c, s_rec, s_q, qn, lo, self_off, pi, vi, pres_i, fx, fy, tix, pq_off);
    }
}

// The neighbour sum driven by the in-range bit masks k_density left behind (one word per 32 candidates
// of the lane's range, first candidate in the top bit) instead of a distance scan: every trip pops the
// next FORCE_HEAVY_UNROLL set bits of every lane and evaluates those pairs.  The current word is
// refilled from global memory (the next word is fetched one refill ahead); a lane that finds an empty
// word, or has run out, evaluates itself for that slot -- a term of exactly +0 (see FAST above).
// Only for FAST tiles: the masks contain the particle itself.
// One group of FORCE_HEAVY_UNROLL pairs as raw LDS records.
struct PairGroup {
    RecVals rec[FORCE_HEAVY_UNROLL];
};

template <int THREADS, bool ZEROP>
__device__ __forceinline__ void force_by_mask(const DevConsts& c, const PairRec* s_rec, uint32_t* s_mask,
                                              const uint32_t* __restrict__ mask_col, int n_max, int lo, int nw,
                                              int self, float2 pi, float2 vi, float pres_i, float& fx, float& fy, int tix,
                                              int pq_off) {
    constexpr int U = FORCE_HEAVY_UNROLL;
    const int tid = tix;
    // The lane's mask words are in LDS, s_mask[w * THREADS + tid] (force_tile puts the first batch's there
    // before staging; later batches are fetched here): the refills inside the pop loop are divergent, and a
    // global load there would stall the wave for a full memory round trip at almost every pop.
    if (mask_col) {
#pragma unroll
        for (int k = 0; k < PAIR_MASK_WORDS; k++)
            if (k < nw) s_mask[k * THREADS + tid] = mask_col[(size_t)k * n_max];
    }
    int w = 0;
    int base_i = lo;  // record index of the candidate in the top bit of the current word
    uint32_t m = 0, m_next = 0;
    if (nw > 0) m = s_mask[tid];
    if (nw > 1) m_next = s_mask[THREADS + tid];
    // Trips of the wave: the lane with the most neighbours sets them (its pops, FORCE_HEAVY_UNROLL per trip).  Counting
    // the bits up front makes the loop counter a scalar: the per-trip vote "any lane with a bit left?" (two compares,
    // an or and the vote itself) is gone.
    // (a pop that finds its word empty moves on to the next word once; if that one is empty too the pop is spent on the
    // particle itself: every empty word behind the first costs the lane one pop)
    int left = 0;
#pragma unroll
    for (int k = 0; k < PAIR_MASK_WORDS; k++)
        if (k < nw) {
            const uint32_t wk = s_mask[k * THREADS + tid];
            left += __builtin_popcount(wk) + ((k > 0 && wk == 0) ? 1 : 0);
        }
    int trips = (left + U - 1) / U;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) trips = max(trips, __shfl_xor(trips, d, 64));
    trips = __builtin_amdgcn_readfirstlane(trips);
    // pops the next U set bits of the lane and starts the LDS reads of those records
    auto fetch = [&](PairGroup& g) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            // (Measured dead end: moving the empty-word case entirely into the refill branch -- no compare / select per
            // pop, five VALU operations instead of eight -- made the kernel 9 % slower: the extra branch structure costs
            // more issue slots than the three operations save.)
            if (m == 0 && w + 1 < nw) {
                w++;
                base_i += 32;
                m = m_next;
                if (w + 1 < nw) m_next = s_mask[(w + 1) * THREADS + tid];
            }
            // v_ffbh_u32 of 0 is -1: shifted and added it gives an address that is not used (m == 0: the slot evaluates
            // the particle itself), and 0x7fffffff >> (-1 & 31) = 0 leaves m at 0
            int b;
            asm("v_ffbh_u32 %0, %1" : "=v"(b) : "v"(m));
            const int r = (m != 0) ? base_i + b : self;
            m &= 0x7fffffffu >> (b & 31);
            g.rec[u] = rec_load<true, !ZEROP>(s_rec, r, pq_off);
        }
    };
    auto eval = [&](const PairGroup& g) {
        float tx[U], ty[U], rx[U], ry[U], r2[U], yd[U];
        float4 aj[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            aj[u] = g.rec[u].aux;
            yd[u] = g.rec[u].yd;
            rx[u] = pi.x - g.rec[u].pos.x;
            ry[u] = pi.y - g.rec[u].pos.y;
            r2[u] = rx[u] * rx[u] + ry[u] * ry[u];
        }
        force_terms<U, true, ZEROP>(c, rx, ry, r2, pres_i, vi.x, vi.y, aj, yd, tx, ty);
#pragma unroll
        for (int u = 0; u < U; u++) {  // same order as the sequential sum
            fx += tx[u];
            fy += ty[u];
        }
    };
    // (A two-register-set software pipeline -- pops and LDS reads of the next group issued before the
    // current one is evaluated -- was measured: 30 more VGPRs, spills at 4 waves per SIMD, 1 % slower.)
    PairGroup g;
    for (int t = 0; t < trips; t++) {
        fetch(g);
        eval(g);
    }
}

#ifndef FORCE_WAVES_MAX
#define FORCE_WAVES_MAX 4
#endif
// One own particle of a force tile (registers).
struct OwnParticle {
    float2 p, v;
    float d, pr;
    uint32_t idv, key;
    __device__ __forceinline__ void load(int s, bool have, const float2* __restrict__ pos,
                                         const float2* __restrict__ vel, const float* __restrict__ rho,
                                         const float* __restrict__ prs, const uint32_t* __restrict__ idvs,
                                         const uint32_t* __restrict__ skeys) {
        p = v = make_float2(0.0f, 0.0f);
        d = 1.0f;
        pr = 0.0f;
        idv = SPH_INVALID_BIT;
        key = 0;
        if (have) {
            p = pos[s];
            v = vel[s];
            d = rho[s];
            pr = prs[s];
            idv = idvs[s];
            key = skeys[s];
        }
    }
};

// One list entry of the force pass.  Prologue as in density_entry: the candidate-cell table, the own particles, their
// mask words (only as many as the entry's longest stencil range needs) and the candidates' records (contiguous
// slot ranges straight from the entry) are all requested before the first wait; three barriers in all.
template <bool FUSE, int THREADS>
__device__ __forceinline__ void force_entry(const DevConsts& c, const float2* __restrict__ pos,
                                            const float2* __restrict__ vel, const float* __restrict__ rho,
                                            const float* __restrict__ prs, const uint32_t* __restrict__ idv,
                                            const uint32_t* __restrict__ skeys, const int* __restrict__ cell_start,
                                            float2* __restrict__ pos_o, float2* __restrict__ vel_o, const TileDesc& d,
                                            int cap, int own_lo_col, int own_hi_col, float mx, float my, float strength,
                                            int allow_fast, const uint32_t* __restrict__ nmask, int n_max,
                                            const DenseSink& sink, const ForceExtraDev* cdev, PairRec* s_rec, int* s_off, int* s_gs, int* s_wave,
                                            uint8_t* s_q, int tix STAMP_PARAM) {
    STAMP(0);  // entry start
    // ghost column of a slab: pass the particles through
    if (d.col < own_lo_col || d.col >= own_hi_col) {
        for (int s = d.own_lo + tix; s < d.own_hi; s += THREADS) {
            vel_o[s] = vel[s];
            if (FUSE) pos_o[s] = pos[s];
        }
        return;
    }
    TileGeom g = entry_geom(c, d);
    // a "long" tile (some stencil range above FORCE_MAX_RANGE records) needs the masks: two passes over their words;
    // without masks it goes the way of the dense sub-tiles
    const int nw_all = d.info & 0xff;
    const bool long_tile = nw_all > PAIR_MASK_WORDS;
    bool use_lds = (d.info & TILE_DENSE) == 0 && !(long_tile && !nmask);
    if (!use_lds && dense_defer(sink, g, tix, c.nx, c.ny)) return;
    const float inf = __builtin_inff();
    int* s_flags = s_wave + 7;
    uint32_t* s_mask = reinterpret_cast<uint32_t*>(s_q);
    const int pq_off = cap * 16;  // split record layout: hot[cap] (16 bytes), vel[cap] (8), prs[cap] (4): rec_load / rec_store

    // ---- every global load of the prologue
    int g0, g1;
    entry_cells_load(c, cell_start, g, tix, g0, g1);
    OwnParticle own0;
    const bool have0 = g.own_lo + tix < g.own_hi;
    own0.load(g.own_lo + tix, have0, pos, vel, rho, prs, idv, skeys);
    const int nw_tile = min(nw_all, PAIR_MASK_WORDS);
    uint32_t mw[PAIR_MASK_WORDS];
#pragma unroll
    for (int k = 0; k < PAIR_MASK_WORDS; k++) {
        mw[k] = 0;
        if (nmask && have0 && k < nw_tile) mw[k] = nmask[(size_t)k * n_max + g.own_lo + tix];
    }
    const int n_cand = use_lds ? entry_cand_count(d) : 0;
    float2 cq[PAIR_STAGE_BATCH], cw[PAIR_STAGE_BATCH];
    float cdj[PAIR_STAGE_BATCH], cpj[PAIR_STAGE_BATCH];
    uint32_t cid[PAIR_STAGE_BATCH], ck[PAIR_STAGE_BATCH];
    int cslot[PAIR_STAGE_BATCH], cdd[PAIR_STAGE_BATCH];
#pragma unroll
    for (int b = 0; b < PAIR_STAGE_BATCH; b++) {
        const int i = tix + b * THREADS;
        cslot[b] = -1;
        cdd[b] = 0;
        if (i < n_cand) {
            const int slot = entry_cand_slot(d, i, cdd[b]);
            cslot[b] = slot;
            ck[b] = skeys[slot];
            cq[b] = pos[slot];
            cw[b] = vel[slot];
            cdj[b] = rho[slot];
            cpj[b] = prs[slot];
            cid[b] = idv[slot];
        }
    }

    // ---- candidate-cell table
    entry_cells_table<THREADS>(g, g0, g1, s_off, s_gs, s_wave, tix);
    if (tix == 0) *s_flags = 0;
    __syncthreads();
    STAMP(1);  // table ready

    bool fast = false, by_mask = false, zero_p = false;
    if (use_lds) {
        // the lane's mask words go to its LDS slots, s_mask[w * THREADS + tix] (read back by the same lane only)
        if (nmask) {
#pragma unroll
            for (int k = 0; k < PAIR_MASK_WORDS; k++)
                if (k < nw_tile) s_mask[k * THREADS + tix] = mw[k];
        }
        // bit 0: an operand outside exact_div.h's safe range; bit 1: no masks; bit 2: a pressure that is not +0
        int flags = ((allow_fast & 1) ? 0 : 1) | ((allow_fast & 2) ? 0 : 4);
        if (!nmask) flags |= 2;          // k_density's masks: only if no candidate is invalid or non-finite
        auto stage = [&](int dst, float2 q, float2 w, float dj, float pj, uint32_t idj) {
            // exact_div.h is used only if every density / pressure of the tile is in its safe range
            flags |= (dj == 0.0f || !sph_safe_operand(dj) || !sph_safe_operand(pj)) ? 1 : 0;
            flags |= __float_as_uint(pj) ? 4 : 0;  // (-0 counts as a pressure: the shortcut is argued for +0 only)
            // `!particles[j].valid` (sph.cpp:63): an invalid j is staged at +inf so that
            // r2 = +inf >= H*H skips it with the same compare that skips far particles.
            flags |= (!(fabsf(q.x) < inf) || !(fabsf(q.y) < inf) || (idj & SPH_INVALID_BIT)) ? 2 : 0;
            if (idj & SPH_INVALID_BIT) q = make_float2(inf, inf);
            rec_store<true>(s_rec, dst, pq_off, make_float4(w.x, w.y, dj, pj), q, sph_rcp_refined(dj));
        };
#pragma unroll
        for (int b = 0; b < PAIR_STAGE_BATCH; b++)
            if (cslot[b] >= 0)
                stage(entry_cand_dst(c, g, s_off, s_gs, cslot[b], cdd[b], ck[b], cap), cq[b], cw[b], cdj[b], cpj[b], cid[b]);
        for (int i = tix + PAIR_STAGE_BATCH * THREADS; i < n_cand; i += THREADS) {  // (LDS budgets beyond 4 x THREADS)
            int dd;
            const int slot = entry_cand_slot(d, i, dd);
            stage(entry_cand_dst(c, g, s_off, s_gs, slot, dd, skeys[slot], cap), pos[slot], vel[slot], rho[slot], prs[slot],
                  idv[slot]);
        }
        for (int t = tix; t < (g.ncells / 3) * PAIR_ROW_PAD; t += THREADS) {
            int idx = s_off[3 * (t / PAIR_ROW_PAD) + 3] - PAIR_ROW_PAD + (t % PAIR_ROW_PAD);
            rec_store<true>(s_rec, idx, pq_off, make_float4(0.0f, 0.0f, 1.0f, 0.0f), make_float2(inf, inf), 1.0f);
        }
        if (flags) atomicOr(s_flags, flags);
        __syncthreads();
        const int all = *s_flags;
        fast = (all & 1) == 0;
        by_mask = fast && (all & 2) == 0;
        zero_p = (all & 4) == 0;
        if (long_tile && !by_mask) {
            // the masks cannot be used after all (an operand outside the safe range, an invalid or non-finite candidate)
            // and the scan path's byte queue does not reach beyond FORCE_MAX_RANGE: the way of the dense sub-tiles
            if (dense_defer(sink, g, tix, c.nx, c.ny)) return;
            use_lds = false;
        }
    }
    STAMP(2);  // staged
    STAMP_VAL(5, (unsigned long long)(g.own_hi - g.own_lo) | ((unsigned long long)s_off[g.ncells] << 32));

    // All lanes of a wave stay in the loop together (wave-level votes below); lanes without a
    // particle, or with an invalid one, simply have an empty candidate range.  On the LDS path an entry holds
    // at most THREADS own particles: one trip.
    const int n_iter = use_lds ? 1 : (g.own_hi - g.own_lo + THREADS - 1) / THREADS;
    for (int it = 0; it < n_iter; it++) {
        const int s = g.own_lo + it * THREADS + tix;
        const bool have = s < g.own_hi;
        OwnParticle own = own0;
        if (it > 0) own.load(s, have, pos, vel, rho, prs, idv, skeys);
        float2 pi = own.p, vi = own.v;
        const float dens_i = own.d, pres_i = own.pr;
        const bool valid = have && !(own.idv & SPH_INVALID_BIT);
        int r0 = 0, r1 = -1, self_cell = 0;
        if (have) {
            const int cy = (int)own.key - g.col * c.ny;
            r0 = max(cy - 1, 0) - g.ylo;
            r1 = min(cy + 1, c.ny - 1) - g.ylo;
            self_cell = 3 * (cy - g.ylo) + 1;
        }
        float fx = 0.0f, fy = 0.0f;
        if (use_lds) {
            int j = 0, hi = 0, self = 0;
            if (have && valid) {  // sph.cpp:59: invalid particles get no force
                j = s_off[3 * r0];
                hi = s_off[3 * r1 + 3] - PAIR_ROW_PAD;
                self = s_off[self_cell] + (s - s_gs[self_cell]);
            }
            if (by_mask) {
                // A long tile takes a second pass: the candidates beyond the first 8 mask words follow with their words
                // fetched from global memory (ascending candidates, the same order; lanes whose range is shorter
                // evaluate themselves meanwhile: terms of +0).  Two call sites, not one loop over the passes: with
                // the loop every tile paid for the second pass (k_force 6 % slower at 16M particles, long tiles or
                // not -- which is why the kernel once had an instantiation without it, picked by the host); with the
                // first pass's word pointer a compile-time null it costs nothing (1.78 -> 1.77 ms there, the evolved
                // 1M dam break with its 8 long tiles 0.274 -> 0.258 ms per step).
                const int nw_i = (hi - j + 31) >> 5;
                if (zero_p)
                    force_by_mask<THREADS, true>(c, s_rec, s_mask, nullptr, n_max, j, min(nw_i, PAIR_MASK_WORDS), self, pi, vi, pres_i, fx, fy, tix, pq_off);
                else
                    force_by_mask<THREADS, false>(c, s_rec, s_mask, nullptr, n_max, j, min(nw_i, PAIR_MASK_WORDS), self, pi, vi, pres_i, fx, fy, tix, pq_off);
                if (long_tile) {  // (the array of words 8..13: its address from device memory, only when it is needed)
                    const ForceExtraDev* xp = cdev;
                    asm volatile("" : "+s"(xp) : : "memory");
                    const uint32_t* hi_words = xp->nmask_hi + s;
                    const int nw_hi = min(max(nw_i - PAIR_MASK_WORDS, 0), PAIR_MASK_WORDS);
                    if (zero_p)
                        force_by_mask<THREADS, true>(c, s_rec, s_mask, hi_words, n_max, j + FORCE_MAX_RANGE, nw_hi, self, pi, vi, pres_i, fx, fy, tix, pq_off);
                    else
                        force_by_mask<THREADS, false>(c, s_rec, s_mask, hi_words, n_max, j + FORCE_MAX_RANGE, nw_hi, self, pi, vi, pres_i, fx, fy, tix, pq_off);
                }
            } else if (fast)
                force_rounds<THREADS, true, false, FORCE_HEAVY_UNROLL, FORCE_Q, true>(c, s_rec, s_q, j, hi, self, pi, vi, pres_i, fx, fy, tix, pq_off);
            else
                force_rounds<THREADS, false, false, FORCE_HEAVY_UNROLL, FORCE_Q, true>(c, s_rec, s_q, j, hi, self, pi, vi, pres_i, fx, fy, tix, pq_off);
        } else if (have && valid) {
            for (int cc = 3 * r0; cc < 3 * r1 + 3; cc++) {
                int gs = s_gs[cc], n = cand_count(s_off, cc);
                for (int j = gs; j < gs + n; j++) {
                    float2 q = pos[j];
                    if (idv[j] & SPH_INVALID_BIT) q = make_float2(inf, inf);
                    float rx = pi.x - q.x, ry = pi.y - q.y;
                    float r2 = rx * rx + ry * ry;
                    if (!(r2 >= c.h2) && j != s) {
                        float2 w = vel[j];
                        float ax = 0.0f, ay = 0.0f;
                        force_term(c, rx, ry, r2, pres_i, vi.x, vi.y, make_float4(w.x, w.y, rho[j], prs[j]), ax, ay);
                        fx += ax;
                        fy += ay;
                    }
                }
            }
        }
        STAMP(3);  // pair loop done
        if (have) {
            if (valid) {
                // The epilogue's dozen constants (gravity, force cap, mouse, dt, velocity cap, damping, walls) are read
                // from the context's device copy here, behind an opaque pointer, instead of sitting in scalar registers
                // through the whole pair loop as kernel arguments do: k_force has no scalar register to spare.
                const ForceExtraDev* cp = cdev;
                asm volatile("" : "+s"(cp) : : "memory");
                const DevConsts ce = cp->c;
                force_epilogue<FUSE>(ce, fx, fy, pi, vi, dens_i, mx, my, strength);
                if (FUSE) integrate_one(ce, pi, vi);
            }
            if (FUSE) pos_o[s] = pi;  // sph.cpp:59,103: invalid particles are left alone
            vel_o[s] = vi;
        }
    }
    STAMP(4);  // epilogue done
    STAMP_VAL(6, __builtin_amdgcn_s_memrealtime());
}

template <bool FUSE, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(FORCE_WAVES_MAX, FORCE_WAVES_MAX))) void k_force(
    DevConsts c, const float2* __restrict__ pos, const float2* __restrict__ vel, const float* __restrict__ rho,
    const float* __restrict__ prs, const uint32_t* __restrict__ idv, const uint32_t* __restrict__ skeys,
    const int* __restrict__ cell_start, float2* __restrict__ pos_o, float2* __restrict__ vel_o, TileQueue q, int col_lo,
    int col_hi, int skip_lo, int skip_hi, int cap, int own_lo_col, int own_hi_col, float mx, float my,
    float strength, int allow_fast, const uint32_t* __restrict__ nmask, int n_max,
    DenseSink sink, const ForceExtraDev* __restrict__ cdev) {
    extern __shared__ __align__(16) float4 s_dyn[];
    PairRec* s_rec = reinterpret_cast<PairRec*>(s_dyn);
    int* s_off = reinterpret_cast<int*>(s_rec + cap);
    int* s_gs = s_off + PAIR_MAXC + 1;
    int* s_wave = s_gs + PAIR_MAXC;
    uint8_t* s_q = reinterpret_cast<uint8_t*>(s_wave + 8);
    for_each_tile(q, [&](const TileDesc& d, int tix, int t_list) {
#ifdef SPH_STAMPS
        const int stamp_slot = t_list < STAMP_TILES ? t_list : -1;
        if (stamp_slot >= 0 && (tix & 63) == 0) {
            g_stamps[((size_t)stamp_slot * 4 + (tix >> 6)) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        }
#else
        (void)t_list;
#endif
        // columns [col_lo, col_hi) without [skip_lo, skip_hi): a slab step runs the columns next to the slab
        // boundaries first and the interior in a second launch (sph_force_launch)
        if (d.col < col_lo || d.col >= col_hi || (d.col >= skip_lo && d.col < skip_hi)) return;
        force_entry<FUSE, THREADS>(c, pos, vel, rho, prs, idv, skeys, cell_start, pos_o, vel_o, d, cap, own_lo_col,
                                   own_hi_col, mx, my, strength, allow_fast, nmask, n_max, sink, cdev, s_rec, s_off, s_gs, s_wave,
                                   s_q, tix STAMP_ARG);
    });
}

// ---------------------------------------------------------------------------------------------
// The dense sub-tiles (see DenseSink): one workgroup per list entry = up to DENSE_THREADS own particles
// of rows [a, a + h) of a column.  The candidate sequence of the sub-tile (row-major, row sentinels, as
// in the tuned kernels) goes through the LDS buffer in chunks, one after the other, and every particle
// walks its part of each chunk: ascending chunks = the canonical order.
// ---------------------------------------------------------------------------------------------
// The force sum of a particle at a NaN position is NaN in both components as soon as its stencil holds one other
// candidate: rx = ry = NaN, so r2 = NaN fails sph.cpp:66's `r2 >= H*H` and every candidate is evaluated, every term
// being NaN.  (NaNs compare equal in the parity tests whatever their sign or payload: x86 and gfx950 differ there
// anyway.)  `real` = candidates of the particle's stencil rows, itself included.
__device__ __forceinline__ bool sph_nan_sum(float2 p, int real) { return (p.x != p.x || p.y != p.y) && real >= 2; }

struct DenseQueue {
    const int4* list;
    const int* count;
    int capacity;
    int* host_count;  // mapped host memory: the count, for the host's next decision and grid size
};

template <class Body>
__device__ __forceinline__ void for_each_dense(const DenseQueue& q, Body&& body) {
    const int total = min(*q.count, q.capacity);
    if (blockIdx.x == 0 && threadIdx.x == 0) *(volatile int*)q.host_count = total;
    for (int e = blockIdx.x; e < total; e += gridDim.x) {
        __syncthreads();  // the previous entry is done with the LDS image
        body(q.list[e], (int)threadIdx.x);
    }
}

// Calls put(dst, cc, k) for every real candidate (cell cc, k-th particle) whose position in the
// candidate sequence lies in [base, base + len), dst = position - base, and pad(dst) for the sentinels.
template <int THREADS, class Put, class Pad>
__device__ __forceinline__ void dense_stage_chunk(const TileGeom& g, const int* s_off, int base, int len, int tix,
                                                  Put&& put, Pad&& pad) {
    for (int cc = tix >> 4; cc < g.ncells; cc += THREADS / 16) {  // 16 lanes per candidate cell
        const int o = s_off[cc], n = cand_count(s_off, cc);
        const int k0 = max(0, base - o), k1 = min(n, base + len - o);
        for (int k = k0 + (tix & 15); k < k1; k += 16) put(o + k - base, cc, k);
    }
    for (int t = tix; t < (g.ncells / 3) * PAIR_ROW_PAD; t += THREADS) {
        const int seq = s_off[3 * (t / PAIR_ROW_PAD) + 3] - PAIR_ROW_PAD + (t % PAIR_ROW_PAD);
        if (seq >= base && seq < base + len) pad(seq - base);
    }
}

__global__ __launch_bounds__(DENSE_THREADS) void k_density_dense(DevConsts c, const float2* __restrict__ pos,
                                                                  const uint32_t* __restrict__ skeys,
                                                                  const int* __restrict__ cell_start,
                                                                  float* __restrict__ rho, float* __restrict__ prs,
                                                                  int* __restrict__ ncount, DenseQueue q, int cap) {
    constexpr int THREADS = DENSE_THREADS;
    extern __shared__ __align__(16) float4 s_dyn[];
    float2* s_pos = reinterpret_cast<float2*>(s_dyn);
    int* s_off = reinterpret_cast<int*>(s_pos + cap);
    int* s_gs = s_off + PAIR_MAXC + 1;
    int* s_wave = s_gs + PAIR_MAXC;
    for_each_dense(q, [&](int4 ent, int tix) {
        TileGeom g;
        tile_setup<THREADS>(c, cell_start, ent.x, ent.y, ent.y + ent.z, g, s_off, s_gs, s_wave, tix);
        __syncthreads();
        const int s = g.own_lo + ent.w * THREADS + tix;
        const bool have = s < g.own_hi;
        float2 pi = make_float2(0.0f, 0.0f);
        int lo_seq = 0, hi_seq = 0;
        if (have) {
            pi = pos[s];
            const int cy = (int)skeys[s] - g.col * c.ny;
            const int r0 = max(cy - 1, 0) - g.ylo, r1 = min(cy + 1, c.ny - 1) - g.ylo;
            lo_seq = s_off[3 * r0];
            hi_seq = s_off[3 * r1 + 3] - PAIR_ROW_PAD;
        }
        float dens = 0.0f;
        int cnt = 0;
        const float inf = __builtin_inff();
        // A particle at a NaN position is in range of nothing (r2 = NaN fails sph.cpp:46's `r2 < H*H`): density 0, no
        // neighbours, whatever the candidates.  Such particles are binned into cell 0 and pile up there once a scene
        // has blown up (the reference box at N = 65536 holds 38 000 of them after 1000 steps); an entry whose own
        // particles are all of that kind skips the candidate loop -- 38 000 x 38 000 rejected pairs otherwise.
        const int total = __syncthreads_or(have && !(pi.x != pi.x || pi.y != pi.y)) ? g.total : 0;
        for (int base = 0; base < total; base += cap) {
            if (base) __syncthreads();  // the previous chunk has been read
            int finite = 0;
            dense_stage_chunk<THREADS>(
                g, s_off, base, cap, tix,
                [&](int dst, int cc, int k) {
                    const float2 qj = pos[s_gs[cc] + k];
                    finite |= (fabsf(qj.x) < inf && fabsf(qj.y) < inf) ? 1 : 0;
                    s_pos[dst] = qj;
                },
                [&](int dst) { s_pos[dst] = make_float2(inf, inf); });
            // (a chunk of nothing but particles at NaN / infinite positions -- the pile in cell 0, walked by the few
            // finite particles of the cells around it -- is in range of nobody: 2 ms of the blown-up reference box)
            const bool chunk_live = __syncthreads_or(finite);
            const int k0 = max(lo_seq, base), k1 = chunk_live ? min(hi_seq, base + cap) : 0;
            // eight candidates per trip, loads first: at one or two waves per SIMD the LDS latency is not hidden
            for (int k = k0; k < k1; k += 8) {
                float2 qv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) qv[u] = s_pos[min(k + u, k1 - 1) - base];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    density_acc(c, pi, k + u < k1 ? qv[u] : make_float2(inf, inf), dens, cnt);
            }
        }
        if (have) {
            float pr = c.stiffness * (dens - c.rest_density);  // sph.cpp:52-53
            if (pr < 0.0f) pr = 0.0f;
            rho[s] = dens;
            prs[s] = pr;
            ncount[s] = cnt;
        }
    });
}

template <bool FUSE>
__global__ __launch_bounds__(DENSE_THREADS) void k_force_dense(
    DevConsts c, const float2* __restrict__ pos, const float2* __restrict__ vel, const float* __restrict__ rho,
    const float* __restrict__ prs, const uint32_t* __restrict__ idv, const uint32_t* __restrict__ skeys,
    const int* __restrict__ cell_start, float2* __restrict__ pos_o, float2* __restrict__ vel_o, DenseQueue q, int cap,
    float mx, float my, float strength, int allow_fast) {
    constexpr int THREADS = DENSE_THREADS;
    extern __shared__ __align__(16) float4 s_dyn[];
    PairRec* s_rec = reinterpret_cast<PairRec*>(s_dyn);
    int* s_off = reinterpret_cast<int*>(s_rec + cap);
    int* s_gs = s_off + PAIR_MAXC + 1;
    int* s_wave = s_gs + PAIR_MAXC;
    uint8_t* s_q = reinterpret_cast<uint8_t*>(s_wave + 8);
    const int chunk_len = cap - FORCE_SCAN_UNROLL;  // (the scan reads a few records past its window)
    for_each_dense(q, [&](int4 ent, int tix) {
        TileGeom g;
        tile_setup<THREADS>(c, cell_start, ent.x, ent.y, ent.y + ent.z, g, s_off, s_gs, s_wave, tix);
        __syncthreads();
        const int s = g.own_lo + ent.w * THREADS + tix;
        const bool have = s < g.own_hi;
        OwnParticle own;
        own.load(s, have, pos, vel, rho, prs, idv, skeys);
        float2 pi = own.p, vi = own.v;
        const float dens_i = own.d, pres_i = own.pr;
        const bool valid = have && !(own.idv & SPH_INVALID_BIT);
        int self_seq = -1, lo_seq = 0, hi_seq = 0;  // positions in the candidate sequence
        // exact_div.h only while every density / pressure seen so far is in its safe range (the particle's own
        // record may only arrive with a later chunk, so its operands are checked here)
        int unsafe = allow_fast ? 0 : 1;
        bool nan_own = false;
        if (valid) {  // sph.cpp:59: invalid particles get no force
            const int cy = (int)own.key - g.col * c.ny;
            const int r0 = max(cy - 1, 0) - g.ylo, r1 = min(cy + 1, c.ny - 1) - g.ylo;
            const int self_cell = 3 * (cy - g.ylo) + 1;
            lo_seq = s_off[3 * r0];
            hi_seq = s_off[3 * r1 + 3] - PAIR_ROW_PAD;
            self_seq = s_off[self_cell] + (s - s_gs[self_cell]);
            unsafe |= (dens_i == 0.0f || !sph_safe_operand(dens_i) || !sph_safe_operand(pres_i)) ? 1 : 0;
            nan_own = sph_nan_sum(pi, hi_seq - lo_seq - PAIR_ROW_PAD * (r1 - r0));
        }
        const float inf = __builtin_inff();
        float fx = 0.0f, fy = 0.0f;
        // (see sph_nan_sum: an entry whose particles all sit at NaN positions needs no candidates)
        const int total = __syncthreads_or(valid && !nan_own) ? g.total : 0;
        if (total == 0 && nan_own) fx = fy = __builtin_nanf("");
        for (int base = 0; base < total; base += chunk_len) {
            dense_stage_chunk<THREADS>(
                g, s_off, base, chunk_len, tix,
                [&](int dst, int cc, int k) {
                    const int j = s_gs[cc] + k;
                    float2 qj = pos[j];
                    const float2 w = vel[j];
                    const float dj = rho[j], pj = prs[j];
                    unsafe |= (dj == 0.0f || !sph_safe_operand(dj) || !sph_safe_operand(pj)) ? 1 : 0;
                    // `!particles[j].valid` (sph.cpp:63): an invalid j sits at +inf, out of everyone's range
                    if (idv[j] & SPH_INVALID_BIT) qj = make_float2(inf, inf);
                    PairRec rec;
                    rec.aux = make_float4(w.x, w.y, dj, pj);
                    rec.pos = qj;
                    rec.yd = sph_rcp_refined(dj);
                    rec.pad = 0.0f;
                    s_rec[dst] = rec;
                },
                [&](int dst) {
                    PairRec rec;
                    rec.aux = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
                    rec.pos = make_float2(inf, inf);
                    rec.yd = 1.0f;
                    rec.pad = 0.0f;
                    s_rec[dst] = rec;
                });
            const bool fast = !__syncthreads_or(unsafe);
            // the particle's part of this chunk, in windows the byte queue can address
            const int k0 = max(lo_seq, base), k1 = min(hi_seq, base + chunk_len);
            for (int w0 = k0; __any(w0 < k1); w0 += FORCE_MAX_RANGE) {
                const int w1 = min(w0 + FORCE_MAX_RANGE, k1);
                if (fast)
                    force_rounds<THREADS, true, true, DENSE_U, DENSE_Q>(c, s_rec, s_q, w0 - base, w1 - base, self_seq - base, pi, vi,
                                                                        pres_i, fx, fy, tix);
                else
                    force_rounds<THREADS, false, true, DENSE_U, DENSE_Q>(c, s_rec, s_q, w0 - base, w1 - base, self_seq - base, pi, vi,
                                                                         pres_i, fx, fy, tix);
            }
            // The chunk has been read.  A sum that is NaN in both components stays NaN whatever follows: once that
            // holds for every particle with candidates left (finite particles next to the NaN pile of cell 0, each
            // walking its tens of thousands of candidates: 9 ms of the blown-up reference box), the entry is done.
            if (!__syncthreads_or(valid && hi_seq > base + chunk_len && !(fx != fx && fy != fy))) break;
        }
        if (have) {
            if (valid) {
                force_epilogue<FUSE>(c, fx, fy, pi, vi, dens_i, mx, my, strength);
                if (FUSE) integrate_one(c, pi, vi);
            }
            if (FUSE) pos_o[s] = pi;  // sph.cpp:59,103: invalid particles are left alone
            vel_o[s] = vi;
        }
    });
}

// Term-parallel variant for scenes whose dense sub-tiles are too few to fill the GPU with one lane per
// particle (the reference box: 16 k particles are 256 waves, each walking ~900 candidates alone on its
// SIMD): here a wave takes ONE own particle at a time and its 64 lanes evaluate 64 consecutive candidates
// of the chunk; the in-range terms are then added in lane order (ascending candidate = the canonical
// order; `x + (+0)` for the others is exact, the sum never is -0).  Same terms, same order, same bits --
// about 3x the instructions per pair of the lane-per-particle kernel, but 64x its parallelism.  Entries
// are DENSE_PART_TP own particles; their partial sums live in LDS between the chunks.  Measured on the
// reference box (force pass per step, this kernel / k_force_dense): N=8192 0.10 / 0.17 ms, N=16384
// 0.16 / 0.25 ms, N=65536 1.58 / 1.32 ms -- the host picks it below ~40 k dense particles.
constexpr int DENSE_PART_TP = 16;

template <bool FUSE>
__global__ __launch_bounds__(DENSE_THREADS) void k_force_dense_tp(
    DevConsts c, const float2* __restrict__ pos, const float2* __restrict__ vel, const float* __restrict__ rho,
    const float* __restrict__ prs, const uint32_t* __restrict__ idv, const uint32_t* __restrict__ skeys,
    const int* __restrict__ cell_start, float2* __restrict__ pos_o, float2* __restrict__ vel_o, DenseQueue q, int cap,
    float mx, float my, float strength, int allow_fast) {
    constexpr int THREADS = DENSE_THREADS;
    extern __shared__ __align__(16) float4 s_dyn[];
    PairRec* s_rec = reinterpret_cast<PairRec*>(s_dyn);
    int* s_off = reinterpret_cast<int*>(s_rec + cap);
    int* s_gs = s_off + PAIR_MAXC + 1;
    int* s_wave = s_gs + PAIR_MAXC;
    float2* s_f = reinterpret_cast<float2*>(s_wave + 8);  // (the queue region of the other force kernels)
    for_each_dense(q, [&](int4 ent, int tix) {
        TileGeom g;
        tile_setup<THREADS>(c, cell_start, ent.x, ent.y, ent.y + ent.z, g, s_off, s_gs, s_wave, tix);
        const int lane = tix & 63, wave = tix >> 6;
        const int own0 = g.own_lo + ent.w * DENSE_PART_TP;
        const int n_own = min(DENSE_PART_TP, g.own_hi - own0);
        int unsafe = allow_fast ? 0 : 1;
        bool live = false;  // a particle whose sum needs the candidates
        __syncthreads();    // (the table of tile_setup)
        if (tix < n_own) {
            const int s = own0 + tix;
            const float di = rho[s], pi_ = prs[s];
            unsafe |= (di == 0.0f || !sph_safe_operand(di) || !sph_safe_operand(pi_)) ? 1 : 0;
            const int cy = (int)skeys[s] - g.col * c.ny;
            const int r0 = max(cy - 1, 0) - g.ylo, r1 = min(cy + 1, c.ny - 1) - g.ylo;
            const bool nan_own = sph_nan_sum(pos[s], s_off[3 * r1 + 3] - PAIR_ROW_PAD - s_off[3 * r0] - PAIR_ROW_PAD * (r1 - r0));
            live = !(idv[s] & SPH_INVALID_BIT) && !nan_own;
            const float f0 = nan_own ? __builtin_nanf("") : 0.0f;
            s_f[tix] = make_float2(f0, f0);
        }
        // (see sph_nan_sum: an entry whose particles all sit at NaN positions needs no candidates)
        const int total = __syncthreads_or(live) ? g.total : 0;
        const float inf = __builtin_inff();
        for (int base = 0; base < total; base += cap) {
            int alive = 0;  // a particle of this wave with candidates left and a sum that can still change
            dense_stage_chunk<THREADS>(
                g, s_off, base, cap, tix,
                [&](int dst, int cc, int k) {
                    const int j = s_gs[cc] + k;
                    float2 qj = pos[j];
                    const float2 w = vel[j];
                    const float dj = rho[j], pj = prs[j];
                    unsafe |= (dj == 0.0f || !sph_safe_operand(dj) || !sph_safe_operand(pj)) ? 1 : 0;
                    if (idv[j] & SPH_INVALID_BIT) qj = make_float2(inf, inf);  // sph.cpp:63
                    PairRec rec;
                    rec.aux = make_float4(w.x, w.y, dj, pj);
                    rec.pos = qj;
                    rec.yd = sph_rcp_refined(dj);
                    rec.pad = 0.0f;
                    s_rec[dst] = rec;
                },
                [&](int dst) {
                    PairRec rec;
                    rec.aux = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
                    rec.pos = make_float2(inf, inf);
                    rec.yd = 1.0f;
                    rec.pad = 0.0f;
                    s_rec[dst] = rec;
                });
            const bool fast = !__syncthreads_or(unsafe);
            for (int o = wave; o < n_own; o += THREADS / 64) {  // (everything about particle o is wave-uniform)
                const int s = own0 + o;
                if (idv[s] & SPH_INVALID_BIT) continue;  // sph.cpp:59
                const float2 pi = pos[s], vi = vel[s];
                const float pres_i = prs[s];
                const int cy = (int)skeys[s] - g.col * c.ny;
                const int r0 = max(cy - 1, 0) - g.ylo, r1 = min(cy + 1, c.ny - 1) - g.ylo;
                const int self_cell = 3 * (cy - g.ylo) + 1;
                const int self_seq = s_off[self_cell] + (s - s_gs[self_cell]);
                const int hi_seq = s_off[3 * r1 + 3] - PAIR_ROW_PAD;
                const int k0 = max(s_off[3 * r0], base), k1 = min(hi_seq, base + cap);
                float2 f = s_f[o];
                if (f.x != f.x && f.y != f.y) continue;  // NaN in both components stays NaN (see k_force_dense)
                for (int kk = k0; kk < k1; kk += 64) {
                    const int k = kk + lane;
                    float tx = 0.0f, ty = 0.0f;
                    bool in = false;
                    if (k < k1 && k != self_seq) {  // sph.cpp:63
                        const float4* rec = reinterpret_cast<const float4*>(s_rec + (k - base));
                        const float4 qv = rec[1];
                        const float rx = pi.x - qv.x, ry = pi.y - qv.y;
                        const float r2 = rx * rx + ry * ry;
                        if (!(r2 >= c.h2)) {  // sph.cpp:66
                            in = true;
                            if (fast) {
                                const float rxa[1] = {rx}, rya[1] = {ry}, r2a[1] = {r2}, yda[1] = {qv.z};
                                const float4 aja[1] = {rec[0]};
                                float txa[1], tya[1];
                                force_terms<1, true>(c, rxa, rya, r2a, pres_i, vi.x, vi.y, aja, yda, txa, tya);
                                tx = txa[0];
                                ty = tya[0];
                            } else {
                                force_term(c, rx, ry, r2, pres_i, vi.x, vi.y, rec[0], tx, ty);
                            }
                        }
                    }
                    unsigned long long bal = __ballot(in);
                    while (bal) {  // ascending candidate; v_readlane (the lane index is wave-uniform), not an LDS permute
                        const int b = __ffsll((long long)bal) - 1;
                        bal &= bal - 1;
                        f.x += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tx), b));
                        f.y += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty), b));
                    }
                }
                if (lane == 0) s_f[o] = f;
                alive |= (hi_seq > base + cap && !(f.x != f.x && f.y != f.y)) ? 1 : 0;
            }
            if (!__syncthreads_or(alive)) break;  // (also: the chunk has been read)
        }
        __syncthreads();
        if (tix < n_own) {
            const int s = own0 + tix;
            OwnParticle own;
            own.load(s, true, pos, vel, rho, prs, idv, skeys);
            float2 pi = own.p, vi = own.v;
            if (!(own.idv & SPH_INVALID_BIT)) {
                float2 f = s_f[tix];
                force_epilogue<FUSE>(c, f.x, f.y, pi, vi, own.d, mx, my, strength);
                if (FUSE) integrate_one(c, pi, vi);
            }
            if (FUSE) pos_o[s] = pi;  // sph.cpp:59,103: invalid particles are left alone
            vel_o[s] = vi;
        }
    });
}

// Stand-alone integrate (serial_integrate, sph.cpp:101-130).  R pos 8 + vel 8 + idv 4, W 16.
__global__ __launch_bounds__(256) void k_integrate(DevConsts c, float2* __restrict__ pos, float2* __restrict__ vel,
                                                    const uint32_t* __restrict__ idv,
                                                    const uint32_t* __restrict__ skeys,
                                                    const int* __restrict__ d_n, int use_cols, int own_lo_col,
                                                    int own_hi_col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *d_n) return;
    if (idv[i] & SPH_INVALID_BIT) return;
    if (use_cols) {
        int col = (int)skeys[i] / c.ny;
        if (col < own_lo_col || col >= own_hi_col) return;
    }
    float2 p = pos[i], v = vel[i];
    integrate_one(c, p, v);
    pos[i] = p;
    vel[i] = v;
}

// ---------------------------------------------------------------------------------------------
// Reference-order mode (parity tool): the reference's serial step exactly as written --
// all pairs in ascending j, and the in-place velocity update of serial_computeForces (sph.cpp:96-97:
// particle i sees the already-updated velocity of every j < i).  That is sequential in i, so it is
// for small N only: the state is put in id order (slot == id) and one wavefront walks i = 0..n-1 with
// the whole state in LDS; for every i the lanes evaluate 64 consecutive j at a time and the in-range
// terms are added in lane order (v_readlane), i.e. in ascending j like the reference's loop.  Same
// arithmetic helpers as the production kernels (the generic `/`, sqrtf instantiation).  Results equal
// the compiled reference bit for bit (tests: golden F2 trajectory, F6 mouse).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_to_id_order(const int* __restrict__ d_n, const float2* __restrict__ pos,
                                                      const float2* __restrict__ vel, const float* __restrict__ rho,
                                                      const float* __restrict__ prs, const uint32_t* __restrict__ idv,
                                                      float2* __restrict__ pos_o, float2* __restrict__ vel_o,
                                                      float* __restrict__ rho_o, float* __restrict__ prs_o,
                                                      uint32_t* __restrict__ idv_o, int* __restrict__ bad) {
    const int n = *d_n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = idv[i];
    const uint32_t id = w & SPH_ID_MASK;
    if (id >= (uint32_t)n) {  // ids must be a permutation of 0..n-1
        atomicOr(bad, 1);
        return;
    }
    pos_o[id] = pos[i];
    vel_o[id] = vel[i];
    rho_o[id] = rho[i];
    prs_o[id] = prs[i];
    idv_o[id] = w;
}

// serial_computeDensityPressure (sph.cpp:41-55): every i in parallel, j ascending over ALL particles.
__global__ __launch_bounds__(256) void k_ref_density(DevConsts c, int n, const float2* __restrict__ pos,
                                                      float* __restrict__ rho, float* __restrict__ prs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 pi = pos[i];
    float dens = 0.0f;
    int cnt = 0;
    for (int j = 0; j < n; j++) density_acc(c, pi, pos[j], dens, cnt);
    float pr = c.stiffness * (dens - c.rest_density);
    if (pr < 0.0f) pr = 0.0f;
    rho[i] = dens;
    prs[i] = pr;
}

constexpr int REF_MAX_N = 4096;  // 24 B of LDS per particle

// serial_computeForces (sph.cpp:57-99) + serial_integrate (sph.cpp:101-130); one wavefront.
__global__ __launch_bounds__(64) void k_ref_forces_integrate(DevConsts c, int n, float2* __restrict__ pos,
                                                              float2* __restrict__ vel, const float* __restrict__ rho,
                                                              const float* __restrict__ prs,
                                                              const uint32_t* __restrict__ idv, float mx, float my,
                                                              float strength) {
    extern __shared__ __align__(16) float4 s_dyn[];
    float2* s_pos = reinterpret_cast<float2*>(s_dyn);
    float2* s_vel = s_pos + n;
    float* s_rho = reinterpret_cast<float*>(s_vel + n);
    float* s_prs = s_rho + n;
    uint32_t* s_ok = reinterpret_cast<uint32_t*>(s_prs + n);
    const int lane = threadIdx.x;
    for (int k = lane; k < n; k += 64) {
        s_pos[k] = pos[k];
        s_vel[k] = vel[k];
        s_rho[k] = rho[k];
        s_prs[k] = prs[k];
        s_ok[k] = (idv[k] & SPH_INVALID_BIT) ? 0u : 1u;
    }
    __syncthreads();
    const int nchunk = (n + 63) / 64;
    for (int i = 0; i < n; i++) {
        if (!s_ok[i]) continue;  // sph.cpp:59 (uniform)
        const float2 pi = s_pos[i];
        float2 vi = s_vel[i];
        const float dens_i = s_rho[i], pres_i = s_prs[i];
        float fx = 0.0f, fy = 0.0f;
        for (int ch = 0; ch < nchunk; ch++) {
            const int j = ch * 64 + lane;
            float tx = 0.0f, ty = 0.0f;
            bool in = false;
            if (j < n && j != i && s_ok[j]) {  // sph.cpp:63
                const float2 q = s_pos[j];
                const float rx = pi.x - q.x, ry = pi.y - q.y;
                const float r2 = rx * rx + ry * ry;
                if (!(r2 >= c.h2)) {  // sph.cpp:66
                    in = true;
                    const float2 w = s_vel[j];  // already updated for j < i: the reference's in-place update
                    force_term(c, rx, ry, r2, pres_i, vi.x, vi.y, make_float4(w.x, w.y, s_rho[j], s_prs[j]), tx, ty);
                }
            }
            unsigned long long bal = __ballot(in);
            while (bal) {  // ascending j
                const int b = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                fx += __shfl(tx, b, 64);
                fy += __shfl(ty, b, 64);
            }
        }
        float2 p = pi;
        force_epilogue<false>(c, fx, fy, p, vi, dens_i, mx, my, strength);
        if (lane == 0) s_vel[i] = vi;  // sph.cpp:96-97
        __syncthreads();                // one wavefront: orders the LDS write before the next i reads it
    }
    for (int k = lane; k < n; k += 64) {  // serial_integrate
        float2 p = s_pos[k], v = s_vel[k];
        if (s_ok[k]) integrate_one(c, p, v);
        pos[k] = p;
        vel[k] = v;
    }
}

int sph_step_reference_order(SphCtx* c, float mx, float my, float strength, int nsteps) {
    const int n = c->n;
    if (n > REF_MAX_N) {
        sph_set_error("sphhip_step_reference_order: more than 4096 particles (this mode is a sequential parity tool)");
        return SPHHIP_ECAPACITY;
    }
    if (n == 0 || nsteps == 0) return SPHHIP_OK;
    int* bad = c->counters + 8;
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_to_id_order, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_n, c->pos, c->vel, c->rho, c->prs,
                       c->idv, c->pos2, c->vel2, c->rho2, c->prs2, c->idv2, bad);
    LAUNCH_CHECK();
    int h_bad = 0;
    HIP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_bad) {
        sph_set_error("sphhip_step_reference_order: particle ids are not a permutation of 0..n-1");
        return SPHHIP_ESTATE;
    }
    std::swap(c->pos, c->pos2);
    std::swap(c->vel, c->vel2);
    std::swap(c->rho, c->rho2);
    std::swap(c->prs, c->prs2);
    std::swap(c->idv, c->idv2);
    c->grid_valid = false;
    c->counts_valid = false;
    c->slice_valid = false;
    c->pending_drop = false;
    c->tail_appended = false;
    const size_t lds = (size_t)n * 28;
    for (int s = 0; s < nsteps; s++) {
        hipLaunchKernelGGL(k_ref_density, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->dc, n, c->pos, c->rho, c->prs);
        hipLaunchKernelGGL(k_ref_forces_integrate, dim3(1), dim3(64), lds, c->stream, c->dc, n, c->pos, c->vel, c->rho,
                           c->prs, c->idv, mx, my, strength);
        LAUNCH_CHECK();
    }
    return SPHHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Builds the work list of the pair kernels (sphhip_internal.h: sph_tile_list_thread); the grid build normally does
// this in the head of its reorder kernel (grid.hip), this kernel is the fallback (another tiling, LDS budget or
// column range than the list was built for).
__global__ __launch_bounds__(256) void k_tile_list(const int* __restrict__ cell_start, int nx, int ny, TileListArgs tl,
                                                    int* __restrict__ err) {
    sph_tile_list_thread(blockIdx.x * blockDim.x + threadIdx.x, cell_start, nx, ny, tl, err);
}

// (Re)builds the tile list for the current grid, tiling, LDS budget and column range if it is not current.
static int ensure_tile_list(SphCtx* c, TileQueue& q) {
    TileListArgs tl;
    int lo, hi;
    sph_tile_list_args(c, tl, lo, hi);
    if (!sph_tiles_current(c, lo, hi)) {
        HIP_TRY(hipMemsetAsync(tl.count, 0, sizeof(int), c->stream));
        HIP_TRY(hipMemsetAsync(tl.count + SPH_LONG_COUNT_OFFSET, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_tile_list, dim3((tl.nthreads + 255) / 256), dim3(256), 0, c->stream, c->cell_start, c->dc.nx,
                           c->dc.ny, tl, c->d_err);
        LAUNCH_CHECK();
        sph_tiles_mark(c, lo, hi);
    }
    q.list = c->tile_list;
    q.count = tl.count;
    q.host_count = c->d_host_tiles;
    q.tile_cells = tl.tile_cells;
    return SPHHIP_OK;
}

// Grid of a pair kernel: one workgroup per listed tile going by the last count that reached the host
// (plus slack; the kernels stride inside their XCD's chunk, so any grid is correct and balanced: for_each_tile),
// else one resident wave of workgroups.  Always a multiple of 8 (one chunk per XCD), at least 8, and never more
// workgroups than the list has room for entries.
static int pair_grid(SphCtx* c, const void* kern, int threads, size_t lds, int& grid) {
    const int grid_max = c->tile_list_cap & ~7;
    const int est = ((volatile int*)c->h_host_tiles)[0], est_cells = ((volatile int*)c->h_host_tiles)[1];
    if (est >= 0 && est_cells > 0) {
        // (a count taken under another tiling is rescaled; any grid size is correct)
        long long e = est_cells == c->tile_cells ? est : ((long long)est * est_cells + c->tile_cells - 1) / c->tile_cells;
        e = (e + e / 64 + 8 + 7) & ~7ll;
        grid = (int)std::min<long long>(e, grid_max);
        if (grid < 8) grid = 8;
        return SPHHIP_OK;
    }
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
    if (per_cu < 1) per_cu = 1;
    grid = (c->num_cus * per_cu) & ~7;
    if (grid > grid_max) grid = grid_max;
    if (grid < 8) grid = 8;
    return SPHHIP_OK;
}

// Height of the base tiles for the coming grid build.  The device-side list builder cuts a base tile that holds
// more particles than a workgroup has lanes (or more candidates than the LDS budget) into several entries, so any
// height is correct; the best one fills the lanes: about pair_threads own particles per entry (both pair kernels
// are VALU-bound, an idle lane is lost time).  Going by the entry count the last launch reported (mapped host
// memory, so a step or two old) the host estimates the particles per row of the occupied region and picks the
// height from that.  A tiny grid (the reference box has 169 cells) gets one row per workgroup, the shortest
// critical path.  Results do not depend on the tiling.
// (Measured dead end: choosing the height so that the entry count lands just below a whole number of rounds of
// resident workgroups -- 13 instead of 15 rows on the 1M dam break -- made k_force 19 % slower: the kernel is bound
// by VALU throughput, not by rounds; the last, partly filled round runs faster per tile, and shorter tiles leave
// lanes idle.)
void sph_adapt_tile_cells(SphCtx* c, int n) {
    if (!c->tile_cells_auto) return;
    const int tiles = ((volatile int*)c->h_host_tiles)[0], cells = ((volatile int*)c->h_host_tiles)[1];
    if (tiles <= 0 || cells <= 0 || n <= 0) return;
    int want = c->tile_cells;
    if ((long long)tiles * cells <= 4ll * c->num_cus) {
        want = 1;
    } else {
        // own particles per entry of the last list, at `cells` rows per base tile
        const double per_entry = (double)n / (double)tiles;
        const double per_row = per_entry / (double)cells;  // (an underestimate where base tiles were cut or trimmed)
        int fit = (int)(0.95 * c->pair_threads / (per_row > 1e-3 ? per_row : 1e-3));
        fit = fit < 1 ? 1 : fit > c->tile_cells_max ? c->tile_cells_max : fit;
        // per_row is only meaningful while most base tiles give one entry: when they are cut (per_entry well below
        // the lanes although the rows are full) the estimate would keep growing the height; follow it downwards
        // freely, upwards only while entries are clearly under-filled
        if (fit < want || per_entry < 0.8 * c->pair_threads) want = fit;
        if (want > c->tile_cells && want < c->tile_cells + 2 && want < c->tile_cells_max) want = c->tile_cells;  // (no dithering)
    }
    if (want != c->tile_cells) {
        c->tile_cells = want;
        c->tiles_valid = false;
    }
}

// Sink / queue of the dense sub-tiles for one pair-kernel launch (which: 0 density, 1 force) and the
// decision whether the follow-up kernel runs: yes unless the last one reported an empty list (the
// mapped ints start at -1 = unknown).
static int dense_prepare(SphCtx* c, int which, DenseSink& sink, DenseQueue& dq, int& grid, bool& term_parallel) {
    const int seen = ((volatile int*)c->h_dense)[which];
    int* count = c->counters + SPH_DENSE_COUNT_SLOT;
    const int defer = seen != 0 ? 1 : 0;
    sink.dev = c->d_sinks + which;  // (list, counter, capacity, mapped flag, error word: written once, ctx.hip)
    // force pass: the term-parallel kernel while the dense particles (going by the last list) would not fill
    // the GPU with a lane each
    term_parallel = which == 1 && c->dense_tp && (seen < 0 || (long long)seen * c->dense_force_part < 40ll * 1024);
    const int sink_part = term_parallel ? DENSE_PART_TP : DENSE_THREADS;
    sink.mode = defer | (sink_part << 1);
    dq.list = c->dense_list;
    dq.count = count;
    dq.capacity = c->dense_cap;
    dq.host_count = c->d_dense + which;
    grid = 0;
    if (defer) {
        HIP_TRY(hipMemsetAsync(count, 0, sizeof(int), c->stream));
        long long want = seen > 0 ? (long long)seen + seen / 8 + 8 : 0;
        if (which == 1) {
            want = want * c->dense_force_part / sink_part;  // (the last count was in entries of the last part size)
            c->dense_force_part = sink_part;
        }
        grid = (int)std::min<long long>(std::max<long long>(want, 2 * c->num_cus), 1 << 20);
    }
    return SPHHIP_OK;
}

int sph_density_launch(SphCtx* c) {
    if (sph_n_grid(c) == 0) return SPHHIP_OK;
    PhaseScope ps(c, SPHHIP_PH_DENSITY);
    // one column wider than the owned range: ghost densities are recomputed locally
    int lo = c->col_lo > 0 ? c->col_lo - 1 : 0;
    int hi = c->col_hi < c->dc.nx ? c->col_hi + 1 : c->dc.nx;
    TileQueue q;
    int rc = ensure_tile_list(c, q);
    if (rc) return rc;
    const int th = c->pair_threads;
    size_t lds = pair_lds_bytes(c->pair_cap, false, th);
    auto kern = th == 128 ? k_density<128> : k_density<256>;
    int grid = 0;
    if ((rc = pair_grid(c, reinterpret_cast<const void*>(kern), th, lds, grid))) return rc;
    DenseSink sink;
    DenseQueue dq;
    int dense_grid = 0;
    bool tp = false;
    if ((rc = dense_prepare(c, 0, sink, dq, dense_grid, tp))) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(th), lds, c->stream, c->dc, c->pos, c->keys, c->cell_start, c->rho,
                       c->prs, c->ncount, c->nmask, c->nmask_hi, c->n_max, q, lo, hi, c->pair_cap, sink);
    LAUNCH_CHECK();
    if (sink.mode & 1) {
        hipLaunchKernelGGL(k_density_dense, dim3(dense_grid), dim3(DENSE_THREADS), pair_lds_bytes(c->pair_cap, false, DENSE_THREADS),
                           c->stream, c->dc, c->pos, c->keys, c->cell_start, c->rho, c->prs, c->ncount, dq, c->pair_cap);
        LAUNCH_CHECK();
    }
    c->counts_valid = true;
    c->mask_tile_cells = c->tile_cells;
    c->mask_cap = c->pair_cap;
    return SPHHIP_OK;
}

// part 0: the whole pass.  Parts 1 and 2 split a fused pass for the slab step: part 1 runs the ghost columns
// and the first `edge_l` / last `edge_r` owned columns -- everything the halo messages of the next step are
// packed from -- and swaps the state buffers; part 2 runs the remaining (interior) columns from the old
// buffers into the new ones while the messages travel.  Between the two only sphhip_pack_halo_async may be
// called (SphCtx::split.pending).
int sph_force_launch(SphCtx* c, float mx, float my, float strength, bool fuse_integrate, int part, int edge_l, int edge_r) {
    const bool second = part == 2;
    if (second && !c->split.pending) return SPHHIP_OK;
    if (sph_n_grid(c) == 0) return SPHHIP_OK;
    if (second && c->split.interior_enqueued) {  // the interior pass is already running beside the boundary pass: join
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->split.interior_enqueued = false;
        c->split.pending = false;
        return SPHHIP_OK;
    }
    PhaseScope ps(c, SPHHIP_PH_FORCE);
    // owned columns plus the halo on each side (ghost tiles pass their particles through); on a slab
    // every particle lies inside that range (sphhip_pack_halo keeps nothing farther out)
    const int flo = c->col_lo - SPHHIP_HALO_COLS > 0 ? c->col_lo - SPHHIP_HALO_COLS : 0;
    const int fhi = c->col_hi + SPHHIP_HALO_COLS < c->dc.nx ? c->col_hi + SPHHIP_HALO_COLS : c->dc.nx;
    int il = second ? c->split.il : c->col_lo + edge_l, ih = second ? c->split.ih : c->col_hi - edge_r;
    if (part == 1 && (il >= ih || !fuse_integrate)) part = 0;  // no interior: one launch does it all
    TileQueue q;
    int rc = ensure_tile_list(c, q);
    if (rc) return rc;
    const int th = c->pair_threads;
    size_t lds = pair_lds_bytes(c->pair_cap, true, th);
    // the masks of the last density pass describe this grid and this tiling
    const bool masks_ok = second ? c->split.masks_ok
                                 : c->use_masks && c->counts_valid && c->grid_valid && c->mask_tile_cells == c->tile_cells &&
                                       c->mask_cap == c->pair_cap;
    auto kern = fuse_integrate ? (th == 128 ? k_force<true, 128> : k_force<true, 256>)
                               : (th == 128 ? k_force<false, 128> : k_force<false, 256>);
    int grid = 0;
    if ((rc = pair_grid(c, reinterpret_cast<const void*>(kern), th, lds, grid))) return rc;
    DenseSink sink;
    DenseQueue dq;
    int dense_grid = 0;
    bool tp = false;
    if ((rc = dense_prepare(c, 1, sink, dq, dense_grid, tp))) return rc;
    // (part 2 runs after part 1 has swapped the buffers: it reads the old state, now in pos2 / vel2)
    const float2 *pin = second ? c->pos2 : c->pos, *vin = second ? c->vel2 : c->vel;
    float2 *pout = second ? c->pos : c->pos2, *vout = second ? c->vel : c->vel2;
    const int lo = second ? il : flo, hi = second ? ih : fhi;
    const int skip_lo = part == 1 ? il : 0, skip_hi = part == 1 ? ih : 0;
    // Slab step in two halves: the boundary columns' pass is a few hundred workgroups, one tile's latency (20 us) of a
    // nearly empty GPU.  So the interior pass is started first, on the side stream behind everything enqueued so
    // far, and the boundary pass runs beside it on the context's stream; sphhip_step_finish only joins.  Same kernels, same tiles, disjoint output slots.  Not while dense
    // sub-tiles are listed (the two passes would share the list).  When phases are timed the interior launch is
    // bracketed on the side stream (SPHHIP_PH_FORCE_INTERIOR), the boundary launch on the context's (SPHHIP_PH_FORCE).
    bool beside = false;
    if (part == 1 && c->split_concurrent && c->side_stream && !(sink.mode & 1)) {
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        sph_phase_begin(c, SPHHIP_PH_FORCE_INTERIOR, c->side_stream);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(th), lds, c->side_stream, c->dc, pin, vin, c->rho, c->prs, c->idv, c->keys,
                           c->cell_start, pout, vout, q, il, ih, 0, 0, c->pair_cap, c->col_lo, c->col_hi, mx, my, strength,
                           (c->fast_div ? 1 : 0) | (c->zero_p ? 2 : 0), masks_ok ? c->nmask : (const uint32_t*)nullptr, c->n_max, sink, c->d_consts);
        LAUNCH_CHECK();
        sph_phase_end(c, SPHHIP_PH_FORCE_INTERIOR, c->side_stream);
        HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
        beside = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(th), lds, c->stream, c->dc, pin, vin, c->rho, c->prs, c->idv, c->keys,
                       c->cell_start, pout, vout, q, lo, hi, skip_lo, skip_hi, c->pair_cap, c->col_lo, c->col_hi,
                       mx, my, strength, (c->fast_div ? 1 : 0) | (c->zero_p ? 2 : 0), masks_ok ? c->nmask : (const uint32_t*)nullptr, c->n_max, sink, c->d_consts);
    LAUNCH_CHECK();
    if (sink.mode & 1) {
        auto dkern = tp ? (fuse_integrate ? k_force_dense_tp<true> : k_force_dense_tp<false>)
                        : (fuse_integrate ? k_force_dense<true> : k_force_dense<false>);
        hipLaunchKernelGGL(dkern, dim3(dense_grid), dim3(DENSE_THREADS), pair_lds_bytes(c->pair_cap, true, DENSE_THREADS),
                           c->stream, c->dc, pin, vin, c->rho, c->prs, c->idv, c->keys, c->cell_start, pout, vout, dq,
                           c->pair_cap, mx, my, strength, c->fast_div ? 1 : 0);
        LAUNCH_CHECK();
    }
    if (second) {
        c->split.pending = false;
        return SPHHIP_OK;
    }
    std::swap(c->vel, c->vel2);
    if (fuse_integrate) {
        std::swap(c->pos, c->pos2);
        c->grid_valid = false;
    }
    if (part == 1) {
        c->split.pending = true;
        c->split.interior_enqueued = beside;
        c->split.masks_ok = masks_ok;
        c->split.il = il;
        c->split.ih = ih;
        c->split.mx = mx;
        c->split.my = my;
        c->split.strength = strength;
    }
    return SPHHIP_OK;
}

int sph_integrate_launch(SphCtx* c) {
    if (sph_n_grid(c) == 0) return SPHHIP_OK;
    PhaseScope ps(c, SPHHIP_PH_INTEGRATE);
    bool restrict_cols = c->grid_valid && (c->col_lo > 0 || c->col_hi < c->dc.nx);
    hipLaunchKernelGGL(k_integrate, dim3((sph_n_grid(c) + 255) / 256), dim3(256), 0, c->stream, c->dc, c->pos, c->vel,
                       c->idv, c->keys, c->d_n, restrict_cols ? 1 : 0, c->col_lo, c->col_hi);
    LAUNCH_CHECK();
    c->grid_valid = false;
    return SPHHIP_OK;
}
