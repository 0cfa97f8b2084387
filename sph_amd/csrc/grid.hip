// grid.hip -- uniform-grid cell list for gfx950: cell hash, sort by cell (default: one counting pass,
// k_cell_count .. k_cell_scatter, or the whole build in one workgroup for small scenes, k_grid_small;
// SPHHIP_GRID=radix: the LDS-local LSD radix sort, k_sort_*), monotone cell start array, in-cell id
// ranking and the reorder of the SoA state into (key, id) order.
//
// There is no reference code for this phase: the reference only names `int* grid, int* cellStart,
// int* cellEnd` as dead kernel parameters (SPH/sph.cuh:33-34,38; always nullptr, SPH/sph.cu:222-223).
// The spec (shared with oracle/sph_oracle.c, orc_build_cells) is:
//   cx = clamp((int)floorf((x - xmin) * inv_cell), 0, nx-1), cy likewise, key = cx * ny + cy
//   sorted order = ascending (key, id);  cell_start[k] = #{particles with key < k}, k = 0..ncell
// All of it is HBM-bound integer work: coalesced 4/8-byte streams, LDS for the per-tile ranking.
#include "sphhip_internal.h"

constexpr int SCAN_CHUNK = 2048;  // cells per workgroup of the scans over the cell table

// ---------------------------------------------------------------------------------------------
// cell hash: key per slot + identity permutation.   R pos 8 B, W key 4 B + idx 4 B per particle
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash(DevConsts c, const float2* __restrict__ pos, const int* __restrict__ d_n,
                                               uint32_t* __restrict__ keys, uint32_t* __restrict__ perm) {
    const int n = *d_n;  // particle count lives on the device (asynchronous slab steps)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 p = pos[i];
    int cx = sph_cell_coord(p.x, c.xmin, c.inv_cell, c.nx);
    int cy = sph_cell_coord(p.y, c.ymin, c.inv_cell, c.ny);
    keys[i] = (uint32_t)(cx * c.ny + cy);
    perm[i] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------
// Radix sort pass, kernel 1: per-workgroup digit histogram, stored digit-major
// (block_hist[d * nblocks + b]) so that kernel 2 scans contiguous rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint32_t* __restrict__ keys, const int* __restrict__ d_n, int shift,
                                                             uint32_t mask, int* __restrict__ block_hist,
                                                             int nblocks) {
    const int n = *d_n;  // particle count lives on the device (asynchronous slab steps)
    __shared__ int hist[SORT_RADIX];
    int tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    int base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        int idx = base + r * SORT_THREADS + tid;
        if (idx < n) atomicAdd(&hist[(keys[idx] >> shift) & mask], 1);
    }
    __syncthreads();
    block_hist[tid * nblocks + blockIdx.x] = hist[tid];
}

// Exclusive scan of 256 values, one per thread of a 256-thread workgroup.
__device__ __forceinline__ int block_excl_scan_256(int v, int* lds_wave /*[4]*/, int* total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int w0 = lds_wave[0], w1 = lds_wave[1], w2 = lds_wave[2], w3 = lds_wave[3];
    int off = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    if (total) *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return off + incl - v;
}

// Kernel 2a: workgroup d scans row d of block_hist in place (exclusive) and stores the row total.
__global__ __launch_bounds__(256) void k_sort_scan_rows(int* __restrict__ block_hist, int nblocks,
                                                         int* __restrict__ digit_total) {
    __shared__ int lds_wave[4];
    int* row = block_hist + (size_t)blockIdx.x * nblocks;
    int carry = 0;
    for (int base = 0; base < nblocks; base += 256) {
        int i = base + (int)threadIdx.x;
        int v = (i < nblocks) ? row[i] : 0;
        int tot;
        int ex = block_excl_scan_256(v, lds_wave, &tot);
        if (i < nblocks) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// Kernel 2b: exclusive scan of the 256 digit totals -> digit_total[256 + d] = global base of digit d.
__global__ __launch_bounds__(256) void k_sort_scan_digits(int* __restrict__ digit_total) {
    __shared__ int lds_wave[4];
    int v = digit_total[threadIdx.x];
    int ex = block_excl_scan_256(v, lds_wave, nullptr);
    digit_total[SORT_RADIX + threadIdx.x] = ex;
}

// Kernel 3: stable LDS-local ranking of the tile by digit, then a run-coalesced scatter.
// Stability (needed by the LSD passes) follows from ranking items in index order: rounds in
// order, waves in order inside a round, lanes in order inside a wave.
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, const int* __restrict__ d_n, int shift,
                                                                uint32_t mask, const int* __restrict__ block_hist,
                                                                const int* __restrict__ digit_total, int nblocks) {
    const int n = *d_n;  // particle count lives on the device (asynchronous slab steps)
    __shared__ unsigned short cnt[SORT_ITEMS][4][SORT_RADIX];  // 16 KB
    __shared__ int digit_start[SORT_RADIX];
    __shared__ int global_off[SORT_RADIX];
    __shared__ int lds_wave[4];
    __shared__ uint2 stage[SORT_TILE];  // 16 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * SORT_TILE;
    const int tile_n = min(SORT_TILE, n - base);

    {  // zero the counters: 8*4*256 shorts = 4096 ints
        int* z = reinterpret_cast<int*>(&cnt[0][0][0]);
#pragma unroll
        for (int i = 0; i < (SORT_ITEMS * 4 * SORT_RADIX / 2) / SORT_THREADS; i++) z[i * SORT_THREADS + tid] = 0;
    }
    __syncthreads();

    uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
    int rank_in_wave[SORT_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        int idx = base + r * SORT_THREADS + tid;
        bool ok = idx < n;
        key[r] = ok ? keys_in[idx] : 0xffffffffu;
        val[r] = ok ? vals_in[idx] : 0u;
        uint32_t d = (key[r] >> shift) & mask;
        // lanes of this wave holding the same digit (and the same in-range status)
        unsigned long long peers = __ballot(ok) ^ (ok ? 0ull : ~0ull);  // ok lanes: ballot(ok); others: ~ballot(ok)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        rank_in_wave[r] = __popcll(peers & lt_mask);
        if (ok && rank_in_wave[r] == 0) cnt[r][wave][d] = (unsigned short)__popcll(peers);
    }
    __syncthreads();

    {  // thread d: exclusive prefix of digit d's counts over (round, wave); then the digit bases
        int running = 0;
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; r++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                int cval = cnt[r][w][tid];
                cnt[r][w][tid] = (unsigned short)running;
                running += cval;
            }
        int ex = block_excl_scan_256(running, lds_wave, nullptr);
        digit_start[tid] = ex;
        global_off[tid] = digit_total[SORT_RADIX + tid] + block_hist[(size_t)tid * nblocks + blockIdx.x] - ex;
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        int idx = base + r * SORT_THREADS + tid;
        if (idx < n) {
            uint32_t d = (key[r] >> shift) & mask;
            int p = digit_start[d] + (int)cnt[r][wave][d] + rank_in_wave[r];
            stage[p] = make_uint2(key[r], val[r]);
        }
    }
    __syncthreads();

    for (int p = tid; p < tile_n; p += SORT_THREADS) {
        uint2 kv = stage[p];
        int dst = global_off[(kv.x >> shift) & mask] + p;
        keys_out[dst] = kv.x;
        vals_out[dst] = kv.y;
    }
}

// ---------------------------------------------------------------------------------------------
// cell marks: mark[k] = first sorted slot holding key k (array pre-filled with 0x7f7f7f7f), and
// the particle ids in sorted-slot order for the in-cell ranking.  R key 4+4, perm 4, id 4; W 4+4.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cell_marks(const uint32_t* __restrict__ skeys,
                                                     const uint32_t* __restrict__ perm,
                                                     const uint32_t* __restrict__ idv, const int* __restrict__ d_n,
                                                     int* __restrict__ cell_start,
                                                     uint32_t* __restrict__ ids_sorted) {
    const int n = *d_n;  // particle count lives on the device (asynchronous slab steps)
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    uint32_t k = skeys[s];
    if (s == 0 || skeys[s - 1] != k) cell_start[k] = s;
    ids_sorted[s] = idv[perm[s]] & SPH_ID_MASK;
}

// Suffix-min scan over the ncell+1 marks -> monotone cell_start (empty cells inherit the start of
// the next non-empty cell).  Three small kernels: chunk minima, suffix-min of the chunk minima
// (one workgroup), apply.  Chunk = 2048 cells.

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ __launch_bounds__(256) void k_sufmin_reduce(const int* __restrict__ a, int len, int* __restrict__ part) {
    __shared__ int w[4];
    int base = blockIdx.x * SCAN_CHUNK;
    int m = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < SCAN_CHUNK / 256; r++) {
        int i = base + r * 256 + (int)threadIdx.x;
        if (i < len) m = min(m, a[i]);
    }
    m = wave_min(m);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = min(min(w[0], w[1]), min(w[2], w[3]));
}

// part[b] <- min(part[b+1 ..]) (exclusive suffix min); single workgroup, sequential over 256-chunks
// from the back.
__global__ __launch_bounds__(256) void k_sufmin_parts(int* __restrict__ part, int nparts) {
    __shared__ int buf[256];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0x7fffffff;
    __syncthreads();
    int nchunks = (nparts + 255) / 256;
    for (int ch = nchunks - 1; ch >= 0; ch--) {
        int i = ch * 256 + (int)threadIdx.x;
        int v = (i < nparts) ? part[i] : 0x7fffffff;
        buf[threadIdx.x] = v;
        __syncthreads();
        // inclusive suffix min within the chunk (Hillis-Steele, 8 steps)
        for (int d = 1; d < 256; d <<= 1) {
            int o = (threadIdx.x + d < 256) ? buf[threadIdx.x + d] : 0x7fffffff;
            __syncthreads();
            buf[threadIdx.x] = min(buf[threadIdx.x], o);
            __syncthreads();
        }
        int carry = carry_s;
        int excl = (threadIdx.x + 1 < 256) ? buf[threadIdx.x + 1] : 0x7fffffff;
        int chunk_min = buf[0];
        __syncthreads();
        if (i < nparts) part[i] = min(excl, carry);
        if (threadIdx.x == 0) carry_s = min(carry, chunk_min);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_sufmin_apply(int* __restrict__ a, int len, const int* __restrict__ part,
                                                       const int* __restrict__ d_n, int* __restrict__ n_sorted) {
    const int n = *d_n;  // particle count lives on the device (asynchronous slab steps)
    // the count at this sort: slots below it are in sorted order (k_split tells stale ghosts from appended records by it)
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_sorted = n;
    __shared__ int buf[SCAN_CHUNK];
    __shared__ int wmin[4];
    int base = blockIdx.x * SCAN_CHUNK;
    int tid = threadIdx.x;
    // thread t owns the 8 consecutive cells base + 8t .. base + 8t + 7
    int v[8];
    int m = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int i = base + r * 256 + tid;  // coalesced load into LDS
        buf[r * 256 + tid] = (i < len) ? a[i] : 0x7fffffff;
    }
    __syncthreads();
#pragma unroll
    for (int r = 7; r >= 0; r--) {
        m = min(m, buf[tid * 8 + r]);
        v[r] = m;  // inclusive suffix min within the thread's 8 cells
    }
    // exclusive suffix min across threads: wave level then workgroup level
    int lane = tid & 63, wave = tid >> 6;
    int incl = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_down(incl, d, 64);
        if (lane + d < 64) incl = min(incl, t);
    }
    if (lane == 0) wmin[wave] = incl;
    int excl_in_wave = __shfl_down(incl, 1, 64);
    if (lane == 63) excl_in_wave = 0x7fffffff;
    __syncthreads();
    int after = part[blockIdx.x];
    for (int w = 3; w > wave; w--) after = min(after, wmin[w]);
    int excl = min(excl_in_wave, after);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) buf[tid * 8 + r] = min(min(v[r], excl), n);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int i = base + r * 256 + tid;
        if (i < len) a[i] = buf[r * 256 + tid];
    }
}

// ---------------------------------------------------------------------------------------------
// In-cell ranking by id + reorder.  The LSD sort leaves ties (same cell) in the order of the
// previous step; ranking each particle among its cell-mates by id makes the order canonical
// (key, id) whatever the input order was -- which is what makes results independent of history,
// of the slab partition and of the arrival order of migrated particles.
// R: key 4, cell_start 8, ids ~cell size * 4 (cached), perm 4, state 16|24 + idv 4;  W: 16|24 + 4.
// ---------------------------------------------------------------------------------------------
#ifndef REORDER_ROUNDS
#define REORDER_ROUNDS 2
#endif
template <bool CARRY_DENSITY>
__global__ __launch_bounds__(256) void k_rank_reorder(const uint32_t* __restrict__ skeys,
                                                       const uint32_t* __restrict__ perm,
                                                       const uint32_t* __restrict__ ids_sorted,
                                                       const int* __restrict__ cell_start, const int* __restrict__ d_n,
                                                       const float2* __restrict__ pos, const float2* __restrict__ vel,
                                                       const float* __restrict__ rho, const float* __restrict__ prs,
                                                       const uint32_t* __restrict__ idv, float2* __restrict__ pos_o,
                                                       float2* __restrict__ vel_o, float* __restrict__ rho_o,
                                                       float* __restrict__ prs_o, uint32_t* __restrict__ idv_o,
                                                       int head_blocks, int* __restrict__ zero_buf, int zero_words,
                                                       int* __restrict__ zero_buf2, int zero_words2, int ncell, int skip_big) {
    if ((int)blockIdx.x < head_blocks) {
        // head workgroups: zero the idle cell table for the next build -- a small launch saved per step.  (The work
        // list of the pair kernels was built here too: its 60 registers per thread cost the latency-bound reorder
        // part half its occupancy, 33 instead of 17 us -- it is a kernel of its own now, k_tile_list_build.)
        const int i = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
        const int stride = head_blocks * (int)blockDim.x;
        for (int k = i; k < zero_words; k += stride) zero_buf[k] = 0;
        for (int k = i; k < zero_words2; k += stride) zero_buf2[k] = 0;
        return;
    }
    const int n = *d_n;  // particle count lives on the device (asynchronous slab steps)
    // The ids of the workgroup's slots go to LDS: cell-mates sit in consecutive slots, so nearly every id the
    // ranking loop below compares with is one of them.  (Read from global memory, one load instruction per cell-mate
    // with the lanes of a wave in four or five different cells, the loop kept the texture path busy for half of the
    // kernel's time: 16 M particles 0.287 ms, 2.6 TB/s.)
    // REORDER_ROUNDS slots per thread, a workgroup-width apart: the gathers of all rounds are issued before the first
    // use (the kernel is a chain of dependent loads -- perm -> state, key -> cell table -> ids -- and ran at their
    // latency with one slot per thread).
    constexpr int TILE = 256 * REORDER_ROUNDS;
    __shared__ uint32_t s_id[TILE];
    const int base = ((int)blockIdx.x - head_blocks) * TILE;
    bool live[REORDER_ROUNDS];
    uint32_t my[REORDER_ROUNDS], k[REORDER_ROUNDS], id_in[REORDER_ROUNDS];
    float2 p_in[REORDER_ROUNDS], v_in[REORDER_ROUNDS];
    float d_in[REORDER_ROUNDS], pr_in[REORDER_ROUNDS];
#pragma unroll
    for (int r = 0; r < REORDER_ROUNDS; r++) {
        const int s = base + r * 256 + (int)threadIdx.x;
        live[r] = s < n;
        my[r] = live[r] ? ids_sorted[s] : 0xffffffffu;
        s_id[r * 256 + threadIdx.x] = my[r];
    }
#pragma unroll
    for (int r = 0; r < REORDER_ROUNDS; r++) {
        const int s = base + r * 256 + (int)threadIdx.x;
        k[r] = 0xffffffffu;
        id_in[r] = 0;
        p_in[r] = v_in[r] = make_float2(0.0f, 0.0f);
        d_in[r] = pr_in[r] = 0.0f;
        if (live[r]) {
            k[r] = skeys[s];
            // (the state is gathered first: its loads depend on perm only and travel while the chain key -> cell table -> ids runs)
            const uint32_t src = perm[s];
            p_in[r] = pos[src];
            v_in[r] = vel[src];
            id_in[r] = idv[src];
            if (CARRY_DENSITY) {
                d_in[r] = rho[src];
                pr_in[r] = prs[src];
            }
        }
    }
    __syncthreads();
    int a[REORDER_ROUNDS], b[REORDER_ROUNDS];
#pragma unroll
    for (int r = 0; r < REORDER_ROUNDS; r++) {
        const bool ok = live[r] && k[r] < (uint32_t)ncell;  // (never false for a slot below the count; a stale key must not become an address)
        a[r] = ok ? cell_start[k[r]] : 0;
        b[r] = ok ? cell_start[k[r] + 1] : 0;
        live[r] = ok;
    }
#pragma unroll
    for (int r = 0; r < REORDER_ROUNDS; r++) {
        if (!live[r]) continue;
        // a crowded cell is ranked by k_rank_big (sorted ids instead of this loop over all cell-mates)
        if (skip_big && b[r] - a[r] > RANK_BIG_MIN) continue;
        int rank = 0;
        int t = a[r];
        for (const int e = min(b[r], base); t < e; t++) rank += (ids_sorted[t] < my[r]) ? 1 : 0;         // cell-mates before this workgroup's slots
        for (const int e = min(b[r], base + TILE); t < e; t++) rank += (s_id[t - base] < my[r]) ? 1 : 0;  // ... among them
        for (; t < b[r]; t++) rank += (ids_sorted[t] < my[r]) ? 1 : 0;                                  // ... behind them
        const int dst = a[r] + rank;
        pos_o[dst] = p_in[r];
        vel_o[dst] = v_in[r];
        if (CARRY_DENSITY) {
            rho_o[dst] = d_in[r];
            prs_o[dst] = pr_in[r];
        }
        idv_o[dst] = id_in[r];
    }
}

// The same ranking for crowded cells.  The loop above reads every cell-mate's id per particle: quadratic in the cell
// population (a pile-up of 16 k particles in one cell: 0.8 ms, a tenth of that scene's step).  Cells above
// RANK_BIG_MIN particles are listed by k_scan_apply; here one workgroup per listed cell sorts the cell's ids in LDS
// (bitonic, padded to a power of two) and every particle finds its rank by binary search.  Same destination slots,
// bit for bit.  A cell of more than RANK_BIG_MAX ids (128 KB of LDS) is sorted chunk by chunk, the ranks adding up.
// Launched only while the host has seen dense sub-tiles (a crowded cell always makes some), so ordinary scenes pay
// nothing.
template <bool CARRY_DENSITY>
__global__ __launch_bounds__(RANK_BIG_THREADS) void k_rank_big(
    const int* __restrict__ big_list, const int* __restrict__ big_count, int big_cap, const uint32_t* __restrict__ perm,
    const uint32_t* __restrict__ ids_sorted, const int* __restrict__ cell_start, uint32_t* __restrict__ rank_acc,
    const float2* __restrict__ pos, const float2* __restrict__ vel, const float* __restrict__ rho,
    const float* __restrict__ prs, const uint32_t* __restrict__ idv, float2* __restrict__ pos_o, float2* __restrict__ vel_o,
    float* __restrict__ rho_o, float* __restrict__ prs_o, uint32_t* __restrict__ idv_o) {
    extern __shared__ uint32_t s_ids[];
    const int tid = (int)threadIdx.x;
    const int total = min(*big_count, big_cap);
    for (int e = (int)blockIdx.x; e < total; e += (int)gridDim.x) {
        const int k = big_list[e];
        const int a = cell_start[k], b = cell_start[k + 1];
        const int np = b - a;
        if (np <= RANK_BIG_MIN) continue;  // (uniform over the workgroup; never: the list holds crowded cells only)
        // chunks of RANK_BIG_MAX ids: rank = sum over the chunks of the ids below mine (ids are unique), the partial
        // sums of a cell of more than one chunk in rank_acc (a per-slot scratch array nobody else uses here)
        for (int c0 = 0; c0 < np; c0 += RANK_BIG_MAX) {
            const int cs = min(RANK_BIG_MAX, np - c0);
            int m = RANK_BIG_MIN;
            while (m < cs) m <<= 1;
            __syncthreads();  // the previous chunk's searches are done
            for (int i = tid; i < m; i += RANK_BIG_THREADS) s_ids[i] = i < cs ? ids_sorted[a + c0 + i] : 0xffffffffu;
            __syncthreads();
            for (int size = 2; size <= m; size <<= 1) {
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = tid; t < (m >> 1); t += RANK_BIG_THREADS) {
                        const int lo = 2 * t - (t & (stride - 1));
                        const int hi = lo + stride;
                        const bool up = (lo & size) == 0;
                        const uint32_t x = s_ids[lo], y = s_ids[hi];
                        if ((x > y) == up) {
                            s_ids[lo] = y;
                            s_ids[hi] = x;
                        }
                    }
                    __syncthreads();
                }
            }
            const bool last = c0 + cs >= np;
            for (int i = tid; i < np; i += RANK_BIG_THREADS) {
                const uint32_t my = ids_sorted[a + i];
                int lo = 0, hi = cs;  // the first index whose id is not below mine = the ids of this chunk below mine
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_ids[mid] < my)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                if (c0 > 0) lo += (int)rank_acc[a + i];
                if (!last) {
                    rank_acc[a + i] = (uint32_t)lo;
                    continue;
                }
                const int dst = a + lo;
                const uint32_t src = perm[a + i];
                pos_o[dst] = pos[src];
                vel_o[dst] = vel[src];
                if (CARRY_DENSITY) {
                    rho_o[dst] = rho[src];
                    prs_o[dst] = prs[src];
                }
                idv_o[dst] = idv[src];
            }
        }
    }
}

int sph_rank_big_setup() {
    // 128 KB of dynamic LDS needs the opt-in (64 KB is the default limit)
    const int bytes = RANK_BIG_MAX * (int)sizeof(uint32_t);
    if (hipFuncSetAttribute((const void*)k_rank_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_rank_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return SPHHIP_EHIP;
    }
    return SPHHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Counting sort by cell (the default grid build).  The radix sort above orders (key, slot) pairs in
// three or four passes; but nothing downstream needs the slot order inside a cell -- k_rank_reorder
// ranks cell-mates by id anyway -- so one counting pass is enough:
//   k_cell_count    key per particle + its arrival index within the cell (atomic counter per cell;
//                   a run of equal keys in a wave -- the common case, the state is still in last
//                   step's cell order -- is counted by one atomic of its head lane)
//   k_scan_*        exclusive prefix sum of the counts -> cell_start (monotone, ncell + 1 entries)
//   k_cell_scatter  sorted slot = cell_start[key] + arrival index: keys / source slot / id per slot
// The arrival order is not deterministic; the (key, id) order after k_rank_reorder is.
// R pos 8, W key 4 + rank 4; scan 2 x 8 B per cell; scatter R 12, W 12 per particle.
// ---------------------------------------------------------------------------------------------
#ifndef COUNT_ROUNDS
#define COUNT_ROUNDS 4
#endif
// COUNT_ROUNDS particles per thread, a workgroup-width apart, stage by stage (position loads, then the run heads'
// atomics, then the stores), so that the loads and the returning atomics of the rounds are in flight together.
__global__ __launch_bounds__(256) void k_cell_count(DevConsts c, const float2* __restrict__ pos,
                                                     const int* __restrict__ d_n, uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ rank, int* __restrict__ cell_cnt,
                                                     const int* __restrict__ prev_table,
                                                     const int* __restrict__ n_sorted, int drop_lo_cell,
                                                     int drop_hi_cell, int col_min, int col_max,
                                                     int* __restrict__ err) {
    const int n = *d_n;
    const int i0 = blockIdx.x * (256 * COUNT_ROUNDS) + threadIdx.x;
    // Lazy ghost drop (slab steps): the first *n_sorted slots are still in the order of the previous sort,
    // whose table says which slot range the owned columns were; everything else among them is a stale
    // ghost copy.  Such a particle gets the key 0xffffffff, is not counted and not scattered.
    int keep_lo = 0, keep_hi = 0, sorted = 0;
    if (prev_table) {
        sorted = *n_sorted;
        keep_lo = prev_table[drop_lo_cell];
        keep_hi = prev_table[drop_hi_cell];
    }
    const int lane = sph_lane_id();
    const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);  // lanes 0..lane
    float2 p[COUNT_ROUNDS];
    bool live[COUNT_ROUNDS];  // in range and not a stale ghost
#pragma unroll
    for (int r = 0; r < COUNT_ROUNDS; r++) {
        const int i = i0 + r * 256;
        live[r] = i < n && !(i < sorted && (i < keep_lo || i >= keep_hi));
        p[r] = live[r] ? pos[i] : make_float2(0.0f, 0.0f);
    }
    uint32_t key[COUNT_ROUNDS];
    int base[COUNT_ROUNDS], start[COUNT_ROUNDS];
#pragma unroll
    for (int r = 0; r < COUNT_ROUNDS; r++) {
        const bool ok = i0 + r * 256 < n;
        bool dead = !live[r];
        key[r] = 0xffffffffu;
        if (live[r]) {
            int cx = sph_cell_coord(p[r].x, c.xmin, c.inv_cell, c.nx);
            int cy = sph_cell_coord(p[r].y, c.ymin, c.inv_cell, c.ny);
            if (cx < col_min || cx >= col_max) {  // outside the columns the cell table covers (a slab and its reach)
                dead = true;
                atomicOr(err, SPH_ERR_OUT_OF_SLAB);
            } else {
                key[r] = (uint32_t)(cx * c.ny + cy);
            }
        }
        const uint32_t prev = __shfl_up(key[r], 1, 64);
        const bool head = ok && (lane == 0 || prev != key[r]);  // (runs of dead particles are runs too, but count nothing)
        const unsigned long long heads = __ballot(head);
        const int n_ok = __popcll(__ballot(ok));  // the in-range lanes are a prefix of the wave
        const unsigned long long below = heads & upto;
        start[r] = below ? 63 - __clzll((long long)below) : 0;  // head lane of this lane's run
        base[r] = 0;
        if (head && !dead) {
            const unsigned long long above = heads & ~upto;
            const int next = above ? __ffsll((long long)above) - 1 : n_ok;
            base[r] = atomicAdd(&cell_cnt[key[r]], next - lane);
        }
    }
#pragma unroll
    for (int r = 0; r < COUNT_ROUNDS; r++) {
        const int i = i0 + r * 256;
        const int b = __shfl(base[r], start[r], 64);
        if (i < n) {
            keys[i] = key[r];
            rank[i] = (uint32_t)(b + lane - start[r]);
        }
    }
}

#ifndef SCATTER_ROUNDS
#define SCATTER_ROUNDS 4
#endif
__global__ __launch_bounds__(256) void k_cell_scatter(const uint32_t* __restrict__ keys,
                                                       const uint32_t* __restrict__ rank,
                                                       const uint32_t* __restrict__ idv, const int* __restrict__ d_n,
                                                       const int* __restrict__ cell_start,
                                                       uint32_t* __restrict__ skeys, uint32_t* __restrict__ perm,
                                                       uint32_t* __restrict__ ids_sorted, int head_blocks, int nx, int ny,
                                                       TileListArgs tl, int* __restrict__ err) {
    if ((int)blockIdx.x < head_blocks) {
        // head workgroups: the work list of the pair kernels (needs only the finished cell table, like the scatter) --
        // a launch of its own took 9 us, twice the launch floor, for 35 k threads
        sph_tile_list_thread((int)(blockIdx.x * blockDim.x + threadIdx.x), cell_start, nx, ny, tl, err);
        return;
    }
    // SCATTER_ROUNDS particles per thread, a workgroup-width apart (every access stays coalesced), all loads of a stage
    // issued before the first use: with one particle per thread the kernel ran at the latency of its chain
    // key -> cell_start[key] -> store (2.5 TB/s at 16M particles with every wave slot of the GPU filled), not at the
    // bandwidth of HBM.
    const int n = *d_n;
    const int base = ((int)blockIdx.x - head_blocks) * (256 * SCATTER_ROUNDS) + (int)threadIdx.x;
    uint32_t k[SCATTER_ROUNDS], rk[SCATTER_ROUNDS], id[SCATTER_ROUNDS];
    int cs[SCATTER_ROUNDS];
#pragma unroll
    for (int r = 0; r < SCATTER_ROUNDS; r++) {
        const int i = base + r * 256;
        const bool ok = i < n;
        k[r] = ok ? keys[i] : 0xffffffffu;  // 0xffffffff: dropped by k_cell_count, or behind the end
        rk[r] = ok ? rank[i] : 0u;
        id[r] = ok ? idv[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SCATTER_ROUNDS; r++) cs[r] = k[r] != 0xffffffffu ? cell_start[k[r]] : 0;
#pragma unroll
    for (int r = 0; r < SCATTER_ROUNDS; r++) {
        if (k[r] == 0xffffffffu) continue;
        const int slot = cs[r] + (int)rk[r];
        skeys[slot] = k[r];
        perm[slot] = (uint32_t)(base + r * 256);
        ids_sorted[slot] = id[r] & SPH_ID_MASK;
    }
}

// Exclusive prefix sum over len ints in place, chunks of SCAN_CHUNK: chunk sums, then every chunk adds
// the sums of the chunks before it (summed by the workgroup itself: at most a few thousand values).
__global__ __launch_bounds__(256) void k_scan_reduce(const int* __restrict__ a, int len, int* __restrict__ part,
                                                      int* __restrict__ zero_me) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0;  // the crowded-cell counter k_scan_apply adds to
    __shared__ int w[4];
    const int base = blockIdx.x * SCAN_CHUNK;
    int sum = 0;
#pragma unroll
    for (int r = 0; r < SCAN_CHUNK / 256; r++) {
        int i = base + r * 256 + (int)threadIdx.x;
        if (i < len) sum += a[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(256) void k_scan_apply(int* __restrict__ a, int len, const int* __restrict__ part,
                                                     int* __restrict__ zero_me, int* __restrict__ n_sorted,
                                                     int* __restrict__ n_out, int first_cell, int* __restrict__ big_list,
                                                     int* __restrict__ big_count, int big_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the tile counters of the pair kernels
        zero_me[0] = 0;
        zero_me[SPH_LONG_COUNT_OFFSET] = 0;
    }
    __shared__ int buf[SCAN_CHUNK];
    __shared__ int wsum[4];
    __shared__ int wtot[4];
    const int base = blockIdx.x * SCAN_CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int before = 0;  // sum of the chunks before this one
    for (int b = tid; b < (int)blockIdx.x; b += 256) before += part[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    if (lane == 0) wsum[wave] = before;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int i = base + r * 256 + tid;  // coalesced load into LDS
        buf[r * 256 + tid] = (i < len) ? a[i] : 0;
    }
    __syncthreads();
    before = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // thread t owns the 8 consecutive cells base + 8t .. base + 8t + 7
    int v[8];
    int run = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        v[r] = run;
        const int cnt = buf[tid * 8 + r];
        run += cnt;
        if (cnt > RANK_BIG_MIN && base + tid * 8 + r < len - 1) {  // crowded cell: ranked by k_rank_big
            const int e = atomicAdd(big_count, 1);
            if (e < big_cap) big_list[e] = first_cell + base + tid * 8 + r;
        }
    }
    int incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int off = before + incl - run;
    for (int w = 0; w < wave; w++) off += wtot[w];
#pragma unroll
    for (int r = 0; r < 8; r++) buf[tid * 8 + r] = v[r] + off;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int i = base + r * 256 + tid;
        if (i < len) a[i] = buf[r * 256 + tid];
        if (i == len - 1) {  // a[ncell] = number of particles sorted: the count after a lazy ghost drop
            *n_sorted = buf[r * 256 + tid];
            if (n_out) *n_out = buf[r * 256 + tid];
        }
    }
}

// (Measured in round 3 and not kept: the same prefix sum in ONE launch, every workgroup publishing its chunk sum and
// looking back over its predecessors' -- decoupled look-back, status bits in the sum's word, bounded polling.  All GPU
// tests stayed bit-exact; the step did not get faster at 1 M particles (0.2201 against 0.2200 ms: the look-back's
// polling round trips cost what the saved launch gains) and the scan got slower at 16 M (48 against 30 us, 4066
// chunks).)

// ---------------------------------------------------------------------------------------------
// The whole grid build of a small scene in one workgroup.  Below a few thousand particles the five
// kernels above are pure launch latency (~5 us each, dependent): here one workgroup of 1024 threads
// counts into an LDS cell table, scans it, scatters (id, source slot) into LDS, ranks cell-mates by id
// and writes the reordered state, the cell table, the tile list of the pair kernels and the zeros of the
// idle table.  Same results as the kernels above (the (key, id) order is canonical); used when the
// particles and the maintained cell range fit (SMALL_MAX_N, SMALL_MAX_CELLS) and no ghost drop is pending.
// ---------------------------------------------------------------------------------------------
constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_MAX_N = 4 * SMALL_THREADS;
constexpr int SMALL_MAX_CELLS = 4 * SMALL_THREADS;  // entries of the maintained range incl. the closing one

template <bool CARRY_DENSITY>
__global__ __launch_bounds__(SMALL_THREADS) void k_grid_small(
    DevConsts c, const int* __restrict__ d_n, const float2* __restrict__ pos, const float2* __restrict__ vel,
    const float* __restrict__ rho, const float* __restrict__ prs, const uint32_t* __restrict__ idv,
    float2* __restrict__ pos_o, float2* __restrict__ vel_o, float* __restrict__ rho_o, float* __restrict__ prs_o,
    uint32_t* __restrict__ idv_o, uint32_t* __restrict__ skeys_o, int* __restrict__ cell_start, int r_first, int r_len,
    int col_min, int col_max, int* __restrict__ tile_count, int* __restrict__ n_sorted, TileListArgs tl,
    int* __restrict__ zero_buf, int zero_words, int* __restrict__ zero_buf2, int zero_words2, int* __restrict__ err) {
    __shared__ int s_start[SMALL_MAX_CELLS + 1];  // counts, then exclusive starts; index = cell - r_first
    __shared__ uint32_t s_id[SMALL_MAX_N];
    __shared__ uint16_t s_src[SMALL_MAX_N], s_key[SMALL_MAX_N];
    __shared__ int s_wtot[SMALL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(*d_n, SMALL_MAX_N);
    for (int k = tid; k <= SMALL_MAX_CELLS; k += SMALL_THREADS) s_start[k] = 0;
    if (tid == 0) {
        tile_count[0] = 0;
        tile_count[SPH_LONG_COUNT_OFFSET] = 0;
    }
    __syncthreads();
    // keys + arrival index within the cell (k_cell_count)
    uint32_t key[4];
    int arrival[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = r * SMALL_THREADS + tid;
        key[r] = 0xffffffffu;
        arrival[r] = 0;
        if (i < n) {
            const float2 p = pos[i];
            const int cx = sph_cell_coord(p.x, c.xmin, c.inv_cell, c.nx);
            const int cy = sph_cell_coord(p.y, c.ymin, c.inv_cell, c.ny);
            if (cx < col_min || cx >= col_max) {
                atomicOr(err, SPH_ERR_OUT_OF_SLAB);
            } else {
                key[r] = (uint32_t)(cx * c.ny + cy);
                arrival[r] = atomicAdd(&s_start[(int)key[r] - r_first], 1);
            }
        }
    }
    __syncthreads();
    // exclusive scan: thread t owns entries 4t .. 4t+3 (k_scan_*)
    int v[4], run = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        v[r] = run;
        run += s_start[tid * 4 + r];
    }
    int incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    int off = incl - run;
    for (int w = 0; w < wave; w++) off += s_wtot[w];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int idx = tid * 4 + r;
        s_start[idx] = v[r] + off;
        if (idx < r_len) cell_start[r_first + idx] = v[r] + off;
        if (idx == r_len - 1) *n_sorted = v[r] + off;  // the closing entry: particles sorted
    }
    __syncthreads();
    // scatter into LDS in arrival order (k_cell_scatter)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (key[r] != 0xffffffffu) {
            const int i = r * SMALL_THREADS + tid;
            const int k = (int)key[r] - r_first;
            const int slot = s_start[k] + arrival[r];
            s_id[slot] = idv[i] & SPH_ID_MASK;
            s_src[slot] = (uint16_t)i;
            s_key[slot] = (uint16_t)k;
        }
    }
    __syncthreads();
    // rank among the cell-mates by id, reorder (k_rank_reorder)
    const int n_s = s_start[r_len - 1];
    for (int s = tid; s < n_s; s += SMALL_THREADS) {
        const int k = s_key[s];
        const int a = s_start[k], b = s_start[k + 1];
        const uint32_t my = s_id[s];
        int rank = 0;
        for (int t = a; t < b; t++) rank += (s_id[t] < my) ? 1 : 0;
        const int dst = a + rank;
        const int src = s_src[s];
        pos_o[dst] = pos[src];
        vel_o[dst] = vel[src];
        if (CARRY_DENSITY) {
            rho_o[dst] = rho[src];
            prs_o[dst] = prs[src];
        }
        idv_o[dst] = idv[src];
        skeys_o[dst] = (uint32_t)(k + r_first);
    }
    __syncthreads();  // the cell table is complete (and visible to this workgroup)
    for (int i0 = 0; i0 < tl.nthreads; i0 += SMALL_THREADS)
        sph_tile_list_thread(i0 + tid, cell_start, c.nx, c.ny, tl, err);
    for (int k = tid; k < zero_words; k += SMALL_THREADS) zero_buf[k] = 0;
    for (int k = tid; k < zero_words2; k += SMALL_THREADS) zero_buf2[k] = 0;
}

// The work list of the pair kernels for the grid just built (sphhip_internal.h: sph_tile_list_thread).  Needs only
// the finished cell table, so it runs right after the scan.
__global__ __launch_bounds__(256) void k_tile_list_build(const int* __restrict__ cell_start, int nx, int ny, TileListArgs tl,
                                                          int* __restrict__ err) {
    sph_tile_list_thread((int)(blockIdx.x * blockDim.x + threadIdx.x), cell_start, nx, ny, tl, err);
}

// ---------------------------------------------------------------------------------------------
static inline int ilog2_ceil(unsigned v) {
    int b = 0;
    while ((1u << b) < v && b < 31) b++;
    return b;
}

int sph_grid_build(SphCtx* c, bool carry_density) {
    const int n = sph_n_grid(c);  // exact count, or the capacity while the count is only known on the device
    // (for the tile-height heuristic the last count the host saw beats the capacity)
    sph_adapt_tile_cells(c, c->n_exact || c->n <= 0 ? n : c->n);
    hipStream_t st = c->stream;
    const int ncell = c->ncell;
    const int len = ncell + 1;
    const int nparts = (len + SCAN_CHUNK - 1) / SCAN_CHUNK;
    const int table_words = len + nparts;  // cell table + the partial sums of its scan
    // The counting path maintains the table only for the columns that can hold particles: the owned
    // columns, the halo and one more column on each side (what the tiles of the halo columns read).  On a
    // slab of a wide global grid that is a fraction of the table.  Entries below the range stay 0; entries
    // above it are never read by the kernels (sphhip_get_cells completes them on the host).
    const int reach = SPHHIP_HALO_COLS + 1;
    const int clo = c->col_lo - reach > 0 ? c->col_lo - reach : 0;
    const int chi = c->col_hi + reach < c->dc.nx ? c->col_hi + reach : c->dc.nx;
    const int r_first = clo * c->dc.ny;                       // first cell of the range
    const int r_len = (chi - clo) * c->dc.ny + 1;             // cells of the range + the closing entry
    const int r_parts = (r_len + SCAN_CHUNK - 1) / SCAN_CHUNK;
    bool range_moved = false;
    bool fused = false;  // k_grid_small did the whole build
    bool count_update = false;
    if (c->grid_counting) {
        {
            PhaseScope ps(c, SPHHIP_PH_HASH);
            // the range moved (rebalance): the idle table is cleaned in full now, the current one (still needed
            // by a pending ghost drop) in full after this build
            range_moved = clo != c->tab_clo || chi != c->tab_chi;
            if (range_moved) c->cell_next_zeroed = false;
            c->tab_clo = clo;
            c->tab_chi = chi;
            // switch to the idle cell table; it was zeroed by the previous build's reorder kernel
            const int next = 1 - c->cell_cur;
            if (!c->cell_next_zeroed) HIP_TRY(hipMemsetAsync(c->cell_buf[next], 0, sizeof(int) * (size_t)table_words, st));
            const int* prev_table = (c->pending_drop && c->slice_valid) ? c->cell_start : nullptr;
            c->cell_cur = next;
            c->cell_start = c->cell_buf[next];
            c->cell_next_zeroed = false;
            // The build can drop particles: the stale ghosts of a pending drop, and -- on a slab, whose cell table
            // covers only its own column range -- any particle outside that range (a non-finite position has
            // column 0).  Then the scan's total, not the old count, is the particle count from here on: the
            // slots behind it hold nothing (k_rank_reorder must not read their keys).
            count_update = c->pending_drop || clo > 0 || chi < c->dc.nx;
            fused = n > 0 && n <= SMALL_MAX_N && r_len <= SMALL_MAX_CELLS && !prev_table && !count_update && c->small_grid;
            if (fused) {
                TileListArgs tl = {};
                int lo = 0, hi = 0;
                sph_tile_list_args(c, tl, lo, hi);
                int* zero_buf = c->cell_buf[1 - c->cell_cur];
                auto kern = carry_density ? k_grid_small<true> : k_grid_small<false>;
                hipLaunchKernelGGL(kern, dim3(1), dim3(SMALL_THREADS), 0, st, c->dc, c->d_n, c->pos, c->vel, c->rho, c->prs, c->idv,
                                   c->pos2, c->vel2, c->rho2, c->prs2, c->idv2, c->keys, c->cell_start, r_first, r_len, clo, chi,
                                   c->counters + SPH_TILE_COUNT_SLOT, c->counters + SPH_N_SORTED_SLOT, tl, zero_buf + r_first,
                                   r_len, zero_buf + len, r_parts, c->d_err);
                LAUNCH_CHECK();
            } else if (n > 0) {
                hipLaunchKernelGGL(k_cell_count, dim3((n + 256 * COUNT_ROUNDS - 1) / (256 * COUNT_ROUNDS)), dim3(256), 0, st, c->dc, c->pos, c->d_n, c->keys2,
                                   c->perm2, c->cell_start, prev_table, c->counters + SPH_N_SORTED_SLOT,
                                   c->slice_lo * c->dc.ny, c->slice_hi * c->dc.ny, clo, chi, c->d_err);
                LAUNCH_CHECK();
            }
        }
        if (!fused) {
            PhaseScope ps(c, SPHHIP_PH_CELLS);
            // (the chunk sums are a kernel of their own: accumulating them with atomics in k_cell_count puts
            // thousands of atomics on a few hundred addresses, 0.4 us each -- measured 0.1 ms)
            hipLaunchKernelGGL(k_scan_reduce, dim3(r_parts), dim3(256), 0, st, c->cell_start + r_first, r_len,
                               c->cell_start + len, c->counters + SPH_BIG_COUNT_SLOT);
            hipLaunchKernelGGL(k_scan_apply, dim3(r_parts), dim3(256), 0, st, c->cell_start + r_first, r_len, c->cell_start + len,
                               c->counters + SPH_TILE_COUNT_SLOT, c->counters + SPH_N_SORTED_SLOT,
                               count_update ? sph_n_slot_next(c) : (int*)nullptr, r_first, c->big_list,
                               c->counters + SPH_BIG_COUNT_SLOT, c->big_cap);
            LAUNCH_CHECK();
            if (n > 0 && !c->tile_list_fused) {  // (k_scan_apply has zeroed the list's counter)
                TileListArgs tl = {};
                int lo = 0, hi = 0;
                sph_tile_list_args(c, tl, lo, hi);
                hipLaunchKernelGGL(k_tile_list_build, dim3((tl.nthreads + 255) / 256), dim3(256), 0, st, c->cell_start, c->dc.nx,
                                   c->dc.ny, tl, c->d_err);
                LAUNCH_CHECK();
            }
        }
        if (n > 0 && !fused) {
            PhaseScope ps(c, SPHHIP_PH_SORT);
            TileListArgs tl = {};
            int lo = 0, hi = 0, head = 0;
            if (c->tile_list_fused) {
                sph_tile_list_args(c, tl, lo, hi);
                head = (tl.nthreads + 255) / 256;
            }
            hipLaunchKernelGGL(k_cell_scatter, dim3((n + 256 * SCATTER_ROUNDS - 1) / (256 * SCATTER_ROUNDS) + head), dim3(256), 0, st, c->keys2, c->perm2, c->idv, c->d_n,
                               c->cell_start, c->keys, c->perm, c->ids_sorted, head, c->dc.nx, c->dc.ny, tl, c->d_err);
            LAUNCH_CHECK();
        }
        if (count_update) {  // the scan wrote the count without the dropped particles into the next ring slot
            sph_n_advance(c);
            c->n_exact = false;
            c->pending_drop = false;
        }
    } else {
        if (n > 0) {
            {
                PhaseScope ps(c, SPHHIP_PH_HASH);
                hipLaunchKernelGGL(k_hash, dim3((n + 255) / 256), dim3(256), 0, st, c->dc, c->pos, c->d_n, c->keys, c->perm);
                LAUNCH_CHECK();
            }
            {
                PhaseScope ps(c, SPHHIP_PH_SORT);
                int key_bits = ilog2_ceil((unsigned)ncell);
                if (key_bits < 1) key_bits = 1;
                int passes = (key_bits + 7) / 8;
                int bits_per_pass = (key_bits + passes - 1) / passes;
                int nblocks = (n + SORT_TILE - 1) / SORT_TILE;
                int shift = 0;
                for (int p = 0; p < passes; p++) {
                    int bits = min(bits_per_pass, key_bits - shift);
                    uint32_t mask = (1u << bits) - 1u;
                    hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(SORT_THREADS), 0, st, c->keys, c->d_n, shift, mask,
                                       c->block_hist, nblocks);
                    hipLaunchKernelGGL(k_sort_scan_rows, dim3(SORT_RADIX), dim3(256), 0, st, c->block_hist, nblocks,
                                       c->digit_total);
                    hipLaunchKernelGGL(k_sort_scan_digits, dim3(1), dim3(256), 0, st, c->digit_total);
                    hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(SORT_THREADS), 0, st, c->keys, c->perm,
                                       c->keys2, c->perm2, c->d_n, shift, mask, c->block_hist, c->digit_total, nblocks);
                    LAUNCH_CHECK();
                    std::swap(c->keys, c->keys2);
                    std::swap(c->perm, c->perm2);
                    shift += bits;
                }
            }
        }
        {
            PhaseScope ps(c, SPHHIP_PH_CELLS);
            c->cell_next_zeroed = false;  // (this path leaves both tables dirty)
            c->tab_clo = c->tab_chi = -1;
            HIP_TRY(hipMemsetAsync(c->cell_start, 0x7f, sizeof(int) * (size_t)len, st));
            if (n > 0) {
                hipLaunchKernelGGL(k_cell_marks, dim3((n + 255) / 256), dim3(256), 0, st, c->keys, c->perm, c->idv, c->d_n,
                                   c->cell_start, c->ids_sorted);
            }
            hipLaunchKernelGGL(k_sufmin_reduce, dim3(nparts), dim3(256), 0, st, c->cell_start, len, c->scan_tmp);
            hipLaunchKernelGGL(k_sufmin_parts, dim3(1), dim3(256), 0, st, c->scan_tmp, nparts);
            hipLaunchKernelGGL(k_sufmin_apply, dim3(nparts), dim3(256), 0, st, c->cell_start, len, c->scan_tmp, c->d_n,
                               c->counters + SPH_N_SORTED_SLOT);
            LAUNCH_CHECK();
        }
    }
    c->tiles_valid = false;
    if (n > 0) {
        PhaseScope ps(c, SPHHIP_PH_REORDER);
        const int nb = (n + 256 * REORDER_ROUNDS - 1) / (256 * REORDER_ROUNDS);
        // head workgroups (counting path only): zeroing of the idle cell table
        int head = 0, lo = 0, hi = 0;
        if (c->grid_counting) {
            TileListArgs tl = {};
            sph_tile_list_args(c, tl, lo, hi);  // (lo, hi: the column range of the list built above)
            head = std::min(64, (r_len + 2047) / 2048 + 1);
        }
        int* zero_buf = c->cell_buf[1 - c->cell_cur];
        // crowded cells (listed by this build's k_scan_apply) go to k_rank_big while the host has seen dense sub-tiles
        // -- a crowded cell always makes some -- or does not know yet; a cell that becomes crowded in between is
        // ranked by k_rank_reorder's loop for a step: the same result, only slower
        const bool big = c->grid_counting && !fused && c->rank_big && ((volatile int*)c->h_dense)[0] != 0;
        if (fused) {
            // (k_grid_small has reordered, listed the tiles and zeroed the idle table)
        } else if (carry_density)
            hipLaunchKernelGGL(k_rank_reorder<true>, dim3(nb + head), dim3(256), 0, st, c->keys, c->perm, c->ids_sorted,
                               c->cell_start, c->d_n, c->pos, c->vel, c->rho, c->prs, c->idv, c->pos2, c->vel2, c->rho2,
                               c->prs2, c->idv2, head, zero_buf + r_first, r_len, zero_buf + len, r_parts, ncell, big ? 1 : 0);
        else
            hipLaunchKernelGGL(k_rank_reorder<false>, dim3(nb + head), dim3(256), 0, st, c->keys, c->perm, c->ids_sorted,
                               c->cell_start, c->d_n, c->pos, c->vel, c->rho, c->prs, c->idv, c->pos2, c->vel2, c->rho2,
                               c->prs2, c->idv2, head, zero_buf + r_first, r_len, zero_buf + len, r_parts, ncell, big ? 1 : 0);
        LAUNCH_CHECK();
        if (big) {
            const int blocks = std::min(c->big_cap, 2 * c->num_cus);
            const size_t lds = sizeof(uint32_t) * RANK_BIG_MAX;
            if (carry_density)
                hipLaunchKernelGGL(k_rank_big<true>, dim3(blocks), dim3(RANK_BIG_THREADS), lds, st, c->big_list, c->counters + SPH_BIG_COUNT_SLOT,
                                   c->big_cap, c->perm, c->ids_sorted, c->cell_start, c->keys2, c->pos, c->vel, c->rho, c->prs, c->idv, c->pos2,
                                   c->vel2, c->rho2, c->prs2, c->idv2);
            else
                hipLaunchKernelGGL(k_rank_big<false>, dim3(blocks), dim3(RANK_BIG_THREADS), lds, st, c->big_list, c->counters + SPH_BIG_COUNT_SLOT,
                                   c->big_cap, c->perm, c->ids_sorted, c->cell_start, c->keys2, c->pos, c->vel, c->rho, c->prs, c->idv, c->pos2,
                                   c->vel2, c->rho2, c->prs2, c->idv2);
            LAUNCH_CHECK();
        }
        if (head > 0 && range_moved) {  // the head zeroed only the new range of the old table
            HIP_TRY(hipMemsetAsync(zero_buf, 0, sizeof(int) * (size_t)table_words, st));
        }
        if (head > 0) {
            c->cell_next_zeroed = true;
            sph_tiles_mark(c, lo, hi);
        }
        std::swap(c->pos, c->pos2);
        std::swap(c->vel, c->vel2);
        if (carry_density) {
            std::swap(c->rho, c->rho2);
            std::swap(c->prs, c->prs2);
        }
        std::swap(c->idv, c->idv2);
    }
    c->grid_valid = true;
    c->counts_valid = false;
    c->tail_appended = false;
    // remember which slot range the owned columns are (ghost dropping on slabs)
    c->slice_valid = true;
    c->slice_lo = c->col_lo;
    c->slice_hi = c->col_hi;
    return SPHHIP_OK;
}
