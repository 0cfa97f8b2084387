// pairs.hip -- density/pressure and force pair sums over the cell list, plus integrate, for gfx950.
//
// Arithmetic follows the reference's serial step expression by expression in IEEE binary32
// (built with -ffp-contract=off; hipcc's `/` and sqrtf are correctly rounded by default):
//   density/pressure  serial_computeDensityPressure  SPH/sph.cpp:41-55  (kernels :7-12)
//   forces            serial_computeForces           SPH/sph.cpp:57-99  (kernels :15-39)
//   integrate         serial_integrate               SPH/sph.cpp:101-130
// The only difference from the serial loop is the one every parallel variant of the reference has
// (SPH/MPI_sph.cpp:144-147,181): neighbour velocities come from the state before the force pass.
// The j-sum of each particle runs sequentially in the canonical neighbour order (columns cx-1, cx,
// cx+1; inside a column the contiguous slot range of cells cy-1..cy+1), which is also the order of
// oracle/sph_oracle.c's cell-list functions: the two agree bit for bit.
//
// Work decomposition: one workgroup per tile = `tile_cells` consecutive cells of one grid column.
// With column-major keys the tile's particles are one contiguous slot range, and its neighbour
// candidates are three contiguous slot ranges (one per column), staged once in LDS as SoA records
// and read by every lane of the tile.  Lanes of one cell read the same LDS addresses (broadcast).
// If a tile's candidates exceed the LDS budget the same loops read global memory instead.
// No MFMA: the pair sums are irregular, distance-predicated gathers, not a contraction.
#include "sphhip_internal.h"

// All LDS of the pair kernels is one dynamic array (16-byte aligned base, no static __shared__ in
// front of it): [float4 aux[cap]] [float2 pos[cap]] [int cs[3][CS_STRIDE]].
constexpr int CS_STRIDE = PAIR_MAX_TILE_CELLS + 3;

struct TileGeom {
    int col;          // grid column of the tile
    int cy0, cy1;     // owned cells [cy0, cy1)
    int ylo, yhi;     // candidate cells [ylo, yhi)
    int own_lo, own_hi;
    int g_lo[3], len[3], base[3];  // candidate slot ranges per column and their LDS offsets
    int total;
};

__device__ __forceinline__ void tile_geometry(const DevConsts& c, const int* __restrict__ cell_start, int col,
                                              int trow, int tile_cells, TileGeom& g) {
    g.col = col;
    g.cy0 = trow * tile_cells;
    g.cy1 = min(g.cy0 + tile_cells, c.ny);
    g.ylo = max(g.cy0 - 1, 0);
    g.yhi = min(g.cy1 + 1, c.ny);
    g.own_lo = cell_start[g.col * c.ny + g.cy0];
    g.own_hi = cell_start[g.col * c.ny + g.cy1];
    g.total = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int cc = g.col - 1 + d;
        bool ok = (cc >= 0) && (cc < c.nx);
        int lo = ok ? cell_start[cc * c.ny + g.ylo] : 0;
        int hi = ok ? cell_start[cc * c.ny + g.yhi] : 0;
        g.g_lo[d] = lo;
        g.len[d] = hi - lo;
        g.base[d] = g.total;
        g.total += hi - lo;
    }
}

// Stage the cell_start entries of the three candidate columns: s_cs[d][t] = cell_start of cell
// (col-1+d, ylo+t), t = 0..(yhi-ylo) inclusive.
__device__ __forceinline__ void stage_cell_starts(const DevConsts& c, const int* __restrict__ cell_start,
                                                  const TileGeom& g, int* s_cs) {
    int rows = g.yhi - g.ylo + 1;
    for (int t = threadIdx.x; t < 3 * rows; t += PAIR_THREADS) {
        int d = t / rows, y = t - d * rows;
        int cc = g.col - 1 + d;
        bool ok = (cc >= 0) && (cc < c.nx);
        s_cs[d * CS_STRIDE + y] = ok ? cell_start[cc * c.ny + g.ylo + y] : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// density + pressure + neighbour count.   Algorithmic HBM bytes / particle: R pos 8, W rho 4 + P 4
// (+ 4 for the neighbour count) = 16 (+4).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PAIR_THREADS) void k_density(DevConsts c, const float2* __restrict__ pos,
                                                           const uint32_t* __restrict__ skeys,
                                                           const int* __restrict__ cell_start,
                                                           float* __restrict__ rho, float* __restrict__ prs,
                                                           int* __restrict__ ncount, int col0, int ncols,
                                                           int tiles_per_col, int tile_cells, int cap) {
    extern __shared__ __align__(16) float4 s_dyn[];
    float2* s_pos = reinterpret_cast<float2*>(s_dyn);
    int* s_cs = reinterpret_cast<int*>(s_pos + cap);
    int tcol, trow;
    if (!sph_tile_of_block(blockIdx.x, ncols, tiles_per_col, tcol, trow)) return;
    TileGeom g;
    tile_geometry(c, cell_start, col0 + tcol, trow, tile_cells, g);
    if (g.own_lo == g.own_hi) return;  // uniform: empty tile
    const bool use_lds = g.total <= cap;
    if (use_lds) {
#pragma unroll
        for (int d = 0; d < 3; d++)
            for (int t = threadIdx.x; t < g.len[d]; t += PAIR_THREADS) s_pos[g.base[d] + t] = pos[g.g_lo[d] + t];
    }
    stage_cell_starts(c, cell_start, g, s_cs);
    __syncthreads();

    for (int s = g.own_lo + (int)threadIdx.x; s < g.own_hi; s += PAIR_THREADS) {
        const float2 pi = pos[s];
        const int cy = (int)skeys[s] - g.col * c.ny;
        const int y0 = max(cy - 1, 0) - g.ylo;
        const int y1 = min(cy + 1, c.ny - 1) + 1 - g.ylo;
        float dens = 0.0f;
        int cnt = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (g.len[d] == 0) continue;
            int lo = s_cs[d * CS_STRIDE + y0], hi = s_cs[d * CS_STRIDE + y1];
            if (use_lds) {
                int off = g.base[d] - g.g_lo[d];
                for (int j = lo + off; j < hi + off; j++) {
                    float2 q = s_pos[j];
                    float rx = pi.x - q.x, ry = pi.y - q.y;
                    float r2 = rx * rx + ry * ry;
                    float term = c.h2 - r2;
                    float w = c.mass * (c.poly6 * term * term * term);  // MASS * poly6(r2, H), sph.cpp:11,49
                    bool in = r2 < c.h2;                                // sph.cpp:48
                    dens += in ? w : 0.0f;                              // x + 0.0f == x: same bits as the branch
                    cnt += in ? 1 : 0;
                }
            } else {
                for (int j = lo; j < hi; j++) {
                    float2 q = pos[j];
                    float rx = pi.x - q.x, ry = pi.y - q.y;
                    float r2 = rx * rx + ry * ry;
                    float term = c.h2 - r2;
                    float w = c.mass * (c.poly6 * term * term * term);
                    bool in = r2 < c.h2;
                    dens += in ? w : 0.0f;
                    cnt += in ? 1 : 0;
                }
            }
        }
        float pr = c.stiffness * (dens - c.rest_density);  // sph.cpp:52-53
        if (pr < 0.0f) pr = 0.0f;
        rho[s] = dens;
        prs[s] = pr;
        ncount[s] = cnt;
    }
}

// ---------------------------------------------------------------------------------------------
// One j-term of the force sum, SPH/sph.cpp:64-75.  (rx, ry) = p_i - p_j.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void force_term(const DevConsts& c, float rx, float ry, float r2, float pres_i,
                                           float vix, float viy, float4 aj /* vx, vy, rho, P of j */, float& fx,
                                           float& fy) {
    float r_len = sqrtf(r2);                                   // sph.cpp:67
    // spiky gradient, sph.cpp:15-20.  `r_len < 1e-6` there is a double compare: true for
    // r_len <= 1e-6f (the float below 1e-6), false from the next float up.
    bool zero_grad = (r_len > c.h) || (r_len <= 1e-6f);
    float term = c.h - r_len;
    float gcoeff = c.spiky * term * term / r_len;
    float gx = zero_grad ? 0.0f : gcoeff * rx;
    float gy = zero_grad ? 0.0f : gcoeff * ry;
    float pressureTerm = (pres_i + aj.w) / (2.0f * aj.z);     // sph.cpp:69
    float pfx = c.neg_mass * pressureTerm * gx;                // sph.cpp:70
    float pfy = c.neg_mass * pressureTerm * gy;
    float relx = aj.x - vix, rely = aj.y - viy;                // sph.cpp:71
    float lap = (r_len > c.h) ? 0.0f : c.visc_lap * (c.h - r_len);  // sph.cpp:23-26
    float viscForce = c.visc_mass * lap / aj.z;                // sph.cpp:72
    // Lennard-Jones, sph.cpp:29-39
    bool zero_lj = (r_len >= c.lj_cut) || (r_len < 1e-6f);
    float inv_r = 1.0f / r_len;
    float sr = c.sigma * inv_r;
    float sr2 = sr * sr;
    float sr6 = sr2 * sr2 * sr2;
    float sr12 = sr6 * sr6;
    float inv_r2 = inv_r * inv_r;
    float lcoeff = c.lj24 * (2.0f * sr12 - sr6) * inv_r2;
    float ljx = zero_lj ? 0.0f : lcoeff * rx;
    float ljy = zero_lj ? 0.0f : lcoeff * ry;
    fx += pfx + viscForce * relx + ljx;                        // sph.cpp:74-75
    fy += pfy + viscForce * rely + ljy;
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
template <bool FUSE>
__global__ __launch_bounds__(PAIR_THREADS) void k_force(DevConsts c, const float2* __restrict__ pos,
                                                         const float2* __restrict__ vel,
                                                         const float* __restrict__ rho,
                                                         const float* __restrict__ prs,
                                                         const uint32_t* __restrict__ idv,
                                                         const uint32_t* __restrict__ skeys,
                                                         const int* __restrict__ cell_start,
                                                         float2* __restrict__ pos_o, float2* __restrict__ vel_o,
                                                         int col0, int ncols, int tiles_per_col, int tile_cells,
                                                         int cap, int own_lo_col, int own_hi_col, float mx, float my,
                                                         float strength) {
    extern __shared__ __align__(16) float4 s_dyn[];  // [cap] float4 aux, [cap] float2 pos, cell starts
    float4* s_aux = s_dyn;
    float2* s_pos = reinterpret_cast<float2*>(s_dyn + cap);
    int* s_cs = reinterpret_cast<int*>(s_pos + cap);

    int tcol, trow;
    if (!sph_tile_of_block(blockIdx.x, ncols, tiles_per_col, tcol, trow)) return;
    TileGeom g;
    tile_geometry(c, cell_start, col0 + tcol, trow, tile_cells, g);
    if (g.own_lo == g.own_hi) return;
    if (g.col < own_lo_col || g.col >= own_hi_col) {  // ghost column: pass through
        for (int s = g.own_lo + (int)threadIdx.x; s < g.own_hi; s += PAIR_THREADS) {
            vel_o[s] = vel[s];
            if (FUSE) pos_o[s] = pos[s];
        }
        return;
    }
    const bool use_lds = g.total <= cap;
    const float inf = __builtin_inff();
    if (use_lds) {
#pragma unroll
        for (int d = 0; d < 3; d++)
            for (int t = threadIdx.x; t < g.len[d]; t += PAIR_THREADS) {
                int j = g.g_lo[d] + t;
                float2 q = pos[j];
                float2 w = vel[j];
                // `!particles[j].valid` (sph.cpp:63): an invalid j is staged at +inf so that
                // r2 = +inf >= H*H skips it with the same compare that skips far particles.
                if (idv[j] & SPH_INVALID_BIT) q = make_float2(inf, inf);
                s_pos[g.base[d] + t] = q;
                s_aux[g.base[d] + t] = make_float4(w.x, w.y, rho[j], prs[j]);
            }
    }
    stage_cell_starts(c, cell_start, g, s_cs);
    __syncthreads();

    for (int s = g.own_lo + (int)threadIdx.x; s < g.own_hi; s += PAIR_THREADS) {
        float2 pi = pos[s];
        float2 vi = vel[s];
        if (idv[s] & SPH_INVALID_BIT) {  // sph.cpp:59,103: invalid particles are left alone
            vel_o[s] = vi;
            if (FUSE) pos_o[s] = pi;
            continue;
        }
        const float dens_i = rho[s], pres_i = prs[s];
        const int cy = (int)skeys[s] - g.col * c.ny;
        const int y0 = max(cy - 1, 0) - g.ylo;
        const int y1 = min(cy + 1, c.ny - 1) + 1 - g.ylo;
        float fx = 0.0f, fy = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (g.len[d] == 0) continue;
            int lo = s_cs[d * CS_STRIDE + y0], hi = s_cs[d * CS_STRIDE + y1];
            if (use_lds) {
                int off = g.base[d] - g.g_lo[d];
                int self = (d == 1) ? s + off : -1;
                for (int j = lo + off; j < hi + off; j++) {
                    float2 q = s_pos[j];
                    float rx = pi.x - q.x, ry = pi.y - q.y;
                    float r2 = rx * rx + ry * ry;
                    if (!(r2 >= c.h2) && j != self)  // sph.cpp:63,66
                        force_term(c, rx, ry, r2, pres_i, vi.x, vi.y, s_aux[j], fx, fy);
                }
            } else {
                int self = (d == 1) ? s : -1;
                for (int j = lo; j < hi; j++) {
                    float2 q = pos[j];
                    if (idv[j] & SPH_INVALID_BIT) q = make_float2(inf, inf);
                    float rx = pi.x - q.x, ry = pi.y - q.y;
                    float r2 = rx * rx + ry * ry;
                    if (!(r2 >= c.h2) && j != self) {
                        float2 w = vel[j];
                        force_term(c, rx, ry, r2, pres_i, vi.x, vi.y, make_float4(w.x, w.y, rho[j], prs[j]), fx, fy);
                    }
                }
            }
        }
        force_epilogue<FUSE>(c, fx, fy, pi, vi, dens_i, mx, my, strength);
        if (FUSE) {
            integrate_one(c, pi, vi);
            pos_o[s] = pi;
        }
        vel_o[s] = vi;
    }
}

// Stand-alone integrate (serial_integrate, sph.cpp:101-130).  R pos 8 + vel 8 + idv 4, W 16.
__global__ __launch_bounds__(256) void k_integrate(DevConsts c, float2* __restrict__ pos, float2* __restrict__ vel,
                                                    const uint32_t* __restrict__ idv,
                                                    const uint32_t* __restrict__ skeys, int n, int use_cols,
                                                    int own_lo_col, int own_hi_col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
static TileMap tile_launch_dims(const SphCtx* c, int col_lo, int col_hi) {
    return sph_tile_map(col_hi - col_lo, (c->dc.ny + c->tile_cells - 1) / c->tile_cells);
}

int sph_density_launch(SphCtx* c) {
    if (c->n == 0) return SPHHIP_OK;
    PhaseScope ps(c, SPHHIP_PH_DENSITY);
    // one column wider than the owned range: ghost densities are recomputed locally
    int lo = c->col_lo > 0 ? c->col_lo - 1 : 0;
    int hi = c->col_hi < c->dc.nx ? c->col_hi + 1 : c->dc.nx;
    TileMap tm = tile_launch_dims(c, lo, hi);
    size_t lds = sizeof(float2) * (size_t)c->pair_cap + sizeof(int) * 3 * CS_STRIDE;
    hipLaunchKernelGGL(k_density, dim3(tm.nblocks), dim3(PAIR_THREADS), lds, c->stream, c->dc, c->pos, c->keys,
                       c->cell_start, c->rho, c->prs, c->ncount, lo, tm.ncols, tm.tiles_per_col, c->tile_cells,
                       c->pair_cap);
    LAUNCH_CHECK();
    c->counts_valid = true;
    return SPHHIP_OK;
}

int sph_force_launch(SphCtx* c, float mx, float my, float strength, bool fuse_integrate) {
    if (c->n == 0) return SPHHIP_OK;
    PhaseScope ps(c, SPHHIP_PH_FORCE);
    TileMap tm = tile_launch_dims(c, 0, c->dc.nx);  // all columns: ghost tiles pass through
    size_t lds = (sizeof(float4) + sizeof(float2)) * (size_t)c->pair_cap + sizeof(int) * 3 * CS_STRIDE;
    if (fuse_integrate) {
        hipLaunchKernelGGL(k_force<true>, dim3(tm.nblocks), dim3(PAIR_THREADS), lds, c->stream, c->dc, c->pos, c->vel,
                           c->rho, c->prs, c->idv, c->keys, c->cell_start, c->pos2, c->vel2, 0, tm.ncols,
                           tm.tiles_per_col, c->tile_cells, c->pair_cap, c->col_lo, c->col_hi, mx, my, strength);
        LAUNCH_CHECK();
        std::swap(c->pos, c->pos2);
        std::swap(c->vel, c->vel2);
        c->grid_valid = false;
    } else {
        hipLaunchKernelGGL(k_force<false>, dim3(tm.nblocks), dim3(PAIR_THREADS), lds, c->stream, c->dc, c->pos, c->vel,
                           c->rho, c->prs, c->idv, c->keys, c->cell_start, c->pos2, c->vel2, 0, tm.ncols,
                           tm.tiles_per_col, c->tile_cells, c->pair_cap, c->col_lo, c->col_hi, mx, my, strength);
        LAUNCH_CHECK();
        std::swap(c->vel, c->vel2);
    }
    return SPHHIP_OK;
}

int sph_integrate_launch(SphCtx* c) {
    if (c->n == 0) return SPHHIP_OK;
    PhaseScope ps(c, SPHHIP_PH_INTEGRATE);
    bool restrict_cols = c->grid_valid && (c->col_lo > 0 || c->col_hi < c->dc.nx);
    hipLaunchKernelGGL(k_integrate, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->dc, c->pos, c->vel, c->idv,
                       c->keys, c->n, restrict_cols ? 1 : 0, c->col_lo, c->col_hi);
    LAUNCH_CHECK();
    c->grid_valid = false;
    return SPHHIP_OK;
}
