/*
 * sph_oracle.c -- CPU oracle for the SPH step.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference's algorithm for the step hot path.  It is
 * the checker the HIP path is compared with; it is never the thing measured or shipped.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (sph_amd/, libsphhip.so) must not import, link or call anything in oracle/.
 *
 * Parity status: PINNED.  The Gauss-Seidel all-pairs functions are checked bit for bit against
 * the reference's own serial source (SPH/sph.cpp, compiled unmodified into oracle/_ref by
 * oracle/Makefile) in tests/test_oracle_pins.py, and against the committed golden vectors under
 * tests/golden/ (generated from oracle/_ref by oracle/make_golden.py).  The reference ships no
 * tests or fixtures of its own.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * tree).  Arithmetic is IEEE binary32 evaluated in the reference's expression order; build with
 * -ffp-contract=off and without -march / -ffast-math (FMA contraction changes bits).
 *
 * Semantics switches (flags argument):
 *   ORC_GAUSS_SEIDEL  velocity updated in place inside the force loop, exactly as
 *                     serial_computeForces does (SPH/sph.cpp:96-97): particle i sees the already
 *                     updated velocity of every j < i.
 *   ORC_SNAPSHOT      neighbour velocities are read from a snapshot taken before the force loop,
 *                     as mpi_computeStep does (SPH/MPI_sph.cpp:144-147,181).  This is what any
 *                     parallel implementation (the HIP path included) computes.
 *   ORC_EPS_F32       spell the spiky-gradient guard as `r_len < 1e-6f` (MPI_sph.cpp:71,
 *                     openMP_sph.cpp:57) instead of the serial `r_len < 1e-6` (double literal,
 *                     sph.cpp:16).  They differ only for r_len == 1e-6f exactly.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/sphhip.h" /* SphParams POD only */

#define ORC_GAUSS_SEIDEL 0
#define ORC_SNAPSHOT 1
#define ORC_EPS_F32 2

typedef struct {
    float px, py, vx, vy, density, pressure;
    unsigned char valid;
    unsigned char pad[3];
} orc_particle; /* SPH/sph.h:27-33, sizeof == 28 */

typedef char orc_particle_size_check[(sizeof(orc_particle) == 28) ? 1 : -1];

/* SPH/sph.h:11-20 and the literals of SPH/sph.cpp:77,81-84,91-92,113-128 */
void orc_default_params(SphParams* p) {
    p->h = 0.15f;
    p->dt = 0.0005f;
    p->mass = 0.05f;
    p->viscosity = 0.3f;
    p->stiffness = 1000.0f;
    p->rest_density = 1000.0f;
    p->damping = 0.9f;
    p->max_vel = 5.0f;
    p->sigma = 0.05f;
    p->epsilon = 1.0f;
    p->gravity = 10.0f;
    p->force_cap = 1000.0f;
    p->mouse_radius = 0.25f;
    p->xmin = 0.05f;
    p->xmax = 1.95f;
    p->ymin = 0.05f;
    p->ymax = 1.95f;
    p->cell_size = 0.0f;
}

/* ---- the four pair kernels (SPH/sph.cpp:7-39) ------------------------------------------ */

/* serial_poly6, SPH/sph.cpp:7-12 */
float orc_poly6(float r2, float h) {
    float h2 = h * h;
    if (r2 > h2) return 0.0f;
    float term = h2 - r2;
    return (315.0f / (64.0f * 3.14159f * powf(h, 9))) * term * term * term;
}

/* serial_spikyGrad, SPH/sph.cpp:15-20; returns the scalar coefficient (gradient = coeff * r).
 * eps_f32 != 0 selects the MPI/OpenMP spelling of the guard. */
float orc_spiky_coeff(float r_len, float h, int eps_f32) {
    if (eps_f32) {
        if (r_len > h || r_len < 1e-6f) return 0.0f;
    } else {
        if (r_len > h || r_len < 1e-6) return 0.0f; /* double compare, as written */
    }
    float term = h - r_len;
    float coeff = -45.0f / (3.14159f * powf(h, 6)) * term * term / r_len;
    return coeff;
}

/* serial_viscosityLaplacian, SPH/sph.cpp:23-26 */
float orc_visc_lap(float r, float h) {
    if (r > h) return 0.0f;
    return 45.0f / (3.14159f * powf(h, 6)) * (h - r);
}

/* serial_ljForce, SPH/sph.cpp:29-39; returns the scalar coefficient (force = coeff * r) */
float orc_lj_coeff(float r_len, float sigma, float epsilon) {
    if (r_len >= 2.5f * sigma || r_len < 1e-6f) return 0.0f;
    float inv_r = 1.0f / r_len;
    float sigma_over_r = sigma * inv_r;
    float sigma_over_r2 = sigma_over_r * sigma_over_r;
    float sigma_over_r6 = sigma_over_r2 * sigma_over_r2 * sigma_over_r2;
    float sigma_over_r12 = sigma_over_r6 * sigma_over_r6;
    float inv_r2 = inv_r * inv_r;
    float coeff = 24.0f * epsilon * (2.0f * sigma_over_r12 - sigma_over_r6) * inv_r2;
    return coeff;
}

/* Constants hoisted out of the loops with the same expression order; the reference recomputes
 * them per call from compile-time constants, which folds to the same binary32 values. */
typedef struct {
    float h, h2, poly6c, spikyc, viscc, visc_mass, neg_mass, lj_cut, lj24;
    int eps_f32;
} orc_consts;

static orc_consts orc_make_consts(const SphParams* p, int flags) {
    orc_consts c;
    c.h = p->h;
    c.h2 = p->h * p->h;
    c.poly6c = 315.0f / (64.0f * 3.14159f * powf(p->h, 9));
    c.spikyc = -45.0f / (3.14159f * powf(p->h, 6));
    c.viscc = 45.0f / (3.14159f * powf(p->h, 6));
    c.visc_mass = p->viscosity * p->mass;
    c.neg_mass = -p->mass;
    c.lj_cut = 2.5f * p->sigma;
    c.lj24 = 24.0f * p->epsilon;
    c.eps_f32 = (flags & ORC_EPS_F32) ? 1 : 0;
    return c;
}

/* One j-term of the density sum, SPH/sph.cpp:46-50 */
static inline float orc_density_term(const orc_consts* c, const SphParams* p, float r2) {
    float term = c->h2 - r2; /* poly6: r2 > h2 cannot hold here because r2 < H*H */
    return p->mass * (c->poly6c * term * term * term);
}

/* One j-term of the force sum, SPH/sph.cpp:64-75.  (rx,ry) = p_i - p_j with r2 < H*H. */
static inline void orc_force_term(const orc_consts* c, const SphParams* p, float rx, float ry, float r2,
                                  float pres_i, float pres_j, float dens_j, float vix, float viy, float vjx,
                                  float vjy, float* fx, float* fy) {
    float r_len = sqrtf(r2);
    float gx, gy;
    int zero_grad;
    if (c->eps_f32)
        zero_grad = (r_len > c->h || r_len < 1e-6f);
    else
        zero_grad = (r_len > c->h || r_len < 1e-6);
    if (zero_grad) {
        gx = 0.0f;
        gy = 0.0f;
    } else {
        float term = c->h - r_len;
        float coeff = c->spikyc * term * term / r_len;
        gx = coeff * rx;
        gy = coeff * ry;
    }
    float pressureTerm = (pres_i + pres_j) / (2.0f * dens_j);
    float pfx = c->neg_mass * pressureTerm * gx;
    float pfy = c->neg_mass * pressureTerm * gy;
    float relx = vjx - vix;
    float rely = vjy - viy;
    float lap = (r_len > c->h) ? 0.0f : c->viscc * (c->h - r_len);
    float viscForce = c->visc_mass * lap / dens_j;
    float ljx, ljy;
    if (r_len >= c->lj_cut || r_len < 1e-6f) {
        ljx = 0.0f;
        ljy = 0.0f;
    } else {
        float inv_r = 1.0f / r_len;
        float s = p->sigma * inv_r;
        float s2 = s * s;
        float s6 = s2 * s2 * s2;
        float s12 = s6 * s6;
        float inv_r2 = inv_r * inv_r;
        float coeff = c->lj24 * (2.0f * s12 - s6) * inv_r2;
        ljx = coeff * rx;
        ljy = coeff * ry;
    }
    *fx += pfx + viscForce * relx + ljx;
    *fy += pfy + viscForce * rely + ljy;
}

/* Everything after the j loop of serial_computeForces, SPH/sph.cpp:77-97.  Updates (vx,vy). */
static inline void orc_force_epilogue(const SphParams* p, float fx, float fy, float px, float py, float dens,
                                      float mx, float my, float strength, float* vx, float* vy) {
    fy -= p->gravity * dens;
    if (strength != 0.0f) {
        float rx = px - mx, ry = py - my;
        float r2 = rx * rx + ry * ry;
        if (r2 < p->mouse_radius * p->mouse_radius) {
            float r_len = sqrtf(r2);
            if (r_len > 1e-6f) {
                float forceMag = strength * (p->mouse_radius - r_len) / p->mouse_radius;
                fx += forceMag * rx / r_len;
                fy += forceMag * ry / r_len;
            }
        }
    }
    float forceMag = sqrtf(fx * fx + fy * fy);
    if (forceMag > p->force_cap * dens) {
        float scale = p->force_cap * dens / forceMag;
        fx *= scale;
        fy *= scale;
    }
    *vx += p->dt * fx / dens;
    *vy += p->dt * fy / dens;
}

/* Body of serial_integrate for one particle, SPH/sph.cpp:105-128 */
static inline void orc_integrate_one(const SphParams* p, float* px, float* py, float* vx, float* vy) {
    *px += p->dt * *vx;
    *py += p->dt * *vy;
    float velMag = sqrtf(*vx * *vx + *vy * *vy);
    if (velMag > p->max_vel) {
        float scale = p->max_vel / velMag;
        *vx *= scale;
        *vy *= scale;
    }
    if (*px < p->xmin) {
        *px = p->xmin;
        *vx = fabsf(*vx) * p->damping;
    }
    if (*px > p->xmax) {
        *px = p->xmax;
        *vx = -fabsf(*vx) * p->damping;
    }
    if (*py < p->ymin) {
        *py = p->ymin;
        *vy = fabsf(*vy) * p->damping;
    }
    if (*py > p->ymax) {
        *py = p->ymax;
        *vy = -fabsf(*vy) * p->damping;
    }
}

/* ---- all-pairs step, reference order (j = 0..N-1) -------------------------------------- */

/* serial_computeDensityPressure, SPH/sph.cpp:41-55 (no `valid` test, self included) */
void orc_density_allpairs(const SphParams* p, orc_particle* a, int n) {
    orc_consts c = orc_make_consts(p, 0);
    for (int i = 0; i < n; i++) {
        float dens = 0.0f;
        for (int j = 0; j < n; j++) {
            float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
            float r2 = rx * rx + ry * ry;
            if (r2 < c.h2) dens += orc_density_term(&c, p, r2);
        }
        a[i].density = dens;
        a[i].pressure = p->stiffness * (dens - p->rest_density);
        if (a[i].pressure < 0.0f) a[i].pressure = 0.0f;
    }
}

/* #{j : r2 < H*H}, self included: the reference predicate of SPH/sph.cpp:46-48 */
void orc_neighbor_counts_allpairs(const SphParams* p, const orc_particle* a, int n, int* counts) {
    float h2 = p->h * p->h;
    for (int i = 0; i < n; i++) {
        int cnt = 0;
        for (int j = 0; j < n; j++) {
            float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
            float r2 = rx * rx + ry * ry;
            if (r2 < h2) cnt++;
        }
        counts[i] = cnt;
    }
}

/* serial_computeForces, SPH/sph.cpp:57-99 (ORC_GAUSS_SEIDEL) or the force part of
 * mpi_computeStep, SPH/MPI_sph.cpp:169-214 (ORC_SNAPSHOT) */
void orc_forces_allpairs(const SphParams* p, orc_particle* a, int n, float mx, float my, float strength,
                         int flags) {
    orc_consts c = orc_make_consts(p, flags);
    float* svx = NULL;
    float* svy = NULL;
    if (flags & ORC_SNAPSHOT) {
        svx = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        svy = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < n; i++) {
            svx[i] = a[i].vx;
            svy[i] = a[i].vy;
        }
    }
    for (int i = 0; i < n; i++) {
        if (!a[i].valid) continue;
        float fx = 0.0f, fy = 0.0f;
        for (int j = 0; j < n; j++) {
            if (j == i || !a[j].valid) continue;
            float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
            float r2 = rx * rx + ry * ry;
            if (r2 >= c.h2) continue;
            float vjx = svx ? svx[j] : a[j].vx;
            float vjy = svy ? svy[j] : a[j].vy;
            orc_force_term(&c, p, rx, ry, r2, a[i].pressure, a[j].pressure, a[j].density, a[i].vx, a[i].vy, vjx,
                           vjy, &fx, &fy);
        }
        orc_force_epilogue(p, fx, fy, a[i].px, a[i].py, a[i].density, mx, my, strength, &a[i].vx, &a[i].vy);
    }
    free(svx);
    free(svy);
}

/* serial_integrate, SPH/sph.cpp:101-130 */
void orc_integrate(const SphParams* p, orc_particle* a, int n) {
    for (int i = 0; i < n; i++) {
        if (!a[i].valid) continue;
        orc_integrate_one(p, &a[i].px, &a[i].py, &a[i].vx, &a[i].vy);
    }
}

/* serial_stepSimulation, SPH/sph.cpp:166-171, nsteps times */
void orc_step_allpairs(const SphParams* p, orc_particle* a, int n, float mx, float my, float strength, int flags,
                       int nsteps) {
    for (int s = 0; s < nsteps; s++) {
        orc_density_allpairs(p, a, n);
        orc_forces_allpairs(p, a, n, mx, my, strength, flags);
        orc_integrate(p, a, n);
    }
}

/* serial_initSimulation, SPH/sph.cpp:132-164, with the square box [min_pos,max_pos]^2 as a
 * parameter (0.05f / 1.95f in the reference).  sqrt_f32 != 0 uses the MPI copy's
 * ceil(sqrtf((float)N)) (MPI_sph.cpp:97) instead of ceil(sqrt((double)N)). */
void orc_init_lattice(orc_particle* a, int n, float min_pos, float max_pos, float rest_density, int sqrt_f32) {
    int cols = sqrt_f32 ? (int)ceilf(sqrtf((float)n)) : (int)ceil(sqrt((double)n));
    int rows = (cols > 0) ? (n + cols - 1) / cols : 0;
    if (rows == 0) rows = 1;
    const float range = max_pos - min_pos;
    float spacingX = range / (float)(cols > 1 ? cols - 1 : 1);
    float spacingY = range / (float)(rows > 1 ? rows - 1 : 1);
    float spacing = (spacingY < spacingX) ? spacingY : spacingX; /* std::min(spacingX, spacingY) */
    float startX = min_pos + (range - (float)(cols - 1) * spacing) / 2.0f;
    float startY = max_pos - (range - (float)(rows - 1) * spacing) / 2.0f;
    for (int i = 0; i < n; i++) {
        int x = i % cols;
        int y = i / cols;
        memset(&a[i], 0, sizeof(orc_particle));
        a[i].px = startX + (float)x * spacing;
        a[i].py = startY - (float)y * spacing;
        a[i].vx = 0.0f;
        a[i].vy = 0.0f;
        a[i].density = rest_density;
        a[i].pressure = 0.0f;
        a[i].valid = 1;
    }
}

/* ---- uniform grid (no reference code: the reference only names `grid, cellStart, cellEnd`,
 *      SPH/sph.cuh:33-34,38).  Spec, shared with the HIP path:
 *        cell   = cell_size ? cell_size : h * (1 + 1/256)
 *        nx     = floor((xmax - xmin) / cell) + 1, ny likewise
 *        cx     = clamp((int)floorf((x - xmin) * (1.0f / cell)), 0, nx - 1), cy likewise
 *        key    = cx * ny + cy            (column-major: an x-slab is a contiguous key range)
 *        order  = ascending (key, id)
 *        cell_start[k] = #{particles with key < k}, k = 0..nx*ny  (cell_end[k] = cell_start[k+1])
 */
void orc_grid_dims(const SphParams* p, float* cell, float* inv_cell, int* nx, int* ny) {
    float c = (p->cell_size > 0.0f) ? p->cell_size : p->h * 1.00390625f;
    float inv = 1.0f / c;
    *cell = c;
    *inv_cell = inv;
    *nx = (int)floorf((p->xmax - p->xmin) * inv) + 1;
    *ny = (int)floorf((p->ymax - p->ymin) * inv) + 1;
    if (*nx < 1) *nx = 1;
    if (*ny < 1) *ny = 1;
}

static inline int orc_cell_coord(float x, float x0, float inv_cell, int n) {
    float f = floorf((x - x0) * inv_cell);
    int c;
    if (!(f >= 0.0f)) /* negative or NaN */
        c = 0;
    else if (f >= (float)n)
        c = n - 1;
    else
        c = (int)f;
    return c;
}

typedef struct {
    int key;
    int id;
} orc_keyid;

static int orc_keyid_cmp(const void* a, const void* b) {
    const orc_keyid* x = (const orc_keyid*)a;
    const orc_keyid* y = (const orc_keyid*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

/* keys[i]: key of particle i (original order); order[s]: particle index at sorted slot s;
 * sorted_keys[s] (may be NULL); cell_start[nx*ny + 1]. */
void orc_build_cells(const SphParams* p, const orc_particle* a, int n, int* keys, int* order, int* sorted_keys,
                     int* cell_start) {
    float cell, inv;
    int nx, ny;
    orc_grid_dims(p, &cell, &inv, &nx, &ny);
    orc_keyid* ki = (orc_keyid*)malloc(sizeof(orc_keyid) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int cx = orc_cell_coord(a[i].px, p->xmin, inv, nx);
        int cy = orc_cell_coord(a[i].py, p->ymin, inv, ny);
        ki[i].key = cx * ny + cy;
        ki[i].id = i;
        if (keys) keys[i] = ki[i].key;
    }
    qsort(ki, (size_t)n, sizeof(orc_keyid), orc_keyid_cmp);
    int ncell = nx * ny;
    int s = 0;
    for (int k = 0; k <= ncell; k++) {
        while (s < n && ki[s].key < k) s++;
        cell_start[k] = s;
    }
    for (int i = 0; i < n; i++) {
        order[i] = ki[i].id;
        if (sorted_keys) sorted_keys[i] = ki[i].key;
    }
    free(ki);
}

/* Canonical neighbour order of the HIP path, for the particle in cell (cx,cy): cell rows cy-1, cy,
 * cy+1 (clamped to the grid); inside a row the cells of columns cx-1, cx, cx+1 (skipping columns
 * outside the grid); inside a cell ascending slot.  `lo/hi` receive up to nine half-open ranges. */
static inline int orc_stencil_ranges(const int* cell_start, int nx, int ny, int cx, int cy, int lo[9], int hi[9]) {
    int nr = 0;
    int y0 = cy > 0 ? cy - 1 : 0;
    int y1 = cy < ny - 1 ? cy + 1 : ny - 1;
    for (int y = y0; y <= y1; y++)
        for (int dc = -1; dc <= 1; dc++) {
            int c = cx + dc;
            if (c < 0 || c >= nx) continue;
            lo[nr] = cell_start[c * ny + y];
            hi[nr] = cell_start[c * ny + y + 1];
            nr++;
        }
    return nr;
}

/* Density / pressure with the cell list; same per-term arithmetic as orc_density_allpairs, the
 * sum runs in the canonical neighbour order.  counts (may be NULL) receives #{j: r2 < H*H}.
 * Results are stored in original particle order. */
void orc_density_cells(const SphParams* p, orc_particle* a, int n, int* counts) {
    orc_consts c = orc_make_consts(p, 0);
    float cell, inv;
    int nx, ny;
    orc_grid_dims(p, &cell, &inv, &nx, &ny);
    int* keys = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* cs = (int*)malloc(sizeof(int) * (size_t)(nx * ny + 1));
    orc_build_cells(p, a, n, keys, order, NULL, cs);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        int cx = keys[i] / ny, cy = keys[i] % ny;
        int lo[9], hi[9];
        int nr = orc_stencil_ranges(cs, nx, ny, cx, cy, lo, hi);
        float dens = 0.0f;
        int cnt = 0;
        for (int r = 0; r < nr; r++)
            for (int s = lo[r]; s < hi[r]; s++) {
                int j = order[s];
                float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
                float r2 = rx * rx + ry * ry;
                if (r2 < c.h2) {
                    dens += orc_density_term(&c, p, r2);
                    cnt++;
                }
            }
        a[i].density = dens;
        float pr = p->stiffness * (dens - p->rest_density);
        if (pr < 0.0f) pr = 0.0f;
        a[i].pressure = pr;
        if (counts) counts[i] = cnt;
    }
    free(keys);
    free(order);
    free(cs);
}

/* Forces + velocity update with the cell list, snapshot semantics (the only semantics a
 * parallel sweep can have), canonical neighbour order. */
void orc_forces_cells(const SphParams* p, orc_particle* a, int n, float mx, float my, float strength, int flags) {
    orc_consts c = orc_make_consts(p, flags);
    float cell, inv;
    int nx, ny;
    orc_grid_dims(p, &cell, &inv, &nx, &ny);
    int* keys = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* cs = (int*)malloc(sizeof(int) * (size_t)(nx * ny + 1));
    float* nvx = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* nvy = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    orc_build_cells(p, a, n, keys, order, NULL, cs);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        nvx[i] = a[i].vx;
        nvy[i] = a[i].vy;
        if (!a[i].valid) continue;
        int cx = keys[i] / ny, cy = keys[i] % ny;
        int lo[9], hi[9];
        int nr = orc_stencil_ranges(cs, nx, ny, cx, cy, lo, hi);
        float fx = 0.0f, fy = 0.0f;
        for (int r = 0; r < nr; r++)
            for (int s = lo[r]; s < hi[r]; s++) {
                int j = order[s];
                if (j == i || !a[j].valid) continue;
                float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
                float r2 = rx * rx + ry * ry;
                if (r2 >= c.h2) continue;
                orc_force_term(&c, p, rx, ry, r2, a[i].pressure, a[j].pressure, a[j].density, a[i].vx, a[i].vy,
                               a[j].vx, a[j].vy, &fx, &fy);
            }
        orc_force_epilogue(p, fx, fy, a[i].px, a[i].py, a[i].density, mx, my, strength, &nvx[i], &nvy[i]);
    }
    for (int i = 0; i < n; i++) {
        a[i].vx = nvx[i];
        a[i].vy = nvy[i];
    }
    free(keys);
    free(order);
    free(cs);
    free(nvx);
    free(nvy);
}

void orc_step_cells(const SphParams* p, orc_particle* a, int n, float mx, float my, float strength, int flags,
                    int nsteps) {
    for (int s = 0; s < nsteps; s++) {
        orc_density_cells(p, a, n, NULL);
        orc_forces_cells(p, a, n, mx, my, strength, flags);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) {
            if (!a[i].valid) continue;
            orc_integrate_one(p, &a[i].px, &a[i].py, &a[i].vx, &a[i].vy);
        }
    }
}

/* ---- CPU baseline: portable restatement of the OpenMP variant --------------------------
 * SPH/openMP_sph.cpp:80-205 (computeDensityPressureOMP, computeForcesOMP, integrateOMP): the
 * serial all-pairs loops under `#pragma omp parallel for` (default static schedule), velocity
 * written in place while other threads read it (:114 vs :144-145; benign for timing, not
 * deterministic).  Its spikyGrad squares with powf(h - r_len, 2) (:58), i.e. coeff =
 * C * (t*t) / r instead of the serial C * t * t / r, and guards with 1e-6f (:57).  The original
 * file is Windows-only (windows.h, DirectXMath.h) and cannot be compiled on Linux. */
void orc_omp_step_allpairs(const SphParams* p, orc_particle* a, int n, int nsteps) {
    orc_consts c = orc_make_consts(p, ORC_EPS_F32);
    for (int step = 0; step < nsteps; step++) {
#pragma omp parallel for
        for (int i = 0; i < n; i++) {
            float dens = 0.0f;
            for (int j = 0; j < n; j++) {
                float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
                float r2 = rx * rx + ry * ry;
                if (r2 < c.h2) dens += orc_density_term(&c, p, r2);
            }
            a[i].density = dens;
            float pr = p->stiffness * (dens - p->rest_density);
            if (pr < 0.0f) pr = 0.0f;
            a[i].pressure = pr;
        }
#pragma omp parallel for
        for (int i = 0; i < n; i++) {
            if (!a[i].valid) continue;
            float fx = 0.0f, fy = 0.0f;
            for (int j = 0; j < n; j++) {
                if (i == j || !a[j].valid) continue;
                float rx = a[i].px - a[j].px, ry = a[i].py - a[j].py;
                float r2 = rx * rx + ry * ry;
                if (r2 >= c.h2) continue;
                float r_len = sqrtf(r2);
                float gx = 0.0f, gy = 0.0f;
                if (!(r_len > c.h || r_len < 1e-6f)) {
                    float t = c.h - r_len;
                    float coeff = c.spikyc * (t * t) / r_len; /* powf(h - r_len, 2) == t*t */
                    gx = coeff * rx;
                    gy = coeff * ry;
                }
                float pressureTerm = (a[i].pressure + a[j].pressure) / (2.0f * a[j].density);
                float pfx = c.neg_mass * pressureTerm * gx, pfy = c.neg_mass * pressureTerm * gy;
                float relx = a[j].vx - a[i].vx, rely = a[j].vy - a[i].vy;
                float lap = (r_len > c.h) ? 0.0f : c.viscc * (c.h - r_len);
                float viscForce = c.visc_mass * lap / a[j].density;
                float ljx = 0.0f, ljy = 0.0f;
                if (!(r_len >= c.lj_cut || r_len < 1e-6f)) {
                    float inv_r = 1.0f / r_len;
                    float s = p->sigma * inv_r;
                    float s2 = s * s;
                    float s6 = s2 * s2 * s2;
                    float s12 = s6 * s6;
                    float inv_r2 = inv_r * inv_r;
                    float coeff = c.lj24 * (2.0f * s12 - s6) * inv_r2;
                    ljx = coeff * rx;
                    ljy = coeff * ry;
                }
                fx += pfx + viscForce * relx + ljx;
                fy += pfy + viscForce * rely + ljy;
            }
            /* openMP_sph.cpp:121-145; mouse strength is 0 in its performance test (:44-45) */
            orc_force_epilogue(p, fx, fy, a[i].px, a[i].py, a[i].density, 0.0f, 0.0f, 0.0f, &a[i].vx, &a[i].vy);
        }
#pragma omp parallel for
        for (int i = 0; i < n; i++) {
            if (!a[i].valid) continue;
            orc_integrate_one(p, &a[i].px, &a[i].py, &a[i].vx, &a[i].vy);
        }
    }
}

/* FNV-1a-64 over the six f32 words of each particle (32-bit word-wise xor-then-multiply), the
 * checksum SURVEY.md section 8(c) recorded its provisional pins with. */
unsigned long long orc_checksum(const orc_particle* a, int n) {
    unsigned long long hsh = 0xcbf29ce484222325ULL;
    for (int i = 0; i < n; i++) {
        unsigned int w[6];
        memcpy(w, &a[i], 24);
        for (int k = 0; k < 6; k++) {
            hsh ^= (unsigned long long)w[k];
            hsh *= 0x100000001b3ULL;
        }
    }
    return hsh;
}

int orc_num_threads(void) {
    int nt = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp master
        nt = omp_get_num_threads();
    }
#endif
    return nt;
}
