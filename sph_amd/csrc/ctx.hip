// ctx.hip -- context, memory, state transfer, slab hooks and the extern "C" surface of libsphhip.so.
// Entry points and the reference interfaces they replace are documented in include/sphhip.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include "sphhip_internal.h"

// ---- error plumbing -------------------------------------------------------------------------
static thread_local std::string g_last_error;

void sph_set_error(const std::string& msg) { g_last_error = msg; }

int sph_hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    return SPHHIP_EHIP;
}

static int fail(int code, const char* msg) {
    g_last_error = msg;
    return code;
}

// ---- phase timing ---------------------------------------------------------------------------
static inline bool phase_timed(const SphCtx* c, int ph) {
    return c->profiling && ((c->profile_mask >> ph) & 1u) && (c->profile_period <= 1 || c->profile_step == 0 || (c->profile_step - 1) % c->profile_period == 0);  // steps 1, 1 + period, ...
}

void sph_phase_begin(SphCtx* c, int ph, hipStream_t on) {
    if (!phase_timed(c, ph)) return;
    PhaseTimer& t = c->timer;
    if (t.used[ph] == t.start[ph].size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        t.start[ph].push_back(a);
        t.stop[ph].push_back(b);
    }
    (void)hipEventRecord(t.start[ph][t.used[ph]], on);  // timing only: a failure shows up as a missing sample
}

void sph_phase_begin(SphCtx* c, int ph) { sph_phase_begin(c, ph, c->stream); }
void sph_phase_end(SphCtx* c, int ph) { sph_phase_end(c, ph, c->stream); }

void sph_phase_end(SphCtx* c, int ph, hipStream_t on) {
    if (!phase_timed(c, ph)) return;
    PhaseTimer& t = c->timer;
    if (t.used[ph] >= t.stop[ph].size()) return;
    (void)hipEventRecord(t.stop[ph][t.used[ph]], on);
    t.used[ph]++;
}

static void timer_flush(SphCtx* c) {
    PhaseTimer& t = c->timer;
    for (int ph = 0; ph < SPHHIP_PH_COUNT; ph++) {
        for (size_t i = 0; i < t.used[ph]; i++) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, t.start[ph][i], t.stop[ph][i]) == hipSuccess) {
                t.ms[ph] += ms;
                t.launches[ph]++;
            }
        }
        t.used[ph] = 0;
    }
}

// ---- parameters -----------------------------------------------------------------------------
extern "C" int sphhip_default_params(SphParams* p) {
    if (!p) return fail(SPHHIP_EINVAL, "sphhip_default_params: null");
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
    return SPHHIP_OK;
}

extern "C" int sphhip_derive(const SphParams* p, SphDerived* d) {
    if (!p || !d) return fail(SPHHIP_EINVAL, "sphhip_derive: null");
    if (!(p->h > 0.0f) || !(p->xmax > p->xmin) || !(p->ymax > p->ymin))
        return fail(SPHHIP_EINVAL, "sphhip_derive: need h > 0 and max > min walls");
    // expression order of SPH/sph.cpp:11,18,25,30,37,70,72,81
    d->h2 = p->h * p->h;
    d->poly6 = 315.0f / (64.0f * 3.14159f * powf(p->h, 9));
    d->spiky = -45.0f / (3.14159f * powf(p->h, 6));
    d->visc_lap = 45.0f / (3.14159f * powf(p->h, 6));
    d->visc_mass = p->viscosity * p->mass;
    d->lj_cut = 2.5f * p->sigma;
    d->lj_24eps = 24.0f * p->epsilon;
    d->neg_mass = -p->mass;
    d->mouse_r2 = p->mouse_radius * p->mouse_radius;
    d->cell = (p->cell_size > 0.0f) ? p->cell_size : p->h * 1.00390625f;
    if (!(d->cell > p->h)) return fail(SPHHIP_EINVAL, "sphhip_derive: cell_size must exceed h");
    d->inv_cell = 1.0f / d->cell;
    double fx = std::floor((double)((p->xmax - p->xmin) * d->inv_cell)) + 1.0;
    double fy = std::floor((double)((p->ymax - p->ymin) * d->inv_cell)) + 1.0;
    if (fx * fy > 1.0e9) return fail(SPHHIP_EINVAL, "sphhip_derive: grid too large (> 1e9 cells)");
    d->nx = (int)floorf((p->xmax - p->xmin) * d->inv_cell) + 1;
    d->ny = (int)floorf((p->ymax - p->ymin) * d->inv_cell) + 1;
    if (d->nx < 1) d->nx = 1;
    if (d->ny < 1) d->ny = 1;
    return SPHHIP_OK;
}

static void fill_dev_consts(const SphParams& p, const SphDerived& d, DevConsts& c) {
    c.h = p.h;
    c.h2 = d.h2;
    c.poly6 = d.poly6;
    c.spiky = d.spiky;
    c.visc_lap = d.visc_lap;
    c.visc_mass = d.visc_mass;
    c.neg_mass = d.neg_mass;
    c.lj_cut = d.lj_cut;
    c.lj24 = d.lj_24eps;
    c.sigma = p.sigma;
    c.mass = p.mass;
    c.stiffness = p.stiffness;
    c.rest_density = p.rest_density;
    c.gravity = p.gravity;
    c.force_cap = p.force_cap;
    c.mouse_radius = p.mouse_radius;
    c.mouse_r2 = d.mouse_r2;
    c.dt = p.dt;
    c.max_vel = p.max_vel;
    c.damping = p.damping;
    c.xmin = p.xmin;
    c.xmax = p.xmax;
    c.ymin = p.ymin;
    c.ymax = p.ymax;
    c.inv_cell = d.inv_cell;
    c.nx = d.nx;
    c.ny = d.ny;
}

// ---- context --------------------------------------------------------------------------------
template <typename T>
static hipError_t dmalloc(T** p, size_t count) {
    return hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1));
}

static void free_all(SphCtx* c) {
    void* ptrs[] = {c->pos, c->vel, c->pos2, c->vel2, c->rho, c->prs, c->rho2, c->prs2, c->idv, c->idv2,
                    c->ncount, c->nmask, c->keys, c->keys2, c->perm, c->perm2, c->ids_sorted, c->cell_buf[0], c->cell_buf[1], c->block_hist,
                    c->digit_total, c->scan_tmp, c->counters, c->tile_list, c->staging, c->dense_list, c->big_list, c->nmask_hi, c->d_sinks, c->d_consts};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int ph = 0; ph < SPHHIP_PH_COUNT; ph++) {
        for (hipEvent_t e : c->timer.start[ph]) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->timer.stop[ph]) (void)hipEventDestroy(e);
    }
    if (c->h_host_tiles) (void)hipHostFree(c->h_host_tiles);
    if (c->h_dense) (void)hipHostFree(c->h_dense);
    if (c->step_graph) (void)hipGraphExecDestroy(c->step_graph);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->ev_pace)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
}

// Everything of a context that is not device memory: constants, geometry, tuning switches (environment).
static void ctx_configure(SphCtx* c, const SphParams& p, const SphDerived& d, int n_max, int device) {
    c->params = p;
    c->drv = d;
    fill_dev_consts(p, d, c->dc);
    c->device = device;
    c->n_max = n_max;
    c->ncell = d.nx * d.ny;
    c->col_lo = 0;
    c->col_hi = d.nx;
    if (const char* e = getenv("SPHHIP_TILE_CELLS")) {
        int v = atoi(e);
        if (v >= 1 && v <= PAIR_MAX_TILE_CELLS) {
            c->tile_cells = c->tile_cells_max = v;
            c->tile_cells_auto = false;
        }
    }
    {   // exact_div.h needs every numerator of the pair force inside its safe range
        // inside [SPH_SAFE_LO, SPH_SAFE_HI] of exact_div.h with margin
        auto safe = [](float x) { x = std::fabs(x); return x >= 4e-12f && x <= 2.5e11f; };
        float t = p.h;  // term = h - r_len <= h
        // (p.h >= 2^-5: k_force's clamp of tiny pair distances relies on h - 2^-30 == h, see force_terms)
        c->fast_div = p.h >= 0.03125f && safe(d.spiky * t * t) && safe(d.visc_mass * d.visc_lap * t) && safe(p.h) && safe(p.stiffness) &&
                      safe(p.rest_density) && safe(d.spiky * 1e-8f * 1e-8f) && safe(d.visc_mass * d.visc_lap * 1e-8f);
        if (const char* e = getenv("SPHHIP_FAST_DIV")) c->fast_div = c->fast_div && atoi(e) != 0;
    }
    if (const char* e = getenv("SPHHIP_USE_MASKS")) c->use_masks = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_ZERO_P")) c->zero_p = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_GRAPH")) c->graph_off = atoi(e) == 0;
    if (const char* e = getenv("SPHHIP_SMALL_GRID")) c->small_grid = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_RANK_BIG")) c->rank_big = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_TILE_LIST_FUSED")) c->tile_list_fused = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_SPLIT_CONCURRENT")) c->split_concurrent = atoi(e) != 0;
    if (c->rank_big && sph_rank_big_setup() != SPHHIP_OK) c->rank_big = false;  // (no 128 KB LDS opt-in: the loop ranks everything)
    if (const char* e = getenv("SPHHIP_DENSE_TP")) c->dense_tp = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_LAZY_PACK")) c->lazy_pack = atoi(e) != 0;
    if (const char* e = getenv("SPHHIP_GRID")) c->grid_counting = std::string(e) != "radix";
    if (const char* e = getenv("SPHHIP_PAIR_THREADS")) {
        int v = atoi(e);
        if (v == 128 || v == 256) c->pair_threads = v;
    }
    if (const char* e = getenv("SPHHIP_PAIR_CAP")) {
        int v = atoi(e);
        if (v >= 64 && v <= 4608) c->pair_cap = v;
    }
}

// The device buffers of a context, one call of f(slot, bytes, zero) each: sphhip_create gives every buffer a
// hipMalloc of its own, sphhip_step_static carves them out of the library's static scratch array.
// zero: the buffer must start as zeros (index arrays start defined: a stale entry must never be an out-of-range
// index; counters and cell tables start at 0).
template <class F>
static void ctx_buffers(SphCtx* c, bool with_staging, F&& f) {
    const size_t n = (size_t)c->n_max;
    const int nblocks = (c->n_max + SORT_TILE - 1) / SORT_TILE;
    for (float2** p : {&c->pos, &c->vel, &c->pos2, &c->vel2}) f((void**)p, sizeof(float2) * n, false);
    for (float** p : {&c->rho, &c->prs, &c->rho2, &c->prs2}) f((void**)p, sizeof(float) * n, false);
    f((void**)&c->ncount, sizeof(int) * n, false);
    f((void**)&c->nmask, sizeof(uint32_t) * n * 8, false);
    for (uint32_t** p : {&c->keys, &c->keys2, &c->perm, &c->perm2, &c->ids_sorted, &c->idv, &c->idv2})
        f((void**)p, sizeof(uint32_t) * n, true);
    // two cell tables, each followed by the partial sums of its scan (see grid.hip)
    const size_t words = (size_t)c->ncell + 1 + ((size_t)c->ncell + 1 + 2047) / 2048 + 8;
    for (int b = 0; b < 2; b++) f((void**)&c->cell_buf[b], sizeof(int) * words, true);
    f((void**)&c->block_hist, sizeof(int) * (size_t)SORT_RADIX * nblocks, false);
    f((void**)&c->digit_total, sizeof(int) * 2 * SORT_RADIX, false);
    f((void**)&c->scan_tmp, sizeof(int) * ((size_t)(c->ncell + 1 + 2047) / 2048 + 1), false);
    f((void**)&c->counters, sizeof(int) * 64, true);
    // every entry of the work list holds at least one particle and at least one cell row
    c->tile_list_cap = (int)std::min<long long>((long long)c->ncell, (long long)c->n_max) + 64;
    f((void**)&c->tile_list, sizeof(TileDesc) * (size_t)c->tile_list_cap, true);
    // a dense sub-tile holds more candidates than the smallest LDS budget (64 records, 9 of them sentinels) and
    // a particle is a candidate of at most 9 one-row sub-tiles; one entry per sub-tile and DENSE_THREADS own
    // particles: 9 n / 55 + n / 256 < n / 4 entries cover it
    c->dense_cap = c->n_max / 4 + 64;
    if (const char* e = getenv("SPHHIP_DENSE_CAP")) c->dense_cap = std::max(1, atoi(e));  // (tests: force an overflow)
    f((void**)&c->dense_list, sizeof(int4) * (size_t)c->dense_cap, false);
    f((void**)&c->d_sinks, sizeof(DenseSinkDev) * 2, true);
    f((void**)&c->d_consts, sizeof(ForceExtraDev), true);
    c->big_cap = c->n_max / RANK_BIG_MIN + 1;  // (no more cells than that can be above RANK_BIG_MIN)
    f((void**)&c->big_list, sizeof(int) * (size_t)c->big_cap, true);
    // mask words 8..13 of long stencil ranges (allocated last: the layout of everything above is as it was without them)
    f((void**)&c->nmask_hi, sizeof(uint32_t) * n * (PAIR_MASK_WORDS_MAX - 8), false);
    if (with_staging) f(&c->staging, std::max<size_t>(n * SPHHIP_PARTICLE_BYTES, 64), false);
}

// The constant halves of the two dense-sub-tile sinks (pairs.hip: DenseSink) go to device memory once per context;
// by kernel, with the values as kernel arguments: no host buffer has to outlive an asynchronous copy, nothing allocates.
__global__ void k_write_sinks(DenseSinkDev* out, DenseSinkDev density, DenseSinkDev force, ForceExtraDev* consts_out, ForceExtraDev consts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = density;
        out[1] = force;
        *consts_out = consts;
    }
}

static int ctx_write_sinks(SphCtx* c) {
    DenseSinkDev s[2];
    for (int which = 0; which < 2; which++) {
        s[which].list = c->dense_list;
        s[which].count = c->counters + SPH_DENSE_COUNT_SLOT;
        s[which].capacity = c->dense_cap;
        s[which].host_flag = c->d_dense + which;
        s[which].err = c->d_err;
    }
    ForceExtraDev x;
    x.c = c->dc;
    x.nmask_hi = c->nmask_hi;
    hipLaunchKernelGGL(k_write_sinks, dim3(1), dim3(64), 0, c->stream, c->d_sinks, s[0], s[1], c->d_consts, x);
    LAUNCH_CHECK();
    return SPHHIP_OK;
}

// Pointers into those buffers, and the state of a context that holds no particles yet.
static void ctx_wire(SphCtx* c) {
    c->cell_start = c->cell_buf[0];
    c->cell_cur = 0;
    c->cell_next_zeroed = true;
    c->d_err = c->counters + 16;
    c->n_ring = c->counters + 32;
    c->n_slot = 0;
    c->d_n = c->n_ring;
}

extern "C" int sphhip_create(const SphParams* p, int n_max, int device, SphCtx** out) {
    if (!p || !out) return fail(SPHHIP_EINVAL, "sphhip_create: null argument");
    if (n_max < 1 || n_max > (1 << 30)) return fail(SPHHIP_EINVAL, "sphhip_create: n_max out of range");
    SphDerived d;
    int rc = sphhip_derive(p, &d);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(SPHHIP_EINVAL, "sphhip_create: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    SphCtx* c = new SphCtx();
    ctx_configure(c, *p, d, n_max, device);
    const size_t n = (size_t)n_max;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    ok(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    {
        // Stream invariant ("Streams and progress"): a stream that another stream waits for never has
        // a lower priority than the waiter.  The side stream and the context's stream wait for each other (ev_fork,
        // ev_join), so they have the SAME priority -- here the default, and sphhip_set_stream re-creates the side stream
        // at the priority of a caller's stream.  (Round 2 tried the lowest priority for the side stream: four slabs on
        // one GPU stopped making progress, the higher-priority queues parked on barrier packets that only the starved
        // stream's kernel could release.)
        ok(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        ok(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&c->ev_pace[0], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&c->ev_pace[1], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    ctx_buffers(c, true, [&](void** slot, size_t bytes, bool zero) {
        ok(hipMalloc(slot, std::max<size_t>(bytes, 4)));
        if (e == hipSuccess && zero) ok(hipMemset(*slot, 0, std::max<size_t>(bytes, 4)));
    });
    ctx_wire(c);
    ok(hipHostMalloc((void**)&c->h_host_tiles, 3 * sizeof(int), hipHostMallocMapped));
    if (e == hipSuccess) {
        c->h_host_tiles[0] = -1;
        c->h_host_tiles[1] = 0;
        c->h_host_tiles[2] = -1;
        ok(hipHostGetDevicePointer((void**)&c->d_host_tiles, c->h_host_tiles, 0));
    }
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->num_cus = v; }
    ok(hipHostMalloc((void**)&c->h_dense, 2 * sizeof(int), hipHostMallocMapped));
    if (e == hipSuccess) {
        c->h_dense[0] = c->h_dense[1] = -1;
        ok(hipHostGetDevicePointer((void**)&c->d_dense, c->h_dense, 0));
    }
    if (e == hipSuccess && getenv("SPHHIP_POISON")) {
        // debugging aid: data buffers start as 0xFF bytes (NaN floats, huge integers), so that a read of something
        // never written shows up in the results instead of depending on what the allocator recycled.  The value
        // is a bit mask: 1 pos/vel, 2 pos2/vel2, 4 rho/prs, 8 rho2/prs2, 16 ncount, 32 nmask, 64 staging.
        const int which = atoi(getenv("SPHHIP_POISON"));
        auto poison = [&](void* p, size_t bytes) { ok(hipMemset(p, 0xFF, bytes)); };
        if (which & 1) for (float2* p : {c->pos, c->vel}) poison(p, sizeof(float2) * n);
        if (which & 2) for (float2* p : {c->pos2, c->vel2}) poison(p, sizeof(float2) * n);
        if (which & 4) for (float* p : {c->rho, c->prs}) poison(p, sizeof(float) * n);
        if (which & 8) for (float* p : {c->rho2, c->prs2}) poison(p, sizeof(float) * n);
        if (which & 16) poison(c->ncount, sizeof(int) * n);
        if (which & 32) poison(c->nmask, sizeof(uint32_t) * n * 8);
        if (which & 64) poison(c->staging, std::max<size_t>(n * SPHHIP_PARTICLE_BYTES, 64));
    }
    // the memsets above ran on the null stream, the context's own stream does not wait for it: finish them
    // before anything (an upload) can be enqueued there
    if (e == hipSuccess) ok(hipDeviceSynchronize());
    if (e != hipSuccess) {
        free_all(c);
        delete c;
        return sph_hip_fail(e, "sphhip_create allocation", __FILE__, __LINE__);
    }
    if (int rc = ctx_write_sinks(c)) {
        free_all(c);
        delete c;
        return rc;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        free_all(c);
        delete c;
        return fail(SPHHIP_EHIP, "sphhip_create: hipStreamSynchronize failed");
    }
    *out = c;
    return SPHHIP_OK;
}

extern "C" int sphhip_destroy(SphCtx* c) {
    if (!c) return SPHHIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_all(c);
    delete c;
    return SPHHIP_OK;
}

// Keeps the stream invariant of sphhip_create: the side stream at the priority of the stream it forks from / joins.
static int side_stream_follow(SphCtx* c) {
    if (!c->side_stream) return SPHHIP_OK;
    int want = 0, have = 0;
    HIP_TRY(hipStreamGetPriority(c->stream, &want));  // (the null stream reports the default priority)
    HIP_TRY(hipStreamGetPriority(c->side_stream, &have));
    if (want == have) return SPHHIP_OK;
    HIP_TRY(hipStreamSynchronize(c->side_stream));
    HIP_TRY(hipStreamDestroy(c->side_stream));
    c->side_stream = nullptr;
    HIP_TRY(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, want));
    return SPHHIP_OK;
}

extern "C" int sphhip_set_stream(SphCtx* c, void* s) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return side_stream_follow(c);
}

extern "C" int sphhip_use_default_stream(SphCtx* c) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = nullptr;  // the legacy default (null) stream
    return side_stream_follow(c);
}

extern "C" int sphhip_get_stream(SphCtx* c, void** s) {
    if (!c || !s) return fail(SPHHIP_EINVAL, "null argument");
    *s = (void*)c->stream;
    return SPHHIP_OK;
}

extern "C" int sphhip_synchronize(SphCtx* c) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPHHIP_OK;
}

extern "C" const char* sphhip_last_error(void) { return g_last_error.c_str(); }

extern "C" int sphhip_set_tuning(SphCtx* c, int tile_cells, int pair_cap) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    if (tile_cells > 0) {
        if (tile_cells > PAIR_MAX_TILE_CELLS) return fail(SPHHIP_EINVAL, "tile_cells too large");
        c->tile_cells = c->tile_cells_max = tile_cells;
        c->tile_cells_auto = false;
    }
    if (3 * (c->tile_cells + 2) > c->pair_threads) c->pair_threads = 256;  // one thread per candidate cell
    if (pair_cap > 0) {
        if (pair_cap < 64 || pair_cap > 4608) return fail(SPHHIP_EINVAL, "pair_cap out of range [64, 4608]");
        c->pair_cap = pair_cap;
    }
    return SPHHIP_OK;
}

// ---- AoS <-> SoA ----------------------------------------------------------------------------
// 28-byte records are read/written as 7 dwords (SPH/sph.h:27-33: pos, vel, density, pressure,
// valid byte + 3 pad).  Off the timed path.
__global__ __launch_bounds__(256) void k_unpack_aos(const uint32_t* __restrict__ aos, int n, uint32_t id_base,
                                                     float2* __restrict__ pos, float2* __restrict__ vel,
                                                     float* __restrict__ rho, float* __restrict__ prs,
                                                     uint32_t* __restrict__ idv) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* r = aos + (size_t)i * 7;
    pos[i] = make_float2(__uint_as_float(r[0]), __uint_as_float(r[1]));
    vel[i] = make_float2(__uint_as_float(r[2]), __uint_as_float(r[3]));
    rho[i] = __uint_as_float(r[4]);
    prs[i] = __uint_as_float(r[5]);
    bool valid = (r[6] & 0xffu) != 0;
    idv[i] = (id_base + (uint32_t)i) | (valid ? 0u : SPH_INVALID_BIT);
}

__global__ __launch_bounds__(256) void k_pack_aos(uint32_t* __restrict__ aos, int n, const float2* __restrict__ pos,
                                                   const float2* __restrict__ vel, const float* __restrict__ rho,
                                                   const float* __restrict__ prs, const uint32_t* __restrict__ idv,
                                                   uint32_t* __restrict__ ids_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* r = aos + (size_t)i * 7;
    float2 p = pos[i], v = vel[i];
    r[0] = __float_as_uint(p.x);
    r[1] = __float_as_uint(p.y);
    r[2] = __float_as_uint(v.x);
    r[3] = __float_as_uint(v.y);
    r[4] = __float_as_uint(rho[i]);
    r[5] = __float_as_uint(prs[i]);
    uint32_t w = idv[i];
    r[6] = (w & SPH_INVALID_BIT) ? 0u : 1u;
    ids_out[i] = w & SPH_ID_MASK;
}

extern "C" int sphhip_upload(SphCtx* c, const void* aos28, int n, int id_base) {
    if (!c || (!aos28 && n > 0)) return fail(SPHHIP_EINVAL, "sphhip_upload: null argument");
    if (n < 0 || id_base < 0 || (long long)id_base + n > 0x7fffffffLL)
        return fail(SPHHIP_EINVAL, "sphhip_upload: bad n / id_base");
    if (n > c->n_max) return fail(SPHHIP_ECAPACITY, "sphhip_upload: n exceeds the context capacity");
    HIP_TRY(hipSetDevice(c->device));
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(c->staging, aos28, (size_t)n * SPHHIP_PARTICLE_BYTES, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_unpack_aos, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)c->staging,
                           n, (uint32_t)id_base, c->pos, c->vel, c->rho, c->prs, c->idv);
        LAUNCH_CHECK();
    }
    {
        int rc = sph_set_n_host(c, n);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // the host buffer is only borrowed for the call
    c->have_state = true;
    c->split.pending = false;  // (a new state: nothing of an unfinished step is left)
    c->grid_valid = false;
    c->counts_valid = false;
    c->slice_valid = false;
    c->pending_drop = false;
    c->tail_appended = false;
    return SPHHIP_OK;
}

static int download_impl(SphCtx* c, void* aos28, int* ids, int n) {
    if (!c || (!aos28 && n > 0)) return fail(SPHHIP_EINVAL, "sphhip_download: null argument");
    if (!c->have_state) return fail(SPHHIP_ESTATE, "sphhip_download: nothing uploaded");
    if (c->split.pending) return fail(SPHHIP_ESTATE, "sphhip_download: sphhip_step_begin without sphhip_step_finish");
    HIP_TRY(hipSetDevice(c->device));
    {
        int rc = sph_sync_n(c);
        if (rc) return rc;
    }
    if (n != c->n) return fail(SPHHIP_EINVAL, "sphhip_download: n differs from the particle count");
    if (n == 0) return SPHHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    // ids_sorted doubles as the id staging buffer (it is rebuilt by every grid build)
    hipLaunchKernelGGL(k_pack_aos, dim3((n + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)c->staging, n, c->pos,
                       c->vel, c->rho, c->prs, c->idv, c->ids_sorted);
    LAUNCH_CHECK();
    std::vector<unsigned char> rec((size_t)n * SPHHIP_PARTICLE_BYTES);
    std::vector<uint32_t> id((size_t)n);
    HIP_TRY(hipMemcpyAsync(rec.data(), c->staging, rec.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(id.data(), c->ids_sorted, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // un-permute to ascending id (the reference never permutes particles; index i is stable)
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return id[a] < id[b]; });
    unsigned char* out = (unsigned char*)aos28;
    for (int k = 0; k < n; k++) {
        memcpy(out + (size_t)k * SPHHIP_PARTICLE_BYTES, rec.data() + (size_t)order[k] * SPHHIP_PARTICLE_BYTES,
               SPHHIP_PARTICLE_BYTES);
        if (ids) ids[k] = (int)id[order[k]];
    }
    return SPHHIP_OK;
}

extern "C" int sphhip_download(SphCtx* c, void* aos28, int n) { return download_impl(c, aos28, nullptr, n); }
extern "C" int sphhip_download_ids(SphCtx* c, void* aos28, int* ids, int n) { return download_impl(c, aos28, ids, n); }

extern "C" int sphhip_count(SphCtx* c, int* n) {
    if (!c || !n) return fail(SPHHIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sph_sync_n(c);
    *n = c->n;
    return rc;
}

// ---- phases ---------------------------------------------------------------------------------
#define NEED_STATE(c)                                                            \
    do {                                                                         \
        if (!(c)) return fail(SPHHIP_EINVAL, "null context");                    \
        if (!(c)->have_state) return fail(SPHHIP_ESTATE, "no particles uploaded"); \
        if ((c)->split.pending) return fail(SPHHIP_ESTATE, "sphhip_step_begin without sphhip_step_finish"); \
        HIP_TRY(hipSetDevice((c)->device));                                      \
    } while (0)

// ... for the calls allowed between sphhip_step_begin and sphhip_step_finish
#define NEED_STATE_MIDSTEP(c)                                                    \
    do {                                                                         \
        if (!(c)) return fail(SPHHIP_EINVAL, "null context");                    \
        if (!(c)->have_state) return fail(SPHHIP_ESTATE, "no particles uploaded"); \
        HIP_TRY(hipSetDevice((c)->device));                                      \
    } while (0)

// for calls that need the particle count on the host
#define NEED_EXACT(c)                    \
    do {                                 \
        NEED_STATE(c);                   \
        int _rc = sph_sync_n(c);         \
        if (_rc) return _rc;             \
    } while (0)

extern "C" int sphhip_build_grid(SphCtx* c) {
    NEED_STATE(c);
    return sph_grid_build(c, true);
}

extern "C" int sphhip_density(SphCtx* c) {
    NEED_STATE(c);
    if (!c->grid_valid) {
        int rc = sph_grid_build(c, true);
        if (rc) return rc;
    }
    return sph_density_launch(c);
}

extern "C" int sphhip_forces(SphCtx* c, float mx, float my, float strength) {
    NEED_STATE(c);
    if (!c->grid_valid) {
        int rc = sph_grid_build(c, true);
        if (rc) return rc;
    }
    return sph_force_launch(c, mx, my, strength, false);
}

extern "C" int sphhip_integrate(SphCtx* c) {
    NEED_STATE(c);
    return sph_integrate_launch(c);
}

// The host enqueues a step in ~20 us, the GPU needs ten times as long: a call of a thousand steps would be enqueued in
// full while the GPU is still at step 100 -- and everything the host decides from what the kernels report back through
// mapped memory (grid sizes, tile height, whether the dense follow-up kernels and the long-tile force kernel run) would
// be decided on news that is 900 steps old (measured: the evolved dam break ran 6 % slower inside one long call than
// in calls of ten steps).  So every THROTTLE_STEPS steps the host waits until the GPU has passed the mark set
// THROTTLE_STEPS steps earlier: it stays 32 to 64 steps ahead, never idle-making.  (Not under stream capture, and not
// for the static context, which has no events.)
constexpr int THROTTLE_STEPS = 32;
static int sph_throttle(SphCtx* c) {
    if (!c->ev_pace[0] || (++c->pace_steps % THROTTLE_STEPS) != 0) return SPHHIP_OK;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (c->stream && (hipStreamIsCapturing(c->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone)) {
        (void)hipGetLastError();
        return SPHHIP_OK;
    }
    const int cur = (int)((c->pace_steps / THROTTLE_STEPS) & 1);
    if (c->pace_armed[1 - cur]) HIP_TRY(hipEventSynchronize(c->ev_pace[1 - cur]));
    HIP_TRY(hipEventRecord(c->ev_pace[cur], c->stream));
    c->pace_armed[cur] = true;
    return SPHHIP_OK;
}

static int step_once(SphCtx* c, float mx, float my, float strength) {
    c->profile_step++;
    int rc = sph_throttle(c);
    if (rc) return rc;
    rc = sph_grid_build(c, false);
    if (rc) return rc;
    rc = sph_density_launch(c);
    if (rc) return rc;
    return sph_force_launch(c, mx, my, strength, true);
}

// ---- graph replay of small scenes -----------------------------------------------------------
// A step is 7 launches.  Below ~100 k particles they take less time on the GPU than the host needs to
// enqueue them (N = 1024, the reference's default: 38 us per step, all of it launch cost), so
// sphhip_step captures GRAPH_STEPS consecutive steps of the stream into a hipGraph once and replays it.
// GRAPH_STEPS = 8 because the host-side state a step advances (ping-pong buffers, the two cell tables,
// the ring of 8 particle-count words) is back where it started after 8 steps -- which is not assumed but
// checked: the signature of that state before and after the capture must agree, else the capture is
// dropped and the steps are enqueued one by one as before.  What a step decides from mapped host memory
// (grid sizes, tile height, whether the dense follow-up kernels run) is frozen into the graph; all of
// it only affects speed, never results, and the signature includes those decisions, so a scene that
// changes regime gets a new capture at the next call.
constexpr int GRAPH_STEPS = 8;
constexpr int GRAPH_MAX_N = 1 << 17;

static std::vector<uint64_t> step_sig(const SphCtx* c, float mx, float my, float strength) {
    auto u = [](const void* p) { return (uint64_t)(uintptr_t)p; };
    auto f = [](float x) { uint32_t b; memcpy(&b, &x, 4); return (uint64_t)b; };
    const int dd = ((volatile int*)c->h_dense)[0], df = ((volatile int*)c->h_dense)[1];
    return {u(c->pos), u(c->vel), u(c->pos2), u(c->vel2), u(c->rho), u(c->prs), u(c->rho2), u(c->prs2), u(c->idv), u(c->idv2),
            u(c->keys), u(c->keys2), u(c->perm), u(c->perm2), u(c->cell_start), u(c->d_n), u(c->stream),
            (uint64_t)c->cell_cur, (uint64_t)c->n_slot, (uint64_t)c->n, (uint64_t)c->tile_cells, (uint64_t)c->pair_cap,
            (uint64_t)c->pair_threads, (uint64_t)c->grid_valid, (uint64_t)c->counts_valid, (uint64_t)c->slice_valid,
            (uint64_t)c->tiles_valid, (uint64_t)c->cell_next_zeroed, (uint64_t)c->mask_tile_cells, (uint64_t)c->mask_cap,
            (uint64_t)c->tab_clo, (uint64_t)c->tab_chi, (uint64_t)c->use_masks, (uint64_t)c->fast_div, (uint64_t)c->zero_p,
            (uint64_t)(dd != 0), (uint64_t)(df != 0), f(mx), f(my), f(strength)};
}

static bool graph_possible(const SphCtx* c) {
    if (c->graph_off || c->stream == nullptr || !c->n_exact || c->n <= 0 || c->n > GRAPH_MAX_N || c->profiling ||
        !c->grid_counting || c->pending_drop || c->tail_appended || c->col_lo != 0 || c->col_hi != c->dc.nx)
        return false;
    // (a caller that is itself capturing this stream gets the plain launches recorded into its graph)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st == hipStreamCaptureStatusNone;
}

static void drop_step_graph(SphCtx* c) {
    if (c->step_graph) (void)hipGraphExecDestroy(c->step_graph);
    c->step_graph = nullptr;
    c->step_graph_sig.clear();
}

// Enqueues GRAPH_STEPS steps as one graph launch (after one plain step if a new capture is needed: it
// leaves the host state in its steady after-a-step form).  advanced = steps enqueued; 0 (host state
// untouched): the caller steps one by one.
static int graph_steps(SphCtx* c, float mx, float my, float strength, int& advanced) {
    advanced = 0;
    sph_adapt_tile_cells(c, c->n);
    std::vector<uint64_t> sig = step_sig(c, mx, my, strength);
    if (!(c->step_graph && sig == c->step_graph_sig)) {
        drop_step_graph(c);
        int rc = step_once(c, mx, my, strength);
        if (rc) return rc;
        advanced = 1;
        sph_adapt_tile_cells(c, c->n);
        sig = step_sig(c, mx, my, strength);
        const SphCtx saved = *c;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
            (void)hipGetLastError();
            c->graph_off = true;
            return SPHHIP_OK;
        }
        for (int s = 0; s < GRAPH_STEPS && !rc; s++) rc = step_once(c, mx, my, strength);
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        hipGraphExec_t exec = nullptr;
        bool ok = !rc && e == hipSuccess && graph && step_sig(c, mx, my, strength) == sig;
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {  // nothing of the capture was enqueued: back to the state before it, no more graphs
            (void)hipGetLastError();
            *c = saved;
            c->graph_off = true;
            return SPHHIP_OK;
        }
        c->step_graph = exec;
        c->step_graph_sig = sig;
    }
    HIP_TRY(hipGraphLaunch(c->step_graph, c->stream));
    advanced += GRAPH_STEPS;
    return SPHHIP_OK;
}

extern "C" int sphhip_step(SphCtx* c, float mx, float my, float strength, int nsteps) {
    NEED_STATE(c);
    if (nsteps < 0) return fail(SPHHIP_EINVAL, "sphhip_step: nsteps < 0");
    int s = 0;
    if (nsteps >= 2 * GRAPH_STEPS + 2 && graph_possible(c)) {
        // the mapped counters must describe this scene: one plain step and a wait, once
        if (((volatile int*)c->h_host_tiles)[0] < 0 || ((volatile int*)c->h_dense)[0] < 0 || ((volatile int*)c->h_dense)[1] < 0) {
            int rc = step_once(c, mx, my, strength);
            if (rc) return rc;
            s++;
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        while (nsteps - s >= GRAPH_STEPS + 1 && graph_possible(c)) {
            int advanced = 0;
            int rc = graph_steps(c, mx, my, strength, advanced);
            if (rc) return rc;
            s += advanced;
            if (advanced < GRAPH_STEPS) break;
        }
    }
    for (; s < nsteps; s++) {
        int rc = step_once(c, mx, my, strength);
        if (rc) return rc;
    }
    return SPHHIP_OK;
}

// One step in two halves for the slab driver: begin = grid build, density, and the force / integrate pass of
// the ghost columns and of the first `edge_left` / last `edge_right` owned columns; finish = the same pass
// over the interior columns.  In between the caller packs the halo messages (sphhip_pack_halo_async reads
// only columns begin has finished) and starts their exchange, which then overlaps the interior pass.
extern "C" int sphhip_step_begin(SphCtx* c, float mx, float my, float strength, int edge_left, int edge_right) {
    NEED_STATE(c);
    if (edge_left < 0 || edge_right < 0) return fail(SPHHIP_EINVAL, "sphhip_step_begin: negative edge width");
    c->profile_step++;
    int rc = sph_throttle(c);
    if (rc) return rc;
    rc = sph_grid_build(c, false);
    if (rc) return rc;
    rc = sph_density_launch(c);
    if (rc) return rc;
    // The pass is split only if the pack that follows can take its boundary-only path (it reads nothing but the
    // columns this half finishes); otherwise the compacting split would run on a half-written state.  With
    // SPHHIP_LAZY_PACK=0 / SPHHIP_GRID=radix the whole pass runs here and sphhip_step_finish has nothing to do.
    const bool can_split = c->lazy_pack && c->grid_counting;
    c->split.edge_l = edge_left;
    c->split.edge_r = edge_right;
    return sph_force_launch(c, mx, my, strength, true, can_split ? 1 : 0, edge_left, edge_right);
}

extern "C" int sphhip_step_finish(SphCtx* c) {
    NEED_STATE_MIDSTEP(c);
    return sph_force_launch(c, c->split.mx, c->split.my, c->split.strength, true, 2);
}

extern "C" int sphhip_step_reference_order(SphCtx* c, float mx, float my, float strength, int nsteps) {
    NEED_EXACT(c);
    if (nsteps < 0) return fail(SPHHIP_EINVAL, "sphhip_step_reference_order: nsteps < 0");
    return sph_step_reference_order(c, mx, my, strength, nsteps);
}

// ---- diagnostics ----------------------------------------------------------------------------
extern "C" int sphhip_get_cells(SphCtx* c, int* keys, int* order, int* cell_start, int* cell_end) {
    NEED_EXACT(c);
    if (!c->grid_valid) return fail(SPHHIP_ESTATE, "sphhip_get_cells: no valid grid (positions changed since the last build)");
    const int n = c->n;
    if (keys && n) HIP_TRY(hipMemcpyAsync(keys, c->keys, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    std::vector<uint32_t> idv;
    if (order && n) {
        idv.resize(n);
        HIP_TRY(hipMemcpyAsync(idv.data(), c->idv, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    }
    std::vector<int> cs;
    if (cell_start || cell_end) {
        cs.resize((size_t)c->ncell + 1);
        HIP_TRY(hipMemcpyAsync(cs.data(), c->cell_start, sizeof(int) * cs.size(), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!cs.empty() && c->tab_chi >= 0) {  // the table is maintained up to column tab_chi only: complete it
        const size_t last = (size_t)c->tab_chi * c->dc.ny;
        for (size_t k = last + 1; k < cs.size(); k++) cs[k] = cs[last];
    }
    if (order)
        for (int i = 0; i < n; i++) order[i] = (int)(idv[i] & SPH_ID_MASK);
    if (cell_start) memcpy(cell_start, cs.data(), sizeof(int) * (size_t)c->ncell);
    if (cell_end) memcpy(cell_end, cs.data() + 1, sizeof(int) * (size_t)c->ncell);
    return SPHHIP_OK;
}

static int fetch_counts_sorted_by_id(SphCtx* c, std::vector<int>& cnt, std::vector<uint32_t>& id) {
    const int n = c->n;
    cnt.resize(n);
    id.resize(n);
    if (n == 0) return SPHHIP_OK;
    HIP_TRY(hipMemcpyAsync(cnt.data(), c->ncount, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(id.data(), c->idv, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPHHIP_OK;
}

extern "C" int sphhip_get_neighbor_counts(SphCtx* c, int* counts, int n) {
    NEED_EXACT(c);
    if (!counts && n > 0) return fail(SPHHIP_EINVAL, "null argument");
    if (n != c->n) return fail(SPHHIP_EINVAL, "sphhip_get_neighbor_counts: n differs from the particle count");
    if (!c->counts_valid || !c->grid_valid)
        return fail(SPHHIP_ESTATE, "sphhip_get_neighbor_counts: call after a density pass, before positions change");
    std::vector<int> cnt;
    std::vector<uint32_t> id;
    int rc = fetch_counts_sorted_by_id(c, cnt, id);
    if (rc) return rc;
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return (id[a] & SPH_ID_MASK) < (id[b] & SPH_ID_MASK); });
    for (int k = 0; k < n; k++) counts[k] = cnt[order[k]];
    return SPHHIP_OK;
}

extern "C" int sphhip_get_tile_stats(SphCtx* c, int* stats) {
    if (!c || !stats) return fail(SPHHIP_EINVAL, "sphhip_get_tile_stats: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    stats[0] = ((volatile int*)c->h_host_tiles)[0];
    stats[1] = ((volatile int*)c->h_host_tiles)[1];
    stats[2] = ((volatile int*)c->h_host_tiles)[2];
    stats[3] = ((volatile int*)c->h_dense)[0];
    stats[4] = ((volatile int*)c->h_dense)[1];
    return SPHHIP_OK;
}

extern "C" int sphhip_get_pair_count(SphCtx* c, long long* pairs) {
    NEED_EXACT(c);
    if (!pairs) return fail(SPHHIP_EINVAL, "null argument");
    if (!c->counts_valid) return fail(SPHHIP_ESTATE, "sphhip_get_pair_count: no density pass yet");
    std::vector<int> cnt(c->n);
    int lo = 0, hi = c->n;
    if (c->n) HIP_TRY(hipMemcpyAsync(cnt.data(), c->ncount, sizeof(int) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    if (c->slice_valid) {  // on a slab: owned particles only (ghost pairs belong to the neighbour rank)
        HIP_TRY(hipMemcpyAsync(&lo, c->cell_start + (size_t)c->slice_lo * c->dc.ny, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&hi, c->cell_start + (size_t)c->slice_hi * c->dc.ny, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    long long t = 0;
    for (int i = lo; i < hi; i++) t += (cnt[i] > 0 ? cnt[i] - 1 : 0);  // self excluded
    *pairs = t;
    return SPHHIP_OK;
}

// ---- slab hooks -----------------------------------------------------------------------------
extern "C" int sphhip_set_owned_columns(SphCtx* c, int lo, int hi) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    if (lo < 0 || hi > c->dc.nx || lo >= hi) return fail(SPHHIP_EINVAL, "sphhip_set_owned_columns: bad range");
    c->col_lo = lo;
    c->col_hi = hi;
    return SPHHIP_OK;
}

struct HaloRec {
    float x, y, vx, vy;
    uint32_t idv;
};
static_assert(sizeof(HaloRec) == 20, "halo record");

// ---- the particle count lives on the device ---------------------------------------------------
// Slab steps change the particle count every step (migration, ghosts).  To keep the host from
// synchronising once per step the count is a device word (c->d_n, one slot of a small ring so that
// a kernel can read the old count and write the new one without a race); kernels read it
// themselves and launches are sized by an upper bound (sph_n_grid).  The host value c->n is exact
// only while c->n_exact; sph_sync_n reads it back (one stream synchronisation) when a caller needs it.

static int* n_slot_next(SphCtx* c) { return sph_n_slot_next(c); }
static void n_advance(SphCtx* c) { sph_n_advance(c); }

int sph_set_n_host(SphCtx* c, int n) {
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->d_n, n, 1, c->stream));
    c->n = n;
    c->n_exact = true;
    return SPHHIP_OK;
}

int sph_sync_n(SphCtx* c) {
    // (the sticky error word is read whether or not the count is known: a single domain can raise it too, e.g.
    // an overflow of the dense sub-tile list, pairs.hip: dense_defer)
    int h[2] = {c->n, 0};
    if (!c->n_exact) HIP_TRY(hipMemcpyAsync(&h[0], c->d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h[1], c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n = h[0];
    c->n_exact = true;
    if (h[1]) {
        HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
        if (h[1] & SPH_ERR_CAPACITY) return fail(SPHHIP_ECAPACITY, "a capacity of the context was exceeded during asynchronous steps (particle capacity: particles were dropped; dense sub-tile list: their particles were not updated)");
        if (h[1] & SPH_ERR_HALO_OVERFLOW) return fail(SPHHIP_ECAPACITY, "halo message buffer too small during asynchronous steps (records were dropped)");
        if (h[1] & SPH_ERR_DEBUG_BOUNDS) return fail(SPHHIP_ESTATE, "an index check of a debugging build failed (SPH_DENSE_DEFER_NOINLINE)");
        if (h[1] & SPH_ERR_OUT_OF_SLAB) return fail(SPHHIP_ESTATE, "particles farther than the halo from the owned columns were found (and dropped) by a grid build");
        return fail(SPHHIP_ECAPACITY, "received halo message with a bad record count");
    }
    return SPHHIP_OK;
}

// Classifies every particle by the column of its current position: copies it to the left buffer
// if col < send_hi_left, to the right buffer if col >= send_lo_right, and keeps it (compacted into
// the *_o arrays) if keep_lo <= col < keep_hi.  Slot order of the outputs is arbitrary (atomics):
// the next grid build sorts by (key, id), so results do not depend on it.
constexpr int SPLIT_THREADS = 1024;

// One output stream of k_split: every lane with `flag` gets a unique slot.  Slots are claimed with
// ONE global atomic per workgroup (wave ballots -> LDS -> atomicAdd by one lane): a single-address
// atomic per wave would serialise at ~11 ns each (190 us per million particles, measured).
__device__ __forceinline__ int split_claim(bool flag, int* __restrict__ counter, int* s_cnt /*[16]*/, int* s_base) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < SPLIT_THREADS / 64; w++) {
            int v = s_cnt[w];
            s_cnt[w] = tot;
            tot += v;
        }
        *s_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int slot = *s_base + s_cnt[wave] + in_wave;
    __syncthreads();
    return slot;
}

__global__ __launch_bounds__(SPLIT_THREADS) void k_split(DevConsts c, const int* __restrict__ d_n,
                                                const float2* __restrict__ pos,
                                                const float2* __restrict__ vel, const float* __restrict__ rho,
                                                const float* __restrict__ prs, const uint32_t* __restrict__ idv,
                                                int send_hi_left, int send_lo_right, int keep_lo, int keep_hi,
                                                HaloRec* __restrict__ left, int cap_left,
                                                HaloRec* __restrict__ right, int cap_right,
                                                float2* __restrict__ pos_o, float2* __restrict__ vel_o,
                                                float* __restrict__ rho_o, float* __restrict__ prs_o,
                                                uint32_t* __restrict__ idv_o, int* __restrict__ cnt_left,
                                                int* __restrict__ cnt_right, int* __restrict__ cnt_keep,
                                                int* __restrict__ err, const int* __restrict__ cell_start,
                                                const int* __restrict__ n_sorted, int slice_lo_cell,
                                                int slice_hi_cell) {
    __shared__ int s_cnt[SPLIT_THREADS / 64];
    __shared__ int s_base;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < *d_n;
    // After a step on a slab the state still holds the ghost copies (unmoved).  Slots are still in
    // the order of that step's sort, so the particles this rank integrated are exactly the slot
    // range of its owned columns; everything else is a stale ghost and is dropped here.
    // (Slots from *n_sorted on were appended after that sort: received records, always kept.)
    if (live && cell_start && i < *n_sorted && (i < cell_start[slice_lo_cell] || i >= cell_start[slice_hi_cell]))
        live = false;
    float2 p = make_float2(0.0f, 0.0f), v = p;
    uint32_t w = 0;
    int col = 0;
    if (live) {
        p = pos[i];
        v = vel[i];
        w = idv[i];
        col = sph_cell_coord(p.x, c.xmin, c.inv_cell, c.nx);
    }
    const HaloRec r = {p.x, p.y, v.x, v.y, w};
    const bool to_left = live && col < send_hi_left;
    const bool to_right = live && col >= send_lo_right;
    const bool keep = live && col >= keep_lo && col < keep_hi;
    if (__syncthreads_or(to_left)) {  // uniform over the workgroup
        int k = split_claim(to_left, cnt_left, s_cnt, &s_base);
        if (to_left) {
            if (k < cap_left) left[k] = r;
            else atomicOr(err, SPH_ERR_HALO_OVERFLOW);
        }
    }
    if (__syncthreads_or(to_right)) {
        int k = split_claim(to_right, cnt_right, s_cnt, &s_base);
        if (to_right) {
            if (k < cap_right) right[k] = r;
            else atomicOr(err, SPH_ERR_HALO_OVERFLOW);
        }
    }
    if (__syncthreads_or(keep)) {
        int k = split_claim(keep, cnt_keep, s_cnt, &s_base);
        if (keep) {
            pos_o[k] = p;
            vel_o[k] = v;
            rho_o[k] = rho[i];
            prs_o[k] = prs[i];
            idv_o[k] = w;
        }
    }
}

// Enqueues the split (no host synchronisation).  cnt_left / cnt_right live in the message headers
// (fixed-size exchange) or in c->counters[0..1]; the kept count becomes the new device-side n.
static int split_enqueue(SphCtx* c, int send_hi_left, int send_lo_right, int keep_lo, int keep_hi, HaloRec* d_left,
                         int cap_left, int* cnt_left, HaloRec* d_right, int cap_right, int* cnt_right) {
    int* keep = n_slot_next(c);
    HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(int) * 4, c->stream));
    HIP_TRY(hipMemsetAsync(keep, 0, sizeof(int), c->stream));
    if (cnt_left != c->counters) HIP_TRY(hipMemsetAsync(cnt_left, 0, sizeof(HaloRec), c->stream));
    if (cnt_right != c->counters + 1) HIP_TRY(hipMemsetAsync(cnt_right, 0, sizeof(HaloRec), c->stream));
    const int ng = sph_n_grid(c);
    if (ng > 0) {
        hipLaunchKernelGGL(k_split, dim3((ng + SPLIT_THREADS - 1) / SPLIT_THREADS), dim3(SPLIT_THREADS), 0, c->stream, c->dc, c->d_n, c->pos, c->vel,
                           c->rho, c->prs, c->idv, send_hi_left, send_lo_right, keep_lo, keep_hi, d_left, cap_left,
                           d_right, cap_right, c->pos2, c->vel2, c->rho2, c->prs2, c->idv2, cnt_left, cnt_right, keep,
                           c->d_err, c->slice_valid ? c->cell_start : nullptr, c->counters + SPH_N_SORTED_SLOT,
                           c->slice_lo * c->dc.ny, c->slice_hi * c->dc.ny);
        LAUNCH_CHECK();
    }
    n_advance(c);
    c->n_exact = false;
    std::swap(c->pos, c->pos2);
    std::swap(c->vel, c->vel2);
    std::swap(c->rho, c->rho2);
    std::swap(c->prs, c->prs2);
    std::swap(c->idv, c->idv2);
    c->grid_valid = false;
    c->counts_valid = false;
    c->slice_valid = false;
    c->pending_drop = false;
    return SPHHIP_OK;
}

// The same classification for the exchange only, WITHOUT compacting the state: after a step the slots are
// still in the order of the step's sort, and a particle moves less than a cell per step, so everything
// that must go left sits in the slot range of the first SPHHIP_HALO_COLS + 1 owned columns (likewise
// right).  Only those ranges are read; the stale ghost copies stay where they are and are dropped by the
// next grid build (SphCtx::pending_drop, k_cell_count).  Workgroups [0, nb) serve the left message,
// [nb, 2 nb) the right one; record slots are claimed with one atomic per workgroup pass on a counter in
// the context, and the last workgroup to finish (ticket) writes both counts into the message headers and
// re-arms the counters: no memset, no host round trip.
constexpr int PACK_BLOCKS = 64;  // (3 boundary columns of a 1 M slab are 12 k particles: one pass of 64 workgroups; 16 took three passes, 9 us)

__global__ __launch_bounds__(256) void k_pack_boundary(DevConsts c, const float2* __restrict__ pos,
                                                        const float2* __restrict__ vel,
                                                        const uint32_t* __restrict__ idv,
                                                        const int* __restrict__ cell_start, int left_lo_cell,
                                                        int left_hi_cell, int right_lo_cell, int right_hi_cell,
                                                        int send_hi_left, int send_lo_right, HaloRec* __restrict__ left,
                                                        int cap_left, HaloRec* __restrict__ right, int cap_right,
                                                        int* __restrict__ msg_cnt, int* __restrict__ ticket,
                                                        int* __restrict__ err) {
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const int side = blockIdx.x / PACK_BLOCKS, b = blockIdx.x % PACK_BLOCKS;
    HaloRec* out = side ? right : left;
    const int cap = side ? cap_right : cap_left;
    if (out) {
        const int lo = cell_start[side ? right_lo_cell : left_lo_cell], hi = cell_start[side ? right_hi_cell : left_hi_cell];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int i0 = lo + b * 256; i0 < hi; i0 += PACK_BLOCKS * 256) {  // uniform trip count over the workgroup
            const int i = i0 + (int)threadIdx.x;
            bool send = false;
            HaloRec r = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
            if (i < hi) {
                const float2 p = pos[i], v = vel[i];
                const int col = sph_cell_coord(p.x, c.xmin, c.inv_cell, c.nx);
                send = side ? col >= send_lo_right : col < send_hi_left;
                r = {p.x, p.y, v.x, v.y, idv[i]};
            }
            const unsigned long long bal = __ballot(send);
            if (lane == 0) s_cnt[wave] = __popcll(bal);
            __syncthreads();
            if (threadIdx.x == 0) {
                const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
                s_base = tot ? atomicAdd(&msg_cnt[side], tot) : 0;
            }
            __syncthreads();
            int k = s_base + __popcll(bal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; w++) k += s_cnt[w];
            if (send) {
                if (k < cap) out[1 + k] = r;  // record 0 is the header
                else atomicOr(err, SPH_ERR_HALO_OVERFLOW);
            }
            __syncthreads();
        }
    }
    // last workgroup out: counts into the headers, counters back to zero
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = atomicAdd(ticket, 1);
        if (done == (int)gridDim.x - 1) {
            __threadfence();
            const int nl = atomicExch(&msg_cnt[0], 0), nr = atomicExch(&msg_cnt[1], 0);
            if (left) left[0] = {__int_as_float(nl < cap_left ? nl : cap_left), 0.0f, 0.0f, 0.0f, 0u};
            if (right) right[0] = {__int_as_float(nr < cap_right ? nr : cap_right), 0.0f, 0.0f, 0.0f, 0u};
            atomicExch(ticket, 0);
        }
    }
}

static int split_impl(SphCtx* c, int send_hi_left, int send_lo_right, int keep_lo, int keep_hi, void* d_left,
                      int cap_left, int* n_left, void* d_right, int cap_right, int* n_right) {
    if (!n_left || !n_right) return fail(SPHHIP_EINVAL, "null argument");
    if ((cap_left > 0 && !d_left) || (cap_right > 0 && !d_right)) return fail(SPHHIP_EINVAL, "null buffer");
    int rc = split_enqueue(c, send_hi_left, send_lo_right, keep_lo, keep_hi, (HaloRec*)d_left, cap_left, c->counters,
                           (HaloRec*)d_right, cap_right, c->counters + 1);
    if (rc) return rc;
    int h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, c->counters, sizeof(int) * 2, hipMemcpyDeviceToHost, c->stream));
    rc = sph_sync_n(c);  // synchronises the stream
    if (rc) return rc;
    *n_left = h[0];
    *n_right = h[1];
    return SPHHIP_OK;
}

extern "C" int sphhip_pack_halo_async(SphCtx* c, int send_hi_left, int send_lo_right, void* d_msg_left, int cap_left,
                                      void* d_msg_right, int cap_right) {
    NEED_STATE_MIDSTEP(c);
    if (cap_left < 0 || cap_right < 0) return fail(SPHHIP_EINVAL, "sphhip_pack_halo_async: negative capacity");
    HaloRec* l = (HaloRec*)d_msg_left;
    HaloRec* r = (HaloRec*)d_msg_right;
    // record 0 of a message is its header (word 0 = record count); records follow from row 1
    if (c->slice_valid && !c->pending_drop && !c->tail_appended && c->grid_counting && c->lazy_pack && c->slice_lo == c->col_lo &&
        c->slice_hi == c->col_hi && (!l || send_hi_left <= c->col_lo + SPHHIP_HALO_COLS) &&
        (!r || send_lo_right >= c->col_hi - SPHHIP_HALO_COLS)) {
        // fast path: the state is in the slot order of the last sort (a step has run since, nothing else)
        const int ny = c->dc.ny;
        const int reach = SPHHIP_PACK_REACH;  // a particle crosses less than one column per step
        // between the two halves of a step only the columns sphhip_step_begin finished hold the new state
        if (c->split.pending && ((l && c->split.edge_l < reach) || (r && c->split.edge_r < reach)))
            return fail(SPHHIP_ESTATE, "sphhip_pack_halo_async: sphhip_step_begin finished fewer than SPHHIP_PACK_REACH "
                                       "columns on a side that has a message");
        const int l_hi = c->col_lo + reach < c->col_hi ? c->col_lo + reach : c->col_hi;
        const int r_lo = c->col_hi - reach > c->col_lo ? c->col_hi - reach : c->col_lo;
        hipLaunchKernelGGL(k_pack_boundary, dim3(2 * PACK_BLOCKS), dim3(256), 0, c->stream, c->dc, c->pos, c->vel, c->idv,
                           c->cell_start, c->col_lo * ny, l_hi * ny, r_lo * ny, c->col_hi * ny, send_hi_left, send_lo_right,
                           l, cap_left, r, cap_right, c->counters + SPH_MSG_COUNT_SLOT, c->counters + SPH_MSG_TICKET_SLOT, c->d_err);
        LAUNCH_CHECK();
        c->pending_drop = true;
        c->grid_valid = false;  // positions have moved since the sort (the fused step leaves it false anyway)
        c->counts_valid = false;
        return SPHHIP_OK;
    }
    // The compacting split reads and rewrites the whole state: not between the two halves of a step.
    if (c->split.pending)
        return fail(SPHHIP_ESTATE, "sphhip_pack_halo_async between sphhip_step_begin and sphhip_step_finish: the send ranges "
                                   "exceed the halo (only the boundary-only pack may run there)");
    return split_enqueue(c, l ? send_hi_left : -1, r ? send_lo_right : 0x7fffffff, c->col_lo - SPHHIP_HALO_COLS,
                         c->col_hi + SPHHIP_HALO_COLS, l ? l + 1 : nullptr, l ? cap_left : 0, l ? (int*)l : c->counters,
                         r ? r + 1 : nullptr, r ? cap_right : 0, r ? (int*)r : c->counters + 1);
}

extern "C" int sphhip_pack_halo(SphCtx* c, int send_hi_left, int send_lo_right, void* d_left, int cap_left,
                                int* n_left, void* d_right, int cap_right, int* n_right) {
    NEED_STATE(c);
    // A particle that has just left the owned columns is handed to the neighbour in this very call,
    // but it is still a neighbour of the particles that stay: keep it (as a ghost, one more step)
    // while it lies inside the halo.  The next call drops it through the owned-slot slice.
    return split_impl(c, send_hi_left, send_lo_right, c->col_lo - SPHHIP_HALO_COLS, c->col_hi + SPHHIP_HALO_COLS,
                      d_left, cap_left, n_left, d_right, cap_right, n_right);
}

extern "C" int sphhip_keep_columns(SphCtx* c, int lo, int hi) {
    NEED_STATE(c);
    int nl, nr;
    return split_impl(c, -1, 0x7fffffff, lo, hi, nullptr, 0, &nl, nullptr, 0, &nr);
}

extern "C" int sphhip_drop_ghosts(SphCtx* c) {
    NEED_STATE(c);
    if (!c->slice_valid) return SPHHIP_OK;  // nothing was stepped with ghosts
    int nl, nr;
    return split_impl(c, -1, 0x7fffffff, 0, c->dc.nx, nullptr, 0, &nl, nullptr, 0, &nr);
}

// Appends records behind the current particles.  The record count is either immediate or the
// header word of a fixed-size message; the new particle count goes to the next ring slot.
__global__ __launch_bounds__(256) void k_append(const HaloRec* __restrict__ rec, const int* __restrict__ d_cnt,
                                                 int cnt_imm, int cap, const int* __restrict__ n_in,
                                                 int* __restrict__ n_out, int n_max, int* __restrict__ err,
                                                 float2* __restrict__ pos, float2* __restrict__ vel,
                                                 float* __restrict__ rho, float* __restrict__ prs,
                                                 uint32_t* __restrict__ idv, float rest_density) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cnt = d_cnt ? *d_cnt : cnt_imm;
    const int at = *n_in;
    int flag = 0;
    if (cnt < 0 || cnt > cap) {
        flag |= SPH_ERR_MSG_BAD;
        cnt = cnt < 0 ? 0 : cap;
    }
    if (at + cnt > n_max) {
        flag |= SPH_ERR_CAPACITY;
        cnt = n_max - at;
    }
    if (i == 0) {
        *n_out = at + cnt;
        if (flag) atomicOr(err, flag);
    }
    if (i >= cnt) return;
    HaloRec r = rec[i];
    pos[at + i] = make_float2(r.x, r.y);
    vel[at + i] = make_float2(r.vx, r.vy);
    rho[at + i] = rest_density;
    prs[at + i] = 0.0f;
    idv[at + i] = r.idv;
}

static int append_enqueue(SphCtx* c, const HaloRec* rec, const int* d_cnt, int cnt_imm, int cap) {
    // (stale ghost copies of the last step may still be in the arrays: they are told from the appended
    // records by their slot, below / from the count at the last sort -- see k_split, k_cell_count)
    int* out = n_slot_next(c);
    int grid = (cap + 255) / 256;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_append, dim3(grid), dim3(256), 0, c->stream, rec, d_cnt, cnt_imm, cap, c->d_n, out, c->n_max,
                       c->d_err, c->pos, c->vel, c->rho, c->prs, c->idv, c->params.rest_density);
    LAUNCH_CHECK();
    n_advance(c);
    c->have_state = true;
    c->grid_valid = false;
    c->counts_valid = false;
    c->tail_appended = true;
    return SPHHIP_OK;
}

extern "C" int sphhip_append(SphCtx* c, const void* d_records, int n) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    if (n < 0 || (n > 0 && !d_records)) return fail(SPHHIP_EINVAL, "sphhip_append: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_exact && c->n + n > c->n_max) return fail(SPHHIP_ECAPACITY, "sphhip_append: exceeds the context capacity");
    if (n == 0) return SPHHIP_OK;
    int rc = append_enqueue(c, (const HaloRec*)d_records, nullptr, n, n);
    if (rc) return rc;
    if (c->n_exact) c->n += n;  // stays exact: the kernel cannot have clamped
    return SPHHIP_OK;
}

extern "C" int sphhip_append_message(SphCtx* c, const void* d_msg, int cap) {
    if (!c || !d_msg) return fail(SPHHIP_EINVAL, "sphhip_append_message: null argument");
    if (cap < 0) return fail(SPHHIP_EINVAL, "sphhip_append_message: negative capacity");
    HIP_TRY(hipSetDevice(c->device));
    const HaloRec* m = (const HaloRec*)d_msg;
    int rc = append_enqueue(c, m + 1, (const int*)m, 0, cap);
    c->n_exact = false;
    return rc;
}

// Two received messages in one launch (a slab step appends one from each side: two launches at the ~4.5 us floor
// each became one).  Message a lands behind the current particles, message b behind message a.
__global__ __launch_bounds__(256) void k_append2(const HaloRec* __restrict__ msg_a, int cap_a, const HaloRec* __restrict__ msg_b, int cap_b,
                                                  const int* __restrict__ n_in, int* __restrict__ n_out, int n_max,
                                                  int* __restrict__ err, float2* __restrict__ pos, float2* __restrict__ vel,
                                                  float* __restrict__ rho, float* __restrict__ prs,
                                                  uint32_t* __restrict__ idv, float rest_density) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int at = *n_in;
    int ca = *(const int*)msg_a, cb = *(const int*)msg_b;
    int flag = 0;
    if (ca < 0 || ca > cap_a) {
        flag |= SPH_ERR_MSG_BAD;
        ca = ca < 0 ? 0 : cap_a;
    }
    if (cb < 0 || cb > cap_b) {
        flag |= SPH_ERR_MSG_BAD;
        cb = cb < 0 ? 0 : cap_b;
    }
    if (at + ca > n_max) {
        flag |= SPH_ERR_CAPACITY;
        ca = n_max - at;
    }
    if (at + ca + cb > n_max) {
        flag |= SPH_ERR_CAPACITY;
        cb = n_max - at - ca;
    }
    if (t == 0) {
        *n_out = at + ca + cb;
        if (flag) atomicOr(err, flag);
    }
    const bool second = t >= cap_a;  // threads [0, cap_a): message a; [cap_a, cap_a + cap_b): message b
    const int i = second ? t - cap_a : t;
    if (i >= (second ? cb : ca)) return;
    const HaloRec r = (second ? msg_b : msg_a)[1 + i];
    const int dst = at + (second ? ca : 0) + i;
    pos[dst] = make_float2(r.x, r.y);
    vel[dst] = make_float2(r.vx, r.vy);
    rho[dst] = rest_density;
    prs[dst] = 0.0f;
    idv[dst] = r.idv;
}

extern "C" int sphhip_append_messages(SphCtx* c, const void* d_msg_a, int cap_a, const void* d_msg_b, int cap_b) {
    if (!c) return fail(SPHHIP_EINVAL, "sphhip_append_messages: null context");
    if (cap_a < 0 || cap_b < 0) return fail(SPHHIP_EINVAL, "sphhip_append_messages: negative capacity");
    if (!d_msg_a && !d_msg_b) return SPHHIP_OK;
    if (!d_msg_a || !d_msg_b) return sphhip_append_message(c, d_msg_a ? d_msg_a : d_msg_b, d_msg_a ? cap_a : cap_b);
    HIP_TRY(hipSetDevice(c->device));
    int* out = n_slot_next(c);
    int grid = (int)(((long long)cap_a + cap_b + 255) / 256);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_append2, dim3(grid), dim3(256), 0, c->stream, (const HaloRec*)d_msg_a, cap_a, (const HaloRec*)d_msg_b, cap_b, c->d_n,
                       out, c->n_max, c->d_err, c->pos, c->vel, c->rho, c->prs, c->idv, c->params.rest_density);
    LAUNCH_CHECK();
    n_advance(c);
    c->have_state = true;
    c->grid_valid = false;
    c->counts_valid = false;
    c->tail_appended = true;
    c->n_exact = false;
    return SPHHIP_OK;
}

extern "C" int sphhip_sync_counts(SphCtx* c, int* n) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sph_sync_n(c);
    if (n) *n = c->n;
    return rc;
}

__global__ __launch_bounds__(256) void k_col_hist(DevConsts c, int n, const float2* __restrict__ pos,
                                                   int* __restrict__ hist) {
    // The state is in (or near) cell order, and the cell keys are column-major: the 64 particles of a wave sit in one or
    // two columns.  One atomic per run of equal columns in a wave (as k_cell_count does for cells) instead of one per
    // particle -- 4000 serialised atomics per column address made a rebalance of 1 M particles take 1.9 ms.
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const int col = ok ? sph_cell_coord(pos[i].x, c.xmin, c.inv_cell, c.nx) : -1;
    const int lane = sph_lane_id();
    const int prev = __shfl_up(col, 1, 64);
    const bool head = ok && (lane == 0 || prev != col);
    const unsigned long long heads = __ballot(head);
    const int n_ok = __popcll(__ballot(ok));  // the in-range lanes are a prefix of the wave
    if (head) {
        const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const unsigned long long above = heads & ~upto;
        const int next = above ? __ffsll((long long)above) - 1 : n_ok;
        atomicAdd(&hist[col], next - lane);
    }
}

extern "C" int sphhip_column_histogram(SphCtx* c, int* hist) {
    NEED_EXACT(c);
    if (!hist) return fail(SPHHIP_EINVAL, "null argument");
    // cell_start is free between steps only when the grid is stale; use keys2 as scratch instead
    int* d_hist = (int*)c->keys2;
    if (c->dc.nx > c->n_max) return fail(SPHHIP_ECAPACITY, "sphhip_column_histogram: nx exceeds scratch");
    HIP_TRY(hipMemsetAsync(d_hist, 0, sizeof(int) * (size_t)c->dc.nx, c->stream));
    if (c->n > 0) {
        hipLaunchKernelGGL(k_col_hist, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->dc, c->n, c->pos, d_hist);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipMemcpyAsync(hist, d_hist, sizeof(int) * (size_t)c->dc.nx, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPHHIP_OK;
}

extern "C" int sphhip_selftest_division(int device, unsigned long long* checked, unsigned long long* mismatches) {
    if (!checked || !mismatches) return fail(SPHHIP_EINVAL, "null argument");
    return sph_selftest_division(device, checked, mismatches);
}

// ---- measurement ----------------------------------------------------------------------------
extern "C" int sphhip_set_profiling(SphCtx* c, int enable) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    timer_flush(c);
    c->profiling = enable != 0;
    c->profile_mask = ~0u;
    return SPHHIP_OK;
}

extern "C" int sphhip_set_profiling_period(SphCtx* c, int period) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    if (period < 1) return fail(SPHHIP_EINVAL, "sphhip_set_profiling_period: period < 1");
    c->profile_period = period;
    c->profile_step = 0;
    return SPHHIP_OK;
}

extern "C" int sphhip_set_profiling_mask(SphCtx* c, unsigned mask) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    timer_flush(c);
    c->profile_mask = mask;
    c->profiling = mask != 0;
    return SPHHIP_OK;
}

extern "C" int sphhip_get_phase_times(SphCtx* c, float* ms, int* launches) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    timer_flush(c);
    for (int ph = 0; ph < SPHHIP_PH_COUNT; ph++) {
        if (ms) ms[ph] = (float)c->timer.ms[ph];
        if (launches) launches[ph] = (int)c->timer.launches[ph];
    }
    return SPHHIP_OK;
}

extern "C" int sphhip_reset_phase_times(SphCtx* c) {
    if (!c) return fail(SPHHIP_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    timer_flush(c);
    for (int ph = 0; ph < SPHHIP_PH_COUNT; ph++) {
        c->timer.ms[ph] = 0;
        c->timer.launches[ph] = 0;
    }
    return SPHHIP_OK;
}

// ---- the entry point that allocates nothing (include/sphhip.h: sphhip_step_static) --------------
// Every buffer sphhip_create would allocate for a context of SPHHIP_STATIC_MAX_N particles on a grid of
// SPHHIP_STATIC_MAX_CELLS cells is carved out of one file-scope device array.  Every GPU the library runs on
// holds its own copy of that array (at its own address) until the process ends; calls on one GPU share it and
// therefore must not overlap.
constexpr size_t STATIC_SCRATCH_BYTES = 16u << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {
struct StaticSlot {
    unsigned char* ptr = nullptr;  // device address of `scratch` on this GPU (hipGetSymbolAddress), never the symbol itself
    SphCtx ctx;
    int host_tiles[3] = {-1, 0, -1};   // stand-ins for the mapped host words of a normal context: always "unknown"
    int host_dense[2] = {-1, -1};
};
constexpr int STATIC_MAX_DEVICES = 64;
StaticSlot* g_static[STATIC_MAX_DEVICES] = {nullptr};

// ids are 0..n-1 here: record k of the output is the particle with id k (the caller's index order)
__global__ __launch_bounds__(256) void k_pack_aos_by_id(uint32_t* __restrict__ aos, const int* __restrict__ d_n,
                                                         const float2* __restrict__ pos, const float2* __restrict__ vel,
                                                         const float* __restrict__ rho, const float* __restrict__ prs,
                                                         const uint32_t* __restrict__ idv) {
    const int n = *d_n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = idv[i];
    const uint32_t id = w & SPH_ID_MASK;
    if (id >= (uint32_t)n) return;  // (cannot happen: ids are assigned by this call)
    uint32_t* r = aos + (size_t)id * 7;
    const float2 p = pos[i], v = vel[i];
    r[0] = __float_as_uint(p.x);
    r[1] = __float_as_uint(p.y);
    r[2] = __float_as_uint(v.x);
    r[3] = __float_as_uint(v.y);
    r[4] = __float_as_uint(rho[i]);
    r[5] = __float_as_uint(prs[i]);
    r[6] = (w & SPH_INVALID_BIT) ? 0u : 1u;
}
}  // namespace

extern "C" int sphhip_step_static(const SphParams* p, int device, const void* d_aos_in, void* d_aos_out, int n, float mx,
                                  float my, float strength, int nsteps, void* hip_stream) {
    if (!p || (n > 0 && (!d_aos_in || !d_aos_out))) return fail(SPHHIP_EINVAL, "sphhip_step_static: null argument");
    if (n < 0 || nsteps < 0) return fail(SPHHIP_EINVAL, "sphhip_step_static: negative n / nsteps");
    if (n > SPHHIP_STATIC_MAX_N)
        return fail(SPHHIP_ECAPACITY, "sphhip_step_static: n exceeds SPHHIP_STATIC_MAX_N (use a context: sphhip_create / sphhip_step)");
    SphDerived d;
    int rc = sphhip_derive(p, &d);
    if (rc) return rc;
    if ((long long)d.nx * d.ny > SPHHIP_STATIC_MAX_CELLS)
        return fail(SPHHIP_ECAPACITY, "sphhip_step_static: the grid has more than SPHHIP_STATIC_MAX_CELLS cells (use a context)");
    if (n == 0) return SPHHIP_OK;
    if (device < 0 || device >= STATIC_MAX_DEVICES) return fail(SPHHIP_EINVAL, "sphhip_step_static: device index out of range");
    HIP_TRY(hipSetDevice(device));
    // set-up, once per device, with that device current: the address of this GPU's copy of the array
    if (!g_static[device]) {
        unsigned char* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(SPHHIP_EHIP, "sphhip_step_static: hipGetSymbolAddress returned a null address");
        static StaticSlot slots[STATIC_MAX_DEVICES];  // host memory of the library image: nothing is allocated
        slots[device].ptr = ptr;
        g_static[device] = &slots[device];
    }
    StaticSlot& sl = *g_static[device];
    SphCtx* c = &sl.ctx;
    *c = SphCtx();
    // the buffers are sized for the maxima, whatever this call's n and grid are (one layout, checked once)
    ctx_configure(c, *p, d, SPHHIP_STATIC_MAX_N, device);
    const int ncell = c->ncell;
    c->ncell = SPHHIP_STATIC_MAX_CELLS;
    size_t off = 0;
    ctx_buffers(c, false, [&](void** slot, size_t bytes, bool) {
        *slot = sl.ptr + off;
        off += (std::max<size_t>(bytes, 4) + 255) & ~(size_t)255;
    });
    c->ncell = ncell;
    // two words for what a normal context keeps in mapped host memory (the kernels write there, this path never reads it)
    c->d_host_tiles = (int*)(sl.ptr + off);
    c->d_dense = (int*)(sl.ptr + off + 256);
    off += 512;
    if (off > STATIC_SCRATCH_BYTES) return fail(SPHHIP_ECAPACITY, "sphhip_step_static: internal layout exceeds the static scratch array");
    ctx_wire(c);
    sl.host_tiles[0] = -1;
    sl.host_tiles[2] = -1;
    sl.host_tiles[1] = 0;
    sl.host_dense[0] = sl.host_dense[1] = -1;
    c->h_host_tiles = sl.host_tiles;
    c->h_dense = sl.host_dense;
    c->own_stream = nullptr;
    c->stream = (hipStream_t)hip_stream;
    c->graph_off = true;      // (instantiating a graph allocates)
    c->profiling = false;     // (so does creating events)
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->num_cus = v; }
    // The array keeps the last call's contents: everything that must start at zero (counters, cell tables, index
    // arrays, tile list) is cleared -- simply all of the part in use.
    HIP_TRY(hipMemsetAsync(sl.ptr, 0, off, c->stream));
    if ((rc = ctx_write_sinks(c))) return rc;
    hipLaunchKernelGGL(k_unpack_aos, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)d_aos_in, n, 0u, c->pos,
                       c->vel, c->rho, c->prs, c->idv);
    LAUNCH_CHECK();
    if ((rc = sph_set_n_host(c, n))) return rc;
    c->have_state = true;
    for (int s = 0; s < nsteps; s++)
        if ((rc = step_once(c, mx, my, strength))) return rc;
    hipLaunchKernelGGL(k_pack_aos_by_id, dim3((n + 255) / 256), dim3(256), 0, c->stream, (uint32_t*)d_aos_out, c->d_n, c->pos,
                       c->vel, c->rho, c->prs, c->idv);
    LAUNCH_CHECK();
    return SPHHIP_OK;
}
