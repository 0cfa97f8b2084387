// host_shim.cpp -- C++ host side above the C ABI: the reference's step interface and dispatcher
// entry points for the MI355X path (see include/sph_hip.h).  Plain C++ (no HIP headers: the
// reference's global `float2` collides with HIP's); links against libsphhip.so.
#include "../../include/sph_hip.h"
#include "../../include/sphhip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

namespace {
SphCtx* g_ctx = nullptr;
int g_cap = 0;
int g_resident = -1;  // particle count of the device-flavour interface's resident state (-1: none)
std::string g_err;

bool fail(const char* where, int rc) {
    g_err = std::string(where) + ": " + sphhip_last_error() + " (code " + std::to_string(rc) + ")";
    std::fprintf(stderr, "[sph_hip] %s\n", g_err.c_str());
    return false;
}

// One cached context per process (the reference's CUDA flavour keeps a hidden global device buffer,
// SPH/sph.cu:10); re-created when a larger N arrives.
bool ensure_ctx(int n) {
    if (g_ctx && n <= g_cap) return true;
    if (g_ctx) {
        sphhip_destroy(g_ctx);
        g_ctx = nullptr;
    }
    SphParams p;
    sphhip_default_params(&p);
    int cap = n > 1024 ? n : 1024;
    int rc = sphhip_create(&p, cap, 0, &g_ctx);
    if (rc) return fail("sphhip_create", rc);
    g_cap = cap;
    return true;
}

bool upload(std::vector<Particle>& v, int n) {
    if ((int)v.size() < n) {
        g_err = "particle vector smaller than N";
        return false;
    }
    if (!ensure_ctx(n)) return false;
    g_resident = -1;  // the vector interface replaces whatever state was resident
    int rc = sphhip_upload(g_ctx, v.data(), n, 0);
    return rc ? fail("sphhip_upload", rc) : true;
}

bool download(std::vector<Particle>& v, int n) {
    int rc = sphhip_download(g_ctx, v.data(), n);
    return rc ? fail("sphhip_download", rc) : true;
}
}  // namespace

// Message of the most recent shim call's failure (empty if it succeeded): every public call clears it first.
const char* hip_lastError() { return g_err.c_str(); }

void hip_cleanupSimulation() {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (g_ctx) sphhip_destroy(g_ctx);
    g_ctx = nullptr;
    g_cap = 0;
    g_resident = -1;
}

void hip_initSimulation(std::vector<Particle>& particles, int N) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    particles.resize(N);
    // SPH/sph.cpp:136-152, same expression order in binary32
    int cols = static_cast<int>(std::ceil(std::sqrt((double)N)));
    int rows = cols > 0 ? (N + cols - 1) / cols : 0;
    if (rows == 0) rows = 1;
    const float minPos = 0.05f, maxPos = 1.95f;
    const float range = maxPos - minPos;
    float spacingX = range / (cols > 1 ? cols - 1 : 1);
    float spacingY = range / (rows > 1 ? rows - 1 : 1);
    float spacing = spacingY < spacingX ? spacingY : spacingX;
    float startX = minPos + (range - (cols - 1) * spacing) / 2.0f;
    float startY = maxPos - (range - (rows - 1) * spacing) / 2.0f;
    for (int i = 0; i < N; i++) {
        std::memset((void*)&particles[i], 0, sizeof(Particle));
        particles[i].pos = float2(startX + (i % cols) * spacing, startY - (i / cols) * spacing);
        particles[i].vel = float2(0.0f, 0.0f);
        particles[i].density = 1000.0f;  // REST_DENSITY
        particles[i].pressure = 0.0f;
        particles[i].valid = true;
    }
}

void hip_buildNeighbors(std::vector<Particle>& particles, int N, std::vector<int>* keys, std::vector<int>* order,
                        std::vector<int>* cellStart, std::vector<int>* cellEnd) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!upload(particles, N)) return;
    int rc = sphhip_build_grid(g_ctx);
    if (rc) {
        fail("sphhip_build_grid", rc);
        return;
    }
    SphParams p;
    SphDerived d;
    sphhip_default_params(&p);
    sphhip_derive(&p, &d);
    if (keys) keys->resize(N);
    if (order) order->resize(N);
    if (cellStart) cellStart->resize((size_t)d.nx * d.ny);
    if (cellEnd) cellEnd->resize((size_t)d.nx * d.ny);
    rc = sphhip_get_cells(g_ctx, keys ? keys->data() : nullptr, order ? order->data() : nullptr,
                          cellStart ? cellStart->data() : nullptr, cellEnd ? cellEnd->data() : nullptr);
    if (rc) fail("sphhip_get_cells", rc);
}

void hip_computeDensityPressure(std::vector<Particle>& particles, int N) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!upload(particles, N)) return;
    int rc = sphhip_density(g_ctx);
    if (rc) {
        fail("sphhip_density", rc);
        return;
    }
    download(particles, N);
}

void hip_computeForces(std::vector<Particle>& particles, int N, float2 mousePos, float interactionStrength) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!upload(particles, N)) return;
    int rc = sphhip_forces(g_ctx, mousePos.x, mousePos.y, interactionStrength);
    if (rc) {
        fail("sphhip_forces", rc);
        return;
    }
    download(particles, N);
}

void hip_integrate(std::vector<Particle>& particles, int N) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!upload(particles, N)) return;
    int rc = sphhip_integrate(g_ctx);
    if (rc) {
        fail("sphhip_integrate", rc);
        return;
    }
    download(particles, N);
}

void hip_stepSimulationResident(std::vector<Particle>& particles, int N, float2 mousePos, float interactionStrength,
                                int nsteps) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!upload(particles, N)) return;
    int rc = sphhip_step(g_ctx, mousePos.x, mousePos.y, interactionStrength, nsteps);
    if (rc) {
        fail("sphhip_step", rc);
        return;
    }
    download(particles, N);
}

void hip_stepSimulation(std::vector<Particle>& particles, int N, float2 mousePos, float interactionStrength) {
    hip_stepSimulationResident(particles, N, mousePos, interactionStrength, 1);
}

// ---- device-flavour interface (SPH/sph.cuh:37-39) ---------------------------------------------
namespace {
void write_vbo(const Particle* particles, int N, void* vbo) {  // updateVBO, SPH/sph.cu:207-216
    if (!vbo) return;
    float2* out = static_cast<float2*>(vbo);
    for (int i = 0; i < N; i++) out[i] = particles[i].valid ? particles[i].pos : float2(-1.0f, -1.0f);
}
}  // namespace

void hip_initSimulation(Particle* particles, int N, void* vbo) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!particles || N < 0) {
        g_err = "hip_initSimulation: bad argument";
        return;
    }
    std::vector<Particle> v;
    hip_initSimulation(v, N);
    if (N > 0) std::memcpy((void*)particles, (const void*)v.data(), sizeof(Particle) * (size_t)N);
    if (!ensure_ctx(N)) return;
    int rc = sphhip_upload(g_ctx, particles, N, 0);  // sph.cu:202
    if (rc) {
        fail("sphhip_upload", rc);
        return;
    }
    g_resident = N;
    write_vbo(particles, N, vbo);
}

void hip_stepSimulation(Particle* particles, int N, int* grid, int* cellStart, int* cellEnd, void* vbo, float2 mousePos,
                        float interactionStrength) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    if (!particles || N < 0) {
        g_err = "hip_stepSimulation: bad argument";
        return;
    }
    if (g_resident != N) {  // no matching initSimulation before: adopt the caller's array as the state
        if (!ensure_ctx(N)) return;
        int rc = sphhip_upload(g_ctx, particles, N, 0);
        if (rc) {
            fail("sphhip_upload", rc);
            return;
        }
        g_resident = N;
    }
    int rc = sphhip_step(g_ctx, mousePos.x, mousePos.y, interactionStrength, 1);  // sph.cu:222-225
    if (rc) {
        fail("sphhip_step", rc);
        return;
    }
    rc = sphhip_download(g_ctx, particles, N);  // sph.cu:235
    if (rc) {
        fail("sphhip_download", rc);
        return;
    }
    write_vbo(particles, N, vbo);
    if (grid || cellStart || cellEnd) {
        rc = sphhip_build_grid(g_ctx);
        if (rc) {
            fail("sphhip_build_grid", rc);
            return;
        }
        std::vector<int> keys(grid ? N : 0), order(grid ? N : 0);
        rc = sphhip_get_cells(g_ctx, grid ? keys.data() : nullptr, grid ? order.data() : nullptr, cellStart, cellEnd);
        if (rc) {
            fail("sphhip_get_cells", rc);
            return;
        }
        for (int s = 0; grid && s < N; s++) grid[order[s]] = keys[s];
    }
}

float HIP_performance_test(int N) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    std::vector<Particle> particles;
    hip_initSimulation(particles, N);
    if (!upload(particles, N)) return 0.0f;
    const int num_steps = 100, num_runs = 10;  // SPH/serial_main.cpp:124-125
    double total_ups = 0.0;
    for (int run = 0; run < num_runs; ++run) {
        std::cout << "Run: " << run + 1 << " / " << num_runs << std::endl;  // outside the timed region
        if (sphhip_synchronize(g_ctx)) return 0.0f;
        auto t0 = std::chrono::high_resolution_clock::now();
        int rc = sphhip_step(g_ctx, 0.0f, 0.0f, 0.0f, num_steps);  // mouse off, serial_main.cpp:129-130
        if (!rc) rc = sphhip_synchronize(g_ctx);
        auto t1 = std::chrono::high_resolution_clock::now();
        if (rc) {
            fail("sphhip_step", rc);
            return 0.0f;
        }
        std::chrono::duration<double> elapsed = t1 - t0;
        total_ups += num_steps / elapsed.count();
    }
    return (float)(total_ups / num_runs);
}

int HIP_main_headless(int N, int steps, const char* dump_path) {
    g_err.clear();  // hip_lastError() describes the most recent call only
    std::vector<Particle> particles;
    hip_initSimulation(particles, N);
    hip_stepSimulationResident(particles, N, float2(0.0f, 0.0f), 0.0f, steps);
    if (!g_err.empty()) return -1;
    if (dump_path && *dump_path) {
        FILE* f = std::fopen(dump_path, "wb");
        if (!f) return -1;
        std::fwrite(particles.data(), sizeof(Particle), (size_t)N, f);
        std::fclose(f);
    }
    return 0;
}

int HIP_main(int N) { return HIP_main_headless(N, 1000, nullptr); }

namespace {
// One frame: 800x800 binary PPM, y up as glOrtho(0, 2, 0, 2) (serial_main.cpp:60), point colour of
// display() in SPH/openMP_sph.cpp:218-219.
bool write_frame(const std::string& path, const std::vector<Particle>& v, int n) {
    const int W = 800, H = 800;
    std::vector<unsigned char> img((size_t)W * H * 3, 0);
    const float stiffness = 1000.0f, rest = 1000.0f;  // SPH/sph.h:15-16
    for (int i = 0; i < n; i++) {
        const Particle& p = v[i];
        if (!p.valid || std::isnan(p.pos.x) || std::isnan(p.pos.y)) continue;  // serial_main.cpp:94
        int x = (int)std::floor(p.pos.x / 2.0f * W), y = (int)std::floor((1.0f - p.pos.y / 2.0f) * H);
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        float t = p.pressure / (stiffness * rest * 0.5f);
        if (!(t < 1.0f)) t = 1.0f;
        if (!(t > 0.0f)) t = 0.0f;
        unsigned char* px = &img[((size_t)y * W + x) * 3];
        px[0] = (unsigned char)std::lround(255.0f * (0.2f + 0.8f * t));
        px[1] = (unsigned char)std::lround(255.0f * (0.6f + 0.4f * t));
        px[2] = 255;
    }
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", W, H);
    bool ok = std::fwrite(img.data(), 1, img.size(), f) == img.size();
    std::fclose(f);
    return ok;
}
}  // namespace

int HIP_main_scripted(int N, int steps, const HipMouseEvent* script, int n_events, int frame_every,
                      const char* frame_prefix, const char* dump_path) {
    g_err.clear();
    if (N <= 0 || steps < 0 || n_events < 0 || (n_events > 0 && !script) || frame_every < 0) {
        g_err = "HIP_main_scripted: bad argument";
        return -1;
    }
    for (int e = 1; e < n_events; e++)
        if (script[e].step < script[e - 1].step) {
            g_err = "HIP_main_scripted: events must be in ascending step order";
            return -1;
        }
    std::vector<Particle> particles;
    hip_initSimulation(particles, N);
    if (!upload(particles, N)) return -1;
    const bool frames = frame_every > 0 && frame_prefix && *frame_prefix;
    auto frame = [&](int step) {
        if (!download(particles, N)) return false;
        char name[32];
        std::snprintf(name, sizeof name, "%06d.ppm", step);
        if (!write_frame(std::string(frame_prefix) + name, particles, N)) {
            g_err = "HIP_main_scripted: cannot write frame";
            return false;
        }
        return true;
    };
    int done = 0, ev = 0;
    float mx = 0.0f, my = 0.0f, strength = 0.0f;
    while (done < steps) {
        while (ev < n_events && script[ev].step <= done) {  // serial_main.cpp:72-75
            mx = (float)((script[ev].px / 800.0) * 2.0);
            my = (float)(2.0 - (script[ev].py / 800.0) * 2.0);
            strength = script[ev].button == 1 ? 50000.0f : script[ev].button == 2 ? -50000.0f : 0.0f;
            ev++;
        }
        int until = steps;  // run with this mouse state until the next event, frame or the end
        if (ev < n_events && script[ev].step < until) until = script[ev].step;
        if (frames) {
            int next_frame = (done / frame_every + 1) * frame_every;
            if (next_frame < until) until = next_frame;
        }
        int rc = sphhip_step(g_ctx, mx, my, strength, until - done);
        if (rc) {
            fail("sphhip_step", rc);
            return -1;
        }
        done = until;
        if (frames && (done % frame_every == 0 || done == steps) && !frame(done)) return -1;
    }
    if (!download(particles, N)) return -1;
    if (dump_path && *dump_path) {
        FILE* f = std::fopen(dump_path, "wb");
        if (!f) return -1;
        std::fwrite(particles.data(), sizeof(Particle), (size_t)N, f);
        std::fclose(f);
    }
    return 0;
}
