// ref_shim.cpp -- extern "C" doorway into the reference's own serial step.  TEST INFRASTRUCTURE.
//
// Compiled together with the reference source where it lies (/root/reference/SPH/sph.cpp, never
// copied) into oracle/_ref/libsph_ref.so by oracle/Makefile.  It only forwards to the functions
// declared in the reference's SPH/sph.h:35-39 and to the four pair kernels defined with external
// linkage in SPH/sph.cpp:7-39, so that tests can pin oracle/sph_oracle.c against the real thing
// and oracle/make_golden.py can generate tests/golden/.
#include "sph.h"  // the reference's header, found through -I/root/reference/SPH
#include <cstring>

float serial_poly6(float r2, float h);
float2 serial_spikyGrad(float2 r, float r_len, float h);
float serial_viscosityLaplacian(float r, float h);
float2 serial_ljForce(float2 r, float r_len, float sigma, float epsilon);

static_assert(sizeof(Particle) == 28, "reference Particle layout");

static std::vector<Particle> to_vec(const void* aos, int n) {
    std::vector<Particle> v(n);
    if (n) std::memcpy(v.data(), aos, sizeof(Particle) * (size_t)n);
    return v;
}
static void from_vec(const std::vector<Particle>& v, void* aos, int n) {
    if (n) std::memcpy(aos, v.data(), sizeof(Particle) * (size_t)n);
}

extern "C" {
int ref_sizeof_particle() { return (int)sizeof(Particle); }
void ref_init(void* aos, int n) {
    std::vector<Particle> v;
    serial_initSimulation(v, n);
    // zero the padding bytes so byte-wise comparisons are meaningful
    std::vector<Particle> z(n);
    std::memset((void*)z.data(), 0, sizeof(Particle) * (size_t)n);
    for (int i = 0; i < n; i++) {
        z[i].pos = v[i].pos; z[i].vel = v[i].vel; z[i].density = v[i].density;
        z[i].pressure = v[i].pressure; z[i].valid = v[i].valid;
    }
    from_vec(z, aos, n);
}
void ref_density(void* aos, int n) { auto v = to_vec(aos, n); serial_computeDensityPressure(v, n); from_vec(v, aos, n); }
void ref_forces(void* aos, int n, float mx, float my, float strength) {
    auto v = to_vec(aos, n); serial_computeForces(v, n, float2(mx, my), strength); from_vec(v, aos, n);
}
void ref_integrate(void* aos, int n) { auto v = to_vec(aos, n); serial_integrate(v, n); from_vec(v, aos, n); }
void ref_step(void* aos, int n, float mx, float my, float strength, int nsteps) {
    auto v = to_vec(aos, n);
    for (int s = 0; s < nsteps; s++) serial_stepSimulation(v, n, float2(mx, my), strength);
    from_vec(v, aos, n);
}
float ref_poly6(float r2, float h) { return serial_poly6(r2, h); }
void ref_spiky_grad(float rx, float ry, float r_len, float h, float* gx, float* gy) {
    float2 g = serial_spikyGrad(float2(rx, ry), r_len, h); *gx = g.x; *gy = g.y;
}
float ref_visc_lap(float r, float h) { return serial_viscosityLaplacian(r, h); }
void ref_lj(float rx, float ry, float r_len, float sigma, float epsilon, float* fx, float* fy) {
    float2 f = serial_ljForce(float2(rx, ry), r_len, sigma, epsilon); *fx = f.x; *fy = f.y;
}
float ref_const(int which) {
    switch (which) {
        case 0: return H; case 1: return DT; case 2: return MASS; case 3: return VISCOSITY;
        case 4: return STIFFNESS; case 5: return REST_DENSITY; case 6: return DAMPING;
        case 7: return MAX_VEL; case 8: return SIGMA; case 9: return EPSILON;
    }
    return 0.0f;
}
}
