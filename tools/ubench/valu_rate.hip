// valu_rate.hip -- microbenchmark: sustained issue rate of the VALU instructions the SPH pair
// kernels are made of (gfx950).  Prints cycles per wave-instruction per SIMD for several
// occupancies.  Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;

#define KERNEL(name, BODY)                                                              \
    __global__ void name(float* out, float seed) {                                     \
        float a[UNROLL];                                                               \
        float b = seed + threadIdx.x * 1e-3f, c2 = seed * 0.5f + 1.0f;                 \
        for (int u = 0; u < UNROLL; u++) a[u] = seed + u + threadIdx.x;                \
        for (int i = 0; i < ITERS; i++) {                                              \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { BODY; }               \
        }                                                                              \
        float s = 0;                                                                   \
        for (int u = 0; u < UNROLL; u++) s += a[u];                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + b + c2;                       \
    }

KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c2)))
KERNEL(k_add, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[u])))
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[u])))
KERNEL(k_rsq, asm volatile("v_rsq_f32 %0, %0" : "+v"(a[u])))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b)))
KERNEL(k_divfixup, asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c2)))
KERNEL(k_divfmas, asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c2)))
KERNEL(k_divscale, asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c2) : "vcc"))
KERNEL(k_cnd64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "s"(0x5555555555555555ull)))
KERNEL(k_cmpcnd, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : "vcc"))
KERNEL(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[u]), "v"(b) : "vcc"))
KERNEL(k_addu32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b)))

__global__ void k_pkmul(float* out, float seed) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a[UNROLL];
    v2 b = {seed, seed + 1.0f};
    for (int u = 0; u < UNROLL; u++) a[u] = (v2){seed + u, seed + threadIdx.x};
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
    }
    float s = 0;
    for (int u = 0; u < UNROLL; u++) s += a[u].x + a[u].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// dependent chain: one accumulator
__global__ void k_mul_dep(float* out, float seed) {
    float a = seed + threadIdx.x, b = seed;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <typename K>
int run(const char* name, K kern, float* d_out, double clock_ghz) {
    const int waves_per_simd[] = {1, 2, 4, 8};
    printf("%-12s", name);
    for (int w : waves_per_simd) {
        int blocks = 256 * w;  // 256-thread blocks = 4 waves = one wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double instr_per_simd = (double)ITERS * UNROLL * w;  // wave-instructions issued on each SIMD
        double cycles = ms * 1e-3 * clock_ghz * 1e9;
        printf("  w=%d: %5.2f cyc/instr", w, cycles / instr_per_simd);
    }
    printf("\n");
    return 0;
}

int main() {
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    int khz = 0;
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    double ghz = khz * 1e-6;
    printf("clock %.2f GHz (nominal; cycles below assume it)\n", ghz);
    run("v_mul_f32", k_mul, d_out, ghz); run("v_mul dep", k_mul_dep, d_out, ghz);
    run("v_fma_f32", k_fma, d_out, ghz); run("v_add_f32", k_add, d_out, ghz);
    run("v_pk_mul_f32", k_pkmul, d_out, ghz); run("v_rcp_f32", k_rcp, d_out, ghz);
    run("v_sqrt_f32", k_sqrt, d_out, ghz); run("v_rsq_f32", k_rsq, d_out, ghz);
    run("v_cndmask", k_cndmask, d_out, ghz); run("cndmask e64", k_cnd64, d_out, ghz); run("cmp+cndmask", k_cmpcnd, d_out, ghz); run("v_div_scale", k_divscale, d_out, ghz);
    run("v_div_fmas", k_divfmas, d_out, ghz); run("v_div_fixup", k_divfixup, d_out, ghz);
    run("v_cmp_lt_f32", k_cmp, d_out, ghz); run("v_add_u32", k_addu32, d_out, ghz);
    run("v_mul_lo_u32", k_mullo, d_out, ghz);
    return 0;
}
