// valu_issue.hip -- in-kernel (s_memtime) issue cost of the VALU instructions the SPH pair kernels are made of,
// gfx950.  Every wave times its own loop; no launch overhead and no clock assumption in the result (cycles are
// shader cycles).  Prints cycles per wave-instruction per SIMD at 1, 2, 4 and 8 waves per SIMD, for independent
// streams and for dependent chains of length 1, 2 and 4.
// Build: hipcc --offload-arch=gfx950 -O2 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4000;

// CHAINS independent accumulators, 16 instructions per iteration
#define KERNEL(name, CHAINS, INSTR)                                                              \
    __global__ void name(unsigned long long* cyc, float* out, float seed) {                     \
        float a[CHAINS];                                                                        \
        float b = seed + threadIdx.x * 1e-3f, c2 = seed * 0.5f + 1.0f;                          \
        for (int u = 0; u < CHAINS; u++) a[u] = seed + u + threadIdx.x;                         \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int i = 0; i < ITERS; i++) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 16; u++) {                                    \
                asm volatile(INSTR : "+v"(a[u % CHAINS]) : "v"(b), "v"(c2));                    \
            }                                                                                   \
        }                                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        float s = 0;                                                                            \
        for (int u = 0; u < CHAINS; u++) s += a[u];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

KERNEL(k_fma16, 16, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma4, 4, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma2, 2, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma1, 1, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul16, 16, "v_mul_f32 %0, %0, %1")
KERNEL(k_add16, 16, "v_add_f32 %0, %0, %1")
KERNEL(k_rcp16, 16, "v_rcp_f32 %0, %0")
KERNEL(k_rsq16, 16, "v_rsq_f32 %0, %0")
KERNEL(k_max16, 16, "v_max_f32 %0, %0, %1")
KERNEL(k_alignbit16, 16, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(k_cmpcnd16, 16, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_ffbh16, 16, "v_ffbh_u32 %0, %0")
KERNEL(k_lshl_add16, 16, "v_lshl_add_u32 %0, %0, 3, %1")

template <typename K>
int run(const char* name, K kern, unsigned long long* d_cyc, float* d_out, int per_iter = 16) {
    printf("%-14s", name);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * w;  // 256-thread blocks = 4 waves, one per SIMD
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_cyc, d_out, 1.0f);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * 4);
        CHECK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        // a wave's loop took `med` cycles while w waves shared its SIMD: cycles per wave-instruction of the SIMD
        printf("  w=%d: %5.2f", w, med / ((double)ITERS * per_iter * w));
    }
    printf("\n");
    return 0;
}

int main() {
    unsigned long long* d_cyc;
    float* d_out;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 256 * 8 * 4));
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    printf("cycles per wave-instruction per SIMD (s_memtime; median wave)\n");
    run("fma indep16", k_fma16, d_cyc, d_out);
    run("fma chains=4", k_fma4, d_cyc, d_out);
    run("fma chains=2", k_fma2, d_cyc, d_out);
    run("fma chains=1", k_fma1, d_cyc, d_out);
    run("mul indep16", k_mul16, d_cyc, d_out);
    run("add indep16", k_add16, d_cyc, d_out);
    run("max indep16", k_max16, d_cyc, d_out);
    run("rcp indep16", k_rcp16, d_cyc, d_out);
    run("rsq indep16", k_rsq16, d_cyc, d_out);
    run("alignbit", k_alignbit16, d_cyc, d_out);
    run("cmp+cndmask", k_cmpcnd16, d_cyc, d_out, 32);
    run("ffbh", k_ffbh16, d_cyc, d_out);
    run("lshl_add", k_lshl_add16, d_cyc, d_out);
    return 0;
}
