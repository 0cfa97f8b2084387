// div_probe.hip -- exhaustive on-GPU probes of SHORTER correctly-rounded sequences for the univariate functions of
// the pair term (candidates for sph_amd/csrc/exact_div.h).  Each probe counts, over every float of its range, the
// inputs for which the candidate differs from hipcc's correctly rounded result.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off div_probe.hip -o div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float rcp_refined(float b) {
    float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
// current: Markstein x2 on top of the refined reciprocal
__device__ __forceinline__ float div_core(float a, float b, float y) {
    float q0 = a * y;
    float r0 = __builtin_fmaf(-b, q0, a);
    float q1 = __builtin_fmaf(r0, y, q0);
    float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}
// candidates
__device__ __forceinline__ float rcp_short(float b, float y1) {  // one correction of the refined reciprocal
    float r = __builtin_fmaf(-b, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
__device__ __forceinline__ float div_short(float a, float b, float y) {  // one Markstein correction
    float q0 = a * y;
    float r0 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r0, y, q0);
}
__device__ __forceinline__ float sqrt_cur(float x, float& hh) {
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    float d = __builtin_fmaf(-g, g, x);
    hh = h;
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_short(float x) {  // no Goldschmidt step: one residual correction of x * rsq(x)
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_two(float x) {  // two residual corrections with the unrefined half reciprocal
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_hw(float x) {  // v_sqrt_f32 + one residual correction using rcp
    float g = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rcpf(g);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// mode 0: sqrt_short, 1: sqrt_two, 2: sqrt_hw, 3: rcp_short, 4: rcp refined only, 5: sqrt_cur (sanity)
__global__ void k_probe1(int mode, unsigned lo_bits, unsigned hi_bits, unsigned long long* bad, unsigned* example) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nbad = 0;
    for (unsigned long long b = lo_bits + i; b <= hi_bits; b += stride) {
        float x = __uint_as_float((unsigned)b);
        float want, got, hh;
        if (mode <= 2 || mode == 5) {
            want = sqrtf(x);
            got = mode == 0 ? sqrt_short(x) : mode == 1 ? sqrt_two(x) : mode == 2 ? sqrt_hw(x) : sqrt_cur(x, hh);
        } else {
            want = 1.0f / x;
            float y1 = rcp_refined(x);
            got = mode == 3 ? rcp_short(x, y1) : y1;
        }
        if (__float_as_uint(want) != __float_as_uint(got)) {
            nbad++;
            *example = (unsigned)b;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// gradient coefficient (sph.cpp:17-18): ((spiky * term) * term) / r_len, term = h - r_len, for every float r_len of
// [lo, h]; candidates: mode 0 one correction with the REFINED reciprocal, mode 1 one correction with the EXACT
// reciprocal (inv_r, which the pair term needs anyway), mode 2 current (two corrections, refined): sanity
__global__ void k_probe_g(int mode, float h, float spiky, unsigned lo_bits, unsigned hi_bits, unsigned long long* bad,
                          unsigned* example) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nbad = 0;
    for (unsigned long long b = lo_bits + i; b <= hi_bits; b += stride) {
        float r = __uint_as_float((unsigned)b);
        float term = h - r;
        float num = spiky * term * term;
        float want = num / r;
        float y1 = rcp_refined(r);
        float got;
        if (mode == 0) got = div_short(num, r, y1);
        else if (mode == 1) got = div_short(num, r, 1.0f / r);
        else got = div_core(num, r, y1);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            nbad++;
            *example = (unsigned)b;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main() {
    unsigned long long* d_bad;
    unsigned* d_ex;
    CHECK(hipMalloc(&d_bad, 8));
    CHECK(hipMalloc(&d_ex, 4));
    auto run1 = [&](const char* name, int mode, float lo, float hi) {
        hipMemset(d_bad, 0, 8);
        hipMemset(d_ex, 0, 4);
        hipLaunchKernelGGL(k_probe1, dim3(4096), dim3(256), 0, 0, mode, bits(lo), bits(hi), d_bad, d_ex);
        unsigned long long h = 0; unsigned ex = 0;
        hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&ex, d_ex, 4, hipMemcpyDeviceToHost);
        printf("%-44s [%g, %g]: %llu mismatches of %llu (e.g. 0x%08x)\n", name, lo, hi, h, (unsigned long long)(bits(hi) - bits(lo)) + 1, ex);
    };
    const float two_m60 = 8.67361738e-19f, two_40 = 1.099511627776e12f, two_m40 = 9.094947017729282e-13f;
    run1("sqrt current (sanity)", 5, two_m60, two_40);
    run1("sqrt_short  rsq + residual", 0, two_m60, two_40);
    run1("sqrt_two    rsq + 2 residuals", 1, two_m60, two_40);
    run1("sqrt_hw     v_sqrt + rcp residual", 2, two_m60, two_40);
    run1("rcp refined only", 4, two_m40, two_40);
    run1("rcp_short   refined + 1 correction", 3, two_m40, two_40);
    const float h = 0.15f;
    const float spiky = -45.0f / (3.14159f * powf(h, 6));
    auto rung = [&](const char* name, int mode, float lo) {
        hipMemset(d_bad, 0, 8);
        hipMemset(d_ex, 0, 4);
        hipLaunchKernelGGL(k_probe_g, dim3(4096), dim3(256), 0, 0, mode, h, spiky, bits(lo), bits(h), d_bad, d_ex);
        unsigned long long hb = 0; unsigned ex = 0;
        hipMemcpy(&hb, d_bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&ex, d_ex, 4, hipMemcpyDeviceToHost);
        printf("%-44s r in [%g, h]: %llu mismatches of %llu (e.g. 0x%08x)\n", name, lo, hb, (unsigned long long)(bits(h) - bits(lo)) + 1, ex);
    };
    rung("gcoeff current (sanity)", 2, 9.3e-10f);
    rung("gcoeff 1 correction, refined rcp", 0, 9.3e-10f);
    rung("gcoeff 1 correction, exact rcp", 1, 9.3e-10f);
    return 0;
}
