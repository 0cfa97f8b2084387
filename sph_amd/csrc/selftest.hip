// selftest.hip -- on-GPU validation of exact_div.h against hipcc's correctly rounded `/`.
#include <cstring>
#include "sphhip_internal.h"
#include "exact_div.h"

__device__ __forceinline__ unsigned long long splitmix(unsigned long long x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// every float in [lo_bits, hi_bits], both signs: 1.0f / x
__global__ void k_check_rcp(unsigned lo_bits, unsigned hi_bits, unsigned long long* bad) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nbad = 0;
    for (unsigned long long b = lo_bits + i; b <= hi_bits; b += stride) {
        for (int sgn = 0; sgn < 2; sgn++) {
            float x = __uint_as_float((unsigned)b | (sgn ? 0x80000000u : 0u));
            float want = 1.0f / x;
            float got = sph_div_core(1.0f, x, sph_rcp_refined(x));  // as k_force uses it: no special-case fixup
            float got1 = sph_rcp_core(x, sph_rcp_refined(x));       // ... and the one-correction form it uses for 1 / r_len
            nbad += (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(want) != __float_as_uint(got1)) ? 1 : 0;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// a mantissa/exponent pair drawn log-uniformly from the safe range (exponent -40..39)
__device__ __forceinline__ float rnd_safe(unsigned long long r, int emin, int emax) {
    unsigned mant = (unsigned)(r & 0x7fffffu);
    int e = emin + (int)((r >> 23) % (unsigned long long)(emax - emin + 1));
    unsigned sign = (unsigned)((r >> 63) & 1u) << 31;
    return __uint_as_float(sign | ((unsigned)(e + 127) << 23) | mant);
}

// mode 0: wide range; mode 1: realistic (rho in [2^4, 2^17], numerators in [2^-30, 2^24]);
// mode 2: (a / (2b)) == 0.5 * (a / b); mode 3: special denominators / numerators
__global__ void k_check_div(unsigned long long n, int mode, unsigned long long seed, unsigned long long* bad) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nbad = 0;
    for (; i < n; i += stride) {
        unsigned long long r1 = splitmix(seed + 2 * i), r2 = splitmix(seed + 2 * i + 1);
        float a, b;
        if (mode == 0 || mode == 2) {
            a = rnd_safe(r1, -40, 39);
            b = rnd_safe(r2, -40, 39);
        } else if (mode == 1) {
            a = rnd_safe(r1, -30, 24);
            b = fabsf(rnd_safe(r2, 4, 17));
        } else {
            const float sp[8] = {0.0f, -0.0f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1.0f, 1e-6f, 0.15f};
            a = (r1 & 8) ? rnd_safe(r1 >> 4, -40, 39) : sp[r1 & 7];
            b = sp[r2 & 7];
            if ((r2 & 7) >= 5) a = 0.0f;  // zero numerators over ordinary denominators
        }
        float y = sph_rcp_refined(b);
        float want, got;
        if (mode == 2) {
            want = a / (2.0f * b);
            got = 0.5f * sph_div_core(a, b, y);
        } else {
            want = a / b;
            got = mode == 3 ? sph_div_with_rcp(a, b, y) : sph_div_core(a, b, y);
        }
        bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want != want && got != got);
        nbad += same ? 0 : 1;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// every float in [lo_bits, hi_bits]: sqrt, and 1/sqrt(x), a/sqrt(x) through the fused reciprocal
__global__ void k_check_sqrt(unsigned lo_bits, unsigned hi_bits, unsigned long long* bad) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nbad = 0;
    for (unsigned long long b = lo_bits + i; b <= hi_bits; b += stride) {
        float x = __uint_as_float((unsigned)b);
        float y;
        float s = x == 0.0f ? sph_sqrt_exact(x) : sph_sqrt_core(x, y);  // y: the reciprocal k_force divides with
        if (x == 0.0f) y = sph_rcp_refined(s);
        float want = sqrtf(x);
        bool ok = __float_as_uint(want) == __float_as_uint(s);
        float a = rnd_safe(splitmix(b), -30, 24);
        if (x != 0.0f) {
            ok = ok && (__float_as_uint(1.0f / want) == __float_as_uint(sph_div_core(1.0f, s, y)));
            ok = ok && (__float_as_uint(1.0f / want) == __float_as_uint(sph_rcp_core(s, y)));
            ok = ok && (__float_as_uint(a / want) == __float_as_uint(sph_div_core(a, s, y)));
        }
        nbad += ok ? 0 : 1;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int sph_selftest_division(int device, unsigned long long* checked, unsigned long long* mismatches) {
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d_bad;
    HIP_TRY(hipMalloc((void**)&d_bad, sizeof(unsigned long long) * 8));
    HIP_TRY(hipMemset(d_bad, 0, sizeof(unsigned long long) * 8));
    unsigned lo, hi;
    float flo = SPH_SAFE_LO, fhi = SPH_SAFE_HI;
    std::memcpy(&lo, &flo, 4);
    std::memcpy(&hi, &fhi, 4);
    const unsigned long long npairs = 1ull << 29;
    hipLaunchKernelGGL(k_check_rcp, dim3(4096), dim3(256), 0, 0, lo, hi, d_bad + 0);
    hipLaunchKernelGGL(k_check_div, dim3(4096), dim3(256), 0, 0, npairs * 2, 0, 0x1234ull, d_bad + 1);
    hipLaunchKernelGGL(k_check_div, dim3(4096), dim3(256), 0, 0, npairs, 1, 0x5678ull, d_bad + 2);
    hipLaunchKernelGGL(k_check_div, dim3(4096), dim3(256), 0, 0, npairs, 2, 0x9abcull, d_bad + 3);
    hipLaunchKernelGGL(k_check_div, dim3(256), dim3(256), 0, 0, 1ull << 20, 3, 0xdef0ull, d_bad + 4);
    unsigned slo, shi;
    float fslo = SPH_SQRT_LO;
    std::memcpy(&slo, &fslo, 4);
    shi = hi;
    hipLaunchKernelGGL(k_check_sqrt, dim3(4096), dim3(256), 0, 0, slo, shi, d_bad + 5);
    hipLaunchKernelGGL(k_check_sqrt, dim3(1), dim3(64), 0, 0, 0u, 0u, d_bad + 6);  // x = 0
    LAUNCH_CHECK();
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, d_bad, sizeof h, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d_bad));
    checked[0] = 2ull * (hi - lo + 1);
    checked[1] = npairs * 2;
    checked[2] = npairs;
    checked[3] = npairs;
    checked[4] = 1ull << 20;
    checked[5] = (unsigned long long)(shi - slo + 1);
    checked[6] = 1;
    for (int i = 0; i < 7; i++) mismatches[i] = h[i];
    return SPHHIP_OK;
}
