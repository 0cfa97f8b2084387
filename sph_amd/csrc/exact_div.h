// exact_div.h -- correctly rounded fp32 reciprocal / division built on v_rcp_f32 + FMA corrections,
// for operands in a "safe" range (no overflow / underflow of quotient or remainders).
//
// hipcc's generic `a / b` (correctly rounded, any operands) costs ~44 cycles per wave on gfx950:
// 2 v_div_scale, v_rcp, 6 FMAs, v_div_fmas, v_div_fixup.  The pair force needs four divisions per
// in-range pair, two of them by r_len and two by rho_j, so sharing one reciprocal per denominator
// and skipping the range scaling saves ~100 of ~400 cycles per pair.  The sequences below are the
// hardware expansion minus the scaling: Newton step on the reciprocal, then two remainder
// corrections of the quotient (Markstein).  They are NOT assumed correct: sphhip_selftest_division
// (ctx.hip) checks them on the GPU against `1.0f / x` for EVERY float in the safe range and against
// `a / b` for 2^31 random pairs, and tests/test_gpu_parity.py keeps the whole step bit-exact against
// the oracle.  Special denominators (0, inf, NaN) go through v_div_fixup like the generic path.
// Outside the safe range (checked per tile when candidates are staged) the kernel uses plain `/`.
#pragma once
#include <hip/hip_runtime.h>

// safe range for denominators and non-zero numerators: [2^-40, 2^40]
#define SPH_SAFE_LO 9.094947017729282e-13f
#define SPH_SAFE_HI 1.099511627776e12f

__device__ __forceinline__ bool sph_safe_operand(float x) {  // 0 is a safe numerator too
    float a = fabsf(x);
    return (a == 0.0f) || (a >= SPH_SAFE_LO && a <= SPH_SAFE_HI);
}

// refined reciprocal (not yet guaranteed correctly rounded: use sph_rcp_exact for a result)
__device__ __forceinline__ float sph_rcp_refined(float b) {
    float y0 = __builtin_amdgcn_rcpf(b);          // <= 1 ulp
    float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

// a / b given y ~ 1/b (refined): two remainder corrections (Markstein).  Domain: b and a non-zero in the
// safe range, or a == 0.  Anything else (b = 0, inf, NaN ...) yields NaN / garbage, never a trap.
__device__ __forceinline__ float sph_div_core(float a, float b, float y) {
    float q0 = a * y;
    float r0 = __builtin_fmaf(-b, q0, a);
    float q1 = __builtin_fmaf(r0, y, q0);
    float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}

// 1.0f / b given y = sph_rcp_refined(b): ONE correction step is enough for the reciprocal -- checked for every
// float of the safe range by sphhip_selftest_division (671 M values, 0 mismatches; tools/ubench/div_probe.hip).
__device__ __forceinline__ float sph_rcp_core(float b, float y) {
    float r = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(r, y, y);
}

// the same with the hardware's special-case fixup (0, inf, NaN operands behave like `/`)
__device__ __forceinline__ float sph_div_with_rcp(float a, float b, float y) {
    return __builtin_amdgcn_div_fixupf(sph_div_core(a, b, y), b, a);
}

// 1.0f / b
__device__ __forceinline__ float sph_rcp_exact(float b, float y) { return sph_div_with_rcp(1.0f, b, y); }

// sqrt(x) from one v_rsq_f32 and ONE residual correction, g = x*r; g += (x - g*g) * (r/2): correctly rounded for
// every float of [SPH_SQRT_LO, SPH_SAFE_HI] (checked exhaustively by sphhip_selftest_division and
// tools/ubench/div_probe.hip: 839 M values, 0 mismatches -- the Goldschmidt iteration the first version ran in
// front of the correction is not needed), without the denormal-range scaling and compare/select fix-ups of hipcc's
// generic sqrtf.  Outside that range (0, denormals, inf) the result is garbage: callers clamp (k_force) or use
// sph_sqrt_exact.
// Also returns y = sph_rcp_refined(s), the `y` of sph_div_core(., s, y) / sph_rcp_core(s, y); the self-test checks
// 1/s and a/s through it for every x.  (Neither the by-product r/2 nor a Newton step from the v_rsq value can
// replace that v_rcp: with either, 1/s came out wrong for the 100 roots s = 1.11..1b * 2^k of the range -- a tie
// that only a seed rounded upward resolves.)
#define SPH_SQRT_LO 8.67361738e-19f /* 2^-60 */
__device__ __forceinline__ float sph_sqrt_core(float x, float& y) {
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r;
    float h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    y = sph_rcp_refined(g);
    return g;
}

// x = 0 and +inf return x
__device__ __forceinline__ float sph_sqrt_exact(float x) {
    float y;
    float g = sph_sqrt_core(x, y);
    bool special = (x == 0.0f) || (x == __builtin_inff());
    return special ? x : g;
}
