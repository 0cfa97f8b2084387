// density_loop.hip -- microbenchmark: the inner loop of k_density (distance test, poly6 term, in-range bit) over
// candidates staged in LDS, in three codings: scalar fp32 (what pairs.hip had), packed fp32 over an array-of-pairs
// image (x,y of one candidate per instruction), packed fp32 over a structure-of-arrays image (two candidates per
// instruction).  All three produce the same bits.  Prints ns per (lane, candidate) slot at 4 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize density_loop.hip -o density_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int NCAND = 512, REPS = 200;

__global__ __launch_bounds__(256) void k_scalar(const float2* q, float* out, uint32_t* mout, float h2, float mass, float poly6, int n) {
    __shared__ __align__(16) float2 s[NCAND];
    for (int i = threadIdx.x; i < NCAND; i += blockDim.x) s[i] = q[i];
    __syncthreads();
    const float2 pi = q[NCAND + threadIdx.x];
    float dens = 0;
    uint32_t m = 0, acc = 0;
    for (int rep = 0; rep < REPS; rep++) {
        for (int j = 0; j < n; j += 4) {
            float2 c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) c[u] = s[j + u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float rx = pi.x - c[u].x, ry = pi.y - c[u].y;
                float r2 = rx * rx + ry * ry;
                float t = r2 - h2;
                float term = fmaxf(-t, 0.0f);
                dens += mass * (poly6 * term * term * term);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);
            }
            if ((j & 31) == 28) acc ^= m;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = dens;
    mout[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_clamp(const float2* q, float* out, uint32_t* mout, float h2, float mass, float poly6, int n) {
    __shared__ __align__(16) float2 s[NCAND];
    for (int i = threadIdx.x; i < NCAND; i += blockDim.x) s[i] = q[i];
    __syncthreads();
    const float2 pi = q[NCAND + threadIdx.x];
    float dens = 0;
    uint32_t m = 0, acc = 0;
    for (int rep = 0; rep < REPS; rep++) {
        for (int j = 0; j < n; j += 4) {
            float2 c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) c[u] = s[j + u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float rx = pi.x - c[u].x, ry = pi.y - c[u].y;
                float r2 = rx * rx + ry * ry;
                float t = r2 - h2;
                float term = __builtin_amdgcn_fmed3f(h2 - r2, 0.0f, 1.0f);
                dens += mass * (poly6 * term * term * term);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);
            }
            if ((j & 31) == 28) acc ^= m;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = dens;
    mout[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_aos(const float2* q, float* out, uint32_t* mout, float h2, float mass, float poly6, int n) {
    __shared__ __align__(16) float2 s[NCAND];
    for (int i = threadIdx.x; i < NCAND; i += blockDim.x) s[i] = q[i];
    __syncthreads();
    const float2 pi = q[NCAND + threadIdx.x];
    const v2 p = {pi.x, pi.y};
    float dens = 0;
    uint32_t m = 0, acc = 0;
    for (int rep = 0; rep < REPS; rep++) {
        for (int j = 0; j < n; j += 4) {
            v2 c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) c[u] = (v2){s[j + u].x, s[j + u].y};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v2 r = p - c[u];
                v2 sq = r * r;
                float r2 = sq.x + sq.y;
                float t = r2 - h2;
                float term = fmaxf(-t, 0.0f);
                dens += mass * (poly6 * term * term * term);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);
            }
            if ((j & 31) == 28) acc ^= m;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = dens;
    mout[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_soa(const float2* q, float* out, uint32_t* mout, float h2, float mass, float poly6, int n) {
    __shared__ __align__(16) float sx[NCAND];
    __shared__ __align__(16) float sy[NCAND];
    for (int i = threadIdx.x; i < NCAND; i += blockDim.x) {
        sx[i] = q[i].x;
        sy[i] = q[i].y;
    }
    __syncthreads();
    const float2 pi = q[NCAND + threadIdx.x];
    const v2 px = {pi.x, pi.x}, py = {pi.y, pi.y}, hh = {h2, h2}, pp = {poly6, poly6}, mm = {mass, mass};
    float dens = 0;
    uint32_t m = 0, acc = 0;
    for (int rep = 0; rep < REPS; rep++) {
        for (int j = 0; j < n; j += 4) {
            const v4 x4 = *reinterpret_cast<const v4*>(sx + j), y4 = *reinterpret_cast<const v4*>(sy + j);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const v2 cx = u ? (v2){x4.z, x4.w} : (v2){x4.x, x4.y};
                const v2 cy = u ? (v2){y4.z, y4.w} : (v2){y4.x, y4.y};
                v2 rx = px - cx, ry = py - cy;
                v2 r2 = rx * rx + ry * ry;
                v2 t = r2 - hh;
                v2 term = {fmaxf(-t.x, 0.0f), fmaxf(-t.y, 0.0f)};
                v2 w = pp * term;
                w = w * term;
                w = w * term;
                w = mm * w;
                dens += w.x;
                dens += w.y;
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(t.x), 31);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(t.y), 31);
            }
            if ((j & 31) == 28) acc ^= m;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = dens;
    mout[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int blocks = 256 * 4 * 4, threads = 256;
    std::vector<float2> h(NCAND + threads);
    uint32_t r = 12345;
    for (auto& v : h) {
        r = r * 1664525u + 1013904223u;
        v.x = (r >> 8) * (1.0f / 16777216.0f) * 0.05f;
        r = r * 1664525u + 1013904223u;
        v.y = (r >> 8) * (1.0f / 16777216.0f) * 0.05f;
    }
    float2* d_q;
    float* d_out[4];
    uint32_t* d_m[4];
    CHECK(hipMalloc(&d_q, h.size() * sizeof(float2)));
    CHECK(hipMemcpy(d_q, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    for (int k = 0; k < 4; k++) {
        CHECK(hipMalloc(&d_out[k], blocks * threads * 4));
        CHECK(hipMalloc(&d_m[k], blocks * threads * 4));
    }
    const float hh = 0.016f * 0.016f, mass = 0.02f, poly6 = 315.0f / (65.0f * 3.14159265f * powf(0.016f, 9.0f));
    typedef void (*K)(const float2*, float*, uint32_t*, float, float, float, int);
    K kern[4] = {k_scalar, k_aos, k_soa, k_clamp};
    const char* name[4] = {"scalar", "packed, array of pairs", "packed, structure of arrays", "scalar, clamp for max"};
    for (int n : {160, 512}) {
        for (int k = 0; k < 4; k++) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(kern[k], dim3(blocks), dim3(threads), 0, 0, d_q, d_out[k], d_m[k], hh, mass, poly6, n);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern[k], dim3(blocks), dim3(threads), 0, 0, d_q, d_out[k], d_m[k], hh, mass, poly6, n);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double slots = (double)blocks * threads * REPS * n;
            printf("n=%3d %-28s %8.3f ms  %6.2f ps per lane-candidate  (%.2f cycles per wave-candidate per SIMD at 2.1 GHz)\n", n, name[k], ms,
                   ms * 1e9 / slots, ms * 1e-3 * 2.1e9 / (slots / 64 / 1024));
        }
    }
    std::vector<float> o0(blocks * threads), o1(blocks * threads);
    std::vector<uint32_t> m0(blocks * threads), m1(blocks * threads);
    CHECK(hipMemcpy(o0.data(), d_out[0], o0.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(m0.data(), d_m[0], m0.size() * 4, hipMemcpyDeviceToHost));
    for (int k = 1; k < 4; k++) {
        CHECK(hipMemcpy(o1.data(), d_out[k], o1.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(m1.data(), d_m[k], m1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < o0.size(); i++) bad += (memcmp(&o0[i], &o1[i], 4) != 0) + (m0[i] != m1[i]);
        printf("%s vs scalar: %zu differing words (sum %g)\n", name[k], bad, (double)o0[5]);
    }
    return 0;
}
