// lds_gather.hip -- microbenchmark: cost of the force kernel's LDS access pattern on gfx950.  Every lane reads records at
// lane-varying positions inside a window of W records (as the mask pops of k_force do), in several layouts:
//   aos32   one 32-byte record = two ds_read_b128 at +0 / +16 (what pairs.hip had)
//   soa16   two arrays of 16-byte half records
//   soa8    four arrays of 8-byte quarter records (ds_read_b64 x 4)
//   private every lane its own granule (no conflicts: the floor)
// Prints LDS cycles per wave and record (one wave per SIMD, nothing else running: pure LDS latency+throughput) and with
// 4 workgroups per CU (throughput).  Build: hipcc --offload-arch=gfx950 -O3 lds_gather.hip -o lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int CAP = 960, ITERS = 4000, NIDX = 64;

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint16_t* idx, float* out) {
    __shared__ __align__(16) float4 s[CAP * 2];
    for (int i = threadIdx.x; i < CAP * 2; i += 256) s[i] = make_float4(i, 1.0f, 2.0f, 3.0f);
    uint16_t my[NIDX];
    for (int k2 = 0; k2 < NIDX; k2++) my[k2] = idx[(size_t)k2 * 256 + threadIdx.x];
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    const char* base = reinterpret_cast<const char*>(s);
    for (int it = 0; it < ITERS / NIDX; it++) {
#pragma unroll
        for (int k2 = 0; k2 < NIDX; k2++) {
            const int r = my[k2] + (it & 15);  // (depends on the trip: the reads cannot be hoisted)
            float4 a, b;
            if (MODE == 0) {  // aos32
                const float4* p = reinterpret_cast<const float4*>(base + r * 32);
                a = p[0];
                b = p[1];
            } else if (MODE == 1) {  // soa16
                const float4* p = reinterpret_cast<const float4*>(base + r * 16);
                a = p[0];
                b = p[CAP];
            } else if (MODE == 2) {  // soa8
                const float2* p = reinterpret_cast<const float2*>(base + r * 8);
                const float2 q0 = p[0], q1 = p[CAP], q2 = p[2 * CAP], q3 = p[3 * CAP];
                a = make_float4(q0.x, q0.y, q1.x, q1.y);
                b = make_float4(q2.x, q2.y, q3.x, q3.y);
            } else if (MODE == 4) {  // soa16, 12 + 12 bytes used (the zero-pressure path of k_force: vx vy rho | x y yd)
                const float* p = reinterpret_cast<const float*>(base + r * 16);
                a = make_float4(p[0], p[1], p[2], 0.0f);
                b = make_float4(p[4 * CAP], p[4 * CAP + 1], p[4 * CAP + 2], 0.0f);
            } else if (MODE == 5) {  // 16-byte hot half (x y yd rho) + an 8-byte array (vx vy)
                const float4* p = reinterpret_cast<const float4*>(base + r * 16);
                const float2* q = reinterpret_cast<const float2*>(base + CAP * 16 + r * 8);
                a = p[0];
                const float2 v = q[0];
                b = make_float4(v.x, v.y, 0.0f, 0.0f);
            } else if (MODE == 6) {  // 16-byte hot half + the first 8 bytes of the other 16-byte half
                const float4* p = reinterpret_cast<const float4*>(base + r * 16);
                const float2* q = reinterpret_cast<const float2*>(base + CAP * 16 + r * 16);
                a = p[0];
                const float2 v = q[0];
                b = make_float4(v.x, v.y, 0.0f, 0.0f);
            } else {  // private
                const float4* p = reinterpret_cast<const float4*>(base + ((threadIdx.x & 63) << 4) + ((r & 1) << 11));
                a = p[0];
                b = p[64];
            }
            acc.x += a.x + b.x;
            acc.y += a.y + b.y;
            acc.z += a.z + b.z;
            acc.w += a.w + b.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    // index pattern: lane l of a wave belongs to own particle l; its k-th record is a random member of a window of W records
    // that drifts with k (neighbouring lanes share the window, as cell-mates share their candidate cells)
    for (int W : {16, 48, 128, 400}) {
        std::vector<uint16_t> h((size_t)NIDX * 256);
        uint32_t r = 777;
        for (int k2 = 0; k2 < NIDX; k2++)
            for (int t = 0; t < 256; t++) {
                r = r * 1664525u + 1013904223u;
                const int win0 = ((t >> 4) * 37 + k2 * 3) % (CAP - W - 16);  // 16 lanes (one cell) share a window
                h[(size_t)k2 * 256 + t] = (uint16_t)(win0 + (r >> 8) % W);
            }
        uint16_t* d_idx;
        float* d_out;
        CHECK(hipMalloc(&d_idx, h.size() * 2));
        CHECK(hipMemcpy(d_idx, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&d_out, 4096 * 256 * 4));
        typedef void (*K)(const uint16_t*, float*);
        K kern[7] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
        const char* name[7] = {"aos32", "soa16", "soa8", "private", "soa16_12", "hot16+8", "hot16+8of16"};
        for (int blocks : {256, 1024}) {
            for (int m = 0; m < 7; m++) {
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0));
                CHECK(hipEventCreate(&e1));
                hipLaunchKernelGGL(kern[m], dim3(blocks), dim3(256), 0, 0, d_idx, d_out);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern[m], dim3(blocks), dim3(256), 0, 0, d_idx, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                // per CU: blocks/256 workgroups x 4 waves x ITERS records
                const double recs_per_cu = (double)blocks / 256 * 4 * (ITERS / NIDX * NIDX);
                printf("W=%3d  %d WG/CU  %-8s %7.3f ms  %6.1f cycles per wave-record per CU (2.1 GHz)\n", W, blocks / 256, name[m], ms,
                       ms * 1e-3 * 2.1e9 / recs_per_cu);
            }
        }
        CHECK(hipFree(d_idx));
        CHECK(hipFree(d_out));
    }
    return 0;
}
