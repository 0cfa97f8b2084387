// shim_driver.cpp -- test helper: applies the C++ drop-in functions of include/sph_hip.h to a raw
// 28-byte-record file so that pytest can compare the result with the oracle / golden vectors.
//   shim_driver <op> <in.bin|-> <out.bin> N [nsteps mx my strength]
//   op: init | density | forces | integrate | step | resident | cells | device
//   device: the pointer interface (initSimulation / stepSimulation shapes of SPH/sph.cuh:37-39): init, nsteps
//           steps; writes the particles, then N ints grid, nc, cellStart, cellEnd, then the N float2 of the vbo
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../include/sph_hip.h"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    std::string op = argv[1];
    int N = std::atoi(argv[4]);
    int nsteps = argc > 5 ? std::atoi(argv[5]) : 1;
    float mx = argc > 6 ? (float)std::atof(argv[6]) : 0.0f;
    float my = argc > 7 ? (float)std::atof(argv[7]) : 0.0f;
    float st = argc > 8 ? (float)std::atof(argv[8]) : 0.0f;
    std::vector<Particle> p(N);
    if (op != "init" && op != "device") {
        FILE* f = std::fopen(argv[2], "rb");
        if (!f) return 3;
        if (std::fread((void*)p.data(), sizeof(Particle), (size_t)N, f) != (size_t)N) return 3;
        std::fclose(f);
    }
    std::vector<int> keys, order, cs, ce;
    if (op == "init") hip_initSimulation(p, N);
    else if (op == "density") hip_computeDensityPressure(p, N);
    else if (op == "forces") hip_computeForces(p, N, float2(mx, my), st);
    else if (op == "integrate") hip_integrate(p, N);
    else if (op == "step") for (int s = 0; s < nsteps; s++) hip_stepSimulation(p, N, float2(mx, my), st);
    else if (op == "resident") hip_stepSimulationResident(p, N, float2(mx, my), st, nsteps);
    else if (op == "cells") hip_buildNeighbors(p, N, &keys, &order, &cs, &ce);
    else if (op == "device") {
        std::vector<float2> vbo(N);
        std::vector<int> grid(N);
        cs.assign(1 << 16, -1);
        ce.assign(1 << 16, -1);
        hip_initSimulation(p.data(), N, vbo.data());
        for (int s = 0; s < nsteps && !*hip_lastError(); s++)
            hip_stepSimulation(p.data(), N, s == nsteps - 1 ? grid.data() : nullptr, s == nsteps - 1 ? cs.data() : nullptr,
                               s == nsteps - 1 ? ce.data() : nullptr, vbo.data(), float2(mx, my), st);
        if (*hip_lastError()) {
            std::fprintf(stderr, "%s\n", hip_lastError());
            return 1;
        }
        FILE* f = std::fopen(argv[3], "wb");
        if (!f) return 3;
        int nc = 0;
        while (nc < (int)cs.size() && cs[nc] >= 0) nc++;
        std::fwrite((void*)p.data(), sizeof(Particle), (size_t)N, f);
        std::fwrite(grid.data(), 4, (size_t)N, f);
        std::fwrite(&nc, 4, 1, f);
        std::fwrite(cs.data(), 4, (size_t)nc, f);
        std::fwrite(ce.data(), 4, (size_t)nc, f);
        std::fwrite((void*)vbo.data(), sizeof(float2), (size_t)N, f);
        std::fclose(f);
        hip_cleanupSimulation();
        return 0;
    }
    else return 2;
    if (*hip_lastError()) {
        std::fprintf(stderr, "%s\n", hip_lastError());
        return 1;
    }
    FILE* f = std::fopen(argv[3], "wb");
    if (!f) return 3;
    if (op == "cells") {
        int nc = (int)cs.size();
        std::fwrite(&nc, 4, 1, f);
        std::fwrite(keys.data(), 4, keys.size(), f);
        std::fwrite(order.data(), 4, order.size(), f);
        std::fwrite(cs.data(), 4, cs.size(), f);
        std::fwrite(ce.data(), 4, ce.size(), f);
    } else {
        std::fwrite((void*)p.data(), sizeof(Particle), (size_t)N, f);
    }
    std::fclose(f);
    hip_cleanupSimulation();
    return 0;
}
