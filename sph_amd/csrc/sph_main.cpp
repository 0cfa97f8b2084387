// sph_main.cpp -- headless, non-interactive stand-in for the reference's menu dispatcher
// (SPH/main.cpp:63-162, which needs MPI, GLFW and std::cin): runs the MI355X path as the fifth
// method and prints the same "<method> Average UPS" line (main.cpp:115,127,139,151).
//   sph_main --n 1024 --mode perf
//   sph_main --n 1024 --mode run --steps 200 --dump state.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "../../include/sph_hip.h"

int main(int argc, char** argv) {
    int N = 1024;  // default N of the reference, main.cpp:68
    int steps = 1000;
    const char* mode = "perf";
    const char* dump = nullptr;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--n") && i + 1 < argc) N = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--mode") && i + 1 < argc) mode = argv[++i];
        else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else {
            std::fprintf(stderr, "usage: sph_main [--n N] [--mode perf|run] [--steps K] [--dump file]\n");
            return 2;
        }
    }
    if (N <= 0) {
        std::fprintf(stderr, "N must be positive\n");
        return 2;
    }
    if (!std::strcmp(mode, "perf")) {
        float ups = HIP_performance_test(N);
        if (*hip_lastError()) return 1;
        std::cout << "\nHIP Average UPS: " << ups << std::endl;
        std::cout << "HIP particle-updates/s: " << (double)ups * N << std::endl;
        return 0;
    }
    int rc = HIP_main_headless(N, steps, dump);
    if (rc || *hip_lastError()) return 1;
    std::cout << "HIP ran " << steps << " steps of N=" << N << std::endl;
    return 0;
}
