// sph_main.cpp -- headless, non-interactive stand-in for the reference's menu dispatcher
// (SPH/main.cpp:63-162, which needs MPI, GLFW and std::cin): runs the MI355X path as the fifth
// method and prints the same "<method> Average UPS" line (main.cpp:115,127,139,151).
//   sph_main --n 1024 --mode perf
//   sph_main --n 1024 --mode run --steps 200 --dump state.bin
//   sph_main --n 1024 --mode run --steps 400 --mouse-script path.txt --frames 100 --frame-prefix out/frame_
// Script lines: "<step> <pixel x> <pixel y> <button 0|1|2>" (the reference's 800x800 window, serial_main.cpp:72-75).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include "../../include/sph_hip.h"

int main(int argc, char** argv) {
    int N = 1024;  // default N of the reference, main.cpp:68
    int steps = 1000;
    const char* mode = "perf";
    const char* dump = nullptr;
    const char* script_path = nullptr;
    const char* frame_prefix = nullptr;
    int frame_every = 0;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--n") && i + 1 < argc) N = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--mode") && i + 1 < argc) mode = argv[++i];
        else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!std::strcmp(argv[i], "--mouse-script") && i + 1 < argc) script_path = argv[++i];
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frame_every = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--frame-prefix") && i + 1 < argc) frame_prefix = argv[++i];
        else {
            std::fprintf(stderr, "usage: sph_main [--n N] [--mode perf|run] [--steps K] [--dump file] [--mouse-script file] "
                                 "[--frames K --frame-prefix path]\n");
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
    int rc;
    if (script_path || frame_every > 0) {
        std::vector<HipMouseEvent> script;
        if (script_path) {
            FILE* f = std::fopen(script_path, "r");
            if (!f) {
                std::fprintf(stderr, "cannot open %s\n", script_path);
                return 2;
            }
            HipMouseEvent e;
            while (std::fscanf(f, "%d %f %f %d", &e.step, &e.px, &e.py, &e.button) == 4) script.push_back(e);
            std::fclose(f);
        }
        rc = HIP_main_scripted(N, steps, script.data(), (int)script.size(), frame_every, frame_prefix, dump);
    } else {
        rc = HIP_main_headless(N, steps, dump);
    }
    if (rc || *hip_lastError()) return 1;
    std::cout << "HIP ran " << steps << " steps of N=" << N << std::endl;
    return 0;
}
