// sph_main.cpp -- headless, non-interactive stand-in for the reference's menu dispatcher
// (SPH/main.cpp:63-162, which needs MPI, GLFW and std::cin): runs the MI355X paths as further methods of the menu
// and prints the same "<method> Average UPS" line (main.cpp:115,127,139,151).
//   sph_main --method hip    --n 1024 --mode perf                      (case 5: one GPU)
//   sph_main --method hip-mg --gpus 8 --scene dam --n 16777216 --mode perf   (case 6: x-slabs over the node's GPUs, RCCL)
//   sph_main --n 1024 --mode run --steps 200 --dump state.bin
//   sph_main --n 1024 --mode run --steps 400 --mouse-script path.txt --frames 100 --frame-prefix out/frame_
// The reference's own methods (serial, openMP, CUDA, MPI: cases 1-4) stay where they are, in the reference's
// main.cpp; INTEGRATION.md shows the two `case` lines that add these.
// --scene box (default): the reference's lattice in its fixed 1.9 x 1.9 box (serial_initSimulation); --scene dam: the
// dam-break column at spacing H/4 with scaled walls (BASELINE configs 2-5; --n is rounded down to a square).
// Script lines: "<step> <pixel x> <pixel y> <button 0|1|2>" (the reference's 800x800 window, serial_main.cpp:72-75).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include "../../include/sph_hip.h"
#include "../../include/sph_mg.h"

namespace {
// The dam-break scene of sph_amd/scenes.py (same binary32 arithmetic): a cols x cols lattice at spacing H/4 against
// the left wall and the floor of a box 4 W wide and 2 W high.
void dam_scene(int cols, SphParams& p, std::vector<Particle>& a) {
    sphhip_default_params(&p);
    const float sp = p.h / 4.0f;
    const float w = (float)cols * sp;
    p.xmin = 0.05f;
    p.ymin = 0.05f;
    p.xmax = 0.05f + 4.0f * w;
    p.ymax = 0.05f + 2.0f * w;
    a.resize((size_t)cols * cols);
    for (size_t i = 0; i < a.size(); i++) {
        std::memset((void*)&a[i], 0, sizeof(Particle));
        a[i].pos = float2(p.xmin + (float)(i % cols) * sp, p.ymin + (float)(i / cols) * sp);
        a[i].density = p.rest_density;
        a[i].valid = true;
    }
}

// X_performance_test protocol (SPH/serial_main.cpp:120-147): 10 runs x 100 steps, mean of 100 / elapsed.
int mg_perf(const SphParams& p, std::vector<Particle>& a, int gpus, int rebalance_every, float& ups) {
    SphMg* mg = nullptr;
    int rc = sphmg_create(&p, a.data(), (int)a.size(), gpus, nullptr, &mg);
    if (rc) return rc;
    const int num_steps = 100, num_runs = 10;
    double total = 0.0;
    for (int run = 0; run < num_runs && !rc; run++) {
        std::cout << "Run: " << run + 1 << " / " << num_runs << std::endl;  // outside the timed region
        rc = sphmg_synchronize(mg);
        auto t0 = std::chrono::high_resolution_clock::now();
        if (!rc) rc = sphmg_step(mg, 0.0f, 0.0f, 0.0f, num_steps, rebalance_every);
        if (!rc) rc = sphmg_synchronize(mg);
        auto t1 = std::chrono::high_resolution_clock::now();
        total += num_steps / std::chrono::duration<double>(t1 - t0).count();
    }
    if (!rc) {
        std::vector<int> cnt(gpus), b(gpus + 1);
        if (!sphmg_counts(mg, cnt.data()) && !sphmg_bounds(mg, b.data())) {
            std::cout << "particles per GPU:";
            for (int g = 0; g < gpus; g++) std::cout << " " << cnt[g];
            std::cout << "  (rebalances: " << sphmg_rebalance_count(mg) << ")" << std::endl;
        }
    }
    sphmg_destroy(mg);
    ups = (float)(total / num_runs);
    return rc;
}
}  // namespace

int main(int argc, char** argv) {
    int N = 1024;  // default N of the reference, main.cpp:68
    int steps = 1000, gpus = 1, rebalance_every = 50;
    const char* mode = "perf";
    const char* method = "hip";
    const char* scene = "box";
    const char* dump = nullptr;
    const char* script_path = nullptr;
    const char* frame_prefix = nullptr;
    int frame_every = 0;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--n") && i + 1 < argc) N = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--mode") && i + 1 < argc) mode = argv[++i];
        else if (!std::strcmp(argv[i], "--method") && i + 1 < argc) method = argv[++i];
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--scene") && i + 1 < argc) scene = argv[++i];
        else if (!std::strcmp(argv[i], "--rebalance-every") && i + 1 < argc) rebalance_every = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!std::strcmp(argv[i], "--mouse-script") && i + 1 < argc) script_path = argv[++i];
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frame_every = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--frame-prefix") && i + 1 < argc) frame_prefix = argv[++i];
        else {
            std::fprintf(stderr, "usage: sph_main [--method hip|hip-mg] [--gpus G] [--scene box|dam] [--n N] [--mode perf|run] "
                                 "[--steps K] [--rebalance-every K] [--dump file] [--mouse-script file] [--frames K --frame-prefix path]\n");
            return 2;
        }
    }
    const bool is_mg = !std::strcmp(method, "hip-mg");
    if (!is_mg && std::strcmp(method, "hip")) {
        std::fprintf(stderr, "unknown method %s: this dispatcher adds hip and hip-mg; serial, openMP, CUDA and MPI are the "
                             "reference's own menu entries (SPH/main.cpp:106-154)\n", method);
        return 2;
    }
    if (N <= 0 || gpus < 1) {
        std::fprintf(stderr, "N and --gpus must be positive\n");
        return 2;
    }
    const bool dam = !std::strcmp(scene, "dam");
    if (is_mg || dam) {
        // scenes with scaled walls and the multi-GPU method go through the C ABI (parameters are part of the context)
        SphParams p;
        std::vector<Particle> a;
        if (dam) {
            const int cols = (int)std::floor(std::sqrt((double)N));
            dam_scene(cols, p, a);
            N = (int)a.size();
        } else {
            sphhip_default_params(&p);
            hip_initSimulation(a, N);
        }
        if (!std::strcmp(mode, "perf")) {
            float ups = 0.0f;
            int rc = mg_perf(p, a, is_mg ? gpus : 1, rebalance_every, ups);
            if (rc) {
                std::fprintf(stderr, "[sph_main] %s (code %d)\n", sphmg_last_error(), rc);
                return 1;
            }
            const char* label = is_mg ? "HIP-MG" : "HIP";
            std::cout << "\n" << label << " Average UPS: " << ups << std::endl;
            std::cout << label << " particle-updates/s: " << (double)ups * N << "  (N = " << N << ", GPUs = " << (is_mg ? gpus : 1) << ")" << std::endl;
            return 0;
        }
        SphMg* mg = nullptr;
        int rc = sphmg_create(&p, a.data(), N, is_mg ? gpus : 1, nullptr, &mg);
        if (!rc) rc = sphmg_step(mg, 0.0f, 0.0f, 0.0f, steps, rebalance_every);
        if (!rc) rc = sphmg_synchronize(mg);
        if (!rc) rc = sphmg_download(mg, a.data(), N);
        if (mg) sphmg_destroy(mg);
        if (rc) {
            std::fprintf(stderr, "[sph_main] %s (code %d)\n", sphmg_last_error(), rc);
            return 1;
        }
        if (dump && *dump) {
            FILE* f = std::fopen(dump, "wb");
            if (!f) return 1;
            std::fwrite(a.data(), sizeof(Particle), (size_t)N, f);
            std::fclose(f);
        }
        std::cout << (is_mg ? "HIP-MG" : "HIP") << " ran " << steps << " steps of N=" << N << std::endl;
        return 0;
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
