/* sph_cpu_main.c -- TEST INFRASTRUCTURE (like everything under oracle/): the CPU entries of the reference's menu
 * (SPH/main.cpp:106-141: case 1 serial, case 2 OpenMP) as a headless program over the oracle's restatements, with the
 * reference's own performance protocol (serial_performance_test, SPH/serial_main.cpp:120-147; openMP_performance_test,
 * SPH/openMP_sph.cpp:320-345): lattice init, 10 runs of 100 steps, mean of 100 / elapsed, printed as the menu prints it
 * ("Serial Average UPS: ...", main.cpp:115,127).  The product's dispatcher (sph_amd/sph_main) refuses these methods on
 * purpose -- it must not link the oracle; this binary is what bench.py's cpu_baseline leg and the tests can run instead.
 *   oracle/sph_cpu_main --method serial|omp [--n 1024] [--mode perf|run] [--runs 10] [--steps 100] [--dump file]
 * Unlike the reference, nothing is printed inside the timed region (it prints every step, serial_main.cpp:137). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/sphhip.h"

typedef struct {
    float px, py, vx, vy, density, pressure;
    unsigned char valid;
    unsigned char pad[3];
} orc_particle;

void orc_default_params(SphParams* p);
void orc_init_lattice(orc_particle* a, int n, float min_pos, float max_pos, float rest_density, int sqrt_f32);
void orc_step_allpairs(const SphParams* p, orc_particle* a, int n, float mx, float my, float strength, int flags, int nsteps);
void orc_omp_step_allpairs(const SphParams* p, orc_particle* a, int n, int nsteps);
int orc_num_threads(void);
unsigned long long orc_checksum(const orc_particle* a, int n);

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char** argv) {
    const char *method = "serial", *mode = "perf", *dump = NULL;
    int n = 1024, runs = 10, steps = 100; /* N = 1024: the menu's default, main.cpp:68 */
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--method") && i + 1 < argc) method = argv[++i];
        else if (!strcmp(argv[i], "--mode") && i + 1 < argc) mode = argv[++i];
        else if (!strcmp(argv[i], "--n") && i + 1 < argc) n = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--runs") && i + 1 < argc) runs = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else {
            fprintf(stderr, "usage: sph_cpu_main --method serial|omp [--n N] [--mode perf|run] [--runs R] [--steps S] [--dump file]\n");
            return 2;
        }
    }
    const int omp = !strcmp(method, "omp") || !strcmp(method, "openmp");
    if (!omp && strcmp(method, "serial")) {
        fprintf(stderr, "sph_cpu_main: method must be serial or omp (the GPU methods are sph_amd/sph_main's)\n");
        return 2;
    }
    if (n < 1 || runs < 1 || steps < 0) {
        fprintf(stderr, "sph_cpu_main: bad --n / --runs / --steps\n");
        return 2;
    }
    SphParams p;
    orc_default_params(&p);
    orc_particle* a = (orc_particle*)calloc((size_t)n, sizeof(orc_particle));
    if (!a) return 1;
    orc_init_lattice(a, n, p.xmin, p.xmax, p.rest_density, 0); /* serial_initSimulation, sph.cpp:132-164 */
    const char* label = omp ? "OpenMP" : "Serial";
    if (!strcmp(mode, "perf")) {
        double total = 0.0;
        for (int run = 0; run < runs; run++) {
            printf("Run: %d / %d\n", run + 1, runs);
            fflush(stdout);
            const double t0 = now_s();
            if (omp) orc_omp_step_allpairs(&p, a, n, steps);
            else orc_step_allpairs(&p, a, n, 0.0f, 0.0f, 0.0f, 0 /* ORC_GAUSS_SEIDEL: sph.cpp's in-place update */, steps);
            total += (double)steps / (now_s() - t0);
        }
        printf("%s Average UPS: %g\n", label, total / runs);
        printf("(N = %d, %d runs x %d steps, %d thread%s, particle-updates/s %.4g)\n", n, runs, steps, omp ? orc_num_threads() : 1,
               (omp ? orc_num_threads() : 1) == 1 ? "" : "s", total / runs * (double)n);
    } else {
        if (omp) orc_omp_step_allpairs(&p, a, n, steps);
        else orc_step_allpairs(&p, a, n, 0.0f, 0.0f, 0.0f, 0, steps);
        printf("%s: %d steps of N = %d, checksum %016llx\n", label, steps, n, orc_checksum(a, n));
    }
    if (dump) {
        FILE* f = fopen(dump, "wb");
        if (!f || fwrite(a, sizeof(orc_particle), (size_t)n, f) != (size_t)n) {
            fprintf(stderr, "sph_cpu_main: cannot write %s\n", dump);
            return 1;
        }
        fclose(f);
    }
    free(a);
    return 0;
}
