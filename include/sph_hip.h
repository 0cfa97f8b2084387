// sph_hip.h -- C++ host-side drop-in: the reference's step interface (SPH/sph.h:35-39) and
// dispatcher entry points (SPH/main.h:52-59) for the MI355X path, implemented over the C ABI of
// include/sphhip.h (libsphhip.so) by sph_amd/csrc/host_shim.cpp.
//
// Same shapes, argument meaning and ownership as the reference's functions: free functions on the
// caller-owned `std::vector<Particle>&`, all `void`, index i stable.  Include the reference's
// "sph.h" BEFORE this header when building inside the reference tree; standalone builds get
// layout-compatible `float2` / `Particle` definitions from here.
#pragma once
#ifndef SPH_H  // standalone: interface types of SPH/sph.h:22-33 (layout fixed by the reference)
#include <vector>
struct float2 {
    float x, y;
    float2(float x_ = 0.0f, float y_ = 0.0f) : x(x_), y(y_) {}
};
struct Particle {
    float2 pos;
    float2 vel;
    float density;
    float pressure;
    bool valid;
};
#endif
static_assert(sizeof(Particle) == 28, "Particle must keep the reference's 28-byte layout (SPH/sph.h:27-33)");

// serial_initSimulation (SPH/sph.cpp:132-164) / initSimulation (SPH/sph.cu:166-205): lattice fill of
// [0.05,1.95]^2; resizes the vector.  Host arithmetic, identical bits.
void hip_initSimulation(std::vector<Particle>& particles, int N);
// Neighbour search (no reference counterpart beyond the dead grid/cellStart/cellEnd parameters,
// SPH/sph.cuh:33-34,38).  Outputs are optional (nullptr): cell key per sorted slot, particle index
// per sorted slot, half-open slot range of each of the nx*ny cells.
void hip_buildNeighbors(std::vector<Particle>& particles, int N, std::vector<int>* keys, std::vector<int>* order,
                        std::vector<int>* cellStart, std::vector<int>* cellEnd);
// serial_computeDensityPressure (SPH/sph.cpp:41-55)
void hip_computeDensityPressure(std::vector<Particle>& particles, int N);
// serial_computeForces (SPH/sph.cpp:57-99), snapshot neighbour velocities (SPH/MPI_sph.cpp:144-147)
void hip_computeForces(std::vector<Particle>& particles, int N, float2 mousePos, float interactionStrength);
// serial_integrate (SPH/sph.cpp:101-130)
void hip_integrate(std::vector<Particle>& particles, int N);
// serial_stepSimulation (SPH/sph.cpp:166-171); like stepSimulation of the CUDA flavour it refreshes
// the caller's host array on every call (SPH/sph.cu:235).
void hip_stepSimulation(std::vector<Particle>& particles, int N, float2 mousePos, float interactionStrength);
// nsteps steps with the state resident in HBM, one upload and one download.
void hip_stepSimulationResident(std::vector<Particle>& particles, int N, float2 mousePos, float interactionStrength,
                                int nsteps);
// Device-flavour interface of the reference's CUDA variant (SPH/sph.cuh:37-39), same shapes with `void*` in
// place of `cudaGraphicsResource*`.  The state lives on the GPU between the calls (the reference's hidden
// `d_particles`, SPH/sph.cu:10); the caller's array is an output, refreshed by every call (sph.cu:204,235).
//   hip_initSimulation(Particle*, N, vbo): initSimulation, sph.cu:166-205 -- lattice fill, upload, copy back.
//   hip_stepSimulation(Particle*, N, grid, cellStart, cellEnd, vbo, mousePos, strength): stepSimulation,
//     sph.cu:218-236 -- one step of the resident state (no upload), then the full download.  The three int
//     arrays are dead parameters in the reference (always passed through as nullptr, sph.cu:222-223); here
//     each may be nullptr or receive the neighbour search of the RETURNED state: grid[i] = cell key of
//     particle i (N ints), cellStart/cellEnd = half-open sorted-slot range of every cell (nx*ny ints each).
//   `vbo`: nullptr, or a host `float2[N]` that receives what updateVBO writes (sph.cu:207-216): the position
//     of every valid particle, (-1, -1) for invalid ones.
void hip_initSimulation(Particle* particles, int N, void* vbo);
void hip_stepSimulation(Particle* particles, int N, int* grid, int* cellStart, int* cellEnd, void* vbo, float2 mousePos,
                        float interactionStrength);
// cleanupSimulation (SPH/sph.cu:275-280): releases the cached device context.
void hip_cleanupSimulation();

// Dispatcher family of SPH/main.h:52-59.  HIP_main is headless (rendering stays off the timed
// path): runs `steps` steps of the default scene and optionally dumps the final state (28-byte
// records) to `dump_path`.  HIP_performance_test follows serial_performance_test
// (SPH/serial_main.cpp:120-147): lattice init, 10 runs x 100 steps, mean of 100/elapsed (UPS);
// the per-step prints of the reference are kept out of the timed region.
int HIP_main(int N);
int HIP_main_headless(int N, int steps, const char* dump_path);

// Scripted stand-in for the interactive loop of X_main (SPH/serial_main.cpp:68-105): the mouse state is
// a list of events instead of GLFW callbacks, the frames go to image files instead of a window.
// From step `step` on the cursor is at window pixel (px, py) of the reference's 800x800 window with
// button 0 = none, 1 = left (strength +50000), 2 = right (-50000); mapped to simulation space exactly
// as serial_main.cpp:72-75.  Events must be in ascending step order.
struct HipMouseEvent {
    int step;
    float px, py;
    int button;
};
// Runs `steps` steps of the default scene under the script.  Every `frame_every` steps (0: never), and
// after the last step, the state is fetched (off any timed path; stands in for updateVBO,
// SPH/sph.cu:207-216) and written as a binary PPM "<frame_prefix><step, 6 digits>.ppm": 800x800, one
// pixel per valid particle with finite position (serial_main.cpp:92-99), coloured by pressure as
// SPH/openMP_sph.cpp:218-219.  `dump_path` as in HIP_main_headless.  Returns 0, or -1 on error.
int HIP_main_scripted(int N, int steps, const HipMouseEvent* script, int n_events, int frame_every,
                      const char* frame_prefix, const char* dump_path);
float HIP_performance_test(int N);
// Error text of the MOST RECENT shim call ("" if it succeeded: every call above clears it first); the
// reference's functions have no error channel, so failures are reported here and on stderr, never by
// exit() (contrast SPH/sph.cuh:17-23).
const char* hip_lastError();
