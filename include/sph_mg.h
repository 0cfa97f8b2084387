/*
 * sph_mg.h -- C ABI of the multi-GPU slab driver (libsph_mg.so): x-slabs over the GPUs of one node, ghost-column
 * halo exchange and particle migration with RCCL send/recv over xGMI, one host thread driving all devices
 * (ncclCommInitAll).  Replaces the MPI variant of the reference -- MPI_main / MPI_performance_test and
 * mpi_computeStep's index-block + 4 x MPI_Allgather scheme (SPH/MPI_sph.cpp:135-166,245-450; ranks started by
 * mpiexec, SPH/main.cpp:64-66) -- with the slab hooks of include/sphhip.h.  Plain C, no HIP / RCCL types.
 *
 * Results equal the single-GPU path (sphhip_step) bit for bit whatever the number of GPUs: the sorted order inside
 * every context is canonical (cell key, particle id), ghosts are sorted into the same cells, and every owned
 * particle sums the same neighbours in the same order.
 */
#ifndef SPH_MG_H
#define SPH_MG_H
#include "sphhip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct SphMg SphMg; /* opaque: one SphCtx, a communication stream and four message buffers per GPU, one RCCL communicator each */

/* Distributes the n particles (28-byte records, reference index order = particle id) over n_gpus x-slabs with about
 * equal particle counts, one per HIP device devices[g] (NULL: 0 .. n_gpus-1).  Replaces the MPI_Bcast of the whole
 * particle array to every rank (MPI_sph.cpp:417-418): no GPU holds all particles.  Unlike MPI_sph.cpp:250-254, n need
 * not be divisible by n_gpus.  SPHHIP_EINVAL if the grid has fewer than 3 columns per slab.
 * Transport: RCCL send/recv when every slab has its own device; device-to-device hipMemcpyAsync on the
 * communication streams when slabs share a device (several slabs on one GPU: what a one-GPU box can run of the
 * whole protocol) or when SPHMG_TRANSPORT=copy asks for it.  With RCCL the peer connections are set up here (one
 * header-sized exchange per neighbour pair), so a transport that does not work fails this call.  If a slab fails
 * during an RCCL run the communicators are aborted and every later call but sphmg_destroy returns SPHHIP_ESTATE. */
int sphmg_create(const SphParams* p, const void* aos28, int n, int n_gpus, const int* devices, SphMg** out);
/* nsteps steps of serial_stepSimulation (SPH/sph.cpp:166-171) on all slabs; per step and direction ONE
 * nearest-neighbour message (ghost columns + migrants) instead of mpi_computeStep's four Allgathers
 * (MPI_sph.cpp:144-147,163-166), overlapped with the interior columns' force pass.  rebalance_every > 0: the slab
 * boundaries (and the message size) are re-derived from the per-column histogram every that many steps.
 * Enqueues only (except at a rebalance); sphmg_synchronize / sphmg_download wait. */
int sphmg_step(SphMg* mg, float mouse_x, float mouse_y, float strength, int nsteps, int rebalance_every);
int sphmg_rebalance(SphMg* mg);
/* Waits for all GPUs; reports (SPHHIP_ECAPACITY / SPHHIP_ESTATE) any overflow the asynchronous steps flagged. */
int sphmg_synchronize(SphMg* mg);
/* All particles back in index order (the gather of MPI_sph.cpp:356-359, off the timed path). */
int sphmg_download(SphMg* mg, void* aos28, int n);
/* Particles each GPU owns (n_gpus ints; synchronises) and the slab boundaries (n_gpus + 1 grid columns). */
int sphmg_counts(SphMg* mg, int* per_gpu);
int sphmg_bounds(SphMg* mg, int* bounds);
int sphmg_rebalance_count(SphMg* mg);
/* What the exchanges shipped so far: their number, the average bytes one slab sent per exchange (both directions,
 * headers included: messages travel at a size fixed per slab boundary from the global column histogram -- 1.25 x the
 * population of the halo columns, SPHMG_HALO_MARGIN -- not at their record count) and the largest steady-state
 * message in records.  Any pointer may be NULL. */
int sphmg_halo_stats(SphMg* mg, long long* exchanges, double* bytes_per_exchange_per_slab, int* max_message_records);
/* "rccl", "copy" (slabs sharing a device / SPHMG_TRANSPORT=copy) or "none" (one slab). */
const char* sphmg_transport(SphMg* mg);
int sphmg_destroy(SphMg* mg);
const char* sphmg_last_error(void);
/* Loopback check of the transport on one device: a grouped ncclSend / ncclRecv of `bytes` bytes to the rank itself,
 * compared byte for byte (what a one-GPU box can verify of the RCCL path).  0 on success. */
int sphmg_selftest_transport(int device, int bytes);

#ifdef __cplusplus
}
#endif
#endif /* SPH_MG_H */
