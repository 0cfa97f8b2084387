// slab_driver.cpp -- libsph_mg.so: the multi-GPU slab driver of include/sph_mg.h.  C++ host code over the C ABI of
// libsphhip.so (slab hooks) and RCCL point-to-point (ncclSend / ncclRecv in one group per step: every slab talks
// to its two x-neighbours only, each pair over its own xGMI link).  The same protocol as sph_amd/slab.py (one
// process per GPU over torch.distributed), here with one host thread driving all devices.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sph_mg.h"

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_OK(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return fail(SPHHIP_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)
#define NCCL_OK(call)                                                                                         \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) return fail(SPHHIP_EHIP, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define SPH_OK(call)                                                                     \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_) return fail(rc_, std::string(#call) + ": " + sphhip_last_error());      \
    } while (0)

constexpr int HALO = SPHHIP_HALO_COLS;
constexpr int MIN_SLAB_COLS = HALO + 1;  // a slab wider than the halo: a halo never spans more than one neighbour
constexpr int BIG = 0x7fffffff;

struct HaloRec {
    float x, y, vx, vy;
    uint32_t idv;
};
static_assert(sizeof(HaloRec) == 20, "halo record");

struct Slab {
    int device = 0;
    SphCtx* ctx = nullptr;
    // compute = the context's own stream (sphhip_get_stream; not created here: one stream fewer per slab on the
    // hardware queues of the process); comm_stream carries the transfers.  Both have the default priority -- the
    // stream invariant of DESIGN.md section 6: no stream that another one waits for has a lower priority than the waiter.
    hipStream_t compute = nullptr, comm_stream = nullptr;
    hipEvent_t packed = nullptr, received = nullptr;
    ncclComm_t comm = nullptr;
    void *send_l = nullptr, *send_r = nullptr, *recv_l = nullptr, *recv_r = nullptr;
    int c0 = 0, c1 = 0;
};
}  // namespace

struct SphMg {
    SphParams params;
    SphDerived drv;
    int n = 0, world = 0, rebalances = 0;
    // Message sizes in records, per slab boundary g = 1 .. world-1 (entries 0 and world unused): `cap` between two
    // rebalances, `cap_first` for the first exchange of the next run of steps (after a rebalance: the hand-over of the
    // columns between the old and the new boundary); cap_alloc = what the four buffers of every slab hold.
    std::vector<int> cap, cap_first;
    int cap_alloc = 0;
    double halo_margin = 1.25;  // SPHMG_HALO_MARGIN
    bool broken = false;        // a slab failed inside an RCCL run: the communicators were aborted, nothing runs any more
    long long records_sent = 0, exchanges = 0;  // sum over slabs and exchanges of the records shipped (headers included)
    bool have_halo = false;  // the ghosts of the next step are already appended
    bool threaded = true;         // one host thread per slab enqueues its steps (SPHMG_THREADS=0: one thread drives all slabs)
    bool copy_transport = false;  // messages by hipMemcpyAsync instead of RCCL (slabs sharing a device; SPHMG_TRANSPORT=copy)
    std::vector<Slab> slabs;
    std::vector<int> bounds;
    // SPHMG_TIMING=1: host time spent enqueuing, per phase, printed to stderr by sphmg_destroy
    bool timing = false;
    double host_us[6] = {0, 0, 0, 0, 0, 0};
    long long timed_steps = 0;
};

namespace {
// Grid column of x, same binary32 arithmetic as the device (sphhip_internal.h: sph_cell_coord).
int grid_column(float x, const SphParams& p, const SphDerived& d) {
    const float f = std::floor((x - p.xmin) * d.inv_cell);
    if (!(f >= 0.0f)) return 0;
    if (f >= (float)d.nx) return d.nx - 1;
    return (int)f;
}

// Column boundaries giving every slab about the same particle count; every slab at least MIN_SLAB_COLS wide; with
// `prev`, boundary g stays strictly between prev[g-1] and prev[g+1] so that every particle's new owner is its old
// owner or an adjacent slab (migration is nearest-neighbour only).
std::vector<int> balanced_bounds(const std::vector<long long>& hist, int world, const std::vector<int>* prev) {
    const int nx = (int)hist.size();
    std::vector<long long> csum(nx + 1, 0);
    for (int i = 0; i < nx; i++) csum[i + 1] = csum[i] + hist[i];
    const long long total = csum[nx];
    std::vector<int> b{0};
    for (int g = 1; g < world; g++) {
        const double target = (double)total * g / world;
        int c = (int)(std::lower_bound(csum.begin(), csum.end(), target, [](long long v, double t) { return (double)v < t; }) - csum.begin());
        int lo = b.back() + MIN_SLAB_COLS, hi = nx - (world - g) * MIN_SLAB_COLS;
        if (prev) {
            lo = std::max(lo, (*prev)[g - 1] + MIN_SLAB_COLS);
            hi = std::min(hi, (*prev)[g + 1] - MIN_SLAB_COLS);
        }
        b.push_back(std::min(std::max(c, lo), hi));
    }
    b.push_back(nx);
    return b;
}

// Records per message across every inner boundary between two rebalances: the HALO columns next to the boundary plus
// the migrants of one step, sized from the fullest column within reach of the boundary (the halo and two columns more
// on each side) times `margin`.  Every slab derives the same numbers from the same global histogram, so a message
// travels at a size the host fixes beforehand without a count round trip -- and ships about what is needed (the first
// version sent 2.5 x HALO x the fullest column of the whole domain, every step).  Same rule as sph_amd/slab.py.
constexpr int HALO_FLOOR = 1024;
std::vector<int> steady_caps(const std::vector<long long>& hist, const std::vector<int>& b, double margin, int floor_records) {
    std::vector<int> caps(b.size(), 0);
    const int nx = (int)hist.size();
    for (size_t g = 1; g + 1 < b.size(); g++) {
        long long m = 0;
        for (int c = std::max(b[g] - HALO - 2, 0); c < std::min(b[g] + HALO + 2, nx); c++) m = std::max(m, hist[c]);
        caps[g] = (int)std::max<long long>((long long)(margin * HALO * (double)m), floor_records);
    }
    return caps;
}

// ... and of the FIRST exchange after the boundaries moved from `old` to `now`: it hands over every column between the
// two besides the halo.  The histogram was taken at the rebalance itself, so its sum over those columns is what the
// message will hold (+ 1/16 + 256: migrants still with their old owner are counted at their new column).
std::vector<int> handover_caps(const std::vector<long long>& hist, const std::vector<int>& old, const std::vector<int>& now,
                               const std::vector<int>& steady) {
    std::vector<int> caps = steady;
    const int nx = (int)hist.size();
    for (size_t g = 1; g + 1 < now.size(); g++) {
        if (old[g] == now[g]) continue;
        long long need = 0;
        for (int c = std::max(std::min(old[g], now[g]) - HALO, 0); c < std::min(std::max(old[g], now[g]) + HALO, nx); c++) need += hist[c];
        caps[g] = (int)std::max<long long>(steady[g], need + need / 16 + 256);
    }
    return caps;
}

int floor_records() {
    const char* e = std::getenv("SPHMG_HALO_FLOOR");
    return e && std::atoi(e) > 0 ? std::atoi(e) : HALO_FLOOR;
}

struct HostTimer {
    SphMg* mg;
    int slot;
    std::chrono::steady_clock::time_point t0;
    HostTimer(SphMg* m, int s) : mg(m), slot(s) {
        if (mg->timing) t0 = std::chrono::steady_clock::now();
    }
    ~HostTimer() {
        if (mg->timing) mg->host_us[slot] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
};
const char* const HOST_PHASE[6] = {"pack", "exchange", "append", "step_begin", "step_finish", "step"};

// Makes the four message buffers of every slab hold (cap + 1) records (grow only; the callers have synchronised).
int alloc_messages(SphMg* mg, int cap) {
    if (cap <= mg->cap_alloc) return SPHHIP_OK;
    const size_t bytes = sizeof(HaloRec) * ((size_t)cap + 1);
    for (Slab& s : mg->slabs) {
        HIP_OK(hipSetDevice(s.device));
        for (void** p : {&s.send_l, &s.send_r, &s.recv_l, &s.recv_r}) {
            if (*p) HIP_OK(hipFree(*p));
            HIP_OK(hipMalloc(p, bytes));
            HIP_OK(hipMemset(*p, 0, bytes));
        }
    }
    mg->cap_alloc = cap;
    return SPHHIP_OK;
}

size_t msg_bytes(int cap) { return sizeof(HaloRec) * ((size_t)cap + 1); }

int set_bounds(SphMg* mg, const std::vector<int>& b) {
    mg->bounds = b;
    for (int g = 0; g < mg->world; g++) {
        Slab& s = mg->slabs[g];
        s.c0 = b[g];
        s.c1 = b[g + 1];
        SPH_OK(sphhip_set_owned_columns(s.ctx, s.c0, s.c1));
    }
    return SPHHIP_OK;
}

// One fixed-size message per direction (header + records) between x-neighbours, all slabs in one group, on the
// communication streams, ordered behind the packs by events; the compute streams wait for `received` before they
// append.  Replaces MPI_Allgather x 4 per step (MPI_sph.cpp:144-147,163-166).
int exchange_all(SphMg* mg, const std::vector<int>& cap) {
    HostTimer ht(mg, 1);
    for (int g = 0; g < mg->world; g++) mg->records_sent += (g > 0 ? cap[g] + 1 : 0) + (g < mg->world - 1 ? cap[g + 1] + 1 : 0);
    mg->exchanges++;
    for (Slab& s : mg->slabs) {
        HIP_OK(hipSetDevice(s.device));
        HIP_OK(hipEventRecord(s.packed, s.compute));
    }
    if (mg->copy_transport) {
        // The sender pushes: its communication stream waits until its own and both receivers' packs are done (a
        // receiver's pack is ordered behind its last read of the receive buffers), then copies device to device.
        for (int g = 0; g < mg->world; g++) {
            Slab& s = mg->slabs[g];
            HIP_OK(hipSetDevice(s.device));
            for (int o = std::max(g - 1, 0); o <= std::min(g + 1, mg->world - 1); o++) HIP_OK(hipStreamWaitEvent(s.comm_stream, mg->slabs[o].packed, 0));
            if (g > 0) HIP_OK(hipMemcpyAsync(mg->slabs[g - 1].recv_r, s.send_l, msg_bytes(cap[g]), hipMemcpyDefault, s.comm_stream));
            if (g < mg->world - 1) HIP_OK(hipMemcpyAsync(mg->slabs[g + 1].recv_l, s.send_r, msg_bytes(cap[g + 1]), hipMemcpyDefault, s.comm_stream));
            HIP_OK(hipEventRecord(s.received, s.comm_stream));
        }
        return SPHHIP_OK;
    }
    for (Slab& s : mg->slabs) {
        HIP_OK(hipSetDevice(s.device));
        HIP_OK(hipStreamWaitEvent(s.comm_stream, s.packed, 0));
    }
    NCCL_OK(ncclGroupStart());
    for (int g = 0; g < mg->world; g++) {
        Slab& s = mg->slabs[g];
        if (g > 0) {
            NCCL_OK(ncclSend(s.send_l, msg_bytes(cap[g]), ncclInt8, g - 1, s.comm, s.comm_stream));
            NCCL_OK(ncclRecv(s.recv_l, msg_bytes(cap[g]), ncclInt8, g - 1, s.comm, s.comm_stream));
        }
        if (g < mg->world - 1) {
            NCCL_OK(ncclSend(s.send_r, msg_bytes(cap[g + 1]), ncclInt8, g + 1, s.comm, s.comm_stream));
            NCCL_OK(ncclRecv(s.recv_r, msg_bytes(cap[g + 1]), ncclInt8, g + 1, s.comm, s.comm_stream));
        }
    }
    NCCL_OK(ncclGroupEnd());
    for (Slab& s : mg->slabs) {
        HIP_OK(hipSetDevice(s.device));
        HIP_OK(hipEventRecord(s.received, s.comm_stream));
    }
    return SPHHIP_OK;
}

// The first exchange of every neighbour pair, header records only (the message buffers are zero: "0 records"), then a
// wait for all of it.
int warm_up_transport(SphMg* mg) {
    NCCL_OK(ncclGroupStart());
    for (int g = 0; g < mg->world; g++) {
        Slab& s = mg->slabs[g];
        if (g > 0) {
            NCCL_OK(ncclSend(s.send_l, sizeof(HaloRec), ncclInt8, g - 1, s.comm, s.comm_stream));
            NCCL_OK(ncclRecv(s.recv_l, sizeof(HaloRec), ncclInt8, g - 1, s.comm, s.comm_stream));
        }
        if (g < mg->world - 1) {
            NCCL_OK(ncclSend(s.send_r, sizeof(HaloRec), ncclInt8, g + 1, s.comm, s.comm_stream));
            NCCL_OK(ncclRecv(s.recv_r, sizeof(HaloRec), ncclInt8, g + 1, s.comm, s.comm_stream));
        }
    }
    NCCL_OK(ncclGroupEnd());
    for (Slab& s : mg->slabs) {
        HIP_OK(hipSetDevice(s.device));
        HIP_OK(hipStreamSynchronize(s.comm_stream));
    }
    return SPHHIP_OK;
}

int append_received(SphMg* mg, const std::vector<int>& cap) {
    HostTimer ht(mg, 2);
    for (int g = 0; g < mg->world; g++) {
        Slab& s = mg->slabs[g];
        HIP_OK(hipSetDevice(s.device));
        // own transfers done (the send buffers are free again) and, for the copy transport, the neighbours' pushes too
        for (int o = std::max(g - 1, 0); o <= std::min(g + 1, mg->world - 1); o++) HIP_OK(hipStreamWaitEvent(s.compute, mg->slabs[o].received, 0));
        SPH_OK(sphhip_append_messages(s.ctx, g > 0 ? s.recv_l : nullptr, cap[g], g < mg->world - 1 ? s.recv_r : nullptr, cap[g + 1]));
    }
    return SPHHIP_OK;
}

int pack_all(SphMg* mg, const std::vector<int>& cap) {
    HostTimer ht(mg, 0);
    for (int g = 0; g < mg->world; g++) {
        Slab& s = mg->slabs[g];
        const bool has_l = g > 0, has_r = g < mg->world - 1;
        SPH_OK(sphhip_pack_halo_async(s.ctx, has_l ? s.c0 + HALO : -1, has_r ? s.c1 - HALO : BIG, has_l ? s.send_l : nullptr,
                                      has_l ? cap[g] : 0, has_r ? s.send_r : nullptr, has_r ? cap[g + 1] : 0));
    }
    return SPHHIP_OK;
}

int run_steps_threaded(SphMg* mg, float mx, float my, float strength, int k);

// k steps between two rebalances.  Every step but the last is split: the columns next to the slab boundaries first
// (sphhip_step_begin), then the messages of the NEXT step are packed from them and travel while the interior
// columns are computed (sphhip_step_finish).
int run_steps(SphMg* mg, float mx, float my, float strength, int k) {
    if (mg->world == 1) {
        SPH_OK(sphhip_step(mg->slabs[0].ctx, mx, my, strength, k));
        return SPHHIP_OK;
    }
    if (mg->threaded) return run_steps_threaded(mg, mx, my, strength, k);
    for (int i = 0; i < k; i++) {
        if (!mg->have_halo) {
            int rc;
            const std::vector<int>& cap = i == 0 ? mg->cap_first : mg->cap;
            if ((rc = pack_all(mg, cap)) || (rc = exchange_all(mg, cap)) || (rc = append_received(mg, cap))) return rc;
            mg->cap_first = mg->cap;  // (the hand-over after a rebalance is done)
        }
        mg->have_halo = false;
        mg->timed_steps++;
        if (i == k - 1) {
            HostTimer ht(mg, 5);
            for (Slab& s : mg->slabs) SPH_OK(sphhip_step(s.ctx, mx, my, strength, 1));
            break;
        }
        {
            HostTimer ht(mg, 3);
            for (int g = 0; g < mg->world; g++)
                SPH_OK(sphhip_step_begin(mg->slabs[g].ctx, mx, my, strength, g > 0 ? SPHHIP_PACK_REACH : 0,
                                         g < mg->world - 1 ? SPHHIP_PACK_REACH : 0));
        }
        int rc;
        if ((rc = pack_all(mg, mg->cap)) || (rc = exchange_all(mg, mg->cap))) return rc;
        {
            HostTimer ht(mg, 4);
            for (Slab& s : mg->slabs) SPH_OK(sphhip_step_finish(s.ctx));
        }
        if ((rc = append_received(mg, mg->cap))) return rc;
        mg->have_halo = true;
    }
    return SPHHIP_OK;
}

// ---- one host thread per slab ---------------------------------------------------------------------------------------
// Enqueuing a slab step costs the host 60-120 us (profiles/r02_mg_host_overhead.txt): about ten launches, event
// records / waits and the transport call.  One thread for eight GPUs would make that 0.5-1 ms per step, more than the
// GPU needs for a step at 1-2 M particles per GPU; with one thread per slab the slabs are enqueued side by side.
// With RCCL the threads never meet on the host (every thread groups its own send / recv pair; the ranks meet on the
// devices); the copy transport waits on the neighbours' events, which have to be recorded first: two host barriers
// per exchange.
class HostBarrier {
   public:
    explicit HostBarrier(int n) : n_(n) {}
    // false: some thread failed and broke the barrier (nobody waits any more)
    bool wait() {
        std::unique_lock<std::mutex> lk(m_);
        if (broken_) return false;
        const long gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            gen_++;
            cv_.notify_all();
            return true;
        }
        cv_.wait(lk, [&] { return gen_ != gen || broken_; });
        return !broken_;
    }
    void breakup() {
        std::lock_guard<std::mutex> lk(m_);
        broken_ = true;
        cv_.notify_all();
    }

   private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    long gen_ = 0;
    bool broken_ = false;
};

int slab_pack(SphMg* mg, int g, const std::vector<int>& cap) {
    Slab& s = mg->slabs[g];
    const bool has_l = g > 0, has_r = g < mg->world - 1;
    SPH_OK(sphhip_pack_halo_async(s.ctx, has_l ? s.c0 + HALO : -1, has_r ? s.c1 - HALO : BIG, has_l ? s.send_l : nullptr,
                                  has_l ? cap[g] : 0, has_r ? s.send_r : nullptr, has_r ? cap[g + 1] : 0));
    return SPHHIP_OK;
}

// Starts slab g's transfers on its communication stream, ordered behind the pack just enqueued.
int slab_send(SphMg* mg, int g, HostBarrier& bar, const std::vector<int>& cap) {
    Slab& s = mg->slabs[g];
    const size_t bytes_l = msg_bytes(cap[g]), bytes_r = msg_bytes(cap[g + 1]);
    HIP_OK(hipEventRecord(s.packed, s.compute));
    if (mg->copy_transport) {
        if (!bar.wait()) return fail(SPHHIP_ESTATE, "another slab failed");  // the neighbours' `packed` are recorded
        for (int o = std::max(g - 1, 0); o <= std::min(g + 1, mg->world - 1); o++) HIP_OK(hipStreamWaitEvent(s.comm_stream, mg->slabs[o].packed, 0));
        if (g > 0) HIP_OK(hipMemcpyAsync(mg->slabs[g - 1].recv_r, s.send_l, bytes_l, hipMemcpyDefault, s.comm_stream));
        if (g < mg->world - 1) HIP_OK(hipMemcpyAsync(mg->slabs[g + 1].recv_l, s.send_r, bytes_r, hipMemcpyDefault, s.comm_stream));
        HIP_OK(hipEventRecord(s.received, s.comm_stream));
        return SPHHIP_OK;
    }
    HIP_OK(hipStreamWaitEvent(s.comm_stream, s.packed, 0));
    NCCL_OK(ncclGroupStart());
    if (g > 0) {
        NCCL_OK(ncclSend(s.send_l, bytes_l, ncclInt8, g - 1, s.comm, s.comm_stream));
        NCCL_OK(ncclRecv(s.recv_l, bytes_l, ncclInt8, g - 1, s.comm, s.comm_stream));
    }
    if (g < mg->world - 1) {
        NCCL_OK(ncclSend(s.send_r, bytes_r, ncclInt8, g + 1, s.comm, s.comm_stream));
        NCCL_OK(ncclRecv(s.recv_r, bytes_r, ncclInt8, g + 1, s.comm, s.comm_stream));
    }
    NCCL_OK(ncclGroupEnd());
    HIP_OK(hipEventRecord(s.received, s.comm_stream));
    return SPHHIP_OK;
}

// Makes slab g's compute stream wait for its messages and appends them.
int slab_receive(SphMg* mg, int g, HostBarrier& bar, const std::vector<int>& cap) {
    Slab& s = mg->slabs[g];
    if (mg->copy_transport) {
        if (!bar.wait()) return fail(SPHHIP_ESTATE, "another slab failed");  // the neighbours' `received` are recorded
        for (int o = std::max(g - 1, 0); o <= std::min(g + 1, mg->world - 1); o++) HIP_OK(hipStreamWaitEvent(s.compute, mg->slabs[o].received, 0));
    } else {
        HIP_OK(hipStreamWaitEvent(s.compute, s.received, 0));
    }
    SPH_OK(sphhip_append_messages(s.ctx, g > 0 ? s.recv_l : nullptr, cap[g], g < mg->world - 1 ? s.recv_r : nullptr, cap[g + 1]));
    return SPHHIP_OK;
}

// The k steps of run_steps for slab g: per slab the same order of calls as the one-thread loop.
int slab_steps(SphMg* mg, int g, HostBarrier& bar, float mx, float my, float strength, int k, bool have_halo) {
    Slab& s = mg->slabs[g];
    HIP_OK(hipSetDevice(s.device));
    struct Lap {  // the host time of slab 0's thread stands for all (they run side by side)
        SphMg* mg;
        bool on;
        std::chrono::steady_clock::time_point t;
        void mark(int slot) {
            if (!on) return;
            const auto now = std::chrono::steady_clock::now();
            mg->host_us[slot] += std::chrono::duration<double, std::micro>(now - t).count();
            t = now;
        }
    } lap{mg, mg->timing && g == 0, std::chrono::steady_clock::now()};
    int rc;
    for (int i = 0; i < k; i++) {
        if (!have_halo) {
            // (the first exchange of a run: after a rebalance it hands whole columns over and has a size of its own;
            // cap_first is reset by run_steps_threaded once every thread is done with it)
            const std::vector<int>& cap = i == 0 ? mg->cap_first : mg->cap;
            if ((rc = slab_pack(mg, g, cap))) return rc;
            lap.mark(0);
            if ((rc = slab_send(mg, g, bar, cap))) return rc;
            lap.mark(1);
            if ((rc = slab_receive(mg, g, bar, cap))) return rc;
            lap.mark(2);
        }
        have_halo = false;
        if (g == 0) mg->timed_steps++;
        if (i == k - 1) {
            SPH_OK(sphhip_step(s.ctx, mx, my, strength, 1));
            lap.mark(5);
            break;
        }
        SPH_OK(sphhip_step_begin(s.ctx, mx, my, strength, g > 0 ? SPHHIP_PACK_REACH : 0, g < mg->world - 1 ? SPHHIP_PACK_REACH : 0));
        lap.mark(3);
        if ((rc = slab_pack(mg, g, mg->cap))) return rc;
        lap.mark(0);
        if ((rc = slab_send(mg, g, bar, mg->cap))) return rc;  // travels while the interior columns are computed
        lap.mark(1);
        SPH_OK(sphhip_step_finish(s.ctx));
        lap.mark(4);
        if ((rc = slab_receive(mg, g, bar, mg->cap))) return rc;
        lap.mark(2);
        have_halo = true;
    }
    return SPHHIP_OK;
}

int run_steps_threaded(SphMg* mg, float mx, float my, float strength, int k) {
    HostBarrier bar(mg->world);
    std::vector<int> rcs(mg->world, 0);
    std::vector<std::string> msgs(mg->world);
    std::vector<std::thread> team;
    const bool have_halo = mg->have_halo;
    for (int g = 0; g < mg->world; g++)
        team.emplace_back([&, g] {
            rcs[g] = slab_steps(mg, g, bar, mx, my, strength, k, have_halo);
            if (rcs[g]) {
                msgs[g] = g_err;  // (thread-local in this thread)
                bar.breakup();
            }
        });
    for (std::thread& t : team) t.join();
    mg->have_halo = false;  // the last step of a run is never split
    {
        long long per_exchange = 0;
        for (int g = 0; g < mg->world; g++) per_exchange += (g > 0 ? mg->cap[g] + 1 : 0) + (g < mg->world - 1 ? mg->cap[g + 1] + 1 : 0);
        long long first = 0;
        for (int g = 0; g < mg->world; g++) first += (g > 0 ? mg->cap_first[g] + 1 : 0) + (g < mg->world - 1 ? mg->cap_first[g + 1] + 1 : 0);
        mg->records_sent += first + per_exchange * (k - 1);
        mg->exchanges += k;
    }
    mg->cap_first = mg->cap;  // (the hand-over after a rebalance is done)
    bool failed = false;
    for (int g = 0; g < mg->world; g++) failed = failed || rcs[g] != 0;
    if (failed && !mg->copy_transport) {
        // A slab that failed returned before its ncclGroupEnd, its neighbours' send / recv kernels would wait for it
        // for ever (and with them sphmg_synchronize / sphmg_destroy): abort the communicators, which makes those
        // kernels exit, and refuse further work on this object.
        for (Slab& s : mg->slabs)
            if (s.comm) {
                (void)ncclCommAbort(s.comm);
                s.comm = nullptr;
            }
        mg->broken = true;
    }
    for (int g = 0; g < mg->world; g++)
        if (rcs[g] && msgs[g] != "another slab failed") return fail(rcs[g], "slab " + std::to_string(g) + ": " + msgs[g]);
    for (int g = 0; g < mg->world; g++)
        if (rcs[g]) return fail(rcs[g], msgs[g]);
    return SPHHIP_OK;
}

void destroy(SphMg* mg) {
    if (mg->timing && mg->timed_steps > 0) {
        std::fprintf(stderr, "[sphmg] host enqueue time per step over %lld steps, %d slabs (us):", mg->timed_steps, mg->world);
        for (int i = 0; i < 6; i++) std::fprintf(stderr, " %s %.1f", HOST_PHASE[i], mg->host_us[i] / (double)mg->timed_steps);
        std::fprintf(stderr, "\n");
    }
    for (Slab& s : mg->slabs) {
        (void)hipSetDevice(s.device);
        if (s.ctx) sphhip_destroy(s.ctx);
        if (s.comm) (void)ncclCommDestroy(s.comm);
        for (void* p : {s.send_l, s.send_r, s.recv_l, s.recv_r})
            if (p) (void)hipFree(p);
        if (s.packed) (void)hipEventDestroy(s.packed);
        if (s.received) (void)hipEventDestroy(s.received);
        if (s.comm_stream) (void)hipStreamDestroy(s.comm_stream);
    }
    delete mg;
}

int histogram(SphMg* mg, std::vector<long long>& hist) {
    hist.assign(mg->drv.nx, 0);
    std::vector<int> h(mg->drv.nx);
    for (Slab& s : mg->slabs) {
        SPH_OK(sphhip_drop_ghosts(s.ctx));
        SPH_OK(sphhip_column_histogram(s.ctx, h.data()));
        for (int i = 0; i < mg->drv.nx; i++) hist[i] += h[i];
    }
    mg->have_halo = false;
    return SPHHIP_OK;
}
}  // namespace

extern "C" const char* sphmg_last_error(void) { return g_err.c_str(); }

extern "C" int sphmg_create(const SphParams* p, const void* aos28, int n, int n_gpus, const int* devices, SphMg** out) {
    if (!p || !out || (n > 0 && !aos28)) return fail(SPHHIP_EINVAL, "sphmg_create: null argument");
    if (n < 0 || n_gpus < 1) return fail(SPHHIP_EINVAL, "sphmg_create: bad n / n_gpus");
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    SphMg* mg = new SphMg();
    mg->params = *p;
    mg->n = n;
    mg->world = n_gpus;
    const char* tm = std::getenv("SPHMG_TIMING");
    mg->timing = tm && *tm && std::strcmp(tm, "0");
    const char* th = std::getenv("SPHMG_THREADS");
    mg->threaded = !(th && !std::strcmp(th, "0"));
    if (int rc = sphhip_derive(p, &mg->drv)) {
        delete mg;
        return fail(rc, sphhip_last_error());
    }
    if (mg->drv.nx < n_gpus * MIN_SLAB_COLS) {
        delete mg;
        return fail(SPHHIP_EINVAL, "sphmg_create: the grid has too few columns for that many slabs (3 per slab)");
    }
    mg->slabs.resize(n_gpus);
    for (int g = 0; g < n_gpus; g++) {
        mg->slabs[g].device = devices ? devices[g] : g;
        if (mg->slabs[g].device < 0 || mg->slabs[g].device >= ndev) {
            delete mg;
            return fail(SPHHIP_EINVAL, "sphmg_create: no such HIP device");
        }
    }
    // owner of every particle: the slab of its grid column under count-balanced boundaries
    const unsigned char* rec = (const unsigned char*)aos28;
    std::vector<int> col(n);
    std::vector<long long> hist(mg->drv.nx, 0);
    for (int i = 0; i < n; i++) {
        float x;
        std::memcpy(&x, rec + (size_t)i * SPHHIP_PARTICLE_BYTES, 4);
        col[i] = grid_column(x, *p, mg->drv);
        hist[col[i]]++;
    }
    const std::vector<int> bounds = balanced_bounds(hist, n_gpus, nullptr);
    long long per = 1;
    for (int g = 0; g < n_gpus; g++) {
        long long c = 0;
        for (int k = bounds[g]; k < bounds[g + 1]; k++) c += hist[k];
        per = std::max(per, c);
    }
    if (const char* e = std::getenv("SPHMG_HALO_MARGIN"))
        if (std::atof(e) >= 1.0) mg->halo_margin = std::atof(e);
    mg->cap = steady_caps(hist, bounds, mg->halo_margin, floor_records());
    mg->cap_first = mg->cap;
    const int cap = std::max(*std::max_element(mg->cap.begin(), mg->cap.end()), HALO_FLOOR);
    // the balanced share with room for imbalance between two rebalances, plus ghosts (two halos: last step's stale
    // copies and the fresh ones) with room to grow
    const long long n_max = n_gpus == 1 ? std::max(n, 1) : per + per / 2 + 8ll * cap + 1024;
    auto bail = [&](int rc) {
        destroy(mg);
        return rc;
    };
    for (int g = 0; g < n_gpus; g++) {
        Slab& s = mg->slabs[g];
        if (hipSetDevice(s.device) != hipSuccess) return bail(fail(SPHHIP_EHIP, "hipSetDevice failed"));
        if (int rc = sphhip_create(p, (int)std::min<long long>(n_max, 1 << 30), s.device, &s.ctx)) return bail(fail(rc, sphhip_last_error()));
        if (hipStreamCreateWithFlags(&s.comm_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.packed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.received, hipEventDisableTiming) != hipSuccess)
            return bail(fail(SPHHIP_EHIP, "stream / event creation failed"));
        void* own = nullptr;
        if (int rc = sphhip_get_stream(s.ctx, &own)) return bail(fail(rc, sphhip_last_error()));
        s.compute = (hipStream_t)own;
    }
    if (int rc = set_bounds(mg, bounds)) return bail(rc);
    if (n_gpus > 1) {
        if (int rc = alloc_messages(mg, cap)) return bail(rc);
        bool shared = false;
        for (int g = 0; g < n_gpus; g++)
            for (int o = 0; o < g; o++) shared = shared || mg->slabs[g].device == mg->slabs[o].device;
        const char* tr = std::getenv("SPHMG_TRANSPORT");
        if (tr && std::strcmp(tr, "copy") && std::strcmp(tr, "rccl")) return bail(fail(SPHHIP_EINVAL, "SPHMG_TRANSPORT must be rccl or copy"));
        if (tr && !std::strcmp(tr, "rccl") && shared) return bail(fail(SPHHIP_EINVAL, "RCCL needs one distinct device per slab"));
        mg->copy_transport = shared || (tr && !std::strcmp(tr, "copy"));
    }
    if (n_gpus > 1 && !mg->copy_transport) {
        std::vector<ncclComm_t> comms(n_gpus);
        std::vector<int> devs(n_gpus);
        for (int g = 0; g < n_gpus; g++) devs[g] = mg->slabs[g].device;
        ncclResult_t r = ncclCommInitAll(comms.data(), n_gpus, devs.data());
        if (r != ncclSuccess) return bail(fail(SPHHIP_EHIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r)));
        for (int g = 0; g < n_gpus; g++) mg->slabs[g].comm = comms[g];
        // One header-sized exchange with every neighbour now, from this one thread: RCCL sets its peer connections up
        // inside the first ncclGroupEnd that uses them, blocking the host -- that must not happen inside the per-slab
        // enqueue threads (a thread that fails before its group call would leave its neighbours blocked there), and a
        // transport that does not work is reported here instead of hanging a run.
        if (int rc = warm_up_transport(mg)) return bail(rc);
    }
    // every slab gets its own particles as 20-byte records {x, y, vx, vy, id | valid}
    for (int g = 0; g < n_gpus; g++) {
        Slab& s = mg->slabs[g];
        std::vector<HaloRec> mine;
        for (int i = 0; i < n; i++) {
            if (col[i] < s.c0 || col[i] >= s.c1) continue;
            const unsigned char* r = rec + (size_t)i * SPHHIP_PARTICLE_BYTES;
            HaloRec h;
            std::memcpy(&h, r, 16);
            h.idv = (uint32_t)i | (r[24] ? 0u : 0x80000000u);
            mine.push_back(h);
        }
        if (hipSetDevice(s.device) != hipSuccess) return bail(fail(SPHHIP_EHIP, "hipSetDevice failed"));
        if (!mine.empty()) {
            void* d = nullptr;
            if (hipMalloc(&d, sizeof(HaloRec) * mine.size()) != hipSuccess) return bail(fail(SPHHIP_EHIP, "hipMalloc failed"));
            hipError_t e = hipMemcpy(d, mine.data(), sizeof(HaloRec) * mine.size(), hipMemcpyHostToDevice);
            int rc = e == hipSuccess ? sphhip_append(s.ctx, d, (int)mine.size()) : SPHHIP_EHIP;
            if (!rc) rc = sphhip_synchronize(s.ctx);
            (void)hipFree(d);
            if (rc) return bail(fail(rc, e == hipSuccess ? sphhip_last_error() : "hipMemcpy failed"));
        } else {
            // (an empty slab still needs a state: zero particles)
            if (int rc = sphhip_upload(s.ctx, nullptr, 0, 0)) return bail(fail(rc, sphhip_last_error()));
        }
    }
    *out = mg;
    return SPHHIP_OK;
}

#define NOT_BROKEN(mg) \
    if ((mg)->broken) return fail(SPHHIP_ESTATE, "a slab failed during an RCCL run: the communicators were aborted; destroy this object")

extern "C" int sphmg_rebalance(SphMg* mg) {
    if (!mg) return fail(SPHHIP_EINVAL, "null");
    NOT_BROKEN(mg);
    std::vector<long long> hist;
    if (int rc = histogram(mg, hist)) return rc;  // (synchronises every slab)
    const std::vector<int> prev = mg->bounds;
    if (int rc = set_bounds(mg, balanced_bounds(hist, mg->world, &prev))) return rc;
    if (mg->world > 1) {
        // message sizes of the new boundaries; the next exchange hands the columns between old and new over
        mg->cap = steady_caps(hist, mg->bounds, mg->halo_margin, floor_records());
        mg->cap_first = handover_caps(hist, prev, mg->bounds, mg->cap);
        const int want = std::max(*std::max_element(mg->cap.begin(), mg->cap.end()), *std::max_element(mg->cap_first.begin(), mg->cap_first.end()));
        if (want > mg->cap_alloc) {
            for (Slab& s : mg->slabs) {
                HIP_OK(hipSetDevice(s.device));
                HIP_OK(hipStreamSynchronize(s.comm_stream));
                SPH_OK(sphhip_synchronize(s.ctx));
            }
            if (int rc = alloc_messages(mg, want)) return rc;
        }
    }
    mg->rebalances++;
    return SPHHIP_OK;
}

extern "C" int sphmg_halo_stats(SphMg* mg, long long* exchanges, double* bytes_per_exchange_per_slab, int* max_message_records) {
    if (!mg) return fail(SPHHIP_EINVAL, "null");
    if (exchanges) *exchanges = mg->exchanges;
    if (bytes_per_exchange_per_slab)
        *bytes_per_exchange_per_slab = mg->exchanges ? (double)sizeof(HaloRec) * (double)mg->records_sent / ((double)mg->exchanges * mg->world) : 0.0;
    if (max_message_records) *max_message_records = mg->cap.empty() ? 0 : *std::max_element(mg->cap.begin(), mg->cap.end());
    return SPHHIP_OK;
}

extern "C" const char* sphmg_transport(SphMg* mg) { return !mg || mg->world < 2 ? "none" : mg->copy_transport ? "copy" : "rccl"; }

extern "C" int sphmg_step(SphMg* mg, float mx, float my, float strength, int nsteps, int rebalance_every) {
    if (!mg || nsteps < 0) return fail(SPHHIP_EINVAL, "sphmg_step: bad argument");
    NOT_BROKEN(mg);
    int done = 0;
    while (done < nsteps) {
        const int k = rebalance_every > 0 ? std::min(rebalance_every, nsteps - done) : nsteps - done;
        if (int rc = run_steps(mg, mx, my, strength, k)) return rc;
        done += k;
        if (rebalance_every > 0 && done < nsteps && mg->world > 1)
            if (int rc = sphmg_rebalance(mg)) return rc;
    }
    return SPHHIP_OK;
}

extern "C" int sphmg_synchronize(SphMg* mg) {
    if (!mg) return fail(SPHHIP_EINVAL, "null");
    NOT_BROKEN(mg);
    for (Slab& s : mg->slabs) {
        HIP_OK(hipSetDevice(s.device));
        HIP_OK(hipStreamSynchronize(s.comm_stream));
        int n = 0;
        SPH_OK(sphhip_sync_counts(s.ctx, &n));
    }
    return SPHHIP_OK;
}

extern "C" int sphmg_counts(SphMg* mg, int* per_gpu) {
    if (!mg || !per_gpu) return fail(SPHHIP_EINVAL, "null");
    for (int g = 0; g < mg->world; g++) {
        Slab& s = mg->slabs[g];
        // as sphmg_download: a particle that left the slab in the last step still counts here until it is handed on
        if (mg->world > 1) {
            if (mg->have_halo) {
                SPH_OK(sphhip_keep_columns(s.ctx, s.c0, s.c1));
            } else {
                SPH_OK(sphhip_drop_ghosts(s.ctx));
            }
        }
        SPH_OK(sphhip_count(s.ctx, &per_gpu[g]));
    }
    mg->have_halo = false;
    return SPHHIP_OK;
}

extern "C" int sphmg_bounds(SphMg* mg, int* bounds) {
    if (!mg || !bounds) return fail(SPHHIP_EINVAL, "null");
    std::copy(mg->bounds.begin(), mg->bounds.end(), bounds);
    return SPHHIP_OK;
}

extern "C" int sphmg_rebalance_count(SphMg* mg) { return mg ? mg->rebalances : 0; }

extern "C" int sphmg_download(SphMg* mg, void* aos28, int n) {
    if (!mg || (n > 0 && !aos28)) return fail(SPHHIP_EINVAL, "null");
    if (n != mg->n) return fail(SPHHIP_EINVAL, "sphmg_download: n differs from the particle count");
    std::vector<unsigned char> seen(n, 0);
    unsigned char* out = (unsigned char*)aos28;
    for (Slab& s : mg->slabs) {
        // owned particles only: whatever lies outside the slab's columns is a ghost copy (stale or just received)
        // or a migrant that has already been handed on
        if (mg->world > 1) {
            if (mg->have_halo) {
                SPH_OK(sphhip_keep_columns(s.ctx, s.c0, s.c1));
            } else {
                SPH_OK(sphhip_drop_ghosts(s.ctx));
            }
        }
        int cnt = 0;
        SPH_OK(sphhip_count(s.ctx, &cnt));
        std::vector<unsigned char> rec((size_t)cnt * SPHHIP_PARTICLE_BYTES);
        std::vector<int> ids(cnt);
        SPH_OK(sphhip_download_ids(s.ctx, rec.data(), ids.data(), cnt));
        for (int k = 0; k < cnt; k++) {
            if (ids[k] < 0 || ids[k] >= n || seen[ids[k]]) return fail(SPHHIP_ESTATE, "sphmg_download: slab ownership broken (particle duplicated or id out of range)");
            seen[ids[k]] = 1;
            std::memcpy(out + (size_t)ids[k] * SPHHIP_PARTICLE_BYTES, rec.data() + (size_t)k * SPHHIP_PARTICLE_BYTES, SPHHIP_PARTICLE_BYTES);
        }
    }
    mg->have_halo = false;
    for (int i = 0; i < n; i++)
        if (!seen[i]) return fail(SPHHIP_ESTATE, "sphmg_download: slab ownership broken (particle lost)");
    return SPHHIP_OK;
}

extern "C" int sphmg_destroy(SphMg* mg) {
    if (mg) destroy(mg);
    return SPHHIP_OK;
}

extern "C" int sphmg_selftest_transport(int device, int bytes) {
    if (bytes < 1) return fail(SPHHIP_EINVAL, "bytes < 1");
    HIP_OK(hipSetDevice(device));
    ncclComm_t comm;
    NCCL_OK(ncclCommInitAll(&comm, 1, &device));
    hipStream_t st;
    HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned char *a = nullptr, *b = nullptr;
    HIP_OK(hipMalloc((void**)&a, bytes));
    HIP_OK(hipMalloc((void**)&b, bytes));
    std::vector<unsigned char> h(bytes), back(bytes, 0);
    for (int i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 131 + 7);
    HIP_OK(hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(b, 0, bytes));
    NCCL_OK(ncclGroupStart());
    NCCL_OK(ncclSend(a, bytes, ncclInt8, 0, comm, st));
    NCCL_OK(ncclRecv(b, bytes, ncclInt8, 0, comm, st));
    NCCL_OK(ncclGroupEnd());
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(back.data(), b, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(a);
    (void)hipFree(b);
    (void)hipStreamDestroy(st);
    (void)ncclCommDestroy(comm);
    return back == h ? SPHHIP_OK : fail(SPHHIP_EHIP, "sphmg_selftest_transport: received bytes differ from the bytes sent");
}
