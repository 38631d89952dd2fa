// TEST INFRASTRUCTURE - not part of the product, never loaded unless GNN_RCCL_LIBRARY points at it.
//
// A stand-in for the handful of RCCL entry points the engine uses (csrc/gnn_engine.hip, rccl_load), so that the REAL multi-process path
// of the engine - one process per rank, unique-id rendezvous, gnn_comm_create, the grouped all-gathers of state rows and flags, the
// all-to-alls of the feature-sliced exchange and the point-to-point schedule of its pipelined return - can be executed end to end on a
// box with ONE GPU (RCCL itself refuses two ranks on one device).  Ranks are processes of one host; data moves through files under
// /dev/shm: every call first waits for the caller's stream, copies device -> shared file, synchronises with the peers through a small
// control segment (sequence counters, spin + yield), and copies shared file -> device.  Blocking where RCCL is asynchronous: timing
// says nothing, ORDER and CONTENT of the calls are what it checks - a wrong peer, count, offset or call order deadlocks (bounded waits:
// error 6 after 120 s) or produces wrong states, which the tests compare with the oracle.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {
constexpr int MAXR = 16;
struct Ctrl {
    uint64_t arrive;                 // barrier: total arrivals so far
    uint64_t sent[MAXR][MAXR];       // messages src -> dst published so far
    uint64_t taken[MAXR][MAXR];      // ... and consumed
    double red[MAXR];                // all-reduce operands
};
struct Comm {
    std::string name;
    int rank = 0, world = 1;
    Ctrl *ctrl = nullptr;
    uint64_t barriers = 0;           // barriers this rank has passed
    uint64_t coll = 0;               // collectives issued (same order on every rank)
    uint64_t psent[MAXR] = {0}, ptaken[MAXR] = {0};
};
struct Op { int kind; const void *send; void *recv; size_t bytes; int peer; Comm *c; hipStream_t st; };   // kind 0 send, 1 recv
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dsize(int dt) { return dt == 8 ? 8 : (dt == 7 || dt == 2 || dt == 3) ? 4 : (dt == 0 || dt == 1) ? 1 : (dt == 6 || dt == 9) ? 2 : 8; }

double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

bool wait_ge(volatile uint64_t *p, uint64_t v)
{
    const double t0 = now();
    for (unsigned spins = 0; __atomic_load_n(p, __ATOMIC_ACQUIRE) < v; ++spins) {
        if ((spins & 255) == 255) { sched_yield(); if (now() - t0 > 120.0) return false; }
    }
    return true;
}

bool barrier(Comm *c)
{
    ++c->barriers;
    __atomic_fetch_add(&c->ctrl->arrive, 1, __ATOMIC_ACQ_REL);
    return wait_ge(&c->ctrl->arrive, c->barriers * (uint64_t)c->world);
}

// device -> a fresh shared file / shared file -> device
bool put(const std::string &file, const void *dev, size_t bytes)
{
    const int fd = shm_open(file.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return false;
    bool ok = bytes == 0 || ftruncate(fd, (off_t)bytes) == 0;
    if (ok && bytes) {
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ok = m != MAP_FAILED && hipMemcpy(m, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess;
        if (m != MAP_FAILED) munmap(m, bytes);
    }
    close(fd);
    return ok;
}
bool get(const std::string &file, void *dev, size_t bytes, bool unlink_after)
{
    const int fd = shm_open(file.c_str(), O_RDONLY, 0600);
    if (fd < 0) return false;
    bool ok = true;
    if (bytes) {
        struct stat sb;
        ok = fstat(fd, &sb) == 0 && (size_t)sb.st_size == bytes;          // a count mismatch between sender and receiver is an error
        if (ok) {
            void *m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
            // (the device synchronisation: a host-to-device copy from pageable memory may return before the device side has it, and the
            //  consumers run on non-blocking streams that the null stream does not order)
            ok = m != MAP_FAILED && hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            if (m != MAP_FAILED) munmap(m, bytes);
        }
    }
    close(fd);
    if (unlink_after) shm_unlink(file.c_str());
    return ok;
}

int run_ops(std::vector<Op> &ops)
{
    // sends first (they never wait), then the receives
    for (Op &o : ops) {
        if (o.kind != 0) continue;
        Comm *c = o.c;
        if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
        const uint64_t seq = c->psent[o.peer]++;
        const std::string f = c->name + ".p2p." + std::to_string(c->rank) + "." + std::to_string(o.peer) + "." + std::to_string(seq);
        if (!put(f, o.send, o.bytes)) return 2;
        __atomic_store_n(&c->ctrl->sent[c->rank][o.peer], seq + 1, __ATOMIC_RELEASE);
    }
    for (Op &o : ops) {
        if (o.kind != 1) continue;
        Comm *c = o.c;
        if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
        const uint64_t seq = c->ptaken[o.peer]++;
        if (!wait_ge(&c->ctrl->sent[o.peer][c->rank], seq + 1)) return 6;
        const std::string f = c->name + ".p2p." + std::to_string(o.peer) + "." + std::to_string(c->rank) + "." + std::to_string(seq);
        if (!get(f, o.recv, o.bytes, true)) return 3;
        __atomic_store_n(&c->ctrl->taken[o.peer][c->rank], seq + 1, __ATOMIC_RELEASE);
    }
    ops.clear();
    return 0;
}
}   // namespace

extern "C" {

const char *ncclGetErrorString(int r)
{
    switch (r) {
    case 0: return "ok";
    case 1: return "mock transport: HIP error";
    case 2: return "mock transport: cannot write a shared file";
    case 3: return "mock transport: shared file missing or of another size (count mismatch between the ranks?)";
    case 4: return "mock transport: bad argument";
    case 6: return "mock transport: timed out waiting for a peer (call order or peer mismatch?)";
    default: return "mock transport: error";
    }
}

int ncclGetUniqueId(void *id)
{
    memset(id, 0, 128);
    timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf((char *)id, 128, "/gnnmock_%d_%ld_%ld", (int)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return 0;
}

struct Id128 { char b[128]; };
int ncclCommInitRank(void **comm, int world, Id128 id, int rank)
{
    if (!comm || world < 1 || world > MAXR || rank < 0 || rank >= world) return 4;
    id.b[127] = 0;
    Comm *c = new Comm();
    c->name = id.b; c->rank = rank; c->world = world;
    const std::string f = c->name + ".ctrl";
    const int fd = shm_open(f.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Ctrl)) != 0) { delete c; return 2; }
    c->ctrl = (Ctrl *)mmap(nullptr, sizeof(Ctrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->ctrl == MAP_FAILED) { delete c; return 2; }
    if (!barrier(c)) { delete c; return 6; }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    barrier(c);                                         // nobody unlinks the control segment while a peer still needs it
    munmap(c->ctrl, sizeof(Ctrl));
    if (c->rank == 0) shm_unlink((c->name + ".ctrl").c_str());
    delete c;
    return 0;
}

int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd()
{
    if (--g_depth > 0) return 0;
    g_depth = 0;
    return run_ops(g_ops);
}

int ncclSend(const void *send, size_t count, int dt, int peer, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return 4;
    g_ops.push_back(Op{0, send, nullptr, count * dsize(dt), peer, c, st});
    return g_depth > 0 ? 0 : run_ops(g_ops);
}
int ncclRecv(void *recv, size_t count, int dt, int peer, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return 4;
    g_ops.push_back(Op{1, nullptr, recv, count * dsize(dt), peer, c, st});
    return g_depth > 0 ? 0 : run_ops(g_ops);
}

// collectives run at once, also inside a group (the engine groups all-gathers of different buffers: their order is the same on all ranks)
int ncclAllGather(const void *send, void *recv, size_t count, int dt, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c) return 4;
    const size_t bytes = count * dsize(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    const uint64_t seq = c->coll++;
    const std::string base = c->name + ".ag." + std::to_string(seq) + ".";
    if (!put(base + std::to_string(c->rank), send, bytes)) return 2;
    if (!barrier(c)) return 6;
    for (int p = 0; p < c->world; ++p)
        if (!get(base + std::to_string(p), (char *)recv + (size_t)p * bytes, bytes, false)) return 3;
    if (!barrier(c)) return 6;
    shm_unlink((base + std::to_string(c->rank)).c_str());
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dt, int op, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || count != 1 || dt != 8 || op != 2) return 4;            // the engine reduces one double with max
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    double v = 0;
    if (hipMemcpy(&v, send, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    c->coll++;
    c->ctrl->red[c->rank] = v;
    if (!barrier(c)) return 6;
    double m = c->ctrl->red[0];
    for (int p = 1; p < c->world; ++p) m = c->ctrl->red[p] > m ? c->ctrl->red[p] : m;
    if (!barrier(c)) return 6;
    return hipMemcpy(recv, &m, sizeof(double), hipMemcpyHostToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}

}   // extern "C"
