/*
 * fake_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for the ten RCCL entry points csrc/rpt_comm.hip resolves with dlsym, so that
 * the product's N-rank gather (rpt_comm_init -> rpt_render_async -> rpt_gather_async -> rpt_read_gathered) runs UNCHANGED with
 * N PROCESSES on a box that has ONE GPU.  Real RCCL refuses two ranks on a device; every GPU box this build can reach has one.
 *
 * Selected only through RPT_RCCL_LIBRARY=<path of this .so> (csrc/rpt_comm.hip, RcclApi::load); bench.py refuses to run with
 * that variable set, and nothing in the product links or names this file (tests/test_contracts.py).
 *
 * Semantics kept from RCCL, because the code under test relies on them:
 *   - ncclSend / ncclRecv are STREAM-ORDERED: they return at once; the payload is read / written when the stream reaches them;
 *   - a send completes (in stream order) only once the matching receive has taken the payload of the PREVIOUS send on that
 *     channel, a receive only once the matching send has delivered — point-to-point rendezvous per (source, destination);
 *   - ncclCommCount reports the size the ranks agreed on at ncclCommInitRank (a barrier over all of them).
 * Transport: per (source, destination) one POSIX shared-memory segment registered with hipHostRegister; the sender's stream
 * runs  [host function: wait until the receiver consumed message k-1] -> hipMemcpyAsync D2H -> [host function: publish k],
 * the receiver's  [wait until k is published] -> hipMemcpyAsync H2D -> [mark k consumed].  Waits give up after
 * RPT_FAKE_RCCL_TIMEOUT_S (default 60) seconds and poison the communicator instead of hanging the box.
 */
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 64;
constexpr uint32_t MAGIC = 0x52505446u;   /* "RPTF" */

struct Channel {
    std::atomic<uint64_t> published;      /* messages the sender has delivered into the segment */
    std::atomic<uint64_t> consumed;       /* messages the receiver has copied out of it */
};
struct Control {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> world;
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> left;
    std::atomic<uint32_t> poisoned;
    Channel chan[MAX_RANKS][MAX_RANKS];   /* [source][destination] */
};

struct Segment {
    void *p = nullptr;
    size_t bytes = 0;
};

double timeout_seconds() {
    const char *e = getenv("RPT_FAKE_RCCL_TIMEOUT_S");
    return e ? atof(e) : 60.0;
}

/* spin (politely) until *word >= want; false on timeout or a poisoned communicator */
bool wait_for(std::atomic<uint64_t> *word, uint64_t want, Control *ctl) {
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = timeout_seconds();
    uint32_t spins = 0;
    while (word->load(std::memory_order_acquire) < want) {
        if (ctl->poisoned.load(std::memory_order_relaxed)) return false;
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            ctl->poisoned.store(1);
            fprintf(stderr, "fake_rccl: gave up waiting for a peer after %.0f s\n", limit);
            return false;
        }
    }
    return true;
}

}  // namespace

struct ncclComm {
    std::string name;                     /* shared-memory name stem from the unique id */
    Control *ctl = nullptr;
    int rank = 0, world = 1;
    std::map<std::pair<int, int>, Segment> seg;           /* (source, destination) -> current mapping */
    std::vector<Segment> retired;                         /* grown-out mappings stay valid until the communicator dies */
    uint64_t sent[MAX_RANKS] = {}, received[MAX_RANKS] = {};
};

namespace {

struct HostOp {
    Control *ctl;
    std::atomic<uint64_t> *word;
    uint64_t value;
    bool is_wait;                         /* wait until *word >= value, or store value into *word */
};

void host_op(void *arg) {
    HostOp *op = static_cast<HostOp *>(arg);
    if (op->is_wait) (void)wait_for(op->word, op->value, op->ctl);
    else op->word->store(op->value, std::memory_order_release);
    delete op;
}

std::string segment_name(const ncclComm *c, int src, int dst) { return c->name + "_" + std::to_string(src) + "_" + std::to_string(dst); }

/* both ends of a channel call this with the same byte count before they enqueue: the segment only ever grows */
ncclResult_t channel_buffer(ncclComm *c, int src, int dst, size_t bytes, void **out) {
    Segment &s = c->seg[{src, dst}];
    if (s.p && s.bytes >= bytes) { *out = s.p; return ncclSuccess; }
    if (s.p) c->retired.push_back(s);
    const size_t want = std::max<size_t>((bytes + 4095) & ~size_t(4095), 4096);
    int fd = shm_open(segment_name(c, src, dst).c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    flock(fd, LOCK_EX);
    struct stat st {};
    if (fstat(fd, &st) != 0 || ((size_t)st.st_size < want && ftruncate(fd, (off_t)want) != 0)) { flock(fd, LOCK_UN); close(fd); return ncclSystemError; }
    flock(fd, LOCK_UN);
    void *p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    /* pinned: the copies below must be true stream-ordered DMA — a pageable hipMemcpyAsync may touch the buffer at call time */
    if (hipHostRegister(p, want, hipHostRegisterDefault) != hipSuccess) { munmap(p, want); return ncclUnhandledCudaError; }
    s.p = p;
    s.bytes = want;
    *out = p;
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t enqueue_host(hipStream_t stream, Control *ctl, std::atomic<uint64_t> *word, uint64_t value, bool is_wait) {
    HostOp *op = new HostOp{ctl, word, value, is_wait};
    if (hipLaunchHostFunc(stream, host_op, op) != hipSuccess) { delete op; return ncclUnhandledCudaError; }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned char rnd[8] = {};
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { (void)!fread(rnd, 1, sizeof(rnd), f); fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "/rptfake_%d_%02x%02x%02x%02x%02x%02x%02x%02x", (int)getpid(), rnd[0], rnd[1], rnd[2], rnd[3], rnd[4],
             rnd[5], rnd[6], rnd[7]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (strncmp(id.internal, "/rptfake_", 9) != 0) return ncclInvalidArgument;
    auto *c = new ncclComm();
    c->name = id.internal;
    c->rank = rank;
    c->world = nranks;
    int fd = shm_open((c->name + "_ctl").c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    flock(fd, LOCK_EX);
    struct stat st {};
    if (fstat(fd, &st) != 0 || ((size_t)st.st_size < sizeof(Control) && ftruncate(fd, (off_t)sizeof(Control)) != 0)) { flock(fd, LOCK_UN); close(fd); delete c; return ncclSystemError; }
    flock(fd, LOCK_UN);
    void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = static_cast<Control *>(p);   /* a fresh segment is all zeros: every counter starts at 0 */
    c->ctl->magic.store(MAGIC);
    uint32_t expected = 0;
    if (!c->ctl->world.compare_exchange_strong(expected, (uint32_t)nranks) && expected != (uint32_t)nranks) { munmap(p, sizeof(Control)); delete c; return ncclInvalidArgument; }
    c->ctl->joined.fetch_add(1);
    /* like ncclCommInitRank: returns once every rank has arrived */
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->joined.load() < (uint32_t)nranks) {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds() * 2) {
            fprintf(stderr, "fake_rccl: only %u of %d ranks arrived\n", c->ctl->joined.load(), nranks);
            munmap(p, sizeof(Control));
            delete c;
            return ncclSystemError;
        }
    }
    *out = c;
    return ncclSuccess;
}

/* one process driving several ranks would need its host functions to wait on each other; the stand-in is one process per rank */
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    for (auto &kv : c->seg) {
        if (kv.second.p) { (void)hipHostUnregister(kv.second.p); munmap(kv.second.p, kv.second.bytes); }
        shm_unlink(segment_name(c, kv.first.first, kv.first.second).c_str());
    }
    for (Segment &s : c->retired) { (void)hipHostUnregister(s.p); munmap(s.p, s.bytes); }
    if (c->ctl) {
        if (c->ctl->left.fetch_add(1) + 1 == (uint32_t)c->world) shm_unlink((c->name + "_ctl").c_str());
        munmap(c->ctl, sizeof(Control));
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) {
    if (!c || !count) return ncclInvalidArgument;
    *count = (int)c->ctl->world.load();    /* what the ranks agreed on in shared memory, not this rank's own argument */
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    if (!c || !buf || peer < 0 || peer >= c->world || peer == c->rank || !type_bytes(type)) return ncclInvalidArgument;
    if (c->ctl->poisoned.load()) return ncclRemoteError;
    const size_t bytes = count * type_bytes(type);
    void *shm = nullptr;
    ncclResult_t r = channel_buffer(c, c->rank, peer, bytes, &shm);
    if (r != ncclSuccess) return r;
    Channel &ch = c->ctl->chan[c->rank][peer];
    const uint64_t k = ++c->sent[peer];
    if ((r = enqueue_host(stream, c->ctl, &ch.consumed, k - 1, true)) != ncclSuccess) return r;
    if (hipMemcpyAsync(shm, buf, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    return enqueue_host(stream, c->ctl, &ch.published, k, false);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    if (!c || !buf || peer < 0 || peer >= c->world || peer == c->rank || !type_bytes(type)) return ncclInvalidArgument;
    if (c->ctl->poisoned.load()) return ncclRemoteError;
    const size_t bytes = count * type_bytes(type);
    void *shm = nullptr;
    ncclResult_t r = channel_buffer(c, peer, c->rank, bytes, &shm);
    if (r != ncclSuccess) return r;
    Channel &ch = c->ctl->chan[peer][c->rank];
    const uint64_t k = ++c->received[peer];
    if ((r = enqueue_host(stream, c->ctl, &ch.published, k, true)) != ncclSuccess) return r;
    if (hipMemcpyAsync(buf, shm, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return enqueue_host(stream, c->ctl, &ch.consumed, k, false);
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: shared memory / rendezvous failed";
        case ncclInvalidArgument: return "fake_rccl: invalid argument";
        case ncclInvalidUsage: return "fake_rccl: one process per rank only";
        case ncclRemoteError: return "fake_rccl: a peer gave up (poisoned communicator)";
        default: return "fake_rccl: error";
    }
}

/* lets a test prove which library the product resolved */
int rpt_fake_rccl_marker(void) { return 0x46414b45; }

}  // extern "C"
