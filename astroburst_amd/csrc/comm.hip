// Collectives behind the C ABI: RCCL over xGMI, one communicator rank per context / GPU.
//
// The reference is a single-process desktop application (rayon threads, no collectives of any kind); what is sharded
// here is its per-pixel loop (core/stacking/combine.rs:160-182, rows are independent) and its frame list.  A Rust host
// has no torch.distributed, so the communicator lives in the library: rank 0 calls ab_comm_get_unique_id, the host
// carries the 128 bytes to the other ranks by whatever it has (Tauri IPC, a pipe, a file), every rank calls
// ab_comm_init_rank, and the sharded entry points (sharded.hip) enqueue their all-reduces on the context's own
// stream -- nothing is staged through the host.  A single-process host driving several GPUs uses ab_comm_init_all
// and one calling thread per context (or brackets the calls with ab_comm_group_start / _end).
//
// librccl is opened lazily with dlopen: libastroburst_hip.so itself loads (and every single-GPU entry point works)
// on a machine without RCCL, and inside a PyTorch process the already-loaded librccl.so.1 is reused.
//
// Second transport (round 3): HOST-STAGED collectives through a POSIX shared-memory segment (ab_comm_init_rank_host).
// RCCL refuses two ranks on one device, so this is what lets the sharded entry points run with N > 1 ranks on a one-GPU
// box (tests/test_gpu_multirank.py), and what a host falls back to when librccl is absent.  Same semantics -- the result
// is in `buf_dev` for whatever the stream runs next -- but the call blocks: chunk by chunk, device -> this rank's slot
// of the segment, barrier, every rank reduces all slots in rank order (so f64 sums are identical on every rank),
// barrier, host -> device.
//
// Failure handling (both transports): ab_comm_agree exchanges a status word BEFORE a sharded entry point's data
// collectives, so a rank that failed locally (out of memory, cancelled, bad shard) makes EVERY rank return an error and
// leaves the communicator usable; waits on a collective are bounded (AB_COMM_TIMEOUT_MS, default 300 000): a peer that
// died turns into AB_ERR_COMM and a dead communicator instead of a hang; ab_comm_abort releases the peers of a host
// communicator at once and tears an RCCL communicator down with ncclCommAbort.
#include "ab_common.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <mutex>

namespace {

// the part of rccl.h this file needs (ABI-stable since NCCL 2.0)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclMax = 2, ncclMin = 3 };
enum { ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat32 = 7, ncclFloat64 = 8, ncclUint8 = 1 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;                 // optional
    int (*CommGetAsyncError)(ncclComm_t, int *) = nullptr;  // optional
    std::string why;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        const char *e = dlerror();
        g_rccl.why = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
        return;
    }
#define AB_SYM(field, name)                                                              \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.handle, name);                   \
    if (!g_rccl.field) {                                                                 \
        g_rccl.why = std::string("librccl lacks ") + name;                               \
        return;                                                                          \
    }
    AB_SYM(GetUniqueId, "ncclGetUniqueId")
    AB_SYM(CommInitRank, "ncclCommInitRank")
    AB_SYM(CommInitAll, "ncclCommInitAll")
    AB_SYM(CommDestroy, "ncclCommDestroy")
    AB_SYM(AllReduce, "ncclAllReduce")
    AB_SYM(AllGather, "ncclAllGather")
    AB_SYM(Broadcast, "ncclBroadcast")
    AB_SYM(GroupStart, "ncclGroupStart")
    AB_SYM(GroupEnd, "ncclGroupEnd")
    AB_SYM(GetErrorString, "ncclGetErrorString")
#undef AB_SYM
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(g_rccl.handle, "ncclCommAbort");
    g_rccl.CommGetAsyncError = (decltype(g_rccl.CommGetAsyncError))dlsym(g_rccl.handle, "ncclCommGetAsyncError");
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.why.empty() ? &g_rccl : nullptr;
}

int to_nccl_type(int dt, size_t *elem) {
    switch (dt) {
        case AB_DT_I32: *elem = 4; return ncclInt32;
        case AB_DT_U32: *elem = 4; return ncclUint32;
        case AB_DT_I64: *elem = 8; return ncclInt64;
        case AB_DT_U64: *elem = 8; return ncclUint64;
        case AB_DT_F32: *elem = 4; return ncclFloat32;
        case AB_DT_F64: *elem = 8; return ncclFloat64;
        default: return -1;
    }
}

// ---- host-staged transport: one shared-memory segment per communicator -------------------------------------------------
struct HostShm {  // lives at the start of the segment; lock-free atomics on plain words work across processes
    std::atomic<uint32_t> magic;    // set last by rank 0
    int32_t nranks;
    uint64_t slot_bytes;
    std::atomic<uint32_t> joined;   // ranks attached so far
    std::atomic<uint32_t> aborted;  // ab_comm_abort, or a rank that gave up waiting: every wait returns AB_ERR_COMM from now on
    std::atomic<uint32_t> arrived;  // barrier: arrivals of the current generation
    std::atomic<uint32_t> generation;
    int32_t status[64];             // ab_comm_agree: one word per rank
    int32_t owner_pid;              // rank 0's process: a segment whose owner is gone is a dead job's, not this one's
    uint64_t owner_start;           // that process's start time (/proc/<pid>/stat field 22, clock ticks since boot; 0 = unknown):
                                    // a recycled pid has another start time, so a dead job's segment cannot pass for a live one
    uint64_t owner_pidns;           // inode of rank 0's /proc/self/ns/pid: the pid test only means something inside one namespace
};
constexpr uint32_t kShmMagic = 0x41424d43u;  // "ABMC"
constexpr size_t kShmHeader = 4096;

int64_t now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}

// start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat; the command name in field 2 may hold
// blanks and parentheses, so fields are counted from the LAST ')'); 0 when it cannot be read
uint64_t proc_start_time(pid_t pid) {
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%d/stat", (int)pid);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return 0;
    const ssize_t n = read(fd, buf, sizeof buf - 1);
    close(fd);
    if (n <= 0) return 0;
    buf[n] = 0;
    const char *p = strrchr(buf, ')');
    if (!p) return 0;
    ++p;
    for (int field = 3; field <= 22; ++field) {  // p sits on the blank before field `field`
        while (*p == ' ') ++p;
        if (field == 22) return strtoull(p, nullptr, 10);
        while (*p && *p != ' ') ++p;
        if (!*p) return 0;
    }
    return 0;
}

uint64_t own_pidns() {
    struct stat st;
    return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : 0;
}

// Is the segment's owner (its rank 0) gone?  Only answerable inside the owner's pid namespace: ranks in separate containers
// that share /dev/shm see each other's pids as nonexistent, so across namespaces the answer is "cannot tell" = alive, and a dead
// job's segment is then caught by the join time-out instead (AB_COMM_HOST_PIDCHECK=0 forces that behaviour everywhere).
bool owner_is_dead(const HostShm *h) {
    static const bool check = [] { const char *e = ab_dev_env("AB_COMM_HOST_PIDCHECK"); return !(e && *e == '0'); }();
    if (!check) return false;
    if (h->owner_pid <= 0) return true;
    const uint64_t ns = own_pidns();
    if (h->owner_pidns && ns && h->owner_pidns != ns) return false;
    if (kill((pid_t)h->owner_pid, 0) != 0 && errno == ESRCH) return true;
    const uint64_t st = proc_start_time((pid_t)h->owner_pid);
    return h->owner_start && st && st != h->owner_start;  // the pid lives on in another process
}

int64_t default_timeout_ms() {
    const char *e = ab_env("AB_COMM_TIMEOUT_MS");
    const long long v = e ? atoll(e) : 300000;
    return v > 0 ? (int64_t)v : 300000;
}

}  // namespace

struct ab_comm {
    ncclComm_t comm = nullptr;  // RCCL transport
    int rank = 0, size = 1, device = 0;
    uint64_t collectives = 0;   // issued so far (bench / tests report it)
    bool dead = false;          // aborted or timed out: every further collective fails at once
    int64_t timeout_ms = 300000;
    // host transport
    bool host = false;
    HostShm *shm = nullptr;
    size_t shm_bytes = 0;
    std::string shm_name;
    bool shm_registered = false;  // hipHostRegister'ed: the copies to and from the slots are true async DMA
    void *tmp = nullptr;          // pinned, slot_bytes: the reduced chunk on its way back to the device
    char *slot(int r) const { return (char *)shm + kShmHeader + (size_t)r * shm->slot_bytes; }
};

namespace {

std::atomic<int> g_rccl_comms{0};  // live RCCL communicators: ab_comm_group_start / _end are no-ops without one

// sense-reversing barrier over the segment; bounded, and released by `aborted`
int host_barrier(ab_ctx *ctx, ab_comm *c, const char *what) {
    HostShm *h = c->shm;
    if (c->dead || h->aborted.load(std::memory_order_acquire)) {
        c->dead = true;
        return ab_set_error(ctx, AB_ERR_COMM, "%s: the communicator was aborted", what);
    }
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->size) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
        return AB_OK;
    }
    const int64_t t0 = now_ms();
    for (unsigned spin = 0;; ++spin) {
        if (h->generation.load(std::memory_order_acquire) != gen) return AB_OK;
        if (h->aborted.load(std::memory_order_acquire)) {
            c->dead = true;
            return ab_set_error(ctx, AB_ERR_COMM, "%s: the communicator was aborted by a peer", what);
        }
        if (spin < 2000) {
            sched_yield();
        } else {
            timespec ts = {0, 50000};
            nanosleep(&ts, nullptr);
            if ((spin & 255) == 0 && now_ms() - t0 > c->timeout_ms) {
                h->aborted.store(1, std::memory_order_release);  // whoever arrives late must not wait for us
                c->dead = true;
                return ab_set_error(ctx, AB_ERR_COMM, "%s: rank %d waited %lld ms for its peers (a rank died or never joined)", what, c->rank,
                                    (long long)(now_ms() - t0));
            }
        }
    }
}

template <typename T>
void reduce_into(T *acc, const T *src, size_t n, int op) {
    if (op == AB_RED_SUM)
        for (size_t i = 0; i < n; ++i) acc[i] += src[i];
    else if (op == AB_RED_MAX)
        for (size_t i = 0; i < n; ++i) acc[i] = src[i] > acc[i] ? src[i] : acc[i];
    else
        for (size_t i = 0; i < n; ++i) acc[i] = src[i] < acc[i] ? src[i] : acc[i];
}

void reduce_slots(ab_comm *c, void *out, size_t bytes, int dtype, int op) {
    memcpy(out, c->slot(0), bytes);
    for (int r = 1; r < c->size; ++r) {  // rank order on every rank: identical f64 sums everywhere
        const void *src = c->slot(r);
        switch (dtype) {
            case AB_DT_I32: reduce_into((int32_t *)out, (const int32_t *)src, bytes / 4, op); break;
            case AB_DT_U32: reduce_into((uint32_t *)out, (const uint32_t *)src, bytes / 4, op); break;
            case AB_DT_I64: reduce_into((int64_t *)out, (const int64_t *)src, bytes / 8, op); break;
            case AB_DT_U64: reduce_into((uint64_t *)out, (const uint64_t *)src, bytes / 8, op); break;
            case AB_DT_F32: reduce_into((float *)out, (const float *)src, bytes / 4, op); break;
            default: reduce_into((double *)out, (const double *)src, bytes / 8, op); break;
        }
    }
}

// a HIP call inside a host collective that fails on THIS rank: the peers are (or will be) waiting in the collective's barrier for a
// rank that has left -- raise the segment's abort flag first, so that they return AB_ERR_COMM at once instead of after the time-out
#define AB_HIP_COMM(ctx, c, call)                                                                       \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (c)->dead = true;                                                                         \
            if ((c)->shm) (c)->shm->aborted.store(1, std::memory_order_release);                      \
            return ab_set_error((ctx), AB_ERR_HIP, "%s failed inside a host-staged collective: %s (%s:%d); communicator aborted", #call, \
                                hipGetErrorString(e_), __FILE__, __LINE__);                           \
        }                                                                                             \
    } while (0)

int host_allreduce(ab_ctx *ctx, ab_comm *c, void *buf_dev, size_t count, size_t elem, int dtype, int op) {
    const size_t slot = (size_t)c->shm->slot_bytes, total = count * elem;
    for (size_t off = 0; off < total; off += slot) {
        const size_t n = std::min(slot, total - off);
        AB_HIP_COMM(ctx, c, hipMemcpyAsync(c->slot(c->rank), (char *)buf_dev + off, n, hipMemcpyDeviceToHost, ctx->stream));
        AB_HIP_COMM(ctx, c, hipStreamSynchronize(ctx->stream));  // (also: the previous chunk has left c->tmp)
        AB_TRY(host_barrier(ctx, c, "all-reduce"));
        reduce_slots(c, c->tmp, n, dtype, op);
        AB_TRY(host_barrier(ctx, c, "all-reduce"));  // every rank has read every slot: they may be overwritten
        AB_HIP_COMM(ctx, c, hipMemcpyAsync((char *)buf_dev + off, c->tmp, n, hipMemcpyHostToDevice, ctx->stream));
    }
    AB_HIP_COMM(ctx, c, hipStreamSynchronize(ctx->stream));  // c->tmp is free again when the call returns
    return AB_OK;
}

int host_broadcast(ab_ctx *ctx, ab_comm *c, void *buf_dev, size_t bytes, int root) {
    const size_t slot = (size_t)c->shm->slot_bytes;
    for (size_t off = 0; off < bytes; off += slot) {
        const size_t n = std::min(slot, bytes - off);
        if (c->rank == root) {
            AB_HIP_COMM(ctx, c, hipMemcpyAsync(c->slot(root), (char *)buf_dev + off, n, hipMemcpyDeviceToHost, ctx->stream));
            AB_HIP_COMM(ctx, c, hipStreamSynchronize(ctx->stream));
        }
        AB_TRY(host_barrier(ctx, c, "broadcast"));
        if (c->rank != root) {
            AB_HIP_COMM(ctx, c, hipMemcpyAsync((char *)buf_dev + off, c->slot(root), n, hipMemcpyHostToDevice, ctx->stream));
            AB_HIP_COMM(ctx, c, hipStreamSynchronize(ctx->stream));
        }
        AB_TRY(host_barrier(ctx, c, "broadcast"));
    }
    return AB_OK;
}

int host_allgather(ab_ctx *ctx, ab_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank) {
    const size_t slot = (size_t)c->shm->slot_bytes;
    for (size_t off = 0; off < bytes_per_rank; off += slot) {
        const size_t n = std::min(slot, bytes_per_rank - off);
        AB_HIP_COMM(ctx, c, hipMemcpyAsync(c->slot(c->rank), (const char *)send_dev + off, n, hipMemcpyDeviceToHost, ctx->stream));
        AB_HIP_COMM(ctx, c, hipStreamSynchronize(ctx->stream));
        AB_TRY(host_barrier(ctx, c, "all-gather"));
        for (int r = 0; r < c->size; ++r)
            AB_HIP_COMM(ctx, c, hipMemcpyAsync((char *)recv_dev + (size_t)r * bytes_per_rank + off, c->slot(r), n, hipMemcpyHostToDevice, ctx->stream));
        AB_HIP_COMM(ctx, c, hipStreamSynchronize(ctx->stream));
        AB_TRY(host_barrier(ctx, c, "all-gather"));
    }
    return AB_OK;
}

void host_detach(ab_comm *c) {
    if (c->tmp) (void)hipHostFree(c->tmp);
    if (c->shm) {
        if (c->shm_registered) (void)hipHostUnregister(c->shm);
        munmap(c->shm, c->shm_bytes);
    }
    if (c->rank == 0 && !c->shm_name.empty()) shm_unlink(c->shm_name.c_str());  // (already gone after a complete join)
    c->tmp = nullptr;
    c->shm = nullptr;
}

}  // namespace

// Wait for the context's stream with a communicator in play: bounded, and watching RCCL's asynchronous error state -- a peer
// that died inside a collective leaves the kernel spinning on the device for ever.
int ab_comm_stream_wait(ab_ctx *ctx, ab_comm *c) {
    if (!c || c->host) {
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return AB_OK;
    }
    if (c->dead) return ab_set_error(ctx, AB_ERR_COMM, "the communicator was aborted");
    const Rccl *r = rccl();
    const int64_t t0 = now_ms();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) return AB_OK;
        if (q != hipErrorNotReady) return ab_set_error(ctx, AB_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        if (spin < 20000) {
            sched_yield();
            continue;
        }
        timespec ts = {0, 100000};
        nanosleep(&ts, nullptr);
        if ((spin & 127) != 0) continue;
        int async_err = ncclSuccess;
        const bool bad = r && r->CommGetAsyncError && c->comm && r->CommGetAsyncError(c->comm, &async_err) == ncclSuccess && async_err != ncclSuccess;
        if (bad || now_ms() - t0 > c->timeout_ms) {
            c->dead = true;
            if (r && r->CommAbort && c->comm) {
                (void)r->CommAbort(c->comm);  // frees the communicator and stops its kernels
                c->comm = nullptr;
                g_rccl_comms.fetch_sub(1);
            }
            return ab_set_error(ctx, AB_ERR_COMM, bad ? "RCCL reported an asynchronous error (%s): communicator aborted" : "a collective did not finish within %s ms: communicator aborted",
                                bad ? r->GetErrorString(async_err) : std::to_string((long long)c->timeout_ms).c_str());
        }
    }
}

#define AB_NCCL(ctx, r, call)                                                                                            \
    do {                                                                                                                 \
        int e_ = (call);                                                                                                 \
        if (e_ != ncclSuccess) return ab_set_error((ctx), AB_ERR_COMM, "%s failed: %s", #call, (r)->GetErrorString(e_)); \
    } while (0)
#define AB_COMM_ALIVE(ctx, c)                                                                         \
    do {                                                                                              \
        if ((c)->dead) return ab_set_error((ctx), AB_ERR_COMM, "the communicator was aborted");    \
    } while (0)

extern "C" {

int ab_comm_get_unique_id(uint8_t id[AB_COMM_ID_BYTES]) try {
    if (!id) return AB_ERR_INVALID;
    const Rccl *r = rccl();
    if (!r) return AB_ERR_COMM;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return AB_ERR_COMM;
    static_assert(sizeof u == AB_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, sizeof u);
    return AB_OK;
} AB_CATCH_NOCTX

int ab_comm_init_rank(ab_ctx *ctx, const uint8_t id[AB_COMM_ID_BYTES], int nranks, int rank, ab_comm **out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, id && out, "null id or output");
    AB_CHECK(ctx, nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    *out = nullptr;
    const Rccl *r = rccl();
    if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
    AB_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ab_comm *c = new (std::nothrow) ab_comm();
    if (!c) return ab_set_error(ctx, AB_ERR_NOMEM, "out of host memory");
    const int e = r->CommInitRank(&c->comm, nranks, u, rank);
    if (e != ncclSuccess) {
        delete c;
        return ab_set_error(ctx, AB_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, r->GetErrorString(e));
    }
    c->rank = rank;
    c->size = nranks;
    c->device = ctx->device;
    c->timeout_ms = default_timeout_ms();
    g_rccl_comms.fetch_add(1);
    *out = c;
    return AB_OK;
} AB_CATCH(ctx)

// Host-staged communicator: `name` identifies the job (every rank passes the same string; rank 0 creates the segment
// /abcomm_<name>, the others wait for it).  Ranks may share a device.  Blocks until all nranks have joined (bounded).
int ab_comm_init_rank_host(ab_ctx *ctx, const char *name, int nranks, int rank, ab_comm **out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, name && *name && out, "null name or output");
    AB_CHECK(ctx, nranks >= 1 && nranks <= 64 && rank >= 0 && rank < nranks, "rank %d of %d (a host communicator has 1 .. 64 ranks)", rank, nranks);
    AB_CHECK(ctx, strlen(name) < 200 && !strchr(name, '/'), "communicator name must be a short string without '/'");
    *out = nullptr;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const char *se = ab_env("AB_COMM_HOST_SLOT_MB");
    const long long slot_mb = se ? atoll(se) : 4;
    const size_t slot = (size_t)(slot_mb >= 1 && slot_mb <= 1024 ? slot_mb : 4) << 20;
    const size_t bytes = kShmHeader + (size_t)nranks * slot;
    static_assert(sizeof(HostShm) <= kShmHeader, "header outgrew its page");
    ab_comm *c = new (std::nothrow) ab_comm();
    if (!c) return ab_set_error(ctx, AB_ERR_NOMEM, "out of host memory");
    c->host = true;
    c->rank = rank;
    c->size = nranks;
    c->device = ctx->device;
    c->timeout_ms = default_timeout_ms();
    c->shm_name = std::string("/abcomm_") + name;
    c->shm_bytes = bytes;
    const int64_t t0 = now_ms();  // ONE clock for the whole call, whatever is found and dropped on the way
    void *m = nullptr;
    HostShm *h = nullptr;
    for (;;) {  // one pass = open, map, validate; a pass that finds a dead job's segment drops it and looks again
        int fd = -1;
        if (rank == 0) {
            shm_unlink(c->shm_name.c_str());  // a stale segment of a crashed job
            fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
                if (fd >= 0) close(fd);
                const std::string nm = c->shm_name;
                c->shm_name.clear();
                delete c;
                return ab_set_error(ctx, AB_ERR_COMM, "cannot create the shared segment %s (%zu bytes): %s", nm.c_str(), bytes, strerror(errno));
            }
        } else {
            for (;;) {  // the segment appears when rank 0 gets here; its size is final once ftruncate has run
                fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
                if (fd >= 0) close(fd);
                fd = -1;
                if (now_ms() - t0 > c->timeout_ms) {
                    const std::string nm = c->shm_name;
                    delete c;
                    return ab_set_error(ctx, AB_ERR_COMM, "rank %d: no segment %s after %lld ms (rank 0 never started?)", rank, nm.c_str(), (long long)(now_ms() - t0));
                }
                timespec ts = {0, 2000000};
                nanosleep(&ts, nullptr);
            }
        }
        m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) {
            if (rank == 0) shm_unlink(c->shm_name.c_str());
            c->shm_name.clear();
            delete c;
            return ab_set_error(ctx, AB_ERR_COMM, "mmap of the shared segment failed: %s", strerror(errno));
        }
        c->shm = (HostShm *)m;
        h = c->shm;
        if (rank == 0) {  // a fresh segment is zero-filled
            h->nranks = nranks;
            h->slot_bytes = slot;
            h->owner_pid = (int32_t)getpid();
            h->owner_start = proc_start_time(getpid());
            h->owner_pidns = own_pidns();
            h->magic.store(kShmMagic, std::memory_order_release);
            break;
        }
        bool initialised = true;
        while (h->magic.load(std::memory_order_acquire) != kShmMagic) {
            if (now_ms() - t0 > c->timeout_ms) {
                initialised = false;
                break;
            }
            sched_yield();
        }
        if (!initialised) {
            host_detach(c);
            delete c;
            return ab_set_error(ctx, AB_ERR_COMM, "rank %d: the shared segment was never initialised", rank);
        }
        // A segment of this name whose rank 0 no longer exists was left by a job that died (before all its ranks had joined, or it
        // would have been unlinked).  This job's rank 0 has not got to replacing it yet: drop the mapping and look again, on the
        // same clock.  (Joining it -- its `joined` count may even make the communicator look complete -- would put this rank in a
        // different segment from its own rank 0 for good.)
        if (!owner_is_dead(h)) break;
        munmap(m, bytes);
        c->shm = nullptr;
        m = nullptr;
        if (now_ms() - t0 > c->timeout_ms) {
            const std::string nm = c->shm_name;
            delete c;
            return ab_set_error(ctx, AB_ERR_COMM, "rank %d: only a dead job's segment %s to join after %lld ms (this job's rank 0 never started?)",
                                rank, nm.c_str(), (long long)(now_ms() - t0));
        }
        timespec ts = {0, 5000000};
        nanosleep(&ts, nullptr);
    }
    if (rank != 0 && (h->nranks != nranks || h->slot_bytes != slot)) {
        const int hn = h->nranks;
        host_detach(c);
        delete c;
        return ab_set_error(ctx, AB_ERR_COMM, "rank %d joined a communicator of %d ranks as one of %d (or with another AB_COMM_HOST_SLOT_MB)", rank, hn, nranks);
    }
    c->shm_registered = hipHostRegister(m, bytes, hipHostRegisterDefault) == hipSuccess;
    if (!c->shm_registered) (void)hipGetLastError();  // pageable copies still work (staged by the runtime)
    if (hipHostMalloc(&c->tmp, slot, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->tmp = nullptr;
        host_detach(c);
        delete c;
        return ab_set_error(ctx, AB_ERR_NOMEM, "cannot allocate %zu bytes of pinned memory for the host communicator", slot);
    }
    h->joined.fetch_add(1, std::memory_order_acq_rel);
    while (h->joined.load(std::memory_order_acquire) < (uint32_t)nranks) {
        if (h->aborted.load(std::memory_order_acquire) || now_ms() - t0 > c->timeout_ms) {
            h->aborted.store(1, std::memory_order_release);
            const uint32_t got = h->joined.load();
            host_detach(c);
            delete c;
            return ab_set_error(ctx, AB_ERR_COMM, "rank %d: only %u of %d ranks joined within %lld ms", rank, got, nranks, (long long)(now_ms() - t0));
        }
        timespec ts = {0, 200000};
        nanosleep(&ts, nullptr);
    }
    if (rank == 0) {  // everybody holds a mapping: the name can go (nothing is left behind whatever happens later)
        shm_unlink(c->shm_name.c_str());
        c->shm_name.clear();
    }
    *out = c;
    return AB_OK;
} AB_CATCH(ctx)

int ab_comm_init_all(ab_ctx *const *ctxs, int n, ab_comm **out_comms) try {
    if (!ctxs || !out_comms || n < 1 || n > 64) return AB_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return AB_ERR_INVALID;
        out_comms[i] = nullptr;
    }
    ab_ctx *c0 = ctxs[0];
    const Rccl *r = rccl();
    if (!r) return ab_set_error(c0, AB_ERR_COMM, "%s", g_rccl.why.c_str());
    int devs[64];
    ncclComm_t comms[64];
    for (int i = 0; i < n; ++i) {
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return ab_set_error(c0, AB_ERR_INVALID, "contexts %d and %d share device %d", j, i, devs[i]);
    }
    AB_NCCL(c0, r, r->CommInitAll(comms, n, devs));
    for (int i = 0; i < n; ++i) {
        ab_comm *c = new (std::nothrow) ab_comm();
        if (!c) return ab_set_error(c0, AB_ERR_NOMEM, "out of host memory");
        c->comm = comms[i];
        c->rank = i;
        c->size = n;
        c->device = devs[i];
        c->timeout_ms = default_timeout_ms();
        g_rccl_comms.fetch_add(1);
        out_comms[i] = c;
    }
    return AB_OK;
} AB_CATCH_NOCTX

void ab_comm_destroy(ab_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->host) {
        host_detach(c);
    } else {
        const Rccl *r = rccl();
        if (r && c->comm) {
            (void)r->CommDestroy(c->comm);
            g_rccl_comms.fetch_sub(1);
        }
    }
    delete c;
}

// Give up on the communicator NOW: peers blocked in a host-staged collective return AB_ERR_COMM at once; an RCCL
// communicator is torn down with ncclCommAbort (its pending kernels stop; peers see it through their own bounded waits).
// The handle stays valid for ab_comm_destroy; every further collective on it fails with AB_ERR_COMM.
int ab_comm_abort(ab_comm *c) try {
    if (!c) return AB_OK;
    c->dead = true;
    if (c->host) {
        if (c->shm) c->shm->aborted.store(1, std::memory_order_release);
        return AB_OK;
    }
    const Rccl *r = rccl();
    if (r && c->comm) {
        (void)hipSetDevice(c->device);
        if (r->CommAbort)
            (void)r->CommAbort(c->comm);
        else
            (void)r->CommDestroy(c->comm);
        c->comm = nullptr;
        g_rccl_comms.fetch_sub(1);
    }
    return AB_OK;
} AB_CATCH_NOCTX

int ab_comm_set_timeout_ms(ab_comm *c, int64_t ms) try {
    if (!c || ms <= 0) return AB_ERR_INVALID;
    c->timeout_ms = ms;
    return AB_OK;
} AB_CATCH_NOCTX

int ab_comm_rank(const ab_comm *c) { return c ? c->rank : 0; }
int ab_comm_size(const ab_comm *c) { return c ? c->size : 1; }
int ab_comm_is_host(const ab_comm *c) try { return c && c->host ? 1 : 0; } AB_CATCH_NOCTX
uint64_t ab_comm_collectives_issued(const ab_comm *c) { return c ? c->collectives : 0; }

// (no-ops unless an RCCL communicator exists: host-staged collectives run in call order on every rank)
int ab_comm_group_start(void) try {
    if (g_rccl_comms.load() == 0) return AB_OK;
    const Rccl *r = rccl();
    return (r && r->GroupStart() == ncclSuccess) ? AB_OK : AB_ERR_COMM;
} AB_CATCH_NOCTX
int ab_comm_group_end(void) try {
    if (g_rccl_comms.load() == 0) return AB_OK;
    const Rccl *r = rccl();
    return (r && r->GroupEnd() == ncclSuccess) ? AB_OK : AB_ERR_COMM;
} AB_CATCH_NOCTX

// Every rank passes the status of what it did locally; all ranks return AB_OK only if all passed AB_OK.  A rank that
// failed gets its own status back; the others get AB_ERR_CANCELLED if a peer was cancelled, else AB_ERR_COMM naming the
// peer.  Call it BEFORE the data collectives of a sharded operation: nobody enters a collective a peer will never join,
// and the communicator stays usable.  Synchronises the stream (RCCL transport) -- one small all-reduce.
int ab_comm_agree(ab_ctx *ctx, ab_comm *c, int local_status) try {
    if (!ctx) return AB_ERR_INVALID;
    if (!c || c->size == 1) return local_status;
    AB_COMM_ALIVE(ctx, c);
    int worst = 0, who = -1;
    if (c->host) {
        c->shm->status[c->rank] = local_status;
        AB_TRY(host_barrier(ctx, c, "status agreement"));
        for (int r = 0; r < c->size; ++r)
            if (c->shm->status[r] > worst) worst = c->shm->status[r], who = r;
        AB_TRY(host_barrier(ctx, c, "status agreement"));  // the words may be rewritten
    } else {
        AB_CHECK(ctx, c->device == ctx->device, "communicator is bound to device %d, context to %d", c->device, ctx->device);
        const Rccl *r = rccl();
        if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
        AB_HIP(ctx, hipSetDevice(ctx->device));
        // MAX of (status << 8 | rank + 1): the highest status and one rank that reported it
        void *pin = nullptr;
        AB_TRY(ab_pinned(ctx, 16, &pin));
        int32_t *word = (int32_t *)pin;
        *word = local_status > 0 ? ((local_status << 8) | (c->rank + 1)) : 0;
        int32_t *dev = (int32_t *)(ctx->counters + AB_REJ_SLOTS + 1);  // a spare word behind the rejection counters
        AB_HIP(ctx, hipMemcpyAsync(dev, word, sizeof *word, hipMemcpyHostToDevice, ctx->stream));
        AB_NCCL(ctx, r, r->AllReduce(dev, dev, 1, ncclInt32, ncclMax, c->comm, ctx->stream));
        AB_HIP(ctx, hipMemcpyAsync(word, dev, sizeof *word, hipMemcpyDeviceToHost, ctx->stream));
        AB_TRY(ab_comm_stream_wait(ctx, c));
        worst = *word >> 8;
        who = (*word & 0xff) - 1;
    }
    c->collectives++;
    if (local_status != AB_OK) return local_status;  // (its message is already in ab_last_error)
    if (worst == AB_OK) return AB_OK;
    if (worst == AB_ERR_CANCELLED) return ab_set_error(ctx, AB_ERR_CANCELLED, "Operation cancelled (on rank %d)", who);
    return ab_set_error(ctx, AB_ERR_COMM, "rank %d failed (ab_status %d) before the collective: nothing was exchanged", who, worst);
} AB_CATCH(ctx)

// in place, on the context's stream; a NULL communicator is a world of one (no-op).  RCCL: asynchronous.  Host-staged: the
// call returns when the result is in buf_dev.
int ab_comm_allreduce(ab_ctx *ctx, ab_comm *c, void *buf_dev, size_t count, int dtype, int op) try {
    if (!ctx) return AB_ERR_INVALID;
    if (!c) return AB_OK;
    AB_COMM_ALIVE(ctx, c);
    AB_CHECK(ctx, buf_dev && count > 0, "null or empty all-reduce buffer");
    AB_CHECK(ctx, c->device == ctx->device, "communicator is bound to device %d, context to %d", c->device, ctx->device);
    size_t elem = 0;
    const int nt = to_nccl_type(dtype, &elem);
    AB_CHECK(ctx, nt >= 0, "bad ab_dtype %d", dtype);
    const int nop = op == AB_RED_SUM ? ncclSum : (op == AB_RED_MAX ? ncclMax : (op == AB_RED_MIN ? ncclMin : -1));
    AB_CHECK(ctx, nop >= 0, "bad ab_redop %d", op);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (c->host) {
        AB_TRY(host_allreduce(ctx, c, buf_dev, count, elem, dtype, op));
    } else {
        const Rccl *r = rccl();
        if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
        AB_NCCL(ctx, r, r->AllReduce(buf_dev, buf_dev, count, nt, nop, c->comm, ctx->stream));
    }
    c->collectives++;
    return AB_OK;
} AB_CATCH(ctx)

// recv_dev holds size x bytes_per_rank bytes, rank r's block at r * bytes_per_rank; send_dev may be its own block
int ab_comm_allgather(ab_ctx *ctx, ab_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, send_dev && recv_dev && bytes_per_rank > 0, "null or empty all-gather buffer");
    if (!c) {
        if (send_dev != recv_dev) AB_HIP(ctx, hipMemcpyAsync(recv_dev, send_dev, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return AB_OK;
    }
    AB_COMM_ALIVE(ctx, c);
    AB_CHECK(ctx, c->device == ctx->device, "communicator is bound to device %d, context to %d", c->device, ctx->device);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (c->host) {
        AB_TRY(host_allgather(ctx, c, send_dev, recv_dev, bytes_per_rank));
    } else {
        const Rccl *r = rccl();
        if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
        AB_NCCL(ctx, r, r->AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, ctx->stream));
    }
    c->collectives++;
    return AB_OK;
} AB_CATCH(ctx)

int ab_comm_broadcast(ab_ctx *ctx, ab_comm *c, void *buf_dev, size_t bytes, int root) try {
    if (!ctx) return AB_ERR_INVALID;
    if (!c) return AB_OK;
    AB_COMM_ALIVE(ctx, c);
    AB_CHECK(ctx, buf_dev && bytes > 0 && root >= 0 && root < c->size, "bad broadcast arguments");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (c->host) {
        AB_TRY(host_broadcast(ctx, c, buf_dev, bytes, root));
    } else {
        const Rccl *r = rccl();
        if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
        AB_NCCL(ctx, r, r->Broadcast(buf_dev, buf_dev, bytes, ncclUint8, root, c->comm, ctx->stream));
    }
    c->collectives++;
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
